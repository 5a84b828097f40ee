# round 6, session 2: BatchNorm passes as one launch (grid-wide barrier) -- parity, whole-step A/B, trace of the BN family
cd $GRAFT_REPO_ROOT; O=gpurun_out/r6s2; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_round6.py -x -q 2>&1 | tail -4
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -k "bn_relu or net_golden or block_golden or train" 2>&1 | tail -3
bash tools/ab_env.sh REPMODE_BN_FUSED 0 1 2>&1 | tee $O/step_ab.txt
