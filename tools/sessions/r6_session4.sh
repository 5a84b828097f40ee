# round 6, session 4: one-launch BatchNorm passes with the two-level barrier -- parity, per-level micro-benchmark, step A/B
cd $GRAFT_REPO_ROOT; O=gpurun_out/r6s4; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_round6.py -x -q 2>&1 | tail -4
for f in 0 1; do echo "REPMODE_BN_FUSED=$f"; REPMODE_BN_FUSED=$f python tools/bn_microbench.py cold 2>&1 | grep "^bn"; done | tee $O/bn_micro.txt
bash tools/ab_env.sh REPMODE_BN_FUSED 0 1 2>&1 | tee $O/step_ab.txt
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
