# round 6, session 6: the column walk's dual form on the per-expert levels -- parity, per-layer micro-benchmark, step A/B
cd $GRAFT_REPO_ROOT; O=gpurun_out/r6s6; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_round6.py -x -q -k "dual or one_launch" 2>&1 | tail -5
python tools/wgrad_deep_microbench.py 8 2>&1 | grep wgrad | tee $O/micro_b8.txt
python tools/wgrad_deep_microbench.py 24 100 2>&1 | grep wgrad | tee $O/micro_b24.txt
