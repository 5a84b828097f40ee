# round 6, session 1: the column-walking filter gradient -- parity, per-layer micro-benchmark, whole-step A/B
cd $GRAFT_REPO_ROOT; O=gpurun_out/r6s1; mkdir -p $O
python -m pytest tests/test_hip_round6.py tests/test_hip_round3.py -x -q -k "wgrad" 2>&1 | tail -4
python tools/wgrad_microbench.py 8 2>&1 | grep wgrad | tee $O/micro_b8.txt
python tools/wgrad_microbench.py 24 100 2>&1 | grep wgrad | tee $O/micro_b24.txt
bash tools/ab_env.sh REPMODE_WGRAD_COL 0 1 2>&1 | tee $O/step_ab.txt
