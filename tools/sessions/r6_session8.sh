# round 6, session 8: the column form's tap split on level 1 -- parity, per-layer micro-benchmark (q = 1 / 2 / 4), step A/B
cd $GRAFT_REPO_ROOT; O=gpurun_out/r6s8; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_round6.py -x -q -k "wgrad" 2>&1 | tail -4
for q in 1 2 4; do echo "REPMODE_WGRAD_COL_Q=$q"; REPMODE_WGRAD_COL_Q=$q python tools/wgrad_microbench.py 8 200 2>&1 | grep wgrad | cut -c1-140; done | tee $O/micro_b8.txt
echo "batch 24"; for q in 1 2; do echo "REPMODE_WGRAD_COL_Q=$q"; REPMODE_WGRAD_COL_Q=$q python tools/wgrad_microbench.py 24 60 2>&1 | grep wgrad | cut -c1-140; done | tee $O/micro_b24.txt
