# Round-4 GPU session 5: stream-K filter gradient incl. the 16-voxel tile (level 2): parity, per-launch tables, train step A/B;
# the gradient collectives beside backward with the persistent grids leaving CUs free (REPMODE_RESERVE_CUS)
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r4s5; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_round3.py tests/test_hip_parity.py -m gpu -q --maxfail=10 --tb=short -k "wgrad or mode_conv3d_op or full_size or train_step" > $O/pytest_new_full.log 2>&1; tail -6 $O/pytest_new_full.log
for mode in 1 3; do
  REPMODE_WGRAD_WS=$mode timeout 300 python bench.py --no-cpu-baseline --no-fwd --prof-all --dump-launches $O/launches_ws$mode.json --steps 12 --warmup 6 > /dev/null 2>$O/err_l$mode.txt
  python profiles/launch_table.py $O/launches_ws$mode.json > $O/launch_table_ws$mode.txt
  grep -E "conv5_wgrad " $O/launch_table_ws$mode.txt | head -30
done
for mode in 1 3 1 3; do
  echo -n "WGRAD_WS=$mode: "; REPMODE_WGRAD_WS=$mode timeout 300 python bench.py --no-cpu-baseline --no-fwd --steps 40 --warmup 15 2>$O/err_$mode.txt | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'], d['config']['final_loss'], {k: (round(v['ms_per_step'],3), v['launches'], round(v['rate'] or 0,1)) for k, v in d['kernels'].items()})"
done | tee $O/bench.log
for r in 0 8 16 32; do
  echo "== REPMODE_RESERVE_CUS=$r"; REPMODE_RESERVE_CUS=$r timeout 300 python tools/comm_overlap_probe.py 30 plain beside 2>/dev/null | grep -E "ms/step|over plain"
done | tee $O/comm_probe.txt
