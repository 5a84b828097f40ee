# Round-4 GPU session 6: stream-K loader waves with two tiles in flight: parity, per-launch table, the 16-voxel tile on / off
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r4s6; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_round3.py tests/test_hip_parity.py -m gpu -q --maxfail=10 --tb=short -k "wgrad or mode_conv3d_op or full_size" > $O/pytest_new_full.log 2>&1; tail -4 $O/pytest_new_full.log
for sk in 0 1; do
  REPMODE_WGRAD_SK16=$sk timeout 300 python bench.py --no-cpu-baseline --no-fwd --prof-all --dump-launches $O/launches_sk$sk.json --steps 12 --warmup 6 > /dev/null 2>$O/err_l$sk.txt
  python profiles/launch_table.py $O/launches_sk$sk.json > $O/launch_table_sk$sk.txt
  echo "== REPMODE_WGRAD_SK16=$sk"; grep -E "conv5_wgrad " $O/launch_table_sk$sk.txt | head -30
done | tee $O/tables.txt
for sk in 0 1 0 1; do
  echo -n "SK16=$sk: "; REPMODE_WGRAD_SK16=$sk timeout 300 python bench.py --no-cpu-baseline --no-fwd --steps 40 --warmup 15 2>$O/err_$sk.txt | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'], d['config']['final_loss'], {k: (round(v['ms_per_step'],3), v['launches'], round(v['rate'] or 0,1)) for k, v in d['kernels'].items()})"
done | tee $O/bench.log
