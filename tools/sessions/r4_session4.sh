# Round-4 GPU session 4: stream-K filter gradient (REPMODE_WGRAD_WS=3 vs 1): parity, train step A/B with the family's time
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r4s4; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_round3.py tests/test_hip_parity.py tests/test_hip_round4.py -m gpu -q --maxfail=10 --tb=short -k "wgrad or own_adam or expert_operands or mode_conv3d_op or full_size" > $O/pytest_new_full.log 2>&1; tail -8 $O/pytest_new_full.log
for mode in 1 3 1 3; do
  echo -n "WGRAD_WS=$mode: "; REPMODE_WGRAD_WS=$mode timeout 300 python bench.py --no-cpu-baseline --no-fwd --steps 40 --warmup 15 2>$O/err_$mode.txt | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'], d['config']['final_loss'], {k: (round(v['ms_per_step'],3), v['launches'], round(v['rate'] or 0,1)) for k, v in d['kernels'].items()})"
done | tee $O/bench.log
tail -3 $O/err_3.txt
