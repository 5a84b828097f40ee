# Round-3 GPU session 23: pipelined conv on by default -- the full GPU suite, bench
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/s23; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --maxfail=20 2>&1 | tail -8 | tee $O/pytest.log
for pipe in 1 0 1 0; do
  echo -n "PIPE=$pipe: "; REPMODE_CONV_PIPE=$pipe timeout 300 python bench.py --no-cpu-baseline --steps 40 --warmup 15 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'], d['roofline']['all_conv_kernels']['frac'], d['config']['final_loss'], d['fwd']['gatrep_conv_unit']['frac'], d['fwd']['ms_per_pass'])"
done | tee $O/bench.log
timeout 600 python tools/predict_bench.py 2>&1 | tail -2 | tee $O/predict.log
