# Round-3 GPU session 35: row-stationary tap order in the wave-specialised conv (REPMODE_CONV_PIPE bit 5, 32-channel layers)
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/s35; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_round3.py -m gpu -q --maxfail=10 -k "pipelined" 2>&1 | tail -3 | tee $O/pytest.log
for shape in "32 32 32 64 64 800" "64 32 32 64 64 500"; do
  for rep in 1 2 3; do for pipe in 25 57; do
    echo -n "PIPE=$pipe  "
    REPMODE_CONV_PIPE=$pipe timeout 120 python tools/conv_microbench.py $shape 2>&1 | tail -1
  done; done
done | tee $O/rs_ab.log
for pipe in 25 57 25 57; do
  echo -n "PIPE=$pipe: "; REPMODE_CONV_PIPE=$pipe timeout 300 python bench.py --no-cpu-baseline --steps 40 --warmup 15 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'], d['config']['final_loss'], d['fwd']['gatrep_conv_unit']['frac'])"
done | tee $O/bench.log
