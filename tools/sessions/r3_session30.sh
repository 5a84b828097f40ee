# Round-3 GPU session 30: wave-specialised conv with the straight-line training epilogue and two-row-ahead filters on the
# two-sub-tile form: parity, A/B against PIPE=1 (pipelined) and 0, train step
cd $GRAFT_REPO_ROOT; O=gpurun_out/s30; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_round3.py -m gpu -q --maxfail=10 -k "conv5 or mode_conv or block or pipelined or net_golden" 2>&1 | tail -4 | tee $O/pytest.log
for shape in "32 32 32 64 64 800" "64 32 32 64 64 500" "64 64 16 32 32 1200" "128 64 16 32 32 800"; do
  for rep in 1 2; do for pipe in 0 9; do
    echo -n "PIPE=$pipe  "
    REPMODE_CONV_PIPE=$pipe timeout 120 python tools/conv_microbench.py $shape 2>&1 | tail -1
  done; done
done | tee $O/ws_ab.log
for pipe in 9 0 9 0; do
  echo -n "PIPE=$pipe: "; REPMODE_CONV_PIPE=$pipe timeout 300 python bench.py --no-cpu-baseline --steps 40 --warmup 15 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'], d['roofline']['all_conv_kernels']['frac'], d['config']['final_loss'], d['fwd']['gatrep_conv_unit']['frac'], d['fwd']['ms_per_pass'])"
done | tee $O/bench.log
