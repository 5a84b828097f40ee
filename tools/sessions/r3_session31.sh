# Round-3 GPU session 31: wave-specialised conv with lane-read slots and the carry-through brick walk: parity, A/B, stamps
cd $GRAFT_REPO_ROOT; O=gpurun_out/s31; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_round3.py -m gpu -q --maxfail=10 -k "conv5 or mode_conv or block or pipelined or net_golden or train_step" 2>&1 | tail -4 | tee $O/pytest.log
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
for shape in "32 32 32 64 64" "64 64 16 32 32"; do
  echo "== $shape"
  REPMODE_LIB=$GRAFT_REPO_ROOT/repmode_amd/librepmode_hip_timing.so timeout 120 python tools/conv_phase_timing.py $shape 2>&1 | grep "^wg  0" | cut -c1-700
done | tee $O/stamps.log
