# Round-3 GPU session 19: the levels 0-1 conv brick on 8 waves (REPMODE_CONV_W8=1: four waves per SIMD at 128 registers) --
# parity, per-layer A/B, train step
cd $GRAFT_REPO_ROOT; O=gpurun_out/s19; mkdir -p $O
REPMODE_CONV_W8=1 timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_round3.py -m gpu -q --maxfail=20 -k "conv5 or conv or mode_conv or block or net_golden" 2>&1 | tail -5 | tee $O/pytest.log
for shape in "32 32 32 64 64 800" "64 32 32 64 64 500" "64 64 16 32 32 1200" "128 64 16 32 32 800"; do
  for rep in 1 2; do for w8 in 0 1; do
    echo -n "W8=$w8  "
    REPMODE_CONV_W8=$w8 timeout 120 python tools/conv_microbench.py $shape 2>&1 | tail -1
  done; done
done | tee $O/w8_ab.log
for w8 in 0 1 0 1; do
  echo -n "W8=$w8: "; REPMODE_CONV_W8=$w8 timeout 300 python bench.py --no-cpu-baseline --steps 40 --warmup 15 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'], d['config']['final_loss'], d['fwd']['gatrep_conv_unit']['frac'])"
done | tee $O/bench.log
