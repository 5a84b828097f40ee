# Round-3 GPU session 32: the wave-specialised filter gradient (REPMODE_WGRAD_WS=1): parity, per-layer A/B, train step
cd $GRAFT_REPO_ROOT; O=gpurun_out/s32; mkdir -p $O
REPMODE_WGRAD_WS=1 timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_round3.py tests/test_bf16_end_to_end_gpu.py -m gpu -q --maxfail=10 -k "wgrad or filter_grad or mode_conv or train or block or golden or pair" 2>&1 | tail -4 | tee $O/pytest.log
for shape in "32 32 32 64 64" "64 32 32 64 64" "64 64 16 32 32" "128 64 16 32 32"; do
  for rep in 1 2; do for ws in 0 1; do
    echo -n "WS=$ws  "
    REPMODE_WGRAD_WS=$ws WGRAD_ITERS=300,500 timeout 120 python tools/wgrad_phase_timing.py $shape 2>&1 | grep "^wgrad"
  done; done
done | tee $O/ws_ab.log
for ws in 1 0 1 0; do
  echo -n "WGRAD_WS=$ws: "; REPMODE_WGRAD_WS=$ws timeout 300 python bench.py --no-cpu-baseline --steps 40 --warmup 15 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'], d['config']['final_loss'], d['fwd']['gatrep_conv_unit']['frac'])"
done | tee $O/bench.log
