#!/bin/bash
# round 5, session 4: the forward's split target (its six outputs pay for a split with float atomics), the GPU suite with the new
# tests (bench.py's two-rank branch, the bf16 epilogue at the benchmarked shapes, injected errors), the bench line
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5s4; mkdir -p $O
{
for t in 64 128 256; do
echo "== forward split target $t"; REPMODE_DEEP_MODE_TARGET_FWD=$t timeout 120 python tools/deep_mode_microbench.py 8 200
done
echo "== batch 24, target 64"; timeout 120 python tools/deep_mode_microbench.py 24 100
echo "== batch 24, target 256"; REPMODE_DEEP_MODE_TARGET_FWD=256 timeout 120 python tools/deep_mode_microbench.py 24 100
} 2>&1 | grep -v "^$\|amdgpu.ids" | tee $O/micro.txt
timeout 1500 python -m pytest tests -m gpu -q -x > $O/gpu_suite.log 2>&1
echo "gpu suite rc=$?" | tee $O/summary.txt
tail -8 $O/gpu_suite.log
for m in 0 3; do
  REPMODE_DEEP_MODE=$m timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_fwd_m$m.json 2> $O/bench_fwd_m$m.err
  python - <<PY
import json
try:
    d = json.loads(open('$O/bench_fwd_m$m.json').read().strip().splitlines()[-1])
    u = d['fwd']['gatrep_conv_unit']
    print('mode $m: %.3f ms/step roofline %.3f; fwd unit frac %.3f conv %.3f ms gatrep %.3f helper %.3f launches %d+%d+%d; fwd pass %.3f ms' % (d['ms_per_step'], d['roofline']['frac'], u['frac'], u['conv_ms'], u['gatrep_ms'], u['helper_ms'], u['conv_launches'], u['gatrep_launches'], u['helper_launches'], d['fwd']['ms_per_pass']))
except Exception as e:
    print('mode $m fwd: FAILED', e)
PY
done | tee -a $O/summary.txt
