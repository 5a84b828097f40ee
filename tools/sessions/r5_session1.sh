#!/bin/bash
# round 5, session 1: the one-launch per-expert block (csrc/deep_mode.hip) -- parity, then the train step with / without it
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r5s1
O=gpurun_out/r5s1
timeout 600 python -m pytest tests/test_hip_round5.py -x -q > $O/new_tests.log 2>&1
echo "new tests rc=$?" | tee -a $O/summary.txt
tail -5 $O/new_tests.log
timeout 900 python -m pytest tests -m gpu -q -x > $O/gpu_suite.log 2>&1
echo "gpu suite rc=$?" | tee -a $O/summary.txt
tail -5 $O/gpu_suite.log
for rep in 1 2; do
  for m in 0 3; do
    REPMODE_DEEP_MODE=$m timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-fwd > $O/bench_m${m}_$rep.json 2> $O/bench_m${m}_$rep.err
    python - <<PY
import json
try:
    d = json.loads(open('$O/bench_m${m}_$rep.json').read().strip().splitlines()[-1])
    print('mode $m rep $rep: %.3f ms/step  roofline %.3f' % (d['ms_per_step'], d['roofline']['frac']), {k: round(v['frac'], 3) for k, v in d['roofline']['by_kernel'].items()})
except Exception as e:
    print('mode $m rep $rep: FAILED', e)
PY
  done
done | tee -a $O/summary.txt
for m in 0 3; do
  REPMODE_DEEP_MODE=$m timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_fwd_m$m.json 2> $O/bench_fwd_m$m.err
  python - <<PY
import json
try:
    d = json.loads(open('$O/bench_fwd_m$m.json').read().strip().splitlines()[-1])
    u = d['fwd']['gatrep_conv_unit']
    print('mode $m: %.3f ms/step; fwd unit frac %.3f conv %.3f ms gatrep %.3f helper %.3f launches %d+%d+%d; fwd pass %.3f ms' % (d['ms_per_step'], u['frac'], u['conv_ms'], u['gatrep_ms'], u['helper_ms'], u['conv_launches'], u['gatrep_launches'], u['helper_launches'], d['fwd']['ms_per_pass']))
except Exception as e:
    print('mode $m fwd: FAILED', e)
PY
done | tee -a $O/summary.txt
