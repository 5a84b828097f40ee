#!/bin/bash
# round 5, session 3: deep_mode_kernel with the slab reduction; timing builds (no 1x1 pass, no stores); the step per mode
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5s3; mkdir -p $O
V=$GRAFT_REPO_ROOT/variants
timeout 300 python -m pytest tests/test_hip_round5.py -x -q 2>&1 | tail -3 | tee $O/tests.txt
{
echo "== product (slab reduction), plan's waves"; timeout 120 python tools/deep_mode_microbench.py 8 200
echo "== product, 4 waves"; REPMODE_DEEP_MODE_WAVES=4 timeout 120 python tools/deep_mode_microbench.py 8 200
echo "== product, 8 waves"; REPMODE_DEEP_MODE_WAVES=8 timeout 120 python tools/deep_mode_microbench.py 8 200
echo "== timing build: no 1x1 pass"; REPMODE_LIB=$V/no1x1/librepmode_hip.so timeout 120 python tools/deep_mode_microbench.py 8 200
echo "== timing build: no stores"; REPMODE_LIB=$V/nostore/librepmode_hip.so timeout 120 python tools/deep_mode_microbench.py 8 200
echo "== timing build: neither"; REPMODE_LIB=$V/nothing/librepmode_hip.so timeout 120 python tools/deep_mode_microbench.py 8 200
} 2>&1 | grep -v "^$\|amdgpu.ids" | tee $O/micro.txt
for m in 0 1 2 3; do
  REPMODE_DEEP_MODE=$m timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-fwd > $O/bench_m$m.json 2> $O/bench_m$m.err
  python - <<PY
import json
try:
    d = json.loads(open('$O/bench_m$m.json').read().strip().splitlines()[-1])
    print('mode $m: %.3f ms/step  roofline %.3f' % (d['ms_per_step'], d['roofline']['frac']), {k: round(v['frac'], 3) for k, v in d['roofline']['by_kernel'].items()})
except Exception as e:
    print('mode $m: FAILED', e)
PY
done | tee $O/bench.txt
