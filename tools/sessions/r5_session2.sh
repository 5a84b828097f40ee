#!/bin/bash
# round 5, session 2: deep_mode_kernel per layer -- rotation of the tap rows / chunks (the lockstep diagnosis), prefetch depth, waves
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5s2; mkdir -p $O
V=$GRAFT_REPO_ROOT/variants
{
echo "== product (rotation, 1 row ahead)"; timeout 120 python tools/deep_mode_microbench.py 8 200 old
echo "== no rotation"; REPMODE_LIB=$V/norot/librepmode_hip.so timeout 120 python tools/deep_mode_microbench.py 8 200
echo "== 2 rows ahead"; REPMODE_LIB=$V/pf2/librepmode_hip.so timeout 120 python tools/deep_mode_microbench.py 8 200
echo "== product, 4 waves"; REPMODE_DEEP_MODE_WAVES=4 timeout 120 python tools/deep_mode_microbench.py 8 200
echo "== product, 8 waves"; REPMODE_DEEP_MODE_WAVES=8 timeout 120 python tools/deep_mode_microbench.py 8 200
echo "== product, target 512"; REPMODE_DEEP_MODE_TARGET=512 timeout 120 python tools/deep_mode_microbench.py 8 200
echo "== product, batch 24"; timeout 120 python tools/deep_mode_microbench.py 24 100 old
} 2>&1 | grep -v "^$" | tee $O/micro.txt
timeout 300 python -m pytest tests/test_hip_round5.py -x -q 2>&1 | tail -3 | tee $O/tests.txt
for m in 0 3; do
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
