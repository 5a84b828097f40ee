#!/bin/bash
# round 5, session 7: 1x1 experts inside the data gradient's chunk loop; one-sample tiles for the level-4 forward
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5s7; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_round5.py -x -q 2>&1 | tail -4 | tee $O/tests.txt
{
echo "== product"; timeout 120 python tools/deep_mode_microbench.py 8 200
echo "== two-sample tiles kept (REPMODE_DEEP_MODE_S1=0)"; REPMODE_DEEP_MODE_S1=0 timeout 120 python tools/deep_mode_microbench.py 8 200
echo "== batch 24"; timeout 120 python tools/deep_mode_microbench.py 24 100
} 2>&1 | grep -v "^$\|amdgpu.ids" | tee $O/micro.txt
for rep in 1 2; do
for m in 0 3; do
  REPMODE_DEEP_MODE=$m timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-fwd --no-prof > $O/bench_m${m}_$rep.json 2> $O/bench_m${m}_$rep.err
  python -c "
import json
d = json.loads(open('$O/bench_m${m}_$rep.json').read().strip().splitlines()[-1])
print('mode $m rep $rep: %.3f ms/step' % d['ms_per_step'])"
done; done 2>&1 | tee $O/bench.txt
