#!/bin/bash
# round 5, session 9: the rotated tap-row / chunk order again, now that the LDS atomics are gone (HBM traffic of deep_mode is 4x
# its filters: do the rotated phases of an XCD's workgroups defeat the L2?)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5s9; mkdir -p $O
V=$GRAFT_REPO_ROOT/variants
{
for rep in 1 2; do
echo "== product (rotation) $rep"; timeout 120 python tools/deep_mode_microbench.py 8 300
echo "== no rotation $rep"; REPMODE_LIB=$V/norot/librepmode_hip.so timeout 120 python tools/deep_mode_microbench.py 8 300
done
} 2>&1 | grep -v "^$\|amdgpu.ids" | tee $O/micro.txt
for rep in 1 2; do
for v in prod norot; do
  if [ $v = norot ]; then export REPMODE_LIB=$V/norot/librepmode_hip.so REPMODE_TORCH_LIB=$V/norot/librepmode_torch.so; else unset REPMODE_LIB REPMODE_TORCH_LIB; fi
  timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-fwd --no-prof > $O/bench_${v}_$rep.json 2> $O/bench_${v}_$rep.err
  python -c "
import json
d = json.loads(open('$O/bench_${v}_$rep.json').read().strip().splitlines()[-1])
print('$v rep $rep: %.3f ms/step' % d['ms_per_step'])"
done; done 2>&1 | tee $O/bench.txt
