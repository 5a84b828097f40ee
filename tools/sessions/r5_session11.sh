#!/bin/bash
# round 5, session 11: the 1x1 experts' filter gradients (gemm3's filter-gradient shape) as the first workgroups of the one-launch
# data gradient (REPMODE_DEEP_MODE bit 4)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5s11; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_round5.py tests/test_hip_parity.py -x -q -k "deep_mode or batchnorm_statistics or full_size_bf16 or graph" 2>&1 | tail -4 | tee $O/tests.txt
for rep in 1 2 3; do
for m in 15 31; do
  REPMODE_DEEP_MODE=$m timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-fwd --no-prof > $O/bench_m${m}_$rep.json 2> $O/bench_m${m}_$rep.err
  python -c "
import json
d = json.loads(open('$O/bench_m${m}_$rep.json').read().strip().splitlines()[-1])
print('mode $m rep $rep: %.3f ms/step loss %.5f' % (d['ms_per_step'], d['config']['final_loss']))"
done; done 2>&1 | tee $O/bench.txt
