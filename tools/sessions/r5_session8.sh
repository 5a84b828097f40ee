#!/bin/bash
# round 5, session 8: BatchNorm statistics from the one-launch forward's epilogue (no statistics pass behind a per-expert block)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5s8; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_round5.py tests/test_bf16_end_to_end_gpu.py -x -q 2>&1 | tail -6 | tee $O/tests.txt
for rep in 1 2 3; do
for m in 3 7; do
  REPMODE_DEEP_MODE=$m timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-fwd --no-prof > $O/bench_m${m}_$rep.json 2> $O/bench_m${m}_$rep.err
  python -c "
import json
d = json.loads(open('$O/bench_m${m}_$rep.json').read().strip().splitlines()[-1])
print('mode $m rep $rep: %.3f ms/step loss %.5f' % (d['ms_per_step'], d['config']['final_loss']))"
done; done 2>&1 | tee $O/bench.txt
