#!/bin/bash
# round 5, session 6: one-launch operand check, graph test; kernel trace of the step (where the time goes now)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5s6; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_round5.py -x -q -k "graph or stale" 2>&1 | tail -4 | tee $O/tests.txt
for rep in 1 2; do
for v in 1 0; do
  REPMODE_FRAG_VERIFY=$v timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-fwd --no-prof > $O/bench_v${v}_$rep.json 2> $O/bench_v${v}_$rep.err
  python -c "
import json
d = json.loads(open('$O/bench_v${v}_$rep.json').read().strip().splitlines()[-1])
print('verify $v rep $rep: %.3f ms/step' % d['ms_per_step'])"
done; done 2>&1 | tee $O/bench.txt
bash tools/trace_bench.sh r5s6
python profiles/analyze_trace.py gpurun_out/prof_r5s6 13 > $O/trace_summary.txt 2>&1
tail -45 $O/trace_summary.txt
rm -rf gpurun_out/prof_r5s6/*/*_kernel_trace.csv.bak
