#!/bin/bash
# round 5, session 5: capturable own optimizer (graph step), on-device operand check, then the step: eager vs graph, check on / off
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5s5; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_round5.py tests/test_hip_round4.py tests/test_hip_parity.py -x -q -k "graph or stale or adam or deep_mode or checkpoint" > $O/tests.log 2>&1
echo "tests rc=$?" | tee $O/summary.txt
tail -5 $O/tests.log
run() {  # tag, env..., -- args
  tag=$1; shift
  env "$@" timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-fwd $EXTRA > $O/bench_$tag.json 2> $O/bench_$tag.err
  python - <<PY
import json
try:
    d = json.loads(open('$O/bench_$tag.json').read().strip().splitlines()[-1])
    print('$tag: %.3f ms/step  roofline %.3f  host enqueue %.2f ms graph=%s' % (d['ms_per_step'], d['roofline']['frac'] if 'roofline' in d else -1, d['config']['host_enqueue_ms_per_step'], d['config']['hip_graph']))
except Exception as e:
    print('$tag: FAILED', e)
PY
}
for rep in 1 2; do
  EXTRA="" run eager_$rep A=1
  EXTRA="" run noverify_$rep REPMODE_FRAG_VERIFY=0
  EXTRA="--graph --no-prof" run graph_$rep A=1
  EXTRA="--no-prof" run eager_noprof_$rep A=1
done 2>&1 | tee -a $O/summary.txt
