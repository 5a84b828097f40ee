# same-box A/B of the whole train step: _base (a built copy of an earlier commit) vs the working tree, interleaved
for rep in 1 2; do
for d in _base .; do
  (cd $GRAFT_REPO_ROOT/$d && timeout 300 python bench.py --no-cpu-baseline --steps 60 --warmup 20 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$d', round(d['ms_per_step'],2), 'ms/step', round(d['roofline']['achieved'],1), 'TF conv')")
done; done
