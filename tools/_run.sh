cd $GRAFT_REPO_ROOT
for rep in 1 2; do
timeout 300 python bench.py --no-cpu-baseline --no-fwd --no-prof --steps 60 --warmup 20 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('eager', round(d['ms_per_step'],3), 'ms/step host_issue', round(d['config']['host_issue_ms_per_step'],2))"
timeout 300 python bench.py --no-cpu-baseline --no-fwd --no-prof --graph --steps 60 --warmup 20 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('graph', round(d['ms_per_step'],3), 'ms/step host_issue', round(d['config']['host_issue_ms_per_step'],2))"
done
