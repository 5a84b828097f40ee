cd $GRAFT_REPO_ROOT; O=gpurun_out/r6s5; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_round6.py -x -q -k "one_launch" 2>&1 | grep -E "^E |assert" | head -12
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-prof --no-fwd --steps 10 --warmup 3 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && python profiles/analyze_trace.py /tmp/prof 13 > $O/trace_summary.txt 2>&1; tail -60 $O/trace_summary.txt
