set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2i; rm -rf $O; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -6 $O/pytest.log
for rep in 1 2; do
timeout 300 python bench.py --no-cpu-baseline --no-fwd --steps 60 --warmup 20 > $O/bench_$rep.json 2>> $O/bench.err; cut -c100-260 $O/bench_$rep.json
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python $R/bench.py --no-cpu-baseline --no-prof --no-fwd --steps 10 --warmup 3 > $O/trace.log 2>&1
cd $R; python profiles/analyze_trace.py $O/trace 13 > $O/trace_summary.txt 2>&1; sed -n 1,30p $O/trace_summary.txt
rm -rf $O/trace
