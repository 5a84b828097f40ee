set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2g; rm -rf $O; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -4 $O/pytest.log
for rep in 1 2; do
REPMODE_DUAL_LAUNCH=0 timeout 300 python bench.py --no-cpu-baseline --no-fwd --steps 60 --warmup 20 > $O/bench_nodual_$rep.json 2>> $O/bench.err; echo nodual; cut -c100-260 $O/bench_nodual_$rep.json
timeout 300 python bench.py --no-cpu-baseline --no-fwd --steps 60 --warmup 20 > $O/bench_dual_$rep.json 2>> $O/bench.err; echo dual; cut -c100-260 $O/bench_dual_$rep.json
done
