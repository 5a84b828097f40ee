set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2l; rm -rf $O; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -6 $O/pytest.log
for rep in 1 2; do
timeout 300 python bench.py --no-cpu-baseline --no-fwd --steps 60 --warmup 20 > $O/bench_$rep.json 2>> $O/bench.err; cut -c100-260 $O/bench_$rep.json
done
timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/bench_driver.json 2>> $O/bench.err; cut -c1-2500 $O/bench_driver.json
