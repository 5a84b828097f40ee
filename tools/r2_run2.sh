# round 2: the C++ operator seam -- full GPU tests, then the bench with the driver's arguments and the defaults
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2b; rm -rf $O; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q --durations=5 > $O/pytest.log 2>&1; tail -25 $O/pytest.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/bench.err; cut -c1-700 $O/bench_driver_args.json
timeout 300 python bench.py --no-cpu-baseline > $O/bench_default.json 2>> $O/bench.err; cut -c1-700 $O/bench_default.json
REPMODE_FORK_MAX_W=16 timeout 300 python bench.py --no-cpu-baseline --no-fwd > $O/bench_fork16.json 2>> $O/bench.err; cut -c1-300 $O/bench_fork16.json
timeout 300 python bench.py --no-cpu-baseline --no-fwd --graph > $O/bench_graph.json 2>> $O/bench.err; cut -c1-300 $O/bench_graph.json
tail -5 $O/bench.err
