# The round's standard GPU call: the -m gpu suite, then the bench with the driver's arguments and with the defaults.
#   gpurun --timeout 2400 -- 'bash tools/gpu_check.sh'
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/check; rm -rf $O; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q --durations=5 > $O/pytest.log 2>&1; tail -12 $O/pytest.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/bench.err; cut -c1-400 $O/bench_driver_args.json
timeout 300 python bench.py --no-cpu-baseline > $O/bench_default.json 2>> $O/bench.err; cut -c1-400 $O/bench_default.json
tail -3 $O/bench.err
