# round 2, first GPU call: the new bench-configuration parity tests + a baseline of HEAD with the driver's arguments
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2a; rm -rf $O; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 > $O/pytest.log 2>&1; tail -15 $O/pytest.log
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/bench.err; cut -c1-1500 $O/bench_driver_args.json
timeout 300 python bench.py --no-cpu-baseline > $O/bench_default.json 2>> $O/bench.err; cut -c1-600 $O/bench_default.json
timeout 300 python bench.py --no-cpu-baseline --batch 24 --steps 40 --warmup 10 > $O/bench_b24.json 2>> $O/bench.err; cut -c1-600 $O/bench_b24.json
tail -5 $O/bench.err
