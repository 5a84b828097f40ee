set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2e; rm -rf $O; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q --durations=5 > $O/pytest.log 2>&1; tail -12 $O/pytest.log
for rep in 1 2; do
REPMODE_BN_EPILOGUE=0 timeout 300 python bench.py --no-cpu-baseline --no-fwd --steps 60 --warmup 20 > $O/bench_noepi_$rep.json 2>> $O/bench.err; cut -c1-330 $O/bench_noepi_$rep.json
timeout 300 python bench.py --no-cpu-baseline --no-fwd --steps 60 --warmup 20 > $O/bench_epi_$rep.json 2>> $O/bench.err; cut -c1-330 $O/bench_epi_$rep.json
done
REPMODE_BN_EPILOGUE=0 timeout 300 python tools/predict_bench.py > $O/predict_noepi.txt 2>&1; tail -2 $O/predict_noepi.txt
timeout 300 python tools/predict_bench.py > $O/predict_epi.txt 2>&1; tail -2 $O/predict_epi.txt
tail -3 $O/bench.err
