set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2f; rm -rf $O; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q -k "not full_size" > $O/pytest.log 2>&1; tail -4 $O/pytest.log
for w in 8 4 0 8 4; do
REPMODE_UNMERGED_MAX_W=$w timeout 300 python bench.py --no-cpu-baseline --no-fwd --steps 60 --warmup 20 > $O/bench_umw$w.json 2>> $O/bench.err; echo "UNMERGED_MAX_W=$w"; cut -c100-260 $O/bench_umw$w.json
done
REPMODE_BN_EPILOGUE=0 timeout 300 python tools/predict_bench.py > $O/predict_noepi.txt 2>&1; tail -1 $O/predict_noepi.txt
timeout 300 python tools/predict_bench.py > $O/predict_epi.txt 2>&1; tail -1 $O/predict_epi.txt
REPMODE_BN_EPILOGUE=0 timeout 300 python tools/predict_bench.py > $O/predict_noepi.txt 2>&1; tail -1 $O/predict_noepi.txt
timeout 300 python tools/predict_bench.py > $O/predict_epi.txt 2>&1; tail -1 $O/predict_epi.txt
