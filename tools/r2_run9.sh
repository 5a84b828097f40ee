set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2k; rm -rf $O; mkdir -p $O
cd $R
python tools/gemm3_microbench.py 2>&1 | grep -v amdgpu.ids | tee $O/gemm3.txt
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -6 $O/pytest.log
for rep in 1 2; do
timeout 300 python bench.py --no-cpu-baseline --no-fwd --steps 60 --warmup 20 > $O/bench_$rep.json 2>> $O/bench.err; cut -c100-260 $O/bench_$rep.json
done
