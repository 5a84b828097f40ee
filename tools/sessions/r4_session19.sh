# gemm3 (the three 1x1 experts): vector fast path against round 3's kernel (variants/k2old), GPU time per launch
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4s19; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
prof() { t=$1; shift
  env "$@" rocprofv3 --kernel-trace --output-format csv -d $O/tr_$t -- python $R/tools/gemm3_microbench.py > $O/log_$t.txt 2>&1
  echo "== $t"; python $R/tools/trace_by_grid.py $O/tr_$t 'gemm3_kernel<true' 5 | cut -c1-40,70-
  rm -rf $O/tr_$t; }
prof old REPMODE_LIB=$R/variants/k2old/librepmode_hip.so
prof new A=1
cd $R; timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "gemm3 or unmerged or per_expert" 2>&1 | tail -2
