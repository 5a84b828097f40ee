# the stride-2 forward / data-gradient kernels with 32-bit position decode (round 3's = variants/k2old), per launch
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4s18; rm -rf $O; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_hip_round4.py tests/test_hip_parity.py tests/test_hip_round3.py -x -q -m gpu -k "k2 or down_up" 2>&1 | tail -2
cd /tmp; export TMPDIR=/tmp
prof() { t=$1; shift
  env "$@" rocprofv3 --kernel-trace --output-format csv -d $O/tr_$t -- python $R/tools/k2s2_microbench.py > $O/log_$t.txt 2>&1
  echo "== $t"; python $R/tools/trace_by_grid.py $O/tr_$t 'k2s2_kernel|k2s2_split' 5 | cut -c1-50,70-
  rm -rf $O/tr_$t; }
prof old REPMODE_LIB=$R/variants/k2old/librepmode_hip.so
prof new A=1
