# k2s2_wgrad_kernel: one address decode per tile + the next tile's rows in flight + a priced split, against round 3's kernel
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4s15; rm -rf $O; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_hip_parity.py tests/test_hip_round3.py -x -q -m gpu -k "k2 or down_up" 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
prof() { # tag, env...
  t=$1; shift
  env "$@" rocprofv3 --kernel-trace --output-format csv -d $O/tr_$t -- python $R/tools/k2s2_microbench.py > $O/log_$t.txt 2>&1
  echo "== $t"; python $R/tools/trace_by_grid.py $O/tr_$t 'k2s2_wgrad' 5 | cut -c1-20,70-
  rm -rf $O/tr_$t
}
prof old REPMODE_LIB=$R/variants/k2old/librepmode_hip.so
prof new_auto A=1
prof new_512 REPMODE_K2W_BLOCKS=512
prof new_256 REPMODE_K2W_BLOCKS=256
prof new_128 REPMODE_K2W_BLOCKS=128
prof new_64 REPMODE_K2W_BLOCKS=64
