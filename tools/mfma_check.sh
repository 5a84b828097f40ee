R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02m; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/a -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-prof --no-fwd > $O/a.log 2>&1
rocprofv3 --pmc MfmaUtil --kernel-trace --output-format csv -d $O/b -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-prof --no-fwd > $O/b.log 2>&1
for d in a b; do f=$(find $O/$d -name '*counter_collection.csv' | head -1); head -1 $f > $O/$d.csv; grep -E "conv5_igemm|conv5_wgrad" $f | tail -60 >> $O/$d.csv; t=$(find $O/$d -name '*kernel_trace.csv' | head -1); head -1 $t > $O/${d}_trace.csv; grep -E "conv5_igemm|conv5_wgrad" $t | tail -30 >> $O/${d}_trace.csv; rm -rf $O/$d; done
