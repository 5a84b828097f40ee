# L2 / fabric counters of the level-0 / level-1 filter gradient (batch 8): hit rate of the XCDs' L2s, requests that go to the
# fabric (Infinity Cache / HBM).   gpurun -- 'bash tools/pmc_wgrad_l2.sh'
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_wgrad_l2; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
P1="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum"
P2="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCP_TCC_READ_REQ_sum FETCH_SIZE"
P3="TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TA_TCP_STATE_READ_sum TCC_TAG_STALL_sum"
for shape in "32 32 32 64 64" "64 64 16 32 32"; do
  tag=$(echo $shape | tr ' ' '_')
  for pass in 1 2 3; do
    eval "P=\$P$pass"
    WGRAD_ITERS=30,20 timeout 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $O/${tag}_p$pass -- python $R/tools/wgrad_phase_timing.py $shape > $O/${tag}_p$pass.log 2>&1
    echo "wgrad $shape pass $pass"; python3 $R/tools/pmc_kernel.py $O/${tag}_p$pass conv5_wgrad
  done
done 2>&1 | tee $O/summary.txt
rm -rf $O/*_p1/ $O/*_p2/ $O/*_p3/
