# Collectives beside the backward pass on one box (tools/comm_overlap_probe.py), then the RCCL kernels' own durations under
# rocprofv3 in the "beside" configuration.   gpurun -- 'bash tools/r3_comm.sh'
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/comm; mkdir -p $O
timeout 600 python tools/comm_overlap_probe.py 30 2>&1 | grep -v amdgpu | tee $O/probe.txt
timeout 300 python tools/comm_overlap_probe.py 30 bf16 plain beside alone 2>&1 | grep -v amdgpu | tee $O/probe_bf16.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python $GRAFT_REPO_ROOT/tools/comm_overlap_probe.py 10 beside > $O/trace.log 2>&1
python3 - "$(find $O/trace -name '*_kernel_stats.csv' | head -1)" <<'PY' | tee $O/rccl_kernels.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n = r['Name']
    if 'ccl' in n.lower() or 'conv5_igemm' in n or 'conv5_wgrad_bf16' in n:
        print('%-90s calls %5s  avg %9.1f us  total %9.1f ms' % (n.replace('(anonymous namespace)::', '')[:90], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6))
PY
rm -rf $O/trace
