# PMC diagnostics of the level-0 / level-1 convolution (batch 8): where the wave cycles go, for the two-workgroup form
# (REPMODE_CONV_PIPE=0) and the pipelined one (1).  Two passes of 8 SQ counters each (no trace domains besides --kernel-trace).
#   gpurun -- 'bash tools/pmc_conv.sh'
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_conv; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
P2="SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM"
for shape in "32 32 32 64 64" "64 64 16 32 32"; do
 for pipe in 0 1; do
  for pass in 1 2; do
    eval "P=\$P$pass"
    REPMODE_CONV_PIPE=$pipe CONV_WARM=50 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $O/p${pipe}_p$pass -- python $R/tools/conv_microbench.py $shape 20 > $O/p${pipe}_p$pass.log 2>&1
    f=$(find $O/p${pipe}_p$pass -name '*counter_collection.csv' | head -1)
    python3 - "$f" "conv $shape pipe=$pipe pass $pass" <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'conv5_' in r['Kernel_Name']]
acc = collections.defaultdict(float); n = collections.Counter()
for r in rows:
    acc[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
print(sys.argv[2], '(%d dispatches)' % max(n.values(), default=0))
for k in sorted(acc): print('  %-34s %.4g per dispatch' % (k, acc[k] / n[k]))
PY
    rm -rf $O/p${pipe}_p$pass
  done
 done
done 2>&1 | tee $O/summary.txt
