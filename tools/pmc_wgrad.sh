# PMC diagnostics + shader-clock phase stamps of the level-0 / level-1 filter gradient (batch 8, 8 slots): where the wave
# cycles go.  Two passes of 8 SQ counters (no trace domains besides --kernel-trace), then the timing build's stamps.
#   gpurun -- 'bash tools/pmc_wgrad.sh'
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_wgrad; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
P2="SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM"
for shape in "32 32 32 64 64" "64 64 16 32 32"; do
  tag=$(echo $shape | tr ' ' '_')
  for pass in 1 2; do
    eval "P=\$P$pass"
    WGRAD_ITERS=30,20 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $O/${tag}_p$pass -- python $R/tools/wgrad_phase_timing.py $shape > $O/${tag}_p$pass.log 2>&1
    f=$(find $O/${tag}_p$pass -name '*counter_collection.csv' | head -1)
    python3 - "$f" "wgrad $shape pass $pass" <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'conv5_wgrad' in r['Kernel_Name']]
acc = collections.defaultdict(float); n = collections.Counter()
for r in rows:
    acc[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
print(sys.argv[2], '(%d dispatches)' % max(n.values(), default=0))
for k in sorted(acc): print('  %-34s %.4g per dispatch' % (k, acc[k] / n[k]))
PY
  done
  echo "== phase stamps (timing build), $shape"
  REPMODE_LIB=$R/repmode_amd/librepmode_hip_timing.so WGRAD_ITERS=300,300 python $R/tools/wgrad_phase_timing.py $shape 2>&1 | tail -7
  echo "== product build rate, $shape"
  python $R/tools/wgrad_phase_timing.py $shape 2>&1 | tail -1
done 2>&1 | tee $O/summary.txt
rm -rf $O/*_p1/ $O/*_p2/
