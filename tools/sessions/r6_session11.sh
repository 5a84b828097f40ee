# round 6, session 11: where do the GatRep forward's wave cycles go?  kernel trace + two passes of SQ counters over tools/gatrep_microbench.py
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6s11; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr -- python $R/tools/gatrep_microbench.py > $O/tr.log 2>&1
python3 $R/tools/trace_by_grid.py $O/tr gatrep_fwd 2>&1 | tail -12 | tee $O/by_grid.txt
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_VALU"
P2="SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD"
for pass in 1 2; do
  eval "P=\$P$pass"
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d $O/p$pass -- python $R/tools/gatrep_microbench.py > $O/p$pass.log 2>&1
  f=$(find $O/p$pass -name '*counter_collection.csv' | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'gatrep_fwd' in r['Kernel_Name']]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
for r in rows:
    g = r.get('Grid_Size', r.get('Grid_Size_X', '?'))
    acc[g][r['Counter_Name']] += float(r['Counter_Value']); n[g][r['Counter_Name']] += 1
for g in sorted(acc, key=lambda x: int(x) if x.isdigit() else 0)[-2:]:
    print('grid', g)
    for k in sorted(acc[g]): print('  %-26s %.4g per dispatch' % (k, acc[g][k] / n[g][k]))
PY
  rm -rf $O/p$pass
done 2>&1 | tee $O/pmc.txt
rm -rf $O/tr
