cd /tmp && export TMPDIR=/tmp
for lib in librepmode_hip.so; do
rm -rf /tmp/pg; REPMODE_LIB=$GRAFT_REPO_ROOT/repmode_amd/$lib rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pg -- python $GRAFT_REPO_ROOT/tools/gatrep_microbench.py > /tmp/pg.log 2>&1
python - <<PY
import csv,glob,collections
tr=list(csv.DictReader(open(glob.glob('/tmp/pg/**/*_kernel_trace.csv',recursive=True)[0])))
tr.sort(key=lambda r:int(r['Start_Timestamp']))
agg=collections.OrderedDict()
for r in tr:
    n=r['Kernel_Name']
    if 'gatrep' not in n and 'gate_bwd' not in n: continue
    key=(n.split('(')[0][-40:], r['Grid_Size_X'], r['Grid_Size_Y'], r['Grid_Size_Z'])
    agg.setdefault(key,[]).append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
print('$lib')
for k,v in agg.items():
    v=sorted(v); print('  %-42s grid %7s %5s %4s n=%3d median %7.1f us'%(k[0],k[1],k[2],k[3],len(v),v[len(v)//2]))
PY
done
