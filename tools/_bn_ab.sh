cd /tmp && export TMPDIR=/tmp
for d in _base .; do
rm -rf /tmp/pb; (cd $GRAFT_REPO_ROOT/$d && rocprofv3 --kernel-trace --output-format csv -d /tmp/pb -- python tools/bn_microbench.py > /tmp/pb.log 2>&1)
python - <<PY
import csv,glob,collections,re
tr=list(csv.DictReader(open(glob.glob('/tmp/pb/**/*_kernel_trace.csv',recursive=True)[0])))
agg=collections.OrderedDict()
for r in tr:
    n=r['Kernel_Name']
    m=re.search(r'(bn_[a-z_]+)_kernel<([a-z ]+)',n)
    if not m: continue
    key=(m.group(1), m.group(2)[:8], r['Grid_Size_X'])
    agg.setdefault(key,[]).append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
print('$d')
for k,v in agg.items(): print('   %-16s %-9s grid %7s  median %6.1f us'%(k[0],k[1],k[2],sorted(v)[len(v)//2]))
PY
done
