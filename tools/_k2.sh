cd /tmp && export TMPDIR=/tmp
for lib in librepmode_hip.so librepmode_hip_k2w512.so librepmode_hip_k2w256.so; do
rm -rf /tmp/pk; REPMODE_LIB=$GRAFT_REPO_ROOT/repmode_amd/$lib rocprofv3 --kernel-trace --output-format csv -d /tmp/pk -- python $GRAFT_REPO_ROOT/tools/k2s2_microbench.py > /tmp/pk.log 2>&1
python - <<PY
import csv,glob,collections,re
tr=list(csv.DictReader(open(glob.glob('/tmp/pk/**/*_kernel_trace.csv',recursive=True)[0])))
agg=collections.OrderedDict()
for r in tr:
    n=r['Kernel_Name']
    m=re.search(r'(k2s2_wgrad_kernel|k2s2_kernel<[a-z ]+, (true|false)>)',n)
    if not m: continue
    key=(m.group(1)[-14:], r['Grid_Size_X'], r['Grid_Size_Y'])
    agg.setdefault(key,[]).append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
print('$lib: '+'  '.join('%s/%s:%.1f'%(k[0],k[1],sorted(v)[len(v)//2]) for k,v in agg.items()))
PY
done
