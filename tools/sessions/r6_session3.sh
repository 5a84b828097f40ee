cd $GRAFT_REPO_ROOT; O=gpurun_out/r6s2; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_round6.py -x -q -k "one_launch" 2>&1 | grep -E "assert|Error|error|^E " | head -20
cd /tmp && export TMPDIR=/tmp
for f in 0 1; do
REPMODE_BN_FUSED=$f rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof$f -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-prof --no-fwd --steps 10 --warmup 3 > /dev/null 2>&1
python - <<PY
import csv,glob,collections
f=glob.glob('/tmp/prof$f/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=0
for r in rows:
    n=r['Name']
    if 'bn_' in n:
        print('$f', n[:60], r['Calls'], r['TotalDurationNs'], float(r['TotalDurationNs'])/13/1e3,'us/step', float(r['AverageNs'])/1e3)
        tot+=float(r['TotalDurationNs'])
print('BN total us/step', tot/13/1e3)
PY
done
