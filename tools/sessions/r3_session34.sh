# Round-3 GPU session 34: item order of the wave-specialised conv (REPMODE_CONV_PIPE bit 4: z first): parity, per-layer time,
# HBM traffic (FETCH_SIZE) at batch 8 and 24, train step
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/s34; mkdir -p $O
REPMODE_CONV_PIPE=29 timeout 600 python -m pytest tests/test_hip_round3.py -m gpu -q --maxfail=10 -k "pipelined" 2>&1 | tail -3 | tee $O/pytest.log
for shape in "32 32 32 64 64 800" "64 32 32 64 64 500" "64 64 16 32 32 1200"; do
  for rep in 1 2; do for pipe in 9 25; do
    echo -n "PIPE=$pipe  "
    REPMODE_CONV_PIPE=$pipe timeout 120 python tools/conv_microbench.py $shape 2>&1 | tail -1
  done; done
done | tee $O/order_ab.log
cd /tmp; export TMPDIR=/tmp
for b in 8 24; do for pipe in 9 25; do
  rm -rf $O/f; REPMODE_TAIL=0 REPMODE_CONV_PIPE=$pipe timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/f -- python $GRAFT_REPO_ROOT/bench.py --batch $b --steps 1 --warmup 1 --no-cpu-baseline --no-prof --no-fwd > $O/f.log 2>&1
  python3 - "$(find $O/f -name '*counter_collection.csv' | head -1)" "batch $b PIPE=$pipe" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'conv5_ws' in r['Kernel_Name'] and r['Counter_Name'] == 'FETCH_SIZE']
print(sys.argv[2], 'conv5_ws: %d launches, HBM read %.1f MB per launch (FETCH_SIZE x 2 KiB)' % (len(rows), sum(float(r['Counter_Value']) for r in rows) * 2 * 1024 / max(len(rows), 1) / 1e6))
PY
done; done | tee $O/fetch.log
rm -rf $O/f
cd $GRAFT_REPO_ROOT
for pipe in 9 25 9 25; do
  echo -n "PIPE=$pipe: "; REPMODE_CONV_PIPE=$pipe timeout 300 python bench.py --no-cpu-baseline --steps 40 --warmup 15 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'], d['config']['final_loss'], d['fwd']['gatrep_conv_unit']['frac'])"
done | tee $O/bench.log
for pipe in 9 25; do
  echo -n "batch 24 PIPE=$pipe: "; REPMODE_CONV_PIPE=$pipe timeout 300 python bench.py --no-cpu-baseline --batch 24 --steps 20 --warmup 8 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'], d['fwd']['gatrep_conv_unit']['frac'])"
done | tee -a $O/bench.log
