# Round-3 GPU session 3: new tests (bf16 end to end, reference-harness call order, lazy wd), PMC diagnostics, A/Bs
cd $GRAFT_REPO_ROOT; O=gpurun_out/s3; mkdir -p $O
timeout 1500 python -m pytest tests/test_bf16_end_to_end_gpu.py tests/test_hip_round3.py tests/test_distributed_gpu.py -m gpu -q --maxfail=30 2>&1 | tail -60 > $O/pytest_new.log
tail -30 $O/pytest_new.log
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q --maxfail=30 2>&1 | tail -15 > $O/pytest_parity.log
tail -5 $O/pytest_parity.log
cat gpurun_out/test_measurements.jsonl
timeout 400 bash tools/pmc_conv.sh > $O/pmc_conv.log 2>&1; cat gpurun_out/pmc_conv/summary.txt
for rep in 1 2; do
  for cfg in "REPMODE_WD_LAZY=0" "REPMODE_WD_LAZY=1" "REPMODE_UNMERGED_MAX_W=16" "REPMODE_DEEP_FWD_MIN=0"; do
    echo -n "$cfg  "
    env $cfg timeout 300 python bench.py --no-cpu-baseline --steps 60 --warmup 20 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; f=d['fwd']['gatrep_conv_unit']
print(round(d['ms_per_step'],3),'ms/step  igemm',round(r['achieved'],1),'TF  all conv',round(r['all_conv_kernels']['achieved'],1),'TF', round(r['all_conv_kernels']['ms_per_step'],3),'ms   fwd unit',round(f['frac'],4),'conv',round(f['conv_ms'],3),'gatrep',round(f['gatrep_ms'],3), 'fwd ms', round(d['fwd']['ms_per_pass'],3), 'loss', round(d['config']['final_loss'],4))"
  done
done | tee $O/bench_ab.log
