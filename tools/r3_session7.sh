# Round-3 GPU session 7: dual-expert launch with the two jobs interleaved across XCDs
cd $GRAFT_REPO_ROOT; O=gpurun_out/s7; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q --maxfail=30 2>&1 | tail -6 | tee $O/pytest.log
timeout 300 python tools/deep_microbench.py 8 2>&1 | grep -v amdgpu | tee $O/deep8.log
for rep in 1 2; do
  timeout 300 python bench.py --no-cpu-baseline --steps 60 --warmup 20 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; f=d['fwd']['gatrep_conv_unit']
print(round(d['ms_per_step'],3),'ms/step  igemm',round(r['achieved'],1),'TF  all conv',round(r['all_conv_kernels']['achieved'],1),'TF', round(r['all_conv_kernels']['ms_per_step'],3),'ms   fwd unit',round(f['frac'],4),'conv',round(f['conv_ms'],3),'gatrep',round(f['gatrep_ms'],3), 'fwd ms', round(d['fwd']['ms_per_pass'],3))"
done | tee $O/bench.log
