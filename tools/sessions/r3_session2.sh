# Round-3 GPU session 2: the whole -m gpu suite (no -x), thin kernels vs the fold, atomics vs partial stores, train step A/B
cd $GRAFT_REPO_ROOT; O=gpurun_out/s2; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q --maxfail=30 2>&1 | tail -60 > $O/pytest.log
tail -40 $O/pytest.log
for per in 0 2 4 8; do echo "per_wg=$per"; REPMODE_THIN_PER_WG=$per timeout 200 python tools/thin_microbench.py 8 2>&1 | grep -v amdgpu; done | tee $O/thin.log
hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics -w tools/atomic_scope.hip -o /tmp/atomic_scope && timeout 120 /tmp/atomic_scope | tee $O/atomic.log
for rep in 1 2; do
  for cfg in "0 0" "1 0" "0 1" "1 1"; do
    set -- $cfg
    echo -n "thin=$1 deep=$2  "
    REPMODE_THIN=$1 REPMODE_DEEP=$2 timeout 300 python bench.py --no-cpu-baseline --steps 60 --warmup 20 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; f=d['fwd']['gatrep_conv_unit']
print(round(d['ms_per_step'],3),'ms/step  igemm',round(r['achieved'],1),'TF  all conv',round(r['all_conv_kernels']['achieved'],1),'TF', round(r['all_conv_kernels']['ms_per_step'],3),'ms   fwd unit',round(f['frac'],4),'conv',round(f['conv_ms'],3),'gatrep',round(f['gatrep_ms'],3), 'loss', round(d['config']['final_loss'],4))"
  done
done | tee $O/bench_ab.log
