# Round-3 GPU session 1: parity of the new kernels, then same-box A/Bs (deep kernel vs dual launch; row-stationary loop and
# 16-byte stores on levels 0-1; the whole train step under each switch).   gpurun -- 'bash tools/r3_session1.sh'
cd $GRAFT_REPO_ROOT; O=gpurun_out/s1; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > $O/pytest.log
tail -3 $O/pytest.log
for t in 512 256 1024; do
  REPMODE_DEEP_TARGET=$t timeout 300 python tools/deep_microbench.py 8 > $O/deep8_t$t.log 2>&1; echo "target $t"; cat $O/deep8_t$t.log | grep -v amdgpu
done
timeout 300 python tools/deep_microbench.py 24 150 2>&1 | grep -v amdgpu | tee $O/deep24.log
for shape in "32 32 32 64 64 800" "64 32 32 64 64 500" "64 64 16 32 32 1200" "128 64 16 32 32 800"; do
  for rs in 0 1; do for wd in 0 1; do
    echo -n "rowstat=$rs wide=$wd  "
    REPMODE_CONV_ROWSTAT=$rs REPMODE_CONV_WIDE=$wd timeout 120 python tools/conv_microbench.py $shape 2>&1 | tail -1
  done; done
done | tee $O/conv_ab.log
for rep in 1 2; do
  for cfg in "0 0 0" "1 0 0" "0 1 0" "0 0 1" "1 1 1"; do
    set -- $cfg
    echo -n "rowstat=$1 wide=$2 deep=$3  "
    REPMODE_CONV_ROWSTAT=$1 REPMODE_CONV_WIDE=$2 REPMODE_DEEP=$3 timeout 300 python bench.py --no-cpu-baseline --steps 60 --warmup 20 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; f=d['fwd']['gatrep_conv_unit']
print(round(d['ms_per_step'],3),'ms/step  igemm',round(r['achieved'],1),'TF  all conv',round(r['all_conv_kernels']['achieved'],1),'TF', round(r['all_conv_kernels']['ms_per_step'],3),'ms   fwd unit',round(f['frac'],4),'conv',round(f['conv_ms'],3),'gatrep',round(f['gatrep_ms'],3))"
  done
done | tee $O/bench_ab.log
