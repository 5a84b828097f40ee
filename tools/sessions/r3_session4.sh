# Round-3 GPU session 4: whole suite (hardware bf16 pack everywhere), GatRep-in-conv A/B, prefetch-depth variants, bf16 gradient diagnostics
cd $GRAFT_REPO_ROOT; O=gpurun_out/s4; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --maxfail=30 2>&1 | tail -40 > $O/pytest.log
tail -12 $O/pytest.log
timeout 300 python tools/merge_ab.py 2>&1 | grep -v amdgpu | tee $O/merge_ab.txt
for shape in "32 32 32 64 64 800" "64 32 32 64 64 500" "64 64 16 32 32 1200" "128 64 16 32 32 800"; do
  for rep in 1 2; do for lib in "" pre2 pre2b; do
    echo -n "lib=${lib:-product}  "
    REPMODE_LIB=${lib:+$GRAFT_REPO_ROOT/repmode_amd/librepmode_hip_$lib.so} timeout 120 python tools/conv_microbench.py $shape 2>&1 | tail -1
  done; done
done | tee $O/pre2_ab.log
for rep in 1 2; do for lib in "" pre2 pre2b; do
  echo -n "float-output level 2 128->128, lib=${lib:-product}  "
  CONV_OUT_F32=1 REPMODE_LIB=${lib:+$GRAFT_REPO_ROOT/repmode_amd/librepmode_hip_$lib.so} timeout 120 python tools/conv_microbench.py 128 128 8 16 16 1500 2>&1 | tail -1
done; done | tee -a $O/pre2_ab.log
timeout 600 python tools/bf16_grad_diag.py 2>&1 | grep -v amdgpu > $O/bf16_grad_diag.txt; tail -4 $O/bf16_grad_diag.txt
timeout 300 python bench.py --no-cpu-baseline --steps 60 --warmup 20 2>/dev/null | tail -1 > $O/bench.json; python -c "
import json; d=json.load(open('$O/bench.json')); print(d['ms_per_step'], d['roofline']['frac'], d['roofline']['all_conv_kernels'], d['fwd']['gatrep_conv_unit']['frac'])"
