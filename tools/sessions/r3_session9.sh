# Round-3 GPU session 9: LDS-DMA halo staging in conv5_igemm -- parity, then same-box A/B against the register-path build
cd $GRAFT_REPO_ROOT; O=gpurun_out/s9; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_round3.py -m gpu -q --maxfail=20 2>&1 | tail -8 | tee $O/pytest.log
for shape in "32 32 32 64 64 800" "64 32 32 64 64 500" "64 64 16 32 32 1200" "128 64 16 32 32 800"; do
  for rep in 1 2; do for lib in "" nodma; do
    echo -n "lib=${lib:-product(dma)}  "
    REPMODE_LIB=${lib:+$GRAFT_REPO_ROOT/repmode_amd/librepmode_hip_$lib.so} timeout 120 python tools/conv_microbench.py $shape 2>&1 | tail -1
  done; done
done | tee $O/dma_ab.log
for rep in 1 2; do for lib in "" nodma; do
  echo -n "float-output level 2 128->128, lib=${lib:-product(dma)}  "
  CONV_OUT_F32=1 REPMODE_LIB=${lib:+$GRAFT_REPO_ROOT/repmode_amd/librepmode_hip_$lib.so} timeout 120 python tools/conv_microbench.py 128 128 8 16 16 1500 2>&1 | tail -1
done; done | tee -a $O/dma_ab.log
timeout 300 python bench.py --no-cpu-baseline --steps 60 --warmup 20 2>/dev/null | tail -1 > $O/bench.json; python -c "
import json; d=json.load(open('$O/bench.json')); print(d['ms_per_step'], d['roofline']['frac'], d['roofline']['all_conv_kernels'], d['fwd']['gatrep_conv_unit']['frac'], d['config']['final_loss'])"
