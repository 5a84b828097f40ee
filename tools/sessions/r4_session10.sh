# Round-4 GPU session 10: row-stationary tap order on the 16-voxel bricks of conv5_ws_kernel (bit 7 of REPMODE_CONV_PIPE):
# parity, level-2 layers tap-major (121) vs row-stationary (249), train step
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r4s10; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_round3.py -m gpu -q --maxfail=10 --tb=short -k "16_voxel" 2>&1 | tail -4 | tee $O/pytest.log
for shape in "64 128" "128 128" "256 128" "128 64" "128 256"; do
  for rep in 1 2; do
    echo -n "tap-major (121)       $shape: "; REPMODE_CONV_PIPE=121 timeout 120 python tools/conv_microbench.py $shape 8 16 16 1500 2>&1 | tail -1
    echo -n "row-stationary (249)  $shape: "; REPMODE_CONV_PIPE=249 timeout 120 python tools/conv_microbench.py $shape 8 16 16 1500 2>&1 | tail -1
  done
done | tee $O/micro.log
for mode in 121 249 121 249; do
  echo -n "PIPE=$mode: "; REPMODE_CONV_PIPE=$mode timeout 300 python bench.py --no-cpu-baseline --steps 40 --warmup 15 2>$O/err_$mode.txt | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'], d['config']['final_loss'], d['fwd']['gatrep_conv_unit']['frac'], {k: (round(v['ms_per_step'],3), v['launches']) for k, v in d['kernels'].items()})"
done | tee $O/bench.log
