# Round-4 GPU session 1: conv5_ws_kernel's 16-voxel-brick form (level 2, bf16 output, no split reduction): parity, per-layer
# A/B against round 3's split-K float-output launches, train step A/B (REPMODE_CONV_PIPE=57 restores round 3's behaviour)
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r4s1; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_round3.py -m gpu -q --maxfail=10 -k "16_voxel or pipelined" 2>&1 | tail -5 | tee $O/pytest_new.log
for shape in "64 128" "128 128" "256 128" "128 64" "128 256"; do
  for rep in 1 2; do
    echo -n "old (split-K float out) $shape: "; REPMODE_CONV_PIPE=57 CONV_OUT_F32=1 timeout 120 python tools/conv_microbench.py $shape 8 16 16 1500 2>&1 | tail -1
    echo -n "new (ws16 bf16 out)      $shape: "; timeout 120 python tools/conv_microbench.py $shape 8 16 16 1500 2>&1 | tail -1
  done
done | tee $O/micro.log
for mode in 57 121 57 121; do
  echo -n "PIPE=$mode: "; REPMODE_CONV_PIPE=$mode timeout 300 python bench.py --no-cpu-baseline --steps 40 --warmup 15 2>$O/err_$mode.txt | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'], d['config']['final_loss'], d['fwd']['gatrep_conv_unit']['frac'], d['fwd']['gatrep_conv_unit']['conv_ms'], {k: (round(v['ms_per_step'],3), v['launches']) for k, v in d['kernels'].items()})"
done | tee $O/bench.log
timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -5 | tee $O/pytest_all.log
