# Round-4 GPU session 2: the optimizer pass (csrc/adam.hip) -- parity, the reference-size tests, the bf16-emulating oracle's
# measured errors, and the train step A/B against torch's fused Adam (REPMODE_ADAM=0)
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r4s2; mkdir -p $O
rm -f gpurun_out/test_measurements.jsonl
timeout 1500 python -m pytest tests/test_hip_round4.py tests/test_bf16_end_to_end_gpu.py -m gpu -q --maxfail=20 --durations=12 2>&1 | tail -40 | tee $O/pytest_new.log
cp gpurun_out/test_measurements.jsonl $O/ 2>/dev/null
for mode in 0 1 0 1; do
  echo -n "ADAM=$mode: "; REPMODE_ADAM=$mode timeout 300 python bench.py --no-cpu-baseline --steps 40 --warmup 15 2>$O/err_$mode.txt | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'], d['config']['final_loss'], d['fwd']['gatrep_conv_unit']['frac'], d['fwd']['gatrep_conv_unit']['gatrep_ms'], {k: (round(v['ms_per_step'],3), v['launches'], round(v['rate'] or 0,1)) for k, v in d['kernels'].items()}, {k: round(v['frac'],3) for k,v in d['roofline']['by_kernel'].items()})"
done | tee $O/bench.log
tail -5 $O/err_1.txt
