# Round-4 GPU session 3: session 2 again after its findings (torch's fused Adam does not move version counters -> optimizer-step
# hook; noise-calibrated bf16 emulation bounds): tests with full failure output, whole suite, train step A/B vs torch's Adam
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r4s3; mkdir -p $O
rm -f gpurun_out/test_measurements.jsonl
timeout 1500 python -m pytest tests/test_hip_round4.py tests/test_bf16_end_to_end_gpu.py -m gpu -q --maxfail=20 --tb=short > $O/pytest_new_full.log 2>&1; tail -15 $O/pytest_new_full.log
cp gpurun_out/test_measurements.jsonl $O/ 2>/dev/null
for mode in 0 1 0 1; do
  echo -n "ADAM=$mode: "; REPMODE_ADAM=$mode timeout 300 python bench.py --no-cpu-baseline --steps 40 --warmup 15 2>$O/err_$mode.txt | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'], d['config']['final_loss'], d['fwd']['gatrep_conv_unit']['frac'], d['fwd']['gatrep_conv_unit']['gatrep_ms'], {k: (round(v['ms_per_step'],3), v['launches'], round(v['rate'] or 0,1)) for k, v in d['kernels'].items()}, {k: round(v['frac'],3) for k,v in d['roofline']['by_kernel'].items()})"
done | tee $O/bench.log
timeout 1500 python -m pytest tests -m gpu -q -x --deselect tests/test_hip_round4.py --deselect tests/test_bf16_end_to_end_gpu.py 2>&1 | tail -5 | tee $O/pytest_rest.log
