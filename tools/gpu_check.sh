# The standard GPU call of a work session (what the driver runs at round end, in one gpurun call): the -m gpu suite, smoke(),
# then bench.py with the driver's arguments.     gpurun --timeout 1800 -- 'bash tools/gpu_check.sh'
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/check; mkdir -p $O
rm -f gpurun_out/test_measurements.jsonl
timeout 1500 python -m pytest tests -m gpu -q -x --tb=short > $O/pytest_gpu.log 2>&1; tail -6 $O/pytest_gpu.log
cp gpurun_out/test_measurements.jsonl $O/ 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py --gpus 1 --steps 20 --warmup 5 2>$O/bench_err.txt | tail -1 > $O/bench_line.json; python -c "
import json; d=json.load(open('$O/bench_line.json')); print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['traffic_over_algorithmic'], d['fwd']['gatrep_conv_unit']['frac'], d['cpu_baseline']['value'], d['cpu_baseline']['sample'][-90:])"
