# round 6, session 10: GatRep forward with two tap classes -- parity (everything that merges filters), the launch inside a step
cd $GRAFT_REPO_ROOT; O=gpurun_out/r6s10; mkdir -p $O
timeout 1200 python -m pytest tests/test_hip_parity.py tests/test_hip_round3.py tests/test_hip_round4.py -x -q 2>&1 | tail -3
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --prof-all --dump-launches $O/l.json > $O/bench.json 2>/dev/null
python profiles/launch_table.py $O/l.json | grep -E "gatrep_fwd" | head -6 | tee $O/gatrep_fwd.txt
python -c "import json; d=json.load(open('$O/bench.json')); print(d['ms_per_step'], d['config']['ms_per_step_unprofiled'], d['fwd']['gatrep_conv_unit']['frac'], d['fwd']['gatrep_conv_unit']['gatrep_ms'])" | tee -a $O/gatrep_fwd.txt
