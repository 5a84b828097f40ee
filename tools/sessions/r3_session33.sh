# Round-3 GPU session 33: full GPU suite + bench with the wave-specialised conv and filter gradient on by default
cd $GRAFT_REPO_ROOT; O=gpurun_out/s33; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --maxfail=20 2>&1 | tail -5 | tee $O/pytest.log
for rep in 1 2; do
  timeout 300 python bench.py --no-cpu-baseline --steps 40 --warmup 15 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'], d['roofline']['all_conv_kernels']['frac'], d['config']['final_loss'], d['fwd']['gatrep_conv_unit']['frac'], d['fwd']['ms_per_pass'])"
done | tee $O/bench.log
REPMODE_DETERMINISTIC=1 timeout 300 python bench.py --no-cpu-baseline --no-fwd --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('deterministic', d['ms_per_step'], d['config']['final_loss'])" | tee -a $O/bench.log
