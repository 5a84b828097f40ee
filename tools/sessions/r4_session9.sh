# Round-4 GPU session 9: filter gradients that are written by plain stores stay out of the pooled memset (REPMODE_WGRAD_PLAN=0/1)
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r4s9; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x --tb=short > $O/pytest_gpu.log 2>&1; tail -5 $O/pytest_gpu.log
for m in 0 1 0 1; do
  echo -n "WGRAD_PLAN=$m: "; REPMODE_WGRAD_PLAN=$m timeout 300 python bench.py --no-cpu-baseline --no-fwd --steps 40 --warmup 15 2>$O/err_$m.txt | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'], d['config']['final_loss'], {k: (round(v['ms_per_step'],3), v['launches']) for k, v in d['kernels'].items()})"
done | tee $O/bench.log
