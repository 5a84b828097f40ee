cd $GRAFT_REPO_ROOT; O=gpurun_out/s11; mkdir -p $O
timeout 900 python -m pytest tests/test_bf16_end_to_end_gpu.py -m gpu -q -k "deterministic" 2>&1 | tail -3 | tee $O/pytest.log
for det in 0 1; do
  echo -n "REPMODE_DETERMINISTIC=$det  "
  REPMODE_DETERMINISTIC=$det timeout 300 python bench.py --no-cpu-baseline --no-fwd --steps 40 --warmup 10 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3),'ms/step', 'loss', d['config']['final_loss'])"
done | tee $O/det_bench.log
