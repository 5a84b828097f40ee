# Round-4 GPU session 7: per-job split factors of the dual-expert conv launch (REPMODE_DUAL_KS=0/1): parity, per-layer, train step
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r4s7; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_round3.py -m gpu -q --maxfail=10 --tb=short -k "dual or unmerged or deep or full_size or wgrad_stream" > $O/pytest_new_full.log 2>&1; tail -4 $O/pytest_new_full.log
for ks in 0 1 0 1; do echo "== REPMODE_DUAL_KS=$ks"; REPMODE_DUAL_KS=$ks timeout 200 python tools/deep_microbench.py 8 300 2>/dev/null | tail -6; done | tee $O/micro.txt
for ks in 0 1 0 1; do
  echo -n "DUAL_KS=$ks: "; REPMODE_DUAL_KS=$ks timeout 300 python bench.py --no-cpu-baseline --steps 40 --warmup 15 2>$O/err_$ks.txt | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'], d['config']['final_loss'], d['fwd']['gatrep_conv_unit']['frac'], {k: (round(v['ms_per_step'],3), v['launches'], round(v['rate'] or 0,1)) for k, v in d['kernels'].items()})"
done | tee $O/bench.log
