# Round-3 GPU session 15: gatrep_bwd thread shapes (REPMODE_GATREP_BWD_MODE) warm and cold, per layer size; per-launch
# durations of the stride-2 kernels (what is there to gain)
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/s15; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -q -k "gatrep or filter_grad or train_step or deterministic" --maxfail=10 2>&1 | tail -5 | tee $O/pytest.log
cd /tmp; export TMPDIR=/tmp
for mode in 0 1 2 3 10 20; do
  for cold in "" 1; do
    rm -rf $O/g_$mode$cold
    GATREP_BWD_ONLY=1 GATREP_COLD=$cold REPMODE_GATREP_BWD_MODE=$mode timeout 200 rocprofv3 --kernel-trace --output-format csv -d $O/g_$mode$cold -- python $GRAFT_REPO_ROOT/tools/gatrep_microbench.py > $O/g_$mode$cold.log 2>&1
    echo "== mode $mode cold=${cold:-0}"; python $GRAFT_REPO_ROOT/tools/trace_by_grid.py $O/g_$mode$cold gatrep_bwd 5
  done
done 2>&1 | tee $O/gatrep_bwd_ab.log
rm -rf $O/k2; timeout 200 rocprofv3 --kernel-trace --output-format csv -d $O/k2 -- python $GRAFT_REPO_ROOT/tools/k2s2_microbench.py > $O/k2.log 2>&1
python $GRAFT_REPO_ROOT/tools/trace_by_grid.py $O/k2 'k2s2|tap_transpose' 5 | tee $O/k2s2_launches.log
for mode in 0 1 2 3; do
  echo -n "mode $mode: "; REPMODE_GATREP_BWD_MODE=$mode timeout 300 python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 40 --warmup 15 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'], d['config']['final_loss'], d['fwd']['gatrep_conv_unit']['frac'])"
done | tee $O/bench_modes.log
find $O -name "*.csv" -size +3M -delete; find $O -name "*.db" -delete
