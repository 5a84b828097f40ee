# Round-3 GPU session 17: the stride-2 kernels' wave-split variant for under-filled launches (parity, per-launch durations),
# gatrep_bwd's final two shapes (parity), where the step's small PyTorch launches come from, the step's launch list
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/s17; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q -k "k2s2 or k2_frags or gatrep or filter_grad or train_step or deterministic or golden" --maxfail=10 2>&1 | tail -5 | tee $O/pytest.log
cd /tmp; export TMPDIR=/tmp
for sb in 0 256; do
  rm -rf $O/k2_$sb; REPMODE_K2S2_SPLIT_BELOW=$sb timeout 200 rocprofv3 --kernel-trace --output-format csv -d $O/k2_$sb -- python $GRAFT_REPO_ROOT/tools/k2s2_microbench.py > $O/k2_$sb.log 2>&1
  echo "== split below $sb"; python $GRAFT_REPO_ROOT/tools/trace_by_grid.py $O/k2_$sb 'k2s2_(split_)?kernel' 5
done | tee $O/k2s2_launches.log
for sb in 0 256 0 256; do
  echo -n "split below $sb: "; REPMODE_K2S2_SPLIT_BELOW=$sb timeout 300 python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 40 --warmup 15 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'], d['config']['final_loss'], d['fwd']['gatrep_conv_unit']['frac'])"
done | tee $O/bench.log
cd $GRAFT_REPO_ROOT; timeout 300 python tools/find_fills.py 2>&1 | tail -75 > $O/fills.log
cd /tmp; rm -rf $O/trace; timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-prof --no-fwd --steps 6 --warmup 3 > $O/trace.log 2>&1
python $GRAFT_REPO_ROOT/tools/step_launches.py $O/trace > $O/step_launches.txt; tail -1 $O/step_launches.txt
find $O -name "*.csv" -size +3M -delete; find $O -name "*.db" -delete
