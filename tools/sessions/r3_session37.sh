# Round-3 GPU session 37: workgroup -> tile order of the GatRep / expert-layout launches (REPMODE_GATREP_ORDER): parity, kernel
# durations inside a traced step, train step
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/s37; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -q --maxfail=10 -k "gatrep or prepared or expert_frags or unmerged or golden or train_step" 2>&1 | tail -3 | tee $O/pytest.log
cd /tmp; export TMPDIR=/tmp
for ord in 0 1; do
  rm -rf $O/t$ord; REPMODE_GATREP_ORDER=$ord timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/t$ord -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-prof --no-fwd --steps 8 --warmup 4 > $O/t$ord.log 2>&1
  echo "== order $ord"; python $GRAFT_REPO_ROOT/tools/trace_by_grid.py $O/t$ord 'gatrep_fwd_multi|expert_frags_multi' 2
done | tee $O/order.log
cd $GRAFT_REPO_ROOT
for ord in 0 1 0 1; do
  echo -n "ORDER=$ord: "; REPMODE_GATREP_ORDER=$ord timeout 300 python bench.py --no-cpu-baseline --steps 40 --warmup 15 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'], d['config']['final_loss'], d['fwd']['gatrep_conv_unit']['frac'], d['fwd']['gatrep_conv_unit']['gatrep_ms'])"
done | tee $O/bench.log
find $O -name "*.csv" -size +3M -delete; find $O -name "*.db" -delete
