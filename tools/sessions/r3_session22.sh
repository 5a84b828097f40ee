# Round-3 GPU session 22: pipelined conv (CW=2 with one-row-ahead filters): parity, A/B, and where the step's time goes with
# and without it (kernel trace of 6 steps each)
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/s22; mkdir -p $O
REPMODE_CONV_PIPE=5 timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_round3.py -m gpu -q --maxfail=10 -k "conv5 or mode_conv or block" 2>&1 | tail -4 | tee $O/pytest.log
for shape in "64 64 16 32 32 1200" "128 64 16 32 32 800"; do
  for rep in 1 2; do for pipe in 0 3 1; do
    echo -n "PIPE=$pipe  "
    REPMODE_CONV_PIPE=$pipe timeout 120 python tools/conv_microbench.py $shape 2>&1 | tail -1
  done; done
done | tee $O/pipe_ab.log
for pipe in 0 1 0 1; do
  echo -n "PIPE=$pipe: "; REPMODE_CONV_PIPE=$pipe timeout 300 python bench.py --no-cpu-baseline --steps 40 --warmup 15 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'], d['config']['final_loss'], d['fwd']['gatrep_conv_unit']['frac'], d['fwd']['ms_per_pass'])"
done | tee $O/bench.log
cd /tmp; export TMPDIR=/tmp
for pipe in 0 1; do
  rm -rf $O/trace$pipe; REPMODE_CONV_PIPE=$pipe timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace$pipe -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-prof --no-fwd --steps 8 --warmup 4 > $O/trace$pipe.log 2>&1
  python $GRAFT_REPO_ROOT/tools/step_launches.py $O/trace$pipe > $O/step_launches_$pipe.txt
done
python $GRAFT_REPO_ROOT/tools/kernel_totals.py $O/step_launches_0.txt $O/step_launches_1.txt | tee $O/totals.txt
find $O -name "*.csv" -size +3M -delete; find $O -name "*.db" -delete
