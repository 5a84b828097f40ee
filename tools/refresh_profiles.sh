# Regenerates, under gpurun_out/${REPMODE_PROFILE_TAG:-r06}/, every artefact that gets copied into profiles/ (run through gpurun, ~6 GPU-minutes):
#   bash tools/refresh_profiles.sh          then locally:  bash tools/collect_profiles.sh
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${REPMODE_PROFILE_TAG:-r06}; rm -rf $O; mkdir -p $O
cd $R
python bench.py --no-cpu-baseline --no-fwd --prof-all --dump-launches $O/launches_last_step.json > $O/bench_profall.json 2>> $O/bench.err
cd /tmp && export TMPDIR=/tmp
for b in 8 24; do
  # (1) the step as shipped: time per step by kernel family
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_b$b -- python $R/bench.py --batch $b --no-cpu-baseline --no-prof --no-fwd --steps 10 --warmup 3 > $O/trace_b$b.log 2>&1
  python $R/profiles/analyze_trace.py $O/trace_b$b 13 > $O/trace_summary_b$b.txt 2>&1
  cp $(find $O/trace_b$b -name '*_kernel_stats.csv' | head -1) $O/kernel_stats_b$b.csv
  rm -rf $O/trace_b$b
  # (2) the conv launches ALONE (REPMODE_TAIL=0: the gate backward / layout transposes that otherwise ride in the first
  # workgroups of the data-gradient conv launches run as kernels of their own) -- what bench.py's `roofline` times on its
  # event-timed steps: per-kernel averages for the agreement check, and the PMC passes
  REPMODE_TAIL=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_alone_b$b -- python $R/bench.py --batch $b --no-cpu-baseline --no-prof --no-fwd --steps 10 --warmup 3 > $O/trace_alone_b$b.log 2>&1
  python $R/profiles/analyze_trace.py $O/trace_alone_b$b 13 > $O/trace_summary_alone_b$b.txt 2>&1
  cp $(find $O/trace_alone_b$b -name '*_kernel_stats.csv' | head -1) $O/kernel_stats_alone_b$b.csv
  rm -rf $O/trace_alone_b$b
  mkdir -p $O/pmc_b$b
  export REPMODE_TAIL=0
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_b$b/pmc_FETCH_SIZE -- python $R/bench.py --batch $b --steps 1 --warmup 1 --no-cpu-baseline --no-prof --no-fwd > $O/pmc_b$b/fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_b$b/pmc_WRITE_SIZE -- python $R/bench.py --batch $b --steps 1 --warmup 1 --no-cpu-baseline --no-prof --no-fwd > $O/pmc_b$b/write.log 2>&1
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_b$b/pmc_MFMA -- python $R/bench.py --batch $b --steps 1 --warmup 1 --no-cpu-baseline --no-prof --no-fwd > $O/pmc_b$b/mfma.log 2>&1
  unset REPMODE_TAIL
  python $R/profiles/pmc_summary.py $O/pmc_b$b $b bf16 > $O/pmc_traffic_b$b.json 2> $O/pmc_b$b/summary.err
  rm -rf $O/pmc_b$b/pmc_*
done
# the bench lines last: roofline.traffic is read from profiles/*_pmc_traffic.json, which must be THIS build's pass
cp $O/pmc_traffic_b8.json $R/profiles/${REPMODE_PROFILE_TAG:-r06}_pmc_traffic.json; cp $O/pmc_traffic_b24.json $R/profiles/${REPMODE_PROFILE_TAG:-r06}_b24_pmc_traffic.json
cd $R
python bench.py > $O/bench_line.json 2> $O/bench.err
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_line_driver_args.json 2>> $O/bench.err
python bench.py --no-cpu-baseline --batch 24 --steps 40 --warmup 10 > $O/bench_line_b24.json 2>> $O/bench.err
python bench.py --no-cpu-baseline --no-fwd --device-volumes > $O/bench_line_device_volumes.json 2>> $O/bench.err
python bench.py --no-cpu-baseline --no-fwd --no-prof --graph --steps 40 --warmup 10 > $O/bench_line_graph.json 2>> $O/bench.err
bash $R/tools/power_cap.sh > $O/power_cap.txt 2>&1
cd /tmp
python $R/profiles/launch_table.py $O/launches_last_step.json > $O/launch_table.txt
timeout 300 python $R/tools/predict_bench.py > $O/predict.txt 2>&1
tail -c 400 $O/bench_line.json; tail -3 $O/trace_summary_b8.txt; head -c 600 $O/pmc_traffic_b8.json; tail -1 $O/predict.txt
