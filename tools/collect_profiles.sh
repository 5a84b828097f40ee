# copies the judged summaries from gpurun_out/${REPMODE_PROFILE_TAG:-r06}/ (scratch) into profiles/ (tracked)
set -e
cd "$(dirname "$0")/.."; S=gpurun_out/${REPMODE_PROFILE_TAG:-r06}; D=profiles
cp $S/bench_line.json $D/${REPMODE_PROFILE_TAG:-r06}_bench_line.json
cp $S/bench_line_driver_args.json $D/${REPMODE_PROFILE_TAG:-r06}_bench_line_driver_args.json
cp $S/bench_line_b24.json $D/${REPMODE_PROFILE_TAG:-r06}_bench_line_b24.json
cp $S/bench_line_device_volumes.json $D/${REPMODE_PROFILE_TAG:-r06}_bench_line_device_volumes.json
cp $S/kernel_stats_b8.csv $D/${REPMODE_PROFILE_TAG:-r06}_bench_kernel_stats.csv
cp $S/kernel_stats_b24.csv $D/${REPMODE_PROFILE_TAG:-r06}_b24_kernel_stats.csv
cp $S/kernel_stats_alone_b8.csv $D/${REPMODE_PROFILE_TAG:-r06}_bench_kernel_stats_conv_alone.csv
cp $S/kernel_stats_alone_b24.csv $D/${REPMODE_PROFILE_TAG:-r06}_b24_kernel_stats_conv_alone.csv
cp $S/trace_summary_alone_b8.txt $D/${REPMODE_PROFILE_TAG:-r06}_bench_trace_summary_conv_alone.txt
cp $S/trace_summary_alone_b24.txt $D/${REPMODE_PROFILE_TAG:-r06}_b24_trace_summary_conv_alone.txt
cp $S/trace_summary_b8.txt $D/${REPMODE_PROFILE_TAG:-r06}_bench_trace_summary.txt
cp $S/trace_summary_b24.txt $D/${REPMODE_PROFILE_TAG:-r06}_b24_trace_summary.txt
cp $S/pmc_traffic_b8.json $D/${REPMODE_PROFILE_TAG:-r06}_pmc_traffic.json
cp $S/pmc_traffic_b24.json $D/${REPMODE_PROFILE_TAG:-r06}_b24_pmc_traffic.json
cp $S/launches_last_step.json $D/${REPMODE_PROFILE_TAG:-r06}_launches_last_step.json
cp $S/launch_table.txt $D/${REPMODE_PROFILE_TAG:-r06}_launch_table.txt
cp $S/predict.txt $D/${REPMODE_PROFILE_TAG:-r06}_predict.txt
cp $S/bench_line_graph.json $D/${REPMODE_PROFILE_TAG:-r06}_bench_line_graph.json
cp $S/power_cap.txt $D/${REPMODE_PROFILE_TAG:-r06}_power_cap.txt
ls -la $D
