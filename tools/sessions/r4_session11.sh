# BatchNorm, the stride-2 stages, 1x1 experts, gatrep: per-launch durations inside the real step, by grid size
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4s11; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/tr -- python $R/bench.py --no-cpu-baseline --no-prof --no-fwd --steps 10 --warmup 3 > $O/trace.log 2>&1
python $R/tools/trace_by_grid.py $O/tr 'bn_|k2s2|gemm3|gatrep|box_sum|expert_mix|Fill|elementwise|tap_tr|thin' 30 > $O/by_grid.txt 2>&1
rm -rf $O/tr
cat $O/by_grid.txt
