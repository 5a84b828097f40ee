cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-prof --no-fwd --steps 10 --warmup 3 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r6s7 && python tools/step_launches.py /tmp/prof > gpurun_out/r6s7/step_launches.txt 2>&1; wc -l gpurun_out/r6s7/step_launches.txt
