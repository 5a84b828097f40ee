# regenerates the artefacts under gpurun_out/r01/ that get copied into profiles/ (run through gpurun)
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r01; rm -rf $O; mkdir -p $O   # (gpurun merges into the local gpurun_out/: delete the local copy first too)
cd $R
python bench.py > $O/bench_line.json 2> $O/bench.err
python bench.py --no-cpu-baseline --prof-all --dump-launches $O/launches_last_step.json > $O/bench_profall.json 2>> $O/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python $R/bench.py --no-cpu-baseline --no-prof --steps 10 --warmup 3 > $O/trace.log 2>&1
tail -c 600 $O/bench_line.json
