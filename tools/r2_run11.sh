set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2m; rm -rf $O; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -6 $O/pytest.log
for rep in 1 2; do
REPMODE_PREPARE=0 timeout 300 python bench.py --no-cpu-baseline --no-fwd --steps 60 --warmup 20 > $O/bench_noprep_$rep.json 2>> $O/bench.err; echo noprep; cut -c100-260 $O/bench_noprep_$rep.json
timeout 300 python bench.py --no-cpu-baseline --no-fwd --steps 60 --warmup 20 > $O/bench_prep_$rep.json 2>> $O/bench.err; echo prep; cut -c100-260 $O/bench_prep_$rep.json
done
timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/bench_driver.json 2>> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench_driver.json')); print(d['ms_per_step'], d['fwd']['ms_per_pass'], d['fwd']['gatrep_conv_unit']['frac'], d['fwd']['gatrep_conv_unit']['gatrep_ms'], d['fwd']['gatrep_conv_unit']['conv_ms'], d['roofline']['frac'])"
