# Round-3 GPU session 18: full GPU suite on the new gatrep_bwd / k2s2 split / pinned plan copy / patch gather+blend kernels,
# the zero pool's contents, the bench line, sliding-window inference time
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/s18; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --maxfail=20 2>&1 | tail -8 | tee $O/pytest.log
REPMODE_POOL_DEBUG=1 timeout 300 python bench.py --no-cpu-baseline --steps 40 --warmup 15 2> $O/bench.err | tail -1 > $O/bench.json
grep "zero pool" $O/bench.err | sort | uniq -c | cut -c1-2500 > $O/pool.txt
python -c "
import json; d=json.load(open('$O/bench.json')); print(d['ms_per_step'], d['roofline']['frac'], d['roofline']['all_conv_kernels']['frac'], d['fwd']['gatrep_conv_unit']['frac'], d['config']['final_loss'])" | tee $O/bench.txt
timeout 600 python tools/predict_bench.py 2>&1 | tail -5 | tee $O/predict.log
