# Round-4 GPU session 8: kernel trace of the train step (where the time goes now) + the N > 1 code path of bench.py with two
# ranks sharing the GPU over gloo (developer check of the data-parallel path with the build's optimizer; not a measurement)
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4s8; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python $R/bench.py --no-cpu-baseline --no-prof --no-fwd --steps 10 --warmup 3 > $O/trace.log 2>&1
python $R/profiles/analyze_trace.py $O/trace 13 > $O/trace_summary.txt 2>&1
rm -rf $O/trace
cat $O/trace_summary.txt
cd $R
REPMODE_BENCH_SHARE_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 6 --warmup 3 --no-fwd 2>$O/ddp_err.txt | tail -1 | cut -c1-600
tail -3 $O/ddp_err.txt
