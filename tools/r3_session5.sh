# Round-3 GPU session 5: thin_in1 rework, bf16 tests, a fresh per-family trace of the step
cd $GRAFT_REPO_ROOT; O=gpurun_out/s5; mkdir -p $O
timeout 900 python -m pytest tests/test_bf16_end_to_end_gpu.py tests/test_hip_parity.py -m gpu -q --maxfail=30 -k "bf16_train_iter or net_golden_bf16 or thin or mode_block_golden or net_golden or mode_conv3d_op or eval_batchnorm" 2>&1 | tail -15 > $O/pytest.log
tail -6 $O/pytest.log
for per in 0 2 8; do echo "per_wg=$per"; REPMODE_THIN_PER_WG=$per timeout 200 python tools/thin_microbench.py 8 2>&1 | grep -v amdgpu; done | tee $O/thin.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-prof --no-fwd --steps 10 --warmup 3 > $GRAFT_REPO_ROOT/$O/trace.log 2>&1
python $GRAFT_REPO_ROOT/profiles/analyze_trace.py $GRAFT_REPO_ROOT/$O/trace > $GRAFT_REPO_ROOT/$O/trace_summary.txt 2>&1
rm -rf $GRAFT_REPO_ROOT/$O/trace
cat $GRAFT_REPO_ROOT/$O/trace_summary.txt | tail -42
