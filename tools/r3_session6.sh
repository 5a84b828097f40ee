# Round-3 GPU session 6: BatchNorm apply kernels with 4 rows in flight -- _base (HEAD before the change) vs the working tree
cd $GRAFT_REPO_ROOT; O=gpurun_out/s6; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -q -k "bn_relu or mode_block_golden or net_golden" 2>&1 | tail -3
for rep in 1 2; do for d in _base .; do echo "== $d"; (cd $GRAFT_REPO_ROOT/$d && timeout 200 python tools/bn_microbench.py 2>&1 | grep -v amdgpu); done; done | tee $O/bn.log
bash tools/ab_bench.sh 2>&1 | grep -v amdgpu | tee $O/bench_ab.log
