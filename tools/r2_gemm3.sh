cd $GRAFT_REPO_ROOT; O=gpurun_out/r2j; mkdir -p $O
python tools/gemm3_microbench.py 2>&1 | grep -v amdgpu.ids | tee $O/gk32.txt
REPMODE_SKIP_TORCH_OPS=1 REPMODE_EXTRA_FLAGS=-DGEMM3_GK=64 bash repmode_amd/csrc/build.sh > /dev/null 2>&1; touch repmode_amd/csrc/gemm3.hip
REPMODE_SKIP_TORCH_OPS=1 REPMODE_EXTRA_FLAGS=-DGEMM3_GK=64 bash repmode_amd/csrc/build.sh | tail -1
python tools/gemm3_microbench.py 2>&1 | grep -v amdgpu.ids | tee $O/gk64.txt
