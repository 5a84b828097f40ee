# Round-3 GPU session 36: full GPU suite on the final defaults, then the profile refresh
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s36
timeout 1500 python -m pytest tests -m gpu -q --maxfail=20 2>&1 | tail -4 | tee gpurun_out/s36/pytest.log
bash tools/refresh_profiles.sh > gpurun_out/s36/refresh.log 2>&1
tail -3 gpurun_out/s36/refresh.log | cut -c1-300
