cd $GRAFT_REPO_ROOT; O=gpurun_out/s13; mkdir -p $O
timeout 900 python -m pytest tests/test_distributed_gpu.py tests/test_hip_parity.py -m gpu -q --maxfail=20 -k "two_ranks or reducer or prepared_in_one or deferred" 2>&1 | tail -30 | tee $O/pytest.log
