cd $GRAFT_REPO_ROOT; O=gpurun_out/s8; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_parity.py tests/test_distributed_gpu.py -m gpu -q --maxfail=20 -k "k2s2 or k2_frags or net_golden or train_iter or full_size_net or two_ranks or reducer" 2>&1 | tail -8 | tee $O/pytest.log
bash tools/r3_comm.sh
