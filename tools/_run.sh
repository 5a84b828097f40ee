cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_last.log 2>&1; tail -3 gpurun_out/pytest_last.log
bash tools/ab_env.sh REPMODE_DUAL_WGRAD 0 1
