# adam_frags_kernel: LDS staging in parameter order (conflict-free 8-byte writes) against the [tap][row][col] staging
R=$GRAFT_REPO_ROOT; cd $R
timeout 600 python -m pytest tests/test_hip_round4.py -x -q -m gpu -k "adam or operand" 2>&1 | tail -3
for rep in 1 2 3; do
  echo "== new"; python tools/adam_microbench.py 2>/dev/null | head -1
  echo "== old"; REPMODE_LIB=$R/variants/adam_old/librepmode_hip.so python tools/adam_microbench.py 2>/dev/null | head -1
done
