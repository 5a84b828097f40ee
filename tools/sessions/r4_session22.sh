R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests/test_hip_round3.py tests/test_hip_parity.py -x -q -m gpu -k "deep or unmerged or per_expert" 2>&1 | tail -2
python tools/deep_microbench.py 8 200 2>&1 | tail -6 | cut -c1-150
BASE_VARIANT=k2old bash tools/sessions/r4_session20.sh
