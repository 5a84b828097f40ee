# round 6, session 12: the stride-2 stages' operands from one launch per forward pass -- parity, step A/B against the previous commit's build (_base)
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_hip_round6.py tests/test_hip_parity.py -x -q -k "stride2 or k2s2 or net_golden or train or down_up" 2>&1 | tail -3
bash tools/ab_bench.sh 2>&1 | tail -4
