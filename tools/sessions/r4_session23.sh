R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests -x -q -m gpu -k "wgrad or dual or unmerged or per_expert or expert_layout" 2>&1 | tail -2
for rep in 1 2; do
echo "== old (variants/k2old)"; REPMODE_LIB=$R/variants/k2old/librepmode_hip.so python tools/deep_microbench.py 8 200 2>&1 | tail -6 | cut -c1-22,150-
echo "== new"; python tools/deep_microbench.py 8 200 2>&1 | tail -6 | cut -c1-22,150-
done
