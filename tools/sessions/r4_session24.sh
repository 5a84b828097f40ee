# the per-expert levels' filter gradient: two-workgroup form (REPMODE_WGRAD_WS8=0), wave-specialised per unit (1), persistent (2)
R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests -x -q -m gpu -k "wgrad or dual or unmerged or per_expert or expert_layout" 2>&1 | tail -2
for rep in 1 2; do for v in 0 1 2; do
echo "== REPMODE_WGRAD_WS8=$v"; REPMODE_WGRAD_WS8=$v timeout 120 python tools/deep_microbench.py 8 200 2>&1 | tail -6 | cut -c1-22,150-
done; done
