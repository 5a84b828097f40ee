# the per-expert levels' filter gradient: wave-specialised form (REPMODE_WGRAD_WS8=1) against the two-workgroup form
R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests -x -q -m gpu -k "wgrad or dual or unmerged or per_expert or expert_layout" 2>&1 | tail -2
for rep in 1 2; do for v in 0 1; do
echo "== REPMODE_WGRAD_WS8=$v"; REPMODE_WGRAD_WS8=$v python tools/deep_microbench.py 8 200 2>&1 | tail -6 | cut -c1-22,150-
done; done
