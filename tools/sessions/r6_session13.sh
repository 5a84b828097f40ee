# round 6, session 13 (second half): the filter gradient's loader waves and MFMA phase -- the calls behind profiles/r06_wgrad_loader.txt.
# Variant builds (built locally, they travel with the snapshot; .gpurunignore must not list variants/ for these calls):
#   bash tools/build_variant.sh timing  "-DRM_CONV_TIMING"                  conv5_wgrad.hip     # phase stamps
#   bash tools/build_variant.sh noload  "-DRM_WG_NOLOAD"                    conv5_wgrad.hip     # tile loop without its global loads
#   bash tools/build_variant.sh nostage "-DRM_WG_NOSTAGE"                   conv5_wgrad.hip     # ... without the transposition into LDS
#   bash tools/build_variant.sh floor   "-DRM_WG_NOLOAD -DRM_WG_NOSTAGE"    conv5_wgrad.hip
#   hipcc --offload-arch=gfx950 -O3 tools/mfma_rate.hip -o tools/build/mfma_rate
cd $GRAFT_REPO_ROOT
V() { export REPMODE_LIB=$GRAFT_REPO_ROOT/variants/$1/librepmode_hip.so REPMODE_TORCH_LIB=$GRAFT_REPO_ROOT/variants/$1/librepmode_torch.so; }
# 1. per layer shape, product and ablations (same call, same box)
for v in product noload nostage floor; do
  [ $v != product ] && V $v
  echo "== $v"; WGRAD_MODES=0 python tools/wgrad_microbench.py 8 300 2>&1 | tail -5
done
unset REPMODE_LIB REPMODE_TORCH_LIB
# 2. stamps: the MFMA waves' barrier wait / MFMA phase per tile, the loader waves' per-tile sums
V timing
python tools/wgrad_phase_timing.py 32 32 32 64 64 2>&1 | tail -6
python tools/wgrad_loader_timing.py 32 32 32 64 64 2>&1 | tail -3
WGRAD_STAMPS=1 python tools/wgrad_deep_microbench.py 8 50 2>&1 | head -8
unset REPMODE_LIB REPMODE_TORCH_LIB
# 3. L2 / fabric counters of the level-0 / level-1 launches; the MFMA issue-rate table; the forms against each other
bash tools/pmc_wgrad_l2.sh
tools/build/mfma_rate
python tools/wgrad_fuzz.py 120 7 | tail -3
python tools/wgrad_deep_microbench.py 8 200 | tail -5
