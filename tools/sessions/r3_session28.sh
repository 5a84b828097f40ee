# Round-3 GPU session 28: shader-clock stamps of the pipelined conv's images (timing build)
cd $GRAFT_REPO_ROOT
for shape in "32 32 32 64 64" "64 64 16 32 32"; do
  echo "== $shape"
  REPMODE_LIB=$GRAFT_REPO_ROOT/repmode_amd/librepmode_hip_timing.so timeout 120 python tools/conv_phase_timing.py $shape 2>&1 | grep "^wg" | cut -c1-1800
done
