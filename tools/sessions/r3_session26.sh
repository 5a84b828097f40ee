# Round-3 GPU session 26: what the pipelined conv's waits are -- timing builds without the halo fetches / with zero filter fragments
cd $GRAFT_REPO_ROOT
for shape in "32 32 32 64 64 600" "64 64 16 32 32 900"; do
  for lib in "" NOHALO NOFILT NONE; do
    echo -n "lib=${lib:-product}  "
    REPMODE_LIB=${lib:+$GRAFT_REPO_ROOT/repmode_amd/librepmode_hip_$lib.so} timeout 120 python tools/conv_microbench.py $shape 2>&1 | tail -1
  done
done
