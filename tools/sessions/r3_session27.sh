# Round-3 GPU session 27: the pipelined conv with every halo image fetched from one L2-resident brick (real data): the cost of
# halo fetches that miss
cd $GRAFT_REPO_ROOT
for shape in "32 32 32 64 64 600" "64 64 16 32 32 900"; do
  for rep in 1 2; do for lib in "" SAMEHALO; do
    echo -n "lib=${lib:-product}  "
    REPMODE_LIB=${lib:+$GRAFT_REPO_ROOT/repmode_amd/librepmode_hip_$lib.so} timeout 120 python tools/conv_microbench.py $shape 2>&1 | tail -1
  done; done
done
