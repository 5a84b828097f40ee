cd $GRAFT_REPO_ROOT
for shape in "32 32 32 64 64 600" "64 64 16 32 32 900"; do
  for pipe in 1 9 17 25 0; do
    echo -n "PIPE=$pipe  "
    REPMODE_CONV_PIPE=$pipe timeout 120 python tools/conv_microbench.py $shape 2>&1 | tail -1
  done
done
