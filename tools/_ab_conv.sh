for shape in "32 32 32 64 64" "64 32 32 64 64" "64 64 16 32 32" "128 64 16 32 32"; do
  for d in _base .; do (cd $GRAFT_REPO_ROOT/$d && python tools/conv_microbench.py $shape 2>&1 | tail -1 | sed "s|^|$d: |"); done
done
