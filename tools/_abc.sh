for shape in "32 32 32 64 64" "64 64 16 32 32"; do
  for d in _base .; do (cd $GRAFT_REPO_ROOT/$d && python tools/conv_microbench.py $shape 2>&1 | tail -1 | sed "s|^|$d: |"); done
done
for shape in "128 128 8 16 16" "256 256 4 8 8" "512 512 2 4 4"; do
  for d in _base .; do (cd $GRAFT_REPO_ROOT/$d && CONV_OUT_F32=1 python tools/conv_microbench.py $shape 2>&1 | tail -1 | sed "s|^|$d: |"); done
done
