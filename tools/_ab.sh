for shape in "32 32 32 64 64" "64 64 16 32 32" "64 32 32 64 64" "128 128 8 16 16" "256 256 4 8 8"; do
  echo "== $shape"
  REPMODE_LIB=$PWD/repmode_amd/librepmode_hip_old.so python tools/conv_microbench.py $shape 2>&1 | tail -1 | sed 's/^/old: /'
  python tools/conv_microbench.py $shape 2>&1 | tail -1 | sed 's/^/new: /'
done
