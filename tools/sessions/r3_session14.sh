# the 4x4x16 tile on level-1 layers (Cout 64), with two and with three waves per SIMD; level 2 with three
cd $GRAFT_REPO_ROOT; O=gpurun_out/s14; mkdir -p $O
W3=$GRAFT_REPO_ROOT/repmode_amd/librepmode_hip_w3.so
for shape in "64 64 16 32 32 1200" "128 64 16 32 32 800" "32 64 16 32 32 1500"; do
  for rep in 1 2; do
    echo -n "X32 tile (product)        "; timeout 120 python tools/conv_microbench.py $shape 2>&1 | tail -1
    echo -n "X16 tile, 2 waves/SIMD    "; REPMODE_CONV_X16_AT=64 timeout 120 python tools/conv_microbench.py $shape 2>&1 | tail -1
    echo -n "X16 tile, 3 waves/SIMD    "; REPMODE_CONV_X16_AT=64 REPMODE_LIB=$W3 timeout 120 python tools/conv_microbench.py $shape 2>&1 | tail -1
  done
done | tee $O/x16.log
for rep in 1 2; do
  echo -n "level 2 float out, 2 waves "; CONV_OUT_F32=1 timeout 120 python tools/conv_microbench.py 128 128 8 16 16 1500 2>&1 | tail -1
  echo -n "level 2 float out, 3 waves "; CONV_OUT_F32=1 REPMODE_LIB=$W3 timeout 120 python tools/conv_microbench.py 128 128 8 16 16 1500 2>&1 | tail -1
done | tee -a $O/x16.log
