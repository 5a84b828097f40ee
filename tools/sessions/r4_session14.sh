# adam_frags_kernel: where the time goes (timing-only builds: 1 = no operand write, 2 = no staging either, 3 = and no LDS reservation)
R=$GRAFT_REPO_ROOT; cd $R
for rep in 1 2; do for v in product r1 r4 t1 t2 t2r4 t3 t3r4 t3n256; do
  if [ $v = product ]; then L=; else L=$R/variants/$v/librepmode_hip.so; fi
  echo "$v: $(REPMODE_LIB=$L python tools/adam_microbench.py 2>/dev/null | head -1)"
done; done
