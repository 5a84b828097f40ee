# BatchNorm rows in flight per thread: cold (HBM) / warm (Infinity Cache) micro-benchmark and the whole step, variants interleaved
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4s12; rm -rf $O; mkdir -p $O; cd $R
run() { # name -> env
  if [ "$1" = product ]; then env "${@:2}"; else env REPMODE_LIB=$R/variants/$1/librepmode_hip.so REPMODE_TORCH_LIB=$R/variants/$1/librepmode_torch.so "${@:2}"; fi
}
for v in bn111 product bn444 bn824; do
  echo "== $v (apply/bwd-apply/bwd-reduce rows)"; run $v python tools/bn_microbench.py cold 2>/dev/null | head -4
  run $v python tools/bn_microbench.py 2>/dev/null | head -4
done
for rep in 1 2; do for v in bn111 product bn444 bn824; do
  run $v timeout 300 python bench.py --no-cpu-baseline --no-fwd --steps 60 --warmup 20 > $O/b_${v}_$rep.json 2>> $O/err.txt
  python -c "
import json; d=json.load(open('$O/b_${v}_$rep.json')); print('$v', round(d['ms_per_step'],3), 'ms/step')"
done; done
