# BatchNorm grid caps (reduce / apply workgroups): 1024/2048 (product), A 512/1024, B 256/512, C 2048/4096 -- cold micro-benchmark + whole step
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4s25; rm -rf $O; mkdir -p $O; cd $R
run() { if [ "$1" = product ]; then env "${@:2}"; else env REPMODE_LIB=$R/variants/$1/librepmode_hip.so REPMODE_TORCH_LIB=$R/variants/$1/librepmode_torch.so "${@:2}"; fi; }
for v in product bnA bnB bnC; do echo "== $v"; run $v python tools/bn_microbench.py cold 2>/dev/null | head -6; done
for rep in 1 2; do for v in product bnA bnB bnC; do
  run $v timeout 300 python bench.py --no-cpu-baseline --no-fwd --steps 60 --warmup 20 > $O/b_${v}_$rep.json 2>> $O/err.txt
  python -c "
import json; d=json.load(open('$O/b_${v}_$rep.json')); print('$v', round(d['ms_per_step'],3), 'ms/step')"
done; done
