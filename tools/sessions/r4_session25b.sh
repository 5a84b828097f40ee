# BatchNorm grid caps: round 3's (bnOld: 1024 / 2048), 512 / 1024 everywhere (bnFlat), 512 / 1024 with the wide grids kept for level 0 (product)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4s25; mkdir -p $O; cd $R
run() { if [ "$1" = product ]; then env "${@:2}"; else env REPMODE_LIB=$R/variants/$1/librepmode_hip.so REPMODE_TORCH_LIB=$R/variants/$1/librepmode_torch.so "${@:2}"; fi; }
for rep in 1 2 3; do for v in bnOld bnFlat product; do
  run $v timeout 300 python bench.py --no-cpu-baseline --no-fwd --steps 60 --warmup 20 > $O/b_${v}_$rep.json 2>> $O/err.txt
  python -c "
import json; d=json.load(open('$O/b_${v}_$rep.json')); print('$v', round(d['ms_per_step'],3), 'ms/step')"
done; done
timeout 600 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "bn" 2>&1 | tail -2
