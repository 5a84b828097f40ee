# launch-geometry probes: box kernels with 8 channels per workgroup (box8), BatchNorm partial-sum slices 8 / 32 (sl8 / sl32; product 16)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4s26; rm -rf $O; mkdir -p $O; cd $R
run() { if [ "$1" = product ]; then env "${@:2}"; else env REPMODE_LIB=$R/variants/$1/librepmode_hip.so REPMODE_TORCH_LIB=$R/variants/$1/librepmode_torch.so "${@:2}"; fi; }
for rep in 1 2; do for v in product box8 sl8 sl32; do
  run $v timeout 300 python bench.py --no-cpu-baseline --no-fwd --steps 60 --warmup 20 > $O/b_${v}_$rep.json 2>> $O/err.txt
  python -c "
import json; d=json.load(open('$O/b_${v}_$rep.json')); print('$v', round(d['ms_per_step'],3), 'ms/step', d['config'].get('final_loss'))"
done; done
