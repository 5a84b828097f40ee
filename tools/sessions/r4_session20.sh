# whole step: working tree against variants/base (a build of an earlier commit), interleaved
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4s20; rm -rf $O; mkdir -p $O; cd $R
B=${BASE_VARIANT:-k2old}
for rep in 1 2 3; do for v in $B product; do
  if [ $v = product ]; then E=""; else E="REPMODE_LIB=$R/variants/$v/librepmode_hip.so REPMODE_TORCH_LIB=$R/variants/$v/librepmode_torch.so"; fi
  env $E timeout 300 python bench.py --no-cpu-baseline --no-fwd --steps 60 --warmup 20 > $O/b_${v}_$rep.json 2>> $O/err.txt
  python -c "
import json; d=json.load(open('$O/b_${v}_$rep.json')); print('$v', round(d['ms_per_step'],3), 'ms/step', d['config'].get('final_loss'))"
done; done
