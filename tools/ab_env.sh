# same-box A/B of the whole train step between two settings of ONE environment switch, interleaved:
#   gpurun -- 'bash tools/ab_env.sh REPMODE_DUAL_WGRAD 0 1'
V=$1; A=$2; B=$3
cd $GRAFT_REPO_ROOT; O=gpurun_out/ab_$V; mkdir -p $O
for rep in 1 2; do for val in $A $B; do
  env $V=$val timeout 300 python bench.py --no-cpu-baseline --no-fwd --steps 60 --warmup 20 > $O/b_${val}_$rep.json 2>> $O/err.txt
  python -c "
import json; d=json.load(open('$O/b_${val}_$rep.json')); print('$V=$val', round(d['ms_per_step'],3), 'ms/step   conv5_igemm', round(d['roofline']['achieved'],1), 'TFLOP/s')"
done; done
