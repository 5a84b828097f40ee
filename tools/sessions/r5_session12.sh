#!/bin/bash
# round 5, session 12: bench.py's N > 1 branch on a one-GPU box (two ranks on cuda:0 over gloo -- a code-path record, never a
# measurement): the line with config.comm as the rule leaves it, and with the rule made to take its bf16 branch
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5s12; mkdir -p $O
export REPMODE_BENCH_SHARE_GPU=1 HSA_ENABLE_IPC_MODE_LEGACY=0
run() { timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $1 bench.py --gpus 2 \
        --steps 10 --warmup 2 --batch 4 --no-cpu-baseline --no-fwd 2>$O/err_$2.txt | grep '^{' > $O/line_$2.json; }
run 29611 rule
REPMODE_COMPRESS_IF_RING_OVER=0 REPMODE_GRAD_RULE_BACKENDS=nccl,gloo run 29612 bf16_branch
python - <<'PY'
import json
for k in ('rule', 'bf16_branch'):
    d = json.load(open('gpurun_out/r5s12/line_%s.json' % k))
    print(k, d['n_gpus'], round(d['ms_per_step'], 2), json.dumps(d['config']['comm']), d['config']['setup_steps_before_warmup'])
PY
