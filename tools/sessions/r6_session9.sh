# round 6, session 9: the thin filter gradient's pipelined loop -- parity, per-launch time (1 -> 32 and 32 -> 1 at 32 x 64 x 64, batch 8)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r6s9; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_round6.py tests/test_hip_parity.py -x -q -k "thin or wgrad_kernel" 2>&1 | tail -3
python - <<'PY' | tee $O/thin_wgrad.txt
import torch, sys
sys.path.insert(0, '.')
from repmode_amd import ops
dev = 'cuda:0'
plan = ops.TaskPlan(torch.arange(8) % 12, 12, dev, training=True)
for cin, cout in ((1, 32), (32, 1)):
    x = torch.randn(8, 32, 64, 64, cin, device=dev).bfloat16()
    dy = torch.randn(8, 32, 64, 64, cout, device=dev).bfloat16()
    for _ in range(50): ops.conv5_wgrad(x, dy, plan, cout)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(300): ops.conv5_wgrad(x, dy, plan, cout)
    e1.record(); torch.cuda.synchronize()
    print('conv5_wgrad_thin %d -> %d: %.1f us per call (incl. the memset)' % (cin, cout, e0.elapsed_time(e1) / 300 * 1e3))
PY
