set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2c; rm -rf $O; mkdir -p $O
cd $R
python tools/host_profile.py > $O/host_profile.txt 2>&1; head -40 $O/host_profile.txt
python - > $O/reducer_dbg.txt 2>&1 <<'PY'
import sys, torch
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
from conftest import Opts
from repmode_amd.model import Model
g = torch.Generator().manual_seed(5)
x = torch.randn(4, 1, 16, 64, 64, generator=g); t = torch.randn(4, 1, 16, 64, 64, generator=g)
tasks = torch.tensor([1, 4, 9, 4])
res = []
for distributed in ('reducer', False, False, 'reducer'):
    torch.manual_seed(0)
    m = Model(Opts(), lr=1e-4, gpu_ids=0, mult_chan=4, dtype=torch.float32, distributed=distributed)
    m.do_train_iter(x, t, tasks)
    res.append({k: p.grad.detach().cpu().clone() for k, p in m.net.named_parameters()})
gmax = max(float(v.abs().max()) for v in res[1].values())
for a, b, nm in ((0, 1, 'reducer vs plain'), (2, 1, 'plain vs plain'), (3, 0, 'reducer vs reducer')):
    errs = sorted(((float((res[a][k] - res[b][k]).abs().max()) / max(float(res[b][k].abs().max()), 1e-2 * gmax), k) for k in res[b]), reverse=True)
    print(nm, errs[:6])
PY
cat $O/reducer_dbg.txt | tail -8
