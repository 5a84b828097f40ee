#!/bin/bash
# Per-kernel register / LDS / spill summary of one .hip source (cross-compiles here, no GPU needed):
#   bash tools/kernel_resources.sh repmode_amd/csrc/conv5_igemm.hip [extra hipcc flags]
SRC=$1; shift
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -I$ROOT/include -I$ROOT/repmode_amd/csrc "$@" \
  -c "$SRC" -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys, re, subprocess
cur = None; rows = []
for line in sys.stdin:
    m = re.search(r'remark: +(Function Name|[A-Za-z ]+): *(.*?) \[-Rpass', line)
    if not m: continue
    k, v = m.group(1).strip(), m.group(2).strip()
    if k == 'Function Name':
        cur = {'name': v}; rows.append(cur)
    elif cur is not None:
        cur[k] = v
for r in rows:
    try: name = subprocess.run(['c++filt', r['name']], capture_output=True, text=True).stdout.strip()
    except Exception: name = r['name']
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    print('%-110s vgpr %-4s agpr %-4s sgpr %-4s spill %-3s lds %-6s occ %s' % (name[:110], r.get('VGPRs'), r.get('AGPRs'), r.get('TotalSGPRs', r.get('SGPRs')), r.get('VGPRs Spill', r.get('VGPR Spill','?')), r.get('LDS Size [bytes/block]', '?'), r.get('Occupancy [waves/SIMD]', '?')))
"
