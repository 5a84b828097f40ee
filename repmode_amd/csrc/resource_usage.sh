#!/bin/bash
# Print per-kernel register / LDS / spill figures for one .hip file (from the code-object metadata).
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
f="$1"; tmp=$(mktemp -d)
( cd "$tmp" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -I"$HERE/../../include" -I"$HERE" ${REPMODE_EXTRA_FLAGS:-} -save-temps -c "$HERE/$f" -o x.o 2>/dev/null )
s=$(ls "$tmp"/*gfx950*.s | head -1)
python3 - "$s" <<'PY'
import re, sys, subprocess
txt = open(sys.argv[1]).read()
meta = txt[txt.find('amdhsa.kernels'):]
for blk in re.split(r'\n  - \.agpr_count', meta)[1:]:
    blk = '.agpr_count' + blk
    get = lambda k: (re.search(r'\.%s:\s+(\S+)' % k, blk) or [None, '?'])[1]
    name = get('name')
    dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    dem = re.sub(r'\(anonymous namespace\)::', '', dem)[:100]
    print('%-100s vgpr %s agpr %s sgpr %s spill v%s s%s lds %s scratch %s' % (dem, get('vgpr_count'), get('agpr_count'), get('sgpr_count'), get('vgpr_spill_count'), get('sgpr_spill_count'), get('group_segment_fixed_size'), get('private_segment_fixed_size')))
PY
mkdir -p "$HERE/build"; cp "$s" "$HERE/build/$(basename "${f%.hip}").s"
rm -rf "$tmp"
