#!/bin/bash
# Build librepmode_hip.so for gfx950 (MI355X) in-tree.  hipcc cross-compiles without a GPU.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
OUT="${REPMODE_OUT:-$HERE/../librepmode_hip.so}"      # (REPMODE_OUT / REPMODE_BUILD_DIR: a variant build next to the product one)
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -I$ROOT/include -I$HERE ${REPMODE_EXTRA_FLAGS:-}"
BUILD="${REPMODE_BUILD_DIR:-$HERE/build}"
mkdir -p "$BUILD"
pids=()
objs=()
for f in "$HERE"/*.hip; do
  o="$BUILD/$(basename "${f%.hip}").o"
  objs+=("$o")
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ "$HERE/common.h" -nt "$o" ] || [ "$HERE/tail_jobs.h" -nt "$o" ] || [ "$HERE/box_body.h" -nt "$o" ] || [ "$HERE/wgrad_col.h" -nt "$o" ] || [ "$ROOT/include/repmode_hip.h" -nt "$o" ]; then
    $HIPCC $FLAGS -c "$f" -o "$o" &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
# only the objects of sources that exist now (a stale .o of a removed / renamed file must not be linked)
$HIPCC --offload-arch=gfx950 -shared -fPIC "${objs[@]}" -o "$OUT"
echo "built $OUT"

# ---- the operator seam: TORCH_LIBRARY ops + C++ autograd over the C ABI (host code only: g++, no device code)
if [ "${REPMODE_SKIP_TORCH_OPS:-0}" != "1" ]; then
  TOUT="${REPMODE_TORCH_OUT:-$(dirname "$OUT")/librepmode_torch.so}"
  TSRC="$HERE/torch/repmode_ops.cpp"
  if [ ! -f "$TOUT" ] || [ "$TSRC" -nt "$TOUT" ] || [ "$ROOT/include/repmode_hip.h" -nt "$TOUT" ] || [ "$OUT" -nt "$TOUT" ]; then
    TDIR="$(python3 -c 'import os, torch; print(os.path.dirname(torch.__file__))')"
    ABI="$(python3 -c 'import torch; print(int(torch._C._GLIBCXX_USE_CXX11_ABI))')"
    g++ -O2 -std=c++17 -fPIC -shared -D__HIP_PLATFORM_AMD__ -DUSE_ROCM -D_GLIBCXX_USE_CXX11_ABI=$ABI \
        -I"$TDIR/include" -I"$TDIR/include/torch/csrc/api/include" -I/opt/rocm/include -I"$ROOT/include" \
        "$TSRC" -o "$TOUT" \
        -L"$TDIR/lib" -ltorch -ltorch_cpu -lc10 -lc10_hip -ltorch_hip \
        -L"$(dirname "$OUT")" -l:"$(basename "$OUT")" -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,'$ORIGIN'
  fi
  echo "built $TOUT"
fi
