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
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ "$HERE/common.h" -nt "$o" ] || [ "$ROOT/include/repmode_hip.h" -nt "$o" ]; then
    $HIPCC $FLAGS -c "$f" -o "$o" &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
# only the objects of sources that exist now (a stale .o of a removed / renamed file must not be linked)
$HIPCC --offload-arch=gfx950 -shared -fPIC "${objs[@]}" -o "$OUT"
echo "built $OUT"
