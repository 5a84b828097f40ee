#!/bin/bash
# The package power cap as the ceiling of the MFMA kernels: the register-only MFMA loop of
# tools/mfma_peak.hip with near-constant and with random operands, then a loop of the level-0 convolution, each with rocm-smi's
# clock / power read every half second beside it.  Run on the GPU box (through gpurun); prints to stdout.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p tools/build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o tools/build/mfma_peak || exit 1
sample() {   # rocm-smi's shader clock and package power while $1 runs
  ( while kill -0 $1 2>/dev/null; do
      rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' ' | sed 's/  */ /g'; echo
      sleep 0.5
    done ) | awk 'NR % 2 == 1' | head -12
}
echo "== tools/mfma_peak.hip: v_mfma_f32_32x32x16_bf16 from registers only (4 accumulators per wave, 2 waves per SIMD), 400 launches per line"
tools/build/mfma_peak > /tmp/mfma_peak.out &
P=$!; sample $P; wait $P; cat /tmp/mfma_peak.out
echo "== the level-0 / level-1 convolution (conv5_ws_kernel) in a loop: tools/conv_microbench.py"
timeout 120 python tools/conv_microbench.py 32 32 32 64 64 20000 > /tmp/conv_loop.out 2>&1 &
P=$!; sample $P; wait $P; tail -12 /tmp/conv_loop.out
