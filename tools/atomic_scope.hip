// atomic_scope.hip -- what the split-K epilogues of the deep levels pay: 512 workgroups, each adding its 256 x 64 float tile
// (64 KB, 128-byte contiguous pieces per half-wave, conv5_igemm's SWAP epilogue pattern) into a 4 MB output shared by 8
// K-slices, with (a) agent-scope float atomics (what unsafeAtomicAdd emits), (b) workgroup-scope atomics (performed in the
// XCD's own L2: only correct when every slice of an output tile runs on ONE XCD -- here slice-major placement, wrong results
// expected, timing only), (c) plain stores of the same bytes into per-slice partial buffers, (d) those partials read back
// and summed by a second kernel.     hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/atomic_scope.hip -o /tmp/atomic_scope
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int TILE_V = 256, TILE_C = 64, COUT = 256, NVOX = 2048, KS = 8;

template <int MODE>
__global__ __launch_bounds__(256) void add_tiles(float* y, float* part, int same_xcd) {
  // workgroup -> (tile, slice): slice fastest (a tile's slices on consecutive workgroups = different XCDs), or
  // same_xcd: tile t's slices all on XCD t % 8
  const int b = blockIdx.x;
  int tile, kz;
  if (same_xcd) { const int xcd = b & 7, j = b >> 3; tile = (j / KS) * 8 + xcd; kz = j % KS; }
  else { tile = b / KS; kz = b % KS; }
  const int ntc = COUT / TILE_C;
  const int v0 = (tile / ntc) * TILE_V, c0 = (tile % ntc) * TILE_C;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l31 = lane & 31, khalf = lane >> 5;
  const int wv = wave & 1, wc = wave >> 1;
  float acc = (float)(b + 1);
#pragma unroll
  for (int vs = 0; vs < 4; ++vs)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = (wv * 4 + vs) * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
      float* p = y + (size_t)(v0 + m) * COUT + c0 + wc * 32 + l31;
      if (MODE == 0) unsafeAtomicAdd(p, acc);
      else if (MODE == 1) __hip_atomic_fetch_add(p, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      else part[(size_t)kz * NVOX * COUT + (size_t)(v0 + m) * COUT + c0 + wc * 32 + l31] = acc;
    }
}

__global__ void sum_parts(const float4* part, float4* y, int n4) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  float4 s = part[i];
  for (int k = 1; k < KS; ++k) { const float4 t = part[(size_t)k * n4 + i]; s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w; }
  y[i] = s;
}

int main() {
  const size_t n = (size_t)NVOX * COUT;
  float *y, *part;
  hipMalloc(&y, n * 4);
  hipMalloc(&part, n * 4 * KS);
  hipMemset(y, 0, n * 4);
  const int grid = (NVOX / TILE_V) * (COUT / TILE_C) * KS;   // 8 x 4 x 8 = 256 tiles-slices ... x2 experts below
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](const char* name, auto launch) {
    for (int i = 0; i < 20; ++i) launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 200; ++i) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %7.2f us per launch  (%.2f TB/s of added floats)\n", name, ms * 5.0, (double)grid * TILE_V * TILE_C * 4 / (ms * 5e-6) / 1e12);
  };
  printf("%d workgroups x 64 KB into a %zu MB output, %d slices per tile\n", grid, n * 4 >> 20, KS);
  run("agent-scope atomics, slices across XCDs", [&] { hipLaunchKernelGGL(add_tiles<0>, dim3(grid), dim3(256), 0, 0, y, part, 0); });
  run("agent-scope atomics, a tile's slices on 1 XCD", [&] { hipLaunchKernelGGL(add_tiles<0>, dim3(grid), dim3(256), 0, 0, y, part, 1); });
  run("workgroup-scope atomics, a tile on 1 XCD", [&] { hipLaunchKernelGGL(add_tiles<1>, dim3(grid), dim3(256), 0, 0, y, part, 1); });
  run("plain stores to per-slice partials", [&] { hipLaunchKernelGGL(add_tiles<2>, dim3(grid), dim3(256), 0, 0, y, part, 0); });
  run("partials + summing kernel", [&] {
    hipLaunchKernelGGL(add_tiles<2>, dim3(grid), dim3(256), 0, 0, y, part, 0);
    hipLaunchKernelGGL(sum_parts, dim3((unsigned)(n / 4 + 255) / 256), dim3(256), 0, 0, (const float4*)part, (float4*)y, (int)(n / 4));
  });
  // is the workgroup-scope result right when a tile's slices share an XCD?  (expected sum per element: sum over its slices of b + 1)
  hipMemset(y, 0, n * 4);
  hipLaunchKernelGGL(add_tiles<1>, dim3(grid), dim3(256), 0, 0, y, part, 1);
  hipDeviceSynchronize();
  std::vector<float> h(n);
  hipMemcpy(h.data(), y, n * 4, hipMemcpyDeviceToHost);
  long bad = 0;
  for (int tile = 0; tile < grid / KS; ++tile) {
    double want = 0;
    for (int kz = 0; kz < KS; ++kz) { const int xcd = tile & 7, j = (tile >> 3) * KS + kz; want += (double)(j * 8 + xcd + 1); }
    const int ntc = COUT / TILE_C, v0 = (tile / ntc) * TILE_V, c0 = (tile % ntc) * TILE_C;
    for (int v = 0; v < TILE_V; ++v) for (int c = 0; c < TILE_C; ++c) if (h[(size_t)(v0 + v) * COUT + c0 + c] != (float)want) ++bad;
  }
  printf("workgroup-scope atomics with same-XCD placement: %ld wrong elements of %zu\n", bad, n);
  return 0;
}
