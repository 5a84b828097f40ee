// store_pattern.hip -- how fast do scattered contiguous pieces of P bytes (at a large stride) stream to HBM?  The GatRep forward
// writes its merged filters as 128-byte pieces, eight neighbouring workgroups completing a 1 KiB tile; this probe writes
// `total` bytes as pieces of P = 128 .. 4096 bytes, a workgroup per piece sequence, to price a rewrite with larger pieces.
//   hipcc --offload-arch=gfx950 -O3 tools/store_pattern.hip -o /tmp/store_pattern && /tmp/store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

// workgroup b writes, for j = 0 .. npieces-1, the piece at byte offset ((j * ngroups + b) * P): pieces of one workgroup are
// ngroups * P apart (the tap stride of the filter tensor), neighbouring workgroups' pieces are adjacent.
__global__ __launch_bounds__(256) void k_store(unsigned* out, int P, int npieces, int ngroups, int per_store) {
  const int lanes_per_piece = P / per_store;                     // threads that cover one piece with one store each
  const int pieces_per_iter = 256 / lanes_per_piece;
  const int t = threadIdx.x % lanes_per_piece, pi = threadIdx.x / lanes_per_piece;
  for (int j = pi; j < npieces; j += pieces_per_iter) {
    char* p = reinterpret_cast<char*>(out) + ((size_t)j * ngroups + blockIdx.x) * P + (size_t)t * per_store;
    if (per_store == 4) *reinterpret_cast<unsigned*>(p) = j;
    else *reinterpret_cast<uint4*>(p) = uint4{(unsigned)j, 1u, 2u, 3u};
  }
}

int main() {
  const size_t total = (size_t)512 << 20;
  unsigned* buf;
  hipMalloc(&buf, total);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int per_store : {4, 16})
    for (int P : {128, 256, 512, 1024, 4096}) {
      if (P / per_store > 256) continue;
      const int ngroups = 4096;                                  // workgroups
      const int npieces = (int)(total / ((size_t)ngroups * P));
      for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k_store, dim3(ngroups), dim3(256), 0, 0, buf, P, npieces, ngroups, per_store);
      hipEventRecord(e0);
      for (int w = 0; w < 10; ++w) hipLaunchKernelGGL(k_store, dim3(ngroups), dim3(256), 0, 0, buf, P, npieces, ngroups, per_store);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      printf("pieces of %4d B, %2d-byte stores: %.2f TB/s\n", P, per_store, total * 10.0 / ms / 1e9);
    }
  return 0;
}
