// Random-line read microbenchmark: how many aligned 128-byte (or 256-byte) lines per second can the chip fetch from random
// positions of a multi-GB buffer?  (What the planned scatter does when a (row, slice) block is a line or two: N = 1M neurons
// with 1000 synapses each.)  Every 16-byte load instruction of a wave touches 64 / LPL random lines (LPL lanes per line), and a
// wave keeps DEPTH such instructions in flight.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x;
}
template <int LPL, int DEPTH>
__global__ void __launch_bounds__(1024) k_rand_lines(const uint4* __restrict__ buf, uint64_t n_lines, int steps, uint32_t* __restrict__ sink) {
  const int lane = threadIdx.x & 63;
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint32_t grp = lane / LPL, l = lane % LPL;               // line group within the wave, lane within the line
  uint4 acc = make_uint4(0, 0, 0, 0);
  for (int s = 0; s < steps; s += DEPTH) {
    uint4 v[DEPTH];
#pragma unroll
    for (int u = 0; u < DEPTH; ++u) {
      const uint64_t line = ((uint64_t)mix32((wave * (64 / LPL) + grp) * 7919u + (uint32_t)(s + u) * 104729u + 1u) * 2654435761ull) % n_lines;
      v[u] = buf[line * LPL + l];                                  // LPL lanes x 16 B = one line of LPL * 16 bytes
    }
#pragma unroll
    for (int u = 0; u < DEPTH; ++u) { acc.x ^= v[u].x; acc.y ^= v[u].y; acc.z ^= v[u].z; acc.w ^= v[u].w; }
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}
template <int LPL, int DEPTH>
void run(const uint4* buf, uint64_t bytes, int grid, uint32_t* sink) {
  const uint64_t n_lines = bytes / (LPL * 16ull);
  const int steps = 256;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float ms = 0;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_rand_lines<LPL, DEPTH>), dim3(grid), dim3(1024), 0, 0, buf, n_lines, steps, sink);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
  }
  const double lines = (double)grid * 16 * steps * (64 / LPL);
  printf("random lines of %4d B, buffer %6.1f GB, %2d in flight per wave x %d lines, %4d workgroups: %.3f ms  %.1f G lines/s  %.0f GB/s\n",
         LPL * 16, bytes / 1e9, DEPTH, 64 / LPL, grid, ms, lines / ms / 1e6, lines * LPL * 16 / ms / 1e6);
}
int main() {
  uint32_t* sink; CK(hipMalloc(&sink, 4));
  for (uint64_t gb : {1ull, 6ull, 60ull}) {
    const uint64_t bytes = gb << 30;
    uint4* buf; CK(hipMalloc(&buf, bytes)); CK(hipMemset(buf, 1, bytes));
    run<8, 4>(buf, bytes, 256, sink);
    run<8, 8>(buf, bytes, 256, sink);
    run<8, 16>(buf, bytes, 256, sink);
    run<16, 8>(buf, bytes, 256, sink);
    run<16, 16>(buf, bytes, 256, sink);
    run<4, 8>(buf, bytes, 256, sink);
    CK(hipFree(buf));
  }
  return 0;
}
