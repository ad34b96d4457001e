// Random-block read microbenchmark: how fast can the chip read aligned blocks of B bytes at random positions of a
// buffer of G bytes?  (What the planned scatter does: ~1 KB blocks of the active rows out of a 65 GB layout.)
// Each wave reads one block per step as 64 lanes x 16 B x (B / 1024) loads, 8 blocks in flight per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x;
}
// LOADS = 16-B loads per lane per block (block bytes = LOADS * 1024); `group` consecutive waves read neighbouring
// blocks of the same 64 KB region when group > 1 (what the row-major layout gives the slices of one row)
template <int LOADS>
__global__ void __launch_bounds__(1024) k_rand_blocks(const uint4* __restrict__ buf, uint64_t n_blocks, int steps, int group,
                                                      uint32_t* __restrict__ sink) {
  const int lane = threadIdx.x & 63;
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  uint4 acc = make_uint4(0, 0, 0, 0);
  for (int s = 0; s < steps; s += 8) {
    uint4 v[8][LOADS];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      uint64_t b;
      if (group == -1 || group == -2) {
        // the planned kernel's pattern: workgroup = (slice, part) with 64 slices x 4 parts; wave w of a workgroup walks
        // its own rows.  -1: row-major layout (the 64 slices' blocks of one row are adjacent: 64 workgroups read one
        // 64 KB window at about the same time); -2: slice-major layout (every slice has its own contiguous region)
        const uint32_t slice = blockIdx.x & 63, part = blockIdx.x >> 6, wv = threadIdx.x >> 6;
        const uint64_t row = ((uint64_t)mix32((part * 16u + wv) * 7919u + (uint32_t)(s + u) * 104729u + 1u) * 2654435761ull) % (n_blocks / 64);
        b = (group == -1) ? row * 64 + slice : (uint64_t)slice * (n_blocks / 64) + row;
      } else if (group > 1) {   // waves of one group share a random base region, each takes its own block in it
        const uint64_t region = ((uint64_t)mix32((wave / group) * 7919u + (uint32_t)(s + u) * 104729u + 1u) * 2654435761ull) % (n_blocks / group);
        b = region * group + (wave % group);
      } else {
        b = ((uint64_t)mix32(wave * 7919u + (uint32_t)(s + u) * 104729u + 1u) * 2654435761ull) % n_blocks;
      }
      const uint4* p = buf + b * (LOADS * 64) + lane;
#pragma unroll
      for (int l = 0; l < LOADS; ++l) v[u][l] = p[l * 64];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int l = 0; l < LOADS; ++l) { acc.x ^= v[u][l].x; acc.y ^= v[u][l].y; acc.z ^= v[u][l].z; acc.w ^= v[u][l].w; }
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}

template <int LOADS>
void run(const uint4* buf, uint64_t gbytes, int group, uint32_t* sink) {
  const uint64_t n_blocks = gbytes / (LOADS * 1024ull);
  const int grid = 256, steps = 4096 / LOADS;     // 4096 waves x steps blocks
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_rand_blocks<LOADS>, dim3(grid), dim3(1024), 0, 0, buf, n_blocks, steps, group, sink);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  }
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double bytes = 4096.0 * steps * LOADS * 1024.0;
  printf("random blocks of %5d B, buffer %6.1f GB, group %2d: %.3f ms  %.0f GB/s\n", LOADS * 1024, gbytes / 1e9, group, ms,
         bytes / ms / 1e6);
  fflush(stdout);
}

int main() {
  const uint64_t big = 64ull << 30;
  uint4* buf; uint32_t* sink;
  CK(hipMalloc(&buf, big)); CK(hipMalloc(&sink, 4));
  CK(hipMemset(buf, 1, big));
  for (uint64_t g : {1ull << 30, 8ull << 30, 64ull << 30}) {
    run<1>(buf, g, 1, sink);
    run<2>(buf, g, 1, sink);
    run<4>(buf, g, 1, sink);
    run<1>(buf, g, 64, sink);
  }
  printf("planned-kernel pattern (64 slices x 4 parts), group -1 = row-major blocks, -2 = slice-major blocks\n");
  run<1>(buf, big, -1, sink);
  run<1>(buf, big, -2, sink);
  run<1>(buf, big, -1, sink);
  run<1>(buf, big, -2, sink);
  return 0;
}
