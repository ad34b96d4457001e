// Microbenchmark 8 (round 6): can the JITC scatter's residue-class workgroups write the output themselves?
// Question: at C3 a workgroup owns the accumulators of one (chunk, lane residue) class — 31 250 positions q of
// out[chunk_start + l + 32 q] — and today stores them contiguously ([class][part][q], 126 KB per workgroup) for a transposing reduce
// kernel (13.3 us: 32 MB read, 16 MB written).  Writing `out` directly from the walk kernel's epilogue means 4-byte stores at a
// 128-byte stride: every wave store touches 64 different lines, and the 32 classes that share a line run in different workgroups
// (different XCDs, different times).  Measured here with the C3 geometry (4 chunks x 32 classes, 2 parts per class = 256 workgroups of
// 1024 threads, 4M outputs): the epilogue alone, no walk —
//   contiguous : each workgroup stores its 31 488 u32 contiguously (what k_jit_mv_scatter does now: 32 MB)
//   strided    : each workgroup of a class stores HALF of the class's positions at out[cs + l + 32 q] (16 MB of useful bytes; the
//                cross-part merge this would also need is not even modelled — the floor of the idea)
//   strided_nt : the same with non-temporal stores
// Build: hipcc --offload-arch=gfx950 -O3 ubench8.hip -o ubench8
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

constexpr int kClasses = 128, kParts = 2, kQ = 31250, kPiece = 31488, kStride = 32;
constexpr int64_t kChunk = 1000000;

__global__ void __launch_bounds__(1024) k_contig(uint32_t* __restrict__ partial, uint32_t v) {
  uint32_t* dst = partial + (int64_t)blockIdx.x * kPiece;
  for (int i = threadIdx.x; i < kPiece; i += 1024) dst[i] = v + i;
}

template <bool NT>
__global__ void __launch_bounds__(1024) k_strided(float* __restrict__ out, float v) {
  const int cls = blockIdx.x / kParts, part = blockIdx.x % kParts;
  const int chunk = cls / kStride, l = cls % kStride;
  float* dst = out + (int64_t)chunk * kChunk + l;
  const int q0 = part * (kQ / kParts), q1 = part == kParts - 1 ? kQ : q0 + kQ / kParts;
  for (int q = q0 + threadIdx.x; q < q1; q += 1024) {
    if (NT) __builtin_nontemporal_store(v + q, dst + (int64_t)kStride * q);
    else dst[(int64_t)kStride * q] = v + q;
  }
}

__global__ void __launch_bounds__(256) k_touch(const float* __restrict__ p, size_t n, float* sink) {   // evicts: reads 1 GiB
  float a = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) a += p[i];
  if (a == 1.2345f) sink[0] = a;
}

int main() {
  uint32_t* partial; float* out; float* big; float* sink;
  CK(hipMalloc(&partial, (size_t)kClasses * kParts * kPiece * 4));
  CK(hipMalloc(&out, (size_t)4 * kChunk * 4));
  CK(hipMalloc(&big, (size_t)1 << 30)); CK(hipMalloc(&sink, 64));
  CK(hipMemset(big, 0, (size_t)1 << 30));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int REP = 50;
  auto run = [&](const char* name, auto launch) {
    float tot = 0.f;
    for (int r = 0; r < REP + 5; ++r) {
      hipLaunchKernelGGL(k_touch, dim3(2048), dim3(256), 0, 0, big, ((size_t)1 << 30) / 4, sink);    // cold caches, as behind a 100-us walk
      CK(hipEventRecord(e0));
      launch();
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (r >= 5) tot += ms;
    }
    printf("%-12s %8.2f us per launch\n", name, tot / REP * 1e3);
  };
  run("contiguous", [&] { hipLaunchKernelGGL(k_contig, dim3(kClasses * kParts), dim3(1024), 0, 0, partial, 1u); });
  run("strided", [&] { hipLaunchKernelGGL(k_strided<false>, dim3(kClasses * kParts), dim3(1024), 0, 0, out, 1.f); });
  run("strided_nt", [&] { hipLaunchKernelGGL(k_strided<true>, dim3(kClasses * kParts), dim3(1024), 0, 0, out, 1.f); });
  return 0;
}
