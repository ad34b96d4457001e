// Microbenchmark 6 (round 4): can a scatter accumulate in the XCD's L2 instead of LDS?
// Question: default (agent-scope) global atomics execute at the memory side on this chip (~21-27 G/s).  If every
// adder of an address runs on ONE XCD, a narrower scope may let that XCD's L2 execute the add in cache.  The kernel reads its
// XCC id from the hardware register, hits random addresses of its own eighth of the table, and adds with the requested scope
// (workgroup / agent) and type (u32, u64, f32).  Each launch is verified (sum of the table == number of adds for integers) —
// a fast rate that loses updates is no rate.
// Build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics ubench6.hip -o ubench6
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

__device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x;
}
__device__ __forceinline__ uint32_t xcc_id() {       // HW_REG_XCC_ID = 20, bits [3:0]
  return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u;
}

// SCOPE: 0 = workgroup, 1 = agent.  LOCAL: 1 = own eighth of the table only, 0 = whole table.
template <typename T, int SCOPE, int LOCAL>
__global__ void __launch_bounds__(256) k_adds(T* table, uint32_t per_xcc, int iters, uint32_t* xcc_hist) {
  const uint32_t x = xcc_id();
  if (threadIdx.x == 0) atomicAdd(xcc_hist + x, 1u);
  T* base = LOCAL ? table + (size_t)x * per_xcc : table;
  const uint32_t range = LOCAL ? per_xcc : per_xcc * 8u;
  uint32_t s = mix32(blockIdx.x * 256u + threadIdx.x + 12345u);
  for (int i = 0; i < iters; ++i) {
    s = s * 1664525u + 1013904223u;
    const uint32_t j = (uint32_t)(((uint64_t)mix32(s) * range) >> 32);
    if (SCOPE == 0) __hip_atomic_fetch_add(base + j, (T)1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else __hip_atomic_fetch_add(base + j, (T)1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

template <typename T>
__global__ void k_sum(const T* t, size_t n, double* out) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  double a = 0;
  for (; i < n; i += (size_t)gridDim.x * blockDim.x) a += (double)t[i];
  atomicAdd(out, a);
}

template <typename T, int SCOPE, int LOCAL>
void run(const char* name, size_t table_bytes, int wgs, int iters) {
  const uint32_t per_xcc = (uint32_t)(table_bytes / sizeof(T) / 8);
  T* table; uint32_t* hist; double* total;
  CK(hipMalloc(&table, (size_t)per_xcc * 8 * sizeof(T))); CK(hipMalloc(&hist, 64)); CK(hipMalloc(&total, 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f; double sum = 0; uint32_t h[8];
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipMemset(table, 0, (size_t)per_xcc * 8 * sizeof(T))); CK(hipMemset(hist, 0, 64)); CK(hipMemset(total, 0, 8));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    k_adds<T, SCOPE, LOCAL><<<wgs, 256>>>(table, per_xcc, iters, hist);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    k_sum<T><<<1024, 256>>>(table, (size_t)per_xcc * 8, total);
    CK(hipMemcpy(&sum, total, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(h, hist, 32, hipMemcpyDeviceToHost));
  }
  const double adds = (double)wgs * 256 * iters;
  printf("%-28s table %7.2f MB  %6.1f G adds/s  (%.3f ms)  sum/adds = %.6f  wgs per xcc %u %u %u %u %u %u %u %u\n", name,
         table_bytes / 1048576.0, adds / best / 1e6, best, sum / adds, h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
  fflush(stdout);
  CK(hipFree(table)); CK(hipFree(hist)); CK(hipFree(total));
}

int main() {
  const int wgs = 256 * 8, iters = 512;           // 2.7e8 adds per launch
  for (size_t mb : {1, 4, 16, 64}) {
    const size_t b = mb << 20;
    run<uint32_t, 1, 0>("u32 agent, whole table", b, wgs, iters);
    run<uint32_t, 1, 1>("u32 agent, own eighth", b, wgs, iters);
    run<uint32_t, 0, 1>("u32 workgroup, own eighth", b, wgs, iters);
    run<unsigned long long, 1, 1>("u64 agent, own eighth", b, wgs, iters);
    run<unsigned long long, 0, 1>("u64 workgroup, own eighth", b, wgs, iters);
    run<float, 1, 1>("f32 agent, own eighth", b, wgs, iters);
    run<float, 0, 1>("f32 workgroup, own eighth", b, wgs, iters);
  }
  return 0;
}
