// LDS microbenchmarks behind the write-combining append of the binned route's pass B (be_csr_binned.hip):
// returning / non-returning ds_add_u32 over few addresses (bins), sub-dword stores, and the reserve-write-commit sequence.
// Build: hipcc --offload-arch=gfx950 -O3 ubench5.hip -o ubench5
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x;
}
constexpr int WORDS = 36 * 1024;   // 144 KB
// MODE 0: ds_add_rtn_u32 on ctr[r % nb]          1: ds_add_u32 (no return) on ctr[r % nb]
//      2: ds_write_b16 random in WORDS*2 halves  3: ds_write_b32 random
//      4: reserve (rtn) + b16 + b32 stores into block r % nb at slot (ret % 32) + commit (rtn)
//      5: as 4 but 4 independent entries per lane and iteration
//      6: ds_add_rtn_u32 with stride-48 addresses (ctr[(r % nb) * 48]) : the bank pattern of block-embedded counters
template <int MODE>
__global__ void __launch_bounds__(1024) k_lds(int iters, int nb, uint32_t* out) {
  extern __shared__ uint32_t s[];
  for (int i = threadIdx.x; i < WORDS; i += blockDim.x) s[i] = 0u;
  __syncthreads();
  uint32_t st = mix32(blockIdx.x * 1024u + threadIdx.x + 1u);
  uint32_t racc = 0;
  uint32_t* ctr = s;                 // [nb] reserve
  uint32_t* done = s + 2048;         // [nb] commit
  uint32_t* buf = s + 4096;          // [nb][48]
  for (int it = 0; it < iters; ++it) {
    st = st * 1664525u + 1013904223u;
    const uint32_t r = st >> 8;
    if (MODE == 0) racc += atomicAdd(&ctr[r % nb], 1u);
    else if (MODE == 1) atomicAdd(&ctr[r % nb], 1u);
    else if (MODE == 2) reinterpret_cast<uint16_t*>(s)[r % (WORDS * 2)] = (uint16_t)r;
    else if (MODE == 3) s[r % WORDS] = r;
    else if (MODE == 4) {
      const uint32_t b = r % nb;
      const uint32_t slot = atomicAdd(&ctr[b], 1u) & 31u;
      reinterpret_cast<uint16_t*>(buf + b * 48 + 32)[slot] = (uint16_t)r;
      buf[b * 48 + slot] = r;
      __atomic_signal_fence(__ATOMIC_SEQ_CST);
      racc += atomicAdd(&done[b], 1u);
    } else if (MODE == 5) {
      uint32_t b[4], slot[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { b[u] = (r >> (2 * u)) % nb; slot[u] = atomicAdd(&ctr[b[u]], 1u) & 31u; }
#pragma unroll
      for (int u = 0; u < 4; ++u) { reinterpret_cast<uint16_t*>(buf + b[u] * 48 + 32)[slot[u]] = (uint16_t)r; buf[b[u] * 48 + slot[u]] = r; }
      __atomic_signal_fence(__ATOMIC_SEQ_CST);
#pragma unroll
      for (int u = 0; u < 4; ++u) racc += atomicAdd(&done[b[u]], 1u);
    } else if (MODE == 6) racc += atomicAdd(&buf[(r % nb) * 48], 1u);
  }
  __syncthreads();
  if (s[threadIdx.x] == 0xdeadbeefu || racc == 0xdeadbeefu) out[0] = 1u;
}
template <int MODE>
void run(const char* name, int nb, int upd_per_iter, hipEvent_t e0, hipEvent_t e1, uint32_t* o) {
  const int iters = 1024, REP = 5;
  CK(hipFuncSetAttribute((const void*)k_lds<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, WORDS * 4));
  for (int threads : {256, 1024}) {
    k_lds<MODE><<<256, threads, WORDS * 4>>>(iters, nb, o);
    CK(hipEventRecord(e0)); for (int r = 0; r < REP; ++r) k_lds<MODE><<<256, threads, WORDS * 4>>>(iters, nb, o);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= REP;
    double n = 256.0 * threads * iters * upd_per_iter;
    printf("%-40s nb=%5d threads=%4d: %.3f ms  %8.1f G/s  (%.1f cyc per wave-wide entry-group per CU @2.4GHz)\n", name, nb, threads, ms,
           n / ms / 1e6, 2.4e9 * (ms * 1e-3) / (iters * upd_per_iter * (threads / 64.0)));
  }
}
int main() {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  uint32_t* o; CK(hipMalloc(&o, 4));
  for (int nb : {2048, 611, 306, 77}) {
    run<0>("ds_add_rtn_u32 ctr[r % nb]", nb, 1, e0, e1, o);
    run<1>("ds_add_u32 ctr[r % nb]", nb, 1, e0, e1, o);
    run<6>("ds_add_rtn_u32 stride-48 ctr", nb > 611 ? 611 : nb, 1, e0, e1, o);
    run<4>("reserve + b16 + b32 + commit", nb > 611 ? 611 : nb, 1, e0, e1, o);
    run<5>("reserve + b16 + b32 + commit, x4", nb > 611 ? 611 : nb, 4, e0, e1, o);
  }
  run<2>("ds_write_b16 random", 0, 1, e0, e1, o);
  run<3>("ds_write_b32 random", 0, 1, e0, e1, o);
  printf("done\n");
  return 0;
}
