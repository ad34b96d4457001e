// LDS accumulate microbenchmarks (which LDS update form is fast on gfx950?)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x;
}
constexpr int SLICE = 32768;
// MODE 0: ds_add_f32 random; 1: ds_add_f32 conflict-free (lane-linear); 2: ds_add_u32 random; 3: ds_add_u64 random (SLICE/2 slots)
// 4: non-atomic RMW random inside a wave-private region; 5: non-atomic RMW, 4 groups in flight; 6: ds_add_rtn_f32 random
// 7: ds_add_f32 random but each wave inside its private region
template <int MODE>
__global__ void k_lds(int iters, float* out) {
  extern __shared__ float s[];
  for (int i = threadIdx.x; i < SLICE; i += blockDim.x) s[i] = 0.f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int region = SLICE / nw;            // wave-private region (power of two for nw in {4,8,16})
  uint32_t st = mix32(blockIdx.x * 1024u + threadIdx.x + 1u);
  float racc = 0.f;
  for (int it = 0; it < iters; ++it) {
    st = st * 1664525u + 1013904223u;
    uint32_t r = st >> 8;
    if (MODE == 0) atomicAdd(&s[r % SLICE], 1.0f);
    else if (MODE == 1) atomicAdd(&s[((r & ~63u) + lane) % SLICE], 1.0f);
    else if (MODE == 2) atomicAdd(reinterpret_cast<unsigned*>(s) + (r % SLICE), 1u);
    else if (MODE == 3) atomicAdd(reinterpret_cast<unsigned long long*>(s) + (r % (SLICE / 2)), 1ull);
    else if (MODE == 4) { float* p = &s[wave * region + (r % region)]; *p = *p + 1.0f; }
    else if (MODE == 5) {
      float* p0 = &s[wave * region + (r % region)];
      float* p1 = &s[wave * region + ((r >> 3) % region)];
      float* p2 = &s[wave * region + ((r >> 5) % region)];
      float* p3 = &s[wave * region + ((r >> 7) % region)];
      float a = *p0, b = *p1, c = *p2, d = *p3;
      *p0 = a + 1.f; *p1 = b + 1.f; *p2 = c + 1.f; *p3 = d + 1.f;
    }
    else if (MODE == 6) racc += atomicAdd(&s[r % SLICE], 1.0f);
    else if (MODE == 7) atomicAdd(&s[wave * region + (r % region)], 1.0f);
  }
  __syncthreads();
  if (s[threadIdx.x] == -1.f || racc == -1.f) out[0] = 1.f;
}
template <int MODE>
void run(const char* name, int upd_per_iter, hipEvent_t e0, hipEvent_t e1, float* o) {
  const int iters = 2048, REP = 5;
  CK(hipFuncSetAttribute((const void*)k_lds<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, SLICE * 4));
  for (int threads : {256, 1024}) {
    k_lds<MODE><<<256, threads, SLICE * 4>>>(iters, o);
    CK(hipEventRecord(e0)); for (int r = 0; r < REP; ++r) k_lds<MODE><<<256, threads, SLICE * 4>>>(iters, o);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= REP;
    double n = 256.0 * threads * iters * upd_per_iter;
    printf("%-34s threads=%4d: %.3f ms  %8.1f Gupd/s  (%.1f cyc/wave-instr/CU @2.4GHz)\n", name, threads, ms, n / ms / 1e6,
           2.4e9 * (ms * 1e-3) / (iters * upd_per_iter * (threads / 64.0)));
  }
}
int main() {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float* o; CK(hipMalloc(&o, 4));
  run<0>("ds_add_f32 random", 1, e0, e1, o);
  run<1>("ds_add_f32 lane-linear", 1, e0, e1, o);
  run<2>("ds_add_u32 random", 1, e0, e1, o);
  run<3>("ds_add_u64 random", 1, e0, e1, o);
  run<4>("rmw f32 wave-private random", 1, e0, e1, o);
  run<5>("rmw f32 wave-private x4 in flight", 4, e0, e1, o);
  run<6>("ds_add_rtn_f32 random", 1, e0, e1, o);
  run<7>("ds_add_f32 wave-private random", 1, e0, e1, o);
  printf("done\n");
  return 0;
}
