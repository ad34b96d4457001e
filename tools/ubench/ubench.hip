// Microbenchmarks that size the scatter design on MI355X (gfx950):
//   1. streaming read bandwidth (16 B / lane)
//   2. random-address global f32 atomic add rate (the reference's scatter shape)
//   3. random-address LDS f32 atomic add rate (ds_add_f32)
//   4. u16-index + f32-weight stream from HBM accumulated into an LDS slice
// Build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics ubench.hip -o ubench
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

__global__ void k_fill_u32(uint32_t* p, size_t n, uint32_t seed, uint32_t mod) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) p[i] = mix32((uint32_t)i * 0x9e3779b9u + seed) % mod;
}
__global__ void k_fill_u16(uint16_t* p, size_t n, uint32_t seed, uint32_t mod) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) p[i] = (uint16_t)(mix32((uint32_t)i * 0x9e3779b9u + seed) % mod);
}
__global__ void k_fill_f32(float* p, size_t n, float v) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) p[i] = v;
}

// 1. stream read
__global__ void k_stream(const float4* __restrict__ p, size_t n4, float* out) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  float acc = 0.f;
  for (; i < n4; i += stride) { float4 v = p[i]; acc += v.x + v.y + v.z + v.w; }
  if (acc == 123.456f) out[0] = acc;
}

// 2. random global atomics: idx + weight streamed, atomicAdd to out[idx]
__global__ void k_gatomic(const uint32_t* __restrict__ idx, const float* __restrict__ w, size_t n, float* out) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) atomicAdd(out + idx[i], w[i]);
}
__global__ void k_gatomic_int(const uint32_t* __restrict__ idx, size_t n, int* out) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) atomicAdd(out + idx[i], 1);
}

// 3. LDS random atomics, no global traffic
template <int SLICE>
__global__ void k_lds_atomic(int iters, float* out) {
  extern __shared__ float s[];
  for (int i = threadIdx.x; i < SLICE; i += blockDim.x) s[i] = 0.f;
  __syncthreads();
  uint32_t st = mix32(blockIdx.x * 1024u + threadIdx.x + 1u);
  for (int it = 0; it < iters; ++it) {
    st = st * 1664525u + 1013904223u;
    atomicAdd(&s[(st >> 8) % SLICE], 1.0f);
  }
  __syncthreads();
  if (s[threadIdx.x] == -1.f) out[0] = 1.f;
}

// 4. slice-accumulate prototype: each block owns (slice, part); streams `seg` entries per active row
//    layout: idx16[slice][row][seg], w[slice][row][seg]  (fixed seg for the microbench)
template <int SLICE, int VEC>
__global__ void __launch_bounds__(1024)
k_slice_acc(const uint16_t* __restrict__ idx16, const float* __restrict__ w, const uint32_t* __restrict__ active,
            int n_active, int n_rows, int seg, int parts, float* __restrict__ partial) {
  extern __shared__ float s[];
  const int slice = blockIdx.x / parts, part = blockIdx.x % parts;
  for (int i = threadIdx.x; i < SLICE; i += blockDim.x) s[i] = 0.f;
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
  const size_t slice_base = (size_t)slice * n_rows * seg;
  // waves of this block take active rows round-robin within the block's part
  for (int a = part * nw + wave; a < n_active; a += parts * nw) {
    const size_t base = slice_base + (size_t)active[a] * seg;
    if (VEC == 1) {
      for (int j = lane; j < seg; j += 64) atomicAdd(&s[idx16[base + j]], w[base + j]);
    } else {  // 4 entries per lane: 8 B of idx + 16 B of weight
      const int seg4 = seg >> 2;
      const uint2* ip = reinterpret_cast<const uint2*>(idx16 + base);
      const float4* wp = reinterpret_cast<const float4*>(w + base);
      for (int j = lane; j < seg4; j += 64) {
        uint2 iv = ip[j]; float4 wv = wp[j];
        atomicAdd(&s[iv.x & 0xffff], wv.x); atomicAdd(&s[iv.x >> 16], wv.y);
        atomicAdd(&s[iv.y & 0xffff], wv.z); atomicAdd(&s[iv.y >> 16], wv.w);
      }
    }
  }
  __syncthreads();
  float* dst = partial + ((size_t)part * gridDim.x / parts + slice) * SLICE;
  for (int i = threadIdx.x; i < SLICE; i += blockDim.x) dst[i] = s[i];
}

static float time_ms(hipEvent_t a, hipEvent_t b) { float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms; }

int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  printf("device: %s CUs=%d clock=%d MHz LDS/block=%zu\n", prop.name, prop.multiProcessorCount, prop.clockRate / 1000, prop.sharedMemPerBlock);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int REP = 10;

  { // 1. stream
    size_t bytes = (size_t)4 << 30; float4* p; float* o; CK(hipMalloc(&p, bytes)); CK(hipMalloc(&o, 4));
    k_fill_f32<<<2048, 256>>>((float*)p, bytes / 4, 1.f);
    for (int g : {2048, 4096, 8192}) {
      k_stream<<<g, 256>>>(p, bytes / 16, o);
      CK(hipEventRecord(e0)); for (int r = 0; r < REP; ++r) k_stream<<<g, 256>>>(p, bytes / 16, o); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms = time_ms(e0, e1) / REP; printf("stream_read grid=%d: %.3f ms  %.1f GB/s\n", g, ms, bytes / ms / 1e6);
    }
    CK(hipFree(p)); CK(hipFree(o));
  }

  const size_t NUPD = 100000000;  // 1e8 updates = C2 step
  uint32_t* idx; float* w; CK(hipMalloc(&idx, NUPD * 4)); CK(hipMalloc(&w, NUPD * 4));
  k_fill_f32<<<2048, 256>>>(w, NUPD, 1.f);
  { // 2. random global atomics into tables of several sizes
    for (uint32_t tbl : {1000000u, 32768u, 16000000u}) {
      float* out; CK(hipMalloc(&out, (size_t)tbl * 4)); CK(hipMemset(out, 0, (size_t)tbl * 4));
      k_fill_u32<<<2048, 256>>>(idx, NUPD, 7u, tbl);
      k_gatomic<<<4096, 256>>>(idx, w, NUPD, out);
      CK(hipEventRecord(e0)); for (int r = 0; r < 3; ++r) k_gatomic<<<4096, 256>>>(idx, w, NUPD, out); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms = time_ms(e0, e1) / 3; printf("global_atomic_f32 random table=%u: %.3f ms  %.2f Gupd/s\n", tbl, ms, NUPD / ms / 1e6);
      CK(hipEventRecord(e0)); for (int r = 0; r < 3; ++r) k_gatomic_int<<<4096, 256>>>(idx, NUPD, (int*)out); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      ms = time_ms(e0, e1) / 3; printf("global_atomic_i32 random table=%u: %.3f ms  %.2f Gupd/s\n", tbl, ms, NUPD / ms / 1e6);
      CK(hipFree(out));
    }
  }
  { // 3. LDS atomics
    float* o; CK(hipMalloc(&o, 4));
    const int iters = 4096;
    for (int threads : {256, 512, 1024}) {
      CK(hipFuncSetAttribute((const void*)k_lds_atomic<32768>, hipFuncAttributeMaxDynamicSharedMemorySize, 32768 * 4));
      k_lds_atomic<32768><<<256, threads, 32768 * 4>>>(iters, o);
      CK(hipEventRecord(e0)); for (int r = 0; r < REP; ++r) k_lds_atomic<32768><<<256, threads, 32768 * 4>>>(iters, o); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms = time_ms(e0, e1) / REP; double n = 256.0 * threads * iters;
      printf("lds_atomic_f32 random slice=32768 threads=%d: %.3f ms  %.1f Gupd/s\n", threads, ms, n / ms / 1e6);
    }
    CK(hipFree(o));
  }
  { // 4. slice accumulate prototype: C2-like. n_rows reduced (layout only needs active rows to be spread)
    constexpr int SLICE = 32768; const int n_slices = 31, n_active = 10000, seg = 320;  // 31*320 = 9920 nnz / row
    const int n_rows = 200000;  // 200k rows * 31 slices * 320 * 6 B = 11.9 GB
    size_t n_ent = (size_t)n_slices * n_rows * seg;
    uint16_t* i16; float* ww; uint32_t* act; float* partial;
    CK(hipMalloc(&i16, n_ent * 2)); CK(hipMalloc(&ww, n_ent * 4)); CK(hipMalloc(&act, n_active * 4));
    k_fill_u16<<<4096, 256>>>(i16, n_ent, 3u, SLICE); k_fill_f32<<<4096, 256>>>(ww, n_ent, 1.f);
    k_fill_u32<<<64, 256>>>(act, n_active, 11u, n_rows);
    CK(hipFuncSetAttribute((const void*)k_slice_acc<SLICE, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, SLICE * 4));
    CK(hipFuncSetAttribute((const void*)k_slice_acc<SLICE, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, SLICE * 4));
    for (int parts : {8, 16}) {
      CK(hipMalloc(&partial, (size_t)parts * n_slices * SLICE * 4));
      for (int threads : {512, 1024}) {
        for (int vec : {1, 4}) {
          auto launch = [&]() {
            if (vec == 1) k_slice_acc<SLICE, 1><<<n_slices * parts, threads, SLICE * 4>>>(i16, ww, act, n_active, n_rows, seg, parts, partial);
            else k_slice_acc<SLICE, 4><<<n_slices * parts, threads, SLICE * 4>>>(i16, ww, act, n_active, n_rows, seg, parts, partial);
          };
          launch(); CK(hipDeviceSynchronize());
          CK(hipEventRecord(e0)); for (int r = 0; r < REP; ++r) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
          float ms = time_ms(e0, e1) / REP; double upd = (double)n_active * n_slices * seg;
          printf("slice_acc parts=%d threads=%d vec=%d: %.3f ms  %.1f Gupd/s  stream %.1f GB/s (6 B/upd)\n", parts, threads, vec, ms, upd / ms / 1e6, upd * 6 / ms / 1e6);
        }
      }
      CK(hipFree(partial));
    }
    // sanity: sum of partial == number of updates
    CK(hipFree(i16)); CK(hipFree(ww)); CK(hipFree(act));
  }
  CK(hipFree(idx)); CK(hipFree(w));
  printf("done\n");
  return 0;
}
