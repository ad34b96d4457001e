// be_common.h — shared host/device helpers for the brainevent_amd HIP kernels (gfx950 / CDNA4 only).
//
// Re-states, for wave64 HIP, the *semantics* the reference keeps in
// brainevent/include/cuda_common.h:86-385 (reference, read as text):
//   * a bool/int8 spike is active when != 0          (cuda_common.h:120-125  IS_ACTIVE_BOOL)
//   * a float spike is active when > 0               (cuda_common.h:126-131  IS_ACTIVE_FLOAT)
//   * f16 / bf16 weights accumulate in f32           (cuda_common.h:62-68, 204-216)
// Nothing here is shared with the oracle: the oracle is test infrastructure (see oracle/README.md).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <hip/hip_bf16.h>
#include <cstdint>
#include <cstdio>
#include <string>

// ---------------------------------------------------------------- error plumbing (never abort)
#include "../../include/brainevent_amd.h"   // BE_OK / BE_ERR_* codes

void be_set_error(const std::string& msg);
// Byte fill by a kernel.  Used instead of hipMemsetAsync everywhere on the step paths: a captured memset node of a few
// hundred KB replayed wrongly on this ROCm (half of a 160 KB buffer kept its old contents), kernels capture reliably.
hipError_t be_fill_async(void* p, int byte_value, size_t bytes, hipStream_t st);
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel, device, size): the call costs host time on every launch
hipError_t be_allow_lds(const void* kernel, int bytes);
// static LDS bytes of a kernel, cached per kernel (-1: the query failed)
int be_static_lds_bytes(const void* kernel);
// optional HIP-event timing of an op's dominant kernel (be_api.hip); slot -1 = profiling off
int be_prof_begin(hipStream_t st);
void be_prof_end(int slot, hipStream_t st);

#define BE_REQUIRE(cond, code, msg)                                                    \
  do {                                                                                 \
    if (!(cond)) {                                                                     \
      be_set_error(std::string(__func__) + ": " + (msg));                              \
      return (code);                                                                   \
    }                                                                                  \
  } while (0)

#define BE_HIP(call)                                                                   \
  do {                                                                                 \
    hipError_t e__ = (call);                                                           \
    if (e__ != hipSuccess) {                                                           \
      be_set_error(std::string(__func__) + ": " #call " -> " + hipGetErrorString(e__)); \
      return BE_ERR_HIP;                                                               \
    }                                                                                  \
  } while (0)

#define BE_LAUNCH_CHECK() BE_HIP(hipGetLastError())

// ---------------------------------------------------------------- dtype tags
struct be_f16 { using storage = __half; };
struct be_bf16 { using storage = __hip_bfloat16; };

template <typename T> struct WTraits;   // weight / output element
template <> struct WTraits<float> {
  using acc = float;
  __device__ static __forceinline__ float load(const float* p, int64_t i) { return p[i]; }
  __device__ static __forceinline__ void store(float* p, int64_t i, float v) { p[i] = v; }
  __device__ static __forceinline__ void store_d(float* p, int64_t i, double v) { p[i] = (float)v; }
};
template <> struct WTraits<double> {
  using acc = double;
  __device__ static __forceinline__ double load(const double* p, int64_t i) { return p[i]; }
  __device__ static __forceinline__ void store(double* p, int64_t i, double v) { p[i] = v; }
  __device__ static __forceinline__ void store_d(double* p, int64_t i, double v) { p[i] = v; }
};
template <> struct WTraits<__half> {
  using acc = float;
  __device__ static __forceinline__ float load(const __half* p, int64_t i) { return __half2float(p[i]); }
  __device__ static __forceinline__ void store(__half* p, int64_t i, float v) { p[i] = __float2half(v); }
  __device__ static __forceinline__ void store_d(__half* p, int64_t i, double v) { p[i] = __float2half((float)v); }
};
template <> struct WTraits<__hip_bfloat16> {
  using acc = float;
  __device__ static __forceinline__ float load(const __hip_bfloat16* p, int64_t i) { return __bfloat162float(p[i]); }
  __device__ static __forceinline__ void store(__hip_bfloat16* p, int64_t i, float v) { p[i] = __float2bfloat16(v); }
  __device__ static __forceinline__ void store_d(__hip_bfloat16* p, int64_t i, double v) { p[i] = __float2bfloat16((float)v); }
};

// spike element: bool tag = any 1-byte integer (!= 0), float tag = f32 (> 0)
struct SpikeBool {
  using type = uint8_t;
  __device__ static __forceinline__ bool active(uint8_t v) { return v != 0; }
};
struct SpikeFloat {
  using type = float;
  __device__ static __forceinline__ bool active(float v) { return v > 0.f; }
};

// ---------------------------------------------------------------- row pointer accessor
// CSR rows come from an int32 or int64 indptr; fixed-number connectivity (FixedNumPerPre) has no
// indptr at all: row r spans [r*fixed, (r+1)*fixed).
struct RowPtr {
  const void* p;
  int is64;
  int64_t fixed;
  __host__ __device__ __forceinline__ int64_t at(int64_t r) const {
    if (p == nullptr) return r * fixed;
    return is64 ? static_cast<const int64_t*>(p)[r] : (int64_t) static_cast<const int32_t*>(p)[r];
  }
};

// a += x with the sum forced back into a's own register: written as plain C++ inside the `mask != 0` branch, the compiler
// gives the updated accumulators new registers and copies ALL of them at the branch's merge point on every generated edge
// (16 v_mov_b64 per edge in the JITC mm gather — as much as the generator itself; 14 % of the f32 dense
// gather with 32 batch rows)
__device__ __forceinline__ void acc_add_inplace(float& a, float x) { asm volatile("v_add_f32_e32 %0, %0, %1" : "+v"(a) : "v"(x)); }
__device__ __forceinline__ void acc_add_inplace(double& a, double x) { asm volatile("v_add_f64 %0, %0, %1" : "+v"(a) : "v"(x)); }

// ---------------------------------------------------------------- wave64 helpers
constexpr int kWave = 64;

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;   // valid in lane 0
}

__device__ __forceinline__ uint32_t be_mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x;
}

static inline int64_t be_align_up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }
static inline int be_cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
