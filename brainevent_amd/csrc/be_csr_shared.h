// be_csr_shared.h — helpers shared by the CSR translation units (be_csr.hip: compaction, direct scatter, gather;
// be_csr_plan.hip: the scatter plan; be_csr_binned.hip: the binned route).  Device helpers are inline; the one host function
// with a single definition (be_resolve_active, be_csr.hip) is declared here.
#pragma once
#include "be_common.h"
#include <cmath>
#include <type_traits>

// ---------------------------------------------------------------- host-side launch helpers
static inline int grid_for(int64_t n, int block, int cap) {
  int64_t g = (n + block - 1) / block;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (int)g;
}

constexpr int kMaxBatch = 65535;
constexpr int kFusedMinBatch = 4;      // batched gather: fuse over the batch from this many columns ...
constexpr int64_t kFusedMinWork = 768; // ... when (average row length x columns) reaches this (and rows average >= 16 entries)

static inline int64_t counts_bytes(int64_t nb) { return be_align_up(nb * 4, 256); }
static inline int64_t active_stride_of(int64_t m) { return be_align_up(m * 4, 256) / 4; }   // in uint32 elements

#define BE_DISPATCH_W(wdtype, HOMO_FLAG, CALL)                                  \
  switch (wdtype) {                                                              \
    case BE_F32:  { using W = float;          if (HOMO_FLAG) { constexpr bool HOMO = true; CALL; } else { constexpr bool HOMO = false; CALL; } } break; \
    case BE_F64:  { using W = double;         if (HOMO_FLAG) { constexpr bool HOMO = true; CALL; } else { constexpr bool HOMO = false; CALL; } } break; \
    case BE_F16:  { using W = __half;         if (HOMO_FLAG) { constexpr bool HOMO = true; CALL; } else { constexpr bool HOMO = false; CALL; } } break; \
    case BE_BF16: { using W = __hip_bfloat16; if (HOMO_FLAG) { constexpr bool HOMO = true; CALL; } else { constexpr bool HOMO = false; CALL; } } break; \
    default: be_set_error("unknown weight dtype"); return BE_ERR_INVALID;       \
  }

static inline bool check_rows(const void* indptr, int64_t row_len) { return indptr != nullptr || row_len >= 0; }

static inline int64_t width_of(int slice_shift, int slice_width) { return slice_width > 0 ? slice_width : (1ll << slice_shift); }
static inline int n_slices_of(int64_t k, int slice_shift, int slice_width = 0) {
  const int64_t w = width_of(slice_shift, slice_width);
  return (int)((k + w - 1) / w);
}

// Active-row list of a scatter call: the spikes compacted into the workspace, or (BE_SPIKE_IDS, n_batch = 1) the
// caller's own list — `spikes` is then a HOST pointer to a be_spike_ids_t holding two device pointers.
struct ActiveList {
  const uint32_t* ids;
  const uint32_t* count;
};
int be_resolve_active(const void* spikes, int sd, int64_t n, int64_t nb, uint32_t* ws_active, int64_t astride,
                      uint32_t* ws_count, hipStream_t st, bool zero_first, ActiveList* al);

// ---------------------------------------------------------------- device helpers
// block-wide inclusive scan over 1024 threads (wave shuffles + one LDS hop): 2 barriers instead of 20
__device__ __forceinline__ uint32_t block_scan_1024(uint32_t v, uint32_t* wave_tot /* [16] in LDS */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t incl = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t t = __shfl_up(incl, off, 64);
    if (lane >= off) incl += t;
  }
  if (lane == 63) wave_tot[wave] = incl;
  __syncthreads();
  uint32_t base = 0;
#pragma unroll
  for (int w = 0; w < 16; ++w)
    if (w < wave) base += wave_tot[w];
  __syncthreads();
  return base + incl;
}

// ---------------------------------------------------------------- integer accumulation in LDS (plan and binned route)
template <bool HOMO> struct PlanAcc;
template <> struct PlanAcc<true> { using type = uint32_t; };
template <> struct PlanAcc<false> { using type = unsigned long long; };

// w * 2^scale_exp as a 64-bit two's-complement integer, built from f32 operations only:
//   t = w * 2^(scale_exp-32);  hi = floor(t);  lo = (t - hi) * 2^32   (all three steps are exact in f32:
//   power-of-two scaling, and t - floor(t) has no more significant bits than t).
// The caller guarantees |w| * 2^scale_exp < 2^62 / m, so hi fits an int32.  `scale` = 2^(scale_exp-32).
__device__ __forceinline__ unsigned long long fixed_from_f32(float w, float scale) {
  const float t = w * scale;
  const float hf = floorf(t);
  const int hi = (int)hf;
  const unsigned lo = (unsigned)((t - hf) * 4294967296.0f);
  return ((unsigned long long)(unsigned)hi << 32) | lo;
}

template <bool HOMO>
__device__ __forceinline__ void plan_add4(typename PlanAcc<HOMO>::type* acc, uint2 iv, float4 wv, float scale) {
  const uint32_t i0 = iv.x & 0xffffu, i1 = iv.x >> 16, i2 = iv.y & 0xffffu, i3 = iv.y >> 16;
  if (HOMO) {
    atomicAdd(&acc[i0], 1u);
    atomicAdd(&acc[i1], 1u);
    atomicAdd(&acc[i2], 1u);
    atomicAdd(&acc[i3], 1u);
  } else {
    atomicAdd(&acc[i0], fixed_from_f32(wv.x, scale));
    atomicAdd(&acc[i1], fixed_from_f32(wv.y, scale));
    atomicAdd(&acc[i2], fixed_from_f32(wv.z, scale));
    atomicAdd(&acc[i3], fixed_from_f32(wv.w, scale));
  }
}

__device__ __forceinline__ void plan_count8(uint32_t* acc, uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
  atomicAdd(&acc[a & 0xffffu], 1u); atomicAdd(&acc[a >> 16], 1u);
  atomicAdd(&acc[b & 0xffffu], 1u); atomicAdd(&acc[b >> 16], 1u);
  atomicAdd(&acc[c & 0xffffu], 1u); atomicAdd(&acc[c >> 16], 1u);
  atomicAdd(&acc[d & 0xffffu], 1u); atomicAdd(&acc[d >> 16], 1u);
}

typedef unsigned be_v2u __attribute__((ext_vector_type(2)));
typedef unsigned be_v4u __attribute__((ext_vector_type(4)));
constexpr int kBufFlags = 0x00020000;   // raw buffer, 32-bit data format (guide T8)
