// be_dense.hip — event-driven dense products  BinaryArray @ ndarray  for gfx950 (vector path).
//
// Computes what the reference's CPU kernels compute (brainevent/_dense/binary.py:168-211 mv,
// :579-632 mm, read as text):
//   transpose=True : weights[k, n], spikes[k, nb]  ->  out[n, nb] = sum_{i : e(s[i, b])} weights[i, :]
//   transpose=False: weights[m, k], spikes[k, nb]  ->  out[m, nb] = sum_{j : e(s[j, b])} weights[:, j]
// Physical layouts here are batch-major (spikes_bm [nb, k], out_bm [nb, n|m]) like the reference's CUDA
// kernels emit (brainevent/_dense/binary_densemm.cu:50-93, "Python transposes", _dense/binary.py:980-987).
//
// The contraction is HBM-bound on the weight rows that carry at least one spike (arithmetic intensity
// <= 2*nb flop per weight byte), so the design goal is to read every needed weight byte exactly once,
// 16 B per lane, and to skip rows without spikes:
//   transpose=True : spikes -> per-row batch masks -> ordered per-batch-group row lists; one wave owns
//                    (column strip, group of 4 batches, row part) and keeps 4 x VEC f32 accumulators in
//                    registers; partial sums per row part are reduced in a fixed order (deterministic).
//   transpose=False: one wave per weight row streams it with 16 B loads against the mask vector
//                    (or gathers only the active columns when fewer than 1/16 of them are active).
// The MFMA kernels for the batched fp16 / bf16 case (k_densemm_mfma, k_densemm_nt_mfma) follow the vector kernels below.
#include "be_common.h"
#include <algorithm>
#include <type_traits>

#ifndef BE_DENSE_NT
#define BE_DENSE_NT 1     // nt policy on the row gathers of the S @ W MFMA kernels: each row piece is
                          // used once, and keeping it out of L2's way measured +8 % (C5 0.467 -> 0.427 ms).
                          // NOT on the W @ S.T kernels, whose strided 16-B pieces share sectors (1.8 -> 2.7 ms).
#endif
typedef unsigned be_dv4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 dense_row_load(const void* p) {
  if (BE_DENSE_NT) { const be_dv4u t = __builtin_nontemporal_load(reinterpret_cast<const be_dv4u*>(p)); return make_uint4(t.x, t.y, t.z, t.w); }
  return *reinterpret_cast<const uint4*>(p);
}

namespace {

constexpr int kGroup = 4;        // batches per wave in the transpose=True kernel
constexpr int kMaxChunk = 32;    // batches per pass (one uint32 mask per row)

template <typename W> struct Vec16 { static constexpr int n = 16 / sizeof(W); };

// ------------------------------------------------------------------------------------------------
// spikes_bm[nb, k] (rows b0 .. b0+nc) -> mask[k], bit b set iff spike (b0+b, k) active
// ------------------------------------------------------------------------------------------------
template <typename SP>
__global__ void __launch_bounds__(256) k_dense_masks(const typename SP::type* __restrict__ spikes, int64_t k, int nc,
                                                     uint32_t* __restrict__ mask) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < k; i += stride) {
    uint32_t mk = 0;
    for (int b0 = 0; b0 < nc; b0 += 8) {       // eight batch rows' loads in flight (one at a time: a round trip per batch row)
      typename SP::type v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = spikes[(int64_t)(b0 + u < nc ? b0 + u : nc - 1) * k + i];
#pragma unroll
      for (int u = 0; u < 8; ++u) mk |= ((b0 + u < nc && SP::active(v[u])) ? 1u : 0u) << (b0 + u);
    }
    mask[i] = mk;
  }
}

// ordered compaction of the rows whose sub-mask for batch group g (= blockIdx.y) is non-zero.
// entry = row | submask << 28  (rows < 2^28).  Three passes: per-tile counts, scan, write.
constexpr int kTile = 2048;   // rows per workgroup (256 threads x 8)

__device__ __forceinline__ uint32_t submask_of(uint32_t mk, int g) { return (mk >> (kGroup * g)) & ((1u << kGroup) - 1u); }

// WHOLE: one list of the rows with any bit set (the union of the batch rows) instead of one list per batch group
template <bool WHOLE = false>
__global__ void __launch_bounds__(256) k_gl_count(const uint32_t* __restrict__ mask, int64_t k, uint32_t* __restrict__ tile_cnt) {
  __shared__ uint32_t red[4];
  const int g = blockIdx.y;
  const int64_t base = (int64_t)blockIdx.x * kTile + (int64_t)threadIdx.x * 8;
  uint32_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i)
    if (base + i < k && (WHOLE ? mask[base + i] : submask_of(mask[base + i], g))) ++c;
  c = wave_sum(c);
  if (lane_id() == 0) red[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) tile_cnt[(int64_t)g * gridDim.x + blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// k_dense_masks and k_gl_count<true> in one launch: workgroup = one tile of kTile rows, a thread = 8 consecutive rows.
// One-byte spikes are read 8 rows at a time (one 8-byte load per batch row and thread, eight batch rows in flight): with a
// byte per load the 32 workgroups of a 65536-row operand took 20 us.
template <typename SP>
__global__ void __launch_bounds__(256) k_dense_masks_count(const typename SP::type* __restrict__ spikes, int64_t k, int nc,
                                                           uint32_t* __restrict__ mask, uint32_t* __restrict__ tile_cnt) {
  __shared__ uint32_t red[4];
  const int64_t base = (int64_t)blockIdx.x * kTile + (int64_t)threadIdx.x * 8;
  uint32_t mk[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
  bool done = false;
  if constexpr (sizeof(typename SP::type) == 1) {
    if (base + 8 <= k && (k & 7) == 0 && (reinterpret_cast<uintptr_t>(spikes) & 7) == 0) {
      for (int b0 = 0; b0 < nc; b0 += 8) {
        uint2 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u)
          v[u] = *reinterpret_cast<const uint2*>(spikes + (int64_t)(b0 + u < nc ? b0 + u : nc - 1) * k + base);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const uint32_t bit = b0 + u < nc ? 1u << (b0 + u) : 0u;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            mk[i] |= ((v[u].x >> (8 * i)) & 0xffu) ? bit : 0u;
            mk[4 + i] |= ((v[u].y >> (8 * i)) & 0xffu) ? bit : 0u;
          }
        }
      }
      done = true;
    }
  }
  if (!done) {
    for (int i = 0; i < 8; ++i) {
      if (base + i >= k) break;
      for (int b0 = 0; b0 < nc; b0 += 8) {
        typename SP::type v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = spikes[(int64_t)(b0 + u < nc ? b0 + u : nc - 1) * k + base + i];
#pragma unroll
        for (int u = 0; u < 8; ++u) mk[i] |= ((b0 + u < nc && SP::active(v[u])) ? 1u : 0u) << (b0 + u);
      }
    }
  }
  uint32_t c = 0;
  if (base + 8 <= k) {
    *reinterpret_cast<uint4*>(mask + base) = make_uint4(mk[0], mk[1], mk[2], mk[3]);
    *reinterpret_cast<uint4*>(mask + base + 4) = make_uint4(mk[4], mk[5], mk[6], mk[7]);
#pragma unroll
    for (int i = 0; i < 8; ++i) c += mk[i] ? 1u : 0u;
  } else {
    for (int i = 0; i < 8 && base + i < k; ++i) { mask[base + i] = mk[i]; c += mk[i] ? 1u : 0u; }
  }
  c = wave_sum(c);
  if (lane_id() == 0) red[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) tile_cnt[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// BE_SPIKE_BITS operands (a batch of bit-packed event vectors, nc rows of ceil(k / 32) words — what BitPackedBinary holds and
// what the spike exchange delivers): the same masks by a bit transpose, read from the words as they are.  The reference's
// kernels pack on entry (brainevent/_jit_scalar/binary_jitsmm.cu:15-26); here a packed operand is never unpacked.
__global__ void __launch_bounds__(256) k_dense_masks_bits(const uint32_t* __restrict__ words, int64_t k, int nc,
                                                          uint32_t* __restrict__ mask) {
  const int64_t n_words = (k + 31) / 32;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < k; i += stride) {
    uint32_t mk = 0;
    for (int b0 = 0; b0 < nc; b0 += 8) {
      uint32_t v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = words[(int64_t)(b0 + u < nc ? b0 + u : nc - 1) * n_words + (i >> 5)];
#pragma unroll
      for (int u = 0; u < 8; ++u) mk |= (b0 + u < nc ? (v[u] >> (i & 31)) & 1u : 0u) << (b0 + u);
    }
    mask[i] = mk;
  }
}

__global__ void __launch_bounds__(256) k_dense_masks_count_bits(const uint32_t* __restrict__ words, int64_t k, int nc,
                                                                uint32_t* __restrict__ mask, uint32_t* __restrict__ tile_cnt) {
  __shared__ uint32_t red[4];
  const int64_t n_words = (k + 31) / 32;
  const int64_t base = (int64_t)blockIdx.x * kTile + (int64_t)threadIdx.x * 8;      // 8 consecutive rows: one byte of a word
  uint32_t mk[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
  if (base < k) {
    for (int b0 = 0; b0 < nc; b0 += 8) {
      uint32_t v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = words[(int64_t)(b0 + u < nc ? b0 + u : nc - 1) * n_words + (base >> 5)];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const uint32_t byte = b0 + u < nc ? (v[u] >> (base & 31)) & 0xffu : 0u;
#pragma unroll
        for (int i = 0; i < 8; ++i) mk[i] |= ((byte >> i) & 1u) << (b0 + u);
      }
    }
  }
  uint32_t c = 0;
  if (base + 8 <= k) {
    *reinterpret_cast<uint4*>(mask + base) = make_uint4(mk[0], mk[1], mk[2], mk[3]);
    *reinterpret_cast<uint4*>(mask + base + 4) = make_uint4(mk[4], mk[5], mk[6], mk[7]);
#pragma unroll
    for (int i = 0; i < 8; ++i) c += mk[i] ? 1u : 0u;
  } else {
    for (int i = 0; i < 8 && base + i < k; ++i) { mask[base + i] = mk[i]; c += mk[i] ? 1u : 0u; }
  }
  c = wave_sum(c);
  if (lane_id() == 0) red[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) tile_cnt[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// bytes of one batch row of a spike operand
static inline size_t spike_row_bytes(int sd, int64_t k) {
  return sd == BE_SPIKE_BITS ? (size_t)((k + 31) / 32) * 4 : (size_t)k * (sd == BE_SPIKE_FLOAT ? 4 : 1);
}
static inline int grid_cap_fwd(int64_t n, int block, int cap) { int64_t g = (n + block - 1) / block; return (int)(g < 1 ? 1 : (g > cap ? cap : g)); }
// masks of a chunk of batch rows, any spike encoding
static inline void launch_dense_masks(const void* chunk, int sd, int64_t k, int nc, uint32_t* mask, hipStream_t st) {
  const dim3 grid(grid_cap_fwd(k, 256, 2048));
  if (sd == BE_SPIKE_BITS) hipLaunchKernelGGL(k_dense_masks_bits, grid, dim3(256), 0, st, static_cast<const uint32_t*>(chunk), k, nc, mask);
  else if (sd == BE_SPIKE_FLOAT) hipLaunchKernelGGL(k_dense_masks<SpikeFloat>, grid, dim3(256), 0, st, static_cast<const float*>(chunk), k, nc, mask);
  else hipLaunchKernelGGL(k_dense_masks<SpikeBool>, grid, dim3(256), 0, st, static_cast<const uint8_t*>(chunk), k, nc, mask);
}
static inline void launch_dense_masks_count(const void* chunk, int sd, int64_t k, int nc, uint32_t* mask, uint32_t* tile_cnt, int64_t nt,
                                            hipStream_t st) {
  if (sd == BE_SPIKE_BITS)
    hipLaunchKernelGGL(k_dense_masks_count_bits, dim3((unsigned)nt), dim3(256), 0, st, static_cast<const uint32_t*>(chunk), k, nc, mask, tile_cnt);
  else if (sd == BE_SPIKE_FLOAT)
    hipLaunchKernelGGL(k_dense_masks_count<SpikeFloat>, dim3((unsigned)nt), dim3(256), 0, st, static_cast<const float*>(chunk), k, nc, mask, tile_cnt);
  else
    hipLaunchKernelGGL(k_dense_masks_count<SpikeBool>, dim3((unsigned)nt), dim3(256), 0, st, static_cast<const uint8_t*>(chunk), k, nc, mask, tile_cnt);
}

// one workgroup per group: exclusive scan of that group's tile counts; count[g] = total
__global__ void __launch_bounds__(1024) k_gl_scan(uint32_t* __restrict__ tile_cnt, int64_t n_tiles, uint32_t* __restrict__ count) {
  __shared__ uint32_t part[1024];
  __shared__ uint32_t carry;
  uint32_t* tc = tile_cnt + (int64_t)blockIdx.x * n_tiles;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int64_t base = 0; base < n_tiles; base += 1024) {
    const int64_t i = base + threadIdx.x;
    const uint32_t v = (i < n_tiles) ? tc[i] : 0u;
    part[threadIdx.x] = v;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
      uint32_t t = 0;
      if ((int)threadIdx.x >= off) t = part[threadIdx.x - off];
      __syncthreads();
      part[threadIdx.x] += t;
      __syncthreads();
    }
    if (i < n_tiles) tc[i] = carry + part[threadIdx.x] - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry += part[1023];
    __syncthreads();
  }
  if (threadIdx.x == 0) count[blockIdx.x] = carry;
}

// SELF_SCAN: tile_off holds the raw per-tile counts (at most 1024 tiles) and every workgroup sums the tiles in front of
// it itself; the last one also writes the total to *count — no scan launch.
template <bool WHOLE = false, bool SELF_SCAN = false>
__global__ void __launch_bounds__(256) k_gl_write(const uint32_t* __restrict__ mask, int64_t k,
                                                  const uint32_t* __restrict__ tile_off, uint32_t* __restrict__ lists,
                                                  int64_t list_stride, uint32_t* __restrict__ count = nullptr) {
  __shared__ uint32_t wave_tot[4];
  __shared__ uint32_t front_s[4];
  uint32_t front = 0;
  if (SELF_SCAN) {
    uint32_t f = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int t = q * 256 + (int)threadIdx.x;
      f += t < (int)blockIdx.x ? tile_off[t] : 0u;
    }
    f = wave_sum(f);
    if (lane_id() == 0) front_s[threadIdx.x >> 6] = f;
    __syncthreads();
    front = front_s[0] + front_s[1] + front_s[2] + front_s[3];
  }
  const int g = blockIdx.y;
  const int64_t base = (int64_t)blockIdx.x * kTile + (int64_t)threadIdx.x * 8;
  uint32_t sm[8];
  uint32_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    sm[i] = (base + i < k) ? (WHOLE ? (mask[base + i] ? 1u : 0u) : submask_of(mask[base + i], g)) : 0u;
    c += sm[i] ? 1u : 0u;
  }
  const int lane = lane_id(), wave = threadIdx.x >> 6;
  uint32_t incl = c;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t t = __shfl_up(incl, off, 64);
    if (lane >= off) incl += t;
  }
  if (lane == 63) wave_tot[wave] = incl;
  __syncthreads();
  uint32_t wave_off = 0;
  for (int w = 0; w < wave; ++w) wave_off += wave_tot[w];
  uint32_t pos = (SELF_SCAN ? front : tile_off[(int64_t)g * gridDim.x + blockIdx.x]) + wave_off + incl - c;
  if (SELF_SCAN && blockIdx.x == gridDim.x - 1 && threadIdx.x == 255) count[0] = front + wave_off + incl;   // the total
  uint32_t* out = lists + (int64_t)g * list_stride;
#pragma unroll
  for (int i = 0; i < 8; ++i)
    if (sm[i]) out[pos++] = (uint32_t)(base + i) | (sm[i] << 28);
}

// ------------------------------------------------------------------------------------------------
// transpose=True accumulate: wave task = (strip, group); blockIdx.y = row part
// ------------------------------------------------------------------------------------------------
#ifndef BE_DENSE_T_NT
#define BE_DENSE_T_NT 1   // nt on the row-piece gathers of the vector S @ W kernel (f32 n=32768 p=0.5: 0.39 -> 0.35 ms)
#endif
#ifndef BE_DENSE_S_NT
#define BE_DENSE_S_NT 1   // nt on the streamed rows of the vector W @ S.T kernel (f32 n=32768 p=0.5: 0.81 -> 0.72 ms)
#endif
template <int POL> __device__ __forceinline__ uint4 load16(const void* p) {
  if (POL) { const be_dv4u t = __builtin_nontemporal_load(reinterpret_cast<const be_dv4u*>(p)); return make_uint4(t.x, t.y, t.z, t.w); }
  return *reinterpret_cast<const uint4*>(p);
}
template <typename W, int VEC> struct RowLoad;
template <typename W> struct RowLoad<W, 1> {
  using ACC = typename WTraits<W>::acc;
  template <int POL = 0> __device__ static __forceinline__ void load(const W* p, ACC (&v)[1]) { v[0] = (ACC)WTraits<W>::load(p, 0); }
};
template <> struct RowLoad<float, 4> {
  template <int POL = 0> __device__ static __forceinline__ void load(const float* p, float (&v)[4]) {
    const uint4 t = load16<POL>(p);
    v[0] = __uint_as_float(t.x); v[1] = __uint_as_float(t.y); v[2] = __uint_as_float(t.z); v[3] = __uint_as_float(t.w);
  }
};
template <> struct RowLoad<double, 2> {
  template <int POL = 0> __device__ static __forceinline__ void load(const double* p, double (&v)[2]) {
    const uint4 t = load16<POL>(p);
    v[0] = __hiloint2double((int)t.y, (int)t.x); v[1] = __hiloint2double((int)t.w, (int)t.z);
  }
};
template <> struct RowLoad<__half, 8> {
  template <int POL = 0> __device__ static __forceinline__ void load(const __half* p, float (&v)[8]) {
    const uint4 t = load16<POL>(p);
    const uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const __half2 h = *reinterpret_cast<const __half2*>(&w[i]);
      const float2 f = __half22float2(h);
      v[2 * i] = f.x; v[2 * i + 1] = f.y;
    }
  }
};
template <> struct RowLoad<__hip_bfloat16, 8> {
  template <int POL = 0> __device__ static __forceinline__ void load(const __hip_bfloat16* p, float (&v)[8]) {
    const uint4 t = load16<POL>(p);
    const uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v[2 * i] = __uint_as_float(w[i] << 16);
      v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
  }
};

template <typename W, int VEC>
__global__ void __launch_bounds__(256) k_densemm_t(const W* __restrict__ weights, int64_t n, const uint32_t* __restrict__ lists,
                                                   int64_t list_stride, const uint32_t* __restrict__ count, int n_groups,
                                                   int nc, typename WTraits<W>::acc* __restrict__ partial, int64_t nb_total,
                                                   int b0) {
  using ACC = typename WTraits<W>::acc;
  const int lane = lane_id();
  const int64_t task = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t n_strips = (n + 64 * VEC - 1) / (64 * VEC);
  if (task >= n_strips * n_groups) return;
  const int g = (int)(task % n_groups);
  const int64_t strip = task / n_groups;
  const int64_t col = strip * (64 * VEC) + (int64_t)lane * VEC;
  const bool in = col < n;            // VEC > 1 only when n % VEC == 0, so a lane is all-in or all-out
  const uint32_t cnt = count[g];
  const uint32_t* list = lists + (int64_t)g * list_stride;
  const int part = blockIdx.y, parts = gridDim.y;

  ACC acc[kGroup][VEC];
#pragma unroll
  for (int b = 0; b < kGroup; ++b)
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[b][v] = ACC(0);

  // rows of this part: a = part, part + parts, ...  (UNR row loads in flight per lane)
  constexpr int UNR = 8;       // (4 in flight: f32 32k^2 at 50 % firing 3.5 TB/s; 8: 4.5; with 32 row parts 5.5)
  uint32_t a = part;
  for (; a + (uint32_t)(UNR - 1) * parts < cnt; a += (uint32_t)UNR * parts) {
    uint32_t e[UNR];
    ACC w[UNR][VEC];
#pragma unroll
    for (int q = 0; q < UNR; ++q) e[q] = list[a + q * parts];
#pragma unroll
    for (int q = 0; q < UNR; ++q)
      if (in) RowLoad<W, VEC>::template load<BE_DENSE_T_NT>(weights + (int64_t)(e[q] & 0x0fffffffu) * n + col, w[q]);
#pragma unroll
    for (int q = 0; q < UNR; ++q) {
      const uint32_t sm = e[q] >> 28;
#pragma unroll
      for (int b = 0; b < kGroup; ++b)
        if (sm & (1u << b)) {
#pragma unroll
          for (int v = 0; v < VEC; ++v) acc[b][v] += w[q][v];
        }
    }
  }
  for (; a < cnt; a += parts) {      // (a predicated last round instead of this tail was measured: no faster at 1 % firing, slower dense)
    const uint32_t e = list[a];
    ACC w[VEC];
    if (in) RowLoad<W, VEC>::template load<BE_DENSE_T_NT>(weights + (int64_t)(e & 0x0fffffffu) * n + col, w);
    const uint32_t sm = e >> 28;
#pragma unroll
    for (int b = 0; b < kGroup; ++b)
      if (sm & (1u << b)) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[b][v] += w[v];
      }
  }
  if (!in) return;
#pragma unroll
  for (int b = 0; b < kGroup; ++b) {
    const int bb = kGroup * g + b;
    if (bb >= nc) break;
    ACC* dst = partial + ((int64_t)part * nb_total + b0 + bb) * n + col;
#pragma unroll
    for (int v = 0; v < VEC; ++v) dst[v] = acc[b][v];
  }
}

template <typename W>
__global__ void __launch_bounds__(256) k_dense_reduce(const typename WTraits<W>::acc* __restrict__ partial, int parts,
                                                      int64_t total, W* __restrict__ out) {
  using ACC = typename WTraits<W>::acc;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    ACC s = ACC(0);
    for (int p0 = 0; p0 < parts; p0 += 8) {      // eight parts' loads in flight; the sum keeps the order of the parts
      ACC v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = partial[(int64_t)(p0 + u < parts ? p0 + u : parts - 1) * total + i];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += p0 + u < parts ? v[u] : ACC(0);
    }
    WTraits<W>::store(out, i, s);
  }
}

// ------------------------------------------------------------------------------------------------
// transpose=False: one wave per weight row
// ------------------------------------------------------------------------------------------------
template <typename W, int VEC, int NBT>
__global__ void __launch_bounds__(256) k_densemm_nt(const W* __restrict__ weights, int64_t m, int64_t k,
                                                    const uint32_t* __restrict__ mask, const uint32_t* __restrict__ ulist,
                                                    const uint32_t* __restrict__ ucount, int nc, W* __restrict__ out_bm,
                                                    int b0) {
  using ACC = typename WTraits<W>::acc;
  const int lane = lane_id();
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const uint32_t n_union = ucount[0];
  const bool gather = (int64_t)n_union * 16 < k;   // few active columns: touch only their 64-B sectors
  for (int64_t r = wave; r < m; r += n_waves) {
    const W* row = weights + r * k;
    ACC acc[NBT];
#pragma unroll
    for (int b = 0; b < NBT; ++b) acc[b] = ACC(0);
    if (gather) {
      uint32_t a = lane;
      for (; a + 3u * 64u < n_union; a += 4u * 64u) {      // four scattered elements in flight per lane
        uint32_t j[4], mk[4];
        ACC w[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) j[u] = ulist[a + 64u * u] & 0x0fffffffu;
#pragma unroll
        for (int u = 0; u < 4; ++u) { mk[u] = mask[j[u]]; w[u] = (ACC)WTraits<W>::load(row, j[u]); }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int b = 0; b < NBT; ++b) acc_add_inplace(acc[b], ((mk[u] >> b) & 1u) ? w[u] : ACC(0));
      }
      for (; a < n_union; a += 64) {
        const uint32_t j = ulist[a] & 0x0fffffffu;
        const uint32_t mk = mask[j];
        const ACC w = (ACC)WTraits<W>::load(row, j);
#pragma unroll
        for (int b = 0; b < NBT; ++b) acc_add_inplace(acc[b], ((mk >> b) & 1u) ? w : ACC(0));
      }
    } else {
      // U row pieces of 16 B and their masks (one vector load per piece) in flight per lane, then the tail piece by piece
      constexpr int U = 4;
      int64_t j = (int64_t)lane * VEC;
      for (; j + (int64_t)(U - 1) * 64 * VEC < k; j += (int64_t)U * 64 * VEC) {
        ACC w[U][VEC];
        uint32_t mk[U][VEC];
#pragma unroll
        for (int u = 0; u < U; ++u) RowLoad<W, VEC>::template load<BE_DENSE_S_NT>(row + j + (int64_t)u * 64 * VEC, w[u]);
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const uint32_t* mp = mask + j + (int64_t)u * 64 * VEC;
          if constexpr (VEC % 4 == 0) {
#pragma unroll
            for (int v4 = 0; v4 < VEC / 4; ++v4) {
              const uint4 t = reinterpret_cast<const uint4*>(mp)[v4];
              mk[u][4 * v4] = t.x; mk[u][4 * v4 + 1] = t.y; mk[u][4 * v4 + 2] = t.z; mk[u][4 * v4 + 3] = t.w;
            }
          } else {
#pragma unroll
            for (int v = 0; v < VEC; ++v) mk[u][v] = mp[v];
          }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
          for (int v = 0; v < VEC; ++v) {
            if (mk[u][v]) {
#pragma unroll
              for (int b = 0; b < NBT; ++b) acc_add_inplace(acc[b], ((mk[u][v] >> b) & 1u) ? w[u][v] : ACC(0));
            }
          }
      }
      for (; j < k; j += 64 * VEC) {
        ACC w[VEC];
        RowLoad<W, VEC>::template load<BE_DENSE_S_NT>(row + j, w);
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          const uint32_t mk = mask[j + v];
          if (mk) {
#pragma unroll
            for (int b = 0; b < NBT; ++b) acc_add_inplace(acc[b], ((mk >> b) & 1u) ? w[v] : ACC(0));
          }
        }
      }
    }
#pragma unroll
    for (int b = 0; b < NBT; ++b) {
      ACC s = acc[b];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
      if (lane == 0 && b < nc) WTraits<W>::store(out_bm, (int64_t)(b0 + b) * m + r, s);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// host
// ------------------------------------------------------------------------------------------------
inline int grid_cap(int64_t n, int block, int cap) {
  int64_t g = (n + block - 1) / block;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

struct DenseWs {
  uint32_t* mask;      // k
  uint32_t* lists;     // (n_groups_max + 1) * k   (group lists; the last one is the union list for NT)
  uint32_t* tile_cnt;  // (n_groups_max + 1) * n_tiles
  uint32_t* count;     // 16
  void* partial;
};

inline int64_t n_tiles_of(int64_t k) { return (k + kTile - 1) / kTile; }
constexpr int kMaxGroups = kMaxChunk / kGroup;   // 8

inline int parts_for(int64_t n, int vec, int n_groups) {
  const int64_t tasks = ((n + 64 * vec - 1) / (64 * vec)) * n_groups;
  const int64_t wgs = (tasks + 3) / 4;
  // workgroups to aim for / row parts at most.  A single vector (one batch group) has few strips: 2048 workgroups of four
  // waves, each lane with 8 row loads in flight, are what it takes to keep HBM busy (fp16 64k^2 at 50 % firing: 16 parts
  // 2.9 TB/s, 32 parts 4.6; f32 32k^2: 3.5 -> 5.5); with several groups the partial sums of more parts cost more than they
  // buy (f32, 8 batch rows: 16 parts 3.8 TB/s, 32 parts 3.4).  The partial buffer is sized for 16 parts x 32 batch rows.
  const int64_t target = n_groups == 1 ? 2048 : 1024, cap = n_groups == 1 ? 32 : 16;
  int64_t p = target / (wgs > 0 ? wgs : 1);
  if (p < 1) p = 1;
  if (p > cap) p = cap;
  return (int)p;
}

inline int64_t dense_ws_bytes(int64_t rows_w, int64_t cols_w, int64_t nb, int transpose, int wdtype) {
  // the contraction dimension: rows (transpose) or columns of the weight matrix
  const int64_t k = transpose ? rows_w : cols_w;
  int64_t b = be_align_up(k * 4, 256);                                   // mask
  b += be_align_up((int64_t)(kMaxGroups + 1) * k * 4, 256);              // lists
  b += be_align_up((int64_t)(kMaxGroups + 1) * n_tiles_of(k) * 4, 256);  // tile counts
  b += 256;                                                              // counts
  if (transpose) {
    const int64_t acc = (wdtype == BE_F64) ? 8 : 4;
    const int64_t rows_p = nb < 32 ? 32 : nb;                            // the MFMA path keeps 32 batch rows per part
    b += be_align_up((int64_t)16 * rows_p * cols_w * acc, 256);          // partial (<= 16 parts x >= 32 rows: also 32 parts x <= 4 rows)
  }
  return b;
}

inline DenseWs carve(void* ws, int64_t k) {
  unsigned char* p = static_cast<unsigned char*>(ws);
  DenseWs d;
  d.mask = reinterpret_cast<uint32_t*>(p); p += be_align_up(k * 4, 256);
  d.lists = reinterpret_cast<uint32_t*>(p); p += be_align_up((int64_t)(kMaxGroups + 1) * k * 4, 256);
  d.tile_cnt = reinterpret_cast<uint32_t*>(p); p += be_align_up((int64_t)(kMaxGroups + 1) * n_tiles_of(k) * 4, 256);
  d.count = reinterpret_cast<uint32_t*>(p); p += 256;
  d.partial = p;
  return d;
}

int build_lists(const void* spikes_chunk, int sd, int64_t k, int nc, int n_groups, const DenseWs& d, hipStream_t st) {
  launch_dense_masks(spikes_chunk, sd, k, nc, d.mask, st);
  BE_LAUNCH_CHECK();
  const int64_t nt = n_tiles_of(k);
  hipLaunchKernelGGL(k_gl_count<false>, dim3((unsigned)nt, n_groups), dim3(256), 0, st, d.mask, k, d.tile_cnt);
  BE_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_gl_scan, dim3(n_groups), dim3(1024), 0, st, d.tile_cnt, nt, d.count);
  BE_LAUNCH_CHECK();
  hipLaunchKernelGGL((k_gl_write<false, false>), dim3((unsigned)nt, n_groups), dim3(256), 0, st, d.mask, k, d.tile_cnt, d.lists, k,
                     static_cast<uint32_t*>(nullptr));
  BE_LAUNCH_CHECK();
  return BE_OK;
}

template <typename W, int VEC>
int densemm_t_vec(const W* weights, const void* spikes_bm, int sd, W* out_bm, int64_t k, int64_t n, int64_t nb, void* ws,
                  hipStream_t st) {
  using ACC = typename WTraits<W>::acc;
  DenseWs d = carve(ws, k);
  ACC* partial = static_cast<ACC*>(d.partial);
  int parts_used = 1;
  // parts must be the same for every chunk so that one reduce pass serves the whole batch
  parts_used = parts_for(n, VEC, (int)((std::min<int64_t>(nb, kMaxChunk) + kGroup - 1) / kGroup));
  const int prof = be_prof_begin(st);
  for (int64_t b0 = 0; b0 < nb; b0 += kMaxChunk) {
    const int nc = (int)std::min<int64_t>(kMaxChunk, nb - b0);
    const int n_groups = (nc + kGroup - 1) / kGroup;
    const void* chunk = static_cast<const unsigned char*>(spikes_bm) + (size_t)b0 * spike_row_bytes(sd, k);
    int rc = build_lists(chunk, sd, k, nc, n_groups, d, st);
    if (rc != BE_OK) return rc;
    const int64_t tasks = ((n + 64 * VEC - 1) / (64 * VEC)) * n_groups;
    hipLaunchKernelGGL((k_densemm_t<W, VEC>), dim3((unsigned)((tasks + 3) / 4), parts_used), dim3(256), 0, st, weights, n,
                       d.lists, k, d.count, n_groups, nc, partial, nb, (int)b0);
    BE_LAUNCH_CHECK();
  }
  be_prof_end(prof, st);
  hipLaunchKernelGGL(k_dense_reduce<W>, dim3(grid_cap(nb * n, 256, 2048)), dim3(256), 0, st, partial, parts_used, nb * n,
                     out_bm);
  BE_LAUNCH_CHECK();
  return BE_OK;
}

template <typename W, int VEC, int NBT>
int densemm_nt_launch(const W* weights, int64_t m, int64_t k, const DenseWs& d, int nc, W* out_bm, int b0, hipStream_t st) {
  hipLaunchKernelGGL((k_densemm_nt<W, VEC, NBT>), dim3(grid_cap(m, 4, 256 * 8)), dim3(256), 0, st, weights, m, k, d.mask,
                     d.lists + (int64_t)kMaxGroups * k, d.count + kMaxGroups, nc, out_bm, b0);
  BE_LAUNCH_CHECK();
  return BE_OK;
}

template <typename W, int VEC>
int densemm_nt_vec(const W* weights, const void* spikes_bm, int sd, W* out_bm, int64_t m, int64_t k, int64_t nb, void* ws,
                   hipStream_t st) {
  DenseWs d = carve(ws, k);
  const int64_t nt = n_tiles_of(k);
  uint32_t* ulist = d.lists + (int64_t)kMaxGroups * k;          // union list in the last slot
  uint32_t* utile = d.tile_cnt + (int64_t)kMaxGroups * nt;
  const int prof = be_prof_begin(st);
  for (int64_t b0 = 0; b0 < nb; b0 += kMaxChunk) {
    const int nc = (int)std::min<int64_t>(kMaxChunk, nb - b0);
    const void* chunk = static_cast<const unsigned char*>(spikes_bm) + (size_t)b0 * spike_row_bytes(sd, k);
    // masks + per-tile counts of the active columns in one launch, scan, ordered union list: three launches (five before:
    // masks, 0/1 flags, count, scan, write — 16 us more per call at C5)
    launch_dense_masks_count(chunk, sd, k, nc, d.mask, utile, nt, st);
    BE_LAUNCH_CHECK();
    if (nt <= 1024) {      // every workgroup of the write sums the tiles in front of it: no scan launch
      hipLaunchKernelGGL((k_gl_write<true, true>), dim3((unsigned)nt, 1), dim3(256), 0, st, d.mask, k, utile, ulist, k,
                         d.count + kMaxGroups);
    } else {
      hipLaunchKernelGGL(k_gl_scan, dim3(1), dim3(1024), 0, st, utile, nt, d.count + kMaxGroups);
      BE_LAUNCH_CHECK();
      hipLaunchKernelGGL((k_gl_write<true, false>), dim3((unsigned)nt, 1), dim3(256), 0, st, d.mask, k, utile, ulist, k,
                         static_cast<uint32_t*>(nullptr));
    }
    BE_LAUNCH_CHECK();
    int rc;
    if (nc == 1) rc = densemm_nt_launch<W, VEC, 1>(weights, m, k, d, nc, out_bm, (int)b0, st);
    else if (nc <= 8) rc = densemm_nt_launch<W, VEC, 8>(weights, m, k, d, nc, out_bm, (int)b0, st);
    else rc = densemm_nt_launch<W, VEC, 32>(weights, m, k, d, nc, out_bm, (int)b0, st);
    if (rc != BE_OK) return rc;
  }
  be_prof_end(prof, st);
  return BE_OK;
}

// ------------------------------------------------------------------------------------------------
// MFMA path (f16 / bf16 weights, transpose=True, batched):  out[b, n] = sum_k S[b, k] * W[k, n]
//
//   v_mfma_f32_32x32x16_{f16,bf16}:  M = 32 batch rows, N = 32 weight columns, K = 16 weight rows per step.
//   K runs over the *union list* of rows that carry a spike in any batch row (rows without spikes are never
//   read); A (the 0/1 spike tile) is rebuilt in registers from the 16 row masks of the step; B (the weight
//   tile) is loaded row-major with 16-B loads, staged in LDS with a 64-B row pad, and read back through
//   ds_read_b64_tr_b16 — the hardware transpose read — which is conflict-free at that pad
//   (row stride 576 B = 144 dwords = 16 mod 64 banks; the two 16-column groups of a half-wave sit 8 banks apart).
//   Workgroup = 4 waves = 256 columns x one K range; each wave owns 64 columns (two 32x32 accumulators).
//   K is split over gridDim.y parts; f32 partials are reduced in fixed order.
// NOTE: 0 * inf = NaN inside an MFMA, so a non-finite weight in an active row reaches every batch row of its tile;
//       such outputs are found by their non-finite value and redone by selection (be_nonfinite, k_mfma_reduce, nt_repair_rows).
// ------------------------------------------------------------------------------------------------
typedef _Float16 be_v8h __attribute__((ext_vector_type(8)));
typedef __bf16 be_v8bf __attribute__((ext_vector_type(8)));
typedef short be_v8s __attribute__((ext_vector_type(8)));
typedef short be_v4s __attribute__((__vector_size__(4 * sizeof(short))));
typedef float be_v16f __attribute__((ext_vector_type(16)));

#ifndef BE_MFMA_COLS
#define BE_MFMA_COLS 256
#endif
constexpr int kMfmaCols = BE_MFMA_COLS;              // columns per workgroup (256 or 512): a tile row is 512 B / 1 KB of one weight row
constexpr int kMfmaRowBytes = kMfmaCols * 2 + 64;    // LDS row stride (padded)
constexpr int kMfmaTileBytes = 16 * kMfmaRowBytes;   // one K-step tile

template <typename W> struct MfmaOne;
template <> struct MfmaOne<__half> { static constexpr uint32_t one = 0x3C00u; };
template <> struct MfmaOne<__hip_bfloat16> { static constexpr uint32_t one = 0x3F80u; };

template <typename W>
__device__ __forceinline__ be_v16f mfma_32x32x16(be_v8s a, be_v8s b, be_v16f c) {
  if (std::is_same<W, __half>::value)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(be_v8h, a), __builtin_bit_cast(be_v8h, b), c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(be_v8bf, a), __builtin_bit_cast(be_v8bf, b), c, 0, 0, 0);
}

constexpr int kMfmaChunk = 64;   // K-steps whose row ids / masks are staged in LDS at a time

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt(0), i.e. every global
// load in flight — which would serialise the register ring of the MFMA kernel to one HBM round trip per step.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ---- non-finite weights.  Inside an MFMA 0 * inf = NaN, so an inf / NaN weight reaches every batch row of its tile,
// where the reference (brainevent/_dense/binary.py:589-632) only ever ADDS the rows a batch row selects.  Finite weights
// cannot produce a non-finite f32 sum here except by a genuine overflow, so a non-finite result marks the outputs to
// redo: the kernels below recompute exactly those outputs by selection (no products), whatever it costs — the case is rare.
__device__ __forceinline__ bool be_nonfinite(float x) { return (__float_as_uint(x) & 0x7f800000u) == 0x7f800000u; }

// W @ S.T: the outputs of weight row `row` for the nc batch columns of this pass, by the whole wave
template <typename W>
__device__ __forceinline__ void nt_repair_row(const W* __restrict__ weights, int64_t m, int64_t k, int64_t row,
                                              const uint32_t* __restrict__ mask, W* __restrict__ out_bm, int nc, int b0, int lane) {
  for (int b8 = 0; b8 < nc; b8 += 8) {
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int64_t kk = lane; kk < k; kk += 64) {
      const float w = (float)WTraits<W>::load(weights, row * k + kk);
      const uint32_t mk = mask[kk] >> b8;
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if ((mk >> j) & 1u) a[j] += w;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float t = wave_sum(a[j]);
      if (lane == 0 && b8 + j < nc) WTraits<W>::store(out_bm, (int64_t)(b0 + b8 + j) * m + row, t);
    }
  }
}
template <typename W>
__device__ __forceinline__ void nt_repair_rows(const W* __restrict__ weights, int64_t m, int64_t k, int64_t m0, uint32_t rows_bad,
                                               const uint32_t* __restrict__ mask, W* __restrict__ out_bm, int nc, int b0, int lane) {
  while (rows_bad) {                     // (wave-uniform)
    const int rr = __ffs(rows_bad) - 1;
    rows_bad &= rows_bad - 1;
    if (m0 + rr < m) nt_repair_row<W>(weights, m, k, m0 + rr, mask, out_bm, nc, b0, lane);
  }
}

template <typename W>
__global__ void __launch_bounds__(256) k_densemm_mfma(const W* __restrict__ weights, int64_t n,
                                                      const uint32_t* __restrict__ mask, const uint32_t* __restrict__ ulist,
                                                      const uint32_t* __restrict__ ucount, float* __restrict__ partial) {
  __shared__ __align__(16) unsigned char tile[2][kMfmaTileBytes];
#ifndef BE_MFMA_D
#define BE_MFMA_D 2              // (round 4, tools/ab_dense_knobs.sh: 2 / 4 / 6 / 8 steps in flight 0.398 / 0.421 / 0.421 / 0.423 ms per C5
#endif                           //  step, 1.307 / 1.393 ms at 50 % firing — three workgroups per CU already keep 48 KB in flight at 2)
  constexpr int D = BE_MFMA_D;     // K-steps of weight rows in flight per thread (register ring); even
  // D extra all-zero steps behind every chunk: the ring runs past the chunk end without any branch
  __shared__ uint32_t rows_s[(kMfmaChunk + D) * 16];
  __shared__ uint32_t masks_s[(kMfmaChunk + D) * 16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t col0 = (int64_t)blockIdx.x * kMfmaCols;
  const uint32_t n_union = ucount[0];
  const uint32_t steps_total = (n_union + 15u) >> 4;
  const uint32_t per_part = (steps_total + gridDim.y - 1) / gridDim.y;
  const uint32_t t_begin = blockIdx.y * per_part;
  const uint32_t t_end = t_begin + per_part < steps_total ? t_begin + per_part : steps_total;

  // staging role: thread loads 16-B chunk `ch` of tile rows `r0` and `r0 + 8`.  Loads are unconditional
  // (clamped address, result zeroed by a select) so that hipcc keeps counted vmcnt waits across the ring.
  constexpr int CPR = kMfmaCols / 8;                   // 16-byte chunks per tile row
  constexpr int NLD = 16 * CPR / 256;                 // loads per thread and step (2 or 4): tile rows r0, r0 + RSTEP, ...
  constexpr int RSTEP = 256 / CPR;
  constexpr int NB = kMfmaCols / 128;                 // 32-column accumulator blocks per wave
  const int ch = tid % CPR, r0 = tid / CPR;
  const bool col_ok = col0 + (int64_t)ch * 8 < n;     // n % 8 == 0: a chunk is all-in or all-out
  const W* wcol = weights + (col_ok ? col0 + (int64_t)ch * 8 : 0);

  // MFMA role
  const int grp = lane >> 4, gi = lane & 15, q = gi >> 2, pq = gi & 3;
  const int b_row = lane & 31, h = lane >> 5;
  const int tr_off = ((grp >> 1) * 8 + q) * kMfmaRowBytes + (wave * (kMfmaCols / 4) + (grp & 1) * 16 + 4 * pq) * 2;

  be_v16f acc[NB] = {};
  uint4 rr[D][NLD];
  uint32_t rk[D];

  for (uint32_t c0 = t_begin; c0 < t_end; c0 += kMfmaChunk) {
    const uint32_t c_end = c0 + kMfmaChunk < t_end ? c0 + kMfmaChunk : t_end;
    __syncthreads();                                   // previous chunk fully consumed
    for (int j = tid; j < (kMfmaChunk + D) * 16; j += 256) {
      const uint32_t i = c0 * 16u + j;
      const bool v = i < n_union && (c0 + (j >> 4)) < c_end;
      const uint32_t rid = v ? (ulist[i] & 0x0fffffffu) : 0u;
      rows_s[j] = rid;
      masks_s[j] = v ? mask[rid] : 0u;               // mask 0 => the row contributes nothing
    }
    __syncthreads();

    // fetch only issues the loads (nothing may consume a loaded register here, or the wait lands right
    // behind the load); padded rows / columns are turned into exact zeros when the slot is stashed
    auto fetch = [&](uint32_t t, uint4 (&r)[NLD], uint32_t& ok) {     // steps past c_end are zero steps
      const uint32_t s = (t - c0) < (uint32_t)(kMfmaChunk + D - 1) ? (t - c0) : (uint32_t)(kMfmaChunk + D - 1);
      ok = 0u;
#pragma unroll
      for (int u = 0; u < NLD; ++u) {
        r[u] = dense_row_load(wcol + (int64_t)rows_s[s * 16 + r0 + u * RSTEP] * n);
        ok |= (col_ok && masks_s[s * 16 + r0 + u * RSTEP] != 0u ? 1u : 0u) << u;
      }
    };
    auto stash = [&](int buf, const uint4 (&r)[NLD], uint32_t ok) {        // registers -> LDS
#pragma unroll
      for (int u = 0; u < NLD; ++u)
        *reinterpret_cast<uint4*>(&tile[buf][(r0 + u * RSTEP) * kMfmaRowBytes + ch * 16]) = ((ok >> u) & 1u) ? r[u] : make_uint4(0, 0, 0, 0);
    };

    // prologue: step c0 -> LDS[0]; steps c0+1 .. c0+D -> ring slots 1 .. D-1, 0
    fetch(c0, rr[0], rk[0]);
    stash(0, rr[0], rk[0]);
#pragma unroll
    for (int s = 1; s <= D; ++s) fetch(c0 + s, rr[s % D], rk[s % D]);
    lds_barrier();
    for (uint32_t t0 = c0; t0 < c_end; t0 += D) {
#pragma unroll
      for (int ii = 0; ii < D; ++ii) {
        const uint32_t t = t0 + ii;
        {                     // no branch here: steps in [c_end, c_end + D) multiply zeros (see rows_s / masks_s)
          const int buf = ii & 1;            // D is even and chunks start at ring position 0
          const uint32_t* mk = &masks_s[(t - c0) * 16 + 8 * h];
          // A operand: element j of this lane = spike bit of batch row b_row at tile row 8h + j
          be_v8s a;
#pragma unroll
          for (int j = 0; j < 8; ++j) a[j] = (short)(((mk[j] >> b_row) & 1u) * MfmaOne<W>::one);
          // B operands through the hardware transpose read (all 64 lanes active here: EXEC is full)
          const unsigned char* tb = &tile[buf][0] + tr_off;
          typedef __attribute__((address_space(3))) be_v4s* lds_v4s;
#pragma unroll
          for (int j = 0; j < NB; ++j) {
            const be_v4s blo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s)(tb + 64 * j));
            const be_v4s bhi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s)(tb + 64 * j + 4 * kMfmaRowBytes));
            const be_v8s b = {blo[0], blo[1], blo[2], blo[3], bhi[0], bhi[1], bhi[2], bhi[3]};
            acc[j] = mfma_32x32x16<W>(a, b, acc[j]);
          }
          // next step's rows go to the other LDS buffer; their ring slot is refilled D steps ahead
          stash(buf ^ 1, rr[(ii + 1) % D], rk[(ii + 1) % D]);
          fetch(t + 1 + D, rr[(ii + 1) % D], rk[(ii + 1) % D]);
          lds_barrier();
        }
      }
    }
  }
  // C layout: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
  float* pbase = partial + (int64_t)blockIdx.y * 32 * n;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
    const int64_t c0 = col0 + wave * (kMfmaCols / 4) + (lane & 31);
#pragma unroll
    for (int j = 0; j < NB; ++j)
      if (c0 + 32 * j < n) pbase[(int64_t)row * n + c0 + 32 * j] = acc[j][r];
  }
}

// ------------------------------------------------------------------------------------------------
// MFMA path, transpose=False (batched):  out[i, b] = sum_k W[i, k] * S[k, b]
//   A = 32 weight rows x 16 k: exactly the MFMA A-operand layout (lane: row = lane & 31, 8 consecutive k at
//   8 * (lane >> 5)), so a lane loads its fragment with ONE 16-byte global load — no LDS, no transpose.
//   B = the 0/1 spike tile (16 k x 32 batch rows), rebuilt in registers from 16 masks staged in LDS per K chunk.
//   One wave = 32 weight rows over the whole K range, 8 K-steps of weight rows in flight (register ring).
//   The whole matrix is streamed (this direction cannot skip rows), so the bound is k * m * sizeof(W) / HBM.
// ------------------------------------------------------------------------------------------------
#ifndef BE_MFMA_WG_TARGET
#define BE_MFMA_WG_TARGET 768    // workgroups of the S @ W MFMA kernels (column tiles x row parts): 3 per CU.  C5, ms per step:
                                 // 2048: 0.478, 1536 (round 1): 0.466, 1024: 0.449, 768: 0.438, 512: 0.441 (1.48 at 50 % firing), 256: 0.571
                                 // — fewer parts = fewer partial sums to write and reduce, until the chip runs out of waves
#endif
#ifndef BE_NT_MFMA_MIN_NB
#define BE_NT_MFMA_MIN_NB 8
#endif
#ifndef BE_NT_MFMA16
#define BE_NT_MFMA16 1
#endif
constexpr int kNtChunk = 4096;   // k per mask chunk staged in LDS (256 steps)
#ifndef BE_NT_RING
#define BE_NT_RING 8
#endif
constexpr int kNtRing = BE_NT_RING;

template <typename W>
__global__ void __launch_bounds__(256) k_densemm_nt_mfma(const W* __restrict__ weights, int64_t m, int64_t k,
                                                         const uint32_t* __restrict__ mask, W* __restrict__ out_bm, int nc,
                                                         int b0) {
  __shared__ uint32_t masks_s[kNtChunk + 16 * kNtRing];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t m0 = ((int64_t)blockIdx.x * 4 + wave) * 32;
  const int r_lane = lane & 31, h = lane >> 5;
  const int64_t row = m0 + r_lane < m ? m0 + r_lane : m - 1;            // clamped: results of rows >= m are dropped
  const W* wrow = weights + row * k + 8 * h;
  be_v16f acc = {};
  uint4 ring[kNtRing];
  for (int64_t c0 = 0; c0 < k; c0 += kNtChunk) {
    const int64_t clen = k - c0 < kNtChunk ? k - c0 : kNtChunk;          // multiple of 8 (k % 8 == 0)
    const int steps = (int)((clen + 15) >> 4);
    __syncthreads();
    for (int j = tid; j < kNtChunk + 16 * kNtRing; j += 256) masks_s[j] = (j < clen) ? mask[c0 + j] : 0u;
    __syncthreads();
    // fetch only issues the load (address clamped into the row); a half-step past the end of k is zeroed at use
    auto fetch = [&](int t, uint4& a) {
      int64_t kpos = c0 + 16 * (int64_t)t;
      kpos = kpos + 8 * h + 8 <= k ? kpos : (k - 8 - 8 * h > 0 ? k - 8 - 8 * h : 0);
      a = *reinterpret_cast<const uint4*>(wrow + kpos);
    };
#pragma unroll
    for (int s = 0; s < kNtRing; ++s) {
      fetch(s, ring[s]);
      __builtin_amdgcn_sched_barrier(0);      // keep the issue order = the consume order (counted vmcnt needs it)
    }
    for (int t0 = 0; t0 < steps; t0 += kNtRing) {
#pragma unroll
      for (int ii = 0; ii < kNtRing; ++ii) {
        const int t = t0 + ii;            // steps beyond `steps` multiply zero masks (padding of masks_s)
        const uint32_t* mk = &masks_s[16 * t + 8 * h];
        be_v8s bfrag;
#pragma unroll
        for (int j = 0; j < 8; ++j) bfrag[j] = (short)(((mk[j] >> r_lane) & 1u) * MfmaOne<W>::one);
        const bool in_k = c0 + 16 * (int64_t)t + 8 * h + 8 <= k;
        const uint4 av = in_k ? ring[ii] : make_uint4(0, 0, 0, 0);
        const be_v8s afrag = __builtin_bit_cast(be_v8s, av);
        acc = mfma_32x32x16<W>(afrag, bfrag, acc);
        fetch(t + kNtRing, ring[ii]);
        // pin the refill here: hipcc otherwise sinks the loads next to their uses and the ring collapses to depth 2
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  // C layout: col (batch) = lane & 31, row (weight row) = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
  const int b = lane & 31;
  uint32_t rows_bad = 0;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int64_t i = m0 + (r & 3) + 8 * (r >> 2) + 4 * h;
    if (b < nc && i < m) WTraits<W>::store(out_bm, (int64_t)(b0 + b) * m + i, acc[r]);
    const uint64_t bad = __ballot(b < nc && be_nonfinite(acc[r]));
    if (bad & 0xffffffffull) rows_bad |= 1u << ((r & 3) + 8 * (r >> 2));
    if (bad >> 32) rows_bad |= 1u << ((r & 3) + 8 * (r >> 2) + 4);
  }
  if (rows_bad) nt_repair_rows<W>(weights, m, k, m0, rows_bad, mask, out_bm, nc, b0, lane);
}

// The same product on v_mfma_f32_16x16x32_{f16,bf16}: an A fragment is 16 weight rows x 32 k, lane (row i = lane % 16,
// k block kb = lane / 16) loads 16 B at W[row][k0 + 8 kb] — the four lanes of a row read one whole 64-byte sector per
// instruction.  The 32x32x16 shape above reads 32 rows x 32 B per instruction: every sector is requested by two
// instructions, 268 M sector requests for the 8.6 GB of a 65536^2 fp16 matrix in 1.9 ms = 143 G/s, close to what the L2 -> L1
// path delivers (~190 G sectors/s), and the kernel sat at 4.5 TB/s whatever the ring depth.  A wave still owns 32 rows x 32
// batch columns: two A fragments (rows 0-15, 16-31) x two B fragments (columns 0-15, 16-31), four MFMAs per 32 k.
// fp16 65536^2, 32 columns: 1.90-1.97 -> 1.545 ms (5.55 TB/s).  The f32 twin on v_mfma_f32_16x16x4_f32 measured no gain
// (0.89 -> 0.91 ms at 32768^2) and is not kept.
typedef float be_v4f __attribute__((ext_vector_type(4)));
template <typename W>
__device__ __forceinline__ be_v4f mfma_16x16x32(be_v8s a, be_v8s b, be_v4f c) {
  if (std::is_same<W, __half>::value)
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(be_v8h, a), __builtin_bit_cast(be_v8h, b), c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(be_v8bf, a), __builtin_bit_cast(be_v8bf, b), c, 0, 0, 0);
}
template <typename W>
__global__ void __launch_bounds__(256) k_densemm_nt_mfma16(const W* __restrict__ weights, int64_t m, int64_t k,
                                                           const uint32_t* __restrict__ mask, W* __restrict__ out_bm, int nc,
                                                           int b0) {
  __shared__ uint32_t masks_s[kNtChunk + 32 * kNtRing];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t m0 = ((int64_t)blockIdx.x * 4 + wave) * 32;
  const int i_lane = lane & 15, kb = lane >> 4;
  const int64_t row0 = m0 + i_lane < m ? m0 + i_lane : m - 1;           // clamped: results of rows >= m are dropped
  const int64_t row1 = m0 + 16 + i_lane < m ? m0 + 16 + i_lane : m - 1;
  const W* w0 = weights + row0 * k + 8 * kb;
  const W* w1 = weights + row1 * k + 8 * kb;
  be_v4f acc[2][2] = {};
  uint4 ring[kNtRing][2];
  for (int64_t c0 = 0; c0 < k; c0 += kNtChunk) {
    const int64_t clen = k - c0 < kNtChunk ? k - c0 : kNtChunk;          // multiple of 8 (k % 8 == 0)
    const int steps = (int)((clen + 31) >> 5);
    __syncthreads();
    for (int j = tid; j < kNtChunk + 32 * kNtRing; j += 256) masks_s[j] = (j < clen) ? mask[c0 + j] : 0u;
    __syncthreads();
    // fetch only issues the loads (address clamped into the row); a piece past the end of k is zeroed at use
    auto fetch = [&](int t, uint4 (&a)[2]) {
      int64_t kpos = c0 + 32 * (int64_t)t;
      kpos = kpos + 8 * kb + 8 <= k ? kpos : (k - 8 - 8 * kb > 0 ? k - 8 - 8 * kb : 0);
      a[0] = *reinterpret_cast<const uint4*>(w0 + kpos);      // (non-temporal: 1.53 -> 1.75 ms — the two sectors of a 128-byte
      a[1] = *reinterpret_cast<const uint4*>(w1 + kpos);      //  line are read by consecutive steps)
    };
#pragma unroll
    for (int s = 0; s < kNtRing; ++s) {
      fetch(s, ring[s]);
      __builtin_amdgcn_sched_barrier(0);      // keep the issue order = the consume order (counted vmcnt needs it)
    }
    for (int t0 = 0; t0 < steps; t0 += kNtRing) {
#pragma unroll
      for (int ii = 0; ii < kNtRing; ++ii) {
        const int t = t0 + ii;            // steps beyond `steps` multiply zero masks (padding of masks_s)
        const uint32_t* mk = &masks_s[32 * t + 8 * kb];
        be_v8s bf0, bf1;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const uint32_t w = mk[j] >> i_lane;
          bf0[j] = (short)((w & 1u) * MfmaOne<W>::one);
          bf1[j] = (short)(((w >> 16) & 1u) * MfmaOne<W>::one);
        }
        const bool in_k = c0 + 32 * (int64_t)t + 8 * kb + 8 <= k;
        const be_v8s a0 = __builtin_bit_cast(be_v8s, in_k ? ring[ii][0] : make_uint4(0, 0, 0, 0));
        const be_v8s a1 = __builtin_bit_cast(be_v8s, in_k ? ring[ii][1] : make_uint4(0, 0, 0, 0));
        acc[0][0] = mfma_16x16x32<W>(a0, bf0, acc[0][0]);
        acc[0][1] = mfma_16x16x32<W>(a0, bf1, acc[0][1]);
        acc[1][0] = mfma_16x16x32<W>(a1, bf0, acc[1][0]);
        acc[1][1] = mfma_16x16x32<W>(a1, bf1, acc[1][1]);
        fetch(t + kNtRing, ring[ii]);
        // pin the refill here: hipcc otherwise sinks the loads next to their uses and the ring collapses to depth 2
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  // D layout of a 16x16 tile: column (batch) = lane % 16, rows 4 * (lane / 16) + reg
  uint32_t rows_bad = 0;
#pragma unroll
  for (int rg = 0; rg < 2; ++rg)
#pragma unroll
    for (int cg = 0; cg < 2; ++cg) {
      const int b = i_lane + 16 * cg;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t i = m0 + 16 * rg + 4 * kb + r;
        if (b < nc && i < m) WTraits<W>::store(out_bm, (int64_t)(b0 + b) * m + i, acc[rg][cg][r]);
        const uint64_t bad = __ballot(b < nc && be_nonfinite(acc[rg][cg][r]));
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if ((bad >> (16 * q)) & 0xffffull) rows_bad |= 1u << (16 * rg + 4 * q + r);
      }
    }
  if (rows_bad) nt_repair_rows<W>(weights, m, k, m0, rows_bad, mask, out_bm, nc, b0, lane);
}

// The same kernel for f32 weights: v_mfma_f32_32x32x2_f32 (M = 32 weight rows, N = 32 batch rows, K = 2 per instruction).
// A lane (row r, half h) loads 16 B = 4 consecutive k of its row at 4h; instruction j of a step multiplies element j of every
// lane, i.e. k = k0 + j (h = 0) and k0 + 4 + j (h = 1) — the two K slots of an instruction need not be neighbours, only the
// same for A and B — so a step covers 8 k with one load per lane and four MFMAs.  The products are w * 1 or w * 0 in f32:
// exact, and the sums differ from the vector kernel's only in their order.  With 32 batch rows the vector kernel spends 64
// select / add operations per weight and is VALU-bound at 2.4 ms for 32768^2 (1.0 ms with 8 batch rows).
template <int DUMMY = 0>
__global__ void __launch_bounds__(256) k_densemm_nt_mfma_f32(const float* __restrict__ weights, int64_t m, int64_t k,
                                                             const uint32_t* __restrict__ mask, float* __restrict__ out_bm,
                                                             int nc, int b0) {
  __shared__ uint32_t masks_s[kNtChunk + 16 * kNtRing];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t m0 = ((int64_t)blockIdx.x * 4 + wave) * 32;
  const int r_lane = lane & 31, h = lane >> 5;
  const int64_t row = m0 + r_lane < m ? m0 + r_lane : m - 1;            // clamped: results of rows >= m are dropped
  const float* wrow = weights + row * k + 4 * h;
  be_v16f acc = {};
  uint4 ring[kNtRing];
  for (int64_t c0 = 0; c0 < k; c0 += kNtChunk) {
    const int64_t clen = k - c0 < kNtChunk ? k - c0 : kNtChunk;          // multiple of 4 (k % 4 == 0)
    const int steps = (int)((clen + 7) >> 3);
    __syncthreads();
    for (int j = tid; j < kNtChunk + 16 * kNtRing; j += 256) masks_s[j] = (j < clen) ? mask[c0 + j] : 0u;
    __syncthreads();
    auto fetch = [&](int t, uint4& a) {      // only issues the load (address clamped into the row; zeroed at use)
      int64_t kpos = c0 + 8 * (int64_t)t;
      kpos = kpos + 4 * h + 4 <= k ? kpos : (k - 4 - 4 * h > 0 ? k - 4 - 4 * h : 0);
      a = *reinterpret_cast<const uint4*>(wrow + kpos);
    };
#pragma unroll
    for (int s = 0; s < kNtRing; ++s) {
      fetch(s, ring[s]);
      __builtin_amdgcn_sched_barrier(0);      // keep the issue order = the consume order (counted vmcnt needs it)
    }
    for (int t0 = 0; t0 < steps; t0 += kNtRing) {
#pragma unroll
      for (int ii = 0; ii < kNtRing; ++ii) {
        const int t = t0 + ii;            // steps beyond `steps` multiply zero masks (padding of masks_s)
        const uint32_t* mk = &masks_s[8 * t + 4 * h];
        const bool in_k = c0 + 8 * (int64_t)t + 4 * h + 4 <= k;
        const uint4 av = in_k ? ring[ii] : make_uint4(0, 0, 0, 0);
        const float a4[4] = {__uint_as_float(av.x), __uint_as_float(av.y), __uint_as_float(av.z), __uint_as_float(av.w)};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float bj = ((mk[j] >> r_lane) & 1u) ? 1.0f : 0.0f;
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[j], bj, acc, 0, 0, 0);
        }
        fetch(t + kNtRing, ring[ii]);
        __builtin_amdgcn_sched_barrier(0);    // pin the refill here (see k_densemm_nt_mfma)
      }
    }
  }
  // C layout: col (batch) = lane & 31, row (weight row) = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
  const int b = lane & 31;
  uint32_t rows_bad = 0;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int64_t i = m0 + (r & 3) + 8 * (r >> 2) + 4 * h;
    if (b < nc && i < m) out_bm[(int64_t)(b0 + b) * m + i] = acc[r];
    const uint64_t bad = __ballot(b < nc && be_nonfinite(acc[r]));
    if (bad & 0xffffffffull) rows_bad |= 1u << ((r & 3) + 8 * (r >> 2));
    if (bad >> 32) rows_bad |= 1u << ((r & 3) + 8 * (r >> 2) + 4);
  }
  if (rows_bad) nt_repair_rows<float>(weights, m, k, m0, rows_bad, mask, out_bm, nc, b0, lane);
}

// transpose=True for f32 weights on v_mfma_f32_32x32x2_f32: M = 32 batch rows (A = the 0/1 spike tile), N = 32 weight columns,
// K = 2 union rows per instruction (lanes 0-31 the first, 32-63 the second).  A lane loads 16 B = 4 consecutive columns of
// its row at 4 (lane & 31): instruction j of a step takes element j, so a wave owns 128 columns as four 32 x 32 accumulators
// over the column sets {4 i + j}.  No LDS staging of the weights: the B operand wants one column per lane, which is what a
// row-major load gives.  The vector kernel re-reads the matrix once per group of 4 batch rows (32 rows at 50 % firing:
// 3.6 ms for 32768^2, 8 passes); this one reads the union rows once.
constexpr int kTfChunk = 128;     // steps (pairs of union rows) staged in LDS at a time
constexpr int kTfRing = 8;
template <int DUMMY = 0>
__global__ void __launch_bounds__(256) k_densemm_t_mfma_f32(const float* __restrict__ weights, int64_t n,
                                                            const uint32_t* __restrict__ mask, const uint32_t* __restrict__ ulist,
                                                            const uint32_t* __restrict__ ucount, float* __restrict__ partial) {
  __shared__ uint32_t rows_s[2 * (kTfChunk + kTfRing)];
  __shared__ uint32_t masks_s[2 * (kTfChunk + kTfRing)];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c_lane = lane & 31, h = lane >> 5;
  const int64_t col = ((int64_t)blockIdx.x * 4 + wave) * 128 + 4 * c_lane;
  const bool col_ok = col < n;                            // n % 4 == 0: a lane's four columns are all in or all out
  const float* wcol = weights + (col_ok ? col : 0);
  const uint32_t n_union = ucount[0];
  const uint32_t steps_total = (n_union + 1u) >> 1;
  const uint32_t per_part = (steps_total + gridDim.y - 1) / gridDim.y;
  const uint32_t t_begin = blockIdx.y * per_part;
  const uint32_t t_end = t_begin + per_part < steps_total ? t_begin + per_part : steps_total;
  be_v16f acc[4] = {};
  uint4 ring[kTfRing];
  for (uint32_t c0 = t_begin; c0 < t_end; c0 += kTfChunk) {
    const uint32_t c_end = c0 + kTfChunk < t_end ? c0 + kTfChunk : t_end;
    __syncthreads();
    for (int j = tid; j < 2 * (kTfChunk + kTfRing); j += 256) {
      const uint32_t i = 2u * c0 + (uint32_t)j;
      const bool v = i < n_union && (c0 + ((uint32_t)j >> 1)) < c_end;
      const uint32_t rid = v ? (ulist[i] & 0x0fffffffu) : 0u;
      rows_s[j] = rid;
      masks_s[j] = v ? mask[rid] : 0u;                     // mask 0: the row contributes nothing (its weights are zeroed at use)
    }
    __syncthreads();
    auto fetch = [&](uint32_t t, uint4& b) {              // only issues the load; steps past c_end read row 0 with mask 0
      const uint32_t s = (t - c0) < (uint32_t)(kTfChunk + kTfRing - 1) ? (t - c0) : (uint32_t)(kTfChunk + kTfRing - 1);
      b = dense_row_load(wcol + (int64_t)rows_s[2 * s + h] * n);
    };
#pragma unroll
    for (int s = 0; s < kTfRing; ++s) {
      fetch(c0 + s, ring[s]);
      __builtin_amdgcn_sched_barrier(0);
    }
    for (uint32_t t0 = c0; t0 < c_end; t0 += kTfRing) {
#pragma unroll
      for (int ii = 0; ii < kTfRing; ++ii) {
        const uint32_t t = t0 + ii;                       // steps in [c_end, c_end + ring) multiply zeros
        const uint32_t mk = masks_s[2 * (t - c0) + h];
        const float a = ((mk >> c_lane) & 1u) ? 1.0f : 0.0f;
        const bool use = col_ok && mk != 0u;               // (0 * inf = NaN inside an MFMA: rows without a spike must be exact zeros)
        const uint4 bv = use ? ring[ii] : make_uint4(0, 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, __uint_as_float(bv.x), acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, __uint_as_float(bv.y), acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, __uint_as_float(bv.z), acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, __uint_as_float(bv.w), acc[3], 0, 0, 0);
        fetch(t + kTfRing, ring[ii]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  // C layout: N (column set index) = lane & 31, M (batch row) = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
  if (col_ok) {
    float* pbase = partial + (int64_t)blockIdx.y * 32 * n + col;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
      *reinterpret_cast<float4*>(pbase + (int64_t)row * n) = make_float4(acc[0][r], acc[1][r], acc[2][r], acc[3][r]);
    }
  }
}

template <typename W>
__global__ void __launch_bounds__(256) k_mfma_reduce(const float* __restrict__ partial, int parts, int64_t part_stride,
                                                     int64_t total, W* __restrict__ out, const W* __restrict__ weights, int64_t n,
                                                     const uint32_t* __restrict__ mask, const uint32_t* __restrict__ ulist,
                                                     const uint32_t* __restrict__ ucount) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int lane = threadIdx.x & 63;
  // (whole waves stay in the loop: a non-finite sum is redone by its wave together)
  for (int64_t base = (int64_t)blockIdx.x * blockDim.x + (threadIdx.x & ~63); base < total; base += stride) {
    const int64_t i = base + lane;
    const bool in = i < total;
    const int64_t ii = in ? i : total - 1;
    float s = 0.f;
    for (int p0 = 0; p0 < parts; p0 += 8) {      // eight parts' loads in flight; the sum keeps the order of the parts
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = partial[(int64_t)(p0 + u < parts ? p0 + u : parts - 1) * part_stride + ii];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += p0 + u < parts ? v[u] : 0.f;
    }
    uint64_t bad = __ballot(in && be_nonfinite(s));
    while (bad) {                                  // (rare: see be_nonfinite) out[b, c] = sum of the rows batch row b selects
      const int src = __ffsll((unsigned long long)bad) - 1;
      bad &= bad - 1;
      const int64_t e = base + src;
      const int64_t b = e / n, c = e - b * n;
      const uint32_t n_union = ucount[0];
      float a = 0.f;
      for (uint32_t p = (uint32_t)lane; p < n_union; p += 64) {
        const uint32_t r = ulist[p] & 0x0fffffffu;
        if ((mask[r] >> b) & 1u) a += (float)WTraits<W>::load(weights, (int64_t)r * n + c);
      }
      a = wave_sum(a);
      const float fixed = __shfl(a, 0, 64);
      if (lane == src) s = fixed;
    }
    if (in) WTraits<W>::store(out, i, s);
  }
}

inline int mfma_parts(int64_t n) {
  const int64_t tiles = (n + kMfmaCols - 1) / kMfmaCols;
  int64_t p = BE_MFMA_WG_TARGET / (tiles > 0 ? tiles : 1);
  return (int)(p < 1 ? 1 : (p > 16 ? 16 : p));
}

template <typename W>
int densemm_t_mfma(const W* weights, const void* spikes_bm, int sd, W* out_bm, int64_t k, int64_t n, int64_t nb, void* ws,
                   hipStream_t st) {
  DenseWs d = carve(ws, k);
  float* partial = static_cast<float*>(d.partial);     // sized for 16 * nb * n floats >= parts * 32 * n when nb >= 8 ... see ws
  const int64_t nt = n_tiles_of(k);
  uint32_t* ulist = d.lists + (int64_t)kMaxGroups * k;
  uint32_t* utile = d.tile_cnt + (int64_t)kMaxGroups * nt;
  const int parts = std::is_same<W, float>::value ? mfma_parts(n / 2) : mfma_parts(n);      // f32: workgroups of 512 columns
  const int prof = be_prof_begin(st);
  for (int64_t b0 = 0; b0 < nb; b0 += kMaxChunk) {
    const int nc = (int)std::min<int64_t>(kMaxChunk, nb - b0);
    const void* chunk = static_cast<const unsigned char*>(spikes_bm) + (size_t)b0 * spike_row_bytes(sd, k);
    // masks + per-tile counts of the active columns in one launch, scan, ordered union list: three launches (five before:
    // masks, 0/1 flags, count, scan, write — 16 us more per call at C5)
    launch_dense_masks_count(chunk, sd, k, nc, d.mask, utile, nt, st);
    BE_LAUNCH_CHECK();
    if (nt <= 1024) {      // every workgroup of the write sums the tiles in front of it: no scan launch
      hipLaunchKernelGGL((k_gl_write<true, true>), dim3((unsigned)nt, 1), dim3(256), 0, st, d.mask, k, utile, ulist, k,
                         d.count + kMaxGroups);
    } else {
      hipLaunchKernelGGL(k_gl_scan, dim3(1), dim3(1024), 0, st, utile, nt, d.count + kMaxGroups);
      BE_LAUNCH_CHECK();
      hipLaunchKernelGGL((k_gl_write<true, false>), dim3((unsigned)nt, 1), dim3(256), 0, st, d.mask, k, utile, ulist, k,
                         static_cast<uint32_t*>(nullptr));
    }
    BE_LAUNCH_CHECK();
    if constexpr (std::is_same<W, float>::value)
      hipLaunchKernelGGL(k_densemm_t_mfma_f32<0>, dim3((unsigned)((n + 511) / 512), parts), dim3(256), 0, st, weights, n, d.mask,
                         ulist, d.count + kMaxGroups, partial);
    else
      hipLaunchKernelGGL(k_densemm_mfma<W>, dim3((unsigned)((n + kMfmaCols - 1) / kMfmaCols), parts), dim3(256), 0, st, weights,
                         n, d.mask, ulist, d.count + kMaxGroups, partial);
    BE_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_mfma_reduce<W>, dim3(grid_cap((int64_t)nc * n, 256, 2048)), dim3(256), 0, st, partial, parts,
                       (int64_t)32 * n, (int64_t)nc * n, out_bm + b0 * n, weights, n, d.mask, ulist, d.count + kMaxGroups);
    BE_LAUNCH_CHECK();
  }
  be_prof_end(prof, st);
  return BE_OK;
}

template <typename W>
int densemm_nt_mfma(const W* weights, const void* spikes_bm, int sd, W* out_bm, int64_t m, int64_t k, int64_t nb, void* ws,
                    hipStream_t st) {
  DenseWs d = carve(ws, k);
  const int prof = be_prof_begin(st);
  for (int64_t b0 = 0; b0 < nb; b0 += kMaxChunk) {
    const int nc = (int)std::min<int64_t>(kMaxChunk, nb - b0);
    const void* chunk = static_cast<const unsigned char*>(spikes_bm) + (size_t)b0 * spike_row_bytes(sd, k);
    launch_dense_masks(chunk, sd, k, nc, d.mask, st);
    BE_LAUNCH_CHECK();
    if constexpr (std::is_same<W, float>::value)
      hipLaunchKernelGGL(k_densemm_nt_mfma_f32<0>, dim3((unsigned)((m + 127) / 128)), dim3(256), 0, st, weights, m, k, d.mask,
                         out_bm, nc, (int)b0);
    else
      hipLaunchKernelGGL(BE_NT_MFMA16 ? k_densemm_nt_mfma16<W> : k_densemm_nt_mfma<W>, dim3((unsigned)((m + 127) / 128)), dim3(256), 0, st,
                         weights, m, k, d.mask, out_bm, nc, (int)b0);
    BE_LAUNCH_CHECK();
  }
  be_prof_end(prof, st);
  return BE_OK;
}

template <typename W>
int densemm_any(const void* weights, const void* spikes_bm, int sd, void* out_bm, int64_t rows_w, int64_t cols_w,
                int64_t nb, int transpose, void* ws, hipStream_t st) {
  const W* w = static_cast<const W*>(weights);
  W* o = static_cast<W*>(out_bm);
  constexpr int V = Vec16<W>::n;
  const bool vec_ok = (cols_w % V == 0) && ((reinterpret_cast<uintptr_t>(weights) & 15) == 0);
  if (transpose) {
    if constexpr (std::is_same<W, __half>::value || std::is_same<W, __hip_bfloat16>::value || std::is_same<W, float>::value) {
      if (vec_ok && nb >= 8) return densemm_t_mfma<W>(w, spikes_bm, sd, o, rows_w, cols_w, nb, ws, st);
    }
    if (vec_ok) return densemm_t_vec<W, V>(w, spikes_bm, sd, o, rows_w, cols_w, nb, ws, st);
    return densemm_t_vec<W, 1>(w, spikes_bm, sd, o, rows_w, cols_w, nb, ws, st);
  }
  if constexpr (std::is_same<W, __half>::value || std::is_same<W, __hip_bfloat16>::value) {
    // enough rows to fill the chip with 32-row waves; k >= 32 so that the clamped tail load stays inside the row
    if (vec_ok && nb >= BE_NT_MFMA_MIN_NB && rows_w >= 4096 && cols_w >= 32)
      return densemm_nt_mfma<W>(w, spikes_bm, sd, o, rows_w, cols_w, nb, ws, st);
  }
  if constexpr (std::is_same<W, float>::value) {
    // f32, more than 8 batch rows: the vector kernel is VALU-bound there (see k_densemm_nt_mfma_f32)
    if (vec_ok && nb >= 8 && rows_w >= 4096 && cols_w >= 8)
      return densemm_nt_mfma<W>(w, spikes_bm, sd, o, rows_w, cols_w, nb, ws, st);
  }
  if (vec_ok) return densemm_nt_vec<W, V>(w, spikes_bm, sd, o, rows_w, cols_w, nb, ws, st);
  return densemm_nt_vec<W, 1>(w, spikes_bm, sd, o, rows_w, cols_w, nb, ws, st);
}

}  // namespace

extern "C" {

int64_t be_binary_densemm_workspace_bytes(int64_t rows_w, int64_t cols_w, int64_t n_batch, int transpose, int wdtype) {
  return dense_ws_bytes(rows_w, cols_w, n_batch, transpose, wdtype);
}

int be_binary_densemm(const void* weights, int wdtype, const void* spikes_bm, int spike_dtype, void* out_bm,
                      int64_t rows_w, int64_t cols_w, int64_t n_batch, int transpose, void* workspace,
                      int64_t workspace_bytes, be_stream_t stream) {
  BE_REQUIRE(rows_w >= 0 && cols_w >= 0 && n_batch >= 0, BE_ERR_INVALID, "bad shape");
  const int64_t k = transpose ? rows_w : cols_w;
  BE_REQUIRE(k < (1ll << 28), BE_ERR_RANGE, "contraction dimension must be < 2^28");
  const int64_t out_len = transpose ? cols_w : rows_w;
  if (out_len == 0 || n_batch == 0) return BE_OK;
  BE_REQUIRE(out_bm != nullptr, BE_ERR_INVALID, "out is NULL");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const size_t esz = (wdtype == BE_F64) ? 8 : (wdtype == BE_F32 ? 4 : 2);
  if (k == 0) {
    BE_HIP(be_fill_async(out_bm, 0, (size_t)out_len * n_batch * esz, st));
    return BE_OK;
  }
  BE_REQUIRE(weights && spikes_bm, BE_ERR_INVALID, "null pointer");
  BE_REQUIRE(spike_dtype == BE_SPIKE_BOOL || spike_dtype == BE_SPIKE_FLOAT || spike_dtype == BE_SPIKE_BITS, BE_ERR_INVALID,
             "unknown spike dtype");
  BE_REQUIRE(workspace != nullptr && workspace_bytes >= dense_ws_bytes(rows_w, cols_w, n_batch, transpose, wdtype),
             BE_ERR_WORKSPACE, "workspace too small");
  switch (wdtype) {
    case BE_F32: return densemm_any<float>(weights, spikes_bm, spike_dtype, out_bm, rows_w, cols_w, n_batch, transpose, workspace, st);
    case BE_F64: return densemm_any<double>(weights, spikes_bm, spike_dtype, out_bm, rows_w, cols_w, n_batch, transpose, workspace, st);
    case BE_F16: return densemm_any<__half>(weights, spikes_bm, spike_dtype, out_bm, rows_w, cols_w, n_batch, transpose, workspace, st);
    case BE_BF16: return densemm_any<__hip_bfloat16>(weights, spikes_bm, spike_dtype, out_bm, rows_w, cols_w, n_batch, transpose, workspace, st);
    default: be_set_error("be_binary_densemm: unknown weight dtype"); return BE_ERR_INVALID;
  }
}

#define BE_DEF_DENSE_VARIANT(W, WD, S, SD)                                                                              \
  int be_binary_densemv_transpose_##W##_##S(BE_DENSE_MV_ARGS) {                                                          \
    return be_binary_densemm(weights, WD, spikes, SD, out, rows_w, cols_w, 1, 1, workspace, workspace_bytes, stream);    \
  }                                                                                                                      \
  int be_binary_densemv_no_transpose_##W##_##S(BE_DENSE_MV_ARGS) {                                                       \
    return be_binary_densemm(weights, WD, spikes, SD, out, rows_w, cols_w, 1, 0, workspace, workspace_bytes, stream);    \
  }                                                                                                                      \
  int be_binary_densemm_transpose_##W##_##S(BE_DENSE_MM_ARGS) {                                                          \
    return be_binary_densemm(weights, WD, spikes_bm, SD, out_bm, rows_w, cols_w, n_batch, 1, workspace, workspace_bytes, \
                             stream);                                                                                    \
  }                                                                                                                      \
  int be_binary_densemm_no_transpose_##W##_##S(BE_DENSE_MM_ARGS) {                                                       \
    return be_binary_densemm(weights, WD, spikes_bm, SD, out_bm, rows_w, cols_w, n_batch, 0, workspace, workspace_bytes, \
                             stream);                                                                                    \
  }

BE_FOR_ALL_VARIANTS(BE_DEF_DENSE_VARIANT)

}  // extern "C"
