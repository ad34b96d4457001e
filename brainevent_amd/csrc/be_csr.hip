// be_csr.hip — event-driven CSR / fixed-number-connectivity matrix-vector products for gfx950.
//
// Computes what the reference's CPU kernel `_csrmv_numba_kernel` computes
// (reference brainevent/_csr/binary.py:387-489, read as text):
//   transpose=True  : out[:] = 0; for active row i: for j in row i: out[indices[j]] += w[j]
//   transpose=False : out[i] = sum_{j in row i} w[j] * e(spikes[indices[j]])
// with w[j] = weights[0] for homogeneous weights.
//
// Two scatter routes exist because of one measured fact (tools/ubench, MI355X): random-address
// global f32 atomics retire at ~21 G/s chip-wide (they execute memory-side), i.e. ~2 % of what the
// HBM stream of indices+weights could feed.  LDS *integer* atomics retire at ~3-4.7 T/s
// (ds_add_u64 / ds_add_u32), LDS *float* atomics only at ~0.2 T/s.  So:
//   * direct route  : no preprocessing, one wave per active row, global atomics.  Any CSR.
//   * planned route : the matrix is re-laid out once per matrix by output slice
//                     ("post-sliced row segments", uint16 local columns).  One workgroup owns one
//                     slice's accumulators in LDS and streams only the active rows' segments for
//                     that slice; sums are 64-bit fixed point (hetero) or integer counts (homo), so
//                     results are order independent and bitwise reproducible.
#include "be_common.h"
#include <cmath>
#include <type_traits>

namespace {

// =================================================================================================
// spike vector helpers
// =================================================================================================
// spikes -> unordered list of active ids.  Each 256-thread workgroup scans a contiguous tile of
// 256 * kCompactPerThread elements: per-thread activity bits stay in registers, one LDS scan gives the
// offsets and ONE global atomic per workgroup reserves the output range (a returning atomic per wave
// serialises at ~11 ns each on one address: 85 us for 1M spikes at 1 % firing, measured).
template <typename SP, int kCompactPerThread, int kThreads>
__global__ void __launch_bounds__(kThreads) k_compact_spikes(const typename SP::type* __restrict__ spikes, int64_t n,
                                                        uint32_t* __restrict__ active, uint32_t* __restrict__ count,
                                                        int64_t active_stride) {
  spikes += (int64_t)blockIdx.y * n;          // batch-major spike matrix [n_batch, n]
  active += (int64_t)blockIdx.y * active_stride;
  count += blockIdx.y;
  constexpr int kWaves = kThreads / 64;
  __shared__ uint32_t wave_tot[kWaves];
  __shared__ uint32_t block_base;
  const int64_t tile = (int64_t)blockIdx.x * (kThreads * kCompactPerThread);
  const int64_t first = tile + (int64_t)threadIdx.x * kCompactPerThread;
  // activity bits of this thread's kCompactPerThread consecutive elements (16 per word)
  constexpr int NW = kCompactPerThread / 16;
  uint32_t bits[NW];
#pragma unroll
  for (int u = 0; u < NW; ++u) {
    const int64_t f = first + 16 * u;
    uint32_t bw = 0;
    if (sizeof(typename SP::type) == 1 && f + 16 <= n && (reinterpret_cast<uintptr_t>(spikes) & 15) == 0) {
      const uint4 v = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint8_t*>(spikes) + f);
      const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int b = 0; b < 4; ++b) bw |= (((w[q] >> (8 * b)) & 0xffu) != 0u ? 1u : 0u) << (4 * q + b);
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i)
        if (f + i < n && SP::active(spikes[f + i])) bw |= 1u << i;
    }
    bits[u] = bw;
  }
  uint32_t cnt_all = 0;
#pragma unroll
  for (int u = 0; u < NW; ++u) cnt_all += __popc(bits[u]);
  const uint32_t cnt = cnt_all;
  // inclusive scan over the wave, then over the 4 waves
  const int lane = lane_id(), wave = threadIdx.x >> 6;
  uint32_t incl = cnt;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t t = __shfl_up(incl, off, 64);
    if (lane >= off) incl += t;
  }
  if (lane == 63) wave_tot[wave] = incl;
  __syncthreads();
  uint32_t wave_off = 0, total = 0;
#pragma unroll
  for (int w = 0; w < kWaves; ++w) {
    if (w < wave) wave_off += wave_tot[w];
    total += wave_tot[w];
  }
  if (threadIdx.x == 0) block_base = total ? atomicAdd(count, total) : 0u;
  __syncthreads();
  uint32_t pos = block_base + wave_off + incl - cnt;
#pragma unroll
  for (int u = 0; u < NW; ++u) {
    uint32_t bw = bits[u];
    while (bw) {
      const int b = __ffs(bw) - 1;
      bw &= bw - 1;
      active[pos++] = (uint32_t)(first + 16 * u + b);
    }
  }
}

// Compaction straight from a bit-packed spike vector (BE_SPIKE_BITS: bit i%32 of word i/32, rows of
// ceil(n/32) words in a batch).  Same block-aggregated reservation as above; a thread owns kWords words.
template <int kWords>
__global__ void __launch_bounds__(256) k_compact_bits(const uint32_t* __restrict__ words, int64_t n, int64_t n_words,
                                                      uint32_t* __restrict__ active, uint32_t* __restrict__ count,
                                                      int64_t active_stride) {
  words += (int64_t)blockIdx.y * n_words;
  active += (int64_t)blockIdx.y * active_stride;
  count += blockIdx.y;
  __shared__ uint32_t wave_tot[4];
  __shared__ uint32_t block_base;
  const int64_t first_word = ((int64_t)blockIdx.x * 256 + threadIdx.x) * kWords;
  uint32_t bits[kWords];
  uint32_t cnt = 0;
#pragma unroll
  for (int u = 0; u < kWords; ++u) {
    const int64_t w = first_word + u;
    uint32_t bw = (w < n_words) ? words[w] : 0u;
    const int64_t left = n - w * 32;                      // bits of this word that are inside the vector
    if (left < 32) bw &= (left <= 0) ? 0u : ((1u << left) - 1u);
    bits[u] = bw;
    cnt += __popc(bw);
  }
  const int lane = lane_id(), wave = threadIdx.x >> 6;
  uint32_t incl = cnt;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t t = __shfl_up(incl, off, 64);
    if (lane >= off) incl += t;
  }
  if (lane == 63) wave_tot[wave] = incl;
  __syncthreads();
  uint32_t wave_off = 0, total = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    if (w < wave) wave_off += wave_tot[w];
    total += wave_tot[w];
  }
  if (threadIdx.x == 0) block_base = total ? atomicAdd(count, total) : 0u;
  __syncthreads();
  uint32_t pos = block_base + wave_off + incl - cnt;
#pragma unroll
  for (int u = 0; u < kWords; ++u) {
    uint32_t bw = bits[u];
    while (bw) {
      const int b = __ffs(bw) - 1;
      bw &= bw - 1;
      active[pos++] = (uint32_t)((first_word + u) * 32 + b);
    }
  }
}

// bits -> one 0/1 byte per spike (for the consumers that index spikes by position)
__global__ void __launch_bounds__(256) k_unpack_spikes(const uint32_t* __restrict__ words, int64_t n,
                                                       uint8_t* __restrict__ out) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    out[i] = (uint8_t)((words[i >> 5] >> (i & 31)) & 1u);
}

// 1-byte spikes, 16-byte aligned rows: a thread turns 32 bytes (two 16-B loads) into one word
__global__ void __launch_bounds__(256) k_pack_spikes_vec(const uint8_t* __restrict__ spikes, int64_t n, uint32_t* __restrict__ bits,
                                                         int64_t words_stride) {
  spikes += (int64_t)blockIdx.y * n;
  bits += (int64_t)blockIdx.y * words_stride;
  const int64_t n_words = (n + 31) >> 5;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < n_words; w += stride) {
    uint32_t word = 0;
    if (w * 32 + 32 <= n) {
      const uint4 a = reinterpret_cast<const uint4*>(spikes)[2 * w], b = reinterpret_cast<const uint4*>(spikes)[2 * w + 1];
      const uint32_t v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
      for (int q = 0; q < 8; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) word |= (((v[q] >> (8 * e)) & 0xffu) != 0u ? 1u : 0u) << (4 * q + e);
    } else {
      for (int e = 0; e < 32 && w * 32 + e < n; ++e) word |= (spikes[w * 32 + e] != 0 ? 1u : 0u) << e;
    }
    bits[w] = word;
  }
}

template <typename SP>
__global__ void __launch_bounds__(256) k_pack_spikes(const typename SP::type* __restrict__ spikes, int64_t n,
                                                     uint32_t* __restrict__ bits, int64_t words_stride) {
  spikes += (int64_t)blockIdx.y * n;
  bits += (int64_t)blockIdx.y * words_stride;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t n_round = (n + 63) & ~(int64_t)63;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_round; i += stride) {
    const bool a = (i < n) && SP::active(spikes[i]);
    const unsigned long long mask = __ballot(a);
    const int lane = lane_id();
    const int64_t word = i >> 5;   // lane 0 -> low word, lane 32 -> high word
    if (lane == 0) bits[word] = (uint32_t)mask;
    if (lane == 32 && (i < n)) bits[word] = (uint32_t)(mask >> 32);
  }
}

// =================================================================================================
// direct scatter: one wave per active row, global atomics
// =================================================================================================
template <typename W, bool HOMO, typename ACC>
__global__ void __launch_bounds__(256) k_csrmv_t_direct(const W* __restrict__ weights, const int32_t* __restrict__ indices,
                                                        RowPtr rp, const uint32_t* __restrict__ active,
                                                        const uint32_t* __restrict__ n_active_p, ACC* __restrict__ out,
                                                        int64_t active_stride, int64_t k) {
  active += (int64_t)blockIdx.y * active_stride;
  out += (int64_t)blockIdx.y * k;
  const uint32_t n_active = n_active_p[blockIdx.y];
  const int lane = lane_id();
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint32_t n_waves = (gridDim.x * blockDim.x) >> 6;
  ACC w0 = ACC(0);
  if (HOMO) w0 = (ACC)WTraits<W>::load(weights, 0);
  for (uint32_t a = wave; a < n_active; a += n_waves) {
    const int64_t r = active[a];
    const int64_t b = rp.at(r), e = rp.at(r + 1);
    for (int64_t j = b + lane; j < e; j += 64) {
      const ACC w = HOMO ? w0 : (ACC)WTraits<W>::load(weights, j);
      atomicAdd(out + indices[j], w);
    }
  }
}

template <typename W>
__global__ void __launch_bounds__(256) k_convert_from_f32(const float* __restrict__ src, W* __restrict__ dst, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) WTraits<W>::store(dst, i, src[i]);
}

// =================================================================================================
// gather (transpose=False): lanes-per-row groups, bit-packed spikes in LDS when they fit
// =================================================================================================
template <typename W, bool HOMO, int LPR, bool BITS_IN_LDS>
__global__ void __launch_bounds__(256) k_csrmv_nt(const W* __restrict__ weights, const int32_t* __restrict__ indices,
                                                  RowPtr rp, const uint32_t* __restrict__ bits_g, int64_t n_words,
                                                  W* __restrict__ out, int64_t m) {
  bits_g += (int64_t)blockIdx.y * n_words;    // batch-major bitmaps and outputs
  out += (int64_t)blockIdx.y * m;
  extern __shared__ uint32_t bits_s[];
  const uint32_t* bits = bits_g;
  if (BITS_IN_LDS) {
    for (int64_t i = threadIdx.x; i < n_words; i += blockDim.x) bits_s[i] = bits_g[i];
    __syncthreads();
    bits = bits_s;
  }
  using ACC = typename WTraits<W>::acc;
  constexpr int GROUPS = 256 / LPR;
  const int sub = threadIdx.x % LPR;
  const int64_t group = (int64_t)blockIdx.x * GROUPS + threadIdx.x / LPR;
  const int64_t n_groups = (int64_t)gridDim.x * GROUPS;
  ACC w0 = ACC(0);
  if (HOMO) w0 = (ACC)WTraits<W>::load(weights, 0);
  // rows are dealt to groups round-robin; all lanes of a wave run the same number of iterations
  const int64_t m_round = (m + n_groups - 1) / n_groups * n_groups;
  for (int64_t r = group; r < m_round; r += n_groups) {
    ACC acc = ACC(0);
    int cnt = 0;
    if (r < m) {
      const int64_t b = rp.at(r), e = rp.at(r + 1);
      for (int64_t j = b + sub; j < e; j += LPR) {
        const uint32_t c = (uint32_t)indices[j];
        const bool on = (bits[c >> 5] >> (c & 31)) & 1u;
        if (HOMO) cnt += on ? 1 : 0;
        else if (on) acc += (ACC)WTraits<W>::load(weights, j);
      }
    }
    if (HOMO) {
#pragma unroll
      for (int off = LPR / 2; off > 0; off >>= 1) cnt += __shfl_down(cnt, off, LPR);
      acc = (ACC)cnt * w0;
    } else {
#pragma unroll
      for (int off = LPR / 2; off > 0; off >>= 1) acc += __shfl_down(acc, off, LPR);
    }
    if (sub == 0 && r < m) WTraits<W>::store(out, r, acc);
  }
}

// long rows: one wave per row, 16 waves per workgroup (they share the LDS bitmap, which takes most of the LDS, so
// these 16 waves are all the latency hiding a CU gets), 4 consecutive entries per lane per load through raw buffer
// descriptors (range-checked: no tail branches), two iterations (2 KB of indices [+ 2 KB of f32 weights]) in flight.
// f32 weights are streamed unconditionally with the same vector loads — a dependent load per active entry would
// serialise the row at one HBM round trip per hit; other weight dtypes keep the conditional scalar load.
typedef unsigned be_nt_v4u __attribute__((ext_vector_type(4)));

template <typename W, bool HOMO, bool BITS_IN_LDS>
__global__ void __launch_bounds__(1024) k_csrmv_nt_wave(const W* __restrict__ weights, const int32_t* __restrict__ indices,
                                                        RowPtr rp, const uint32_t* __restrict__ bits_g, int64_t n_words,
                                                        W* __restrict__ out, int64_t m) {
  bits_g += (int64_t)blockIdx.y * n_words;
  out += (int64_t)blockIdx.y * m;
  extern __shared__ uint32_t bits_s[];
  const uint32_t* bits = bits_g;
  if (BITS_IN_LDS) {
    for (int64_t i = threadIdx.x; i < n_words; i += blockDim.x) bits_s[i] = bits_g[i];
    __syncthreads();
    bits = bits_s;
  }
  using ACC = typename WTraits<W>::acc;
  constexpr bool VECW = std::is_same<W, float>::value && !HOMO;
  const int lane = lane_id();
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  ACC w0 = ACC(0);
  if (HOMO) w0 = (ACC)WTraits<W>::load(weights, 0);

  // one row: entries [j_begin, len) of the row starting at b, through descriptors of at most 2^30 entries
  auto row_tail = [&](int64_t b, int64_t len, int64_t j_begin, ACC& acc, int& cnt) {
    for (int64_t p0 = (j_begin >> 30) << 30; p0 < len; p0 += (1ll << 30)) {
      const int64_t plen = len - p0 < (1ll << 30) ? len - p0 : (1ll << 30);
      auto ri = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(indices + b + p0), 0, (int)(plen * 4), 0x00020000);
      auto rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<W*>(VECW ? weights + b + p0 : weights), 0,
                                                  VECW ? (int)(plen * 4) : 0, 0x00020000);
      for (int64_t j0 = (p0 == ((j_begin >> 30) << 30)) ? (j_begin - p0) : 0; j0 < plen; j0 += 512) {
        be_nt_v4u c[2], wv[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int off = (int)(j0 + 256 * u + 4 * lane) * 4;
          c[u] = __builtin_amdgcn_raw_buffer_load_b128(ri, off, 0, 0);
          if (VECW) wv[u] = __builtin_amdgcn_raw_buffer_load_b128(rw, off, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int64_t j = j0 + 256 * u + 4 * lane + q;
            if (j < plen) {
              const uint32_t col = c[u][q];
              const bool on = (bits[col >> 5] >> (col & 31)) & 1u;
              if (HOMO) cnt += on ? 1 : 0;
              else if (VECW) acc += on ? (ACC)__uint_as_float(wv[u][q]) : ACC(0);
              else if (on) acc += (ACC)WTraits<W>::load(weights, b + p0 + j);
            }
          }
        }
      }
    }
  };

  // rows of this wave: wave, wave + n_waves, ...; their bounds are fetched 64 at a time (lane l -> l-th next row) and
  // four rows' first 256 entries are in flight together; longer rows continue in row_tail
  for (int64_t t0 = 0;; t0 += 64) {
    const int64_t my_r = wave + n_waves * (t0 + lane);
    const bool valid = my_r < m;
    const unsigned long long vmask = __ballot(valid);
    if (vmask == 0ull) break;
    const int64_t rc = valid ? my_r : 0;
    int64_t rb = rp.at(rc), re = rp.at(rc + 1);
    if (!valid) { rb = 0; re = 0; }
    const uint32_t b_lo = (uint32_t)rb, b_hi = (uint32_t)((uint64_t)rb >> 32);
    const uint64_t rl = (uint64_t)(re - rb);
    const uint32_t l_lo = (uint32_t)rl, l_hi = (uint32_t)(rl >> 32);
    const int nvalid = __popcll(vmask);
    for (int i = 0; i < nvalid; i += 4) {
      int64_t gb[4], gl[4];
      be_nt_v4u c[4], wv[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int src = (i + q) & 63;
        gb[q] = (int64_t)(((uint64_t)__builtin_amdgcn_readlane(b_hi, src) << 32) | __builtin_amdgcn_readlane(b_lo, src));
        gl[q] = (i + q < nvalid)
                    ? (int64_t)(((uint64_t)__builtin_amdgcn_readlane(l_hi, src) << 32) | __builtin_amdgcn_readlane(l_lo, src))
                    : 0;
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int64_t hl = gl[q] < 256 ? gl[q] : 256;       // head: first 256 entries
        auto ri = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(indices + gb[q]), 0, (int)(hl * 4), 0x00020000);
        c[q] = __builtin_amdgcn_raw_buffer_load_b128(ri, lane * 16, 0, 0);
        if (VECW) {
          auto rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<W*>(weights + gb[q]), 0, (int)(hl * 4), 0x00020000);
          wv[q] = __builtin_amdgcn_raw_buffer_load_b128(rw, lane * 16, 0, 0);
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (i + q >= nvalid) break;                           // uniform
        ACC acc = ACC(0);
        int cnt = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int64_t j = 4 * lane + e;
          if (j < gl[q] && j < 256) {
            const uint32_t col = c[q][e];
            const bool on = (bits[col >> 5] >> (col & 31)) & 1u;
            if (HOMO) cnt += on ? 1 : 0;
            else if (VECW) acc += on ? (ACC)__uint_as_float(wv[q][e]) : ACC(0);
            else if (on) acc += (ACC)WTraits<W>::load(weights, gb[q] + j);
          }
        }
        if (gl[q] > 256) row_tail(gb[q], gl[q], 256, acc, cnt);   // uniform
        if (HOMO) {
#pragma unroll
          for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down(cnt, off, 64);
          acc = (ACC)cnt * w0;
        } else {
#pragma unroll
          for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
        }
        if (lane == 0) WTraits<W>::store(out, wave + n_waves * (t0 + i + q), acc);
      }
    }
  }
}

// =================================================================================================
// batched gather fused over the batch (binary_csrmm, transpose=False, long rows): ONE pass over the matrix for up to 32
// spike columns instead of one pass per column.  mask[j] holds the 32 columns' spikes of neuron j as one word (global /
// L2: k words do not fit LDS); a lane keeps its 32 per-column partial sums in private LDS slots and visits only the set
// bits of each entry's mask (at 1 % firing three entries in four have none), so the per-entry cost does not grow with
// the batch.  One wave per row; at the row end the wave folds its 64 x 32 slots and writes out_bm[c, row].
// =================================================================================================
template <typename SP>
__global__ void __launch_bounds__(256) k_batch_masks(const typename SP::type* __restrict__ spikes_bm, int64_t len, int nc,
                                                     uint32_t* __restrict__ mask) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < len; i += stride) {
    uint32_t mk = 0;
    for (int b = 0; b < nc; ++b) mk |= (SP::active(spikes_bm[(int64_t)b * len + i]) ? 1u : 0u) << b;
    mask[i] = mk;
  }
}

constexpr int kFusedSlots = 33;      // 32 columns + 1 pad word: lane l's slots start at bank l

template <typename W, bool HOMO>
__global__ void __launch_bounds__(1024) k_csrmm_nt_fused(const W* __restrict__ weights, const int32_t* __restrict__ indices,
                                                         RowPtr rp, const uint32_t* __restrict__ mask, int nc,
                                                         W* __restrict__ out_bm, int64_t m) {
  extern __shared__ float fused_s[];                 // [1024][kFusedSlots]; uint32 counts when HOMO
  constexpr bool VECW = std::is_same<W, float>::value && !HOMO;
  float* my = fused_s + threadIdx.x * kFusedSlots;
  uint32_t* my_u = reinterpret_cast<uint32_t*>(my);
#pragma unroll
  for (int b = 0; b < 32; ++b) my[b] = 0.0f;
  const int lane = lane_id();
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const float* wave_slots = fused_s + (threadIdx.x & ~63) * kFusedSlots;
  float w0 = 0.0f;
  if (HOMO) w0 = (float)WTraits<W>::load(weights, 0);
  for (int64_t r = wave; r < m; r += n_waves) {
    const int64_t rb = rp.at(r), len = rp.at(r + 1) - rb;
    for (int64_t p0 = 0; p0 < len; p0 += (1ll << 28)) {
      const int64_t plen = len - p0 < (1ll << 28) ? len - p0 : (1ll << 28);
      auto ri = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(indices + rb + p0), 0, (int)(plen * 4), 0x00020000);
      auto rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<W*>(VECW ? weights + rb + p0 : weights), 0,
                                                  VECW ? (int)(plen * 4) : 0, 0x00020000);
      for (int64_t j0 = 0; j0 < plen; j0 += 512) {
        be_nt_v4u c[2], wv[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int off = (int)(j0 + 256 * u + 4 * lane) * 4;
          c[u] = __builtin_amdgcn_raw_buffer_load_b128(ri, off, 0, 0);       // out-of-range lanes read 0
          if (VECW) wv[u] = __builtin_amdgcn_raw_buffer_load_b128(rw, off, 0, 0);
        }
        uint32_t mk[8];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int q = 0; q < 4; ++q) mk[4 * u + q] = mask[c[u][q]];          // unconditional gathers (column 0 for the tail)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int64_t j = j0 + 256 * u + 4 * lane + q;
            uint32_t bits = j < plen ? mk[4 * u + q] : 0u;
            if (bits) {
              float w = 0.0f;
              if (!HOMO) w = VECW ? __uint_as_float(wv[u][q]) : (float)WTraits<W>::load(weights, rb + p0 + j);
              do {
                const int b = __ffs(bits) - 1;
                bits &= bits - 1;
                if (HOMO) my_u[b] += 1u;
                else my[b] += w;
              } while (bits);
            }
          }
        }
      }
    }
    // fold the wave's 64 x 32 slots: lane l sums column (l & 31) over lanes [32 * (l >> 5), +32)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    {
      const int col = lane & 31, half = lane >> 5;
      const float* src = wave_slots + (half * 32) * kFusedSlots + col;
      float sum = 0.0f;
      uint32_t cnt = 0;
#pragma unroll 8
      for (int t = 0; t < 32; ++t) {
        if (HOMO) cnt += reinterpret_cast<const uint32_t*>(src)[t * kFusedSlots];
        else sum += src[t * kFusedSlots];
      }
      if (HOMO) {
        cnt += __shfl_xor(cnt, 32, 64);
        sum = (float)cnt * w0;
      } else {
        sum += __shfl_xor(sum, 32, 64);
      }
      if (lane < nc) WTraits<W>::store(out_bm, (int64_t)lane * m + r, sum);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int b = 0; b < 32; ++b) my[b] = 0.0f;
  }
}

// =================================================================================================
// scatter plan: count -> scan -> fill
//
// Layout ("post-sliced row segments", row-major): for row r and output slice s the entries of row r whose
// column falls in slice s form one *block*, 128-byte aligned, stored as
//     [ f32 weight x 4*n4 ][ uint16 local column x 4*n4 ]        (hetero;  homo: the uint16 part only)
// where n4 = ceil(count / 4) and pads are (local column = 2^slice_shift, weight = 0).
// seg[r * n_slices + s] = { start of the block in 128-B units, n4 }.
// Row-major order keeps the 60-odd blocks of one active row within one ~64 KB window (they are read at about
// the same time by the workgroups of all slices), and 128-B alignment makes a block of b bytes cost
// ceil(b / 128) cache lines instead of ~b/128 + 1.5.
// =================================================================================================
constexpr int kMaxSlices = 4096;   // LDS histogram capacity of the plan kernels

// entries are stored in groups: 4 per group with weights (8 B of columns + 16 B of weights per lane),
// 8 per group without (16 B of columns per lane) — one group is what one lane loads
__host__ __device__ __forceinline__ uint32_t plan_group(bool homo) { return homo ? 8u : 4u; }
__host__ __device__ __forceinline__ uint32_t plan_block_units(uint32_t n_groups, bool homo) {
  const uint32_t bytes = n_groups * (homo ? 16u : 24u);
  return (bytes + 127u) >> 7;
}

// one workgroup per row (grid-stride): per-slice histogram of the row -> seg[r][s] = { block units, n4 }
__global__ void __launch_bounds__(256) k_plan_count(const int32_t* __restrict__ indices, RowPtr rp, int64_t m,
                                                    uint32_t slice_width, int n_slices, int homo, uint2* __restrict__ seg) {
  __shared__ uint32_t hist[kMaxSlices];
  for (int64_t r = blockIdx.x; r < m; r += gridDim.x) {
    for (int s = threadIdx.x; s < n_slices; s += blockDim.x) hist[s] = 0;
    __syncthreads();
    const int64_t b = rp.at(r), e = rp.at(r + 1);
    for (int64_t j = b + threadIdx.x; j < e; j += blockDim.x) atomicAdd(&hist[((uint32_t)indices[j]) / slice_width], 1u);
    __syncthreads();
    for (int s = threadIdx.x; s < n_slices; s += blockDim.x) {
      const uint32_t gsz = plan_group(homo != 0);
      const uint32_t n4 = (hist[s] + gsz - 1u) / gsz;
      seg[r * n_slices + s] = make_uint2(plan_block_units(n4, homo != 0), n4);
    }
    __syncthreads();
  }
}

// three-pass exclusive scan of the .x fields of a uint2 array (sums carried in uint64: overflow is detectable)
constexpr int kScanChunk = 2048;   // elements per workgroup of 256 threads (8 each)

__global__ void __launch_bounds__(256) k_scan_block_sums(const uint2* __restrict__ a, int64_t n, uint64_t* __restrict__ sums) {
  __shared__ uint64_t red[256];
  const int64_t base = (int64_t)blockIdx.x * kScanChunk;
  uint64_t s = 0;
  for (int i = threadIdx.x; i < kScanChunk; i += 256) {
    const int64_t j = base + i;
    if (j < n) s += a[j].x;
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) sums[blockIdx.x] = red[0];
}

// single workgroup: exclusive scan of the block sums in place; sums[n_blocks] = grand total
__global__ void __launch_bounds__(1024) k_scan_sums(uint64_t* __restrict__ sums, int64_t n_blocks) {
  __shared__ uint64_t part[1024];
  __shared__ uint64_t carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int64_t base = 0; base < n_blocks; base += 1024) {
    const int64_t i = base + threadIdx.x;
    const uint64_t v = (i < n_blocks) ? sums[i] : 0;
    part[threadIdx.x] = v;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {   // Hillis-Steele inclusive scan
      uint64_t t = 0;
      if ((int)threadIdx.x >= off) t = part[threadIdx.x - off];
      __syncthreads();
      part[threadIdx.x] += t;
      __syncthreads();
    }
    if (i < n_blocks) sums[i] = carry + part[threadIdx.x] - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry += part[1023];
    __syncthreads();
  }
  if (threadIdx.x == 0) sums[n_blocks] = carry;
}

__global__ void __launch_bounds__(256) k_scan_apply(uint2* __restrict__ a, int64_t n, const uint64_t* __restrict__ sums) {
  __shared__ uint32_t tsum[256];
  const int64_t base = (int64_t)blockIdx.x * kScanChunk + (int64_t)threadIdx.x * 8;
  uint32_t v[8];
  uint32_t s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    v[i] = (base + i < n) ? a[base + i].x : 0u;
    s += v[i];
  }
  tsum[threadIdx.x] = s;
  __syncthreads();
  for (int off = 1; off < 256; off <<= 1) {
    uint32_t t = 0;
    if ((int)threadIdx.x >= off) t = tsum[threadIdx.x - off];
    __syncthreads();
    tsum[threadIdx.x] += t;
    __syncthreads();
  }
  uint32_t run = (uint32_t)sums[blockIdx.x] + tsum[threadIdx.x] - s;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    if (base + i < n) a[base + i].x = run;
    run += v[i];
  }
}

// one workgroup per row (grid-stride): place every entry of the row into its block, then write the pads
template <typename W, bool HOMO>
__global__ void __launch_bounds__(256) k_plan_fill(const W* __restrict__ weights, const int32_t* __restrict__ indices, RowPtr rp,
                                                   int64_t m, int slice_shift, uint32_t slice_width, int n_slices,
                                                   const uint2* __restrict__ seg, unsigned char* __restrict__ blob,
                                                   uint32_t* __restrict__ maxabs_bits) {
  __shared__ uint32_t cur[kMaxSlices];
  __shared__ uint32_t seg_start[kMaxSlices];
  __shared__ uint32_t seg_n4[kMaxSlices];
  uint32_t my_max = 0, my_min = 0xffffffffu;
  for (int64_t r = blockIdx.x; r < m; r += gridDim.x) {
    for (int s = threadIdx.x; s < n_slices; s += blockDim.x) {
      const uint2 sg = seg[r * n_slices + s];
      cur[s] = 0;
      seg_start[s] = sg.x;
      seg_n4[s] = sg.y;
    }
    __syncthreads();
    const int64_t b = rp.at(r), e = rp.at(r + 1);
    for (int64_t j = b + threadIdx.x; j < e; j += blockDim.x) {
      const uint32_t c = (uint32_t)indices[j];
      const uint32_t s = c / slice_width;
      const uint32_t loc = c - s * slice_width;
      const uint32_t rank = atomicAdd(&cur[s], 1u);
      unsigned char* blk = blob + ((int64_t)seg_start[s] << 7);
      if (HOMO) {
        reinterpret_cast<uint16_t*>(blk)[rank] = (uint16_t)loc;
      } else {
        const float w = (float)WTraits<W>::load(weights, j);
        reinterpret_cast<float*>(blk)[rank] = w;
        reinterpret_cast<uint16_t*>(blk + (size_t)seg_n4[s] * 16)[rank] = (uint16_t)loc;
        const uint32_t ab = __float_as_uint(w) & 0x7fffffffu;
        my_max = ab > my_max ? ab : my_max;
        if (ab != 0u) my_min = ab < my_min ? ab : my_min;
      }
    }
    __syncthreads();
    for (int s = threadIdx.x; s < n_slices; s += blockDim.x) {   // pads: dummy slot, zero weight
      unsigned char* blk = blob + ((int64_t)seg_start[s] << 7);
      const uint32_t n = seg_n4[s] * plan_group(HOMO);
      for (uint32_t i = cur[s]; i < n; ++i) {
        if (HOMO) {
          reinterpret_cast<uint16_t*>(blk)[i] = (uint16_t)(1u << slice_shift);
        } else {
          reinterpret_cast<float*>(blk)[i] = 0.f;
          reinterpret_cast<uint16_t*>(blk + (size_t)seg_n4[s] * 16)[i] = (uint16_t)(1u << slice_shift);
        }
      }
    }
    __syncthreads();
  }
  if (!HOMO) {
    // non-negative float bit patterns order like unsigned integers (NaN/Inf sort above every finite value)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const uint32_t o = __shfl_down(my_max, off, 64);
      my_max = o > my_max ? o : my_max;
    }
    if (lane_id() == 0 && my_max != 0) atomicMax(maxabs_bits, my_max);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const uint32_t o = __shfl_down(my_min, off, 64);
      my_min = o < my_min ? o : my_min;
    }
    if (lane_id() == 0 && my_min != 0xffffffffu) atomicMin(maxabs_bits + 1, my_min);   // smallest non-zero |w|
  }
}

// =================================================================================================
// planned scatter step
// =================================================================================================
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

// One group = up to 4 row segments whose first 64 lane-groups are in flight together.
// Loads go through raw buffer descriptors built per segment from wave-uniform (base, length): lanes
// past the end of a segment are range-checked by the hardware (no traffic, zeros returned), so the
// loads need no exec-mask branches and hipcc can keep *counted* vmcnt waits — with conditional
// global loads it falls back to vmcnt(0) before every load and the kernel runs one segment at a time.
typedef unsigned be_v2u __attribute__((ext_vector_type(2)));
typedef unsigned be_v4u __attribute__((ext_vector_type(4)));
constexpr int kBufFlags = 0x00020000;   // raw buffer, 32-bit data format (guide T8)

struct SegGroup {
  uint32_t start[4], n4[4];   // block start (128-B units), number of 4-entry groups
  be_v2u iv[4];
  be_v4u wv[4];
};

template <bool HOMO>
__device__ __forceinline__ void seg_issue(SegGroup& g, int i, int nvalid, uint32_t st_v, uint32_t n4_v, int lane,
                                          const unsigned char* __restrict__ blob) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int src = (i + q) & 63;
    g.start[q] = __builtin_amdgcn_readlane(st_v, src);
    g.n4[q] = (i + q < nvalid) ? __builtin_amdgcn_readlane(n4_v, src) : 0u;
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    // a descriptor addresses < 4 GiB: longer segments are clamped here and finished by the tail loop
    const uint32_t l = g.n4[q] < (1u << 26) ? g.n4[q] : (1u << 26);
    unsigned char* blk = const_cast<unsigned char*>(blob) + ((uint64_t)g.start[q] << 7);
    if (HOMO) {   // 8 uint16 columns per lane; they travel in the wv registers
      auto ri = __builtin_amdgcn_make_buffer_rsrc(blk, 0, (int)(l * 16u), kBufFlags);
      g.wv[q] = __builtin_amdgcn_raw_buffer_load_b128(ri, lane * 16, 0, 0);
    } else {
      auto rw = __builtin_amdgcn_make_buffer_rsrc(blk, 0, (int)(l * 16u), kBufFlags);
      g.wv[q] = __builtin_amdgcn_raw_buffer_load_b128(rw, lane * 16, 0, 0);
      auto ri = __builtin_amdgcn_make_buffer_rsrc(blk + (uint64_t)g.n4[q] * 16u, 0, (int)(l * 8u), kBufFlags);
      g.iv[q] = __builtin_amdgcn_raw_buffer_load_b64(ri, lane * 8, 0, 0);
    }
  }
}

template <bool HOMO>
__device__ __forceinline__ void seg_consume(const SegGroup& g, typename PlanAcc<HOMO>::type* acc, int lane, float scale,
                                            const unsigned char* __restrict__ blob) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if ((uint32_t)lane < g.n4[q]) {
      if (HOMO) {
        plan_count8(reinterpret_cast<uint32_t*>(acc), g.wv[q].x, g.wv[q].y, g.wv[q].z, g.wv[q].w);
      } else {
        const uint2 iv = make_uint2(g.iv[q].x, g.iv[q].y);
        const float4 wv = make_float4(__uint_as_float(g.wv[q].x), __uint_as_float(g.wv[q].y), __uint_as_float(g.wv[q].z),
                                      __uint_as_float(g.wv[q].w));
        plan_add4<HOMO>(acc, iv, wv, scale);
      }
    }
  }
  // long segments (> 256 entries): remaining chunks, wave-uniform guard
  if ((g.n4[0] | g.n4[1] | g.n4[2] | g.n4[3]) > 64u) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const unsigned char* blk = blob + ((uint64_t)g.start[q] << 7);
      for (uint32_t o = 64 + lane; o < g.n4[q]; o += 64) {
        if (HOMO) {
          const uint4 c = reinterpret_cast<const uint4*>(blk)[o];
          plan_count8(reinterpret_cast<uint32_t*>(acc), c.x, c.y, c.z, c.w);
        } else {
          const uint2 ivt = reinterpret_cast<const uint2*>(blk + (uint64_t)g.n4[q] * 16u)[o];
          const float4 wvt = reinterpret_cast<const float4*>(blk)[o];
          plan_add4<HOMO>(acc, ivt, wvt, scale);
        }
      }
    }
  }
}

// Workgroup -> (part, slice).  Blocks b and b + 8 share an XCD (round-robin dispatch, a speed assumption only):
// block b handles linear task L = (b % 8) * (gridDim.x / 8) + b / 8 with part = L / n_slices, slice = L % n_slices,
// so the workgroups of one XCD work on the same part (same active rows) and neighbouring slices: the rows'
// segment-pointer lines (n_slices x 8 B per row, contiguous) are fetched into that XCD's L2 once.
// gridDim.x is a multiple of 8; tasks L >= n_slices * parts are idle.
// Each wave keeps two groups of 4 segments in flight (register double buffer) and prefetches the segment
// pointers of its next 64 rows and the row ids of the 64 after those.
template <bool HOMO>
__global__ void __launch_bounds__(1024) k_plan_accumulate(const unsigned char* __restrict__ blob, const uint2* __restrict__ seg,
                                                          const uint32_t* __restrict__ active,
                                                          const uint32_t* __restrict__ n_active_p, int n_slices,
                                                          int slice_shift, int parts, float scale,
                                                          typename PlanAcc<HOMO>::type* __restrict__ partial,
                                                          int64_t active_stride, int stride) {
  // `stride` = accumulators a task hands to the reduce (slice width rounded up to 16 bytes): the LDS holds 2^slice_shift
  // slots + the pad slot whatever the width, but only the slice's own columns travel through memory
  using acc_t = typename PlanAcc<HOMO>::type;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  acc_t* acc = reinterpret_cast<acc_t*>(smem_raw);
  const int S = 1 << slice_shift;
  const int per_xcd = gridDim.x >> 3;
  const int L = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  const int n_tasks = n_slices * parts;
  if (L >= n_tasks) return;
  const int part = L / n_slices;
  const int slice = L - part * n_slices;
  active += (int64_t)blockIdx.y * active_stride;
  partial += ((int64_t)blockIdx.y * n_tasks + L) * stride;
  {
    uint4* z = reinterpret_cast<uint4*>(smem_raw);
    const int n16 = (int)(((size_t)(S + 1) * sizeof(acc_t) + 15) / 16);
    for (int i = threadIdx.x; i < n16; i += blockDim.x) z[i] = make_uint4(0u, 0u, 0u, 0u);
  }
  __syncthreads();

  const uint32_t n_active = n_active_p[blockIdx.y];
  const uint2* sp = seg + slice;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
  // list position of this lane's row in batch b of this wave
  const uint64_t a0 = (uint64_t)part + (uint64_t)parts * ((uint64_t)wave + (uint64_t)nw * lane);
  const uint64_t a_step = (uint64_t)parts * nw * 64;

  // pointer pipeline: rows of batch b+2 | bounds of batch b+1 | work on batch b.  All pointer loads are
  // unconditional (clamped index, result masked) for the same counted-vmcnt reason as above.
  if (n_active > 0) {
    const uint64_t last = n_active - 1;
    uint64_t a = a0;
    bool v_n = a < n_active;
    uint32_t r_n = active[a < last ? a : last];
    a += a_step;
    uint2 sg = sp[(uint64_t)r_n * n_slices];
    uint32_t st_v = sg.x, n4_v = v_n ? sg.y : 0u;
    bool v_c = v_n;
    v_n = a < n_active;
    r_n = active[a < last ? a : last];
    a += a_step;

    while (__ballot(v_c) != 0ull) {
      const int nvalid = __popcll(__ballot(v_c));   // valid lanes form a prefix: a grows with the lane
      // issue next batch's bounds and the batch-after-next's row ids before touching this batch's data
      const uint2 sgn = sp[(uint64_t)r_n * n_slices];
      const bool v_nn = a < n_active;
      const uint32_t r_nn = active[a < last ? a : last];
      a += a_step;

      SegGroup gA, gB;
      seg_issue<HOMO>(gA, 0, nvalid, st_v, n4_v, lane, blob);
      for (int i = 0; i < nvalid; i += 8) {
        seg_issue<HOMO>(gB, i + 4, nvalid, st_v, n4_v, lane, blob);
        seg_consume<HOMO>(gA, acc, lane, scale, blob);
        seg_issue<HOMO>(gA, i + 8, nvalid, st_v, n4_v, lane, blob);
        seg_consume<HOMO>(gB, acc, lane, scale, blob);
      }
      st_v = sgn.x;
      n4_v = v_n ? sgn.y : 0u;
      v_c = v_n;
      v_n = v_nn;
      r_n = r_nn;
    }
  }
  __syncthreads();
  {
    // stride * sizeof(acc_t) is a multiple of 16
    const uint4* src = reinterpret_cast<const uint4*>(smem_raw);
    uint4* dst = reinterpret_cast<uint4*>(partial);
    const int n16 = (int)((size_t)stride * sizeof(acc_t) / 16);
    for (int i = threadIdx.x; i < n16; i += blockDim.x) dst[i] = src[i];
  }
}

// sum of p[q * pstep], q < parts, with up to eight loads in flight (two per iteration left a 64-part reduce latency-bound;
// a switch on the remainder keeps a 5-part reduce at one round trip too)
template <typename A, int N>
__device__ __forceinline__ A sum_n(const A* __restrict__ b, int64_t pstep) {
  A v[N];
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = b[(int64_t)i * pstep];
  A s = 0;
#pragma unroll
  for (int i = 0; i < N; ++i) s += v[i];
  return s;
}
template <typename A>
__device__ __forceinline__ A sum_parts(const A* __restrict__ p, int parts, int64_t pstep) {
  A s = 0;
  int q = 0;
  for (; q + 8 <= parts; q += 8) s += sum_n<A, 8>(p + (int64_t)q * pstep, pstep);
  const A* b = p + (int64_t)q * pstep;
  switch (parts - q) {      // wave-uniform
    case 1: s += sum_n<A, 1>(b, pstep); break;
    case 2: s += sum_n<A, 2>(b, pstep); break;
    case 3: s += sum_n<A, 3>(b, pstep); break;
    case 4: s += sum_n<A, 4>(b, pstep); break;
    case 5: s += sum_n<A, 5>(b, pstep); break;
    case 6: s += sum_n<A, 6>(b, pstep); break;
    case 7: s += sum_n<A, 7>(b, pstep); break;
    default: break;
  }
  return s;
}

// out[j] = sum over the parts of slice(j); partial is [batch][part][slice][S]
template <typename W, bool HOMO>
__global__ void __launch_bounds__(256) k_plan_reduce(const typename PlanAcc<HOMO>::type* __restrict__ partial, int parts,
                                                     int n_slices, int cap, uint32_t slice_width, int64_t k,
                                                     double inv_scale, const W* __restrict__ weights, W* __restrict__ out,
                                                     int64_t partial_stride, uint32_t* __restrict__ count) {
  if (blockIdx.x == 0 && threadIdx.x == 0) count[blockIdx.y] = 0u;   // re-arm the spike counter for the next call
  partial += (int64_t)blockIdx.y * partial_stride;
  out += (int64_t)blockIdx.y * k;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int S = cap;                     // accumulators per (slice, part) task = stride of its partial sums
  typename WTraits<W>::acc w0 = 0;
  if (HOMO) w0 = WTraits<W>::load(weights, 0);
  const int64_t pstep = (int64_t)n_slices * S;
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < k; j += stride) {
    const int64_t slice = (int64_t)((uint32_t)j / slice_width);       // k <= 2^32 (column ids are int32)
    const int loc = (int)((uint32_t)j - (uint32_t)slice * slice_width);
    const typename PlanAcc<HOMO>::type* p = partial + slice * S + loc;
    if (HOMO) {
      WTraits<W>::store(out, j, (typename WTraits<W>::acc)sum_parts(p, parts, pstep) * w0);
    } else {
      WTraits<W>::store_d(out, j, (double)(long long)sum_parts(p, parts, pstep) * inv_scale);
    }
  }
}

// =================================================================================================
// host-side launch helpers.  Every op takes a batch: spikes are batch-major [n_batch, len], outputs
// batch-major [n_batch, out_len]; the *mv entry points are the n_batch = 1 case, the *mm entry points
// launch the same kernels with gridDim.y = n_batch (one launch per stage for the whole batch — the
// reference loops over the columns on the host, brainevent/_csr/binary_csrmm_hybrid.cu:16-57).
// =================================================================================================
inline int grid_for(int64_t n, int block, int cap) {
  int64_t g = (n + block - 1) / block;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (int)g;
}

constexpr int kMaxBatch = 65535;
constexpr int kFusedMinBatch = 4;      // batched gather: fuse over the batch from this many columns ...
constexpr int64_t kFusedMinRow = 256;  // ... when rows average at least this many entries

inline int64_t counts_bytes(int64_t nb) { return be_align_up(nb * 4, 256); }
inline int64_t active_stride_of(int64_t m) { return be_align_up(m * 4, 256) / 4; }   // in uint32 elements

inline int64_t direct_ws_bytes(int64_t m, int64_t k, int wdtype, int64_t nb) {
  int64_t b = counts_bytes(nb) + nb * active_stride_of(m) * 4;
  if (wdtype == BE_F16 || wdtype == BE_BF16) b += be_align_up(nb * k * 4, 256);
  return b;
}

template <typename SP>
int launch_compact(const void* spikes, int64_t n, int64_t nb, uint32_t* active, int64_t active_stride, uint32_t* count,
                   hipStream_t st, bool zero_first) {
  if (zero_first) BE_HIP(be_fill_async(count, 0, (size_t)nb * 4, st));
  if (n == 0 || nb == 0) return BE_OK;
  // one returning atomic per workgroup serialises at ~11 ns each on one address: keep the number of workgroups
  // per spike vector in the hundreds (4096 elements per workgroup up to 2M spikes, 16384 beyond)
  if (n <= (64ll << 10)) {          // small vectors: more, smaller workgroups
    const int64_t tiles = (n + 256 * 16 - 1) / (256 * 16);
    hipLaunchKernelGGL((k_compact_spikes<SP, 16, 256>), dim3((unsigned)tiles, (unsigned)nb), dim3(256), 0, st,
                       static_cast<const typename SP::type*>(spikes), n, active, count, active_stride);
  } else if (n <= (2ll << 20)) {    // 16384 elements per workgroup: 61 reservations for 1M spikes instead of 244
    const int64_t tiles = (n + 1024 * 16 - 1) / (1024 * 16);
    hipLaunchKernelGGL((k_compact_spikes<SP, 16, 1024>), dim3((unsigned)tiles, (unsigned)nb), dim3(1024), 0, st,
                       static_cast<const typename SP::type*>(spikes), n, active, count, active_stride);
  } else {
    const int64_t tiles = (n + 1024 * 64 - 1) / (1024 * 64);
    hipLaunchKernelGGL((k_compact_spikes<SP, 64, 1024>), dim3((unsigned)tiles, (unsigned)nb), dim3(1024), 0, st,
                       static_cast<const typename SP::type*>(spikes), n, active, count, active_stride);
  }
  BE_LAUNCH_CHECK();
  return BE_OK;
}

int compact_any(const void* spikes, int sd, int64_t n, int64_t nb, uint32_t* active, int64_t active_stride,
                uint32_t* count, hipStream_t st, bool zero_first = true) {
  if (sd == BE_SPIKE_BOOL) return launch_compact<SpikeBool>(spikes, n, nb, active, active_stride, count, st, zero_first);
  if (sd == BE_SPIKE_FLOAT) return launch_compact<SpikeFloat>(spikes, n, nb, active, active_stride, count, st, zero_first);
  if (sd == BE_SPIKE_BITS) {
    if (zero_first) BE_HIP(be_fill_async(count, 0, (size_t)nb * 4, st));
    if (n == 0 || nb == 0) return BE_OK;
    const int64_t n_words = (n + 31) / 32;
    if (n <= (2ll << 20)) {
      hipLaunchKernelGGL((k_compact_bits<1>), dim3((unsigned)((n_words + 255) / 256), (unsigned)nb), dim3(256), 0, st,
                         static_cast<const uint32_t*>(spikes), n, n_words, active, count, active_stride);
    } else {
      hipLaunchKernelGGL((k_compact_bits<4>), dim3((unsigned)((n_words + 1023) / 1024), (unsigned)nb), dim3(256), 0, st,
                         static_cast<const uint32_t*>(spikes), n, n_words, active, count, active_stride);
    }
    BE_LAUNCH_CHECK();
    return BE_OK;
  }
  be_set_error("unknown spike dtype");
  return BE_ERR_INVALID;
}

// Active-row list of a scatter call: the spikes compacted into the workspace, or (BE_SPIKE_IDS, n_batch = 1) the
// caller's own list — `spikes` is then a HOST pointer to a be_spike_ids_t holding two device pointers.
struct ActiveList {
  const uint32_t* ids;
  const uint32_t* count;
};
int resolve_active(const void* spikes, int sd, int64_t n, int64_t nb, uint32_t* ws_active, int64_t astride,
                   uint32_t* ws_count, hipStream_t st, bool zero_first, ActiveList* al) {
  if (sd == BE_SPIKE_IDS) {
    BE_REQUIRE(nb == 1, BE_ERR_UNSUPPORTED, "BE_SPIKE_IDS takes a single event vector (n_batch = 1)");
    const be_spike_ids_t* s = static_cast<const be_spike_ids_t*>(spikes);
    BE_REQUIRE(s->active_ids != nullptr && s->n_active != nullptr, BE_ERR_INVALID, "null id list");
    al->ids = s->active_ids;
    al->count = s->n_active;
    if (zero_first) BE_HIP(be_fill_async(ws_count, 0, 4, st));
    return BE_OK;
  }
  al->ids = ws_active;
  al->count = ws_count;
  return compact_any(spikes, sd, n, nb, ws_active, astride, ws_count, st, zero_first);
}

template <typename SP>
int launch_pack(const void* spikes, int64_t n, int64_t nb, uint32_t* bits, int64_t words_stride, hipStream_t st) {
  if (n == 0 || nb == 0) return BE_OK;
  if (sizeof(typename SP::type) == 1 && (reinterpret_cast<uintptr_t>(spikes) & 15) == 0 && (nb == 1 || (n & 15) == 0)) {
    hipLaunchKernelGGL(k_pack_spikes_vec, dim3(grid_for((n + 31) / 32, 256, 4096), (unsigned)nb), dim3(256), 0, st,
                       static_cast<const uint8_t*>(spikes), n, bits, words_stride);
    BE_LAUNCH_CHECK();
    return BE_OK;
  }
  hipLaunchKernelGGL(k_pack_spikes<SP>, dim3(grid_for(n, 256, 2048), (unsigned)nb), dim3(256), 0, st,
                     static_cast<const typename SP::type*>(spikes), n, bits, words_stride);
  BE_LAUNCH_CHECK();
  return BE_OK;
}

int pack_any(const void* spikes, int sd, int64_t n, int64_t nb, uint32_t* bits, int64_t words_stride, hipStream_t st) {
  if (sd == BE_SPIKE_BOOL) return launch_pack<SpikeBool>(spikes, n, nb, bits, words_stride, st);
  if (sd == BE_SPIKE_FLOAT) return launch_pack<SpikeFloat>(spikes, n, nb, bits, words_stride, st);
  be_set_error("unknown spike dtype");
  return BE_ERR_INVALID;
}

template <typename W, bool HOMO>
int csrmv_t_direct(const void* weights, const int32_t* indices, RowPtr rp, const void* spikes, int sd, void* out,
                   int64_t m, int64_t k, int64_t nb, void* ws, hipStream_t st) {
  unsigned char* wsb = static_cast<unsigned char*>(ws);
  uint32_t* count = reinterpret_cast<uint32_t*>(wsb);
  uint32_t* active = reinterpret_cast<uint32_t*>(wsb + counts_bytes(nb));
  const int64_t astride = active_stride_of(m);
  constexpr bool via_f32 = std::is_same<W, __half>::value || std::is_same<W, __hip_bfloat16>::value;
  using ACC = typename std::conditional<std::is_same<W, double>::value, double, float>::type;
  ACC* acc = via_f32 ? reinterpret_cast<ACC*>(wsb + counts_bytes(nb) + nb * astride * 4) : static_cast<ACC*>(out);
  if (k > 0 && nb > 0) BE_HIP(be_fill_async(acc, 0, (size_t)k * nb * sizeof(ACC), st));
  ActiveList al;
  int rc = resolve_active(spikes, sd, m, nb, active, astride, count, st, true, &al);
  if (rc != BE_OK) return rc;
  if (m > 0 && k > 0 && nb > 0) {
    const int gx = nb >= 8 ? 512 : 2048;
    const int prof = be_prof_begin(st);
    hipLaunchKernelGGL((k_csrmv_t_direct<W, HOMO, ACC>), dim3(gx, (unsigned)nb), dim3(256), 0, st,
                       static_cast<const W*>(weights), indices, rp, al.ids, al.count, acc, astride, k);
    be_prof_end(prof, st);
    BE_LAUNCH_CHECK();
  }
  if (via_f32 && k > 0 && nb > 0) {
    hipLaunchKernelGGL(k_convert_from_f32<W>, dim3(grid_for(k * nb, 256, 2048)), dim3(256), 0, st,
                       reinterpret_cast<const float*>(acc), static_cast<W*>(out), k * nb);
    BE_LAUNCH_CHECK();
  }
  return BE_OK;
}

template <typename W, bool HOMO, int LPR>
int csrmv_nt_launch(const void* weights, const int32_t* indices, RowPtr rp, const uint32_t* bits, int64_t n_words,
                    void* out, int64_t m, int64_t nb, hipStream_t st) {
  constexpr int GROUPS = 256 / LPR;
  const size_t lds = (size_t)n_words * 4;
  const int grid = grid_for(m, GROUPS, nb >= 8 ? 256 : 256 * 8);
  const int prof = be_prof_begin(st);
  if (lds <= 150 * 1024) {
    auto kern = k_csrmv_nt<W, HOMO, LPR, true>;
    BE_HIP(be_allow_lds(reinterpret_cast<const void*>(kern), (int)lds));
    hipLaunchKernelGGL(kern, dim3(grid, (unsigned)nb), dim3(256), lds, st, static_cast<const W*>(weights), indices, rp,
                       bits, n_words, static_cast<W*>(out), m);
  } else {
    hipLaunchKernelGGL((k_csrmv_nt<W, HOMO, LPR, false>), dim3(grid, (unsigned)nb), dim3(256), 0, st,
                       static_cast<const W*>(weights), indices, rp, bits, n_words, static_cast<W*>(out), m);
  }
  be_prof_end(prof, st);
  BE_LAUNCH_CHECK();
  return BE_OK;
}

template <typename W, bool HOMO>
int csrmv_nt(const void* weights, const int32_t* indices, RowPtr rp, int64_t nnz_hint, const void* spikes, int sd,
             void* out, int64_t m, int64_t k, int64_t nb, void* ws, hipStream_t st) {
  const int64_t n_words = (k + 31) / 32;
  if (nb >= kFusedMinBatch && m > 0 && nnz_hint / m >= kFusedMinRow && !std::is_same<W, double>::value &&
      (sd == BE_SPIKE_BOOL || sd == BE_SPIKE_FLOAT)) {
    // long rows, several columns: one pass over the matrix per 32 columns (k_csrmm_nt_fused)
    uint32_t* mask = static_cast<uint32_t*>(ws);
    const size_t lds = (size_t)1024 * kFusedSlots * 4;
    auto kern = k_csrmm_nt_fused<W, HOMO>;
    BE_HIP(be_allow_lds(reinterpret_cast<const void*>(kern), (int)lds));
    const size_t ssz = sd == BE_SPIKE_FLOAT ? 4 : 1;
    for (int64_t c0 = 0; c0 < nb; c0 += 32) {
      const int nc = (int)std::min<int64_t>(32, nb - c0);
      const unsigned char* sp = static_cast<const unsigned char*>(spikes) + (size_t)c0 * k * ssz;
      if (sd == BE_SPIKE_FLOAT)
        hipLaunchKernelGGL(k_batch_masks<SpikeFloat>, dim3(grid_for(k, 256, 2048)), dim3(256), 0, st,
                           reinterpret_cast<const float*>(sp), k, nc, mask);
      else
        hipLaunchKernelGGL(k_batch_masks<SpikeBool>, dim3(grid_for(k, 256, 2048)), dim3(256), 0, st, sp, k, nc, mask);
      BE_LAUNCH_CHECK();
      const int prof = be_prof_begin(st);
      hipLaunchKernelGGL(kern, dim3(grid_for(m, 16, 256)), dim3(1024), lds, st, static_cast<const W*>(weights), indices, rp,
                         mask, nc, static_cast<W*>(out) + c0 * m, m);
      be_prof_end(prof, st);
      BE_LAUNCH_CHECK();
    }
    return BE_OK;
  }
  const uint32_t* bits = static_cast<const uint32_t*>(spikes);   // BE_SPIKE_BITS: already in the kernels' format
  if (sd != BE_SPIKE_BITS) {
    int rc = pack_any(spikes, sd, k, nb, static_cast<uint32_t*>(ws), n_words, st);
    if (rc != BE_OK) return rc;
    bits = static_cast<const uint32_t*>(ws);
  }
  if (m == 0 || nb == 0) return BE_OK;
  const int64_t avg = nnz_hint / (m > 0 ? m : 1);
  if (avg <= 8) return csrmv_nt_launch<W, HOMO, 4>(weights, indices, rp, bits, n_words, out, m, nb, st);
  if (avg <= 48) return csrmv_nt_launch<W, HOMO, 16>(weights, indices, rp, bits, n_words, out, m, nb, st);
  {
    const size_t lds = (size_t)n_words * 4;
    const int grid = grid_for(m, 16, nb >= 8 ? 256 : 512);
    const int prof = be_prof_begin(st);
    if (lds <= 150 * 1024) {
      auto kern = k_csrmv_nt_wave<W, HOMO, true>;
      BE_HIP(be_allow_lds(reinterpret_cast<const void*>(kern), (int)lds));
      hipLaunchKernelGGL(kern, dim3(grid, (unsigned)nb), dim3(1024), lds, st, static_cast<const W*>(weights), indices, rp,
                         bits, n_words, static_cast<W*>(out), m);
    } else {
      hipLaunchKernelGGL((k_csrmv_nt_wave<W, HOMO, false>), dim3(grid, (unsigned)nb), dim3(1024), 0, st,
                         static_cast<const W*>(weights), indices, rp, bits, n_words, static_cast<W*>(out), m);
    }
    be_prof_end(prof, st);
    BE_LAUNCH_CHECK();
    return BE_OK;
  }
}

#define BE_DISPATCH_W(wdtype, HOMO_FLAG, CALL)                                  \
  switch (wdtype) {                                                              \
    case BE_F32:  { using W = float;          if (HOMO_FLAG) { constexpr bool HOMO = true; CALL; } else { constexpr bool HOMO = false; CALL; } } break; \
    case BE_F64:  { using W = double;         if (HOMO_FLAG) { constexpr bool HOMO = true; CALL; } else { constexpr bool HOMO = false; CALL; } } break; \
    case BE_F16:  { using W = __half;         if (HOMO_FLAG) { constexpr bool HOMO = true; CALL; } else { constexpr bool HOMO = false; CALL; } } break; \
    case BE_BF16: { using W = __hip_bfloat16; if (HOMO_FLAG) { constexpr bool HOMO = true; CALL; } else { constexpr bool HOMO = false; CALL; } } break; \
    default: be_set_error("unknown weight dtype"); return BE_ERR_INVALID;       \
  }

inline bool check_rows(const void* indptr, int64_t row_len) { return indptr != nullptr || row_len >= 0; }

// =================================================================================================
// binned route: event-driven scatter for matrices WITHOUT a plan (or whose rows put too few entries into
// one output slice for the plan to pay: FixedNumPerPre K=1000 over 10M outputs has 1.6 entries per (row, slice)).
//
//   pass B (k_bin_rows)   : persistent workgroups take active rows round-robin, fill an LDS batch of <= kBinBatch
//                           entries, counting-sort it by output slice ("bin") in LDS, reserve one range per
//                           (workgroup, bin) in that bin's global region (one returning atomic each) and copy the
//                           runs out coalesced as (uint16 local column, f32 weight).
//   pass C (k_bin_accumulate): one workgroup per (bin, part) streams the bin and accumulates in LDS with integer
//                           atomics exactly like the planned route, then adds its slice to the output.
//   A bin region that overflows its capacity never corrupts anything: that run is delivered with global float
//   atomics instead (slow path, still correct).
// HBM traffic per update (hetero): 8 B read (pass B) + 6 B write + 6 B read = 20 B  vs  8 B algorithmic.
// =================================================================================================
constexpr int kMaxBins = 2048;       // 4 x 4 B x 2048 = 32 KiB of LDS bookkeeping
// entries per LDS batch of (uint16 column [, f32 weight]) payload
template <bool HOMO> struct BinBatch { static constexpr int n = HOMO ? 32768 : 16384; };   // 64 / 96 KiB of payload

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

// one launch instead of three memset nodes: output <- 0, bin cursors <- 0, bin valid extents <- "all of it"
__global__ void __launch_bounds__(256) k_bin_reset(float* __restrict__ out, int64_t k, uint32_t* __restrict__ cursor,
                                                   uint32_t* __restrict__ valid, int n_bins, uint32_t* __restrict__ count) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t == 0) count[0] = 0u;                       // the spike counter of the compaction that follows
  float4* o4 = reinterpret_cast<float4*>(out);
  const int64_t k4 = k >> 2;                       // out comes from the caller's allocator: 16-byte aligned
  for (int64_t i = t; i < k4; i += stride) o4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int64_t i = (k4 << 2) + t; i < k; i += stride) out[i] = 0.f;
  for (int64_t i = t; i < n_bins; i += stride) { cursor[i] = 0u; valid[i] = 0xffffffffu; }
}

// A batch is cut into chunks of 64 consecutive entries of one row piece (one per lane); wave w owns chunks
// w * SLOTS .. w * SLOTS + SLOTS - 1 of the batch and keeps them in REGISTERS from the first read to the placement:
// every load of the batch is in flight at once (16 / 48 independent 256-byte reads per wave — a row is a random
// 0.5 .. 4 KB read, and the first version, four chunks in flight and a second pass over the rows for the placement,
// spent most of a batch waiting for HBM round trips: 28 us per 16384 entries, 2.5 TB/s), and the rows are read once.
template <bool HOMO> struct BinSlots { static constexpr int n = HOMO ? 32 : 16; };     // registers per lane: 32 / 16 + 16

template <typename W, bool HOMO>
__global__ void __launch_bounds__(1024) k_bin_rows(const W* __restrict__ weights, const int32_t* __restrict__ indices, RowPtr rp,
                                                   const uint32_t* __restrict__ active, const uint32_t* __restrict__ n_active_p,
                                                   int slice_shift, int n_bins, uint32_t cap, uint32_t* __restrict__ bin_cursor,
                                                   uint32_t* __restrict__ bin_valid, uint16_t* __restrict__ bin_idx,
                                                   float* __restrict__ bin_w, float* __restrict__ out) {
  constexpr int SLOTS = BinSlots<HOMO>::n;
  constexpr uint32_t kChunks = 16u * SLOTS;            // chunks per batch (16 waves)
  constexpr uint32_t kBatch = kChunks * 64u;           // entries per batch
  static_assert(kBatch == (uint32_t)BinBatch<HOMO>::n, "LDS batch size");
  __shared__ uint32_t hist[kMaxBins], offs[kMaxBins], fill[kMaxBins], gpos[kMaxBins];
  __shared__ uint16_t s_idx[kBatch];
  __shared__ float s_w[HOMO ? 1 : kBatch];
  __shared__ uint32_t s_lens[1024];       // batch: piece length
  __shared__ uint32_t s_cstart[1024];     //        first chunk of the piece
  __shared__ int64_t s_begin[1024];       //        first entry of the piece
  __shared__ uint32_t s_wtot[16];
  __shared__ uint32_t s_nrows, s_nchunks;
  __shared__ uint64_t s_next;             // next list position of this workgroup
  __shared__ int64_t s_carry_begin;       // unfinished tail of a long row
  __shared__ uint32_t s_carry_len;

  const uint32_t n_active = *n_active_p;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
  const uint32_t mask = (1u << slice_shift) - 1u;
  float w0 = 0.f;
  if (HOMO) w0 = (float)WTraits<W>::load(weights, 0);
  if (tid == 0) { s_next = blockIdx.x; s_carry_len = 0; }
  __syncthreads();

  for (;;) {
    // ---- form a batch: thread t looks at this workgroup's t-th next row (loads in parallel), a block scan of
    //      the rows' chunk counts picks the longest prefix that fits one batch; a row longer than a batch is
    //      processed alone, one batch-sized piece at a time (carry)
    for (int b = tid; b < n_bins; b += blockDim.x) { hist[b] = 0; fill[b] = 0; }
    if (s_carry_len) {               // uniform: shared state
      __syncthreads();
      if (tid == 0) {
        const uint32_t take = s_carry_len < kBatch ? s_carry_len : kBatch;
        s_begin[0] = s_carry_begin; s_lens[0] = take; s_cstart[0] = 0;
        s_carry_begin += take; s_carry_len -= take;
        s_nrows = 1; s_nchunks = (take + 63u) >> 6;
      }
      __syncthreads();
    } else {
      const uint64_t a = s_next + (uint64_t)tid * gridDim.x;
      int64_t rb = 0; uint64_t len = 0;
      if (a < n_active) {
        const uint32_t r = active[a];
        rb = rp.at(r);
        len = (uint64_t)(rp.at((int64_t)r + 1) - rb);
      }
      // chunk counts saturate at kChunks + 1 per row; 1024 of them cannot overflow 32 bits
      const uint32_t nch = len > (uint64_t)kBatch ? kChunks + 1u : (uint32_t)((len + 63u) >> 6);
      const uint32_t incl = block_scan_1024(nch, s_wtot);
      const bool in_list = a < n_active;
      const bool fits = in_list && incl <= kChunks;
      const int nfit = __syncthreads_count(fits);          // rows 0 .. nfit-1 (a prefix: the scan is monotone)
      if (fits) { s_begin[tid] = rb; s_lens[tid] = (uint32_t)len; s_cstart[tid] = incl - nch; }
      if (fits && tid == nfit - 1) s_nchunks = incl;
      if (tid == 0) {
        if (nfit > 0) {
          s_nrows = nfit;
          s_next += (uint64_t)nfit * gridDim.x;
        } else if (in_list) {                               // the first row alone exceeds a batch: start carrying it
          s_begin[0] = rb; s_lens[0] = kBatch; s_cstart[0] = 0;
          s_carry_begin = rb + kBatch;
          s_carry_len = (len - kBatch) > 0xffffffffull ? 0xffffffffu : (uint32_t)(len - kBatch);
          s_nrows = 1; s_nchunks = kChunks;
          s_next += gridDim.x;
        } else {
          s_nrows = 0; s_nchunks = 0;
        }
      }
      __syncthreads();
    }
    const uint32_t nrows = s_nrows;
    if (nrows == 0) break;

    // ---- this wave's chunks: lane s < SLOTS finds the piece of chunk wave * SLOTS + s (last piece whose first chunk
    //      is <= the chunk id: empty pieces share their start with the piece that follows them)
    int64_t my_e0 = 0;
    uint32_t my_n = 0;
    {
      const uint32_t cid = (uint32_t)wave * SLOTS + (uint32_t)lane;
      if (lane < SLOTS && cid < s_nchunks) {
        uint32_t lo = 0, hi = nrows;                        // first piece with cstart > cid
        while (lo < hi) {
          const uint32_t mid = (lo + hi) >> 1;
          if (s_cstart[mid] > cid) hi = mid; else lo = mid + 1;
        }
        const uint32_t pc = lo - 1u;
        const uint32_t j0 = (cid - s_cstart[pc]) << 6;
        my_e0 = s_begin[pc] + j0;
        const uint32_t left = s_lens[pc] - j0;
        my_n = left < 64u ? left : 64u;
      }
    }
    // ---- every load of the batch at once (clamped index + predicate instead of conditional loads)
    uint32_t col[SLOTS];
    float wv[HOMO ? 1 : SLOTS];
    uint32_t cnt_mask_lo = 0, cnt_mask_hi = 0;              // bit s: this lane holds an entry in slot s
#pragma unroll
    for (int sl = 0; sl < SLOTS; ++sl) {
      const uint32_t n_s = (uint32_t)__builtin_amdgcn_readlane((int)my_n, sl);
      const uint32_t e_lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(my_e0 & 0xffffffffll), sl);
      const uint32_t e_hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(my_e0 >> 32), sl);
      const int64_t e0 = (int64_t)(((uint64_t)e_hi << 32) | e_lo);
      const uint32_t l = n_s ? ((uint32_t)lane < n_s ? (uint32_t)lane : n_s - 1u) : 0u;
      const int64_t e = n_s ? e0 + l : 0;
      col[sl] = (uint32_t)indices[e];
      if (!HOMO) wv[sl] = (float)WTraits<W>::load(weights, e);
      if ((uint32_t)lane < n_s) { if (sl < 32) cnt_mask_lo |= 1u << (sl & 31); else cnt_mask_hi |= 1u << (sl & 31); }
    }
    // ---- phase 1: histogram of the batch over the bins
#pragma unroll
    for (int sl = 0; sl < SLOTS; ++sl)
      if ((sl < 32 ? cnt_mask_lo : cnt_mask_hi) >> (sl & 31) & 1u) atomicAdd(&hist[col[sl] >> slice_shift], 1u);
    __syncthreads();
    // ---- phase 2: exclusive scan of hist (n_bins <= 2048: two per thread) + one range reservation per bin
    {
      const uint32_t v0 = (2 * tid < n_bins) ? hist[2 * tid] : 0u, v1 = (2 * tid + 1 < n_bins) ? hist[2 * tid + 1] : 0u;
      const uint32_t excl = block_scan_1024(v0 + v1, s_wtot) - (v0 + v1);
      if (2 * tid < n_bins) {
        offs[2 * tid] = excl;
        gpos[2 * tid] = v0 ? atomicAdd(&bin_cursor[2 * tid], v0) : 0u;
      }
      if (2 * tid + 1 < n_bins) {
        offs[2 * tid + 1] = excl + v0;
        gpos[2 * tid + 1] = v1 ? atomicAdd(&bin_cursor[2 * tid + 1], v1) : 0u;
      }
    }
    __syncthreads();
    // ---- phase 3: place the entries into the LDS batch sorted by bin, straight from the registers
#pragma unroll
    for (int sl = 0; sl < SLOTS; ++sl) {
      if ((sl < 32 ? cnt_mask_lo : cnt_mask_hi) >> (sl & 31) & 1u) {
        const uint32_t bin = col[sl] >> slice_shift;
        const uint32_t pos = offs[bin] + atomicAdd(&fill[bin], 1u);
        s_idx[pos] = (uint16_t)(col[sl] & mask);
        if (!HOMO) s_w[pos] = wv[sl];
      }
    }
    __syncthreads();
    // ---- phase 4: copy the runs out, 16 lanes per bin (a run has ~27 entries at C4: one wave per bin spent its time in
    //      the chain of dependent LDS reads per bin, 38 bins per wave: 16 us of a 28 us batch); runs that do not fit
    //      go through global atomics
    {
      // lanes per bin from the expected run length (batch entries / bins): 16 ... 64
      const uint32_t run = kBatch / (uint32_t)n_bins;
      const int lpb_shift = run >= 192u ? 6 : (run >= 64u ? 5 : 4);
      const int LPB = 1 << lpb_shift, BPW = 64 >> lpb_shift;      // lanes per bin, bins per wave and iteration
      const int grp = lane >> lpb_shift, gl = lane & (LPB - 1);
      for (int bin0 = wave * BPW; bin0 < n_bins; bin0 += nw * BPW) {
        const int bin = bin0 + grp;
        uint32_t cnt = 0, o = 0, g = 0;
        if (bin < n_bins) { cnt = hist[bin]; o = offs[bin]; g = gpos[bin]; }
        const bool fits = (uint64_t)g + cnt <= cap;
        // a full bin: everything from position g on is NOT in the bin (later reservations start even higher)
        if (cnt && !fits && gl == 0) atomicMin(&bin_valid[bin], g);
        uint16_t* di = bin_idx + (int64_t)bin * cap + g;
        float* dw = bin_w + (int64_t)bin * cap + g;
        float* dst = out + ((int64_t)bin << slice_shift);
        for (uint32_t j = gl; j < cnt; j += LPB) {
          const uint16_t c = s_idx[o + j];
          const float w = HOMO ? w0 : s_w[o + j];
          if (fits) {
            di[j] = c;
            if (!HOMO) dw[j] = w;
          } else {
            atomicAdd(dst + c, w);
          }
        }
      }
    }
    __syncthreads();
  }
}

template <bool HOMO>
__global__ void __launch_bounds__(1024) k_bin_accumulate(const uint16_t* __restrict__ bin_idx, const float* __restrict__ bin_w,
                                                         const uint32_t* __restrict__ bin_cursor,
                                                         const uint32_t* __restrict__ bin_valid, uint32_t cap, int slice_shift,
                                                         int parts, int64_t k, float scale, double inv_scale,
                                                         const float* __restrict__ w0p, float* __restrict__ out) {
  using acc_t = typename PlanAcc<HOMO>::type;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  acc_t* acc = reinterpret_cast<acc_t*>(smem_raw);
  const int S = 1 << slice_shift;
  const int bin = blockIdx.x / parts, part = blockIdx.x - bin * parts;
  uint32_t cnt = bin_cursor[bin];
  const uint32_t valid = bin_valid[bin];         // first position that was NOT written (cap if the bin never overflowed)
  cnt = cnt < valid ? cnt : valid;
  cnt = cnt < cap ? cnt : cap;
  const float w0 = HOMO ? w0p[0] : 0.f;
  if (cnt == 0) return;                     // nothing was binned here (out already holds zeros / overflow adds)
  for (int i = threadIdx.x; i < S; i += blockDim.x) acc[i] = 0;
  __syncthreads();
  // this part's share, in units of 8 entries
  const uint32_t n8 = (cnt + 7u) >> 3;
  const uint32_t per = (n8 + parts - 1) / parts;
  const uint32_t g_begin = part * per, g_end = g_begin + per < n8 ? g_begin + per : n8;
  const uint16_t* bi = bin_idx + (int64_t)bin * cap;
  const float* bw = bin_w + (int64_t)bin * cap;
  for (uint32_t g = g_begin + threadIdx.x; g < g_end; g += blockDim.x) {
    const uint32_t e0 = g * 8u;
    if (e0 + 8u <= cnt) {        // cap is a multiple of 8: bin regions are 16-byte aligned
      const uint4 iv = *reinterpret_cast<const uint4*>(bi + e0);
      if (HOMO) {
        plan_count8(reinterpret_cast<uint32_t*>(acc), iv.x, iv.y, iv.z, iv.w);
      } else {
        const float4 wa = *reinterpret_cast<const float4*>(bw + e0), wb = *reinterpret_cast<const float4*>(bw + e0 + 4);
        plan_add4<HOMO>(acc, make_uint2(iv.x, iv.y), wa, scale);
        plan_add4<HOMO>(acc, make_uint2(iv.z, iv.w), wb, scale);
      }
    } else {
      for (uint32_t e = e0; e < cnt; ++e) {
        if (HOMO) atomicAdd(reinterpret_cast<uint32_t*>(acc) + bi[e], 1u);
        else atomicAdd(reinterpret_cast<unsigned long long*>(acc) + bi[e], fixed_from_f32(bw[e], scale));
      }
    }
  }
  __syncthreads();
  const int64_t j0 = (int64_t)bin << slice_shift;
  for (int i = threadIdx.x; i < S; i += blockDim.x) {
    if (j0 + i >= k) break;
    float v;
    if (HOMO) v = (float)reinterpret_cast<uint32_t*>(acc)[i] * w0;
    else v = (float)((double)(long long)reinterpret_cast<unsigned long long*>(acc)[i] * inv_scale);
    if (v != 0.f) atomicAdd(out + j0 + i, v);     // contiguous float atomics; one add per output unless parts > 1
  }
}


// =================================================================================================
// "d8" layout of the scatter plan (heterogeneous weights): 5 bytes per entry instead of 6.
//   block(r, s) = [ f32 weight x 4*ng ][ uint8 delta x 4*ng ]   (20 * ng bytes, padded to 128)
//   seg[r][s]   = { block start / 128 B,  ng | (local column of the first entry << 16) }
// The entries of a block are sorted by column; an entry's column is the previous one's plus its delta (the first delta
// is 0).  A gap above 255 is bridged by escape entries (weight 0, delta 255): they add zero to some accumulator of the
// slice, so the step kernel has no special case at all — column = base + inclusive prefix sum of the deltas, add
// weight.  Tail pads are (weight 0, delta 0).  A 1-KB block of the u16 layout becomes ~0.9 KB: one 128-B line less.
// Build: one workgroup sorts a row in LDS (bitonic, 64-bit keys column << 16 | position), rows of at most kD8MaxRow
// entries and at most kD8MaxSlices slices (the caller falls back to the u16 layout otherwise).
// =================================================================================================
constexpr int kD8MaxRow = 16384;
constexpr int kD8MaxSlices = 1024;

__host__ __device__ __forceinline__ uint32_t d8_block_units(uint32_t ng) { return (ng * 20u + 127u) >> 7; }

// loads the row's (column, position) keys into LDS and sorts them; returns the row length
__device__ __forceinline__ int d8_sort_row(unsigned long long* keys, const int32_t* __restrict__ indices, int64_t b, int64_t e) {
  // caller contract: rows have at most kD8MaxRow entries (the Python side checks it and falls back to the u16 layout);
  // a longer row is cut here rather than written past the LDS array
  const int len = (e - b) > (int64_t)kD8MaxRow ? kD8MaxRow : (int)(e - b);
  int n2 = 2;
  while (n2 < len) n2 <<= 1;
  for (int i = threadIdx.x; i < n2; i += blockDim.x)
    keys[i] = i < len ? (((unsigned long long)(uint32_t)indices[b + i] << 16) | (unsigned long long)i) : ~0ull;
  __syncthreads();
  for (int k2 = 2; k2 <= n2; k2 <<= 1) {
    for (int j = k2 >> 1; j > 0; j >>= 1) {
      for (int t = threadIdx.x; t < (n2 >> 1); t += blockDim.x) {
        const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));
        const int hi = lo | j;
        const unsigned long long a = keys[lo], c = keys[hi];
        if ((a > c) == ((lo & k2) == 0)) {
          keys[lo] = c;
          keys[hi] = a;
        }
      }
      __syncthreads();
    }
  }
  return len;
}

// (slice, local column, first-of-block, escapes before it, delta byte) of sorted position i
struct D8Item { uint32_t s, loc, esc, rem; bool first; };
// H8 = the homogeneous-weight variant of the encoding (see "h8" below): code 255 is reserved for the escape itself,
// so a gap is esc x 255 + rem with rem in [0, 254]; d8 keeps rem in [1, 255] for a non-zero gap.
template <bool H8 = false>
__device__ __forceinline__ D8Item d8_item(const unsigned long long* keys, int i, uint32_t W) {
  D8Item it;
  const uint32_t col = (uint32_t)(keys[i] >> 16);
  it.s = col / W;
  it.loc = col - it.s * W;
  uint32_t gap = 0;
  it.first = true;
  if (i > 0) {
    const uint32_t prev = (uint32_t)(keys[i - 1] >> 16);
    if (prev / W == it.s) {
      it.first = false;
      gap = col - prev;
    }
  }
  it.esc = H8 ? gap / 255u : (gap ? (gap - 1u) / 255u : 0u);
  it.rem = gap - 255u * it.esc;
  return it;
}
__host__ __device__ __forceinline__ uint32_t h8_block_units(uint32_t ng) { return (ng * 8u + 127u) >> 7; }

template <bool H8>
__global__ void __launch_bounds__(1024) k_plan_d8_count(const int32_t* __restrict__ indices, RowPtr rp, int64_t m,
                                                        uint32_t slice_width, int n_slices, uint2* __restrict__ seg) {
  extern __shared__ unsigned long long d8_keys[];
  __shared__ uint32_t tot[kD8MaxSlices], base_s[kD8MaxSlices];
  for (int64_t r = blockIdx.x; r < m; r += gridDim.x) {
    for (int s = threadIdx.x; s < n_slices; s += blockDim.x) { tot[s] = 0; base_s[s] = 0; }
    const int len = d8_sort_row(d8_keys, indices, rp.at(r), rp.at(r + 1));   // ends with a barrier
    for (int i = threadIdx.x; i < len; i += blockDim.x) {
      const D8Item it = d8_item<H8>(d8_keys, i, slice_width);
      atomicAdd(&tot[it.s], 1u + it.esc);
      if (it.first) base_s[it.s] = it.loc;
    }
    __syncthreads();
    for (int s = threadIdx.x; s < n_slices; s += blockDim.x) {
      // ng < 2^16: a row has at most kD8MaxRow entries + W/255 escapes
      const uint32_t ng = H8 ? (tot[s] + 7u) >> 3 : (tot[s] + 3u) >> 2;
      seg[r * n_slices + s] = make_uint2(H8 ? h8_block_units(ng) : d8_block_units(ng), ng | (base_s[s] << 16));
    }
    __syncthreads();
  }
}

template <typename W>
__global__ void __launch_bounds__(1024) k_plan_d8_fill(const W* __restrict__ weights, const int32_t* __restrict__ indices,
                                                       RowPtr rp, int64_t m, uint32_t slice_width, int n_slices,
                                                       const uint2* __restrict__ seg, unsigned char* __restrict__ blob,
                                                       uint32_t* __restrict__ maxabs_bits) {
  extern __shared__ unsigned long long d8_keys[];
  __shared__ uint32_t first_idx[kD8MaxSlices], e_first[kD8MaxSlices], seg_start[kD8MaxSlices], seg_ng[kD8MaxSlices],
      tot[kD8MaxSlices];
  __shared__ uint32_t wtot[16];
  uint32_t my_max = 0, my_min = 0xffffffffu;
  for (int64_t r = blockIdx.x; r < m; r += gridDim.x) {
    for (int s = threadIdx.x; s < n_slices; s += blockDim.x) {
      const uint2 sg = seg[r * n_slices + s];
      seg_start[s] = sg.x;
      seg_ng[s] = sg.y & 0xffffu;
      tot[s] = 0;
    }
    const int64_t rb = rp.at(r);
    const int len = d8_sort_row(d8_keys, indices, rb, rp.at(r + 1));
    // inclusive prefix sums of the escape counts over the sorted row: a thread owns `per` consecutive positions
    const int per = (len + 1023) >> 10;
    const int i0 = threadIdx.x * per, i1 = (i0 + per < len) ? i0 + per : len;
    uint32_t mine = 0;
    for (int i = i0; i < i1; ++i) mine += d8_item(d8_keys, i, slice_width).esc;
    const uint32_t excl = block_scan_1024(mine, wtot) - mine;
    uint32_t run = excl;
    for (int i = i0; i < i1; ++i) {
      const D8Item it = d8_item(d8_keys, i, slice_width);
      run += it.esc;
      if (it.first) { first_idx[it.s] = (uint32_t)i; e_first[it.s] = run; }
    }
    __syncthreads();
    run = excl;
    for (int i = i0; i < i1; ++i) {
      const D8Item it = d8_item(d8_keys, i, slice_width);
      run += it.esc;
      const uint32_t pos = ((uint32_t)i - first_idx[it.s]) + (run - e_first[it.s]);
      unsigned char* blk = blob + ((int64_t)seg_start[it.s] << 7);
      float* wp = reinterpret_cast<float*>(blk);
      unsigned char* dp = blk + (size_t)seg_ng[it.s] * 16;
      for (uint32_t q = pos - it.esc; q < pos; ++q) { wp[q] = 0.f; dp[q] = 255; }
      const float w = (float)WTraits<W>::load(weights, rb + (int64_t)(d8_keys[i] & 0xffffull));
      wp[pos] = w;
      dp[pos] = (unsigned char)it.rem;
      atomicMax(&tot[it.s], pos + 1u);
      const uint32_t ab = __float_as_uint(w) & 0x7fffffffu;
      my_max = ab > my_max ? ab : my_max;
      if (ab != 0u) my_min = ab < my_min ? ab : my_min;
    }
    __syncthreads();
    for (int s = threadIdx.x; s < n_slices; s += blockDim.x) {   // tail pads
      unsigned char* blk = blob + ((int64_t)seg_start[s] << 7);
      for (uint32_t q = tot[s]; q < seg_ng[s] * 4u; ++q) {
        reinterpret_cast<float*>(blk)[q] = 0.f;
        (blk + (size_t)seg_ng[s] * 16)[q] = 0;
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const uint32_t o = __shfl_down(my_max, off, 64);
    my_max = o > my_max ? o : my_max;
  }
  if (lane_id() == 0 && my_max != 0) atomicMax(maxabs_bits, my_max);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const uint32_t o = __shfl_down(my_min, off, 64);
    my_min = o < my_min ? o : my_min;
  }
  if (lane_id() == 0) atomicMin(maxabs_bits + 1, my_min);
}

// ---- step kernel for the d8 layout: k_plan_accumulate<false> with the column decode in front of the adds
// wave-wide inclusive prefix sum (DPP row shifts + row broadcasts: 6 VALU adds, no LDS traffic); all 64 lanes active
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v) {
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);   // row_shr:1
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);   // row_shr:2
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);   // row_shr:4
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);   // row_shr:8
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1, 3
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2, 3
  return v;
}

// one lane-group: 4 deltas (one dword) + 4 weights; `before` = column of the entry preceding this lane-group
__device__ __forceinline__ void d8_add4(unsigned long long* acc, uint32_t before, uint32_t d, const be_v4u& wv, float scale) {
  const uint32_t c0 = before + (d & 0xffu), c1 = c0 + ((d >> 8) & 0xffu), c2 = c1 + ((d >> 16) & 0xffu), c3 = c2 + (d >> 24);
  atomicAdd(&acc[c0], fixed_from_f32(__uint_as_float(wv.x), scale));
  atomicAdd(&acc[c1], fixed_from_f32(__uint_as_float(wv.y), scale));
  atomicAdd(&acc[c2], fixed_from_f32(__uint_as_float(wv.z), scale));
  atomicAdd(&acc[c3], fixed_from_f32(__uint_as_float(wv.w), scale));
}
__device__ __forceinline__ uint32_t d8_sum4(uint32_t d) { return (d & 0xffu) + ((d >> 8) & 0xffu) + ((d >> 16) & 0xffu) + (d >> 24); }

struct SegGroupD8 {
  uint32_t start[4], ng[4], base[4];
  uint32_t dv[4];
  be_v4u wv[4];
};

__device__ __forceinline__ void d8_issue(SegGroupD8& g, int i, int nvalid, uint32_t st_v, uint32_t n4_v, int lane,
                                         const unsigned char* __restrict__ blob) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int src = (i + q) & 63;
    g.start[q] = __builtin_amdgcn_readlane(st_v, src);
    const uint32_t y = (i + q < nvalid) ? __builtin_amdgcn_readlane(n4_v, src) : 0u;
    g.ng[q] = y & 0xffffu;
    g.base[q] = y >> 16;
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    unsigned char* blk = const_cast<unsigned char*>(blob) + ((uint64_t)g.start[q] << 7);
    auto rw = __builtin_amdgcn_make_buffer_rsrc(blk, 0, (int)(g.ng[q] * 16u), kBufFlags);
    g.wv[q] = __builtin_amdgcn_raw_buffer_load_b128(rw, lane * 16, 0, 0);
    auto rd = __builtin_amdgcn_make_buffer_rsrc(blk + (uint64_t)g.ng[q] * 16u, 0, (int)(g.ng[q] * 4u), kBufFlags);
    g.dv[q] = __builtin_amdgcn_raw_buffer_load_b32(rd, lane * 4, 0, 0);     // lanes past the block read 0
  }
}

__device__ __forceinline__ void d8_consume(const SegGroupD8& g, unsigned long long* acc, int lane, float scale,
                                           const unsigned char* __restrict__ blob) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    // the prefix sum runs on all 64 lanes (out-of-range lanes hold delta 0: range-checked loads); only the adds are masked
    const uint32_t t = d8_sum4(g.dv[q]);
    const uint32_t before = g.base[q] + wave_incl_scan_u32(t) - t;
    if ((uint32_t)lane < g.ng[q]) d8_add4(acc, before, g.dv[q], g.wv[q], scale);
  }
  if ((g.ng[0] | g.ng[1] | g.ng[2] | g.ng[3]) > 64u) {      // long blocks: remaining chunks of 64 lane-groups
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (g.ng[q] <= 64u) continue;
      const unsigned char* blk = blob + ((uint64_t)g.start[q] << 7);
      const uint32_t t0 = d8_sum4(g.dv[q]);
      uint32_t carry = g.base[q] + __builtin_amdgcn_readlane(wave_incl_scan_u32(t0), 63);
      for (uint32_t o0 = 64; o0 < g.ng[q]; o0 += 64) {
        const uint32_t o = o0 + lane;
        const bool in = o < g.ng[q];
        const uint32_t d = in ? reinterpret_cast<const uint32_t*>(blk + (uint64_t)g.ng[q] * 16u)[o] : 0u;
        be_v4u wv = {0u, 0u, 0u, 0u};
        if (in) {
          const uint4 x = reinterpret_cast<const uint4*>(blk)[o];
          wv = be_v4u{x.x, x.y, x.z, x.w};
        }
        const uint32_t t = d8_sum4(d);
        const uint32_t incl = wave_incl_scan_u32(t);
        if (in) d8_add4(acc, carry + incl - t, d, wv, scale);
        carry += __builtin_amdgcn_readlane(incl, 63);
      }
    }
  }
}

__global__ void __launch_bounds__(1024) k_plan_accumulate_d8(const unsigned char* __restrict__ blob, const uint2* __restrict__ seg,
                                                             const uint32_t* __restrict__ active,
                                                             const uint32_t* __restrict__ n_active_p, int n_slices,
                                                             int cap, int parts, float scale,
                                                             unsigned long long* __restrict__ partial, int64_t active_stride) {
  using acc_t = unsigned long long;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  acc_t* acc = reinterpret_cast<acc_t*>(smem_raw);
  const int S = cap;                     // even; >= slice width (the d8 layout needs no pad slot)
  const int per_xcd = gridDim.x >> 3;
  const int L = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  const int n_tasks = n_slices * parts;
  if (L >= n_tasks) return;
  const int part = L / n_slices;
  const int slice = L - part * n_slices;
  active += (int64_t)blockIdx.y * active_stride;
  partial += ((int64_t)blockIdx.y * n_tasks + L) * S;
  {
    uint4* z = reinterpret_cast<uint4*>(smem_raw);
    const int n16 = (int)(((size_t)(S + 1) * sizeof(acc_t) + 15) / 16);
    for (int i = threadIdx.x; i < n16; i += blockDim.x) z[i] = make_uint4(0u, 0u, 0u, 0u);
  }
  __syncthreads();
  const uint32_t n_active = n_active_p[blockIdx.y];
  const uint2* sp = seg + slice;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
  const uint64_t a0 = (uint64_t)part + (uint64_t)parts * ((uint64_t)wave + (uint64_t)nw * lane);
  const uint64_t a_step = (uint64_t)parts * nw * 64;
  if (n_active > 0) {
    const uint64_t last = n_active - 1;
    uint64_t a = a0;
    bool v_n = a < n_active;
    uint32_t r_n = active[a < last ? a : last];
    a += a_step;
    uint2 sg = sp[(uint64_t)r_n * n_slices];
    uint32_t st_v = sg.x, n4_v = v_n ? sg.y : 0u;
    bool v_c = v_n;
    v_n = a < n_active;
    r_n = active[a < last ? a : last];
    a += a_step;
    while (__ballot(v_c) != 0ull) {
      const int nvalid = __popcll(__ballot(v_c));
      const uint2 sgn = sp[(uint64_t)r_n * n_slices];
      const bool v_nn = a < n_active;
      const uint32_t r_nn = active[a < last ? a : last];
      a += a_step;
      SegGroupD8 gA, gB;
      d8_issue(gA, 0, nvalid, st_v, n4_v, lane, blob);
      for (int i = 0; i < nvalid; i += 8) {
        d8_issue(gB, i + 4, nvalid, st_v, n4_v, lane, blob);
        d8_consume(gA, acc, lane, scale, blob);
        d8_issue(gA, i + 8, nvalid, st_v, n4_v, lane, blob);
        d8_consume(gB, acc, lane, scale, blob);
      }
      st_v = sgn.x;
      n4_v = v_n ? sgn.y : 0u;
      v_c = v_n;
      v_n = v_nn;
      r_n = r_nn;
    }
  }
  __syncthreads();
  {
    const uint4* src = reinterpret_cast<const uint4*>(smem_raw);
    uint4* dst = reinterpret_cast<uint4*>(partial);
    const int n16 = (int)((size_t)S * sizeof(acc_t) / 16);
    for (int i = threadIdx.x; i < n16; i += blockDim.x) dst[i] = src[i];
  }
}

// =================================================================================================
// "h8" layout of the scatter plan (homogeneous weight): 1 byte per entry instead of 2.
//   block(r, s) = [ uint8 code x 8*ng ]   (padded to 128 B);   seg[r][s] = { start / 128 B, ng | first local column << 16 }
// Same sorted-column delta idea as d8, but a counted entry has no weight to zero out, so the escape is its own code:
//   code c < 255: advance c columns and count one entry;   c = 255: advance 255 columns, count nothing (also the tail pad).
// A gap g is g / 255 escapes followed by the code g % 255.  One lane decodes 8 codes (one 8-byte load): a C2 block
// (312 entries + ~25 escapes) is three 128-B lines instead of five.
// =================================================================================================
__global__ void __launch_bounds__(1024) k_plan_h8_fill(const int32_t* __restrict__ indices, RowPtr rp, int64_t m,
                                                       uint32_t slice_width, int n_slices, const uint2* __restrict__ seg,
                                                       unsigned char* __restrict__ blob) {
  extern __shared__ unsigned long long d8_keys[];
  __shared__ uint32_t first_idx[kD8MaxSlices], e_first[kD8MaxSlices], seg_start[kD8MaxSlices], seg_ng[kD8MaxSlices],
      tot[kD8MaxSlices];
  __shared__ uint32_t wtot[16];
  for (int64_t r = blockIdx.x; r < m; r += gridDim.x) {
    for (int s = threadIdx.x; s < n_slices; s += blockDim.x) {
      const uint2 sg = seg[r * n_slices + s];
      seg_start[s] = sg.x;
      seg_ng[s] = sg.y & 0xffffu;
      tot[s] = 0;
    }
    const int len = d8_sort_row(d8_keys, indices, rp.at(r), rp.at(r + 1));
    const int per = (len + 1023) >> 10;
    const int i0 = threadIdx.x * per, i1 = (i0 + per < len) ? i0 + per : len;
    uint32_t mine = 0;
    for (int i = i0; i < i1; ++i) mine += d8_item<true>(d8_keys, i, slice_width).esc;
    const uint32_t excl = block_scan_1024(mine, wtot) - mine;
    uint32_t run = excl;
    for (int i = i0; i < i1; ++i) {
      const D8Item it = d8_item<true>(d8_keys, i, slice_width);
      run += it.esc;
      if (it.first) { first_idx[it.s] = (uint32_t)i; e_first[it.s] = run; }
    }
    __syncthreads();
    run = excl;
    for (int i = i0; i < i1; ++i) {
      const D8Item it = d8_item<true>(d8_keys, i, slice_width);
      run += it.esc;
      const uint32_t pos = ((uint32_t)i - first_idx[it.s]) + (run - e_first[it.s]);
      unsigned char* blk = blob + ((int64_t)seg_start[it.s] << 7);
      for (uint32_t q = pos - it.esc; q < pos; ++q) blk[q] = 255;
      blk[pos] = (unsigned char)it.rem;
      atomicMax(&tot[it.s], pos + 1u);
    }
    __syncthreads();
    for (int s = threadIdx.x; s < n_slices; s += blockDim.x) {   // tail pads
      unsigned char* blk = blob + ((int64_t)seg_start[s] << 7);
      for (uint32_t q = tot[s]; q < seg_ng[s] * 8u; ++q) blk[q] = 255;
    }
    __syncthreads();
  }
}

// one lane-group: 8 codes (two dwords); `before` = column reached before this lane-group
__device__ __forceinline__ void h8_add8(uint32_t* acc, uint32_t before, uint32_t d0, uint32_t d1) {
  uint32_t pos = before;
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    const uint32_t c = ((b < 4 ? d0 : d1) >> (8 * (b & 3))) & 0xffu;
    pos += c;
    if (c != 255u) atomicAdd(&acc[pos], 1u);
  }
}
__device__ __forceinline__ uint32_t h8_sum8(uint32_t d0, uint32_t d1) {
  return __builtin_amdgcn_sad_u8(d1, 0u, __builtin_amdgcn_sad_u8(d0, 0u, 0u));
}

struct SegGroupH8 {
  uint32_t start[4], ng[4], base[4];
  be_v2u dv[4];
};

__device__ __forceinline__ void h8_issue(SegGroupH8& g, int i, int nvalid, uint32_t st_v, uint32_t n4_v, int lane,
                                         const unsigned char* __restrict__ blob) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int src = (i + q) & 63;
    g.start[q] = __builtin_amdgcn_readlane(st_v, src);
    const uint32_t y = (i + q < nvalid) ? __builtin_amdgcn_readlane(n4_v, src) : 0u;
    g.ng[q] = y & 0xffffu;
    g.base[q] = y >> 16;
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    unsigned char* blk = const_cast<unsigned char*>(blob) + ((uint64_t)g.start[q] << 7);
    auto rd = __builtin_amdgcn_make_buffer_rsrc(blk, 0, (int)(g.ng[q] * 8u), kBufFlags);
    g.dv[q] = __builtin_amdgcn_raw_buffer_load_b64(rd, lane * 8, 0, 0);     // lanes past the block read 0
  }
}

__device__ __forceinline__ void h8_consume(const SegGroupH8& g, uint32_t* acc, int lane, const unsigned char* __restrict__ blob) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const uint32_t t = h8_sum8(g.dv[q].x, g.dv[q].y);
    const uint32_t before = g.base[q] + wave_incl_scan_u32(t) - t;
    if ((uint32_t)lane < g.ng[q]) h8_add8(acc, before, g.dv[q].x, g.dv[q].y);
  }
  if ((g.ng[0] | g.ng[1] | g.ng[2] | g.ng[3]) > 64u) {      // long blocks: remaining chunks of 64 lane-groups
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (g.ng[q] <= 64u) continue;
      const uint2* blk = reinterpret_cast<const uint2*>(blob + ((uint64_t)g.start[q] << 7));
      uint32_t carry = g.base[q] + __builtin_amdgcn_readlane(wave_incl_scan_u32(h8_sum8(g.dv[q].x, g.dv[q].y)), 63);
      for (uint32_t o0 = 64; o0 < g.ng[q]; o0 += 64) {
        const uint32_t o = o0 + lane;
        const bool in = o < g.ng[q];
        uint2 d = make_uint2(0u, 0u);
        if (in) d = blk[o];
        const uint32_t t = h8_sum8(d.x, d.y);
        const uint32_t incl = wave_incl_scan_u32(t);
        if (in) h8_add8(acc, carry + incl - t, d.x, d.y);
        carry += __builtin_amdgcn_readlane(incl, 63);
      }
    }
  }
}

__global__ void __launch_bounds__(1024) k_plan_accumulate_h8(const unsigned char* __restrict__ blob, const uint2* __restrict__ seg,
                                                             const uint32_t* __restrict__ active,
                                                             const uint32_t* __restrict__ n_active_p, int n_slices,
                                                             int cap, int parts, uint32_t* __restrict__ partial,
                                                             int64_t active_stride) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  uint32_t* acc = reinterpret_cast<uint32_t*>(smem_raw);
  const int S = cap;                     // multiple of 4; >= slice width
  const int per_xcd = gridDim.x >> 3;
  const int L = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  const int n_tasks = n_slices * parts;
  if (L >= n_tasks) return;
  const int part = L / n_slices;
  const int slice = L - part * n_slices;
  active += (int64_t)blockIdx.y * active_stride;
  partial += ((int64_t)blockIdx.y * n_tasks + L) * S;
  {
    uint4* z = reinterpret_cast<uint4*>(smem_raw);
    const int n16 = (int)((size_t)S * 4 / 16);
    for (int i = threadIdx.x; i < n16; i += blockDim.x) z[i] = make_uint4(0u, 0u, 0u, 0u);
  }
  __syncthreads();
  const uint32_t n_active = n_active_p[blockIdx.y];
  const uint2* sp = seg + slice;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
  const uint64_t a0 = (uint64_t)part + (uint64_t)parts * ((uint64_t)wave + (uint64_t)nw * lane);
  const uint64_t a_step = (uint64_t)parts * nw * 64;
  if (n_active > 0) {
    const uint64_t last = n_active - 1;
    uint64_t a = a0;
    bool v_n = a < n_active;
    uint32_t r_n = active[a < last ? a : last];
    a += a_step;
    uint2 sg = sp[(uint64_t)r_n * n_slices];
    uint32_t st_v = sg.x, n4_v = v_n ? sg.y : 0u;
    bool v_c = v_n;
    v_n = a < n_active;
    r_n = active[a < last ? a : last];
    a += a_step;
    while (__ballot(v_c) != 0ull) {
      const int nvalid = __popcll(__ballot(v_c));
      const uint2 sgn = sp[(uint64_t)r_n * n_slices];
      const bool v_nn = a < n_active;
      const uint32_t r_nn = active[a < last ? a : last];
      a += a_step;
      SegGroupH8 gA, gB;
      h8_issue(gA, 0, nvalid, st_v, n4_v, lane, blob);
      for (int i = 0; i < nvalid; i += 8) {
        h8_issue(gB, i + 4, nvalid, st_v, n4_v, lane, blob);
        h8_consume(gA, acc, lane, blob);
        h8_issue(gA, i + 8, nvalid, st_v, n4_v, lane, blob);
        h8_consume(gB, acc, lane, blob);
      }
      st_v = sgn.x;
      n4_v = v_n ? sgn.y : 0u;
      v_c = v_n;
      v_n = v_nn;
      r_n = r_nn;
    }
  }
  __syncthreads();
  {
    const uint4* src = reinterpret_cast<const uint4*>(smem_raw);
    uint4* dst = reinterpret_cast<uint4*>(partial);
    const int n16 = (int)((size_t)S * 4 / 16);
    for (int i = threadIdx.x; i < n16; i += blockDim.x) dst[i] = src[i];
  }
}

// =================================================================================================
// Small matrices: the whole planned step in ONE launch.  When the output is a single slice and one part is enough
// (k <= 16384 weighted / 32768 counted accumulators, up to ~1M entries), one workgroup compacts the spikes itself (4096
// at a time, into LDS), walks the active rows' blocks, and converts its accumulators straight into the output: no
// active-list round trip through memory, no partial sums, no reduce — 1 launch instead of 3 (a COBA-sized projection:
// 21 -> 13 us of host time per call, and the step is host-bound at that size).
// =================================================================================================
constexpr int kSingleChunk = 4096;       // spikes compacted per pass (4 per thread)

template <int LAYOUT /* 0: u16 weighted, 1: u16 counted, 2: d8, 3: h8 */, typename SP, typename W>
__global__ void __launch_bounds__(1024) k_plan_single(const unsigned char* __restrict__ blob, const uint2* __restrict__ seg,
                                                      const typename SP::type* __restrict__ spikes, int64_t m, int64_t k,
                                                      int cap, float scale, double inv_scale, const W* __restrict__ weights,
                                                      W* __restrict__ out) {
  constexpr bool HOMO = LAYOUT == 1 || LAYOUT == 3;
  using acc_t = typename PlanAcc<HOMO>::type;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  acc_t* acc = reinterpret_cast<acc_t*>(smem_raw);
  __shared__ uint32_t list[kSingleChunk];
  __shared__ uint32_t n_list;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i <= cap; i += 1024) acc[i] = 0;
  for (int64_t base = 0; base < m; base += kSingleChunk) {
    if (tid == 0) n_list = 0;
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int64_t i = base + tid + 1024 * e;
      if (i < m && SP::active(spikes[i])) list[atomicAdd(&n_list, 1u)] = (uint32_t)i;
    }
    __syncthreads();
    const uint32_t n = n_list;
    for (uint32_t a = wave; a < n; a += 16) {
      const uint2 sg = seg[list[a]];                       // one slice: seg[row]
      const unsigned char* blk = blob + ((uint64_t)sg.x << 7);
      if (LAYOUT == 2) {
        const uint32_t ng = sg.y & 0xffffu;
        uint32_t carry = sg.y >> 16;
        for (uint32_t o0 = 0; o0 < ng; o0 += 64) {
          const uint32_t o = o0 + lane;
          const bool in = o < ng;
          const uint32_t d = in ? reinterpret_cast<const uint32_t*>(blk + (uint64_t)ng * 16u)[o] : 0u;
          be_v4u wv = {0u, 0u, 0u, 0u};
          if (in) {
            const uint4 x = reinterpret_cast<const uint4*>(blk)[o];
            wv = be_v4u{x.x, x.y, x.z, x.w};
          }
          const uint32_t t = d8_sum4(d);
          const uint32_t incl = wave_incl_scan_u32(t);
          if (in) d8_add4(reinterpret_cast<unsigned long long*>(acc), carry + incl - t, d, wv, scale);
          carry += __builtin_amdgcn_readlane(incl, 63);
        }
      } else if (LAYOUT == 3) {
        const uint32_t ng = sg.y & 0xffffu;
        uint32_t carry = sg.y >> 16;
        for (uint32_t o0 = 0; o0 < ng; o0 += 64) {
          const uint32_t o = o0 + lane;
          const bool in = o < ng;
          uint2 d = make_uint2(0u, 0u);
          if (in) d = reinterpret_cast<const uint2*>(blk)[o];
          const uint32_t t = h8_sum8(d.x, d.y);
          const uint32_t incl = wave_incl_scan_u32(t);
          if (in) h8_add8(reinterpret_cast<uint32_t*>(acc), carry + incl - t, d.x, d.y);
          carry += __builtin_amdgcn_readlane(incl, 63);
        }
      } else {
        const uint32_t ng = sg.y;
        for (uint32_t o = lane; o < ng; o += 64) {
          if (HOMO) {
            const uint4 c = reinterpret_cast<const uint4*>(blk)[o];
            plan_count8(reinterpret_cast<uint32_t*>(acc), c.x, c.y, c.z, c.w);
          } else {
            const uint2 iv = reinterpret_cast<const uint2*>(blk + (uint64_t)ng * 16u)[o];
            const float4 wv = reinterpret_cast<const float4*>(blk)[o];
            plan_add4<false>(reinterpret_cast<unsigned long long*>(acc), iv, wv, scale);
          }
        }
      }
    }
    __syncthreads();
  }
  typename WTraits<W>::acc w0 = 0;
  if (HOMO) w0 = WTraits<W>::load(weights, 0);
  for (int64_t j = tid; j < k; j += 1024) {
    if (HOMO) WTraits<W>::store(out, j, (typename WTraits<W>::acc)(uint32_t)acc[j] * w0);
    else WTraits<W>::store_d(out, j, (double)(long long)acc[j] * inv_scale);
  }
}

template <int LAYOUT, typename W>
int launch_plan_single(const void* blob, const void* seg, const void* spikes, int sd, int64_t m, int64_t k, int cap,
                       float scale, double inv_scale, const void* weights, void* out, size_t lds, hipStream_t st) {
  if (sd == BE_SPIKE_FLOAT) {
    auto kern = k_plan_single<LAYOUT, SpikeFloat, W>;
    BE_HIP(be_allow_lds(reinterpret_cast<const void*>(kern), (int)lds));
    hipLaunchKernelGGL(kern, dim3(1), dim3(1024), lds, st, static_cast<const unsigned char*>(blob), static_cast<const uint2*>(seg),
                       static_cast<const float*>(spikes), m, k, cap, scale, inv_scale, static_cast<const W*>(weights),
                       static_cast<W*>(out));
  } else {
    auto kern = k_plan_single<LAYOUT, SpikeBool, W>;
    BE_HIP(be_allow_lds(reinterpret_cast<const void*>(kern), (int)lds));
    hipLaunchKernelGGL(kern, dim3(1), dim3(1024), lds, st, static_cast<const unsigned char*>(blob), static_cast<const uint2*>(seg),
                       static_cast<const uint8_t*>(spikes), m, k, cap, scale, inv_scale, static_cast<const W*>(weights),
                       static_cast<W*>(out));
  }
  BE_LAUNCH_CHECK();
  return BE_OK;
}

}  // namespace

// =================================================================================================
// C ABI
// =================================================================================================
extern "C" {

int be_pack_spikes(const void* spikes, int spike_dtype, int64_t n, uint32_t* bits, be_stream_t stream) {
  BE_REQUIRE(n >= 0, BE_ERR_INVALID, "n < 0");
  BE_REQUIRE(n == 0 || (spikes && bits), BE_ERR_INVALID, "null pointer");
  return pack_any(spikes, spike_dtype, n, 1, bits, (n + 31) / 32, static_cast<hipStream_t>(stream));
}

int be_pack_spikes_batched(const void* spikes_bm, int spike_dtype, int64_t n, int64_t n_batch, uint32_t* bits,
                           be_stream_t stream) {
  BE_REQUIRE(n >= 0 && n_batch >= 0 && n_batch <= kMaxBatch, BE_ERR_INVALID, "bad n / n_batch");
  BE_REQUIRE(n == 0 || n_batch == 0 || (spikes_bm && bits), BE_ERR_INVALID, "null buffer");
  return pack_any(spikes_bm, spike_dtype, n, n_batch, bits, (n + 31) / 32, static_cast<hipStream_t>(stream));
}

int be_unpack_spikes(const uint32_t* bits, int64_t n, uint8_t* spikes_out, be_stream_t stream) {
  BE_REQUIRE(n >= 0, BE_ERR_INVALID, "n must be >= 0");
  BE_REQUIRE(n == 0 || (bits && spikes_out), BE_ERR_INVALID, "null buffer");
  if (n == 0) return BE_OK;
  hipLaunchKernelGGL(k_unpack_spikes, dim3(grid_for(n, 256, 4096)), dim3(256), 0, static_cast<hipStream_t>(stream), bits, n,
                     spikes_out);
  BE_LAUNCH_CHECK();
  return BE_OK;
}

int be_compact_spikes(const void* spikes, int spike_dtype, int64_t n, uint32_t* active_ids, uint32_t* count,
                      be_stream_t stream) {
  BE_REQUIRE(n >= 0 && n <= 0xffffffffll, BE_ERR_INVALID, "n out of range");
  BE_REQUIRE(count && (n == 0 || (spikes && active_ids)), BE_ERR_INVALID, "null pointer");
  return compact_any(spikes, spike_dtype, n, 1, active_ids, 0, count, static_cast<hipStream_t>(stream));
}

// batched compaction for the other translation units (be_jitc.hip): spikes_bm [nb, n] -> active[b * stride ...], count[b]
int be_compact_spikes_batched(const void* spikes_bm, int spike_dtype, int64_t n, int64_t n_batch, uint32_t* active_ids,
                              int64_t active_stride, uint32_t* counts, be_stream_t stream) {
  BE_REQUIRE(n >= 0 && n <= 0xffffffffll && n_batch >= 0 && n_batch <= kMaxBatch, BE_ERR_INVALID, "shape out of range");
  BE_REQUIRE(counts && (n == 0 || n_batch == 0 || (spikes_bm && active_ids)), BE_ERR_INVALID, "null pointer");
  return compact_any(spikes_bm, spike_dtype, n, n_batch, active_ids, active_stride, counts, static_cast<hipStream_t>(stream));
}

int64_t be_binary_csrmm_t_workspace_bytes(int64_t m, int64_t k, int64_t n_batch, int wdtype) {
  return direct_ws_bytes(m, k, wdtype, n_batch);
}
int64_t be_binary_csrmv_t_workspace_bytes(int64_t m, int64_t k, int wdtype) { return direct_ws_bytes(m, k, wdtype, 1); }

int be_binary_csrmm_t(const void* weights, int homo, int wdtype, const int32_t* indices, const void* indptr,
                      int indptr_is_i64, int64_t row_len, const void* spikes, int spike_dtype, void* out, int64_t m,
                      int64_t k, int64_t n_batch, void* workspace, int64_t workspace_bytes, be_stream_t stream) {
  BE_REQUIRE(m >= 0 && k >= 0 && m <= 0xffffffffll, BE_ERR_INVALID, "bad shape");
  BE_REQUIRE(n_batch >= 0 && n_batch <= kMaxBatch, BE_ERR_INVALID, "n_batch out of range");
  BE_REQUIRE(check_rows(indptr, row_len), BE_ERR_INVALID, "indptr is NULL and row_len < 0");
  BE_REQUIRE(weights != nullptr, BE_ERR_INVALID, "weights is NULL");
  BE_REQUIRE(k == 0 || n_batch == 0 || out != nullptr, BE_ERR_INVALID, "out is NULL");
  BE_REQUIRE(m == 0 || n_batch == 0 || spikes != nullptr, BE_ERR_INVALID, "spikes is NULL");
  BE_REQUIRE(workspace != nullptr && workspace_bytes >= direct_ws_bytes(m, k, wdtype, n_batch), BE_ERR_WORKSPACE,
             "workspace too small");
  RowPtr rp{indptr, indptr_is_i64, row_len};
  hipStream_t st = static_cast<hipStream_t>(stream);
  BE_DISPATCH_W(wdtype, homo, return (csrmv_t_direct<W, HOMO>(weights, indices, rp, spikes, spike_dtype, out, m, k, n_batch, workspace, st)));
  return BE_OK;
}

int be_binary_csrmv_t(const void* weights, int homo, int wdtype, const int32_t* indices, const void* indptr,
                      int indptr_is_i64, int64_t row_len, const void* spikes, int spike_dtype, void* out, int64_t m,
                      int64_t k, void* workspace, int64_t workspace_bytes, be_stream_t stream) {
  return be_binary_csrmm_t(weights, homo, wdtype, indices, indptr, indptr_is_i64, row_len, spikes, spike_dtype, out, m, k,
                           1, workspace, workspace_bytes, stream);
}

int64_t be_binary_csrmm_nt_workspace_bytes(int64_t m, int64_t k, int64_t n_batch) {
  (void)m;
  const int64_t bits = (((k + 31) / 32) * n_batch + 2) * 4;            // per-column bitmaps
  const int64_t masks = n_batch >= kFusedMinBatch ? (k + 2) * 4 : 0;   // per-neuron column masks of the fused kernel
  return be_align_up(std::max(bits, masks), 256);
}
int64_t be_binary_csrmv_nt_workspace_bytes(int64_t m, int64_t k) { return be_binary_csrmm_nt_workspace_bytes(m, k, 1); }

int be_binary_csrmm_nt(const void* weights, int homo, int wdtype, const int32_t* indices, const void* indptr,
                       int indptr_is_i64, int64_t row_len, const void* spikes, int spike_dtype, void* out, int64_t m,
                       int64_t k, int64_t n_batch, void* workspace, int64_t workspace_bytes, be_stream_t stream) {
  BE_REQUIRE(m >= 0 && k >= 0, BE_ERR_INVALID, "bad shape");
  BE_REQUIRE(n_batch >= 0 && n_batch <= kMaxBatch, BE_ERR_INVALID, "n_batch out of range");
  BE_REQUIRE(check_rows(indptr, row_len), BE_ERR_INVALID, "indptr is NULL and row_len < 0");
  BE_REQUIRE(weights != nullptr, BE_ERR_INVALID, "weights is NULL");
  BE_REQUIRE(m == 0 || n_batch == 0 || out != nullptr, BE_ERR_INVALID, "out is NULL");
  BE_REQUIRE(k == 0 || n_batch == 0 || spikes != nullptr, BE_ERR_INVALID, "spikes is NULL");
  BE_REQUIRE(workspace != nullptr && workspace_bytes >= be_binary_csrmm_nt_workspace_bytes(m, k, n_batch),
             BE_ERR_WORKSPACE, "workspace too small");
  RowPtr rp{indptr, indptr_is_i64, row_len};
  hipStream_t st = static_cast<hipStream_t>(stream);
  // average row length steers the lanes-per-row choice; for CSR it needs indptr[m] which lives on the
  // device, so the caller's row_len doubles as a hint (row_len >= 0: exact for fixed rows, a hint for CSR)
  const int64_t nnz_hint = (row_len >= 0 ? row_len : 64) * m;
  BE_DISPATCH_W(wdtype, homo, return (csrmv_nt<W, HOMO>(weights, indices, rp, nnz_hint, spikes, spike_dtype, out, m, k, n_batch, workspace, st)));
  return BE_OK;
}

int be_binary_csrmv_nt(const void* weights, int homo, int wdtype, const int32_t* indices, const void* indptr,
                       int indptr_is_i64, int64_t row_len, const void* spikes, int spike_dtype, void* out, int64_t m,
                       int64_t k, void* workspace, int64_t workspace_bytes, be_stream_t stream) {
  return be_binary_csrmm_nt(weights, homo, wdtype, indices, indptr, indptr_is_i64, row_len, spikes, spike_dtype, out, m,
                            k, 1, workspace, workspace_bytes, stream);
}

// ---------------------------------------------------------------- scatter plan
// slices are `slice_width` output neurons wide (0 = the LDS capacity 2^slice_shift); a width below the capacity lets
// the caller balance the slices (k = 1M: 64 slices of 15625 instead of 61 full ones and a sliver)
static inline int64_t width_of(int slice_shift, int slice_width) { return slice_width > 0 ? slice_width : (1ll << slice_shift); }
static inline int n_slices_of(int64_t k, int slice_shift, int slice_width = 0) {
  const int64_t w = width_of(slice_shift, slice_width);
  return (int)((k + w - 1) / w);
}
constexpr int kD8MaxWidth = 20000;     // d8 blocks need no pad slot and no power-of-two capacity: 20000 x 8 B = 156 KiB of LDS
constexpr int kH8MaxWidth = 40000;     // h8: the same LDS in 4-byte counters
static inline bool width_ok(int slice_shift, int slice_width, int layout = BE_PLAN_U16) {
  return slice_width >= 0 && (slice_width <= (1 << slice_shift) || (layout == BE_PLAN_D8 && slice_width <= kD8MaxWidth) ||
                              (layout == BE_PLAN_H8 && slice_width <= kH8MaxWidth));
}
// accumulators per task (= stride of a task's partial sums; a multiple of 16 bytes)
static inline int64_t cap_of(int slice_shift, int slice_width, int layout, int homo) {
  if (layout == BE_PLAN_D8) return (width_of(slice_shift, slice_width) + 1) & ~1ll;
  if (layout == BE_PLAN_H8) return (width_of(slice_shift, slice_width) + 3) & ~3ll;
  const int64_t w = width_of(slice_shift, slice_width);      // u16: the slice's own columns (the LDS holds 2^shift + pad)
  return homo ? (w + 3) & ~3ll : (w + 1) & ~1ll;
}
// accumulator slots a workgroup keeps in LDS (the u16 layouts address a pad slot at 2^slice_shift)
static inline int64_t lds_slots_of(int slice_shift, int slice_width, int layout, int homo) {
  return layout == BE_PLAN_U16 ? (1ll << slice_shift) : cap_of(slice_shift, slice_width, layout, homo);
}

int64_t be_scatter_plan_scratch_bytes(int64_t m, int64_t k, int slice_shift, int slice_width) {
  const int64_t n = (int64_t)n_slices_of(k, slice_shift, slice_width) * m;
  const int64_t n_blocks = (n + kScanChunk - 1) / kScanChunk;
  return be_align_up((n_blocks + 2) * 8, 256);
}

int be_scatter_plan_count(const int32_t* indices, const void* indptr, int indptr_is_i64, int64_t row_len, int64_t m,
                          int64_t k, int slice_shift, int slice_width, int homo, int layout, void* seg, void* scratch,
                          int64_t scratch_bytes, int64_t* blob_bytes_host, be_stream_t stream) {
  BE_REQUIRE(m > 0 && k > 0, BE_ERR_INVALID, "empty matrix has no plan");
  BE_REQUIRE(slice_shift >= 4 && slice_shift <= 15, BE_ERR_INVALID, "slice_shift must be in [4, 15]");
  BE_REQUIRE(width_ok(slice_shift, slice_width, layout), BE_ERR_INVALID, "slice_width out of range for this layout");
  BE_REQUIRE(check_rows(indptr, row_len), BE_ERR_INVALID, "indptr is NULL and row_len < 0");
  const int n_slices = n_slices_of(k, slice_shift, slice_width);
  BE_REQUIRE(n_slices <= kMaxSlices, BE_ERR_RANGE, "too many slices for the plan kernels");
  BE_REQUIRE(seg && scratch && blob_bytes_host, BE_ERR_INVALID, "null pointer");
  BE_REQUIRE(scratch_bytes >= be_scatter_plan_scratch_bytes(m, k, slice_shift, slice_width), BE_ERR_WORKSPACE,
             "scratch too small");
  hipStream_t st = static_cast<hipStream_t>(stream);
  RowPtr rp{indptr, indptr_is_i64, row_len};
  const int64_t n = (int64_t)n_slices * m;
  uint2* sg = static_cast<uint2*>(seg);
  if (layout == BE_PLAN_D8 || layout == BE_PLAN_H8) {
    BE_REQUIRE((layout == BE_PLAN_H8) == (homo != 0), BE_ERR_INVALID, "d8 is the heterogeneous layout, h8 the homogeneous one");
    BE_REQUIRE(indptr != nullptr || row_len <= kD8MaxRow, BE_ERR_RANGE, "d8 / h8 layout: rows of at most 16384 entries");
    BE_REQUIRE(n_slices <= kD8MaxSlices, BE_ERR_RANGE, "too many slices for the d8 / h8 layout");
    auto kern = layout == BE_PLAN_H8 ? k_plan_d8_count<true> : k_plan_d8_count<false>;
    BE_HIP(be_allow_lds(reinterpret_cast<const void*>(kern), kD8MaxRow * 8));
    hipLaunchKernelGGL(kern, dim3(grid_for(m, 1, 256 * 8)), dim3(1024), kD8MaxRow * 8, st, indices, rp, m,
                       (uint32_t)width_of(slice_shift, slice_width), n_slices, sg);
  } else {
    BE_REQUIRE(layout == BE_PLAN_U16, BE_ERR_INVALID, "unknown plan layout");
    hipLaunchKernelGGL(k_plan_count, dim3(grid_for(m, 1, 256 * 16)), dim3(256), 0, st, indices, rp, m,
                       (uint32_t)width_of(slice_shift, slice_width), n_slices, homo, sg);
  }
  BE_LAUNCH_CHECK();
  uint64_t* sums = static_cast<uint64_t*>(scratch);
  const int64_t n_blocks = (n + kScanChunk - 1) / kScanChunk;
  BE_REQUIRE(n_blocks < (1ll << 31), BE_ERR_RANGE, "plan index too large");
  hipLaunchKernelGGL(k_scan_block_sums, dim3((unsigned)n_blocks), dim3(256), 0, st, sg, n, sums);
  BE_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(1024), 0, st, sums, n_blocks);
  BE_LAUNCH_CHECK();
  uint64_t total_units = 0;
  BE_HIP(hipMemcpyAsync(&total_units, sums + n_blocks, 8, hipMemcpyDeviceToHost, st));
  BE_HIP(hipStreamSynchronize(st));
  BE_REQUIRE(total_units < (1ull << 32), BE_ERR_RANGE, "matrix too large for a 32-bit block index (512 GiB)");
  hipLaunchKernelGGL(k_scan_apply, dim3((unsigned)n_blocks), dim3(256), 0, st, sg, n, sums);
  BE_LAUNCH_CHECK();
  *blob_bytes_host = (int64_t)(total_units << 7);
  return BE_OK;
}

int be_scatter_plan_fill(const void* weights, int homo, int wdtype, const int32_t* indices, const void* indptr,
                         int indptr_is_i64, int64_t row_len, int64_t m, int64_t k, int slice_shift, int slice_width,
                         int layout, const void* seg, void* blob, uint32_t* maxabs_bits, be_stream_t stream) {
  BE_REQUIRE(m > 0 && k > 0, BE_ERR_INVALID, "empty matrix has no plan");
  BE_REQUIRE(slice_shift >= 4 && slice_shift <= 15, BE_ERR_INVALID, "slice_shift must be in [4, 15]");
  BE_REQUIRE(width_ok(slice_shift, slice_width, layout), BE_ERR_INVALID, "slice_width out of range for this layout");
  BE_REQUIRE(check_rows(indptr, row_len), BE_ERR_INVALID, "indptr is NULL and row_len < 0");
  BE_REQUIRE(seg && maxabs_bits && blob, BE_ERR_INVALID, "null pointer");
  BE_REQUIRE(homo || weights, BE_ERR_INVALID, "hetero plan needs weights");
  const int n_slices = n_slices_of(k, slice_shift, slice_width);
  BE_REQUIRE(n_slices <= kMaxSlices, BE_ERR_RANGE, "too many slices for the plan kernels");
  hipStream_t st = static_cast<hipStream_t>(stream);
  RowPtr rp{indptr, indptr_is_i64, row_len};
  BE_HIP(be_fill_async(maxabs_bits, 0, 4, st));
  BE_HIP(be_fill_async(maxabs_bits + 1, 0xff, 4, st));
  if (layout == BE_PLAN_H8) {
    BE_REQUIRE(homo && n_slices <= kD8MaxSlices, BE_ERR_INVALID, "h8 layout: homogeneous weight, <= 1024 slices");
    BE_REQUIRE(indptr != nullptr || row_len <= kD8MaxRow, BE_ERR_RANGE, "h8 layout: rows of at most 16384 entries");
    auto kern = k_plan_h8_fill;
    BE_HIP(be_allow_lds(reinterpret_cast<const void*>(kern), kD8MaxRow * 8));
    hipLaunchKernelGGL(kern, dim3(grid_for(m, 1, 256 * 8)), dim3(1024), kD8MaxRow * 8, st, indices, rp, m,
                       (uint32_t)width_of(slice_shift, slice_width), n_slices, static_cast<const uint2*>(seg),
                       static_cast<unsigned char*>(blob));
    BE_LAUNCH_CHECK();
    return BE_OK;
  }
  if (layout == BE_PLAN_D8) {
    BE_REQUIRE(!homo && n_slices <= kD8MaxSlices, BE_ERR_INVALID, "d8 layout: heterogeneous weights, <= 1024 slices");
    BE_REQUIRE(indptr != nullptr || row_len <= kD8MaxRow, BE_ERR_RANGE, "d8 layout: rows of at most 16384 entries");
    const int g8 = grid_for(m, 1, 256 * 8);
    const uint32_t wdt = (uint32_t)width_of(slice_shift, slice_width);
#define BE_D8_FILL(WT)                                                                                                   \
    {                                                                                                                    \
      auto kern = k_plan_d8_fill<WT>;                                                                                    \
      BE_HIP(be_allow_lds(reinterpret_cast<const void*>(kern),        \
                                 kD8MaxRow * 8));                                                                        \
      hipLaunchKernelGGL(kern, dim3(g8), dim3(1024), kD8MaxRow * 8, st, static_cast<const WT*>(weights), indices, rp, m, \
                         wdt, n_slices, static_cast<const uint2*>(seg), static_cast<unsigned char*>(blob), maxabs_bits); \
    }
    switch (wdtype) {
      case BE_F32: BE_D8_FILL(float) break;
      case BE_F16: BE_D8_FILL(__half) break;
      case BE_BF16: BE_D8_FILL(__hip_bfloat16) break;
      default: be_set_error("d8 layout: f32 / f16 / bf16 weights"); return BE_ERR_UNSUPPORTED;
    }
#undef BE_D8_FILL
    BE_LAUNCH_CHECK();
    return BE_OK;
  }
  BE_REQUIRE(layout == BE_PLAN_U16, BE_ERR_INVALID, "unknown plan layout");
  const int grid = grid_for(m, 1, 256 * 16);
  BE_DISPATCH_W(wdtype, homo,
                hipLaunchKernelGGL((k_plan_fill<W, HOMO>), dim3(grid), dim3(256), 0, st, static_cast<const W*>(weights),
                                   indices, rp, m, slice_shift, (uint32_t)width_of(slice_shift, slice_width), n_slices,
                                   static_cast<const uint2*>(seg), static_cast<unsigned char*>(blob), maxabs_bits));
  BE_LAUNCH_CHECK();
  return BE_OK;
}

int64_t be_binary_csrmm_t_plan_workspace_bytes(int64_t m, int64_t k, int64_t n_batch, int slice_shift, int slice_width,
                                               int parts, int homo) {
  const int64_t n_slices = n_slices_of(k, slice_shift, slice_width);
  const int64_t acc_bytes = homo ? 4 : 8;
  return counts_bytes(n_batch) + n_batch * active_stride_of(m) * 4 +
         be_align_up(n_batch * n_slices * parts * std::max<int64_t>(1ll << slice_shift, slice_width + 1) * acc_bytes, 256);
}
int64_t be_binary_csrmv_t_plan_workspace_bytes(int64_t m, int64_t k, int slice_shift, int slice_width, int parts, int homo) {
  return be_binary_csrmm_t_plan_workspace_bytes(m, k, 1, slice_shift, slice_width, parts, homo);
}

int be_binary_csrmm_t_plan(const void* weights, int homo, int wdtype, const void* blob, const void* seg,
                           const void* spikes, int spike_dtype, void* out, int64_t m, int64_t k, int64_t n_batch,
                           int slice_shift, int slice_width, int layout, int parts, int scale_exp, void* workspace,
                           int64_t workspace_bytes, be_stream_t stream) {
  BE_REQUIRE(m > 0 && k > 0 && m <= 0xffffffffll && k <= 0xffffffffll, BE_ERR_INVALID, "bad shape");
  BE_REQUIRE(layout == BE_PLAN_U16 || (layout == BE_PLAN_D8 && !homo) || (layout == BE_PLAN_H8 && homo), BE_ERR_INVALID,
             "bad plan layout");
  BE_REQUIRE(n_batch >= 1 && n_batch <= kMaxBatch, BE_ERR_INVALID, "n_batch out of range");
  BE_REQUIRE(slice_shift >= 4 && slice_shift <= 15, BE_ERR_INVALID, "slice_shift must be in [4, 15]");
  BE_REQUIRE(width_ok(slice_shift, slice_width, layout), BE_ERR_INVALID, "slice_width out of range for this layout");
  BE_REQUIRE(parts >= 1 && parts <= 64, BE_ERR_INVALID, "parts must be in [1, 64]");
  BE_REQUIRE(seg && spikes && out && blob, BE_ERR_INVALID, "null pointer");
  BE_REQUIRE(!homo || weights != nullptr, BE_ERR_INVALID, "missing weights");
  BE_REQUIRE(homo || (scale_exp - 32 > -126 && scale_exp - 32 < 127), BE_ERR_INVALID, "scale_exp out of range");
  const int64_t S = cap_of(slice_shift, slice_width, layout, homo);                 // stride of a task's partial sums
  const int64_t slots = lds_slots_of(slice_shift, slice_width, layout, homo);
  const size_t lds = ((size_t)(slots + 1) * (homo ? 4 : 8) + 15) & ~(size_t)15;
  BE_REQUIRE(lds <= 160 * 1024, BE_ERR_RANGE, "slice does not fit LDS (hetero: slice_shift <= 14)");
  BE_REQUIRE(workspace != nullptr &&
                 workspace_bytes >= be_binary_csrmm_t_plan_workspace_bytes(m, k, n_batch, slice_shift, slice_width, parts, homo),
             BE_ERR_WORKSPACE, "workspace too small");
  hipStream_t st = static_cast<hipStream_t>(stream);
  unsigned char* wsb = static_cast<unsigned char*>(workspace);
  uint32_t* count = reinterpret_cast<uint32_t*>(wsb);
  uint32_t* active = reinterpret_cast<uint32_t*>(wsb + counts_bytes(n_batch));
  const int64_t astride = active_stride_of(m);
  void* partial = wsb + counts_bytes(n_batch) + n_batch * astride * 4;
  const int n_slices = n_slices_of(k, slice_shift, slice_width);
  if (n_slices == 1 && parts == 1 && n_batch == 1 && wdtype != BE_F64 &&
      (spike_dtype == BE_SPIKE_BOOL || spike_dtype == BE_SPIKE_FLOAT) &&
      lds + kSingleChunk * 4 + 64 <= 160 * 1024) {
    // small matrix: compaction + accumulate + output conversion in one single-workgroup launch (k_plan_single)
    const float sc1 = ldexpf(1.0f, scale_exp - 32);
    const double isc1 = ldexp(1.0, -scale_exp);
    const int lay = homo ? (layout == BE_PLAN_H8 ? 3 : 1) : (layout == BE_PLAN_D8 ? 2 : 0);
    const int prof1 = be_prof_begin(st);
    int rc1 = BE_ERR_INVALID;
#define BE_SINGLE(LAY, WT) rc1 = launch_plan_single<LAY, WT>(blob, seg, spikes, spike_dtype, m, k, (int)slots, sc1, isc1, weights, out, lds, st)
#define BE_SINGLE_W(LAY)                                                    \
    switch (wdtype) {                                                        \
      case BE_F32: BE_SINGLE(LAY, float); break;                             \
      case BE_F16: BE_SINGLE(LAY, __half); break;                            \
      case BE_BF16: BE_SINGLE(LAY, __hip_bfloat16); break;                   \
      default: be_set_error("planned scatter: f32 / f16 / bf16 outputs"); rc1 = BE_ERR_UNSUPPORTED; \
    }
    if (lay == 0) { BE_SINGLE_W(0) } else if (lay == 1) { BE_SINGLE_W(1) } else if (lay == 2) { BE_SINGLE_W(2) } else { BE_SINGLE_W(3) }
#undef BE_SINGLE_W
#undef BE_SINGLE
    be_prof_end(prof1, st);
    return rc1;
  }
  // the per-batch counters at the head of the workspace are zero on entry (caller contract) and are zeroed
  // again by k_plan_reduce once the accumulate kernel has consumed them: saves a 5 us memset node per step
  ActiveList al;
  int rc = resolve_active(spikes, spike_dtype, m, n_batch, active, astride, count, st, /*zero_first=*/false, &al);
  if (rc != BE_OK) return rc;
  const float scale = ldexpf(1.0f, scale_exp - 32);   // see fixed_from_f32
  const double inv_scale = ldexp(1.0, -scale_exp);
  const int n_tasks = n_slices * parts;
  const dim3 grid((unsigned)((n_tasks + 7) / 8 * 8), (unsigned)n_batch), block(1024);
  const int prof = be_prof_begin(st);
  if (layout == BE_PLAN_H8) {
    auto kern = k_plan_accumulate_h8;
    BE_HIP(be_allow_lds(reinterpret_cast<const void*>(kern), (int)lds));
    hipLaunchKernelGGL(kern, grid, block, lds, st, static_cast<const unsigned char*>(blob), static_cast<const uint2*>(seg),
                       al.ids, al.count, n_slices, (int)S, parts, static_cast<uint32_t*>(partial), astride);
  } else if (homo) {
    auto kern = k_plan_accumulate<true>;
    BE_HIP(be_allow_lds(reinterpret_cast<const void*>(kern), (int)lds));
    hipLaunchKernelGGL(kern, grid, block, lds, st, static_cast<const unsigned char*>(blob), static_cast<const uint2*>(seg),
                       al.ids, al.count, n_slices, slice_shift, parts, scale, static_cast<uint32_t*>(partial), astride, (int)S);
  } else if (layout == BE_PLAN_D8) {
    auto kern = k_plan_accumulate_d8;
    BE_HIP(be_allow_lds(reinterpret_cast<const void*>(kern), (int)lds));
    hipLaunchKernelGGL(kern, grid, block, lds, st, static_cast<const unsigned char*>(blob), static_cast<const uint2*>(seg),
                       al.ids, al.count, n_slices, (int)S, parts, scale, static_cast<unsigned long long*>(partial), astride);
  } else {
    auto kern = k_plan_accumulate<false>;
    BE_HIP(be_allow_lds(reinterpret_cast<const void*>(kern), (int)lds));
    hipLaunchKernelGGL(kern, grid, block, lds, st, static_cast<const unsigned char*>(blob), static_cast<const uint2*>(seg),
                       al.ids, al.count, n_slices, slice_shift, parts, scale, static_cast<unsigned long long*>(partial),
                       astride, (int)S);
  }
  be_prof_end(prof, st);
  BE_LAUNCH_CHECK();
  const int rgrid = grid_for(k, 256, n_batch >= 8 ? 256 : 2048);
  const int64_t pstride = (int64_t)n_tasks * S;
  BE_DISPATCH_W(wdtype, homo,
                hipLaunchKernelGGL((k_plan_reduce<W, HOMO>), dim3(rgrid, (unsigned)n_batch), dim3(256), 0, st,
                                   static_cast<const typename PlanAcc<HOMO>::type*>(partial), parts, n_slices, (int)S,
                                   (uint32_t)width_of(slice_shift, slice_width), k, inv_scale, static_cast<const W*>(weights), static_cast<W*>(out), pstride, count));
  BE_LAUNCH_CHECK();
  return BE_OK;
}

int be_binary_csrmv_t_plan(const void* weights, int homo, int wdtype, const void* blob, const void* seg,
                           const void* spikes, int spike_dtype, void* out, int64_t m, int64_t k, int slice_shift,
                           int slice_width, int layout, int parts, int scale_exp, void* workspace, int64_t workspace_bytes,
                           be_stream_t stream) {
  return be_binary_csrmm_t_plan(weights, homo, wdtype, blob, seg, spikes, spike_dtype, out, m, k, 1, slice_shift, slice_width,
                                layout, parts, scale_exp, workspace, workspace_bytes, stream);
}


// ---------------------------------------------------------------- binned route (no plan)
static inline int64_t binned_cap_align(int64_t cap) { return (cap + 7) & ~7ll; }

int64_t be_binary_csrmv_t_binned_workspace_bytes(int64_t m, int64_t k, int slice_shift, int64_t bin_capacity) {
  const int64_t n_bins = n_slices_of(k, slice_shift);
  const int64_t cap = binned_cap_align(bin_capacity);
  return 256 + be_align_up(m * 4, 256) + 2 * be_align_up(n_bins * 4, 256) + be_align_up(n_bins * cap * 2, 256) +
         be_align_up(n_bins * cap * 4, 256);
}

int be_binary_csrmv_t_binned(const void* weights, int homo, int wdtype, const int32_t* indices, const void* indptr,
                             int indptr_is_i64, int64_t row_len, const void* spikes, int spike_dtype, void* out, int64_t m,
                             int64_t k, int slice_shift, int64_t bin_capacity, int scale_exp, void* workspace,
                             int64_t workspace_bytes, be_stream_t stream) {
  BE_REQUIRE(m > 0 && k > 0 && m <= 0xffffffffll, BE_ERR_INVALID, "bad shape");
  BE_REQUIRE(wdtype == BE_F32, BE_ERR_UNSUPPORTED, "the binned route supports f32 weights / outputs");
  BE_REQUIRE(slice_shift >= 4 && slice_shift <= 15, BE_ERR_INVALID, "slice_shift must be in [4, 15]");
  BE_REQUIRE(check_rows(indptr, row_len), BE_ERR_INVALID, "indptr is NULL and row_len < 0");
  BE_REQUIRE(weights && indices && spikes && out, BE_ERR_INVALID, "null pointer");
  const int n_bins = n_slices_of(k, slice_shift);
  BE_REQUIRE(n_bins <= kMaxBins, BE_ERR_RANGE, "too many bins for the binned route");
  const int64_t cap = binned_cap_align(bin_capacity);
  BE_REQUIRE(cap >= 8 && cap < (1ll << 32), BE_ERR_INVALID, "bin_capacity out of range");
  BE_REQUIRE(homo || (scale_exp - 32 > -126 && scale_exp - 32 < 127), BE_ERR_INVALID, "scale_exp out of range");
  const int64_t S = 1ll << slice_shift;
  const size_t lds = (size_t)S * (homo ? 4 : 8);
  BE_REQUIRE(lds <= 160 * 1024, BE_ERR_RANGE, "slice does not fit LDS (hetero: slice_shift <= 14)");
  BE_REQUIRE(workspace != nullptr &&
                 workspace_bytes >= be_binary_csrmv_t_binned_workspace_bytes(m, k, slice_shift, bin_capacity),
             BE_ERR_WORKSPACE, "workspace too small");
  hipStream_t st = static_cast<hipStream_t>(stream);
  unsigned char* wsb = static_cast<unsigned char*>(workspace);
  uint32_t* count = reinterpret_cast<uint32_t*>(wsb);
  uint32_t* active = reinterpret_cast<uint32_t*>(wsb + 256);
  uint32_t* cursor = reinterpret_cast<uint32_t*>(wsb + 256 + be_align_up(m * 4, 256));
  uint32_t* valid = reinterpret_cast<uint32_t*>(reinterpret_cast<unsigned char*>(cursor) + be_align_up((int64_t)n_bins * 4, 256));
  uint16_t* bin_idx = reinterpret_cast<uint16_t*>(reinterpret_cast<unsigned char*>(valid) + be_align_up((int64_t)n_bins * 4, 256));
  float* bin_w = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(bin_idx) + be_align_up((int64_t)n_bins * cap * 2, 256));
  RowPtr rp{indptr, indptr_is_i64, row_len};
  if ((reinterpret_cast<uintptr_t>(out) & 15) == 0) {
    hipLaunchKernelGGL(k_bin_reset, dim3(grid_for(k / 4 + 1, 256, 1024)), dim3(256), 0, st, static_cast<float*>(out), k, cursor,
                       valid, n_bins, count);
    BE_LAUNCH_CHECK();
  } else {
    BE_HIP(be_fill_async(out, 0, (size_t)k * 4, st));
    BE_HIP(be_fill_async(cursor, 0, (size_t)n_bins * 4, st));
    BE_HIP(be_fill_async(valid, 0xff, (size_t)n_bins * 4, st));
    BE_HIP(be_fill_async(count, 0, 4, st));
  }
  ActiveList al;
  int rc = resolve_active(spikes, spike_dtype, m, 1, active, 0, count, st, /*zero_first=*/false, &al);
  if (rc != BE_OK) return rc;
  const int prof = be_prof_begin(st);
  if (homo)
    hipLaunchKernelGGL((k_bin_rows<float, true>), dim3(256), dim3(1024), 0, st, static_cast<const float*>(weights), indices, rp,
                       al.ids, al.count, slice_shift, n_bins, (uint32_t)cap, cursor, valid, bin_idx, bin_w, static_cast<float*>(out));
  else
    hipLaunchKernelGGL((k_bin_rows<float, false>), dim3(256), dim3(1024), 0, st, static_cast<const float*>(weights), indices, rp,
                       al.ids, al.count, slice_shift, n_bins, (uint32_t)cap, cursor, valid, bin_idx, bin_w, static_cast<float*>(out));
  BE_LAUNCH_CHECK();
  // n_bins * parts ~ 256: every workgroup fills a CU (128 KB of LDS) and its slice is merged into the output with float
  // atomics, one per non-zero accumulator, so more parts than CUs only add merge traffic (39 bins: 13 parts took 55 us, 6 take 30)
  int parts = 256 / (n_bins > 0 ? n_bins : 1);
  parts = parts < 1 ? 1 : (parts > 16 ? 16 : parts);
  const float scale = ldexpf(1.0f, scale_exp - 32);
  const double inv_scale = ldexp(1.0, -scale_exp);
  if (homo) {
    auto kern = k_bin_accumulate<true>;
    BE_HIP(be_allow_lds(reinterpret_cast<const void*>(kern), (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)(n_bins * parts)), dim3(1024), lds, st, bin_idx, bin_w, cursor, valid, (uint32_t)cap,
                       slice_shift, parts, k, scale, inv_scale, static_cast<const float*>(weights), static_cast<float*>(out));
  } else {
    auto kern = k_bin_accumulate<false>;
    BE_HIP(be_allow_lds(reinterpret_cast<const void*>(kern), (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)(n_bins * parts)), dim3(1024), lds, st, bin_idx, bin_w, cursor, valid, (uint32_t)cap,
                       slice_shift, parts, k, scale, inv_scale, static_cast<const float*>(nullptr), static_cast<float*>(out));
  }
  be_prof_end(prof, st);
  BE_LAUNCH_CHECK();
  return BE_OK;
}

// ---------------------------------------------------------------- per-variant symbols
#define BE_DEF_CSR_VARIANT(W, WD, S, SD)                                                                               \
  int be_binary_csrmv_t_homo_##W##_##S(BE_CSR_MV_ARGS) {                                                                \
    return be_binary_csrmm_t(weights, 1, WD, indices, indptr, indptr_is_i64, -1, spikes, SD, out, m, k, 1, workspace,   \
                             workspace_bytes, stream);                                                                  \
  }                                                                                                                     \
  int be_binary_csrmv_t_hetero_##W##_##S(BE_CSR_MV_ARGS) {                                                              \
    return be_binary_csrmm_t(weights, 0, WD, indices, indptr, indptr_is_i64, -1, spikes, SD, out, m, k, 1, workspace,   \
                             workspace_bytes, stream);                                                                  \
  }                                                                                                                     \
  int be_binary_csrmv_nt_homo_##W##_##S(BE_CSR_MV_ARGS) {                                                               \
    return be_binary_csrmm_nt(weights, 1, WD, indices, indptr, indptr_is_i64, -1, spikes, SD, out, m, k, 1, workspace,  \
                              workspace_bytes, stream);                                                                 \
  }                                                                                                                     \
  int be_binary_csrmv_nt_hetero_##W##_##S(BE_CSR_MV_ARGS) {                                                             \
    return be_binary_csrmm_nt(weights, 0, WD, indices, indptr, indptr_is_i64, -1, spikes, SD, out, m, k, 1, workspace,  \
                              workspace_bytes, stream);                                                                 \
  }                                                                                                                     \
  int be_binary_csrmm_t_homo_##W##_##S(BE_CSR_MM_ARGS) {                                                                \
    return be_binary_csrmm_t(weights, 1, WD, indices, indptr, indptr_is_i64, -1, spikes_bm, SD, out_bm, m, k, n_batch,  \
                             workspace, workspace_bytes, stream);                                                       \
  }                                                                                                                     \
  int be_binary_csrmm_t_hetero_##W##_##S(BE_CSR_MM_ARGS) {                                                              \
    return be_binary_csrmm_t(weights, 0, WD, indices, indptr, indptr_is_i64, -1, spikes_bm, SD, out_bm, m, k, n_batch,  \
                             workspace, workspace_bytes, stream);                                                       \
  }                                                                                                                     \
  int be_binary_csrmm_nt_homo_##W##_##S(BE_CSR_MM_ARGS) {                                                               \
    return be_binary_csrmm_nt(weights, 1, WD, indices, indptr, indptr_is_i64, -1, spikes_bm, SD, out_bm, m, k, n_batch, \
                              workspace, workspace_bytes, stream);                                                      \
  }                                                                                                                     \
  int be_binary_csrmm_nt_hetero_##W##_##S(BE_CSR_MM_ARGS) {                                                             \
    return be_binary_csrmm_nt(weights, 0, WD, indices, indptr, indptr_is_i64, -1, spikes_bm, SD, out_bm, m, k, n_batch, \
                              workspace, workspace_bytes, stream);                                                      \
  }

BE_FOR_ALL_VARIANTS(BE_DEF_CSR_VARIANT)

}  // extern "C"
