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
#include "be_csr_shared.h"

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

// the same walk over a RE-INDEXED structure: slot j carries weights[perm[j]] (reference: the perm-fused hybrid kernel,
// brainevent/_csr/binary_indexed_csrmv_hybrid.cu:16-23) — only the weights of active rows are ever read, no data[perm] pass
template <typename W, typename ACC, typename PT>
__global__ void __launch_bounds__(256) k_csrmv_t_direct_indexed(const W* __restrict__ weights, const int32_t* __restrict__ indices,
                                                                RowPtr rp, const PT* __restrict__ perm,
                                                                const uint32_t* __restrict__ active,
                                                                const uint32_t* __restrict__ n_active_p, ACC* __restrict__ out,
                                                                int64_t active_stride, int64_t k) {
  active += (int64_t)blockIdx.y * active_stride;
  out += (int64_t)blockIdx.y * k;
  const uint32_t n_active = n_active_p[blockIdx.y];
  const int lane = lane_id();
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint32_t n_waves = (gridDim.x * blockDim.x) >> 6;
  for (uint32_t a = wave; a < n_active; a += n_waves) {
    const int64_t r = active[a];
    const int64_t b = rp.at(r), e = rp.at(r + 1);
    for (int64_t j = b + lane; j < e; j += 64) atomicAdd(out + indices[j], (ACC)WTraits<W>::load(weights, (int64_t)perm[j]));
  }
}

// gather over a re-indexed structure: one wave per row, the bit-packed input vector read from L2, a weight is loaded only
// behind its spike test (through perm the weights cannot be streamed with the indices anyway)
template <typename W, typename PT>
__global__ void __launch_bounds__(256) k_csrmv_nt_indexed(const W* __restrict__ weights, const int32_t* __restrict__ indices,
                                                          RowPtr rp, const PT* __restrict__ perm,
                                                          const uint32_t* __restrict__ bits, int64_t n_words,
                                                          W* __restrict__ out, int64_t m) {
  using ACC = typename WTraits<W>::acc;
  bits += (int64_t)blockIdx.y * n_words;
  out += (int64_t)blockIdx.y * m;
  const int lane = lane_id();
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t r = wave; r < m; r += n_waves) {
    const int64_t b = rp.at(r), e = rp.at(r + 1);
    ACC acc = ACC(0);
    for (int64_t j = b + lane; j < e; j += 64) {
      const uint32_t c = (uint32_t)indices[j];
      if ((bits[c >> 5] >> (c & 31)) & 1u) acc += (ACC)WTraits<W>::load(weights, (int64_t)perm[j]);
    }
    acc = wave_sum(acc);
    if (lane == 0) WTraits<W>::store(out, r, acc);
  }
}

template <typename W>
__global__ void __launch_bounds__(256) k_convert_from_f32(const float* __restrict__ src, W* __restrict__ dst, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) WTraits<W>::store(dst, i, src[i]);
}

// Spike test of the gather kernels.  The bit-packed input vector sits in LDS when it fits (k <= 1.2M columns: IN_LDS); a
// longer one would cost a 64-byte L2 sector per stored entry (2M columns, 100 per row: 0.83 ms against 0.39 ms at 1M).
// Then LDS holds a COARSE bitmap instead — one bit per group of 2^g_shift columns, set iff any of them fired — and only
// entries whose group fired go on to the exact bitmap in global memory: at 1 % firing and groups of 2 ... 8 columns that
// is 2 ... 8 % of them.
constexpr int64_t kLdsBitmapBytes = 150 * 1024;
// N entries at once (bit e of `valid`: entry e exists; returns bit e set iff its input fired): the exact-bitmap reads of the
// coarse route are issued together and waited for once — one global round trip per group of entries, not one per hit
// AT0: the caller's bitmap starts at LDS address 0 (checked there), so a word's address is its byte offset — through the
// generic pointer every read paid an add of the allocation's (link-time) base, which is 0.
typedef __attribute__((address_space(3))) const uint32_t LdsWord;
template <bool AT0>
__device__ __forceinline__ uint32_t lds_bit_word(const uint32_t* __restrict__ lds_bits, uint32_t word) {
  if (AT0) return *(LdsWord*)(word << 2);
  return lds_bits[word];
}
template <bool IN_LDS, int N, bool AT0 = false>
__device__ __forceinline__ uint32_t spike_mask(const uint32_t* __restrict__ lds_bits, const uint32_t* __restrict__ fine_g,
                                               const uint32_t (&col)[N], uint32_t valid, int g_shift) {
  uint32_t m = 0;
  if (IN_LDS) {
#pragma unroll
    for (int e = 0; e < N; ++e) m |= __builtin_amdgcn_ubfe(lds_bit_word<AT0>(lds_bits, col[e] >> 5), col[e], 1u) << e;   // (v_bfe takes the offset mod 32)
    return m & valid;      // entries that are not the row's hold other rows' ids or zeros: read them, then drop them —
                           // a test per entry put every LDS read behind its own branch and wait
  }
  uint32_t ch = 0;
#pragma unroll
  for (int e = 0; e < N; ++e) {
    const uint32_t cc = col[e] >> g_shift;
    ch |= __builtin_amdgcn_ubfe(lds_bit_word<AT0>(lds_bits, cc >> 5), cc, 1u) << e;
  }
  ch &= valid;
  if (ch == 0u) return 0u;
  uint32_t fw[N];
#pragma unroll
  for (int e = 0; e < N; ++e) fw[e] = ((ch >> e) & 1u) ? fine_g[col[e] >> 5] : 0u;
#pragma unroll
  for (int e = 0; e < N; ++e) m |= __builtin_amdgcn_ubfe(fw[e], col[e], 1u) << e;
  return m;
}

// coarse[b][w] bit i = any fine bit of batch row b in columns [(32 w + i) << g_shift, (32 w + i + 1) << g_shift)
__global__ void __launch_bounds__(256) k_coarsen_bits(const uint32_t* __restrict__ fine, int64_t n_words, uint32_t* __restrict__ coarse,
                                                      int64_t n_cwords, int g_shift) {
  fine += (int64_t)blockIdx.y * n_words;
  coarse += (int64_t)blockIdx.y * n_cwords;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < n_cwords; w += stride) {
    uint32_t cw = 0;
    for (int i = 0; i < 32; ++i) {
      const int64_t c0 = (w * 32 + i) << g_shift;          // first fine bit of the group (a multiple of the group size)
      bool any = false;
      if (g_shift < 5) {
        const int64_t fw = c0 >> 5;
        if (fw < n_words) any = ((fine[fw] >> (c0 & 31)) & ((1u << (1 << g_shift)) - 1u)) != 0u;
      } else {
        const int64_t fw0 = c0 >> 5, nfw = (int64_t)1 << (g_shift - 5);
        for (int64_t t = 0; t < nfw && fw0 + t < n_words; ++t) any = any || fine[fw0 + t] != 0u;
      }
      cw |= (any ? 1u : 0u) << i;
    }
    coarse[w] = cw;
  }
}

// =================================================================================================
// gather (transpose=False): bit-packed spikes in LDS when they fit; a wave per long row, 2 ... 32 lanes per shorter row
// =================================================================================================
// long rows: one wave per row, 16 waves per workgroup (they share the LDS bitmap, which takes most of the LDS, so
// these 16 waves are all the latency hiding a CU gets), 4 consecutive entries per lane per load through raw buffer
// descriptors (range-checked: no tail branches), two iterations (2 KB of indices [+ 2 KB of f32 weights]) in flight.
// f32 weights are streamed unconditionally with the same vector loads — a dependent load per active entry would
// serialise the row at one HBM round trip per hit; other weight dtypes keep the conditional scalar load.
typedef unsigned be_nt_v4u __attribute__((ext_vector_type(4)));
#ifndef BE_GATHER_AUX
#define BE_GATHER_AUX 2      // cache policy of the wave-per-row matrix streams: nt (read once; +3 ... 8 % at 2e8 entries)
#endif
#ifndef BE_GATHER_DEPTH_HOMO
#define BE_GATHER_DEPTH_HOMO 4   // groups ahead without per-entry weights (11 registers per group): 2 -> 4 measured 0 ... 6 %, 8 nothing more
#endif
#ifndef BE_GATHER_VEC_NT
#define BE_GATHER_VEC_NT 1   // nt on the lanes-per-row matrix streams too (+2 ... 8 %; the 16-lane tier at 100 entries
                             // per row loses 6 % because a row's last piece reads into the next row)
#endif
__device__ __forceinline__ uint4 gather_ld16(const void* p) {
  if (BE_GATHER_VEC_NT) { const be_nt_v4u t = __builtin_nontemporal_load(reinterpret_cast<const be_nt_v4u*>(p)); return make_uint4(t.x, t.y, t.z, t.w); }
  return *reinterpret_cast<const uint4*>(p);
}
constexpr int64_t kGatherVecMaxRow = 200;   // average row length up to which the lanes-per-row vector kernel is used (32 lanes: 256 entries in two passes)

template <typename W, bool HOMO, bool BITS_IN_LDS>
__global__ void __launch_bounds__(1024) k_csrmv_nt_wave(const W* __restrict__ weights, const int32_t* __restrict__ indices,
                                                        RowPtr rp, const uint32_t* __restrict__ bits_g, int64_t n_words,
                                                        W* __restrict__ out, int64_t m, const uint32_t* __restrict__ coarse_g,
                                                        int64_t n_cwords, int g_shift) {
  bits_g += (int64_t)blockIdx.y * n_words;
  out += (int64_t)blockIdx.y * m;
  extern __shared__ uint32_t bits_s[];
  {   // the exact bitmap (BITS_IN_LDS) or the coarse one (spike_on)
    const uint32_t* src = BITS_IN_LDS ? bits_g : coarse_g + (int64_t)blockIdx.y * n_cwords;
    const int64_t nw = BITS_IN_LDS ? n_words : n_cwords;
    for (int64_t i = threadIdx.x; i < nw; i += blockDim.x) bits_s[i] = src[i];
    __syncthreads();
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
          c[u] = __builtin_amdgcn_raw_buffer_load_b128(ri, off, 0, BE_GATHER_AUX);
          if (VECW) wv[u] = __builtin_amdgcn_raw_buffer_load_b128(rw, off, 0, BE_GATHER_AUX);
        }
        uint32_t cols[8], valid = 0;
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            cols[4 * u + q] = c[u][q];
            valid |= (j0 + 256 * u + 4 * lane + q < plen ? 1u : 0u) << (4 * u + q);
          }
        const uint32_t onm = spike_mask<BITS_IN_LDS, 8>(bits_s, bits_g, cols, valid, g_shift);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int64_t j = j0 + 256 * u + 4 * lane + q;
            const bool on = (onm >> (4 * u + q)) & 1u;
            if (HOMO) cnt += on ? 1 : 0;
            else if (VECW) acc += on ? (ACC)__uint_as_float(wv[u][q]) : ACC(0);
            else if (on) acc += (ACC)WTraits<W>::load(weights, b + p0 + j);
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
        // (readlane returns a signed int: without the uint32_t casts a low word >= 2^31 sign-extends over the high word — rows
        //  starting beyond 2^31 entries read from a wild address; found at C2, 1e10 entries, in round 4)
        gb[q] = (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)b_hi, src) << 32) |
                          (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)b_lo, src));
        gl[q] = (i + q < nvalid)
                    ? (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)l_hi, src) << 32) |
                                (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)l_lo, src))
                    : 0;
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int64_t hl = gl[q] < 256 ? gl[q] : 256;       // head: first 256 entries
        auto ri = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(indices + gb[q]), 0, (int)(hl * 4), 0x00020000);
        c[q] = __builtin_amdgcn_raw_buffer_load_b128(ri, lane * 16, 0, BE_GATHER_AUX);
        if (VECW) {
          auto rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<W*>(weights + gb[q]), 0, (int)(hl * 4), 0x00020000);
          wv[q] = __builtin_amdgcn_raw_buffer_load_b128(rw, lane * 16, 0, BE_GATHER_AUX);
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (i + q >= nvalid) break;                           // uniform
        ACC acc = ACC(0);
        int cnt = 0;
        uint32_t cols[4], valid = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          cols[e] = c[q][e];
          valid |= ((4 * lane + e < gl[q] && 4 * lane + e < 256) ? 1u : 0u) << e;
        }
        const uint32_t onm = spike_mask<BITS_IN_LDS, 4>(bits_s, bits_g, cols, valid, g_shift);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int64_t j = 4 * lane + e;
          const bool on = (onm >> e) & 1u;
          if (HOMO) cnt += on ? 1 : 0;
          else if (VECW) acc += on ? (ACC)__uint_as_float(wv[q][e]) : ACC(0);
          else if (on) acc += (ACC)WTraits<W>::load(weights, gb[q] + j);
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

// short and medium rows (a few to ~200 entries on average): LPR lanes per row, 64 / LPR rows per group (one load
// instruction), four consecutive entries per lane and load (16 B of indices [+ 16 B of f32 weights]), two passes of a row
// per group.  A wave takes 64 CONSECUTIVE rows at a time (coalesced row pointers; results leave as one 256-byte store
// through a per-wave LDS row) and walks their groups in straight-line code two groups ahead — the first version looped
// over the groups and paid one full memory latency per group.  Rows longer than two passes finish in a serial tail.
// The scalar lanes-per-row kernel of round 1 loaded a weight only behind its spike test (a dependent round trip per hit).
// 2e8 entries, 1 % of the inputs active, ms per product (tools/bench_gather_rows.py), weighted: 4 per row 3.77 -> 0.57,
// 12: 4.86 -> 0.35, 48: 2.52 -> 0.33, 100: 0.50 -> 0.31 (5.2 TB/s); counted: 12: 4.45 -> 0.28, 100: 0.41 -> 0.19.
// Then: no branch per entry or per piece (unconditional LDS reads masked afterwards, one buffer descriptor per batch with
// out-of-range offsets for the pieces a row does not reach), bitmap at LDS address 0 — counted 12: 0.21, 48: 0.19, 100: 0.17.
template <typename W, bool HOMO, int LPR, bool BITS_IN_LDS>
__global__ void __launch_bounds__(1024) k_csrmv_nt_vec(const W* __restrict__ weights, const int32_t* __restrict__ indices,
                                                       RowPtr rp, const uint32_t* __restrict__ bits_g, int64_t n_words,
                                                       W* __restrict__ out, int64_t m, const uint32_t* __restrict__ coarse_g,
                                                       int64_t n_cwords, int g_shift) {
  bits_g += (int64_t)blockIdx.y * n_words;
  out += (int64_t)blockIdx.y * m;
  extern __shared__ uint32_t bits_s[];
  {   // the exact bitmap (BITS_IN_LDS) or the coarse one (spike_mask)
    const uint32_t* src = BITS_IN_LDS ? bits_g : coarse_g + (int64_t)blockIdx.y * n_cwords;
    const int64_t nw = BITS_IN_LDS ? n_words : n_cwords;
    for (int64_t i = threadIdx.x; i < nw; i += blockDim.x) bits_s[i] = src[i];
    __syncthreads();
  }
  using ACC = typename WTraits<W>::acc;
  constexpr bool VECW = std::is_same<W, float>::value && !HOMO;
  constexpr int RPW = 64 / LPR;                  // rows per group (one load instruction)
  constexpr int NGRP = 64 / RPW;                 // groups of a 64-row batch
  constexpr int PASS = 4 * LPR;                  // entries of a row per pass; a group holds two passes of its rows
  constexpr int DEPTH_WANT = (HOMO || !VECW) ? BE_GATHER_DEPTH_HOMO : 2;
  constexpr int DEPTH = NGRP < DEPTH_WANT ? NGRP : DEPTH_WANT;   // groups in flight ahead of the one being consumed (registers: 19 per group, 11 without weights)
  // per wave: the 64 results of a batch (long rows add their tails here).  Behind the bitmap in the dynamic allocation, so
  // that the bitmap starts at LDS address 0 and a spike test needs no base add.
  ACC (*res_s)[64] = reinterpret_cast<ACC (*)[64]>(bits_s + (((BITS_IN_LDS ? n_words : n_cwords) + 3) & ~(int64_t)3));
  const int lane = lane_id(), sub = lane % LPR, slot = lane / LPR, wv_id = threadIdx.x >> 6;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const int64_t nnz_end = rp.at(m);              // a 16-byte load may not run past the arrays (the descriptors end there)
  ACC w0 = ACC(0);
  if (HOMO) w0 = (ACC)WTraits<W>::load(weights, 0);

  struct Grp {
    be_nt_v4u c[2], wv[2];
    int32_t rel;               // row start, entries from the batch's first row
    int32_t len;               // clamped to two passes: all a group looks at
  };
  // a batch = 64 CONSECUTIVE rows (coalesced row pointers and results); batches are dealt to the waves round-robin
  for (int64_t r0 = wave * 64; r0 < m; r0 += n_waves * 64) {
    const int64_t my_r = r0 + lane;
    const bool valid = my_r < m;
    const int64_t rc = valid ? my_r : m - 1;
    const int64_t rb = rp.at(rc);
    const int64_t rl = valid ? rp.at(rc + 1) - rb : 0;
    // The batch's entries are read through ONE descriptor over [first row's start, +2^29 entries) with 32-bit offsets
    // relative to it: loads past the arrays' end return zeros (no end-of-array special case), a piece a row does not reach
    // gets an out-of-range offset instead of a branch, and the address arithmetic is one add-shift per piece.  A batch that
    // spans more than 2^29 entries (`far`) leaves all its rows to the tail loop below.
    const int64_t b_first = (int64_t)(((uint64_t)__builtin_amdgcn_readfirstlane((int)(rb >> 32)) << 32) |
                                      (uint32_t)__builtin_amdgcn_readfirstlane((int)rb));
    const int32_t cap = (int32_t)(rl < 2 * PASS ? rl : 2 * PASS);
    const bool far = __ballot(rb - b_first + cap > (1ll << 29)) != 0ull;
    const int64_t span = nnz_end - b_first < (1ll << 29) ? nnz_end - b_first : (1ll << 29);
    auto ri = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(indices + b_first), 0, (int)(span * 4), 0x00020000);
    auto rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<W*>(VECW ? weights + b_first : weights), 0,
                                                VECW ? (int)(span * 4) : 0, 0x00020000);
    const int32_t my_rel = far ? 0 : (int32_t)(rb - b_first), my_cap = far ? 0 : cap;
    // The groups of the batch are walked in straight-line code, DEPTH groups ahead: no load crosses a back edge, so the
    // wait counts stay exact (a loop around issue / consume makes hipcc wait for every load in flight: be_csr_plan.hip).
    auto issue = [&](Grp& g, int q) {
      const int src = q * RPW + slot;
      g.rel = __shfl(my_rel, src, 64);
      g.len = __shfl(my_cap, src, 64);
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int32_t j = u * PASS + 4 * sub;
        const int off = j < g.len ? (g.rel + j) * 4 : (int)0x80000000;     // (a row's last piece reads into the next row)
        g.c[u] = __builtin_amdgcn_raw_buffer_load_b128(ri, off, 0, BE_GATHER_AUX);
        if (VECW) g.wv[u] = __builtin_amdgcn_raw_buffer_load_b128(rw, off, 0, BE_GATHER_AUX);
        else g.wv[u] = be_nt_v4u{0u, 0u, 0u, 0u};
      }
    };
    auto consume = [&](const Grp& g, int q) {
      uint32_t cols[8];
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) cols[4 * u + e] = g.c[u][e];
      // entries of the row in this lane's two pieces: 0 ... 4 each (a clamp, not a compare per entry)
      const int32_t left = g.len - 4 * sub;
      const int32_t n0 = left < 0 ? 0 : (left > 4 ? 4 : left);
      const int32_t n1 = left - PASS < 0 ? 0 : (left - PASS > 4 ? 4 : left - PASS);
      const uint32_t vmask = ((1u << n0) - 1u) | (((1u << n1) - 1u) << 4);
      const uint32_t onm = spike_mask<BITS_IN_LDS, 8, true>(bits_s, bits_g, cols, vmask, g_shift);
      ACC acc = ACC(0);
      int cnt = HOMO ? __popc(onm) : 0;
      if (!HOMO) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const bool on = (onm >> (4 * u + e)) & 1u;
            if (VECW) acc += on ? (ACC)__uint_as_float(g.wv[u][e]) : ACC(0);
            else if (on) acc += (ACC)WTraits<W>::load(weights, b_first + g.rel + (int64_t)u * PASS + 4 * sub + e);
          }
      }
      if (HOMO) {
#pragma unroll
        for (int off = LPR / 2; off > 0; off >>= 1) cnt += __shfl_down(cnt, off, LPR);
        acc = (ACC)cnt;
      } else {
#pragma unroll
        for (int off = LPR / 2; off > 0; off >>= 1) acc += __shfl_down(acc, off, LPR);
      }
      if (sub == 0) res_s[wv_id][q * RPW + slot] = acc;
    };
    Grp g[NGRP];
#pragma unroll
    for (int q = 0; q < DEPTH; ++q) issue(g[q], q);
#pragma unroll
    for (int q = 0; q < NGRP; ++q) {
      if (q + DEPTH < NGRP) issue(g[q + DEPTH], q + DEPTH);
      consume(g[q], q);
    }
    // rows longer than two passes: the rest of the row, a wave per row (the host picks LPR so that these are the exception)
    const int64_t tail_from = far ? 0 : 2 * PASS;     // (a `far` batch took nothing above)
    unsigned long long longm = __ballot(rl > tail_from);
    while (longm) {
      const int src = __ffsll((long long)longm) - 1;
      longm &= longm - 1;
      const int64_t b = __shfl(rb, src, 64), len = __shfl(rl, src, 64);
      ACC acc = ACC(0);
      int cnt = 0;
      for (int64_t j0 = tail_from; j0 < len; j0 += 2 * 256) {
        be_nt_v4u c[2] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}}, wv[2] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int64_t j = j0 + (int64_t)u * 256 + 4 * lane;
          if (j + 4 <= len || (j < len && b + j + 4 <= nnz_end)) {
            const uint4 t = gather_ld16(indices + b + j);
            c[u] = be_nt_v4u{t.x, t.y, t.z, t.w};
            if (VECW) {
              const uint4 tw = gather_ld16(reinterpret_cast<const float*>(weights) + b + j);
              wv[u] = be_nt_v4u{tw.x, tw.y, tw.z, tw.w};
            }
          } else if (j < len) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (j + e < len) {
                c[u][e] = (uint32_t)indices[b + j + e];
                if (VECW) wv[u][e] = __float_as_uint(reinterpret_cast<const float*>(weights)[b + j + e]);
              }
          }
        }
        uint32_t cols[8], vmask = 0;
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            cols[4 * u + e] = c[u][e];
            vmask |= (j0 + (int64_t)u * 256 + 4 * lane + e < len ? 1u : 0u) << (4 * u + e);
          }
        const uint32_t onm = spike_mask<BITS_IN_LDS, 8, true>(bits_s, bits_g, cols, vmask, g_shift);
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const bool on = (onm >> (4 * u + e)) & 1u;
            if (HOMO) cnt += on ? 1 : 0;
            else if (VECW) acc += on ? (ACC)__uint_as_float(wv[u][e]) : ACC(0);
            else if (on) acc += (ACC)WTraits<W>::load(weights, b + j0 + (int64_t)u * 256 + 4 * lane + e);
          }
      }
      if (HOMO) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down(cnt, off, 64);
        acc = (ACC)cnt;
      } else {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
      }
      if (lane == 0) res_s[wv_id][src] += acc;        // (same wave wrote the slot: program order)
    }
    // one coalesced store of the batch's 64 results (LDS traffic of one wave needs no barrier, only the wait below)
    __builtin_amdgcn_s_waitcnt(0xc07f);               // lgkmcnt(0)
    if (valid) {
      const ACC r = res_s[wv_id][lane];
      WTraits<W>::store(out, my_r, HOMO ? r * w0 : r);
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
    for (int b0 = 0; b0 < nc; b0 += 8) {       // eight batch rows' loads in flight
      typename SP::type v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = spikes_bm[(int64_t)(b0 + u < nc ? b0 + u : nc - 1) * len + i];
#pragma unroll
      for (int u = 0; u < 8; ++u) mk |= ((b0 + u < nc && SP::active(v[u])) ? 1u : 0u) << (b0 + u);
    }
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
          c[u] = __builtin_amdgcn_raw_buffer_load_b128(ri, off, 0, BE_GATHER_AUX);       // out-of-range lanes read 0
          if (VECW) wv[u] = __builtin_amdgcn_raw_buffer_load_b128(rw, off, 0, BE_GATHER_AUX);
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

// The same product for rows of up to ~200 entries: LPR lanes per row and 64 / LPR rows per group, as in k_csrmv_nt_vec
// (64 consecutive rows per wave batch, two passes of a row per group, the next group's index / weight loads issued before
// the current group's mask gathers are waited for).  The wave-per-row kernel above pays both dependent round trips (entries,
// then masks) and a 64-lane fold + clear of the private slots for every row: 16 ps per entry on rows of 100 against 7 ps
// on rows of 1000.  Rows longer than two passes are finished by the whole wave before the group's fold.
template <typename W, bool HOMO, int LPR>
__global__ void __launch_bounds__(1024) k_csrmm_nt_fused_vec(const W* __restrict__ weights, const int32_t* __restrict__ indices,
                                                             RowPtr rp, const uint32_t* __restrict__ mask, int nc,
                                                             W* __restrict__ out_bm, int64_t m) {
  extern __shared__ float fused_s[];                 // [1024][kFusedSlots]; uint32 counts when HOMO
  constexpr bool VECW = std::is_same<W, float>::value && !HOMO;
  constexpr int RPW = 64 / LPR, NGRP = 64 / RPW, PASS = 4 * LPR;
  float* my = fused_s + threadIdx.x * kFusedSlots;
  uint32_t* my_u = reinterpret_cast<uint32_t*>(my);
#pragma unroll
  for (int b = 0; b < 32; ++b) my[b] = 0.0f;
  const int lane = lane_id(), sub = lane % LPR, slot = lane / LPR;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  float* wave_slots = fused_s + (threadIdx.x & ~63) * kFusedSlots;
  const int64_t nnz_end = rp.at(m);
  float w0 = 0.0f;
  if (HOMO) w0 = (float)WTraits<W>::load(weights, 0);

  struct Grp {
    be_nt_v4u c[2], wv[2];
    int64_t b;
    int32_t len;               // clamped to two passes
  };
  auto add_entry = [&](uint32_t bits, uint32_t wbits, int64_t pos) {          // visit the set bits of one entry's mask
    if (bits) {
      float w = 0.0f;
      if (!HOMO) w = VECW ? __uint_as_float(wbits) : (float)WTraits<W>::load(weights, pos);
      do {
        const int b = __ffs(bits) - 1;
        bits &= bits - 1;
        if (HOMO) my_u[b] += 1u;
        else my[b] += w;
      } while (bits);
    }
  };
  for (int64_t r0 = wave * 64; r0 < m; r0 += n_waves * 64) {
    const int64_t my_r = r0 + lane;
    const bool valid = my_r < m;
    const int64_t rc = valid ? my_r : m - 1;
    const int64_t rb = rp.at(rc);
    const int64_t rl = valid ? rp.at(rc + 1) - rb : 0;
    auto issue = [&](Grp& g, int q) {
      const int src = q * RPW + slot;
      g.b = __shfl(rb, src, 64);
      {
        const int64_t l64 = __shfl(rl, src, 64);
        g.len = (int32_t)(l64 < 2 * PASS ? l64 : 2 * PASS);
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        g.c[u] = be_nt_v4u{0u, 0u, 0u, 0u};
        g.wv[u] = be_nt_v4u{0u, 0u, 0u, 0u};
        const int32_t j = u * PASS + 4 * sub;
        if (j + 4 <= g.len || (j < g.len && g.b + j + 4 <= nnz_end)) {
          const uint4 t = gather_ld16(indices + g.b + j);
          g.c[u] = be_nt_v4u{t.x, t.y, t.z, t.w};
          if (VECW) {
            const uint4 tw = gather_ld16(reinterpret_cast<const float*>(weights) + g.b + j);
            g.wv[u] = be_nt_v4u{tw.x, tw.y, tw.z, tw.w};
          }
        } else if (j < g.len) {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (j + e < g.len) {
              g.c[u][e] = (uint32_t)indices[g.b + j + e];
              if (VECW) g.wv[u][e] = __float_as_uint(reinterpret_cast<const float*>(weights)[g.b + j + e]);
            }
        }
      }
    };
    // One group at a time, in a real loop (the body — tails, fold — is too long to unroll 8 ... 32 times): the mask gathers
    // of group q and the index / weight loads of group q + 1 are issued together and waited for once.
    Grp g;
    issue(g, 0);
#pragma unroll 1
    for (int q = 0; q < NGRP; ++q) {
      uint32_t mk[8];
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) mk[4 * u + e] = mask[g.c[u][e]];          // unconditional gathers (column 0 past a row's end)
      Grp gn = g;
      if (q + 1 < NGRP) issue(gn, q + 1);
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int32_t j = u * PASS + 4 * sub + e;
          add_entry(j < g.len ? mk[4 * u + e] : 0u, g.wv[u][e], g.b + j);
        }
      // rows of this group longer than two passes: the whole wave takes the rest of each (the slots of ALL its lanes are
      // folded into that row below, so the tail entries go to the slots of the row's own lanes: lane l -> slot lane l % LPR)
      const int64_t my_row_len = __shfl(rl, q * RPW + slot, 64);             // (every lane takes part in the shuffle)
      unsigned long long longm = __ballot(sub == 0 && my_row_len > 2 * PASS);
      while (longm) {
        const int src_lane = __ffsll((long long)longm) - 1;                       // first lane of the long row's lane group
        longm &= longm - 1;
        const int src = q * RPW + src_lane / LPR;
        const int64_t b = __shfl(rb, src, 64), len = __shfl(rl, src, 64);
        float* tgt = wave_slots + (src_lane + sub) * kFusedSlots;                // a slot row of that group, by this lane's sub index
        uint32_t* tgt_u = reinterpret_cast<uint32_t*>(tgt);
        for (int64_t j0 = 2 * PASS; j0 < len; j0 += 64) {
          const int64_t j = j0 + lane;
          uint32_t bits = 0;
          float w = 0.0f;
          if (j < len) {
            bits = mask[indices[b + j]];
            if (!HOMO && bits) w = (float)WTraits<W>::load(weights, b + j);
          }
          // lanes with the same sub index share a slot row here: serialise them by their slot number (RPW rounds)
#pragma unroll
          for (int turn = 0; turn < RPW; ++turn) {
            if (slot == turn) {
              uint32_t bb = bits;
              while (bb) {
                const int bi = __ffs(bb) - 1;
                bb &= bb - 1;
                if (HOMO) tgt_u[bi] += 1u;
                else tgt[bi] += w;
              }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          }
        }
      }
      // fold the group's rows: lane (slot, sub) sums columns sub, sub + LPR, ... over the LPR lanes of its row
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      {
        const int64_t row = r0 + q * RPW + slot;
        const float* src = wave_slots + (slot * LPR) * kFusedSlots;
        for (int col = sub; col < 32; col += LPR) {
          float sum = 0.0f;
          uint32_t cnt = 0;
#pragma unroll 8
          for (int t = 0; t < LPR; ++t) {
            if (HOMO) cnt += reinterpret_cast<const uint32_t*>(src)[t * kFusedSlots + col];
            else sum += src[t * kFusedSlots + col];
          }
          if (HOMO) sum = (float)cnt * w0;
          if (col < nc && row < m) WTraits<W>::store(out_bm, (int64_t)col * m + row, sum);
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int b = 0; b < 32; ++b) my[b] = 0.0f;
      g = gn;
    }
  }
}

// =================================================================================================
// host-side launch helpers.  Every op takes a batch: spikes are batch-major [n_batch, len], outputs
// batch-major [n_batch, out_len]; the *mv entry points are the n_batch = 1 case, the *mm entry points
// launch the same kernels with gridDim.y = n_batch (one launch per stage for the whole batch — the
// reference loops over the columns on the host, brainevent/_csr/binary_csrmm_hybrid.cu:16-57).
// =================================================================================================
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
                   int64_t m, int64_t k, int64_t nb, void* ws, hipStream_t st, const void* perm = nullptr, int perm64 = 0) {
  unsigned char* wsb = static_cast<unsigned char*>(ws);
  uint32_t* count = reinterpret_cast<uint32_t*>(wsb);
  uint32_t* active = reinterpret_cast<uint32_t*>(wsb + counts_bytes(nb));
  const int64_t astride = active_stride_of(m);
  constexpr bool via_f32 = std::is_same<W, __half>::value || std::is_same<W, __hip_bfloat16>::value;
  using ACC = typename std::conditional<std::is_same<W, double>::value, double, float>::type;
  ACC* acc = via_f32 ? reinterpret_cast<ACC*>(wsb + counts_bytes(nb) + nb * astride * 4) : static_cast<ACC*>(out);
  if (k > 0 && nb > 0) BE_HIP(be_fill_async(acc, 0, (size_t)k * nb * sizeof(ACC), st));
  ActiveList al;
  int rc = be_resolve_active(spikes, sd, m, nb, active, astride, count, st, true, &al);
  if (rc != BE_OK) return rc;
  if (m > 0 && k > 0 && nb > 0) {
    const int gx = nb >= 8 ? 512 : 2048;
    const int prof = be_prof_begin(st);
    if (perm != nullptr && !HOMO) {       // (one shared weight: perm is irrelevant, as in the reference)
      if (perm64)
        hipLaunchKernelGGL((k_csrmv_t_direct_indexed<W, ACC, int64_t>), dim3(gx, (unsigned)nb), dim3(256), 0, st,
                           static_cast<const W*>(weights), indices, rp, static_cast<const int64_t*>(perm), al.ids, al.count, acc,
                           astride, k);
      else
        hipLaunchKernelGGL((k_csrmv_t_direct_indexed<W, ACC, int32_t>), dim3(gx, (unsigned)nb), dim3(256), 0, st,
                           static_cast<const W*>(weights), indices, rp, static_cast<const int32_t*>(perm), al.ids, al.count, acc,
                           astride, k);
    } else
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

// coarse bitmap geometry of a gather over k input columns: group size 2^g_shift (0: the exact bitmap fits LDS)
static inline int gather_g_shift(int64_t k) {
  int g = 0;
  while ((((k + ((int64_t)1 << g) - 1) >> g) + 31) / 32 * 4 > kLdsBitmapBytes) ++g;
  return g;
}
static inline int64_t gather_cwords(int64_t k, int g_shift) { return g_shift ? (((k + ((int64_t)1 << g_shift) - 1) >> g_shift) + 31) / 32 : 0; }

template <typename W, bool HOMO>
int csrmv_nt(const void* weights, const int32_t* indices, RowPtr rp, int64_t nnz_hint, const void* spikes, int sd,
             void* out, int64_t m, int64_t k, int64_t nb, void* ws, hipStream_t st) {
  const int64_t n_words = (k + 31) / 32;
  // one pass over the matrix for all columns against one pass of the vector kernel per column (2e8 entries, 8 / 32
  // columns, ms: rows of 24: 4.8 / 7.0 fused, 2.7 / 9.7 per column; 100: 1.9 / 3.2 against 2.4 / 9.3; 250: 1.5 / 2.1
  // against 2.3 / 9.4; tools/bench_gather_batched.py): fused from (average row length x columns) >= kFusedMinWork
  if (nb >= kFusedMinBatch && m > 0 && (nnz_hint / m) * nb >= kFusedMinWork && nnz_hint / m >= 16 && !std::is_same<W, double>::value &&
      (sd == BE_SPIKE_BOOL || sd == BE_SPIKE_FLOAT)) {
    // long rows, several columns: one pass over the matrix per 32 columns (k_csrmm_nt_fused)
    uint32_t* mask = static_cast<uint32_t*>(ws);
    const size_t lds = (size_t)1024 * kFusedSlots * 4;
    auto kern = k_csrmm_nt_fused<W, HOMO>;
    const int64_t avg_row = nnz_hint / m;
    const int grid_rows = avg_row <= kGatherVecMaxRow ? 16 * 64 : 16;      // rows a workgroup takes per round
    if (avg_row <= 45) kern = k_csrmm_nt_fused_vec<W, HOMO, 8>;
    else if (avg_row <= 100) kern = k_csrmm_nt_fused_vec<W, HOMO, 16>;
    else if (avg_row <= kGatherVecMaxRow) kern = k_csrmm_nt_fused_vec<W, HOMO, 32>;
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
      hipLaunchKernelGGL(kern, dim3(grid_for(m, grid_rows, 256)), dim3(1024), lds, st, static_cast<const W*>(weights), indices, rp,
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
  // more columns than the LDS holds bits for: a coarse bitmap (behind the exact one in the workspace) goes to LDS instead
  const int g_shift = gather_g_shift(k);
  const int64_t n_cwords = gather_cwords(k, g_shift);
  const uint32_t* coarse = nullptr;
  if (g_shift) {
    uint32_t* cw = static_cast<uint32_t*>(ws) + n_words * nb + 2;
    hipLaunchKernelGGL(k_coarsen_bits, dim3(grid_for(n_cwords, 256, 1024), (unsigned)nb), dim3(256), 0, st, bits, n_words, cw, n_cwords,
                       g_shift);
    BE_LAUNCH_CHECK();
    coarse = cw;
  }
  const bool in_lds = g_shift == 0;
  const size_t lds = (size_t)(in_lds ? n_words : n_cwords) * 4;
  const int64_t avg = nnz_hint / (m > 0 ? m : 1);
  if (avg <= kGatherVecMaxRow) {      // short and medium rows: 2 ... 32 lanes per row, four entries per lane and load
    const int prof = be_prof_begin(st);
    // bitmap (rounded to 16 bytes) + the 16 waves' result rows
    const size_t vlds = (size_t)((((in_lds ? n_words : n_cwords) + 3) & ~(int64_t)3) * 4) + 16 * 64 * sizeof(typename WTraits<W>::acc);
    // the kernels address the bitmap from LDS address 0 (spike_mask AT0): true while the very instantiation that is launched
    // declares no static LDS (checked once per kernel; a static __shared__ added later fails here, loudly, instead of
    // testing the wrong bits)
#define BE_NT_VEC_AT0(KERN)                                                                                                 \
    do {                                                                                                                    \
      if (be_static_lds_bytes(reinterpret_cast<const void*>(KERN)) != 0) {                                                  \
        be_set_error("k_csrmv_nt_vec: static LDS in front of the bitmap (spike_mask AT0 addresses it from 0)");             \
        return BE_ERR_UNSUPPORTED;                                                                                          \
      }                                                                                                                     \
    } while (0)
#define BE_NT_VEC(LPR_)                                                                                                     \
    do {                                                                                                                    \
      const int grid = grid_for(m, 16 * 64, nb >= 8 ? 256 : 512);                                                   \
      if (in_lds) {                                                                                                          \
        auto kern = k_csrmv_nt_vec<W, HOMO, LPR_, true>;                                                                    \
        BE_NT_VEC_AT0(kern);                                                                                                 \
        BE_HIP(be_allow_lds(reinterpret_cast<const void*>(kern), (int)vlds));                                               \
        hipLaunchKernelGGL(kern, dim3(grid, (unsigned)nb), dim3(1024), vlds, st, static_cast<const W*>(weights), indices, rp, \
                           bits, n_words, static_cast<W*>(out), m, coarse, n_cwords, g_shift);                              \
      } else {                                                                                                               \
        auto kern = k_csrmv_nt_vec<W, HOMO, LPR_, false>;                                                                   \
        BE_NT_VEC_AT0(kern);                                                                                                 \
        BE_HIP(be_allow_lds(reinterpret_cast<const void*>(kern), (int)vlds));                                               \
        hipLaunchKernelGGL(kern, dim3(grid, (unsigned)nb), dim3(1024), vlds, st, static_cast<const W*>(weights), indices, rp, \
                           bits, n_words, static_cast<W*>(out), m, coarse, n_cwords, g_shift);                              \
      }                                                                                                                      \
    } while (0)
    // two passes of a row (8 x LPR entries) should hold nearly every row: LPR by the average length
    // — rows of one known length (no indptr: FixedNumConn) have no long tail to provide for: 8 / 16 lanes up to exactly
    // two passes (64 / 128 entries; measured +5 % at 48 ... 64 per row).  Fewer lanes with both passes in use lose
    // (12 ... 32 per row on 2 / 4 lanes instead of 4 / 8: 0.30-0.33 -> 0.36 ms at 2e8 weighted entries).
    const bool fixed = rp.p == nullptr;
    if (avg <= 8) BE_NT_VEC(2);
    else if (avg <= 20) BE_NT_VEC(4);
    else if (avg <= (fixed ? 64 : 45)) BE_NT_VEC(8);
    else if (avg <= (fixed ? 128 : 100)) BE_NT_VEC(16);
    else BE_NT_VEC(32);
#undef BE_NT_VEC
#undef BE_NT_VEC_AT0
    be_prof_end(prof, st);
    BE_LAUNCH_CHECK();
    return BE_OK;
  }
  {
    const int grid = grid_for(m, 16, nb >= 8 ? 256 : 512);
    const int prof = be_prof_begin(st);
    if (in_lds) {
      auto kern = k_csrmv_nt_wave<W, HOMO, true>;
      BE_HIP(be_allow_lds(reinterpret_cast<const void*>(kern), (int)lds));
      hipLaunchKernelGGL(kern, dim3(grid, (unsigned)nb), dim3(1024), lds, st, static_cast<const W*>(weights), indices, rp,
                         bits, n_words, static_cast<W*>(out), m, coarse, n_cwords, g_shift);
    } else {
      auto kern = k_csrmv_nt_wave<W, HOMO, false>;
      BE_HIP(be_allow_lds(reinterpret_cast<const void*>(kern), (int)lds));
      hipLaunchKernelGGL(kern, dim3(grid, (unsigned)nb), dim3(1024), lds, st, static_cast<const W*>(weights), indices, rp,
                         bits, n_words, static_cast<W*>(out), m, coarse, n_cwords, g_shift);
    }
    be_prof_end(prof, st);
    BE_LAUNCH_CHECK();
    return BE_OK;
  }
}


template <typename W>
int csrmv_nt_indexed(const void* weights, const int32_t* indices, RowPtr rp, const void* perm, int perm64, const void* spikes,
                     int sd, void* out, int64_t m, int64_t k, int64_t nb, void* ws, hipStream_t st) {
  const int64_t n_words = (k + 31) / 32;
  const uint32_t* bits = static_cast<const uint32_t*>(spikes);
  if (sd != BE_SPIKE_BITS) {
    int rc = pack_any(spikes, sd, k, nb, static_cast<uint32_t*>(ws), n_words, st);
    if (rc != BE_OK) return rc;
    bits = static_cast<const uint32_t*>(ws);
  }
  if (m == 0 || nb == 0) return BE_OK;
  const int grid = grid_for(m, 4, nb >= 8 ? 1024 : 4096);
  const int prof = be_prof_begin(st);
  if (perm64)
    hipLaunchKernelGGL((k_csrmv_nt_indexed<W, int64_t>), dim3(grid, (unsigned)nb), dim3(256), 0, st, static_cast<const W*>(weights),
                       indices, rp, static_cast<const int64_t*>(perm), bits, n_words, static_cast<W*>(out), m);
  else
    hipLaunchKernelGGL((k_csrmv_nt_indexed<W, int32_t>), dim3(grid, (unsigned)nb), dim3(256), 0, st, static_cast<const W*>(weights),
                       indices, rp, static_cast<const int32_t*>(perm), bits, n_words, static_cast<W*>(out), m);
  be_prof_end(prof, st);
  BE_LAUNCH_CHECK();
  return BE_OK;
}

}  // namespace

// the single definition of the shared active-list resolver (declared in be_csr_shared.h)
int be_resolve_active(const void* spikes, int sd, int64_t n, int64_t nb, uint32_t* ws_active, int64_t astride,
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

// the same with the zeroing of the counters left to the caller (be_jitc.hip: a workspace armed by be_jit_scatter_workspace_arm
// holds zero counters on entry and is re-armed by the step's last kernel); not part of the public header
int be_internal_compact_spikes_batched(const void* spikes_bm, int spike_dtype, int64_t n, int64_t n_batch, uint32_t* active_ids,
                                       int64_t active_stride, uint32_t* counts, int zero_first, be_stream_t stream) {
  BE_REQUIRE(n >= 0 && n <= 0xffffffffll && n_batch >= 0 && n_batch <= kMaxBatch, BE_ERR_INVALID, "shape out of range");
  BE_REQUIRE(counts && (n == 0 || n_batch == 0 || (spikes_bm && active_ids)), BE_ERR_INVALID, "null pointer");
  return compact_any(spikes_bm, spike_dtype, n, n_batch, active_ids, active_stride, counts, static_cast<hipStream_t>(stream),
                     zero_first != 0);
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
  const int64_t bits = (((k + 31) / 32) * n_batch + 2 + gather_cwords(k, gather_g_shift(k)) * n_batch) * 4;   // per-column bitmaps (+ coarse ones)
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

// ---------------------------------------------------------------- perm-fused ("indexed") products
int be_binary_csrmm_t_indexed(const void* weights, int homo, int wdtype, const int32_t* indices, const void* indptr,
                              int indptr_is_i64, int64_t row_len, const void* perm, int perm_is_i64, const void* spikes,
                              int spike_dtype, void* out, int64_t m, int64_t k, int64_t n_batch, void* workspace,
                              int64_t workspace_bytes, be_stream_t stream) {
  if (homo || perm == nullptr)
    return be_binary_csrmm_t(weights, homo, wdtype, indices, indptr, indptr_is_i64, row_len, spikes, spike_dtype, out, m, k,
                             n_batch, workspace, workspace_bytes, stream);
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
  BE_DISPATCH_W(wdtype, 0, return (csrmv_t_direct<W, HOMO>(weights, indices, rp, spikes, spike_dtype, out, m, k, n_batch, workspace, st, perm, perm_is_i64)));
  return BE_OK;
}

int be_binary_csrmm_nt_indexed(const void* weights, int homo, int wdtype, const int32_t* indices, const void* indptr,
                               int indptr_is_i64, int64_t row_len, const void* perm, int perm_is_i64, const void* spikes,
                               int spike_dtype, void* out, int64_t m, int64_t k, int64_t n_batch, void* workspace,
                               int64_t workspace_bytes, be_stream_t stream) {
  if (homo || perm == nullptr)
    return be_binary_csrmm_nt(weights, homo, wdtype, indices, indptr, indptr_is_i64, row_len, spikes, spike_dtype, out, m, k,
                              n_batch, workspace, workspace_bytes, stream);
  BE_REQUIRE(m >= 0 && k >= 0, BE_ERR_INVALID, "bad shape");
  BE_REQUIRE(n_batch >= 0 && n_batch <= kMaxBatch, BE_ERR_INVALID, "n_batch out of range");
  BE_REQUIRE(check_rows(indptr, row_len), BE_ERR_INVALID, "indptr is NULL and row_len < 0");
  BE_REQUIRE(weights != nullptr, BE_ERR_INVALID, "weights is NULL");
  BE_REQUIRE(m == 0 || n_batch == 0 || out != nullptr, BE_ERR_INVALID, "out is NULL");
  BE_REQUIRE(k == 0 || n_batch == 0 || spikes != nullptr, BE_ERR_INVALID, "spikes is NULL");
  BE_REQUIRE(spike_dtype == BE_SPIKE_BOOL || spike_dtype == BE_SPIKE_FLOAT || spike_dtype == BE_SPIKE_BITS, BE_ERR_INVALID,
             "unknown spike dtype");
  BE_REQUIRE(workspace != nullptr && workspace_bytes >= be_binary_csrmm_nt_workspace_bytes(m, k, n_batch),
             BE_ERR_WORKSPACE, "workspace too small");
  RowPtr rp{indptr, indptr_is_i64, row_len};
  hipStream_t st = static_cast<hipStream_t>(stream);
  switch (wdtype) {
    case BE_F32: return csrmv_nt_indexed<float>(weights, indices, rp, perm, perm_is_i64, spikes, spike_dtype, out, m, k, n_batch, workspace, st);
    case BE_F64: return csrmv_nt_indexed<double>(weights, indices, rp, perm, perm_is_i64, spikes, spike_dtype, out, m, k, n_batch, workspace, st);
    case BE_F16: return csrmv_nt_indexed<__half>(weights, indices, rp, perm, perm_is_i64, spikes, spike_dtype, out, m, k, n_batch, workspace, st);
    case BE_BF16: return csrmv_nt_indexed<__hip_bfloat16>(weights, indices, rp, perm, perm_is_i64, spikes, spike_dtype, out, m, k, n_batch, workspace, st);
    default: be_set_error("unknown weight dtype"); return BE_ERR_INVALID;
  }
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
