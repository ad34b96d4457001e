// be_csr_plan.hip — the post-sliced scatter plan behind `spk @ CSR` / `spk @ FixedNumPerPre` on gfx950: build kernels
// (count -> scan -> fill, the uint16 / d8 / h8 block layouts), the LDS-accumulating step kernels, the reduce and the
// single-launch kernel for small matrices.  Semantics as in be_csr.hip (reference brainevent/_csr/binary.py:387-489,
// transpose=True); the role of the reference's per-matrix task workspace (brainevent/_csr/main.py:58-88).
#include "be_csr_shared.h"
#include <climits>
#include <cstring>
#ifndef BE_BLOCK_AUX
#define BE_BLOCK_AUX 2      // cache policy of the wave-per-block d8 loads: nt — blocks are streamed once per step (C2: kernel 119.6 -> 108-112 us,
                            // 716 -> 750-780 Geff/s; sc0 / sc1 on top change nothing).  NOT on the other decoders: uint16 blocks by part of a wave
                            // 26 -> 32 us at N = 350k, 162 -> 186 at 2.5M; h8 at C2 1522 -> 1489 Geff/s.  And only for long blocks
                            // (kD8NtMinBlock items on average): on one post slice of an 8-way cut of C2 (blocks of ~114 items, 5 lines
                            // each) the kernel takes 32.5 us with nt and 30.3 without.
#endif
constexpr int kD8NtMinBlock = 160;

namespace {

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
// fused step: the accumulate kernel lists the active rows of ITS part itself (no compaction launch, no active list in
// memory, one dependent round trip less).  Rows are dealt to the parts in stripes of 1024 (32 words of the bit-packed
// vector / 1024 spike bytes): part p owns stripes p, p + parts, ... — a different split of the rows than the interleaved
// list positions of the unfused path, which integer sums do not see.  Every workgroup of a part builds the same list, in
// ascending row order (block scans, no atomics), into LDS behind its accumulators; a list that does not fit LDS goes to
// the part's region of the workspace instead (all workgroups of the part write identical values there).
// FUSED: 0 = list from the workspace (compaction kernel or the caller's ids), 1 = bit-packed spikes, 2 = 1-byte spikes.
// =================================================================================================
template <int FUSED>
__device__ __forceinline__ uint32_t fused_word(const void* __restrict__ spikes, int64_t gw, int64_t m) {
  uint32_t w = 0;
  if (FUSED == 1) {
    w = static_cast<const uint32_t*>(spikes)[gw];
  } else {
    const uint8_t* sp = static_cast<const uint8_t*>(spikes) + gw * 32;
    if (gw * 32 + 32 <= m) {
      const uint4 a = reinterpret_cast<const uint4*>(sp)[0], b = reinterpret_cast<const uint4*>(sp)[1];
      const uint32_t v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
      for (int q = 0; q < 8; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) w |= (((v[q] >> (8 * e)) & 0xffu) != 0u ? 1u : 0u) << (4 * q + e);
    } else {
      for (int e = 0; e < 32 && gw * 32 + e < m; ++e) w |= (sp[e] != 0 ? 1u : 0u) << e;
    }
  }
  const int64_t left = m - gw * 32;                        // spikes of the population in this word
  return left >= 32 ? w : (left > 0 ? (w & ((1u << (uint32_t)left) - 1u)) : 0u);
}

// inclusive scan over the 1024 threads + the block total with ONE barrier: consecutive calls alternate between two
// arrays of wave totals, so a call never overwrites totals a slower thread of the previous call may still be reading
__device__ __forceinline__ uint32_t block_scan_total_1024(uint32_t v, uint32_t* wave_tot /* [16] of this call */, uint32_t* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t incl = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t t = __shfl_up(incl, off, 64);
    if (lane >= off) incl += t;
  }
  if (lane == 63) wave_tot[wave] = incl;
  __syncthreads();
  uint32_t base = 0, t = 0;
#pragma unroll
  for (int w = 0; w < 16; ++w) {
    const uint32_t x = wave_tot[w];
    t += x;
    if (w < wave) base += x;
  }
  *total = t;
  return base + incl;
}

template <int FUSED>
__device__ __forceinline__ uint32_t build_part_list(const void* __restrict__ spikes, int64_t m, int part, int parts,
                                                    uint32_t* lds_list, uint32_t lds_cap, uint32_t* glob_list,
                                                    uint32_t* wave_tot /* [32] */, const uint32_t** list_out) {
  const int64_t n_words = (m + 31) >> 5;
  const int64_t n_stripes = (n_words + 31) >> 5;
  const int64_t my_words = part < n_stripes ? ((n_stripes - part + parts - 1) / parts) * 32 : 0;
  uint32_t* dst = lds_list;
  uint32_t base = 0;
  int buf = 0;
  for (int64_t u0 = 0; u0 < my_words; u0 += 1024) {
    const int64_t u = u0 + threadIdx.x;
    const int64_t gw = ((u >> 5) * parts + part) * 32 + (u & 31);
    uint32_t w = (u < my_words && gw < n_words) ? fused_word<FUSED>(spikes, gw, m) : 0u;
    const uint32_t v = __popc(w);
    uint32_t it_total;
    uint32_t pos = base + block_scan_total_1024(v, wave_tot + 16 * buf, &it_total) - v;
    buf ^= 1;
    if (dst == lds_list && base + it_total > lds_cap) {     // uniform: the list outgrows LDS -> start over into the workspace
      dst = glob_list;
      base = 0;
      u0 = -1024;
      continue;
    }
    while (w) {
      const uint32_t b = __ffs(w) - 1u;
      dst[pos++] = (uint32_t)(gw * 32) + b;
      w &= w - 1u;
    }
    base += it_total;
  }
  __syncthreads();
  *list_out = dst;
  return base;
}

// =================================================================================================
// Pre-gathered segment table (short blocks).  With ~20 entries per block the step is bound by its segment-table gather:
// one load instruction of a wave fetches the 8-byte entries of 64 different rows — 64 different 128-byte lines — and
// compiling that gather out (made-up block addresses) cut the weighted kernel at N = 1M, K = 1000 from 65 to 17 us.
// k_gather_seg runs between the compaction and the accumulate kernel: it reads the active rows' table rows (n_slices x
// 8 B contiguous: coalesced) and writes them transposed, dense by LIST POSITION: dense[slice][position].  The accumulate
// kernel then gives every part a contiguous range of positions and every wave 64 consecutive ones, reads its 64 entries as
// one 512-byte load, and never loads a row id at all (FUSED = 3).
// =================================================================================================
__global__ void __launch_bounds__(256) k_gather_seg(const uint2* __restrict__ seg, const uint32_t* __restrict__ active,
                                                    const uint32_t* __restrict__ n_active_p, int n_slices, int64_t a_stride,
                                                    uint2* __restrict__ dense) {
  __shared__ uint2 tile[64][65];
  const uint32_t n_active = *n_active_p;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (uint32_t c = blockIdx.x; (uint64_t)c * 64 < n_active; c += gridDim.x) {
    const uint32_t a0 = c * 64;
    const uint32_t n_here = n_active - a0 < 64u ? n_active - a0 : 64u;
    // a wave takes 16 of the 64 rows: their ids in one load, then 16 independent row reads in flight (two memory latencies
    // per tile instead of 32)
    const uint32_t i_mine = (uint32_t)wave * 16u + (uint32_t)(lane & 15);
    const uint32_t rid = active[a0 + (i_mine < n_here ? i_mine : 0u)];
    for (int s0 = 0; s0 < n_slices; s0 += 64) {
      const int sl = s0 + lane < n_slices ? s0 + lane : n_slices - 1;
      uint2 v[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) v[j] = seg[(uint64_t)__builtin_amdgcn_readlane((int)rid, j) * n_slices + sl];
#pragma unroll
      for (int j = 0; j < 16; ++j) tile[wave * 16 + j][lane] = v[j];
      __syncthreads();
      for (int s = wave; s < 64 && s0 + s < n_slices; s += 4)      // wave: one slice; lanes: 64 consecutive positions
        if ((uint32_t)lane < n_here) dense[(int64_t)(s0 + s) * a_stride + a0 + lane] = tile[lane][s];
      __syncthreads();
    }
  }
}

// activity bits of rows [16 g, 16 g + 16) (bit-packed words or 1-byte spikes)
template <int FUSED>
__device__ __forceinline__ uint32_t fused_half(const void* __restrict__ spikes, int64_t g, int64_t m) {
  uint32_t w = 0;
  if (FUSED == 1) {
    w = (static_cast<const uint32_t*>(spikes)[g >> 1] >> ((uint32_t)(g & 1) * 16u)) & 0xffffu;
  } else {
    const uint8_t* sp = static_cast<const uint8_t*>(spikes) + g * 16;
    if (g * 16 + 16 <= m) {
      const uint4 a = reinterpret_cast<const uint4*>(sp)[0];
      const uint32_t v[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) w |= (((v[q] >> (8 * e)) & 0xffu) != 0u ? 1u : 0u) << (4 * q + e);
    } else {
      for (int e = 0; e < 16 && g * 16 + e < m; ++e) w |= (sp[e] != 0 ? 1u : 0u) << e;
    }
  }
  const int64_t left = m - g * 16;
  return left >= 16 ? w : (left > 0 ? (w & ((1u << (uint32_t)left) - 1u)) : 0u);
}

// Compaction and gather in one launch (bit-packed or 1-byte spikes): a workgroup scans 4096 rows, reserves its range of list
// positions with one atomic — any order of the positions gives the same sums: the accumulators are integers — and gathers its
// own rows' table entries straight from the ids it holds in LDS; the id list itself never reaches memory.
template <int FUSED>
__global__ void __launch_bounds__(256) k_compact_gather_seg(const void* __restrict__ spikes, int64_t m,
                                                            const uint2* __restrict__ seg, int n_slices, int64_t a_stride,
                                                            uint2* __restrict__ dense, uint32_t* __restrict__ count) {
  __shared__ uint2 tile[64][65];
  __shared__ uint32_t ids_s[4096];
  __shared__ uint32_t wtot[4], s_base;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t n_groups = (m + 15) >> 4;
  const int64_t gw = (int64_t)blockIdx.x * 256 + threadIdx.x;
  uint32_t w = gw < n_groups ? fused_half<FUSED>(spikes, gw, m) : 0u;
  const uint32_t v = __popc(w);
  uint32_t incl = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t t = __shfl_up(incl, off, 64);
    if (lane >= off) incl += t;
  }
  if (lane == 63) wtot[wave] = incl;
  __syncthreads();
  uint32_t wave_off = 0, total = 0;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if (q < wave) wave_off += wtot[q];
    total += wtot[q];
  }
  if (total == 0) return;                                   // uniform
  if (threadIdx.x == 0) s_base = atomicAdd(count, total);
  uint32_t pos = wave_off + incl - v;
  while (w) {
    const uint32_t b = __ffs(w) - 1u;
    ids_s[pos++] = (uint32_t)(gw * 16) + b;
    w &= w - 1u;
  }
  __syncthreads();
  const uint32_t base = s_base;
  for (uint32_t c0 = 0; c0 < total; c0 += 64) {
    const uint32_t n_here = total - c0 < 64u ? total - c0 : 64u;
    const uint32_t i_mine = (uint32_t)wave * 16u + (uint32_t)(lane & 15);
    const uint32_t rid = ids_s[c0 + (i_mine < n_here ? i_mine : 0u)];
    for (int s0 = 0; s0 < n_slices; s0 += 64) {
      const int sl = s0 + lane < n_slices ? s0 + lane : n_slices - 1;
      uint2 x[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) x[j] = seg[(uint64_t)__builtin_amdgcn_readlane((int)rid, j) * n_slices + sl];
#pragma unroll
      for (int j = 0; j < 16; ++j) tile[wave * 16 + j][lane] = x[j];
      __syncthreads();
      for (int s = wave; s < 64 && s0 + s < n_slices; s += 4)
        if ((uint32_t)lane < n_here) dense[(int64_t)(s0 + s) * a_stride + base + c0 + lane] = tile[lane][s];
      __syncthreads();
    }
  }
}

// =================================================================================================
// planned scatter step
// =================================================================================================
// One group = up to 4 row segments whose first 64 lane-groups are in flight together.
// Loads go through raw buffer descriptors built per segment from wave-uniform (base, length): lanes
// past the end of a segment are range-checked by the hardware (no traffic, zeros returned), so the
// loads need no exec-mask branches and hipcc can keep *counted* vmcnt waits — with conditional
// global loads it falls back to vmcnt(0) before every load and the kernel runs one segment at a time.

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
// ---- counted entries, short blocks: LPB lanes per block (8 columns per lane), 64 / LPB blocks per load instruction.
// With 31 entries per block (N = 1M, K = 1000, one shared weight) the wave-per-block path keeps 4 of 64 lanes busy.
struct SubGroup {
  uint32_t n;                   // lane-groups in this lane's block (8 counted columns, or 4 weighted entries, per lane)
  be_v4u c;                     // 8 uint16 columns (counted) / 4 weights
  be_v2u iv;                    // weighted: 4 uint16 columns
  const unsigned char* blk;
};
template <bool HOMO, int LPB>
__device__ __forceinline__ void sub_issue(SubGroup& g, int i, int nvalid, uint32_t st_v, uint32_t n4_v, int lane,
                                          const unsigned char* __restrict__ blob) {
  const int row = i + lane / LPB;
  const uint32_t st = (uint32_t)__shfl((int)st_v, row & 63, 64);
  const uint32_t n = (uint32_t)__shfl((int)n4_v, row & 63, 64);
  g.n = row < nvalid ? n : 0u;
  g.blk = g.n ? blob + ((uint64_t)st << 7) : blob;          // nothing to do: the head of the blob (in bounds, cached)
  const uint32_t l = (uint32_t)(lane % LPB);
  const uint32_t o = l < g.n ? l : 0u;                       // clamped: the loads stay unconditional (counted vmcnt waits)
  g.c = *(reinterpret_cast<const be_v4u*>(g.blk) + o);
  if constexpr (!HOMO) g.iv = *(reinterpret_cast<const be_v2u*>(g.blk + (uint64_t)g.n * 16u) + o);
}
template <bool HOMO, int LPB>
__device__ __forceinline__ void sub_consume(SubGroup& g, typename PlanAcc<HOMO>::type* acc, int lane, float scale) {
  // an opaque use of the loaded registers in straight-line code: without it the compiler may sink a load into the `l < n`
  // branch below, where its wait (vmcnt(0)) drains every load the pipeline has in flight
  if constexpr (HOMO) asm volatile("" : "+v"(g.c.x), "+v"(g.c.y), "+v"(g.c.z), "+v"(g.c.w));
  else asm volatile("" : "+v"(g.c.x), "+v"(g.c.y), "+v"(g.c.z), "+v"(g.c.w), "+v"(g.iv.x), "+v"(g.iv.y));
  const uint32_t l = (uint32_t)(lane % LPB);
  if (l < g.n) {                                             // the first LPB lane-groups; longer blocks: sub_tails
    if constexpr (HOMO) {
      plan_count8(reinterpret_cast<uint32_t*>(acc), g.c.x, g.c.y, g.c.z, g.c.w);
    } else {
      plan_add4<false>(acc, make_uint2(g.iv.x, g.iv.y),
                       make_float4(__uint_as_float(g.c.x), __uint_as_float(g.c.y), __uint_as_float(g.c.z), __uint_as_float(g.c.w)), scale);
    }
  }
}
// The rare blocks longer than one sub-wave pass (n > LPB lane-groups), after the batch's pipelined loop, a wave per block.
// The lanes-per-block variant is chosen so that these are ~0.1 % of the blocks (host: mean + 3 sigma fits one pass).
template <bool HOMO, int LPB>
__device__ __forceinline__ void sub_tails(uint32_t st_v, uint32_t n4_v, int nvalid, typename PlanAcc<HOMO>::type* acc, int lane,
                                          float scale, const unsigned char* __restrict__ blob) {
  unsigned long long longm = __ballot(lane < nvalid && n4_v > (uint32_t)LPB);
  while (longm) {
    const int src = __ffsll((long long)longm) - 1;
    longm &= longm - 1;
    const uint32_t st = (uint32_t)__builtin_amdgcn_readlane((int)st_v, src), n = (uint32_t)__builtin_amdgcn_readlane((int)n4_v, src);
    const unsigned char* blk = blob + ((uint64_t)st << 7);
    for (uint32_t o = LPB + lane; o < n; o += 64) {
      const uint4 x = reinterpret_cast<const uint4*>(blk)[o];
      if constexpr (HOMO) {
        plan_count8(reinterpret_cast<uint32_t*>(acc), x.x, x.y, x.z, x.w);
      } else {
        const uint2 iv = reinterpret_cast<const uint2*>(blk + (uint64_t)n * 16u)[o];
        plan_add4<false>(acc, iv, make_float4(__uint_as_float(x.x), __uint_as_float(x.y), __uint_as_float(x.z), __uint_as_float(x.w)), scale);
      }
    }
  }
}

#ifndef BE_PARTIAL_NT
#define BE_PARTIAL_NT 1     // bit 0: the d8 kernel's partial sums leave with non-temporal stores, bit 1: the u16 kernels',
                            // bit 2: the h8 kernel's (nobody reads them before the reduce launch; C2: 754-761 -> 777-781 Geff/s)
#endif
// a task's accumulators, LDS -> its partial sums (n16 pieces of 16 bytes)
template <int BIT>
__device__ __forceinline__ void store_partials(const unsigned char* smem_raw, void* partial, int n16) {
  const uint4* src = reinterpret_cast<const uint4*>(smem_raw);
  uint4* dst = reinterpret_cast<uint4*>(partial);
  typedef unsigned pv4 __attribute__((ext_vector_type(4)));
  for (int i = threadIdx.x; i < n16; i += blockDim.x) {
    const uint4 v = src[i];
    if (BE_PARTIAL_NT & BIT) __builtin_nontemporal_store(pv4{v.x, v.y, v.z, v.w}, reinterpret_cast<pv4*>(dst) + i);
    else dst[i] = v;
  }
}
template <bool HOMO, int LPB = 0, int FUSED = 0>
__global__ void __launch_bounds__(1024) k_plan_accumulate(const unsigned char* __restrict__ blob, const uint2* __restrict__ seg,
                                                          const uint32_t* active,
                                                          const uint32_t* __restrict__ n_active_p, int n_slices,
                                                          int slice_shift, int parts, float scale,
                                                          typename PlanAcc<HOMO>::type* __restrict__ partial,
                                                          int64_t active_stride, int stride,
                                                          const void* __restrict__ fused_spikes, int64_t fused_m, int64_t fused_stride,
                                                          uint32_t list_off, uint32_t lds_cap, uint32_t* __restrict__ glob_lists,
                                                          int64_t glob_region) {
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
  if constexpr (FUSED == 0) active += (int64_t)blockIdx.y * active_stride;
  partial += ((int64_t)blockIdx.y * n_tasks + L) * stride;
  {
    uint4* z = reinterpret_cast<uint4*>(smem_raw);
    const int n16 = (int)(((size_t)(S + 1) * sizeof(acc_t) + 15) / 16);
    for (int i = threadIdx.x; i < n16; i += blockDim.x) z[i] = make_uint4(0u, 0u, 0u, 0u);
  }
  __syncthreads();

  __shared__ uint32_t fused_wtot[32];
  uint32_t n_active;
  if constexpr (FUSED == 1 || FUSED == 2) {       // list the active rows of this part in LDS behind the accumulators (or in the part's region)
    n_active = build_part_list<FUSED>(static_cast<const unsigned char*>(fused_spikes) + (int64_t)blockIdx.y * fused_stride, fused_m,
                                      part, parts, reinterpret_cast<uint32_t*>(smem_raw + list_off), lds_cap,
                                      glob_lists + ((int64_t)blockIdx.y * parts + part) * glob_region, fused_wtot, &active);
  } else {
    n_active = n_active_p[blockIdx.y];
  }
  // FUSED == 3 (pre-gathered table): part p owns the list positions [p * npp, (p + 1) * npp), a wave 64 consecutive ones
  uint32_t pos_lo = 0;
  if constexpr (FUSED == 3) {
    const uint32_t npp = (n_active + (uint32_t)parts - 1u) / (uint32_t)parts;
    pos_lo = (uint32_t)part * npp;
    const uint32_t hi = pos_lo + npp < n_active ? pos_lo + npp : n_active;
    n_active = hi > pos_lo ? hi : pos_lo;            // from here on: one past this part's last position
  }
  // segment table: [row][slice] as built, or (FUSED == 3) the step's pre-gathered [slice][list position]
  const uint2* sp = FUSED == 3 ? seg + (int64_t)slice * glob_region : seg + slice;
  const uint64_t seg_rs = FUSED == 3 ? 1u : (uint64_t)n_slices;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
  // list position of this lane's row in batch b of this wave
  const uint64_t a0 = FUSED == 3 ? (uint64_t)pos_lo + (uint64_t)wave * 64 + lane
                      : FUSED  ? (uint64_t)wave + (uint64_t)nw * lane
                               : (uint64_t)part + (uint64_t)parts * ((uint64_t)wave + (uint64_t)nw * lane);
  const uint64_t a_step = FUSED ? (uint64_t)nw * 64 : (uint64_t)parts * nw * 64;

  // pointer pipeline: rows of batch b+2 | bounds of batch b+1 | work on batch b.  All pointer loads are
  // unconditional (clamped index, result masked) for the same counted-vmcnt reason as above.
  if (n_active > pos_lo) {
    const uint64_t last = n_active - 1;
    uint64_t a = a0;
    bool v_n = a < n_active;
    uint32_t r_n = FUSED == 3 ? (uint32_t)(a < last ? a : last) : active[a < last ? a : last];
    a += a_step;
    uint2 sg = sp[(uint64_t)r_n * seg_rs];
    uint32_t st_v = sg.x, n4_v = v_n ? sg.y : 0u;
    bool v_c = v_n;
    v_n = a < n_active;
    r_n = FUSED == 3 ? (uint32_t)(a < last ? a : last) : active[a < last ? a : last];
    a += a_step;

    while (__ballot(v_c) != 0ull) {
      const int nvalid = __popcll(__ballot(v_c));   // valid lanes form a prefix: a grows with the lane
      // issue next batch's bounds and the batch-after-next's row ids before touching this batch's data
      const uint2 sgn = sp[(uint64_t)r_n * seg_rs];
      const bool v_nn = a < n_active;
      const uint32_t r_nn = FUSED == 3 ? (uint32_t)(a < last ? a : last) : active[a < last ? a : last];
      a += a_step;

      if constexpr (LPB > 0) {
        // straight-line over the batch's 64 / BPI groups, DEPTH groups ahead (see k_plan_accumulate_d8)
        constexpr int BPI = 64 / LPB;                       // blocks per load instruction
        constexpr int NGRP = 64 / BPI;
        constexpr int DEPTH = NGRP < 4 ? NGRP : 4;
        SubGroup g[NGRP];
#pragma unroll
        for (int q = 0; q < DEPTH; ++q) sub_issue<HOMO, LPB>(g[q], q * BPI, nvalid, st_v, n4_v, lane, blob);
#pragma unroll
        for (int q = 0; q < NGRP; ++q) {
          if (q + DEPTH < NGRP) sub_issue<HOMO, LPB>(g[q + DEPTH], (q + DEPTH) * BPI, nvalid, st_v, n4_v, lane, blob);
          sub_consume<HOMO, LPB>(g[q], acc, lane, scale);
        }
        sub_tails<HOMO, LPB>(st_v, n4_v, nvalid, acc, lane, scale, blob);
      } else {
        SegGroup gA, gB;
        seg_issue<HOMO>(gA, 0, nvalid, st_v, n4_v, lane, blob);
        for (int i = 0; i < nvalid; i += 8) {
          seg_issue<HOMO>(gB, i + 4, nvalid, st_v, n4_v, lane, blob);
          seg_consume<HOMO>(gA, acc, lane, scale, blob);
          seg_issue<HOMO>(gA, i + 8, nvalid, st_v, n4_v, lane, blob);
          seg_consume<HOMO>(gB, acc, lane, scale, blob);
        }
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
    store_partials<2>(smem_raw, partial, (int)((size_t)stride * sizeof(acc_t) / 16));
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

// one compare-exchange stage (stride j of merge width k2) of the bitonic network over keys[0, n2) in LDS
__device__ __forceinline__ void d8_lds_stage(unsigned long long* keys, int n2, int k2, int j) {
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
// loads the row's (column, position) keys into LDS and sorts them; returns the row length.  `order_out` (count pass): the
// sorted order — the row-local position of the i-th smallest column, uint16 — is written to order_out[b + i]; `order_in`
// (fill / weight refresh): the order is read back instead of sorting again (a gather of the row's columns: the row is one
// 40-KB window at most).  One of the two sorts of a build and every sort of a weight refresh disappear.
// Round 4: the sort is a counting sort by column range followed by a rank sort inside each bucket — the row's columns are cut
// into ~len / 48 equal ranges, every entry draws its place in its bucket from an LDS counter, and a wave then orders a bucket
// in place by counting, for each of its (at most 4 per lane) keys, the keys of the bucket below it (every key of the bucket is
// read once by the whole wave: a broadcast LDS read; no barrier inside a bucket).  One bitonic sort of 16384 keys was 105
// compare-exchange stages with a barrier each (C2: 648 ms of a 850 ms build); this is 4 barriers per row.  Keys are unique
// (the position is part of the key), so the result is THE sorted order whatever the arrival order at the counters.  Rows
// whose columns cluster (a bucket above 256 keys) take the bitonic network as before; rows that already ascend skip the sort.
__device__ __forceinline__ int d8_sort_row(unsigned long long* keys, const int32_t* __restrict__ indices, int64_t b, int64_t e,
                                           const uint16_t* __restrict__ order_in = nullptr,
                                           uint16_t* __restrict__ order_out = nullptr, uint32_t col_range = 0) {
  // caller contract: rows have at most kD8MaxRow entries (the Python side checks it and falls back to the u16 layout);
  // a longer row is cut here rather than written past the LDS array
  const int len = (e - b) > (int64_t)kD8MaxRow ? kD8MaxRow : (int)(e - b);
  constexpr int kPer = kD8MaxRow / 1024;            // entries a thread holds (blockDim.x == 1024)
  if (order_in != nullptr) {
    if (blockDim.x == 1024 && len > 0) {
      // all positions first, then all column gathers: 16 loads in flight per thread instead of a dependent pair per iteration
      // (a load under `if (i < len)` is waited for on the spot — vmcnt(0) — before the next one is issued)
      uint32_t pv[kPer], cv[kPer];
#pragma unroll
      for (int q = 0; q < kPer; ++q) {
        const int i = threadIdx.x + q * 1024;
        pv[q] = order_in[b + (i < len ? i : len - 1)];
      }
#pragma unroll
      for (int q = 0; q < kPer; ++q) cv[q] = (uint32_t)indices[b + pv[q]];
#pragma unroll
      for (int q = 0; q < kPer; ++q) {
        const int i = threadIdx.x + q * 1024;
        if (i < len) keys[i] = ((unsigned long long)cv[q] << 16) | (unsigned long long)pv[q];
      }
    } else {
      for (int i = threadIdx.x; i < len; i += blockDim.x) {
        const uint32_t p = order_in[b + i];
        keys[i] = ((unsigned long long)(uint32_t)indices[b + p] << 16) | (unsigned long long)p;
      }
    }
    __syncthreads();
    return len;
  }
  __shared__ uint32_t bk_cnt[1024], bk_start[1024], bk_wtot[16];
  bool sorted_done = false;
  if (col_range != 0u && len > 256 && blockDim.x == 1024) {
    // buckets of ~48 keys: the rank sort costs (bucket size) compares per key, and a bucket of up to 64 keys is one key per lane
    const uint32_t nb = (uint32_t)((len + 47) / 48) < 1024u ? (uint32_t)((len + 47) / 48) : 1024u;
    const uint32_t bw = (col_range + nb - 1u) / nb;                                                    // columns per bucket
    const uint32_t magic = (uint32_t)(0xffffffffull / bw);                // floor((2^32 - 1) / bw): the quotient below is <= 1 short
    bk_cnt[threadIdx.x] = 0u;
    __syncthreads();
    uint32_t colv[kPer], rk[kPer], prevv[kPer];
    int unsorted = 0;
    // every load of the row is issued before the first use (clamped addresses instead of branches: see above)
#pragma unroll
    for (int q = 0; q < kPer; ++q) {
      const int i = threadIdx.x + q * 1024;
      const int ii = i < len ? i : len - 1;
      colv[q] = (uint32_t)indices[b + ii];
      prevv[q] = (uint32_t)indices[b + (ii > 0 ? ii - 1 : 0)];
    }
#pragma unroll
    for (int q = 0; q < kPer; ++q) {
      const int i = threadIdx.x + q * 1024;
      rk[q] = 0u;
      if (i < len) {
        const uint32_t c = colv[q];
        uint32_t sb = __umulhi(c, magic);
        if (c - sb * bw >= bw) ++sb;
        sb = sb < nb ? sb : nb - 1u;                  // (a column >= col_range is the caller's error: kept in the last bucket)
        rk[q] = (sb << 16) | atomicAdd(&bk_cnt[sb], 1u);
        if (i > 0 && c < prevv[q]) unsorted = 1;
      }
    }
    if (!__syncthreads_or(unsorted)) {                // canonical rows: (column << 16 | position) ascends with the columns
#pragma unroll
      for (int q = 0; q < kPer; ++q) {
        const int i = threadIdx.x + q * 1024;
        if (i < len) keys[i] = ((unsigned long long)colv[q] << 16) | (unsigned long long)i;
      }
      __syncthreads();
      sorted_done = true;
    } else {
      const uint32_t mine = threadIdx.x < nb ? bk_cnt[threadIdx.x] : 0u;
      const int too_big = mine > 256u ? 1 : 0;
      const uint32_t incl = block_scan_1024(mine, bk_wtot);              // (two barriers)
      bk_start[threadIdx.x] = incl - mine;
      if (!__syncthreads_or(too_big)) {
#pragma unroll
        for (int q = 0; q < kPer; ++q) {
          const int i = threadIdx.x + q * 1024;
          if (i < len) keys[bk_start[rk[q] >> 16] + (rk[q] & 0xffffu)] = ((unsigned long long)colv[q] << 16) | (unsigned long long)i;
        }
        __syncthreads();
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        for (uint32_t bk = (uint32_t)wave; bk < nb; bk += 16u) {        // a wave orders a bucket in place
          const uint32_t B = bk_cnt[bk];
          unsigned long long* kb = keys + bk_start[bk];
          // the bucket's keys are read from LDS once, 64 at a time (one per lane), and handed round the wave by v_readlane: the
          // inner loop has no memory access (reading key j from LDS per iteration was bound by the LDS latency: 86 us per
          // 10 000-entry row)
          if (B <= 64u) {                                                 // nearly every bucket: one key per lane
            const unsigned long long mk = (uint32_t)lane < B ? kb[lane] : ~0ull;
            uint32_t below = 0u;
            if (bw < (1u << 18)) {
              // inside a bucket the column is < bucket start + bw: (column - start) << 14 | position is a 32-bit key with the order
              // of the 64-bit one (positions < 16384) — one v_readlane and one 32-bit compare per key instead of two and a 64-bit one
              const uint32_t ck = (uint32_t)lane < B ? ((((uint32_t)(mk >> 16) - bk * bw) << 14) | ((uint32_t)mk & 0x3fffu)) : 0xffffffffu;
              for (uint32_t j = 0; j < B; ++j) below += (uint32_t)__builtin_amdgcn_readlane((int)ck, (int)j) < ck ? 1u : 0u;
            } else {
              const uint32_t lo = (uint32_t)mk, hi = (uint32_t)(mk >> 32);
              for (uint32_t j = 0; j < B; ++j) {
                const unsigned long long kj = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)hi, (int)j) << 32) |
                                              (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)lo, (int)j);
                below += kj < mk ? 1u : 0u;
              }
            }
            __builtin_amdgcn_wave_barrier();                              // (every lane holds its key: nothing reads the bucket any more)
            if ((uint32_t)lane < B) kb[below] = mk;
          } else {                                                        // up to 256 keys: four per lane
            unsigned long long mk[4];
            uint32_t below[4] = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int q = 0; q < 4; ++q) mk[q] = (uint32_t)(lane + 64 * q) < B ? kb[lane + 64 * q] : ~0ull;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              if ((uint32_t)(64 * c) >= B) break;                          // (uniform)
              const uint32_t lo = (uint32_t)mk[c], hi = (uint32_t)(mk[c] >> 32);
              const uint32_t nj = B - 64u * (uint32_t)c < 64u ? B - 64u * (uint32_t)c : 64u;
              for (uint32_t j = 0; j < nj; ++j) {
                const unsigned long long kj = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)hi, (int)j) << 32) |
                                              (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)lo, (int)j);
#pragma unroll
                for (int q = 0; q < 4; ++q) below[q] += kj < mk[q] ? 1u : 0u;
              }
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int q = 0; q < 4; ++q)
              if ((uint32_t)(lane + 64 * q) < B) kb[below[q]] = mk[q];
          }
        }
        __syncthreads();
        sorted_done = true;
      }
    }
  }
  if (!sorted_done) {
    int n2 = 2;
    while (n2 < len) n2 <<= 1;
    // a row whose columns already ascend (canonical CSR) needs no sort: (column << 16 | position) ascends with the columns
    int unsorted = 0;
    for (int i = threadIdx.x; i < n2; i += blockDim.x) {
      keys[i] = i < len ? (((unsigned long long)(uint32_t)indices[b + i] << 16) | (unsigned long long)i) : ~0ull;
      if (i > 0 && i < len && (uint32_t)indices[b + i] < (uint32_t)indices[b + i - 1]) unsorted = 1;
    }
    if (__syncthreads_or(unsorted)) {
      // (a thread owning 16 consecutive keys and running the strides below 16 in registers was tried: its strided LDS reads
      //  conflict 32-way and the build took 1.8 x as long)
      for (int k2 = 2; k2 <= n2; k2 <<= 1)
        for (int j = k2 >> 1; j > 0; j >>= 1) d8_lds_stage(keys, n2, k2, j);
    }
  }
  if (order_out != nullptr)
    for (int i = threadIdx.x; i < len; i += blockDim.x) order_out[b + i] = (uint16_t)(keys[i] & 0xffffull);
  return len;
}

// (slice, local column, first-of-block, escapes before it, delta byte) of sorted position i
struct D8Item { uint32_t s, loc, esc, rem; bool first; };
// H8 = the homogeneous-weight variant of the encoding (see "h8" below): code 255 is reserved for the escape itself,
// so a gap is esc x 255 + rem with rem in [0, 254]; d8 keeps rem in [1, 255] for a non-zero gap.
// col / W without the ~35-instruction integer division (three loops of the fill walk every sorted position through it, twice):
// the quotient by floor((2^32 - 1) / W) is at most one short
__device__ __forceinline__ uint32_t d8_magic(uint32_t W) { return (uint32_t)(0xffffffffull / W); }     // once per kernel
__device__ __forceinline__ uint32_t d8_div(uint32_t col, uint32_t W, uint32_t magic) {
  uint32_t q = __umulhi(col, magic);
  if (col - q * W >= W) ++q;
  return q;
}
template <bool H8 = false>
__device__ __forceinline__ D8Item d8_item(const unsigned long long* keys, int i, uint32_t W, uint32_t magic) {
  D8Item it;
  const uint32_t col = (uint32_t)(keys[i] >> 16);
  it.s = d8_div(col, W, magic);
  it.loc = col - it.s * W;
  uint32_t gap = 0;
  it.first = true;
  if (i > 0) {
    const uint32_t prev = (uint32_t)(keys[i - 1] >> 16);
    if (d8_div(prev, W, magic) == it.s) {
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
                                                        uint32_t slice_width, int n_slices, uint2* __restrict__ seg,
                                                        unsigned long long* __restrict__ too_long,
                                                        uint16_t* __restrict__ order_out) {
  const uint32_t wmagic = d8_magic(slice_width);
  extern __shared__ unsigned long long d8_keys[];
  __shared__ uint32_t tot[kD8MaxSlices], base_s[kD8MaxSlices];
  for (int64_t r = blockIdx.x; r < m; r += gridDim.x) {
    for (int s = threadIdx.x; s < n_slices; s += blockDim.x) { tot[s] = 0; base_s[s] = 0; }
    const int64_t rb = rp.at(r), re = rp.at(r + 1);
    // a row the LDS sort cannot hold: reported to the host, which fails the build (d8_sort_row would cut it)
    if (threadIdx.x == 0 && re - rb > (int64_t)kD8MaxRow) atomicMax(too_long, (unsigned long long)(re - rb));
    const int len = d8_sort_row(d8_keys, indices, rb, re, nullptr, order_out, slice_width * (uint32_t)n_slices);   // ends with a barrier
    for (int i = threadIdx.x; i < len; i += blockDim.x) {
      const D8Item it = d8_item<H8>(d8_keys, i, slice_width, wmagic);
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
                                                       uint32_t* __restrict__ maxabs_bits, const uint16_t* __restrict__ order) {
  const uint32_t wmagic = d8_magic(slice_width);
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
    const int len = d8_sort_row(d8_keys, indices, rb, rp.at(r + 1), order, nullptr, slice_width * (uint32_t)n_slices);
    // inclusive prefix sums of the escape counts over the sorted row: a thread owns `per` consecutive positions
    const int per = (len + 1023) >> 10;
    const int i0 = threadIdx.x * per, i1 = (i0 + per < len) ? i0 + per : len;
    uint32_t mine = 0;
    for (int i = i0; i < i1; ++i) mine += d8_item(d8_keys, i, slice_width, wmagic).esc;
    const uint32_t excl = block_scan_1024(mine, wtot) - mine;
    uint32_t run = excl;
    for (int i = i0; i < i1; ++i) {
      const D8Item it = d8_item(d8_keys, i, slice_width, wmagic);
      run += it.esc;
      if (it.first) { first_idx[it.s] = (uint32_t)i; e_first[it.s] = run; }
    }
    __syncthreads();
    // the thread's weights first, all in flight together (clamped positions instead of a branch per load: a gather under
    // `if` is waited for on the spot, one memory round trip per entry — 16 per row and thread)
    constexpr int kOwn = kD8MaxRow / 1024;
    float wown[kOwn];
    if (len > 0) {
#pragma unroll
      for (int t = 0; t < kOwn; ++t) {
        const int i = i0 + t < i1 ? i0 + t : (i1 > i0 ? i1 - 1 : len - 1);
        wown[t] = (float)WTraits<W>::load(weights, rb + (int64_t)(d8_keys[i] & 0xffffull));
      }
    }
    run = excl;
#pragma unroll
    for (int t = 0; t < kOwn; ++t) {
      const int i = i0 + t;
      if (i >= i1) break;
      const D8Item it = d8_item(d8_keys, i, slice_width, wmagic);
      run += it.esc;
      const uint32_t pos = ((uint32_t)i - first_idx[it.s]) + (run - e_first[it.s]);
      unsigned char* blk = blob + ((int64_t)seg_start[it.s] << 7);
      float* wp = reinterpret_cast<float*>(blk);
      unsigned char* dp = blk + (size_t)seg_ng[it.s] * 16;
      for (uint32_t q = pos - it.esc; q < pos; ++q) { wp[q] = 0.f; dp[q] = 255; }
      const float w = wown[t];
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
#ifdef BE_ABL_NOATOMIC      // ablation builds: everything but the LDS atomics (operands kept alive)
  const unsigned long long v0 = fixed_from_f32(__uint_as_float(wv.x), scale), v1 = fixed_from_f32(__uint_as_float(wv.y), scale),
                           v2 = fixed_from_f32(__uint_as_float(wv.z), scale), v3 = fixed_from_f32(__uint_as_float(wv.w), scale);
  asm volatile("" ::"v"(c0), "v"(c1), "v"(c2), "v"(c3), "v"(v0), "v"(v1), "v"(v2), "v"(v3));
#else
  atomicAdd(&acc[c0], fixed_from_f32(__uint_as_float(wv.x), scale));
  atomicAdd(&acc[c1], fixed_from_f32(__uint_as_float(wv.y), scale));
  atomicAdd(&acc[c2], fixed_from_f32(__uint_as_float(wv.z), scale));
  atomicAdd(&acc[c3], fixed_from_f32(__uint_as_float(wv.w), scale));
#endif
}
__device__ __forceinline__ uint32_t d8_sum4(uint32_t d) { return (d & 0xffu) + ((d >> 8) & 0xffu) + ((d >> 16) & 0xffu) + (d >> 24); }

struct SegGroupD8 {
  uint32_t start[4], ng[4], base[4];
  uint32_t dv[4];
  be_v4u wv[4];
};

template <int AUX>
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
    g.wv[q] = __builtin_amdgcn_raw_buffer_load_b128(rw, lane * 16, 0, AUX);
    auto rd = __builtin_amdgcn_make_buffer_rsrc(blk + (uint64_t)g.ng[q] * 16u, 0, (int)(g.ng[q] * 4u), kBufFlags);
    g.dv[q] = __builtin_amdgcn_raw_buffer_load_b32(rd, lane * 4, 0, AUX);     // lanes past the block read 0
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

// ---- short blocks: a quarter wave per block.  With ~20 entries per block (N = 1M, K = 1000) only 5 of 64 lanes work in
// the one-block-per-wave path and the kernel is bound by its instruction count (~60 wave instructions per block).  Here the
// four 16-lane rows of a wave decode four blocks at once (the DPP row shifts scan within 16 lanes natively), the per-row
// segment entries travel by ds_bpermute, and 16 blocks are in flight per wave (9 registers per group of four).
template <int LPB>
__device__ __forceinline__ uint32_t sub_incl_scan_u32(uint32_t v, uint32_t l /* lane % LPB */) {
  if (LPB == 16) {          // a DPP row: the shifts stop at the row boundary by themselves
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);   // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);   // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);   // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);   // row_shr:8
  } else {                  // half a row: mask what would cross the 8-lane boundary
    uint32_t t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
    v += l >= 1u ? t : 0u;
    t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
    v += l >= 2u ? t : 0u;
    t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);
    v += l >= 4u ? t : 0u;
  }
  return v;
}

struct QGroupD8 {
  uint32_t ng, base, d;
  be_v4u w;
  const unsigned char* blk;
};

template <int LPB>
__device__ __forceinline__ void d8q_issue(QGroupD8& g, int i, int nvalid, uint32_t st_v, uint32_t n4_v, int lane,
                                          const unsigned char* __restrict__ blob) {
  const int row = i + lane / LPB;
  const uint32_t st = (uint32_t)__shfl((int)st_v, row & 63, 64);
  const uint32_t y = (uint32_t)__shfl((int)n4_v, row & 63, 64);
  g.ng = row < nvalid ? (y & 0xffffu) : 0u;
  g.base = y >> 16;
  g.blk = g.ng ? blob + ((uint64_t)st << 7) : blob;          // nothing to do: read the head of the blob (in bounds, cached)
  const uint32_t l = (uint32_t)(lane % LPB);
  const uint32_t o = l < g.ng ? l : 0u;                       // clamped: loads stay unconditional (counted vmcnt waits)
  g.w = *(reinterpret_cast<const be_v4u*>(g.blk) + o);
  g.d = *(reinterpret_cast<const uint32_t*>(g.blk + (uint64_t)g.ng * 16u) + o);
}

template <int LPB>
__device__ __forceinline__ void d8q_consume(QGroupD8& g, unsigned long long* acc, int lane, float scale) {
  // an opaque use of the loaded registers in straight-line code: without it the compiler sinks the weight load into the
  // `l < ng` branch below, where its wait (vmcnt(0)) drains every load the pipeline has in flight
  asm volatile("" : "+v"(g.w.x), "+v"(g.w.y), "+v"(g.w.z), "+v"(g.w.w), "+v"(g.d));
  const uint32_t l = (uint32_t)(lane % LPB);
  const uint32_t d0 = l < g.ng ? g.d : 0u;
  const uint32_t t0 = d8_sum4(d0);
  const uint32_t incl0 = sub_incl_scan_u32<LPB>(t0, l);
  if (l < g.ng) d8_add4(acc, g.base + incl0 - t0, d0, g.w, scale);
}

// the rare blocks longer than one sub-wave pass (ng > LPB lane-groups), after the batch's pipelined loop, a wave per block
// (see sub_tails).  The columns reached by the first LPB lane-groups are re-derived from their deltas.
template <int LPB>
__device__ __forceinline__ void d8q_tails(uint32_t st_v, uint32_t n4_v, int nvalid, unsigned long long* acc, int lane, float scale,
                                          const unsigned char* __restrict__ blob) {
  unsigned long long longm = __ballot(lane < nvalid && (n4_v & 0xffffu) > (uint32_t)LPB);
  while (longm) {
    const int src = __ffsll((long long)longm) - 1;
    longm &= longm - 1;
    const uint32_t st = (uint32_t)__builtin_amdgcn_readlane((int)st_v, src), y = (uint32_t)__builtin_amdgcn_readlane((int)n4_v, src);
    const uint32_t ng = y & 0xffffu;
    const unsigned char* blk = blob + ((uint64_t)st << 7);
    const uint32_t* dp = reinterpret_cast<const uint32_t*>(blk + (uint64_t)ng * 16u);
    const uint32_t d_head = lane < LPB ? dp[lane] : 0u;                 // deltas the pipelined pass has already applied
    uint32_t carry = (y >> 16) + (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan_u32(d8_sum4(d_head)), 63);
    for (uint32_t o0 = LPB; o0 < ng; o0 += 64) {
      const uint32_t o = o0 + lane;
      const bool in = o < ng;
      uint32_t d = 0u;
      be_v4u wv = {0u, 0u, 0u, 0u};
      if (in) {
        d = dp[o];
        const uint4 x = reinterpret_cast<const uint4*>(blk)[o];
        wv = be_v4u{x.x, x.y, x.z, x.w};
      }
      const uint32_t t = d8_sum4(d);
      const uint32_t incl = wave_incl_scan_u32(t);
      if (in) d8_add4(acc, carry + incl - t, d, wv, scale);
      carry += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    }
  }
}

// phase stamps of k_plan_accumulate_d8 (diagnostic builds only: BE_HIPCC_FLAGS=-DBE_PLAN_PROF; tools/plan_phase_prof.py)
#ifdef BE_PLAN_PROF
__device__ unsigned long long g_plan_prof[256 * 8];
#define PLAN_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 256) { const unsigned long long t__ = __builtin_amdgcn_s_memtime(); g_plan_prof[blockIdx.x * 8 + (i)] += t__ - plan_t; plan_t = t__; } } while (0)
#else
#define PLAN_STAMP(i) do { } while (0)
#endif

template <int LPB /* 0: a wave per block; 8 / 16: lanes per block */, int FUSED = 0, int AUX = 0 /* cache policy of the wave-per-block loads */>
__global__ void __launch_bounds__(1024) k_plan_accumulate_d8(const unsigned char* __restrict__ blob, const uint2* __restrict__ seg,
                                                             const uint32_t* active,
                                                             const uint32_t* __restrict__ n_active_p, int n_slices,
                                                             int cap, int parts, float scale,
                                                             unsigned long long* __restrict__ partial, int64_t active_stride,
                                                          const void* __restrict__ fused_spikes, int64_t fused_m, int64_t fused_stride,
                                                          uint32_t list_off, uint32_t lds_cap, uint32_t* __restrict__ glob_lists,
                                                          int64_t glob_region) {
  using acc_t = unsigned long long;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  acc_t* acc = reinterpret_cast<acc_t*>(smem_raw);
  const int S = cap;                     // even; >= slice width (the d8 layout needs no pad slot)
#ifdef BE_PLAN_PROF
  unsigned long long plan_t = __builtin_amdgcn_s_memtime();
#endif
  const int per_xcd = gridDim.x >> 3;
  const int L = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  const int n_tasks = n_slices * parts;
  if (L >= n_tasks) return;
  const int part = L / n_slices;
  const int slice = L - part * n_slices;
  if constexpr (FUSED == 0) active += (int64_t)blockIdx.y * active_stride;
  partial += ((int64_t)blockIdx.y * n_tasks + L) * S;
  {
    uint4* z = reinterpret_cast<uint4*>(smem_raw);
    const int n16 = (int)(((size_t)(S + 1) * sizeof(acc_t) + 15) / 16);
    for (int i = threadIdx.x; i < n16; i += blockDim.x) z[i] = make_uint4(0u, 0u, 0u, 0u);
  }
  __syncthreads();
  PLAN_STAMP(0);     // LDS zeroed
  __shared__ uint32_t fused_wtot[32];
  uint32_t n_active;
  if constexpr (FUSED == 1 || FUSED == 2) {       // list the active rows of this part in LDS behind the accumulators (or in the part's region)
    n_active = build_part_list<FUSED>(static_cast<const unsigned char*>(fused_spikes) + (int64_t)blockIdx.y * fused_stride, fused_m,
                                      part, parts, reinterpret_cast<uint32_t*>(smem_raw + list_off), lds_cap,
                                      glob_lists + ((int64_t)blockIdx.y * parts + part) * glob_region, fused_wtot, &active);
  } else {
    n_active = n_active_p[blockIdx.y];
  }
  // FUSED == 3 (pre-gathered table): part p owns the list positions [p * npp, (p + 1) * npp), a wave 64 consecutive ones
  uint32_t pos_lo = 0;
  if constexpr (FUSED == 3) {
    const uint32_t npp = (n_active + (uint32_t)parts - 1u) / (uint32_t)parts;
    pos_lo = (uint32_t)part * npp;
    const uint32_t hi = pos_lo + npp < n_active ? pos_lo + npp : n_active;
    n_active = hi > pos_lo ? hi : pos_lo;            // from here on: one past this part's last position
  }
  PLAN_STAMP(1);     // spike count / list known
  // segment table: [row][slice] as built, or (FUSED == 3) the step's pre-gathered [slice][list position]
  const uint2* sp = FUSED == 3 ? seg + (int64_t)slice * glob_region : seg + slice;
  const uint64_t seg_rs = FUSED == 3 ? 1u : (uint64_t)n_slices;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
  const uint64_t a0 = FUSED == 3 ? (uint64_t)pos_lo + (uint64_t)wave * 64 + lane
                      : FUSED  ? (uint64_t)wave + (uint64_t)nw * lane
                               : (uint64_t)part + (uint64_t)parts * ((uint64_t)wave + (uint64_t)nw * lane);
  const uint64_t a_step = FUSED ? (uint64_t)nw * 64 : (uint64_t)parts * nw * 64;
  if (n_active > pos_lo) {
    const uint64_t last = n_active - 1;
    uint64_t a = a0;
    bool v_n = a < n_active;
    uint32_t r_n = FUSED == 3 ? (uint32_t)(a < last ? a : last) : active[a < last ? a : last];
    a += a_step;
    uint2 sg = sp[(uint64_t)r_n * seg_rs];
    uint32_t st_v = sg.x, n4_v = v_n ? sg.y : 0u;
    bool v_c = v_n;
    v_n = a < n_active;
    r_n = FUSED == 3 ? (uint32_t)(a < last ? a : last) : active[a < last ? a : last];
    a += a_step;
    PLAN_STAMP(2);   // first row ids and segment entries landed (wave 0)
    while (__ballot(v_c) != 0ull) {
      const int nvalid = __popcll(__ballot(v_c));
      const uint2 sgn = sp[(uint64_t)r_n * seg_rs];
      const bool v_nn = a < n_active;
      const uint32_t r_nn = FUSED == 3 ? (uint32_t)(a < last ? a : last) : active[a < last ? a : last];
      a += a_step;
      if constexpr (LPB > 0) {
        // The 64 rows of a batch are 64 / BPI groups; they are walked in STRAIGHT-LINE code, DEPTH groups ahead: hipcc's
        // wait-count pass is exact without a back edge (the rotating four-group loop this replaces drained the queue to one
        // group at the first consume of every iteration: s_waitcnt vmcnt(2) with eight loads outstanding).
        constexpr int BPI = 64 / LPB;                       // blocks per load instruction
        constexpr int NGRP = 64 / BPI;                      // groups of a 64-row batch
        constexpr int DEPTH = 4;
        QGroupD8 g[NGRP];
#pragma unroll
        for (int q = 0; q < DEPTH; ++q) d8q_issue<LPB>(g[q], q * BPI, nvalid, st_v, n4_v, lane, blob);
#pragma unroll
        for (int q = 0; q < NGRP; ++q) {
          if (q + DEPTH < NGRP) d8q_issue<LPB>(g[q + DEPTH], (q + DEPTH) * BPI, nvalid, st_v, n4_v, lane, blob);
          d8q_consume<LPB>(g[q], acc, lane, scale);
        }
        d8q_tails<LPB>(st_v, n4_v, nvalid, acc, lane, scale, blob);
      } else {
        SegGroupD8 gA, gB;
        d8_issue<AUX>(gA, 0, nvalid, st_v, n4_v, lane, blob);
        for (int i = 0; i < nvalid; i += 8) {
          d8_issue<AUX>(gB, i + 4, nvalid, st_v, n4_v, lane, blob);
          d8_consume(gA, acc, lane, scale, blob);
          d8_issue<AUX>(gA, i + 8, nvalid, st_v, n4_v, lane, blob);
          d8_consume(gB, acc, lane, scale, blob);
        }
      }
      st_v = sgn.x;
      n4_v = v_n ? sgn.y : 0u;
      v_c = v_n;
      v_n = v_nn;
      r_n = r_nn;
    }
  }
  PLAN_STAMP(3);     // wave 0 through its blocks
  __syncthreads();
  PLAN_STAMP(4);     // every wave through
  {
    store_partials<1>(smem_raw, partial, (int)((size_t)S * sizeof(acc_t) / 16));
  }
#ifdef BE_PLAN_PROF
  __builtin_amdgcn_s_waitcnt(0);
  PLAN_STAMP(5);     // partial sums stored
#endif
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
                                                       unsigned char* __restrict__ blob, const uint16_t* __restrict__ order) {
  const uint32_t wmagic = d8_magic(slice_width);
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
    const int len = d8_sort_row(d8_keys, indices, rp.at(r), rp.at(r + 1), order, nullptr, slice_width * (uint32_t)n_slices);
    const int per = (len + 1023) >> 10;
    const int i0 = threadIdx.x * per, i1 = (i0 + per < len) ? i0 + per : len;
    uint32_t mine = 0;
    for (int i = i0; i < i1; ++i) mine += d8_item<true>(d8_keys, i, slice_width, wmagic).esc;
    const uint32_t excl = block_scan_1024(mine, wtot) - mine;
    uint32_t run = excl;
    for (int i = i0; i < i1; ++i) {
      const D8Item it = d8_item<true>(d8_keys, i, slice_width, wmagic);
      run += it.esc;
      if (it.first) { first_idx[it.s] = (uint32_t)i; e_first[it.s] = run; }
    }
    __syncthreads();
    run = excl;
    for (int i = i0; i < i1; ++i) {
      const D8Item it = d8_item<true>(d8_keys, i, slice_width, wmagic);
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

template <int FUSED = 0>
__global__ void __launch_bounds__(1024) k_plan_accumulate_h8(const unsigned char* __restrict__ blob, const uint2* __restrict__ seg,
                                                             const uint32_t* active,
                                                             const uint32_t* __restrict__ n_active_p, int n_slices,
                                                             int cap, int parts, uint32_t* __restrict__ partial,
                                                             int64_t active_stride,
                                                          const void* __restrict__ fused_spikes, int64_t fused_m, int64_t fused_stride,
                                                          uint32_t list_off, uint32_t lds_cap, uint32_t* __restrict__ glob_lists,
                                                          int64_t glob_region) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  uint32_t* acc = reinterpret_cast<uint32_t*>(smem_raw);
  const int S = cap;                     // multiple of 4; >= slice width
  const int per_xcd = gridDim.x >> 3;
  const int L = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  const int n_tasks = n_slices * parts;
  if (L >= n_tasks) return;
  const int part = L / n_slices;
  const int slice = L - part * n_slices;
  if constexpr (FUSED == 0) active += (int64_t)blockIdx.y * active_stride;
  partial += ((int64_t)blockIdx.y * n_tasks + L) * S;
  {
    uint4* z = reinterpret_cast<uint4*>(smem_raw);
    const int n16 = (int)((size_t)S * 4 / 16);
    for (int i = threadIdx.x; i < n16; i += blockDim.x) z[i] = make_uint4(0u, 0u, 0u, 0u);
  }
  __syncthreads();
  __shared__ uint32_t fused_wtot[32];
  uint32_t n_active;
  if constexpr (FUSED == 1 || FUSED == 2) {       // list the active rows of this part in LDS behind the accumulators (or in the part's region)
    n_active = build_part_list<FUSED>(static_cast<const unsigned char*>(fused_spikes) + (int64_t)blockIdx.y * fused_stride, fused_m,
                                      part, parts, reinterpret_cast<uint32_t*>(smem_raw + list_off), lds_cap,
                                      glob_lists + ((int64_t)blockIdx.y * parts + part) * glob_region, fused_wtot, &active);
  } else {
    n_active = n_active_p[blockIdx.y];
  }
  // FUSED == 3 (pre-gathered table): part p owns the list positions [p * npp, (p + 1) * npp), a wave 64 consecutive ones
  uint32_t pos_lo = 0;
  if constexpr (FUSED == 3) {
    const uint32_t npp = (n_active + (uint32_t)parts - 1u) / (uint32_t)parts;
    pos_lo = (uint32_t)part * npp;
    const uint32_t hi = pos_lo + npp < n_active ? pos_lo + npp : n_active;
    n_active = hi > pos_lo ? hi : pos_lo;            // from here on: one past this part's last position
  }
  // segment table: [row][slice] as built, or (FUSED == 3) the step's pre-gathered [slice][list position]
  const uint2* sp = FUSED == 3 ? seg + (int64_t)slice * glob_region : seg + slice;
  const uint64_t seg_rs = FUSED == 3 ? 1u : (uint64_t)n_slices;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
  const uint64_t a0 = FUSED == 3 ? (uint64_t)pos_lo + (uint64_t)wave * 64 + lane
                      : FUSED  ? (uint64_t)wave + (uint64_t)nw * lane
                               : (uint64_t)part + (uint64_t)parts * ((uint64_t)wave + (uint64_t)nw * lane);
  const uint64_t a_step = FUSED ? (uint64_t)nw * 64 : (uint64_t)parts * nw * 64;
  if (n_active > pos_lo) {
    const uint64_t last = n_active - 1;
    uint64_t a = a0;
    bool v_n = a < n_active;
    uint32_t r_n = FUSED == 3 ? (uint32_t)(a < last ? a : last) : active[a < last ? a : last];
    a += a_step;
    uint2 sg = sp[(uint64_t)r_n * seg_rs];
    uint32_t st_v = sg.x, n4_v = v_n ? sg.y : 0u;
    bool v_c = v_n;
    v_n = a < n_active;
    r_n = FUSED == 3 ? (uint32_t)(a < last ? a : last) : active[a < last ? a : last];
    a += a_step;
    while (__ballot(v_c) != 0ull) {
      const int nvalid = __popcll(__ballot(v_c));
      const uint2 sgn = sp[(uint64_t)r_n * seg_rs];
      const bool v_nn = a < n_active;
      const uint32_t r_nn = FUSED == 3 ? (uint32_t)(a < last ? a : last) : active[a < last ? a : last];
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
    store_partials<4>(smem_raw, partial, (int)((size_t)S * 4 / 16));
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

// =================================================================================================
// Column statistics of a weighted plan, read from its own blocks: what the fixed-point exponent needs — the largest column sum
// of |w| (overflow bound) and the smallest of the columns' largest |w| (accuracy gate) — without the two passes of global
// float atomics over the raw entries that be_fixed_point_exponent makes (0.47 s at 1e10 entries: the chip's 21 G/s).  One
// workgroup per slice owns the slice's accumulators in LDS and walks EVERY row's block of that slice: a planned step with all
// rows active, over |w|.  MODE 0: 64-bit fixed-point sums at a safe exponent, every addend rounded UP, so the bound never
// falls short of the true sum; out[0] = the largest.  MODE 1: ds_max of the |w| bit patterns; out[1] = the smallest non-zero one.
// Build-time code: a wave per block, four blocks in flight per wave, nothing tuned beyond that.
// =================================================================================================
template <int MODE> struct StatAcc { using type = unsigned long long; };
template <> struct StatAcc<1> { using type = uint32_t; };
template <int MODE>
__device__ __forceinline__ void stat_add(typename StatAcc<MODE>::type* acc, uint32_t col, uint32_t wbits, float scale) {
  const uint32_t ab = wbits & 0x7fffffffu;
  if (ab == 0u) return;
  if constexpr (MODE == 0) atomicAdd(&acc[col], fixed_from_f32(__uint_as_float(ab), scale) + 1ull);
  else atomicMax(&acc[col], ab);
}
template <int LAYOUT /* BE_PLAN_U16 (weighted) or BE_PLAN_D8 */, int MODE>
__global__ void __launch_bounds__(1024) k_plan_colstats(const unsigned char* __restrict__ blob, const uint2* __restrict__ seg,
                                                        int64_t m, int n_slices, int n_slots, float scale,
                                                        unsigned long long* __restrict__ out) {
  using acc_t = typename StatAcc<MODE>::type;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  acc_t* acc = reinterpret_cast<acc_t*>(smem_raw);
  for (int i = threadIdx.x; i < n_slots; i += blockDim.x) acc[i] = 0;
  __syncthreads();
  const int slice = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
  constexpr int U = 4;
  // a wave takes 64 consecutive rows at a time: lane l holds the table entry of row r0 + l
  for (int64_t r0 = (int64_t)wave * 64; r0 < m; r0 += (int64_t)nw * 64) {
    uint2 sg = make_uint2(0u, 0u);
    if (r0 + lane < m) sg = seg[(r0 + lane) * n_slices + slice];
    const int n_rows = (int)(m - r0 < 64 ? m - r0 : 64);
    for (int i0 = 0; i0 < n_rows; i0 += U) {
      be_v4u wv[U];
      uint32_t cv0[U], cv1[U], ngs[U], bases[U];
      const unsigned char* blks[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {              // the first 64 lane-groups of U blocks: every load issued before the first add
        const int i = i0 + u < n_rows ? i0 + u : n_rows - 1;
        const uint32_t st = (uint32_t)__builtin_amdgcn_readlane((int)sg.x, i), y = (uint32_t)__builtin_amdgcn_readlane((int)sg.y, i);
        ngs[u] = i0 + u < n_rows ? (LAYOUT == BE_PLAN_D8 ? y & 0xffffu : y) : 0u;
        bases[u] = y >> 16;
        blks[u] = blob + ((int64_t)st << 7);
        wv[u] = be_v4u{0u, 0u, 0u, 0u};
        cv0[u] = cv1[u] = 0u;
        if ((uint32_t)lane < ngs[u]) {
          wv[u] = reinterpret_cast<const be_v4u*>(blks[u])[lane];
          if (LAYOUT == BE_PLAN_D8) {
            cv0[u] = reinterpret_cast<const uint32_t*>(blks[u] + (size_t)ngs[u] * 16)[lane];
          } else {
            const uint2 c = reinterpret_cast<const uint2*>(blks[u] + (size_t)ngs[u] * 16)[lane];
            cv0[u] = c.x;
            cv1[u] = c.y;
          }
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        uint32_t before = bases[u];                 // d8: the column in front of this pass's first entry
        for (uint32_t g0 = 0; g0 < ngs[u]; g0 += 64) {
          be_v4u w = wv[u];
          uint32_t c0 = cv0[u], c1 = cv1[u];
          const bool in = g0 + (uint32_t)lane < ngs[u];
          if (g0 > 0) {                             // (rare) a block of more than 256 items: the later passes load here
            w = be_v4u{0u, 0u, 0u, 0u};
            c0 = c1 = 0u;
            if (in) {
              w = reinterpret_cast<const be_v4u*>(blks[u])[g0 + lane];
              if (LAYOUT == BE_PLAN_D8) {
                c0 = reinterpret_cast<const uint32_t*>(blks[u] + (size_t)ngs[u] * 16)[g0 + lane];
              } else {
                const uint2 c = reinterpret_cast<const uint2*>(blks[u] + (size_t)ngs[u] * 16)[g0 + lane];
                c0 = c.x;
                c1 = c.y;
              }
            }
          }
          if (LAYOUT == BE_PLAN_D8) {
            const uint32_t d = in ? c0 : 0u, mine = d8_sum4(d);
            const uint32_t incl = wave_incl_scan_u32(mine);
            const uint32_t b = before + incl - mine;
            const uint32_t k0 = b + (d & 0xffu), k1 = k0 + ((d >> 8) & 0xffu), k2 = k1 + ((d >> 16) & 0xffu), k3 = k2 + (d >> 24);
            if (in) {
              stat_add<MODE>(acc, k0, w.x, scale); stat_add<MODE>(acc, k1, w.y, scale);
              stat_add<MODE>(acc, k2, w.z, scale); stat_add<MODE>(acc, k3, w.w, scale);
            }
            before += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
          } else if (in) {
            stat_add<MODE>(acc, c0 & 0xffffu, w.x, scale); stat_add<MODE>(acc, c0 >> 16, w.y, scale);
            stat_add<MODE>(acc, c1 & 0xffffu, w.z, scale); stat_add<MODE>(acc, c1 >> 16, w.w, scale);
          }
        }
      }
    }
  }
  __syncthreads();
  unsigned long long best = MODE == 0 ? 0ull : ~0ull;
  for (int i = threadIdx.x; i < n_slots; i += blockDim.x) {
    const unsigned long long v = (unsigned long long)acc[i];
    if (MODE == 0) best = v > best ? v : best;
    else if (v != 0ull) best = v < best ? v : best;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const unsigned long long o = ((unsigned long long)(uint32_t)__shfl_down((int)(best >> 32), off, 64) << 32) |
                                 (uint32_t)__shfl_down((int)(uint32_t)best, off, 64);
    best = MODE == 0 ? (o > best ? o : best) : (o < best ? o : best);
  }
  if (lane == 0) {
    if (MODE == 0) { if (best) atomicMax(&out[0], best); }
    else if (best != ~0ull) atomicMin(&out[1], best);
  }
}

}  // namespace

// =================================================================================================
// C ABI
// =================================================================================================
#ifdef BE_PLAN_PROF
extern "C" int be_debug_plan_prof(unsigned long long* host, int reset) {
  if (host) (void)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_plan_prof), sizeof(unsigned long long) * 256 * 8);
  if (reset) { static unsigned long long z[256 * 8]; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_plan_prof), z, sizeof(z)); }
  return 0;
}
#endif

extern "C" {

// ---------------------------------------------------------------- scatter plan
// slices are `slice_width` output neurons wide (0 = the LDS capacity 2^slice_shift); a width below the capacity lets
// the caller balance the slices (k = 1M: 64 slices of 15625 instead of 61 full ones and a sliver)
// Lanes per block by the average block length: a sub-wave pass covers LPB lane-groups (4 entries each for d8, 8 for counted
// u16); longer blocks wait for the serial tails pass, so a variant is used while mean + 3 sigma (Poisson) fits one pass.
constexpr int kD8QuarterMaxBlock = 43;  // average entries per block up to which 16 lanes per block decode the d8 layout (64 per pass)
constexpr int kD8EighthMaxBlock = 18;   // ... and 8 lanes per block (32 per pass)
constexpr int kSub4MaxBlock = 18;       // counted u16: 4 lanes per block (32 per pass)
constexpr int kSub8MaxBlock = 43;       // 8 lanes (64 per pass)
constexpr int kSub16MaxBlock = 96;      // 16 lanes (128 per pass)
constexpr int kSubW4MaxBlock = 7;       // weighted u16: 4 lanes per block (16 per pass)
constexpr int kSubW8MaxBlock = 18;      // weighted u16: 8 lanes per block (32 per pass)
constexpr int kSubW16MaxBlock = 43;     // 16 lanes (64 per pass)
constexpr int kSubW32MaxBlock = 96;     // 32 lanes (128 per pass)
constexpr int kSub32MaxBlock = 210;     // counted u16: 32 lanes (256 per pass)
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

#define COMMA ,
// The segment table of a step is pre-gathered (k_gather_seg) for short blocks in many slices: the two extra small launches
// cost ~5 us, the gather they take out of the accumulate kernel grows with the slice count (FixedNumPerPre K = 1000, 1 %
// firing, weighted / counted, us per step: 25 slices 26 -> 28 / 21 -> 25; 32 slices 30 -> 33 / 37 -> 35; 51 slices
// 43 -> 38 / 60 -> 48; 64 slices 55 -> 46; 85 slices counted 172 -> 89; 128 slices weighted 125 -> 84).
constexpr int kPreGatherMaxBlock = 64;   // average entries per block up to which ...
constexpr int kPreGatherMinSlices = 40;  // ... and slices from which the segment table is pre-gathered
constexpr size_t kFusedMinList = 2048;   // LDS room (row ids) from which the step lists its active rows in the kernel
// rows a part can own: its stripes of 1024 rows (the overflow destination of a part's list in the workspace)
static inline int64_t fused_region_of(int64_t m, int parts) {
  const int64_t n_stripes = (((m + 31) >> 5) + 31) >> 5;
  return ((n_stripes + parts - 1) / parts) * 1024;
}

int64_t be_scatter_plan_scratch_bytes(int64_t m, int64_t k, int slice_shift, int slice_width) {
  const int64_t n = (int64_t)n_slices_of(k, slice_shift, slice_width) * m;
  const int64_t n_blocks = (n + kScanChunk - 1) / kScanChunk;
  return be_align_up((n_blocks + 2) * 8, 256);
}

// ---- the count pass in three pieces, so that a matrix can be planned from BLOCKS OF ROWS that are resident one at a time
//      (be_scatter_plan_begin, then be_scatter_plan_count_rows per block, then be_scatter_plan_scan; the fill runs per block too:
//      everything the count and fill kernels touch is local to a row except the block starts, which the scan provides)
static inline unsigned long long* plan_too_long(void* scratch, int64_t m, int n_slices) {
  const int64_t n_blocks = ((int64_t)n_slices * m + kScanChunk - 1) / kScanChunk;
  return reinterpret_cast<unsigned long long*>(static_cast<uint64_t*>(scratch) + n_blocks + 1);   // longest over-long row
}

int be_scatter_plan_begin(int64_t m, int64_t k, int slice_shift, int slice_width, void* scratch, int64_t scratch_bytes,
                          be_stream_t stream) {
  BE_REQUIRE(m > 0 && k > 0, BE_ERR_INVALID, "empty matrix has no plan");
  BE_REQUIRE(slice_shift >= 4 && slice_shift <= 15, BE_ERR_INVALID, "slice_shift must be in [4, 15]");
  BE_REQUIRE(scratch && scratch_bytes >= be_scatter_plan_scratch_bytes(m, k, slice_shift, slice_width), BE_ERR_WORKSPACE, "scratch too small");
  const int n_slices = n_slices_of(k, slice_shift, slice_width);
  BE_HIP(be_fill_async(plan_too_long(scratch, m, n_slices), 0, 8, static_cast<hipStream_t>(stream)));
  return BE_OK;
}

int be_scatter_plan_count_rows(const int32_t* indices, const void* indptr, int indptr_is_i64, int64_t row_len, int64_t m_rows,
                               int64_t m_total, int64_t k, int slice_shift, int slice_width, int homo, int layout, void* seg_rows,
                               void* scratch, int64_t scratch_bytes, uint16_t* order_out, be_stream_t stream) {
  BE_REQUIRE(m_rows > 0 && m_rows <= m_total && k > 0, BE_ERR_INVALID, "bad row block");
  BE_REQUIRE(slice_shift >= 4 && slice_shift <= 15, BE_ERR_INVALID, "slice_shift must be in [4, 15]");
  BE_REQUIRE(width_ok(slice_shift, slice_width, layout), BE_ERR_INVALID, "slice_width out of range for this layout");
  BE_REQUIRE(check_rows(indptr, row_len), BE_ERR_INVALID, "indptr is NULL and row_len < 0");
  const int n_slices = n_slices_of(k, slice_shift, slice_width);
  BE_REQUIRE(n_slices <= kMaxSlices, BE_ERR_RANGE, "too many slices for the plan kernels");
  BE_REQUIRE(seg_rows && scratch, BE_ERR_INVALID, "null pointer");
  BE_REQUIRE(scratch_bytes >= be_scatter_plan_scratch_bytes(m_total, k, slice_shift, slice_width), BE_ERR_WORKSPACE,
             "scratch too small");
  hipStream_t st = static_cast<hipStream_t>(stream);
  RowPtr rp{indptr, indptr_is_i64, row_len};
  uint2* sg = static_cast<uint2*>(seg_rows);
  unsigned long long* too_long = plan_too_long(scratch, m_total, n_slices);
  if (layout == BE_PLAN_D8 || layout == BE_PLAN_H8) {
    BE_REQUIRE((layout == BE_PLAN_H8) == (homo != 0), BE_ERR_INVALID, "d8 is the heterogeneous layout, h8 the homogeneous one");
    BE_REQUIRE(indptr != nullptr || row_len <= kD8MaxRow, BE_ERR_RANGE, "d8 / h8 layout: rows of at most 16384 entries");
    BE_REQUIRE(n_slices <= kD8MaxSlices, BE_ERR_RANGE, "too many slices for the d8 / h8 layout");
    auto kern = layout == BE_PLAN_H8 ? k_plan_d8_count<true> : k_plan_d8_count<false>;
    BE_HIP(be_allow_lds(reinterpret_cast<const void*>(kern), kD8MaxRow * 8));
    hipLaunchKernelGGL(kern, dim3(grid_for(m_rows, 1, 256 * 8)), dim3(1024), kD8MaxRow * 8, st, indices, rp, m_rows,
                       (uint32_t)width_of(slice_shift, slice_width), n_slices, sg, too_long, order_out);
  } else {
    BE_REQUIRE(layout == BE_PLAN_U16, BE_ERR_INVALID, "unknown plan layout");
    hipLaunchKernelGGL(k_plan_count, dim3(grid_for(m_rows, 1, 256 * 16)), dim3(256), 0, st, indices, rp, m_rows,
                       (uint32_t)width_of(slice_shift, slice_width), n_slices, homo, sg);
  }
  BE_LAUNCH_CHECK();
  return BE_OK;
}

int be_scatter_plan_scan(int64_t m, int64_t k, int slice_shift, int slice_width, void* seg, void* scratch, int64_t scratch_bytes,
                         int64_t* blob_bytes_host, be_stream_t stream) {
  BE_REQUIRE(m > 0 && k > 0, BE_ERR_INVALID, "empty matrix has no plan");
  BE_REQUIRE(slice_shift >= 4 && slice_shift <= 15, BE_ERR_INVALID, "slice_shift must be in [4, 15]");
  BE_REQUIRE(seg && scratch && blob_bytes_host, BE_ERR_INVALID, "null pointer");
  BE_REQUIRE(scratch_bytes >= be_scatter_plan_scratch_bytes(m, k, slice_shift, slice_width), BE_ERR_WORKSPACE, "scratch too small");
  const int n_slices = n_slices_of(k, slice_shift, slice_width);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int64_t n = (int64_t)n_slices * m;
  uint2* sg = static_cast<uint2*>(seg);
  uint64_t* sums = static_cast<uint64_t*>(scratch);
  const int64_t n_blocks = (n + kScanChunk - 1) / kScanChunk;
  BE_REQUIRE(n_blocks < (1ll << 31), BE_ERR_RANGE, "plan index too large");
  hipLaunchKernelGGL(k_scan_block_sums, dim3((unsigned)n_blocks), dim3(256), 0, st, sg, n, sums);
  BE_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(1024), 0, st, sums, n_blocks);
  BE_LAUNCH_CHECK();
  uint64_t back[2] = {0, 0};        // { total block units, longest row above the d8 / h8 limit }
  BE_HIP(hipMemcpyAsync(back, sums + n_blocks, 16, hipMemcpyDeviceToHost, st));
  BE_HIP(hipStreamSynchronize(st));
  BE_REQUIRE(back[1] == 0, BE_ERR_RANGE,
             "d8 / h8 layout: a row has " + std::to_string(back[1]) + " entries (at most 16384); use BE_PLAN_U16");
  const uint64_t total_units = back[0];
  BE_REQUIRE(total_units < (1ull << 32), BE_ERR_RANGE, "matrix too large for a 32-bit block index (512 GiB)");
  hipLaunchKernelGGL(k_scan_apply, dim3((unsigned)n_blocks), dim3(256), 0, st, sg, n, sums);
  BE_LAUNCH_CHECK();
  *blob_bytes_host = (int64_t)(total_units << 7);
  return BE_OK;
}

int be_scatter_plan_count_ordered(const int32_t* indices, const void* indptr, int indptr_is_i64, int64_t row_len, int64_t m,
                                  int64_t k, int slice_shift, int slice_width, int homo, int layout, void* seg, void* scratch,
                                  int64_t scratch_bytes, int64_t* blob_bytes_host, uint16_t* order_out, be_stream_t stream) {
  BE_REQUIRE(blob_bytes_host, BE_ERR_INVALID, "null pointer");
  int rc = be_scatter_plan_begin(m, k, slice_shift, slice_width, scratch, scratch_bytes, stream);
  if (rc != BE_OK) return rc;
  rc = be_scatter_plan_count_rows(indices, indptr, indptr_is_i64, row_len, m, m, k, slice_shift, slice_width, homo, layout, seg, scratch,
                                  scratch_bytes, order_out, stream);
  if (rc != BE_OK) return rc;
  return be_scatter_plan_scan(m, k, slice_shift, slice_width, seg, scratch, scratch_bytes, blob_bytes_host, stream);
}

int be_scatter_plan_count(const int32_t* indices, const void* indptr, int indptr_is_i64, int64_t row_len, int64_t m,
                          int64_t k, int slice_shift, int slice_width, int homo, int layout, void* seg, void* scratch,
                          int64_t scratch_bytes, int64_t* blob_bytes_host, be_stream_t stream) {
  return be_scatter_plan_count_ordered(indices, indptr, indptr_is_i64, row_len, m, k, slice_shift, slice_width, homo, layout, seg,
                                       scratch, scratch_bytes, blob_bytes_host, nullptr, stream);
}

int be_scatter_plan_fill_ordered(const void* weights, int homo, int wdtype, const int32_t* indices, const void* indptr,
                                 int indptr_is_i64, int64_t row_len, int64_t m, int64_t k, int slice_shift, int slice_width,
                                 int layout, const void* seg, void* blob, uint32_t* maxabs_bits, const uint16_t* order,
                                 be_stream_t stream) {
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
                       static_cast<unsigned char*>(blob), order);
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
                         wdt, n_slices, static_cast<const uint2*>(seg), static_cast<unsigned char*>(blob), maxabs_bits,  \
                         order);                                                                                         \
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

int be_scatter_plan_fill(const void* weights, int homo, int wdtype, const int32_t* indices, const void* indptr,
                         int indptr_is_i64, int64_t row_len, int64_t m, int64_t k, int slice_shift, int slice_width,
                         int layout, const void* seg, void* blob, uint32_t* maxabs_bits, be_stream_t stream) {
  return be_scatter_plan_fill_ordered(weights, homo, wdtype, indices, indptr, indptr_is_i64, row_len, m, k, slice_shift, slice_width,
                                      layout, seg, blob, maxabs_bits, nullptr, stream);
}

int be_scatter_plan_refresh_weights_ordered(const void* weights, int homo, int wdtype, const int32_t* indices, const void* indptr,
                                            int indptr_is_i64, int64_t row_len, int64_t m, int64_t k, int slice_shift,
                                            int slice_width, int layout, const void* seg, void* blob, uint32_t* maxabs_bits,
                                            const uint16_t* order, be_stream_t stream) {
  // block starts and lengths (seg) depend on the structure only: re-running the fill over the same seg / blob rewrites
  // every block with the new weights.  The d8 fill needs every row in column order: with the order the count pass stored it
  // is a gather-copy, without it the rows are sorted again (positions inside a u16 block may differ from the first fill,
  // which integer accumulation does not see)
  return be_scatter_plan_fill_ordered(weights, homo, wdtype, indices, indptr, indptr_is_i64, row_len, m, k, slice_shift,
                                      slice_width, layout, seg, blob, maxabs_bits, order, stream);
}
int be_scatter_plan_refresh_weights(const void* weights, int homo, int wdtype, const int32_t* indices, const void* indptr,
                                    int indptr_is_i64, int64_t row_len, int64_t m, int64_t k, int slice_shift,
                                    int slice_width, int layout, const void* seg, void* blob, uint32_t* maxabs_bits,
                                    be_stream_t stream) {
  return be_scatter_plan_refresh_weights_ordered(weights, homo, wdtype, indices, indptr, indptr_is_i64, row_len, m, k, slice_shift,
                                                 slice_width, layout, seg, blob, maxabs_bits, nullptr, stream);
}

// The fixed-point exponent of a weighted plan from the plan itself (k_plan_colstats): same bound, same gate and same keep_exp rule
// as be_fixed_point_exponent, whose global-atomic passes over the raw entries it replaces for matrices that have a plan.
// maxabs_bits: what the fill left (max |w| bits, smallest non-zero |w| bits).  scratch >= 256 bytes.  SYNCHRONOUS.
int be_scatter_plan_exponent(const void* blob, const void* seg, int64_t m, int64_t k, int slice_shift, int slice_width, int layout,
                             int64_t nnz, const uint32_t* maxabs_bits, int min_weight_bits, int keep_exp, void* scratch,
                             int64_t scratch_bytes, int* scale_exp_host, be_stream_t stream) {
  BE_REQUIRE(blob && seg && maxabs_bits && scratch && scale_exp_host, BE_ERR_INVALID, "null pointer");
  BE_REQUIRE(m > 0 && k > 0 && nnz >= 0, BE_ERR_INVALID, "bad shape");
  BE_REQUIRE(layout == BE_PLAN_U16 || layout == BE_PLAN_D8, BE_ERR_INVALID, "a weighted plan has the u16 or the d8 layout");
  BE_REQUIRE(slice_shift >= 4 && slice_shift <= 15 && width_ok(slice_shift, slice_width, layout), BE_ERR_INVALID, "bad slice geometry");
  BE_REQUIRE(min_weight_bits >= 0 && min_weight_bits <= 40, BE_ERR_INVALID, "min_weight_bits out of range");
  BE_REQUIRE(scratch_bytes >= 256, BE_ERR_WORKSPACE, "scratch too small");
  const int n_slices = n_slices_of(k, slice_shift, slice_width);
  hipStream_t st = static_cast<hipStream_t>(stream);
  unsigned long long* out = static_cast<unsigned long long*>(scratch);           // [0] largest column sum, [1] smallest column maximum
  uint32_t mb[2];
  BE_HIP(hipMemcpyAsync(mb, maxabs_bits, 8, hipMemcpyDeviceToHost, st));
  BE_HIP(hipStreamSynchronize(st));
  BE_REQUIRE(mb[0] < 0x7f800000u, BE_ERR_RANGE, "weights contain inf / nan: the fixed-point routes do not apply");
  float wmax, wmin = 0.f;
  memcpy(&wmax, &mb[0], 4);
  const bool has_min = mb[1] != 0xffffffffu;
  if (has_min) memcpy(&wmin, &mb[1], 4);
  // LDS slots: the u16 layout addresses 2^shift columns + the pad slot, the d8 layout the slice's own columns
  const int n_slots = layout == BE_PLAN_D8 ? (int)((width_of(slice_shift, slice_width) + 3) & ~3ll) : (1 << slice_shift) + 1;
  auto launch = [&](int mode, float scale) -> int {
    const size_t lds = (size_t)n_slots * (mode == 0 ? 8 : 4);
#define BE_STAT(L_, M_)                                                                                          \
    {                                                                                                            \
      auto kern = k_plan_colstats<L_, M_>;                                                                       \
      BE_HIP(be_allow_lds(reinterpret_cast<const void*>(kern), (int)lds));                                       \
      hipLaunchKernelGGL(kern, dim3(n_slices), dim3(1024), lds, st, static_cast<const unsigned char*>(blob),     \
                         static_cast<const uint2*>(seg), m, n_slices, n_slots, scale, out);                      \
    }
    if (layout == BE_PLAN_D8) { if (mode == 0) BE_STAT(BE_PLAN_D8, 0) else BE_STAT(BE_PLAN_D8, 1) }
    else { if (mode == 0) BE_STAT(BE_PLAN_U16, 0) else BE_STAT(BE_PLAN_U16, 1) }
#undef BE_STAT
    BE_LAUNCH_CHECK();
    return BE_OK;
  };
  double bound = 0.0;
  if (wmax > 0.f && nnz > 0) {
    // a safe exponent for the sums of |w| rounded up: nnz * (wmax * 2^e0 + 1) < 2^62
    int ew = 0, en = 0;
    (void)frexp((double)wmax, &ew);                                          // wmax < 2^ew
    (void)frexp((double)nnz, &en);                                           // nnz  < 2^en
    int e0 = 61 - ew - en;
    e0 = e0 < -90 ? -90 : (e0 > 150 ? 150 : e0);
    BE_HIP(be_fill_async(out, 0, 8, st));
    int rc = launch(0, ldexpf(1.0f, e0 - 32));
    if (rc != BE_OK) return rc;
    unsigned long long smax = 0;
    BE_HIP(hipMemcpyAsync(&smax, out, 8, hipMemcpyDeviceToHost, st));
    BE_HIP(hipStreamSynchronize(st));
    bound = ldexp((double)smax + 1.0, -e0);                                 // >= the true largest column sum (addends rounded up)
  }
  int eb = 0;
  if (bound > 0) (void)frexp(bound, &eb);                                    // bound < 2^eb
  int need = 62 - eb;
  need = need < -90 ? -90 : (need > 150 ? 150 : need);                       // 2^(e - 32) must be a normal f32
  bool have_colmax = false;
  float colmax_min = 0.f;
  const int cand[2] = {keep_exp, need};
  for (int c = (keep_exp != INT_MIN && keep_exp <= need) ? 0 : 1; c < 2; ++c) {
    const int e = cand[c];
    const double thr = ldexp(1.0, min_weight_bits - e);
    bool ok = !has_min || (double)wmin >= thr;
    if (!ok) {
      if (!have_colmax) {
        BE_HIP(be_fill_async(out + 1, 0xff, 8, st));
        int rc = launch(1, 0.f);
        if (rc != BE_OK) return rc;
        unsigned long long cm = ~0ull;
        BE_HIP(hipMemcpyAsync(&cm, out + 1, 8, hipMemcpyDeviceToHost, st));
        BE_HIP(hipStreamSynchronize(st));
        have_colmax = true;
        if (cm == ~0ull) colmax_min = INFINITY;
        else { const uint32_t b = (uint32_t)cm; memcpy(&colmax_min, &b, 4); }
      }
      ok = (double)colmax_min >= thr;
    }
    if (ok) {
      *scale_exp_host = e;
      return BE_OK;
    }
  }
  be_set_error("be_scatter_plan_exponent: the dynamic range of the weights (" + std::to_string(wmin) + " .. " + std::to_string(wmax) +
               ") exceeds what 64-bit fixed-point sums resolve; use the direct route");
  return BE_ERR_RANGE;
}

// active-list area per batch row: the compacted list (m ids) or, in the fused step, one region per part (its stripes of
// 1024 rows: at most 64 x 1024 ids more than m in all)
static inline int64_t plan_active_stride(int64_t m) { return active_stride_of(m + 64 * 1024 + 1024); }

// room for the pre-gathered segment table (k_gather_seg: n_slices x positions x 8 B, sized for every row active: 8 B per
// block of the plan, which itself holds >= 128 B per block) of a single-vector step, as long as it stays below 8 GiB;
// larger tables (and batches) gather from the plan's own table
static inline int pregather_min_slices() {
  static const int v = [] { const char* e = getenv("BE_PLAN_PREGATHER_SLICES"); return e ? atoi(e) : kPreGatherMinSlices; }();
  return v > 2 ? v : 2;
}
static inline int64_t plan_dense_seg_bytes(int64_t m, int64_t n_slices, int64_t n_batch) {
  const int64_t bytes = n_slices * plan_active_stride(m) * 8;
  return (n_batch == 1 && n_slices >= pregather_min_slices() && bytes <= (8ll << 30)) ? be_align_up(bytes, 256) : 0;
}

static inline int pregather_max_block() {
  static const int v = [] { const char* e = getenv("BE_PLAN_PREGATHER_MAX"); return e ? atoi(e) : kPreGatherMaxBlock; }();
  return v;
}

// the workspace every form of the planned step needs: counters, active lists, partial sums
int64_t be_binary_csrmm_t_plan_workspace_bytes(int64_t m, int64_t k, int64_t n_batch, int slice_shift, int slice_width,
                                               int parts, int homo) {
  const int64_t n_slices = n_slices_of(k, slice_shift, slice_width);
  const int64_t acc_bytes = homo ? 4 : 8;
  // a task's partial sums are cap_of() accumulators wide; the layout is not an argument here, so size for the widest
  // rounding any layout applies to the width (h8 / homo u16: up to the next multiple of 4)
  const int64_t task = std::max<int64_t>(1ll << slice_shift, (width_of(slice_shift, slice_width) + 3) & ~3ll);
  return counts_bytes(n_batch) + n_batch * plan_active_stride(m) * 4 +
         be_align_up(n_batch * n_slices * parts * task * acc_bytes, 256);
}
// ... plus room for the pre-gathered segment table when the step would use it: single vector, >= 40 slices, short blocks
// (block_hint = the plan's average items per block, as passed to the step).  A step that finds only the smaller workspace
// runs without the table — correct, slower for short blocks.
int64_t be_binary_csrmm_t_plan_workspace_bytes_for(int64_t m, int64_t k, int64_t n_batch, int slice_shift, int slice_width,
                                                   int parts, int homo, int block_hint) {
  const int64_t base = be_binary_csrmm_t_plan_workspace_bytes(m, k, n_batch, slice_shift, slice_width, parts, homo);
  const bool pre = block_hint > 0 && block_hint <= pregather_max_block();
  return base + (pre ? plan_dense_seg_bytes(m, n_slices_of(k, slice_shift, slice_width), n_batch) : 0);
}
int64_t be_binary_csrmv_t_plan_workspace_bytes(int64_t m, int64_t k, int slice_shift, int slice_width, int parts, int homo) {
  return be_binary_csrmm_t_plan_workspace_bytes(m, k, 1, slice_shift, slice_width, parts, homo);
}

int be_binary_csrmm_t_plan(const void* weights, int homo, int wdtype, const void* blob, const void* seg,
                           const void* spikes, int spike_dtype, void* out, int64_t m, int64_t k, int64_t n_batch,
                           int slice_shift, int slice_width, int layout, int block_hint, int parts, int scale_exp,
                           void* workspace, int64_t workspace_bytes, be_stream_t stream) {
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
  // what the launch below actually writes (independent of the sizing function above)
  BE_REQUIRE(workspace_bytes >= counts_bytes(n_batch) + n_batch * plan_active_stride(m) * 4 +
                                    n_batch * (int64_t)n_slices_of(k, slice_shift, slice_width) * parts * S * (homo ? 4 : 8),
             BE_ERR_WORKSPACE, "workspace too small for this layout's partial sums");
  hipStream_t st = static_cast<hipStream_t>(stream);
  unsigned char* wsb = static_cast<unsigned char*>(workspace);
  uint32_t* count = reinterpret_cast<uint32_t*>(wsb);
  uint32_t* active = reinterpret_cast<uint32_t*>(wsb + counts_bytes(n_batch));
  const int64_t astride = plan_active_stride(m);
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
  // Fused step: when the accumulators leave room in LDS and the spikes are bit-packed or 1-byte, every workgroup lists the
  // active rows of its part itself (build_part_list): no compaction launch, no active list through memory.  Otherwise the
  // spikes are compacted into the workspace first (or the caller's id list is used).  The per-batch counters at the head
  // of the workspace are zero on entry (caller contract) and are zeroed again by k_plan_reduce: no memset node per step.
  static const int fused_off = [] { const char* e = getenv("BE_PLAN_FUSED"); return e && e[0] == '0'; }();
  const size_t lds_room = 160 * 1024 - 1024 - lds;                    // bytes of LDS the accumulators leave
  int fused = 0;
  if (!fused_off && lds + 1024 <= 160 * 1024 && lds_room >= kFusedMinList * 4) {
    if (spike_dtype == BE_SPIKE_BITS) fused = 1;
    else if (spike_dtype == BE_SPIKE_BOOL && (reinterpret_cast<uintptr_t>(spikes) & 15) == 0 && (n_batch == 1 || m % 16 == 0)) fused = 2;
  }
  // short blocks: the segment-table gather dominates the step -> pre-gather it (k_gather_seg, FUSED = 3) instead of fusing
  // the compaction (a workgroup that lists its own rows would have to gather its table entries itself)
  const int64_t dense_bytes = plan_dense_seg_bytes(m, n_slices, n_batch);
  if (dense_bytes > 0 && block_hint > 0 && block_hint <= pregather_max_block() &&
      workspace_bytes >= be_binary_csrmm_t_plan_workspace_bytes(m, k, n_batch, slice_shift, slice_width, parts, homo) + dense_bytes)
    fused = 3;           // (only with a workspace sized by ..._workspace_bytes_for: the table sits behind the partial sums)
  const uint32_t lds_cap = (fused == 1 || fused == 2) ? (uint32_t)std::min<size_t>(lds_room / 4, 32768) : 0u;
  const uint32_t list_off = (uint32_t)lds;
  const size_t lds_dyn = lds + (size_t)lds_cap * 4;
  const int64_t fused_stride = spike_dtype == BE_SPIKE_BITS ? ((m + 31) / 32) * 4 : m;       // bytes per batch row of spikes
  const int64_t glob_region = fused == 3 ? astride : fused_region_of(m, parts);
  ActiveList al{active, count};
  // fused == 3: bit-packed / aligned 1-byte spikes are compacted by the gather launch itself
  const int pre_mode = fused != 3 ? 0 : (spike_dtype == BE_SPIKE_BITS ? 1 :
                       (spike_dtype == BE_SPIKE_BOOL && (reinterpret_cast<uintptr_t>(spikes) & 15) == 0) ? 2 : 0);
  if (fused == 0 || (fused == 3 && pre_mode == 0)) {
    int rc = be_resolve_active(spikes, spike_dtype, m, n_batch, active, astride, count, st, /*zero_first=*/false, &al);
    if (rc != BE_OK) return rc;
  }
  const void* seg_used = seg;
  if (fused == 3) {       // dense[slice][list position] behind the partial sums
    uint2* dense = reinterpret_cast<uint2*>(static_cast<unsigned char*>(partial) +
                                            be_align_up((int64_t)n_slices * parts * S * (homo ? 4 : 8), 256));
    const unsigned cg_grid = (unsigned)(((m + 15) / 16 + 255) / 256);
    if (pre_mode == 1)
      hipLaunchKernelGGL(k_compact_gather_seg<1>, dim3(cg_grid), dim3(256), 0, st, spikes, m, static_cast<const uint2*>(seg), n_slices,
                         astride, dense, count);
    else if (pre_mode == 2)
      hipLaunchKernelGGL(k_compact_gather_seg<2>, dim3(cg_grid), dim3(256), 0, st, spikes, m, static_cast<const uint2*>(seg), n_slices,
                         astride, dense, count);
    else
      hipLaunchKernelGGL(k_gather_seg, dim3(512), dim3(256), 0, st, static_cast<const uint2*>(seg), al.ids, al.count, n_slices,
                         astride, dense);
    BE_LAUNCH_CHECK();
    seg_used = dense;
  }
  const float scale = ldexpf(1.0f, scale_exp - 32);   // see fixed_from_f32
  const double inv_scale = ldexp(1.0, -scale_exp);
  const int n_tasks = n_slices * parts;
  const dim3 grid((unsigned)((n_tasks + 7) / 8 * 8), (unsigned)n_batch), block(1024);
  const int prof = be_prof_begin(st);
#define BE_FUSED_ARGS spikes, m, fused_stride, list_off, lds_cap, active, glob_region
#define BE_PLAN_LAUNCH(KERN, ...)                                                                              \
  do {                                                                                                         \
    auto kern__ = KERN;                                                                                        \
    BE_HIP(be_allow_lds(reinterpret_cast<const void*>(kern__), (int)lds_dyn));                                 \
    hipLaunchKernelGGL(kern__, grid, block, lds_dyn, st, static_cast<const unsigned char*>(blob),              \
                       static_cast<const uint2*>(seg_used), al.ids, al.count, __VA_ARGS__, BE_FUSED_ARGS);     \
  } while (0)
#define BE_PLAN_BY_FUSED(TMPL_A, TMPL_B, ...)                                                                 \
  do {                                                                                                         \
    if (fused == 1) BE_PLAN_LAUNCH((TMPL_A 1 TMPL_B), __VA_ARGS__);                                           \
    else if (fused == 2) BE_PLAN_LAUNCH((TMPL_A 2 TMPL_B), __VA_ARGS__);                                      \
    else if (fused == 3) BE_PLAN_LAUNCH((TMPL_A 3 TMPL_B), __VA_ARGS__);                                      \
    else BE_PLAN_LAUNCH((TMPL_A 0 TMPL_B), __VA_ARGS__);                                                      \
  } while (0)
  if (layout == BE_PLAN_H8) {
    BE_PLAN_BY_FUSED(k_plan_accumulate_h8<, >, n_slices, (int)S, parts, static_cast<uint32_t*>(partial), astride);
  } else if (homo) {
    // short blocks: 4 lanes (<= 32 entries on average) or 16 lanes (<= 128) per block instead of a wave
    uint32_t* pp = static_cast<uint32_t*>(partial);
    if (block_hint > 0 && block_hint <= kSub4MaxBlock) BE_PLAN_BY_FUSED(k_plan_accumulate<true COMMA 4 COMMA, >, n_slices, slice_shift, parts, scale, pp, astride, (int)S);
    else if (block_hint > 0 && block_hint <= kSub8MaxBlock) BE_PLAN_BY_FUSED(k_plan_accumulate<true COMMA 8 COMMA, >, n_slices, slice_shift, parts, scale, pp, astride, (int)S);
    else if (block_hint > 0 && block_hint <= kSub16MaxBlock) BE_PLAN_BY_FUSED(k_plan_accumulate<true COMMA 16 COMMA, >, n_slices, slice_shift, parts, scale, pp, astride, (int)S);
    else if (block_hint > 0 && block_hint <= kSub32MaxBlock) BE_PLAN_BY_FUSED(k_plan_accumulate<true COMMA 32 COMMA, >, n_slices, slice_shift, parts, scale, pp, astride, (int)S);
    else BE_PLAN_BY_FUSED(k_plan_accumulate<true COMMA 0 COMMA, >, n_slices, slice_shift, parts, scale, pp, astride, (int)S);
  } else if (layout == BE_PLAN_D8) {
    // short blocks: 8 lanes (<= 24 entries on average) or 16 lanes (<= 48) per block instead of a wave
    unsigned long long* pp = static_cast<unsigned long long*>(partial);
    if (block_hint > 0 && block_hint <= kD8EighthMaxBlock) BE_PLAN_BY_FUSED(k_plan_accumulate_d8<8 COMMA, >, n_slices, (int)S, parts, scale, pp, astride);
    else if (block_hint > 0 && block_hint <= kD8QuarterMaxBlock) BE_PLAN_BY_FUSED(k_plan_accumulate_d8<16 COMMA, >, n_slices, (int)S, parts, scale, pp, astride);
    else if (block_hint >= kD8NtMinBlock) BE_PLAN_BY_FUSED(k_plan_accumulate_d8<0 COMMA, COMMA BE_BLOCK_AUX>, n_slices, (int)S, parts, scale, pp, astride);
    else BE_PLAN_BY_FUSED(k_plan_accumulate_d8<0 COMMA, COMMA 0>, n_slices, (int)S, parts, scale, pp, astride);
  } else {
    unsigned long long* pw = static_cast<unsigned long long*>(partial);
    if (block_hint > 0 && block_hint <= kSubW4MaxBlock) BE_PLAN_BY_FUSED(k_plan_accumulate<false COMMA 4 COMMA, >, n_slices, slice_shift, parts, scale, pw, astride, (int)S);
    else if (block_hint > 0 && block_hint <= kSubW8MaxBlock) BE_PLAN_BY_FUSED(k_plan_accumulate<false COMMA 8 COMMA, >, n_slices, slice_shift, parts, scale, pw, astride, (int)S);
    else if (block_hint > 0 && block_hint <= kSubW16MaxBlock) BE_PLAN_BY_FUSED(k_plan_accumulate<false COMMA 16 COMMA, >, n_slices, slice_shift, parts, scale, pw, astride, (int)S);
    else if (block_hint > 0 && block_hint <= kSubW32MaxBlock) BE_PLAN_BY_FUSED(k_plan_accumulate<false COMMA 32 COMMA, >, n_slices, slice_shift, parts, scale, pw, astride, (int)S);
    else BE_PLAN_BY_FUSED(k_plan_accumulate<false COMMA 0 COMMA, >, n_slices, slice_shift, parts, scale, pw, astride, (int)S);
  }
#undef BE_PLAN_BY_FUSED
#undef BE_PLAN_LAUNCH
#undef BE_FUSED_ARGS
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
                           int slice_width, int layout, int block_hint, int parts, int scale_exp, void* workspace,
                           int64_t workspace_bytes, be_stream_t stream) {
  return be_binary_csrmm_t_plan(weights, homo, wdtype, blob, seg, spikes, spike_dtype, out, m, k, 1, slice_shift, slice_width,
                                layout, block_hint, parts, scale_exp, workspace, workspace_bytes, stream);
}

}  // extern "C"
