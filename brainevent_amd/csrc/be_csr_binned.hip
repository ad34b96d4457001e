// be_csr_binned.hip — the binned scatter route (no per-matrix layout) for gfx950; see the section comment below.
#include "be_csr_shared.h"
#include <cstdlib>

namespace {

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
//   Every bin has EIGHT regions, one per XCD (the workgroup reads its XCC id): the writers of a 128-byte line then all
//   sit behind the same L2, which merges their partial writes into whole lines before they leave for HBM — with one
//   region per bin the short runs of different XCDs met in the same lines and each L2 wrote its own masked copy.
// HBM traffic per update (hetero): 8 B read (pass B) + 6 B write + 6 B read = 20 B  vs  8 B algorithmic.
// =================================================================================================
#ifndef BE_BIN_U
#define BE_BIN_U 4       // groups of 8 binned entries in flight per thread of pass C (counted C4: 0.305 -> 0.282 ms; weighted: no change)
#endif
constexpr int kMaxBins = 2048;       // 3 x 4 B x 2048 = 24 KiB of LDS bookkeeping (16-wave kernel; the 8-wave one takes 1024)
constexpr int kBinRegions = 8;       // regions per bin: one per XCD
constexpr uint16_t kBinPad = 0xffffu; // column marker of a pad entry in a bin of counted entries (local columns are < 2^15)

// phase stamps of k_bin_rows (diagnostic builds only: -DBE_BIN_PROF; BE_HIPCC_FLAGS of brainevent_amd._lib.build)
#ifdef BE_BIN_PROF
__device__ unsigned long long g_bin_prof[256 * 8];
#define BIN_STAMP(i) do { if (tid == 0) { const unsigned long long t__ = __builtin_amdgcn_s_memtime(); prof_acc[i] += t__ - prof_t; prof_t = t__; } } while (0)
#else
#define BIN_STAMP(i) do { } while (0)
#endif

__device__ __forceinline__ uint32_t xcc_id() {
  uint32_t x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  return x & 7u;
}

// block-wide inclusive scan over NW waves (wave shuffles + one LDS hop)
template <int NW>
__device__ __forceinline__ uint32_t block_scan_nw(uint32_t v, uint32_t* wave_tot /* [NW] in LDS */) {
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
  for (int w = 0; w < NW; ++w)
    if (w < wave) base += wave_tot[w];
  __syncthreads();
  return base + incl;
}
// entries per LDS batch of (uint16 column [, f32 weight]) payload
template <bool HOMO> struct BinBatch { static constexpr int n = HOMO ? 32768 : 16384; };   // 64 / 96 KiB of payload

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
  for (int64_t i = t; i < (int64_t)n_bins * kBinRegions; i += stride) { cursor[i] = 0u; valid[i] = 0xffffffffu; }
}

// A batch is cut into chunks of 64 consecutive entries of one row piece (one per lane); wave w owns chunks
// w * SLOTS .. w * SLOTS + SLOTS - 1 of the batch and keeps them in REGISTERS from the first read to the placement:
// every load of the batch is in flight at once (16 / 48 independent 256-byte reads per wave — a row is a random
// 0.5 .. 4 KB read, and the first version, four chunks in flight and a second pass over the rows for the placement,
// spent most of a batch waiting for HBM round trips: 28 us per 16384 entries, 2.5 TB/s), and the rows are read once.
template <bool HOMO> struct BinSlots { static constexpr int n = HOMO ? 32 : 16; };     // registers per lane: 32 / 16 + 16

// NW = waves per workgroup (16: one workgroup per CU; the bookkeeping arrays and the batch scale with it).
template <typename W, bool HOMO, int NW>
__global__ void __launch_bounds__(NW * 64, 4) k_bin_rows(const W* __restrict__ weights, const int32_t* __restrict__ indices, RowPtr rp,
                                                       const uint32_t* __restrict__ active, const uint32_t* __restrict__ n_active_p,
                                                       int slice_shift, int n_bins, uint32_t cap_x, uint32_t* __restrict__ bin_cursor,
                                                       uint32_t* __restrict__ bin_valid, uint16_t* __restrict__ bin_idx,
                                                       float* __restrict__ bin_w, float* __restrict__ out, uint32_t kChunks,
                                                       uint32_t payload_entries) {
  // kChunks = chunks per batch (<= NW * SLOTS, what the registers hold; fewer when many bins leave less LDS for the batch);
  // payload_entries = kChunks * 64 + 3 * n_bins rounded up to 8: every run is padded to a multiple of 4 entries
  constexpr int SLOTS = BinSlots<HOMO>::n;
  const uint32_t kBatch = kChunks * 64u;               // entries per batch
  constexpr int NT = NW * 64;                          // threads = rows a batch can hold (one thread per row)
  constexpr int MAXB = NW == 16 ? kMaxBins : kMaxBins / 2;
  // weighted entries keep the rank the histogram atomic returns (one LDS atomic per entry); counted entries hold twice the
  // slots per lane and have no registers left for it: they draw the rank with a second atomic at placement time
  constexpr bool RANKED = !HOMO;
  __shared__ uint32_t hist[MAXB], offs[MAXB], gpos[MAXB];
  __shared__ uint32_t fill[RANKED ? 1 : MAXB];
  extern __shared__ __align__(16) unsigned char bin_payload[];        // [f32 weight x payload_entries][u16 column x payload_entries]
  float* s_w = reinterpret_cast<float*>(bin_payload);
  uint16_t* s_idx = reinterpret_cast<uint16_t*>(bin_payload + (HOMO ? 0 : (size_t)payload_entries * 4));
  __shared__ uint32_t s_lens[NT];         // batch: piece length
  __shared__ uint32_t s_cstart[NT];       //        first chunk of the piece
  __shared__ int64_t s_begin[NT];         //        first entry of the piece
  __shared__ uint32_t s_wtot[NW];
  __shared__ uint32_t s_nrows, s_nchunks;
  __shared__ uint64_t s_next;             // next list position of this workgroup
  __shared__ int64_t s_carry_begin;       // unfinished tail of a long row
  __shared__ uint32_t s_carry_len;

  const uint32_t n_active = *n_active_p;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t mask = (1u << slice_shift) - 1u;
  const uint32_t xcc = xcc_id();          // this workgroup's XCD: it writes the regions (bin, xcc)
  float w0 = 0.f;
  if (HOMO) w0 = (float)WTraits<W>::load(weights, 0);
  if (tid == 0) { s_next = blockIdx.x; s_carry_len = 0; }
  __syncthreads();
#ifdef BE_BIN_PROF
  unsigned long long prof_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, prof_t = __builtin_amdgcn_s_memtime();
#endif
  // fixed-length rows that fit a batch (at most NT per batch: one thread per row)
  const bool fixed_rows = rp.p == nullptr && rp.fixed > 0 && rp.fixed <= (int64_t)kBatch;
  const uint32_t fixed_nch = fixed_rows ? (uint32_t)((rp.fixed + 63) >> 6) : 1u;
  const uint32_t fixed_per_batch = fixed_rows ? (kChunks / fixed_nch < (uint32_t)NT ? kChunks / fixed_nch : (uint32_t)NT) : 0u;

  for (;;) {
    // ---- form a batch: thread t looks at this workgroup's t-th next row (loads in parallel), a block scan of
    //      the rows' chunk counts picks the longest prefix that fits one batch; a row longer than a batch is
    //      processed alone, one batch-sized piece at a time (carry)
    for (int b = tid; b < n_bins; b += NT) {
      hist[b] = 0;
      if (!RANKED) fill[b] = 0;
    }
    if (fixed_rows) {                // rows of one length (FixedNumPerPre): the batch is arithmetic — no scan, no search
      const uint64_t nx = s_next;
      const uint64_t left = nx < n_active ? (n_active - nx + gridDim.x - 1) / gridDim.x : 0;
      const uint32_t nr = left < fixed_per_batch ? (uint32_t)left : fixed_per_batch;
      if ((uint32_t)tid < nr) s_begin[tid] = (int64_t)active[nx + (uint64_t)tid * gridDim.x] * rp.fixed;
      __syncthreads();
      if (tid == 0) { s_next = nx + (uint64_t)nr * gridDim.x; s_nrows = nr; s_nchunks = nr * fixed_nch; }
    } else if (s_carry_len) {        // uniform: shared state
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
      // chunk counts saturate at kChunks + 1 per row; NT of them cannot overflow 32 bits
      const uint32_t nch = len > (uint64_t)kBatch ? kChunks + 1u : (uint32_t)((len + 63u) >> 6);
      const uint32_t incl = block_scan_nw<NW>(nch, s_wtot);
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
    if (fixed_rows) __syncthreads();                         // publish s_nrows / s_nchunks of the arithmetic path
    const uint32_t nrows = s_nrows;
    if (nrows == 0) break;
    BIN_STAMP(0);      // batch formed

    // ---- this wave's chunks: lane s < SLOTS finds the piece of chunk wave * SLOTS + s (last piece whose first chunk
    //      is <= the chunk id: empty pieces share their start with the piece that follows them)
    int64_t my_e0 = 0;
    uint32_t my_n = 0;
    {
      const uint32_t cid = (uint32_t)wave * SLOTS + (uint32_t)lane;
      if (fixed_rows) {
        if (lane < SLOTS && cid < s_nchunks) {
          const uint32_t pc = cid / fixed_nch;
          const uint32_t j0 = (cid - pc * fixed_nch) << 6;
          my_e0 = s_begin[pc] + j0;
          const uint32_t left = (uint32_t)rp.fixed - j0;
          my_n = left < 64u ? left : 64u;
        }
      } else if (lane < SLOTS && cid < s_nchunks) {
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
    BIN_STAMP(1);      // loads issued
    // ---- phase 1: histogram of the batch over the bins; the value the atomic returns is the entry's rank inside its bin
    //      (weighted entries: one LDS atomic per entry instead of one in a histogram pass and one in the placement pass)
    uint32_t rank[RANKED ? SLOTS : 1];
#pragma unroll
    for (int sl = 0; sl < SLOTS; ++sl) {
      if (RANKED) rank[sl] = 0u;
      if ((sl < 32 ? cnt_mask_lo : cnt_mask_hi) >> (sl & 31) & 1u) {
        if (RANKED) rank[sl] = atomicAdd(&hist[col[sl] >> slice_shift], 1u);
        else atomicAdd(&hist[col[sl] >> slice_shift], 1u);
      }
    }
    __syncthreads();
    BIN_STAMP(2);      // loads landed + histogram
    // ---- phase 2: exclusive scan of hist (two bins per thread) + one range reservation per (bin, XCD)
    uint32_t g0 = 0u, g1 = 0u;
    {
      // runs are padded to multiples of 4 entries in LDS and in the bin (8-byte / 16-byte aligned pieces: the copy-out
      // moves 4 entries per lane and instruction instead of one).  A pad is (some column of the slice, weight 0) with
      // weights — it adds zero to an accumulator; the columns are spread because equal addresses serialise an LDS atomic —
      // and the marker kBinPad without (counted entries: the accumulate pass skips it).
      const uint32_t v0 = (2 * tid < n_bins) ? hist[2 * tid] : 0u, v1 = (2 * tid + 1 < n_bins) ? hist[2 * tid + 1] : 0u;
      const uint32_t p0 = (v0 + 3u) & ~3u, p1 = (v1 + 3u) & ~3u;
      const uint32_t excl = block_scan_nw<NW>(p0 + p1, s_wtot) - (p0 + p1);
      // the range reservations are returning global atomics (a memory-side round trip of 1-3 us under load): they are
      // issued here and their values are stored to LDS only after the placement phase, behind a barrier that waits for
      // LDS traffic alone (__syncthreads() would drain them first)
      if (2 * tid < n_bins) {
        offs[2 * tid] = excl;
        g0 = v0 ? atomicAdd(&bin_cursor[(2 * tid) * kBinRegions + xcc], p0) : 0u;
        for (uint32_t q = excl + v0; q < excl + p0; ++q) { s_idx[q] = HOMO ? kBinPad : (uint16_t)((q * 40503u) & mask); if (!HOMO) s_w[q] = 0.f; }
      }
      if (2 * tid + 1 < n_bins) {
        offs[2 * tid + 1] = excl + p0;
        g1 = v1 ? atomicAdd(&bin_cursor[(2 * tid + 1) * kBinRegions + xcc], p1) : 0u;
        for (uint32_t q = excl + p0 + v1; q < excl + p0 + p1; ++q) { s_idx[q] = HOMO ? kBinPad : (uint16_t)((q * 40503u) & mask); if (!HOMO) s_w[q] = 0.f; }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    BIN_STAMP(3);      // scan + reservations issued
    // ---- phase 3: place the entries into the LDS batch sorted by bin, straight from the registers
#pragma unroll
    for (int sl = 0; sl < SLOTS; ++sl) {
      if ((sl < 32 ? cnt_mask_lo : cnt_mask_hi) >> (sl & 31) & 1u) {
        const uint32_t bin = col[sl] >> slice_shift;
        const uint32_t pos = offs[bin] + (RANKED ? rank[sl] : atomicAdd(&fill[bin], 1u));
        s_idx[pos] = (uint16_t)(col[sl] & mask);
        if (!HOMO) s_w[pos] = wv[sl];
      }
    }
    if (2 * tid < n_bins) gpos[2 * tid] = g0;
    if (2 * tid + 1 < n_bins) gpos[2 * tid + 1] = g1;
    __syncthreads();
    BIN_STAMP(4);      // placement
    // ---- phase 4: copy the padded runs out, 4 entries per lane (one 8-byte column piece + one 16-byte weight piece) and
    //      8 ... 64 lanes per bin; runs that do not fit their region go through global atomics
    {
      const uint32_t run = kBatch / (uint32_t)n_bins;             // expected run length
      const int lpb_shift = run > 128u ? 6 : (run > 64u ? 5 : (run > 32u ? 4 : 3));
      const int LPB = 1 << lpb_shift, BPW = 64 >> lpb_shift;      // lanes per bin, bins per wave and iteration
      const int grp = lane >> lpb_shift, gl = lane & (LPB - 1);
      for (int bin0 = wave * BPW; bin0 < n_bins; bin0 += NW * BPW) {
        const int bin = bin0 + grp;
        uint32_t cnt = 0, o = 0, g = 0;
        if (bin < n_bins) { cnt = hist[bin]; o = offs[bin]; g = gpos[bin]; }
        const uint32_t cntp = (cnt + 3u) & ~3u;
        const bool fits = (uint64_t)g + cntp <= cap_x;
        // a full region: everything from position g on is NOT in it (later reservations start even higher)
        if (cnt && !fits && gl == 0) atomicMin(&bin_valid[bin * kBinRegions + xcc], g);
        const int64_t base = ((int64_t)bin * kBinRegions + xcc) * cap_x + g;
        if (fits) {
          uint2* di = reinterpret_cast<uint2*>(bin_idx + base);            // g is a multiple of 4, regions are 128-B aligned
          float4* dw = reinterpret_cast<float4*>(bin_w + base);
          const uint2* si = reinterpret_cast<const uint2*>(s_idx + o);     // o is a multiple of 4
          const float4* sw = reinterpret_cast<const float4*>(s_w + o);
          for (uint32_t j = gl; j < (cntp >> 2); j += LPB) {
            di[j] = si[j];
            if (!HOMO) dw[j] = sw[j];
          }
        } else {
          float* dst = out + ((int64_t)bin << slice_shift);
          for (uint32_t j = gl; j < cnt; j += LPB) atomicAdd(dst + s_idx[o + j], HOMO ? w0 : s_w[o + j]);
        }
      }
    }
    __syncthreads();
    BIN_STAMP(5);      // copy-out
  }
#ifdef BE_BIN_PROF
  if (tid == 0 && blockIdx.x < 256)
    for (int i = 0; i < 8; ++i) g_bin_prof[blockIdx.x * 8 + i] += prof_acc[i];
#endif
}

// eight counted entries (uint16 columns, two per dword); pads carry kBinPad and are skipped
__device__ __forceinline__ void bin_count8(uint32_t* acc, uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
  const uint32_t v[4] = {a, b, c, d};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const uint32_t lo = v[i] & 0xffffu, hi = v[i] >> 16;
    if (lo != kBinPad) atomicAdd(&acc[lo], 1u);
    if (hi != kBinPad) atomicAdd(&acc[hi], 1u);
  }
}

template <bool HOMO>
__global__ void __launch_bounds__(1024) k_bin_accumulate(const uint16_t* __restrict__ bin_idx, const float* __restrict__ bin_w,
                                                         const uint32_t* __restrict__ bin_cursor,
                                                         const uint32_t* __restrict__ bin_valid, uint32_t cap, int slice_shift,
                                                         int parts, int64_t k, float scale, double inv_scale,
                                                         const void* __restrict__ w0p, int wdtype, float* __restrict__ out) {
  using acc_t = typename PlanAcc<HOMO>::type;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  acc_t* acc = reinterpret_cast<acc_t*>(smem_raw);
  const int S = 1 << slice_shift;
  const int bin = blockIdx.x / parts, part = blockIdx.x - bin * parts;
  // entries of each of the bin's regions (one per XCD); valid = first position that was NOT written (cap if it never overflowed)
  uint32_t cnts[kBinRegions];
  uint32_t any = 0;
  bool overflowed = false;                  // some run of this bin went through global atomics instead
#pragma unroll
  for (int x = 0; x < kBinRegions; ++x) {
    uint32_t c = bin_cursor[bin * kBinRegions + x];
    const uint32_t valid = bin_valid[bin * kBinRegions + x];
    overflowed |= valid != 0xffffffffu;
    c = c < valid ? c : valid;
    cnts[x] = c < cap ? c : cap;
    any |= cnts[x];
  }
  float w0 = 0.f;
  if (HOMO) {
    if (wdtype == BE_F16) w0 = __half2float(static_cast<const __half*>(w0p)[0]);
    else if (wdtype == BE_BF16) w0 = __bfloat162float(static_cast<const __hip_bfloat16*>(w0p)[0]);
    else w0 = static_cast<const float*>(w0p)[0];
  }
  if (any == 0) return;                     // nothing was binned here (out already holds zeros / overflow adds)
  for (int i = threadIdx.x; i < S; i += blockDim.x) acc[i] = 0;
  __syncthreads();
  // one flat loop over the groups of 8 entries of all eight regions (all loads independent: a loop per region started
  // every region with a dependent round trip and left most of the workgroup idle on its tail)
  uint32_t gstart[kBinRegions + 1];
  gstart[0] = 0;
#pragma unroll
  for (int x = 0; x < kBinRegions; ++x) gstart[x + 1] = gstart[x] + ((cnts[x] + 7u) >> 3);
  const uint32_t n8 = gstart[kBinRegions];
  const uint32_t per = (n8 + parts - 1) / parts;
  const uint32_t g_begin = part * per, g_end = g_begin + per < n8 ? g_begin + per : n8;
  const uint16_t* bin_i = bin_idx + (int64_t)bin * kBinRegions * cap;
  const float* bin_f = bin_w + (int64_t)bin * kBinRegions * cap;
  // BE_BIN_U groups of 8 entries per thread and round, every load of the round issued before the first add: one group per
  // round left the workgroup waiting a full memory latency per 48 bytes and thread (80 rounds of ~2 us at C4: the kernel
  // ran at 3.7 TB/s of its 600 MB).  A group that is not a whole one (a region's last, or past the end) loads the bin's
  // first entries instead — no branch around the loads — and goes through the entry-by-entry tail.
  constexpr int U = BE_BIN_U;
  for (uint32_t g0 = g_begin + threadIdx.x; g0 < g_end; g0 += U * blockDim.x) {
    uint4 iv[U];
    float4 wa[U], wb[U];
    uint32_t e0s[U], cnts_u[U];
    int xs[U];
    bool whole[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t g = g0 + (uint32_t)u * blockDim.x;
      int x = 0;
#pragma unroll
      for (int q = 1; q < kBinRegions; ++q) x += g >= gstart[q] ? 1 : 0;
      uint32_t gs = gstart[0], cnt = cnts[0];
#pragma unroll
      for (int q = 1; q < kBinRegions; ++q) if (x == q) { gs = gstart[q]; cnt = cnts[q]; }
      const uint32_t e0 = (g - gs) * 8u;
      whole[u] = g < g_end && e0 + 8u <= cnt;        // cap is a multiple of 64: the regions are 128-byte aligned
      xs[u] = x; e0s[u] = e0; cnts_u[u] = g < g_end ? cnt : 0u;
      const int64_t at = whole[u] ? (int64_t)x * cap + e0 : 0;
      iv[u] = *reinterpret_cast<const uint4*>(bin_i + at);
      if (!HOMO) {
        wa[u] = *reinterpret_cast<const float4*>(bin_f + at);
        wb[u] = *reinterpret_cast<const float4*>(bin_f + at + 4);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (whole[u]) {
        if (HOMO) {
          bin_count8(reinterpret_cast<uint32_t*>(acc), iv[u].x, iv[u].y, iv[u].z, iv[u].w);
        } else {
          plan_add4<HOMO>(acc, make_uint2(iv[u].x, iv[u].y), wa[u], scale);
          plan_add4<HOMO>(acc, make_uint2(iv[u].z, iv[u].w), wb[u], scale);
        }
      } else {
        const uint16_t* bi = bin_i + (int64_t)xs[u] * cap;
        const float* bw = bin_f + (int64_t)xs[u] * cap;
        for (uint32_t e = e0s[u]; e < cnts_u[u]; ++e) {
          if (HOMO) { if (bi[e] != kBinPad) atomicAdd(reinterpret_cast<uint32_t*>(acc) + bi[e], 1u); }
          else atomicAdd(reinterpret_cast<unsigned long long*>(acc) + bi[e], fixed_from_f32(bw[e], scale));
        }
      }
    }
  }
  __syncthreads();
  const int64_t j0 = (int64_t)bin << slice_shift;
  // parts == 1: this workgroup is the only writer of its slice after k_bin_rows has finished, so a plain store does —
  // or a plain read-modify-write when an overflowing run has already added into the slice with global atomics
  // (10M contiguous float atomics cost ~30 us at C4; the chip retires them at 1.3 TB/s of added bytes)
  const bool plain = parts == 1;
  for (int i = threadIdx.x; i < S; i += blockDim.x) {
    if (j0 + i >= k) break;
    float v;
    if (HOMO) v = (float)reinterpret_cast<uint32_t*>(acc)[i] * w0;
    else v = (float)((double)(long long)reinterpret_cast<unsigned long long*>(acc)[i] * inv_scale);
    if (plain) {
      if (!overflowed) out[j0 + i] = v;
      else if (v != 0.f) out[j0 + i] += v;
    } else if (v != 0.f) {
      atomicAdd(out + j0 + i, v);     // contiguous float atomics: the parts of a bin merge here
    }
  }
}

// f16 / bf16 outputs: the bins accumulate into an f32 image of the output (the overflow path adds f32 atomically), rounded once
template <typename W>
__global__ void __launch_bounds__(256) k_bin_round(const float* __restrict__ src, W* __restrict__ dst, int64_t k) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < k; i += stride) WTraits<W>::store(dst, i, src[i]);
}

}  // namespace

// =================================================================================================
// C ABI
// =================================================================================================
#ifdef BE_BIN_PROF
extern "C" int be_debug_bin_prof(unsigned long long* host, int reset) {
  if (host) (void)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_bin_prof), sizeof(unsigned long long) * 256 * 8);
  if (reset) { static unsigned long long z[256 * 8]; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_bin_prof), z, sizeof(z)); }
  return 0;
}
#endif

extern "C" {

// ---------------------------------------------------------------- binned route (no plan)
// entries per (bin, XCD) region: an eighth of the bin's capacity plus slack for the uneven split, in whole 128-byte lines
static inline int64_t binned_cap_x(int64_t cap) { return ((cap + 7) / 8 * 9 / 8 + 128 + 63) & ~63ll; }

int64_t be_binary_csrmv_t_binned_workspace_bytes(int64_t m, int64_t k, int slice_shift, int64_t bin_capacity) {
  const int64_t n_bins = n_slices_of(k, slice_shift);
  const int64_t cap = binned_cap_x(bin_capacity) * kBinRegions;
  return 256 + be_align_up(m * 4, 256) + 2 * be_align_up(n_bins * kBinRegions * 4, 256) + be_align_up(n_bins * cap * 2, 256) +
         be_align_up(n_bins * cap * 4, 256) + be_align_up(k * 4, 256);      // the tail: f32 image of an f16 / bf16 output
}

int be_binary_csrmv_t_binned(const void* weights, int homo, int wdtype, const int32_t* indices, const void* indptr,
                             int indptr_is_i64, int64_t row_len, const void* spikes, int spike_dtype, void* out, int64_t m,
                             int64_t k, int slice_shift, int64_t bin_capacity, int scale_exp, void* workspace,
                             int64_t workspace_bytes, be_stream_t stream) {
  BE_REQUIRE(m > 0 && k > 0 && m <= 0xffffffffll, BE_ERR_INVALID, "bad shape");
  BE_REQUIRE(wdtype == BE_F32 || wdtype == BE_F16 || wdtype == BE_BF16, BE_ERR_UNSUPPORTED,
             "the binned route supports f32 / f16 / bf16 weights (its bins carry f32; f64 weights take the planned route)");
  BE_REQUIRE(slice_shift >= 4 && slice_shift <= 15, BE_ERR_INVALID, "slice_shift must be in [4, 15]");
  BE_REQUIRE(check_rows(indptr, row_len), BE_ERR_INVALID, "indptr is NULL and row_len < 0");
  BE_REQUIRE(weights && indices && spikes && out, BE_ERR_INVALID, "null pointer");
  const int n_bins = n_slices_of(k, slice_shift);
  BE_REQUIRE(n_bins <= kMaxBins, BE_ERR_RANGE, "too many bins for the binned route");
  BE_REQUIRE(bin_capacity >= 8 && bin_capacity < (1ll << 32), BE_ERR_INVALID, "bin_capacity out of range");
  const int64_t cap_x = binned_cap_x(bin_capacity);            // per (bin, XCD) region
  const int64_t cap = cap_x * kBinRegions;
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
  uint32_t* valid = reinterpret_cast<uint32_t*>(reinterpret_cast<unsigned char*>(cursor) + be_align_up((int64_t)n_bins * kBinRegions * 4, 256));
  uint16_t* bin_idx = reinterpret_cast<uint16_t*>(reinterpret_cast<unsigned char*>(valid) + be_align_up((int64_t)n_bins * kBinRegions * 4, 256));
  float* bin_w = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(bin_idx) + be_align_up((int64_t)n_bins * cap * 2, 256));
  float* out32 = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(bin_w) + be_align_up((int64_t)n_bins * cap * 4, 256));
  void* out_user = out;
  if (wdtype != BE_F32) out = out32;                 // accumulate in f32, round once at the end
  RowPtr rp{indptr, indptr_is_i64, row_len};
  if ((reinterpret_cast<uintptr_t>(out) & 15) == 0) {
    hipLaunchKernelGGL(k_bin_reset, dim3(grid_for(k / 4 + 1, 256, 1024)), dim3(256), 0, st, static_cast<float*>(out), k, cursor,
                       valid, n_bins, count);
    BE_LAUNCH_CHECK();
  } else {
    BE_HIP(be_fill_async(out, 0, (size_t)k * 4, st));
    BE_HIP(be_fill_async(cursor, 0, (size_t)n_bins * kBinRegions * 4, st));
    BE_HIP(be_fill_async(valid, 0xff, (size_t)n_bins * kBinRegions * 4, st));
    BE_HIP(be_fill_async(count, 0, 4, st));
  }
  ActiveList al;
  int rc = be_resolve_active(spikes, spike_dtype, m, 1, active, 0, count, st, /*zero_first=*/false, &al);
  if (rc != BE_OK) return rc;
  const int prof = be_prof_begin(st);
  // one 1024-thread workgroup per CU.  (Two 512-thread workgroups with half-size batches, so that one could sort in LDS while
  // the other has its loads or its copy-out in flight, measured slower at C4 — 533 vs 422 us weighted, 284 vs 204 counted:
  // twice the batches pay twice the barriers and fixed latencies, and the runs are half as long.)
  {
    const void* kern = nullptr;
#define BE_BIN_KERN(WT) kern = homo ? (const void*)k_bin_rows<WT, true, 16> : (const void*)k_bin_rows<WT, false, 16>
    if (wdtype == BE_F16) BE_BIN_KERN(__half); else if (wdtype == BE_BF16) BE_BIN_KERN(__hip_bfloat16); else BE_BIN_KERN(float);
#undef BE_BIN_KERN
    hipFuncAttributes fa;
    BE_HIP(hipFuncGetAttributes(&fa, kern));
    // the batch payload lives in dynamic LDS next to the kernel's static bookkeeping: every run is padded to 4 entries
    const int64_t budget = 160 * 1024 - (int64_t)fa.sharedSizeBytes - 256;
    const int64_t per_entry = homo ? 2 : 6;
    const int64_t max_chunks = 16ll * (homo ? 32 : 16);
    int64_t chunks = (budget / per_entry - 3ll * n_bins - 8) / 64;
    chunks = chunks > max_chunks ? max_chunks : chunks;
    BE_REQUIRE(chunks >= 1, BE_ERR_RANGE, "too many bins for the LDS batch of the binned route");
    const uint32_t payload_entries = (uint32_t)((chunks * 64 + 3ll * n_bins + 7) & ~7ll);
    const size_t dyn = (size_t)payload_entries * per_entry;
    BE_HIP(be_allow_lds(kern, (int)dyn));
#define BE_BIN_ROWS(WT, HOMO_)                                                                                              \
  hipLaunchKernelGGL((k_bin_rows<WT, HOMO_, 16>), dim3(256), dim3(1024), dyn, st, static_cast<const WT*>(weights), indices, rp, \
                     al.ids, al.count, slice_shift, n_bins, (uint32_t)cap_x, cursor, valid, bin_idx, bin_w,                    \
                     static_cast<float*>(out), (uint32_t)chunks, payload_entries)
#define BE_BIN_ROWS_W(WT) do { if (homo) BE_BIN_ROWS(WT, true); else BE_BIN_ROWS(WT, false); } while (0)
    if (wdtype == BE_F16) BE_BIN_ROWS_W(__half); else if (wdtype == BE_BF16) BE_BIN_ROWS_W(__hip_bfloat16); else BE_BIN_ROWS_W(float);
#undef BE_BIN_ROWS_W
#undef BE_BIN_ROWS
  }
  BE_LAUNCH_CHECK();
  // n_bins * parts ~ 256: every workgroup fills a CU (128 KB of LDS) and its slice is merged into the output with float
  // atomics, one per non-zero accumulator, so more parts than CUs only add merge traffic (39 bins: 13 parts took 55 us, 6 take 30).
  // (Beyond 256 bins the last round of workgroups is partly empty — C4: 611 bins = 2 rounds + 99 — but splitting only its
  //  bins into parts that fill the round measured nothing, 165 -> 156 us at best: the pass is bound by its byte stream.)
  int parts = 256 / (n_bins > 0 ? n_bins : 1);
  parts = parts < 1 ? 1 : (parts > 16 ? 16 : parts);
  const unsigned acc_grid = (unsigned)(n_bins * parts);
  const float scale = ldexpf(1.0f, scale_exp - 32);
  const double inv_scale = ldexp(1.0, -scale_exp);
  if (homo) {
    auto kern = k_bin_accumulate<true>;
    BE_HIP(be_allow_lds(reinterpret_cast<const void*>(kern), (int)lds));
    hipLaunchKernelGGL(kern, dim3(acc_grid), dim3(1024), lds, st, bin_idx, bin_w, cursor, valid, (uint32_t)cap_x,
                       slice_shift, parts, k, scale, inv_scale, weights, wdtype, static_cast<float*>(out));
  } else {
    auto kern = k_bin_accumulate<false>;
    BE_HIP(be_allow_lds(reinterpret_cast<const void*>(kern), (int)lds));
    hipLaunchKernelGGL(kern, dim3(acc_grid), dim3(1024), lds, st, bin_idx, bin_w, cursor, valid, (uint32_t)cap_x,
                       slice_shift, parts, k, scale, inv_scale, static_cast<const void*>(nullptr), wdtype, static_cast<float*>(out));
  }
  be_prof_end(prof, st);
  BE_LAUNCH_CHECK();
  if (wdtype == BE_F16)
    hipLaunchKernelGGL(k_bin_round<__half>, dim3(grid_for(k, 256, 2048)), dim3(256), 0, st, out32, static_cast<__half*>(out_user), k);
  else if (wdtype == BE_BF16)
    hipLaunchKernelGGL(k_bin_round<__hip_bfloat16>, dim3(grid_for(k, 256, 2048)), dim3(256), 0, st, out32,
                       static_cast<__hip_bfloat16*>(out_user), k);
  BE_LAUNCH_CHECK();
  return BE_OK;
}

}  // extern "C"
