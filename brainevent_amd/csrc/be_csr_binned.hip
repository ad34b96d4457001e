// be_csr_binned.hip — the binned scatter route (no per-matrix layout) for gfx950; see the section comment below.
#include "be_csr_shared.h"
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {

// =================================================================================================
// binned route: event-driven scatter for matrices WITHOUT a plan (or whose rows put too few entries into
// one output slice for the plan to pay: FixedNumPerPre K=1000 over 10M outputs has 1.6 entries per (row, slice)).
//
//   pass B (k_bin_stream)    : 256 workgroups of 16 waves stream the active rows.  Every workgroup keeps one
//                              write-combining block of CAP entries per output slice ("bin") in LDS; a wave reads 64
//                              entries, every lane appends its entry to its bin's block (reserve a slot with a returning
//                              LDS atomic, write, commit with a second one) and the lane whose commit completes a block
//                              has its wave copy the block to the workgroup's OWN region of that bin in memory — one
//                              contiguous store of 48 ... 768 bytes, no global atomics, no workgroup barrier anywhere
//                              in the stream: loads, LDS traffic and stores of the 16 waves overlap all the time.
//                              (Rounds 1-2 sorted whole batches of 16384 entries per workgroup: histogram, scan,
//                              placement and copy-out phases behind barriers left HBM idle 41 % of the kernel.)
//   pass C (k_bin_accumulate): one workgroup per (bin, part) streams the bin's 256 regions (entry counts from the
//                              directory pass B leaves behind) and accumulates in LDS with integer atomics exactly
//                              like the planned route, then writes its slice of the output.
//   A region that overflows its capacity never corrupts anything: the block is delivered with global float
//   atomics instead (slow path, still correct).
// Block layout in LDS and in memory: CAP / 2 units of [f32 w0][f32 w1][u16 c0 | u16 c1] (one weight: [u16 column x CAP]).
// HBM traffic per update (hetero): 8 B read (pass B) + 6 B write + 6 B read = 20 B  vs  8 B algorithmic.
// =================================================================================================
#ifndef BE_BIN_U
#define BE_BIN_U 4       // groups of 8 binned entries in flight per thread of pass C
#endif
#ifndef BE_RING
#define BE_RING 2         // write-combining blocks per bin (a ring: an entry of block q waits for block q - BE_RING to leave)
#endif
#ifndef BE_FLUSH_PJ
#define BE_FLUSH_PJ 1    // (2 / 4 passes per LDS round trip measured slower: 418 -> 455 / 464 us at C4, registers)
#endif
#ifndef BE_STREAM_THREADS
#define BE_STREAM_THREADS 1024
#endif
#ifndef BE_STREAM_U
#define BE_STREAM_U 2    // steps (64 lanes x 4 entries) a wave of pass B keeps in flight next to the ones it is appending
#endif
constexpr int64_t kShortRow = 256;       // rows up to this many entries (on average) take pass B with one step of loads in flight
constexpr int kTaskRowsFloor = 4, kTaskGroupsCap = 1024;   // pass B: a task has at least 4 rows while they stay within 1024 groups of four
constexpr int kMaxBins = 2048;
constexpr int kStreamGrid = 256;     // workgroups of pass B = regions per bin (one per CU)
constexpr int kStreamWaves = 16;
constexpr int kRing = BE_RING, kRingLog = BE_RING == 2 ? 1 : 0;
static_assert(BE_RING == 1 || BE_RING == 2, "ring of one or two blocks per bin");
// A lane whose ring slot is not freed within kWaitLimitTicks of the constant 100 MHz clock (20 ms — four orders of magnitude
// above a flush; delayed co-resident waves, a profiler or several processes sharing the card stay far inside it) gives up:
// it raises the workspace's sticky protocol flag and drops its pending entries.  Pass C then writes NaN into every output of
// the step and be_binned_workspace_status() reports BE_ERR_HIP with the cause — an error code, never a trap: a device-side
// abort would poison the HIP context of the whole process (the header promises codes).  Never observed.
constexpr uint64_t kWaitLimitTicks = 2000000ull;
constexpr int kBinErrWord = 16;             // word of the workspace head that holds the sticky flag (word 0: the spike counter)
// Conservation counters (64-bit, behind the workspace head; they only grow until be_binned_workspace_status
// clears them).  Every step adds: [0] the stored entries of its active rows (x the batch rows a row is active in), summed from
// the row bounds pass B reads; [1] the tickets pass B handed out (its LDS counters at the drain); [2] the entries pass C added
// to its accumulators (counted where they are added, not taken from the directory); [3] the entries pass B delivered through
// the overflow image.  After any number of complete steps [0] == [1] == [2] + [3]: an entry lost or delivered twice anywhere
// between the row bounds and the accumulators breaks one of the equalities and be_binned_workspace_status says which.
// ([0] > [1] also when a column id is >= k: the caller's error, such entries are dropped.)  They are kept PER WORKGROUP — pass B's
// workgroup g owns words [3 g, 3 g + 3) of the first array, workgroup w of pass C word w of the second — and added up by the
// status call on the host: a first version with one global atomic per wave on four shared words cost pass C of a C4 step 37 us
// (9776 same-address atomics serialise in L2; the post slice of an 8-way cut: 20 -> 63 us).  Cost now: one wave reduction per
// task, two LDS atomics per wave and one plain 8-byte read-modify-write per workgroup and counter.
constexpr int kBinAuditGridC = 2048;                                      // workgroups of pass C at most (kMaxBins, or 256 with parts)
constexpr int64_t kBinAuditOff = 256;                                     // byte offset of the counters in the workspace
constexpr int64_t kBinAuditBytes = (3 * 256 + kBinAuditGridC) * 8;        // [256 workgroups of pass B][3] + [kBinAuditGridC]

template <bool HOMO, int CAP> struct BinBlock {
  static constexpr int bytes = CAP * (HOMO ? 2 : 6);
  static constexpr int dwords = bytes / 4;
  // stride of a block in the GLOBAL regions (blocks starting on 128-byte lines were measured in round 4 and lose: LABNOTES.md)
  static constexpr int gdwords = dwords;
  // weighted entries sit in UNITS of 12 bytes, two entries each: [f32 w0][f32 w1][u16 c0 | u16 c1] — pass C reads the units of a
  // region as one contiguous stream, a unit per lane and load; one weight: [u16 column x CAP]
  __device__ static __forceinline__ uint32_t w_dw(uint32_t s) { return 3u * (s >> 1) + (s & 1u); }                 // dword of entry s's weight
  __device__ static __forceinline__ uint32_t col_hw(uint32_t s) { return HOMO ? s : 6u * (s >> 1) + 4u + (s & 1u); }   // halfword of its column
  static constexpr int groups = CAP / 8;               // groups of 8 entries (pass C: one per thread and round)
  // a block leaves LDS 16 bytes per lane (ds_read_b128 + one aligned 16-byte store): `lpf` lanes per block, 64 / lpf blocks
  // per pass (weighted blocks of 16 entries: 6 lanes, 10 blocks per pass)
  static constexpr int lpf = bytes / 16;
};

// exact n / d for 32-bit n by multiply-high (Granlund-Montgomery, round-up method); d >= 1.  A power of two is a shift (m == 0).
struct DivU32 {
  uint32_t m, sh1, sh2;
  __device__ __forceinline__ uint32_t div(uint32_t n) const {
    if (m == 0u) return n >> sh2;                  // (uniform)
    const uint32_t t = __umulhi(m, n);
    return (t + ((n - t) >> sh1)) >> sh2;
  }
};
static inline DivU32 make_div(uint32_t d) {
  uint32_t l = 0;
  while (l < 32 && (1ull << l) < d) ++l;
  DivU32 r;
  if ((1ull << l) == d) { r.m = 0; r.sh1 = 0; r.sh2 = l; return r; }
  r.m = (uint32_t)(((1ull << 32) * ((1ull << l) - d)) / d + 1);
  r.sh1 = l < 1 ? l : 1;
  r.sh2 = l > 1 ? l - 1 : 0;
  return r;
}

// one launch instead of several memset nodes: output <- 0 and the spike counter of the compaction that follows <- 0
__global__ void __launch_bounds__(256) k_bin_reset(float* __restrict__ out, int64_t k, uint32_t* __restrict__ count) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t == 0) count[0] = 0u;
  if ((reinterpret_cast<uintptr_t>(out) & 15) == 0) {
    float4* o4 = reinterpret_cast<float4*>(out);
    const int64_t k4 = k >> 2;
    for (int64_t i = t; i < k4; i += stride) o4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t i = (k4 << 2) + t; i < k; i += stride) out[i] = 0.f;
  } else {
    for (int64_t i = t; i < k; i += stride) out[i] = 0.f;
  }
}

// phase stamps of k_bin_stream (diagnostic builds only: -DBE_BIN_PROF; BE_HIPCC_FLAGS of brainevent_amd._lib.build)
#ifdef BE_BIN_PROF
__device__ unsigned long long g_bin_prof[256 * 8];
struct StreamProf {
  unsigned long long acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t = __builtin_amdgcn_s_memtime();
  __device__ __forceinline__ void stamp(int i) { const unsigned long long n = __builtin_amdgcn_s_memtime(); acc[i] += n - t; t = n; }
  __device__ __forceinline__ void count(int i, unsigned long long n = 1) { acc[i] += n; }
  __device__ __forceinline__ void flush_out(int lane) {
    if (lane == 0 && blockIdx.x < 256)
      for (int i = 0; i < 8; ++i) atomicAdd(&g_bin_prof[blockIdx.x * 8 + i], acc[i]);
  }
};
#else
struct StreamProf {
  __device__ __forceinline__ void stamp(int) {}
  __device__ __forceinline__ void count(int, unsigned long long = 1) {}
  __device__ __forceinline__ void flush_out(int) {}
};
#endif

__device__ __forceinline__ void lds_fence() { __atomic_signal_fence(__ATOMIC_SEQ_CST); }
__device__ __forceinline__ uint32_t rfl(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint32_t rl(uint32_t v, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, l); }

// The per-workgroup state of pass B in LDS.  Every bin has a RING OF TWO write-combining blocks of CB entries.  An entry
// draws a ticket t from its bin's counter: it belongs to block q = t / CB of the bin's stream (position q of the
// workgroup's region of that bin), slot t % CB, ring slot q & 1 — and may be written once that ring slot's previous
// occupant (block q - 2) has left: gen[ring slot] == q >> 1.  With one block per bin a full block stalled every later
// entry of the bin until its flush had finished (one wave in two met such a lane per 64 entries and went round again).
template <bool HOMO, int CB>
struct StreamLds {
  uint32_t* tick;   // [n_bins + 64]      tickets handed out (+ one dummy counter per lane for lanes without an entry)
  uint32_t* done;   // [2 * n_bins + 64]  entries written into the ring slot's current block (+ dummies)
  uint32_t* gen;    // [2 * n_bins]       blocks that have left the ring slot
  uint32_t* ovf;    // [n_bins]           != 0: some entries of the bin went through global atomics
  uint32_t* dummy;  // [128]              where lanes without a writable entry put their two stores
  uint32_t* buf;    // [2 * n_bins][B::dwords]
  uint32_t* err;    // GLOBAL: the workspace's sticky protocol flag (kBinErrWord)
};
// LDS words of pass B in front of the per-bin state: per wave a task table (64 row starts as int64, 66 prefix sums of
// the rows' 4-entry groups, 64 row lengths), a flush list of 64 (ring slot, block number) items, the task ticket, the
// dummies (128 words + 64 ticket counters + 64 commit counters)
constexpr int kStreamWl = kStreamWaves * (64 * 2 + 66 + 64);
constexpr int kStreamFixedWords = kStreamWl + kStreamWaves * 128 + 12 + 256 + 4;   // (12: task ticket + 3 conservation sums; + 4: the blocks start 16-byte aligned)

typedef uint32_t be_u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
typedef uint32_t be_u32x3_a4 __attribute__((ext_vector_type(3), aligned(4)));
typedef uint32_t be_u32x2_a4 __attribute__((ext_vector_type(2), aligned(4)));
typedef float be_f32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));

// The wave copies the `n` completed blocks of its flush list (wl[j] = ring slot id, wl[64 + j] = block number) to their places in
// the workgroup's regions, B::fpp blocks per pass (B::lpf lanes per block, B::vec dwords per lane), and frees the ring slots.
template <bool HOMO, int CB>
__device__ __forceinline__ void stream_flush_list(const StreamLds<HOMO, CB>& S, const uint32_t* wl, uint32_t n, uint32_t cap_blocks,
                                                  uint32_t* __restrict__ wg_regions, size_t bin_stride_dw, float* __restrict__ out,
                                                  uint32_t width, int n_bins_b, int64_t k, float w0, int lane) {
  using B = BinBlock<HOMO, CB>;
  constexpr int LPF = B::lpf, FPP = 64 / LPF;
  static_assert(B::bytes % 16 == 0 && LPF >= 1 && LPF <= 64, "blocks are copied 16 bytes per lane");
  constexpr int PJ = BE_FLUSH_PJ;                           // passes whose list reads, block reads and stores each go out together
  const uint32_t g = (uint32_t)lane / LPF, gl = (uint32_t)lane % LPF;
  for (uint32_t j0 = 0; j0 < n; j0 += PJ * FPP) {
    bool have[PJ];
    uint32_t slotid[PJ], q[PJ];
    uint4 v[PJ];
#pragma unroll
    for (int p = 0; p < PJ; ++p) {
      const uint32_t j = j0 + (uint32_t)p * FPP + g;
      have[p] = g < (uint32_t)FPP && j < n;
      slotid[p] = wl[have[p] ? j : 0u];
      q[p] = wl[64u + (have[p] ? j : 0u)];
    }
#pragma unroll
    for (int p = 0; p < PJ; ++p)
      if (p == 0 || j0 + (uint32_t)p * FPP < n)       // (wave-uniform)
        v[p] = *reinterpret_cast<const uint4*>(S.buf + (size_t)slotid[p] * B::dwords + gl * 4);
#pragma unroll
    for (int p = 0; p < PJ; ++p) {
      if (p > 0 && j0 + (uint32_t)p * FPP >= n) break;
      uint64_t slow = __ballot(have[p] && q[p] >= cap_blocks);
      while (slow) {                                  // (rare) the region is full: the block goes out through float atomics
        const int sl = __ffsll((unsigned long long)slow) - 1;
        slow &= slow - 1;
        if ((sl % LPF) != 0) continue;
        const uint32_t sid = rl(slotid[p], sl);
        const uint32_t* blk = S.buf + (size_t)sid * B::dwords;
        const uint16_t* bi = reinterpret_cast<const uint16_t*>(blk);
        const float* bw = reinterpret_cast<const float*>(blk);
        const uint32_t vb = sid >> kRingLog, bb = vb / (uint32_t)n_bins_b;          // (batch row, bin) of the virtual bin
        float* dst = out + (int64_t)bb * k + (int64_t)(vb - bb * (uint32_t)n_bins_b) * width;
        for (int j = lane; j < CB; j += 64) {
          const uint32_t c = bi[B::col_hw(j)];          // (< width by construction; the test keeps a float atomic inside the bin's slice whatever LDS holds)
          if (c < width) atomicAdd(dst + c, HOMO ? w0 : bw[B::w_dw(j)]);
        }
        if (lane == 0) S.ovf[sid >> kRingLog] = 1u;
      }
    }
    // LDS executes a wave's instructions in order: the blocks have been read before these stores hand their ring slots to
    // the blocks two further on (done first: an entry written after gen moves on must find the count at 0)
    lds_fence();
#pragma unroll
    for (int p = 0; p < PJ; ++p)
      if (have[p] && gl == 0) __hip_atomic_store(&S.done[slotid[p]], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    lds_fence();
#pragma unroll
    for (int p = 0; p < PJ; ++p)
      if (have[p] && gl == 0) __hip_atomic_store(&S.gen[slotid[p]], (q[p] >> kRingLog) + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    lds_fence();
#pragma unroll
    for (int p = 0; p < PJ; ++p) {
      if (have[p] && q[p] < cap_blocks)
        *reinterpret_cast<uint4*>(wg_regions + (size_t)(slotid[p] >> kRingLog) * bin_stride_dw + (size_t)q[p] * B::gdwords + gl * 4) = v[p];
    }
  }
}

// The NE entries of a lane (a lane's missing entries carry the column 0xffffffff): tickets, then — until every
// entry is in its block — the stores of the entries whose ring slot is free, their commits, and the flushes of the blocks
// those commits completed.  All NE entries go through every stage together (independent LDS operations in flight, no
// branches: a lane without an entry, or with one that has to wait, aims at counters and words of its own that nobody reads).
template <bool HOMO, int CB, int NE, bool BATCH>
__device__ __forceinline__ void stream_append(const StreamLds<HOMO, CB>& S, const uint32_t (&col)[NE],
                                               const float (&w)[HOMO ? 1 : NE], const uint32_t (&boff)[NE / 4], uint32_t width,
                                               DivU32 wdiv, int n_bins, int n_bins_b, int64_t k,
                                               uint32_t cap_blocks, uint32_t* wl, uint32_t* __restrict__ wg_regions,
                                               size_t bin_stride_dw, float* __restrict__ out, float w0, int lane, StreamProf& prof) {
  using B = BinBlock<HOMO, CB>;
  constexpr int LOG_CB = CB == 128 ? 7 : CB == 64 ? 6 : CB == 32 ? 5 : CB == 16 ? 4 : 3;
  // a column >= k (the caller's error) and a missing entry both land on the lane's own dummy counter n_bins + lane
  const uint32_t dummy_bin = (uint32_t)n_bins + (uint32_t)lane;
  uint32_t bin[NE], t[NE], lc[NE];
  uint32_t pend = 0;
#pragma unroll
  for (int u = 0; u < NE; ++u) {
    const uint32_t bl = wdiv.div(col[u]);                      // the bin inside its batch row: a missing entry's is huge
    lc[u] = col[u] - __umul24(bl, width);
    // (a single vector has no virtual bins: the bin is the bin — pass B counted 201 -> 188 us, weighted 442 -> 428 at C4 without
    //  the batch arithmetic in this path)
    const uint32_t b = !BATCH ? bl : (bl < (uint32_t)n_bins_b ? bl + boff[u / 4] : 0xffffffffu);
    bin[u] = b < dummy_bin ? b : dummy_bin;
    pend |= (b < (uint32_t)n_bins ? 1u : 0u) << u;
  }
#pragma unroll
  for (int u = 0; u < NE; ++u) t[u] = atomicAdd(&S.tick[bin[u]], 1u);
  __builtin_amdgcn_sched_barrier(0);
  uint32_t spins = 0;
  uint64_t t_wait = 0;
  for (;;) {
    uint32_t g[NE], slotid[NE];
#pragma unroll
    for (int u = 0; u < NE; ++u) {
      slotid[u] = bin[u] * (uint32_t)kRing + ((t[u] >> LOG_CB) & (uint32_t)(kRing - 1));
      g[u] = __hip_atomic_load(&S.gen[slotid[u]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __builtin_amdgcn_sched_barrier(0);
    uint32_t wr = 0;
    uint32_t d[NE];
#pragma unroll
    for (int u = 0; u < NE; ++u) {
      const bool ok = ((pend >> u) & 1u) && g[u] == (t[u] >> (LOG_CB + kRingLog));
      wr |= (ok ? 1u : 0u) << u;
      uint32_t* blk = S.buf + (size_t)slotid[u] * B::dwords;
      const uint32_t s = t[u] & (uint32_t)(CB - 1);
      uint16_t* pi = ok ? reinterpret_cast<uint16_t*>(blk) + B::col_hw(s) : reinterpret_cast<uint16_t*>(S.dummy + 64 + lane);
      *pi = (uint16_t)lc[u];
      if (!HOMO) {
        float* pw = ok ? reinterpret_cast<float*>(blk) + B::w_dw(s) : reinterpret_cast<float*>(S.dummy + lane);
        *pw = w[HOMO ? 0 : u];
      }
    }
    lds_fence();
#pragma unroll
    for (int u = 0; u < NE; ++u) d[u] = atomicAdd(&S.done[(wr >> u) & 1u ? slotid[u] : (uint32_t)kRing * (uint32_t)n_bins + (uint32_t)lane], 1u);
    __builtin_amdgcn_sched_barrier(0);
    // the blocks these commits completed go on the wave's list (positions from the ballots)
    uint32_t flm = 0;
#pragma unroll
    for (int u = 0; u < NE; ++u) flm |= (((wr >> u) & 1u) && d[u] == (uint32_t)CB - 1u ? 1u : 0u) << u;
    pend &= ~wr;
    prof.count(4);
    // (every completing lane copying its own block, 16 bytes per instruction and no list, measured slower: 406 -> 554 us at
    //  C4 — a store instruction whose few active lanes write 16 bytes each to unrelated lines costs far more than its issue slot)
    // (Leaving the list to the top of the wave's next round, so that its stores have a round of appends to complete in before
    //  the full wait for that round's loads, bought nothing at C4 — 412 -> 421 us — and made one post slice of an 8-way cut
    //  wait for ring slots: 2.3 trips through this loop per round instead of 1.)
    while (__ballot(flm != 0)) {                     // (one pass, unless more than 64 blocks completed at once)
      uint32_t nfl = 0;
#pragma unroll
      for (int u = 0; u < NE; ++u) {
        const bool fl = (flm >> u) & 1u;
        const uint64_t m = __ballot(fl);
        if (m) {                                     // (wave-uniform)
          const uint32_t at = nfl + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
          if (fl && at < 64u) {
            wl[at] = slotid[u];
            wl[64 + at] = t[u] >> LOG_CB;
            flm &= ~(1u << u);
          }
          nfl += (uint32_t)__popcll((unsigned long long)m);
        }
      }
      prof.count(5, nfl < 64u ? nfl : 64u);
      lds_fence();
      stream_flush_list<HOMO, CB>(S, wl, nfl < 64u ? nfl : 64u, cap_blocks, wg_regions, bin_stride_dw, out, width, n_bins_b, k, w0,
                                  lane);
    }
    if (__ballot(pend != 0) == 0) break;
    if ((++spins & 63u) == 0u) {                     // (wave-uniform) look at the clock every 64 rounds
      if (spins == 64u) t_wait = wall_clock64();
      else if (wall_clock64() - t_wait > kWaitLimitTicks) {
        if (pend != 0) atomicOr(S.err, 1u);          // sticky: pass C poisons the outputs, be_binned_workspace_status names it
        pend = 0;
      }
    }
    __builtin_amdgcn_s_sleep(1);
  }
}

// pass B.  Tasks of 2^rshift consecutive active rows are handed to the waves by an LDS ticket inside the workgroup's
// contiguous share of the task list; a wave flattens its task's rows into groups of four consecutive entries (prefix
// sums of the rows' group counts in a per-wave LDS table), a lane takes one group per step — one 16-byte load of columns
// and one of weights — and U steps are loading while the previous ones are appended (U = BE_STREAM_U = 2; short rows, whose
// tasks are four steps long, run U = 1: kShortRow below).
template <typename W, bool HOMO, int CB, bool BATCH, int U = BE_STREAM_U>
__global__ void __launch_bounds__(1024) k_bin_stream(const W* __restrict__ weights, const int32_t* __restrict__ indices, RowPtr rp,
                                                     const uint32_t* __restrict__ active, const uint32_t* __restrict__ n_active_p,
                                                     uint32_t width, DivU32 wdiv, int n_bins, uint32_t cap_blocks,
                                                     uint32_t* __restrict__ regions, uint32_t* __restrict__ dir,
                                                     float* __restrict__ out, DivU32 fixdiv,
                                                     const uint32_t* __restrict__ row_masks, int n_bins_b, int64_t k,
                                                     uint32_t min_tasks, uint32_t task_groups, int64_t m_rows,
                                                     uint32_t* __restrict__ err_flag, uint32_t sign_mask,
                                                     unsigned long long* __restrict__ audit) {
  // row_masks != NULL: a batch.  `active` lists the rows with a spike in ANY of the (<= 32) batch rows of this pass and
  // row_masks[j] says in which; the bins are virtual — batch row b's bin i is n_bins_b * b + i of n_bins — and an entry is
  // appended once per batch row that has its row active (the rows are read once for the whole batch).
  using B = BinBlock<HOMO, CB>;
  extern __shared__ __align__(16) uint32_t lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int64_t* tbeg = reinterpret_cast<int64_t*>(lds) + wave * 64;                  // this wave's task table
  uint32_t* tpre = lds + kStreamWaves * 64 * 2 + wave * 66;
  uint32_t* tlen = lds + kStreamWaves * (64 * 2 + 66) + wave * 64;
  uint32_t* wl = lds + kStreamWl + wave * 128;
  uint32_t* s_ticket = lds + kStreamWl + kStreamWaves * 128;
  StreamLds<HOMO, CB> S;
  unsigned long long* s_audit = reinterpret_cast<unsigned long long*>(s_ticket + 4);       // [3] (8-byte aligned: kStreamWl is even)
  S.dummy = s_ticket + 12;                       // [128], then the lanes' dummy counters live behind tick / done
  S.tick = S.dummy + 128;
  S.done = S.tick + n_bins + 64;
  S.gen = S.done + kRing * n_bins + 64;
  S.ovf = S.gen + kRing * n_bins;
  // the blocks are read 16 bytes at a time (ds_read_b128): their start is rounded up to a 4-word boundary of the allocation —
  // the arrays in front total 5 * n_bins + constant words, so rounding only the size of `ovf` left C4's 611 bins 12 bytes off
  S.buf = lds + (((S.ovf + n_bins) - lds + 3) & ~(ptrdiff_t)3);
  S.err = err_flag;
  for (int i = tid; i < (2 + 2 * kRing) * n_bins + 128; i += (int)blockDim.x) S.tick[i] = 0u;
  if (tid == 0) { s_ticket[0] = 0u; s_audit[0] = 0ull; s_audit[1] = 0ull; s_audit[2] = 0ull; }
  __syncthreads();
  unsigned long long n_expected = 0;              // stored entries of the rows this lane read the bounds of (x batch rows)

  const uint32_t n_active = *n_active_p;
  float w0 = 0.f;
  if (HOMO) w0 = (float)WTraits<W>::load(weights, 0);
  const bool fixed = rp.p == nullptr && rp.fixed > 0 && rp.fixed < (1ll << 26);
  const uint32_t K = fixed ? (uint32_t)rp.fixed : 0u, K4 = (K + 3u) >> 2;
  // rows per task (1 ... 64): about `task_groups` groups of four entries, but at least kTaskRowsFloor rows while those stay
  // within kTaskGroupsCap groups.  Measured on the round-4 kernel (tools/ab_c4_rank_tasks.sh, tools/ab_c4_tasks.sh): rows of
  // ~125 entries (one post slice of an 8-way cut of C4), 32 / 16 / 8 / 4 rows per task 104.9 / 95.5 / 93.1 / 94.3 us per step;
  // rows of 1000 entries (C4), 4 / 2 / 1 rows per task 0.597-0.629 / 0.592 / 0.590 ms weighted and 0.236 / 0.242 / 0.250 ms
  // counted: a handful of rows per task either way — shorter tasks pay their header (ticket, row ids, row bounds: dependent
  // loads) too often, longer ones leave waves without work at the end of the workgroup's share (12 tasks for 16 waves at 32
  // rows per task on the post slice).  Fewer active rows than `min_tasks` tasks of that size: smaller tasks.
  uint64_t avg_g4 = K4;
  if (!fixed) {
    const int64_t nnz = rp.at(m_rows) - rp.at(0);
    avg_g4 = m_rows > 0 ? ((uint64_t)(nnz > 0 ? nnz : 0) / (uint64_t)m_rows + 3u) >> 2 : 1u;
  }
  avg_g4 = avg_g4 ? avg_g4 : 1u;
  int rshift = 6;
  while (rshift > 0 && (avg_g4 << rshift) > (uint64_t)task_groups) --rshift;
  while ((1 << rshift) < kTaskRowsFloor && (avg_g4 << (rshift + 1)) <= (uint64_t)kTaskGroupsCap) ++rshift;
  while (rshift > 0 && (n_active >> rshift) < min_tasks) --rshift;
  while (fixed && rshift > 0 && (rp.fixed << rshift) >= (1ll << 31)) --rshift;
  const uint32_t R = 1u << rshift;
  const uint64_t n_tasks = ((uint64_t)n_active + R - 1) >> rshift;
  const uint64_t per = (n_tasks + gridDim.x - 1) / gridDim.x;
  const uint64_t t_begin = (uint64_t)blockIdx.x * per;
  const uint64_t t_end = t_begin + per < n_tasks ? t_begin + per : n_tasks;
  // (Round 5: handing out the last 5/16 of a workgroup's share in tasks of R / 4 rows — so that 16 waves drawing 49 equal tasks
  //  do not end with one wave in a fourth round — measured SLOWER on the post slice of an 8-way cut of C4: pass B 59.0 -> 63.9 us.
  //  A task's time is its header's dependent round trips (ticket -> row ids -> row bounds -> first loads), not its rows.)
  const size_t bin_stride_dw = (size_t)kStreamGrid * cap_blocks * B::gdwords;
  uint32_t* wg_regions = regions + (size_t)blockIdx.x * cap_blocks * B::gdwords;

  StreamProf prof;
  // (Round 5, measured and not kept: the task headers — ticket -> row id -> row bounds, dependent round trips — issued two tasks
  //  ahead of the appends: pass B 59.0 -> 61.4 us on one post slice of an 8-way cut of C4, 414 -> 415 us at C4.  The 16 waves of
  //  the workgroup already hide each other's headers.)
  for (;;) {
    prof.stamp(7);
    uint32_t tk = 0;
    if (lane == 0) tk = atomicAdd(s_ticket, 1u);
    tk = rfl(tk);
    const uint64_t task = t_begin + tk;
    if (task >= t_end) break;
    const uint64_t a0 = task << rshift;
    int64_t rb = 0;
    uint64_t len = 0;
    uint32_t rmask = 1u;                              // batch rows in which this lane's row is active
    if ((uint32_t)lane < R && a0 + lane < n_active) {
      const uint32_t r = active[a0 + lane];
      if (BATCH) rmask = row_masks[a0 + lane];
      rb = rp.at(r);
      len = (uint64_t)(rp.at((int64_t)r + 1) - rb);
      n_expected += len * (BATCH ? (uint64_t)__popc(rmask) : 1ull);
    }
    const bool huge = __ballot(len >= (1ull << 26)) != 0;     // a row the 32-bit flattening cannot hold: one row at a time
    const uint32_t n_pieces = huge ? R : 1u;
    for (uint32_t piece = 0; piece < n_pieces; ++piece) {
      uint64_t sub_len = 0;            // huge rows: walked in pieces of 2^26 entries
      int64_t sub_beg = 0;
      if (huge) {
        const uint32_t lo = rl((uint32_t)(len & 0xffffffffull), (int)piece), hi = rl((uint32_t)(len >> 32), (int)piece);
        sub_len = ((uint64_t)hi << 32) | lo;
        const uint32_t blo = rl((uint32_t)((uint64_t)rb & 0xffffffffull), (int)piece), bhi = rl((uint32_t)((uint64_t)rb >> 32), (int)piece);
        sub_beg = (int64_t)(((uint64_t)bhi << 32) | blo);
      }
      for (uint64_t sub = 0; sub == 0 || sub < sub_len; sub += (1ull << 26)) {
        uint32_t T4;                                          // groups of four entries in this (piece of the) task
        if (!huge) {
          const uint32_t g4 = ((uint32_t)len + 3u) >> 2;
          uint32_t incl = g4;
#pragma unroll
          for (int off = 1; off < 64; off <<= 1) {
            const uint32_t t = __shfl_up(incl, off, 64);
            if (lane >= off) incl += t;
          }
          T4 = rl(incl, 63);
          tpre[lane] = incl - g4;
          tlen[lane] = (uint32_t)len;
          tbeg[lane] = rb;
        } else {
          const uint64_t left = sub_len - sub;
          const uint32_t l32 = left < (1ull << 26) ? (uint32_t)left : (1u << 26);
          T4 = (l32 + 3u) >> 2;
          tpre[lane] = lane == 0 ? 0u : T4;
          tlen[lane] = l32;
          tbeg[lane] = sub_beg + (int64_t)sub;
        }
        if (lane == 0) { tpre[64] = T4; tpre[65] = T4; }
        lds_fence();
        const uint32_t n_steps = (T4 + 63u) >> 6;
        if (n_steps == 0) continue;
        const bool use_div = fixed && !huge;
        prof.stamp(0);
        // ---- the stream: U steps loading while the previous U are appended
        uint32_t colN[U][4];
        float wN[U][HOMO ? 1 : 4];
        uint32_t validN[U], maskN[U];
#define BE_STREAM_LOAD_COLS                                                                                            \
        const be_u32x4_a4 c4 = *reinterpret_cast<const be_u32x4_a4*>(indices + at);                                    \
        colN[u][0] = c4.x; colN[u][1] = c4.y; colN[u][2] = c4.z; colN[u][3] = c4.w;
#define BE_STREAM_ISSUE(C0)                                                                                          \
  do {                                                                                                               \
    _Pragma("unroll") for (int u = 0; u < U; ++u) {                                                                  \
      const uint32_t e = ((C0) + (uint32_t)u) * 64u + (uint32_t)lane;                                                \
      const bool ok = e < T4;                                                                                        \
      const uint32_t ee = ok ? e : 0u;                                                                               \
      uint32_t i = 0, off, rlen;                                                                                     \
      if (use_div) {                                                                                                 \
        i = fixdiv.div(ee);                                                                                          \
        off = (ee - i * K4) << 2;                                                                                    \
        rlen = K;                                                                                                    \
      } else {                                                                                                       \
        for (uint32_t s = R >> 1; s > 0; s >>= 1)                                                                    \
          if (tpre[i + s] <= ee) i += s;                                                                             \
        off = (ee - tpre[i]) << 2;                                                                                   \
        rlen = tlen[i];                                                                                              \
      }                                                                                                              \
      const uint32_t rem = rlen - off;                          /* >= 1 */                                           \
      const bool shrt = rlen < 4u;                                                                                   \
      const uint32_t sh = rem >= 4u || shrt ? 0u : 4u - rem;    /* the row's last group: the window moves back */    \
      const int64_t at = tbeg[i] + (int64_t)off - (int64_t)sh;                                                       \
      uint32_t vm = rem >= 4u ? 0xfu : (shrt ? (1u << rem) - 1u : (0xfu << sh) & 0xfu);                              \
      vm = ok ? vm : 0u;                                                                                             \
      if (__ballot(shrt) == 0) {                                                                                     \
        BE_STREAM_LOAD_COLS                                                                                          \
        if (!HOMO) {                                                                                                 \
          if (sizeof(W) == 4) {                                                                                      \
            const be_f32x4_a4 w4 = *reinterpret_cast<const be_f32x4_a4*>(reinterpret_cast<const float*>(weights) + at); \
            wN[u][0] = w4.x; wN[u][HOMO ? 0 : 1] = w4.y; wN[u][HOMO ? 0 : 2] = w4.z; wN[u][HOMO ? 0 : 3] = w4.w;      \
          } else {                                                                                                   \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) wN[u][HOMO ? 0 : j] = (float)WTraits<W>::load(weights, at + j); \
          }                                                                                                          \
        }                                                                                                            \
      } else {                                       /* (rare) rows of fewer than four entries: entry by entry */     \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                              \
          const bool pj = (vm >> j) & 1u;                                                                            \
          colN[u][j] = pj ? (uint32_t)indices[at + j] : 0u;                                                          \
          if (!HOMO) wN[u][HOMO ? 0 : j] = pj ? (float)WTraits<W>::load(weights, at + j) : 0.f;                      \
        }                                                                                                            \
      }                                                                                                              \
      validN[u] = vm;                                                                                                \
      maskN[u] = BATCH ? (huge ? rl(rmask, (int)piece) : (uint32_t)__shfl((int)rmask, (int)i, 64)) : 1u;             \
    }                                                                                                                \
  } while (0)
        BE_STREAM_ISSUE(0u);
        prof.stamp(1);
        for (uint32_t c0 = 0; c0 < n_steps; c0 += U) {
          // the loads of this round have been in flight for a whole round of appends; pin them here so that the wait sits
          // in front of the next round's loads (only one round is ever outstanding: the full wait is the exact one)
          uint32_t colC[U][4];
          float wC[U][HOMO ? 1 : 4];
          uint32_t validC[U], maskC[U];
#pragma unroll
          for (int u = 0; u < U; ++u) {
            maskC[u] = maskN[u];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              colC[u][j] = colN[u][j];
              asm volatile("" : "+v"(colC[u][j]));
              if (!HOMO) { wC[u][HOMO ? 0 : j] = wN[u][HOMO ? 0 : j]; asm volatile("" : "+v"(wC[u][HOMO ? 0 : j])); }
            }
            validC[u] = c0 + (uint32_t)u < n_steps ? validN[u] : 0u;
          }
          prof.stamp(2);
          if (c0 + U < n_steps) BE_STREAM_ISSUE(c0 + U);
          prof.stamp(1);
          {
            uint32_t colA[U * 4], offA[U];
            float wA[HOMO ? 1 : U * 4];
#pragma unroll
            for (int u = 0; u < U; ++u) {
              if (__ballot(validC[u] != 0xfu)) {         // (wave-uniform) lanes without all four entries: sentinel columns
#pragma unroll
                for (int j = 0; j < 4; ++j) colC[u][j] |= ~(uint32_t)__builtin_amdgcn_sbfe((int)validC[u], j, 1);
              }
              if (!HOMO) {
#pragma unroll
                for (int j = 0; j < 4; ++j) wA[HOMO ? 0 : u * 4 + j] = __uint_as_float(__float_as_uint(wC[u][HOMO ? 0 : j]) & sign_mask);
              }
            }
            do {                                          // once; a batch: once per batch row some lane's row is active in
#pragma unroll
              for (int u = 0; u < U; ++u) {
                if constexpr (!BATCH) {
                  maskC[u] = 0u;
                  offA[u] = 0u;
#pragma unroll
                  for (int j = 0; j < 4; ++j) colA[u * 4 + j] = colC[u][j];
                } else {
                  const uint32_t mk = maskC[u];
                  const uint32_t b = mk ? (uint32_t)__ffs(mk) - 1u : 0u;
                  maskC[u] = mk & (mk - 1u);
                  offA[u] = b * (uint32_t)n_bins_b;
#pragma unroll
                  for (int j = 0; j < 4; ++j) colA[u * 4 + j] = mk ? colC[u][j] : 0xffffffffu;
                }
              }
              stream_append<HOMO, CB, U * 4, BATCH>(S, colA, wA, offA, width, wdiv, n_bins, n_bins_b, k, cap_blocks, wl, wg_regions,
                                             bin_stride_dw, out, w0, lane, prof);
              bool again = false;
#pragma unroll
              for (int u = 0; u < U; ++u) again |= maskC[u] != 0u;
              if (__ballot(again) == 0) break;
            } while (true);
          }
          prof.stamp(3);
          prof.count(6, U);
        }
#undef BE_STREAM_ISSUE
        lds_fence();
      }
    }
  }
  prof.stamp(7);
  prof.flush_out(lane);
  n_expected = wave_sum(n_expected);
  if (lane == 0 && n_expected) atomicAdd(&s_audit[0], n_expected);
  __syncthreads();
  // ---- drain: tickets are handed out in order, so of a bin's two ring slots only the one of block T / CB (T = the tickets
  //      drawn) can hold entries now, T % CB of them from slot 0 on; it goes out as it is, and the directory gets T
  for (int q = tid; q < n_bins * B::dwords; q += (int)blockDim.x) {
    const int bin = q / B::dwords, l = q - bin * B::dwords;
    const uint32_t T = S.tick[bin], blk = T / (uint32_t)CB, d = T % (uint32_t)CB;
    if (d > 0 && blk < cap_blocks)
      regions[(((size_t)bin * kStreamGrid + blockIdx.x) * cap_blocks + blk) * B::gdwords + l] = S.buf[(size_t)(bin * kRing + (blk & (uint32_t)(kRing - 1))) * B::dwords + l];
  }
  unsigned long long n_tickets = 0, n_overflow = 0;
  for (int bin = tid; bin < n_bins; bin += (int)blockDim.x) {
    const uint32_t T = S.tick[bin], blk = T / (uint32_t)CB, d = T % (uint32_t)CB;
    n_tickets += T;
    uint32_t o = S.ovf[bin];
    if (d > 0 && blk >= cap_blocks) {          // a partly filled block of a full region: float atomics
      const uint32_t* bp = S.buf + (size_t)(bin * kRing + (blk & (uint32_t)(kRing - 1))) * B::dwords;
      const uint16_t* bi = reinterpret_cast<const uint16_t*>(bp);
      const float* bw = reinterpret_cast<const float*>(bp);
      const uint32_t bb = (uint32_t)bin / (uint32_t)n_bins_b;
      float* dst = out + (int64_t)bb * k + (int64_t)((uint32_t)bin - bb * (uint32_t)n_bins_b) * width;
      for (uint32_t j = 0; j < d; ++j) {
        const uint32_t c = bi[B::col_hw(j)];
        if (c < width) atomicAdd(dst + c, HOMO ? w0 : bw[B::w_dw(j)]);
      }
      o = 1u;
    }
    const uint64_t room = (uint64_t)cap_blocks * CB;
    n_overflow += T > room ? T - room : 0ull;
    dir[(size_t)bin * kStreamGrid + blockIdx.x] = (uint32_t)(T < room ? T : room) | (o ? 0x80000000u : 0u);
  }
  n_tickets = wave_sum(n_tickets);
  n_overflow = wave_sum(n_overflow);
  if (lane == 0 && n_tickets) atomicAdd(&s_audit[1], n_tickets);
  if (lane == 0 && n_overflow) atomicAdd(&s_audit[2], n_overflow);
  __syncthreads();
  if (tid < 3 && s_audit[tid]) audit[3 * blockIdx.x + tid] += s_audit[tid];      // this workgroup's own words (launches are stream-ordered)
}

// eight counted entries (uint16 columns, two per dword)
__device__ __forceinline__ void bin_count8(uint32_t* acc, uint4 v) {
  atomicAdd(&acc[v.x & 0xffffu], 1u); atomicAdd(&acc[v.x >> 16], 1u);
  atomicAdd(&acc[v.y & 0xffffu], 1u); atomicAdd(&acc[v.y >> 16], 1u);
  atomicAdd(&acc[v.z & 0xffffu], 1u); atomicAdd(&acc[v.z >> 16], 1u);
  atomicAdd(&acc[v.w & 0xffffu], 1u); atomicAdd(&acc[v.w >> 16], 1u);
}

// ACC32 (per-entry weights only): the bin's sums are 32-bit fixed point at 2^scale_exp32 instead of 64-bit — 40000 accumulators
// fit one workgroup's LDS instead of 20000, so 10M outputs are 256 bins (one round of pass C, blocks of 32 entries in pass B)
// instead of 611 (2.4 rounds, blocks of 16).  Chosen by the host only when every column's largest weight keeps >= 18 bits at
// that exponent (be_fixed_point_exponent with min_weight_bits = 18 + 32): every addend is then rounded by at most 2^-19 of its
// column's largest weight — 4 x finer than the gate of the 64-bit sums asks for — and the sum is still an integer sum: order
// independent, bitwise reproducible.  `scale` is then 2^scale_exp32 and inv_scale 2^-scale_exp32.
template <bool HOMO, int CAP, bool ACC32 = false>
__global__ void __launch_bounds__(1024) k_bin_accumulate(const uint32_t* __restrict__ regions, const uint32_t* __restrict__ dir,
                                                         uint32_t cap_blocks, int width, int map_cap, int parts, int64_t k, float scale,
                                                         double inv_scale, const void* __restrict__ w0p, int wdtype,
                                                         float* __restrict__ out, float* __restrict__ ovf_img,
                                                         uint32_t* __restrict__ count_rearm, int n_bins_b,
                                                         const uint32_t* __restrict__ err_flag, unsigned long long* __restrict__ audit) {
  using B = BinBlock<HOMO, CAP>;
  using acc_t = typename PlanAcc<HOMO || ACC32>::type;
  static_assert(!(HOMO && ACC32), "ACC32 is a mode of per-entry weights");
  extern __shared__ __align__(16) unsigned char smem_raw[];
  __shared__ uint32_t s_cnt[kStreamGrid], s_pre[kStreamGrid + 1], s_wtot[16];
  __shared__ unsigned long long s_added;
  // block -> region of the bin, one byte per block, in the LDS the accumulators leave: a group then finds its region with one
  // LDS read instead of an 8-step binary search over the prefix sums (bins of more than map_cap blocks search).
  // LAYOUT: [static: directory, prefix sums][s_map: map_cap bytes][accumulators] — the accumulators come LAST, so that a 16-bit
  // local column >= width (which pass B never writes: lc = col - floor(col / width) * width; only a region read past what was
  // stored could hold one) indexes past the END of the workgroup's LDS, where the hardware drops the access, instead of into
  // s_map.  Round 5's -DBE_DBG_LEVEL=3 timing build (tickets real, block stores compiled out) made pass C add at never-written
  // columns; with the map BEHIND the accumulators those adds rewrote map bytes, a later group read a wrong region number r,
  // lb = blk - s_pre[r] wrapped, and the 12-byte unit load left the workspace by ~2^32 units: the memory access fault of
  // gpurun_out/prof_abl.log.  (tests/test_plan_contracts_gpu.py plants such a column: be_internal_binned_poison_next.)
  uint8_t* s_map = smem_raw;
  acc_t* acc = reinterpret_cast<acc_t*>(smem_raw + map_cap);          // map_cap is a multiple of 16 (binned_geometry)
  const int S = width;
  const int bin = blockIdx.x / parts, part = blockIdx.x - bin * parts;
  const int tid = threadIdx.x;
  // the bin's directory: entries per region -> blocks per region -> prefix sums
  uint32_t raw = 0;
  if (tid < kStreamGrid) raw = dir[(size_t)bin * kStreamGrid + tid];
  const uint32_t cnt_r = raw & 0x7fffffffu;
  const uint32_t nb_r = (cnt_r + (uint32_t)CAP - 1u) / (uint32_t)CAP;
  const uint32_t incl = block_scan_1024(nb_r, s_wtot);
  if (tid < kStreamGrid) { s_cnt[tid] = cnt_r; s_pre[tid + 1] = incl; }
  if (tid == 0) { s_pre[0] = 0u; s_added = 0ull; }
  const bool overflowed = __syncthreads_or((int)(raw >> 31)) != 0;     // some entries of this bin went to the overflow image
  if (blockIdx.x == 0 && tid == 0 && count_rearm) count_rearm[0] = 0u;   // the spike counter of the next call's compaction
  const uint32_t NB = s_pre[kStreamGrid];
  float w0 = 0.f;
  if (HOMO) {
    if (wdtype == BE_F16) w0 = __half2float(static_cast<const __half*>(w0p)[0]);
    else if (wdtype == BE_BF16) w0 = __bfloat162float(static_cast<const __hip_bfloat16*>(w0p)[0]);
    else w0 = static_cast<const float*>(w0p)[0];
  }
  if (NB == 0 && parts > 1 && !(overflowed && part == 0)) return;   // nothing to add (several parts per bin: `out` was zeroed up front)
  for (int i = tid; i < S; i += 1024) acc[i] = 0;
  const bool mapped = NB <= (uint32_t)map_cap;
  if (mapped && tid < kStreamGrid)
    for (uint32_t bk = s_pre[tid]; bk < s_pre[tid + 1]; ++bk) s_map[bk] = (uint8_t)tid;
  __syncthreads();
  const uint32_t* bin_base = regions + (size_t)bin * kStreamGrid * cap_blocks * B::gdwords;
  uint32_t n_added = 0;                 // entries this thread adds to the accumulators (conservation counter [2])
  if (HOMO) {
    // one flat loop over the groups of 8 columns (16 bytes) of all regions, BE_BIN_U groups per thread and round with every
    // load of the round issued before the first add
    const uint32_t n8 = NB * (uint32_t)B::groups;
    const uint32_t per = (n8 + parts - 1) / parts;
    const uint32_t g_begin = part * per, g_end = g_begin + per < n8 ? g_begin + per : n8;
    constexpr int U = BE_BIN_U;
    for (uint32_t g0 = g_begin + tid; g0 < g_end; g0 += U * 1024) {
      uint4 iv[U];
      uint32_t nv[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint32_t g = g0 + (uint32_t)u * 1024u;
        const bool in = g < g_end;
        const uint32_t gg = in ? g : g_begin;
        const uint32_t blk = gg / (uint32_t)B::groups, sub = gg - blk * (uint32_t)B::groups;
        uint32_t r = 0;
        if (mapped) {
          r = s_map[blk];
        } else {
#pragma unroll
          for (uint32_t s = kStreamGrid / 2; s > 0; s >>= 1)
            if (s_pre[r + s] <= blk) r += s;
        }
        const uint32_t lb = blk - s_pre[r];
        const uint32_t first = lb * (uint32_t)CAP + sub * 8u, c = s_cnt[r];
        nv[u] = !in || c <= first ? 0u : (c - first < 8u ? c - first : 8u);
        iv[u] = *reinterpret_cast<const uint4*>(bin_base + ((size_t)r * cap_blocks + lb) * B::gdwords + sub * 4u);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        n_added += nv[u];
        if (nv[u] == 8u) {
          bin_count8(reinterpret_cast<uint32_t*>(acc), iv[u]);
        } else if (nv[u]) {                  // the last group of a region
          const uint32_t ix[4] = {iv[u].x, iv[u].y, iv[u].z, iv[u].w};
#pragma unroll
          for (uint32_t j = 0; j < 8; ++j)
            if (j < nv[u]) atomicAdd(reinterpret_cast<uint32_t*>(acc) + ((ix[j >> 1] >> (16 * (j & 1))) & 0xffffu), 1u);
        }
      }
    }
  } else {
    // weighted: one flat loop over the 12-byte units (two entries) of all regions; consecutive lanes read consecutive units —
    // 768 contiguous bytes per load instruction — 2 * BE_BIN_U units per thread and round in flight
    constexpr uint32_t UB = CAP / 2;                         // units per block
    const uint32_t n_u = NB * UB;
    const uint32_t per = (n_u + parts - 1) / parts;
    const uint32_t g_begin = part * per, g_end = g_begin + per < n_u ? g_begin + per : n_u;
    constexpr int U = 2 * BE_BIN_U;
    for (uint32_t g0 = g_begin + tid; g0 < g_end; g0 += U * 1024) {
      be_u32x3_a4 uv[U];
      uint32_t nv[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint32_t g = g0 + (uint32_t)u * 1024u;
        const bool in = g < g_end;
        const uint32_t gg = in ? g : g_begin;
        const uint32_t blk = gg / UB, ub = gg - blk * UB;
        uint32_t r = 0;
        if (mapped) {
          r = s_map[blk];
        } else {
#pragma unroll
          for (uint32_t s = kStreamGrid / 2; s > 0; s >>= 1)
            if (s_pre[r + s] <= blk) r += s;
        }
        const uint32_t lb = blk - s_pre[r];
        const uint32_t first = lb * (uint32_t)CAP + ub * 2u, c = s_cnt[r];
        nv[u] = !in || c <= first ? 0u : (c - first < 2u ? 1u : 2u);
        uv[u] = *reinterpret_cast<const be_u32x3_a4*>(bin_base + ((size_t)r * cap_blocks + lb) * B::gdwords + ub * 3u);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        n_added += nv[u];
        unsigned long long* a64 = reinterpret_cast<unsigned long long*>(acc);
        if (ACC32) {
          uint32_t* a32 = reinterpret_cast<uint32_t*>(acc);
          if (nv[u] >= 1u) atomicAdd(a32 + (uv[u].z & 0xffffu), (uint32_t)__float2int_rn(__uint_as_float(uv[u].x) * scale));
          if (nv[u] == 2u) atomicAdd(a32 + (uv[u].z >> 16), (uint32_t)__float2int_rn(__uint_as_float(uv[u].y) * scale));
        } else {
          if (nv[u] >= 1u) atomicAdd(a64 + (uv[u].z & 0xffffu), fixed_from_f32(__uint_as_float(uv[u].x), scale));
          if (nv[u] == 2u) atomicAdd(a64 + (uv[u].z >> 16), fixed_from_f32(__uint_as_float(uv[u].y), scale));
        }
      }
    }
  }
  {
    const unsigned long long na = wave_sum((unsigned long long)n_added);
    if ((tid & 63) == 0 && na) atomicAdd(&s_added, na);
  }
  __syncthreads();
  if (tid == 0 && s_added && blockIdx.x < (unsigned)kBinAuditGridC) audit[3 * kStreamGrid + blockIdx.x] += s_added;
  // a batch: virtual bin = batch row * n_bins_b + bin; out / ovf_img are [batch row][k]
  const int64_t j0 = (int64_t)(bin / n_bins_b) * k + (int64_t)(bin % n_bins_b) * width;
  const int64_t j_end = (int64_t)(bin / n_bins_b) * k + k;
  // parts == 1: this workgroup is the only writer of its slice, so a plain store does — every output of the slice is
  // written, `out` needs no zeroing.  What pass B could not place in a region sits in the overflow image (all zeros
  // otherwise: read and cleared here only when the bin's flag says so).
  const bool plain = parts == 1;
  const bool poisoned = *err_flag != 0u;             // pass B gave up on an entry (kWaitLimitTicks): no output of this step is trusted
  for (int i = tid; i < S; i += 1024) {
    if (j0 + i >= j_end) break;
    float v;
    if (HOMO) v = (float)reinterpret_cast<uint32_t*>(acc)[i] * w0;
    else if (ACC32) v = (float)((double)(int)reinterpret_cast<uint32_t*>(acc)[i] * inv_scale);
    else v = (float)((double)(long long)reinterpret_cast<unsigned long long*>(acc)[i] * inv_scale);
    if (poisoned) v = __int_as_float(0x7fc00000);
    if (overflowed && part == 0) {
      const float o = ovf_img[j0 + i];
      if (o != 0.f) { v += o; ovf_img[j0 + i] = 0.f; }
    }
    if (plain) out[j0 + i] = v;
    else if (v != 0.f) atomicAdd(out + j0 + i, v);     // contiguous float atomics: the parts of a bin merge here
  }
}

// TEST HOOK (not in the public header; tests/test_plan_contracts_gpu.py): the next binned step has the local column of entry 0 of
// block 0 of region (bin, workgroup) overwritten between pass B and pass C — what a block read past its stored entries could hold.
__global__ void k_bin_poison(uint32_t* __restrict__ block, int col_halfword, uint32_t column) {
  reinterpret_cast<uint16_t*>(block)[col_halfword] = (uint16_t)column;
}
static std::atomic<uint64_t> g_poison_next{0};      // bit 63: armed; [47:32] column; [31:16] bin; [15:0] region

// f16 / bf16 outputs: the bins accumulate into an f32 image of the output (the overflow path adds f32 atomically), rounded once
template <typename W>
__global__ void __launch_bounds__(256) k_bin_round(const float* __restrict__ src, W* __restrict__ dst, int64_t k) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < k; i += stride) WTraits<W>::store(dst, i, src[i]);
}

// A batch of spike vectors [n_batch, m]: the rows with a spike in any of the batch rows [b0, b0 + nb) (nb <= 32) and, per
// listed row, the mask of those batch rows (bit j = batch row b0 + j).  Same block-aggregated reservation as k_compact_spikes.
template <int SD /* BE_SPIKE_BOOL / BE_SPIKE_FLOAT / BE_SPIKE_BITS */>
__global__ void __launch_bounds__(256) k_bin_union(const void* __restrict__ spikes_bm, int64_t m, int64_t row_stride, int nb,
                                                   uint32_t* __restrict__ active, uint32_t* __restrict__ masks,
                                                   uint32_t* __restrict__ count) {
  __shared__ uint32_t wave_tot[4];
  __shared__ uint32_t block_base;
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  uint32_t mk = 0;
  if (r < m) {
    for (int b = 0; b < nb; ++b) {
      bool on;
      if (SD == BE_SPIKE_BITS) on = (static_cast<const uint32_t*>(spikes_bm)[(int64_t)b * row_stride + (r >> 5)] >> (r & 31)) & 1u;
      else if (SD == BE_SPIKE_FLOAT) on = static_cast<const float*>(spikes_bm)[(int64_t)b * row_stride + r] > 0.f;
      else on = static_cast<const uint8_t*>(spikes_bm)[(int64_t)b * row_stride + r] != 0;
      mk |= (on ? 1u : 0u) << b;
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint64_t bal = __ballot(mk != 0u);
  const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
  if (lane == 0) wave_tot[wave] = (uint32_t)__popcll((unsigned long long)bal);
  __syncthreads();
  uint32_t off = 0, total = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    if (w < wave) off += wave_tot[w];
    total += wave_tot[w];
  }
  if (threadIdx.x == 0) block_base = total ? atomicAdd(count, total) : 0u;
  __syncthreads();
  if (mk) {
    const uint32_t at = block_base + off + rank;
    active[at] = (uint32_t)r;
    masks[at] = mk;
  }
}

// `kind` of a binned step (the `homo` argument of the entry points): 0 = per-entry weights, 64-bit sums; 1 = one shared weight
// (counts); 2 = per-entry weights, 32-bit sums (BE_BINNED_ACC32: twice the bin width).  Entries are 2 bytes (counted) or 6.
static inline bool kind_counted(int kind) { return kind == 1; }
static inline int64_t kind_acc_bytes(int kind) { return kind == 0 ? 8 : 4; }

// entries per write-combining block: the largest the LDS of pass B holds for this many bins
static inline int stream_cap(int n_bins, int homo) {
  const int64_t budget = 160 * 1024 - 512 - (int64_t)kStreamFixedWords * 4;
  static const int forced = [] { const char* e = getenv("BE_BIN_CAP"); return e ? atoi(e) : 0; }();       // (A/B runs)
  // (weighted blocks of 128 entries — 768 bytes — serve outputs of fewer than ~85 bins: 10M rows x 1.25M outputs, 125 per row,
  //  8 % firing, 77 bins: pass B 735 / 567 / 519 / 438 us with blocks of 16 / 32 / 64 / 128, tools/exp_hybrid_estimate.py)
  for (int cap = forced > 0 ? forced : 128; cap >= 8; cap >>= 1) {
    const int64_t per_bin = kRing * (int64_t)cap * (kind_counted(homo) ? 2 : 6) + 8 + 8 * kRing;      // two blocks + ticket, 2 commit counts, 2 generations, flag
    if (per_bin * n_bins <= budget) return cap;
  }
  return 0;
}
static inline int64_t stream_cap_blocks(int64_t bin_capacity, int cap) {
  const int64_t per_region = bin_capacity / kStreamGrid * 5 / 4 + 64;       // the rows are dealt evenly, the columns are not
  return (per_region + cap - 1) / cap + 2;
}

// The bins.  Pass C runs one workgroup per bin, every bin the same work, 256 CUs: a bin count just above a multiple of 256
// leaves most of the chip idle for a whole round (C4 weighted: 611 bins of 2^14 columns = 2.39 rounds, 190 us; 512 bins of
// 19532 columns = 2 rounds).  So the output is cut into a multiple of 256 bins of equal width, as few as the accumulators
// of one bin fit LDS (`slice_shift` caps the width at 2^slice_shift columns; local columns are uint16); outputs of fewer
// than 256 x 256 columns get bins of 256 columns.
constexpr int64_t kAccStaticBytes = 4 * (kStreamGrid + kStreamGrid + 1 + 16) + 512;       // pass C: directory, prefix sums, scan, margin
struct BinGeo { int64_t width; int n_bins, cap, map_cap; };
static inline BinGeo binned_geometry(int64_t k, int slice_shift, int homo /* kind */) {
  const int64_t acc_bytes = kind_acc_bytes(homo);
  int64_t max_w = (160 * 1024 - kAccStaticBytes) / acc_bytes;
  max_w = std::min<int64_t>(std::min<int64_t>(max_w, 1ll << slice_shift), 65535) & ~3ll;
  BinGeo g{0, 0, 0, 0};
  static const int forced = [] { const char* e = getenv("BE_BIN_COUNT"); return e ? atoi(e) : 0; }();     // (A/B runs)
  if (forced > 0 && (k + forced - 1) / forced <= max_w) {
    g.width = ((k + forced - 1) / forced + 3) & ~3ll;
  } else if (k <= 256 * 256) {
    g.width = std::min<int64_t>(256, max_w);
  } else if (homo == 0 && k > 256 * std::min<int64_t>(max_w, 16384)) {
    // weighted entries over many outputs keep bins of a power of two (2^14 columns at most: C4 = 611 bins): pass C's rounds
    // were measured not to matter there (512 bins of 19532: 199 us, 611 of 16384: 190 us — it is bound by its byte stream
    // at ~3.4 TB/s, with or without the LDS atomics), and pass B runs 7 % faster without the division per entry
    int sh = 14;
    while ((1ll << sh) > max_w) --sh;
    g.width = 1ll << sh;
  } else {
    const int64_t rounds = (k + 256 * max_w - 1) / (256 * max_w);
    g.width = ((k + 256 * rounds - 1) / (256 * rounds) + 3) & ~3ll;
  }
  const int64_t nb = (k + g.width - 1) / g.width;
  if (nb > kMaxBins) return g;
  g.n_bins = (int)nb;
  g.cap = stream_cap(g.n_bins, homo);
  g.map_cap = (int)(160 * 1024 - kAccStaticBytes - ((g.width * acc_bytes + 15) & ~15ll));
  g.map_cap = g.map_cap < 0 ? 0 : (g.map_cap & ~15);          // the accumulators start behind the map: 16-byte aligned
  return g;
}

// pass B's task size (be_binned_set_tuning; the environment — BE_BIN_TASK_GROUPS / BE_BIN_TASKS — gives the initial values for A/B runs)
static int env_int(const char* name, int dflt) { const char* e = getenv(name); const int v = e ? atoi(e) : 0; return v > 0 ? v : dflt; }
std::atomic<int> g_task_groups{env_int("BE_BIN_TASK_GROUPS", 256)}, g_min_tasks{env_int("BE_BIN_TASKS", 2048)};

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
int be_binned_set_tuning(int task_groups, int min_tasks) {
  BE_REQUIRE(task_groups >= 1 && min_tasks >= 1, BE_ERR_INVALID, "task_groups and min_tasks must be >= 1");
  g_task_groups.store(task_groups, std::memory_order_relaxed);
  g_min_tasks.store(min_tasks, std::memory_order_relaxed);
  return BE_OK;
}

// bins the route cuts k outputs into for this slice_shift (see binned_geometry); 0: not served (too many bins for pass B's LDS)
int be_binned_bins(int64_t k, int slice_shift, int homo) {
  if (k <= 0 || slice_shift < 4 || slice_shift > 16 || homo < 0 || homo > 2) return 0;
  const BinGeo g = binned_geometry(k, slice_shift, homo);
  return g.cap > 0 ? g.n_bins : 0;
}

}  // extern "C"

namespace {
// A batch reads the rows once for `gb` batch rows at a time: bins as wide as pass C's accumulators allow (few bins per batch
// row), gb batch rows' bins side by side as long as pass B's LDS holds blocks of 16 (weighted) / 32 (counted) entries for
// all of them.  gb == 1: the single-vector geometry, one pass per batch row.
struct BatchGeo { BinGeo g; int gb; };
static inline BatchGeo binned_geometry_batch(int64_t k, int slice_shift, int homo, int64_t n_batch) {
  BatchGeo r{binned_geometry(k, slice_shift, homo), 1};
  if (n_batch <= 1) return r;
  const int64_t acc_bytes = kind_acc_bytes(homo);
  int64_t max_w = (160 * 1024 - kAccStaticBytes) / acc_bytes;
  max_w = std::min<int64_t>(std::min<int64_t>(max_w, 1ll << slice_shift), 65535) & ~3ll;
  const int64_t nbb = (k + max_w - 1) / max_w;
  const int64_t width = ((k + nbb - 1) / nbb + 3) & ~3ll;
  int gb = (int)std::min<int64_t>(n_batch, 32);
  while (gb > 1 && (nbb * gb > kMaxBins || stream_cap((int)(nbb * gb), homo) < (kind_counted(homo) ? 32 : 16))) --gb;
  if (gb <= 1) return r;
  r.gb = gb;
  r.g.width = width;
  r.g.n_bins = (int)nbb;                    // bins per batch row; pass B sees n_bins * gb virtual bins
  r.g.cap = stream_cap((int)(nbb * gb), homo);
  r.g.map_cap = (int)(160 * 1024 - kAccStaticBytes - ((width * acc_bytes + 15) & ~15ll));
  r.g.map_cap = r.g.map_cap < 0 ? 0 : (r.g.map_cap & ~15);
  return r;
}
// entries one bin of this geometry receives when one bin of the single-vector geometry receives bin_capacity
static inline int64_t batch_bin_capacity(int64_t k, int slice_shift, int homo, int64_t bin_capacity, const BatchGeo& bg) {
  if (bg.gb <= 1) return bin_capacity;
  const BinGeo g1 = binned_geometry(k, slice_shift, homo);
  const int64_t c = (int64_t)((double)bin_capacity * (g1.n_bins > 0 ? g1.n_bins : 1) / (bg.g.n_bins > 0 ? bg.g.n_bins : 1)) + 64;
  return c < (1ll << 31) ? c : (1ll << 31) - 1;
}
struct BinWs { int64_t active_off, masks_off, dir_off, regions_off, ovf_off, out32_off, total; };
static inline BinWs binned_ws_layout(int64_t m, int64_t k, int64_t n_batch, int slice_shift, int64_t bin_capacity) {
  // sized for the larger of the two entry kinds: a workspace serves one weight or per-entry weights
  int64_t blocks_bytes = 0, dir_bytes = 0, gb_max = 1;
  for (int homo = 0; homo < 3; ++homo) {          // (the three kinds)
    const BatchGeo bg = binned_geometry_batch(k, slice_shift, homo, n_batch);
    if (bg.g.cap == 0) continue;
    const int64_t vb = (int64_t)bg.g.n_bins * bg.gb;
    const int64_t cb = stream_cap_blocks(batch_bin_capacity(k, slice_shift, homo, bin_capacity, bg), bg.g.cap);
    const int64_t b = vb * kStreamGrid * cb * bg.g.cap * (kind_counted(homo) ? 2 : 6);
    blocks_bytes = b > blocks_bytes ? b : blocks_bytes;
    dir_bytes = std::max<int64_t>(dir_bytes, vb * kStreamGrid * 4);
    gb_max = std::max<int64_t>(gb_max, bg.gb);
  }
  BinWs w;
  w.active_off = kBinAuditOff + be_align_up(kBinAuditBytes, 256);
  w.masks_off = w.active_off + be_align_up(m * 4, 256);
  w.dir_off = w.masks_off + (n_batch > 1 ? be_align_up(m * 4, 256) : 0);
  w.regions_off = w.dir_off + be_align_up(dir_bytes, 256);
  w.ovf_off = w.regions_off + be_align_up(blocks_bytes, 256);              // overflow image: [gb][k] f32
  w.out32_off = w.ovf_off + be_align_up(gb_max * k * 4, 256);             // f32 image of an f16 / bf16 output: [n_batch][k]
  w.total = w.out32_off + be_align_up(n_batch * k * 4, 256);
  return w;
}
}  // namespace

extern "C" {

int64_t be_binary_csrmm_t_binned_workspace_bytes(int64_t m, int64_t k, int64_t n_batch, int slice_shift, int64_t bin_capacity) {
  return binned_ws_layout(m, k, n_batch < 1 ? 1 : n_batch, slice_shift, bin_capacity).total;
}
int64_t be_binary_csrmv_t_binned_workspace_bytes(int64_t m, int64_t k, int slice_shift, int64_t bin_capacity) {
  return be_binary_csrmm_t_binned_workspace_bytes(m, k, 1, slice_shift, bin_capacity);
}

// Once per workspace, before its first step: the spike counter and the overflow image start at zero (every step leaves them so).
int be_binary_csrmm_t_binned_workspace_init(void* workspace, int64_t workspace_bytes, int64_t m, int64_t k, int64_t n_batch,
                                            int slice_shift, int64_t bin_capacity, be_stream_t stream) {
  const BinWs w = binned_ws_layout(m, k, n_batch < 1 ? 1 : n_batch, slice_shift, bin_capacity);
  BE_REQUIRE(workspace != nullptr && workspace_bytes >= w.total, BE_ERR_WORKSPACE, "workspace too small");
  hipStream_t st = static_cast<hipStream_t>(stream);
  unsigned char* wsb = static_cast<unsigned char*>(workspace);
  BE_HIP(be_fill_async(wsb, 0, (size_t)w.active_off, st));          // head (spike counter, sticky flag) + conservation counters
  BE_HIP(be_fill_async(wsb + w.ovf_off, 0, (size_t)(w.out32_off - w.ovf_off), st));
  return BE_OK;
}
int be_binary_csrmv_t_binned_workspace_init(void* workspace, int64_t workspace_bytes, int64_t m, int64_t k, int slice_shift,
                                            int64_t bin_capacity, be_stream_t stream) {
  return be_binary_csrmm_t_binned_workspace_init(workspace, workspace_bytes, m, k, 1, slice_shift, bin_capacity, stream);
}

static int read_audit(const void* workspace, uint64_t c[4], uint32_t* flag, hipStream_t st) {
  static_assert(kBinAuditOff % 8 == 0, "the counters are 8-byte words");
  std::vector<unsigned long long> h((size_t)(kBinAuditOff + kBinAuditBytes) / 8);
  BE_HIP(hipMemcpyAsync(h.data(), workspace, h.size() * 8, hipMemcpyDeviceToHost, st));
  BE_HIP(hipStreamSynchronize(st));
  *flag = reinterpret_cast<const uint32_t*>(h.data())[kBinErrWord];
  const unsigned long long* a = h.data() + kBinAuditOff / 8;
  c[0] = c[1] = c[2] = c[3] = 0;
  for (int g = 0; g < kStreamGrid; ++g) { c[0] += a[3 * g]; c[1] += a[3 * g + 1]; c[3] += a[3 * g + 2]; }
  for (int w = 0; w < kBinAuditGridC; ++w) c[2] += a[3 * kStreamGrid + w];
  return BE_OK;
}

int be_binned_workspace_audit(const void* workspace, uint64_t* counters_host, be_stream_t stream) {
  BE_REQUIRE(workspace != nullptr && counters_host != nullptr, BE_ERR_INVALID, "null pointer");
  uint32_t flag = 0;
  return read_audit(workspace, counters_host, &flag, static_cast<hipStream_t>(stream));
}

int be_binned_workspace_status(const void* workspace, int clear, be_stream_t stream) {
  BE_REQUIRE(workspace != nullptr, BE_ERR_INVALID, "workspace is NULL");
  hipStream_t st = static_cast<hipStream_t>(stream);
  uint32_t flag = 0;
  uint64_t c[4];
  const int rc = read_audit(workspace, c, &flag, st);
  if (rc != BE_OK) return rc;
  const bool conserved = c[0] == c[1] && c[1] == c[2] + c[3];
  if (flag == 0u && conserved) return BE_OK;
  if (clear) {
    unsigned char* w = static_cast<unsigned char*>(const_cast<void*>(workspace));
    BE_HIP(be_fill_async(w + kBinErrWord * 4, 0, 4, st));
    BE_HIP(be_fill_async(w + kBinAuditOff, 0, (size_t)kBinAuditBytes, st));
  }
  if (flag != 0u) {
    be_set_error("be_binned_workspace_status: pass B of a binned step gave up on an entry whose ring slot was not freed within 20 ms "
                 "(append protocol stalled); that step's outputs were written as NaN");
    return BE_ERR_HIP;
  }
  char msg[384];
  snprintf(msg, sizeof(msg), "be_binned_workspace_status: conservation broken since the last clear: %llu entries in the active rows, "
           "%llu tickets drawn by pass B, %llu entries accumulated by pass C + %llu through the overflow image (an entry was lost or "
           "delivered twice%s)", (unsigned long long)c[0], (unsigned long long)c[1], (unsigned long long)c[2], (unsigned long long)c[3],
           c[0] > c[1] && c[1] == c[2] + c[3] ? ", or the matrix holds column ids >= k, which are dropped" : "");
  be_set_error(msg);
  return BE_ERR_RANGE;
}

int be_binary_csrmm_t_binned(const void* weights, int homo, int wdtype, const int32_t* indices, const void* indptr,
                             int indptr_is_i64, int64_t row_len, const void* spikes_bm, int spike_dtype, void* out_bm, int64_t m,
                             int64_t k, int64_t n_batch, int slice_shift, int64_t bin_capacity, int scale_exp, void* workspace,
                             int64_t workspace_bytes, be_stream_t stream) {
  BE_REQUIRE(m > 0 && k > 0 && m <= 0xffffffffll && k < (1ll << 31), BE_ERR_INVALID, "bad shape");
  BE_REQUIRE(n_batch >= 1 && n_batch <= kMaxBatch, BE_ERR_INVALID, "bad n_batch");
  BE_REQUIRE(wdtype == BE_F32 || wdtype == BE_F16 || wdtype == BE_BF16, BE_ERR_UNSUPPORTED,
             "the binned route supports f32 / f16 / bf16 weights (its bins carry f32; f64 weights take the planned route)");
  BE_REQUIRE(slice_shift >= 4 && slice_shift <= 16, BE_ERR_INVALID, "slice_shift must be in [4, 16]");
  BE_REQUIRE(check_rows(indptr, row_len), BE_ERR_INVALID, "indptr is NULL and row_len < 0");
  BE_REQUIRE(weights && indices && spikes_bm && out_bm, BE_ERR_INVALID, "null pointer");
  BE_REQUIRE(bin_capacity >= 8 && bin_capacity < (1ll << 32), BE_ERR_INVALID, "bin_capacity out of range");
  // BE_BINNED_ABS (bit 2, per-entry weights): the step sums |w| (pass B clears the sign of every weight it reads) — the column
  // statistics a fixed-point exponent is derived from
  const uint32_t sign_mask = (homo & 4) ? 0x7fffffffu : 0xffffffffu;
  // BE_BINNED_SHORT_ROWS (bit 3, a hint): the stored rows average at most kShortRow entries.  Rows of one length say so
  // themselves (row_len); with an indptr only the caller knows the entry count without a device read.  Pass B then keeps one
  // step of loads in flight per wave instead of two: a task of such rows is 4 steps long, the second step ahead is mostly
  // wasted issue (one post slice of an 8-way cut of C4, rows of ~125: 92.0 -> 84.8 us per step; C4 itself, rows of 1000,
  // loses with it: 0.592 -> 0.619 ms; tools/ab_binned_knobs.sh).  Any value gives the same bits.
  const bool short_rows = indptr == nullptr ? (row_len > 0 && row_len <= kShortRow) : (homo & 8) != 0;
  homo &= ~(4 | 8);
  BE_REQUIRE(homo >= 0 && homo <= 2, BE_ERR_INVALID, "homo must be 0 (per-entry weights), 1 (one weight) or 2 (BE_BINNED_ACC32)");
  const int kind = homo;
  const bool acc32 = kind == 2;
  homo = kind == 1;
  BE_REQUIRE(homo || (acc32 ? (scale_exp > -126 && scale_exp < 127) : (scale_exp - 32 > -126 && scale_exp - 32 < 127)), BE_ERR_INVALID,
             "scale_exp out of range");
  BE_REQUIRE(n_batch == 1 || spike_dtype != BE_SPIKE_IDS, BE_ERR_UNSUPPORTED, "BE_SPIKE_IDS takes a single event vector");
  const BatchGeo bg = binned_geometry_batch(k, slice_shift, kind, n_batch);
  const BinGeo& geo = bg.g;
  const int cap = geo.cap, n_bins_b = geo.n_bins, gb = bg.gb;
  BE_REQUIRE(cap > 0, BE_ERR_RANGE, "too many bins for the LDS blocks of the binned route (be_binned_bins)");
  const size_t lds = (((size_t)geo.width * (size_t)kind_acc_bytes(kind) + 15) & ~(size_t)15) + (size_t)geo.map_cap;
  const int64_t cap_blocks = stream_cap_blocks(batch_bin_capacity(k, slice_shift, kind, bin_capacity, bg), cap);
  BE_REQUIRE(cap_blocks * cap < (1ll << 31), BE_ERR_RANGE, "bin_capacity too large");
  const BinWs wl = binned_ws_layout(m, k, n_batch, slice_shift, bin_capacity);
  BE_REQUIRE(workspace != nullptr && workspace_bytes >= wl.total, BE_ERR_WORKSPACE, "workspace too small");
  hipStream_t st = static_cast<hipStream_t>(stream);
  unsigned char* wsb = static_cast<unsigned char*>(workspace);
  uint32_t* count = reinterpret_cast<uint32_t*>(wsb);
  unsigned long long* audit = reinterpret_cast<unsigned long long*>(wsb + kBinAuditOff);
  uint32_t* active = reinterpret_cast<uint32_t*>(wsb + wl.active_off);
  uint32_t* masks = reinterpret_cast<uint32_t*>(wsb + wl.masks_off);
  uint32_t* dir = reinterpret_cast<uint32_t*>(wsb + wl.dir_off);
  uint32_t* regions = reinterpret_cast<uint32_t*>(wsb + wl.regions_off);
  float* ovf_img = reinterpret_cast<float*>(wsb + wl.ovf_off);
  float* out32 = reinterpret_cast<float*>(wsb + wl.out32_off);
  void* out_user = out_bm;
  float* out = wdtype != BE_F32 ? out32 : static_cast<float*>(out_bm);       // accumulate in f32, round once at the end
  RowPtr rp{indptr, indptr_is_i64, row_len};
  const size_t spk_sz = spike_dtype == BE_SPIKE_FLOAT ? 4 : 1;
  const int64_t row_stride = spike_dtype == BE_SPIKE_BITS ? (m + 31) / 32 : m;            // elements per batch row of spikes
  const DivU32 wdiv = make_div((uint32_t)geo.width);
  // rows of one length: a lane finds its row by dividing its group index by the row's groups of four
  const DivU32 fixdiv = make_div(indptr == nullptr && row_len > 0 && row_len < (1ll << 26) ? (uint32_t)((row_len + 3) / 4) : 1u);
  const float scale = ldexpf(1.0f, acc32 ? scale_exp : scale_exp - 32);      // (ACC32: the multiplier itself; else 2^(e - 32), fixed_from_f32)
  const double inv_scale = ldexp(1.0, -scale_exp);
  const int prof = be_prof_begin(st);
  for (int64_t b0 = 0; b0 < n_batch; b0 += gb) {
    const int nb = (int)std::min<int64_t>(gb, n_batch - b0);
    const int n_vbins = n_bins_b * nb;
    float* out_p = out + b0 * k;
    const unsigned char* spk_p = static_cast<const unsigned char*>(spikes_bm) +
                                 (size_t)b0 * row_stride * (spike_dtype == BE_SPIKE_BITS ? 4 : spk_sz);
    // n_vbins * parts ~ 256: every workgroup of pass C fills a CU; with several parts per bin (fewer than 129 bins) the parts
    // merge their slices into `out` with float atomics, which then has to start at zero
    int parts = 256 / (n_vbins > 0 ? n_vbins : 1);
    parts = parts < 1 ? 1 : (parts > 16 ? 16 : parts);
    if (parts > 1) {
      hipLaunchKernelGGL(k_bin_reset, dim3(grid_for((int64_t)nb * k / 4 + 1, 256, 1024)), dim3(256), 0, st, out_p, (int64_t)nb * k, count);
      BE_LAUNCH_CHECK();
    }
    ActiveList al;
    const uint32_t* row_masks = nullptr;
    if (gb > 1) {               // the rows with a spike in any batch row of this pass + their masks
      const unsigned ug = (unsigned)((m + 255) / 256);
      if (spike_dtype == BE_SPIKE_BITS)
        hipLaunchKernelGGL(k_bin_union<BE_SPIKE_BITS>, dim3(ug), dim3(256), 0, st, spk_p, m, row_stride, nb, active, masks, count);
      else if (spike_dtype == BE_SPIKE_FLOAT)
        hipLaunchKernelGGL(k_bin_union<BE_SPIKE_FLOAT>, dim3(ug), dim3(256), 0, st, spk_p, m, row_stride, nb, active, masks, count);
      else if (spike_dtype == BE_SPIKE_BOOL)
        hipLaunchKernelGGL(k_bin_union<BE_SPIKE_BOOL>, dim3(ug), dim3(256), 0, st, spk_p, m, row_stride, nb, active, masks, count);
      else { be_set_error("be_binary_csrmm_t_binned: unknown spike dtype"); return BE_ERR_INVALID; }
      BE_LAUNCH_CHECK();
      al.ids = active;
      al.count = count;
      row_masks = masks;
    } else {
      int rc = be_resolve_active(spk_p, spike_dtype, m, 1, active, 0, count, st, /*zero_first=*/false, &al);
      if (rc != BE_OK) return rc;
    }
    {
      // groups of four entries per task, and the tasks a step is cut into at least (k_bin_stream: rows per task)
      const uint32_t min_tasks = (uint32_t)g_min_tasks.load(std::memory_order_relaxed);
      const uint32_t task_groups = (uint32_t)g_task_groups.load(std::memory_order_relaxed);
      const size_t dyn = ((size_t)kStreamFixedWords + (size_t)n_vbins * (kRing * (size_t)cap * (homo ? 2 : 6) / 4 + 2 + 2 * kRing)) * 4;
#define BE_BIN_STREAM(WT, HOMO_, CAP_)                                                                                          \
  do {                                                                                                                          \
    auto kern = row_masks ? k_bin_stream<WT, HOMO_, CAP_, true>                                                                  \
                          : (short_rows ? k_bin_stream<WT, HOMO_, CAP_, false, 1> : k_bin_stream<WT, HOMO_, CAP_, false>);       \
    BE_HIP(be_allow_lds(reinterpret_cast<const void*>(kern), (int)dyn));                                                        \
    hipLaunchKernelGGL(kern, dim3(kStreamGrid), dim3(BE_STREAM_THREADS), dyn, st, static_cast<const WT*>(weights), indices, rp, \
                       al.ids, al.count, (uint32_t)geo.width, wdiv, n_vbins, (uint32_t)cap_blocks, regions, dir, ovf_img,       \
                       fixdiv, row_masks, n_bins_b, k, min_tasks, task_groups, m, count + kBinErrWord, sign_mask, audit);       \
  } while (0)
#define BE_BIN_STREAM_W(WT)                                                                                                     \
  do {                                                                                                                          \
    if (homo) {                                                                                                                 \
      if (cap == 128) BE_BIN_STREAM(WT, true, 128); else if (cap == 64) BE_BIN_STREAM(WT, true, 64);                            \
      else if (cap == 32) BE_BIN_STREAM(WT, true, 32); else if (cap == 16) BE_BIN_STREAM(WT, true, 16);                         \
      else BE_BIN_STREAM(WT, true, 8);                                                                                          \
    } else {                                                                                                                    \
      if (cap == 128) BE_BIN_STREAM(WT, false, 128);                                                                            \
      else if (cap == 64) BE_BIN_STREAM(WT, false, 64); else if (cap == 32) BE_BIN_STREAM(WT, false, 32);                       \
      else if (cap == 16) BE_BIN_STREAM(WT, false, 16); else BE_BIN_STREAM(WT, false, 8);                                       \
    }                                                                                                                           \
  } while (0)
      if (wdtype == BE_F16) BE_BIN_STREAM_W(__half); else if (wdtype == BE_BF16) BE_BIN_STREAM_W(__hip_bfloat16); else BE_BIN_STREAM_W(float);
#undef BE_BIN_STREAM_W
#undef BE_BIN_STREAM
    }
    BE_LAUNCH_CHECK();
    const unsigned acc_grid = (unsigned)(n_vbins * parts);
    uint32_t* rearm = spike_dtype == BE_SPIKE_IDS ? static_cast<uint32_t*>(nullptr) : count;
    const uint64_t poison = g_poison_next.load(std::memory_order_relaxed) >> 63 ? g_poison_next.exchange(0) : 0;      // (test hook)
#define BE_BIN_POISON(HOMO_, CAP_)                                                                                              \
  do {                                                                                                                          \
    using PB = BinBlock<HOMO_, CAP_>;                                                                                           \
    const uint32_t pbin = (uint32_t)(poison >> 16) & 0xffffu, preg = (uint32_t)poison & 0xffffu;                                \
    if ((poison >> 63) && pbin < (uint32_t)n_vbins && preg < (uint32_t)kStreamGrid)                                             \
      hipLaunchKernelGGL(k_bin_poison, dim3(1), dim3(1), 0, st,                                                                 \
                         regions + ((size_t)pbin * kStreamGrid + preg) * cap_blocks * PB::gdwords, (HOMO_ ? 0 : 4) /* B::col_hw(0) */, \
                         (uint32_t)(poison >> 32) & 0xffffu);                                                                   \
  } while (0)
#define BE_BIN_ACC32(CAP_)                                                                                                     \
  do {                                                                                                                          \
    BE_BIN_POISON(false, CAP_);                                                                                                 \
    auto kern = k_bin_accumulate<false, CAP_, true>;                                                                            \
    BE_HIP(be_allow_lds(reinterpret_cast<const void*>(kern), (int)lds));                                                        \
    hipLaunchKernelGGL(kern, dim3(acc_grid), dim3(1024), lds, st, regions, dir, (uint32_t)cap_blocks, (int)geo.width,           \
                       geo.map_cap, parts, k, scale, inv_scale, static_cast<const void*>(nullptr), wdtype,                      \
                       out_p, ovf_img, rearm, n_bins_b, count + kBinErrWord, audit);                                            \
  } while (0)
#define BE_BIN_ACC(HOMO_, CAP_)                                                                                                 \
  do {                                                                                                                          \
    BE_BIN_POISON(HOMO_, CAP_);                                                                                                 \
    auto kern = k_bin_accumulate<HOMO_, CAP_>;                                                                                  \
    BE_HIP(be_allow_lds(reinterpret_cast<const void*>(kern), (int)lds));                                                        \
    hipLaunchKernelGGL(kern, dim3(acc_grid), dim3(1024), lds, st, regions, dir, (uint32_t)cap_blocks, (int)geo.width,           \
                       geo.map_cap, parts, k, scale, inv_scale, HOMO_ ? weights : static_cast<const void*>(nullptr), wdtype,    \
                       out_p, ovf_img, rearm, n_bins_b, count + kBinErrWord, audit);                                            \
  } while (0)
    if (homo) {
      if (cap == 128) BE_BIN_ACC(true, 128); else if (cap == 64) BE_BIN_ACC(true, 64); else if (cap == 32) BE_BIN_ACC(true, 32);
      else if (cap == 16) BE_BIN_ACC(true, 16); else BE_BIN_ACC(true, 8);
    } else if (acc32) {
      if (cap == 128) BE_BIN_ACC32(128); else if (cap == 64) BE_BIN_ACC32(64); else if (cap == 32) BE_BIN_ACC32(32);
      else if (cap == 16) BE_BIN_ACC32(16); else BE_BIN_ACC32(8);
    } else {
      if (cap == 128) BE_BIN_ACC(false, 128); else if (cap == 64) BE_BIN_ACC(false, 64); else if (cap == 32) BE_BIN_ACC(false, 32); else if (cap == 16) BE_BIN_ACC(false, 16); else BE_BIN_ACC(false, 8);
    }
#undef BE_BIN_ACC
#undef BE_BIN_ACC32
#undef BE_BIN_POISON
    BE_LAUNCH_CHECK();
  }
  be_prof_end(prof, st);
  const int64_t total = n_batch * k;
  if (wdtype == BE_F16)
    hipLaunchKernelGGL(k_bin_round<__half>, dim3(grid_for(total, 256, 2048)), dim3(256), 0, st, out32, static_cast<__half*>(out_user), total);
  else if (wdtype == BE_BF16)
    hipLaunchKernelGGL(k_bin_round<__hip_bfloat16>, dim3(grid_for(total, 256, 2048)), dim3(256), 0, st, out32,
                       static_cast<__hip_bfloat16*>(out_user), total);
  BE_LAUNCH_CHECK();
  return BE_OK;
}

/* test hook, deliberately absent from include/brainevent_amd.h: arms k_bin_poison for the next binned step of this process */
int be_internal_binned_poison_next(int bin, int region, uint32_t column) {
  BE_REQUIRE(bin >= 0 && bin < 65536 && region >= 0 && region < 65536 && column < 65536u, BE_ERR_INVALID, "out of range");
  g_poison_next.store((1ull << 63) | ((uint64_t)column << 32) | ((uint64_t)bin << 16) | (uint64_t)region);
  return BE_OK;
}

int be_binary_csrmv_t_binned(const void* weights, int homo, int wdtype, const int32_t* indices, const void* indptr,
                             int indptr_is_i64, int64_t row_len, const void* spikes, int spike_dtype, void* out, int64_t m,
                             int64_t k, int slice_shift, int64_t bin_capacity, int scale_exp, void* workspace,
                             int64_t workspace_bytes, be_stream_t stream) {
  return be_binary_csrmm_t_binned(weights, homo, wdtype, indices, indptr, indptr_is_i64, row_len, spikes, spike_dtype, out, m, k, 1,
                                  slice_shift, bin_capacity, scale_exp, workspace, workspace_bytes, stream);
}

}  // extern "C"
