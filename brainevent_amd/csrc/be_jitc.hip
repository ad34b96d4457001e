// be_jitc.hip — just-in-time connectivity products (BinaryArray @ JITC{Scalar,Uniform,Normal}{R,C}) for gfx950.
//
// The connectivity is never stored: it is regenerated from (seed, row, chunk, lane) with the reference's
// "light_rng" sampler, reproduced here bit for bit (uint32 arithmetic):
//   mix32 / xorshift32 / mulhi-bounded / stationary initial q     brainevent/_numba_random.py:385-421, :489-502
//   per-edge weight hashes uniform01 (24-bit) / normal01 (Acklam)  brainevent/_numba_random.py:424-486
//   walk: for (row, chunk_id, lane < stride): q0 = initial_q; local_j = lane + stride*q; q += 1 + bounded(next, cl-1)
//         brainevent/_jit_scalar/binary.py:340-416 (mv, stride 32), :875-949 (mm, stride 4);
//         chunk_size = ceil(shape[1] / 4) always (brainevent/_misc.py:74-122)
//   `corder` alone picks the kernel: True -> gather over output rows ("notrans"), False -> scatter over
//   active input rows ("trans")  (brainevent/_jit_scalar/binary.py:434-439).
// The lane stride (32 / 4) is part of the drawn matrix, so a 64-wide wavefront never walks with stride 64:
// it runs two 32-lane tasks (gather) or 64 independent single-lane walks (scatter).
//
// MI355X mapping:
//   gather  : workgroup = (row block, chunk); the chunk's bit-packed spikes are staged in LDS (<= 150 KB),
//             each half-wave owns one output row (lane = residue class), deterministic per-chunk partials.
//   scatter : outputs are partitioned by (chunk, residue class lane): a walk (row, chunk, lane) only ever
//             touches columns  chunk_start + lane + 32 q, so one workgroup owns the accumulators of one class
//             in LDS (indexed by q) and each of its threads walks one active row at a time — no redundant RNG
//             work, no global atomics (random global f32 atomics retire at ~21 G/s on this chip, LDS integer
//             atomics at > 3 T/s).  Sums are integer counts (scalar weight) or 64-bit fixed point
//             (uniform / normal), hence order independent and bitwise reproducible.
#include "be_jitc_shared.h"
#include <mutex>
#include <unordered_set>
#include <type_traits>

namespace {

// ------------------------------------------------------------------------------------------------ gather (mv)
// partial[chunk * m + row]: uint32 counts (scalar) or double sums (uniform / normal)
template <int MODE> struct GatherAcc { using type = double; };
template <> struct GatherAcc<MODE_SCALAR> { using type = uint32_t; };

// 1024 threads per workgroup: the staged bitmap takes most of the LDS (one workgroup per CU), so the 16 waves of
// that one workgroup are all the latency hiding the CU gets for the dependent LDS lookups of the walk.
template <int MODE, bool BITS_IN_LDS>
__global__ void __launch_bounds__(1024) k_jit_mv_gather(JitP p, const uint32_t* __restrict__ bits, int64_t m,
                                                       typename GatherAcc<MODE>::type* __restrict__ partial) {
  using AccT = typename GatherAcc<MODE>::type;
  extern __shared__ uint32_t bits_s[];
  const int chunk = blockIdx.y;
  const int64_t cs = (int64_t)chunk * p.chunk_size;
  const int64_t ce = cs + p.chunk_size < p.walk_len ? cs + p.chunk_size : p.walk_len;
  const int64_t width = ce - cs;
  const int64_t w_first = cs >> 5;
  const uint32_t* bsrc = bits;
  int64_t w_off = 0;
  if (BITS_IN_LDS) {
    const int64_t n_w = ((ce + 31) >> 5) - w_first;
    for (int64_t i = threadIdx.x; i < n_w; i += blockDim.x) bits_s[i] = bits[w_first + i];
    __syncthreads();
    bsrc = bits_s;
    w_off = w_first;
  }
  const uint32_t l = threadIdx.x & 31u;
  const int64_t half = threadIdx.x >> 5;
  const int64_t halves = blockDim.x >> 5;
  const int64_t rows_per_iter = (int64_t)gridDim.x * halves;
  const int64_t m_round = (m + rows_per_iter - 1) / rows_per_iter * rows_per_iter;
  for (int64_t row = (int64_t)blockIdx.x * halves + half; row < m_round; row += rows_per_iter) {
    AccT acc = AccT(0);
    if (row < m) {
      // the walk in the q domain: lane l visits chunk-local columns l + 32 q, i.e. always bit ((cs + l) & 31) of
      // consecutive 32-bit words of the packed spike vector — one LDS word per step, no 64-bit arithmetic
      const uint32_t grow = (uint32_t)row + p.row0;          // the generator row (the RNG is keyed by it; `row` indexes the output)
      uint32_t state = lr_init(p.seed, grow, (uint32_t)chunk, l);
      uint32_t q = lr_initial_q(state, p.cl);
      const uint32_t qmax = width > (int64_t)l ? (uint32_t)((width - l + 31) >> 5) : 0u;   // l + 32 q < width
      const int64_t bit0 = cs + l;
      const uint32_t sh = (uint32_t)(bit0 & 31);
      const uint32_t* wp = bsrc + ((bit0 >> 5) - w_off);
      while (q < qmax) {
        const bool on = (wp[q] >> sh) & 1u;
        if (on) {
          if (MODE == MODE_SCALAR) acc += 1;
          else acc += (AccT)edge_weight<MODE, float>(p, grow, (uint32_t)(bit0 + 32ll * q));
        }
        state = lr_next_nz(state);
        q = q + 1u + lr_bounded(state, p.cl - 1u);
      }
    }
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) acc += __shfl_down(acc, off, 32);
    if (l == 0 && row < m) partial[(int64_t)chunk * m + row] = acc;
  }
}

template <int MODE, typename W>
__global__ void __launch_bounds__(256) k_jit_gather_reduce(const typename GatherAcc<MODE>::type* __restrict__ partial,
                                                           int n_chunks, int64_t m, double w0, W* __restrict__ out) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < m; r += stride) {
    if (MODE == MODE_SCALAR) {
      uint64_t c = 0;
      for (int ch = 0; ch < n_chunks; ++ch) c += (uint64_t)partial[(int64_t)ch * m + r];
      WTraits<W>::store_d(out, r, (double)c * w0);     // count * weight, one rounding (numba: f64 count * w)
    } else {
      double s = 0.0;
      for (int ch = 0; ch < n_chunks; ++ch) s += (double)partial[(int64_t)ch * m + r];
      WTraits<W>::store_d(out, r, s);
    }
  }
}

// ------------------------------------------------------------------------------------------------ scatter (mv)
// grid.x = n_classes * pieces * parts.  class = chunk * stride + lane residue (stride 32 for the mv matrix, 4 for mm).
// gridDim.y = batch column: active lists / counters / partials of column b live at b * (their stride).
template <int MODE, bool ONE_PIECE>
__global__ void __launch_bounds__(1024) k_jit_mv_scatter(JitP p, const uint32_t* __restrict__ active,
                                                         const uint32_t* __restrict__ n_active_p, int pieces, int parts,
                                                         uint32_t piece_len, float fx_scale,
                                                         typename ScatterAcc<MODE>::type* __restrict__ partial,
                                                         int64_t active_stride) {
  using AccT = typename ScatterAcc<MODE>::type;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  AccT* acc = reinterpret_cast<AccT*>(smem_raw);
  const int part = blockIdx.x % parts;
  const int piece = (blockIdx.x / parts) % pieces;
  const int cls = p.cls_begin + blockIdx.x / (parts * pieces);
  const uint32_t S = (uint32_t)p.stride;
  const uint32_t chunk = (uint32_t)cls / S, l = (uint32_t)cls - chunk * S;
  active += (int64_t)blockIdx.y * active_stride;
  partial += (int64_t)blockIdx.y * gridDim.x * piece_len;
  const int64_t cs = (int64_t)chunk * p.chunk_size;
  const int64_t ce = cs + p.chunk_size < p.walk_len ? cs + p.chunk_size : p.walk_len;
  const int64_t width = ce - cs;
  // positions q with l + stride * q < width
  const int64_t Q = width > (int64_t)l ? (width - l + S - 1) / S : 0;
  const int64_t q_begin = (int64_t)piece * piece_len;
  const int64_t q_end = q_begin + piece_len < Q ? q_begin + piece_len : Q;
  for (uint32_t i = threadIdx.x; i < piece_len; i += blockDim.x) acc[i] = 0;
  __syncthreads();
  const uint32_t n_active = n_active_p[blockIdx.y];
  if (q_begin < q_end) {
    const uint32_t qb = (uint32_t)q_begin, qe = (uint32_t)q_end;      // Q < 2^27: 32-bit walk arithmetic
    const uint32_t j0 = (uint32_t)(cs + l);
    // A wave starts 64 walks together and waits for the longest (31 +- 6 edges per walk at C3: ~70 % of the lanes busy).
    // Letting idle lanes take their next row early was measured in round 3 and is slower, whatever the threshold: the
    // start-up code (lr_init + the rejection loop of the stationary start, ~7 rounds until the last of the refilling lanes
    // accepts) then runs once per 8 ... 32 walks instead of once per 64 — kernel 103 us as it is, 119 / 150 / 177 / 219 us
    // refilling at 64 (= never early) / 32 / 16 / 8 idle lanes, with the next row id prefetched (tools/ab_c3.sh).
    for (uint64_t a = (uint64_t)part * blockDim.x + threadIdx.x; a < n_active; a += (uint64_t)parts * blockDim.x) {
      const uint32_t row = active[a];
      uint32_t state = lr_init(p.seed, row, chunk, l);
      uint32_t q = lr_initial_q(state, p.cl);
      while (q < qe) {
        if (ONE_PIECE || q >= qb) {        // one piece per class: qb == 0, no test in the loop
          const uint32_t slot = ONE_PIECE ? q : q - qb;       // one piece: qb == 0, one vector operation less per edge
          if (MODE == MODE_SCALAR) atomicAdd(&acc[slot], (AccT)1);
          else atomicAdd(&acc[slot], (AccT)jit_fixed_from_f32(edge_weight<MODE, float>(p, row, j0 + S * q), fx_scale));
        }
        state = lr_next_nz(state);
        q = q + 1u + lr_bounded(state, p.cl - 1u);
      }
    }
  }
  __syncthreads();
  AccT* dst = partial + (int64_t)blockIdx.x * piece_len;
  for (uint32_t i = threadIdx.x; i < piece_len; i += blockDim.x) dst[i] = acc[i];
}

// ------------------------------------------------------------------------------------------------ mm (stride 4)
// gather : one thread per generator row; spike matrix as per-column masks (<= 32 batch columns per pass):
//          out_bm[c, row] = sum over edges j of row with bit c of mask[j] set
// scatter: the residue-class kernel above with lane stride 4 and gridDim.y = batch column (jit_scatter_batched)
// Scalar-weight mm gather: the per-column counts of a generator row are kept BIT-SLICED — plane[b] holds bit b of all (<= 32)
// column counters, one column per bit position — so adding an edge's column mask is a ripple-carry add of one word: on average
// two plane updates (and / xor / move) instead of one predicated add per batch column (32 columns: ~100 vector operations per
// generated edge that hits, with some lane of the wave hitting on nearly every edge at 1 % firing).  The carry chain stops as soon
// as no lane carries any more; its length is log2 of the largest count so far.  Counts are exact integers either way.
struct SlicedCounts {
  uint32_t plane[32];
  __device__ __forceinline__ void clear() {
#pragma unroll
    for (int b = 0; b < 32; ++b) plane[b] = 0u;
  }
  __device__ __forceinline__ void add(uint32_t mk) {
    uint32_t carry = mk;
#pragma unroll
    for (int b = 0; b < 32; ++b) {
      if (__ballot(carry != 0u) == 0ull) break;          // (wave-uniform exit)
      const uint32_t t = plane[b] & carry;
      plane[b] ^= carry;
      carry = t;
    }
  }
  __device__ __forceinline__ uint32_t count(int c) const {
    uint32_t n = 0u;
#pragma unroll
    for (int b = 0; b < 32; ++b) n |= ((plane[b] >> c) & 1u) << b;
    return n;
  }
};

template <int MODE, typename A, int NCOL>
__global__ void __launch_bounds__(256) k_jit_mm_gather(JitP p, const uint32_t* __restrict__ mask, int64_t m, int nc,
                                                       A* __restrict__ out_bm) {
  const int64_t stride_t = (int64_t)gridDim.x * blockDim.x;
  for (int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; row < m; row += stride_t) {
    A acc[MODE == MODE_SCALAR ? 1 : NCOL];         // NCOL = 8 / 16 / 32 >= nc: the columns the masks can hold
    SlicedCounts sc;                               // (scalar weights: bit-sliced counts instead)
    if constexpr (MODE == MODE_SCALAR) sc.clear();
    else {
#pragma unroll
      for (int c = 0; c < NCOL; ++c) acc[c] = A(0);
    }
    for (int chunk = 0; chunk < p.n_chunks; ++chunk) {
      const int64_t cs = (int64_t)chunk * p.chunk_size;
      const int64_t ce = cs + p.chunk_size < p.walk_len ? cs + p.chunk_size : p.walk_len;
      const int64_t width = ce - cs;
      for (uint32_t l = 0; l < (uint32_t)p.stride; ++l) {
        uint32_t state = lr_init(p.seed, (uint32_t)row, (uint32_t)chunk, l);
        uint32_t q = lr_initial_q(state, p.cl);
        uint64_t lj = (uint64_t)l + (uint64_t)p.stride * q;
        while ((int64_t)lj < width) {
          const int64_t j = cs + (int64_t)lj;
          const uint32_t mk = mask[j];
          if constexpr (MODE == MODE_SCALAR) {
            sc.add(mk);
          } else if (mk) {
            const A w = edge_weight<MODE, A>(p, (uint32_t)row, (uint32_t)j);
#pragma unroll
            for (int c = 0; c < NCOL; ++c) acc_add_inplace(acc[c], ((mk >> c) & 1u) ? w : A(0));
          }
          state = lr_next_nz(state);
          q = q + 1u + lr_bounded(state, p.cl - 1u);
          lj = (uint64_t)l + (uint64_t)p.stride * q;
        }
      }
    }
#pragma unroll
    for (int c = 0; c < NCOL; ++c)
      if (c < nc) {
        if constexpr (MODE == MODE_SCALAR) out_bm[(int64_t)c * m + row] = (A)((A)sc.count(c) * (A)p.w0);
        else out_bm[(int64_t)c * m + row] = acc[c];
      }
  }
}

// gather with the chunk's column masks staged in LDS (as the mv gather stages its bits): the 32-bit masks are narrowed to
// the batch width (uint8 for <= 8 columns, uint16 for <= 16) so that a whole chunk — a quarter of the input population —
// fits 128 KB: populations up to 512k / 256k / 128k, and four times that with windows (below).  A thread keeps its row's
// accumulators over the four chunks; the workgroup reloads the masks between chunks / windows.  The global-mask kernel above pays a 64-byte sector per generated edge
// (~5 ps per edge whatever the population: n = 250k ... 2M); this one runs at the generator's rate.
template <int MODE, typename A, typename T>
__global__ void __launch_bounds__(1024) k_jit_mm_gather_lds(JitP p, const uint32_t* __restrict__ mask, int64_t m, int nc,
                                                            A* __restrict__ out_bm, int64_t win_cols) {
  // win_cols: columns of a chunk whose masks LDS holds at a time.  A chunk wider than that is swept in windows.
  extern __shared__ __align__(16) unsigned char jit_mm_lds[];
  T* ms = reinterpret_cast<T*>(jit_mm_lds);
  const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  constexpr int NCOL = 8 * (int)sizeof(T);                 // batch columns a mask of type T holds
  A acc[MODE == MODE_SCALAR ? 1 : NCOL];
  SlicedCounts sc;                                         // (scalar weights: bit-sliced counts instead)
  if constexpr (MODE == MODE_SCALAR) sc.clear();
  else {
#pragma unroll
    for (int c = 0; c < NCOL; ++c) acc[c] = A(0);
  }
  constexpr int kMmStride = 4;                               // lane stride of the mm matrix (brainevent/_misc.py:37-38)
  for (int chunk = 0; chunk < p.n_chunks; ++chunk) {
    const int64_t cs = (int64_t)chunk * p.chunk_size;
    const int64_t ce = cs + p.chunk_size < p.walk_len ? cs + p.chunk_size : p.walk_len;
    const int64_t width = ce - cs;
    // the four walks of this row in this chunk keep their generator state across the windows (round 2 restarted every
    // walk at every window: W windows cost (W + 1) / 2 walks of the chunk, which ruled windows out for wide batches)
    uint32_t wstate[kMmStride], wq[kMmStride];
#pragma unroll
    for (int l = 0; l < kMmStride; ++l) {
      wstate[l] = lr_init(p.seed, (uint32_t)row, (uint32_t)chunk, (uint32_t)l);
      wq[l] = lr_initial_q(wstate[l], p.cl);
    }
    for (int64_t w_lo = 0; w_lo < width; w_lo += win_cols) {
      const int64_t w_hi = w_lo + win_cols < width ? w_lo + win_cols : width;
      const int64_t w_n = w_hi - w_lo;
      __syncthreads();                                       // the previous window's masks are no longer read
      for (int64_t i0 = threadIdx.x; i0 < w_n; i0 += 8 * (int64_t)blockDim.x) {      // eight loads in flight per thread
        uint32_t v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int64_t i = i0 + (int64_t)u * blockDim.x;
          v[u] = mask[cs + w_lo + (i < w_n ? i : 0)];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int64_t i = i0 + (int64_t)u * blockDim.x;
          if (i < w_n) ms[i] = (T)v[u];
        }
      }
      __syncthreads();
      if (row < m) {
#pragma unroll
        for (int l = 0; l < kMmStride; ++l) {
          uint32_t state = wstate[l], q = wq[l];
          uint64_t lj = (uint64_t)l + (uint64_t)kMmStride * q;
          while ((int64_t)lj < w_hi) {
            uint32_t mk = (uint32_t)ms[(int64_t)lj - w_lo];
            if constexpr (MODE == MODE_SCALAR) {
              sc.add(mk);
            } else if (mk) {
              asm volatile("" : "+v"(mk));                   // keeps the per-column adds behind the branch
              const A w = edge_weight<MODE, A>(p, (uint32_t)row, (uint32_t)(cs + (int64_t)lj));
#pragma unroll
              for (int c = 0; c < NCOL; ++c) acc_add_inplace(acc[c], ((mk >> c) & 1u) ? w : A(0));
            }
            state = lr_next_nz(state);
            q = q + 1u + lr_bounded(state, p.cl - 1u);
            lj = (uint64_t)l + (uint64_t)kMmStride * q;
          }
          wstate[l] = state;
          wq[l] = q;
        }
      }
    }
  }
  if (row < m) {
#pragma unroll
    for (int c = 0; c < NCOL; ++c)
      if (c < nc) {
        if constexpr (MODE == MODE_SCALAR) out_bm[(int64_t)c * m + row] = (A)((A)sc.count(c) * (A)p.w0);
        else out_bm[(int64_t)c * m + row] = acc[c];
      }
  }
}

// the same masks from a bit-packed batch (nc rows of ceil(len / 32) words): a bit transpose, no unpack
__global__ void __launch_bounds__(256) k_jit_masks_bits(const uint32_t* __restrict__ words, int64_t len, int nc,
                                                        uint32_t* __restrict__ mask) {
  const int64_t n_words = (len + 31) / 32;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < len; i += stride) {
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

template <typename SP>
__global__ void __launch_bounds__(256) k_jit_masks(const typename SP::type* __restrict__ spikes_bm, int64_t len, int nc,
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

template <typename A, typename W>
__global__ void __launch_bounds__(256) k_jit_convert(const A* __restrict__ src, W* __restrict__ dst, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) WTraits<W>::store_d(dst, i, (double)src[i]);
}

// ------------------------------------------------------------------------------------------------ materialise
// Generator matrix -> CSR (rows = walk owners).  One thread per (row, chunk, lane) task: count its edges, then
// reserve a contiguous range of the row with one atomic and re-walk writing (column [, weight]).
// Within a row the order of the tasks is unspecified; columns inside one task are increasing.
__global__ void __launch_bounds__(256) k_jit_csr_count(JitP p, int64_t n_rows, uint32_t* __restrict__ row_counts) {
  const int64_t tasks_per_row = (int64_t)p.n_chunks * p.stride;
  const int64_t n_tasks = n_rows * tasks_per_row;
  const int64_t stride_t = (int64_t)gridDim.x * blockDim.x;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n_tasks; t += stride_t) {
    const int64_t row = t / tasks_per_row;
    const int rem = (int)(t - row * tasks_per_row);
    const int chunk = rem / p.stride;
    const uint32_t l = (uint32_t)(rem - chunk * p.stride);
    const int64_t cs = (int64_t)chunk * p.chunk_size;
    const int64_t ce = cs + p.chunk_size < p.walk_len ? cs + p.chunk_size : p.walk_len;
    const int64_t width = ce - cs;
    uint32_t state = lr_init(p.seed, (uint32_t)row, (uint32_t)chunk, l);
    uint32_t q = lr_initial_q(state, p.cl);
    uint64_t lj = (uint64_t)l + (uint64_t)p.stride * q;
    uint32_t cnt = 0;
    while ((int64_t)lj < width) {
      ++cnt;
      state = lr_next_nz(state);
      q = q + 1u + lr_bounded(state, p.cl - 1u);
      lj = (uint64_t)l + (uint64_t)p.stride * q;
    }
    if (cnt) atomicAdd(&row_counts[row], cnt);
  }
}

template <int MODE>
__global__ void __launch_bounds__(256) k_jit_csr_fill(JitP p, int64_t n_rows, const int64_t* __restrict__ indptr,
                                                      uint32_t* __restrict__ cursor, int32_t* __restrict__ indices,
                                                      float* __restrict__ weights) {
  const int64_t tasks_per_row = (int64_t)p.n_chunks * p.stride;
  const int64_t n_tasks = n_rows * tasks_per_row;
  const int64_t stride_t = (int64_t)gridDim.x * blockDim.x;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n_tasks; t += stride_t) {
    const int64_t row = t / tasks_per_row;
    const int rem = (int)(t - row * tasks_per_row);
    const int chunk = rem / p.stride;
    const uint32_t l = (uint32_t)(rem - chunk * p.stride);
    const int64_t cs = (int64_t)chunk * p.chunk_size;
    const int64_t ce = cs + p.chunk_size < p.walk_len ? cs + p.chunk_size : p.walk_len;
    const int64_t width = ce - cs;
    const uint32_t state0 = lr_init(p.seed, (uint32_t)row, (uint32_t)chunk, l);
    uint32_t state = state0;
    const uint32_t q0 = lr_initial_q(state, p.cl);
    const uint32_t state1 = state;
    uint32_t q = q0, cnt = 0;
    uint64_t lj = (uint64_t)l + (uint64_t)p.stride * q;
    while ((int64_t)lj < width) {
      ++cnt;
      state = lr_next_nz(state);
      q = q + 1u + lr_bounded(state, p.cl - 1u);
      lj = (uint64_t)l + (uint64_t)p.stride * q;
    }
    if (cnt == 0) continue;
    int64_t pos = indptr[row] + atomicAdd(&cursor[row], cnt);
    state = state1; q = q0;
    lj = (uint64_t)l + (uint64_t)p.stride * q;
    while ((int64_t)lj < width) {
      const int64_t j = cs + (int64_t)lj;
      indices[pos] = (int32_t)j;
      if (MODE != MODE_SCALAR) weights[pos] = edge_weight<MODE, float>(p, (uint32_t)row, (uint32_t)j);
      ++pos;
      state = lr_next_nz(state);
      q = q + 1u + lr_bounded(state, p.cl - 1u);
      lj = (uint64_t)l + (uint64_t)p.stride * q;
    }
  }
}

// ------------------------------------------------------------------------------------------------ host
}  // namespace

// compaction kernel of be_csr.hip (spikes -> active ids + count), n_batch = 1
extern "C" int be_compact_spikes(const void* spikes, int spike_dtype, int64_t n, uint32_t* active_ids, uint32_t* count,
                                 be_stream_t stream);
extern "C" int be_pack_spikes(const void* spikes, int spike_dtype, int64_t n, uint32_t* bits, be_stream_t stream);
extern "C" int be_compact_spikes_batched(const void* spikes_bm, int spike_dtype, int64_t n, int64_t n_batch,
                                         uint32_t* active_ids, int64_t active_stride, uint32_t* counts, be_stream_t stream);
extern "C" int be_internal_compact_spikes_batched(const void* spikes_bm, int spike_dtype, int64_t n, int64_t n_batch,
                                                  uint32_t* active_ids, int64_t active_stride, uint32_t* counts, int zero_first,
                                                  be_stream_t stream);

namespace {

// workspaces armed for the scatter orientation (be_jit_scatter_workspace_arm / _disarm): a handful of pointers, looked up once per call
std::mutex g_armed_mu;
std::unordered_set<const void*> g_armed;
inline bool jit_workspace_is_armed(const void* ws) {
  std::lock_guard<std::mutex> lk(g_armed_mu);
  return g_armed.count(ws) != 0;
}

template <int MODE, typename W>
int jit_mv_gather(const JitP& p, const void* spikes, int sd, void* out, int64_t m, void* ws, hipStream_t st) {
  using AccT = typename GatherAcc<MODE>::type;
  unsigned char* wsb = static_cast<unsigned char*>(ws);
  uint32_t* bits = reinterpret_cast<uint32_t*>(wsb);
  const int64_t n_words = (p.walk_len + 31) / 32;
  AccT* partial = reinterpret_cast<AccT*>(wsb + be_align_up((n_words + 2) * 4, 256));
  if (sd == BE_SPIKE_BITS) {      // already the kernels' format (a BitPackedBinary, the words the spike exchange delivers): no pack launch
    bits = const_cast<uint32_t*>(static_cast<const uint32_t*>(spikes));
  } else {
    int rc = be_pack_spikes(spikes, sd, p.walk_len, bits, st);
    if (rc != BE_OK) return rc;
  }
  const size_t lds = (size_t)(((std::min<int64_t>(p.chunk_size, p.walk_len) + 31) / 32) + 2) * 4;
  const dim3 grid(gcap(m, 32, 512), p.n_chunks);
  const int prof = be_prof_begin(st);
  if (lds <= 150 * 1024) {
    auto kern = k_jit_mv_gather<MODE, true>;
    BE_HIP(be_allow_lds(reinterpret_cast<const void*>(kern), (int)lds));
    hipLaunchKernelGGL(kern, grid, dim3(1024), lds, st, p, bits, m, partial);
  } else {
    hipLaunchKernelGGL((k_jit_mv_gather<MODE, false>), grid, dim3(1024), 0, st, p, bits, m, partial);
  }
  be_prof_end(prof, st);
  BE_LAUNCH_CHECK();
  hipLaunchKernelGGL((k_jit_gather_reduce<MODE, W>), dim3(gcap(m, 256, 2048)), dim3(256), 0, st, partial, p.n_chunks, m,
                     p.w0, static_cast<W*>(out));
  BE_LAUNCH_CHECK();
  return BE_OK;
}

inline int64_t jit_counts_bytes(int64_t nb) { return be_align_up(nb * 4, 256); }
inline int64_t jit_active_stride(int64_t m) { return be_align_up(m * 4, 256) / 4; }

inline int64_t jit_scatter_ws_bytes(const JitP& p, int64_t m, int64_t nb) {
  const ScatterGeom g = scatter_geom(p, false, nb), gs = scatter_geom(p, true, nb);
  const int64_t a = (int64_t)g.n_classes * g.pieces * g.parts * g.piece_len * 8;
  const int64_t b = (int64_t)gs.n_classes * gs.pieces * gs.parts * gs.piece_len * 4;
  return jit_counts_bytes(nb) + nb * jit_active_stride(m) * 4 + be_align_up(nb * std::max(a, b), 256);
}

// scatter ("trans" kernel) for a batch: spikes_bm [nb, m] -> out_bm [nb, walk_len]; stride 32 (mv) or 4 (mm)
template <int MODE, typename W>
int jit_scatter_batched(const JitP& p, const void* spikes_bm, int sd, void* out_bm, int64_t m, int64_t nb, int scale_exp,
                        void* ws, hipStream_t st) {
  using AccT = typename ScatterAcc<MODE>::type;
  unsigned char* wsb = static_cast<unsigned char*>(ws);
  uint32_t* count = reinterpret_cast<uint32_t*>(wsb);
  uint32_t* active = reinterpret_cast<uint32_t*>(wsb + jit_counts_bytes(nb));
  const int64_t astride = jit_active_stride(m);
  AccT* partial = reinterpret_cast<AccT*>(wsb + jit_counts_bytes(nb) + nb * astride * 4);
  // An ARMED workspace (be_jit_scatter_workspace_arm) holds zero counters on entry and the reduce kernel below re-arms them:
  // no zeroing launch in front of the compaction (k_fill_bytes: 4.7 us of a 122-us C3 step).  Any other workspace: zeroed here.
  const bool armed = jit_workspace_is_armed(ws);
  int rc = be_internal_compact_spikes_batched(spikes_bm, sd, m, nb, active, astride, count, armed ? 0 : 1, st);
  if (rc != BE_OK) return rc;
  const ScatterGeom g = scatter_geom(p, MODE == MODE_SCALAR, nb);
  const size_t lds = (size_t)g.piece_len * sizeof(AccT);
  const float fx_scale = ldexpf(1.0f, scale_exp - 32);
  const dim3 sgrid((unsigned)(g.n_classes * g.pieces * g.parts), (unsigned)nb);
  const int prof = be_prof_begin(st);
  if (g.pieces == 1) {
    auto kern = k_jit_mv_scatter<MODE, true>;
    BE_HIP(be_allow_lds(reinterpret_cast<const void*>(kern), (int)lds));
    hipLaunchKernelGGL(kern, sgrid, dim3(1024), lds, st, p, active, count, g.pieces, g.parts, g.piece_len, fx_scale, partial,
                       astride);
  } else {
    auto kern = k_jit_mv_scatter<MODE, false>;
    BE_HIP(be_allow_lds(reinterpret_cast<const void*>(kern), (int)lds));
    hipLaunchKernelGGL(kern, sgrid, dim3(1024), lds, st, p, active, count, g.pieces, g.parts, g.piece_len, fx_scale, partial,
                       astride);
  }
  be_prof_end(prof, st);
  BE_LAUNCH_CHECK();
  {
    const int64_t q_per_chunk = (std::min<int64_t>(p.chunk_size, p.walk_len) + p.stride - 1) / p.stride;
    const dim3 rgrid((unsigned)((q_per_chunk + 255) / 256), (unsigned)p.n_chunks, (unsigned)nb);
    const int64_t pstride = (int64_t)g.n_classes * g.pieces * g.parts * g.piece_len;
    hipLaunchKernelGGL((k_jit_scatter_reduce<MODE, W>), rgrid, dim3(256), 0, st, partial, p, g.pieces, g.parts, g.piece_len,
                       ldexp(1.0, -scale_exp), static_cast<W*>(out_bm), pstride, armed ? count : static_cast<uint32_t*>(nullptr));
  }
  BE_LAUNCH_CHECK();
  return BE_OK;
}

template <int MODE, typename W>
int jit_mv_scatter(const JitP& p, const void* spikes, int sd, void* out, int64_t m, int scale_exp, void* ws,
                   hipStream_t st) {
  return jit_scatter_batched<MODE, W>(p, spikes, sd, out, m, 1, scale_exp, ws, st);
}

inline int64_t jit_mv_ws_bytes(int64_t shape1, int64_t in_len, int64_t out_len, int gather) {
  if (gather) {
    const int64_t n_words = (in_len + 31) / 32;
    const int64_t chunk = std::max<int64_t>(1, (shape1 + 3) / 4);
    const int64_t n_chunks = (in_len + chunk - 1) / chunk;
    return be_align_up((n_words + 2) * 4, 256) + be_align_up(std::max<int64_t>(1, n_chunks) * out_len * 8, 256);
  }
  const JitP p = make_params(shape1, out_len, 0, 2, 32, 0, 0);
  return jit_scatter_ws_bytes(p, in_len, 1);
}

// fixed-point exponent from the weight bound of a family: |w| * 2^e * n_rows < 2^62
inline int jit_scale_exp(int mode, double w0, double w1, int64_t n_rows) {
  double wmax = std::fabs(w0);
  if (mode == MODE_UNIFORM) wmax = std::max(std::fabs(w0), std::fabs(w0 + w1));
  if (mode == MODE_NORMAL) wmax = std::fabs(w0) + 6.5 * std::fabs(w1);     // |normal01| <= 6.37 after the 1e-10 clamp
  int e = 0;
  if (wmax > 0) std::frexp(wmax, &e);
  int lg = 1;
  while ((1ll << lg) < n_rows + 1) ++lg;
  return std::max(-90, std::min(150, 62 - e - lg));
}

template <int MODE>
int jit_mv_dispatch(const JitP& p, int wdtype, const void* spikes, int sd, void* out, int64_t in_len, int64_t out_len,
                    int gather, int scale_exp, void* ws, hipStream_t st) {
#define BE_JIT_CASE(WT)                                                                       \
  return gather ? jit_mv_gather<MODE, WT>(p, spikes, sd, out, out_len, ws, st)                \
                : jit_mv_scatter<MODE, WT>(p, spikes, sd, out, in_len, scale_exp, ws, st)
  switch (wdtype) {
    case BE_F32: BE_JIT_CASE(float);
    case BE_F64: BE_JIT_CASE(double);
    case BE_F16: BE_JIT_CASE(__half);
    case BE_BF16: BE_JIT_CASE(__hip_bfloat16);
    default: be_set_error("unknown weight dtype"); return BE_ERR_INVALID;
  }
#undef BE_JIT_CASE
}

template <int MODE, typename A>
int jit_mm_run(const JitP& p, const uint32_t* mask, int64_t rows, int nc, int gather, A* out_bm, hipStream_t st) {
  (void)gather;   // only the gather ("notrans") direction comes here; the scatter runs jit_scatter_batched
  // a chunk's masks in LDS when they fit 128 KB at the batch's width
  const int64_t mask_sz = nc <= 8 ? 1 : (nc <= 16 ? 2 : 4);
  const int64_t chunk_cols = std::min<int64_t>(p.chunk_size, p.walk_len);
  const int64_t win_cap = 128 * 1024 / mask_sz;                      // columns whose masks fit LDS
  const int64_t n_win = (chunk_cols + win_cap - 1) / win_cap;
  // (the walks keep their state from window to window, so windows cost two barriers and a staging pass each, nothing else)
  if (n_win <= 256 && p.stride == 4 && rows > 0) {
    const int64_t win_cols = (chunk_cols + n_win - 1) / n_win;
    const size_t lds = (size_t)be_align_up(win_cols * mask_sz, 16);
    const unsigned grid = (unsigned)((rows + 1023) / 1024);
#define BE_JIT_MM_LDS(T_)                                                                                     \
    do {                                                                                                      \
      auto kern = k_jit_mm_gather_lds<MODE, A, T_>;                                                          \
      BE_HIP(be_allow_lds(reinterpret_cast<const void*>(kern), (int)lds));                                   \
      hipLaunchKernelGGL(kern, dim3(grid), dim3(1024), lds, st, p, mask, rows, nc, out_bm, win_cols);        \
    } while (0)
    if (mask_sz == 1) BE_JIT_MM_LDS(uint8_t);
    else if (mask_sz == 2) BE_JIT_MM_LDS(uint16_t);
    else BE_JIT_MM_LDS(uint32_t);
#undef BE_JIT_MM_LDS
    BE_LAUNCH_CHECK();
    return BE_OK;
  }
  if (nc <= 8) hipLaunchKernelGGL((k_jit_mm_gather<MODE, A, 8>), dim3(gcap(rows, 256, 4096)), dim3(256), 0, st, p, mask, rows, nc, out_bm);
  else if (nc <= 16) hipLaunchKernelGGL((k_jit_mm_gather<MODE, A, 16>), dim3(gcap(rows, 256, 4096)), dim3(256), 0, st, p, mask, rows, nc, out_bm);
  else hipLaunchKernelGGL((k_jit_mm_gather<MODE, A, 32>), dim3(gcap(rows, 256, 4096)), dim3(256), 0, st, p, mask, rows, nc, out_bm);
  BE_LAUNCH_CHECK();
  return BE_OK;
}

// weights of explicitly listed edges (rows / cols in the RNG orientation): the device hashes exposed as an op
template <int MODE>
__global__ void __launch_bounds__(256) k_jit_edge_weights(JitP p, const int32_t* __restrict__ rows,
                                                          const int32_t* __restrict__ cols, int64_t n, float* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = edge_weight<MODE, float>(p, (uint32_t)rows[i], (uint32_t)cols[i]);
}

}  // namespace

extern "C" {

// Arm a workspace of the scatter orientation (mv or mm): its spike counters are zeroed once, here, and every later scatter call
// on it skips the zeroing launch — the call's last kernel leaves the counters at zero again.  The library remembers the POINTER:
// disarm before freeing the memory (or before handing it to anything else).  A call that fails half-way leaves the counters in an
// unknown state: disarm, or arm again.
int be_jit_scatter_workspace_arm(void* workspace, int64_t workspace_bytes, be_stream_t stream) {
  BE_REQUIRE(workspace != nullptr && workspace_bytes >= 256, BE_ERR_INVALID, "null / tiny workspace");
  // the counters sit at the head: 4 bytes per batch column, at most kMaxBatch columns (whatever else the fill covers is scratch)
  BE_HIP(be_fill_async(workspace, 0, (size_t)std::min<int64_t>(workspace_bytes, 4 * 65536), static_cast<hipStream_t>(stream)));
  std::lock_guard<std::mutex> lk(g_armed_mu);
  g_armed.insert(workspace);
  return BE_OK;
}
int be_jit_scatter_workspace_disarm(void* workspace) {
  std::lock_guard<std::mutex> lk(g_armed_mu);
  g_armed.erase(workspace);
  return BE_OK;
}

int64_t be_binary_jitmv_workspace_bytes(int64_t shape1, int64_t in_len, int64_t out_len, int gather) {
  return jit_mv_ws_bytes(shape1, in_len, out_len, gather);
}

static int jitmv_impl(int mode, double w0, double w1, int wdtype, int64_t clen, uint32_t seed, const void* spikes,
                      int spike_dtype, void* out, int64_t shape1, int64_t in_len, int64_t out_len, int gather, int scale_exp,
                      int class_begin, int class_count, int64_t row_begin, void* workspace, int64_t workspace_bytes, be_stream_t stream);

int be_binary_jitmv(int mode, double w0, double w1, int wdtype, int64_t clen, uint32_t seed, const void* spikes,
                    int spike_dtype, void* out, int64_t shape1, int64_t in_len, int64_t out_len, int gather, int scale_exp,
                    void* workspace, int64_t workspace_bytes, be_stream_t stream) {
  return jitmv_impl(mode, w0, w1, wdtype, clen, seed, spikes, spike_dtype, out, shape1, in_len, out_len, gather, scale_exp, 0,
                    -1, 0, workspace, workspace_bytes, stream);
}

// The gather ("notrans") orientation sharded by OUTPUT ROWS: the generator rows are the outputs there and a row's walk is keyed
// by (seed, row, chunk, lane) alone, so a rank that owns the rows [row_begin, row_begin + row_count) computes exactly those
// outputs from the full (all-gathered) spike vector — nothing stored, outputs disjoint, their concatenation is the unsharded
// result bit for bit.  out: row_count elements.  Workspace: be_binary_jitmv_workspace_bytes(shape1, in_len, row_count, 1).
int be_binary_jitmv_rows(int mode, double w0, double w1, int wdtype, int64_t clen, uint32_t seed, const void* spikes,
                         int spike_dtype, void* out, int64_t shape1, int64_t in_len, int64_t row_begin, int64_t row_count,
                         void* workspace, int64_t workspace_bytes, be_stream_t stream) {
  BE_REQUIRE(row_begin >= 0 && row_count >= 0 && row_begin + row_count < (1ll << 32), BE_ERR_INVALID, "bad row range");
  return jitmv_impl(mode, w0, w1, wdtype, clen, seed, spikes, spike_dtype, out, shape1, in_len, row_count, /*gather=*/1, 0, 0, -1,
                    row_begin, workspace, workspace_bytes, stream);
}

int be_jit_scatter_classes(int64_t shape1, int64_t out_len, int stride) {
  BE_REQUIRE(shape1 >= 0 && out_len >= 0 && (stride == 32 || stride == 4), BE_ERR_INVALID, "bad arguments");
  const JitP p = make_params(shape1, out_len, 0, 2, stride, 0, 0);
  return p.n_chunks * stride;
}

int be_binary_jitmv_sharded(int mode, double w0, double w1, int wdtype, int64_t clen, uint32_t seed, const void* spikes,
                            int spike_dtype, void* out, int64_t shape1, int64_t in_len, int64_t out_len, int class_begin,
                            int class_count, int scale_exp, void* workspace, int64_t workspace_bytes, be_stream_t stream) {
  BE_REQUIRE(class_begin >= 0 && class_count >= 0, BE_ERR_INVALID, "bad class range");
  return jitmv_impl(mode, w0, w1, wdtype, clen, seed, spikes, spike_dtype, out, shape1, in_len, out_len, /*gather=*/0,
                    scale_exp, class_begin, class_count, 0, workspace, workspace_bytes, stream);
}

static int jitmv_impl(int mode, double w0, double w1, int wdtype, int64_t clen, uint32_t seed, const void* spikes,
                      int spike_dtype, void* out, int64_t shape1, int64_t in_len, int64_t out_len, int gather, int scale_exp,
                      int class_begin, int class_count, int64_t row_begin, void* workspace, int64_t workspace_bytes, be_stream_t stream) {
  BE_REQUIRE(mode >= 0 && mode <= 2, BE_ERR_INVALID, "mode must be 0 (scalar), 1 (uniform) or 2 (normal)");
  BE_REQUIRE(in_len >= 0 && out_len >= 0 && shape1 >= 0, BE_ERR_INVALID, "bad shape");
  BE_REQUIRE(in_len < (1ll << 32) && out_len < (1ll << 32), BE_ERR_RANGE, "dimensions must fit uint32 for the RNG keys");
  if (out_len == 0) return BE_OK;
  BE_REQUIRE(out != nullptr, BE_ERR_INVALID, "out is NULL");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const size_t esz = (wdtype == BE_F64) ? 8 : (wdtype == BE_F32 ? 4 : 2);
  if (in_len == 0 || clen <= 0) {   // empty walk or prob == 0: all zeros (documented choice)
    BE_HIP(be_fill_async(out, 0, (size_t)out_len * esz, st));
    return BE_OK;
  }
  BE_REQUIRE(spikes != nullptr, BE_ERR_INVALID, "spikes is NULL");
  BE_REQUIRE(workspace != nullptr && workspace_bytes >= jit_mv_ws_bytes(shape1, in_len, out_len, gather),
             BE_ERR_WORKSPACE, "workspace too small");
  BE_REQUIRE(gather || mode == MODE_SCALAR || (scale_exp - 32 > -126 && scale_exp - 32 < 127), BE_ERR_INVALID,
             "scale_exp out of range");
  JitP p = make_params(shape1, gather ? in_len : out_len, seed, clen, 32, w0, w1);
  p.row0 = (uint32_t)row_begin;
  if (class_count >= 0) {      // sharded scatter: only the classes [class_begin, class_begin + class_count)
    BE_REQUIRE(class_begin + class_count <= p.cls_count, BE_ERR_RANGE, "class range exceeds be_jit_scatter_classes()");
    p.cls_begin = class_begin;
    p.cls_count = class_count;
    if (class_count == 0) {
      BE_HIP(be_fill_async(out, 0, (size_t)out_len * esz, st));
      return BE_OK;
    }
  }
  switch (mode) {
    case MODE_SCALAR: return jit_mv_dispatch<MODE_SCALAR>(p, wdtype, spikes, spike_dtype, out, in_len, out_len, gather, scale_exp, workspace, st);
    case MODE_UNIFORM: return jit_mv_dispatch<MODE_UNIFORM>(p, wdtype, spikes, spike_dtype, out, in_len, out_len, gather, scale_exp, workspace, st);
    default: return jit_mv_dispatch<MODE_NORMAL>(p, wdtype, spikes, spike_dtype, out, in_len, out_len, gather, scale_exp, workspace, st);
  }
}

int64_t be_binary_jitmm_workspace_bytes(int64_t shape1, int64_t in_len, int64_t out_len, int64_t n_batch, int gather) {
  const int64_t nb = std::max<int64_t>(1, n_batch);
  if (gather) return be_align_up(in_len * 4, 256) + be_align_up(nb * out_len * 8, 256);
  const JitP p = make_params(shape1, out_len, 0, 2, 4, 0, 0);
  return jit_scatter_ws_bytes(p, in_len, nb);
}

// spikes_bm [n_batch, in_len] -> out_bm [n_batch, out_len]; gather: generator rows = out_len, walk over in_len;
// scatter: generator rows = in_len, walk over out_len.  Works in passes of 32 batch columns.
int be_binary_jitmm(int mode, double w0, double w1, int wdtype, int64_t clen, uint32_t seed, const void* spikes_bm,
                    int spike_dtype, void* out_bm, int64_t shape1, int64_t in_len, int64_t out_len, int64_t n_batch,
                    int gather, void* workspace, int64_t workspace_bytes, be_stream_t stream) {
  BE_REQUIRE(mode >= 0 && mode <= 2, BE_ERR_INVALID, "mode must be 0 (scalar), 1 (uniform) or 2 (normal)");
  BE_REQUIRE(in_len >= 0 && out_len >= 0 && shape1 >= 0 && n_batch >= 0, BE_ERR_INVALID, "bad shape");
  BE_REQUIRE(in_len < (1ll << 32) && out_len < (1ll << 32), BE_ERR_RANGE, "dimensions must fit uint32 for the RNG keys");
  if (out_len == 0 || n_batch == 0) return BE_OK;
  BE_REQUIRE(out_bm != nullptr, BE_ERR_INVALID, "out is NULL");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const size_t esz = (wdtype == BE_F64) ? 8 : (wdtype == BE_F32 ? 4 : 2);
  if (in_len == 0 || clen <= 0) {
    BE_HIP(be_fill_async(out_bm, 0, (size_t)out_len * n_batch * esz, st));
    return BE_OK;
  }
  BE_REQUIRE(spikes_bm != nullptr, BE_ERR_INVALID, "spikes is NULL");
  BE_REQUIRE(workspace != nullptr &&
                 workspace_bytes >= be_binary_jitmm_workspace_bytes(shape1, in_len, out_len, n_batch, gather),
             BE_ERR_WORKSPACE, "workspace too small");
  if (!gather) {
    // scatter: the batched residue-class kernel of the mv path with lane stride 4 (gridDim.y = batch column)
    const JitP ps = make_params(shape1, out_len, seed, clen, 4, w0, w1);
    const int se = jit_scale_exp(mode, w0, w1, in_len);
#define BE_JITMM_SC(MODE_)                                                                                              \
    switch (wdtype) {                                                                                                   \
      case BE_F32: return jit_scatter_batched<MODE_, float>(ps, spikes_bm, spike_dtype, out_bm, in_len, n_batch, se, workspace, st);          \
      case BE_F64: return jit_scatter_batched<MODE_, double>(ps, spikes_bm, spike_dtype, out_bm, in_len, n_batch, se, workspace, st);         \
      case BE_F16: return jit_scatter_batched<MODE_, __half>(ps, spikes_bm, spike_dtype, out_bm, in_len, n_batch, se, workspace, st);         \
      case BE_BF16: return jit_scatter_batched<MODE_, __hip_bfloat16>(ps, spikes_bm, spike_dtype, out_bm, in_len, n_batch, se, workspace, st); \
      default: be_set_error("unknown weight dtype"); return BE_ERR_INVALID;                                             \
    }
    if (mode == MODE_SCALAR) { BE_JITMM_SC(MODE_SCALAR) }
    else if (mode == MODE_UNIFORM) { BE_JITMM_SC(MODE_UNIFORM) }
    else { BE_JITMM_SC(MODE_NORMAL) }
#undef BE_JITMM_SC
  }
  unsigned char* wsb = static_cast<unsigned char*>(workspace);
  uint32_t* mask = reinterpret_cast<uint32_t*>(wsb);
  void* scratch = wsb + be_align_up(in_len * 4, 256);
  const JitP p = make_params(shape1, gather ? in_len : out_len, seed, clen, 4, w0, w1);
  const int64_t gen_rows = gather ? out_len : in_len;
  const bool f64 = (wdtype == BE_F64);
  const bool direct = (wdtype == BE_F32 || wdtype == BE_F64);   // kernels write f32 / f64; f16 / bf16 via f32 scratch
  void* dst = direct ? out_bm : scratch;
  const size_t asz = f64 ? 8 : 4;
  if (!gather) BE_HIP(be_fill_async(dst, 0, (size_t)out_len * n_batch * asz, st));
  const size_t row_bytes = spike_dtype == BE_SPIKE_BITS ? (size_t)((in_len + 31) / 32) * 4
                                                         : (size_t)in_len * (spike_dtype == BE_SPIKE_FLOAT ? 4 : 1);
  const int prof = be_prof_begin(st);
  // (Cutting a wide batch into passes of 8 columns was measured in round 3 and is slower — n = 1M, 32 columns: 4.7 ms in one pass
  //  against 4 x 2.87 = 11.5 ms, every pass re-walks the whole matrix; what helps is windows whose walks keep their state,
  //  k_jit_mm_gather_lds.)
  for (int64_t b0 = 0; b0 < n_batch; b0 += 32) {
    const int nc = (int)std::min<int64_t>(32, n_batch - b0);
    const void* chunk = static_cast<const unsigned char*>(spikes_bm) + (size_t)b0 * row_bytes;
    if (spike_dtype == BE_SPIKE_BITS)
      hipLaunchKernelGGL(k_jit_masks_bits, dim3(gcap(in_len, 256, 2048)), dim3(256), 0, st, static_cast<const uint32_t*>(chunk),
                         in_len, nc, mask);
    else if (spike_dtype == BE_SPIKE_FLOAT)
      hipLaunchKernelGGL(k_jit_masks<SpikeFloat>, dim3(gcap(in_len, 256, 2048)), dim3(256), 0, st,
                         static_cast<const float*>(chunk), in_len, nc, mask);
    else
      hipLaunchKernelGGL(k_jit_masks<SpikeBool>, dim3(gcap(in_len, 256, 2048)), dim3(256), 0, st,
                         static_cast<const uint8_t*>(chunk), in_len, nc, mask);
    BE_LAUNCH_CHECK();
    void* o = static_cast<unsigned char*>(dst) + (size_t)b0 * out_len * asz;
    int rc;
#define BE_JITMM(MODE_)                                                                                                  \
    rc = f64 ? jit_mm_run<MODE_, double>(p, mask, gen_rows, nc, gather, static_cast<double*>(o), st)                     \
             : jit_mm_run<MODE_, float>(p, mask, gen_rows, nc, gather, static_cast<float*>(o), st)
    if (mode == MODE_SCALAR) { BE_JITMM(MODE_SCALAR); }
    else if (mode == MODE_UNIFORM) { BE_JITMM(MODE_UNIFORM); }
    else { BE_JITMM(MODE_NORMAL); }
#undef BE_JITMM
    if (rc != BE_OK) return rc;
  }
  be_prof_end(prof, st);
  if (!direct) {
    const int64_t n = out_len * n_batch;
    if (wdtype == BE_F16)
      hipLaunchKernelGGL((k_jit_convert<float, __half>), dim3(gcap(n, 256, 2048)), dim3(256), 0, st,
                         static_cast<const float*>(scratch), static_cast<__half*>(out_bm), n);
    else
      hipLaunchKernelGGL((k_jit_convert<float, __hip_bfloat16>), dim3(gcap(n, 256, 2048)), dim3(256), 0, st,
                         static_cast<const float*>(scratch), static_cast<__hip_bfloat16*>(out_bm), n);
    BE_LAUNCH_CHECK();
  }
  return BE_OK;
}


// ---------------------------------------------------------------- per-edge weights
int be_jit_edge_weights(int mode, double w0, double w1, uint32_t seed, const int32_t* rows, const int32_t* cols, int64_t n,
                        float* out, be_stream_t stream) {
  BE_REQUIRE(mode >= 0 && mode <= 2, BE_ERR_INVALID, "mode must be 0 (scalar), 1 (uniform) or 2 (normal)");
  BE_REQUIRE(n >= 0, BE_ERR_INVALID, "negative count");
  if (n == 0) return BE_OK;
  BE_REQUIRE(rows && cols && out, BE_ERR_INVALID, "null pointer");
  hipStream_t st = static_cast<hipStream_t>(stream);
  JitP p{};
  p.seed = seed;
  p.w0 = w0;
  p.w1 = w1;
  const dim3 grid(gcap(n, 256, 8192)), block(256);
  if (mode == MODE_UNIFORM) hipLaunchKernelGGL(k_jit_edge_weights<MODE_UNIFORM>, grid, block, 0, st, p, rows, cols, n, out);
  else if (mode == MODE_NORMAL) hipLaunchKernelGGL(k_jit_edge_weights<MODE_NORMAL>, grid, block, 0, st, p, rows, cols, n, out);
  else hipLaunchKernelGGL(k_jit_edge_weights<MODE_SCALAR>, grid, block, 0, st, p, rows, cols, n, out);
  BE_LAUNCH_CHECK();
  return BE_OK;
}

// ---------------------------------------------------------------- materialisation (generator matrix -> CSR)
// replaces: brainevent/_jit_scalar/csr.cu count + fill (and the uniform / normal twins).
// rows = walk owners: n_rows generator rows, walk over walk_len columns; stride 32 (mv matrix) or 4 (mm matrix).
int be_jitc_csr_count(int64_t clen, uint32_t seed, int64_t shape1, int64_t n_rows, int64_t walk_len, int stride,
                      uint32_t* row_counts, be_stream_t stream) {
  BE_REQUIRE(n_rows >= 0 && walk_len >= 0 && shape1 >= 0, BE_ERR_INVALID, "bad shape");
  BE_REQUIRE(stride == 32 || stride == 4, BE_ERR_INVALID, "stride must be 32 (mv) or 4 (mm)");
  BE_REQUIRE(n_rows < (1ll << 32) && walk_len < (1ll << 31), BE_ERR_RANGE, "dimensions out of range");
  if (n_rows == 0) return BE_OK;
  BE_REQUIRE(row_counts != nullptr, BE_ERR_INVALID, "null pointer");
  hipStream_t st = static_cast<hipStream_t>(stream);
  BE_HIP(be_fill_async(row_counts, 0, (size_t)n_rows * 4, st));
  if (walk_len == 0 || clen <= 0) return BE_OK;
  const JitP p = make_params(shape1, walk_len, seed, clen, stride, 0, 0);
  hipLaunchKernelGGL(k_jit_csr_count, dim3(gcap(n_rows * p.n_chunks * stride, 256, 8192)), dim3(256), 0, st, p, n_rows,
                     row_counts);
  BE_LAUNCH_CHECK();
  return BE_OK;
}

int be_jitc_csr_fill(int mode, double w0, double w1, int64_t clen, uint32_t seed, int64_t shape1, int64_t n_rows,
                     int64_t walk_len, int stride, const int64_t* indptr, uint32_t* cursor, int32_t* indices,
                     float* weights, be_stream_t stream) {
  BE_REQUIRE(mode >= 0 && mode <= 2, BE_ERR_INVALID, "mode must be 0 (scalar), 1 (uniform) or 2 (normal)");
  BE_REQUIRE(n_rows >= 0 && walk_len >= 0 && shape1 >= 0, BE_ERR_INVALID, "bad shape");
  BE_REQUIRE(stride == 32 || stride == 4, BE_ERR_INVALID, "stride must be 32 (mv) or 4 (mm)");
  BE_REQUIRE(n_rows < (1ll << 32) && walk_len < (1ll << 31), BE_ERR_RANGE, "dimensions out of range");
  if (n_rows == 0 || walk_len == 0 || clen <= 0) return BE_OK;
  BE_REQUIRE(indptr && cursor && indices && (mode == MODE_SCALAR || weights), BE_ERR_INVALID, "null pointer");
  hipStream_t st = static_cast<hipStream_t>(stream);
  BE_HIP(be_fill_async(cursor, 0, (size_t)n_rows * 4, st));
  const JitP p = make_params(shape1, walk_len, seed, clen, stride, w0, w1);
  const dim3 grid(gcap(n_rows * p.n_chunks * stride, 256, 8192));
  if (mode == MODE_SCALAR) hipLaunchKernelGGL(k_jit_csr_fill<MODE_SCALAR>, grid, dim3(256), 0, st, p, n_rows, indptr, cursor, indices, weights);
  else if (mode == MODE_UNIFORM) hipLaunchKernelGGL(k_jit_csr_fill<MODE_UNIFORM>, grid, dim3(256), 0, st, p, n_rows, indptr, cursor, indices, weights);
  else hipLaunchKernelGGL(k_jit_csr_fill<MODE_NORMAL>, grid, dim3(256), 0, st, p, n_rows, indptr, cursor, indices, weights);
  BE_LAUNCH_CHECK();
  return BE_OK;
}

#define BE_DEF_JIT_VARIANT(F, M, W, WD)                                                                               \
  int be_binary_jit##F##mv_notrans_##W(BE_JIT_MV_ARGS) {                                                               \
    return be_binary_jitmv(M, w0, w1, WD, clen, seed, spikes, spike_dtype, out, shape1, in_len, out_len, 1, scale_exp,  \
                           workspace, workspace_bytes, stream);                                                        \
  }                                                                                                                    \
  int be_binary_jit##F##mv_trans_##W(BE_JIT_MV_ARGS) {                                                                 \
    return be_binary_jitmv(M, w0, w1, WD, clen, seed, spikes, spike_dtype, out, shape1, in_len, out_len, 0, scale_exp,  \
                           workspace, workspace_bytes, stream);                                                        \
  }                                                                                                                    \
  int be_binary_jit##F##mm_notrans_##W(BE_JIT_MM_ARGS) {                                                               \
    return be_binary_jitmm(M, w0, w1, WD, clen, seed, spikes_bm, spike_dtype, out_bm, shape1, in_len, out_len, n_batch, \
                           1, workspace, workspace_bytes, stream);                                                     \
  }                                                                                                                    \
  int be_binary_jit##F##mm_trans_##W(BE_JIT_MM_ARGS) {                                                                 \
    return be_binary_jitmm(M, w0, w1, WD, clen, seed, spikes_bm, spike_dtype, out_bm, shape1, in_len, out_len, n_batch, \
                           0, workspace, workspace_bytes, stream);                                                     \
  }

BE_FOR_JIT_VARIANTS(BE_DEF_JIT_VARIANT)

}  // extern "C"
