// be_jitc_shared.h — the light_rng sampler, the walk parameters and the per-edge weight hashes shared by the JIT-connectivity
// translation units (be_jitc.hip: the event-driven products; be_jitc_float.hip: their float-operand twins).  Reproduced bit for
// bit from the reference (uint32 arithmetic): brainevent/_numba_random.py:385-502; the walk itself is documented in be_jitc.hip.
#pragma once
#include "be_common.h"
#include <algorithm>
#include <cmath>
#include <type_traits>

namespace {

enum { MODE_SCALAR = 0, MODE_UNIFORM = 1, MODE_NORMAL = 2 };

// ------------------------------------------------------------------------------------------------ light_rng
__device__ __forceinline__ uint32_t lr_next(uint32_t x) {
  x ^= x << 13; x ^= x >> 17; x ^= x << 5;
  return x == 0u ? 0x6d2b79f5u : x;
}
// the same step for a state known to be non-zero: xorshift32 (13, 17, 5) is a bijection of the non-zero 32-bit words, so
// a walk that starts from lr_init (never zero) never reaches zero and the reference's zero fix-up cannot fire — the hot
// loops drop its compare + select (2 of ~15 vector instructions per generated edge; the gather kernel is VALU-bound:
// SQ_ACTIVE_INST_VALU x 4 waves per SIMD = 1.15 of SQ_WAVE_CYCLES at C3)
__device__ __forceinline__ uint32_t lr_next_nz(uint32_t x) {
  x ^= x << 13; x ^= x >> 17; x ^= x << 5;
  return x;
}
__device__ __forceinline__ uint32_t lr_bounded(uint32_t r, uint32_t bound) { return __umulhi(r, bound); }
__device__ __forceinline__ uint32_t lr_init(uint32_t seed, uint32_t row, uint32_t chunk, uint32_t lane) {
  uint32_t x = seed ^ 0xd1b54a35u;
  x ^= row * 0x85ebca6bu;
  x ^= chunk * 0xc2b2ae35u;
  x ^= lane * 0x27d4eb2du;
  x = be_mix32(x);
  return x == 0u ? 0x6d2b79f5u : x;
}
__device__ __forceinline__ uint32_t lr_initial_q(uint32_t& state, uint32_t cl) {
  const uint32_t n = cl - 1u;
  for (;;) {
    state = lr_next(state);
    const uint32_t q = lr_bounded(state, n);
    state = lr_next(state);
    const uint32_t gate = lr_bounded(state, n);
    if (gate < n - q) return q;
  }
}
__device__ __forceinline__ float lr_uniform01(uint32_t seed, uint32_t row, uint32_t col) {
  uint32_t h = seed ^ 0xa0761d65u;
  h ^= row * 0xe7037ed1u;
  h ^= col * 0x8ebc6af1u;
  h = be_mix32(h);
  return (float)(h & 0x00ffffffu) * (1.0f / 16777216.0f);
}
// No FMA contraction in the weight formulas: the reference evaluates them one rounded f32 operation at a
// time (numpy golden model, brainevent/_numba_random.py:433-486); the Acklam rational cancels heavily, so a
// fused multiply-add changes a weight by ~1e-5 relative.
__device__ __forceinline__ float lr_normal01(uint32_t seed, uint32_t row, uint32_t col) {
#pragma clang fp contract(off)
  float u = lr_uniform01(seed, row, col);
  const float lo = 1e-10f, hi = (float)(1.0 - 1e-10);
  u = u < lo ? lo : (u > hi ? hi : u);
  const float a1 = -39.696830f, a2 = 220.94609f, a3 = -275.92851f, a4 = 138.35775f, a5 = -30.664799f, a6 = 2.5066283f;
  const float b1 = -54.476099f, b2 = 161.58584f, b3 = -155.69898f, b4 = 66.801312f, b5 = -13.280681f;
  const float c1 = -0.007784894f, c2 = -0.32239646f, c3 = -2.4007583f, c4 = -2.5497325f, c5 = 4.3746641f, c6 = 2.9381640f;
  const float d1 = 0.007784696f, d2 = 0.32246713f, d3 = 2.4451342f, d4 = 3.7544087f;
  float z;
  if (u < 0.02425f) {
    const float v = sqrtf(-2.0f * logf(u));
    z = -((((((c1 * v + c2) * v + c3) * v + c4) * v + c5) * v + c6) / ((((d1 * v + d2) * v + d3) * v + d4) * v + 1.0f));
  } else if (u > 0.97575f) {
    const float v = sqrtf(-2.0f * logf(1.0f - u));
    z = (((((c1 * v + c2) * v + c3) * v + c4) * v + c5) * v + c6) / ((((d1 * v + d2) * v + d3) * v + d4) * v + 1.0f);
  } else {
    const float v = u - 0.5f, r = v * v;
    z = (((((a1 * r + a2) * r + a3) * r + a4) * r + a5) * r + a6) * v /
        (((((b1 * r + b2) * r + b3) * r + b4) * r + b5) * r + 1.0f);
  }
  return z;
}

struct JitP {
  uint32_t seed, cl;        // cl already clamped to >= 2
  int64_t chunk_size;       // ceil(shape[1] / 4)
  int64_t walk_len;         // length of the walk dimension (vector length for gather, output length for scatter)
  int n_chunks;             // ceil(walk_len / chunk_size)
  int stride;               // 32 (mv) or 4 (mm)
  double w0, w1;            // scalar: weight, -- | uniform: low, span | normal: loc, scale
  int cls_begin, cls_count; // scatter: the (chunk, lane) classes [cls_begin, cls_begin + cls_count) this call owns
  uint32_t row0;            // gather: generator row of output 0 (a rank of a row-sharded gather computes rows [row0, row0 + m))
};

// edge weight in the arithmetic type A (float or double); (row, col) are the RNG-orientation coordinates
template <int MODE, typename A>
__device__ __forceinline__ A edge_weight(const JitP& p, uint32_t row, uint32_t col) {
#pragma clang fp contract(off)
  if (MODE == MODE_UNIFORM) return (A)p.w0 + (A)lr_uniform01(p.seed, row, col) * (A)p.w1;
  if (MODE == MODE_NORMAL) return (A)p.w0 + (A)lr_normal01(p.seed, row, col) * (A)p.w1;
  return (A)p.w0;
}

// ------------------------------------------------------------------------------------------------ host
inline int gcap(int64_t n, int block, int cap) {
  int64_t g = (n + block - 1) / block;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

inline JitP make_params(int64_t shape1, int64_t walk_len, uint32_t seed, int64_t clen, int stride, double w0, double w1) {
  JitP p;
  p.seed = seed;
  p.cl = (uint32_t)(clen < 2 ? 2 : clen);
  p.chunk_size = std::max<int64_t>(1, (shape1 + 3) / 4);
  p.walk_len = walk_len;
  p.n_chunks = (int)((walk_len + p.chunk_size - 1) / p.chunk_size);
  p.stride = stride;
  p.w0 = w0;
  p.w1 = w1;
  p.cls_begin = 0;
  p.row0 = 0u;
  p.cls_count = p.n_chunks * stride;
  return p;
}


// ------------------------------------------------------------------------------------------------ scatter by residue class
// (shared by the event-driven scatter of be_jitc.hip and the float-operand scatter of be_jitc_float.hip)
template <int MODE> struct ScatterAcc { using type = unsigned long long; };
template <> struct ScatterAcc<MODE_SCALAR> { using type = uint32_t; };

__device__ __forceinline__ unsigned long long jit_fixed_from_f32(float w, float scale) {
  // same construction as fixed_from_f32 in be_csr.hip: w * 2^scale_exp split into (hi, lo) words in f32
  const float t = w * scale;
  const float hf = floorf(t);
  const int hi = (int)hf;
  const unsigned lo = (unsigned)((t - hf) * 4294967296.0f);
  return ((unsigned long long)(unsigned)hi << 32) | lo;
}

// partial is class-major ([class][piece][part][piece_len], class = chunk * stride + lane residue) while the output is
// column-major in (q, lane): out[chunk_start + stride * q + l].  One workgroup transposes a tile of `stride` classes x
// 256 q through LDS: coalesced reads per class row, coalesced writes of stride * 256 consecutive outputs.
// piece_len is a multiple of 256, so a tile never straddles two pieces.  gridDim.z = batch column.
template <int MODE, typename W>
__global__ void __launch_bounds__(256) k_jit_scatter_reduce(const typename ScatterAcc<MODE>::type* __restrict__ partial,
                                                            JitP p, int pieces, int parts, uint32_t piece_len,
                                                            double inv_scale, W* __restrict__ out, int64_t partial_stride,
                                                            uint32_t* __restrict__ rearm = nullptr) {
  using TileT = typename std::conditional<std::is_same<W, double>::value, double, float>::type;
  __shared__ TileT tile[32][257];
  // an armed workspace: the step's last kernel leaves the spike counter of its batch column at zero for the next call
  if (rearm != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) rearm[blockIdx.z] = 0u;
  partial += (int64_t)blockIdx.z * partial_stride;
  out += (int64_t)blockIdx.z * p.walk_len;
  const int S = p.stride;
  const int chunk = blockIdx.y;
  const int64_t cs = (int64_t)chunk * p.chunk_size;
  const int64_t ce = cs + p.chunk_size < p.walk_len ? cs + p.chunk_size : p.walk_len;
  const int64_t width = ce - cs;
  const int64_t q0 = (int64_t)blockIdx.x * 256;
  if (q0 * S >= width) return;
  const int64_t piece = q0 / piece_len, i0 = q0 - piece * piece_len;
  const int t = threadIdx.x;
  using AccT = typename ScatterAcc<MODE>::type;
  const int64_t cls_stride = (int64_t)pieces * parts * piece_len;      // between consecutive classes
  // class c = chunk * S + l lives at local index c - cls_begin; classes outside the owned range read as zero
  const int c0 = chunk * S - p.cls_begin;
  // A thread sums FOUR consecutive positions q of one class row per load (16 bytes of counts, 32 of fixed-point sums) — round 5:
  // 4-byte loads left the 48 MB this kernel moves at C3 at 3.3 TB/s (14.5 us); lanes 0..63 cover the tile's 256 positions, the
  // four waves take the class rows l = wave, wave + 4, ...; every row x part load of a thread is issued before the first add.
  const int qi = (t & 63) * 4, rsel = t >> 6;
  const AccT* base = partial + ((int64_t)piece * parts) * (int64_t)piece_len + i0 + qi;
  constexpr int kRows = 8;                                  // class rows per thread and round (S = 32: one round)
  for (int l0 = 0; l0 < S; l0 += 4 * kRows) {
    unsigned long long sum[kRows][4];
#pragma unroll
    for (int u = 0; u < kRows; ++u)
#pragma unroll
      for (int e = 0; e < 4; ++e) sum[u][e] = 0;
    for (int q2 = 0; q2 < parts; ++q2) {
      AccT v[kRows][4];
#pragma unroll
      for (int u = 0; u < kRows; ++u) {      // unconditional loads from a clamped class row; the select happens at the add
        const int c = c0 + l0 + rsel + 4 * u;
        const int cc = c < 0 ? 0 : (c >= p.cls_count ? p.cls_count - 1 : c);
        const AccT* src = base + (int64_t)cc * cls_stride + (int64_t)q2 * piece_len;
        if constexpr (sizeof(AccT) == 4) {
          const uint4 x = *reinterpret_cast<const uint4*>(src);
          v[u][0] = x.x; v[u][1] = x.y; v[u][2] = x.z; v[u][3] = x.w;
        } else {
          const ulonglong2 x = reinterpret_cast<const ulonglong2*>(src)[0], y = reinterpret_cast<const ulonglong2*>(src)[1];
          v[u][0] = x.x; v[u][1] = x.y; v[u][2] = y.x; v[u][3] = y.y;
        }
      }
#pragma unroll
      for (int u = 0; u < kRows; ++u) {
        const int l = l0 + rsel + 4 * u, c = c0 + l;
        const bool ok = l < S && c >= 0 && c < p.cls_count;
#pragma unroll
        for (int e = 0; e < 4; ++e) sum[u][e] += ok ? (unsigned long long)v[u][e] : 0ull;
      }
    }
#pragma unroll
    for (int u = 0; u < kRows; ++u) {
      const int l = l0 + rsel + 4 * u;
      if (l >= S) break;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        double val = 0.0;
        if ((q0 + qi + e) * S + l < width)
          val = (MODE == MODE_SCALAR) ? (double)sum[u][e] * p.w0 : (double)(long long)sum[u][e] * inv_scale;
        tile[l][qi + e] = (TileT)val;
      }
    }
  }
  __syncthreads();
  for (int r = 0; r < S; ++r) {
    const int jl = r * 256 + t;                 // tile-local output index: stride * (q - q0) + l
    const int64_t j_local = q0 * S + jl;
    if (j_local < width) WTraits<W>::store_d(out, cs + j_local, (double)tile[jl % S][jl / S]);
  }
}

#ifndef BE_JIT_WG_TARGET
#define BE_JIT_WG_TARGET 256   // scatter workgroups (classes x pieces x parts): one round over the CUs.  512 (two rounds, twice the
                               // partial sums) until late in round 2: C3 143 -> 136 us per step; 384 / 1024: 165 / 162
#endif
constexpr uint32_t kPieceU32 = 32768, kPieceU64 = 16384;   // LDS accumulators per scatter workgroup (128 KiB); multiples of 256

struct ScatterGeom { int n_classes, pieces, parts; uint32_t piece_len; };
inline ScatterGeom scatter_geom(const JitP& p, bool scalar, int64_t n_batch = 1) {
  ScatterGeom g;
  g.n_classes = p.cls_count;
  const int64_t Qmax = (std::min<int64_t>(p.chunk_size, p.walk_len) + p.stride - 1) / p.stride;
  const uint32_t cap = scalar ? kPieceU32 : kPieceU64;
  g.pieces = (int)std::max<int64_t>(1, (Qmax + cap - 1) / cap);
  const int64_t per_piece = (Qmax + g.pieces - 1) / g.pieces;
  g.piece_len = (uint32_t)std::max<int64_t>(256, (per_piece + 255) & ~255ll);   // multiple of the reduce tile
  int parts = (int)(BE_JIT_WG_TARGET / std::max<int64_t>(1, (int64_t)g.n_classes * g.pieces * n_batch));
  g.parts = std::max(1, std::min(parts, 16));
  return g;
}

}  // namespace
