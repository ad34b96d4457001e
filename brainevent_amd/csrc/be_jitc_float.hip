// be_jitc_float.hip — float-operand twins of the JIT-connectivity products for gfx950 (last clause): the same
// on-the-fly matrices (same walks, same per-edge weights: be_jitc_shared.h) against a dense vector / matrix.
//   reference: brainevent/_jit_scalar/float.py:838-905 (jitsmv CPU loops: gather sums v[col] in float64 and multiplies by the
//              weight once; scatter skips the zeros of the operand), :1331-1420 (jitsmm: stride-4 walk, its own matrix);
//              brainevent/_jit_uniform/float.py, brainevent/_jit_normal/float.py (per-edge weights from the hashes).
//   gather  (corder = True):  generator rows = outputs; out[r, c] = sum over the edges (r, j) of w(r, j) * X[j, c]
//   scatter (corder = False): generator rows = inputs;  out[j, c] += w(r, j) * X[r, c] for every edge of every row r with X[r, :] != 0
// Nothing is event-driven here (every element of the operand counts): the gather reads the operand from memory per edge (the
// event-driven twin holds a bitmap in LDS instead), the scatter adds through float atomics into an f32 / f64 image.
#include "be_jitc_shared.h"

namespace {

// A task (row, chunk) is walked by `stride` lanes (32: the mv matrix, 4: the mm matrix), lane l visiting chunk-local columns
// l + stride * q.  NC columns of the operand per pass.  partial[(chunk * m + row) * NC + c]: float64 sums per chunk.
template <int MODE, typename W, int NC>
__global__ void __launch_bounds__(256) k_jit_f_gather(JitP p, const W* __restrict__ X, int64_t n, int64_t c0, int64_t m,
                                                      double* __restrict__ partial) {
  const int S = p.stride;
  const uint32_t l = threadIdx.x % S;
  const int64_t tpb = 256 / S;
  const int chunk = blockIdx.y;
  const int64_t cs = (int64_t)chunk * p.chunk_size;
  const int64_t ce = cs + p.chunk_size < p.walk_len ? cs + p.chunk_size : p.walk_len;
  const int64_t width = ce - cs;
  const uint32_t qmax = width > (int64_t)l ? (uint32_t)((width - l + S - 1) / S) : 0u;      // l + S q < width
  for (int64_t r0 = (int64_t)blockIdx.x * tpb; r0 < m; r0 += (int64_t)gridDim.x * tpb) {   // (whole waves stay in: shuffles below)
    const int64_t row = r0 + threadIdx.x / S;
    double acc[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[c] = 0.0;
    if (row < m) {
      const uint32_t grow = (uint32_t)row;
      uint32_t state = lr_init(p.seed, grow, (uint32_t)chunk, l);
      uint32_t q = lr_initial_q(state, p.cl);
      while (q < qmax) {
        const int64_t col = cs + l + (int64_t)S * q;
        const float w = MODE == MODE_SCALAR ? 1.0f : edge_weight<MODE, float>(p, grow, (uint32_t)col);
#pragma unroll
        for (int c = 0; c < NC; ++c)
          if (c0 + c < n) acc[c] += (double)w * (double)WTraits<W>::load(X, col * n + c0 + c);
        state = lr_next_nz(state);
        q = q + 1u + lr_bounded(state, p.cl - 1u);
      }
    }
#pragma unroll
    for (int c = 0; c < NC; ++c)
      for (int off = S / 2; off > 0; off >>= 1) acc[c] += __shfl_xor(acc[c], off, 64);
    if (l == 0 && row < m) {
#pragma unroll
      for (int c = 0; c < NC; ++c) partial[((int64_t)chunk * m + row) * NC + c] = acc[c];
    }
  }
}

template <int MODE, typename W, int NC>
__global__ void __launch_bounds__(256) k_jit_f_gather_reduce(const double* __restrict__ partial, int n_chunks, int64_t m, int64_t n,
                                                             int64_t c0, double w0, W* __restrict__ out) {
  const int64_t total = m * NC;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / NC;
    const int c = (int)(i - row * NC);
    if (c0 + c >= n) continue;
    double s = 0.0;
    for (int ch = 0; ch < n_chunks; ++ch) s += partial[((int64_t)ch * m + row) * NC + c];
    WTraits<W>::store_d(out, row * n + c0 + c, MODE == MODE_SCALAR ? s * w0 : s);      // one shared weight: multiplied once (:870)
  }
}

// scatter: one thread per (input row, chunk, lane) walk; img [out_len, n] in the accumulator type
template <int MODE, typename W, int NC>
__global__ void __launch_bounds__(256) k_jit_f_scatter(JitP p, const W* __restrict__ X, int64_t n, int64_t c0, int64_t in_len,
                                                       typename WTraits<W>::acc* __restrict__ img) {
  using ACC = typename WTraits<W>::acc;
  const int S = p.stride;
  const int64_t per_row = (int64_t)p.n_chunks * S;
  const int64_t tasks = in_len * per_row;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < tasks; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = t / per_row;
    const int rem = (int)(t - row * per_row);
    const int chunk = rem / S;
    const uint32_t l = (uint32_t)(rem - chunk * S);
    ACC x[NC];
    bool any = false;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      x[c] = c0 + c < n ? (ACC)WTraits<W>::load(X, row * n + c0 + c) : ACC(0);
      any = any || x[c] != ACC(0);
    }
    if (!any) continue;                                      // a zero of the operand adds nothing (:886-887)
    if (MODE == MODE_SCALAR) {
#pragma unroll
      for (int c = 0; c < NC; ++c) x[c] *= (ACC)p.w0;
    }
    const int64_t cs = (int64_t)chunk * p.chunk_size;
    const int64_t ce = cs + p.chunk_size < p.walk_len ? cs + p.chunk_size : p.walk_len;
    const int64_t width = ce - cs;
    const uint32_t qmax = width > (int64_t)l ? (uint32_t)((width - l + S - 1) / S) : 0u;
    uint32_t state = lr_init(p.seed, (uint32_t)row, (uint32_t)chunk, l);
    uint32_t q = lr_initial_q(state, p.cl);
    while (q < qmax) {
      const int64_t col = cs + l + (int64_t)S * q;
      const ACC w = MODE == MODE_SCALAR ? ACC(1) : (ACC)edge_weight<MODE, float>(p, (uint32_t)row, (uint32_t)col);
#pragma unroll
      for (int c = 0; c < NC; ++c)
        if (c0 + c < n && x[c] != ACC(0)) atomicAdd(img + col * n + c0 + c, w * x[c]);
      state = lr_next_nz(state);
      q = q + 1u + lr_bounded(state, p.cl - 1u);
    }
  }
}

// scatter through LDS, the event-driven scatter's structure (be_jitc.hip: one workgroup owns the accumulators of one residue
// class, 64-bit fixed point, order independent) with the operand's value as a per-row factor.  The exponent comes from the
// caller: |w|max * |x|max * rows * 2^scale_exp < 2^62.  Batch-major operand X_bm [n, in_len] and partial sums per column
// (gridDim.y = column), reduced by k_jit_scatter_reduce into out_bm [n, out_len].  With a dense operand every row walks:
// the kernel runs at the walk's rate instead of the atomic units' (C3 shape: 757 ms -> see tools/bench_jit_float.py).
template <int MODE, typename W, bool ONE_PIECE>
__global__ void __launch_bounds__(1024) k_jit_f_scatter_lds(JitP p, const W* __restrict__ X_bm, int64_t in_len, int pieces, int parts,
                                                            uint32_t piece_len, float fx_scale,
                                                            unsigned long long* __restrict__ partial) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  unsigned long long* acc = reinterpret_cast<unsigned long long*>(smem_raw);
  const int part = blockIdx.x % parts;
  const int piece = (blockIdx.x / parts) % pieces;
  const int cls = p.cls_begin + blockIdx.x / (parts * pieces);
  const uint32_t S = (uint32_t)p.stride;
  const uint32_t chunk = (uint32_t)cls / S, l = (uint32_t)cls - chunk * S;
  const W* x = X_bm + (int64_t)blockIdx.y * in_len;
  partial += (int64_t)blockIdx.y * gridDim.x * piece_len;
  const int64_t cs = (int64_t)chunk * p.chunk_size;
  const int64_t ce = cs + p.chunk_size < p.walk_len ? cs + p.chunk_size : p.walk_len;
  const int64_t width = ce - cs;
  const int64_t Q = width > (int64_t)l ? (width - l + S - 1) / S : 0;
  const int64_t q_begin = (int64_t)piece * piece_len;
  const int64_t q_end = q_begin + piece_len < Q ? q_begin + piece_len : Q;
  for (uint32_t i = threadIdx.x; i < piece_len; i += blockDim.x) acc[i] = 0;
  __syncthreads();
  if (q_begin < q_end) {
    const uint32_t qb = (uint32_t)q_begin, qe = (uint32_t)q_end;
    const uint32_t j0 = (uint32_t)(cs + l);
    for (int64_t row = (int64_t)part * blockDim.x + threadIdx.x; row < in_len; row += (int64_t)parts * blockDim.x) {
      const float xv = (float)WTraits<W>::load(x, row);
      if (xv == 0.f) continue;                                 // a zero of the operand adds nothing
      const unsigned long long fixed_row = jit_fixed_from_f32(xv * (float)p.w0, fx_scale);      // (one shared weight: per row)
      uint32_t state = lr_init(p.seed, (uint32_t)row, chunk, l);
      uint32_t q = lr_initial_q(state, p.cl);
      while (q < qe) {
        if (ONE_PIECE || q >= qb) {
          const uint32_t slot = ONE_PIECE ? q : q - qb;
          if (MODE == MODE_SCALAR) atomicAdd(&acc[slot], fixed_row);
          else atomicAdd(&acc[slot], jit_fixed_from_f32(edge_weight<MODE, float>(p, (uint32_t)row, j0 + S * q) * xv, fx_scale));
        }
        state = lr_next_nz(state);
        q = q + 1u + lr_bounded(state, p.cl - 1u);
      }
    }
  }
  __syncthreads();
  unsigned long long* dst = partial + (int64_t)blockIdx.x * piece_len;
  for (uint32_t i = threadIdx.x; i < piece_len; i += blockDim.x) dst[i] = acc[i];
}

template <typename W>
__global__ void __launch_bounds__(256) k_jit_f_round(const float* __restrict__ img, W* __restrict__ out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    WTraits<W>::store_d(out, i, (double)img[i]);
}

constexpr int kTile = 8;        // columns of a matrix operand per pass (each pass walks the matrix again)

template <int MODE, typename W>
int run_jit_float(const JitP& p, const void* Xv, void* outv, int64_t in_len, int64_t out_len, int64_t n, int gather, void* ws,
                  hipStream_t st) {
  using ACC = typename WTraits<W>::acc;
  const W* X = static_cast<const W*>(Xv);
  W* out = static_cast<W*>(outv);
  if (gather) {
    double* partial = static_cast<double*>(ws);
    const int64_t tpb = 256 / p.stride;
    const dim3 grid((unsigned)gcap(out_len, (int)tpb, 4096), (unsigned)p.n_chunks);
    if (n == 1) {
      hipLaunchKernelGGL((k_jit_f_gather<MODE, W, 1>), grid, dim3(256), 0, st, p, X, n, (int64_t)0, out_len, partial);
      BE_LAUNCH_CHECK();
      hipLaunchKernelGGL((k_jit_f_gather_reduce<MODE, W, 1>), dim3(gcap(out_len, 256, 2048)), dim3(256), 0, st, partial, p.n_chunks,
                         out_len, n, (int64_t)0, p.w0, out);
      BE_LAUNCH_CHECK();
      return BE_OK;
    }
    for (int64_t c0 = 0; c0 < n; c0 += kTile) {
      hipLaunchKernelGGL((k_jit_f_gather<MODE, W, kTile>), grid, dim3(256), 0, st, p, X, n, c0, out_len, partial);
      BE_LAUNCH_CHECK();
      hipLaunchKernelGGL((k_jit_f_gather_reduce<MODE, W, kTile>), dim3(gcap(out_len * kTile, 256, 2048)), dim3(256), 0, st, partial,
                         p.n_chunks, out_len, n, c0, p.w0, out);
      BE_LAUNCH_CHECK();
    }
    return BE_OK;
  }
  ACC* img = sizeof(W) == 2 ? static_cast<ACC*>(ws) : reinterpret_cast<ACC*>(outv);
  BE_HIP(be_fill_async(img, 0, (size_t)out_len * (size_t)n * sizeof(ACC), st));
  const int64_t tasks = in_len * (int64_t)p.n_chunks * p.stride;
  const dim3 grid((unsigned)gcap(tasks, 256, 256 * 32));
  if (n == 1) {
    hipLaunchKernelGGL((k_jit_f_scatter<MODE, W, 1>), grid, dim3(256), 0, st, p, X, n, (int64_t)0, in_len, img);
    BE_LAUNCH_CHECK();
  } else {
    for (int64_t c0 = 0; c0 < n; c0 += kTile) {
      hipLaunchKernelGGL((k_jit_f_scatter<MODE, W, kTile>), grid, dim3(256), 0, st, p, X, n, c0, in_len, img);
      BE_LAUNCH_CHECK();
    }
  }
  if (sizeof(W) == 2) {
    hipLaunchKernelGGL((k_jit_f_round<W>), dim3(gcap(out_len * n, 256, 2048)), dim3(256), 0, st, reinterpret_cast<const float*>(img), out,
                       out_len * n);
    BE_LAUNCH_CHECK();
  }
  return BE_OK;
}

template <int MODE>
int dispatch_jit_float(const JitP& p, int wdtype, const void* X, void* out, int64_t in_len, int64_t out_len, int64_t n, int gather,
                       void* ws, hipStream_t st) {
  switch (wdtype) {
    case BE_F32: return run_jit_float<MODE, float>(p, X, out, in_len, out_len, n, gather, ws, st);
    case BE_F64: return run_jit_float<MODE, double>(p, X, out, in_len, out_len, n, gather, ws, st);
    case BE_F16: return run_jit_float<MODE, __half>(p, X, out, in_len, out_len, n, gather, ws, st);
    case BE_BF16: return run_jit_float<MODE, __hip_bfloat16>(p, X, out, in_len, out_len, n, gather, ws, st);
    default: be_set_error("unknown weight dtype"); return BE_ERR_INVALID;
  }
}

template <int MODE, typename W>
int run_jit_float_scatter_lds(const JitP& p, const void* X_bm, void* out_bm, int64_t in_len, int64_t n, int scale_exp, void* ws,
                              hipStream_t st) {
  const ScatterGeom g = scatter_geom(p, /*scalar=*/false, n);
  const size_t lds = (size_t)g.piece_len * 8;
  const float fx_scale = ldexpf(1.0f, scale_exp - 32);
  unsigned long long* partial = static_cast<unsigned long long*>(ws);
  const dim3 sgrid((unsigned)(g.n_classes * g.pieces * g.parts), (unsigned)n);
  if (g.pieces == 1) {
    auto kern = k_jit_f_scatter_lds<MODE, W, true>;
    BE_HIP(be_allow_lds(reinterpret_cast<const void*>(kern), (int)lds));
    hipLaunchKernelGGL(kern, sgrid, dim3(1024), lds, st, p, static_cast<const W*>(X_bm), in_len, g.pieces, g.parts, g.piece_len,
                       fx_scale, partial);
  } else {
    auto kern = k_jit_f_scatter_lds<MODE, W, false>;
    BE_HIP(be_allow_lds(reinterpret_cast<const void*>(kern), (int)lds));
    hipLaunchKernelGGL(kern, sgrid, dim3(1024), lds, st, p, static_cast<const W*>(X_bm), in_len, g.pieces, g.parts, g.piece_len,
                       fx_scale, partial);
  }
  BE_LAUNCH_CHECK();
  const int64_t q_per_chunk = (std::min<int64_t>(p.chunk_size, p.walk_len) + p.stride - 1) / p.stride;
  const dim3 rgrid((unsigned)((q_per_chunk + 255) / 256), (unsigned)p.n_chunks, (unsigned)n);
  const int64_t pstride = (int64_t)g.n_classes * g.pieces * g.parts * g.piece_len;
  // (the 64-bit fixed-point branch of the shared reduce kernel, whatever the family: MODE_UNIFORM selects it)
  hipLaunchKernelGGL((k_jit_scatter_reduce<MODE_UNIFORM, W>), rgrid, dim3(256), 0, st, partial, p, g.pieces, g.parts, g.piece_len,
                     ldexp(1.0, -scale_exp), static_cast<W*>(out_bm), pstride);
  BE_LAUNCH_CHECK();
  return BE_OK;
}

}  // namespace

extern "C" {

int64_t be_jitmm_float_scatter_workspace_bytes(int64_t shape1, int64_t out_len, int64_t n, int stride) {
  const JitP p = make_params(shape1, out_len, 0, 2, stride == 4 ? 4 : 32, 0, 0);
  const ScatterGeom g = scatter_geom(p, false, std::max<int64_t>(1, n));
  return be_align_up(std::max<int64_t>(1, n) * (int64_t)g.n_classes * g.pieces * g.parts * g.piece_len * 8, 256);
}

// The scatter orientation (generator rows = inputs) through LDS fixed-point sums: X_bm [n, in_len] -> out_bm [n, out_len], both
// batch-major.  scale_exp: |w|max * |x|max * in_len * 2^scale_exp < 2^62 (the caller knows the operand's largest magnitude).
int be_jitmm_float_scatter(int mode, double w0, double w1, int wdtype, int64_t clen, uint32_t seed, const void* X_bm, void* out_bm,
                           int64_t shape1, int64_t in_len, int64_t out_len, int64_t n, int stride, int scale_exp, void* workspace,
                           int64_t workspace_bytes, be_stream_t stream) {
  BE_REQUIRE(mode >= 0 && mode <= 2, BE_ERR_INVALID, "mode must be 0 (scalar), 1 (uniform) or 2 (normal)");
  BE_REQUIRE(in_len >= 0 && out_len >= 0 && shape1 >= 0 && n >= 1 && n <= 65535, BE_ERR_INVALID, "bad shape");
  BE_REQUIRE(stride == 32 || stride == 4, BE_ERR_INVALID, "stride must be 32 or 4");
  BE_REQUIRE(in_len < (1ll << 32) && out_len < (1ll << 32), BE_ERR_RANGE, "dimensions must fit uint32 for the RNG keys");
  BE_REQUIRE(scale_exp - 32 > -126 && scale_exp - 32 < 127, BE_ERR_INVALID, "scale_exp out of range");
  if (out_len == 0) return BE_OK;
  BE_REQUIRE(out_bm != nullptr, BE_ERR_INVALID, "out is NULL");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const size_t esz = wdtype == BE_F64 ? 8 : (wdtype == BE_F32 ? 4 : 2);
  if (in_len == 0 || clen <= 0) {
    BE_HIP(be_fill_async(out_bm, 0, (size_t)out_len * (size_t)n * esz, st));
    return BE_OK;
  }
  BE_REQUIRE(X_bm != nullptr, BE_ERR_INVALID, "operand is NULL");
  BE_REQUIRE(workspace != nullptr && workspace_bytes >= be_jitmm_float_scatter_workspace_bytes(shape1, out_len, n, stride),
             BE_ERR_WORKSPACE, "workspace too small");
  const JitP p = make_params(shape1, out_len, seed, clen, stride, w0, w1);
#define BE_JFS(MODE_)                                                                                                            \
  switch (wdtype) {                                                                                                             \
    case BE_F32: return run_jit_float_scatter_lds<MODE_, float>(p, X_bm, out_bm, in_len, n, scale_exp, workspace, st);          \
    case BE_F64: return run_jit_float_scatter_lds<MODE_, double>(p, X_bm, out_bm, in_len, n, scale_exp, workspace, st);         \
    case BE_F16: return run_jit_float_scatter_lds<MODE_, __half>(p, X_bm, out_bm, in_len, n, scale_exp, workspace, st);         \
    case BE_BF16: return run_jit_float_scatter_lds<MODE_, __hip_bfloat16>(p, X_bm, out_bm, in_len, n, scale_exp, workspace, st); \
    default: be_set_error("unknown weight dtype"); return BE_ERR_INVALID;                                                       \
  }
  if (mode == MODE_SCALAR) { BE_JFS(MODE_SCALAR) }
  else if (mode == MODE_UNIFORM) { BE_JFS(MODE_UNIFORM) }
  else { BE_JFS(MODE_NORMAL) }
#undef BE_JFS
}

int64_t be_jitmm_float_workspace_bytes(int64_t shape1, int64_t in_len, int64_t out_len, int64_t n, int gather, int wdtype) {
  if (gather) {
    const int64_t chunk = std::max<int64_t>(1, (shape1 + 3) / 4);
    const int64_t n_chunks = std::max<int64_t>(1, (in_len + chunk - 1) / chunk);
    return be_align_up(n_chunks * std::max<int64_t>(1, out_len) * (n == 1 ? 1 : kTile) * 8, 256);
  }
  if (wdtype == BE_F16 || wdtype == BE_BF16) return be_align_up(std::max<int64_t>(1, out_len) * std::max<int64_t>(1, n) * 4, 256);
  return 256;
}

int be_jitmm_float(int mode, double w0, double w1, int wdtype, int64_t clen, uint32_t seed, const void* X, void* out, int64_t shape1,
                   int64_t in_len, int64_t out_len, int64_t n, int stride, int gather, void* workspace, int64_t workspace_bytes,
                   be_stream_t stream) {
  BE_REQUIRE(mode >= 0 && mode <= 2, BE_ERR_INVALID, "mode must be 0 (scalar), 1 (uniform) or 2 (normal)");
  BE_REQUIRE(in_len >= 0 && out_len >= 0 && shape1 >= 0 && n >= 1, BE_ERR_INVALID, "bad shape");
  BE_REQUIRE(stride == 32 || stride == 4, BE_ERR_INVALID, "stride must be 32 (the mv matrix) or 4 (the mm matrix)");
  BE_REQUIRE(in_len < (1ll << 32) && out_len < (1ll << 32), BE_ERR_RANGE, "dimensions must fit uint32 for the RNG keys");
  if (out_len == 0) return BE_OK;
  BE_REQUIRE(out != nullptr, BE_ERR_INVALID, "out is NULL");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const size_t esz = wdtype == BE_F64 ? 8 : (wdtype == BE_F32 ? 4 : 2);
  if (in_len == 0 || clen <= 0) {      // empty walk or prob == 0: all zeros (as the event-driven twins)
    BE_HIP(be_fill_async(out, 0, (size_t)out_len * (size_t)n * esz, st));
    return BE_OK;
  }
  BE_REQUIRE(X != nullptr, BE_ERR_INVALID, "operand is NULL");
  BE_REQUIRE(workspace != nullptr && workspace_bytes >= be_jitmm_float_workspace_bytes(shape1, in_len, out_len, n, gather, wdtype),
             BE_ERR_WORKSPACE, "workspace too small");
  const JitP p = make_params(shape1, gather ? in_len : out_len, seed, clen, stride, w0, w1);
  switch (mode) {
    case MODE_SCALAR: return dispatch_jit_float<MODE_SCALAR>(p, wdtype, X, out, in_len, out_len, n, gather, workspace, st);
    case MODE_UNIFORM: return dispatch_jit_float<MODE_UNIFORM>(p, wdtype, X, out, in_len, out_len, n, gather, workspace, st);
    default: return dispatch_jit_float<MODE_NORMAL>(p, wdtype, X, out, in_len, out_len, n, gather, workspace, st);
  }
}

int be_jitmv_float(int mode, double w0, double w1, int wdtype, int64_t clen, uint32_t seed, const void* v, void* out, int64_t shape1,
                   int64_t in_len, int64_t out_len, int gather, void* workspace, int64_t workspace_bytes, be_stream_t stream) {
  return be_jitmm_float(mode, w0, w1, wdtype, clen, seed, v, out, shape1, in_len, out_len, 1, 32, gather, workspace, workspace_bytes,
                        stream);
}

}  // extern "C"
