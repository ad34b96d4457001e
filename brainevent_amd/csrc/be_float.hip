// be_float.hip — float-operand twins of the CSR / fixed-number products for gfx950 (last clause): the same
// matrices against a dense vector or matrix instead of a spike vector.
//   reference: brainevent/_csr/float.py:49-150 (csrmv), :559-668 (csrmm), CPU loops :153-207 / :670-744;
//              brainevent/_fcn/float.py:33-134 (fcnmv), :136-240 (fcnmm) — the same loops over rows of one length.
//   transpose = 0:  out[i, c] = sum_j w_j * B[indices[j], c]        j over row i       (gather: one writer per output row)
//   transpose = 1:  out[indices[j], c] += w_j * B[i, c]             for every row i    (scatter: float atomics — every row
//                                                                                       contributes, nothing to skip but zeros)
// Not event-driven by nature (every element of the operand counts), so the bound is the matrix stream: 4 B of index
// (+ the weight) per entry.  Rows are walked in aligned groups of four entries — one 16-byte load of indices and one of
// weights per lane, entries outside the row masked — by 4, 16 or 64 lanes per row depending on the average row length.
#include "be_csr_shared.h"
#include <algorithm>

namespace {

template <typename W> struct Vec4;      // four consecutive weights as one aligned load
template <> struct Vec4<float> {
  __device__ static __forceinline__ void load(const float* p, int64_t j, float (&o)[4]) {
    const float4 v = *reinterpret_cast<const float4*>(p + j);
    o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
  }
};
template <> struct Vec4<double> {
  __device__ static __forceinline__ void load(const double* p, int64_t j, double (&o)[4]) {
    const double2 a = *reinterpret_cast<const double2*>(p + j), b = *reinterpret_cast<const double2*>(p + j + 2);
    o[0] = a.x; o[1] = a.y; o[2] = b.x; o[3] = b.y;
  }
};
template <> struct Vec4<__half> {
  __device__ static __forceinline__ void load(const __half* p, int64_t j, float (&o)[4]) {
    const uint2 v = *reinterpret_cast<const uint2*>(p + j);
    const __half2 a = *reinterpret_cast<const __half2*>(&v.x), b = *reinterpret_cast<const __half2*>(&v.y);
    o[0] = __low2float(a); o[1] = __high2float(a); o[2] = __low2float(b); o[3] = __high2float(b);
  }
};
template <> struct Vec4<__hip_bfloat16> {
  __device__ static __forceinline__ void load(const __hip_bfloat16* p, int64_t j, float (&o)[4]) {
    const uint2 v = *reinterpret_cast<const uint2*>(p + j);
    o[0] = __uint_as_float(v.x << 16); o[1] = __uint_as_float(v.x & 0xffff0000u);
    o[2] = __uint_as_float(v.y << 16); o[3] = __uint_as_float(v.y & 0xffff0000u);
  }
};

// One aligned group of four entries [4g, 4g + 4) cut to the row [b, e): columns, weights (HOMO: none) and a validity mask.
// The group may reach past the end of the arrays only in the array's last group: read entry by entry there.
template <typename W, bool HOMO>
__device__ __forceinline__ uint32_t load_group(const W* __restrict__ weights, const int32_t* __restrict__ indices, int64_t g,
                                               int64_t b, int64_t e, int64_t nnz, int32_t (&col)[4],
                                               typename WTraits<W>::acc (&w)[4]) {
  const int64_t j0 = g << 2;
  uint32_t ok = 0;
#pragma unroll
  for (int q = 0; q < 4; ++q) ok |= (j0 + q >= b && j0 + q < e ? 1u : 0u) << q;
  if (j0 + 4 <= nnz) {
    const int4 c = *reinterpret_cast<const int4*>(indices + j0);
    col[0] = c.x; col[1] = c.y; col[2] = c.z; col[3] = c.w;
    if (!HOMO) Vec4<W>::load(weights, j0, w);
  } else {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const bool in = j0 + q < nnz;
      col[q] = in ? indices[j0 + q] : 0;
      if (!HOMO) w[q] = in ? (typename WTraits<W>::acc)WTraits<W>::load(weights, j0 + q) : (typename WTraits<W>::acc)0;
    }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) col[q] = (ok >> q) & 1u ? col[q] : 0;       // a masked entry reads operand row 0 and adds nothing
  return ok;
}

// ---------------------------------------------------------------- gather, single vector: LPR lanes per row
template <typename W, bool HOMO, int LPR>
__global__ void __launch_bounds__(256) k_fcsrmv_nt(const W* __restrict__ weights, const int32_t* __restrict__ indices, RowPtr rp,
                                                    const W* __restrict__ v, W* __restrict__ out, int64_t m) {
  using ACC = typename WTraits<W>::acc;
  const int sub = threadIdx.x % LPR;
  const int64_t groups = (int64_t)gridDim.x * (256 / LPR);
  const int64_t nnz = rp.at(m);
  const ACC w0 = HOMO ? (ACC)WTraits<W>::load(weights, 0) : ACC(0);
  for (int64_t r0 = (int64_t)blockIdx.x * (256 / LPR); r0 < m; r0 += groups) {      // (whole waves stay in the loop: shuffles below)
    const int64_t row = r0 + threadIdx.x / LPR;
    int64_t b = 0, e = 0;
    if (row < m) { b = rp.at(row); e = rp.at(row + 1); }
    ACC acc = ACC(0);
    const int64_t g_end = (e + 3) >> 2;
    for (int64_t g = (b >> 2) + sub; g < g_end; g += 2 * LPR) {                     // two groups in flight per lane
      int32_t c0[4], c1[4];
      ACC w_0[4], w_1[4];
      const bool second = g + LPR < g_end;
      const uint32_t ok0 = load_group<W, HOMO>(weights, indices, g, b, e, nnz, c0, w_0);
      const uint32_t ok1 = second ? load_group<W, HOMO>(weights, indices, g + LPR, b, e, nnz, c1, w_1) : 0u;
      ACC x0[4], x1[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) x0[q] = (ACC)WTraits<W>::load(v, c0[q]);
#pragma unroll
      for (int q = 0; q < 4; ++q) x1[q] = second ? (ACC)WTraits<W>::load(v, c1[q]) : ACC(0);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if ((ok0 >> q) & 1u) acc += HOMO ? x0[q] : w_0[q] * x0[q];
        if ((ok1 >> q) & 1u) acc += HOMO ? x1[q] : w_1[q] * x1[q];
      }
    }
#pragma unroll
    for (int off = LPR / 2; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (sub == 0 && row < m) WTraits<W>::store_d(out, row, (double)(HOMO ? acc * w0 : acc));
  }
}

// ---------------------------------------------------------------- gather, matrix operand B [k, n] -> out [m, n]
// A wave owns a row; CPG lanes (a power of two >= the columns of this launch's tile, <= 64) span the columns and 64 / CPG
// entries of the row are taken per step.  B rows are contiguous: one entry's operand row is one coalesced read.
template <typename W, bool HOMO, int CPG>
__global__ void __launch_bounds__(256) k_fcsrmm_nt(const W* __restrict__ weights, const int32_t* __restrict__ indices, RowPtr rp,
                                                    const W* __restrict__ B, W* __restrict__ out, int64_t m, int64_t n) {
  using ACC = typename WTraits<W>::acc;
  constexpr int EPS = 64 / CPG;                       // entries per step
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int cl = lane % CPG, el = lane / CPG;
  const int64_t c = (int64_t)blockIdx.y * CPG + cl;
  const bool c_ok = c < n;
  const ACC w0 = HOMO ? (ACC)WTraits<W>::load(weights, 0) : ACC(0);
  for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < m; row += (int64_t)gridDim.x * 4) {
    const int64_t b = rp.at(row), e = rp.at(row + 1);
    ACC acc = ACC(0);
    for (int64_t j = b + el; j < e; j += 2 * EPS) {
      const bool second = j + EPS < e;
      const int64_t i0 = indices[j], i1 = second ? indices[j + EPS] : 0;
      const ACC a0 = HOMO ? ACC(1) : (ACC)WTraits<W>::load(weights, j);
      const ACC a1 = HOMO || !second ? ACC(second ? 1 : 0) : (ACC)WTraits<W>::load(weights, j + EPS);
      const ACC x0 = c_ok ? (ACC)WTraits<W>::load(B, i0 * n + c) : ACC(0);
      const ACC x1 = c_ok && second ? (ACC)WTraits<W>::load(B, i1 * n + c) : ACC(0);
      acc += a0 * x0;
      acc += a1 * x1;
    }
#pragma unroll
    for (int off = 32; off >= CPG; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (el == 0 && c_ok) WTraits<W>::store_d(out, row * n + c, (double)(HOMO ? acc * w0 : acc));
  }
}

// ---------------------------------------------------------------- scatter (transpose): float atomics into an f32 / f64 image
template <typename T> __device__ __forceinline__ void atomic_add_img(T* p, T x) { atomicAdd(p, x); }

template <typename W, bool HOMO, int LPR>
__global__ void __launch_bounds__(256) k_fcsrmv_t(const W* __restrict__ weights, const int32_t* __restrict__ indices, RowPtr rp,
                                                   const W* __restrict__ v, typename WTraits<W>::acc* __restrict__ img, int64_t m) {
  using ACC = typename WTraits<W>::acc;
  const int sub = threadIdx.x % LPR;
  const int64_t groups = (int64_t)gridDim.x * (256 / LPR);
  const int64_t nnz = rp.at(m);
  const ACC w0 = HOMO ? (ACC)WTraits<W>::load(weights, 0) : ACC(0);
  for (int64_t row = (int64_t)blockIdx.x * (256 / LPR) + threadIdx.x / LPR; row < m; row += groups) {
    const ACC x = (ACC)WTraits<W>::load(v, row);
    if (x == ACC(0)) continue;                                      // a zero of the operand adds nothing
    const int64_t b = rp.at(row), e = rp.at(row + 1);
    const ACC xs = HOMO ? x * w0 : x;
    for (int64_t g = (b >> 2) + sub; g < ((e + 3) >> 2); g += LPR) {
      int32_t c[4];
      ACC w[4];
      const uint32_t ok = load_group<W, HOMO>(weights, indices, g, b, e, nnz, c, w);
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if ((ok >> q) & 1u) atomic_add_img(img + c[q], HOMO ? xs : w[q] * xs);
    }
  }
}

// out [k, n] += w_j * B[i, :]: a wave per row, CPG lanes over the columns, 64 / CPG entries per step; one entry's atomics are
// CPG contiguous elements (the shape the memory-side atomic units take best)
template <typename W, bool HOMO, int CPG>
__global__ void __launch_bounds__(256) k_fcsrmm_t(const W* __restrict__ weights, const int32_t* __restrict__ indices, RowPtr rp,
                                                   const W* __restrict__ B, typename WTraits<W>::acc* __restrict__ img, int64_t m,
                                                   int64_t n) {
  using ACC = typename WTraits<W>::acc;
  constexpr int EPS = 64 / CPG;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int cl = lane % CPG, el = lane / CPG;
  const int64_t c = (int64_t)blockIdx.y * CPG + cl;
  if (c >= n) return;
  const ACC w0 = HOMO ? (ACC)WTraits<W>::load(weights, 0) : ACC(0);
  for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < m; row += (int64_t)gridDim.x * 4) {
    const ACC x = (ACC)WTraits<W>::load(B, row * n + c);
    if (x == ACC(0)) continue;
    const int64_t b = rp.at(row), e = rp.at(row + 1);
    const ACC xs = HOMO ? x * w0 : x;
    for (int64_t j = b + el; j < e; j += EPS) {
      const int64_t i = indices[j];
      atomic_add_img(img + i * n + c, HOMO ? xs : (ACC)WTraits<W>::load(weights, j) * xs);
    }
  }
}

template <typename W>
__global__ void __launch_bounds__(256) k_img_round(const float* __restrict__ img, W* __restrict__ out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    WTraits<W>::store_d(out, i, (double)img[i]);
}

template <typename W> constexpr bool needs_image() { return sizeof(W) == 2; }      // f16 / bf16 outputs accumulate in an f32 image

template <typename W, bool HOMO>
int run_float_csr(const void* weights, const int32_t* indices, RowPtr rp, const void* B, void* out, int64_t m, int64_t k, int64_t n,
                  int64_t avg_row, int transpose, void* ws, hipStream_t st) {
  using ACC = typename WTraits<W>::acc;
  const W* w = static_cast<const W*>(weights);
  const W* b = static_cast<const W*>(B);
  W* o = static_cast<W*>(out);
  if (!transpose) {
    if (n == 1) {
#define BE_F_NT(LPR_) hipLaunchKernelGGL((k_fcsrmv_nt<W, HOMO, LPR_>), dim3(grid_for(m, 256 / LPR_, 256 * 16)), dim3(256), 0, st, w, indices, rp, b, o, m)
      if (avg_row <= 24) BE_F_NT(4); else if (avg_row <= 160) BE_F_NT(16); else BE_F_NT(64);
#undef BE_F_NT
    } else {
#define BE_F_NTM(CPG_) hipLaunchKernelGGL((k_fcsrmm_nt<W, HOMO, CPG_>), dim3(grid_for(m, 4, 256 * 16), (unsigned)((n + CPG_ - 1) / CPG_)), dim3(256), 0, st, w, indices, rp, b, o, m, n)
      if (n <= 2) BE_F_NTM(2); else if (n <= 4) BE_F_NTM(4); else if (n <= 8) BE_F_NTM(8); else if (n <= 16) BE_F_NTM(16);
      else if (n <= 32) BE_F_NTM(32); else BE_F_NTM(64);
#undef BE_F_NTM
    }
    BE_LAUNCH_CHECK();
    return BE_OK;
  }
  ACC* img = needs_image<W>() ? static_cast<ACC*>(ws) : reinterpret_cast<ACC*>(out);
  BE_HIP(be_fill_async(img, 0, (size_t)k * (size_t)n * sizeof(ACC), st));
  if (n == 1) {
#define BE_F_T(LPR_) hipLaunchKernelGGL((k_fcsrmv_t<W, HOMO, LPR_>), dim3(grid_for(m, 256 / LPR_, 256 * 16)), dim3(256), 0, st, w, indices, rp, b, img, m)
    if (avg_row <= 24) BE_F_T(4); else if (avg_row <= 160) BE_F_T(16); else BE_F_T(64);
#undef BE_F_T
  } else {
#define BE_F_TM(CPG_) hipLaunchKernelGGL((k_fcsrmm_t<W, HOMO, CPG_>), dim3(grid_for(m, 4, 256 * 16), (unsigned)((n + CPG_ - 1) / CPG_)), dim3(256), 0, st, w, indices, rp, b, img, m, n)
    if (n <= 2) BE_F_TM(2); else if (n <= 4) BE_F_TM(4); else if (n <= 8) BE_F_TM(8); else if (n <= 16) BE_F_TM(16);
    else if (n <= 32) BE_F_TM(32); else BE_F_TM(64);
#undef BE_F_TM
  }
  BE_LAUNCH_CHECK();
  if (needs_image<W>()) {
    hipLaunchKernelGGL((k_img_round<W>), dim3(grid_for(k * n, 256, 2048)), dim3(256), 0, st, reinterpret_cast<const float*>(img), o, k * n);
    BE_LAUNCH_CHECK();
  }
  return BE_OK;
}

}  // namespace

extern "C" {

int64_t be_csrmm_workspace_bytes(int64_t m, int64_t k, int64_t n, int transpose, int wdtype) {
  (void)m;
  if (!transpose || (wdtype != BE_F16 && wdtype != BE_BF16)) return 256;
  return be_align_up(std::max<int64_t>(1, k) * std::max<int64_t>(1, n) * 4, 256);
}

int be_csrmm(const void* weights, int homo, int wdtype, const int32_t* indices, const void* indptr, int indptr_is_i64, int64_t row_len,
             const void* B, void* out, int64_t m, int64_t k, int64_t n, int64_t nnz_hint, int transpose, void* workspace,
             int64_t workspace_bytes, be_stream_t stream) {
  BE_REQUIRE(m >= 0 && k >= 0 && n >= 1 && m < (1ll << 31) && k < (1ll << 31), BE_ERR_INVALID, "bad shape");
  BE_REQUIRE(check_rows(indptr, row_len), BE_ERR_INVALID, "indptr is NULL and row_len < 0");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int64_t out_rows = transpose ? k : m;
  if (out_rows == 0) return BE_OK;
  BE_REQUIRE(out != nullptr, BE_ERR_INVALID, "out is NULL");
  const size_t esz = wdtype == BE_F64 ? 8 : (wdtype == BE_F32 ? 4 : 2);
  if (m == 0 || (transpose ? m : k) == 0) {             // nothing to sum: zeros
    BE_HIP(be_fill_async(out, 0, (size_t)out_rows * (size_t)n * esz, st));
    return BE_OK;
  }
  BE_REQUIRE(weights && indices && B, BE_ERR_INVALID, "null pointer");
  BE_REQUIRE(workspace_bytes >= be_csrmm_workspace_bytes(m, k, n, transpose, wdtype) && (workspace != nullptr || workspace_bytes == 0),
             BE_ERR_WORKSPACE, "workspace too small");
  RowPtr rp{indptr, indptr_is_i64, row_len};
  const int64_t avg = indptr == nullptr ? row_len : (nnz_hint > 0 ? nnz_hint / m : 64);
  BE_DISPATCH_W(wdtype, homo, return (run_float_csr<W, HOMO>(weights, indices, rp, B, out, m, k, n, avg, transpose, workspace, st)));
  return BE_OK;
}

int be_csrmv(const void* weights, int homo, int wdtype, const int32_t* indices, const void* indptr, int indptr_is_i64, int64_t row_len,
             const void* v, void* out, int64_t m, int64_t k, int64_t nnz_hint, int transpose, void* workspace, int64_t workspace_bytes,
             be_stream_t stream) {
  return be_csrmm(weights, homo, wdtype, indices, indptr, indptr_is_i64, row_len, v, out, m, k, 1, nnz_hint, transpose, workspace,
                  workspace_bytes, stream);
}

}  // extern "C"
