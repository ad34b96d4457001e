// be_fixed_point.hip — the fixed-point exponent of a matrix, chosen by the library (so that a scatter plan or a binned
// workspace can be set up from C alone).  The fast scatter routes accumulate a weight w as round(w * 2^e) in a 64-bit
// integer (be_csr_shared.h: fixed_from_f32); e must be small enough that no output can overflow with every row active,
// and large enough that every output keeps the accuracy of the path.  No counterpart in the reference: its GPU kernels
// add floats with atomics (brainevent/_csr/binary_csrmv_hybrid.cu:199-234), which this chip retires at 21 G/s.
#include "be_csr_shared.h"
#include <climits>
#include <cstring>
#include <cmath>

namespace {

// stats[0] = max |w| bits, stats[1] = smallest non-zero |w| bits (0xffffffff if none); colsum[c] += |w| per entry
template <typename W>
__global__ void __launch_bounds__(256) k_fp_colsum(const W* __restrict__ weights, const int32_t* __restrict__ indices, int64_t nnz,
                                                   float* __restrict__ colsum, uint32_t* __restrict__ stats) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  uint32_t my_max = 0, my_min = 0xffffffffu;
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < nnz; j += stride) {
    const float a = fabsf((float)WTraits<W>::load(weights, j));
    const uint32_t ab = __float_as_uint(a);
    my_max = ab > my_max ? ab : my_max;           // non-negative float bit patterns order like unsigned integers
    if (ab != 0u) my_min = ab < my_min ? ab : my_min;
    if (indices != nullptr && ab != 0u && ab < 0x7f800000u) atomicAdd(&colsum[indices[j]], a);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const uint32_t o = __shfl_down(my_max, off, 64), p = __shfl_down(my_min, off, 64);
    my_max = o > my_max ? o : my_max;
    my_min = p < my_min ? p : my_min;
  }
  if (lane_id() == 0) {
    if (my_max != 0u) atomicMax(&stats[0], my_max);
    if (my_min != 0xffffffffu) atomicMin(&stats[1], my_min);
  }
}

template <typename W>
__global__ void __launch_bounds__(256) k_fp_colmax(const W* __restrict__ weights, const int32_t* __restrict__ indices, int64_t nnz,
                                                   uint32_t* __restrict__ colmax) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < nnz; j += stride) {
    const uint32_t ab = __float_as_uint(fabsf((float)WTraits<W>::load(weights, j)));
    if (ab != 0u) atomicMax(&colmax[indices[j]], ab);
  }
}

// stats[2] = max over the columns of colsum (bits); stats[3] = min over the live columns of colmax (bits)
__global__ void __launch_bounds__(256) k_fp_reduce(const float* __restrict__ colsum, const uint32_t* __restrict__ colmax, int64_t k,
                                                   uint32_t* __restrict__ stats) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  uint32_t mx = 0, mn = 0xffffffffu;
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < k; j += stride) {
    if (colsum) { const uint32_t b = __float_as_uint(colsum[j]); mx = b > mx ? b : mx; }
    if (colmax) { const uint32_t b = colmax[j]; if (b != 0u) mn = b < mn ? b : mn; }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const uint32_t o = __shfl_down(mx, off, 64), p = __shfl_down(mn, off, 64);
    mx = o > mx ? o : mx;
    mn = p < mn ? p : mn;
  }
  if (lane_id() == 0) {
    if (colsum && mx != 0u) atomicMax(&stats[2], mx);
    if (colmax && mn != 0xffffffffu) atomicMin(&stats[3], mn);
  }
}

inline float bits_to_float(uint32_t b) { float f; memcpy(&f, &b, 4); return f; }

}  // namespace

extern "C" {

int64_t be_fixed_point_scratch_bytes(int64_t k) { return 256 + be_align_up((k > 0 ? k : 1) * 4, 256); }

// max |w| and the smallest non-zero |w| of a weight array, as f32 bit patterns (0 / 0xffffffff when there is none): one streaming
// pass, no atomics per entry.  A max at or above 0x7f800000 means inf / nan.  scratch >= 256 bytes.  SYNCHRONOUS.
int be_weight_stats(const void* weights, int wdtype, int64_t n, uint32_t* max_bits_host, uint32_t* min_nonzero_bits_host,
                    void* scratch, int64_t scratch_bytes, be_stream_t stream) {
  BE_REQUIRE(weights && max_bits_host && min_nonzero_bits_host && scratch && scratch_bytes >= 256, BE_ERR_INVALID, "null pointer / scratch");
  BE_REQUIRE(n >= 0, BE_ERR_INVALID, "n < 0");
  hipStream_t st = static_cast<hipStream_t>(stream);
  uint32_t* stats = static_cast<uint32_t*>(scratch);
  BE_HIP(be_fill_async(stats, 0, 16, st));
  BE_HIP(be_fill_async(stats + 1, 0xff, 4, st));
  if (n > 0) {
    const int grid = grid_for(n, 256, 256 * 16);
    const int32_t* no_idx = nullptr;
    float* no_col = nullptr;
    switch (wdtype) {
      case BE_F32: hipLaunchKernelGGL(k_fp_colsum<float>, dim3(grid), dim3(256), 0, st, static_cast<const float*>(weights), no_idx, n, no_col, stats); break;
      case BE_F64: hipLaunchKernelGGL(k_fp_colsum<double>, dim3(grid), dim3(256), 0, st, static_cast<const double*>(weights), no_idx, n, no_col, stats); break;
      case BE_F16: hipLaunchKernelGGL(k_fp_colsum<__half>, dim3(grid), dim3(256), 0, st, static_cast<const __half*>(weights), no_idx, n, no_col, stats); break;
      case BE_BF16: hipLaunchKernelGGL(k_fp_colsum<__hip_bfloat16>, dim3(grid), dim3(256), 0, st, static_cast<const __hip_bfloat16*>(weights), no_idx, n, no_col, stats); break;
      default: be_set_error("unknown weight dtype"); return BE_ERR_INVALID;
    }
    BE_LAUNCH_CHECK();
  }
  uint32_t h[2];
  BE_HIP(hipMemcpyAsync(h, stats, 8, hipMemcpyDeviceToHost, st));
  BE_HIP(hipStreamSynchronize(st));
  *max_bits_host = h[0];
  *min_nonzero_bits_host = h[1];
  return BE_OK;
}

int be_fixed_point_exponent(const void* weights, int wdtype, const int32_t* indices, int64_t nnz, int64_t k, int min_weight_bits,
                            int keep_exp, void* scratch, int64_t scratch_bytes, int* scale_exp_host, be_stream_t stream) {
  BE_REQUIRE(weights && scale_exp_host && scratch, BE_ERR_INVALID, "null pointer");
  BE_REQUIRE(nnz >= 0 && k > 0 && k <= 0xffffffffll, BE_ERR_INVALID, "bad nnz / k");
  BE_REQUIRE(min_weight_bits >= 0 && min_weight_bits <= 60, BE_ERR_INVALID, "min_weight_bits out of range");
  BE_REQUIRE(scratch_bytes >= be_fixed_point_scratch_bytes(k), BE_ERR_WORKSPACE, "scratch too small");
  hipStream_t st = static_cast<hipStream_t>(stream);
  uint32_t* stats = static_cast<uint32_t*>(scratch);                       // 4 words at the head
  float* col = reinterpret_cast<float*>(static_cast<unsigned char*>(scratch) + 256);
  BE_HIP(be_fill_async(stats, 0, 16, st));
  BE_HIP(be_fill_async(stats + 1, 0xff, 4, st));
  BE_HIP(be_fill_async(stats + 3, 0xff, 4, st));
  if (indices) BE_HIP(be_fill_async(col, 0, (size_t)k * 4, st));
  const int grid = grid_for(nnz, 256, 256 * 16);
#define BE_FP_PASS(KERN, ...)                                                                                   \
  switch (wdtype) {                                                                                             \
    case BE_F32: hipLaunchKernelGGL(KERN<float>, dim3(grid), dim3(256), 0, st, static_cast<const float*>(weights), __VA_ARGS__); break;            \
    case BE_F64: hipLaunchKernelGGL(KERN<double>, dim3(grid), dim3(256), 0, st, static_cast<const double*>(weights), __VA_ARGS__); break;          \
    case BE_F16: hipLaunchKernelGGL(KERN<__half>, dim3(grid), dim3(256), 0, st, static_cast<const __half*>(weights), __VA_ARGS__); break;          \
    case BE_BF16: hipLaunchKernelGGL(KERN<__hip_bfloat16>, dim3(grid), dim3(256), 0, st, static_cast<const __hip_bfloat16*>(weights), __VA_ARGS__); break; \
    default: be_set_error("unknown weight dtype"); return BE_ERR_INVALID;                                       \
  }
  if (nnz > 0) {
    BE_FP_PASS(k_fp_colsum, indices, nnz, col, stats)
    BE_LAUNCH_CHECK();
  }
  if (indices) {
    hipLaunchKernelGGL(k_fp_reduce, dim3(grid_for(k, 256, 1024)), dim3(256), 0, st, col, static_cast<const uint32_t*>(nullptr), k, stats);
    BE_LAUNCH_CHECK();
  }
  uint32_t h[4];
  BE_HIP(hipMemcpyAsync(h, stats, 16, hipMemcpyDeviceToHost, st));
  BE_HIP(hipStreamSynchronize(st));
  BE_REQUIRE(h[0] < 0x7f800000u, BE_ERR_RANGE, "weights contain inf / nan: the fixed-point routes do not apply");
  const float wmax = bits_to_float(h[0]);
  // bound on any output with every row active: the largest column sum of |w| (a row may list a column several times,
  // so "rows x max |w|" is not a bound); without the structure, all the weights there are
  double bound = indices ? (double)bits_to_float(h[2]) * 1.001 : (double)wmax * (double)(nnz + 1);
  int eb = 0;
  if (bound > 0) (void)frexp(bound, &eb);                                  // bound < 2^eb
  int need = 62 - eb;
  need = need < -90 ? -90 : (need > 150 ? 150 : need);                     // 2^(e - 32) must be a normal f32
  const bool has_min = h[1] != 0xffffffffu;
  const float wmin = has_min ? bits_to_float(h[1]) : 0.f;
  // accuracy gate: a sum of n weights carries an absolute error below n * 2^-e; accept e when the largest weight of every
  // non-empty output column keeps min_weight_bits bits (cheap sufficient test first: the globally smallest non-zero |w| does)
  bool have_colmax = false;
  float colmax_min = 0.f;
  const int cand[2] = {keep_exp, need};
  for (int c = (keep_exp != INT_MIN && keep_exp <= need) ? 0 : 1; c < 2; ++c) {
    const int e = cand[c];
    const double thr = ldexp(1.0, min_weight_bits - e);
    bool ok = !has_min || (double)wmin >= thr;
    if (!ok && indices) {
      if (!have_colmax) {
        BE_HIP(be_fill_async(col, 0, (size_t)k * 4, st));
        BE_FP_PASS(k_fp_colmax, indices, nnz, reinterpret_cast<uint32_t*>(col))
        BE_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_fp_reduce, dim3(grid_for(k, 256, 1024)), dim3(256), 0, st, static_cast<const float*>(nullptr),
                           reinterpret_cast<const uint32_t*>(col), k, stats);
        BE_LAUNCH_CHECK();
        BE_HIP(hipMemcpyAsync(h, stats, 16, hipMemcpyDeviceToHost, st));
        BE_HIP(hipStreamSynchronize(st));
        have_colmax = true;
        colmax_min = h[3] != 0xffffffffu ? bits_to_float(h[3]) : INFINITY;
      }
      ok = (double)colmax_min >= thr;
    }
    if (ok) {
      *scale_exp_host = e;
      return BE_OK;
    }
  }
#undef BE_FP_PASS
  be_set_error("be_fixed_point_exponent: the dynamic range of the weights (" + std::to_string(wmin) + " .. " + std::to_string(wmax) +
               ") exceeds what 64-bit fixed-point sums resolve; use the direct route");
  return BE_ERR_RANGE;
}

}  // extern "C"
