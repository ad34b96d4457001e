/* brainevent_amd.h — C ABI of libbrainevent_amd.so (MI355X / gfx950 spike-triggered SpMV/SpMM engine).
 *
 * This is the drop-in boundary for the ONE hot path of chaobrain/brainevent that this repository
 * accelerates: BinaryArray @ {CSR, CSC, dense, JITC{Scalar,Normal,Uniform}{R,C}, FixedNumConn}.
 *
 * What it replaces in the reference (paths relative to the reference checkout, read as text only):
 *   the `// @BE <name>` native entry points that brainevent/_op/kernix_codegen.py:617-736 wraps into
 *   `extern "C" XLA_FFI_Error* be_<name>(XLA_FFI_CallFrame*)` and that the Python side reaches through
 *   `jax.ffi.ffi_call("<module>.<name>", …)`.  Here the same per-variant naming grammar is kept
 *   (`<op>_<homo|hetero>_<f32|f64|f16|bf16>_<bool|float>`), but the calling convention is a plain C one:
 *   raw device pointers + sizes + an explicit hipStream_t, `int` status return, no XLA/JAX types,
 *   no torch types.  Each declaration cites the reference interface it stands in for.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the parameter name ends in `_host`;
 *   - the caller owns every buffer (the library allocates nothing that outlives a call);
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream); calls are asynchronous
 *     on that stream unless stated otherwise;
 *   - return value 0 = success, negative = error (see BE_ERR_*); `be_last_error()` returns a
 *     thread-local message.  The library never aborts the process;
 *   - bool spikes are any 1-byte integer buffer, active when != 0; float spikes are f32, active when > 0
 *     (reference: brainevent/include/cuda_common.h:120-131);
 *   - indices are int32; indptr is int32 or int64 (`indptr_is_i64`)
 *     (reference: brainevent/include/brainevent/dispatch.h:184-215).
 */
#ifndef BRAINEVENT_AMD_H
#define BRAINEVENT_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BE_OK 0
#define BE_ERR_INVALID (-1)
#define BE_ERR_WORKSPACE (-2)
#define BE_ERR_HIP (-3)
#define BE_ERR_RANGE (-4)
#define BE_ERR_UNSUPPORTED (-5)

/* weight / output dtype codes and spike dtype codes used by the generic entry points */
#define BE_F32 0
#define BE_F64 1
#define BE_F16 2
#define BE_BF16 3
#define BE_SPIKE_BOOL 0
#define BE_SPIKE_FLOAT 1

typedef void* be_stream_t; /* hipStream_t */

/* ------------------------------------------------------------------------------------------------
 * library / runtime
 * ---------------------------------------------------------------------------------------------- */
int be_version(void);                 /* 10000*major + 100*minor + patch */
const char* be_last_error(void);      /* thread-local, valid until the next failing call on this thread */
int be_device_count(void);            /* number of visible HIP devices, or a negative BE_ERR_* */
const char* be_build_arch(void);      /* "gfx950" */
/* HIP-event timing of each op's dominant kernel, recorded on the op's own stream.
 * enable(n) arms n record slots (0 disarms); read() synchronises and returns the number of
 * records copied to ms_host (kernel durations in milliseconds, in call order) and rearms. */
int be_profile_enable(int max_records);
int be_profile_read(float* ms_host, int capacity);

/* ------------------------------------------------------------------------------------------------
 * event vector helpers (replace: brainevent/_jit_scalar/binary_jitsmv.cu:107-125 `_pack_bool_kern`
 * and the active-row extraction of brainevent/_csr/binary_csrmv_hybrid.cu:275-327)
 * ---------------------------------------------------------------------------------------------- */
/* spikes[n] -> bits[ceil(n/32)] (bit i%32 of word i/32 set iff spike i active) */
int be_pack_spikes(const void* spikes, int spike_dtype, int64_t n, uint32_t* bits, be_stream_t stream);
/* spikes[n] -> active_ids[<=n] (unordered) and *count (device uint32) */
int be_compact_spikes(const void* spikes, int spike_dtype, int64_t n, uint32_t* active_ids, uint32_t* count,
                      be_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * binary_csrmv, transpose=True  (scatter):  out[indices[j]] += w[j]  for every active row
 * replaces: binary_csrmv_wat_hybrid_{homo,hetero}_{f32,f64,f16,bf16}_{bool,float}
 *           (brainevent/_csr/binary_csrmv_hybrid.cu:619-632, 789-821)
 * and, with indptr == NULL and row_len = n_conn, binary_fcnmv_scatter_{homo,hetero}_bool_{…}
 *           (brainevent/_fcn/binary_fcnmv.cu:55-137, 207-217).
 *   weights : [nnz] (hetero) or [1] (homo), dtype wdtype;  out : [k] dtype wdtype (fully written)
 *   spikes  : [m];  indices : [nnz] int32 in [0,k);  indptr : [m+1] or NULL (then rows are row_len long)
 *   workspace : >= be_binary_csrmv_t_workspace_bytes(m, k, wdtype) bytes, 256-byte aligned
 * "direct" route: no preprocessing, global float atomics.
 * ---------------------------------------------------------------------------------------------- */
int64_t be_binary_csrmv_t_workspace_bytes(int64_t m, int64_t k, int wdtype);
int be_binary_csrmv_t(const void* weights, int homo, int wdtype, const int32_t* indices, const void* indptr,
                      int indptr_is_i64, int64_t row_len, const void* spikes, int spike_dtype, void* out,
                      int64_t m, int64_t k, void* workspace, int64_t workspace_bytes, be_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * post-sliced scatter plan (the MI355X-native layout behind `spk @ CSR` / `spk @ FixedNumPerPre`).
 * Plays the role of the reference's per-matrix task workspace
 * (brainevent/_csr/main.py:58-88, brainevent/_csr/hybrid_config.py:298-324): built once per matrix,
 * cached by the CSR object, passed to every call.
 *
 * Layout: output neurons are cut into slices of 2^slice_shift; for slice s and row r the entries of
 * row r whose column falls in slice s are stored contiguously as (uint16 local column, f32 weight),
 * padded to a multiple of 4 entries (pad: local column = 2^slice_shift, weight 0).
 *   seg_ptr[s*m + r] .. seg_ptr[s*m + r + 1]  delimit that segment in units of 4 entries.
 *
 *   step 1  be_scatter_plan_count : fills seg_ptr (n_slices*m + 1 uint32) and returns the total number
 *           of stored entries (multiple of 4) in *total_entries_host.  SYNCHRONOUS (it reads the total back).
 *   step 2  caller allocates idx16[total] (uint16) and, for hetero weights, w32[total] (f32).
 *   step 3  be_scatter_plan_fill  : fills idx16 / w32; writes max |w| as f32 bits to *maxabs_bits (device uint32).
 * ---------------------------------------------------------------------------------------------- */
int64_t be_scatter_plan_scratch_bytes(int64_t m, int64_t k, int slice_shift);
int be_scatter_plan_count(const int32_t* indices, const void* indptr, int indptr_is_i64, int64_t row_len,
                          int64_t m, int64_t k, int slice_shift, uint32_t* seg_ptr, void* scratch,
                          int64_t scratch_bytes, int64_t* total_entries_host, be_stream_t stream);
int be_scatter_plan_fill(const void* weights, int homo, int wdtype, const int32_t* indices, const void* indptr,
                         int indptr_is_i64, int64_t row_len, int64_t m, int64_t k, int slice_shift,
                         const uint32_t* seg_ptr, int64_t total_entries, uint16_t* idx16, float* w32,
                         uint32_t* maxabs_bits, be_stream_t stream);

/* planned scatter step: out[k] (dtype wdtype, fully written) from spikes[m].
 *   weights : device pointer to weights[0] (homo only; may be NULL for hetero)
 *   scale_exp : fixed-point exponent chosen by the caller from max|w| and m (hetero only): every stored
 *               weight is accumulated as round(w * 2^scale_exp) in a 64-bit integer (order independent,
 *               bitwise reproducible); |w|max * 2^scale_exp * m must stay below 2^62.
 *   parts : number of workgroups that share one slice (each takes 1/parts of the active rows)
 *   workspace : >= be_binary_csrmv_t_plan_workspace_bytes(m, k, slice_shift, parts, homo) bytes
 */
int64_t be_binary_csrmv_t_plan_workspace_bytes(int64_t m, int64_t k, int slice_shift, int parts, int homo);
int be_binary_csrmv_t_plan(const void* weights, int homo, int wdtype, const uint16_t* idx16, const float* w32,
                           const uint32_t* seg_ptr, const void* spikes, int spike_dtype, void* out, int64_t m,
                           int64_t k, int slice_shift, int parts, int scale_exp, void* workspace,
                           int64_t workspace_bytes, be_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * binary_csrmv, transpose=False (gather):  out[i] = sum_j w[j] * e(spikes[indices[j]])
 * replaces: binary_csrmv_nt_auto_{homo,hetero}_{…}_{bool,float} (brainevent/_csr/binary_csrmv.cu:437-486)
 *   spikes : [k];  out : [m];  workspace >= be_binary_csrmv_nt_workspace_bytes(m, k)
 * ---------------------------------------------------------------------------------------------- */
int64_t be_binary_csrmv_nt_workspace_bytes(int64_t m, int64_t k);
int be_binary_csrmv_nt(const void* weights, int homo, int wdtype, const int32_t* indices, const void* indptr,
                       int indptr_is_i64, int64_t row_len, const void* spikes, int spike_dtype, void* out,
                       int64_t m, int64_t k, void* workspace, int64_t workspace_bytes, be_stream_t stream);

/* per-variant symbols (same grammar as the reference's `// @BE` names); thin wrappers of the above */
#define BE_DECL_CSRMV_VARIANT(W, WD, S, SD)                                                                   \
  int be_binary_csrmv_t_homo_##W##_##S(const void* weights, const int32_t* indices, const void* indptr,        \
                                       int indptr_is_i64, const void* spikes, void* out, int64_t m, int64_t k, \
                                       void* workspace, int64_t workspace_bytes, be_stream_t stream);          \
  int be_binary_csrmv_t_hetero_##W##_##S(const void* weights, const int32_t* indices, const void* indptr,      \
                                         int indptr_is_i64, const void* spikes, void* out, int64_t m,          \
                                         int64_t k, void* workspace, int64_t workspace_bytes,                  \
                                         be_stream_t stream);                                                  \
  int be_binary_csrmv_nt_homo_##W##_##S(const void* weights, const int32_t* indices, const void* indptr,       \
                                        int indptr_is_i64, const void* spikes, void* out, int64_t m,           \
                                        int64_t k, void* workspace, int64_t workspace_bytes,                   \
                                        be_stream_t stream);                                                   \
  int be_binary_csrmv_nt_hetero_##W##_##S(const void* weights, const int32_t* indices, const void* indptr,     \
                                          int indptr_is_i64, const void* spikes, void* out, int64_t m,         \
                                          int64_t k, void* workspace, int64_t workspace_bytes,                 \
                                          be_stream_t stream);

#define BE_FOR_ALL_VARIANTS(X) \
  X(f32, BE_F32, bool, BE_SPIKE_BOOL)   X(f32, BE_F32, float, BE_SPIKE_FLOAT)   \
  X(f64, BE_F64, bool, BE_SPIKE_BOOL)   X(f64, BE_F64, float, BE_SPIKE_FLOAT)   \
  X(f16, BE_F16, bool, BE_SPIKE_BOOL)   X(f16, BE_F16, float, BE_SPIKE_FLOAT)   \
  X(bf16, BE_BF16, bool, BE_SPIKE_BOOL) X(bf16, BE_BF16, float, BE_SPIKE_FLOAT)

BE_FOR_ALL_VARIANTS(BE_DECL_CSRMV_VARIANT)

#ifdef __cplusplus
}
#endif
#endif /* BRAINEVENT_AMD_H */
