// be_fcn.hip — fixed-number connectivity (ELL) entry points.
//
// A FixedNumPerPre matrix is a CSR matrix whose rows all hold n_conn entries
// (reference brainevent/_fcn/main.py:781-854: "indices (n_pre, n_conn) = post ids"), so these
// entry points run the CSR kernels of be_csr.hip with an implicit indptr (row r = [r*n_conn, (r+1)*n_conn)).
// They keep the reference's per-variant naming (brainevent/_fcn/binary_fcnmv.cu:207-251,
// brainevent/_fcn/binary_fcnmm.cu:993-1023) so a maintainer can map symbol to symbol.
#include "be_common.h"

extern "C" {

#define BE_DEF_FCN_VARIANT(W, WD, S, SD)                                                                              \
  int be_binary_fcnmv_scatter_homo_##W##_##S(BE_FCN_MV_ARGS) {                                                         \
    return be_binary_csrmm_t(weights, 1, WD, indices, nullptr, 0, n_conn, spikes, SD, out, n_pre, n_post, 1,           \
                             workspace, workspace_bytes, stream);                                                      \
  }                                                                                                                    \
  int be_binary_fcnmv_scatter_hetero_##W##_##S(BE_FCN_MV_ARGS) {                                                       \
    return be_binary_csrmm_t(weights, 0, WD, indices, nullptr, 0, n_conn, spikes, SD, out, n_pre, n_post, 1,           \
                             workspace, workspace_bytes, stream);                                                      \
  }                                                                                                                    \
  int be_binary_fcnmv_gather_homo_##W##_##S(BE_FCN_MV_ARGS) {                                                          \
    return be_binary_csrmm_nt(weights, 1, WD, indices, nullptr, 0, n_conn, spikes, SD, out, n_pre, n_post, 1,          \
                              workspace, workspace_bytes, stream);                                                     \
  }                                                                                                                    \
  int be_binary_fcnmv_gather_hetero_##W##_##S(BE_FCN_MV_ARGS) {                                                        \
    return be_binary_csrmm_nt(weights, 0, WD, indices, nullptr, 0, n_conn, spikes, SD, out, n_pre, n_post, 1,          \
                              workspace, workspace_bytes, stream);                                                     \
  }                                                                                                                    \
  int be_binary_fcnmm_scatter_homo_##W##_##S(BE_FCN_MM_ARGS) {                                                         \
    return be_binary_csrmm_t(weights, 1, WD, indices, nullptr, 0, n_conn, spikes_bm, SD, out_bm, n_pre, n_post,        \
                             n_batch, workspace, workspace_bytes, stream);                                             \
  }                                                                                                                    \
  int be_binary_fcnmm_scatter_hetero_##W##_##S(BE_FCN_MM_ARGS) {                                                       \
    return be_binary_csrmm_t(weights, 0, WD, indices, nullptr, 0, n_conn, spikes_bm, SD, out_bm, n_pre, n_post,        \
                             n_batch, workspace, workspace_bytes, stream);                                             \
  }                                                                                                                    \
  int be_binary_fcnmm_gather_homo_##W##_##S(BE_FCN_MM_ARGS) {                                                          \
    return be_binary_csrmm_nt(weights, 1, WD, indices, nullptr, 0, n_conn, spikes_bm, SD, out_bm, n_pre, n_post,       \
                              n_batch, workspace, workspace_bytes, stream);                                            \
  }                                                                                                                    \
  int be_binary_fcnmm_gather_hetero_##W##_##S(BE_FCN_MM_ARGS) {                                                        \
    return be_binary_csrmm_nt(weights, 0, WD, indices, nullptr, 0, n_conn, spikes_bm, SD, out_bm, n_pre, n_post,       \
                              n_batch, workspace, workspace_bytes, stream);                                            \
  }

BE_FOR_ALL_VARIANTS(BE_DEF_FCN_VARIANT)

}  // extern "C"
