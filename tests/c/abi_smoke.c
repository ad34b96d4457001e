/* abi_smoke.c — the INTEGRATION.md call sequence from plain C (gcc, no C++, no torch, no Python): what a non-Python
 * binder of the reference (its FFI replaces `jax.ffi.ffi_call("<module>.<fn>")`, brainevent/_op/kernix_runtime.py:161-189)
 * would do with include/brainevent_amd.h.
 *   1. direct route:   be_binary_csrmv_t_hetero_f32_bool            (no preprocessing, global atomics)
 *   2. planned route:  be_scatter_plan_count_ordered -> _fill_ordered -> be_scatter_plan_exponent (checked against
 *                      be_fixed_point_exponent) -> be_binary_csrmv_t_plan
 *   3. weight refresh: be_scatter_plan_refresh_weights_ordered (a gather-copy through the stored order), then step 2's call again
 *   4. gather:         be_binary_csrmv_nt_hetero_f32_bool
 *   5. neuron step:    be_lif_coba_step, be_lif_cuba_step, be_lif_step_scaled_packed; be_diag_stream_read
 *   7. float twin:     be_csrmv in both directions (a dense operand instead of spikes)
 *   8. binned route:   be_binary_csrmv_t_binned_workspace_bytes -> _init -> be_binary_csrmv_t_binned x 2 -> be_binned_workspace_audit / _status
 *   9. JIT scatter:    be_binary_jitmv over a per-call and an armed workspace (be_jit_scatter_workspace_arm / _disarm)
 *   6. the unfavourable direction made event-driven: be_csr_to_csc_count -> _indptr -> _fill_block (two column
 *      blocks, weights moved along, perm kept) gives the CSC mirror; the gather product of step 4 is then (a) the direct scatter
 *      over the mirror, (b) the perm-fused scatter be_binary_csrmm_t_indexed over the mirror's structure with the weights left in
 *      CSR order, (c) after a weight update: be_gather_by_perm + the same scatter
 * Every result is compared with the serial loop of the reference's CPU kernel (brainevent/_csr/binary.py:446-451 /
 * :466-472), restated inline in double precision; tolerance rtol = atol = 1e-5 (the tolerance of the path).
 * Build:  gcc -std=c11 -D__HIP_PLATFORM_AMD__ -I include -I /opt/rocm/include tests/c/abi_smoke.c \
 *             -L brainevent_amd/lib -lbrainevent_amd -L /opt/rocm/lib -lamdhip64 -lm -o abi_smoke
 */
#include <limits.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <hip/hip_runtime_api.h>

#include "brainevent_amd.h"

#define CHECK_BE(call)                                                                      \
  do {                                                                                      \
    int rc_ = (call);                                                                       \
    if (rc_ != BE_OK) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, be_last_error()); return 1; } \
  } while (0)
#define CHECK_HIP(call)                                                                     \
  do {                                                                                      \
    hipError_t e_ = (call);                                                                 \
    if (e_ != hipSuccess) { fprintf(stderr, "%s -> %s\n", #call, hipGetErrorString(e_)); return 1; } \
  } while (0)

static uint32_t rng_state = 12345u;
static uint32_t rnd(void) { rng_state = rng_state * 1664525u + 1013904223u; return rng_state >> 8; }

static void *dev_copy(const void *host, size_t bytes) {
  void *d = NULL;
  if (hipMalloc(&d, bytes ? bytes : 4) != hipSuccess) return NULL;
  if (host && bytes && hipMemcpy(d, host, bytes, hipMemcpyHostToDevice) != hipSuccess) return NULL;
  return d;
}

/* tolerance of the path: |got - ref| <= 1e-5 + 1e-5 |ref| (f32 accumulated currents; the direct route adds f32 atomically
 * in arrival order, so mixed-sign rows carry the cancellation error of an f32 sum — the planned route is exact to the ulp) */
static int compare(const char *what, const float *got, const double *ref, int64_t n) {
  double worst = 0;
  for (int64_t i = 0; i < n; ++i) {
    const double err = fabs((double)got[i] - ref[i]) / (1e-5 + 1e-5 * fabs(ref[i]));
    if (err > worst) worst = err;
  }
  printf("%-28s worst error / tolerance %.3g %s\n", what, worst, worst <= 1.0 ? "ok" : "FAIL");
  return worst <= 1.0 ? 0 : 1;
}

int main(void) {
  const int64_t m = 3000, k = 50000, row = 400, nnz = m * row;
  if (be_device_count() < 1) { fprintf(stderr, "no HIP device: %s\n", be_last_error()); return 2; }
  printf("libbrainevent_amd %d for %s\n", be_version(), be_build_arch());

  int32_t *idx = malloc(nnz * 4), *ptr = malloc((m + 1) * 4);
  float *w = malloc(nnz * 4), *got = malloc((k > m ? k : m) * 4);
  uint8_t *spk = malloc(m), *spk_k = malloc(k);
  double *ref = malloc((k > m ? k : m) * 8);
  for (int64_t i = 0; i <= m; ++i) ptr[i] = (int32_t)(i * row);
  for (int64_t j = 0; j < nnz; ++j) { idx[j] = (int32_t)(rnd() % k); w[j] = (float)(rnd() % 100000) / 100000.f - 0.3f; }
  for (int64_t i = 0; i < m; ++i) spk[i] = (rnd() % 100) < 10;
  for (int64_t i = 0; i < k; ++i) spk_k[i] = (rnd() % 100) < 10;

  void *d_idx = dev_copy(idx, nnz * 4), *d_ptr = dev_copy(ptr, (m + 1) * 4), *d_w = dev_copy(w, nnz * 4);
  void *d_spk = dev_copy(spk, m), *d_spk_k = dev_copy(spk_k, k), *d_out = dev_copy(NULL, (k > m ? k : m) * 4);
  if (!d_idx || !d_ptr || !d_w || !d_spk || !d_spk_k || !d_out) { fprintf(stderr, "hipMalloc / hipMemcpy failed\n"); return 1; }
  int fails = 0;

  /* reference loop, transpose=True: posts[indices[j]] += weights[j] for every active row */
  memset(ref, 0, k * 8);
  for (int64_t i = 0; i < m; ++i)
    if (spk[i]) for (int64_t j = ptr[i]; j < ptr[i + 1]; ++j) ref[idx[j]] += (double)w[j];

  /* 1. direct route */
  int64_t ws_bytes = be_binary_csrmv_t_workspace_bytes(m, k, BE_F32);
  void *d_ws = dev_copy(NULL, ws_bytes);
  CHECK_BE(be_binary_csrmv_t_hetero_f32_bool(d_w, d_idx, d_ptr, 0, d_spk, d_out, m, k, d_ws, ws_bytes, NULL));
  CHECK_HIP(hipMemcpy(got, d_out, k * 4, hipMemcpyDeviceToHost));
  fails += compare("direct scatter", got, ref, k);

  /* 2. planned route */
  const int shift = 14, width = 0, layout = BE_PLAN_D8, parts = 4;
  const int64_t n_slices = (k + (1 << shift) - 1) >> shift;
  void *d_seg = dev_copy(NULL, m * n_slices * 8);
  int64_t scr_bytes = be_scatter_plan_scratch_bytes(m, k, shift, width), blob_bytes = 0;
  void *d_scr = dev_copy(NULL, scr_bytes);
  /* the count pass leaves the rows' column order behind (2 bytes per entry): the fill and the refresh below read it back */
  void *d_order = dev_copy(NULL, nnz * 2);
  CHECK_BE(be_scatter_plan_count_ordered(d_idx, d_ptr, 0, -1, m, k, shift, width, 0, layout, d_seg, d_scr, scr_bytes, &blob_bytes,
                                         (uint16_t *)d_order, NULL));
  void *d_blob = dev_copy(NULL, blob_bytes + 128), *d_maxabs = dev_copy(NULL, 8);
  CHECK_BE(be_scatter_plan_fill_ordered(d_w, 0, BE_F32, d_idx, d_ptr, 0, -1, m, k, shift, width, layout, d_seg, d_blob, d_maxabs,
                                        (const uint16_t *)d_order, NULL));
  int64_t fp_bytes = be_fixed_point_scratch_bytes(k);
  void *d_fp = dev_copy(NULL, fp_bytes);
  int scale_exp = 0, scale_exp_entries = 0;
  /* the exponent from the plan's own blocks, and — for comparison — from the raw entries (the call a binned workspace uses) */
  CHECK_BE(be_scatter_plan_exponent(d_blob, d_seg, m, k, shift, width, layout, nnz, (const uint32_t *)d_maxabs, 16, INT_MIN, d_fp,
                                    fp_bytes, &scale_exp, NULL));
  CHECK_BE(be_fixed_point_exponent(d_w, BE_F32, d_idx, nnz, k, 16, INT_MIN, d_fp, fp_bytes, &scale_exp_entries, NULL));
  if (scale_exp != scale_exp_entries && scale_exp != scale_exp_entries - 1) {
    printf("FAIL exponent from the plan %d vs from the entries %d\n", scale_exp, scale_exp_entries);
    ++fails;
  }
  int64_t pws_bytes = be_binary_csrmv_t_plan_workspace_bytes(m, k, shift, width, parts, 0);
  void *d_pws = dev_copy(NULL, pws_bytes);
  CHECK_HIP(hipMemset(d_pws, 0, 256));                    /* spike counters: zero on entry, zero again on exit */
  const int block_hint = (int)(nnz / (m * n_slices));
  for (int rep = 0; rep < 2; ++rep) {                     /* twice on one workspace: the counters are re-armed by the call */
    CHECK_BE(be_binary_csrmv_t_plan(NULL, 0, BE_F32, d_blob, d_seg, d_spk, BE_SPIKE_BOOL, d_out, m, k, shift, width, layout,
                                    block_hint, parts, scale_exp, d_pws, pws_bytes, NULL));
    CHECK_HIP(hipMemcpy(got, d_out, k * 4, hipMemcpyDeviceToHost));
    fails += compare(rep ? "planned scatter (again)" : "planned scatter", got, ref, k);
  }
  printf("plan: %lld slices, blob %lld bytes, scale_exp %d\n", (long long)n_slices, (long long)blob_bytes, scale_exp);

  /* 3. weights updated in place (plasticity): refresh the blocks of the unchanged structure */
  for (int64_t j = 0; j < nnz; ++j) w[j] = w[j] * 0.5f + 0.01f;
  CHECK_HIP(hipMemcpy(d_w, w, nnz * 4, hipMemcpyHostToDevice));
  CHECK_BE(be_scatter_plan_refresh_weights_ordered(d_w, 0, BE_F32, d_idx, d_ptr, 0, -1, m, k, shift, width, layout, d_seg, d_blob,
                                                   d_maxabs, (const uint16_t *)d_order, NULL));
  CHECK_BE(be_scatter_plan_exponent(d_blob, d_seg, m, k, shift, width, layout, nnz, (const uint32_t *)d_maxabs, 16, scale_exp, d_fp,
                                    fp_bytes, &scale_exp, NULL));
  memset(ref, 0, k * 8);
  for (int64_t i = 0; i < m; ++i)
    if (spk[i]) for (int64_t j = ptr[i]; j < ptr[i + 1]; ++j) ref[idx[j]] += (double)w[j];
  CHECK_BE(be_binary_csrmv_t_plan(NULL, 0, BE_F32, d_blob, d_seg, d_spk, BE_SPIKE_BOOL, d_out, m, k, shift, width, layout,
                                  block_hint, parts, scale_exp, d_pws, pws_bytes, NULL));
  CHECK_HIP(hipMemcpy(got, d_out, k * 4, hipMemcpyDeviceToHost));
  fails += compare("planned, refreshed weights", got, ref, k);

  /* 4. gather direction: out[i] = sum_j w[j] * e(spikes[indices[j]]) */
  for (int64_t i = 0; i < m; ++i) {
    double s = 0;
    for (int64_t j = ptr[i]; j < ptr[i + 1]; ++j) if (spk_k[idx[j]]) s += (double)w[j];
    ref[i] = s;
  }
  int64_t gws_bytes = be_binary_csrmv_nt_workspace_bytes(m, k);
  void *d_gws = dev_copy(NULL, gws_bytes);
  CHECK_BE(be_binary_csrmv_nt_hetero_f32_bool(d_w, d_idx, d_ptr, 0, d_spk_k, d_out, m, k, d_gws, gws_bytes, NULL));
  CHECK_HIP(hipMemcpy(got, d_out, m * 4, hipMemcpyDeviceToHost));
  fails += compare("gather", got, ref, m);

  /* 5. the neuron half of a time step (be_lif_coba_step) on the gathered currents: one neuron far above threshold, one
   *    refractory, the rest at rest — checked against the formulas of the header evaluated here */
  {
    float hv[4] = {-50.5f, -55.f, -60.f, -40.f}, hge[4] = {0.f, 1.f, 0.f, 0.f}, hgi[4] = {0.f, 0.f, 2.f, 0.f}, hrf[4] = {0.f, 0.f, 0.f, 3.f};
    float hin[4] = {50.f, 0.f, 0.f, 0.f}, hzero[4] = {0.f, 0.f, 0.f, 0.f};
    void *dv = dev_copy(hv, 16), *dge = dev_copy(hge, 16), *dgi = dev_copy(hgi, 16), *drf = dev_copy(hrf, 16);
    void *din = dev_copy(hin, 16), *dz = dev_copy(hzero, 16), *dsp = dev_copy(NULL, 4), *dcnt = dev_copy(hzero, 16);
    const double dt = 0.1, de = exp(-dt / 5.0), di = exp(-dt / 10.0);
    CHECK_BE(be_lif_coba_step(dv, dge, dgi, drf, din, dz, dsp, dcnt, 4, dt, 20.0, -60.0, -50.0, -60.0, 5.0, 0.0, -80.0, de, di, 20.0,
                              1e-3, NULL));
    float gv[4], grf[4]; unsigned char gsp[4];
    CHECK_HIP(hipMemcpy(gv, dv, 16, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(grf, drf, 16, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(gsp, dsp, 4, hipMemcpyDeviceToHost));
    double rv[4]; int rs[4];
    for (int i = 0; i < 4; ++i) {
      const double ge = hge[i] * de + hin[i], gi = hgi[i] * di;
      const double dvv = (-(hv[i] + 60.0) + (ge * (0.0 - hv[i]) + gi * (-80.0 - hv[i])) * 1e-3 + 20.0) * (dt / 20.0);
      const int act = hrf[i] <= 0.f;
      const double vn = act ? hv[i] + dvv : hv[i];
      rs[i] = act && vn >= -50.0;
      rv[i] = rs[i] ? -60.0 : vn;
    }
    int bad = 0;
    for (int i = 0; i < 4; ++i) bad += (gsp[i] != rs[i]) || fabs(gv[i] - rv[i]) > 1e-4 || fabs(grf[i] - (rs[i] ? 5.0 : hrf[i] - dt)) > 1e-5;
    printf("%-28s spikes %d%d%d%d %s\n", "neuron step", gsp[0], gsp[1], gsp[2], gsp[3], bad ? "FAIL" : "ok");
    fails += bad ? 1 : 0;
    /* 5b. the current-based twin (be_lif_cuba_step) and the scaled-input form (be_lif_step_scaled_packed with the weights 2 and -3
     *     applied to unit inputs) must agree with each other when the plain call is handed the already weighted inputs */
    float hA[4] = {-50.5f, -55.f, -49.5f, -40.f}, hone[4] = {1.f, 0.f, 2.f, 0.f}, hw_e[4] = {2.f, 0.f, 4.f, 0.f}, hw_i[4] = {-3.f, 0.f, -6.f, 0.f};
    void *dvA = dev_copy(hA, 16), *dvB = dev_copy(hA, 16), *dgeA = dev_copy(hzero, 16), *dgiA = dev_copy(hzero, 16);
    void *dgeB = dev_copy(hzero, 16), *dgiB = dev_copy(hzero, 16), *drA = dev_copy(hzero, 16), *drB = dev_copy(hzero, 16);
    void *done_ = dev_copy(hone, 16), *dwe = dev_copy(hw_e, 16), *dwi = dev_copy(hw_i, 16), *dsA = dev_copy(NULL, 4), *dsB = dev_copy(NULL, 4);
    void *dbits = dev_copy(NULL, 4);
    CHECK_BE(be_lif_cuba_step(dvA, dgeA, dgiA, drA, dwe, dwi, dsA, NULL, 4, dt, 20.0, -49.0, -50.0, -60.0, 5.0, de, di, 20.0, 1.0, NULL));
    CHECK_BE(be_lif_step_scaled_packed(1, dvB, dgeB, dgiB, drB, done_, done_, 2.0, -3.0, dsB, dbits, NULL, 4, dt, 20.0, -49.0, -50.0, -60.0,
                                       5.0, 0.0, 0.0, de, di, 20.0, 1.0, NULL));
    float vA[4], vB[4]; unsigned char sA[4], sB[4]; uint32_t word = 0;
    CHECK_HIP(hipMemcpy(vA, dvA, 16, hipMemcpyDeviceToHost)); CHECK_HIP(hipMemcpy(vB, dvB, 16, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(sA, dsA, 4, hipMemcpyDeviceToHost)); CHECK_HIP(hipMemcpy(sB, dsB, 4, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(&word, dbits, 4, hipMemcpyDeviceToHost));
    int bad2 = 0;
    for (int i = 0; i < 4; ++i) bad2 += vA[i] != vB[i] || sA[i] != sB[i] || ((word >> i) & 1u) != sB[i];
    printf("%-28s spikes %d%d%d%d, words %x %s\n", "current-based / scaled step", sA[0], sA[1], sA[2], sA[3], word, bad2 ? "FAIL" : "ok");
    fails += bad2 ? 1 : 0;
  }
  /* 5c. the read-only streaming rate of the device (the ceiling bench.py reports) */
  {
    const int64_t bytes = 256ll << 20;
    void *buf = dev_copy(NULL, (size_t)bytes), *sink = dev_copy(NULL, 4);
    CHECK_HIP(hipMemset(buf, 1, (size_t)bytes));
    float ms = 0.f;
    CHECK_BE(be_diag_stream_read(buf, bytes, 3, sink, &ms, NULL));
    const double gbps = bytes / (ms * 1e-3) / 1e9;
    printf("%-28s %.0f GB/s %s\n", "read-only stream (256 MiB)", gbps, gbps > 500.0 && gbps < 9000.0 ? "ok" : "FAIL");
    fails += gbps > 500.0 && gbps < 9000.0 ? 0 : 1;
    if (be_diag_stream_read(buf, 8, 1, sink, &ms, NULL) != BE_ERR_INVALID) { printf("FAIL be_diag_stream_read accepted 8 bytes\n"); ++fails; }
    CHECK_HIP(hipFree(buf));
  }

  /* 6. CSR -> CSC mirror from C, then the gather product of step 4 as a scatter over the active columns */
  {
    void *d_counts = dev_copy(NULL, k * 8), *d_cptr = dev_copy(NULL, (k + 1) * 8);
    int64_t cscr = be_csr_to_csc_scratch_bytes(k), total = 0;
    void *d_cscr = dev_copy(NULL, cscr);
    CHECK_BE(be_csr_to_csc_count((const int32_t *)d_idx, nnz, k, (int64_t *)d_counts, NULL));
    CHECK_BE(be_csr_to_csc_indptr((const int64_t *)d_counts, k, d_cptr, 1, &total, d_cscr, cscr, NULL));
    if (total != nnz) { printf("FAIL csr_to_csc total %lld != nnz %lld\n", (long long)total, (long long)nnz); ++fails; }
    int64_t *cptr = malloc((k + 1) * 8);
    CHECK_HIP(hipMemcpy(cptr, d_cptr, (k + 1) * 8, hipMemcpyDeviceToHost));
    void *d_rows = dev_copy(NULL, nnz * 4), *d_wt = dev_copy(NULL, nnz * 4), *d_perm = dev_copy(NULL, nnz * 4);
    void *d_cursor = dev_copy(NULL, k * 8);
    const int64_t cut = k / 3;                               /* two column blocks: [0, cut) and [cut, k) */
    const int64_t lo[2] = {0, cut}, hi[2] = {cut, k};
    for (int b = 0; b < 2; ++b) {
      const int64_t off = cptr[lo[b]];                       /* a block's arrays start at its first column's offset */
      CHECK_BE(be_csr_to_csc_fill_block((const int32_t *)d_idx, d_ptr, 0, -1, m, nnz, lo[b], hi[b], (const int64_t *)d_cptr,
                                        (int64_t *)d_cursor, (int32_t *)d_rows + off, (int32_t *)d_perm + off, 0, d_w, 4,
                                        (float *)d_wt + off, NULL));
    }
    /* (a) direct scatter over the mirror: rows = the k columns, outputs = the m rows; int64 indptr */
    int64_t mws_bytes = be_binary_csrmv_t_workspace_bytes(k, m, BE_F32);
    void *d_mws = dev_copy(NULL, mws_bytes);
    CHECK_BE(be_binary_csrmv_t(d_wt, 0, BE_F32, (const int32_t *)d_rows, d_cptr, 1, -1, d_spk_k, BE_SPIKE_BOOL, d_out, k, m, d_mws,
                               mws_bytes, NULL));
    CHECK_HIP(hipMemcpy(got, d_out, m * 4, hipMemcpyDeviceToHost));
    fails += compare("mirror scatter (moved w)", got, ref, m);
    /* (b) perm-fused: the weights stay in CSR order, slot j reads w[perm[j]] */
    CHECK_BE(be_binary_csrmm_t_indexed(d_w, 0, BE_F32, (const int32_t *)d_rows, d_cptr, 1, -1, d_perm, 0, d_spk_k, BE_SPIKE_BOOL,
                                       d_out, k, m, 1, d_mws, mws_bytes, NULL));
    CHECK_HIP(hipMemcpy(got, d_out, m * 4, hipMemcpyDeviceToHost));
    fails += compare("mirror scatter (perm-fused)", got, ref, m);
    /* (c) weights updated in place: one gather-copy brings the mirror's copy up to date */
    for (int64_t j = 0; j < nnz; ++j) w[j] = -w[j];
    CHECK_HIP(hipMemcpy(d_w, w, nnz * 4, hipMemcpyHostToDevice));
    CHECK_BE(be_gather_by_perm(d_w, 4, d_perm, 0, nnz, d_wt, NULL));
    for (int64_t i = 0; i < m; ++i) ref[i] = -ref[i];
    CHECK_BE(be_binary_csrmv_t(d_wt, 0, BE_F32, (const int32_t *)d_rows, d_cptr, 1, -1, d_spk_k, BE_SPIKE_BOOL, d_out, k, m, d_mws,
                               mws_bytes, NULL));
    CHECK_HIP(hipMemcpy(got, d_out, m * 4, hipMemcpyDeviceToHost));
    fails += compare("mirror after gather_by_perm", got, ref, m);
    free(cptr);
  }

  /* 7. float-operand twin: out[i] = sum_j w[j] * x[indices[j]] (be_csrmv, gather) and its transpose (float atomics) */
  {
    float *x = malloc((k > m ? k : m) * 4);
    for (int64_t i = 0; i < (k > m ? k : m); ++i) x[i] = (float)(rnd() % 2000) / 1000.f - 1.f;
    void *d_x = dev_copy(x, (k > m ? k : m) * 4);
    const int64_t fws_bytes = be_csrmm_workspace_bytes(m, k, 1, 1, BE_F32);
    void *d_fws = dev_copy(NULL, fws_bytes);
    for (int64_t i = 0; i < m; ++i) {
      ref[i] = 0;
      for (int64_t j = ptr[i]; j < ptr[i + 1]; ++j) ref[i] += (double)w[j] * (double)x[idx[j]];
    }
    CHECK_BE(be_csrmv(d_w, 0, BE_F32, (const int32_t *)d_idx, d_ptr, 0, -1, d_x, d_out, m, k, nnz, 0, d_fws, fws_bytes, NULL));
    CHECK_HIP(hipMemcpy(got, d_out, m * 4, hipMemcpyDeviceToHost));
    {   /* sums of 400 mixed-sign products in f32: compare at the scale of the row's terms, like the direct scatter above */
      double worst = 0;
      for (int64_t i = 0; i < m; ++i) { const double e = fabs((double)got[i] - ref[i]) / (1e-4 + 1e-5 * fabs(ref[i])); if (e > worst) worst = e; }
      printf("%-28s worst error / tolerance %.3g %s\n", "float csrmv (gather)", worst, worst <= 1.0 ? "ok" : "FAIL");
      fails += worst <= 1.0 ? 0 : 1;
    }
    memset(ref, 0, k * 8);
    for (int64_t i = 0; i < m; ++i)
      for (int64_t j = ptr[i]; j < ptr[i + 1]; ++j) ref[idx[j]] += (double)w[j] * (double)x[i];
    CHECK_BE(be_csrmv(d_w, 0, BE_F32, (const int32_t *)d_idx, d_ptr, 0, -1, d_x, d_out, m, k, nnz, 1, d_fws, fws_bytes, NULL));
    CHECK_HIP(hipMemcpy(got, d_out, k * 4, hipMemcpyDeviceToHost));
    {
      double worst = 0;
      for (int64_t i = 0; i < k; ++i) { const double e = fabs((double)got[i] - ref[i]) / (1e-4 + 1e-5 * fabs(ref[i])); if (e > worst) worst = e; }
      printf("%-28s worst error / tolerance %.3g %s\n", "float csrmv (scatter)", worst, worst <= 1.0 ? "ok" : "FAIL");
      fails += worst <= 1.0 ? 0 : 1;
    }
    free(x);
  }

  /* 8. binned route (no per-matrix layout): workspace_bytes -> init -> two steps -> be_binned_workspace_status / _audit:
   *    the conservation counters say that every stored entry of the active rows was delivered exactly once */
  {
    const int64_t cap = 1 << 20;
    const int64_t bws_bytes = be_binary_csrmv_t_binned_workspace_bytes(m, k, 16, cap);
    void *d_bws = dev_copy(NULL, bws_bytes);
    int bexp = 0;
    const int64_t fps = be_fixed_point_scratch_bytes(k);
    void *d_fps = dev_copy(NULL, fps);
    CHECK_BE(be_fixed_point_exponent(d_w, BE_F32, (const int32_t *)d_idx, nnz, k, 16, INT_MIN, d_fps, fps, &bexp, NULL));
    CHECK_BE(be_binary_csrmv_t_binned_workspace_init(d_bws, bws_bytes, m, k, 16, cap, NULL));
    memset(ref, 0, k * 8);
    uint64_t expect = 0;
    for (int64_t i = 0; i < m; ++i)
      if (spk[i]) { expect += (uint64_t)(ptr[i + 1] - ptr[i]); for (int64_t j = ptr[i]; j < ptr[i + 1]; ++j) ref[idx[j]] += (double)w[j]; }
    for (int rep = 0; rep < 2; ++rep)
      CHECK_BE(be_binary_csrmv_t_binned(d_w, 0, BE_F32, (const int32_t *)d_idx, d_ptr, 0, -1, d_spk, BE_SPIKE_BOOL, d_out, m, k, 16, cap,
                                        bexp, d_bws, bws_bytes, NULL));
    CHECK_HIP(hipMemcpy(got, d_out, k * 4, hipMemcpyDeviceToHost));
    fails += compare("binned scatter", got, ref, k);
    uint64_t c[4];
    CHECK_BE(be_binned_workspace_audit(d_bws, c, NULL));
    const int conserved = c[0] == 2 * expect && c[1] == c[0] && c[2] + c[3] == c[1];
    printf("%-28s %llu entries, %llu tickets, %llu accumulated + %llu overflowed %s\n", "binned conservation", (unsigned long long)c[0],
           (unsigned long long)c[1], (unsigned long long)c[2], (unsigned long long)c[3], conserved ? "ok" : "FAIL");
    fails += conserved ? 0 : 1;
    CHECK_BE(be_binned_workspace_status(d_bws, 1, NULL));
  }

  /* 9. JIT connectivity, scatter orientation: a per-call workspace (the library zeroes its counters) and an ARMED one
   *    (be_jit_scatter_workspace_arm: no zeroing launch, re-armed by the call's last kernel) give the same bits, call after call */
  {
    const int64_t clen = 40;                                         /* prob = 0.05 */
    const int64_t jws_bytes = be_binary_jitmv_workspace_bytes(k, m, k, 0);
    void *d_j1 = dev_copy(NULL, jws_bytes), *d_j2 = dev_copy(NULL, jws_bytes), *d_o2 = dev_copy(NULL, k * 4);
    float *got2 = malloc(k * 4);
    CHECK_HIP(hipMemset(d_j1, 0x5a, jws_bytes));
    CHECK_HIP(hipMemset(d_j2, 0x5a, jws_bytes));
    CHECK_BE(be_jit_scatter_workspace_arm(d_j2, jws_bytes, NULL));
    int same = 1;
    for (int rep = 0; rep < 3; ++rep) {
      CHECK_BE(be_binary_jitmv(0, 1.5, 0.0, BE_F32, clen, 7u, d_spk, BE_SPIKE_BOOL, d_out, k, m, k, 0, 0, d_j1, jws_bytes, NULL));
      CHECK_BE(be_binary_jitmv(0, 1.5, 0.0, BE_F32, clen, 7u, d_spk, BE_SPIKE_BOOL, d_o2, k, m, k, 0, 0, d_j2, jws_bytes, NULL));
      CHECK_HIP(hipMemcpy(got, d_out, k * 4, hipMemcpyDeviceToHost));
      CHECK_HIP(hipMemcpy(got2, d_o2, k * 4, hipMemcpyDeviceToHost));
      same = same && memcmp(got, got2, k * 4) == 0;
    }
    double total = 0;
    for (int64_t i = 0; i < k; ++i) total += got[i];
    printf("%-28s %s (sum of the outputs %.1f)\n", "jit scatter, armed workspace", same && total > 0 ? "ok" : "FAIL", total);
    fails += same && total > 0 ? 0 : 1;
    CHECK_BE(be_jit_scatter_workspace_disarm(d_j2));
    free(got2);
  }

  /* error convention: status code + message, never an abort */
  if (be_binary_csrmv_t_plan(NULL, 0, BE_F32, d_blob, d_seg, d_spk, BE_SPIKE_BOOL, d_out, m, k, shift, width, layout, block_hint,
                             parts, scale_exp, d_pws, 16, NULL) != BE_ERR_WORKSPACE) { printf("missing BE_ERR_WORKSPACE\n"); ++fails; }
  be_shutdown();
  printf(fails ? "abi_smoke: %d FAILED\n" : "abi_smoke: all ok\n", fails);
  return fails ? 1 : 0;
}
