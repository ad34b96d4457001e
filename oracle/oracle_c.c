/* oracle_c.c — plain-C restatement of the reference's CPU loops for the hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/README.md): used by tests/ as a second checker and by
 * bench.py's cpu_baseline leg as the timed single-thread CPU baseline ("kind": "port").
 * The product library never links or loads this file.
 *
 * Parity status: pinned (tests/test_oracle.py checks it against the reference's known-answer tests
 * and the golden vectors in tests/golden/).
 *
 * Each function cites the reference lines it follows (paths relative to the reference checkout).
 * Build: make -C oracle   (gcc -O3 -march=native -shared -fPIC)
 */
#include <stdint.h>
#include <string.h>

/* ---- CSR, transpose=True, f32 -------------------------------------------------------------------
 * brainevent/_csr/binary.py:399-405 (homo, bool), :409-416 (homo, float),
 *                           :446-451 (hetero, bool), :455-461 (hetero, float).
 * Serial by construction in the reference ("Cannot parallelize due to race condition", :397/:444). */
void oracle_csrmv_t_f32(const float* w, int homo, const int32_t* indices, const int64_t* indptr,
                        const void* v, int v_is_float, int64_t m, int64_t k, float* posts) {
  memset(posts, 0, (size_t)k * sizeof(float));
  const uint8_t* vb = (const uint8_t*)v;
  const float* vf = (const float*)v;
  const float w0 = w[0];
  for (int64_t i = 0; i < m; ++i) {
    const int on = v_is_float ? (vf[i] > 0.f) : (vb[i] != 0);
    if (!on) continue;
    if (homo) {
      for (int64_t j = indptr[i]; j < indptr[i + 1]; ++j) posts[indices[j]] += w0;
    } else {
      for (int64_t j = indptr[i]; j < indptr[i + 1]; ++j) posts[indices[j]] += w[j];
    }
  }
}

/* ---- CSR, transpose=False, f32 ------------------------------------------------------------------
 * brainevent/_csr/binary.py:421-428, :432-439 (homo), :466-472, :476-482 (hetero): prange over rows. */
void oracle_csrmv_nt_f32(const float* w, int homo, const int32_t* indices, const int64_t* indptr,
                         const void* v, int v_is_float, int64_t m, int64_t k, float* posts) {
  (void)k;
  const uint8_t* vb = (const uint8_t*)v;
  const float* vf = (const float*)v;
  const float w0 = w[0];
  for (int64_t i = 0; i < m; ++i) {
    float r = 0.f;
    for (int64_t j = indptr[i]; j < indptr[i + 1]; ++j) {
      const int32_t c = indices[j];
      const int on = v_is_float ? (vf[c] > 0.f) : (vb[c] != 0);
      if (on) r += homo ? w0 : w[j];
    }
    posts[i] = r;
  }
}
