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

/* ---- CSR, transpose=True, f32, all host cores (NOT what the reference does) ----------------------------------
 * An upper bound for the CPU side: the same loop with the active rows spread over OpenMP threads and
 * the scatter made safe with atomic adds.  The reference's kernel is serial ("Cannot parallelize due to race
 * condition", brainevent/_csr/binary.py:397/:444); this variant only answers "what if it were not". */
void oracle_csrmv_t_f32_parallel(const float* w, int homo, const int32_t* indices, const int64_t* indptr, const void* v,
                                 int v_is_float, int64_t m, int64_t k, float* posts, int n_threads) {
  memset(posts, 0, (size_t)k * sizeof(float));
  const uint8_t* vb = (const uint8_t*)v;
  const float* vf = (const float*)v;
  const float w0 = w[0];
#pragma omp parallel for schedule(dynamic, 4) num_threads(n_threads)
  for (int64_t i = 0; i < m; ++i) {
    const int on = v_is_float ? (vf[i] > 0.f) : (vb[i] != 0);
    if (!on) continue;
    for (int64_t j = indptr[i]; j < indptr[i + 1]; ++j) {
      const float x = homo ? w0 : w[j];
#pragma omp atomic
      posts[indices[j]] += x;
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

/* ---- dense, f32 ------------------------------------------------------------------------------------
 * brainevent/_dense/binary.py:178-190 (transpose: posts += weights[i] for active i),
 *                             :193-206 (no transpose: posts += weights[:, i]). */
void oracle_densemv_f32(const float* w, int64_t rows, int64_t cols, const void* s, int s_is_float, int transpose,
                        float* posts) {
  const uint8_t* sb = (const uint8_t*)s;
  const float* sf = (const float*)s;
  const int64_t n_out = transpose ? cols : rows, k = transpose ? rows : cols;
  memset(posts, 0, (size_t)n_out * sizeof(float));
  for (int64_t i = 0; i < k; ++i) {
    const int on = s_is_float ? (sf[i] > 0.f) : (sb[i] != 0);
    if (!on) continue;
    if (transpose) for (int64_t j = 0; j < cols; ++j) posts[j] += w[i * cols + j];
    else for (int64_t j = 0; j < rows; ++j) posts[j] += w[j * cols + i];
  }
}

/* ---- light_rng (brainevent/_numba_random.py:385-502) ---------------------------------------------- */
#include <math.h>
static inline uint32_t lr_mix32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
static inline uint32_t lr_bounded(uint32_t r, uint32_t b) { return (uint32_t)(((uint64_t)r * (uint64_t)b) >> 32); }
static inline uint32_t lr_next(uint32_t x) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; return x == 0u ? 0x6d2b79f5u : x; }
static inline uint32_t lr_init(uint32_t seed, uint32_t row, uint32_t chunk, uint32_t lane) {
  uint32_t x = seed ^ 0xd1b54a35u;
  x ^= row * 0x85ebca6bu; x ^= chunk * 0xc2b2ae35u; x ^= lane * 0x27d4eb2du;
  x = lr_mix32(x);
  return x == 0u ? 0x6d2b79f5u : x;
}
static inline uint32_t lr_initial_q(uint32_t* state, uint32_t cl) {
  const uint32_t n = cl - 1u;
  for (;;) {
    *state = lr_next(*state); const uint32_t q = lr_bounded(*state, n);
    *state = lr_next(*state); const uint32_t gate = lr_bounded(*state, n);
    if (gate < n - q) return q;
  }
}
static inline float lr_uniform01(uint32_t seed, uint32_t row, uint32_t col) {
  uint32_t h = seed ^ 0xa0761d65u;
  h ^= row * 0xe7037ed1u; h ^= col * 0x8ebc6af1u;
  h = lr_mix32(h);
  return (float)(h & 0x00ffffffu) * (1.0f / 16777216.0f);
}
#pragma GCC push_options
#pragma GCC optimize ("fp-contract=off")
static float lr_normal01(uint32_t seed, uint32_t row, uint32_t col) {
  volatile float u = lr_uniform01(seed, row, col);   /* one rounded f32 operation at a time, as the numpy golden model */
  const float lo = 1e-10f, hi = (float)(1.0 - 1e-10);
  if (u < lo) u = lo; else if (u > hi) u = hi;
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
    z = (((((a1 * r + a2) * r + a3) * r + a4) * r + a5) * r + a6) * v / (((((b1 * r + b2) * r + b3) * r + b4) * r + b5) * r + 1.0f);
  }
  return z;
}
static inline float jit_weight(int mode, float w0, float w1, uint32_t seed, uint32_t row, uint32_t col) {
  if (mode == 1) return w0 + lr_uniform01(seed, row, col) * w1;   /* w1 = high - low */
  if (mode == 2) return w0 + lr_normal01(seed, row, col) * w1;    /* w0 = loc, w1 = scale */
  return w0;
}
#pragma GCC pop_options

/* ---- JITC mv / mm (stride 32 / 4) ---------------------------------------------------------------------
 * brainevent/_jit_scalar/binary.py:340-377 (gather), :381-416 (scatter); _jit_uniform/binary.py:292-415;
 * _jit_normal/binary.py:307-410.  Edge weights are formed in f32, sums run in f64 (numba: out = np.float64(0.)).
 * gather: rows = out_len outputs, walk over in_len;  scatter: rows = in_len inputs, walk over out_len.
 * spikes: uint8 (0/1) [in_len];  out: double [out_len] */
void oracle_jitmv(int mode, float w0, float w1, int64_t clen, uint32_t seed, const uint8_t* spikes, int64_t shape1,
                  int64_t in_len, int64_t out_len, int gather, int stride, double* out) {
  for (int64_t i = 0; i < out_len; ++i) out[i] = 0.0;
  if (clen <= 0) return;
  const uint32_t cl = (uint32_t)(clen < 2 ? 2 : clen);
  int64_t chunk = (shape1 + 3) / 4; if (chunk < 1) chunk = 1;
  const int64_t n_rows = gather ? out_len : in_len, walk = gather ? in_len : out_len;
  const int64_t n_chunks = (walk + chunk - 1) / chunk;
  for (int64_t row = 0; row < n_rows; ++row) {
    if (!gather && !spikes[row]) continue;
    double acc = 0.0;
    for (int64_t c = 0; c < n_chunks; ++c) {
      const int64_t cs = c * chunk, ce = cs + chunk < walk ? cs + chunk : walk, width = ce - cs;
      for (int lane = 0; lane < stride; ++lane) {
        uint32_t state = lr_init(seed, (uint32_t)row, (uint32_t)c, (uint32_t)lane);
        uint32_t q = lr_initial_q(&state, cl);
        int64_t lj = lane + (int64_t)stride * q;
        while (lj < width) {
          const int64_t j = cs + lj;
          if (gather) { if (spikes[j]) acc += (double)jit_weight(mode, w0, w1, seed, (uint32_t)row, (uint32_t)j); }
          else out[j] += (double)jit_weight(mode, w0, w1, seed, (uint32_t)row, (uint32_t)j);
          state = lr_next(state);
          q = q + 1u + lr_bounded(state, cl - 1u);
          lj = lane + (int64_t)stride * q;
        }
      }
    }
    if (gather) out[row] = acc;
  }
}
