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
/* bit-packed events: uint32 words, bit i%32 of word i/32 (rows of ceil(n/32) words in a batch); the layout of the
 * reference's `bitpack` (brainevent/_event/bitpack_binary.py:32-75).  Accepted by be_compact_spikes* and by the
 * scatter entry points that start with a compaction (be_binary_csrmv/mm_t, *_t_plan, *_t_binned). */
#define BE_SPIKE_BITS 2
/* already-compacted events (single vector, scatter entry points only): `spikes` is a HOST pointer to a be_spike_ids_t
 * whose two fields are DEVICE pointers — the first *n_active entries of active_ids are the active positions, each
 * < m and listed once (caller contract; what be_compact_spikes produces).  The call skips its compaction kernel. */
#define BE_SPIKE_IDS 3
typedef struct be_spike_ids {
  const uint32_t* active_ids;
  const uint32_t* n_active;
} be_spike_ids_t;

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
/* Diagnostic: the read-only streaming rate of this device in the caller's own run (the ceiling a read-dominated kernel is held
 * against): `repeats` passes of 16-byte-per-lane loads over buf[0 .. bytes) (16-byte aligned, device memory) between
 * two HIP events on `stream`; synchronises; *ms_per_pass = average pass time.  sink4: 4 writable device bytes. */
int be_diag_stream_read(const void* buf, int64_t bytes, int repeats, void* sink4, float* ms_per_pass, be_stream_t stream);
/* releases the little the library keeps between calls (profiling events); exchange handles have be_exchange_destroy */
int be_shutdown(void);

/* ----------------------------------------------------------------------------------------------
 * Neuron half of the COBA step loop (the reference example composes the same dynamics from brainstate
 * modules, examples/COBA_2005.py:35-87): ONE fused update of n conductance-based LIF neurons with exponential synapses,
 * in place.  Per neuron, every operation rounded separately in this order (so it reproduces the plain elementwise
 * formulation bit for bit):
 *   g_exc = g_exc * decay_exc + in_exc;   g_inh = g_inh * decay_inh + in_inh          (in_*: this step's scatter outputs)
 *   i_syn = (g_exc * (e_exc - v) + g_inh * (e_inh - v)) * syn_scale
 *   dv    = (-(v - v_rest) + i_syn + i_ext) * (dt / tau_m)
 *   active = refractory <= 0;  v' = active ? v + dv : v;  spike = active && v' >= v_th
 *   v = spike ? v_reset : v';  refractory = spike ? t_ref : refractory - dt;  spikes_out = spike (1 byte);
 *   spike_count += spike (optional, may be NULL)
 * A time step of the network is then: scatter(spikes_exc) -> in_exc, scatter(spikes_inh) -> in_inh, be_lif_coba_step.
 * ---------------------------------------------------------------------------------------------- */
int be_lif_coba_step(float* v, float* g_exc, float* g_inh, float* refractory, const float* in_exc, const float* in_inh,
                     uint8_t* spikes_out, float* spike_count, int64_t n, double dt, double tau_m, double v_rest, double v_th,
                     double v_reset, double t_ref, double e_exc, double e_inh, double decay_exc, double decay_inh, double i_ext,
                     double syn_scale, be_stream_t stream);
/* The same step; the spikes are ALSO (or only: spikes_out may then be NULL) written bit-packed, spike_bits_out[ceil(n / 32)]
 * (bit i % 32 of word i / 32; bits past n are 0) — the form every scatter entry point takes as BE_SPIKE_BITS and
 * be_exchange_allgather_bits / be_exchange_post gather from where it lies: a step loop that keeps its spikes as words has no
 * pack launch anywhere (the reference packs per call, brainevent/_jit_scalar/binary_jitsmv.cu:107-125). */
int be_lif_coba_step_packed(float* v, float* g_exc, float* g_inh, float* refractory, const float* in_exc, const float* in_inh,
                            uint8_t* spikes_out, uint32_t* spike_bits_out, float* spike_count, int64_t n, double dt, double tau_m,
                            double v_rest, double v_th, double v_reset, double t_ref, double e_exc, double e_inh,
                            double decay_exc, double decay_inh, double i_ext, double syn_scale, be_stream_t stream);
/* The CURRENT-based twin (reference example examples/CUBA_2005.py:35-66: LIF V_rest -49 mV, V_th -50 mV, V_reset -60 mV, tau 20 ms,
 * refractory 5 ms; Expon synapses tau 5 / 10 ms with CUBA outputs, weights 1.62 / -9.0 mS): the same update with
 *   i_syn = (g_exc + g_inh) * syn_scale
 * in place of the conductance term (no reversal potentials; an inhibitory projection carries a negative weight).  Every other line,
 * the rounding order and the spike outputs are those of be_lif_coba_step / be_lif_coba_step_packed. */
int be_lif_cuba_step(float* v, float* g_exc, float* g_inh, float* refractory, const float* in_exc, const float* in_inh,
                     uint8_t* spikes_out, float* spike_count, int64_t n, double dt, double tau_m, double v_rest, double v_th,
                     double v_reset, double t_ref, double decay_exc, double decay_inh, double i_ext, double syn_scale,
                     be_stream_t stream);
int be_lif_cuba_step_packed(float* v, float* g_exc, float* g_inh, float* refractory, const float* in_exc, const float* in_inh,
                            uint8_t* spikes_out, uint32_t* spike_bits_out, float* spike_count, int64_t n, double dt, double tau_m,
                            double v_rest, double v_th, double v_reset, double t_ref, double decay_exc, double decay_inh, double i_ext,
                            double syn_scale, be_stream_t stream);
/* Either step (current_based = 0: be_lif_coba_step_packed, 1: be_lif_cuba_step_packed; e_exc / e_inh ignored then) with the two
 * inputs multiplied by in_scale_exc / in_scale_inh first (g = g * decay + in * in_scale, each operation rounded separately).  What it
 * is for: ONE scatter for both projections of a network — the excitatory and the inhibitory matrix stacked into one n x 2n matrix
 * with weight 1 (columns [0, n) = targets of the excitatory rows, [n, 2n) = targets of the inhibitory rows): its output halves are
 * the synaptic COUNTS, the weights are applied here — count * w is rounded exactly as it is inside a scatter with weight w, so the
 * step equals the two-projection formulation bit for bit with half the launches (examples/coba_2005.py `combined=True`).
 * in_scale = 1.0 reproduces the plain entry points. */
int be_lif_step_scaled_packed(int current_based, float* v, float* g_exc, float* g_inh, float* refractory, const float* in_exc,
                              const float* in_inh, double in_scale_exc, double in_scale_inh, uint8_t* spikes_out,
                              uint32_t* spike_bits_out, float* spike_count, int64_t n, double dt, double tau_m, double v_rest,
                              double v_th, double v_reset, double t_ref, double e_exc, double e_inh, double decay_exc,
                              double decay_inh, double i_ext, double syn_scale, be_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * event vector helpers (replace: brainevent/_jit_scalar/binary_jitsmv.cu:107-125 `_pack_bool_kern`
 * and the active-row extraction of brainevent/_csr/binary_csrmv_hybrid.cu:275-327)
 * ---------------------------------------------------------------------------------------------- */
/* spikes[n] -> bits[ceil(n/32)] (bit i%32 of word i/32 set iff spike i active) */
int be_pack_spikes(const void* spikes, int spike_dtype, int64_t n, uint32_t* bits, be_stream_t stream);
/* batch-major spikes_bm[n_batch, n] -> bits[n_batch, ceil(n/32)] */
int be_pack_spikes_batched(const void* spikes_bm, int spike_dtype, int64_t n, int64_t n_batch, uint32_t* bits,
                           be_stream_t stream);
/* bits[ceil(n/32)] -> spikes_out[n] (one 0/1 byte per spike) */
int be_unpack_spikes(const uint32_t* bits, int64_t n, uint8_t* spikes_out, be_stream_t stream);
/* spikes[n] -> active_ids[<=n] (unordered) and *count (device uint32) */
int be_compact_spikes(const void* spikes, int spike_dtype, int64_t n, uint32_t* active_ids, uint32_t* count,
                      be_stream_t stream);
/* batch-major spikes_bm[n_batch, n] -> active_ids[b * active_stride + ...] and counts[b] */
int be_compact_spikes_batched(const void* spikes_bm, int spike_dtype, int64_t n, int64_t n_batch, uint32_t* active_ids,
                              int64_t active_stride, uint32_t* counts, be_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * spike exchange of the multi-GPU partition (the reference has no distributed path): one process per
 * GPU, rank g owns a post slice of the matrix and the spikes of its 1/G of the pre population; ONE all-gather of the
 * bit-packed slices (RCCL over xGMI) rebuilds the full spike vector on every rank, which the scatter entry points consume
 * as it is (BE_SPIKE_BITS).  Every rank owns the same whole number of 32-bit words: slice of rank r =
 * [r * w * 32, (r + 1) * w * 32) clipped to n_pre, w = ceil(ceil(n_pre / 32) / world) (be_exchange_slice).
 *   rank 0: be_exchange_get_unique_id(id)  ->  the binder ships the be_exchange_unique_id_bytes() bytes to every process
 *   all   : be_exchange_init(id, world, rank, n_pre, &ex)        (collective: ncclCommInitRank; current HIP device)
 *   step  : be_exchange_allgather_bits(ex, local_spikes, dtype, full_bits, stream)   full_bits: be_exchange_full_words(ex) words
 *           dtype BE_SPIKE_BITS: local_spikes are the slice's own words (bits past the slice 0): a slice that fills its w words
 *           is gathered from where it lies (no pack launch, no copy); with be_exchange_post the words must stay unchanged
 *           until the slot's be_exchange_wait has been queued
 *   end   : be_exchange_destroy(ex)
 * The handle owns the communicator and one device buffer of w words.  RCCL is loaded with dlopen("librccl.so") on first use
 * (override: environment variable BE_RCCL_LIB); BE_ERR_UNSUPPORTED if it cannot be loaded.
 * ---------------------------------------------------------------------------------------------- */
int be_exchange_unique_id_bytes(void);
int be_exchange_get_unique_id(void* id_host);
int be_exchange_init(const void* id_host, int world, int rank, int64_t n_pre, void** exchange_host_out);
int be_exchange_slice(const void* exchange, int rank, int64_t* lo_host, int64_t* hi_host);
/* the same partition as a pure function (no handle, no device, no RCCL): rank's slice [lo, hi) of n_pre spikes and the words
 * every rank owns — what init, slice, allgather and post all compute; any of the three outputs may be NULL */
int be_exchange_slice_for(int64_t n_pre, int world, int rank, int64_t* lo_host, int64_t* hi_host, int64_t* words_per_rank_host);
int64_t be_exchange_full_words(const void* exchange);
int be_exchange_allgather_bits(void* exchange, const void* local_spikes, int spike_dtype, uint32_t* full_bits,
                               be_stream_t stream);
/* pipelined form (synaptic delays >= 2 steps): post step t + 1's exchange on the library's own stream — it waits for what
 * producer_stream has queued so far, i.e. the spikes — then scatter step t; wait makes consumer_stream wait for the posted
 * slot (0 / 1, alternate them) and returns its device buffer, valid until that slot is posted again.
 * Slot reuse: the gather into a slot must not start before the consumer's last read of the slot's previous contents.  When
 * consumer_stream IS producer_stream (one stream issues the scatters and the posts) that order is implied: post waits for
 * everything that stream has queued.  A consumer on another stream calls be_exchange_release(ex, slot, consumer_stream)
 * after queueing its last read of the slot; the next post into the slot then waits for that point too.
 * A failed be_exchange_post (stream / event / buffer creation) leaves the handle usable: the next call starts over. */
int be_exchange_post(void* exchange, const void* local_spikes, int spike_dtype, int slot, be_stream_t producer_stream);
int be_exchange_wait(void* exchange, int slot, const uint32_t** full_bits_out, be_stream_t consumer_stream);
int be_exchange_release(void* exchange, int slot, be_stream_t consumer_stream);
/* The posted exchange that ALSO compacts what it gathered: right behind the all-gather, still on the library's own stream, the
 * slot's words become the list of active pre neurons (ids + a device counter, as be_compact_spikes writes them).
 * be_exchange_wait_ids makes consumer_stream wait for the slot and fills *ids_out — pass a pointer to it as the `spikes` argument
 * of a scatter entry point with spike dtype BE_SPIKE_IDS (n_batch = 1): the scatter then runs no compaction of its own, fused or
 * launched, on the consumer's critical path.  *full_bits_out (optional) = the slot's words, as be_exchange_wait returns them.
 * Same slot, lifetime and release rules as be_exchange_post / _wait; the list holds up to n_pre ids and lives in the handle.
 * The exchange's events are created without a system-scope fence (they order work of one device; environment variable
 * BE_EXCHANGE_SYSTEM_FENCE=1 restores default events). */
int be_exchange_post_ids(void* exchange, const void* local_spikes, int spike_dtype, int slot, be_stream_t producer_stream);
int be_exchange_wait_ids(void* exchange, int slot, be_spike_ids_t* ids_out, const uint32_t** full_bits_out,
                         be_stream_t consumer_stream);
/* Measurement hook (process-wide; also environment variable BE_EXCHANGE_EMULATE_US at start-up): every all-gather issued from now on
 * is followed, on its own stream, by a spin kernel of `us` microseconds — a stand-in for the latency of a real multi-rank all-gather
 * on a box with one GPU, so that the sequential and the pipelined schedule can be timed against an exchange of realistic length
 * (bench.py rank_breakdown.emulated).  0 switches it off (the default). */
int be_exchange_emulate_latency_us(double us);
int be_exchange_destroy(void* exchange);

/* ------------------------------------------------------------------------------------------------
 * Batch convention (all *mm entry points): spikes_bm is batch-major [n_batch, len] and out_bm is
 * batch-major [n_batch, out_len] — the physical layout the reference's SRAW kernels also emit
 * (brainevent/_csr/binary.py:1263-1286, brainevent/_fcn/binary.py:867-889: "Python transposes back").
 * The *mv entry points are the n_batch = 1 case of the same kernels.
 * ---------------------------------------------------------------------------------------------- */

/* ------------------------------------------------------------------------------------------------
 * binary_csrmv / binary_csrmm, transpose=True (scatter):  out[indices[j]] += w[j]  for every active row
 * replaces: binary_csrmv_wat_hybrid_{homo,hetero}_{f32,f64,f16,bf16}_{bool,float}
 *           (brainevent/_csr/binary_csrmv_hybrid.cu:619-632, 789-821),
 *           binary_csrmm_sraw_hybrid_{…} (brainevent/_csr/binary_csrmm_hybrid.cu:16-57, 469-530)
 * and, with indptr == NULL and row_len = n_conn, binary_fcnmv_scatter_{…} / binary_fcnmm_sraw_{…}
 *           (brainevent/_fcn/binary_fcnmv.cu:55-137, 207-217; brainevent/_fcn/binary_fcnmm.cu:486-529, 835-857).
 *   weights : [nnz] (hetero) or [1] (homo), dtype wdtype;  out : [n_batch, k] dtype wdtype (fully written)
 *   spikes  : [n_batch, m];  indices : [nnz] int32 in [0,k);  indptr : [m+1] or NULL (rows are row_len long)
 *   workspace : >= be_binary_csrmm_t_workspace_bytes(m, k, n_batch, wdtype) bytes, 256-byte aligned
 * "direct" route: no preprocessing, global float atomics.
 * ---------------------------------------------------------------------------------------------- */
int64_t be_binary_csrmv_t_workspace_bytes(int64_t m, int64_t k, int wdtype);
int64_t be_binary_csrmm_t_workspace_bytes(int64_t m, int64_t k, int64_t n_batch, int wdtype);
int be_binary_csrmv_t(const void* weights, int homo, int wdtype, const int32_t* indices, const void* indptr,
                      int indptr_is_i64, int64_t row_len, const void* spikes, int spike_dtype, void* out,
                      int64_t m, int64_t k, void* workspace, int64_t workspace_bytes, be_stream_t stream);
int be_binary_csrmm_t(const void* weights, int homo, int wdtype, const int32_t* indices, const void* indptr,
                      int indptr_is_i64, int64_t row_len, const void* spikes_bm, int spike_dtype, void* out_bm,
                      int64_t m, int64_t k, int64_t n_batch, void* workspace, int64_t workspace_bytes,
                      be_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * post-sliced scatter plan (the MI355X-native layout behind `spk @ CSR` / `spk @ FixedNumPerPre`).
 * Plays the role of the reference's per-matrix task workspace
 * (brainevent/_csr/main.py:58-88, brainevent/_csr/hybrid_config.py:298-324): built once per matrix,
 * cached by the CSR object, passed to every call.
 *
 * Layout: output neurons are cut into slices of `slice_width` columns (slice = column / slice_width, local column =
 * column % slice_width); 2^slice_shift is the accumulator capacity of a workgroup (LDS) and the stride of its
 * partial sums, slice_width <= 2^slice_shift, and slice_width = 0 means 2^slice_shift.  A width below the capacity
 * balances the slices (k = 1M, shift 14: 64 slices of 15625 instead of 61 full ones and a sliver of 576).
 * For row r and slice s the entries of row r
 * whose column falls in slice s form one 128-byte-aligned block inside `blob`:
 *     hetero: [ f32 weight x 4*ng ][ uint16 local column x 4*ng ]   with ng = ceil(count / 4) lane groups
 *     homo  : [ uint16 local column x 8*ng ]                          with ng = ceil(count / 8)
 * pads carry local column 2^slice_shift (a dummy accumulator) and weight 0.
 *     seg[(r * n_slices + s)] = { uint32 block start in 128-B units, uint32 ng }     (8 bytes per entry)
 * layout BE_PLAN_D8 (heterogeneous weights, at most 1024 slices, rows of at most 16384 entries — be_scatter_plan_count
 * checks every row on the device and returns BE_ERR_RANGE for a longer one): 5 bytes per entry —
 *     block: [ f32 weight x 4*ng ][ uint8 delta x 4*ng ], entries sorted by column, column = previous column + delta
 *     (first delta 0), gaps above 255 bridged by escape entries (weight 0, delta 255), tail pads (weight 0, delta 0);
 *     seg = { block start in 128-B units, ng | (local column of the first entry << 16) }.
 * layout BE_PLAN_H8 (one homogeneous weight; same limits as d8; slice_width up to 40000): 1 byte per entry —
 *     block: [ uint8 code x 8*ng ], entries sorted by column; code c < 255: advance c columns and count one entry,
 *     c = 255: advance 255 columns and count nothing (a gap g is g / 255 escapes followed by the code g % 255; tail pads
 *     are 255); the first code is 0 and seg is as for d8.
 *
 *   step 1  be_scatter_plan_count : fills seg (m * n_slices entries of 8 B) and returns the size of `blob`
 *           in *blob_bytes_host.  SYNCHRONOUS (it reads the total back).
 *   step 2  caller allocates blob (128-byte aligned, blob_bytes + 128).
 *   step 3  be_scatter_plan_fill  : fills blob; writes the f32 bit patterns of max |w| and of the smallest non-zero |w|
 *           to maxabs_bits[0] and maxabs_bits[1] (device uint32[2]) so that the caller can pick the fixed-point exponent
 *           and refuse matrices whose dynamic range the 64-bit fixed-point sums cannot resolve.
 *   later   be_scatter_plan_refresh_weights : same arguments as the fill, over the SAME seg / blob, after the weights of
 *           an unchanged structure were updated (plasticity).  The reference's cached workspace holds task ranges only
 *           (brainevent/_csr/main.py:58-88, :148-161), so weight updates never invalidate it; here the blocks embed the
 *           weights, and this call is what keeps them current.  Re-derive scale_exp from the new maxabs / column sums.
 * ---------------------------------------------------------------------------------------------- */
#define BE_BINNED_ACC32 2 /* `homo` argument of the binned entry points: per-entry weights, 32-bit fixed-point sums */
#define BE_BINNED_ABS 4   /* ... OR-ed to 0 / 2: the step sums |w| (column statistics: how a caller derives scale_exp with a few
                            binned steps over all the rows instead of a pass of global atomics over the entries) */
#define BE_BINNED_SHORT_ROWS 8 /* ... OR-ed to any kind, a performance hint for matrices WITH an indptr: the stored rows average at
                                 most 256 entries (pass B keeps one step of loads in flight instead of two; rows of one length —
                                 indptr NULL — are classified by row_len).  Results do not depend on it. */
#define BE_PLAN_U16 0 /* uint16 local columns (both weight kinds) */
#define BE_PLAN_D8 1  /* sorted columns as uint8 deltas (heterogeneous weights) */
#define BE_PLAN_H8 2  /* sorted columns as uint8 advance codes (one homogeneous weight) */
int64_t be_scatter_plan_scratch_bytes(int64_t m, int64_t k, int slice_shift, int slice_width);
int be_scatter_plan_count(const int32_t* indices, const void* indptr, int indptr_is_i64, int64_t row_len,
                          int64_t m, int64_t k, int slice_shift, int slice_width, int homo, int layout, void* seg,
                          void* scratch, int64_t scratch_bytes, int64_t* blob_bytes_host, be_stream_t stream);
int be_scatter_plan_fill(const void* weights, int homo, int wdtype, const int32_t* indices, const void* indptr,
                         int indptr_is_i64, int64_t row_len, int64_t m, int64_t k, int slice_shift, int slice_width,
                         int layout, const void* seg, void* blob, uint32_t* maxabs_bits, be_stream_t stream);
int be_scatter_plan_refresh_weights(const void* weights, int homo, int wdtype, const int32_t* indices, const void* indptr,
                                    int indptr_is_i64, int64_t row_len, int64_t m, int64_t k, int slice_shift,
                                    int slice_width, int layout, const void* seg, void* blob, uint32_t* maxabs_bits,
                                    be_stream_t stream);
/* Planning a matrix from BLOCKS OF ROWS that are resident one at a time (a matrix whose raw CSR and plan do not fit the device
 * together).  Everything the count and the fill touch is local to a row except the block starts, which one scan over the whole
 * segment table provides:
 *   be_scatter_plan_begin(m, k, ...)                                once: scratch as for be_scatter_plan_count, whole matrix
 *   be_scatter_plan_count_rows(block, ..., m_rows, m_total, ..., seg + r0 * n_slices * 8 bytes, scratch, ..., order_blk)   per block:
 *                          indices / indptr of rows [r0, r0 + m_rows), indptr RELATIVE to the block (indptr[0] == 0)
 *   be_scatter_plan_scan(m, k, ..., seg, scratch, ..., &blob_bytes)   once, SYNCHRONOUS; the caller allocates blob
 *   be_scatter_plan_fill_ordered(block, ..., m = m_rows, ..., seg + r0 * n_slices * 8, blob, maxabs, order_blk)           per block;
 *                          every call restarts maxabs: combine the blocks' (max, min) on the caller's side
 * be_scatter_plan_count[_ordered] is exactly begin + count_rows over all rows + scan. */
int be_scatter_plan_begin(int64_t m, int64_t k, int slice_shift, int slice_width, void* scratch, int64_t scratch_bytes,
                          be_stream_t stream);
int be_scatter_plan_count_rows(const int32_t* indices, const void* indptr, int indptr_is_i64, int64_t row_len, int64_t m_rows,
                               int64_t m_total, int64_t k, int slice_shift, int slice_width, int homo, int layout, void* seg_rows,
                               void* scratch, int64_t scratch_bytes, uint16_t* order_out, be_stream_t stream);
int be_scatter_plan_scan(int64_t m, int64_t k, int slice_shift, int slice_width, void* seg, void* scratch, int64_t scratch_bytes,
                         int64_t* blob_bytes_host, be_stream_t stream);

/* The sorted layouts (BE_PLAN_D8 / BE_PLAN_H8) need every row in column order.  The *_ordered forms keep that order instead of
 * sorting three times: the count pass writes it — order[nnz] uint16, the row-local position of the i-th smallest column of
 * each row — and the fill and every later weight refresh read it back (a gather-copy: no sort).  order == NULL, or the
 * BE_PLAN_U16 layout: exactly the calls above.  The caller may free `order` after the fill (the build then saved one sort)
 * or keep it — 2 bytes per stored entry — for as long as it wants cheap refreshes. */
int be_scatter_plan_count_ordered(const int32_t* indices, const void* indptr, int indptr_is_i64, int64_t row_len, int64_t m,
                                  int64_t k, int slice_shift, int slice_width, int homo, int layout, void* seg, void* scratch,
                                  int64_t scratch_bytes, int64_t* blob_bytes_host, uint16_t* order_out, be_stream_t stream);
int be_scatter_plan_fill_ordered(const void* weights, int homo, int wdtype, const int32_t* indices, const void* indptr,
                                 int indptr_is_i64, int64_t row_len, int64_t m, int64_t k, int slice_shift, int slice_width,
                                 int layout, const void* seg, void* blob, uint32_t* maxabs_bits, const uint16_t* order,
                                 be_stream_t stream);
int be_scatter_plan_refresh_weights_ordered(const void* weights, int homo, int wdtype, const int32_t* indices, const void* indptr,
                                            int indptr_is_i64, int64_t row_len, int64_t m, int64_t k, int slice_shift,
                                            int slice_width, int layout, const void* seg, void* blob, uint32_t* maxabs_bits,
                                            const uint16_t* order, be_stream_t stream);

/* Fixed-point exponent of a weight array — the `scale_exp` of the planned and the binned step — chosen by the library:
 *   overflow bound : the largest e with (largest column sum of |w|) * 2^e < 2^62 (every row active; a row may list a column
 *                    several times, so "rows x max|w|" is NOT a bound).  indices == NULL: all the weights bound a column.
 *   accuracy gate  : e is accepted when the largest weight of every non-empty output column keeps `min_weight_bits` bits
 *                    (16 gives ~1e-5 of the column's weight scale, the tolerance of the path).
 *   keep_exp       : an exponent to keep when it still cannot overflow (weights refreshed under a captured graph, where
 *                    scale_exp is a recorded launch argument), or INT_MIN.
 * SYNCHRONOUS (reads statistics back).  Returns BE_ERR_RANGE — use the direct route — for inf / nan weights or a dynamic
 * range 64-bit sums cannot resolve.  scratch >= be_fixed_point_scratch_bytes(k).  No reference counterpart: its GPU kernels
 * add floats with global atomics (brainevent/_csr/binary_csrmv_hybrid.cu:199-234). */
int64_t be_fixed_point_scratch_bytes(int64_t k);
/* max |w| and the smallest non-zero |w| of a weight array as f32 bit patterns (0 / 0xffffffff when there is none; a max at or
 * above 0x7f800000 means inf / nan): one streaming pass.  scratch >= 256 bytes.  SYNCHRONOUS. */
int be_weight_stats(const void* weights, int wdtype, int64_t n, uint32_t* max_bits_host, uint32_t* min_nonzero_bits_host,
                    void* scratch, int64_t scratch_bytes, be_stream_t stream);
int be_fixed_point_exponent(const void* weights, int wdtype, const int32_t* indices, int64_t nnz, int64_t k,
                            int min_weight_bits, int keep_exp, void* scratch, int64_t scratch_bytes, int* scale_exp_host,
                            be_stream_t stream);
/* The same exponent — same bound, gate and keep_exp rule — for a matrix that has a WEIGHTED plan (BE_PLAN_U16 with per-entry
 * weights, BE_PLAN_D8), computed from the plan's own blocks: one workgroup per slice sums |w| of every row's block in LDS
 * (64-bit fixed point, addends rounded up: the bound never falls short) — a planned step with all rows active — instead of two
 * passes of global float atomics over the raw entries (1e10 entries: 0.47 s -> tens of ms).  maxabs_bits: what
 * be_scatter_plan_fill left.  scratch >= 256 bytes.  SYNCHRONOUS. */
int be_scatter_plan_exponent(const void* blob, const void* seg, int64_t m, int64_t k, int slice_shift, int slice_width, int layout,
                             int64_t nnz, const uint32_t* maxabs_bits, int min_weight_bits, int keep_exp, void* scratch,
                             int64_t scratch_bytes, int* scale_exp_host, be_stream_t stream);

/* planned scatter step: out[n_batch, k] (dtype wdtype, fully written) from spikes[n_batch, m].
 *   weights : device pointer to weights[0] (homo only; may be NULL for hetero)
 *   scale_exp : fixed-point exponent chosen by the caller (hetero only): every stored weight is accumulated as
 *               round(w * 2^scale_exp) in a 64-bit integer (order independent, bitwise reproducible).  The largest
 *               column sum of |w| times 2^scale_exp must stay below 2^62 (a row may list a column several times, so
 *               "rows x max|w|" is NOT a bound); sums that exceed it wrap silently.
 *   block_hint : average stored items per (row, slice) block of the plan — nnz / (m * slices); for BE_PLAN_D8 the
 *               entries plus its escape items, i.e. 4 x the average of the table's lane-group counts — or 0 (unknown).
 *               A speed hint only, never a correctness input: short blocks are decoded by 4, 8 or 16 lanes each instead
 *               of a wave each (the variant whose single pass holds the average + 3 sigma of a Poisson length; longer
 *               blocks finish in a serial tail), and with >= 40 slices of blocks <= 64 items the step first gathers the
 *               active rows' table entries into the workspace (two small launches more).
 *   parts : number of workgroups that share one slice (each takes 1/parts of the active rows)
 *   workspace : >= be_binary_csrmm_t_plan_workspace_bytes(m, k, n_batch, slice_shift, slice_width, parts, homo) bytes
 *               (spike counters + active-row list + partial sums; for a single vector and >= 40 slices also room for the
 *               pre-gathered segment table, 8 B x m x slices, up to 8 GiB).
 *               Its first 4 * n_batch bytes (the spike counters) must be ZERO on entry; they are zero again when the
 *               call has completed, so a workspace zero-filled once can be reused for every step.
 */
int64_t be_binary_csrmv_t_plan_workspace_bytes(int64_t m, int64_t k, int slice_shift, int slice_width, int parts, int homo);
int64_t be_binary_csrmm_t_plan_workspace_bytes(int64_t m, int64_t k, int64_t n_batch, int slice_shift, int slice_width,
                                               int parts, int homo);
/* the same plus room for the pre-gathered segment table, which the single-vector step uses for plans of >= 40 slices and short
 * blocks (block_hint <= 64, the value passed to the step): without it the step gathers from the plan's own table (correct,
 * slower there); with a block_hint that does not qualify the two sizes are equal */
int64_t be_binary_csrmm_t_plan_workspace_bytes_for(int64_t m, int64_t k, int64_t n_batch, int slice_shift, int slice_width,
                                                   int parts, int homo, int block_hint);
int be_binary_csrmv_t_plan(const void* weights, int homo, int wdtype, const void* blob, const void* seg,
                           const void* spikes, int spike_dtype, void* out, int64_t m, int64_t k, int slice_shift,
                           int slice_width, int layout, int block_hint, int parts, int scale_exp, void* workspace,
                           int64_t workspace_bytes, be_stream_t stream);
int be_binary_csrmm_t_plan(const void* weights, int homo, int wdtype, const void* blob, const void* seg,
                           const void* spikes_bm, int spike_dtype, void* out_bm, int64_t m, int64_t k, int64_t n_batch,
                           int slice_shift, int slice_width, int layout, int block_hint, int parts, int scale_exp,
                           void* workspace, int64_t workspace_bytes, be_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * binned scatter: the event-driven transpose=True product for a matrix WITHOUT a plan (raw CSR / fixed-length rows),
 * f32 / f16 / bf16 weights.  Pass B: 256 workgroups stream the active rows' entries, four per lane and load, and append
 * each to a write-combining block of its output slice ("bin") in LDS (two blocks of 8 ... 128 entries per bin; a ticket
 * per entry, one commit per entry, no workgroup barrier); a completed block is copied to the workgroup's own region of
 * the bin — no global atomics.  Pass C: one workgroup per bin streams the bin's 256 regions and accumulates in LDS
 * (integer atomics, like the planned route).  A region that overflows its share of `bin_capacity` is delivered through
 * global float atomics into an overflow image of the output instead (slower, still correct: pass C adds the image in), so
 * `bin_capacity` is a tuning parameter, not a correctness one:
 * about 1.5 x expected_active_rows x mean_row_length / be_binned_bins(k, slice_shift, homo).  The library cuts the k outputs
 * into a multiple of 256 bins of equal width (one workgroup per bin in pass C, 256 CUs), as few as one bin's accumulators fit
 * LDS and at most 2^slice_shift columns wide (slice_shift in [4, 16]; 16 = as wide as LDS allows); be_binned_bins returns
 * that count, or 0 when pass B's LDS cannot hold two blocks per bin (more than 2048 / 1146 bins counted / weighted).
 * A column index >= k is the caller's error (the entry is dropped).
 * Reproducibility: sums are integers (counts, or 64-bit fixed point at 2^scale_exp) converted once, so the result is
 * bitwise identical from call to call as long as (i) no region overflows and (ii) the matrix has more than 128 bins
 * (k > 32768: one workgroup per bin).  With fewer bins several workgroups share a bin and
 * merge through float atomics, and an overflowing block is added with float atomics too: both are order dependent in
 * the last bit (still within the 1e-5 tolerance of the path).  The planned route has neither exception.
 * Same role as binary_csrmv_wat_hybrid_* (brainevent/_csr/binary_csrmv_hybrid.cu:619-632): no preprocessing.
 * `homo` of the binned entry points names the KIND of a step: 0 = per-entry weights, sums in 64-bit fixed point at 2^scale_exp;
 * 1 = one shared weight (counts); BE_BINNED_ACC32 (2) = per-entry weights, sums in 32-bit fixed point at 2^scale_exp — bins twice
 * as wide (10M outputs: 256 bins, one round of pass C, instead of 611), for matrices whose every column keeps its largest weight
 * at >= 18 bits at that exponent: scale_exp = be_fixed_point_exponent(..., min_weight_bits = 18 + 32, ...) - 32 (BE_ERR_RANGE:
 * use kind 0).  The largest column sum of |w| times 2^scale_exp stays below 2^30 by the same call; sums stay integers
 * (order independent).  be_binned_bins and the workspace sizes take the same kind (a workspace is sized for all three).
 * ---------------------------------------------------------------------------------------------- */
int be_binned_bins(int64_t k, int slice_shift, int homo);
/* Task size of pass B (process-wide, read at every call): a task is about task_groups groups of four consecutive entries — but at
 * least four rows while those stay within 1024 groups —, and a step is cut into at least min_tasks tasks.  Defaults 256 / 2048
 * (measured on gfx950; rounds 1-3: 1024 / 2048); the Python layer sets them from its
 * persisted per-architecture tuning — the counterpart of the thresholds the reference compiles into its hybrid kernel from
 * brainevent/_csr/hybrid_config.py:77-88, :256-295. */
int be_binned_set_tuning(int task_groups, int min_tasks);
/* Sticky protocol flag of a binned workspace.  Pass B's append never blocks for good: a lane whose write-combining slot is not
 * freed within 20 ms (constant-rate clock; never observed) raises the flag and drops its pending entries, pass C then writes
 * NaN into every output of that step — the process keeps its HIP context (the reference's device-side check traps instead,
 * brainevent/include/brainevent/check.h:79-82).  This call reads the flag: BE_OK, or BE_ERR_HIP with the cause in
 * be_last_error(); clear != 0 re-arms the workspace.  SYNCHRONOUS.  A caller that never sees NaN never needs it.
 * The same call checks the workspace's CONSERVATION COUNTERS (four uint64 that every step adds to; be_binned_workspace_audit reads
 * them): [0] stored entries of the active rows (from the row bounds), [1] tickets pass B drew, [2] entries pass C added to its
 * accumulators, [3] entries delivered through the overflow image.  After complete steps [0] == [1] == [2] + [3]; otherwise
 * BE_ERR_RANGE with the four numbers in be_last_error() — an entry was lost or delivered twice (or a column id is >= k: the
 * caller's error, dropped by pass B, shows as [0] > [1]).  clear != 0 zeroes the counters too.  The reference has no counterpart:
 * its scatter adds with global atomics (brainevent/_csr/binary_csrmv_hybrid.cu:330-350), nothing is staged that could be lost. */
int be_binned_workspace_status(const void* workspace, int clear, be_stream_t stream);
/* the four conservation counters of a binned workspace -> counters_host[4].  SYNCHRONOUS. */
int be_binned_workspace_audit(const void* workspace, uint64_t* counters_host, be_stream_t stream);
int64_t be_binary_csrmv_t_binned_workspace_bytes(int64_t m, int64_t k, int slice_shift, int64_t bin_capacity);
/* once per workspace, before its first step: zeroes the spike counter and the overflow image inside it (every step leaves
 * both at zero, so the step itself needs no memset and no zeroing of `out`: pass C writes every output) */
int be_binary_csrmv_t_binned_workspace_init(void* workspace, int64_t workspace_bytes, int64_t m, int64_t k, int slice_shift,
                                            int64_t bin_capacity, be_stream_t stream);
int be_binary_csrmv_t_binned(const void* weights, int homo, int wdtype, const int32_t* indices, const void* indptr,
                             int indptr_is_i64, int64_t row_len, const void* spikes, int spike_dtype, void* out,
                             int64_t m, int64_t k, int slice_shift, int64_t bin_capacity, int scale_exp, void* workspace,
                             int64_t workspace_bytes, be_stream_t stream);
/* the same for a batch: spikes_bm [n_batch, m] (bit-packed: [n_batch, ceil(m / 32)] words), out_bm [n_batch, k].  The rows with a
 * spike in ANY batch row are read once for up to 32 batch rows at a time — each entry is appended once per batch row that has
 * its row active, into that batch row's own bins — as long as pass B's LDS holds blocks for all their bins (wide bins: few
 * per batch row); otherwise one single-vector step per batch row, as the reference's batched scatter does
 * (brainevent/_csr/binary_csrmm_hybrid.cu:16-57, brainevent/_fcn/binary_fcnmm.cu:486-529).  `bin_capacity` as for one vector. */
int64_t be_binary_csrmm_t_binned_workspace_bytes(int64_t m, int64_t k, int64_t n_batch, int slice_shift, int64_t bin_capacity);
int be_binary_csrmm_t_binned_workspace_init(void* workspace, int64_t workspace_bytes, int64_t m, int64_t k, int64_t n_batch,
                                            int slice_shift, int64_t bin_capacity, be_stream_t stream);
int be_binary_csrmm_t_binned(const void* weights, int homo, int wdtype, const int32_t* indices, const void* indptr,
                             int indptr_is_i64, int64_t row_len, const void* spikes_bm, int spike_dtype, void* out_bm,
                             int64_t m, int64_t k, int64_t n_batch, int slice_shift, int64_t bin_capacity, int scale_exp,
                             void* workspace, int64_t workspace_bytes, be_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * binary_csrmv / binary_csrmm, transpose=False (gather):  out[i] = sum_j w[j] * e(spikes[indices[j]])
 * replaces: binary_csrmv_nt_auto_{homo,hetero}_{…}_{bool,float} (brainevent/_csr/binary_csrmv.cu:437-486),
 *           binary_csrmm_nt_auto_{…} (brainevent/_csr/binary_csrmm.cu:328-392) and, with indptr == NULL,
 *           the gather direction of binary_fcnmv / binary_fcnmm (brainevent/_fcn/binary.py:201-253, 731-766).
 *   spikes : [n_batch, k];  out : [n_batch, m];  workspace >= be_binary_csrmm_nt_workspace_bytes(m, k, n_batch)
 * ---------------------------------------------------------------------------------------------- */
int64_t be_binary_csrmv_nt_workspace_bytes(int64_t m, int64_t k);
int64_t be_binary_csrmm_nt_workspace_bytes(int64_t m, int64_t k, int64_t n_batch);
int be_binary_csrmv_nt(const void* weights, int homo, int wdtype, const int32_t* indices, const void* indptr,
                       int indptr_is_i64, int64_t row_len, const void* spikes, int spike_dtype, void* out,
                       int64_t m, int64_t k, void* workspace, int64_t workspace_bytes, be_stream_t stream);
int be_binary_csrmm_nt(const void* weights, int homo, int wdtype, const int32_t* indices, const void* indptr,
                       int indptr_is_i64, int64_t row_len, const void* spikes_bm, int spike_dtype, void* out_bm,
                       int64_t m, int64_t k, int64_t n_batch, void* workspace, int64_t workspace_bytes,
                       be_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * CSR -> CSC structure conversion in column blocks, on the device, 64-bit offsets (no entry-count limit).
 * replaces: csr_to_csc_count / csr_to_csc_fill_block (brainevent/_csr/csr_to_csc.cu:137-199) and the host loop around them
 *           (brainevent/_misc.py:1380-1513 `_csr_to_csc_index_gpu_column_block`; public entry brainevent/_misc.py:1516
 *           `csr_to_csc_index(..., method="gpu_column_block")`), `fixed_conn_num_csc_structure` (brainevent/_misc.py:1255,
 *           with indptr == NULL and row_len = n_conn).  It is what the mirror of the unfavourable direction is built from
 *           (brainevent/_csr/main.py:1321-1357 `_weight_indices`, brainevent/_fcn/main.py:280-300).
 *   1  be_csr_to_csc_count   : counts[n_cols] (int64) <- stored entries per column (one pass over `indices`)
 *   2  be_csr_to_csc_indptr  : exclusive scan -> csc_indptr[n_cols + 1] (int64 or int32); *nnz_host (may be NULL; when given
 *                              the call is SYNCHRONOUS; BE_ERR_RANGE if an int32 indptr cannot hold the total)
 *   3  be_csr_to_csc_fill_block, once per column block [col_lo, col_hi), any partition of the columns: the block's entries go to
 *        rows_out / weights_out / perm_out — arrays of the BLOCK (csc_indptr[col_hi] - csc_indptr[col_lo] elements), slot =
 *        csc_indptr[column] - csc_indptr[col_lo] + position inside the column.  csc_indptr is the int64 form of step 2.
 *        rows_out: the row of every entry (int32);  weights_out: its weight, moved along (weight_bytes = 2 / 4 / 8, or 0: no
 *        weights — one shared weight);  perm_out: its position in the CSR arrays (int32 or int64; NULL = not wanted: 8 bytes
 *        per entry is more than the structure itself).  cursor: scratch of (col_hi - col_lo) int64.
 *   The order of the entries INSIDE a column is unspecified (slots are drawn from per-column cursors with atomics), exactly as
 *   in the reference's kernel (csr_to_csc.cu:26-27); every product over the result is insensitive to it.
 *   be_gather_by_perm: out[i] = src[perm[i]] for 2 / 4 / 8-byte elements — how a mirror that kept perm follows a weight update.
 * ---------------------------------------------------------------------------------------------- */
int64_t be_csr_to_csc_scratch_bytes(int64_t n_cols);
int be_csr_to_csc_count(const int32_t* indices, int64_t nnz, int64_t n_cols, int64_t* counts, be_stream_t stream);
int be_csr_to_csc_indptr(const int64_t* counts, int64_t n_cols, void* csc_indptr_out, int out_is_i64, int64_t* nnz_host,
                         void* scratch, int64_t scratch_bytes, be_stream_t stream);
int be_csr_to_csc_fill_block(const int32_t* indices, const void* indptr, int indptr_is_i64, int64_t row_len, int64_t m,
                             int64_t nnz, int64_t col_lo, int64_t col_hi, const int64_t* csc_indptr, int64_t* cursor,
                             int32_t* rows_out, void* perm_out, int perm_is_i64, const void* weights, int weight_bytes,
                             void* weights_out, be_stream_t stream);
int be_gather_by_perm(const void* src, int elem_bytes, const void* perm, int perm_is_i64, int64_t n, void* out,
                      be_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * float-operand twins (last clause): the same CSR / fixed-number matrices against a dense vector or matrix.
 * replaces: csrmv / csrmm (brainevent/_csr/float.py:49-150, :559-668; CPU loops :153-207, :670-744; float_csrmv.cu,
 *           float_csrmm.cu) and fcnmv / fcnmm (brainevent/_fcn/float.py:33-134, :136-240) — indptr NULL + row_len = n_conn.
 *   transpose = 0:  out[i, c] = sum over row i of w_j * B[indices[j], c]            B [k, n], out [m, n]   (one writer per row)
 *   transpose = 1:  out[indices[j], c] += w_j * B[i, c] for every stored row i      B [m, n], out [k, n]   (float atomics)
 * B, out and the weights share wdtype (f32 / f64 / f16 / bf16; sums in f32 / f64); row-major, n >= 1 (be_csrmv: n = 1).
 * homo != 0: one shared weight weights[0] (the sum is multiplied once, like the reference's `w * r`).  indices and per-entry
 * weights must be 16-byte aligned (rows are read in aligned groups of four).  nnz_hint: the stored entries if the caller knows
 * them (it selects the lanes per row; 0 = unknown).  workspace: be_csrmm_workspace_bytes (an f32 image of the output for the
 * f16 / bf16 scatter; 256 bytes otherwise).  Zeros of the operand are skipped in the scatter direction (they add nothing).
 * ---------------------------------------------------------------------------------------------- */
int64_t be_csrmm_workspace_bytes(int64_t m, int64_t k, int64_t n, int transpose, int wdtype);
int be_csrmm(const void* weights, int homo, int wdtype, const int32_t* indices, const void* indptr, int indptr_is_i64,
             int64_t row_len, const void* B, void* out, int64_t m, int64_t k, int64_t n, int64_t nnz_hint, int transpose,
             void* workspace, int64_t workspace_bytes, be_stream_t stream);
int be_csrmv(const void* weights, int homo, int wdtype, const int32_t* indices, const void* indptr, int indptr_is_i64,
             int64_t row_len, const void* v, void* out, int64_t m, int64_t k, int64_t nnz_hint, int transpose, void* workspace,
             int64_t workspace_bytes, be_stream_t stream);

/* JIT connectivity against a dense operand: the same on-the-fly matrices as be_binary_jitmv / be_binary_jitmm (same walks, same
 * per-edge weight hashes), every element of the operand counting.
 * replaces: jitsmv / jitsmm (brainevent/_jit_scalar/float.py:838-905, :1331-1420), jitumv / jitumm, jitnmv / jitnmm
 *           (brainevent/_jit_uniform/float.py, brainevent/_jit_normal/float.py).
 *   X [in_len, n] -> out [out_len, n], row-major, wdtype = the weight dtype (sums in float64 for the gather orientation, f32 /
 *   f64 atomics for the scatter orientation).  stride: 32 = the matrix a vector operand sees, 4 = the matrix a matrix operand
 *   sees (the reference draws them differently; be_jitmv_float: n = 1, stride 32).  gather != 0: generator rows = outputs
 *   (corder = True), else generator rows = inputs.  mode / w0 / w1 / clen / seed / shape1 as be_binary_jitmv. */
int64_t be_jitmm_float_workspace_bytes(int64_t shape1, int64_t in_len, int64_t out_len, int64_t n, int gather, int wdtype);
int be_jitmm_float(int mode, double w0, double w1, int wdtype, int64_t clen, uint32_t seed, const void* X, void* out, int64_t shape1,
                   int64_t in_len, int64_t out_len, int64_t n, int stride, int gather, void* workspace, int64_t workspace_bytes,
                   be_stream_t stream);
/* The scatter orientation (gather = 0) through LDS fixed-point sums instead of float atomics — the event-driven scatter's structure
 * with the operand's value as a per-row factor (C3 shape, dense operand: the walk's rate instead of 21 G atomics/s).  BATCH-major
 * operand X_bm [n, in_len] and result out_bm [n, out_len]; scale_exp such that |w|max * |x|max * in_len * 2^scale_exp < 2^62 —
 * the caller knows the operand's largest magnitude.  f32 / f16 / bf16 (f64 operands: be_jitmm_float, whose atomics are f64). */
int64_t be_jitmm_float_scatter_workspace_bytes(int64_t shape1, int64_t out_len, int64_t n, int stride);
int be_jitmm_float_scatter(int mode, double w0, double w1, int wdtype, int64_t clen, uint32_t seed, const void* X_bm, void* out_bm,
                           int64_t shape1, int64_t in_len, int64_t out_len, int64_t n, int stride, int scale_exp, void* workspace,
                           int64_t workspace_bytes, be_stream_t stream);
int be_jitmv_float(int mode, double w0, double w1, int wdtype, int64_t clen, uint32_t seed, const void* v, void* out, int64_t shape1,
                   int64_t in_len, int64_t out_len, int gather, void* workspace, int64_t workspace_bytes, be_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * perm-fused ("indexed") products: be_binary_csrmm_{t,nt} over a RE-INDEXED structure whose slot j carries
 * weights[perm[j]] — the weights stay in their canonical order and only the weights of active rows (t) / of entries whose
 * input fired (nt) are read; no data[perm] pass per call.
 * replaces: binary_indexed_csrmv_hybrid / binary_indexed_csrmm_hybrid (brainevent/_csr/binary_indexed_csrmv_hybrid.cu:16-23,
 *           brainevent/_csr/binary_indexed.py:70, :615).  perm: [nnz] int32 or int64; homo != 0 or perm == NULL: the plain
 *           product (one shared weight ignores perm, as in the reference).  Workspaces as for be_binary_csrmm_{t,nt}.
 * These are the preprocessing-free forms (global atomics / one wave per row).  A structure that is used every step is
 * planned once from the permuted weights instead (be_scatter_plan_*), after which a step reads no weight array at all.
 * ---------------------------------------------------------------------------------------------- */
int be_binary_csrmm_t_indexed(const void* weights, int homo, int wdtype, const int32_t* indices, const void* indptr,
                              int indptr_is_i64, int64_t row_len, const void* perm, int perm_is_i64, const void* spikes_bm,
                              int spike_dtype, void* out_bm, int64_t m, int64_t k, int64_t n_batch, void* workspace,
                              int64_t workspace_bytes, be_stream_t stream);
int be_binary_csrmm_nt_indexed(const void* weights, int homo, int wdtype, const int32_t* indices, const void* indptr,
                               int indptr_is_i64, int64_t row_len, const void* perm, int perm_is_i64, const void* spikes_bm,
                               int spike_dtype, void* out_bm, int64_t m, int64_t k, int64_t n_batch, void* workspace,
                               int64_t workspace_bytes, be_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * binary_densemv / binary_densemm  (BinaryArray @ ndarray)
 * replaces: binary_densemv_{transpose,no_transpose}_{f32..bf16}_{bool,float} (brainevent/_dense/binary_densemv.cu:52-149),
 *           binary_densemm_{transpose,no_transpose}_{…} (brainevent/_dense/binary_densemm.cu:50-162) and the cuBLAS
 *           variants binary_dense{mv,mm}_cublas_{nt,t}_f32_bool (brainevent/_dense/binary_dense_cublas.cu:139-212).
 *   weights : [rows_w, cols_w] row-major, dtype wdtype
 *   transpose = 1: spikes_bm [n_batch, rows_w] -> out_bm [n_batch, cols_w],  out[b, j] = sum_{i: e(s[b,i])} W[i, j]
 *   transpose = 0: spikes_bm [n_batch, cols_w] -> out_bm [n_batch, rows_w],  out[b, i] = sum_{j: e(s[b,j])} W[i, j]
 *   workspace >= be_binary_densemm_workspace_bytes(rows_w, cols_w, n_batch, transpose, wdtype)
 * Sums are accumulated in f32 (f64 for f64 weights) in a fixed order: results are reproducible.
 * ---------------------------------------------------------------------------------------------- */
int64_t be_binary_densemm_workspace_bytes(int64_t rows_w, int64_t cols_w, int64_t n_batch, int transpose, int wdtype);
int be_binary_densemm(const void* weights, int wdtype, const void* spikes_bm, int spike_dtype, void* out_bm,
                      int64_t rows_w, int64_t cols_w, int64_t n_batch, int transpose, void* workspace,
                      int64_t workspace_bytes, be_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * JIT-connectivity products (the matrix is regenerated on the fly, never stored)
 * replaces: jit_scalar_binary_jitsmv.{pack_bool,notrans_*,trans_*} (brainevent/_jit_scalar/binary_jitsmv.cu:107-321),
 *           jit_scalar_binary_jitsmm.{pack,notrans_*,trans_*} (brainevent/_jit_scalar/binary_jitsmm.cu),
 *           and the uniform / normal twins (brainevent/_jit_uniform/binary_jitumv.cu:77-213, binary_jitumm.cu;
 *           brainevent/_jit_normal/binary_jitnmv.cu:105-312, binary_jitnmm.cu).
 *   mode    : 0 scalar (w = w0), 1 uniform (w = w0 + u01(seed,row,col) * w1, i.e. w0 = low, w1 = high - low),
 *             2 normal (w = w0 + n01(seed,row,col) * w1, i.e. w0 = loc, w1 = scale)
 *   clen    : ceil(2 / prob) as the reference computes it (brainevent/_data.py:1212-1245); values < 2 act as 2;
 *             clen <= 0 (prob = 0) yields zeros
 *   shape1  : shape[1] of the logical matrix — keys the chunk width ceil(shape1 / 4) (brainevent/_misc.py:74-122)
 *   gather  : 1 = "notrans" kernel (corder=True): RNG rows are the out_len outputs, the walk runs over in_len;
 *             0 = "trans" kernel (corder=False): RNG rows are the in_len inputs (only active ones are walked),
 *             the walk runs over out_len
 *   scale_exp : (scatter, modes 1 and 2) fixed-point exponent: edge weights are summed as round(w * 2^scale_exp)
 *             in 64-bit integers; the caller guarantees |w|max * 2^scale_exp * in_len < 2^62
 *   mv walks with lane stride 32, mm with lane stride 4 (a different matrix, as in the reference:
 *   brainevent/_misc.py:32-38)
 * ---------------------------------------------------------------------------------------------- */
/* Armed workspaces of the SCATTER orientation (mv and mm).  A scatter call compacts the spikes through counters at the head of its
 * workspace, which it zeroes first — one more launch per call (4.7 us of a 122-us step at BASELINE config C3).  A workspace that was
 * armed once (its counters zeroed here) skips that launch: the call's last kernel leaves the counters at zero again.  The library
 * keys this on the workspace POINTER: disarm it before the memory is freed or reused for anything else; after a call that returned
 * an error, disarm or arm again.  Results are identical either way.  (The reference compacts per call inside its FFI target and
 * has no workspace to keep: brainevent/_jit_scalar/binary_jitsmv.cu:107-125, :175-214.) */
int be_jit_scatter_workspace_arm(void* workspace, int64_t workspace_bytes, be_stream_t stream);
int be_jit_scatter_workspace_disarm(void* workspace);
int64_t be_binary_jitmv_workspace_bytes(int64_t shape1, int64_t in_len, int64_t out_len, int gather);
int be_binary_jitmv(int mode, double w0, double w1, int wdtype, int64_t clen, uint32_t seed, const void* spikes,
                    int spike_dtype, void* out, int64_t shape1, int64_t in_len, int64_t out_len, int gather,
                    int scale_exp, void* workspace, int64_t workspace_bytes, be_stream_t stream);
/* Multi-GPU partition of the scatter ("trans") orientation with no stored state: the walk of a generator row is a
 * set of independent streams keyed by (chunk, lane) (brainevent/_jit_scalar/binary_jitsmv.cu:54-66), and stream
 * (chunk, lane) only ever touches the output columns chunk_start + lane + stride * q.  A rank that owns the classes
 * [class_begin, class_begin + class_count) (class = chunk * stride + lane; be_jit_scatter_classes() of them in all)
 * therefore owns those columns outright: `out` (full length out_len) receives the owned columns and zeros elsewhere,
 * the outputs of the ranks are disjoint and their sum is the unsharded result.  Workspace: the unsharded query. */
int be_jit_scatter_classes(int64_t shape1, int64_t out_len, int stride);
int be_binary_jitmv_sharded(int mode, double w0, double w1, int wdtype, int64_t clen, uint32_t seed, const void* spikes,
                            int spike_dtype, void* out, int64_t shape1, int64_t in_len, int64_t out_len, int class_begin,
                            int class_count, int scale_exp, void* workspace, int64_t workspace_bytes, be_stream_t stream);
/* ... and of the gather ("notrans") orientation, by OUTPUT ROWS: the generator rows are the outputs there, keyed by (seed, row,
 * chunk, lane) alone, so a rank that owns rows [row_begin, row_begin + row_count) computes exactly those outputs from the full
 * (all-gathered) spike vector — `out` has row_count elements; the ranks' slices concatenate to the unsharded result bit for bit.
 * Workspace: be_binary_jitmv_workspace_bytes(shape1, in_len, row_count, 1). */
int be_binary_jitmv_rows(int mode, double w0, double w1, int wdtype, int64_t clen, uint32_t seed, const void* spikes,
                         int spike_dtype, void* out, int64_t shape1, int64_t in_len, int64_t row_begin, int64_t row_count,
                         void* workspace, int64_t workspace_bytes, be_stream_t stream);
int64_t be_binary_jitmm_workspace_bytes(int64_t shape1, int64_t in_len, int64_t out_len, int64_t n_batch, int gather);
int be_binary_jitmm(int mode, double w0, double w1, int wdtype, int64_t clen, uint32_t seed, const void* spikes_bm,
                    int spike_dtype, void* out_bm, int64_t shape1, int64_t in_len, int64_t out_len, int64_t n_batch,
                    int gather, void* workspace, int64_t workspace_bytes, be_stream_t stream);

/* weights of explicitly listed edges: out[i] = the weight edge (rows[i], cols[i]) carries if the walk generates it —
 * rows / cols in the RNG orientation (the generator row and the walk coordinate).  mode / w0 / w1 as above.  These are the
 * per-edge hashes of the reference, brainevent/_numba_random.py:424-430 (uniform01), :433-486 (normal01), evaluated by
 * the device code the products use; the reference pins them with exact values (brainevent/_numba_random_test.py:58-93). */
int be_jit_edge_weights(int mode, double w0, double w1, uint32_t seed, const int32_t* rows, const int32_t* cols, int64_t n,
                        float* out, be_stream_t stream);

/* materialisation of the generator matrix as CSR (rows = walk owners; stride 32 = the mv matrix, 4 = the mm matrix)
 * replaces: the count + fill kernels of brainevent/_jit_scalar/csr.cu (and the uniform / normal twins).
 *   count: row_counts[n_rows] (uint32) <- number of generated edges per row
 *   fill : caller scans the counts into indptr (int64, n_rows + 1) and allocates indices[nnz] (+ weights[nnz] f32 for
 *          modes 1 / 2); cursor[n_rows] is scratch.  Order inside a row is unspecified. */
int be_jitc_csr_count(int64_t clen, uint32_t seed, int64_t shape1, int64_t n_rows, int64_t walk_len, int stride,
                      uint32_t* row_counts, be_stream_t stream);
int be_jitc_csr_fill(int mode, double w0, double w1, int64_t clen, uint32_t seed, int64_t shape1, int64_t n_rows,
                     int64_t walk_len, int stride, const int64_t* indptr, uint32_t* cursor, int32_t* indices,
                     float* weights, be_stream_t stream);

/* named per-family / per-dtype symbols: be_binary_jit{s,u,n}{mv,mm}_{notrans,trans}_{f32,f64,f16,bf16} */
#define BE_JIT_MV_ARGS double w0, double w1, int64_t clen, uint32_t seed, const void *spikes, int spike_dtype,    \
                       void *out, int64_t shape1, int64_t in_len, int64_t out_len, int scale_exp,                 \
                       void *workspace, int64_t workspace_bytes, be_stream_t stream
#define BE_JIT_MM_ARGS double w0, double w1, int64_t clen, uint32_t seed, const void *spikes_bm, int spike_dtype, \
                       void *out_bm, int64_t shape1, int64_t in_len, int64_t out_len, int64_t n_batch,            \
                       void *workspace, int64_t workspace_bytes, be_stream_t stream
#define BE_DECL_JIT_FAMILY(F, W)                        \
  int be_binary_jit##F##mv_notrans_##W(BE_JIT_MV_ARGS);  \
  int be_binary_jit##F##mv_trans_##W(BE_JIT_MV_ARGS);    \
  int be_binary_jit##F##mm_notrans_##W(BE_JIT_MM_ARGS);  \
  int be_binary_jit##F##mm_trans_##W(BE_JIT_MM_ARGS);
#define BE_FOR_JIT_VARIANTS(X)                                  \
  X(s, 0, f32, BE_F32) X(s, 0, f64, BE_F64) X(s, 0, f16, BE_F16) X(s, 0, bf16, BE_BF16) \
  X(u, 1, f32, BE_F32) X(u, 1, f64, BE_F64) X(u, 1, f16, BE_F16) X(u, 1, bf16, BE_BF16) \
  X(n, 2, f32, BE_F32) X(n, 2, f64, BE_F64) X(n, 2, f16, BE_F16) X(n, 2, bf16, BE_BF16)
#define BE_DECL_JIT_VARIANT(F, M, W, WD) BE_DECL_JIT_FAMILY(F, W)
BE_FOR_JIT_VARIANTS(BE_DECL_JIT_VARIANT)

/* per-variant symbols (same grammar as the reference's `// @BE` names); thin wrappers of the above */
#define BE_DENSE_MV_ARGS const void *weights, const void *spikes, void *out, int64_t rows_w, int64_t cols_w,      \
                         void *workspace, int64_t workspace_bytes, be_stream_t stream
#define BE_DENSE_MM_ARGS const void *weights, const void *spikes_bm, void *out_bm, int64_t rows_w, int64_t cols_w, \
                         int64_t n_batch, void *workspace, int64_t workspace_bytes, be_stream_t stream
#define BE_CSR_MV_ARGS const void *weights, const int32_t *indices, const void *indptr, int indptr_is_i64,       \
                       const void *spikes, void *out, int64_t m, int64_t k, void *workspace,                      \
                       int64_t workspace_bytes, be_stream_t stream
#define BE_CSR_MM_ARGS const void *weights, const int32_t *indices, const void *indptr, int indptr_is_i64,       \
                       const void *spikes_bm, void *out_bm, int64_t m, int64_t k, int64_t n_batch,                \
                       void *workspace, int64_t workspace_bytes, be_stream_t stream
/* fixed-number connectivity: indices is [n_pre, n_conn] row-major, weights same shape or [1]
 * (brainevent/_fcn/binary_fcnmv.cu:207-251, binary_fcnmm.cu:835-857, 993-1023).
 * scatter: spikes [n_batch, n_pre] -> out [n_batch, n_post];  gather: spikes [n_batch, n_post] -> out [n_batch, n_pre] */
#define BE_FCN_MV_ARGS const void *weights, const int32_t *indices, const void *spikes, void *out,               \
                       int64_t n_pre, int64_t n_post, int64_t n_conn, void *workspace, int64_t workspace_bytes,   \
                       be_stream_t stream
#define BE_FCN_MM_ARGS const void *weights, const int32_t *indices, const void *spikes_bm, void *out_bm,         \
                       int64_t n_pre, int64_t n_post, int64_t n_conn, int64_t n_batch, void *workspace,           \
                       int64_t workspace_bytes, be_stream_t stream

#define BE_DECL_VARIANT(W, WD, S, SD)                            \
  int be_binary_csrmv_t_homo_##W##_##S(BE_CSR_MV_ARGS);           \
  int be_binary_csrmv_t_hetero_##W##_##S(BE_CSR_MV_ARGS);         \
  int be_binary_csrmv_nt_homo_##W##_##S(BE_CSR_MV_ARGS);          \
  int be_binary_csrmv_nt_hetero_##W##_##S(BE_CSR_MV_ARGS);        \
  int be_binary_csrmm_t_homo_##W##_##S(BE_CSR_MM_ARGS);           \
  int be_binary_csrmm_t_hetero_##W##_##S(BE_CSR_MM_ARGS);         \
  int be_binary_csrmm_nt_homo_##W##_##S(BE_CSR_MM_ARGS);          \
  int be_binary_csrmm_nt_hetero_##W##_##S(BE_CSR_MM_ARGS);        \
  int be_binary_fcnmv_scatter_homo_##W##_##S(BE_FCN_MV_ARGS);     \
  int be_binary_fcnmv_scatter_hetero_##W##_##S(BE_FCN_MV_ARGS);   \
  int be_binary_fcnmv_gather_homo_##W##_##S(BE_FCN_MV_ARGS);      \
  int be_binary_fcnmv_gather_hetero_##W##_##S(BE_FCN_MV_ARGS);    \
  int be_binary_fcnmm_scatter_homo_##W##_##S(BE_FCN_MM_ARGS);     \
  int be_binary_fcnmm_scatter_hetero_##W##_##S(BE_FCN_MM_ARGS);   \
  int be_binary_fcnmm_gather_homo_##W##_##S(BE_FCN_MM_ARGS);      \
  int be_binary_fcnmm_gather_hetero_##W##_##S(BE_FCN_MM_ARGS);    \
  int be_binary_densemv_transpose_##W##_##S(BE_DENSE_MV_ARGS);    \
  int be_binary_densemv_no_transpose_##W##_##S(BE_DENSE_MV_ARGS); \
  int be_binary_densemm_transpose_##W##_##S(BE_DENSE_MM_ARGS);    \
  int be_binary_densemm_no_transpose_##W##_##S(BE_DENSE_MM_ARGS);

#define BE_FOR_ALL_VARIANTS(X) \
  X(f32, BE_F32, bool, BE_SPIKE_BOOL)   X(f32, BE_F32, float, BE_SPIKE_FLOAT)   \
  X(f64, BE_F64, bool, BE_SPIKE_BOOL)   X(f64, BE_F64, float, BE_SPIKE_FLOAT)   \
  X(f16, BE_F16, bool, BE_SPIKE_BOOL)   X(f16, BE_F16, float, BE_SPIKE_FLOAT)   \
  X(bf16, BE_BF16, bool, BE_SPIKE_BOOL) X(bf16, BE_BF16, float, BE_SPIKE_FLOAT)

BE_FOR_ALL_VARIANTS(BE_DECL_VARIANT)

#ifdef __cplusplus
}
#endif
#endif /* BRAINEVENT_AMD_H */
