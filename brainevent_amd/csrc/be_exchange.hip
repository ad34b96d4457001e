// be_exchange.hip — the one collective of the multi-GPU path behind the C ABI: a bit-packed all-gather of the spike vector
// over RCCL (xGMI), one process per GPU.  SURVEY.md §8(e): rank g owns a post slice of the matrix and produces the spikes
// of its 1/G of the pre population; one all-gather rebuilds the full vector on every rank; the scatter is local.
// The reference has no distributed path; a JAX-side binder (not torch) reaches the exchange through these entry points:
// rank 0 obtains an id (be_exchange_get_unique_id), the binder ships its 128 bytes to the other processes by its own
// means, every process calls be_exchange_init.  RCCL is loaded at run time (dlopen "librccl.so"): the library itself keeps
// no link-time dependency on it, and a process that has already loaded RCCL (PyTorch ships one) shares that copy.
#include "be_common.h"
#include <dlfcn.h>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <atomic>
#include <chrono>
#include <mutex>
#include <string>

namespace {

struct NcclUniqueId { char internal[128]; };           // ncclUniqueId of rccl.h (NCCL_UNIQUE_ID_BYTES = 128)
typedef void* NcclComm;
constexpr int kNcclUint32 = 3;                          // ncclDataType_t: ncclUint32

struct Rccl {
  void* handle = nullptr;
  int (*GetUniqueId)(NcclUniqueId*) = nullptr;
  int (*CommInitRank)(NcclComm*, int, NcclUniqueId, int) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, NcclComm, hipStream_t) = nullptr;
  int (*CommDestroy)(NcclComm) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};

Rccl* rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    // a copy of RCCL the process has already mapped (PyTorch ships its own under torch/lib) is reused: two copies in one
    // process would each run their own bootstrap
    std::string mapped;
    if (FILE* maps = fopen("/proc/self/maps", "r")) {
      char line[4096];
      while (mapped.empty() && fgets(line, sizeof(line), maps)) {
        const char* path = strchr(line, '/');
        if (path && strstr(path, "librccl.so")) {
          mapped = path;
          while (!mapped.empty() && (mapped.back() == '\n' || mapped.back() == ' ')) mapped.pop_back();
        }
      }
      fclose(maps);
    }
    const char* names[] = {getenv("BE_RCCL_LIB"), mapped.empty() ? nullptr : mapped.c_str(), "librccl.so", "librccl.so.1",
                           "/opt/rocm/lib/librccl.so"};
    for (const char* n : names) {
      if (!n || !*n) continue;
      r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
      if (r.handle) break;
    }
    if (!r.handle) return;
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(r.handle, "ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(r.handle, "ncclCommInitRank"));
    r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(r.handle, "ncclAllGather"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.handle, "ncclCommDestroy"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.handle, "ncclGetErrorString"));
  });
  return (r.handle && r.GetUniqueId && r.CommInitRank && r.AllGather && r.CommDestroy) ? &r : nullptr;
}

struct Exchange {
  NcclComm comm;
  int world, rank;
  int64_t n_pre, words_per_rank;      // every rank owns the same whole number of 32-bit words (the last may hold fewer spikes)
  uint32_t* local_words;              // device: this rank's packed slice (words_per_rank)
  // pipelined exchange (be_exchange_post / _wait): the library's own stream, two result buffers, their events
  hipStream_t side = nullptr;
  hipEvent_t ev_in = nullptr, ev_done[2] = {nullptr, nullptr};
  hipEvent_t ev_free[2] = {nullptr, nullptr};  // recorded by be_exchange_release on the consumer's stream
  bool released[2] = {false, false};
  uint32_t* post_local[2] = {nullptr, nullptr};
  uint32_t* post_full[2] = {nullptr, nullptr};
  // be_exchange_post_ids: the gathered words of a slot compacted into an id list on the exchange's stream (n_pre ids + a counter)
  uint32_t* post_ids[2] = {nullptr, nullptr};
  uint32_t* post_count[2] = {nullptr, nullptr};
  bool has_ids[2] = {false, false};
};

#define BE_RCCL(call)                                                                                     \
  do {                                                                                                    \
    const int rc__ = (call);                                                                              \
    if (rc__ != 0) {                                                                                      \
      be_set_error(std::string(__func__) + ": " #call " -> " + (R->GetErrorString ? R->GetErrorString(rc__) : "RCCL error")); \
      return BE_ERR_HIP;                                                                                  \
    }                                                                                                     \
  } while (0)

}  // namespace

static hipError_t exchange_emulated_latency(hipStream_t st, uint32_t* scratch);      // (measurement hook, defined below)

extern "C" {

int be_pack_spikes(const void* spikes, int spike_dtype, int64_t n, uint32_t* bits, be_stream_t stream);
int be_compact_spikes(const void* spikes, int spike_dtype, int64_t n, uint32_t* active_ids, uint32_t* count, be_stream_t stream);

int be_exchange_unique_id_bytes(void) { return (int)sizeof(NcclUniqueId); }

int be_exchange_get_unique_id(void* id_host) {
  BE_REQUIRE(id_host != nullptr, BE_ERR_INVALID, "null pointer");
  Rccl* R = rccl();
  BE_REQUIRE(R != nullptr, BE_ERR_UNSUPPORTED, "librccl.so could not be loaded (set BE_RCCL_LIB)");
  BE_RCCL(R->GetUniqueId(static_cast<NcclUniqueId*>(id_host)));
  return BE_OK;
}

// The partition of the pre population the exchange rests on, as a pure function of (n_pre, world, rank): every rank owns the same
// whole number of 32-bit words w = ceil(ceil(n_pre / 32) / world); rank r's slice is [r * w * 32, (r + 1) * w * 32) clipped to
// n_pre — the last owners may hold fewer spikes, or none.  Everything below (init, slice, gather, post) goes through it.
struct ExSlice {
  int64_t words_per_rank, lo, hi, n_local, used_words;
};
static inline ExSlice ex_slice_of(int64_t n_pre, int world, int rank) {
  ExSlice s;
  s.words_per_rank = (((n_pre + 31) / 32) + world - 1) / world;
  const int64_t lo = (int64_t)rank * s.words_per_rank * 32, hi = lo + s.words_per_rank * 32;
  s.lo = lo < n_pre ? lo : n_pre;
  s.hi = hi < n_pre ? hi : n_pre;
  s.n_local = s.hi - s.lo;
  s.used_words = (s.n_local + 31) / 32;
  return s;
}

int be_exchange_slice_for(int64_t n_pre, int world, int rank, int64_t* lo_host, int64_t* hi_host, int64_t* words_per_rank_host) {
  BE_REQUIRE(world >= 1 && rank >= 0 && rank < world && n_pre >= 0, BE_ERR_INVALID, "bad world / rank / n_pre");
  const ExSlice s = ex_slice_of(n_pre, world, rank);
  if (lo_host) *lo_host = s.lo;
  if (hi_host) *hi_host = s.hi;
  if (words_per_rank_host) *words_per_rank_host = s.words_per_rank;
  return BE_OK;
}

int be_exchange_init(const void* id_host, int world, int rank, int64_t n_pre, void** exchange_host_out) {
  BE_REQUIRE(id_host && exchange_host_out, BE_ERR_INVALID, "null pointer");
  BE_REQUIRE(world >= 1 && rank >= 0 && rank < world && n_pre >= 0, BE_ERR_INVALID, "bad world / rank / n_pre");
  Rccl* R = rccl();
  BE_REQUIRE(R != nullptr, BE_ERR_UNSUPPORTED, "librccl.so could not be loaded (set BE_RCCL_LIB)");
  Exchange* ex = new Exchange();
  ex->world = world;
  ex->rank = rank;
  ex->n_pre = n_pre;
  ex->words_per_rank = ex_slice_of(n_pre, world, rank).words_per_rank;
  ex->local_words = nullptr;
  NcclUniqueId id;
  memcpy(&id, id_host, sizeof(id));
  const int rc = R->CommInitRank(&ex->comm, world, id, rank);
  if (rc != 0) {
    be_set_error(std::string("be_exchange_init: ncclCommInitRank -> ") + (R->GetErrorString ? R->GetErrorString(rc) : "RCCL error"));
    delete ex;
    return BE_ERR_HIP;
  }
  if (hipMalloc(&ex->local_words, (size_t)(ex->words_per_rank > 0 ? ex->words_per_rank : 1) * 4) != hipSuccess) {
    be_set_error("be_exchange_init: hipMalloc failed");
    R->CommDestroy(ex->comm);
    delete ex;
    return BE_ERR_HIP;
  }
  *exchange_host_out = ex;
  return BE_OK;
}

int be_exchange_slice(const void* exchange, int rank, int64_t* lo_host, int64_t* hi_host) {
  BE_REQUIRE(exchange && lo_host && hi_host, BE_ERR_INVALID, "null pointer");
  const Exchange* ex = static_cast<const Exchange*>(exchange);
  BE_REQUIRE(rank >= 0 && rank < ex->world, BE_ERR_INVALID, "rank out of range");
  const ExSlice sl = ex_slice_of(ex->n_pre, ex->world, rank);
  *lo_host = sl.lo;
  *hi_host = sl.hi;
  return BE_OK;
}

int64_t be_exchange_full_words(const void* exchange) {
  if (!exchange) return 0;
  const Exchange* ex = static_cast<const Exchange*>(exchange);
  return ex->words_per_rank * ex->world;
}

int be_exchange_allgather_bits(void* exchange, const void* local_spikes, int spike_dtype, uint32_t* full_bits,
                               be_stream_t stream) {
  BE_REQUIRE(exchange && full_bits, BE_ERR_INVALID, "null pointer");
  Exchange* ex = static_cast<Exchange*>(exchange);
  Rccl* R = rccl();
  BE_REQUIRE(R != nullptr, BE_ERR_UNSUPPORTED, "librccl.so could not be loaded");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const ExSlice sl = ex_slice_of(ex->n_pre, ex->world, ex->rank);
  const int64_t n_local = sl.n_local, used = sl.used_words;
  BE_REQUIRE(n_local == 0 || local_spikes != nullptr, BE_ERR_INVALID, "null pointer");
  // BE_SPIKE_BITS: the producer already emits the slice as words (be_lif_coba_step's spike_bits_out, a BitPackedBinary): a
  // slice that fills its words is gathered from where it lies — no pack launch, no copy
  const uint32_t* send = ex->local_words;
  if (spike_dtype == BE_SPIKE_BITS && used == ex->words_per_rank && n_local > 0) {
    send = static_cast<const uint32_t*>(local_spikes);
  } else {
    if (used < ex->words_per_rank)      // words of the slice that hold no spike of the population
      BE_HIP(be_fill_async(ex->local_words + used, 0, (size_t)(ex->words_per_rank - used) * 4, st));
    if (n_local > 0) {
      if (spike_dtype == BE_SPIKE_BITS) {
        BE_HIP(hipMemcpyAsync(ex->local_words, local_spikes, (size_t)used * 4, hipMemcpyDeviceToDevice, st));
      } else {
        const int rc = be_pack_spikes(local_spikes, spike_dtype, n_local, ex->local_words, stream);
        if (rc != BE_OK) return rc;
      }
    }
  }
  if (ex->words_per_rank > 0)
    BE_RCCL(R->AllGather(send, full_bits, (size_t)ex->words_per_rank, kNcclUint32, ex->comm, st));
  BE_HIP(exchange_emulated_latency(st, ex->local_words));
  return BE_OK;
}

// Pipelined form: the all-gather of step t + 1 runs on the exchange's own stream while the caller's stream scatters step t
// (legitimate when synaptic delays are at least two steps: the spikes a step delivers were emitted before the previous
// step started).  post: the side stream waits for what `producer_stream` has queued so far (the spikes), packs and gathers into
// the exchange's buffer `slot` (0 / 1, alternate them) and records the slot's event.  wait: `consumer_stream` waits for that
// event; *full_bits_out is the slot's device buffer (be_exchange_full_words words), valid until the slot is posted again.
}  // extern "C"

// ---- a hardware queue of its own for the exchange's stream ------------------------------------------------------------------------
// The runtime multiplexes a process's streams onto a few hardware queues (GPU_MAX_HW_QUEUES, default 4), round-robin in creation
// order.  A side stream that lands on the queue of the stream it is meant to overlap with does not overlap at all: every gather
// serialises behind the scatter in front of it and the cross-stream events only add their packets.  Round 5's driver bench measured
// exactly that (one rank of eight: pipelined 52.2 us against 36.1 sequential; the same legs run alone, or with GPU_MAX_HW_QUEUES=8:
// 39-40 us — profiles/r06_rank_step_schedules.txt): by the time the bench reached those legs the process had used its four queues.
// So the stream is CHOSEN: up to four candidates are created (consecutive candidates sit on consecutive queues) and each is probed
// once — a 40-us spin kernel on the consumer's stream and one on the candidate, timed together; two that overlap finish in about
// one kernel's time, two on one queue in two — and the first that overlaps is kept.  ~0.5 ms, once per exchange handle; skipped
// (first candidate) while the consumer's stream is capturing or when BE_EXCHANGE_PROBE=0.
__global__ void k_exchange_probe_spin(uint32_t ticks, uint32_t* sink) {
  const uint64_t t0 = wall_clock64();            // 100 MHz
  uint32_t it = 0;
  while (wall_clock64() - t0 < ticks && it < (1u << 20)) { __builtin_amdgcn_s_sleep(16); ++it; }
  if (it == 0xffffffffu) sink[0] = it;           // (never)
}

static hipError_t exchange_pick_side_stream(hipStream_t consumer, uint32_t* scratch, hipStream_t* out) {
  static const bool probe_off = [] { const char* e = getenv("BE_EXCHANGE_PROBE"); return e && e[0] == '0'; }();
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  (void)hipStreamIsCapturing(consumer, &cap);
  hipStream_t cand[4] = {nullptr, nullptr, nullptr, nullptr};
  hipError_t e = hipStreamCreateWithFlags(&cand[0], hipStreamNonBlocking);
  if (e != hipSuccess) return e;
  int best = 0;
  if (!probe_off && cap == hipStreamCaptureStatusNone) {
    constexpr uint32_t kTicks = 4000;             // 40 us
    double best_us = 1e30;
    (void)hipStreamSynchronize(consumer);
    for (int c = 0; c < 4; ++c) {
      if (c > 0 && hipStreamCreateWithFlags(&cand[c], hipStreamNonBlocking) != hipSuccess) { cand[c] = nullptr; break; }
      double us = 1e30;
      for (int rep = 0; rep < 2; ++rep) {         // (the first launch of the kernel pays its load: the second pair is the measurement)
        const auto t0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(k_exchange_probe_spin, dim3(1), dim3(64), 0, consumer, kTicks, scratch);
        hipLaunchKernelGGL(k_exchange_probe_spin, dim3(1), dim3(64), 0, cand[c], kTicks, scratch);
        (void)hipStreamSynchronize(consumer);
        (void)hipStreamSynchronize(cand[c]);
        us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
      }
      if (us < best_us) { best_us = us; best = c; }
      if (us < 1.6 * kTicks / 100.0) break;       // overlapped: a queue of its own
    }
    (void)hipGetLastError();
  }
  for (int c = 0; c < 4; ++c)
    if (c != best && cand[c]) (void)hipStreamDestroy(cand[c]);
  *out = cand[best];
  return hipSuccess;
}

// MEASUREMENT HOOK: BE_EXCHANGE_EMULATE_US=<us> appends a spin kernel of that length behind every all-gather, on the stream the
// all-gather runs on — a stand-in for the latency of a REAL multi-rank all-gather (12-15 us for the 16 KB per rank of C2 over
// xGMI) on a box that has one GPU, so that both schedules can be timed against an exchange of realistic length
// (tools/rank_step_lab.py).  Unset (the default): nothing is launched.
static std::atomic<uint32_t> g_emulate_ticks{[] { const char* e = getenv("BE_EXCHANGE_EMULATE_US"); return e ? (uint32_t)(atof(e) * 100.0) : 0u; }()};
static hipError_t exchange_emulated_latency(hipStream_t st, uint32_t* scratch) {
  const uint32_t ticks = g_emulate_ticks.load(std::memory_order_relaxed);
  if (ticks == 0u) return hipSuccess;
  hipLaunchKernelGGL(k_exchange_probe_spin, dim3(1), dim3(64), 0, st, ticks, scratch);
  return hipGetLastError();
}

// Events of the pipelined exchange order work of ONE device (producer kernel -> gather on the side stream; gather -> scatter on the
// consumer's stream): an agent-scope release is all they need.  hipEventDisableSystemFence drops the system-scope fence a default
// event performs when it is recorded — measured 2.4 us per step of the pipelined schedule (tools/ubench/ubench7.hip,
// profiles/r06_ubench7_cross_stream_dependency.txt: 46.4 -> 44.0 us).  Peers never synchronise through these events (RCCL fences
// what it sends itself).  BE_EXCHANGE_SYSTEM_FENCE=1 restores default events.
static unsigned exchange_event_flags() {
  static const bool sys = [] { const char* e = getenv("BE_EXCHANGE_SYSTEM_FENCE"); return e && e[0] == '1'; }();
  return hipEventDisableTiming | (sys ? 0u : (unsigned)hipEventDisableSystemFence);
}

static int exchange_post(void* exchange, const void* local_spikes, int spike_dtype, int slot, be_stream_t producer_stream, bool with_ids) {
  BE_REQUIRE(exchange, BE_ERR_INVALID, "null pointer");
  BE_REQUIRE(slot == 0 || slot == 1, BE_ERR_INVALID, "slot must be 0 or 1");
  Exchange* ex = static_cast<Exchange*>(exchange);
  Rccl* R = rccl();
  BE_REQUIRE(R != nullptr, BE_ERR_UNSUPPORTED, "librccl.so could not be loaded");
  if (!ex->side) {
    // everything is created into locals and published together: a failure half-way leaves the handle as it was
    // (a later call starts over) instead of a stream without its buffers
    hipStream_t side = nullptr;
    hipEvent_t ev_in = nullptr, ev_done[2] = {nullptr, nullptr}, ev_free[2] = {nullptr, nullptr};
    uint32_t* pl[2] = {nullptr, nullptr};
    uint32_t* pf[2] = {nullptr, nullptr};
    const size_t wl = (size_t)(ex->words_per_rank > 0 ? ex->words_per_rank : 1) * 4;
    const unsigned evf = exchange_event_flags();
    hipError_t e = exchange_pick_side_stream(static_cast<hipStream_t>(producer_stream), ex->local_words, &side);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ev_in, evf);
    for (int i = 0; i < 2 && e == hipSuccess; ++i) {
      e = hipEventCreateWithFlags(&ev_done[i], evf);
      if (e == hipSuccess) e = hipEventCreateWithFlags(&ev_free[i], evf);
      if (e == hipSuccess) e = hipMalloc(&pl[i], wl);
      if (e == hipSuccess) e = hipMalloc(&pf[i], wl * ex->world);
    }
    if (e != hipSuccess) {
      for (int i = 0; i < 2; ++i) {
        if (pf[i]) (void)hipFree(pf[i]);
        if (pl[i]) (void)hipFree(pl[i]);
        if (ev_free[i]) (void)hipEventDestroy(ev_free[i]);
        if (ev_done[i]) (void)hipEventDestroy(ev_done[i]);
      }
      if (ev_in) (void)hipEventDestroy(ev_in);
      if (side) (void)hipStreamDestroy(side);
      be_set_error(std::string("be_exchange_post: creating the exchange stream / events / buffers -> ") + hipGetErrorString(e));
      return BE_ERR_HIP;
    }
    ex->ev_in = ev_in;
    for (int i = 0; i < 2; ++i) { ex->ev_done[i] = ev_done[i]; ex->ev_free[i] = ev_free[i]; ex->post_local[i] = pl[i]; ex->post_full[i] = pf[i]; }
    ex->side = side;
  }
  // the side stream waits for what the producer's stream has queued so far: the spikes — and, when the consumer works on that
  // same stream, the consumer's reads of this slot's previous contents.  A consumer on ANOTHER stream says when it is done
  // with a slot through be_exchange_release; the gather into the slot then waits for that as well.
  BE_HIP(hipEventRecord(ex->ev_in, static_cast<hipStream_t>(producer_stream)));
  BE_HIP(hipStreamWaitEvent(ex->side, ex->ev_in, 0));
  if (ex->released[slot]) {
    BE_HIP(hipStreamWaitEvent(ex->side, ex->ev_free[slot], 0));
    ex->released[slot] = false;
  }
  const ExSlice sl = ex_slice_of(ex->n_pre, ex->world, ex->rank);
  const int64_t n_local = sl.n_local, used = sl.used_words;
  BE_REQUIRE(n_local == 0 || local_spikes != nullptr, BE_ERR_INVALID, "null pointer");
  // (BE_SPIKE_BITS: see be_exchange_allgather_bits; the caller keeps the words unchanged until the slot's wait has returned)
  const uint32_t* send = ex->post_local[slot];
  if (spike_dtype == BE_SPIKE_BITS && used == ex->words_per_rank && n_local > 0) {
    send = static_cast<const uint32_t*>(local_spikes);
  } else {
    if (used < ex->words_per_rank) BE_HIP(be_fill_async(ex->post_local[slot] + used, 0, (size_t)(ex->words_per_rank - used) * 4, ex->side));
    if (n_local > 0) {
      if (spike_dtype == BE_SPIKE_BITS) {
        BE_HIP(hipMemcpyAsync(ex->post_local[slot], local_spikes, (size_t)used * 4, hipMemcpyDeviceToDevice, ex->side));
      } else {
        const int rc = be_pack_spikes(local_spikes, spike_dtype, n_local, ex->post_local[slot], ex->side);
        if (rc != BE_OK) return rc;
      }
    }
  }
  if (ex->words_per_rank > 0)
    BE_RCCL(R->AllGather(send, ex->post_full[slot], (size_t)ex->words_per_rank, kNcclUint32, ex->comm, ex->side));
  BE_HIP(exchange_emulated_latency(ex->side, ex->local_words));
  ex->has_ids[slot] = false;
  if (with_ids) {
    // the gathered words -> the list of active pre neurons, right behind the all-gather on the exchange's own stream: the consumer's
    // scatter takes it as BE_SPIKE_IDS and runs no compaction of its own (fused or launched) on its critical path
    for (int i = 0; i < 2; ++i) {
      if (ex->post_ids[i]) continue;
      uint32_t* ids = nullptr;
      uint32_t* cnt = nullptr;
      hipError_t e = hipMalloc(&ids, (size_t)(ex->n_pre > 0 ? ex->n_pre : 1) * 4);
      if (e == hipSuccess) e = hipMalloc(&cnt, 256);
      if (e != hipSuccess) {
        if (ids) (void)hipFree(ids);
        be_set_error(std::string("be_exchange_post_ids: id list buffers -> ") + hipGetErrorString(e));
        return BE_ERR_HIP;
      }
      ex->post_ids[i] = ids;
      ex->post_count[i] = cnt;
    }
    const int rc = be_compact_spikes(ex->post_full[slot], BE_SPIKE_BITS, ex->n_pre, ex->post_ids[slot], ex->post_count[slot], ex->side);
    if (rc != BE_OK) return rc;
    ex->has_ids[slot] = true;
  }
  BE_HIP(hipEventRecord(ex->ev_done[slot], ex->side));
  return BE_OK;
}

extern "C" {

int be_exchange_emulate_latency_us(double us) {
  BE_REQUIRE(us >= 0.0 && us <= 1000.0, BE_ERR_INVALID, "0 <= us <= 1000");
  g_emulate_ticks.store((uint32_t)(us * 100.0), std::memory_order_relaxed);
  return BE_OK;
}

int be_exchange_post(void* exchange, const void* local_spikes, int spike_dtype, int slot, be_stream_t producer_stream) {
  return exchange_post(exchange, local_spikes, spike_dtype, slot, producer_stream, false);
}

int be_exchange_post_ids(void* exchange, const void* local_spikes, int spike_dtype, int slot, be_stream_t producer_stream) {
  return exchange_post(exchange, local_spikes, spike_dtype, slot, producer_stream, true);
}

int be_exchange_wait_ids(void* exchange, int slot, be_spike_ids_t* ids_out, const uint32_t** full_bits_out, be_stream_t consumer_stream) {
  BE_REQUIRE(exchange && ids_out, BE_ERR_INVALID, "null pointer");
  BE_REQUIRE(slot == 0 || slot == 1, BE_ERR_INVALID, "slot must be 0 or 1");
  Exchange* ex = static_cast<Exchange*>(exchange);
  BE_REQUIRE(ex->side != nullptr && ex->has_ids[slot], BE_ERR_INVALID, "the slot was not posted with be_exchange_post_ids");
  BE_HIP(hipStreamWaitEvent(static_cast<hipStream_t>(consumer_stream), ex->ev_done[slot], 0));
  ids_out->active_ids = ex->post_ids[slot];
  ids_out->n_active = ex->post_count[slot];
  if (full_bits_out) *full_bits_out = ex->post_full[slot];
  return BE_OK;
}

int be_exchange_wait(void* exchange, int slot, const uint32_t** full_bits_out, be_stream_t consumer_stream) {
  BE_REQUIRE(exchange && full_bits_out, BE_ERR_INVALID, "null pointer");
  BE_REQUIRE(slot == 0 || slot == 1, BE_ERR_INVALID, "slot must be 0 or 1");
  Exchange* ex = static_cast<Exchange*>(exchange);
  BE_REQUIRE(ex->side != nullptr, BE_ERR_INVALID, "nothing was posted");
  BE_HIP(hipStreamWaitEvent(static_cast<hipStream_t>(consumer_stream), ex->ev_done[slot], 0));
  *full_bits_out = ex->post_full[slot];
  return BE_OK;
}

// The consumer has queued its last read of the slot's buffer on `consumer_stream`: the next be_exchange_post into that slot
// waits for it.  Only needed when the consumer's stream is not the stream passed to be_exchange_post as producer_stream.
int be_exchange_release(void* exchange, int slot, be_stream_t consumer_stream) {
  BE_REQUIRE(exchange, BE_ERR_INVALID, "null pointer");
  BE_REQUIRE(slot == 0 || slot == 1, BE_ERR_INVALID, "slot must be 0 or 1");
  Exchange* ex = static_cast<Exchange*>(exchange);
  BE_REQUIRE(ex->side != nullptr, BE_ERR_INVALID, "nothing was posted");
  BE_HIP(hipEventRecord(ex->ev_free[slot], static_cast<hipStream_t>(consumer_stream)));
  ex->released[slot] = true;
  return BE_OK;
}

int be_exchange_destroy(void* exchange) {
  if (!exchange) return BE_OK;
  Exchange* ex = static_cast<Exchange*>(exchange);
  Rccl* R = rccl();
  if (ex->side) {
    (void)hipStreamSynchronize(ex->side);
    for (int i = 0; i < 2; ++i) {
      if (ex->ev_done[i]) (void)hipEventDestroy(ex->ev_done[i]);
      if (ex->ev_free[i]) (void)hipEventDestroy(ex->ev_free[i]);
      if (ex->post_local[i]) (void)hipFree(ex->post_local[i]);
      if (ex->post_full[i]) (void)hipFree(ex->post_full[i]);
      if (ex->post_ids[i]) (void)hipFree(ex->post_ids[i]);
      if (ex->post_count[i]) (void)hipFree(ex->post_count[i]);
    }
    if (ex->ev_in) (void)hipEventDestroy(ex->ev_in);
    (void)hipStreamDestroy(ex->side);
  }
  if (ex->local_words) (void)hipFree(ex->local_words);
  if (R) (void)R->CommDestroy(ex->comm);
  delete ex;
  return BE_OK;
}

}  // extern "C"
