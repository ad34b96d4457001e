// be_api.hip — library-level entry points of the C ABI (version, error string, device count).
#include "be_common.h"
#include "../../include/brainevent_amd.h"

static thread_local std::string g_last_error;

void be_set_error(const std::string& msg) { g_last_error = msg; }

extern "C" {

int be_version(void) { return 100; }  // 0.1.0

const char* be_last_error(void) { return g_last_error.c_str(); }

const char* be_build_arch(void) { return "gfx950"; }

int be_profile_enable(int max_records);

/* releases what the library holds on to between calls (the profiling events); exchange handles are released by
 * be_exchange_destroy.  Every other buffer belongs to the caller. */
int be_shutdown(void) { return be_profile_enable(0); }

int be_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    be_set_error(std::string("be_device_count: ") + hipGetErrorString(e));
    return BE_ERR_HIP;
  }
  return n;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------
// in-library kernel timing: HIP event pairs recorded around the dominant kernel of an op, on the
// stream that kernel is launched on (bench.py reads them back for the roofline figure).
// ------------------------------------------------------------------------------------------------
#include <vector>
#include <tuple>
#include <mutex>

namespace {
std::mutex g_prof_mu;
std::vector<hipEvent_t> g_prof_ev;   // 2 * capacity events
int g_prof_cap = 0;
int g_prof_n = 0;
}  // namespace

// called by the launch helpers; returns the slot or -1
namespace {
__global__ void __launch_bounds__(256) k_fill_bytes(unsigned char* __restrict__ p, size_t n, unsigned char v) {
  const size_t stride = (size_t)gridDim.x * blockDim.x, t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t head = (16 - (reinterpret_cast<uintptr_t>(p) & 15)) & 15;       // bytes before the first 16-B boundary
  const size_t h = head < n ? head : n;
  const size_t n16 = (n - h) >> 4;
  const unsigned w = 0x01010101u * v;
  uint4* q = reinterpret_cast<uint4*>(p + h);
  for (size_t i = t; i < n16; i += stride) q[i] = make_uint4(w, w, w, w);
  for (size_t i = t; i < h; i += stride) p[i] = v;
  for (size_t i = h + (n16 << 4) + t; i < n; i += stride) p[i] = v;
}
}  // namespace

hipError_t be_allow_lds(const void* kernel, int bytes) {
  static std::mutex mu;
  static std::vector<std::tuple<const void*, int, int>> granted;      // (kernel, device, largest size granted)
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  std::lock_guard<std::mutex> lock(mu);
  for (auto& g : granted) {
    if (std::get<0>(g) == kernel && std::get<1>(g) == dev) {
      if (std::get<2>(g) >= bytes) return hipSuccess;
      e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
      if (e == hipSuccess) std::get<2>(g) = bytes;
      return e;
    }
  }
  e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess) granted.emplace_back(kernel, dev, bytes);
  return e;
}

// static LDS bytes of a kernel (hipFuncGetAttributes once per kernel; -1 on error).  Kernels that address their dynamic LDS
// from address 0 (spike_mask AT0 in be_csr.hip) are only correct while this is 0.
int be_static_lds_bytes(const void* kernel) {
  static std::mutex mu;
  static std::vector<std::pair<const void*, int>> seen;
  std::lock_guard<std::mutex> lock(mu);
  for (auto& k : seen)
    if (k.first == kernel) return k.second;
  hipFuncAttributes fa;
  if (hipFuncGetAttributes(&fa, kernel) != hipSuccess) return -1;
  seen.emplace_back(kernel, (int)fa.sharedSizeBytes);
  return (int)fa.sharedSizeBytes;
}

hipError_t be_fill_async(void* p, int byte_value, size_t bytes, hipStream_t st) {
  if (bytes == 0) return hipSuccess;
  size_t blocks = (bytes / 16 + 255) / 256;
  blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
  hipLaunchKernelGGL(k_fill_bytes, dim3((unsigned)blocks), dim3(256), 0, st, static_cast<unsigned char*>(p), bytes,
                     (unsigned char)byte_value);
  return hipGetLastError();
}

int be_prof_begin(hipStream_t st) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if (g_prof_n >= g_prof_cap) return -1;
  const int slot = g_prof_n++;
  if (hipEventRecord(g_prof_ev[2 * slot], st) != hipSuccess) return -1;
  return slot;
}
void be_prof_end(int slot, hipStream_t st) {
  if (slot < 0) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  (void)hipEventRecord(g_prof_ev[2 * slot + 1], st);
}

extern "C" {

int be_profile_enable(int max_records) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (hipEvent_t e : g_prof_ev) (void)hipEventDestroy(e);
  g_prof_ev.clear();
  g_prof_cap = 0;
  g_prof_n = 0;
  if (max_records <= 0) return BE_OK;
  g_prof_ev.resize(2 * (size_t)max_records);
  for (auto& e : g_prof_ev) {
    hipError_t rc = hipEventCreate(&e);
    if (rc != hipSuccess) {
      be_set_error(std::string("be_profile_enable: ") + hipGetErrorString(rc));
      return BE_ERR_HIP;
    }
  }
  g_prof_cap = max_records;
  return BE_OK;
}

// ------------------------------------------------------------------------------------------------
// read-only streaming ceiling of this device, measured in the caller's run (bench.py: `roofline.read_ceiling_GBps`):
// 16 B per lane, grid-stride, 2048 workgroups of 256 — the shape profiles/r01_ubench.txt found fastest.
// ------------------------------------------------------------------------------------------------
}  // extern "C"
namespace {
typedef unsigned diag_u4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) k_diag_stream_read(const diag_u4* __restrict__ p, size_t n16, unsigned* __restrict__ sink) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  unsigned acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) {
    const diag_u4 v = __builtin_nontemporal_load(p + i);
    acc += v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x9e3779b9u) sink[0] = acc;      // (keeps the loads; practically never taken)
}
}  // namespace
extern "C" {

int be_diag_stream_read(const void* buf, int64_t bytes, int repeats, void* sink4, float* ms_per_pass, be_stream_t stream) {
  if (!buf || !sink4 || !ms_per_pass || bytes < 16 || repeats < 1 || (reinterpret_cast<uintptr_t>(buf) & 15)) {
    be_set_error("be_diag_stream_read: needs a 16-byte-aligned buffer of >= 16 bytes, a 4-byte sink, repeats >= 1");
    return BE_ERR_INVALID;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipEvent_t e0, e1;
  hipError_t rc = hipEventCreate(&e0);
  if (rc == hipSuccess) rc = hipEventCreate(&e1);
  const size_t n16 = (size_t)bytes >> 4;
  if (rc == hipSuccess) {
    hipLaunchKernelGGL(k_diag_stream_read, dim3(2048), dim3(256), 0, st, static_cast<const diag_u4*>(buf), n16,
                       static_cast<unsigned*>(sink4));      // warm
    rc = hipEventRecord(e0, st);
    for (int r = 0; r < repeats; ++r)
      hipLaunchKernelGGL(k_diag_stream_read, dim3(2048), dim3(256), 0, st, static_cast<const diag_u4*>(buf), n16,
                         static_cast<unsigned*>(sink4));
    if (rc == hipSuccess) rc = hipEventRecord(e1, st);
    if (rc == hipSuccess) rc = hipEventSynchronize(e1);
    float ms = 0.f;
    if (rc == hipSuccess) rc = hipEventElapsedTime(&ms, e0, e1);
    if (rc == hipSuccess) rc = hipGetLastError();
    *ms_per_pass = ms / (float)repeats;
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  if (rc != hipSuccess) {
    be_set_error(std::string("be_diag_stream_read: ") + hipGetErrorString(rc));
    return BE_ERR_HIP;
  }
  return BE_OK;
}

int be_profile_read(float* ms_host, int capacity) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  int n = g_prof_n < capacity ? g_prof_n : capacity;
  for (int i = 0; i < n; ++i) {
    hipError_t rc = hipEventSynchronize(g_prof_ev[2 * i + 1]);
    if (rc == hipSuccess) rc = hipEventElapsedTime(&ms_host[i], g_prof_ev[2 * i], g_prof_ev[2 * i + 1]);
    if (rc != hipSuccess) {
      be_set_error(std::string("be_profile_read: ") + hipGetErrorString(rc));
      return BE_ERR_HIP;
    }
  }
  g_prof_n = 0;
  return n;
}

}  // extern "C"
