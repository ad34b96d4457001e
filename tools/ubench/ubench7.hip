// Microbenchmark 7 (round 6): what does a cross-stream dependency cost per step on this runtime, and which form is cheapest?
// Question: the pipelined rank step — all-gather of step t+1 on a side stream while the main stream scatters
// step t — costs +8..17 us per step over the sum of its kernels.  The step is modelled with timed spin kernels: main stream
// K1 (27 us, 256 x 1024 threads) + K2 (5 us), side stream KS (4 us, small).  Variants of the dependency:
//   seq        : K1, K2, KS on one stream (no dependency packets)                      -> the sequential schedule
//   ideal      : K1, K2 on main; KS on side, NO dependencies at all (incorrect; floor)
//   ev         : per step  record(e_in, main); wait(side, e_in); KS; record(e_done, side); wait(main, e_done); K1; K2   (be_exchange_post/_wait)
//   ev_nofence : the same with events created hipEventDisableTiming | hipEventDisableSystemFence
//   ev_ext     : record(e_in) replaced by hipExtLaunchKernelGGL's stopEvent on K2 (no separate marker packet on main)
//   one_wait   : only the side -> main half (record on side, wait on main); no main -> side dependency (incorrect; attributes the halves)
//   one_record : only the main -> side half
//   graph_seq  : seq captured as one graph, replayed
//   graph_fork : {fork: KS on side || K1, K2 on main; join} captured as ONE graph, replayed (no cross-launch events)
//   graph_fork4: four such steps per graph
// Output: us per step of each over N steps.   Build: hipcc --offload-arch=gfx950 -O3 ubench7.hip -o ubench7
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <chrono>
#include <functional>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

// spins until `ticks` of the 100 MHz constant clock have passed (bounded: an exit every wave reaches)
__global__ void __launch_bounds__(1024) k_spin(uint32_t ticks, uint32_t* sink) {
  const uint64_t t0 = wall_clock64();
  uint32_t it = 0;
  while (wall_clock64() - t0 < ticks && it < (1u << 22)) { __builtin_amdgcn_s_sleep(8); ++it; }
  if (it == 0xffffffffu) sink[0] = it;
}

static double time_us(int n, const std::function<void(int)>& step, hipStream_t a, hipStream_t b) {
  for (int i = 0; i < 50; ++i) step(i);
  CK(hipStreamSynchronize(a)); CK(hipStreamSynchronize(b));
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < n; ++i) step(i);
  CK(hipStreamSynchronize(a)); CK(hipStreamSynchronize(b));
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / n;
}

int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 2000;
  const uint32_t T1 = argc > 2 ? atoi(argv[2]) : 2700, T2 = 500, TS = argc > 3 ? atoi(argv[3]) : 400;   // 100 MHz ticks
  uint32_t* sink; CK(hipMalloc(&sink, 64));
  hipStream_t A, B;
  CK(hipStreamCreateWithFlags(&A, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&B, hipStreamNonBlocking));
  auto K1 = [&](hipStream_t s) { hipLaunchKernelGGL(k_spin, dim3(256), dim3(1024), 0, s, T1, sink); };
  auto K2 = [&](hipStream_t s) { hipLaunchKernelGGL(k_spin, dim3(256), dim3(256), 0, s, T2, sink); };
  auto KS = [&](hipStream_t s) { hipLaunchKernelGGL(k_spin, dim3(8), dim3(256), 0, s, TS, sink); };
  printf("model step: K1 %.1f us (256 x 1024) + K2 %.1f us on main, KS %.1f us on side; %d steps each\n", T1 / 100.0, T2 / 100.0, TS / 100.0, N);

  printf("%-12s %8.2f us/step\n", "seq", time_us(N, [&](int) { KS(A); K1(A); K2(A); }, A, B));
  printf("%-12s %8.2f us/step\n", "ideal", time_us(N, [&](int) { KS(B); K1(A); K2(A); }, A, B));

  for (int fl = 0; fl < 3; ++fl) {
    const unsigned flags = hipEventDisableTiming | (fl == 1 ? hipEventDisableSystemFence : 0) | (fl == 2 ? hipEventReleaseToDevice : 0);
    if (fl) printf("-- events with %s\n", fl == 1 ? "hipEventDisableSystemFence" : "hipEventReleaseToDevice");
    hipEvent_t e_in, e_done[2];
    CK(hipEventCreateWithFlags(&e_in, flags)); CK(hipEventCreateWithFlags(&e_done[0], flags)); CK(hipEventCreateWithFlags(&e_done[1], flags));
    // prime: slot 0 posted
    auto post = [&](int slot) { CK(hipEventRecord(e_in, A)); CK(hipStreamWaitEvent(B, e_in, 0)); KS(B); CK(hipEventRecord(e_done[slot], B)); };
    post(0);
    double us = time_us(N, [&](int i) { post((i + 1) & 1); CK(hipStreamWaitEvent(A, e_done[i & 1], 0)); K1(A); K2(A); }, A, B);
    printf("%-12s %8.2f us/step\n", fl ? "ev (flag)" : "ev", us);
    // the record placed between K1 and K2 instead of behind K2
    us = time_us(N, [&](int i) {
      CK(hipStreamWaitEvent(B, e_in, 0)); KS(B); CK(hipEventRecord(e_done[(i + 1) & 1], B));
      CK(hipStreamWaitEvent(A, e_done[i & 1], 0)); K1(A); CK(hipEventRecord(e_in, A)); K2(A); }, A, B);
    printf("%-12s %8.2f us/step   (record between K1 and K2)\n", "ev_mid", us);
    // halves
    CK(hipEventRecord(e_done[0], B)); CK(hipEventRecord(e_done[1], B));
    us = time_us(N, [&](int i) { KS(B); CK(hipEventRecord(e_done[(i + 1) & 1], B)); CK(hipStreamWaitEvent(A, e_done[i & 1], 0)); K1(A); K2(A); }, A, B);
    printf("%-12s %8.2f us/step   (side -> main only%s)\n", "one_wait", us, fl ? ", nofence" : "");
    us = time_us(N, [&](int i) { CK(hipEventRecord(e_in, A)); CK(hipStreamWaitEvent(B, e_in, 0)); KS(B); K1(A); K2(A); }, A, B);
    printf("%-12s %8.2f us/step   (main -> side only%s)\n", "one_record", us, fl ? ", nofence" : "");
    us = time_us(N, [&](int i) { CK(hipEventRecord(e_in, A)); K1(A); K2(A); }, A, B);
    printf("%-12s %8.2f us/step   (a record on main nobody waits for%s)\n", "rec_only", us, fl ? ", nofence" : "");
    // stop event on K2 instead of a separate record
    auto post_ext = [&](int slot) { CK(hipStreamWaitEvent(B, e_in, 0)); KS(B); CK(hipEventRecord(e_done[slot], B)); };
    hipExtLaunchKernelGGL(k_spin, dim3(256), dim3(256), 0, A, nullptr, e_in, 0, T2, sink);
    post_ext(0);
    us = time_us(N, [&](int i) {
      post_ext((i + 1) & 1); CK(hipStreamWaitEvent(A, e_done[i & 1], 0)); K1(A);
      hipExtLaunchKernelGGL(k_spin, dim3(256), dim3(256), 0, A, nullptr, e_in, 0, T2, sink);
    }, A, B);
    printf("%-12s %8.2f us/step   (stop event on K2%s)\n", "ev_ext", us, fl ? ", nofence" : "");
    CK(hipStreamSynchronize(A)); CK(hipStreamSynchronize(B));
    CK(hipEventDestroy(e_in)); CK(hipEventDestroy(e_done[0])); CK(hipEventDestroy(e_done[1]));
  }

  // stream memory operations instead of events: main writes a step number when it gets there, side waits for it (and back)
  {
    int can = 0;
    CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
    printf("-- stream memory operations (hipStreamWriteValue32 / hipStreamWaitValue32): attribute %d\n", can);
    if (can) {
      uint32_t* flag_host = nullptr;            // signal memory must be host-coherent: hipExtMallocWithFlags(hipMallocSignalMemory) or pinned host
      CK(hipExtMallocWithFlags(reinterpret_cast<void**>(&flag_host), 8, hipMallocSignalMemory));
      uint32_t* f_in = flag_host;
      uint32_t* f_done = nullptr;
      CK(hipExtMallocWithFlags(reinterpret_cast<void**>(&f_done), 8, hipMallocSignalMemory));
      *f_in = 0; *f_done = 0;
      uint32_t step_no = 0;
      auto stepf = [&](int) {
        ++step_no;
        CK(hipStreamWriteValue32(A, f_in, step_no, 0));
        CK(hipStreamWaitValue32(B, f_in, step_no, hipStreamWaitValueGte, 0xffffffffu));
        KS(B);
        CK(hipStreamWriteValue32(B, f_done, step_no, 0));
        if (step_no > 1) CK(hipStreamWaitValue32(A, f_done, step_no - 1, hipStreamWaitValueGte, 0xffffffffu));
        K1(A); K2(A);
      };
      printf("%-12s %8.2f us/step\n", "memops", time_us(N, stepf, A, B));
      auto half = [&](int) {
        ++step_no;
        CK(hipStreamWriteValue32(A, f_in, step_no, 0));
        CK(hipStreamWaitValue32(B, f_in, step_no, hipStreamWaitValueGte, 0xffffffffu));
        KS(B); K1(A); K2(A);
      };
      printf("%-12s %8.2f us/step   (main -> side only)\n", "memops_rec", time_us(N, half, A, B));
    }
  }

  // graphs
  auto capture = [&](int unroll, bool fork) {
    hipGraph_t g; hipGraphExec_t ge;
    hipEvent_t ef, ej; CK(hipEventCreateWithFlags(&ef, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ej, hipEventDisableTiming));
    CK(hipStreamBeginCapture(A, hipStreamCaptureModeGlobal));
    for (int u = 0; u < unroll; ++u) {
      if (fork) {
        CK(hipEventRecord(ef, A)); CK(hipStreamWaitEvent(B, ef, 0)); KS(B); CK(hipEventRecord(ej, B));
        K1(A); K2(A);
        CK(hipStreamWaitEvent(A, ej, 0));
      } else { KS(A); K1(A); K2(A); }
    }
    CK(hipStreamEndCapture(A, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    return ge;
  };
  for (int unroll : {1, 4}) {
    hipGraphExec_t gs = capture(unroll, false), gf = capture(unroll, true);
    char nm[32];
    snprintf(nm, sizeof nm, "graph_seq%d", unroll);
    printf("%-12s %8.2f us/step\n", nm, time_us(N / unroll, [&](int) { CK(hipGraphLaunch(gs, A)); }, A, B) / unroll);
    snprintf(nm, sizeof nm, "graph_fork%d", unroll);
    printf("%-12s %8.2f us/step\n", nm, time_us(N / unroll, [&](int) { CK(hipGraphLaunch(gf, A)); }, A, B) / unroll);
  }
  return 0;
}
