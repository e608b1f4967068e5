// Developer microbenchmark: (1) how many HIP streams run kernels at the same time, and
// (2) whether a stream that waits for an event holds up kernels of OTHER streams.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void spin(long long ticks, int* out) {   // wall clock: 100 MHz
  long long t0 = wall_clock64();
  long long n = 0;
  while (wall_clock64() - t0 < ticks && n < (1ll << 40)) ++n;
  if (n == -1) out[0] = 1;
}
static double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
int main(int argc, char** argv) {
  setenv("GPU_MAX_HW_QUEUES", argc > 1 ? argv[1] : "16", 1);
  int* out; (void)hipMalloc(&out, 4);
  for (int ns : {1, 2, 4, 6, 8, 12, 16}) {
    std::vector<hipStream_t> st(ns);
    for (auto& s : st) (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    for (auto& s : st) hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, 1000, out);
    (void)hipDeviceSynchronize();
    double t0 = now_ms();
    for (auto& s : st) hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, 100000 /* 1 ms */, out);
    (void)hipDeviceSynchronize();
    printf("queues %s, %2d streams, one 1 ms kernel each: %.3f ms\n", getenv("GPU_MAX_HW_QUEUES"), ns, now_ms() - t0);
    for (auto& s : st) (void)hipStreamDestroy(s);
  }
  // (2) nw streams wait for an event that fires after 3 ms; nb other streams run a 0.3 ms kernel
  for (int nw : {0, 1, 2, 4, 8}) {
    const int nb = 4;
    hipStream_t src; (void)hipStreamCreateWithFlags(&src, hipStreamNonBlocking);
    std::vector<hipStream_t> w(nw), b(nb);
    for (auto& s : w) (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    for (auto& s : b) (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    hipEvent_t ev; (void)hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    std::vector<hipEvent_t> done(nb);
    for (auto& e : done) (void)hipEventCreateWithFlags(&e, hipEventDisableTiming);
    hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, src, 1000, out);
    for (auto& s : w) hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, 1000, out);
    for (auto& s : b) hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, 1000, out);
    (void)hipDeviceSynchronize();
    double t0 = now_ms();
    hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, src, 300000 /* 3 ms */, out);
    (void)hipEventRecord(ev, src);
    for (auto& s : w) { (void)hipStreamWaitEvent(s, ev, 0); hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, 1000, out); }
    for (int i = 0; i < nb; ++i) { hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, b[i], 30000 /* 0.3 ms */, out); (void)hipEventRecord(done[i], b[i]); }
    double worst = 0;
    for (int i = 0; i < nb; ++i) { (void)hipEventSynchronize(done[i]); double t = now_ms() - t0; if (t > worst) worst = t; }
    (void)hipDeviceSynchronize();
    printf("queues %s, %d streams waiting 3 ms for an event: the 0.3 ms kernels of %d other streams finish after %.3f ms (all done %.3f ms)\n",
           getenv("GPU_MAX_HW_QUEUES"), nw, nb, worst, now_ms() - t0);
    for (auto& s : w) (void)hipStreamDestroy(s);
    for (auto& s : b) (void)hipStreamDestroy(s);
    (void)hipStreamDestroy(src);
  }
  return 0;
}
