// Developer microbenchmark: how many HIP streams run kernels at the same time.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <chrono>
__global__ void spin(long long cycles, int* out) {
  long long t0 = wall_clock64();
  long long n = 0;
  while (wall_clock64() - t0 < cycles && n < (1ll << 40)) ++n;   // wall clock: 100 MHz
  if (n == -1) out[0] = 1;
}
int main(int argc, char** argv) {
  setenv("GPU_MAX_HW_QUEUES", argc > 1 ? argv[1] : "16", 1);
  int* out; (void)hipMalloc(&out, 4);
  for (int ns : {1, 2, 4, 5, 6, 8, 10, 12, 16}) {
    std::vector<hipStream_t> st(ns);
    for (auto& s : st) (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    for (auto& s : st) hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, 1000, out);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, 0);
    (void)hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now();
    for (auto& s : st) hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, 100000 /* 1 ms */, out);
    (void)hipDeviceSynchronize();
    auto t1 = std::chrono::steady_clock::now();
    printf("queues %s streams %2d: %.3f ms for one 1 ms kernel per stream\n", getenv("GPU_MAX_HW_QUEUES"), ns,
           std::chrono::duration<double, std::milli>(t1 - t0).count());
    for (auto& s : st) (void)hipStreamDestroy(s);
  }
  return 0;
}
