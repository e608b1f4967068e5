// Developer microbenchmark: VALU issue rate of long straight-line code made of 4-byte
// (VOP2, register operands) versus 8-byte (VOP2 + 32-bit literal) instructions.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
#define REP512(x) REP8(REP64(x))
#define REP4096(x) REP8(REP512(x))
#define BODY4 \
  "v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n" \
  "v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %8, %9\n v_fmac_f32 %6, %8, %9\n v_fmac_f32 %7, %8, %9\n"
#define BODY8 \
  "v_fmaak_f32 %0, %8, %0, 0x3f800001\n v_fmaak_f32 %1, %8, %1, 0x3f800001\n v_fmaak_f32 %2, %8, %2, 0x3f800001\n v_fmaak_f32 %3, %8, %3, 0x3f800001\n" \
  "v_fmaak_f32 %4, %8, %4, 0x3f800001\n v_fmaak_f32 %5, %8, %5, 0x3f800001\n v_fmaak_f32 %6, %8, %6, 0x3f800001\n v_fmaak_f32 %7, %8, %7, 0x3f800001\n"
#define BODYS \
  "v_fmac_f32 %0, %10, %9\n v_fmac_f32 %1, %10, %9\n v_fmac_f32 %2, %10, %9\n v_fmac_f32 %3, %10, %9\n" \
  "v_fmac_f32 %4, %10, %9\n v_fmac_f32 %5, %10, %9\n v_fmac_f32 %6, %10, %9\n v_fmac_f32 %7, %10, %9\n"
#define BODYV3 \
  "v_fma_f32 %0, %8, %9, %0\n v_fma_f32 %1, %8, %9, %1\n v_fma_f32 %2, %8, %9, %2\n v_fma_f32 %3, %8, %9, %3\n" \
  "v_fma_f32 %4, %8, %9, %4\n v_fma_f32 %5, %8, %9, %5\n v_fma_f32 %6, %8, %9, %6\n v_fma_f32 %7, %8, %9, %7\n"
#define BODYMIX \
  "v_fmac_f32 %0, %8, %9\n v_fmaak_f32 %1, %8, %1, 0x3f800001\n v_fmac_f32 %2, %8, %9\n v_fmaak_f32 %3, %8, %3, 0x3f800001\n" \
  "v_fmac_f32 %4, %8, %9\n v_fmaak_f32 %5, %8, %5, 0x3f800001\n v_fmac_f32 %6, %8, %9\n v_fmaak_f32 %7, %8, %7, 0x3f800001\n"
#define BODYSMUL \
  "v_mul_f32 %0, %10, %0\n v_mul_f32 %1, %10, %1\n v_mul_f32 %2, %10, %2\n v_mul_f32 %3, %10, %3\n" \
  "v_mul_f32 %4, %10, %4\n v_mul_f32 %5, %10, %5\n v_mul_f32 %6, %10, %6\n v_mul_f32 %7, %10, %7\n"
#define BODYVMUL \
  "v_mul_f32 %0, %8, %0\n v_mul_f32 %1, %8, %1\n v_mul_f32 %2, %8, %2\n v_mul_f32 %3, %8, %3\n" \
  "v_mul_f32 %4, %8, %4\n v_mul_f32 %5, %8, %5\n v_mul_f32 %6, %8, %6\n v_mul_f32 %7, %8, %7\n"
#define BODYINL \
  "v_fmac_f32 %0, 0.5, %9\n v_fmac_f32 %1, 0.5, %9\n v_fmac_f32 %2, 0.5, %9\n v_fmac_f32 %3, 0.5, %9\n" \
  "v_fmac_f32 %4, 0.5, %9\n v_fmac_f32 %5, 0.5, %9\n v_fmac_f32 %6, 0.5, %9\n v_fmac_f32 %7, 0.5, %9\n"
#define BODYSMIX \
  "v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %10, %9\n" \
  "v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %8, %9\n v_fmac_f32 %6, %8, %9\n v_fmac_f32 %7, %10, %9\n"
#define BODYLITMUL \
  "v_mul_f32 %0, 0x3f800001, %0\n v_mul_f32 %1, 0x3f800001, %1\n v_mul_f32 %2, 0x3f800001, %2\n v_mul_f32 %3, 0x3f800001, %3\n" \
  "v_mul_f32 %4, 0x3f800001, %4\n v_mul_f32 %5, 0x3f800001, %5\n v_mul_f32 %6, 0x3f800001, %6\n v_mul_f32 %7, 0x3f800001, %7\n"
#define KERNEL(NAME, ASM) \
__global__ void NAME(float* out, int iters, float a, float b, float sc) { \
  float r0 = threadIdx.x, r1 = 1, r2 = 2, r3 = 3, r4 = 4, r5 = 5, r6 = 6, r7 = 7; \
  for (int i = 0; i < iters; ++i) { \
    asm volatile(ASM : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b), "s"(sc)); \
  } \
  if (r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 == 12345.678f) out[0] = r0; \
}
KERNEL(k4_small, REP8(BODY4))            // 64 instructions, 256 B per loop body
KERNEL(k8_small, REP8(BODY8))            // 64 instructions, 512 B
KERNEL(k4_big, REP512(BODY4))            // 4096 instructions, 16 KB
KERNEL(k8_big, REP512(BODY8))            // 4096 instructions, 32 KB
KERNEL(ks_big, REP512(BODYS))            // 4096 instructions with an SGPR operand, 16 KB
KERNEL(k8_huge, REP512(BODY8) REP512(BODY8))  // 8192 instructions, 64 KB

KERNEL(kv3_big, REP512(BODYV3))
KERNEL(kmix_big, REP512(BODYMIX))
KERNEL(ksmul_big, REP512(BODYSMUL))
KERNEL(kvmul_big, REP512(BODYVMUL))
KERNEL(kinl_big, REP512(BODYINL))
KERNEL(ksmix_big, REP512(BODYSMIX))
KERNEL(klitmul_big, REP512(BODYLITMUL))
template <class K>
void run(const char* name, K k, int body, int waves_per_simd) {
  float* out; hipMalloc(&out, 4);
  const long total = 1 << 19;  // instructions per wave
  int iters = (int)(total / body);
  dim3 grid(256 * 4 * waves_per_simd), block(64);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k, grid, block, 0, 0, out, iters / 8, 1.0f, 1e-9f, 1.0f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k, grid, block, 0, 0, out, iters, 1.0f, 1e-9f, 1.0f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double inst = (double)total * grid.x;
  printf("%-9s waves/simd %d: %.3f ms, %.1f G wave-instr/s, %.3f instr/ns/SIMD\n", name, waves_per_simd, ms, inst / ms / 1e6, inst / ms / 1e6 / 1024);
  hipFree(out);
}
int main() {
  for (int w : {4, 7}) {
    run("k4_big", k4_big, 4096, w); run("k8_big", k8_big, 4096, w); run("ks_big", ks_big, 4096, w);
    run("kv3_big", kv3_big, 4096, w); run("kmix_big", kmix_big, 4096, w);
    run("ksmul", ksmul_big, 4096, w); run("kvmul", kvmul_big, 4096, w); run("kinl", kinl_big, 4096, w);
    run("ksmix1/4", ksmix_big, 4096, w); run("klitmul", klitmul_big, 4096, w);
  }
  return 0;
}
