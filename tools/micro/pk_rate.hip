// Issue rate of v_fma_f32 / v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 with 1, 2, 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  float a[8]; f2 b[8];
  for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 1e-3f + i; b[i] = f2{a[i], a[i] + 1.f}; }
  const float m = 1.0001f, c = 1e-6f; const f2 m2 = {m, m}, c2 = {c, c};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
        if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(b[i]) : "v"(m2), "v"(c2));
        if (MODE == 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(b[i]) : "v"(m2));
        if (MODE == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(b[i]) : "v"(c2));
        if (MODE == 4) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
      }
  }
  float s = 0; for (int i = 0; i < 8; ++i) s += a[i] + b[i].x + b[i].y;
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE> void run(const char* name, float* d, int wps) {
  const int iters = 2000, blocks = 256 * wps;  // 4 waves per block -> wps blocks per CU
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<blocks, 256>>>(d, 10);
  hipEventRecord(e0); k<MODE><<<blocks, 256>>>(d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double inst = (double)iters * 64 * wps;  // wave-instructions per SIMD
  printf("%-14s waves/SIMD %d: %.2f ms, %.2f cycles/instr/SIMD @2.4GHz\n", name, wps, ms, ms * 1e-3 * 2.4e9 / inst);
}
int main() {
  float* d; hipMalloc(&d, 256 * 8 * 256 * 4);
  for (int w : {1, 2, 4}) { run<0>("v_fma_f32", d, w); run<4>("v_mul_f32", d, w); run<1>("v_pk_fma_f32", d, w); run<2>("v_pk_mul_f32", d, w); run<3>("v_pk_add_f32", d, w); }
  return 0;
}
