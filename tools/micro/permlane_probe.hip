// Probe of the 4-values-into-one-register wave reduction built from
// v_permlane32_swap / v_permlane16_swap and DPP row adds: prints which row of
// the result holds the total of which input.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void swap32(float& a, float& b) {
  const f2 r = __builtin_bit_cast(f2, __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false));
  a = r.x; b = r.y;
}
__device__ __forceinline__ void swap16(float& a, float& b) {
  const f2 r = __builtin_bit_cast(f2, __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false));
  a = r.x; b = r.y;
}
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  const int m = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false);
  return v + __builtin_bit_cast(float, m);
}
__global__ void k(float* y) {
  const int l = threadIdx.x;
  // value i = 10^i per lane: totals 64, 640, 6400, 64000
  float p = 1.f, q = 10.f, r = 100.f, s = 1000.f;
  swap32(p, q); float u = p + q;
  swap32(r, s); float v = r + s;
  swap16(u, v); float t = u + v;
  t = dpp_add<0xb1>(t); t = dpp_add<0x4e>(t); t = dpp_add<0x124>(t); t = dpp_add<0x128>(t);
  y[l] = t;
}
int main() {
  float* d; hipMalloc(&d, 256); k<<<1, 64>>>(d); float h[64]; hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
  for (int r = 0; r < 4; ++r) printf("row %d: lane0 %.0f lane15 %.0f\n", r, h[16 * r], h[16 * r + 15]);
  return 0;
}
