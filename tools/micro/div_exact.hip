// Checks that the shortened division sequence returns the bits of the IEEE quotient:
//   q = n * y; e = fma(-d, q, n); q = fma(e, y, q); e = fma(-d, q, n); q = fma(e, y, q); q = div_fixup(q, d, n)
// with y = fma(fma(-d, rcp(d), 1), rcp(d), rcp(d)), against n / d as compiled by hipcc.
// Mode 0: every float bit pattern as numerator for a set of integer-valued denominators (image sizes - 1).
// Mode 1: random numerator / denominator pairs over wide exponent ranges (the projection and the SSIM ratio).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ float refined_rcp(float d) {
  const float r = __builtin_amdgcn_rcpf(d);
  return fmaf(fmaf(-d, r, 1.0f), r, r);
}
__device__ __forceinline__ float div_by(float n, float d, float y) {
  float q = n * y;
  float e = fmaf(-d, q, n);
  q = fmaf(e, y, q);
  e = fmaf(-d, q, n);
  q = fmaf(e, y, q);
  return __builtin_amdgcn_div_fixupf(q, d, n);
}
__device__ __forceinline__ bool same(float a, float b) {
  const uint32_t x = __builtin_bit_cast(uint32_t, a), y = __builtin_bit_cast(uint32_t, b);
  return x == y || (a != a && b != b) || (a == 0.f && b == 0.f);  // any NaN = any NaN; the sign of zero is not compared
}
__global__ void all_numerators(float d, unsigned long long* bad, uint32_t* first) {
  const float y = refined_rcp(d);
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 32); i += (uint64_t)gridDim.x * blockDim.x) {
    const float n = __builtin_bit_cast(float, (uint32_t)i);
    const float want = n / d, got = div_by(n, d, y);
    if (!same(want, got)) { if (atomicAdd(bad, 1ull) == 0) *first = (uint32_t)i; }
  }
}
__device__ __forceinline__ uint32_t rng(uint64_t& s) { s = s * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(s >> 32); }
__global__ void random_pairs(int emin_n, int emax_n, int emin_d, int emax_d, int iters, unsigned long long* bad, uint32_t* first) {
  uint64_t s = 0x9E3779B97F4A7C15ull * ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x + 1);
  for (int it = 0; it < iters; ++it) {
    const uint32_t a = rng(s), b = rng(s), c = rng(s);
    const uint32_t en = 127 + emin_n + (c & 0xffff) % (uint32_t)(emax_n - emin_n + 1);
    const uint32_t ed = 127 + emin_d + (c >> 16) % (uint32_t)(emax_d - emin_d + 1);
    const float n = __builtin_bit_cast(float, (a & 0x807fffffu) | (en << 23));
    const float d = __builtin_bit_cast(float, (b & 0x807fffffu) | (ed << 23));
    const float want = n / d, got = div_by(n, d, refined_rcp(d));
    if (!same(want, got)) { if (atomicAdd(bad, 1ull) == 0) { first[0] = __builtin_bit_cast(uint32_t, n); first[1] = __builtin_bit_cast(uint32_t, d); } }
  }
}
int main() {
  unsigned long long* bad; uint32_t* first;
  if (hipMalloc(&bad, 8) != hipSuccess || hipMalloc(&first, 8) != hipSuccess) return 2;
  int rc = 0;
  const int dens[] = {1, 2, 3, 7, 15, 23, 47, 79, 95, 103, 159, 191, 319, 383, 511, 639, 1023, 1241, 1279, 2047, 4095, 16777215};
  for (int dv : dens) {
    (void)hipMemset(bad, 0, 8); (void)hipMemset(first, 0, 8);
    all_numerators<<<4096, 256>>>((float)dv, bad, first);
    unsigned long long hb = 0; uint32_t hf[2] = {0, 0};
    (void)hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost); (void)hipMemcpy(hf, first, 8, hipMemcpyDeviceToHost);
    printf("d = %8d: all 2^32 numerators, mismatches %llu (first numerator bits %08x)\n", dv, hb, hf[0]);
    if (hb) rc = 1;
  }
  struct R { int an, bn, ad, bd; const char* what; } ranges[] = {
      {-60, 60, -48, 40, "projection: c / (z + eps)"}, {-50, 3, -24, 3, "SSIM ratio n / d"}, {-100, 90, -100, 90, "wide"}};
  for (auto r : ranges) {
    (void)hipMemset(bad, 0, 8); (void)hipMemset(first, 0, 8);
    random_pairs<<<4096, 256>>>(r.an, r.bn, r.ad, r.bd, 4096, bad, first);
    unsigned long long hb = 0; uint32_t hf[2] = {0, 0};
    (void)hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost); (void)hipMemcpy(hf, first, 8, hipMemcpyDeviceToHost);
    printf("%-28s 2^32 random pairs, mismatches %llu (first n %08x d %08x)\n", r.what, hb, hf[0], hf[1]);
    if (hb) rc = 1;
  }
  return rc;
}
