// Issue cost of the vector instructions the tile kernel is made of, with 1, 2, 3 and 4 waves per
// SIMD: cycles per wave-instruction per SIMD (2.4 GHz assumed).  Each kernel runs one opcode in
// independent chains (8 registers), so the number is the pipe cost, not a dependency latency.
#include <hip/hip_runtime.h>
#include <cstdio>
#define OPS(X) \
  X(0, "v_fma_f32", "v_fma_f32 %0, %0, %1, %2") \
  X(1, "v_mul_f32", "v_mul_f32 %0, %0, %1") \
  X(2, "v_add_f32", "v_add_f32 %0, %0, %2") \
  X(3, "v_sub_f32", "v_sub_f32 %0, %0, %2") \
  X(4, "v_fmac_f32", "v_fmac_f32 %0, %1, %2") \
  X(5, "v_cndmask_b32", "v_cndmask_b32 %0, %0, %1, vcc") \
  X(6, "v_add_u32", "v_add_u32 %0, %0, %1") \
  X(7, "v_mul_lo_u32", "v_mul_lo_u32 %0, %0, %1") \
  X(8, "v_rcp_f32", "v_rcp_f32 %0, %0") \
  X(9, "v_div_fixup_f32", "v_div_fixup_f32 %0, %0, %1, %2") \
  X(10, "v_med3_f32", "v_med3_f32 %0, %0, %1, %2") \
  X(11, "v_floor_f32", "v_floor_f32 %0, %0") \
  X(12, "v_cvt_i32_f32", "v_cvt_i32_f32 %0, %0") \
  X(13, "v_cmp_lt_f32", "v_cmp_lt_f32 vcc, %0, %1") \
  X(14, "v_mov_b32", "v_mov_b32 %0, %1") \
  X(15, "v_max_i32", "v_max_i32 %0, %0, %1") \
  X(16, "v_lshlrev_b32", "v_lshlrev_b32 %0, 1, %0") \
  X(17, "v_add_f32_dpp", "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf") \
  X(18, "v_exp_f32", "v_exp_f32 %0, %0") \
  X(19, "v_min_f32", "v_min_f32 %0, %0, %1") \
  X(20, "v_max_f32", "v_max_f32 %0, %0, %1") \
  X(21, "v_and_b32", "v_and_b32 %0, %0, %1") \
  X(22, "v_or_b32", "v_or_b32 %0, %0, %1") \
  X(23, "v_mul_u32_u24", "v_mul_u32_u24 %0, %0, %1") \
  X(24, "v_mad_u32_u24", "v_mad_u32_u24 %0, %0, %1, %2") \
  X(25, "v_cvt_f32_i32", "v_cvt_f32_i32 %0, %0") \
  X(26, "v_bfe_u32", "v_bfe_u32 %0, %0, 4, 8") \
  X(27, "v_add3_u32", "v_add3_u32 %0, %0, %1, %2") \
  X(28, "v_lshl_add_u32", "v_lshl_add_u32 %0, %0, 2, %1") \
  X(29, "v_add_lshl_u32", "v_add_lshl_u32 %0, %0, %1, 2") \
  X(30, "v_fract_f32", "v_fract_f32 %0, %0") \
  X(31, "v_trunc_f32", "v_trunc_f32 %0, %0") \
  X(32, "v_cndmask_b32", "v_cndmask_b32 %0, %0, %1, s[10:11]") \
  X(33, "v_cmp_f32 sgpr", "v_cmp_lt_f32 s[12:13], %0, %1") \
  X(34, "v_xor_b32", "v_xor_b32 %0, %0, %1") \
  X(35, "v_sub_u32", "v_sub_u32 %0, %0, %1") \
  X(36, "v_mul_f32 sgpr", "v_mul_f32 %0, s20, %0") \
  X(37, "v_pk_mul_f32", "v_pk_mul_f32 %3, %3, %3") \
  X(38, "v_mul_legacy_f32", "v_mul_legacy_f32 %0, %0, %1") \
  X(39, "v_ldexp_f32", "v_ldexp_f32 %0, %0, 1") \
  X(40, "v_max3_f32", "v_max3_f32 %0, %0, %1, %2") \
  X(41, "v_mad_i32_i24", "v_mad_i32_i24 %0, %0, %1, %2") \
  X(42, "v_ashrrev_i32", "v_ashrrev_i32 %0, 1, %0") \
  X(43, "v_cvt_u32_f32", "v_cvt_u32_f32 %0, %0") \
  X(44, "v_rndne_f32", "v_rndne_f32 %0, %0") \
  X(45, "v_sub_f32 abs", "v_sub_f32 %0, |%0|, %1") \
  X(46, "v_mul_f32 inline", "v_mul_f32 %0, 0.5, %0") \
  X(47, "v_mul_f32 literal", "v_mul_f32 %0, 0x3f9e0419, %0") \
  X(48, "v_add_f32 sgpr", "v_add_f32 %0, s20, %0") \
  X(49, "v_fma_f32 sgpr", "v_fma_f32 %0, %0, s20, %2") \
  X(50, "v_fmaak_f32", "v_fmaak_f32 %0, %0, %1, 0x3f9e0419") \
  X(51, "v_fmamk_f32", "v_fmamk_f32 %0, %0, 0x3f9e0419, %1") \
  X(52, "v_add_f32 neg", "v_add_f32 %0, -%0, %1") \
  X(53, "v_mul_f32 e64", "v_mul_f32_e64 %0, %0, %1 mul:2") \
  X(54, "v_fma_f32 neg", "v_fma_f32 %0, -%0, %1, %2") \
  X(55, "v_add_f32 inline", "v_add_f32 %0, 1.0, %0") \
  X(56, "v_fma_f32 inline", "v_fma_f32 %0, %0, 2.0, %2") \
  X(57, "v_add_u32 sgpr", "v_add_u32 %0, s20, %0") \
  X(58, "v_mov_b32 sgpr", "v_mov_b32 %0, s20") \
  X(59, "v_fmac_f32 sgpr", "v_fmac_f32 %0, s20, %1")
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  float a[8];
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 p2[2] = {f2{1.0001f, 1.0002f}, f2{1.0003f, 1.0004f}};
  for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 1e-3f + i;
  const float m = 1.0001f, c = 1e-6f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
#define X(N, NAME, ASM) if (MODE == N) asm volatile(ASM : "+v"(a[i]) : "v"(m), "v"(c), "v"(p2[i & 1]) : "vcc", "s10", "s11", "s12", "s13", "s20");
        OPS(X)
#undef X
      }
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += a[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
// 64-bit add (address arithmetic)
__global__ __launch_bounds__(256) void k64(unsigned long long* out, int iters) {
  unsigned long long a[8];
  for (int i = 0; i < 8; ++i) a[i] = threadIdx.x + i;
  unsigned long long m = 12345;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(a[i]) : "v"(m));
  }
  unsigned long long s = 0;
  for (int i = 0; i < 8; ++i) s += a[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <typename F> void timeit(const char* name, int wps, F launch) {
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  launch(10);
  hipEventRecord(e0); launch(iters); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double inst = (double)iters * 64 * wps;  // wave-instructions per SIMD
  printf("%-16s waves/SIMD %d: %.2f cycles/instr/SIMD\n", name, wps, ms * 1e-3 * 2.4e9 / inst);
}
int main() {
  float* d; hipMalloc(&d, 256 * 8 * 256 * 8);
  for (int w : {4}) {
    const int blocks = 256 * w;
#define X(N, NAME, ASM) timeit(NAME, w, [&](int it) { k<N><<<blocks, 256>>>(d, it); });
    OPS(X)
#undef X
    timeit("v_lshl_add_u64", w, [&](int it) { k64<<<blocks, 256>>>((unsigned long long*)d, it); });
  }
  return 0;
}
