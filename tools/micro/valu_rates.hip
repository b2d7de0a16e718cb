// Vector-pipe cost of the opcodes (and operand kinds) the tile kernel is made of, with 1, 2, 3 and 4
// waves per SIMD: cycles per wave-instruction per SIMD (2.4 GHz assumed).  Every kernel runs one
// opcode in eight independent register chains; a loop body is a single asm block of 64
// instructions (separate asm statements make the compiler put an s_nop after each one, which
// halves the issue rate of a lone wave).  With four waves per SIMD the figure is the time the
// opcode holds the pipe: about 2.3 cycles for plain float32 / 32-bit add / logic opcodes with
// vector-register, inline-constant or literal operands, about 4.3 for everything else -- including
// the same opcodes with a scalar-register operand -- and about 8.4 for transcendentals.
#include <hip/hip_runtime.h>
#include <cstdio>
// One loop body: the instruction of the mode in eight independent register chains (operand %n,
// n = 0 .. 7), eight times over, as ONE asm statement (see above).
#define CHAINS(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)
#define BODY(OP) CHAINS(OP) CHAINS(OP) CHAINS(OP) CHAINS(OP) CHAINS(OP) CHAINS(OP) CHAINS(OP) CHAINS(OP)
#define RUN(OP)                                                                                                   \
  asm volatile(BODY(OP)                                                                                           \
               : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])   \
               : "v"(m), "v"(c)                                                                                   \
               : "vcc", "s10", "s11", "s20")
// the instruction of every mode; %8 = m and %9 = c are vector registers, s20 / s[10:11] scalar ones
#define OP0(n) "v_fma_f32 %" #n ", %" #n ", %8, %9\n\t"
#define OP1(n) "v_mul_f32 %" #n ", %" #n ", %8\n\t"
#define OP2(n) "v_add_f32 %" #n ", %" #n ", %9\n\t"
#define OP3(n) "v_sub_f32 %" #n ", %" #n ", %9\n\t"
#define OP4(n) "v_fmac_f32 %" #n ", %8, %9\n\t"
#define OP5(n) "v_mov_b32 %" #n ", %8\n\t"
#define OP6(n) "v_add_u32 %" #n ", %" #n ", %8\n\t"
#define OP7(n) "v_sub_u32 %" #n ", %" #n ", %8\n\t"
#define OP8(n) "v_and_b32 %" #n ", %" #n ", %8\n\t"
#define OP9(n) "v_or_b32 %" #n ", %" #n ", %8\n\t"
#define OP10(n) "v_xor_b32 %" #n ", %" #n ", %8\n\t"
#define OP11(n) "v_ashrrev_i32 %" #n ", 1, %" #n "\n\t"
#define OP12(n) "v_mul_f32 %" #n ", 0.5, %" #n "\n\t"
#define OP13(n) "v_mul_f32 %" #n ", 0x3f9e0419, %" #n "\n\t"
#define OP14(n) "v_fmaak_f32 %" #n ", %" #n ", %8, 0x3f9e0419\n\t"
#define OP15(n) "v_add_f32 %" #n ", -%" #n ", %8\n\t"
#define OP16(n) "v_sub_f32 %" #n ", |%" #n "|, %8\n\t"
#define OP17(n) "v_mul_f32 %" #n ", s20, %" #n "\n\t"
#define OP18(n) "v_add_f32 %" #n ", s20, %" #n "\n\t"
#define OP19(n) "v_fma_f32 %" #n ", %" #n ", s20, %9\n\t"
#define OP20(n) "v_add_u32 %" #n ", s20, %" #n "\n\t"
#define OP21(n) "v_mov_b32 %" #n ", s20\n\t"
#define OP22(n) "v_cndmask_b32 %" #n ", %" #n ", %8, s[10:11]\n\t"
#define OP23(n) "v_cmp_lt_f32 vcc, %" #n ", %8\n\t"
#define OP24(n) "v_min_f32 %" #n ", %" #n ", %8\n\t"
#define OP25(n) "v_max_f32 %" #n ", %" #n ", %8\n\t"
#define OP26(n) "v_med3_f32 %" #n ", %" #n ", %8, %9\n\t"
#define OP27(n) "v_floor_f32 %" #n ", %" #n "\n\t"
#define OP28(n) "v_cvt_i32_f32 %" #n ", %" #n "\n\t"
#define OP29(n) "v_cvt_f32_i32 %" #n ", %" #n "\n\t"
#define OP30(n) "v_min_i32 %" #n ", %" #n ", %8\n\t"
#define OP31(n) "v_lshlrev_b32 %" #n ", 1, %" #n "\n\t"
#define OP32(n) "v_lshl_add_u32 %" #n ", %" #n ", 2, %8\n\t"
#define OP33(n) "v_add3_u32 %" #n ", %" #n ", %8, %9\n\t"
#define OP34(n) "v_bfe_u32 %" #n ", %" #n ", 4, 8\n\t"
#define OP35(n) "v_mul_u32_u24 %" #n ", %" #n ", %8\n\t"
#define OP36(n) "v_mad_u32_u24 %" #n ", %" #n ", %8, %9\n\t"
#define OP37(n) "v_mul_lo_u32 %" #n ", %" #n ", %8\n\t"
#define OP38(n) "v_div_fixup_f32 %" #n ", %" #n ", %8, %9\n\t"
#define OP39(n) "v_ldexp_f32 %" #n ", %" #n ", 1\n\t"
#define OP40(n) "v_add_f32_dpp %" #n ", %" #n ", %" #n " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
#define OP41(n) "v_rcp_f32 %" #n ", %" #n "\n\t"
#define OP42(n) "v_exp_f32 %" #n ", %" #n "\n\t"
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  float a[8];
  for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 1e-3f + i;
  const float m = 1.0001f, c = 1e-6f;
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0)
      RUN(OP0);
    else if (MODE == 1)
      RUN(OP1);
    else if (MODE == 2)
      RUN(OP2);
    else if (MODE == 3)
      RUN(OP3);
    else if (MODE == 4)
      RUN(OP4);
    else if (MODE == 5)
      RUN(OP5);
    else if (MODE == 6)
      RUN(OP6);
    else if (MODE == 7)
      RUN(OP7);
    else if (MODE == 8)
      RUN(OP8);
    else if (MODE == 9)
      RUN(OP9);
    else if (MODE == 10)
      RUN(OP10);
    else if (MODE == 11)
      RUN(OP11);
    else if (MODE == 12)
      RUN(OP12);
    else if (MODE == 13)
      RUN(OP13);
    else if (MODE == 14)
      RUN(OP14);
    else if (MODE == 15)
      RUN(OP15);
    else if (MODE == 16)
      RUN(OP16);
    else if (MODE == 17)
      RUN(OP17);
    else if (MODE == 18)
      RUN(OP18);
    else if (MODE == 19)
      RUN(OP19);
    else if (MODE == 20)
      RUN(OP20);
    else if (MODE == 21)
      RUN(OP21);
    else if (MODE == 22)
      RUN(OP22);
    else if (MODE == 23)
      RUN(OP23);
    else if (MODE == 24)
      RUN(OP24);
    else if (MODE == 25)
      RUN(OP25);
    else if (MODE == 26)
      RUN(OP26);
    else if (MODE == 27)
      RUN(OP27);
    else if (MODE == 28)
      RUN(OP28);
    else if (MODE == 29)
      RUN(OP29);
    else if (MODE == 30)
      RUN(OP30);
    else if (MODE == 31)
      RUN(OP31);
    else if (MODE == 32)
      RUN(OP32);
    else if (MODE == 33)
      RUN(OP33);
    else if (MODE == 34)
      RUN(OP34);
    else if (MODE == 35)
      RUN(OP35);
    else if (MODE == 36)
      RUN(OP36);
    else if (MODE == 37)
      RUN(OP37);
    else if (MODE == 38)
      RUN(OP38);
    else if (MODE == 39)
      RUN(OP39);
    else if (MODE == 40)
      RUN(OP40);
    else if (MODE == 41)
      RUN(OP41);
    else if (MODE == 42)
      RUN(OP42);
  }
  float s = 0;
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
  printf("%-20s waves/SIMD %d: %.2f cycles/instr/SIMD\n", name, wps, ms * 1e-3 * 2.4e9 / inst);
}
int main() {
  float* d; hipMalloc(&d, 256 * 8 * 256 * 8);
  for (int w : {1, 2, 3, 4}) {
    const int blocks = 256 * w;  // 4 waves per block: w blocks per CU = w waves per SIMD
    timeit("v_fma_f32", w, [&](int it) { k<0><<<blocks, 256>>>(d, it); });
    timeit("v_mul_f32", w, [&](int it) { k<1><<<blocks, 256>>>(d, it); });
    timeit("v_add_f32", w, [&](int it) { k<2><<<blocks, 256>>>(d, it); });
    timeit("v_sub_f32", w, [&](int it) { k<3><<<blocks, 256>>>(d, it); });
    timeit("v_fmac_f32", w, [&](int it) { k<4><<<blocks, 256>>>(d, it); });
    timeit("v_mov_b32", w, [&](int it) { k<5><<<blocks, 256>>>(d, it); });
    timeit("v_add_u32", w, [&](int it) { k<6><<<blocks, 256>>>(d, it); });
    timeit("v_sub_u32", w, [&](int it) { k<7><<<blocks, 256>>>(d, it); });
    timeit("v_and_b32", w, [&](int it) { k<8><<<blocks, 256>>>(d, it); });
    timeit("v_or_b32", w, [&](int it) { k<9><<<blocks, 256>>>(d, it); });
    timeit("v_xor_b32", w, [&](int it) { k<10><<<blocks, 256>>>(d, it); });
    timeit("v_ashrrev_i32", w, [&](int it) { k<11><<<blocks, 256>>>(d, it); });
    timeit("v_mul_f32 inline", w, [&](int it) { k<12><<<blocks, 256>>>(d, it); });
    timeit("v_mul_f32 literal", w, [&](int it) { k<13><<<blocks, 256>>>(d, it); });
    timeit("v_fmaak_f32", w, [&](int it) { k<14><<<blocks, 256>>>(d, it); });
    timeit("v_add_f32 neg", w, [&](int it) { k<15><<<blocks, 256>>>(d, it); });
    timeit("v_sub_f32 abs", w, [&](int it) { k<16><<<blocks, 256>>>(d, it); });
    timeit("v_mul_f32 sgpr", w, [&](int it) { k<17><<<blocks, 256>>>(d, it); });
    timeit("v_add_f32 sgpr", w, [&](int it) { k<18><<<blocks, 256>>>(d, it); });
    timeit("v_fma_f32 sgpr", w, [&](int it) { k<19><<<blocks, 256>>>(d, it); });
    timeit("v_add_u32 sgpr", w, [&](int it) { k<20><<<blocks, 256>>>(d, it); });
    timeit("v_mov_b32 sgpr", w, [&](int it) { k<21><<<blocks, 256>>>(d, it); });
    timeit("v_cndmask_b32", w, [&](int it) { k<22><<<blocks, 256>>>(d, it); });
    timeit("v_cmp_lt_f32", w, [&](int it) { k<23><<<blocks, 256>>>(d, it); });
    timeit("v_min_f32", w, [&](int it) { k<24><<<blocks, 256>>>(d, it); });
    timeit("v_max_f32", w, [&](int it) { k<25><<<blocks, 256>>>(d, it); });
    timeit("v_med3_f32", w, [&](int it) { k<26><<<blocks, 256>>>(d, it); });
    timeit("v_floor_f32", w, [&](int it) { k<27><<<blocks, 256>>>(d, it); });
    timeit("v_cvt_i32_f32", w, [&](int it) { k<28><<<blocks, 256>>>(d, it); });
    timeit("v_cvt_f32_i32", w, [&](int it) { k<29><<<blocks, 256>>>(d, it); });
    timeit("v_min_i32", w, [&](int it) { k<30><<<blocks, 256>>>(d, it); });
    timeit("v_lshlrev_b32", w, [&](int it) { k<31><<<blocks, 256>>>(d, it); });
    timeit("v_lshl_add_u32", w, [&](int it) { k<32><<<blocks, 256>>>(d, it); });
    timeit("v_add3_u32", w, [&](int it) { k<33><<<blocks, 256>>>(d, it); });
    timeit("v_bfe_u32", w, [&](int it) { k<34><<<blocks, 256>>>(d, it); });
    timeit("v_mul_u32_u24", w, [&](int it) { k<35><<<blocks, 256>>>(d, it); });
    timeit("v_mad_u32_u24", w, [&](int it) { k<36><<<blocks, 256>>>(d, it); });
    timeit("v_mul_lo_u32", w, [&](int it) { k<37><<<blocks, 256>>>(d, it); });
    timeit("v_div_fixup_f32", w, [&](int it) { k<38><<<blocks, 256>>>(d, it); });
    timeit("v_ldexp_f32", w, [&](int it) { k<39><<<blocks, 256>>>(d, it); });
    timeit("v_add_f32_dpp", w, [&](int it) { k<40><<<blocks, 256>>>(d, it); });
    timeit("v_rcp_f32", w, [&](int it) { k<41><<<blocks, 256>>>(d, it); });
    timeit("v_exp_f32", w, [&](int it) { k<42><<<blocks, 256>>>(d, it); });
  }
  return 0;
}
