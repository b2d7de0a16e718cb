// Fused view-synthesis + photometric-loss kernels (forward and backward).
//
// The reference (/root/reference) has no counterpart of these stages; they are
// the warp / SSIM+L1 / minimum-reprojection / auto-mask / smoothness stack
// that follows the depth and pose networks in self-supervised training.
//
// Forward arithmetic that decides a validity bit or an arg-min index is
// written one IEEE operation at a time in a fixed association order and this
// file is compiled with -ffp-contract=off, so those outputs are reproducible
// bit for bit by any IEEE float32 evaluation of the same expressions.
// Backward arithmetic uses explicit fmaf and fast reciprocals where it helps.
//
// Launch sequence of one training step:
//   forward : smooth_kernel (+ depth maps) -> photo_tile_kernel<GRAD> -> loss_finalize_kernel
//   backward: backward_kernel (pose blocks + disparity-map blocks)
// photo_tile_kernel<true> does the whole per-pixel backward at unit upstream
// gradient while the tile is still on chip and leaves, per scale, the gradient
// with respect to the up-sampled disparity (a full-resolution map) and per-tile
// partial sums of the pose gradient in the workspace; the backward kernel
// gathers the adjoint of the up-sampling, adds the partials in a fixed order,
// adds the smoothness term and scales by the upstream gradient.  No
// floating-point atomics anywhere: results are bitwise reproducible.  The
// optional gradient with respect to the warped source images (flag
// DW_FLAG_COLOR_GRAD) is a scatter: photo_tile_color_kernel also stores the
// per-pixel adjoint of the warped colours, color_scatter_kernel accumulates the
// four-tap scatter of a tile in an LDS patch per source frame (64-bit fixed
// point, so the order of the additions does not matter) and writes the patch to
// a slot of its own with plain stores, and grad_color_kernel sums, per source
// pixel, the slots that cover it in a fixed order.  Only samples that leave
// their tile's patch (extreme poses) fall back to integer atomics in global
// memory.
//
// Tile geometry.  A block owns an "error-map region" Q of QH x QW pixels, one
// 1xSW strip per thread (NT = QH * QW/SW threads).  The colour / warped region R is
// Q plus one pixel on every side (3x3 SSIM window).  With gradients, the pixels
// a block is responsible for (its "core") are Q minus one pixel on every side,
// because the adjoint of the window sums at a pixel needs the error terms of
// its eight neighbours; tiles therefore step by (QH-2) x (QW-2).  Forward-only
// blocks have core = Q and step by QH x QW.
#include <atomic>

#include "dw_common.h"

namespace dw {

#ifndef DW_QH_GRAD
#define DW_QH_GRAD 16
#endif
#ifndef DW_BLOCKS_PER_CU
#define DW_BLOCKS_PER_CU 2
#endif
constexpr int QH_GRAD = DW_QH_GRAD;  // error-map rows per block, gradient kernel
constexpr int QH_FWD = 16;   // same, forward-only kernel
constexpr int QW = 32;       // error-map columns per block
#ifndef DW_SW
#define DW_SW 2
#endif
constexpr int SW = DW_SW;    // pixels per thread strip (1 x SW); 1, 2 or 4
constexpr int WW = SW + 2;   // strip window columns
constexpr int SPR = QW / SW; // strips per error-map row
constexpr int BLOCKS_PER_CU = DW_BLOCKS_PER_CU;  // what LDS and registers are budgeted for
#ifndef DW_FWD_BLOCKS_PER_CU
#define DW_FWD_BLOCKS_PER_CU 3
#endif
constexpr int FWD_BLOCKS_PER_CU = DW_FWD_BLOCKS_PER_CU;  // same, forward-only kernel
constexpr int RW = QW + 2;   // region columns
constexpr int RP = QW + 4;  // region row pitch in LDS (16-byte aligned rows)
constexpr int SM_PX = 1024;  // pixels per block in the per-scale map kernels
constexpr int SM_NT = 256;
#ifndef DW_PIPE_A
#define DW_PIPE_A 1
#endif
#ifndef DW_GD_PASSES
#define DW_GD_PASSES 4
#endif
constexpr int GD_PASSES = DW_GD_PASSES;  // coarse-scale pixels per lane group in the backward kernel
#ifndef DW_XCD_SWIZZLE
#define DW_XCD_SWIZZLE 1
#endif

template <bool GRAD>
struct Geom {
  static constexpr int QH = GRAD ? QH_GRAD : QH_FWD;
  static constexpr int NT = QH * SPR;
  // blocks per CU the kernel is compiled for: the forward-only kernel needs less LDS (35 KB) and
  // runs 5 - 7 % faster with three blocks in 168 registers than with two in 244
  static constexpr int PER_CU = GRAD ? BLOCKS_PER_CU : FWD_BLOCKS_PER_CU;
  static constexpr int MINW = NT / 64 * PER_CU / 4;  // waves per SIMD to stay under
  static constexpr int RH = QH + 2;
  static constexpr int CO = GRAD ? 1 : 0;  // core offset inside Q
  static constexpr int CH = QH - 2 * CO;   // core rows = tile step
  static constexpr int CW = QW - 2 * CO;   // core columns = tile step
};
constexpr int G_CH = Geom<true>::CH, G_CW = Geom<true>::CW;

struct ScaleLayout {
  int Hs[DW_MAX_SCALES], Ws[DW_MAX_SCALES];
  int nblk[DW_MAX_SCALES];          // blocks per image at scale s (map kernels)
  int blk_off[DW_MAX_SCALES + 1];   // prefix sum of B * nblk[s]
  long pix_off[DW_MAX_SCALES + 1];  // prefix sum of B * Hs * Ws
};

// Element `i` of a per-scale array of the kernel arguments.  The arguments live in constant
// memory; an index known only at run time is a scalar load of its own, and a block that first
// decodes "which scale am I" through three or four of them, one depending on the other, spends
// its first microsecond there -- as long as the short map blocks take for their work.  Reading
// all entries (fixed offsets: one wide load when the kernel starts) and selecting costs nothing.
template <typename T, int N>
__device__ __forceinline__ T sel(const T (&a)[N], int i) {
  T v = a[0];
#pragma unroll
  for (int j = 1; j < N; ++j) v = (i == j) ? a[j] : v;
  return v;
}

struct PhotoParams {
  int B, H, W, S;
  int tilesX, tilesY;  // of the kernel being launched
  int n_full;          // blocks [0, n_full) run every scale of one tile; the rest one scale each
  int n_tile_blocks;   // blocks of the tile launch that work on tiles; those behind them sum up the images' smoothness partials
  int vec4;            // every scale's width is a multiple of 4 and every map pointer 16-byte aligned
  int du_px, du_nblk;  // depth maps: pixels per thread (4 or 1) and blocks per sample
  // (H >> s) / H and (W >> s) / W as the float32 quotients the up-sampling formula uses: the same
  // for every thread, so the host divides once instead of every lane of every block
  float rsy[DW_MAX_SCALES], rsx[DW_MAX_SCALES];
  const float* disp[DW_MAX_SCALES];
  const float* color[3];
  const float* pyr[DW_MAX_SCALES];
  const float* K;
  const float* invK;
  const float* T;
  const float* tie;
  uint8_t* mask[DW_MAX_SCALES];
  // workspace sections
  float* tile_loss;  // [B][tiles][4]
  float* tile_dP;    // [B][tiles][S][24]
  float* depth_up;   // [S][B][H][W]            depth from the up-sampled disparity of every scale
  float* gdup;       // [S][B][H][W]            d loss / d (up-sampled disp[s]), photometric part
  float* gxout;      // [S][B][2][3][H][W]      d loss / d (warped colour of frame f), unit upstream; optional
  unsigned char* gany;  // [S][B][H][W]          bit f: frame f won somewhere in the pixel's 3 x 3 window
  float* img_scal;   // [S][B][2]               per image: mean of disp[s], sum of G * disp[s]
  double* sm_img;    // [S][B][2]               per image: its smoothness sums x, y times 1 / (mean disparity + 1e-7)
  float* sm_G;       // concatenated per scale [B][Hs][Ws]
  float* sm_part;    // [blk_off[S]][4]         per block: smoothness sums x, y; sum G * disp; sum disp
  ScaleLayout L;
};

// ------------------------------------------------------------------ geometry
__device__ __forceinline__ float upsampled_disp(const float* __restrict__ d, int hs, int ws,
                                                int s, int y, int x, float sy, float sx) {
  if (s == 0) return d[(long)y * ws + x];
  const Taps ty = upsample_taps(y, hs, sy);
  const Taps tx = upsample_taps(x, ws, sx);
  const float v00 = d[(long)ty.i0 * ws + tx.i0], v01 = d[(long)ty.i0 * ws + tx.i1];
  const float v10 = d[(long)ty.i1 * ws + tx.i0], v11 = d[(long)ty.i1 * ws + tx.i1];
  const float top = tx.l0 * v00 + tx.l1 * v01;
  const float bot = tx.l0 * v10 + tx.l1 * v11;
  return ty.l0 * top + ty.l1 * bot;
}

// Weight with which full-resolution index `dst` contributes to low-resolution
// index `j` in the bilinear up-sampling (0 if j is not one of its two taps).
// Equivalent tent form: max(0, 1 - |clamp(src, 0, in-1) - j|).
__device__ __forceinline__ float upsample_weight(int dst, int j, int in_size, float scale) {
  float src = fmaf((float)dst + 0.5f, scale, -0.5f);
  src = fminf(fmaxf(src, 0.f), (float)(in_size - 1));
  return fmaxf(0.f, 1.0f - fabsf(src - (float)j));
}

// Load from a base pointer plus a 32-bit per-lane byte offset.  With the base in
// scalar registers this is one load with no per-lane 64-bit address arithmetic.
typedef const __attribute__((address_space(1))) char* GlobalBytes;
typedef const __attribute__((address_space(1))) float* GlobalFloats;
__device__ __forceinline__ float tap(GlobalFloats base, unsigned byte_off) {
  return *(GlobalFloats)((GlobalBytes)base + byte_off);
}

// A pointer that is the same in every lane, moved to scalar registers (64-bit
// products of uniform values are computed by the vector unit and would
// otherwise stay there, and every load through them pays a 64-bit add).
// Off by default: with the bases in scalar registers the tile kernel spills
// more scalar registers than it saves vector instructions (measured 1 % slower).
#ifndef DW_UNIFORM_PTR
#define DW_UNIFORM_PTR 0
#endif
__device__ __forceinline__ GlobalFloats scalar_ptr(GlobalFloats ptr) {
  const unsigned long long v = reinterpret_cast<unsigned long long>(ptr);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (GlobalFloats)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ GlobalFloats uniform_ptr(const float* ptr) {
  return DW_UNIFORM_PTR ? scalar_ptr((GlobalFloats)ptr) : (GlobalFloats)ptr;
}

// IEEE division in fewer instructions.  hipcc expands n / d into: scale both
// operands if the quotient could leave the normal range (v_div_scale), y =
// reciprocal of d refined once, q = n * y corrected twice with the exact
// residual, undo the scaling (v_div_fmas), special cases (v_div_fixup).  When
// no scaling happens -- d normal, exponents of n and d less than 96 apart, n
// not below 2^-103 -- the two scale instructions are copies and v_div_fmas is a
// plain fma, so the sequence below returns the same bits with y computed once
// per divisor instead of once per division.  Every use in this file divides
// quantities inside that range (projected coordinates by a depth or by an image
// size, the SSIM numerator by its denominator); tools/micro/div_exact.hip checks
// the sequence against `/` over all 2^32 numerators for a set of image sizes
// and over 2^32 random pairs of each use.
__device__ __forceinline__ float div_rcp(float d) {
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

struct Sample {
  float val[3];
  float dx[3], dy[3];  // d val[c] / d (sample x, y) in pixels, before the border mask
};

// The three steps of warping one pixel into one source frame, separated so that
// the caller can put other work between issuing the loads and using them:
// where to sample, the twelve taps, the blend.
struct SampleAddr {
  unsigned o_nw, o_ne, o_sw, o_se;  // byte offsets of the four taps inside a colour plane
  // The two taps of a row are neighbours in memory, so the gather reads them as
  // one 8-byte pair starting at column min(x0, W - 2): o_n / o_s are the byte
  // offsets of the upper / lower pair.  At x0 = W - 1 both taps are the second
  // element of the pair and the fraction is 0; the blend then runs with tx = 1,
  // which gives the first element weight 0 and the same sums (x * 0 + y = y).
  unsigned o_n, o_s;
  int x0, y0, x1, y1;               // integer coordinates of the four taps
  float tx, ty;                     // fractional position inside the pair / cell
  // status bits of the sample, already at their place in the region's status byte
  // for source frame f: bit 2 + f inside the image before clamping and in front of
  // the camera; bits 6f, 6f + 1 strictly inside the image in x, y (the clamp is
  // inactive, so the sample position has a gradient).  Kept as one integer: a
  // boolean that lives across other work costs a pair of scalar registers.
  unsigned vb;
};
typedef float PairU __attribute__((ext_vector_type(2), aligned(4)));  // 8 bytes, dword aligned
typedef float QuadU __attribute__((ext_vector_type(4), aligned(4)));  // 16 bytes, dword aligned
struct SampleTaps {
  PairU n[3], s[3];  // upper and lower pair of every channel
};
// An 8-byte load from a base pointer plus a 32-bit per-lane byte offset (dword aligned).
__device__ __forceinline__ PairU tap2(GlobalFloats base, unsigned byte_off) {
  return *(const __attribute__((address_space(1))) PairU*)((GlobalBytes)base + byte_off);
}

// Project the camera-space point (X,Y,Z) with P (3x4, row-major); border
// padding, corners aligned.
// refined reciprocals of W - 1 and H - 1 (uniform, computed once per block)
struct DivConsts {
  float yw, yh;
};
template <bool WITH_GRAD>
__device__ __forceinline__ void sample_addr(int H, int W, const DivConsts& dc, const float* P, float X, float Y,
                                            float Z, unsigned f, SampleAddr& a) {
  const float c0 = ((P[0] * X + P[1] * Y) + P[2] * Z) + P[3];
  const float c1 = ((P[4] * X + P[5] * Y) + P[6] * Z) + P[7];
  const float c2 = ((P[8] * X + P[9] * Y) + P[10] * Z) + P[11];
  const float zp = c2 + kProjEps;
  const float yz = div_rcp(zp);
  const float u = div_by(c0, zp, yz);  // c0 / zp
  const float v = div_by(c1, zp, yz);  // c1 / zp
  const float wm1 = (float)(W - 1), hm1 = (float)(H - 1);
  // The formulation normalises to gx = (u / (W - 1) - 0.5) * 2, tests gx in [-1, 1] and
  // samples at ix = ((gx + 1) * 0.5) * (W - 1).  Scaling by 2 and by 0.5 is exact in binary
  // floating point, so with hx = u / (W - 1) - 0.5: gx in [-1, 1] <=> |hx| <= 0.5 and
  // (gx + 1) * 0.5 = hx + 0.5, bit for bit (an overflow of 2 * hx only meets clamped samples).
  const float hx = div_by(u, wm1, dc.yw) - 0.5f;  // u / (W - 1) - 0.5
  const float hy = div_by(v, hm1, dc.yh) - 0.5f;  // v / (H - 1) - 0.5
  const bool valid = (fabsf(hx) <= 0.5f) && (fabsf(hy) <= 0.5f) && (c2 > 0.f);
  a.vb = valid ? (4u << f) : 0u;
  float ix = (hx + 0.5f) * wm1;
  float iy = (hy + 0.5f) * hm1;
  if (WITH_GRAD) {
    a.vb |= ((ix > 0.f) && (ix < wm1)) ? (1u << (6u * f)) : 0u;
    a.vb |= ((iy > 0.f) && (iy < hm1)) ? (2u << (6u * f)) : 0u;
  }
  // median of (ix, 0, W - 1) = min(max(ix, 0), W - 1), one instruction (a NaN gives 0 either way)
  ix = __builtin_amdgcn_fmed3f(ix, 0.f, wm1);
  iy = __builtin_amdgcn_fmed3f(iy, 0.f, hm1);
  const float ix0 = floorf(ix), iy0 = floorf(iy);
  a.ty = iy - iy0;
  const int x0 = (int)ix0, y0 = (int)iy0;
  const int x1 = min(x0 + 1, W - 1), y1 = min(y0 + 1, H - 1);
  a.x0 = x0, a.y0 = y0, a.x1 = x1, a.y1 = y1;
  a.o_nw = (unsigned)(y0 * W + x0) * 4u;
  a.o_ne = (unsigned)(y0 * W + x1) * 4u;
  a.o_sw = (unsigned)(y1 * W + x0) * 4u;
  a.o_se = (unsigned)(y1 * W + x1) * 4u;
  const int xb = min(x0, W - 2);
  a.o_n = (unsigned)(y0 * W + xb) * 4u;
  a.o_s = (unsigned)(y1 * W + xb) * 4u;
  a.tx = (x0 > xb) ? 1.0f : ix - ix0;
}

// uniform plane base + 32-bit per-lane offset: no 64-bit address arithmetic per tap
__device__ __forceinline__ void sample_gather(GlobalFloats src, long plane,
                                              const SampleAddr& a, SampleTaps& t) {
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const GlobalFloats p = src + c * plane;
    t.n[c] = tap2(p, a.o_n);
    t.s[c] = tap2(p, a.o_s);
  }
}

template <bool WITH_GRAD>
__device__ __forceinline__ void sample_blend(const SampleAddr& a, const SampleTaps& t, Sample& o) {
  const float tx = a.tx, ty = a.ty;
  const float ex = 1.0f - tx, ey = 1.0f - ty;
  const float w_nw = ex * ey, w_ne = tx * ey, w_sw = ex * ty, w_se = tx * ty;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float nw = t.n[c].x, ne = t.n[c].y, sw = t.s[c].x, se = t.s[c].y;
    o.val[c] = ((nw * w_nw + ne * w_ne) + sw * w_sw) + se * w_se;
    if (WITH_GRAD) {
      o.dx[c] = fmaf(se - sw, ty, (ne - nw) * ey);
      o.dy[c] = fmaf(se - ne, tx, (sw - nw) * ex);
    }
  }
}

// The WW consecutive 32-bit words of one window row in LDS (start aligned to SW words).
template <typename T>
__device__ __forceinline__ void loadw(const T* p, T* o) {
  static_assert(sizeof(T) == 4, "32-bit elements");
  if (SW == 1) {
#pragma unroll
    for (int j = 0; j < WW; ++j) o[j] = p[j];
  } else if (SW == 4) {
    const int4 a = *reinterpret_cast<const int4*>(p);
    const int2 b = *reinterpret_cast<const int2*>(p + 4);
    const int v[6] = {a.x, a.y, a.z, a.w, b.x, b.y};
#pragma unroll
    for (int j = 0; j < WW; ++j) o[j] = __builtin_bit_cast(T, v[j]);
  } else {
    const int2 a = *reinterpret_cast<const int2*>(p);
    const int2 b = *reinterpret_cast<const int2*>(p + 2);
    const int v[4] = {a.x, a.y, b.x, b.y};
#pragma unroll
    for (int j = 0; j < WW; ++j) o[j] = __builtin_bit_cast(T, v[j % 4]);
  }
}

// A value that is the same in every lane, moved to a scalar register.
__device__ __forceinline__ float uniform(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}

struct SsimTerms {
  float A1, A2, B1, B2, r, raw;
};

__device__ __forceinline__ float ssim_loss_px(float mux, float muy, float exx, float eyy,
                                              float exy, SsimTerms& t) {
  const float pxy = mux * muy;
  const float sx = exx - mux * mux;
  const float sy = eyy - muy * muy;
  const float sxy = exy - pxy;
  // (2 * mux) * muy + C1 and 2 * sxy + C2: doubling is exact, so one fused
  // multiply-add by 2 rounds once, like the addition it replaces
  t.A1 = fmaf(2.0f, pxy, kC1);
  t.A2 = fmaf(2.0f, sxy, kC2);
  t.B1 = (mux * mux + muy * muy) + kC1;
  t.B2 = (sx + sy) + kC2;
  const float n = t.A1 * t.A2;
  const float d = t.B1 * t.B2;
  t.r = div_by(n, d, div_rcp(d));
  t.raw = (1.0f - t.r) * 0.5f;
  return __builtin_amdgcn_fmed3f(t.raw, 0.f, 1.f);  // clamp to [0, 1]
}

// 3x3 window sums of one strip: column sums (top + mid) + bot for the six
// columns, then (left + centre) + right per pixel.
__device__ __forceinline__ void colsum3(const float a[3][WW], float cs[WW]) {
#pragma unroll
  for (int j = 0; j < WW; ++j) cs[j] = (a[0][j] + a[1][j]) + a[2][j];
}
__device__ __forceinline__ void colsum3_prod(const float a[3][WW], const float b[3][WW], float cs[WW]) {
#pragma unroll
  for (int j = 0; j < WW; ++j) cs[j] = (a[0][j] * b[0][j] + a[1][j] * b[1][j]) + a[2][j] * b[2][j];
}
__device__ __forceinline__ float rowsum3(const float cs[WW], int p) {
  return ((cs[p] + cs[p + 1]) + cs[p + 2]) * kNinth;
}

// Diagnostic build only (-DDW_STAMPS): per-wave cycle sums of the kernel's
// segments, taken with s_memtime. The sums of one tile overwrite the start of
// the gdup workspace section, so that build's gradients are not usable.
#ifdef DW_STAMPS
#define DW_STAMP_DECL                               \
  unsigned long long t_last, t_acc[10];             \
  for (int k_ = 0; k_ < 10; ++k_) t_acc[k_] = 0;    \
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_last)::"memory");
#define DW_STAMP(k)                                                                  \
  {                                                                                  \
    unsigned long long t_now;                                                        \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_now)::"memory"); \
    t_acc[k] += t_now - t_last;                                                      \
    t_last = t_now;                                                                  \
  }
#else
#define DW_STAMP_DECL
#define DW_STAMP(k)
#endif

// ------------------------------------------------------------ depth of every scale
// depth_up[s][b][y][x] = 1 / (min_disp + range * up(disp[s])(y, x)): the bilinear up-sampling of
// every scale's disparity to full resolution and its inversion, once per pixel.  The tile kernel
// needs these values for its whole region (1.46 pixels per pixel it is responsible for); done
// here they cost one streaming pass instead of a latency-bound burst of 39 gathered taps and
// twelve exact divisions per thread at the start of every tile.
// It has no launch of its own: its blocks ride behind the blocks of the smoothness kernel,
// which precedes the tile kernel and is as short (two short launches cost more than their work).
constexpr int DU_NT = 256;
// PX pixels of a row per thread (4 when the width allows 16-byte stores, else 1); all taps of a
// thread are loaded before the first is used.  `blk`: block of sample `b`.
// (Reading the taps of the four pixels as one dword-aligned 16-byte load per tap row and picking
// them out of it -- 7 loads per thread instead of 52 -- is slower: 28 against 21 us for the
// launch; the lanes' 16-byte windows overlap and are not aligned.  Nor do 8 x 128 tiles whose three
// coarse rectangles are copied to LDS with row-contiguous loads, three per thread: 21.2 us.  The
// launch moves 70 MB in 21 us and is at what a launch this short gets out of the memory system.)
template <int PX>
__device__ __forceinline__ void depth_up_block(const PhotoParams& p, int b, int blk) {
  const int H = p.H, W = p.W;
  const int i = (blk * DU_NT + threadIdx.x) * PX;
  if (i >= H * W) return;
  const int y = i / W, x = i - y * W;
  // (one scale at a time would need half the registers, which also bound the smoothness blocks
  // of the launch; measured 2 us slower than having every tap in flight)
  float v[DW_MAX_SCALES][PX][4], wx0[DW_MAX_SCALES][PX], wx1[DW_MAX_SCALES][PX], wy0[DW_MAX_SCALES], wy1[DW_MAX_SCALES];
#pragma unroll
  for (int s = 0; s < DW_MAX_SCALES; ++s) {
    if (s < p.S) {
      const int hs = H >> s, ws = W >> s;
      const float* d = p.disp[s] + (long)b * hs * ws;
      const Taps ty = upsample_taps(y, hs, p.rsy[s]);
      wy0[s] = ty.l0;
      wy1[s] = ty.l1;
#pragma unroll
      for (int k = 0; k < PX; ++k) {
        if (s == 0) {  // not up-sampled
          if (PX == 4) {
            const float4 q = *reinterpret_cast<const float4*>(d + (long)y * ws + x);
            const float e = k == 0 ? q.x : (k == 1 ? q.y : (k == 2 ? q.z : q.w));
            v[s][k][0] = v[s][k][1] = v[s][k][2] = v[s][k][3] = e;
          } else {
            v[s][k][0] = v[s][k][1] = v[s][k][2] = v[s][k][3] = d[(long)y * ws + x + k];
          }
          wx0[s][k] = 1.f, wx1[s][k] = 0.f;
        } else {
          const Taps tx = upsample_taps(x + k, ws, p.rsx[s]);
          v[s][k][0] = d[ty.i0 * ws + tx.i0];
          v[s][k][1] = d[ty.i0 * ws + tx.i1];
          v[s][k][2] = d[ty.i1 * ws + tx.i0];
          v[s][k][3] = d[ty.i1 * ws + tx.i1];
          wx0[s][k] = tx.l0, wx1[s][k] = tx.l1;
        }
      }
    }
  }
#pragma unroll
  for (int s = 0; s < DW_MAX_SCALES; ++s) {
    if (s < p.S) {
      float out[PX];
#pragma unroll
      for (int k = 0; k < PX; ++k) {
        float du = v[s][k][0];
        if (s > 0) {  // the association of upsampled_disp
          const float top = wx0[s][k] * v[s][k][0] + wx1[s][k] * v[s][k][1];
          const float bot = wx0[s][k] * v[s][k][2] + wx1[s][k] * v[s][k][3];
          du = wy0[s] * top + wy1[s] * bot;
        }
        const float dsp = kMinDisp + kDispRange * du;
        out[k] = div_by(1.0f, dsp, div_rcp(dsp));  // 1 / dsp
      }
      float* o = p.depth_up + ((long)s * p.B + b) * H * W + i;
      if (PX == 4) {
        *reinterpret_cast<float4*>(o) = make_float4(out[0], out[1], out[2], out[3]);
      } else {
#pragma unroll
        for (int k = 0; k < PX; ++k) o[k] = out[k];
      }
    }
  }
}

// ------------------------------------------------------------ the tile kernel
// GRAD: with the per-pixel backward.  CGRAD (needs GRAD): also store the
// adjoint of the warped colours of both frames into p.gxout.
template <bool GRAD, bool CGRAD>
__device__ __forceinline__ void photo_tile_body(const PhotoParams& p) {
  using Gm = Geom<GRAD>;
  constexpr int QH = Gm::QH, NT = Gm::NT, RH = Gm::RH, CO = Gm::CO, CH = Gm::CH, CW = Gm::CW;
  constexpr int RN = RH * RP;  // floats per region plane
  constexpr int GN = QH * RP;  // floats per adjoint-coefficient plane
  constexpr int UNION = (6 * RN > 9 * GN) ? 6 * RN : 9 * GN;
  __shared__ __attribute__((aligned(16))) float s_tgt[3 * RN];
  // warped sources [2][3][RN] during stages A-B; SSIM adjoint coefficients
  // [9][GN] during stage C (written after every thread is done with A-B data)
  __shared__ __attribute__((aligned(16))) float s_un[GRAD ? UNION : 6 * RN];
  __shared__ float s_depth[DW_MAX_SCALES * RN];  // depth of every region pixel at every scale
  // per error-map pixel and frame: warped colour (3), its x / y derivatives (3 + 3); stage A -> C
  __shared__ __attribute__((aligned(16))) float s_st[GRAD ? 2 * 9 * QH * QW : 4];
  // per region pixel: bits 2,3 validity of source 0,1, bits 0,1 / 6,7 sample strictly inside the
  // image in x,y for source 0 / 1 (stage A); bits 4,5 winning frame + 1
  // (stage B; 0 = no re-projection won, or not a pixel of the error map)
  __shared__ __attribute__((aligned(8))) unsigned char s_valid[RN];
  __shared__ __attribute__((aligned(16))) float s_P[2][12];
  __shared__ float s_invK[9];
  __shared__ float s_red[NT / kWave];
  // per-wave partial sums of one scale: 24 pose entries and the loss
  __shared__ float s_dPw[GRAD ? (NT / kWave) * 25 : 1];

  DW_STAMP_DECL
  const int tid = threadIdx.x;
  // Block -> (sample, tile, scale range).  Consecutive block ids run on
  // consecutive XCDs (8 of them, one L2 each): give every XCD a contiguous run
  // of tiles so that neighbouring tiles share an L2.  The last tiles of the
  // launch may be split into one block per scale (see the host side).
  const int ntile = p.tilesX * p.tilesY;
  int tslot_i, s_lo = 0, s_hi = p.S;
  {
    const int L = blockIdx.x;
    if (L < p.n_full) {
#if DW_XCD_SWIZZLE
      const int x = L & 7, q = p.n_full >> 3, rem = p.n_full & 7;
      tslot_i = x * q + min(x, rem) + (L >> 3);
#else
      tslot_i = L;
#endif
    } else {
      const int k = L - p.n_full;
      tslot_i = p.n_full + k / p.S;
      s_lo = k - (k / p.S) * p.S;
      s_hi = s_lo + 1;
    }
  }
  const int b = tslot_i / ntile;
  const int tile = tslot_i - b * ntile;
  const int tY = tile / p.tilesX, tX = tile - tY * p.tilesX;
  const int cy0 = tY * CH, cx0 = tX * CW;        // image coords of the core origin
  const int qy0 = cy0 - CO, qx0 = cx0 - CO;      // ... of Q(0,0)
  const int ry0 = qy0 - 1, rx0 = qx0 - 1;        // ... of R(0,0)
  const int H = p.H, W = p.W;
  const long plane = H * W;  // 32-bit product (the host checks H * W <= 2^28): stays in scalar registers
  const int qr = tid / SPR, ql = tid % SPR;      // this thread's strip: Q row, Q cols SW*ql..
  const int win = qr * RP + SW * ql;              // region index of the strip's window corner

  if (tid < 24) {
    const int f = tid / 12, ij = tid - f * 12, i = ij >> 2, j = ij & 3;
    const float* Kb = p.K + (long)b * 16;
    const float* Tb = p.T + ((long)b * DW_NUM_SRC + f) * 16;
    s_P[f][ij] = ((Kb[i * 4 + 0] * Tb[0 * 4 + j] + Kb[i * 4 + 1] * Tb[1 * 4 + j]) +
                  Kb[i * 4 + 2] * Tb[2 * 4 + j]) +
                 Kb[i * 4 + 3] * Tb[3 * 4 + j];
  } else if (tid < 33) {
    const int k = tid - 24;
    s_invK[k] = p.invK[(long)b * 16 + (k / 3) * 4 + (k % 3)];
  }

  // ---- colour regions (reflect-mapped): target and the two un-warped sources
  const GlobalFloats tgt_b = uniform_ptr(p.color[0] + (long)b * 3 * plane);
  const GlobalFloats src_b[2] = {uniform_ptr(p.color[1] + (long)b * 3 * plane),
                           uniform_ptr(p.color[2] + (long)b * 3 * plane)};
  // Every global load of the prologue is issued before the first one is used
  // (nothing else is live yet, so registers are plentiful): all region pixels
  // of the thread, nine colour values and the depth of every scale each.  Loads
  // are unconditional -- a thread past the end of the region re-reads pixel 0
  // and a scale this block does not process re-reads one it does -- and only
  // the LDS stores are predicated.
  {
    constexpr int NIT = (RH * RW + NT - 1) / NT;
    float col[NIT][9];
    float dep[NIT][DW_MAX_SCALES];
    int lis[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      int i = tid + it * NT;
      const bool act = i < RH * RW;
      i = act ? i : 0;
      const int ry = i / RW, rx = i - ry * RW;
      const int y = reflect_index(ry0 + ry, H), x = reflect_index(rx0 + rx, W);
      const unsigned o = (unsigned)(y * W + x) * 4u;  // byte offset inside a plane
      lis[it] = act ? ry * RP + rx : -1;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        col[it][c] = tap(tgt_b + c * plane, o);
        col[it][3 + c] = tap(src_b[0] + c * plane, o);
        col[it][6 + c] = tap(src_b[1] + c * plane, o);
      }
      // depth of the pixel at every scale (the disparities were up-sampled and inverted once
      // per pixel by the depth blocks of the launch before this one: a tile copies its region)
#pragma unroll
      for (int s = 0; s < DW_MAX_SCALES; ++s) {
        const int sc = (s >= s_lo && s < s_hi) ? s : s_lo;
        dep[it][s] = tap(uniform_ptr(p.depth_up + ((long)sc * p.B + b) * plane), o);
      }
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int li = lis[it];
      if (li >= 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          s_tgt[c * RN + li] = col[it][c];
          s_un[(0 * 3 + c) * RN + li] = col[it][3 + c];
          s_un[(1 * 3 + c) * RN + li] = col[it][6 + c];
        }
      }
#pragma unroll
      for (int s = 0; s < DW_MAX_SCALES; ++s)
        if (s >= s_lo && s < s_hi && li >= 0) s_depth[s * RN + li] = dep[it][s];
    }
  }
  __syncthreads();
  DW_STAMP(8)  // region + depth loads

  // camera constants into scalar registers (uniform over the block)
  float P0[12], P1[12], iK[9];
#pragma unroll
  for (int k = 0; k < 12; ++k) {
    P0[k] = uniform(s_P[0][k]);
    P1[k] = uniform(s_P[1][k]);
  }
#pragma unroll
  for (int k = 0; k < 9; ++k) iK[k] = uniform(s_invK[k]);
  // reference value of the SSIM adjoint (see stage B): the target's colour at the centre of Q
  float ref[3] = {0.f, 0.f, 0.f};
  if (GRAD) {
#pragma unroll
    for (int c = 0; c < 3; ++c) ref[c] = uniform(s_tgt[c * RN + (QH / 2 + 1) * RP + QW / 2 + 1]);
  }
  DivConsts dc;
  dc.yw = uniform(div_rcp((float)(W - 1)));
  dc.yh = uniform(div_rcp((float)(H - 1)));

  // image coordinates and status of the pixels of this thread's strip
  const int sy_img = qy0 + qr;
  const int sx_img = qx0 + SW * ql;
  const bool row_in = (sy_img >= 0 && sy_img < H);
  const bool row_core = row_in && (qr >= CO) && (qr < QH - CO);

  // ---- target window statistics (kept in registers for every scale)
  float muy[3][SW], eyy[3][SW];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float y[3][WW], cs[WW], cyy[WW];
#pragma unroll
    for (int r = 0; r < 3; ++r) loadw(&s_tgt[c * RN + win + r * RP], y[r]);
    colsum3(y, cs);
    colsum3_prod(y, y, cyy);
#pragma unroll
    for (int q = 0; q < SW; ++q) {
      muy[c][q] = rowsum3(cs, q);
      eyy[c][q] = rowsum3(cyy, q);
    }
  }

  // Photometric errors of both frames for the strip, from the region images in
  // LDS. E[f][q]; with gradients also the SSIM adjoint coefficients Gc[f][q][9].
  auto strip_errors = [&](float E[2][SW], float Gc[2][SW][9], bool want_g) __attribute__((always_inline)) {
    float ss[2][SW], l1[2][SW];  // channel sums, accumulated in channel order
    // Six (channel, frame) bodies in sequence.  The window of the next body is
    // read from LDS at the start of the current one (ping-pong registers), so
    // its latency is covered by the arithmetic.
    float xw[2][3][WW], y[3][WW];
#pragma unroll
    for (int r = 0; r < 3; ++r) loadw(&s_un[win + r * RP], xw[0][r]);
#pragma unroll
    for (int idx = 0; idx < 6; ++idx) {
      const int c = idx >> 1, f = idx & 1;
      if (f == 0) {
#pragma unroll
        for (int r = 0; r < 3; ++r) loadw(&s_tgt[c * RN + win + r * RP], y[r]);
      }
      if (idx + 1 < 6) {
        const int cn = (idx + 1) >> 1, fn = (idx + 1) & 1;
#pragma unroll
        for (int r = 0; r < 3; ++r) loadw(&s_un[(fn * 3 + cn) * RN + win + r * RP], xw[(idx + 1) & 1][r]);
      }
      const float (*x)[WW] = xw[idx & 1];
      float cs[WW], cxx[WW], cxy[WW];
      colsum3(x, cs);
      colsum3_prod(x, x, cxx);
      colsum3_prod(x, y, cxy);
#pragma unroll
      for (int q = 0; q < SW; ++q) {
        const float mux = rowsum3(cs, q), exx = rowsum3(cxx, q), exy = rowsum3(cxy, q);
        SsimTerms t;
        const float sv = ssim_loss_px(mux, muy[c][q], exx, eyy[c][q], exy, t);
        const float lv = fabsf(x[1][q + 1] - y[1][q + 1]);
        ss[f][q] = (c == 0) ? sv : ss[f][q] + sv;
        l1[f][q] = (c == 0) ? lv : l1[f][q] + lv;
        if (GRAD && want_g) {
          // upstream: 0.85/3 from the channel mean, -1/2 from (1 - r)/2, 1/d; zero where the clamp is active
          const bool open = (t.raw >= 0.f) && (t.raw <= 1.f);
          // the 1/9 of the window mean is part of the constant, and the plane of the x * x
          // term carries the 2 of d(x * x): stage C then needs two fused multiply-adds per value
          const float up2 = open ? (-kWSsim * kThird * kNinth) * __builtin_amdgcn_rcpf(t.B1 * t.B2) : 0.f;
          // The adjoint of this window with respect to a pixel (x, y) inside it is linear:
          // up2 * (muy (A2 - A1) - r mux (B2 - B1) - r B1 x + A1 y).  Its constant term nearly
          // cancels the other two (B1 and A1 are of the order of the squared means, the result of
          // the order of the variances B2, A2), which in float32 costs a relative error of
          // 2^-24 * B1 / B2 per window -- visible in the pose gradient of flat, dark regions.
          // Written around a reference value near the window's values -- the target's centre
          // pixel of the tile (`ref`, per channel), the same for every window of the block so
          // that stage C can still sum the planes before it multiplies -- the terms that cancel
          // are formed from small differences: the constant term becomes
          // up2 * (muy A2 - r mux B2 - A1 (muy - ref) + r B1 (mux - ref)), to be used with
          // (x - ref) and (y - ref).  Measured on poses that flatten the warped image: pose
          // gradient 1.4e-4 -> 3e-5 from a float64 evaluation, the same as this stage in double.
          const float rB1 = t.r * t.B1;
          const float k0 = fmaf(rB1, mux - ref[c], fmaf(-t.A1, muy[c][q] - ref[c], fmaf(-(t.r * mux), t.B2, muy[c][q] * t.A2)));
          Gc[f][q][3 * c + 0] = up2 * k0;
          Gc[f][q][3 * c + 1] = up2 * -rB1;
          Gc[f][q][3 * c + 2] = up2 * t.A1;
        }
      }
      // keep the scheduler from interleaving the bodies: that would hold every
      // window of the strip in registers at once
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int q = 0; q < SW; ++q) {
        E[f][q] = kWSsim * (ss[f][q] * kThird) + kWL1 * (l1[f][q] * kThird);
      }
  };

  // ---- identity errors (scale independent): min of the two and its index
  float idm[SW];
  int idi[SW];
  {
    float E[2][SW];
    strip_errors(E, nullptr, false);
#pragma unroll
    for (int q = 0; q < SW; ++q) {
      const int x = sx_img + q;
      float e0 = E[0][q], e1 = E[1][q];
      if (p.tie && row_in && x >= 0 && x < W) {
        const long o = (long)sy_img * W + x;
        e0 = e0 + p.tie[((long)b * DW_NUM_SRC + 0) * plane + o];
        e1 = e1 + p.tie[((long)b * DW_NUM_SRC + 1) * plane + o];
      }
      idm[q] = e0;
      idi[q] = 0;
      if (e1 < e0) {
        idm[q] = e1;
        idi[q] = 1;
      }
    }
  }
  __syncthreads();  // un-warped sources no longer needed: stage A may overwrite them
  DW_STAMP(9)  // target statistics + identity errors

  const long tslot = tslot_i;

  for (int s = s_lo; s < s_hi; ++s) {
    float loss_acc = 0.f;

    DW_STAMP(0)
    // ---- stage A: warp both sources over the region.  The region is NFULL full
    // passes of the block plus LEFT pixels; when the rest is small it is shared
    // out over all waves with the two frames of a pixel in the two halves of a
    // wave, so that no wave runs a nearly empty pass while the others wait.
    constexpr int NFULL = (RH * RW) / NT, LEFT = RH * RW - NFULL * NT;
    constexpr int LEFT_PER_WAVE = (LEFT + NT / kWave - 1) / (NT / kWave);
    constexpr bool SPLIT_LEFT = LEFT > 0 && LEFT_PER_WAVE <= 32;
    // With two-pixel strips the full passes are exactly the error-map region Q (2 * NT pixels) and
    // the rest is exactly the one-pixel ring of R around it.  Ring pixels are needed for the
    // windows only: they are never stashed and need no derivatives, and no pixel of a full pass
    // has to ask whether it is stashed.
    constexpr bool QMAP = GRAD && SPLIT_LEFT && NFULL * NT == QH * QW && LEFT == 2 * RW + 2 * QH;
    struct Px {
      int li, sti;
      bool stash, act;
      int f;
    };
    // issue: geometry, projection into both frames and all 24 taps of region pixel i
    auto setup = [&](int i, Px& q, SampleAddr* qa, SampleTaps* qt) __attribute__((always_inline)) {
      // The region coordinates of the two full passes do not depend on the scale.  The gradient
      // kernel lets the compiler keep them in registers across the scale loop (saves their
      // half-rate integer arithmetic: 1 % of the kernel); the forward-only kernel, compiled for
      // 168 registers, recomputes them per scale instead of spilling.
      if (!GRAD) asm volatile("" : "+v"(i));
      const int ry = QMAP ? i / QW + 1 : i / RW, rx = QMAP ? i % QW + 1 : i - ry * RW;
      const int y = reflect_index(ry0 + ry, H), x = reflect_index(rx0 + rx, W);
      q.li = ry * RP + rx;
      q.stash = QMAP || ((unsigned)(ry - 1) < (unsigned)QH && (unsigned)(rx - 1) < (unsigned)QW);
      q.sti = (ry - 1) * QW + (rx - 1);
      const float depth = s_depth[s * RN + q.li];
      const float fx = (float)x, fy = (float)y;
      const float X = depth * ((iK[0] * fx + iK[1] * fy) + iK[2]);
      const float Y = depth * ((iK[3] * fx + iK[4] * fy) + iK[5]);
      const float Z = depth * ((iK[6] * fx + iK[7] * fy) + iK[8]);
      // both frames' taps are in flight before the first blend: frame 1 is
      // projected while the loads of frame 0 travel
      sample_addr<GRAD>(H, W, dc, P0, X, Y, Z, 0u, qa[0]);
      sample_gather(src_b[0], plane, qa[0], qt[0]);
      __builtin_amdgcn_sched_barrier(0);
      sample_addr<GRAD>(H, W, dc, P1, X, Y, Z, 1u, qa[1]);
      sample_gather(src_b[1], plane, qa[1], qt[1]);
      __builtin_amdgcn_sched_barrier(0);
    };
    // consume: blend, warped images / stash / status byte to LDS
    auto finish = [&](const Px& q, const SampleAddr* qa, const SampleTaps* qt) __attribute__((always_inline)) {
      unsigned char vb = 0;
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        Sample sm;
        sample_blend<GRAD>(qa[f], qt[f], sm);
        s_un[(f * 3 + 0) * RN + q.li] = sm.val[0];
        s_un[(f * 3 + 1) * RN + q.li] = sm.val[1];
        s_un[(f * 3 + 2) * RN + q.li] = sm.val[2];
        vb |= qa[f].vb;
        if (GRAD) {
          if (q.stash) {
            float* st = &s_st[f * 9 * QH * QW + q.sti];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
              st[(c + 0) * QH * QW] = sm.val[c];
              st[(c + 3) * QH * QW] = sm.dx[c];
              st[(c + 6) * QH * QW] = sm.dy[c];
            }
          }
        }
      }
      s_valid[q.li] = vb;
    };
    // the same for the shared-out rest: one frame of one pixel per lane
    auto setup_left = [&](Px& q, SampleAddr& qa, SampleTaps& qt) __attribute__((always_inline)) {
      const int lane = tid & (kWave - 1);
      const int wid = __builtin_amdgcn_readfirstlane(tid / kWave);
      const int h = lane & 31;  // pixel slot of the wave
      q.f = lane >> 5;          // source frame
      const int pi = wid * LEFT_PER_WAVE + h;
      q.act = h < LEFT_PER_WAVE && pi < LEFT;
      int i = NFULL * NT + (q.act ? pi : 0);
      asm volatile("" : "+v"(i));  // recomputed per scale: keeping these as well costs scalar-register spills
      int ry = i / RW, rx = i - ry * RW;
      if (QMAP) {
        // ring pixel k: top row, bottom row, then the left / right column pixel of rows 1 .. QH
        const int k = i - NFULL * NT;
        const int ks = k - 2 * RW;
        ry = k < RW ? 0 : (k < 2 * RW ? RH - 1 : 1 + (ks >> 1));
        rx = k < RW ? k : (k < 2 * RW ? k - RW : ((ks & 1) ? RW - 1 : 0));
      }
      const int y = reflect_index(ry0 + ry, H), x = reflect_index(rx0 + rx, W);
      q.li = ry * RP + rx;
      q.stash = !QMAP && (unsigned)(ry - 1) < (unsigned)QH && (unsigned)(rx - 1) < (unsigned)QW;
      q.sti = (ry - 1) * QW + (rx - 1);
      const float depth = s_depth[s * RN + q.li];
      const float fx = (float)x, fy = (float)y;
      const float X = depth * ((iK[0] * fx + iK[1] * fy) + iK[2]);
      const float Y = depth * ((iK[3] * fx + iK[4] * fy) + iK[5]);
      const float Z = depth * ((iK[6] * fx + iK[7] * fy) + iK[8]);
      // this lane's projection matrix, from LDS (a per-lane choice between the two
      // copies in scalar registers would be twelve selects)
      float Pf[12];
      {
        const float4* Pl = reinterpret_cast<const float4*>(&s_P[q.f][0]);
        const float4 a = Pl[0], bq = Pl[1], c = Pl[2];
        Pf[0] = a.x, Pf[1] = a.y, Pf[2] = a.z, Pf[3] = a.w;
        Pf[4] = bq.x, Pf[5] = bq.y, Pf[6] = bq.z, Pf[7] = bq.w;
        Pf[8] = c.x, Pf[9] = c.y, Pf[10] = c.z, Pf[11] = c.w;
      }
      const GlobalFloats srcf = q.f ? src_b[1] : src_b[0];
      sample_addr<GRAD>(H, W, dc, Pf, X, Y, Z, (unsigned)q.f, qa);
      sample_gather(srcf, plane, qa, qt);
    };
    auto finish_left = [&](const Px& q, const SampleAddr& qa, const SampleTaps& qt) __attribute__((always_inline)) {
      const int f = q.f;
      Sample sm;
      sample_blend<GRAD && !QMAP>(qa, qt, sm);
      const unsigned vb = qa.vb;
      const unsigned other = (unsigned)__shfl_xor((int)vb, 32, kWave);  // the other frame's bits of this pixel
      if (q.act) {
        s_un[(f * 3 + 0) * RN + q.li] = sm.val[0];
        s_un[(f * 3 + 1) * RN + q.li] = sm.val[1];
        s_un[(f * 3 + 2) * RN + q.li] = sm.val[2];
        if (GRAD && !QMAP && q.stash) {
          float* st = &s_st[f * 9 * QH * QW + q.sti];
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            st[(c + 0) * QH * QW] = sm.val[c];
            st[(c + 3) * QH * QW] = sm.dx[c];
            st[(c + 6) * QH * QW] = sm.dy[c];
          }
        }
        if (f == 0) s_valid[q.li] = (unsigned char)(vb | other);
      }
    };
    if (DW_PIPE_A && SPLIT_LEFT && NFULL == 2) {
      // software pipeline: the taps of the next pixel are in flight while the
      // current one is blended and stored
      Px q0, q1, ql;
      SampleAddr a0[2], a1[2], al;
      SampleTaps t0[2], t1[2], tl;
      setup(tid, q0, a0, t0);
      setup(tid + NT, q1, a1, t1);
      finish(q0, a0, t0);
      __builtin_amdgcn_sched_barrier(0);
      setup_left(ql, al, tl);
      __builtin_amdgcn_sched_barrier(0);
      finish(q1, a1, t1);
      __builtin_amdgcn_sched_barrier(0);
      finish_left(ql, al, tl);
    } else {
#pragma unroll 1
      for (int i0 = tid; i0 < (SPLIT_LEFT ? NFULL * NT : RH * RW); i0 += NT) {
        Px q;
        SampleAddr qa[2];
        SampleTaps qt[2];
        setup(i0, q, qa, qt);
        finish(q, qa, qt);
      }
      if (SPLIT_LEFT) {
        Px q;
        SampleAddr qa;
        SampleTaps qt;
        setup_left(q, qa, qt);
        finish_left(q, qa, qt);
      }
    }
    DW_STAMP(1)  // stage A
    __syncthreads();
    DW_STAMP(2)  // barrier after A

    // ---- stage B: errors of the strip, minimum, mask; adjoint coefficients
    float E[2][SW];
    float Gc[GRAD ? 2 : 1][SW][9];
    strip_errors(E, GRAD ? Gc : nullptr, true);
    int sel[SW];
#pragma unroll
    for (int q = 0; q < SW; ++q) {
      const int x = sx_img + q;
      const bool inside = row_in && x >= 0 && x < W;
      float m = idm[q];
      int idx = idi[q];
      if (E[0][q] < m) {
        m = E[0][q];
        idx = 2;
      }
      if (E[1][q] < m) {
        m = E[1][q];
        idx = 3;
      }
      sel[q] = (inside && idx >= 2) ? idx - 2 : -1;
      const int qc = SW * ql + q;
      const unsigned char vb = s_valid[win + RP + 1 + q];
      if (inside && row_core && qc >= CO && qc < QW - CO) {
        loss_acc += m;
        if (p.mask[s]) p.mask[s][(long)b * plane + (long)sy_img * W + x] = (uint8_t)(idx | (vb & 0x0c));
      }
      if (GRAD) s_valid[win + RP + 1 + q] = (unsigned char)(vb | ((sel[q] + 1) << 4));
    }

    DW_STAMP(3)  // stage B
    if (GRAD) {
      __syncthreads();  // every thread has read the warped images: reuse their LDS
      // coefficient planes are indexed [k][Q row][Q col + 1]
#pragma unroll
      for (int q = 0; q < SW; ++q) {
        const int gi = qr * RP + SW * ql + q + 1;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
          const float g = (sel[q] == 1) ? Gc[GRAD ? 1 : 0][q][k] : Gc[0][q][k];
          s_un[k * GN + gi] = (sel[q] < 0) ? 0.f : g;
        }
      }
      __syncthreads();

      DW_STAMP(4)  // coefficient planes + barriers
      // ---- stage C: adjoint of the window sums, L1, warp, projection, depth
      float gd[SW];
#pragma unroll
      for (int q = 0; q < SW; ++q) gd[q] = 0.f;
      float dP[2][12];
#pragma unroll
      for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int k = 0; k < 12; ++k) dP[f][k] = 0.f;
      if (row_core) {
        // a window row / column that was reflected onto this pixel's row /
        // column counts twice
        float wy[3], wx[SW][3];
        wy[0] = (sy_img == 1) ? 2.f : 1.f;
        wy[1] = 1.f;
        wy[2] = (sy_img == H - 2) ? 2.f : 1.f;
#pragma unroll
        for (int q = 0; q < SW; ++q) {
          const int x = sx_img + q;
          wx[q][0] = (x == 1) ? 2.f : 1.f;
          wx[q][1] = 1.f;
          wx[q][2] = (x == W - 2) ? 2.f : 1.f;
        }
        // winning frame of the 3 x WW neighbourhood (Q rows qr-1..qr+1, cols SW*ql-1..SW*ql+SW)
        int nsel[3][WW];
        unsigned wrow[3];
        const int gwin = (qr - 1) * RP + SW * ql;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          // region rows qr-1+1 .. qr+1+1, region columns SW*ql .. SW*ql+WW-1
          const unsigned char* vrow = &s_valid[(qr + r) * RP + SW * ql];
          unsigned w[2];
          if (SW == 1) {
            w[0] = (unsigned)vrow[0] | ((unsigned)vrow[1] << 8) | ((unsigned)vrow[2] << 16);
            w[1] = 0;
          } else if (SW == 4) {
            const uint2 w2 = *reinterpret_cast<const uint2*>(vrow);
            w[0] = w2.x;
            w[1] = w2.y;
          } else {
            w[0] = *reinterpret_cast<const unsigned short*>(vrow) |
                   ((unsigned)*reinterpret_cast<const unsigned short*>(vrow + 2) << 16);
            w[1] = 0;
          }
#pragma unroll
          for (int j = 0; j < WW; ++j) nsel[r][j] = (int)((w[j >> 2] >> (8 * (j & 3) + 4)) & 3) - 1;
          wrow[r] = w[0];
        }
        // Bits 4 and 5 of a status byte are one-hot (frame 0 / frame 1 won), so for
        // two-pixel strips, whose neighbourhood row is one 32-bit word, "did frame f
        // win anywhere in the 3 x 3 window of pixel q" is a mask test on the OR of
        // the three row words.
        bool any[2][SW];
        if (SW == 2) {
          const unsigned m_or = wrow[0] | wrow[1] | wrow[2];
          any[0][0] = (m_or & 0x00101010u) != 0;
          any[0][SW - 1] = (m_or & 0x10101000u) != 0;
          any[1][0] = (m_or & 0x00202020u) != 0;
          any[1][SW - 1] = (m_or & 0x20202000u) != 0;
        } else {
#pragma unroll
          for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int q = 0; q < SW; ++q) {
              bool a = false;
#pragma unroll
              for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int j = 0; j < 3; ++j) a = a || (nsel[r][q + j] == f);
              any[f][q] = a;
            }
        }
        // per-pixel geometry shared by both frames
        float depth[SW], r0[SW], r1[SW], r2[SW];
        bool live[SW];
#pragma unroll
        for (int q = 0; q < SW; ++q) {
          const int x = sx_img + q, qc = SW * ql + q;
          live[q] = (qc >= CO && qc < QW - CO && x < W) && (any[0][q] || any[1][q]);
          depth[q] = 0.f;
          r0[q] = r1[q] = r2[q] = 0.f;
          if (live[q]) {
            depth[q] = s_depth[s * RN + win + RP + 1 + q];
            const float fx = (float)x, fy = (float)sy_img;
            r0[q] = (iK[0] * fx + iK[1] * fy) + iK[2];
            r1[q] = (iK[3] * fx + iK[4] * fy) + iK[5];
            r2[q] = (iK[6] * fx + iK[7] * fy) + iK[8];
          }
        }
        float gdepth[SW];
#pragma unroll
        for (int q = 0; q < SW; ++q) gdepth[q] = 0.f;
        // A wave none of whose pixels has a re-projection winner in its 3 x 3 neighbourhood (a region
        // the auto-mask removed as a whole: static scene, texture-less sky) has no photometric
        // gradient at all: the adjoint below would add exact zeros.  Uniform branch, no barrier inside.
        bool wave_live = false;
#pragma unroll
        for (int q = 0; q < SW; ++q) wave_live = wave_live || live[q];
        if (CGRAD) {
          // which frames' warped colours carry a gradient at this pixel: the scatter kernel finds its
          // work in these bytes (12 MB) instead of scanning the adjoint planes (141 MB)
#pragma unroll
          for (int q = 0; q < SW; ++q) {
            const int x = sx_img + q, qc = SW * ql + q;
            if (qc >= CO && qc < QW - CO && x < W)
              p.gany[((long)s * p.B + b) * plane + (long)sy_img * W + x] =
                  (unsigned char)((any[0][q] ? 1 : 0) | (any[1][q] ? 2 : 0));
          }
        }
        if (CGRAD || __builtin_amdgcn_ballot_w64(wave_live) != 0ull) {
        // Window-sum adjoint, one colour channel (three coefficient planes) at a
        // time and both frames in the same pass over LDS: `tot` is the sum over
        // all neighbours, `af0` over the neighbours where frame 0 won; frame 1's
        // sum is their difference (the planes are zero where no frame won).
        // d err / d (sample position) accumulates per frame in gix / giy.
        float gix[2][SW], giy[2][SW];
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
          for (int q = 0; q < SW; ++q) gix[f][q] = giy[f][q] = 0.f;
        // weights of the frame-0 neighbours (row weight times mask), shared by the nine planes
        float w0[3][WW];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
          for (int j = 0; j < WW; ++j) {
            if (SW == 2) {
              // bit 4 of byte j as 0.0 / 1.0 (a byte-to-float conversion), times the row weight
              const unsigned won0 = (wrow[r] >> 4) & 0x01010101u;
              w0[r][j] = (float)((won0 >> (8 * j)) & 0xffu) * wy[r];
            } else {
              w0[r][j] = (nsel[r][j] == 0) ? wy[r] : 0.f;
            }
          }
#ifndef DW_C_PREFETCH
#define DW_C_PREFETCH 1
#endif
        float l1w[2][SW];  // weight of the L1 term: 0.15 / 3 where frame f won at pixel q
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
          for (int q = 0; q < SW; ++q) l1w[f][q] = (sel[q] == f) ? kWL1 * kThird : 0.f;
#if DW_C_PREFETCH
        // the coefficient planes of the next channel are read while the current one is summed
        float gbuf[2][3][3][WW];
#pragma unroll
        for (int k3 = 0; k3 < 3; ++k3)
#pragma unroll
          for (int r = 0; r < 3; ++r) loadw(&s_un[k3 * GN + gwin + r * RP], gbuf[0][k3][r]);
#pragma unroll
#else
#pragma unroll 1
#endif
        for (int c = 0; c < 3; ++c) {
          float tot[SW][3], af0[SW][3];
#if DW_C_PREFETCH
          if (c + 1 < 3) {
#pragma unroll
            for (int k3 = 0; k3 < 3; ++k3)
#pragma unroll
              for (int r = 0; r < 3; ++r)
                loadw(&s_un[(3 * (c + 1) + k3) * GN + gwin + r * RP], gbuf[(c + 1) & 1][k3][r]);
          }
#endif
#pragma unroll
          for (int k3 = 0; k3 < 3; ++k3) {
            float cs[WW], cs0[WW];
#if DW_C_PREFETCH
            const float (*g)[WW] = gbuf[c & 1][k3];
#else
            float g[3][WW];
#pragma unroll
            for (int r = 0; r < 3; ++r) loadw(&s_un[(3 * c + k3) * GN + gwin + r * RP], g[r]);
#endif
#pragma unroll
            for (int j = 0; j < WW; ++j) {
              cs[j] = fmaf(wy[0], g[0][j], fmaf(wy[2], g[2][j], g[1][j]));
              cs0[j] = fmaf(w0[0][j], g[0][j], fmaf(w0[2][j], g[2][j], w0[1][j] * g[1][j]));
            }
#pragma unroll
            for (int q = 0; q < SW; ++q) {
              tot[q][k3] = fmaf(wx[q][0], cs[q], fmaf(wx[q][2], cs[q + 2], cs[q + 1]));
              af0[q][k3] = fmaf(wx[q][0], cs0[q], fmaf(wx[q][2], cs0[q + 2], cs0[q + 1]));
            }
          }
#pragma unroll
          for (int q = 0; q < SW; ++q) {
            const float yq = s_tgt[c * RN + win + RP + 1 + q];
            const float yc = yq - ref[c];  // relative to the block's reference value
#pragma unroll
            for (int f = 0; f < 2; ++f) {
              const float* st = &s_st[f * 9 * QH * QW + qr * QW + SW * ql + q];
              const float a0 = f ? tot[q][0] - af0[q][0] : af0[q][0];
              const float a1 = f ? tot[q][1] - af0[q][1] : af0[q][1];
              const float a2 = f ? tot[q][2] - af0[q][2] : af0[q][2];
              const float xq = st[c * QH * QW];
              float gx = fmaf(xq - ref[c], a1, fmaf(yc, a2, a0));
              // L1 term where this frame won: 0.15 / 3 with the sign of x - y, nothing at x = y
              const float df = xq - yq;
              gx += (df != 0.f) ? __builtin_copysignf(l1w[f][q], df) : 0.f;
              gix[f][q] = fmaf(gx, st[(c + 3) * QH * QW], gix[f][q]);
              giy[f][q] = fmaf(gx, st[(c + 6) * QH * QW], giy[f][q]);
              if (CGRAD) {
                // adjoint of the warped colour itself, for the scatter to the source image
                const int x = sx_img + q, qc = SW * ql + q;
                // (stored for every pixel, zeros included: whole cache lines are cheaper to write than
                // the lanes with a gradient alone, 309 against 345 us for the launch)
                if (qc >= CO && qc < QW - CO && x < W)
                  p.gxout[((((long)s * p.B + b) * 2 + f) * 3 + c) * plane + (long)sy_img * W + x] =
                      (live[q] && any[f][q]) ? gx : 0.f;
              }
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        // Branch-free: a pixel without a contribution runs the same arithmetic on
        // zeros (its depth, its gradient and the reciprocal are zeroed), which adds
        // exact zeros to the sums.  The projection matrix is read from LDS here --
        // in scalar registers it would be live across the whole scale loop.
#pragma unroll
        for (int f = 0; f < 2; ++f) {
          float P[12];
          {
            const float4 a = *reinterpret_cast<const float4*>(&s_P[f][0]);
            const float4 bq = *reinterpret_cast<const float4*>(&s_P[f][4]);
            const float4 c = *reinterpret_cast<const float4*>(&s_P[f][8]);
            P[0] = a.x, P[1] = a.y, P[2] = a.z, P[3] = a.w;
            P[4] = bq.x, P[5] = bq.y, P[6] = bq.z, P[7] = bq.w;
            P[8] = c.x, P[9] = c.y, P[10] = c.z, P[11] = c.w;
          }
#pragma unroll
          for (int q = 0; q < SW; ++q) {
            const bool on = live[q] && any[f][q];
            const float X = depth[q] * r0[q], Y = depth[q] * r1[q], Z = depth[q] * r2[q];
            // projection of the pixel again (registers only: no memory access)
            const float c0 = fmaf(P[0], X, fmaf(P[1], Y, fmaf(P[2], Z, P[3])));
            const float c1 = fmaf(P[4], X, fmaf(P[5], Y, fmaf(P[6], Z, P[7])));
            const float c2 = fmaf(P[8], X, fmaf(P[9], Y, fmaf(P[10], Z, P[11])));
            const float iz = on ? __builtin_amdgcn_rcpf(c2 + kProjEps) : 0.f;
            const float pu = c0 * iz, pv = c1 * iz;
            // status byte of the pixel itself: byte q + 1 of the centre row word read above
            const unsigned vbits = (SW == 2 ? (wrow[1] >> (8 * (q + 1))) & 0xffu
                                            : (unsigned)s_valid[win + RP + 1 + q]) >> (6 * f);
            const float gu = (on && (vbits & 1)) ? gix[f][q] : 0.f;
            const float gv = (on && (vbits & 2)) ? giy[f][q] : 0.f;
            const float gc0 = gu * iz, gc1 = gv * iz;
            const float gc2 = -(gu * pu + gv * pv) * iz;
            dP[f][0] = fmaf(gc0, X, dP[f][0]);
            dP[f][1] = fmaf(gc0, Y, dP[f][1]);
            dP[f][2] = fmaf(gc0, Z, dP[f][2]);
            dP[f][3] += gc0;
            dP[f][4] = fmaf(gc1, X, dP[f][4]);
            dP[f][5] = fmaf(gc1, Y, dP[f][5]);
            dP[f][6] = fmaf(gc1, Z, dP[f][6]);
            dP[f][7] += gc1;
            dP[f][8] = fmaf(gc2, X, dP[f][8]);
            dP[f][9] = fmaf(gc2, Y, dP[f][9]);
            dP[f][10] = fmaf(gc2, Z, dP[f][10]);
            dP[f][11] += gc2;
            const float gX = gc0 * P[0] + gc1 * P[4] + gc2 * P[8];
            const float gY = gc0 * P[1] + gc1 * P[5] + gc2 * P[9];
            const float gZ = gc0 * P[2] + gc1 * P[6] + gc2 * P[10];
            gdepth[q] += gX * r0[q] + gY * r1[q] + gZ * r2[q];
          }
        }
        }  // wave_live
#pragma unroll
        for (int q = 0; q < SW; ++q) gd[q] = gdepth[q] * (-kDispRange * depth[q] * depth[q]);
      }
      DW_STAMP(5)  // stage C
      // pose partials and the loss of this scale: one value per wave and entry.
      // Four entries share a register (two half-wave exchanges fold each onto one
      // row of 16 lanes), then four DPP steps sum the rows: 7 registers instead of
      // 25 full six-step reductions.
      {
        const int lane = tid & (kWave - 1), wid = tid / kWave;
        float v[28];
#pragma unroll
        for (int k = 0; k < 24; ++k) v[k] = dP[k / 12][k % 12];
        v[24] = loss_acc;
        v[25] = v[26] = v[27] = 0.f;
        float t[7];
#pragma unroll
        for (int g = 0; g < 7; ++g) t[g] = wave_sum4_fold(v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]);
#pragma unroll
        for (int g = 0; g < 7; ++g) t[g] = dpp_add<0xb1, 0xf>(t[g]);   // quad_perm:[1,0,3,2]
#pragma unroll
        for (int g = 0; g < 7; ++g) t[g] = dpp_add<0x4e, 0xf>(t[g]);   // quad_perm:[2,3,0,1]
#pragma unroll
        for (int g = 0; g < 7; ++g) t[g] = dpp_add<0x124, 0xf>(t[g]);  // row_ror:4
#pragma unroll
        for (int g = 0; g < 7; ++g) t[g] = dpp_add<0x128, 0xf>(t[g]);  // row_ror:8
        if ((lane & 15) == 0) {
          // row 0 / 1 / 2 / 3 of register g holds entry 4g + 0 / 2 / 1 / 3
          const int row = lane >> 4;
          const int e = ((row & 1) << 1) | (row >> 1);
#pragma unroll
          for (int g = 0; g < 7; ++g)
            if (4 * g + e < 25) s_dPw[wid * 25 + 4 * g + e] = t[g];
        }
      }
      // after the next barrier: per-tile sums over the waves, fixed order
      auto finish_sums = [&]() __attribute__((always_inline)) {
        if (tid < 25) {
          float v = 0.f;
#pragma unroll
          for (int w = 0; w < NT / kWave; ++w) v += s_dPw[w * 25 + tid];
          if (tid < 24)
            p.tile_dP[(tslot * p.S + s) * 24 + tid] = v;  // one entry of the 2x3x4 projection gradient
          else
            p.tile_loss[tslot * DW_MAX_SCALES + s] = v;
        }
      };
      // d loss / d (up-sampled disparity) of the core pixels; the adjoint of the
      // up-sampling itself is a gather in the backward kernel
      if (row_core) {
        float* gout = p.gdup + ((long)s * p.B + b) * plane + (long)sy_img * W;
#pragma unroll
        for (int q = 0; q < SW; ++q) {
          const int x = sx_img + q, qc = SW * ql + q;
          if (qc >= CO && qc < QW - CO && x < W) gout[x] = gd[q];
        }
      }
      __syncthreads();  // also separates this scale's LDS traffic from the next stage A
      finish_sums();
    }
    DW_STAMP(6)  // pose partials, gradient store
    if (!GRAD) {
      // per-tile partial sum (fixed order: reproducible run to run); its barriers
      // also separate this scale's LDS traffic from the next stage A
      const float t = block_sum<NT>(loss_acc, s_red);
      if (tid == 0) p.tile_loss[tslot * DW_MAX_SCALES + s] = t;
    }
    DW_STAMP(7)  // loss partial sum
  }
#ifdef DW_STAMPS
  if (GRAD && (tid & 63) == 0 && tile == p.tilesX * (p.tilesY / 2) + p.tilesX / 2 && b == p.B / 2 && s_lo == 0)
    for (int k = 0; k < 10; ++k) p.gdup[(tid / 64) * 10 + k] = (float)t_acc[k];
#endif
}

// The blocks behind the tile blocks of a tile launch, one image (scale, sample) per wave: the
// image's smoothness partials (left by the smoothness kernel, the launch before) summed up --
// mean disparity, sum of G * disp for the backward pass, and the two smoothness sums scaled by
// 1 / (mean + 1e-7) for the loss.  These few short chains of dependent loads used to be the
// larger part of the loss kernel that follows the tiles; here they run in slots the last,
// partial round of tiles leaves free.
template <int NT>
__device__ __forceinline__ void image_sums_block(const PhotoParams& p, int k) {
  const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;
  const int m = k * (NT / kWave) + wid;
  if (m >= p.S * p.B) return;
  const int s = m / p.B, b = m - s * p.B;
  const int nblk = sel(p.L.nblk, s);
  const long base = (long)sel(p.L.blk_off, s) + (long)b * nblk;
  float ax = 0.f, ay = 0.f, ag = 0.f, ad = 0.f;
#pragma unroll 2
  for (int j = lane; j < nblk; j += kWave) {
    const float4 v = *reinterpret_cast<const float4*>(p.sm_part + (base + j) * 4);
    ax += v.x;
    ay += v.y;
    ag += v.z;
    ad += v.w;
  }
  ax = wave_sum(ax);
  ay = wave_sum(ay);
  ag = wave_sum(ag);
  ad = wave_sum(ad);
  if (lane == kSumLane) {
    const float mean = ad / (float)((long)(p.H >> s) * (p.W >> s));
    const double inv = (double)(1.0f / (mean + kMeanEps));
    p.img_scal[2 * m + 0] = mean;
    p.img_scal[2 * m + 1] = ag;
    p.sm_img[2 * m + 0] = inv * (double)ax;
    p.sm_img[2 * m + 1] = inv * (double)ay;
  }
}

template <bool GRAD>
__global__ __launch_bounds__(Geom<GRAD>::NT, Geom<GRAD>::MINW) void photo_tile_kernel(const PhotoParams p) {
  if ((int)blockIdx.x >= p.n_tile_blocks) {
    image_sums_block<Geom<GRAD>::NT>(p, (int)blockIdx.x - p.n_tile_blocks);
    return;
  }
  photo_tile_body<GRAD, false>(p);
}
__global__ __launch_bounds__(Geom<true>::NT, Geom<true>::MINW) void photo_tile_color_kernel(const PhotoParams p) {
  if ((int)blockIdx.x >= p.n_tile_blocks) {
    image_sums_block<Geom<true>::NT>(p, (int)blockIdx.x - p.n_tile_blocks);
    return;
  }
  photo_tile_body<true, true>(p);
}

// Decode a block of the per-scale map kernels: scale, sample, chunk.
__device__ __forceinline__ void decode_map_block(const ScaleLayout& L, int S, int blk, int& s,
                                                 int& b, int& chunk) {
  s = 0;
#pragma unroll
  for (int j = 1; j < DW_MAX_SCALES; ++j) s += (j < S && blk >= L.blk_off[j]) ? 1 : 0;  // blk_off ascends
  const int r = blk - sel(L.blk_off, s);
  const int nb = sel(L.nblk, s);
  b = r / nb;
  chunk = r - b * nb;
}

// ------------------------------------------------- edge-aware smoothness
// loss_s = mean_x(|d^(x) - d^(x+1)| * exp(-mean_c|I(x) - I(x+1)|)) + same in y,
// d^ = disp / (mean + 1e-7), mean = the mean of disp over the image.  The
// normalisation is a positive factor per image, so this kernel does not need
// it: it sums |d(x) - d(x+1)| * w on the raw disparity (the finalisation kernel
// multiplies the sums of an image by 1 / (mean + 1e-7)), and G = d loss_s / d d^
// depends on the differences only through their signs.  Writes G per pixel and
// per-block partials (sum_x, sum_y, sum G*disp, sum disp): the sum of disp is
// what the mean is made of, so no pass over disp precedes this kernel.
template <bool GRAD>
__global__ __launch_bounds__(SM_NT) void smooth_kernel(const PhotoParams p) {
  static_assert(SM_NT == DU_NT, "the depth blocks run in this kernel's launch");
  const int n_smooth_blocks = sel(p.L.blk_off, p.S);
  if ((int)blockIdx.x >= n_smooth_blocks) {  // the blocks behind the smoothness blocks: depth maps
    const int k = (int)blockIdx.x - n_smooth_blocks;
    if (p.du_px == 4)
      depth_up_block<4>(p, k / p.du_nblk, k % p.du_nblk);
    else
      depth_up_block<1>(p, k / p.du_nblk, k % p.du_nblk);
    return;
  }
  int s, b, chunk;
  decode_map_block(p.L, p.S, blockIdx.x, s, b, chunk);
  const int hs = p.H >> s, ws = p.W >> s;
  const long n = (long)hs * ws;
  const float* d = sel(p.disp, s) + (long)b * n;
  const float* I = sel(p.pyr, s) + (long)b * 3 * n;
  const long pix_off = sel(p.L.pix_off, s);
  const float icx = 1.0f / ((float)p.B * hs * (ws - 1));
  const float icy = 1.0f / ((float)p.B * (hs - 1) * ws);
  float lx = 0.f, ly = 0.f, gd = 0.f, sd = 0.f;
  auto sgn = [](float a) -> float { return (a > 0.f) ? 1.f : ((a < 0.f) ? -1.f : 0.f); };
  if (p.vec4) {
    // Rows are whole float4s: a thread takes four consecutive pixels of one
    // row, reads them and their upper / lower neighbours as 16-byte vectors (a
    // missing neighbour row re-reads the centre row and is masked) and the two
    // pixels beside the group as scalars.  Same per-pixel arithmetic as below.
    const long i0 = (long)chunk * SM_PX + 4 * threadIdx.x;
    const bool act = i0 < n;
    const long ia = act ? i0 : 0;
    const int y = (int)((unsigned)ia / (unsigned)ws), x0 = (int)((unsigned)ia - (unsigned)y * (unsigned)ws);
    const bool up = y > 0, dn = y + 1 < hs, lf = x0 > 0, rt = x0 + 4 < ws;
    const long iu = up ? ia - ws : ia, id = dn ? ia + ws : ia;
    float C[4][6], U[4][4], D[4][4];  // [plane: disp, I0, I1, I2][column]; C holds x0-1 .. x0+4
#pragma unroll
    for (int pl = 0; pl < 4; ++pl) {
      const float* src = (pl == 0) ? d : I + (long)(pl - 1) * n;
      const float4 c = *reinterpret_cast<const float4*>(src + ia);
      const float4 u = *reinterpret_cast<const float4*>(src + iu);
      const float4 w = *reinterpret_cast<const float4*>(src + id);
      C[pl][0] = src[lf ? ia - 1 : ia];
      C[pl][5] = src[rt ? ia + 4 : ia];
      C[pl][1] = c.x; C[pl][2] = c.y; C[pl][3] = c.z; C[pl][4] = c.w;
      U[pl][0] = u.x; U[pl][1] = u.y; U[pl][2] = u.z; U[pl][3] = u.w;
      D[pl][0] = w.x; D[pl][1] = w.y; D[pl][2] = w.z; D[pl][3] = w.w;
    }
    if (act) {
      float Gv[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float dc = C[0][e + 1];
        const float c0 = C[1][e + 1], c1 = C[2][e + 1], c2 = C[3][e + 1];
        auto edge = [&](float n0, float n1, float n2) -> float {  // exp(-mean_c |I(i) - I(j)|)
          const float m = (fabsf(c0 - n0) + fabsf(c1 - n1) + fabsf(c2 - n2)) * kThird;
          return __expf(-m);
        };
        float G = 0.f;
        if (e < 3 || rt) {
          const float a = dc - C[0][e + 2], w = edge(C[1][e + 2], C[2][e + 2], C[3][e + 2]);
          lx += fabsf(a) * w;
          G += sgn(a) * w * icx;
        }
        if (e > 0 || lf) {
          const float a = C[0][e] - dc, w = edge(C[1][e], C[2][e], C[3][e]);
          G -= sgn(a) * w * icx;
        }
        if (dn) {
          const float a = dc - D[0][e], w = edge(D[1][e], D[2][e], D[3][e]);
          ly += fabsf(a) * w;
          G += sgn(a) * w * icy;
        }
        if (up) {
          const float a = U[0][e] - dc, w = edge(U[1][e], U[2][e], U[3][e]);
          G -= sgn(a) * w * icy;
        }
        Gv[e] = G;
        sd += dc;
        if (GRAD) gd = fmaf(G, dc, gd);
      }
      if (GRAD)
        *reinterpret_cast<float4*>(p.sm_G + pix_off + (long)b * n + i0) =
            make_float4(Gv[0], Gv[1], Gv[2], Gv[3]);
    }
  } else {
  for (int k = threadIdx.x; k < SM_PX; k += SM_NT) {
    const long i = (long)chunk * SM_PX + k;
    if (i >= n) break;
    // 32-bit index arithmetic (n < 2^30): a 64-bit division costs more than the rest of the pixel
    const int y = (int)((unsigned)i / (unsigned)ws), x = (int)((unsigned)i - (unsigned)y * (unsigned)ws);
    const float dc = d[i];
    const float c0 = I[i], c1 = I[n + i], c2 = I[2 * n + i];
    float G = 0.f;
    auto edge = [&](long j) -> float {  // exp(-mean_c |I(i) - I(j)|)
      const float m = (fabsf(c0 - I[j]) + fabsf(c1 - I[n + j]) + fabsf(c2 - I[2 * n + j])) * kThird;
      return __expf(-m);
    };
    if (x + 1 < ws) {
      const float a = dc - d[i + 1], w = edge(i + 1);
      lx += fabsf(a) * w;
      G += sgn(a) * w * icx;
    }
    if (x > 0) {
      const float a = d[i - 1] - dc, w = edge(i - 1);
      G -= sgn(a) * w * icx;
    }
    if (y + 1 < hs) {
      const float a = dc - d[i + ws], w = edge(i + ws);
      ly += fabsf(a) * w;
      G += sgn(a) * w * icy;
    }
    if (y > 0) {
      const float a = d[i - ws] - dc, w = edge(i - ws);
      G -= sgn(a) * w * icy;
    }
    sd += dc;
    if (GRAD) {
      p.sm_G[pix_off + (long)b * n + i] = G;
      gd = fmaf(G, dc, gd);
    }
  }
  }
  // the four block sums behind one barrier; thread k adds the wave values of sum k
  __shared__ float red4[4][SM_NT / kWave];
  lx = wave_sum(lx);
  ly = wave_sum(ly);
  gd = GRAD ? wave_sum(gd) : 0.f;
  sd = wave_sum(sd);
  const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;
  if (lane == kSumLane) {
    red4[0][wid] = lx;
    red4[1][wid] = ly;
    red4[2][wid] = gd;
    red4[3][wid] = sd;
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < SM_NT / kWave; ++i) t += red4[threadIdx.x][i];
    p.sm_part[(long)blockIdx.x * 4 + threadIdx.x] = t;
  }
}

// ------------------------------------------------- scalar loss
// One block of 1024 threads: the S tile-loss sums are accumulated in double in one pass over the
// partials (the S partials of a tile are one 16-byte load), reduced per wave with shuffles and
// across waves by one thread; the smoothness sums of the images (left by the blocks behind the
// tiles) are added per scale in sample order.  Fixed order throughout.
constexpr int LF_WAVES = 1024 / kWave;
__global__ __launch_bounds__(1024) void loss_finalize_kernel(const PhotoParams p, float* loss_out) {
  __shared__ double s_part[3 * DW_MAX_SCALES][LF_WAVES];
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wid = tid / kWave;
  const int ntile = p.B * p.tilesX * p.tilesY;
  double acc[DW_MAX_SCALES];
#pragma unroll
  for (int k = 0; k < DW_MAX_SCALES; ++k) acc[k] = 0.0;
  const float4* tl = reinterpret_cast<const float4*>(p.tile_loss);
#pragma unroll 4
  for (int t = tid; t < ntile; t += 1024) {
    const float4 v = tl[t];
    acc[0] += (double)v.x;
    acc[1] += (double)v.y;
    acc[2] += (double)v.z;
    acc[3] += (double)v.w;
  }
#pragma unroll
  for (int k = 0; k < DW_MAX_SCALES; ++k) {
    double a = acc[k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off, kWave);
    if (lane == 0) s_part[3 * k][wid] = a;
  }
  if (wid < 2 * DW_MAX_SCALES) {  // wave 2 s + c: smoothness sum c (x, y) of scale s over the samples
    const int sc = wid >> 1, c = wid & 1;
    double a = 0.0;
    if (sc < p.S)
      for (int b = lane; b < p.B; b += kWave) a += p.sm_img[2 * (sc * p.B + b) + c];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off, kWave);
    if (lane == 0) s_part[3 * sc + 1 + c][0] = a;
  }
  __syncthreads();
  if (tid < DW_MAX_SCALES) {  // thread k finishes the tile-loss sum of scale k
    double a = 0.0;
    for (int w = 0; w < LF_WAVES; ++w) a += s_part[3 * tid][w];
    s_part[3 * tid][0] = a;
  }
  __syncthreads();
  if (tid == 0) {
    double total = 0.0;
    for (int s = 0; s < p.S; ++s) {
      const int hs = p.H >> s, ws = p.W >> s;
      const double photo = s_part[3 * s][0] / ((double)p.B * p.H * p.W);
      const double sm = s_part[3 * s + 1][0] / ((double)p.B * hs * (ws - 1)) +
                        s_part[3 * s + 2][0] / ((double)p.B * (hs - 1) * ws);
      loss_out[1 + s] = (float)photo;
      loss_out[1 + p.S + s] = (float)sm;
      total += photo + (double)kWSmooth * sm / (double)(1 << s);
    }
    loss_out[0] = (float)(total / p.S);
  }
}

// ------------------------------------------------- backward: d loss / d disp[s]
struct BwdParams {
  int B, H, W, S;
  int tilesX, tilesY;
  const float* disp[DW_MAX_SCALES];
  const float* K;
  const float* grad_loss;  // upstream gradient of loss_out: [0] the total; with g_len = 1 + 2S also
  int g_len;               // [1 + s] of L_photo[s] and [1 + S + s] of L_sm[s]
  int vec4;                // full resolution: H * W is a multiple of 4 and grad_disp[0] is 16-byte aligned
  const float* tile_dP;
  const float* gdup;
  const float* img_scal;
  const float* sm_G;
  const float* sm_part;
  float* grad_disp[DW_MAX_SCALES];
  float* grad_T;
  // grid of backward_kernel: n_pose blocks for d loss / d T first, then the blocks
  // of d loss / d disp[s] from the coarsest scale (the longest blocks) to the finest.
  // Pixels per block and blocks per image of every scale; g_off[k] = first block of
  // scale S - 1 - k, counted after the pose blocks
  int n_pose;
  int g_ppb[DW_MAX_SCALES], g_nblk[DW_MAX_SCALES], g_off[DW_MAX_SCALES + 1];
  ScaleLayout L;
};

struct GatherArgs {
  const float* gup;
  const float* G;
  float* out;
  long n;
  int chunk, ppb, hs, ws, H, W;
  float coef, inv, mean_term, g0;
};

// Adjoint of the bilinear up-sampling for scale S_ >= 1 as a gather from the full-resolution
// gradient map: a coarse pixel receives from a window of 2 * 2^S_ rows and columns.  The kernel is
// bound by the number of load instructions (a partly filled 64-lane gather costs the address unit
// the same as a full one), so a lane takes four neighbouring columns as one 16-byte load (dword
// aligned) and walks down the rows; the 2^S_ / 2 lanes of a pixel are summed with DPP, fixed order.
template <int S_>
__device__ __forceinline__ void gather_upsampling_adjoint(const GatherArgs& a) {
  static_assert(S_ >= 1 && S_ <= 3, "window of at most 16 columns: four lanes of a quad");
  constexpr int K2 = 1 << S_;
  constexpr int LP_SHIFT = S_ - 1;  // lanes per pixel = 2 * 2^s / 4
  const int r = threadIdx.x & ((1 << LP_SHIFT) - 1);
  const float sy = (float)a.hs / (float)a.H, sx = (float)a.ws / (float)a.W;
#pragma unroll 1
  for (int pass = 0; pass < GD_PASSES; ++pass) {
    const int pix = pass * (SM_NT >> LP_SHIFT) + (threadIdx.x >> LP_SHIFT);
    const long i = (long)a.chunk * a.ppb + pix;
    const bool act = i < a.n;
    const long ic = act ? i : 0;
    // 32-bit index arithmetic (n < 2^30)
    const int y = (int)((unsigned)ic / (unsigned)a.ws), x = (int)((unsigned)ic - (unsigned)y * (unsigned)a.ws);
    // full-resolution rows / columns with a bilinear tap at (y, x)
    const int Ya = max(0, (y - 1) * K2 + K2 / 2), Yb = min(a.H - 1, (y + 1) * K2 + K2 / 2 - 1);
    const int Xa = max(0, (x - 1) * K2 + K2 / 2), Xb = min(a.W - 1, (x + 1) * K2 + K2 / 2 - 1);
    // lane r answers for the columns Xa + 4r .. Xa + 4r + 3 (up to Xb); its load starts there,
    // or further left where that keeps the four columns inside the row (W >= 4 at these scales'
    // full resolution: W is a multiple of 8); a column it holds but does not answer for gets
    // weight 0
    const int X0 = Xa + 4 * r, Xs = min(X0, a.W - 4);
    float wc[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int X = Xs + e;
      wc[e] = (X >= X0 && X <= Xb) ? upsample_weight(X, x, a.ws, sx) : 0.f;
    }
    float acc = 0.f;
    if (act && X0 <= Xb) {
      const float* col = a.gup + Xs;
#pragma unroll 4
      for (int Y = Ya; Y <= Yb; ++Y) {
        const QuadU q = *reinterpret_cast<const QuadU*>(col + (long)Y * a.W);
        const float rowsum = fmaf(wc[3], q.w, fmaf(wc[2], q.z, fmaf(wc[1], q.y, wc[0] * q.x)));
        acc = fmaf(upsample_weight(Y, y, a.hs, sy), rowsum, acc);
      }
    }
    if (S_ >= 2) acc = dpp_add<0xb1, 0xf>(acc);   // quad_perm:[1,0,3,2]: two lanes
    if (S_ >= 3) acc = dpp_add<0x4e, 0xf>(acc);   // quad_perm:[2,3,0,1]: four lanes
    if (act && r == 0) a.out[i] = a.coef * (a.G[i] * a.inv - a.mean_term) + a.g0 * acc;
  }
}

__device__ __forceinline__ void grad_disp_block(const BwdParams& p, int blk) {
  // Full resolution is a plain map (1024 pixels per block, four per thread).  A
  // coarser scale s gathers the adjoint of the bilinear up-sampling from the
  // full-resolution gradient map: a pixel receives from 2 * 2^s columns, one lane
  // per column, summed over the lanes with DPP (fixed order).
  int k = 0;
#pragma unroll
  for (int j = 1; j < DW_MAX_SCALES; ++j) k += (j < p.S && blk >= p.g_off[j]) ? 1 : 0;  // g_off ascends
  const int s = p.S - 1 - k;
  const int rblk = blk - sel(p.g_off, k);
  const int nblk_s = sel(p.g_nblk, s);
  const int b = rblk / nblk_s, chunk = rblk - b * nblk_s;
  const int ppb = sel(p.g_ppb, s);
  const int hs = p.H >> s, ws = p.W >> s;
  const long n = (long)hs * ws;
  const int H = p.H, W = p.W;
  const long plane = (long)H * W;
  const float mean = p.img_scal[2 * (s * p.B + b) + 0];
  const float sum_gd = p.img_scal[2 * (s * p.B + b) + 1];  // sum_i G_i * disp_i over the image
  const float go = p.grad_loss[0];
  const float inv = 1.0f / (mean + kMeanEps);
  float coef = go * (kWSmooth / (float)(1 << s)) / (float)p.S;
  float g0 = go / ((float)p.S * (float)p.B * (float)plane);
  if (p.g_len > 1) {  // the per-scale terms have upstream gradients of their own
    coef += p.grad_loss[1 + p.S + s];
    g0 = (go / (float)p.S + p.grad_loss[1 + s]) / ((float)p.B * (float)plane);
  }
  const float mean_term = sum_gd * inv * inv / (float)n;
  const float* G = p.sm_G + sel(p.L.pix_off, s) + (long)b * n;
  const float* gup = p.gdup + ((long)s * p.B + b) * plane;
  float* out = sel(p.grad_disp, s) + (long)b * n;
  if (s == 0) {
    if (p.vec4) {  // whole 16-byte groups: four pixels per thread, one load and one store each
      const long i = (long)chunk * ppb + 4 * threadIdx.x;
      if (i < n) {
        const float4 g = *reinterpret_cast<const float4*>(G + i), u = *reinterpret_cast<const float4*>(gup + i);
        float4 o;
        o.x = coef * (g.x * inv - mean_term) + g0 * u.x;
        o.y = coef * (g.y * inv - mean_term) + g0 * u.y;
        o.z = coef * (g.z * inv - mean_term) + g0 * u.z;
        o.w = coef * (g.w * inv - mean_term) + g0 * u.w;
        *reinterpret_cast<float4*>(out + i) = o;
      }
      return;
    }
    for (int k = threadIdx.x; k < ppb; k += SM_NT) {
      const long i = (long)chunk * ppb + k;
      if (i >= n) break;
      out[i] = coef * (G[i] * inv - mean_term) + g0 * gup[i];
    }
    return;
  }
  // the scale as a compile-time constant: the row loop of a lane is then a fixed number of loads,
  // all of them issued before the first is used
  const GatherArgs ga{gup, G, out, n, chunk, ppb, hs, ws, H, W, coef, inv, mean_term, g0};
  switch (s) {
    case 1: gather_upsampling_adjoint<1>(ga); break;
    case 2: gather_upsampling_adjoint<2>(ga); break;
    default: gather_upsampling_adjoint<3>(ga); break;
  }
}

// ------------------------------------------------- backward: d loss / d T
// One block per sample sums the per-(tile, scale) partials of the projection
// gradient in double, fixed order, and applies K^T.
constexpr int GP_GROUPS = SM_NT / 24;  // 10 groups of 24 threads (240 of the 256 in use)
__device__ __forceinline__ void grad_pose_block(const BwdParams& p, int b) {
  __shared__ double s_part[GP_GROUPS][24];
  __shared__ double s_dP[24];
  const int tid = threadIdx.x;
  const int nrow = p.tilesX * p.tilesY * p.S;  // one row of 24 partials per (tile, scale)
  // weight of the rows of scale s: 1 when only the total has an upstream gradient (it is
  // applied to the sum below), else the upstream gradient that reaches L_photo[s]
  double wsc[DW_MAX_SCALES];
#pragma unroll
  for (int s = 0; s < DW_MAX_SCALES; ++s)
    wsc[s] = (p.g_len > 1 && s < p.S) ? (double)p.grad_loss[0] / (double)p.S + (double)p.grad_loss[1 + s] : 1.0;
  if (tid < GP_GROUPS * 24) {
    const int g = tid / 24, k = tid - g * 24;
    double a = 0.0;
    const float* src = p.tile_dP + (long)b * nrow * 24 + k;
    if (p.g_len > 1) {
      int sc = g % p.S;  // row t belongs to scale t % S
      const int step = GP_GROUPS % p.S;
#pragma unroll 8
      for (int t = g; t < nrow; t += GP_GROUPS) {
        const double w = (sc == 0) ? wsc[0] : (sc == 1) ? wsc[1] : (sc == 2) ? wsc[2] : wsc[3];
        a += w * (double)src[(long)t * 24];
        sc += step;
        sc -= (sc >= p.S) ? p.S : 0;
      }
    } else {
#pragma unroll 8
      for (int t = g; t < nrow; t += GP_GROUPS) a += (double)src[(long)t * 24];
    }
    s_part[g][k] = a;
  }
  __syncthreads();
  if (tid < 24) {
    double a = 0.0;
    for (int g = 0; g < GP_GROUPS; ++g) a += s_part[g][tid];
    s_dP[tid] = a;
  }
  __syncthreads();
  if (tid < 32) {
    const int f = tid >> 4, k = (tid >> 2) & 3, j = tid & 3;
    const float* Kb = p.K + (long)b * 16;
    double a = 0.0;
    for (int i = 0; i < 3; ++i) a += (double)Kb[i * 4 + k] * s_dP[f * 12 + i * 4 + j];
    const double g0 = (p.g_len > 1) ? 1.0 / ((double)p.B * p.H * p.W)
                                    : (double)p.grad_loss[0] / ((double)p.S * p.B * p.H * p.W);
    p.grad_T[((long)b * DW_NUM_SRC + f) * 16 + k * 4 + j] = (float)(a * g0);
  }
}

// Both gradients in one launch: they are independent, and the few long pose
// blocks run beside the many short map blocks instead of after them.
__global__ __launch_bounds__(SM_NT) void backward_kernel(const BwdParams p) {
  if ((int)blockIdx.x < p.n_pose)
    grad_pose_block(p, blockIdx.x);
  else
    grad_disp_block(p, (int)blockIdx.x - p.n_pose);
}

// ------------------------------------------------- backward: d loss / d (warped sources)
// The adjoint of the bilinear warp with respect to the image it samples is a
// scatter: target pixel t adds gx(t) * w_tap to its four taps.  A block takes a
// CS_TH x CS_TW tile of target pixels and one source frame and accumulates their
// contributions, for every scale, into a CS_PH x CS_PW patch in LDS.  The patch
// is anchored where the tile's centre pixel samples the frame (scale 0), so for
// ordinary motion every tap of the tile falls inside it; the sums are 64-bit
// fixed point (2^-32 units), which makes them independent of the order in which
// the lanes arrive.  The rows of the patch that received taps then go to a slot of
// its own in the workspace with plain stores, and grad_color_kernel adds up, per
// source pixel, the slots whose touched rectangles cover it, in tile order.  A tap outside the patch (a pose that
// tears the tile's samples apart) is added to a dense fixed-point image in
// global memory with an integer atomic instead -- the rare path.
constexpr int CS_TH = 16, CS_TW = 32, CS_NT = 256;
constexpr int CS_PH = 32, CS_PW = 48;
constexpr int CS_MY = (CS_PH - CS_TH - 1) / 2, CS_MX = (CS_PW - CS_TW - 1) / 2;  // margin around the tile's own taps
constexpr int CS_OFF_MAX = 16;  // largest displacement of a patch from its tile; bounds the search in grad_color_kernel

struct ColorParams {
  int B, H, W, S;
  int tilesX, tilesY;
  const float* disp[DW_MAX_SCALES];
  const float* K;
  const float* invK;
  const float* T;
  const float* gxout;         // [S][B][2][3][H][W]
  const unsigned char* gany;  // [S][B][H][W]
  unsigned long long* gcol;   // [2][B][3][H][W]   fixed-point sums of the taps outside their patch
  float* slots;               // [B][tiles][2][3][CS_PH][CS_PW]
  int* anchors;               // [B][tiles][2][4]  (row, column) of patch element (0, 0) in the source image, and the
                              //                   rectangle of the patch that was touched: rows y0 | y1 << 16,
                              //                   columns x0 | x1 << 16 (inclusive; y1 < y0 when nothing was)
  unsigned* spill_flag;       // set when any tap went to gcol
};

// v in 2^-32 units as a two's-complement 64-bit integer, for |v| < 2^31: the integer part and the
// rest (exact: v - trunc(v)) are converted separately, both towards zero, the rest in steps of 2^-31,
// so every contribution, whatever its sign, is off by less than 2^-31.  Eight instructions; the
// compiler's own float -> int64 conversion takes fourteen, twelve times per item.
__device__ __forceinline__ unsigned long long to_fixed(float v) {
  const int hi = (int)v;
  const int lo = (int)((v - (float)hi) * 2147483648.0f);  // same sign as v, |lo| < 2^31
  const int carry = lo < 0 ? -1 : 0;  // the upper word of 2 * lo as a 64-bit integer
  return ((unsigned long long)(unsigned)(hi + carry) << 32) | ((unsigned)lo << 1);
}

__global__ __launch_bounds__(CS_NT, 3) void color_scatter_kernel(const ColorParams p) {
  // one block per (sample, tile, source frame): a 37 KB patch, three blocks per CU
  __shared__ unsigned long long s_patch[3][CS_PH * CS_PW];
  __shared__ __attribute__((aligned(16))) float s_P[12];
  __shared__ float s_invK[9];
  __shared__ int s_anchor[2];
  __shared__ unsigned short s_queue[CS_TH * CS_TW * DW_MAX_SCALES];  // (pixel of the tile << 2) | scale
  __shared__ int s_count;
  __shared__ int s_rect[4];  // rows and columns of the patch that received a tap: min, max, min, max
  const int tid = threadIdx.x;
  const int ntile = p.tilesX * p.tilesY;
  const int f = blockIdx.x & 1;
  const int bt = blockIdx.x >> 1;  // (sample, tile): the slot index
  const int b = bt / ntile, tile = bt - b * ntile;
  const int tY = tile / p.tilesX, tX = tile - tY * p.tilesX;
  const int ty0 = tY * CS_TH, tx0 = tX * CS_TW;
  const int H = p.H, W = p.W;
  const long plane = (long)H * W;
  if (tid < 12) {  // the projection matrix, with the arithmetic of the forward pass
    const int i = tid >> 2, j = tid & 3;
    const float* Kb = p.K + (long)b * 16;
    const float* Tb = p.T + ((long)b * DW_NUM_SRC + f) * 16;
    s_P[tid] = ((Kb[i * 4 + 0] * Tb[0 * 4 + j] + Kb[i * 4 + 1] * Tb[1 * 4 + j]) + Kb[i * 4 + 2] * Tb[2 * 4 + j]) +
               Kb[i * 4 + 3] * Tb[3 * 4 + j];
  } else if (tid >= 16 && tid < 25) {
    const int k = tid - 16;
    s_invK[k] = p.invK[(long)b * 16 + (k / 3) * 4 + (k % 3)];
  }
  if (tid == 0) {
    s_count = 0;
    s_rect[0] = CS_PH, s_rect[1] = -1, s_rect[2] = CS_PW, s_rect[3] = -1;
  }
  for (int i = tid; i < 3 * CS_PH * CS_PW; i += CS_NT) (&s_patch[0][0])[i] = 0ull;
  // Only pixels that carry a gradient into this frame do any work (elsewhere the adjoint is an
  // exact zero; the tile kernel left one byte per pixel and scale saying so): they are queued first,
  // for all scales at once, so that the heavy part runs on full waves.  The order of the queue varies from run to
  // run; the fixed-point sums do not depend on it.
  const int lane = tid & (kWave - 1);
  {
    constexpr int NK = CS_TH * CS_TW / CS_NT;
    bool live[DW_MAX_SCALES][NK];
#pragma unroll
    for (int s = 0; s < DW_MAX_SCALES; ++s) {
      const int sc = s < p.S ? s : 0;
      const unsigned char* gs = p.gany + ((long)sc * p.B + b) * plane;
#pragma unroll
      for (int it = 0; it < NK; ++it) {
        const int k = tid + it * CS_NT;
        const int y = ty0 + k / CS_TW, x = tx0 + k % CS_TW;
        const bool in_img = y < H && x < W;
        live[s][it] = s < p.S && in_img && ((gs[in_img ? (long)y * W + x : 0] >> f) & 1);
      }
    }
    __syncthreads();  // s_count, the patch and the matrices are set
#pragma unroll
    for (int s = 0; s < DW_MAX_SCALES; ++s)
#pragma unroll
      for (int it = 0; it < NK; ++it) {
        const unsigned long long m = __ballot(live[s][it]);
        int base = 0;
        if (lane == 0 && m) base = atomicAdd(&s_count, __popcll(m));
        base = __shfl(base, 0, kWave);
        if (live[s][it])
          s_queue[base + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)(((tid + it * CS_NT) << 2) | s);
      }
  }
  DivConsts dc;
  dc.yw = div_rcp((float)(W - 1));
  dc.yh = div_rcp((float)(H - 1));
  // where target pixel (y, x) samples the frame at scale s: the forward arithmetic, bit for bit
  auto sample_at = [&](int s, int y, int x, SampleAddr& a) __attribute__((always_inline)) {
    const int hs = H >> s, ws = W >> s;
    const float du = upsampled_disp(p.disp[s] + (long)b * hs * ws, hs, ws, s, y, x, (float)hs / (float)H,
                                    (float)ws / (float)W);
    const float dsp = kMinDisp + kDispRange * du;
    const float depth = div_by(1.0f, dsp, div_rcp(dsp));
    const float fx = (float)x, fy = (float)y;
    const float X = depth * ((s_invK[0] * fx + s_invK[1] * fy) + s_invK[2]);
    const float Y = depth * ((s_invK[3] * fx + s_invK[4] * fy) + s_invK[5]);
    const float Z = depth * ((s_invK[6] * fx + s_invK[7] * fy) + s_invK[8]);
    sample_addr<false>(H, W, dc, s_P, X, Y, Z, (unsigned)f, a);
  };
  if (tid == 0) {
    // the patch is anchored where the tile's centre pixel samples the frame at scale 0
    const int cy = min(ty0 + CS_TH / 2, H - 1), cx = min(tx0 + CS_TW / 2, W - 1);
    SampleAddr a;
    sample_at(0, cy, cx, a);
    const int dy = min(max(a.y0 - cy, -CS_OFF_MAX), CS_OFF_MAX);
    const int dx = min(max(a.x0 - cx, -CS_OFF_MAX), CS_OFF_MAX);
    s_anchor[0] = ty0 + dy - CS_MY;
    s_anchor[1] = tx0 + dx - CS_MX;
  }
  __syncthreads();
  const int oy = s_anchor[0], ox = s_anchor[1];
  const int n_items = s_count;
  bool spilled = false;
  int y_lo = CS_PH, y_hi = -1, x_lo = CS_PW, x_hi = -1;
  // CS_NI items per thread and trip, in two phases: every global load of the trip (the adjoint values and
  // the four disparity taps of each item; the taps of scale 0 are the pixel itself with weights 1 and 0,
  // the same value bit for bit) is issued before the first is used.
#ifndef CS_NI
#define CS_NI 2
#endif
#pragma unroll 1
  for (int i = tid; i < n_items; i += CS_NI * CS_NT) {
    bool on[CS_NI];
    int ys[CS_NI], xs[CS_NI];
    float gv[CS_NI][3], v4[CS_NI][4], lx[CS_NI][2], ly[CS_NI][2];
#pragma unroll
    for (int u = 0; u < CS_NI; ++u) {
      on[u] = i + u * CS_NT < n_items;
      const int item = s_queue[on[u] ? i + u * CS_NT : i];
      const int k = item >> 2, sc = item & 3;
      const int y = ty0 + k / CS_TW, x = tx0 + k % CS_TW;
      ys[u] = y, xs[u] = x;
      const float* g = p.gxout + ((((long)sc * p.B + b) * 2 + f) * 3) * plane + (long)y * W + x;
      gv[u][0] = g[0], gv[u][1] = g[plane], gv[u][2] = g[2 * plane];
      const int hs = H >> sc, ws = W >> sc;
      const float* d = p.disp[sc] + (long)b * hs * ws;
      const Taps ty = upsample_taps(y, hs, (float)hs / (float)H), tx = upsample_taps(x, ws, (float)ws / (float)W);
      v4[u][0] = d[(long)ty.i0 * ws + tx.i0], v4[u][1] = d[(long)ty.i0 * ws + tx.i1];
      v4[u][2] = d[(long)ty.i1 * ws + tx.i0], v4[u][3] = d[(long)ty.i1 * ws + tx.i1];
      lx[u][0] = tx.l0, lx[u][1] = tx.l1, ly[u][0] = ty.l0, ly[u][1] = ty.l1;
    }
#pragma unroll
    for (int u = 0; u < CS_NI; ++u) {
      const float top = lx[u][0] * v4[u][0] + lx[u][1] * v4[u][1];
      const float bot = lx[u][0] * v4[u][2] + lx[u][1] * v4[u][3];
      const float du = ly[u][0] * top + ly[u][1] * bot;
      const float dsp = kMinDisp + kDispRange * du;
      const float depth = div_by(1.0f, dsp, div_rcp(dsp));
      const float fx = (float)xs[u], fy = (float)ys[u];
      const float X = depth * ((s_invK[0] * fx + s_invK[1] * fy) + s_invK[2]);
      const float Y = depth * ((s_invK[3] * fx + s_invK[4] * fy) + s_invK[5]);
      const float Z = depth * ((s_invK[6] * fx + s_invK[7] * fy) + s_invK[8]);
      SampleAddr a;
      sample_addr<false>(H, W, dc, s_P, X, Y, Z, (unsigned)f, a);
      const float ex = 1.0f - a.tx, ey = 1.0f - a.ty;
      const float wt[4] = {ex * ey, a.tx * ey, ex * a.ty, a.tx * a.ty};
      const int yy[4] = {a.y0, a.y0, a.y1, a.y1}, xx[4] = {a.x0, a.x1, a.x0, a.x1};
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int py = yy[t] - oy, px = xx[t] - ox;
        const bool in = (unsigned)py < (unsigned)CS_PH && (unsigned)px < (unsigned)CS_PW;
        if (in && on[u]) {
          y_lo = min(y_lo, py), y_hi = max(y_hi, py);
          x_lo = min(x_lo, px), x_hi = max(x_hi, px);
        }
        unsigned long long q[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) q[c] = to_fixed(gv[u][c] * wt[t]);
        if (!on[u]) continue;
        if (in) {  // one branch per tap, not per channel
#pragma unroll
          for (int c = 0; c < 3; ++c)
            __hip_atomic_fetch_add(&s_patch[c][py * CS_PW + px], q[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
#pragma unroll
          for (int c = 0; c < 3; ++c)
            __hip_atomic_fetch_add(p.gcol + (((long)f * p.B + b) * 3 + c) * plane + (long)yy[t] * W + xx[t], q[c],
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          spilled = true;
        }
      }
    }
  }
  if (spilled) *p.spill_flag = 1u;  // every writer stores the same value
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    y_lo = min(y_lo, __shfl_xor(y_lo, off, kWave)), y_hi = max(y_hi, __shfl_xor(y_hi, off, kWave));
    x_lo = min(x_lo, __shfl_xor(x_lo, off, kWave)), x_hi = max(x_hi, __shfl_xor(x_hi, off, kWave));
  }
  if (lane == 0 && y_hi >= 0) {
    atomicMin(&s_rect[0], y_lo), atomicMax(&s_rect[1], y_hi);
    atomicMin(&s_rect[2], x_lo), atomicMax(&s_rect[3], x_hi);
  }
  __syncthreads();
  // Only the rows of the patch that received taps go to the slot, whole (a contiguous run per channel: full
  // cache lines); the reader gets the touched rectangle with the anchor and reads nothing outside it.
  const int ry0 = s_rect[0], ry1 = s_rect[1], rx0 = s_rect[2], rx1 = s_rect[3];
  float* slot = p.slots + ((long)bt * 2 + f) * (3 * CS_PH * CS_PW);
#pragma unroll
  for (int c = 0; c < 3; ++c)
    for (int i = ry0 * CS_PW + tid; i < (ry1 + 1) * CS_PW; i += CS_NT)
      slot[c * CS_PH * CS_PW + i] = (float)((double)(long long)s_patch[c][i] * (1.0 / 4294967296.0));
  if (tid < 4) {
    int* an = p.anchors + ((long)bt * 2 + f) * 4;
    an[tid] = tid < 2 ? s_anchor[tid] : (tid == 2 ? (ry0 | (ry1 << 16)) : (rx0 | (rx1 << 16)));
  }
}

// Fixed-order sum of the slots that cover a source pixel (+ the rare-path image), scaled by the
// upstream gradient and the 1 / (S * B * H * W) of the mean.  A block takes one CS_TH x CS_TW
// region of one source image (two pixels per thread, three channels each) and walks over the
// tiles whose patch can reach the region -- the same list for every thread, so the loop is uniform.
constexpr int GC_NU = 2;  // (4 and 8 measured the same)
__device__ __forceinline__ int floor_div(int a, int b) { return (a >= 0) ? a / b : -((-a + b - 1) / b); }
__global__ __launch_bounds__(CS_NT) void grad_color_kernel(const ColorParams p, const float* __restrict__ grad_loss,
                                                           float inv_count, float* __restrict__ out0,
                                                           float* __restrict__ out1) {
  const int ntile = p.tilesX * p.tilesY;
  const int f = blockIdx.x / (p.B * ntile);
  const int r = blockIdx.x - f * p.B * ntile;
  const int b = r / ntile, tile = r - b * ntile;
  const int rY = tile / p.tilesX, rX = tile - rY * p.tilesX;
  const int y_lo = rY * CS_TH, x_lo = rX * CS_TW;  // the region: rows y_lo .. y_lo + CS_TH - 1
  const long plane = (long)p.H * p.W;
  // a patch starts within CS_OFF_MAX of (its tile's origin - margin)
  const int tY0 = max(0, floor_div(y_lo - CS_OFF_MAX + CS_MY - CS_PH + 1 + CS_TH - 1, CS_TH));
  const int tY1 = min(p.tilesY - 1, floor_div(y_lo + CS_TH - 1 + CS_OFF_MAX + CS_MY, CS_TH));
  const int tX0 = max(0, floor_div(x_lo - CS_OFF_MAX + CS_MX - CS_PW + 1 + CS_TW - 1, CS_TW));
  const int tX1 = min(p.tilesX - 1, floor_div(x_lo + CS_TW - 1 + CS_OFF_MAX + CS_MX, CS_TW));
  const int ky = threadIdx.x / (CS_TW / 2), kx = 2 * (threadIdx.x % (CS_TW / 2));
  const int y = y_lo + ky, x = x_lo + kx;
  float acc[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
  const long tbase = (long)b * ntile;
  // The candidates' records are fetched by one lane each, all at once, and the ones whose touched
  // rectangle meets the region are then taken in tile order (one dependent load per candidate, one
  // after the other, was most of this kernel's time).
  const int lane = threadIdx.x & (kWave - 1);
  const int ncx = tX1 - tX0 + 1, ncand = (tY1 - tY0 + 1) * ncx;
  for (int base = 0; base < ncand; base += kWave) {
    const int i = base + lane;
    int4 rec = make_int4(0, 0, CS_PH | (-1 << 16), CS_PW | (-1 << 16));  // an empty rectangle
    long t = 0;
    if (i < ncand) {
      t = tbase + (long)(tY0 + i / ncx) * p.tilesX + (tX0 + i % ncx);
      rec = *reinterpret_cast<const int4*>(p.anchors + (t * 2 + f) * 4);
    }
    const bool meets = !((rec.x + (rec.z & 0xffff) > y_lo + CS_TH - 1) || (rec.x + (rec.z >> 16) < y_lo) ||
                         (rec.y + (rec.w & 0xffff) > x_lo + CS_TW - 1) || (rec.y + (rec.w >> 16) < x_lo));
    unsigned long long todo = __ballot(meets);
    while (todo) {
      // GC_NU tiles per trip: their loads are all in flight before the first is added
      float v[GC_NU][2][3];
#pragma unroll
      for (int u = 0; u < GC_NU; ++u)
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
          for (int c = 0; c < 3; ++c) v[u][e][c] = 0.f;
#pragma unroll
      for (int u = 0; u < GC_NU; ++u) {
        if (!todo) break;
        const int k = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const int oy = __builtin_amdgcn_readlane(rec.x, k), ox = __builtin_amdgcn_readlane(rec.y, k);
        const int ry = __builtin_amdgcn_readlane(rec.z, k), rx = __builtin_amdgcn_readlane(rec.w, k);
        const int ry0 = ry & 0xffff, ry1 = ry >> 16, rx0 = rx & 0xffff, rx1 = rx >> 16;
        const int ci = base + k;
        const long tk = tbase + (long)(tY0 + ci / ncx) * p.tilesX + (tX0 + ci % ncx);
        const int py = y - oy, px = x - ox;
        const float* sl = p.slots + (tk * 2 + f) * (3 * CS_PH * CS_PW) + py * CS_PW + px;
        if (py >= ry0 && py <= ry1) {
#pragma unroll
          for (int e = 0; e < 2; ++e)
            if (px + e >= rx0 && px + e <= rx1) {
#pragma unroll
              for (int c = 0; c < 3; ++c) v[u][e][c] = sl[c * CS_PH * CS_PW + e];
            }
        }
      }
      // (a cell outside a tile's rectangle adds an exact zero, which changes no sum)
#pragma unroll
      for (int u = 0; u < GC_NU; ++u)
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
          for (int c = 0; c < 3; ++c) acc[e][c] += v[u][e][c];
    }
  }
  if (y >= p.H) return;
  const bool spill = *p.spill_flag != 0u;
  const float g0 = grad_loss[0] * inv_count;
  float* out = (f ? out1 : out0) + (long)b * 3 * plane + (long)y * p.W + x;
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    if (x + e >= p.W) break;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float v = acc[e][c];
      if (spill)
        v += (float)((double)(long long)p.gcol[(((long)f * p.B + b) * 3 + c) * plane + (long)y * p.W + x + e] *
                     (1.0 / 4294967296.0));
      out[c * plane + e] = g0 * v;
    }
  }
}

// ------------------------------------------------------------- host side
struct WsLayout {
  size_t tile_loss, tile_dP, depth_up, gdup, img_scal, sm_img, sm_G, sm_part, total;
  // only with DW_FLAG_COLOR_GRAD: [color_zero_begin, color_zero_end) is cleared by every forward call
  size_t gxout, gany, slots, anchors, spill_flag, gcol, color_zero_begin, color_zero_end;
  int cX, cY;  // tiles of the colour scatter
};

struct TileCounts {
  int gX, gY;  // gradient kernel
  int fX, fY;  // forward-only kernel
};

static void make_layout(const dw_desc& d, ScaleLayout& L, WsLayout& w, TileCounts& tc) {
  tc.gX = (d.W + Geom<true>::CW - 1) / Geom<true>::CW;
  tc.gY = (d.H + Geom<true>::CH - 1) / Geom<true>::CH;
  tc.fX = (d.W + Geom<false>::CW - 1) / Geom<false>::CW;
  tc.fY = (d.H + Geom<false>::CH - 1) / Geom<false>::CH;
  const size_t tiles_g = (size_t)tc.gX * tc.gY;
  const size_t tiles_max = tiles_g > (size_t)tc.fX * tc.fY ? tiles_g : (size_t)tc.fX * tc.fY;
  L.blk_off[0] = 0;
  L.pix_off[0] = 0;
  for (int s = 0; s < DW_MAX_SCALES; ++s) {
    if (s < d.S) {
      L.Hs[s] = d.H >> s;
      L.Ws[s] = d.W >> s;
      L.nblk[s] = (L.Hs[s] * L.Ws[s] + SM_PX - 1) / SM_PX;
    } else {
      L.Hs[s] = L.Ws[s] = 0;
      L.nblk[s] = 0;
    }
    L.blk_off[s + 1] = L.blk_off[s] + d.B * L.nblk[s];
    L.pix_off[s + 1] = L.pix_off[s] + (long)d.B * L.Hs[s] * L.Ws[s];
  }
  const size_t A = 256;
  size_t o = 0;
  w.tile_loss = o;
  o = align_up(o + sizeof(float) * (size_t)d.B * tiles_max * DW_MAX_SCALES, A);
  w.tile_dP = o;
  o = align_up(o + sizeof(float) * (size_t)d.B * tiles_g * DW_MAX_SCALES * 24, A);
  w.sm_part = o;
  o = align_up(o + sizeof(float) * (size_t)L.blk_off[d.S] * 4, A);
  w.sm_G = o;
  o = align_up(o + sizeof(float) * (size_t)L.pix_off[d.S], A);
  w.img_scal = o;
  o = align_up(o + sizeof(float) * (size_t)DW_MAX_SCALES * d.B * 2, A);
  w.sm_img = o;
  o = align_up(o + sizeof(double) * (size_t)DW_MAX_SCALES * d.B * 2, A);
  w.depth_up = o;
  o = align_up(o + sizeof(float) * (size_t)d.S * d.B * d.H * d.W, A);
  w.gdup = o;  // last section (the diagnostic stamps overwrite its start)
  o = align_up(o + sizeof(float) * (size_t)d.S * d.B * d.H * d.W, A);
  w.cX = (d.W + CS_TW - 1) / CS_TW;
  w.cY = (d.H + CS_TH - 1) / CS_TH;
  w.gxout = w.gany = w.slots = w.anchors = w.spill_flag = w.gcol = w.color_zero_begin = w.color_zero_end = o;
  if (d.flags & DW_FLAG_COLOR_GRAD) {
    const size_t ctiles = (size_t)d.B * w.cX * w.cY;
    w.gxout = o;
    o = align_up(o + sizeof(float) * (size_t)d.S * d.B * 6 * d.H * d.W, A);
    w.gany = o;
    o = align_up(o + (size_t)d.S * d.B * d.H * d.W, A);
    w.slots = o;
    o = align_up(o + sizeof(float) * ctiles * 2 * 3 * CS_PH * CS_PW, A);
    w.anchors = o;
    o = align_up(o + sizeof(int) * ctiles * 8, A);
    w.color_zero_begin = w.spill_flag = o;
    o = align_up(o + sizeof(unsigned), A);
    w.gcol = o;
    o = align_up(o + sizeof(unsigned long long) * (size_t)2 * d.B * 3 * d.H * d.W, A);
    w.color_zero_end = o;
  }
  w.total = o;
}

static int check_desc(const dw_desc* d) {
  if (!d) return DW_ERR_NULL;
  if (d->B < 1 || d->S < 1 || d->S > DW_MAX_SCALES || d->F != DW_NUM_SRC) return DW_ERR_SHAPE;
  if (d->flags & ~(DW_FLAG_TILE_KERNEL_ONLY | DW_FLAG_NO_TAIL_SPLIT | DW_FLAG_COLOR_GRAD)) return DW_ERR_ARG;
  const int m = 1 << (d->S - 1);
  if (d->H % m || d->W % m) return DW_ERR_SHAPE;
  if ((d->H >> (d->S - 1)) < 2 || (d->W >> (d->S - 1)) < 2) return DW_ERR_SHAPE;
  // byte offsets inside a colour plane ((y * W + x) * 4) and H * W itself are 32-bit in the kernels
  if ((long)d->H * d->W > (1l << 28) || d->B > 65535) return DW_ERR_SHAPE;
  return DW_OK;
}

// Resident-block capacity of the current device for a tile kernel compiled for `per_cu` blocks per CU.
// (The compute-unit count of a device is looked up once and kept: relaxed atomics, since
// concurrent first calls all store the same value.)
static int tile_slots(int per_cu) {
  static std::atomic<int> cached[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
  int cus = cached[dev].load(std::memory_order_relaxed);
  if (cus == 0) {
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    cached[dev].store(cus, std::memory_order_relaxed);
  }
  return cus * per_cu;
}

// All tiles take the same time, so a launch runs in rounds of `slots` blocks
// and the last round leaves most of the chip idle when it is short.  If the
// tiles of that round, split into one block per scale, still fit into one
// round, launch them that way: the tail then lasts one scale (plus the
// scale-independent prologue) instead of S.  Every partial result is indexed
// by (tile, scale), so the outputs do not depend on the split.
static int full_blocks(int n_tiles, int S, int flags, int per_cu) {
  if ((flags & DW_FLAG_NO_TAIL_SPLIT) || S < 2) return n_tiles;
  const int slots = tile_slots(per_cu);
  if (slots <= 0) return n_tiles;
  const int r = n_tiles % slots;
  return ((long)r * S <= slots) ? n_tiles - r : n_tiles;
}

}  // namespace dw

using namespace dw;

extern "C" size_t dw_photometric_workspace_bytes(const dw_desc* desc) {
  if (check_desc(desc) != DW_OK) return 0;
  ScaleLayout L;
  WsLayout w;
  TileCounts tc;
  make_layout(*desc, L, w, tc);
  return w.total;
}

extern "C" int dw_photometric_fwd(const dw_desc* desc, const float* const* disp,
                                  const float* const* color, const float* const* color_pyr,
                                  const float* K, const float* inv_K, const float* T,
                                  const float* tie_noise, uint8_t* const* mask, float* loss_out,
                                  void* workspace, size_t workspace_bytes, int with_grad,
                                  void* stream) {
  int rc = check_desc(desc);
  if (rc != DW_OK) return rc;
  if (!disp || !color || !color_pyr || !K || !inv_K || !T || !loss_out || !workspace)
    return DW_ERR_NULL;
  PhotoParams p{};
  WsLayout w;
  TileCounts tc;
  make_layout(*desc, p.L, w, tc);
  if (workspace_bytes < w.total || ((uintptr_t)workspace & 255)) return DW_ERR_WORKSPACE;
  p.B = desc->B;
  p.H = desc->H;
  p.W = desc->W;
  p.S = desc->S;
  p.tilesX = with_grad ? tc.gX : tc.fX;
  p.tilesY = with_grad ? tc.gY : tc.fY;
  p.vec4 = 1;
  for (int s = 0; s < DW_MAX_SCALES; ++s) {
    p.rsy[s] = (float)(desc->H >> s) / (float)desc->H;
    p.rsx[s] = (float)(desc->W >> s) / (float)desc->W;
  }
  for (int s = 0; s < desc->S; ++s) {
    if (!disp[s] || !color_pyr[s]) return DW_ERR_NULL;
    p.disp[s] = disp[s];
    p.pyr[s] = color_pyr[s];
    p.mask[s] = mask ? mask[s] : nullptr;
    if (((desc->W >> s) & 3) || (((uintptr_t)disp[s] | (uintptr_t)color_pyr[s]) & 15)) p.vec4 = 0;
  }
  for (int i = 0; i < 3; ++i) {
    if (!color[i]) return DW_ERR_NULL;
    p.color[i] = color[i];
  }
  p.K = K;
  p.invK = inv_K;
  p.T = T;
  p.tie = tie_noise;
  char* ws = (char*)workspace;
  p.tile_loss = (float*)(ws + w.tile_loss);
  p.tile_dP = (float*)(ws + w.tile_dP);
  p.gdup = (float*)(ws + w.gdup);
  p.depth_up = (float*)(ws + w.depth_up);
  p.img_scal = (float*)(ws + w.img_scal);
  p.sm_img = (double*)(ws + w.sm_img);
  p.sm_G = (float*)(ws + w.sm_G);
  p.sm_part = (float*)(ws + w.sm_part);
  hipStream_t st = (hipStream_t)stream;
  const bool cgrad = with_grad && (desc->flags & DW_FLAG_COLOR_GRAD);
  ColorParams cp{};
  if (cgrad) {
    p.gxout = (float*)(ws + w.gxout);
    p.gany = (unsigned char*)(ws + w.gany);
    cp.B = p.B, cp.H = p.H, cp.W = p.W, cp.S = p.S;
    cp.tilesX = w.cX, cp.tilesY = w.cY;
    for (int s = 0; s < desc->S; ++s) cp.disp[s] = disp[s];
    cp.K = K, cp.invK = inv_K, cp.T = T;
    cp.gxout = p.gxout;
    cp.gany = p.gany;
    cp.gcol = (unsigned long long*)(ws + w.gcol);
    cp.slots = (float*)(ws + w.slots);
    cp.anchors = (int*)(ws + w.anchors);
    cp.spill_flag = (unsigned*)(ws + w.spill_flag);
    // the rare-path image and its flag start at zero
    if (hipMemsetAsync(ws + w.color_zero_begin, 0, w.color_zero_end - w.color_zero_begin, st) != hipSuccess)
      return DW_ERR_LAUNCH;
  }

  const int n_tiles = p.tilesX * p.tilesY * p.B;
  p.n_full = full_blocks(n_tiles, p.S, desc->flags, with_grad ? Geom<true>::PER_CU : Geom<false>::PER_CU);
  p.n_tile_blocks = p.n_full + (n_tiles - p.n_full) * p.S;
  // behind the tile blocks: one wave per image (scale, sample) for the images' smoothness sums
  const int waves_per_block = (with_grad ? Geom<true>::NT : Geom<false>::NT) / kWave;
  const dim3 tgrid(p.n_tile_blocks + (p.S * p.B + waves_per_block - 1) / waves_per_block);
  // depth maps: four pixels per thread when rows are whole 16-byte groups (every plane of the
  // workspace section is 256-byte aligned then)
  p.du_px = (p.W & 3) ? 1 : 4;
  p.du_nblk = (p.H * p.W / p.du_px + DU_NT - 1) / DU_NT;
  const unsigned map_blocks = (unsigned)(p.L.blk_off[p.S] + p.B * p.du_nblk);
  if (desc->flags & DW_FLAG_TILE_KERNEL_ONLY) {
    if (cgrad)
      photo_tile_color_kernel<<<tgrid, Geom<true>::NT, 0, st>>>(p);
    else if (with_grad)
      photo_tile_kernel<true><<<tgrid, Geom<true>::NT, 0, st>>>(p);
    else
      photo_tile_kernel<false><<<tgrid, Geom<false>::NT, 0, st>>>(p);
    return hipGetLastError() == hipSuccess ? DW_OK : DW_ERR_LAUNCH;
  }
  // smoothness sums and the depth maps first (one launch), then the tiles, which read the maps
  if (with_grad) {
    smooth_kernel<true><<<map_blocks, SM_NT, 0, st>>>(p);
    if (cgrad) {
      photo_tile_color_kernel<<<tgrid, Geom<true>::NT, 0, st>>>(p);
      color_scatter_kernel<<<2 * p.B * w.cX * w.cY, CS_NT, 0, st>>>(cp);
    } else {
      photo_tile_kernel<true><<<tgrid, Geom<true>::NT, 0, st>>>(p);
    }
  } else {
    smooth_kernel<false><<<map_blocks, SM_NT, 0, st>>>(p);
    photo_tile_kernel<false><<<tgrid, Geom<false>::NT, 0, st>>>(p);
  }
  loss_finalize_kernel<<<1, 1024, 0, st>>>(p, loss_out);
  return hipGetLastError() == hipSuccess ? DW_OK : DW_ERR_LAUNCH;
}

extern "C" int dw_photometric_bwd(const dw_desc* desc, const float* const* disp, const float* K,
                                  const float* grad_loss, int32_t grad_loss_len, const void* workspace,
                                  size_t workspace_bytes, float* const* grad_disp, float* grad_T,
                                  void* stream) {
  int rc = check_desc(desc);
  if (rc != DW_OK) return rc;
  if (!disp || !K || !grad_loss || !workspace || !grad_disp) return DW_ERR_NULL;
  if (grad_loss_len != 1 && grad_loss_len != 1 + 2 * desc->S) return DW_ERR_ARG;
  BwdParams p{};
  WsLayout w;
  TileCounts tc;
  make_layout(*desc, p.L, w, tc);
  if (workspace_bytes < w.total || ((uintptr_t)workspace & 255)) return DW_ERR_WORKSPACE;
  p.B = desc->B;
  p.H = desc->H;
  p.W = desc->W;
  p.S = desc->S;
  p.tilesX = tc.gX;
  p.tilesY = tc.gY;
  for (int s = 0; s < desc->S; ++s) {
    if (!disp[s] || !grad_disp[s]) return DW_ERR_NULL;
    p.disp[s] = disp[s];
    p.grad_disp[s] = grad_disp[s];
  }

  p.K = K;
  p.grad_loss = grad_loss;
  p.g_len = grad_loss_len;
  static_assert(SM_PX == 4 * SM_NT, "four pixels per thread in the full-resolution blocks");
  p.vec4 = (((long)desc->H * desc->W) % 4 == 0 && ((uintptr_t)grad_disp[0] & 15) == 0) ? 1 : 0;
  const char* ws = (const char*)workspace;
  p.tile_dP = (const float*)(ws + w.tile_dP);
  p.gdup = (const float*)(ws + w.gdup);
  p.img_scal = (const float*)(ws + w.img_scal);
  p.sm_G = (const float*)(ws + w.sm_G);
  p.sm_part = (const float*)(ws + w.sm_part);
  p.grad_T = grad_T;
  hipStream_t st = (hipStream_t)stream;
  p.g_off[0] = 0;
  for (int s = 0; s < DW_MAX_SCALES; ++s) {
    p.g_ppb[s] = (s == 0) ? SM_PX : GD_PASSES * (SM_NT / ((2 << s) / 4));  // coarse scales: 2 * 2^s / 4 lanes per pixel
    p.g_nblk[s] = (s < p.S) ? (p.L.Hs[s] * p.L.Ws[s] + p.g_ppb[s] - 1) / p.g_ppb[s] : 0;
  }
  for (int k = 0; k < DW_MAX_SCALES; ++k)
    p.g_off[k + 1] = p.g_off[k] + (k < p.S ? p.B * p.g_nblk[p.S - 1 - k] : 0);
  p.n_pose = grad_T ? p.B : 0;
  backward_kernel<<<p.n_pose + p.g_off[p.S], SM_NT, 0, st>>>(p);
  return hipGetLastError() == hipSuccess ? DW_OK : DW_ERR_LAUNCH;
}

extern "C" int dw_photometric_bwd_color(const dw_desc* desc, const float* grad_loss, const void* workspace,
                                        size_t workspace_bytes, float* const* grad_color, void* stream) {
  int rc = check_desc(desc);
  if (rc != DW_OK) return rc;
  if (!(desc->flags & DW_FLAG_COLOR_GRAD)) return DW_ERR_ARG;
  if (!grad_loss || !workspace || !grad_color || !grad_color[0] || !grad_color[1]) return DW_ERR_NULL;
  ScaleLayout L;
  WsLayout w;
  TileCounts tc;
  make_layout(*desc, L, w, tc);
  if (workspace_bytes < w.total || ((uintptr_t)workspace & 255)) return DW_ERR_WORKSPACE;
  ColorParams cp{};
  cp.B = desc->B, cp.H = desc->H, cp.W = desc->W, cp.S = desc->S;
  cp.tilesX = w.cX, cp.tilesY = w.cY;
  char* ws = (char*)workspace;  // read only
  cp.gcol = (unsigned long long*)(ws + w.gcol);
  cp.slots = (float*)(ws + w.slots);
  cp.anchors = (int*)(ws + w.anchors);
  cp.spill_flag = (unsigned*)(ws + w.spill_flag);
  const float inv_count = 1.0f / ((float)desc->S * (float)desc->B * (float)((long)desc->H * desc->W));
  const long nblk = 2l * desc->B * w.cX * w.cY;
  if (nblk > 0x7fffffffl) return DW_ERR_SHAPE;
  grad_color_kernel<<<(unsigned)nblk, CS_NT, 0, (hipStream_t)stream>>>(cp, grad_loss, inv_count, grad_color[0],
                                                                      grad_color[1]);
  return hipGetLastError() == hipSuccess ? DW_OK : DW_ERR_LAUNCH;
}
