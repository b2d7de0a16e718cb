// Train.compute_loss(target, pred) of the reference as one strip kernel plus a
// per-sample finalisation (/root/reference/train.py:272-309), forward only as
// there (the reference body runs under no_grad, train.py:274).
//
//   target -> bilinear resize to (H, W)                         train.py:276
//   Sobel x / y of the resized target, /8, replicate border      train.py:284
//   m_x = mean|dx|, m_y = mean|dy|;  smooth = m_x e^{m_x} + m_y e^{m_y}
//        (the "pred" edges are the target's, the weights are scalars:
//         train.py:288-299)
//   ssim_term = 1 - mean clamp((1 - SSIM_7x7)/2, 0, 1)           train.py:302
//   l1 = mean|target - pred|                                     train.py:305
//   loss = 0.75 ssim_term + 0.20 l1 + 0.15 smooth                train.py:307
//
// SSIM follows kornia's definition: 7x7 Gaussian window (sigma 1.5, normalised),
// reflect border, C1 = 0.01^2, C2 = 0.03^2, ssim = num / (den + 1e-12).  The
// window is applied separably: along x on each row as it arrives, along y over
// the last seven filtered rows, which a lane keeps in registers.  The sum of the
// two variances is filtered as one plane (t^2 + p^2): SSIM needs only their sum.
//
// The METRICS variant adds, from the same pass, what the reference's evaluation
// code derives from the same two tensors in separate passes:
//   rmse = sqrt(mean((resize(gt) - clamp(1/pred, lo, hi))^2))    train.py:311-343
//   min(pred), max(pred) for normalize_image                     train.py:248-253
// and normalize_image itself is the element-wise kernel at the end of the file.
#include <algorithm>
#include <cstdlib>
#include <mutex>

#include "dw_common.h"

// Measurement builds only (tools/build_variant.sh compute_loss NAME -DDW_TUNING -DDW_CL_KO=n): 1 leaves the
// row loads out of the walk (the first row is worked on again and again), 2 leaves the filters, the SSIM
// and the Sobel sums out (loads, row exchange and the absolute difference stay).  0 in the product.
#ifndef DW_CL_KO
#define DW_CL_KO 0
#endif

namespace dw {

// One wave owns a strip of CL_CW columns (two adjacent ones per lane) and walks down a chunk of
// rows.  Per step it takes one row of the resized target and of the prediction, exchanges the
// neighbouring columns through a row buffer in LDS, filters the row along x and pushes the
// result into a window of the last seven rows that lives in registers; the filter along y, the
// SSIM, the Sobel sums and the absolute difference all come out of registers.  Nothing but the
// two row buffers goes through LDS, every input value is fetched once per strip (plus three
// halo rows and columns per side), and the halo costs a row filter only.
constexpr int CL_CW = 128, CL_HALO = 3, CL_NT = 64;
constexpr int CL_ROWBUF = CL_CW + 2 * CL_HALO + 2;  // region columns x0-3 .. x0+130, padded
constexpr int CL_MIN_ROWS = 2, CL_MAX_ROWS = 64;

struct ClParams {
  int B, Ht, Wt, H, W, strips, chunks, rows;
  int antialias;  // resize with the area-weighted triangle filter instead of two taps per axis
  const void* target;  // float32, or float64 for the kernels instantiated with TT = double
  const float* pred;
  // pair call: sample 1 lives in buffers of its own (target2, pred2) instead of behind sample 0
  const void* target2;
  const float* pred2;
  float* part;  // [B][chunks * strips][4 or 8]: sum|dx|, sum|dy|, sum ssim-loss, sum|t - p|,
                //                                sum (t - clamp(1/p))^2, min p, max p, unused
  float* out;   // [B][4 or 8]
  float* norm;  // metrics + normalize_image in one call: [B][H * W] receives (pred - min) / (max - min); else null
  float g[4];   // Gaussian taps, centre last (the window is symmetric)
  float clamp_lo, clamp_hi;
};

// One axis of the anti-aliased bilinear resize (the triangle filter that
// F.interpolate(mode="bilinear", antialias=True) applies when it shrinks an
// image): output index `dst` reads the input indices [lo, lo + n) with weight
// max(0, 1 - |(j + lo - centre + 0.5) / max(scale, 1)|), normalised to sum 1.
struct AaSpan {
  int lo, n;
  float centre, inv, norm;
  __device__ __forceinline__ float weight(int j) const {
    const float t = fabsf(((float)(j + lo) - centre + 0.5f) * inv);
    return (t < 1.0f ? 1.0f - t : 0.f) * norm;
  }
};
__device__ __forceinline__ AaSpan aa_span(int dst, int in_size, float scale) {
  AaSpan a;
  const float support = (scale >= 1.0f) ? scale : 1.0f;
  a.centre = scale * ((float)dst + 0.5f);
  a.inv = (scale >= 1.0f) ? 1.0f / scale : 1.0f;
  a.lo = max((int)(a.centre - support + 0.5f), 0);
  a.n = min((int)(a.centre + support + 0.5f), in_size) - a.lo;
  a.n = min(max(a.n, 0), (int)ceilf(support) * 2 + 1);
  a.norm = 1.0f;
  float total = 0.f;
  for (int j = 0; j < a.n; ++j) total += a.weight(j);
  a.norm = (total != 0.f) ? 1.0f / total : 1.0f;
  return a;
}

// Minimum / maximum over a block (result in thread 0); `red` holds NT/64 floats.
template <int NT, bool IS_MAX>
__device__ __forceinline__ float block_extreme(float v, float* red) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float o = __shfl_down(v, off, kWave);
    v = IS_MAX ? fmaxf(v, o) : fminf(v, o);
  }
  __syncthreads();
  if ((threadIdx.x & (kWave - 1)) == 0) red[threadIdx.x / kWave] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < NT / kWave; ++w) v = IS_MAX ? fmaxf(v, red[w]) : fminf(v, red[w]);
  }
  return v;
}

// MODE 0: target already has the size of the prediction; 1: two-tap bilinear resize;
// 2: anti-aliased resize.  TT: element type of the target in memory -- the reference hands over
// float64 ground truth (NumPy pipeline) and computes in float32 after a cast; a double target is
// rounded to float32 value by value as it is read, which is that cast without a pass of its own.
template <int MODE, bool METRICS, typename TT>
__global__ __launch_bounds__(CL_NT) void compute_loss_strip_kernel(const ClParams p) {
  constexpr int PS = METRICS ? 8 : 4;
  __shared__ __attribute__((aligned(16))) float s_row[2][2][CL_ROWBUF];  // [slot][target | prediction][region column]
  const int lane = threadIdx.x, b = blockIdx.y;
  const int chunk = blockIdx.x / p.strips, strip = blockIdx.x - chunk * p.strips;
  const int H = p.H, W = p.W;
  const int x0 = strip * CL_CW, y0 = chunk * p.rows, y1 = min(y0 + p.rows, H);
  const bool second = p.target2 && b == 1;  // uniform
  const TT* tg = second ? static_cast<const TT*>(p.target2) : static_cast<const TT*>(p.target) + (long)b * p.Ht * p.Wt;
  const float* pr = second ? p.pred2 : p.pred + (long)b * H * W;
  const float sy = (float)p.Ht / (float)H, sx = (float)p.Wt / (float)W;

  // the lane's columns: its own two, and for the first six lanes one of the halo columns
  // (region column c holds image column x0 - 3 + c, reflected into the image)
  const int xa = x0 + 2 * lane;
  const bool has_halo = lane < 2 * CL_HALO;
  const int hc = lane < CL_HALO ? lane : CL_CW + lane;
  const bool live_a = xa < W, live_b = xa + 1 < W;
  const int cxa = reflect_index(min(xa, W + CL_HALO - 1), W);
  const int cxb = reflect_index(min(xa + 1, W + CL_HALO - 1), W);
  const int cxh = reflect_index(min(x0 - CL_HALO + hc, W + CL_HALO - 1), W);
  // The two own columns are neighbours in memory (in either order where the image edge
  // reflects them; columns beyond the reflection are never used): one 8-byte load each.
  const int own = min(cxa, cxb);
  const bool swapped = cxb < cxa;
  // A pair of horizontal resize taps is one 8-byte load as well; at the right edge, where
  // both taps are the last column, the pair ends there and its second value is taken twice.
  struct PairTaps {
    int i;     // first column of the pair
    bool end;  // both taps are the pair's second value
    float l0, l1;
  };
  PairTaps tx[3];
  AaSpan ax[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int col = c == 0 ? cxa : (c == 1 ? cxb : cxh);
    if (MODE == 1) {
      const Taps t = upsample_taps(col, p.Wt, sx);
      tx[c].end = t.i1 == t.i0;
      tx[c].i = tx[c].end ? t.i0 - 1 : t.i0;
      tx[c].l0 = t.l0;
      tx[c].l1 = t.l1;
    }
    if (MODE == 2) ax[c] = aa_span(col, p.Wt, sx);
  }

  struct T2 {
    TT x, y;
  };
  struct Row {
    T2 t[3][2];      // MODE 1: tap pairs of the three columns in the two source rows;
                     // otherwise t[0][0] = the own pair, t[2][0].x = the halo column
    float2 q;        // prediction, own pair
    float qh;        // prediction, halo column
    float l0, l1;    // weights of the two source rows (MODE 1)
  };
  auto load2 = [](const float* ptr) __attribute__((always_inline)) {
    float2 v;
    __builtin_memcpy(&v, ptr, sizeof(v));  // 4-byte aligned 8-byte load
    return v;
  };
  auto load2t = [](const TT* ptr) __attribute__((always_inline)) {
    T2 v;
    __builtin_memcpy(&v, ptr, sizeof(v));  // element-aligned load of two neighbours
    return v;
  };
  auto fetch = [&](int r) __attribute__((always_inline)) {
    Row w;
    const int ry = reflect_index(r, H);
    const float* prow = pr + (long)ry * W;
    w.q = load2(prow + own);
    w.qh = 0.f;
    if (has_halo) w.qh = prow[cxh];
    w.l0 = 1.f;
    w.l1 = 0.f;
    if (MODE == 0) {
      const TT* trow = tg + (long)ry * W;
      w.t[0][0] = load2t(trow + own);
      w.t[2][0].x = 0;
      if (has_halo) w.t[2][0].x = trow[cxh];
    } else if (MODE == 1) {
      const Taps ty = upsample_taps(ry, p.Ht, sy);
      const TT* r0 = tg + (long)ty.i0 * p.Wt;
      const TT* r1 = tg + (long)ty.i1 * p.Wt;
      w.l0 = ty.l0;
      w.l1 = ty.l1;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        w.t[c][0].x = w.t[c][0].y = w.t[c][1].x = w.t[c][1].y = 0;
        if (c < 2 || has_halo) {
          w.t[c][0] = load2t(r0 + tx[c].i);
          w.t[c][1] = load2t(r1 + tx[c].i);
        }
      }
    } else {
      // rows of the input first (horizontal pass), then the weighted rows (vertical pass)
      const AaSpan ay = aa_span(ry, p.Ht, sy);
      float v[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        float t = 0.f;
        if (c < 2 || has_halo) {
          for (int j = 0; j < ay.n; ++j) {
            const TT* row = tg + (long)(ay.lo + j) * p.Wt + ax[c].lo;
            float h = 0.f;
            for (int k = 0; k < ax[c].n; ++k) h = fmaf(ax[c].weight(k), (float)row[k], h);
            t = fmaf(ay.weight(j), h, t);
          }
        }
        v[c] = t;
      }
      w.t[0][0].x = (TT)v[0];
      w.t[0][0].y = (TT)v[1];
      w.t[2][0].x = (TT)v[2];
    }
    return w;
  };
  // resized target of column c (0, 1: own; 2: halo) and the prediction, in image order
  auto resized = [&](const Row& w, int c) __attribute__((always_inline)) {
    if (MODE == 1) {
      const float a0 = (float)w.t[c][0].x, a1 = (float)w.t[c][0].y, b0 = (float)w.t[c][1].x, b1 = (float)w.t[c][1].y;
      const float v00 = tx[c].end ? a1 : a0, v01 = a1;
      const float v10 = tx[c].end ? b1 : b0, v11 = b1;
      return w.l0 * (tx[c].l0 * v00 + tx[c].l1 * v01) + w.l1 * (tx[c].l0 * v10 + tx[c].l1 * v11);
    }
    if (c == 2) return (float)w.t[2][0].x;
    if (MODE == 2) return (float)(c ? w.t[0][0].y : w.t[0][0].x);
    return (float)((c == 1) != swapped ? w.t[0][0].y : w.t[0][0].x);
  };

  // The taps in vector registers: an instruction with a scalar-register operand holds the
  // vector pipe for longer than one without, and the taps go into every second instruction.
  float g0, g1, g2, g3;
  asm volatile("v_mov_b32 %0, %1" : "=v"(g0) : "s"(p.g[0]));
  asm volatile("v_mov_b32 %0, %1" : "=v"(g1) : "s"(p.g[1]));
  asm volatile("v_mov_b32 %0, %1" : "=v"(g2) : "s"(p.g[2]));
  asm volatile("v_mov_b32 %0, %1" : "=v"(g3) : "s"(p.g[3]));
  // symmetric seven-tap window over v[0..6]
  auto gauss = [&](float v0, float v1, float v2, float v3, float v4, float v5, float v6) __attribute__((always_inline)) {
    float a = g3 * v3;
    a = fmaf(g2, v2 + v4, a);
    a = fmaf(g1, v1 + v5, a);
    return fmaf(g0, v0 + v6, a);
  };

  // The last seven rows in registers, indexed by the step's phase (the loop is unrolled seven
  // times, so every index is a constant): row-filtered target, prediction, t^2 + p^2, t p, and
  // the row's Sobel partials (right - left, left + 2 centre + right).
  float win[7][6][2];
#pragma unroll
  for (int i = 0; i < 7; ++i)
#pragma unroll
    for (int k = 0; k < 6; ++k) win[i][k][0] = win[i][k][1] = 0.f;

  // sums per own column; a column outside the image is dropped once, at the end
  float sdx[2] = {0.f, 0.f}, sdy[2] = {0.f, 0.f}, sss[2] = {0.f, 0.f}, sl1[2] = {0.f, 0.f};
  float sse[2] = {0.f, 0.f}, pmin[2] = {INFINITY, INFINITY}, pmax[2] = {-INFINITY, -INFINITY};
  const int r_last = y1 + CL_HALO - 1;
  int r = y0 - CL_HALO;
  Row row = fetch(r);
  for (;;) {
#pragma unroll
    for (int ph = 0; ph < 7; ++ph) {
      if (r > r_last) goto done;
      float* st = s_row[r & 1][0];
      float* sq = s_row[r & 1][1];
      st[2 * lane + CL_HALO] = resized(row, 0);
      st[2 * lane + CL_HALO + 1] = resized(row, 1);
      sq[2 * lane + CL_HALO] = swapped ? row.q.y : row.q.x;
      sq[2 * lane + CL_HALO + 1] = swapped ? row.q.x : row.q.y;
      if (has_halo) {
        st[hc] = resized(row, 2);
        sq[hc] = row.qh;
      }
#if DW_CL_KO != 1
      if (r < r_last) row = fetch(r + 1);  // in flight while this row is worked on
#endif
      __syncthreads();
      float a[8], q[8];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float2 va = reinterpret_cast<const float2*>(st)[lane + k];
        const float2 vq = reinterpret_cast<const float2*>(sq)[lane + k];
        a[2 * k] = va.x;
        a[2 * k + 1] = va.y;
        q[2 * k] = vq.x;
        q[2 * k + 1] = vq.y;
      }
#if DW_CL_KO != 2
      float e[8], c[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        e[k] = fmaf(a[k], a[k], q[k] * q[k]);
        c[k] = a[k] * q[k];
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        win[ph][0][j] = gauss(a[j], a[j + 1], a[j + 2], a[j + 3], a[j + 4], a[j + 5], a[j + 6]);
        win[ph][1][j] = gauss(q[j], q[j + 1], q[j + 2], q[j + 3], q[j + 4], q[j + 5], q[j + 6]);
        win[ph][2][j] = gauss(e[j], e[j + 1], e[j + 2], e[j + 3], e[j + 4], e[j + 5], e[j + 6]);
        win[ph][3][j] = gauss(c[j], c[j + 1], c[j + 2], c[j + 3], c[j + 4], c[j + 5], c[j + 6]);
        // Sobel partials of this row, replicate border in x (the row buffer is reflect-mapped)
        const int x = xa + j;
        const float tc = a[CL_HALO + j];
        const float tl = (x == 0) ? tc : a[CL_HALO + j - 1];
        const float tr = (x >= W - 1) ? tc : a[CL_HALO + j + 1];
        win[ph][4][j] = tr - tl;
        win[ph][5][j] = (tl + 2.f * tc) + tr;
      }
#endif
      if (r >= y0 && r < y1) {  // absolute difference (and the metrics) of this row
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const float tv = a[CL_HALO + j], qv = q[CL_HALO + j];
          sl1[j] += fabsf(tv - qv);
          if (METRICS) {
            const float depth = fminf(fmaxf(1.0f / qv, p.clamp_lo), p.clamp_hi);
            const float d = tv - depth;
            sse[j] = fmaf(d, d, sse[j]);
            pmin[j] = fminf(pmin[j], qv);
            pmax[j] = fmaxf(pmax[j], qv);
          }
        }
      }
#if DW_CL_KO != 2
      const int ys = r - 1;  // Sobel of the row above this one, replicate border in y
      if (ys >= y0 && ys < y1) {
        const int UP = (ph + 5) % 7, MID = (ph + 6) % 7, DN = ph;  // constants once unrolled
        const int up = (ys == 0) ? MID : UP, dn = (ys == H - 1) ? MID : DN;
        if (up == UP && dn == DN) {  // the common case on its own path: no per-value selects
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            sdx[j] += fabsf(((win[UP][4][j] + 2.f * win[MID][4][j]) + win[DN][4][j]) * 0.125f);
            sdy[j] += fabsf((win[DN][5][j] - win[UP][5][j]) * 0.125f);
          }
        } else {
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const float d_up = up == UP ? win[UP][4][j] : win[MID][4][j], s_up = up == UP ? win[UP][5][j] : win[MID][5][j];
            const float d_dn = dn == DN ? win[DN][4][j] : win[MID][4][j], s_dn = dn == DN ? win[DN][5][j] : win[MID][5][j];
            sdx[j] += fabsf(((d_up + 2.f * win[MID][4][j]) + d_dn) * 0.125f);
            sdy[j] += fabsf((s_dn - s_up) * 0.125f);
          }
        }
      }
      if (r - CL_HALO >= y0) {  // seven rows are in the window: SSIM of its middle row
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          float m[4];
#pragma unroll
          for (int k = 0; k < 4; ++k)
            m[k] = gauss(win[(ph + 1) % 7][k][j], win[(ph + 2) % 7][k][j], win[(ph + 3) % 7][k][j],
                         win[(ph + 4) % 7][k][j], win[(ph + 5) % 7][k][j], win[(ph + 6) % 7][k][j], win[ph][k][j]);
          const float mm = m[0] * m[1], msq = fmaf(m[0], m[0], m[1] * m[1]);
          const float s_sum = m[2] - msq, s12 = m[3] - mm;
          const float num = fmaf(2.f, mm, 1e-4f) * fmaf(2.f, s12, 9e-4f);
          const float den = (msq + 1e-4f) * (s_sum + 9e-4f);
          const float ssim = num * __builtin_amdgcn_rcpf(den + 1e-12f);
          sss[j] += fminf(fmaxf((1.f - ssim) * 0.5f, 0.f), 1.f);
        }
      }
#endif
      ++r;
    }
  }
done:
  float* out = p.part + ((long)b * p.strips * p.chunks + blockIdx.x) * PS;
  auto both = [&](const float* v) __attribute__((always_inline)) {
    return wave_sum((live_a ? v[0] : 0.f) + (live_b ? v[1] : 0.f));
  };
  const float t_dx = both(sdx), t_dy = both(sdy), t_ss = both(sss), t_l1 = both(sl1);
  float t_se = 0.f, lo = INFINITY, hi = -INFINITY;
  if (METRICS) {
    t_se = both(sse);
    lo = fminf(live_a ? pmin[0] : INFINITY, live_b ? pmin[1] : INFINITY);
    hi = fmaxf(live_a ? pmax[0] : -INFINITY, live_b ? pmax[1] : -INFINITY);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      lo = fminf(lo, __shfl_xor(lo, off, kWave));
      hi = fmaxf(hi, __shfl_xor(hi, off, kWave));
    }
  }
  if (lane == kSumLane) {
    out[0] = t_dx;
    out[1] = t_dy;
    out[2] = t_ss;
    out[3] = t_l1;
    if (METRICS) {
      out[4] = t_se;
      out[5] = lo;
      out[6] = hi;
    }
  }
}

// One block per sample adds the partial sums in double, in a fixed order.  The partials come
// from other compute units' caches: every thread has its few loads in flight at once (16-byte
// loads, unrolled), since the kernel is nothing but their latency.
// PAIR: one block of 512 threads finishes both samples of a pair call, each half exactly as a
// block of its own would, and then adds the two losses (out[8]).
template <bool METRICS, bool PAIR>
__device__ __forceinline__ void compute_loss_finalize_block(const ClParams& p, int sample) {
  constexpr int PS = METRICS ? 8 : 4, NS = METRICS ? 5 : 4;  // partial stride, number of sums
  __shared__ double red_[PAIR ? 2 : 1][NS][256 / kWave];
  __shared__ float ext_[PAIR ? 2 : 1][2][256 / kWave];
  __shared__ float pair_loss[2];
  const int half = PAIR ? threadIdx.x >> 8 : 0;
  double (*red)[256 / kWave] = red_[half];
  float (*ext)[256 / kWave] = ext_[half];
  const int b = PAIR ? half : sample, tid = threadIdx.x & 255, lane = tid & (kWave - 1), wid = tid / kWave;
  const int nt = p.strips * p.chunks;
  double a[NS];
  for (int k = 0; k < NS; ++k) a[k] = 0.0;
  float lo = INFINITY, hi = -INFINITY;
#pragma unroll 4
  for (int t = tid; t < nt; t += 256) {
    const float4* src = reinterpret_cast<const float4*>(p.part + ((long)b * nt + t) * PS);
    const float4 u = src[0];
    a[0] += (double)u.x;
    a[1] += (double)u.y;
    a[2] += (double)u.z;
    a[3] += (double)u.w;
    if (METRICS) {
      const float4 v = src[1];
      a[NS - 1] += (double)v.x;
      lo = fminf(lo, v.y);
      hi = fmaxf(hi, v.z);
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    for (int k = 0; k < NS; ++k) a[k] += __shfl_xor(a[k], off, kWave);
    lo = fminf(lo, __shfl_xor(lo, off, kWave));
    hi = fmaxf(hi, __shfl_xor(hi, off, kWave));
  }
  if (lane == 0) {
    for (int k = 0; k < NS; ++k) red[k][wid] = a[k];
    ext[0][wid] = lo;
    ext[1][wid] = hi;
  }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < 256 / kWave; ++w) {
      for (int k = 0; k < NS; ++k) a[k] += red[k][w];
      lo = fminf(lo, ext[0][w]);
      hi = fmaxf(hi, ext[1][w]);
    }
    const double n = (double)p.H * p.W;
    const float mx = (float)(a[0] / n), my = (float)(a[1] / n);
    const float smooth = mx * expf(mx) + my * expf(my);
    const float ssim_term = 1.f - (float)(a[2] / n);
    const float l1 = (float)(a[3] / n);
    float* o = p.out + (long)b * PS;
    o[0] = 0.75f * ssim_term + 0.20f * l1 + 0.15f * smooth;
    o[1] = ssim_term;
    o[2] = l1;
    o[3] = smooth;
    if (METRICS) {
      o[4] = sqrtf((float)(a[NS - 1] / n));
      o[5] = lo;
      o[6] = hi;
      o[7] = 0.f;
    }
    if (PAIR) pair_loss[half] = o[0];
  }
  if (PAIR) {
    __syncthreads();
    if (threadIdx.x == 0) p.out[2 * PS] = pair_loss[0] + pair_loss[1];
  }
}
template <bool METRICS, bool PAIR>
__global__ __launch_bounds__(PAIR ? 512 : 256) void compute_loss_finalize_kernel(const ClParams p) {
  compute_loss_finalize_block<METRICS, PAIR>(p, (int)blockIdx.x);
}

// ---------------------------------------------------------------- normalize_image
constexpr int NI_NT = 256, NI_PER_BLOCK = 4096;

// per-block minimum / maximum of x[b]: part[(b * nblk + blk) * 2 + {0, 1}]
__global__ __launch_bounds__(NI_NT) void minmax_part_kernel(long n, int nblk, const float* __restrict__ x,
                                                            float* __restrict__ part) {
  __shared__ float red[NI_NT / kWave];
  const int b = blockIdx.y;
  const float* xb = x + (long)b * n;
  float lo = INFINITY, hi = -INFINITY;
  const long base = (long)blockIdx.x * NI_PER_BLOCK;
#pragma unroll 4
  for (int k = threadIdx.x; k < NI_PER_BLOCK; k += NI_NT) {
    const long i = base + k;
    if (i < n) {
      const float v = xb[i];
      lo = fminf(lo, v);
      hi = fmaxf(hi, v);
    }
  }
  lo = block_extreme<NI_NT, false>(lo, red);
  hi = block_extreme<NI_NT, true>(hi, red);
  if (threadIdx.x == 0) {
    part[((long)b * nblk + blockIdx.x) * 2 + 0] = lo;
    part[((long)b * nblk + blockIdx.x) * 2 + 1] = hi;
  }
}

// sqrt(mean((a - b)^2)) per sample (compute_depth_errors, train.py:338-343):
// per-block partial sums, then one block per sample adds them in double.
__global__ __launch_bounds__(NI_NT) void sse_part_kernel(long n, int nblk, const float* __restrict__ a,
                                                         const float* __restrict__ c,
                                                         float* __restrict__ part) {
  __shared__ float red[NI_NT / kWave];
  const int b = blockIdx.y;
  const float* ab = a + (long)b * n;
  const float* cb = c + (long)b * n;
  float acc = 0.f;
  const long base = (long)blockIdx.x * NI_PER_BLOCK;
#pragma unroll 4
  for (int k = threadIdx.x; k < NI_PER_BLOCK; k += NI_NT) {
    const long i = base + k;
    if (i < n) {
      const float e = ab[i] - cb[i];
      acc = fmaf(e, e, acc);
    }
  }
  const float t = block_sum<NI_NT>(acc, red);
  if (threadIdx.x == 0) part[(long)b * nblk + blockIdx.x] = t;
}

__global__ __launch_bounds__(NI_NT) void rmse_finalize_kernel(long n, int nblk, const float* __restrict__ part,
                                                              float* __restrict__ out) {
  __shared__ double red[NI_NT];
  const int b = blockIdx.x, tid = threadIdx.x;
  double a = 0.0;
  for (int k = tid; k < nblk; k += NI_NT) a += (double)part[(long)b * nblk + k];
  red[tid] = a;
  __syncthreads();
  for (int o = NI_NT / 2; o > 0; o >>= 1) {
    if (tid < o) red[tid] += red[tid + o];
    __syncthreads();
  }
  if (tid == 0) out[b] = sqrtf((float)(red[0] / (double)n));
}

// out = (x - mi) / d,  d = ma - mi, or 1e5 if ma == mi (train.py:250-253).  The
// reference takes ma and mi to the host as Python floats: d is their double
// difference rounded to float32, and a division of a device tensor by a host
// scalar is evaluated as a multiplication with the float32 reciprocal.
__global__ __launch_bounds__(NI_NT) void normalize_image_kernel(long n, int nblk, const float* __restrict__ x,
                                                                const float* __restrict__ part,
                                                                const float* __restrict__ minmax,
                                                                int minmax_stride, float* __restrict__ out) {
  __shared__ float red[NI_NT / kWave];
  __shared__ float s_mm[2];
  const int b = blockIdx.y;
  if (minmax) {
    if (threadIdx.x == 0) {
      s_mm[0] = minmax[(long)b * minmax_stride];
      s_mm[1] = minmax[(long)b * minmax_stride + 1];
    }
  } else {
    float lo = INFINITY, hi = -INFINITY;
    for (int k = threadIdx.x; k < nblk; k += NI_NT) {
      lo = fminf(lo, part[((long)b * nblk + k) * 2 + 0]);
      hi = fmaxf(hi, part[((long)b * nblk + k) * 2 + 1]);
    }
    lo = block_extreme<NI_NT, false>(lo, red);
    hi = block_extreme<NI_NT, true>(hi, red);
    if (threadIdx.x == 0) {
      s_mm[0] = lo;
      s_mm[1] = hi;
    }
  }
  __syncthreads();
  const float mi = s_mm[0], ma = s_mm[1];
  const float d = (ma != mi) ? (float)((double)ma - (double)mi) : 1e5f;
  const float inv = 1.0f / d;
  const float* xb = x + (long)b * n;
  float* ob = out + (long)b * n;
  const long base = (long)blockIdx.x * NI_PER_BLOCK;
#pragma unroll 4
  for (int k = threadIdx.x; k < NI_PER_BLOCK; k += NI_NT) {
    const long i = base + k;
    if (i < n) ob[i] = (xb[i] - mi) * inv;
  }
}

// Loss + metrics + normalize_image of the prediction in one call: the launch behind the strip kernel.
// Block (0, b) finishes sample b's sums as compute_loss_finalize_kernel does; block (1 + k, b) takes the
// prediction's extrema from the strip kernel's partial results itself (a few hundred floats) and writes
// chunk k of the normalised image, with the arithmetic of normalize_image_kernel.
__global__ __launch_bounds__(NI_NT) void compute_loss_finalize_norm_kernel(const ClParams p) {
  const int b = blockIdx.y;
  if (blockIdx.x == 0) {
    compute_loss_finalize_block<true, false>(p, b);
    return;
  }
  __shared__ float red[NI_NT / kWave];
  __shared__ float s_mm[2];
  const int nt = p.strips * p.chunks;
  float lo = INFINITY, hi = -INFINITY;
  for (int k = threadIdx.x; k < nt; k += NI_NT) {
    const float* q = p.part + ((long)b * nt + k) * 8;
    lo = fminf(lo, q[5]);
    hi = fmaxf(hi, q[6]);
  }
  lo = block_extreme<NI_NT, false>(lo, red);
  hi = block_extreme<NI_NT, true>(hi, red);
  if (threadIdx.x == 0) {
    s_mm[0] = lo;
    s_mm[1] = hi;
  }
  __syncthreads();
  const float mi = s_mm[0], ma = s_mm[1];
  const float d = (ma != mi) ? (float)((double)ma - (double)mi) : 1e5f;
  const float inv = 1.0f / d;
  const long n = (long)p.H * p.W;
  const float* xb = p.pred + (long)b * n;
  float* ob = p.norm + (long)b * n;
  const long base = (long)(blockIdx.x - 1) * NI_PER_BLOCK;
#pragma unroll 4
  for (int k = threadIdx.x; k < NI_PER_BLOCK; k += NI_NT) {
    const long i = base + k;
    if (i < n) ob[i] = (xb[i] - mi) * inv;
  }
}

}  // namespace dw

using namespace dw;

// Rows per chunk: as tall as leaves about `kWantWaves` waves for the launch (a strip re-filters six
// halo rows per chunk, so taller is cheaper as long as the chip stays full).
static int compute_loss_rows(int B, int H, int W) {
#ifdef DW_TUNING  // tools/cl_sweep.sh: the launch size from the environment
  const long kWantWaves = getenv("DW_CL_WAVES") ? atol(getenv("DW_CL_WAVES")) : 3072;
#else
  constexpr long kWantWaves = 3072;  // three per SIMD; measured flat between two and four
#endif
  const long strips = (W + CL_CW - 1) / CL_CW;
  const long want_chunks = std::max<long>(1, kWantWaves / (strips * B));
  const long rows = (H + want_chunks - 1) / want_chunks;
  return (int)std::min<long>(std::max<long>(rows, CL_MIN_ROWS), CL_MAX_ROWS);
}

extern "C" size_t dw_compute_loss_workspace_bytes(int32_t B, int32_t H, int32_t W) {
  if (B < 1 || H < 4 || W < 4) return 0;
  // the largest number of (strip, chunk) partials any choice of the chunk height can need
  const size_t parts = (size_t)((W + CL_CW - 1) / CL_CW) * ((H + CL_MIN_ROWS - 1) / CL_MIN_ROWS);
  return align_up(sizeof(float) * 8 * parts * B, 256);  // enough for the metrics variant too
}

// target2 / pred2 non-null: a pair call (B must be 2): sample 1 is read from them, the chunk
// height is that of a one-sample call, and out receives a ninth float, the sum of the two losses.
template <bool METRICS>
static int compute_loss_launch(int32_t B, int32_t Ht, int32_t Wt, int32_t H, int32_t W,
                               const void* target, const float* pred, float clamp_lo, float clamp_hi,
                               float* out, int32_t flags, void* workspace, size_t workspace_bytes, void* stream,
                               const void* target2 = nullptr, const float* pred2 = nullptr, float* norm = nullptr) {
  if (!target || !pred || !out || !workspace) return DW_ERR_NULL;
  const bool pair = target2 || pred2;
  if (pair && (!target2 || !pred2 || B != 2)) return DW_ERR_ARG;
  if (flags & ~(DW_CL_FLAG_ANTIALIAS | DW_CL_FLAG_TARGET_F64)) return DW_ERR_ARG;
  // reflect padding of 3 needs at least 4 pixels per side
  if (B < 1 || B > 65535 || H < 4 || W < 4 || Ht < 1 || Wt < 1) return DW_ERR_SHAPE;
  if ((long)H * W > (1l << 28) || (long)Ht * Wt > (1l << 28)) return DW_ERR_SHAPE;
  // the horizontal taps of the resize are fetched as pairs of neighbouring columns
  if ((Ht != H || Wt != W) && Wt < 2) return DW_ERR_SHAPE;
  ClParams p{};
  p.B = B;
  p.Ht = Ht;
  p.Wt = Wt;
  p.H = H;
  p.W = W;
  p.rows = compute_loss_rows(pair ? 1 : B, H, W);
  p.strips = (W + CL_CW - 1) / CL_CW;
  p.chunks = (H + p.rows - 1) / p.rows;
  // one set of partial sums per (strip, chunk) and sample; dw_compute_loss_workspace_bytes is
  // the bound of this over every chunk height
  if (workspace_bytes < sizeof(float) * 8 * (size_t)p.strips * p.chunks * B || ((uintptr_t)workspace & 255))
    return DW_ERR_WORKSPACE;
  p.antialias = (flags & DW_CL_FLAG_ANTIALIAS) ? 1 : 0;
  p.target = target;
  p.pred = pred;
  p.target2 = target2;
  p.pred2 = pred2;
  p.part = (float*)workspace;
  p.out = out;
  p.norm = norm;
  p.clamp_lo = clamp_lo;
  p.clamp_hi = clamp_hi;
  double g[7], sum = 0.0;
  for (int k = 0; k < 7; ++k) {
    const double x = k - 3;
    g[k] = exp(-(x * x) / (2.0 * 1.5 * 1.5));
    sum += g[k];
  }
  for (int k = 0; k < 4; ++k) p.g[k] = (float)(g[k] / sum);
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(p.strips * p.chunks, B);
  const int mode = (Ht == H && Wt == W) ? 0 : (p.antialias ? 2 : 1);
  if (flags & DW_CL_FLAG_TARGET_F64) {
    if (mode == 0) compute_loss_strip_kernel<0, METRICS, double><<<grid, CL_NT, 0, st>>>(p);
    if (mode == 1) compute_loss_strip_kernel<1, METRICS, double><<<grid, CL_NT, 0, st>>>(p);
    if (mode == 2) compute_loss_strip_kernel<2, METRICS, double><<<grid, CL_NT, 0, st>>>(p);
  } else {
    if (mode == 0) compute_loss_strip_kernel<0, METRICS, float><<<grid, CL_NT, 0, st>>>(p);
    if (mode == 1) compute_loss_strip_kernel<1, METRICS, float><<<grid, CL_NT, 0, st>>>(p);
    if (mode == 2) compute_loss_strip_kernel<2, METRICS, float><<<grid, CL_NT, 0, st>>>(p);
  }
  if (pair) {
    compute_loss_finalize_kernel<METRICS, true><<<1, 512, 0, st>>>(p);
  } else if (METRICS && norm) {
    const long nblk = ((long)H * W + NI_PER_BLOCK - 1) / NI_PER_BLOCK;
    compute_loss_finalize_norm_kernel<<<dim3((unsigned)(1 + nblk), B), NI_NT, 0, st>>>(p);
  } else {
    compute_loss_finalize_kernel<METRICS, false><<<B, 256, 0, st>>>(p);
  }
  return hipGetLastError() == hipSuccess ? DW_OK : DW_ERR_LAUNCH;
}

extern "C" int dw_compute_loss_fwd(int32_t B, int32_t Ht, int32_t Wt, int32_t H, int32_t W,
                                   const void* target, const float* pred, float* out, int32_t flags,
                                   void* workspace, size_t workspace_bytes, void* stream) {
  return compute_loss_launch<false>(B, Ht, Wt, H, W, target, pred, 0.f, 0.f, out, flags, workspace,
                                    workspace_bytes, stream);
}

extern "C" int dw_compute_loss_pair_fwd(int32_t Ht, int32_t Wt, int32_t H, int32_t W, const void* target_l,
                                        const float* pred_l, const void* target_r, const float* pred_r,
                                        float* out, int32_t flags, void* workspace, size_t workspace_bytes,
                                        void* stream) {
  if (!target_r || !pred_r) return DW_ERR_NULL;
  return compute_loss_launch<false>(2, Ht, Wt, H, W, target_l, pred_l, 0.f, 0.f, out, flags, workspace,
                                    workspace_bytes, stream, target_r, pred_r);
}

extern "C" int dw_compute_loss_metrics_fwd(int32_t B, int32_t Ht, int32_t Wt, int32_t H, int32_t W,
                                           const void* target, const float* pred, float clamp_min,
                                           float clamp_max, float* out, int32_t flags, void* workspace,
                                           size_t workspace_bytes, void* stream) {
  if (!(clamp_min <= clamp_max)) return DW_ERR_ARG;
  return compute_loss_launch<true>(B, Ht, Wt, H, W, target, pred, clamp_min, clamp_max, out, flags, workspace,
                                   workspace_bytes, stream);
}

extern "C" int dw_compute_loss_metrics_norm_fwd(int32_t B, int32_t Ht, int32_t Wt, int32_t H, int32_t W,
                                                const void* target, const float* pred, float clamp_min,
                                                float clamp_max, float* out, float* pred_normalized, int32_t flags,
                                                void* workspace, size_t workspace_bytes, void* stream) {
  if (!(clamp_min <= clamp_max)) return DW_ERR_ARG;
  if (!pred_normalized) return DW_ERR_NULL;
  return compute_loss_launch<true>(B, Ht, Wt, H, W, target, pred, clamp_min, clamp_max, out, flags, workspace,
                                   workspace_bytes, stream, nullptr, nullptr, pred_normalized);
}

// ---------------------------------------------------------------- allocation-free entry point
// The partial sums of dw_compute_loss_fwd (one set per strip and chunk) live in a fixed device array
// of this library instead of a caller workspace.  Every GPU the library runs on
// holds its own copy of the array at its own address until the process ends;
// `scratch` itself is only a host-side handle of the symbol (not a device
// address), so the address on the current device is looked up once per device
// and that pointer is what the kernels receive.
static __device__ __align__(256) unsigned char scratch[DW_STATIC_SCRATCH_BYTES];

namespace {
constexpr int kMaxDevices = 64;
std::mutex g_scratch_mutex;
void* g_scratch_ptr[kMaxDevices] = {nullptr};

// Device address of `scratch` on the current device (nullptr on failure).
void* static_scratch_ptr() {
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return nullptr;
  std::lock_guard<std::mutex> lock(g_scratch_mutex);
  if (!g_scratch_ptr[dev]) {
    void* ptr = nullptr;
    if (hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)) != hipSuccess) return nullptr;
    g_scratch_ptr[dev] = ptr;
  }
  return g_scratch_ptr[dev];
}
}  // namespace

extern "C" int dw_compute_loss_fwd_static(int32_t B, int32_t Ht, int32_t Wt, int32_t H, int32_t W,
                                          const void* target, const float* pred, float* out, int32_t flags,
                                          void* stream) {
  if (!target || !pred || !out) return DW_ERR_NULL;
  if (B < 1 || B > 65535 || H < 4 || W < 4 || Ht < 1 || Wt < 1) return DW_ERR_SHAPE;
  // what this call writes: one set of partial sums per (strip, chunk) of every sample
  const int rows = compute_loss_rows(B, H, W);
  const size_t need = sizeof(float) * 8 * (size_t)((W + CL_CW - 1) / CL_CW) * ((H + rows - 1) / rows) * B;
  if (need > DW_STATIC_SCRATCH_BYTES) return DW_ERR_TOO_LARGE;
  void* ptr = static_scratch_ptr();
  if (!ptr) return DW_ERR_LAUNCH;
  // every partial the finalisation reads is written by the strip kernel of the same call:
  // nothing in the array has to start at zero
  return compute_loss_launch<false>(B, Ht, Wt, H, W, target, pred, 0.f, 0.f, out, flags, ptr,
                                    DW_STATIC_SCRATCH_BYTES, stream);
}

extern "C" size_t dw_normalize_image_workspace_bytes(int32_t B, int64_t n) {
  if (B < 1 || n < 1) return 0;
  const size_t nblk = (size_t)((n + NI_PER_BLOCK - 1) / NI_PER_BLOCK);
  return align_up(sizeof(float) * 2 * nblk * B, 256);
}

extern "C" int dw_rmse(int32_t B, int64_t n, const float* a, const float* b, float* out,
                       void* workspace, size_t workspace_bytes, void* stream) {
  if (!a || !b || !out || !workspace) return DW_ERR_NULL;
  if (B < 1 || B > 65535 || n < 1 || n > ((int64_t)1 << 40)) return DW_ERR_SHAPE;
  const long nblk = (n + NI_PER_BLOCK - 1) / NI_PER_BLOCK;
  if (nblk > 0x7fffffffL) return DW_ERR_SHAPE;
  if (workspace_bytes < dw_normalize_image_workspace_bytes(B, n) || ((uintptr_t)workspace & 255))
    return DW_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  sse_part_kernel<<<dim3((unsigned)nblk, B), NI_NT, 0, st>>>(n, (int)nblk, a, b, (float*)workspace);
  rmse_finalize_kernel<<<B, NI_NT, 0, st>>>(n, (int)nblk, (const float*)workspace, out);
  return hipGetLastError() == hipSuccess ? DW_OK : DW_ERR_LAUNCH;
}

extern "C" int dw_normalize_image(int32_t B, int64_t n, const float* x, const float* minmax,
                                  int32_t minmax_stride, float* out, void* workspace,
                                  size_t workspace_bytes, void* stream) {
  if (!x || !out) return DW_ERR_NULL;
  if (B < 1 || B > 65535 || n < 1 || n > ((int64_t)1 << 40)) return DW_ERR_SHAPE;
  const long nblk = (n + NI_PER_BLOCK - 1) / NI_PER_BLOCK;
  if (nblk > 0x7fffffffL) return DW_ERR_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  if (minmax) {
    if (minmax_stride < 2) return DW_ERR_ARG;
  } else {
    if (!workspace) return DW_ERR_NULL;
    if (workspace_bytes < dw_normalize_image_workspace_bytes(B, n) || ((uintptr_t)workspace & 255))
      return DW_ERR_WORKSPACE;
    minmax_part_kernel<<<dim3((unsigned)nblk, B), NI_NT, 0, st>>>(n, (int)nblk, x, (float*)workspace);
  }
  normalize_image_kernel<<<dim3((unsigned)nblk, B), NI_NT, 0, st>>>(n, (int)nblk, x, (const float*)workspace,
                                                                    minmax, minmax_stride, out);
  return hipGetLastError() == hipSuccess ? DW_OK : DW_ERR_LAUNCH;
}
