// Train.compute_loss(target, pred) of the reference as one tiled kernel plus a
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
// window is applied separably (rows, then columns) from an LDS tile.
//
// The METRICS variant adds, from the same tile, what the reference's evaluation
// code derives from the same two tensors in separate passes:
//   rmse = sqrt(mean((resize(gt) - clamp(1/pred, lo, hi))^2))    train.py:311-343
//   min(pred), max(pred) for normalize_image                     train.py:248-253
// and normalize_image itself is the element-wise kernel at the end of the file.
#include <mutex>

#include "dw_common.h"

namespace dw {

constexpr int CL_TH = 16, CL_TW = 64, CL_NT = 256, CL_HALO = 3;
constexpr int CL_RW = CL_TW + 2 * CL_HALO, CL_RH = CL_TH + 2 * CL_HALO, CL_R = CL_RW * CL_RH;

struct ClParams {
  int B, Ht, Wt, H, W, tilesX, tilesY;
  int antialias;  // resize with the area-weighted triangle filter instead of two taps per axis
  const float* target;
  const float* pred;
  float* part;  // [B][tiles][4 or 8]: sum|dx|, sum|dy|, sum ssim-loss, sum|t - p|,
                //                       sum (t - clamp(1/p))^2, min p, max p, unused
  float* out;   // [B][4 or 8]
  float g[7];   // Gaussian taps
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

template <bool METRICS>
__global__ __launch_bounds__(CL_NT) void compute_loss_tile_kernel(const ClParams p) {
  constexpr int PS = METRICS ? 8 : 4;
  __shared__ float s_t[CL_R];              // resized target, reflect-mapped halo
  __shared__ float s_p[CL_R];              // prediction
  __shared__ float s_h[5][CL_RH * CL_TW];  // row-filtered t, p, tt, pp, tp
  __shared__ float s_red[CL_NT / kWave];
  const int tid = threadIdx.x, b = blockIdx.y;
  const int tY = blockIdx.x / p.tilesX, tX = blockIdx.x - tY * p.tilesX;
  const int y00 = tY * CL_TH - CL_HALO, x00 = tX * CL_TW - CL_HALO;
  const int H = p.H, W = p.W;
  const float* tg = p.target + (long)b * p.Ht * p.Wt;
  const float* pr = p.pred + (long)b * H * W;
  const bool same = (p.Ht == H && p.Wt == W);
  const float sy = (float)p.Ht / (float)H, sx = (float)p.Wt / (float)W;

  for (int i = tid; i < CL_R; i += CL_NT) {
    const int ry = i / CL_RW, rx = i - ry * CL_RW;
    const int y = reflect_index(y00 + ry, H), x = reflect_index(x00 + rx, W);
    float t;
    if (same) {
      t = tg[(long)y * W + x];
    } else if (p.antialias) {
      // rows of the input first (horizontal pass), then the weighted rows (vertical pass)
      const AaSpan ay = aa_span(y, p.Ht, sy), ax = aa_span(x, p.Wt, sx);
      t = 0.f;
      for (int j = 0; j < ay.n; ++j) {
        const float* row = tg + (long)(ay.lo + j) * p.Wt + ax.lo;
        float h = 0.f;
        for (int k = 0; k < ax.n; ++k) h = fmaf(ax.weight(k), row[k], h);
        t = fmaf(ay.weight(j), h, t);
      }
    } else {
      const Taps ty = upsample_taps(y, p.Ht, sy), tx = upsample_taps(x, p.Wt, sx);
      const float v00 = tg[(long)ty.i0 * p.Wt + tx.i0], v01 = tg[(long)ty.i0 * p.Wt + tx.i1];
      const float v10 = tg[(long)ty.i1 * p.Wt + tx.i0], v11 = tg[(long)ty.i1 * p.Wt + tx.i1];
      t = ty.l0 * (tx.l0 * v00 + tx.l1 * v01) + ty.l1 * (tx.l0 * v10 + tx.l1 * v11);
    }
    s_t[i] = t;
    s_p[i] = pr[(long)y * W + x];
  }
  __syncthreads();

  // rows: 7-tap Gaussian along x for every region row, core columns only
  for (int i = tid; i < CL_RH * CL_TW; i += CL_NT) {
    const int ry = i / CL_TW, cx = i - ry * CL_TW;
    const int base = ry * CL_RW + cx;  // region column cx .. cx+6 is centred on core column cx
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f;
#pragma unroll
    for (int k = 0; k < 7; ++k) {
      const float t = s_t[base + k], q = s_p[base + k], w = p.g[k];
      a0 = fmaf(w, t, a0);
      a1 = fmaf(w, q, a1);
      a2 = fmaf(w, t * t, a2);
      a3 = fmaf(w, q * q, a3);
      a4 = fmaf(w, t * q, a4);
    }
    s_h[0][i] = a0;
    s_h[1][i] = a1;
    s_h[2][i] = a2;
    s_h[3][i] = a3;
    s_h[4][i] = a4;
  }
  __syncthreads();

  float sdx = 0.f, sdy = 0.f, sss = 0.f, sl1 = 0.f;
  float sse = 0.f, pmin = INFINITY, pmax = -INFINITY;
  for (int i = tid; i < CL_TH * CL_TW; i += CL_NT) {
    const int cy = i / CL_TW, cx = i - cy * CL_TW;
    const int y = y00 + CL_HALO + cy, x = x00 + CL_HALO + cx;
    if (y >= H || x >= W) continue;
    // columns: 7 taps along y
    float m2 = 0.f, m1 = 0.f, e22 = 0.f, e11 = 0.f, e12 = 0.f;
#pragma unroll
    for (int k = 0; k < 7; ++k) {
      const int j = (cy + k) * CL_TW + cx;
      const float w = p.g[k];
      m2 = fmaf(w, s_h[0][j], m2);
      m1 = fmaf(w, s_h[1][j], m1);
      e22 = fmaf(w, s_h[2][j], e22);
      e11 = fmaf(w, s_h[3][j], e11);
      e12 = fmaf(w, s_h[4][j], e12);
    }
    const float s11 = e11 - m1 * m1, s22 = e22 - m2 * m2, s12 = e12 - m1 * m2;
    const float num = (2.f * m1 * m2 + 1e-4f) * (2.f * s12 + 9e-4f);
    const float den = (m1 * m1 + m2 * m2 + 1e-4f) * (s11 + s22 + 9e-4f);
    const float ssim = num / (den + 1e-12f);
    sss += fminf(fmaxf((1.f - ssim) * 0.5f, 0.f), 1.f);
    // Sobel on the resized target, replicate border: neighbours by clamped
    // image coordinates (the region stores in-image pixels at their own place)
    const int ym = max(y - 1, 0) - y00, yp = min(y + 1, H - 1) - y00, yc = y - y00;
    const int xm = max(x - 1, 0) - x00, xp = min(x + 1, W - 1) - x00, xc = x - x00;
    const float t00 = s_t[ym * CL_RW + xm], t01 = s_t[ym * CL_RW + xc], t02 = s_t[ym * CL_RW + xp];
    const float t10 = s_t[yc * CL_RW + xm], t12 = s_t[yc * CL_RW + xp];
    const float t20 = s_t[yp * CL_RW + xm], t21 = s_t[yp * CL_RW + xc], t22 = s_t[yp * CL_RW + xp];
    const float dx = ((t02 - t00) + 2.f * (t12 - t10) + (t22 - t20)) * 0.125f;
    const float dy = ((t20 - t00) + 2.f * (t21 - t01) + (t22 - t02)) * 0.125f;
    sdx += fabsf(dx);
    sdy += fabsf(dy);
    const int c = yc * CL_RW + xc;
    sl1 += fabsf(s_t[c] - s_p[c]);
    if (METRICS) {
      const float q = s_p[c];
      const float depth = fminf(fmaxf(1.0f / q, p.clamp_lo), p.clamp_hi);
      const float e = s_t[c] - depth;
      sse = fmaf(e, e, sse);
      pmin = fminf(pmin, q);
      pmax = fmaxf(pmax, q);
    }
  }
  float* out = p.part + ((long)b * p.tilesX * p.tilesY + blockIdx.x) * PS;
  float t = block_sum<CL_NT>(sdx, s_red);
  if (tid == 0) out[0] = t;
  t = block_sum<CL_NT>(sdy, s_red);
  if (tid == 0) out[1] = t;
  t = block_sum<CL_NT>(sss, s_red);
  if (tid == 0) out[2] = t;
  t = block_sum<CL_NT>(sl1, s_red);
  if (tid == 0) out[3] = t;
  if (METRICS) {
    t = block_sum<CL_NT>(sse, s_red);
    if (tid == 0) out[4] = t;
    t = block_extreme<CL_NT, false>(pmin, s_red);
    if (tid == 0) out[5] = t;
    t = block_extreme<CL_NT, true>(pmax, s_red);
    if (tid == 0) out[6] = t;
  }
}

template <bool METRICS>
__global__ __launch_bounds__(256) void compute_loss_finalize_kernel(const ClParams p) {
  constexpr int PS = METRICS ? 8 : 4, NS = METRICS ? 5 : 4;  // partial stride, number of sums
  __shared__ double red[NS][256];
  __shared__ float ext[2][256];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int nt = p.tilesX * p.tilesY;
  double a[NS];
  for (int k = 0; k < NS; ++k) a[k] = 0.0;
  float lo = INFINITY, hi = -INFINITY;
  for (int t = tid; t < nt; t += 256) {
    const float* src = p.part + ((long)b * nt + t) * PS;
    for (int k = 0; k < NS; ++k) a[k] += (double)src[k];
    if (METRICS) {
      lo = fminf(lo, src[5]);
      hi = fmaxf(hi, src[6]);
    }
  }
  for (int k = 0; k < NS; ++k) red[k][tid] = a[k];
  ext[0][tid] = lo;
  ext[1][tid] = hi;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) {
      for (int k = 0; k < NS; ++k) red[k][tid] += red[k][tid + o];
      ext[0][tid] = fminf(ext[0][tid], ext[0][tid + o]);
      ext[1][tid] = fmaxf(ext[1][tid], ext[1][tid + o]);
    }
    __syncthreads();
  }
  if (tid == 0) {
    const double n = (double)p.H * p.W;
    const float mx = (float)(red[0][0] / n), my = (float)(red[1][0] / n);
    const float smooth = mx * expf(mx) + my * expf(my);
    const float ssim_term = 1.f - (float)(red[2][0] / n);
    const float l1 = (float)(red[3][0] / n);
    float* o = p.out + (long)b * PS;
    o[0] = 0.75f * ssim_term + 0.20f * l1 + 0.15f * smooth;
    o[1] = ssim_term;
    o[2] = l1;
    o[3] = smooth;
    if (METRICS) {
      o[4] = sqrtf((float)(red[NS - 1][0] / n));
      o[5] = ext[0][0];
      o[6] = ext[1][0];
      o[7] = 0.f;
    }
  }
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

}  // namespace dw

using namespace dw;

extern "C" size_t dw_compute_loss_workspace_bytes(int32_t B, int32_t H, int32_t W) {
  if (B < 1 || H < 4 || W < 4) return 0;
  const size_t tiles = (size_t)((W + CL_TW - 1) / CL_TW) * ((H + CL_TH - 1) / CL_TH);
  return align_up(sizeof(float) * 8 * tiles * B, 256);  // enough for the metrics variant too
}

template <bool METRICS>
static int compute_loss_launch(int32_t B, int32_t Ht, int32_t Wt, int32_t H, int32_t W,
                               const float* target, const float* pred, float clamp_lo, float clamp_hi,
                               float* out, int32_t flags, void* workspace, size_t workspace_bytes, void* stream) {
  if (!target || !pred || !out || !workspace) return DW_ERR_NULL;
  if (flags & ~DW_CL_FLAG_ANTIALIAS) return DW_ERR_ARG;
  // reflect padding of 3 needs at least 4 pixels per side
  if (B < 1 || B > 65535 || H < 4 || W < 4 || Ht < 1 || Wt < 1) return DW_ERR_SHAPE;
  if (workspace_bytes < dw_compute_loss_workspace_bytes(B, H, W) || ((uintptr_t)workspace & 255))
    return DW_ERR_WORKSPACE;
  ClParams p{};
  p.B = B;
  p.Ht = Ht;
  p.Wt = Wt;
  p.H = H;
  p.W = W;
  p.tilesX = (W + CL_TW - 1) / CL_TW;
  p.tilesY = (H + CL_TH - 1) / CL_TH;
  p.antialias = (flags & DW_CL_FLAG_ANTIALIAS) ? 1 : 0;
  p.target = target;
  p.pred = pred;
  p.part = (float*)workspace;
  p.out = out;
  p.clamp_lo = clamp_lo;
  p.clamp_hi = clamp_hi;
  double g[7], sum = 0.0;
  for (int k = 0; k < 7; ++k) {
    const double x = k - 3;
    g[k] = exp(-(x * x) / (2.0 * 1.5 * 1.5));
    sum += g[k];
  }
  for (int k = 0; k < 7; ++k) p.g[k] = (float)(g[k] / sum);
  hipStream_t st = (hipStream_t)stream;
  compute_loss_tile_kernel<METRICS><<<dim3(p.tilesX * p.tilesY, B), CL_NT, 0, st>>>(p);
  compute_loss_finalize_kernel<METRICS><<<B, 256, 0, st>>>(p);
  return hipGetLastError() == hipSuccess ? DW_OK : DW_ERR_LAUNCH;
}

extern "C" int dw_compute_loss_fwd(int32_t B, int32_t Ht, int32_t Wt, int32_t H, int32_t W,
                                   const float* target, const float* pred, float* out, int32_t flags,
                                   void* workspace, size_t workspace_bytes, void* stream) {
  return compute_loss_launch<false>(B, Ht, Wt, H, W, target, pred, 0.f, 0.f, out, flags, workspace,
                                    workspace_bytes, stream);
}

extern "C" int dw_compute_loss_metrics_fwd(int32_t B, int32_t Ht, int32_t Wt, int32_t H, int32_t W,
                                           const float* target, const float* pred, float clamp_min,
                                           float clamp_max, float* out, int32_t flags, void* workspace,
                                           size_t workspace_bytes, void* stream) {
  if (!(clamp_min <= clamp_max)) return DW_ERR_ARG;
  return compute_loss_launch<true>(B, Ht, Wt, H, W, target, pred, clamp_min, clamp_max, out, flags, workspace,
                                   workspace_bytes, stream);
}

// ---------------------------------------------------------------- allocation-free entry point
// The per-tile partial sums of dw_compute_loss_fwd live in a fixed device array
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
                                          const float* target, const float* pred, float* out, int32_t flags,
                                          void* stream) {
  if (!target || !pred || !out) return DW_ERR_NULL;
  if (B < 1 || B > 65535 || H < 4 || W < 4 || Ht < 1 || Wt < 1) return DW_ERR_SHAPE;
  const size_t need = dw_compute_loss_workspace_bytes(B, H, W);
  if (need == 0) return DW_ERR_SHAPE;
  if (need > DW_STATIC_SCRATCH_BYTES) return DW_ERR_TOO_LARGE;
  void* ptr = static_scratch_ptr();
  if (!ptr) return DW_ERR_LAUNCH;
  // every partial the finalisation reads is written by the tile kernel of the same call:
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
