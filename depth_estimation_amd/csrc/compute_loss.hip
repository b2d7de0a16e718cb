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
#include "dw_common.h"

namespace dw {

constexpr int CL_TH = 16, CL_TW = 64, CL_NT = 256, CL_HALO = 3;
constexpr int CL_RW = CL_TW + 2 * CL_HALO, CL_RH = CL_TH + 2 * CL_HALO, CL_R = CL_RW * CL_RH;

struct ClParams {
  int B, Ht, Wt, H, W, tilesX, tilesY;
  const float* target;
  const float* pred;
  float* part;  // [B][tiles][4]: sum|dx|, sum|dy|, sum ssim-loss, sum|t - p|
  float* out;   // [B][4]
  float g[7];   // Gaussian taps
};

__global__ __launch_bounds__(CL_NT) void compute_loss_tile_kernel(const ClParams p) {
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
  }
  float* out = p.part + ((long)b * p.tilesX * p.tilesY + blockIdx.x) * 4;
  float t = block_sum<CL_NT>(sdx, s_red);
  if (tid == 0) out[0] = t;
  t = block_sum<CL_NT>(sdy, s_red);
  if (tid == 0) out[1] = t;
  t = block_sum<CL_NT>(sss, s_red);
  if (tid == 0) out[2] = t;
  t = block_sum<CL_NT>(sl1, s_red);
  if (tid == 0) out[3] = t;
}

__global__ __launch_bounds__(256) void compute_loss_finalize_kernel(const ClParams p) {
  __shared__ double red[4][256];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int nt = p.tilesX * p.tilesY;
  double a[4] = {0.0, 0.0, 0.0, 0.0};
  for (int t = tid; t < nt; t += 256) {
    const float* src = p.part + ((long)b * nt + t) * 4;
    for (int k = 0; k < 4; ++k) a[k] += (double)src[k];
  }
  for (int k = 0; k < 4; ++k) red[k][tid] = a[k];
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o)
      for (int k = 0; k < 4; ++k) red[k][tid] += red[k][tid + o];
    __syncthreads();
  }
  if (tid == 0) {
    const double n = (double)p.H * p.W;
    const float mx = (float)(red[0][0] / n), my = (float)(red[1][0] / n);
    const float smooth = mx * expf(mx) + my * expf(my);
    const float ssim_term = 1.f - (float)(red[2][0] / n);
    const float l1 = (float)(red[3][0] / n);
    float* o = p.out + (long)b * 4;
    o[0] = 0.75f * ssim_term + 0.20f * l1 + 0.15f * smooth;
    o[1] = ssim_term;
    o[2] = l1;
    o[3] = smooth;
  }
}

}  // namespace dw

using namespace dw;

extern "C" size_t dw_compute_loss_workspace_bytes(int32_t B, int32_t H, int32_t W) {
  if (B < 1 || H < 4 || W < 4) return 0;
  const size_t tiles = (size_t)((W + CL_TW - 1) / CL_TW) * ((H + CL_TH - 1) / CL_TH);
  return align_up(sizeof(float) * 4 * tiles * B, 256);
}

extern "C" int dw_compute_loss_fwd(int32_t B, int32_t Ht, int32_t Wt, int32_t H, int32_t W,
                                   const float* target, const float* pred, float* out,
                                   void* workspace, size_t workspace_bytes, void* stream) {
  if (!target || !pred || !out || !workspace) return DW_ERR_NULL;
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
  p.target = target;
  p.pred = pred;
  p.part = (float*)workspace;
  p.out = out;
  double g[7], sum = 0.0;
  for (int k = 0; k < 7; ++k) {
    const double x = k - 3;
    g[k] = exp(-(x * x) / (2.0 * 1.5 * 1.5));
    sum += g[k];
  }
  for (int k = 0; k < 7; ++k) p.g[k] = (float)(g[k] / sum);
  hipStream_t st = (hipStream_t)stream;
  compute_loss_tile_kernel<<<dim3(p.tilesX * p.tilesY, B), CL_NT, 0, st>>>(p);
  compute_loss_finalize_kernel<<<B, 256, 0, st>>>(p);
  return hipGetLastError() == hipSuccess ? DW_OK : DW_ERR_LAUNCH;
}
