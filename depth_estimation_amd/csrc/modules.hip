// Unfused building blocks of the view-synthesis stack, one kernel pair each:
// disparity->depth, back-projection, projection to a sampling grid, bilinear
// border warp, 3x3 SSIM map, edge-aware smoothness.  They exist so that code
// written against the usual module interfaces (BackprojectDepth, Project3D,
// SSIM, get_smooth_loss, grid_sample) can be switched over piece by piece; the
// fused kernel in photometric.hip is the fast path.  None of these stages is
// present in the reference (/root/reference).
//
// Forward arithmetic follows the same fixed association order as the fused
// kernel (file compiled with -ffp-contract=off); backward uses fmaf freely.
#include "dw_common.h"

namespace dw {

constexpr int EW_NT = 256;

__device__ __forceinline__ float sgnf(float a) { return (a > 0.f) ? 1.f : ((a < 0.f) ? -1.f : 0.f); }

// ------------------------------------------------------------ disp <-> depth
__global__ __launch_bounds__(EW_NT) void disp_to_depth_fwd_kernel(long n, const float* __restrict__ disp,
                                                                  float min_disp, float range,
                                                                  float* __restrict__ scaled,
                                                                  float* __restrict__ depth) {
  for (long i = (long)blockIdx.x * EW_NT + threadIdx.x; i < n; i += (long)gridDim.x * EW_NT) {
    const float s = min_disp + range * disp[i];
    scaled[i] = s;
    depth[i] = 1.0f / s;
  }
}

__global__ __launch_bounds__(EW_NT) void disp_to_depth_bwd_kernel(long n, const float* __restrict__ depth,
                                                                  const float* __restrict__ g_scaled,
                                                                  const float* __restrict__ g_depth,
                                                                  float range, float* __restrict__ g_disp) {
  for (long i = (long)blockIdx.x * EW_NT + threadIdx.x; i < n; i += (long)gridDim.x * EW_NT) {
    float g = 0.f;
    if (g_scaled) g = g_scaled[i] * range;
    if (g_depth) {
      const float d = depth[i];
      g = fmaf(g_depth[i], -range * d * d, g);
    }
    g_disp[i] = g;
  }
}

// ------------------------------------------------------------ back-projection
// out[b][i][n] = depth[b][n] * ((invK_i0 * x + invK_i1 * y) + invK_i2), out[b][3][n] = 1
__global__ __launch_bounds__(EW_NT) void backproject_fwd_kernel(int H, int W, const float* __restrict__ depth,
                                                                const float* __restrict__ invK,
                                                                float* __restrict__ out) {
  const int b = blockIdx.y;
  const long N = (long)H * W;
  const float* k = invK + (long)b * 16;
  for (long n = (long)blockIdx.x * EW_NT + threadIdx.x; n < N; n += (long)gridDim.x * EW_NT) {
    const int y = (int)(n / W), x = (int)(n - (long)y * W);
    const float fx = (float)x, fy = (float)y, d = depth[b * N + n];
    float* o = out + (long)b * 4 * N + n;
    o[0] = d * ((k[0] * fx + k[1] * fy) + k[2]);
    o[N] = d * ((k[4] * fx + k[5] * fy) + k[6]);
    o[2 * N] = d * ((k[8] * fx + k[9] * fy) + k[10]);
    o[3 * N] = 1.0f;
  }
}

__global__ __launch_bounds__(EW_NT) void backproject_bwd_kernel(int H, int W, const float* __restrict__ invK,
                                                                const float* __restrict__ g_out,
                                                                float* __restrict__ g_depth) {
  const int b = blockIdx.y;
  const long N = (long)H * W;
  const float* k = invK + (long)b * 16;
  for (long n = (long)blockIdx.x * EW_NT + threadIdx.x; n < N; n += (long)gridDim.x * EW_NT) {
    const int y = (int)(n / W), x = (int)(n - (long)y * W);
    const float fx = (float)x, fy = (float)y;
    const float* g = g_out + (long)b * 4 * N + n;
    const float r0 = (k[0] * fx + k[1] * fy) + k[2];
    const float r1 = (k[4] * fx + k[5] * fy) + k[6];
    const float r2 = (k[8] * fx + k[9] * fy) + k[10];
    g_depth[b * N + n] = fmaf(g[0], r0, fmaf(g[N], r1, g[2 * N] * r2));
  }
}

// ------------------------------------------------------------ projection
__device__ __forceinline__ void load_P(const float* K, const float* T, int b, float* P) {
  const float* Kb = K + (long)b * 16;
  const float* Tb = T + (long)b * 16;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
      P[i * 4 + j] = ((Kb[i * 4 + 0] * Tb[0 * 4 + j] + Kb[i * 4 + 1] * Tb[1 * 4 + j]) +
                      Kb[i * 4 + 2] * Tb[2 * 4 + j]) +
                     Kb[i * 4 + 3] * Tb[3 * 4 + j];
}

__global__ __launch_bounds__(EW_NT) void project3d_fwd_kernel(int H, int W, const float* __restrict__ pts,
                                                              const float* __restrict__ K,
                                                              const float* __restrict__ T, float eps,
                                                              float* __restrict__ grid) {
  const int b = blockIdx.y;
  const long N = (long)H * W;
  float P[12];
  load_P(K, T, b, P);
  const float wm1 = (float)(W - 1), hm1 = (float)(H - 1);
  for (long n = (long)blockIdx.x * EW_NT + threadIdx.x; n < N; n += (long)gridDim.x * EW_NT) {
    const float* q = pts + (long)b * 4 * N + n;
    const float X = q[0], Y = q[N], Z = q[2 * N], Wh = q[3 * N];
    const float c0 = ((P[0] * X + P[1] * Y) + P[2] * Z) + P[3] * Wh;
    const float c1 = ((P[4] * X + P[5] * Y) + P[6] * Z) + P[7] * Wh;
    const float c2 = ((P[8] * X + P[9] * Y) + P[10] * Z) + P[11] * Wh;
    const float zp = c2 + eps;
    const float gx = ((c0 / zp) / wm1 - 0.5f) * 2.0f;
    const float gy = ((c1 / zp) / hm1 - 0.5f) * 2.0f;
    *reinterpret_cast<float2*>(grid + ((long)b * N + n) * 2) = make_float2(gx, gy);
  }
}

// g_points and per-block partial sums of d loss / d P (12 values)
__global__ __launch_bounds__(EW_NT) void project3d_bwd_kernel(int H, int W, const float* __restrict__ pts,
                                                              const float* __restrict__ K,
                                                              const float* __restrict__ T, float eps,
                                                              const float* __restrict__ g_grid,
                                                              float* __restrict__ g_pts,
                                                              float* __restrict__ part) {
  __shared__ float red[EW_NT / kWave];
  const int b = blockIdx.y;
  const long N = (long)H * W;
  float P[12];
  load_P(K, T, b, P);
  const float sxn = 2.0f / (float)(W - 1), syn = 2.0f / (float)(H - 1);
  float dP[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) dP[k] = 0.f;
  for (long n = (long)blockIdx.x * EW_NT + threadIdx.x; n < N; n += (long)gridDim.x * EW_NT) {
    const float* q = pts + (long)b * 4 * N + n;
    const float pt[4] = {q[0], q[N], q[2 * N], q[3 * N]};
    const float c0 = fmaf(P[0], pt[0], fmaf(P[1], pt[1], fmaf(P[2], pt[2], P[3] * pt[3])));
    const float c1 = fmaf(P[4], pt[0], fmaf(P[5], pt[1], fmaf(P[6], pt[2], P[7] * pt[3])));
    const float c2 = fmaf(P[8], pt[0], fmaf(P[9], pt[1], fmaf(P[10], pt[2], P[11] * pt[3])));
    const float iz = 1.0f / (c2 + eps);
    const float2 g = *reinterpret_cast<const float2*>(g_grid + ((long)b * N + n) * 2);
    const float gu = g.x * sxn, gv = g.y * syn;
    const float gc[3] = {gu * iz, gv * iz, -(gu * c0 * iz + gv * c1 * iz) * iz};
    float* o = g_pts + (long)b * 4 * N + n;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j * N] = fmaf(gc[0], P[j], fmaf(gc[1], P[4 + j], gc[2] * P[8 + j]));
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) dP[i * 4 + j] = fmaf(gc[i], pt[j], dP[i * 4 + j]);
  }
  if (part) {
#pragma unroll
    for (int k = 0; k < 12; ++k) {
      const float t = block_sum<EW_NT>(dP[k], red);
      if (threadIdx.x == 0) part[((long)b * gridDim.x + blockIdx.x) * 12 + k] = t;
    }
  }
}

__global__ __launch_bounds__(64) void project3d_gT_kernel(int nblk, const float* __restrict__ K,
                                                          const float* __restrict__ part,
                                                          float* __restrict__ g_T) {
  __shared__ double s_dP[12];
  const int b = blockIdx.x, tid = threadIdx.x;
  if (tid < 12) {
    double a = 0.0;
    for (int k = 0; k < nblk; ++k) a += (double)part[((long)b * nblk + k) * 12 + tid];
    s_dP[tid] = a;
  }
  __syncthreads();
  if (tid < 16) {
    const int k = tid >> 2, j = tid & 3;
    const float* Kb = K + (long)b * 16;
    double a = 0.0;
    for (int i = 0; i < 3; ++i) a += (double)Kb[i * 4 + k] * s_dP[i * 4 + j];
    g_T[(long)b * 16 + tid] = (float)a;
  }
}

// ------------------------------------------------------------ bilinear warp
// grid_sample(src, grid, bilinear, padding border, align_corners = true)
struct WarpGeom {
  float w_nw, w_ne, w_sw, w_se, tx, ty;
  unsigned o_nw, o_ne, o_sw, o_se;
  bool inx, iny;
};

__device__ __forceinline__ WarpGeom warp_geom(float gx, float gy, int H, int W) {
  WarpGeom g;
  const float wm1 = (float)(W - 1), hm1 = (float)(H - 1);
  float ix = ((gx + 1.0f) * 0.5f) * wm1;
  float iy = ((gy + 1.0f) * 0.5f) * hm1;
  g.inx = (ix > 0.f) && (ix < wm1);
  g.iny = (iy > 0.f) && (iy < hm1);
  ix = fminf(fmaxf(ix, 0.f), wm1);
  iy = fminf(fmaxf(iy, 0.f), hm1);
  const float ix0 = floorf(ix), iy0 = floorf(iy);
  g.tx = ix - ix0;
  g.ty = iy - iy0;
  const float ex = 1.0f - g.tx, ey = 1.0f - g.ty;
  const int x0 = (int)ix0, y0 = (int)iy0;
  const int x1 = min(x0 + 1, W - 1), y1 = min(y0 + 1, H - 1);
  g.w_nw = ex * ey;
  g.w_ne = g.tx * ey;
  g.w_sw = ex * g.ty;
  g.w_se = g.tx * g.ty;
  g.o_nw = (unsigned)(y0 * W + x0);
  g.o_ne = (unsigned)(y0 * W + x1);
  g.o_sw = (unsigned)(y1 * W + x0);
  g.o_se = (unsigned)(y1 * W + x1);
  return g;
}

__global__ __launch_bounds__(EW_NT) void warp_fwd_kernel(int C, int H, int W, long No,
                                                         const float* __restrict__ src,
                                                         const float* __restrict__ grid,
                                                         float* __restrict__ out) {
  const int b = blockIdx.y;
  const long plane = (long)H * W;
  for (long n = (long)blockIdx.x * EW_NT + threadIdx.x; n < No; n += (long)gridDim.x * EW_NT) {
    const float2 g = *reinterpret_cast<const float2*>(grid + ((long)b * No + n) * 2);
    const WarpGeom w = warp_geom(g.x, g.y, H, W);
    for (int c = 0; c < C; ++c) {
      const float* p = src + ((long)b * C + c) * plane;
      out[((long)b * C + c) * No + n] =
          ((p[w.o_nw] * w.w_nw + p[w.o_ne] * w.w_ne) + p[w.o_sw] * w.w_sw) + p[w.o_se] * w.w_se;
    }
  }
}

__global__ __launch_bounds__(EW_NT) void warp_bwd_kernel(int C, int H, int W, long No,
                                                         const float* __restrict__ src,
                                                         const float* __restrict__ grid,
                                                         const float* __restrict__ g_out,
                                                         float* __restrict__ g_grid) {
  const int b = blockIdx.y;
  const long plane = (long)H * W;
  const float mx = 0.5f * (float)(W - 1), my = 0.5f * (float)(H - 1);
  for (long n = (long)blockIdx.x * EW_NT + threadIdx.x; n < No; n += (long)gridDim.x * EW_NT) {
    const float2 g = *reinterpret_cast<const float2*>(grid + ((long)b * No + n) * 2);
    const WarpGeom w = warp_geom(g.x, g.y, H, W);
    float gix = 0.f, giy = 0.f;
    for (int c = 0; c < C; ++c) {
      const float* p = src + ((long)b * C + c) * plane;
      const float nw = p[w.o_nw], ne = p[w.o_ne], sw = p[w.o_sw], se = p[w.o_se];
      const float go = g_out[((long)b * C + c) * No + n];
      gix = fmaf(go, fmaf(se - sw, w.ty, (ne - nw) * (1.f - w.ty)), gix);
      giy = fmaf(go, fmaf(se - ne, w.tx, (sw - nw) * (1.f - w.tx)), giy);
    }
    *reinterpret_cast<float2*>(g_grid + ((long)b * No + n) * 2) =
        make_float2(w.inx ? gix * mx : 0.f, w.iny ? giy * my : 0.f);
  }
}

// ------------------------------------------------------------ SSIM map (3x3)
constexpr int SS_TH = 16, SS_TW = 64, SS_NT = 256;

__device__ __forceinline__ float box3(const float* a, int pos, int pitch) {
  const float c0 = (a[pos - pitch - 1] + a[pos - 1]) + a[pos + pitch - 1];
  const float c1 = (a[pos - pitch] + a[pos]) + a[pos + pitch];
  const float c2 = (a[pos - pitch + 1] + a[pos + 1]) + a[pos + pitch + 1];
  return ((c0 + c1) + c2) * kNinth;
}
__device__ __forceinline__ float box3p(const float* a, const float* b, int pos, int pitch) {
  const float c0 = (a[pos - pitch - 1] * b[pos - pitch - 1] + a[pos - 1] * b[pos - 1]) +
                   a[pos + pitch - 1] * b[pos + pitch - 1];
  const float c1 = (a[pos - pitch] * b[pos - pitch] + a[pos] * b[pos]) + a[pos + pitch] * b[pos + pitch];
  const float c2 = (a[pos - pitch + 1] * b[pos - pitch + 1] + a[pos + 1] * b[pos + 1]) +
                   a[pos + pitch + 1] * b[pos + pitch + 1];
  return ((c0 + c1) + c2) * kNinth;
}

struct SsimPx {
  float mux, muy, A1, A2, B1, B2, r, raw;
};
__device__ __forceinline__ float ssim_px(const float* xs, const float* ys, int pos, int pitch, SsimPx& t) {
  t.mux = box3(xs, pos, pitch);
  t.muy = box3(ys, pos, pitch);
  const float sx = box3p(xs, xs, pos, pitch) - t.mux * t.mux;
  const float sy = box3p(ys, ys, pos, pitch) - t.muy * t.muy;
  const float sxy = box3p(xs, ys, pos, pitch) - t.mux * t.muy;
  t.A1 = (2.0f * t.mux) * t.muy + kC1;
  t.A2 = 2.0f * sxy + kC2;
  t.B1 = (t.mux * t.mux + t.muy * t.muy) + kC1;
  t.B2 = (sx + sy) + kC2;
  t.r = (t.A1 * t.A2) / (t.B1 * t.B2);
  t.raw = (1.0f - t.r) * 0.5f;
  return fminf(fmaxf(t.raw, 0.f), 1.f);
}

template <int HALO>
__device__ __forceinline__ void load_region(const float* __restrict__ img, int H, int W, int y00, int x00,
                                            float* s) {
  constexpr int RWd = SS_TW + 2 * HALO, RHt = SS_TH + 2 * HALO;
  for (int i = threadIdx.x; i < RWd * RHt; i += SS_NT) {
    const int ry = i / RWd, rx = i - ry * RWd;
    s[i] = img[(long)reflect_index(y00 + ry, H) * W + reflect_index(x00 + rx, W)];
  }
}

__global__ __launch_bounds__(SS_NT) void ssim_fwd_kernel(int H, int W, int tilesX,
                                                         const float* __restrict__ x,
                                                         const float* __restrict__ y,
                                                         float* __restrict__ out) {
  constexpr int RWd = SS_TW + 2, RHt = SS_TH + 2;
  __shared__ float sx[RWd * RHt], sy[RWd * RHt];
  const long plane = (long)H * W;
  const int tY = blockIdx.x / tilesX, tX = blockIdx.x - tY * tilesX;
  const int y00 = tY * SS_TH - 1, x00 = tX * SS_TW - 1;
  const float* xp = x + blockIdx.y * plane;
  const float* yp = y + blockIdx.y * plane;
  load_region<1>(xp, H, W, y00, x00, sx);
  load_region<1>(yp, H, W, y00, x00, sy);
  __syncthreads();
  for (int i = threadIdx.x; i < SS_TH * SS_TW; i += SS_NT) {
    const int cy = i / SS_TW, cx = i - cy * SS_TW;
    const int yy = y00 + 1 + cy, xx = x00 + 1 + cx;
    if (yy >= H || xx >= W) continue;
    SsimPx t;
    out[blockIdx.y * plane + (long)yy * W + xx] = ssim_px(sx, sy, (cy + 1) * RWd + cx + 1, RWd, t);
  }
}

__global__ __launch_bounds__(SS_NT) void ssim_bwd_kernel(int H, int W, int tilesX,
                                                         const float* __restrict__ x,
                                                         const float* __restrict__ y,
                                                         const float* __restrict__ g_out,
                                                         float* __restrict__ g_x,
                                                         float* __restrict__ g_y) {
  constexpr int RWd = SS_TW + 4, RHt = SS_TH + 4;  // colours: halo 2
  constexpr int QWd = SS_TW + 2, QHt = SS_TH + 2;  // coefficients: halo 1
  __shared__ float sx[RWd * RHt], sy[RWd * RHt];
  // Per window: the adjoint with respect to a pixel (x, y) inside it is linear,
  //   d/dx = c0 + 2 g1 (x - mux) + g2 (y - muy),   d/dy = c3 + 2 g1 (y - muy) + g2 (x - mux),
  // written around the window's own means: the un-centred form (constant + 2 g1 x + g2 y) sums
  // terms of the order of the squared means to a result of the order of the variances, which
  // in float32 loses B1 / B2 of the precision in flat regions.
  __shared__ float sG[6][QWd * QHt];  // c0, g1, g2, c3, mux, muy
  const long plane = (long)H * W;
  const int tY = blockIdx.x / tilesX, tX = blockIdx.x - tY * tilesX;
  const int y00 = tY * SS_TH - 2, x00 = tX * SS_TW - 2;
  const float* xp = x + blockIdx.y * plane;
  const float* yp = y + blockIdx.y * plane;
  const float* gp = g_out + blockIdx.y * plane;
  load_region<2>(xp, H, W, y00, x00, sx);
  load_region<2>(yp, H, W, y00, x00, sy);
  __syncthreads();
  for (int q = threadIdx.x; q < QWd * QHt; q += SS_NT) {
    const int qy = q / QWd, qx = q - qy * QWd;
    const int yy = y00 + 1 + qy, xx = x00 + 1 + qx;
    float g[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
      SsimPx t;
      ssim_px(sx, sy, (qy + 1) * RWd + qx + 1, RWd, t);
      if (t.raw >= 0.f && t.raw <= 1.f) {
        const float up = -0.5f * gp[(long)yy * W + xx] / (t.B1 * t.B2);
        g[0] = up * (2.0f * (t.muy * t.A2 - (t.r * t.mux) * t.B2));
        g[1] = up * (-t.r * t.B1);
        g[2] = up * (2.0f * t.A1);
        g[3] = up * (2.0f * (t.mux * t.A2 - (t.r * t.muy) * t.B2));
        g[4] = t.mux;
        g[5] = t.muy;
      }
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) sG[k][q] = g[k];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < SS_TH * SS_TW; i += SS_NT) {
    const int cy = i / SS_TW, cx = i - cy * SS_TW;
    const int yy = y00 + 2 + cy, xx = x00 + 2 + cx;
    if (yy >= H || xx >= W) continue;
    const int pos = (cy + 2) * RWd + cx + 2;
    const float xv = sx[pos], yv = sy[pos];
    float ax = 0.f, ay = 0.f;
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy) {
      const float wy = 1.f + ((dy == -1 && yy == 1) ? 1.f : 0.f) + ((dy == 1 && yy == H - 2) ? 1.f : 0.f);
#pragma unroll
      for (int dx = -1; dx <= 1; ++dx) {
        const float w = wy * (1.f + ((dx == -1 && xx == 1) ? 1.f : 0.f) +
                              ((dx == 1 && xx == W - 2) ? 1.f : 0.f));
        const int q = (cy + 1 + dy) * QWd + (cx + 1 + dx);
        const float dxm = xv - sG[4][q], dym = yv - sG[5][q];  // a window without a gradient has all six planes zero
        const float g1 = 2.0f * sG[1][q], g2 = sG[2][q];
        ax = fmaf(w, fmaf(g1, dxm, fmaf(g2, dym, sG[0][q])), ax);
        ay = fmaf(w, fmaf(g1, dym, fmaf(g2, dxm, sG[3][q])), ay);
      }
    }
    const long o = blockIdx.y * plane + (long)yy * W + xx;
    if (g_x) g_x[o] = kNinth * ax;
    if (g_y) g_y[o] = kNinth * ay;
  }
}

// ------------------------------------------------------------ edge-aware smoothness
// loss = mean_x(|d(x) - d(x+1)| e^{-mean_c|I(x) - I(x+1)|}) + the same in y
constexpr int SL_PX = 1024, SL_NT = 256;

template <bool BWD>
__global__ __launch_bounds__(SL_NT) void smooth_loss_kernel(int B, int H, int W, const float* __restrict__ disp,
                                                            const float* __restrict__ img,
                                                            const float* __restrict__ g_loss,
                                                            float* __restrict__ part,
                                                            float* __restrict__ g_disp) {
  __shared__ float red[SL_NT / kWave];
  const int b = blockIdx.y;
  const long n = (long)H * W;
  const float* d = disp + b * n;
  const float* I = img + (long)b * 3 * n;
  const float icx = 1.0f / ((float)B * H * (W - 1)), icy = 1.0f / ((float)B * (H - 1) * W);
  const float go = BWD ? g_loss[0] : 0.f;
  float lx = 0.f, ly = 0.f;
  for (int k = threadIdx.x; k < SL_PX; k += SL_NT) {
    const long i = (long)blockIdx.x * SL_PX + k;
    if (i >= n) break;
    const int y = (int)(i / W), x = (int)(i - (long)y * W);
    const float dc = d[i], c0 = I[i], c1 = I[n + i], c2 = I[2 * n + i];
    auto edge = [&](long j) -> float {
      const float m = (fabsf(c0 - I[j]) + fabsf(c1 - I[n + j]) + fabsf(c2 - I[2 * n + j])) * kThird;
      return __expf(-m);
    };
    float G = 0.f;
    if (x + 1 < W) {
      const float a = dc - d[i + 1], w = edge(i + 1);
      lx += fabsf(a) * w;
      G += sgnf(a) * w * icx;
    }
    if (BWD && x > 0) G -= sgnf(d[i - 1] - dc) * edge(i - 1) * icx;
    if (y + 1 < H) {
      const float a = dc - d[i + W], w = edge(i + W);
      ly += fabsf(a) * w;
      G += sgnf(a) * w * icy;
    }
    if (BWD && y > 0) G -= sgnf(d[i - W] - dc) * edge(i - W) * icy;
    if (BWD) g_disp[b * n + i] = go * G;
  }
  if (!BWD) {
    float* o = part + ((long)b * gridDim.x + blockIdx.x) * 2;
    float t = block_sum<SL_NT>(lx, red);
    if (threadIdx.x == 0) o[0] = t;
    t = block_sum<SL_NT>(ly, red);
    if (threadIdx.x == 0) o[1] = t;
  }
}

__global__ __launch_bounds__(256) void smooth_loss_finalize_kernel(int B, int H, int W, int nblk,
                                                                   const float* __restrict__ part,
                                                                   float* __restrict__ out) {
  __shared__ double red[2][256];
  double ax = 0.0, ay = 0.0;
  for (int k = threadIdx.x; k < B * nblk; k += 256) {
    ax += (double)part[(long)k * 2];
    ay += (double)part[(long)k * 2 + 1];
  }
  red[0][threadIdx.x] = ax;
  red[1][threadIdx.x] = ay;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) {
      red[0][threadIdx.x] += red[0][threadIdx.x + o];
      red[1][threadIdx.x] += red[1][threadIdx.x + o];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0)
    out[0] = (float)(red[0][0] / ((double)B * H * (W - 1)) + red[1][0] / ((double)B * (H - 1) * W));
}

static inline int ew_blocks(long n) {
  long b = (n + EW_NT - 1) / EW_NT;
  return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}
static inline bool bad_dims(int B, int H, int W) {
  return B < 1 || B > 65535 || H < 2 || W < 2 || (long)H * W > (1l << 30);
}

}  // namespace dw

using namespace dw;

#define DW_LAUNCH_RC() (hipGetLastError() == hipSuccess ? DW_OK : DW_ERR_LAUNCH)

extern "C" int dw_disp_to_depth_fwd(int64_t n, const float* disp, float min_disp, float range,
                                    float* scaled, float* depth, void* stream) {
  if (!disp || !scaled || !depth) return DW_ERR_NULL;
  if (n < 1) return DW_ERR_SHAPE;
  disp_to_depth_fwd_kernel<<<ew_blocks(n), EW_NT, 0, (hipStream_t)stream>>>(n, disp, min_disp, range,
                                                                          scaled, depth);
  return DW_LAUNCH_RC();
}

extern "C" int dw_disp_to_depth_bwd(int64_t n, const float* depth, const float* g_scaled,
                                    const float* g_depth, float range, float* g_disp, void* stream) {
  if (!depth || !g_disp) return DW_ERR_NULL;
  if (n < 1) return DW_ERR_SHAPE;
  disp_to_depth_bwd_kernel<<<ew_blocks(n), EW_NT, 0, (hipStream_t)stream>>>(n, depth, g_scaled, g_depth,
                                                                          range, g_disp);
  return DW_LAUNCH_RC();
}

extern "C" int dw_backproject_fwd(int32_t B, int32_t H, int32_t W, const float* depth,
                                  const float* inv_K, float* points, void* stream) {
  if (!depth || !inv_K || !points) return DW_ERR_NULL;
  if (bad_dims(B, H, W)) return DW_ERR_SHAPE;
  backproject_fwd_kernel<<<dim3(ew_blocks((long)H * W), B), EW_NT, 0, (hipStream_t)stream>>>(
      H, W, depth, inv_K, points);
  return DW_LAUNCH_RC();
}

extern "C" int dw_backproject_bwd(int32_t B, int32_t H, int32_t W, const float* inv_K,
                                  const float* g_points, float* g_depth, void* stream) {
  if (!inv_K || !g_points || !g_depth) return DW_ERR_NULL;
  if (bad_dims(B, H, W)) return DW_ERR_SHAPE;
  backproject_bwd_kernel<<<dim3(ew_blocks((long)H * W), B), EW_NT, 0, (hipStream_t)stream>>>(
      H, W, inv_K, g_points, g_depth);
  return DW_LAUNCH_RC();
}

extern "C" int dw_project3d_fwd(int32_t B, int32_t H, int32_t W, const float* points, const float* K,
                                const float* T, float eps, float* grid, void* stream) {
  if (!points || !K || !T || !grid) return DW_ERR_NULL;
  if (bad_dims(B, H, W)) return DW_ERR_SHAPE;
  project3d_fwd_kernel<<<dim3(ew_blocks((long)H * W), B), EW_NT, 0, (hipStream_t)stream>>>(
      H, W, points, K, T, eps, grid);
  return DW_LAUNCH_RC();
}

static int p3d_blocks(int H, int W) {
  const long b = ((long)H * W + 4 * EW_NT - 1) / (4 * EW_NT);
  return (int)(b > 1024 ? 1024 : (b < 1 ? 1 : b));
}

extern "C" size_t dw_project3d_workspace_bytes(int32_t B, int32_t H, int32_t W) {
  if (bad_dims(B, H, W)) return 0;
  return align_up(sizeof(float) * 12 * (size_t)B * p3d_blocks(H, W), 256);
}

extern "C" int dw_project3d_bwd(int32_t B, int32_t H, int32_t W, const float* points, const float* K,
                                const float* T, float eps, const float* g_grid, float* g_points,
                                float* g_T, void* workspace, size_t workspace_bytes, void* stream) {
  if (!points || !K || !T || !g_grid || !g_points) return DW_ERR_NULL;
  if (bad_dims(B, H, W)) return DW_ERR_SHAPE;
  if (g_T && (!workspace || workspace_bytes < dw_project3d_workspace_bytes(B, H, W) ||
              ((uintptr_t)workspace & 255)))
    return DW_ERR_WORKSPACE;
  const int nblk = p3d_blocks(H, W);
  hipStream_t st = (hipStream_t)stream;
  project3d_bwd_kernel<<<dim3(nblk, B), EW_NT, 0, st>>>(H, W, points, K, T, eps, g_grid, g_points,
                                                      g_T ? (float*)workspace : nullptr);
  if (g_T) project3d_gT_kernel<<<B, 64, 0, st>>>(nblk, K, (const float*)workspace, g_T);
  return DW_LAUNCH_RC();
}

extern "C" int dw_warp_fwd(int32_t B, int32_t C, int32_t H, int32_t W, int32_t Ho, int32_t Wo,
                           const float* src, const float* grid, float* out, void* stream) {
  if (!src || !grid || !out) return DW_ERR_NULL;
  if (bad_dims(B, H, W) || Ho < 1 || Wo < 1 || C < 1) return DW_ERR_SHAPE;
  warp_fwd_kernel<<<dim3(ew_blocks((long)Ho * Wo), B), EW_NT, 0, (hipStream_t)stream>>>(
      C, H, W, (long)Ho * Wo, src, grid, out);
  return DW_LAUNCH_RC();
}

extern "C" int dw_warp_bwd(int32_t B, int32_t C, int32_t H, int32_t W, int32_t Ho, int32_t Wo,
                           const float* src, const float* grid, const float* g_out, float* g_grid,
                           void* stream) {
  if (!src || !grid || !g_out || !g_grid) return DW_ERR_NULL;
  if (bad_dims(B, H, W) || Ho < 1 || Wo < 1 || C < 1) return DW_ERR_SHAPE;
  warp_bwd_kernel<<<dim3(ew_blocks((long)Ho * Wo), B), EW_NT, 0, (hipStream_t)stream>>>(
      C, H, W, (long)Ho * Wo, src, grid, g_out, g_grid);
  return DW_LAUNCH_RC();
}

extern "C" int dw_ssim_fwd(int32_t planes, int32_t H, int32_t W, const float* x, const float* y,
                           float* out, void* stream) {
  if (!x || !y || !out) return DW_ERR_NULL;
  if (bad_dims(planes, H, W)) return DW_ERR_SHAPE;
  const int tx = (W + SS_TW - 1) / SS_TW, ty = (H + SS_TH - 1) / SS_TH;
  ssim_fwd_kernel<<<dim3(tx * ty, planes), SS_NT, 0, (hipStream_t)stream>>>(H, W, tx, x, y, out);
  return DW_LAUNCH_RC();
}

extern "C" int dw_ssim_bwd(int32_t planes, int32_t H, int32_t W, const float* x, const float* y,
                           const float* g_out, float* g_x, float* g_y, void* stream) {
  if (!x || !y || !g_out || (!g_x && !g_y)) return DW_ERR_NULL;
  if (bad_dims(planes, H, W)) return DW_ERR_SHAPE;
  const int tx = (W + SS_TW - 1) / SS_TW, ty = (H + SS_TH - 1) / SS_TH;
  ssim_bwd_kernel<<<dim3(tx * ty, planes), SS_NT, 0, (hipStream_t)stream>>>(H, W, tx, x, y, g_out, g_x,
                                                                          g_y);
  return DW_LAUNCH_RC();
}

extern "C" size_t dw_smooth_loss_workspace_bytes(int32_t B, int32_t H, int32_t W) {
  if (bad_dims(B, H, W)) return 0;
  return align_up(sizeof(float) * 2 * (size_t)B * (((long)H * W + SL_PX - 1) / SL_PX), 256);
}

extern "C" int dw_smooth_loss_fwd(int32_t B, int32_t H, int32_t W, const float* disp, const float* img,
                                  float* out, void* workspace, size_t workspace_bytes, void* stream) {
  if (!disp || !img || !out || !workspace) return DW_ERR_NULL;
  if (bad_dims(B, H, W)) return DW_ERR_SHAPE;
  if (workspace_bytes < dw_smooth_loss_workspace_bytes(B, H, W) || ((uintptr_t)workspace & 255))
    return DW_ERR_WORKSPACE;
  const int nblk = (int)(((long)H * W + SL_PX - 1) / SL_PX);
  hipStream_t st = (hipStream_t)stream;
  smooth_loss_kernel<false><<<dim3(nblk, B), SL_NT, 0, st>>>(B, H, W, disp, img, nullptr,
                                                            (float*)workspace, nullptr);
  smooth_loss_finalize_kernel<<<1, 256, 0, st>>>(B, H, W, nblk, (const float*)workspace, out);
  return DW_LAUNCH_RC();
}

extern "C" int dw_smooth_loss_bwd(int32_t B, int32_t H, int32_t W, const float* disp, const float* img,
                                  const float* g_loss, float* g_disp, void* stream) {
  if (!disp || !img || !g_loss || !g_disp) return DW_ERR_NULL;
  if (bad_dims(B, H, W)) return DW_ERR_SHAPE;
  const int nblk = (int)(((long)H * W + SL_PX - 1) / SL_PX);
  smooth_loss_kernel<true><<<dim3(nblk, B), SL_NT, 0, (hipStream_t)stream>>>(B, H, W, disp, img, g_loss,
                                                                           nullptr, g_disp);
  return DW_LAUNCH_RC();
}
