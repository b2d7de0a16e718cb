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
// Backward arithmetic uses explicit fmaf where it helps.
//
// Launch sequence of one training step:
//   forward : disp_mean_kernel -> photo_tile_kernel<GRAD> -> smooth_kernel
//             -> loss_finalize_kernel
//   backward: grad_disp_kernel -> grad_pose_kernel
// photo_tile_kernel<true> does the whole per-pixel backward at unit upstream
// gradient while the tile is still on chip and leaves d loss / d (up-sampled
// disparity) and per-tile pose partials in the workspace; the two backward
// kernels only finish the adjoint of the up-sampling, the smoothness term and
// the deterministic cross-tile sums, scaled by the upstream gradient.
#include "dw_common.h"

namespace dw {

constexpr int TH = 16;   // tile rows
constexpr int TW = 64;   // tile columns
constexpr int NT = 512;  // threads per tile block (2 waves per SIMD; LDS allows one block per CU)
constexpr int SM_PX = 1024;  // pixels per block in the per-scale map kernels
constexpr int SM_NT = 256;

struct ScaleLayout {
  int Hs[DW_MAX_SCALES], Ws[DW_MAX_SCALES];
  int nblk[DW_MAX_SCALES];         // blocks per image at scale s
  int blk_off[DW_MAX_SCALES + 1];  // prefix sum of B * nblk[s]
  long pix_off[DW_MAX_SCALES + 1]; // prefix sum of B * Hs * Ws
};

struct PhotoParams {
  int B, H, W, S;
  int tilesX, tilesY;
  const float* disp[DW_MAX_SCALES];
  const float* color[3];
  const float* pyr[DW_MAX_SCALES];
  const float* K;
  const float* invK;
  const float* T;
  const float* tie;
  uint8_t* mask[DW_MAX_SCALES];
  // workspace sections
  float* mean;       // [S][B]
  float* tile_loss;  // [B][tiles][S]
  float* tile_dP;    // [B][tiles][24]
  float* gdup;       // [S][B][H][W]
  float* sm_G;       // concatenated per scale [B][Hs][Ws]
  float* sm_part;    // [blk_off[S]][3]
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

struct Sample {
  // forward
  float val[3];
  bool valid;
  // what the backward needs
  float u, v, zp;
  float tx, ty;
  bool inx, iny;
  int o_nw, o_ne, o_sw, o_se;
};

// Project the camera-space point (X,Y,Z) with P (3x4, row-major) and sample
// the three colour planes of `src` bilinearly (border padding, corners aligned).
__device__ __forceinline__ void project_sample(const float* __restrict__ src, long plane, int H,
                                               int W, const float* P, float X, float Y, float Z,
                                               Sample& o) {
  const float c0 = ((P[0] * X + P[1] * Y) + P[2] * Z) + P[3];
  const float c1 = ((P[4] * X + P[5] * Y) + P[6] * Z) + P[7];
  const float c2 = ((P[8] * X + P[9] * Y) + P[10] * Z) + P[11];
  const float zp = c2 + kProjEps;
  const float u = c0 / zp;
  const float v = c1 / zp;
  const float wm1 = (float)(W - 1), hm1 = (float)(H - 1);
  const float gx = (u / wm1 - 0.5f) * 2.0f;
  const float gy = (v / hm1 - 0.5f) * 2.0f;
  o.valid = (gx >= -1.f) && (gx <= 1.f) && (gy >= -1.f) && (gy <= 1.f) && (c2 > 0.f);
  float ix = ((gx + 1.0f) * 0.5f) * wm1;
  float iy = ((gy + 1.0f) * 0.5f) * hm1;
  o.inx = (ix > 0.f) && (ix < wm1);
  o.iny = (iy > 0.f) && (iy < hm1);
  ix = fminf(fmaxf(ix, 0.f), wm1);
  iy = fminf(fmaxf(iy, 0.f), hm1);
  const float ix0 = floorf(ix), iy0 = floorf(iy);
  const float tx = ix - ix0, ty = iy - iy0;
  const float ex = 1.0f - tx, ey = 1.0f - ty;
  const int x0 = (int)ix0, y0 = (int)iy0;
  const int x1 = min(x0 + 1, W - 1), y1 = min(y0 + 1, H - 1);
  const float w_nw = ex * ey, w_ne = tx * ey, w_sw = ex * ty, w_se = tx * ty;
  o.o_nw = y0 * W + x0;
  o.o_ne = y0 * W + x1;
  o.o_sw = y1 * W + x0;
  o.o_se = y1 * W + x1;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float* p = src + c * plane;
    o.val[c] = ((p[o.o_nw] * w_nw + p[o.o_ne] * w_ne) + p[o.o_sw] * w_sw) + p[o.o_se] * w_se;
  }
  o.u = u;
  o.v = v;
  o.zp = zp;
  o.tx = tx;
  o.ty = ty;
}

// 3x3 sums around `pos` of a region image with row pitch RW: per column
// (top + mid) + bot, then (left + centre) + right.
__device__ __forceinline__ float box3_sum(const float* a, int pos, int RW) {
  const float c0 = (a[pos - RW - 1] + a[pos - 1]) + a[pos + RW - 1];
  const float c1 = (a[pos - RW] + a[pos]) + a[pos + RW];
  const float c2 = (a[pos - RW + 1] + a[pos + 1]) + a[pos + RW + 1];
  return (c0 + c1) + c2;
}
__device__ __forceinline__ float box3_sum_prod(const float* a, const float* b, int pos, int RW) {
  const float c0 = (a[pos - RW - 1] * b[pos - RW - 1] + a[pos - 1] * b[pos - 1]) +
                   a[pos + RW - 1] * b[pos + RW - 1];
  const float c1 = (a[pos - RW] * b[pos - RW] + a[pos] * b[pos]) + a[pos + RW] * b[pos + RW];
  const float c2 = (a[pos - RW + 1] * b[pos - RW + 1] + a[pos + 1] * b[pos + 1]) +
                   a[pos + RW + 1] * b[pos + RW + 1];
  return (c0 + c1) + c2;
}

struct SsimTerms {
  float A1, A2, B1, B2, raw;
};

__device__ __forceinline__ float ssim_loss_px(float mux, float muy, float exx, float eyy,
                                              float exy, SsimTerms& t) {
  const float sx = exx - mux * mux;
  const float sy = eyy - muy * muy;
  const float sxy = exy - mux * muy;
  t.A1 = (2.0f * mux) * muy + kC1;
  t.A2 = 2.0f * sxy + kC2;
  t.B1 = (mux * mux + muy * muy) + kC1;
  t.B2 = (sx + sy) + kC2;
  const float n = t.A1 * t.A2;
  const float d = t.B1 * t.B2;
  t.raw = (1.0f - n / d) * 0.5f;
  return fminf(fmaxf(t.raw, 0.f), 1.f);
}

// ------------------------------------------------------------ the tile kernel
template <bool GRAD>
struct TileShape {
  static constexpr int HALO = GRAD ? 2 : 1;
  static constexpr int RW = TW + 2 * HALO, RH = TH + 2 * HALO, R = RW * RH;  // warped / colour region
  static constexpr int QW = RW - 2, QH = RH - 2, Q = QW * QH;                // error-map region
};

template <bool GRAD>
__global__ __launch_bounds__(NT) void photo_tile_kernel(const PhotoParams p) {
  using G = TileShape<GRAD>;
  constexpr int HALO = G::HALO, RW = G::RW, RH = G::RH, R = G::R, QW = G::QW, QH = G::QH,
                Q = G::Q;
  __shared__ float s_tgt[3][R];
  __shared__ float s_xw[2][3][R];
  __shared__ float s_ystat[6][Q];  // per channel: mean, E[y^2]
  __shared__ float s_idm[Q];       // min of the two identity errors
  __shared__ unsigned char s_idi[Q];
  __shared__ unsigned char s_valid[R];
  __shared__ float s_G[GRAD ? 9 : 1][GRAD ? Q : 1];
  __shared__ signed char s_sel[GRAD ? Q : 1];
  __shared__ float s_P[2][12];
  __shared__ float s_invK[9];
  __shared__ float s_red[NT / kWave];

  const int tid = threadIdx.x;
  const int tile = blockIdx.x;
  const int b = blockIdx.y;
  const int tY = tile / p.tilesX, tX = tile - tY * p.tilesX;
  const int y00 = tY * TH - HALO, x00 = tX * TW - HALO;  // image coords of region (0,0)
  const int H = p.H, W = p.W;
  const long plane = (long)H * W;

  // ---- per-sample camera constants: P[f] = (K @ T[f])[:3], inv_K[:3,:3]
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
  const float* tgt_b = p.color[0] + (long)b * 3 * plane;
  const float* src_b[2] = {p.color[1] + (long)b * 3 * plane, p.color[2] + (long)b * 3 * plane};
  for (int i = tid; i < R; i += NT) {
    const int ry = i / RW, rx = i - ry * RW;
    const int y = reflect_index(y00 + ry, H), x = reflect_index(x00 + rx, W);
    const long o = (long)y * W + x;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      s_tgt[c][i] = tgt_b[c * plane + o];
      s_xw[0][c][i] = src_b[0][c * plane + o];
      s_xw[1][c][i] = src_b[1][c * plane + o];
    }
  }
  __syncthreads();

  // Photometric error of frame f at error-map position (qy, qx) from the
  // region images; optionally returns the SSIM terms per channel.
  auto photo_error = [&](int f, int pos, int q, SsimTerms* terms, float* mux_out) -> float {
    float ss[3], l1[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float* xa = s_xw[f][c];
      const float* ya = s_tgt[c];
      const float mux = box3_sum(xa, pos, RW) * kNinth;
      const float exx = box3_sum_prod(xa, xa, pos, RW) * kNinth;
      const float exy = box3_sum_prod(xa, ya, pos, RW) * kNinth;
      const float muy = s_ystat[2 * c][q], eyy = s_ystat[2 * c + 1][q];
      SsimTerms t;
      ss[c] = ssim_loss_px(mux, muy, exx, eyy, exy, t);
      l1[c] = fabsf(xa[pos] - ya[pos]);
      if (terms) {
        terms[c] = t;
        mux_out[c] = mux;
      }
    }
    const float ms = ((ss[0] + ss[1]) + ss[2]) * kThird;
    const float ml = ((l1[0] + l1[1]) + l1[2]) * kThird;
    return kWSsim * ms + kWL1 * ml;
  };

  // ---- target statistics and identity errors over the error-map region
  for (int q = tid; q < Q; q += NT) {
    const int qy = q / QW, qx = q - qy * QW;
    const int pos = (qy + 1) * RW + (qx + 1);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      s_ystat[2 * c][q] = box3_sum(s_tgt[c], pos, RW) * kNinth;
      s_ystat[2 * c + 1][q] = box3_sum_prod(s_tgt[c], s_tgt[c], pos, RW) * kNinth;
    }
  }
  __syncthreads();
  for (int q = tid; q < Q; q += NT) {
    const int qy = q / QW, qx = q - qy * QW;
    const int pos = (qy + 1) * RW + (qx + 1);
    const int y = y00 + 1 + qy, x = x00 + 1 + qx;
    float m = 0.f;
    unsigned char mi = 0;
    if (y >= 0 && y < H && x >= 0 && x < W) {
      float e0 = photo_error(0, pos, q, nullptr, nullptr);
      float e1 = photo_error(1, pos, q, nullptr, nullptr);
      if (p.tie) {
        const long o = (long)y * W + x;
        e0 = e0 + p.tie[((long)b * DW_NUM_SRC + 0) * plane + o];
        e1 = e1 + p.tie[((long)b * DW_NUM_SRC + 1) * plane + o];
      }
      m = e0;
      if (e1 < m) {
        m = e1;
        mi = 1;
      }
    }
    s_idm[q] = m;
    s_idi[q] = mi;
  }
  __syncthreads();

  const long tslot = (long)b * p.tilesX * p.tilesY + tile;
  float dP[2][12];
#pragma unroll
  for (int f = 0; f < 2; ++f)
#pragma unroll
    for (int k = 0; k < 12; ++k) dP[f][k] = 0.f;

  for (int s = 0; s < p.S; ++s) {
    const int hs = H >> s, ws = W >> s;
    const float sy = (float)hs / (float)H, sx = (float)ws / (float)W;
    const float* disp_b = p.disp[s] + (long)b * hs * ws;
    float loss_acc = 0.f;

    // ---- stage A: warp both sources over the colour region
    for (int i = tid; i < R; i += NT) {
      const int ry = i / RW, rx = i - ry * RW;
      const int y = reflect_index(y00 + ry, H), x = reflect_index(x00 + rx, W);
      const float du = upsampled_disp(disp_b, hs, ws, s, y, x, sy, sx);
      const float depth = 1.0f / (kMinDisp + kDispRange * du);
      const float fx = (float)x, fy = (float)y;
      const float X = depth * ((s_invK[0] * fx + s_invK[1] * fy) + s_invK[2]);
      const float Y = depth * ((s_invK[3] * fx + s_invK[4] * fy) + s_invK[5]);
      const float Z = depth * ((s_invK[6] * fx + s_invK[7] * fy) + s_invK[8]);
      unsigned char vb = 0;
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        Sample sm;
        project_sample(src_b[f], plane, H, W, s_P[f], X, Y, Z, sm);
        s_xw[f][0][i] = sm.val[0];
        s_xw[f][1][i] = sm.val[1];
        s_xw[f][2][i] = sm.val[2];
        vb |= (sm.valid ? 1 : 0) << (2 + f);
      }
      s_valid[i] = vb;
    }
    __syncthreads();

    // ---- stage B: errors, minimum, mask; SSIM adjoint terms of the winner
    for (int q = tid; q < Q; q += NT) {
      const int qy = q / QW, qx = q - qy * QW;
      const int pos = (qy + 1) * RW + (qx + 1);
      const int y = y00 + 1 + qy, x = x00 + 1 + qx;
      const bool inside = (y >= 0 && y < H && x >= 0 && x < W);
      signed char sel = -1;
      float g[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (inside) {
        SsimTerms t0[3], t1[3];
        float mx0[3], mx1[3];
        const float e0 = photo_error(0, pos, q, GRAD ? t0 : nullptr, mx0);
        const float e1 = photo_error(1, pos, q, GRAD ? t1 : nullptr, mx1);
        float m = s_idm[q];
        int idx = s_idi[q];
        if (e0 < m) {
          m = e0;
          idx = 2;
        }
        if (e1 < m) {
          m = e1;
          idx = 3;
        }
        const bool core = (qy >= HALO - 1) && (qy < HALO - 1 + TH) && (qx >= HALO - 1) &&
                          (qx < HALO - 1 + TW);
        if (core) {
          loss_acc += m;
          if (p.mask[s]) p.mask[s][(long)b * plane + (long)y * W + x] = (uint8_t)(idx | s_valid[pos]);
        }
        if (GRAD && idx >= 2) {
          sel = (signed char)(idx - 2);
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const SsimTerms t = (idx == 2) ? t0[c] : t1[c];
            const float mux = (idx == 2) ? mx0[c] : mx1[c];
            const float muy = s_ystat[2 * c][q];
            if (t.raw >= 0.f && t.raw <= 1.f) {
              const float d = t.B1 * t.B2;
              const float r = (t.A1 * t.A2) / d;
              const float inv_d = 1.0f / d;
              // upstream: 0.85/3 from the channel mean, -1/2 from (1 - r)/2
              const float up = -0.5f * kWSsim * kThird * inv_d;
              g[3 * c + 0] = up * (2.0f * muy * (t.A2 - t.A1) - r * 2.0f * mux * (t.B2 - t.B1));
              g[3 * c + 1] = up * (-r * t.B1);
              g[3 * c + 2] = up * (2.0f * t.A1);
            }
          }
        }
      }
      if (GRAD) {
        s_sel[q] = sel;
#pragma unroll
        for (int k = 0; k < 9; ++k) s_G[k][q] = g[k];
      }
    }
    __syncthreads();

    // ---- stage C: adjoint of window sums, L1, warp, projection, depth
    if (GRAD) {
      for (int i = tid; i < TH * TW; i += NT) {
        const int cy = i / TW, cx = i - cy * TW;
        const int y = y00 + HALO + cy, x = x00 + HALO + cx;
        if (y >= H || x >= W) continue;
        const int q0 = (cy + HALO - 1) * QW + (cx + HALO - 1);  // this pixel in error-map coords
        const int pos = (cy + HALO) * RW + (cx + HALO);         // and in region coords
        float acc[2][9];
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
          for (int k = 0; k < 9; ++k) acc[f][k] = 0.f;
        bool any = false;
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy) {
          // a window row that was reflected onto this pixel's row counts twice
          const float wy = 1.f + ((dy == -1 && y == 1) ? 1.f : 0.f) +
                           ((dy == 1 && y == H - 2) ? 1.f : 0.f);
#pragma unroll
          for (int dx = -1; dx <= 1; ++dx) {
            const float wx = 1.f + ((dx == -1 && x == 1) ? 1.f : 0.f) +
                             ((dx == 1 && x == W - 2) ? 1.f : 0.f);
            const int q = q0 + dy * QW + dx;
            const int sel = s_sel[q];
            if (sel < 0) continue;
            any = true;
            const float w = wy * wx;
#pragma unroll
            for (int k = 0; k < 9; ++k) {
              const float gk = s_G[k][q] * w;
              if (sel == 0) acc[0][k] += gk; else acc[1][k] += gk;
            }
          }
        }
        float gd = 0.f;
        if (any) {
          const float du = upsampled_disp(disp_b, hs, ws, s, y, x, sy, sx);
          const float depth = 1.0f / (kMinDisp + kDispRange * du);
          const float fx = (float)x, fy = (float)y;
          const float r0 = (s_invK[0] * fx + s_invK[1] * fy) + s_invK[2];
          const float r1 = (s_invK[3] * fx + s_invK[4] * fy) + s_invK[5];
          const float r2 = (s_invK[6] * fx + s_invK[7] * fy) + s_invK[8];
          const float X = depth * r0, Y = depth * r1, Z = depth * r2;
          const int my_sel = s_sel[q0];
          float gdepth = 0.f;
#pragma unroll
          for (int f = 0; f < 2; ++f) {
            Sample sm;
            project_sample(src_b[f], plane, H, W, s_P[f], X, Y, Z, sm);
            float gix = 0.f, giy = 0.f;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
              const float xq = s_xw[f][c][pos], yq = s_tgt[c][pos];
              float gx = kNinth * (acc[f][3 * c] + 2.0f * xq * acc[f][3 * c + 1] +
                                   yq * acc[f][3 * c + 2]);
              if (my_sel == f) {
                const float df = xq - yq;
                gx += kWL1 * kThird * ((df > 0.f) ? 1.f : ((df < 0.f) ? -1.f : 0.f));
              }
              const float* pc = src_b[f] + c * plane;
              const float nw = pc[sm.o_nw], ne = pc[sm.o_ne], sw = pc[sm.o_sw], se = pc[sm.o_se];
              gix = fmaf(gx, (ne - nw) * (1.f - sm.ty) + (se - sw) * sm.ty, gix);
              giy = fmaf(gx, (sw - nw) * (1.f - sm.tx) + (se - ne) * sm.tx, giy);
            }
            const float gu = sm.inx ? gix : 0.f;
            const float gv = sm.iny ? giy : 0.f;
            const float iz = 1.0f / sm.zp;
            const float gc0 = gu * iz, gc1 = gv * iz;
            const float gc2 = -(gu * sm.u + gv * sm.v) * iz;
            const float* P = s_P[f];
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
            gdepth += gX * r0 + gY * r1 + gZ * r2;
          }
          gd = gdepth * (-kDispRange * depth * depth);
        }
        p.gdup[((long)s * p.B + b) * plane + (long)y * W + x] = gd;
      }
    }
    // per-tile partial sum (fixed order: reproducible run to run); its
    // barriers also separate this scale's LDS reads from the next stage A
    const float t = block_sum<NT>(loss_acc, s_red);
    if (tid == 0) p.tile_loss[tslot * DW_MAX_SCALES + s] = t;
  }

  if (GRAD) {
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int k = 0; k < 12; ++k) {
        const float t = block_sum<NT>(dP[f][k], s_red);
        if (tid == 0) p.tile_dP[tslot * 24 + f * 12 + k] = t;
      }
  }
}

// ------------------------------------------------- per-image disparity means
__global__ __launch_bounds__(1024) void disp_mean_kernel(const PhotoParams p) {
  __shared__ float red[1024 / kWave];
  const int s = blockIdx.y, b = blockIdx.x;
  if (s >= p.S) return;
  const long n = (long)p.L.Hs[s] * p.L.Ws[s];
  const float* d = p.disp[s] + (long)b * n;
  float acc = 0.f;
  for (long i = threadIdx.x; i < n; i += 1024) acc += d[i];
  const float t = block_sum<1024>(acc, red);
  if (threadIdx.x == 0) p.mean[s * p.B + b] = t / (float)n;
}

// Decode a block of the per-scale map kernels: scale, sample, first pixel.
__device__ __forceinline__ void decode_map_block(const ScaleLayout& L, int S, int B, int blk, int& s,
                                                 int& b, int& chunk) {
  s = 0;
  while (s + 1 < S && blk >= L.blk_off[s + 1]) ++s;
  const int r = blk - L.blk_off[s];
  b = r / L.nblk[s];
  chunk = r - b * L.nblk[s];
}

// ------------------------------------------------- edge-aware smoothness
// loss_s = mean_x(|d^(x) - d^(x+1)| * exp(-mean_c|I(x) - I(x+1)|)) + same in y,
// d^ = disp / (mean + 1e-7).  Writes G = d loss_s / d d^ per pixel and per-block
// partials (sum_x, sum_y, sum G*disp).
template <bool GRAD>
__global__ __launch_bounds__(SM_NT) void smooth_kernel(const PhotoParams p) {
  __shared__ float red[SM_NT / kWave];
  int s, b, chunk;
  decode_map_block(p.L, p.S, p.B, blockIdx.x, s, b, chunk);
  const int hs = p.L.Hs[s], ws = p.L.Ws[s];
  const long n = (long)hs * ws;
  const float* d = p.disp[s] + (long)b * n;
  const float* I = p.pyr[s] + (long)b * 3 * n;
  const float inv = 1.0f / (p.mean[s * p.B + b] + kMeanEps);
  const float icx = 1.0f / ((float)p.B * hs * (ws - 1));
  const float icy = 1.0f / ((float)p.B * (hs - 1) * ws);
  float lx = 0.f, ly = 0.f, gd = 0.f;
  for (int k = threadIdx.x; k < SM_PX; k += SM_NT) {
    const long i = (long)chunk * SM_PX + k;
    if (i >= n) break;
    const int y = (int)(i / ws), x = (int)(i - (long)y * ws);
    const float dc = d[i] * inv;
    const float c0 = I[i], c1 = I[n + i], c2 = I[2 * n + i];
    float G = 0.f;
    auto edge = [&](long j) -> float {  // exp(-mean_c |I(i) - I(j)|)
      const float m = (fabsf(c0 - I[j]) + fabsf(c1 - I[n + j]) + fabsf(c2 - I[2 * n + j])) * kThird;
      return __expf(-m);
    };
    auto sgn = [](float a) -> float { return (a > 0.f) ? 1.f : ((a < 0.f) ? -1.f : 0.f); };
    if (x + 1 < ws) {
      const float a = dc - d[i + 1] * inv, w = edge(i + 1);
      lx += fabsf(a) * w;
      G += sgn(a) * w * icx;
    }
    if (x > 0) {
      const float a = d[i - 1] * inv - dc, w = edge(i - 1);
      G -= sgn(a) * w * icx;
    }
    if (y + 1 < hs) {
      const float a = dc - d[i + ws] * inv, w = edge(i + ws);
      ly += fabsf(a) * w;
      G += sgn(a) * w * icy;
    }
    if (y > 0) {
      const float a = d[i - ws] * inv - dc, w = edge(i - ws);
      G -= sgn(a) * w * icy;
    }
    if (GRAD) {
      p.sm_G[p.L.pix_off[s] + (long)b * n + i] = G;
      gd = fmaf(G, d[i], gd);
    }
  }
  float* out = p.sm_part + (long)blockIdx.x * 3;
  float t = block_sum<SM_NT>(lx, red);
  if (threadIdx.x == 0) out[0] = t;
  t = block_sum<SM_NT>(ly, red);
  if (threadIdx.x == 0) out[1] = t;
  if (GRAD) {
    t = block_sum<SM_NT>(gd, red);
    if (threadIdx.x == 0) out[2] = t;
  }
}

// ------------------------------------------------- scalar loss
__global__ __launch_bounds__(256) void loss_finalize_kernel(const PhotoParams p, float* loss_out) {
  __shared__ double red[256];
  const int tid = threadIdx.x;
  const int ntile = p.B * p.tilesX * p.tilesY;
  double total = 0.0;
  for (int s = 0; s < p.S; ++s) {
    double a = 0.0;
    for (int t = tid; t < ntile; t += 256) a += (double)p.tile_loss[(long)t * DW_MAX_SCALES + s];
    red[tid] = a;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (tid < o) red[tid] += red[tid + o];
      __syncthreads();
    }
    const double photo = red[0] / ((double)p.B * p.H * p.W);
    __syncthreads();
    double ax = 0.0, ay = 0.0;
    for (int k = p.L.blk_off[s] + tid; k < p.L.blk_off[s + 1]; k += 256) {
      ax += (double)p.sm_part[(long)k * 3 + 0];
      ay += (double)p.sm_part[(long)k * 3 + 1];
    }
    red[tid] = ax;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (tid < o) red[tid] += red[tid + o];
      __syncthreads();
    }
    const double sumx = red[0];
    __syncthreads();
    red[tid] = ay;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (tid < o) red[tid] += red[tid + o];
      __syncthreads();
    }
    const double sumy = red[0];
    __syncthreads();
    const int hs = p.L.Hs[s], ws = p.L.Ws[s];
    const double sm = sumx / ((double)p.B * hs * (ws - 1)) + sumy / ((double)p.B * (hs - 1) * ws);
    if (tid == 0) {
      loss_out[1 + s] = (float)photo;
      loss_out[1 + p.S + s] = (float)sm;
    }
    total += photo + (double)kWSmooth * sm / (double)(1 << s);
  }
  if (tid == 0) loss_out[0] = (float)(total / p.S);
}

// ------------------------------------------------- backward: d loss / d disp[s]
struct BwdParams {
  int B, H, W, S;
  int tilesX, tilesY;
  const float* disp[DW_MAX_SCALES];
  const float* K;
  const float* grad_loss;
  const float* mean;
  const float* tile_dP;
  const float* gdup;
  const float* sm_G;
  const float* sm_part;
  float* grad_disp[DW_MAX_SCALES];
  float* grad_T;
  ScaleLayout L;
};

__global__ __launch_bounds__(SM_NT) void grad_disp_kernel(const BwdParams p) {
  __shared__ float s_gd;
  __shared__ float red[SM_NT / kWave];
  int s, b, chunk;
  decode_map_block(p.L, p.S, p.B, blockIdx.x, s, b, chunk);
  const int hs = p.L.Hs[s], ws = p.L.Ws[s];
  const long n = (long)hs * ws;
  const int H = p.H, W = p.W;
  const long plane = (long)H * W;
  // sum_i G_i * disp_i over the whole image, from the per-block partials
  {
    float a = 0.f;
    const long base = (long)p.L.blk_off[s] + (long)b * p.L.nblk[s];
    for (int k = threadIdx.x; k < p.L.nblk[s]; k += SM_NT) a += p.sm_part[(base + k) * 3 + 2];
    const float t = block_sum<SM_NT>(a, red);
    if (threadIdx.x == 0) s_gd = t;
    __syncthreads();
  }
  const float go = p.grad_loss[0];
  const float inv = 1.0f / (p.mean[s * p.B + b] + kMeanEps);
  const float coef = go * (kWSmooth / (float)(1 << s)) / (float)p.S;
  const float g0 = go / ((float)p.S * (float)p.B * (float)plane);
  const float mean_term = s_gd * inv * inv / (float)n;
  const float* gdup = p.gdup + ((long)s * p.B + b) * plane;
  const float* G = p.sm_G + p.L.pix_off[s] + (long)b * n;
  float* out = p.grad_disp[s] + (long)b * n;
  const int k2 = 1 << s;
  const float sy = (float)hs / (float)H, sx = (float)ws / (float)W;
  for (int k = threadIdx.x; k < SM_PX; k += SM_NT) {
    const long i = (long)chunk * SM_PX + k;
    if (i >= n) break;
    const int y = (int)(i / ws), x = (int)(i - (long)y * ws);
    float ph = 0.f;
    if (s == 0) {
      ph = gdup[i];
    } else {
      // adjoint of the bilinear up-sampling in gather form: every full-res
      // pixel one of whose taps is (y, x)
      const int Y0 = max(0, (y - 1) * k2 + k2 / 2), Y1 = min(H - 1, (y + 1) * k2 + k2 / 2 - 1);
      const int X0 = max(0, (x - 1) * k2 + k2 / 2), X1 = min(W - 1, (x + 1) * k2 + k2 / 2 - 1);
      for (int Y = Y0; Y <= Y1; ++Y) {
        const Taps ty = upsample_taps(Y, hs, sy);
        const float wy = (ty.i0 == y ? ty.l0 : 0.f) + (ty.i1 == y ? ty.l1 : 0.f);
        if (wy == 0.f) continue;
        float row = 0.f;
        for (int X = X0; X <= X1; ++X) {
          const Taps tx = upsample_taps(X, ws, sx);
          const float wx = (tx.i0 == x ? tx.l0 : 0.f) + (tx.i1 == x ? tx.l1 : 0.f);
          row = fmaf(wx, gdup[(long)Y * W + X], row);
        }
        ph = fmaf(wy, row, ph);
      }
    }
    out[i] = coef * (G[i] * inv - mean_term) + g0 * ph;
  }
}

// ------------------------------------------------- backward: d loss / d T
__global__ __launch_bounds__(64) void grad_pose_kernel(const BwdParams p) {
  __shared__ double s_dP[24];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int ntile = p.tilesX * p.tilesY;
  if (tid < 24) {
    double a = 0.0;
    const float* src = p.tile_dP + (long)b * ntile * 24 + tid;
    for (int t = 0; t < ntile; ++t) a += (double)src[(long)t * 24];
    s_dP[tid] = a;
  }
  __syncthreads();
  if (tid < 32) {
    const int f = tid >> 4, k = (tid >> 2) & 3, j = tid & 3;
    const float* Kb = p.K + (long)b * 16;
    double a = 0.0;
    for (int i = 0; i < 3; ++i) a += (double)Kb[i * 4 + k] * s_dP[f * 12 + i * 4 + j];
    const double g0 = (double)p.grad_loss[0] / ((double)p.S * p.B * p.H * p.W);
    p.grad_T[((long)b * DW_NUM_SRC + f) * 16 + k * 4 + j] = (float)(a * g0);
  }
}

// ------------------------------------------------------------- host side
struct WsLayout {
  size_t mean, tile_loss, tile_dP, gdup, sm_G, sm_part, total;
};

static void make_layout(const dw_desc& d, ScaleLayout& L, WsLayout& w, int& tilesX, int& tilesY) {
  tilesX = (d.W + TW - 1) / TW;
  tilesY = (d.H + TH - 1) / TH;
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
  w.mean = o;
  o = align_up(o + sizeof(float) * DW_MAX_SCALES * d.B, A);
  w.tile_loss = o;
  o = align_up(o + sizeof(float) * (size_t)d.B * tilesX * tilesY * DW_MAX_SCALES, A);
  w.tile_dP = o;
  o = align_up(o + sizeof(float) * (size_t)d.B * tilesX * tilesY * 24, A);
  w.sm_part = o;
  o = align_up(o + sizeof(float) * (size_t)L.blk_off[d.S] * 3, A);
  w.sm_G = o;
  o = align_up(o + sizeof(float) * (size_t)L.pix_off[d.S], A);
  w.gdup = o;
  o = align_up(o + sizeof(float) * (size_t)d.S * d.B * d.H * d.W, A);
  w.total = o;
}

static int check_desc(const dw_desc* d) {
  if (!d) return DW_ERR_NULL;
  if (d->B < 1 || d->S < 1 || d->S > DW_MAX_SCALES || d->F != DW_NUM_SRC) return DW_ERR_SHAPE;
  if (d->flags & ~DW_FLAG_TILE_KERNEL_ONLY) return DW_ERR_ARG;
  const int m = 1 << (d->S - 1);
  if (d->H % m || d->W % m) return DW_ERR_SHAPE;
  if ((d->H >> (d->S - 1)) < 2 || (d->W >> (d->S - 1)) < 2) return DW_ERR_SHAPE;
  if ((long)d->H * d->W > (1l << 30) || d->B > 65535) return DW_ERR_SHAPE;
  return DW_OK;
}

}  // namespace dw

using namespace dw;

extern "C" size_t dw_photometric_workspace_bytes(const dw_desc* desc) {
  if (check_desc(desc) != DW_OK) return 0;
  ScaleLayout L;
  WsLayout w;
  int tx, ty;
  make_layout(*desc, L, w, tx, ty);
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
  make_layout(*desc, p.L, w, p.tilesX, p.tilesY);
  if (workspace_bytes < w.total || ((uintptr_t)workspace & 255)) return DW_ERR_WORKSPACE;
  p.B = desc->B;
  p.H = desc->H;
  p.W = desc->W;
  p.S = desc->S;
  for (int s = 0; s < desc->S; ++s) {
    if (!disp[s] || !color_pyr[s]) return DW_ERR_NULL;
    p.disp[s] = disp[s];
    p.pyr[s] = color_pyr[s];
    p.mask[s] = mask ? mask[s] : nullptr;
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
  p.mean = (float*)(ws + w.mean);
  p.tile_loss = (float*)(ws + w.tile_loss);
  p.tile_dP = (float*)(ws + w.tile_dP);
  p.gdup = (float*)(ws + w.gdup);
  p.sm_G = (float*)(ws + w.sm_G);
  p.sm_part = (float*)(ws + w.sm_part);
  hipStream_t st = (hipStream_t)stream;

  const dim3 tgrid(p.tilesX * p.tilesY, p.B);
  if (desc->flags & DW_FLAG_TILE_KERNEL_ONLY) {
    if (with_grad)
      photo_tile_kernel<true><<<tgrid, NT, 0, st>>>(p);
    else
      photo_tile_kernel<false><<<tgrid, NT, 0, st>>>(p);
    return hipGetLastError() == hipSuccess ? DW_OK : DW_ERR_LAUNCH;
  }
  disp_mean_kernel<<<dim3(p.B, p.S), 1024, 0, st>>>(p);
  if (with_grad) {
    photo_tile_kernel<true><<<tgrid, NT, 0, st>>>(p);
    smooth_kernel<true><<<p.L.blk_off[p.S], SM_NT, 0, st>>>(p);
  } else {
    photo_tile_kernel<false><<<tgrid, NT, 0, st>>>(p);
    smooth_kernel<false><<<p.L.blk_off[p.S], SM_NT, 0, st>>>(p);
  }
  loss_finalize_kernel<<<1, 256, 0, st>>>(p, loss_out);
  return hipGetLastError() == hipSuccess ? DW_OK : DW_ERR_LAUNCH;
}

extern "C" int dw_photometric_bwd(const dw_desc* desc, const float* const* disp, const float* K,
                                  const float* grad_loss, const void* workspace,
                                  size_t workspace_bytes, float* const* grad_disp, float* grad_T,
                                  void* stream) {
  int rc = check_desc(desc);
  if (rc != DW_OK) return rc;
  if (!disp || !K || !grad_loss || !workspace || !grad_disp) return DW_ERR_NULL;
  BwdParams p{};
  WsLayout w;
  make_layout(*desc, p.L, w, p.tilesX, p.tilesY);
  if (workspace_bytes < w.total || ((uintptr_t)workspace & 255)) return DW_ERR_WORKSPACE;
  p.B = desc->B;
  p.H = desc->H;
  p.W = desc->W;
  p.S = desc->S;
  for (int s = 0; s < desc->S; ++s) {
    if (!disp[s] || !grad_disp[s]) return DW_ERR_NULL;
    p.disp[s] = disp[s];
    p.grad_disp[s] = grad_disp[s];
  }
  p.K = K;
  p.grad_loss = grad_loss;
  const char* ws = (const char*)workspace;
  p.mean = (const float*)(ws + w.mean);
  p.tile_dP = (const float*)(ws + w.tile_dP);
  p.gdup = (const float*)(ws + w.gdup);
  p.sm_G = (const float*)(ws + w.sm_G);
  p.sm_part = (const float*)(ws + w.sm_part);
  p.grad_T = grad_T;
  hipStream_t st = (hipStream_t)stream;
  grad_disp_kernel<<<p.L.blk_off[p.S], SM_NT, 0, st>>>(p);
  if (grad_T) grad_pose_kernel<<<p.B, 64, 0, st>>>(p);
  return hipGetLastError() == hipSuccess ? DW_OK : DW_ERR_LAUNCH;
}
