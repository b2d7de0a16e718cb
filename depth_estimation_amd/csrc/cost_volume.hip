// Cost-volume construction (shift + concatenate), forward and backward.
//
// Reference: DepthBlock.cost_volume, /root/reference/model/model.py:460-508 --
// a 64-iteration Python loop of zeros / cat / slice / cat, then stack, about
// ten launches and five temporaries per disparity.  Here one kernel writes
// both (B, D, 4F, h, w) volumes exactly once; the four (B, F, h, w) feature
// maps are a few MB and stay cache resident, so the pass is bounded by the
// HBM write stream.  Channel blocks, as in the reference:
//   d == 0 : left  = [imgl, carl, imgr, carr]        right = [imgr, carr, imgl, carl]
//   d  > 0 : left  = [imgl, R_d(imgr), carl, R_d(carr)]
//            right = [imgr, L_d(imgl), carr, L_d(carl)]
// with R_d(a)[x] = a[x - d] (0 for x < d) and L_d(a)[x] = a[x + d] (0 for
// x + d >= w).
//
// A thread owns VEC consecutive pixels of one (sample, feature channel, row)
// and loops over a chunk of disparities: the un-shifted values stay in
// registers, the shifted ones are re-read (unaligned vector loads that hit
// cache), and every disparity costs eight vector stores.  Backward is the
// mirror image: the same thread sums, in ascending disparity order, the eight
// places each of its input pixels was copied to (fixed order: reproducible).
#include "dw_common.h"

namespace dw {

constexpr int CV_NT = 256;
constexpr int CV_DCHUNK = 8;   // disparities per block in the forward pass

typedef float cv_f4 __attribute__((ext_vector_type(4)));
typedef cv_f4 cv_f4u __attribute__((aligned(4)));  // same vector, only dword aligned

template <int VEC>
struct CvVec;
template <>
struct CvVec<1> {
  using aligned = float;
  using loose = float;
};
template <>
struct CvVec<4> {
  using aligned = cv_f4;
  using loose = cv_f4u;
};

template <int VEC>
__device__ __forceinline__ void cv_load(const float* p, float* v) {
  using T = typename CvVec<VEC>::aligned;
  const T t = *reinterpret_cast<const T*>(p);
  const float* f = reinterpret_cast<const float*>(&t);
#pragma unroll
  for (int i = 0; i < VEC; ++i) v[i] = f[i];
}
// streaming variant for data that is read exactly once
template <int VEC>
__device__ __forceinline__ void cv_load_nt(const float* p, float* v) {
  using T = typename CvVec<VEC>::aligned;
  const T t = __builtin_nontemporal_load(reinterpret_cast<const T*>(p));
  const float* f = reinterpret_cast<const float*>(&t);
#pragma unroll
  for (int i = 0; i < VEC; ++i) v[i] = f[i];
}
template <int VEC>
__device__ __forceinline__ void cv_store_nt(float* p, const float* v) {
  using T = typename CvVec<VEC>::aligned;
  T t;
  float* f = reinterpret_cast<float*>(&t);
#pragma unroll
  for (int i = 0; i < VEC; ++i) f[i] = v[i];
  __builtin_nontemporal_store(t, reinterpret_cast<T*>(p));
}

// VEC values of row `row` (w floats) starting at column x (any alignment);
// columns outside [0, w) read as zero.
template <int VEC, bool STREAM>
__device__ __forceinline__ void cv_load_shifted(const float* row, int x, int w, float* v) {
  if (x >= 0 && x + VEC <= w) {
    using T = typename CvVec<VEC>::loose;
    const T t = STREAM ? __builtin_nontemporal_load(reinterpret_cast<const T*>(row + x))
                       : *reinterpret_cast<const T*>(row + x);
    const float* f = reinterpret_cast<const float*>(&t);
#pragma unroll
    for (int i = 0; i < VEC; ++i) v[i] = f[i];
  } else {
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      const int xi = x + i;
      v[i] = (xi >= 0 && xi < w) ? row[xi] : 0.f;
    }
  }
}

struct CvParams {
  int F, h, w, D;
  const float* in[4];   // imgl, imgr, carl, carr            (forward inputs)
  float* vol[2];        // left, right                        (forward outputs)
  const float* gvol[2]; // gradients of left, right           (backward inputs)
  float* gin[4];        // gradients of imgl, imgr, carl, carr (backward outputs)
};

// grid: (ceil(h*w/VEC / CV_NT), F * ceil(D / CV_DCHUNK), B)
template <int VEC>
__global__ __launch_bounds__(CV_NT) void cost_volume_fwd_kernel(const CvParams p) {
  const int F = p.F, w = p.w, D = p.D;
  const long n = (long)p.h * w;
  const long i0 = ((long)blockIdx.x * CV_NT + threadIdx.x) * VEC;  // first pixel of this thread
  if (i0 >= n) return;
  const int c = blockIdx.y % F, dz = blockIdx.y / F;
  const int b = blockIdx.z;
  const int y = (int)(i0 / w), x0 = (int)(i0 - (long)y * w);
  const long in_off = ((long)b * F + c) * n;
  const float* rowl = p.in[0] + in_off + (long)y * w;
  const float* rowr = p.in[1] + in_off + (long)y * w;
  const float* rowcl = p.in[2] + in_off + (long)y * w;
  const float* rowcr = p.in[3] + in_off + (long)y * w;
  float vl[VEC], vr[VEC], vcl[VEC], vcr[VEC];
  cv_load<VEC>(rowl + x0, vl);
  cv_load<VEC>(rowr + x0, vr);
  cv_load<VEC>(rowcl + x0, vcl);
  cv_load<VEC>(rowcr + x0, vcr);
  const long blk = (long)F * n;   // one channel block of one disparity
  const int d_lo = dz * CV_DCHUNK, d_hi = min(d_lo + CV_DCHUNK, D);
  for (int d = d_lo; d < d_hi; ++d) {
    const long o = (((long)b * D + d) * 4 * F + c) * n + i0;
    float* L = p.vol[0] + o;
    float* R = p.vol[1] + o;
    if (d == 0) {
      cv_store_nt<VEC>(L, vl);
      cv_store_nt<VEC>(L + blk, vcl);
      cv_store_nt<VEC>(L + 2 * blk, vr);
      cv_store_nt<VEC>(L + 3 * blk, vcr);
      cv_store_nt<VEC>(R, vr);
      cv_store_nt<VEC>(R + blk, vcr);
      cv_store_nt<VEC>(R + 2 * blk, vl);
      cv_store_nt<VEC>(R + 3 * blk, vcl);
    } else {
      float sr[VEC], scr[VEC], sl[VEC], scl[VEC];
      cv_load_shifted<VEC, false>(rowr, x0 - d, w, sr);
      cv_load_shifted<VEC, false>(rowcr, x0 - d, w, scr);
      cv_load_shifted<VEC, false>(rowl, x0 + d, w, sl);
      cv_load_shifted<VEC, false>(rowcl, x0 + d, w, scl);
      cv_store_nt<VEC>(L, vl);
      cv_store_nt<VEC>(L + blk, sr);
      cv_store_nt<VEC>(L + 2 * blk, vcl);
      cv_store_nt<VEC>(L + 3 * blk, scr);
      cv_store_nt<VEC>(R, vr);
      cv_store_nt<VEC>(R + blk, sl);
      cv_store_nt<VEC>(R + 2 * blk, vcr);
      cv_store_nt<VEC>(R + 3 * blk, scl);
    }
  }
}

// grid: (ceil(h*w/VEC / CV_NT), F, B)
template <int VEC>
__global__ __launch_bounds__(CV_NT) void cost_volume_bwd_kernel(const CvParams p) {
  const int F = p.F, w = p.w, D = p.D;
  const long n = (long)p.h * w;
  const long i0 = ((long)blockIdx.x * CV_NT + threadIdx.x) * VEC;
  if (i0 >= n) return;
  const int c = blockIdx.y;
  const int b = blockIdx.z;
  const int y = (int)(i0 / w), x0 = (int)(i0 - (long)y * w);
  const long blk = (long)F * n;
  float al[VEC], ar[VEC], acl[VEC], acr[VEC];
  {
    // d = 0: left = [imgl, carl, imgr, carr], right = [imgr, carr, imgl, carl]
    const long o = (((long)b * D) * 4 * F + c) * n + i0;
    const float* L = p.gvol[0] + o;
    const float* R = p.gvol[1] + o;
    float t[VEC];
    cv_load_nt<VEC>(L, al);
    cv_load_nt<VEC>(R + 2 * blk, t);
#pragma unroll
    for (int i = 0; i < VEC; ++i) al[i] += t[i];
    cv_load_nt<VEC>(L + blk, acl);
    cv_load_nt<VEC>(R + 3 * blk, t);
#pragma unroll
    for (int i = 0; i < VEC; ++i) acl[i] += t[i];
    cv_load_nt<VEC>(L + 2 * blk, ar);
    cv_load_nt<VEC>(R, t);
#pragma unroll
    for (int i = 0; i < VEC; ++i) ar[i] += t[i];
    cv_load_nt<VEC>(L + 3 * blk, acr);
    cv_load_nt<VEC>(R + blk, t);
#pragma unroll
    for (int i = 0; i < VEC; ++i) acr[i] += t[i];
  }
#pragma unroll 2
  for (int d = 1; d < D; ++d) {
    // left  = [imgl, R_d(imgr), carl, R_d(carr)]: imgr[x] was copied to column x + d
    // right = [imgr, L_d(imgl), carr, L_d(carl)]: imgl[x] was copied to column x - d
    const long o = (((long)b * D + d) * 4 * F + c) * n + (long)y * w;
    const float* L = p.gvol[0] + o;
    const float* R = p.gvol[1] + o;
    float t0[VEC], t1[VEC], t2[VEC], t3[VEC], t4[VEC], t5[VEC], t6[VEC], t7[VEC];
    cv_load_nt<VEC>(L + x0, t0);                                 // -> imgl
    cv_load_shifted<VEC, true>(R + blk, x0 - d, w, t1);          // -> imgl
    cv_load_nt<VEC>(L + 2 * blk + x0, t2);                       // -> carl
    cv_load_shifted<VEC, true>(R + 3 * blk, x0 - d, w, t3);      // -> carl
    cv_load_nt<VEC>(R + x0, t4);                                 // -> imgr
    cv_load_shifted<VEC, true>(L + blk, x0 + d, w, t5);          // -> imgr
    cv_load_nt<VEC>(R + 2 * blk + x0, t6);                       // -> carr
    cv_load_shifted<VEC, true>(L + 3 * blk, x0 + d, w, t7);      // -> carr
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      al[i] = (al[i] + t0[i]) + t1[i];
      acl[i] = (acl[i] + t2[i]) + t3[i];
      ar[i] = (ar[i] + t4[i]) + t5[i];
      acr[i] = (acr[i] + t6[i]) + t7[i];
    }
  }
  const long in_off = ((long)b * F + c) * n + i0;
  using T = typename CvVec<VEC>::aligned;
  auto put = [&](float* dst, const float* v) __attribute__((always_inline)) {
    T t;
    float* f = reinterpret_cast<float*>(&t);
#pragma unroll
    for (int i = 0; i < VEC; ++i) f[i] = v[i];
    *reinterpret_cast<T*>(dst + in_off) = t;
  };
  put(p.gin[0], al);
  put(p.gin[1], ar);
  put(p.gin[2], acl);
  put(p.gin[3], acr);
}

static bool cv_aligned16(const void* a) { return ((uintptr_t)a & 15) == 0; }

}  // namespace dw

using namespace dw;

static int cv_check(int B, int F, int h, int w, int D) {
  if (B < 1 || B > 65535 || F < 1 || h < 1 || w < 1 || D < 1) return DW_ERR_SHAPE;
  if ((long)F * ((D + CV_DCHUNK - 1) / CV_DCHUNK) > 65535) return DW_ERR_SHAPE;
  if ((long)h * w > (1L << 30)) return DW_ERR_SHAPE;
  return DW_OK;
}

extern "C" int dw_cost_volume_fwd(int32_t B, int32_t F, int32_t h, int32_t w, int32_t D,
                                  const float* imgl, const float* imgr, const float* carl,
                                  const float* carr, float* left, float* right, void* stream) {
  if (!imgl || !imgr || !carl || !carr || !left || !right) return DW_ERR_NULL;
  if (const int rc = cv_check(B, F, h, w, D)) return rc;
  CvParams p{};
  p.F = F; p.h = h; p.w = w; p.D = D;
  p.in[0] = imgl; p.in[1] = imgr; p.in[2] = carl; p.in[3] = carr;
  p.vol[0] = left; p.vol[1] = right;
  const long n = (long)h * w;
  const bool vec4 = (w % 4 == 0) && cv_aligned16(imgl) && cv_aligned16(imgr) && cv_aligned16(carl) &&
                    cv_aligned16(carr) && cv_aligned16(left) && cv_aligned16(right);
  const int V = vec4 ? 4 : 1;
  const dim3 grid((unsigned)((n / V + CV_NT - 1) / CV_NT), (unsigned)(F * ((D + CV_DCHUNK - 1) / CV_DCHUNK)), B);
  if (vec4)
    cost_volume_fwd_kernel<4><<<grid, CV_NT, 0, (hipStream_t)stream>>>(p);
  else
    cost_volume_fwd_kernel<1><<<grid, CV_NT, 0, (hipStream_t)stream>>>(p);
  return hipGetLastError() == hipSuccess ? DW_OK : DW_ERR_LAUNCH;
}

extern "C" int dw_cost_volume_bwd(int32_t B, int32_t F, int32_t h, int32_t w, int32_t D,
                                  const float* g_left, const float* g_right, float* g_imgl,
                                  float* g_imgr, float* g_carl, float* g_carr, void* stream) {
  if (!g_left || !g_right || !g_imgl || !g_imgr || !g_carl || !g_carr) return DW_ERR_NULL;
  if (const int rc = cv_check(B, F, h, w, D)) return rc;
  if (F > 65535) return DW_ERR_SHAPE;
  CvParams p{};
  p.F = F; p.h = h; p.w = w; p.D = D;
  p.gvol[0] = g_left; p.gvol[1] = g_right;
  p.gin[0] = g_imgl; p.gin[1] = g_imgr; p.gin[2] = g_carl; p.gin[3] = g_carr;
  const long n = (long)h * w;
  const bool vec4 = (w % 4 == 0) && cv_aligned16(g_left) && cv_aligned16(g_right) && cv_aligned16(g_imgl) &&
                    cv_aligned16(g_imgr) && cv_aligned16(g_carl) && cv_aligned16(g_carr);
  const int V = vec4 ? 4 : 1;
  const dim3 grid((unsigned)((n / V + CV_NT - 1) / CV_NT), F, B);
  if (vec4)
    cost_volume_bwd_kernel<4><<<grid, CV_NT, 0, (hipStream_t)stream>>>(p);
  else
    cost_volume_bwd_kernel<1><<<grid, CV_NT, 0, (hipStream_t)stream>>>(p);
  return hipGetLastError() == hipSuccess ? DW_OK : DW_ERR_LAUNCH;
}
