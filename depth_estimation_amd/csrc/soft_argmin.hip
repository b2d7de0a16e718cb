// Fused double-softmax + soft-argmin disparity regression, forward and backward.
//
// Reference: /root/reference/model/model.py:454-456 (view to (1, D, H, W),
// prob = softmax(-out, 1)) and :417-429 (squeeze, softmax(-prob, 0) AGAIN,
// sum_d d * p).  The second softmax of an already normalised volume is a quirk
// of the reference and is kept.  The reference streams the D x H x W volume
// six to eight times (neg, softmax, squeeze, neg, softmax, mul with a
// materialised d-grid, sum); here it is read from HBM exactly once per pass:
//
//   block = NW waves x 64 lanes; a lane owns VEC consecutive pixels, wave w owns
//   the disparities [w*DPW, (w+1)*DPW) of those pixels and keeps its DPW*VEC
//   cost values in registers.  The soft-max statistics (max, sum; then the two
//   sums of the second soft-max) are combined across the NW waves through LDS.
//   For fixed d the lanes of a wave read 64*VEC consecutive floats: coalesced.
//
// Backward recomputes the forward from the registers (nothing is saved) and
// adds one more cross-wave sum:  with p1 = softmax(-x), p2 = softmax(-p1),
// out = sum_d d p2_d:   dp1_d = -g p2_d (d - out);  dx_d = -p1_d (dp1_d - sum_k dp1_k p1_k).
#include "dw_common.h"

namespace dw {

#ifndef SA_NW_
#define SA_NW_ 8
#endif
constexpr int SA_NW = SA_NW_;   // waves per block
constexpr int SA_DPW = 256 / SA_NW;  // disparities per wave -> up to 256 per pixel in registers
constexpr int SA_NT = SA_NW * kWave;
constexpr float kLog2e = 1.4426950408889634f;
// 2^x on the hardware exponential unit (arguments here are <= 0 and far from
// the denormal range that the library wrapper would rescale for)
__device__ __forceinline__ float exp2_fast(float x) { return __builtin_amdgcn_exp2f(x); }
// Occupancy decides this kernel: a block alternates between a load phase and an
// arithmetic phase (two exponentials per element), so a CU needs two resident
// blocks to keep HBM busy -- 4 waves per SIMD, at most 128 registers per lane.
// That bounds the pixels per lane: 2 forward (64 cost values), 1 backward
// (which also keeps the second soft-max numerators).
#ifndef SA_MINW
#define SA_MINW 4
#endif
#ifndef SA_MINW_B
#define SA_MINW_B 4
#endif
#ifndef SA_VEC
#define SA_VEC 2
#endif
#ifndef SA_VEC_B
#define SA_VEC_B 1
#endif

// Buffer addressing: a 128-bit resource (base, size) in scalar registers, a
// wave-uniform byte offset in a scalar register (the disparity plane) and one
// 32-bit per-lane byte offset (the pixel).  No 64-bit per-load address
// registers, which is what lets a lane keep its whole column set in registers.
// Cache policy of the volume accesses: 2 = non-temporal.  The cost volume is read once and its
// gradient written once, 500 MB each: marking them as streaming keeps them from displacing
// each other in the caches (measured: forward 94 -> 90 us, backward 206 -> 202 us).
#ifndef SA_LOAD_AUX
#define SA_LOAD_AUX 2
#endif
#ifndef SA_STORE_AUX
#define SA_STORE_AUX 2
#endif
typedef unsigned sa_u2 __attribute__((ext_vector_type(2)));
typedef unsigned sa_u4 __attribute__((ext_vector_type(4)));
using sa_rsrc = buf_rsrc;

// The builtins return generic integer vectors; they are converted as a whole to
// float vectors before any element is taken (element access on the integer
// vector itself is mis-optimised into a single-dword load by this compiler).
typedef float sa_f2 __attribute__((ext_vector_type(2)));
typedef float sa_f4 __attribute__((ext_vector_type(4)));

template <int VEC>
__device__ __forceinline__ void load_vec(sa_rsrc r, unsigned lane_off, unsigned plane_off, float* v) {
  if constexpr (VEC == 4) {
    const sa_f4 t = __builtin_bit_cast(sa_f4, __builtin_amdgcn_raw_buffer_load_b128(r, lane_off, plane_off, SA_LOAD_AUX));
    v[0] = t.x;
    v[1] = t.y;
    v[2] = t.z;
    v[3] = t.w;
  } else if constexpr (VEC == 2) {
    const sa_f2 t = __builtin_bit_cast(sa_f2, __builtin_amdgcn_raw_buffer_load_b64(r, lane_off, plane_off, SA_LOAD_AUX));
    v[0] = t.x;
    v[1] = t.y;
  } else {
    v[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, lane_off, plane_off, SA_LOAD_AUX));
  }
}
template <int VEC>
__device__ __forceinline__ void store_vec(sa_rsrc r, unsigned lane_off, unsigned plane_off, const float* v) {
  if constexpr (VEC == 4) {
    const sa_f4 t = {v[0], v[1], v[2], v[3]};
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(sa_u4, t), r, lane_off, plane_off, SA_STORE_AUX);
  } else if constexpr (VEC == 2) {
    const sa_f2 t = {v[0], v[1]};
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(sa_u2, t), r, lane_off, plane_off, SA_STORE_AUX);
  } else {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[0]), r, lane_off, plane_off, SA_STORE_AUX);
  }
}

// Combine one value per (wave, pixel) over the NW waves. `buf` is [NW][64*VEC].
template <int VEC, bool IS_MAX>
__device__ __forceinline__ void cross_wave(float* buf, int wid, int lane, float* v) {
#pragma unroll
  for (int i = 0; i < VEC; ++i) buf[wid * (kWave * VEC) + lane * VEC + i] = v[i];
  __syncthreads();
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    float a = buf[lane * VEC + i];
#pragma unroll
    for (int w = 1; w < SA_NW; ++w) {
      const float t = buf[w * (kWave * VEC) + lane * VEC + i];
      a = IS_MAX ? fmaxf(a, t) : a + t;
    }
    v[i] = a;
  }
}

// FULL: D equals the SA_NW * SA_DPW disparities a block holds (the reference's
// 256), so no disparity slot is ever empty and the per-slot tests vanish.
template <int VEC, bool BWD, bool FULL>
__global__ __launch_bounds__(SA_NT, BWD ? SA_MINW_B : SA_MINW) void soft_argmin_kernel(int D, long N, const float* __restrict__ x,
                                                               const float* __restrict__ g_out,
                                                               float* __restrict__ out,
                                                               float* __restrict__ g_x) {
  __shared__ float s_buf[BWD ? 5 : 4][SA_NW * kWave * VEC];
  const int lane = threadIdx.x & (kWave - 1);
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);  // wave-uniform: row pointers stay scalar
  const int b = blockIdx.y;
  const long n0 = ((long)blockIdx.x * kWave + lane) * VEC;  // first pixel of this lane
  const bool act = n0 < N;                                  // N % VEC == 0 is checked on the host
  const unsigned plane_bytes = (unsigned)N * 4u;                 // D * N < 2^30 is checked on the host
  const sa_rsrc rx = make_rsrc(x + (long)b * D * N, (unsigned)D * plane_bytes);
  const int d0 = wid * SA_DPW;

  // z = -x * log2(e) for this wave's disparities (soft-max in base 2); missing ones behave as -inf
  float z[SA_DPW][VEC];
  float m[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) m[i] = -INFINITY;
  // Every load is issued unconditionally (clamped address), all of them before
  // the first use, so that they are in flight together; masking happens on the
  // values.  The empty asm keeps the compiler from sinking a load into the
  // conditional code that consumes it (which serialises the loads).
  const unsigned off = (unsigned)(act ? n0 : 0) * 4u;  // byte offset of this lane inside a disparity plane
#pragma unroll
  for (int k = 0; k < SA_DPW; ++k)
    load_vec<VEC>(rx, off, (unsigned)(FULL ? d0 + k : min(d0 + k, D - 1)) * plane_bytes, z[k]);
#pragma unroll
  for (int k = 0; k < SA_DPW; ++k) {
    const bool have = act && (FULL || (d0 + k) < D);
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      asm volatile("" : "+v"(z[k][i]));
      z[k][i] = have ? z[k][i] * -kLog2e : -INFINITY;
      m[i] = fmaxf(m[i], z[k][i]);
    }
  }
  cross_wave<VEC, true>(s_buf[0], wid, lane, m);
  if (!act) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) m[i] = 0.f;  // keeps the arithmetic below finite on idle lanes
  }
  // first soft-max: e1 = exp(z - m), Z1 = sum e1
  float zsum[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) zsum[i] = 0.f;
#pragma unroll
  for (int k = 0; k < SA_DPW; ++k)
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      z[k][i] = exp2_fast(z[k][i] - m[i]);  // 2^-inf = 0 for missing disparities
      zsum[i] += z[k][i];
    }
  cross_wave<VEC, false>(s_buf[1], wid, lane, zsum);
  float inv1[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) inv1[i] = act ? 1.0f / zsum[i] : 0.f;
  // second soft-max over -p1 (p1 in [0,1]: no shift needed), soft arg-min
  float s2[VEC], n2[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) s2[i] = n2[i] = 0.f;
#pragma unroll
  for (int k = 0; k < SA_DPW; ++k) {
    const bool have = FULL || (d0 + k) < D;
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      const float p1 = z[k][i] * inv1[i];
      const float e2 = have ? exp2_fast(p1 * -kLog2e) : 0.f;
      if (BWD) z[k][i] = p1;  // keep p1; e2 is recomputed in the backward pass
      s2[i] += e2;
      n2[i] = fmaf((float)(d0 + k), e2, n2[i]);
    }
  }
  cross_wave<VEC, false>(s_buf[2], wid, lane, s2);
  cross_wave<VEC, false>(s_buf[3], wid, lane, n2);
  float res[VEC], inv2[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    inv2[i] = act ? 1.0f / s2[i] : 0.f;
    res[i] = n2[i] * inv2[i];
  }
  if (!BWD) {
    if (act && wid == 0) store_vec<VEC>(make_rsrc(out + (long)b * N, plane_bytes), off, 0u, res);
    return;
  }
  // backward
  float g[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) g[i] = 0.f;
  if (act) load_vec<VEC>(make_rsrc(g_out + (long)b * N, plane_bytes), off, 0u, g);
  float t[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) t[i] = 0.f;
  // dp1_d = -g p2_d (d - out);  T = sum_d dp1_d p1_d
#pragma unroll
  for (int k = 0; k < SA_DPW; ++k) {
    const bool have = FULL || (d0 + k) < D;
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      float p1 = z[k][i];
      asm volatile("" : "+v"(p1));  // recompute e2 here instead of keeping all of them in registers
      const float p2 = have ? exp2_fast(p1 * -kLog2e) * inv2[i] : 0.f;
      const float dp1 = -g[i] * p2 * ((float)(d0 + k) - res[i]);
      t[i] = fmaf(dp1, p1, t[i]);
    }
  }
  cross_wave<VEC, false>(s_buf[4], wid, lane, t);
  const sa_rsrc rg = make_rsrc(g_x + (long)b * D * N, (unsigned)D * plane_bytes);
#pragma unroll
  for (int k = 0; k < SA_DPW; ++k) {
    const int d = d0 + k;
    if (act && (FULL || d < D)) {
      float v[VEC];
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        float p1 = z[k][i];
        asm volatile("" : "+v"(p1));
        const float p2 = exp2_fast(p1 * -kLog2e) * inv2[i];
        const float dp1 = -g[i] * p2 * ((float)d - res[i]);
        v[i] = -p1 * (dp1 - t[i]);
      }
      store_vec<VEC>(rg, off, (unsigned)d * plane_bytes, v);
    }
    // one plane at a time: keeps the scheduler from computing many planes ahead
    // of their stores, which would double the live registers
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <bool BWD>
static int launch_soft_argmin(int B, int D, long N, const float* x, const float* g_out, float* out,
                              float* g_x, hipStream_t st) {
  // widest vector the plane stride and the pointers allow
  auto aligned = [&](int bytes) {
    const uintptr_t m = bytes - 1;
    return (((uintptr_t)x & m) | ((uintptr_t)out & m) | ((uintptr_t)g_out & m) | ((uintptr_t)g_x & m)) == 0;
  };
  constexpr int V = BWD ? SA_VEC_B : SA_VEC;
  const bool full = (D == SA_NW * SA_DPW);
  if (N % V == 0 && aligned(4 * V)) {
    const dim3 grid((unsigned)((N / V + kWave - 1) / kWave), B);
    if (full)
      soft_argmin_kernel<V, BWD, true><<<grid, SA_NT, 0, st>>>(D, N, x, g_out, out, g_x);
    else
      soft_argmin_kernel<V, BWD, false><<<grid, SA_NT, 0, st>>>(D, N, x, g_out, out, g_x);
  } else {
    const dim3 grid((unsigned)((N + kWave - 1) / kWave), B);
    soft_argmin_kernel<1, BWD, false><<<grid, SA_NT, 0, st>>>(D, N, x, g_out, out, g_x);
  }
  return hipGetLastError() == hipSuccess ? DW_OK : DW_ERR_LAUNCH;
}

}  // namespace dw

using namespace dw;

extern "C" int dw_soft_argmin_fwd(int32_t B, int32_t D, int64_t N, const float* cost, float* disp,
                                  void* stream) {
  if (!cost || !disp) return DW_ERR_NULL;
  if (B < 1 || B > 65535 || D < 1 || D > SA_NW * SA_DPW || N < 1 || (long)D * N >= (1L << 30)) return DW_ERR_SHAPE;
  return launch_soft_argmin<false>(B, D, N, cost, nullptr, disp, nullptr, (hipStream_t)stream);
}

extern "C" int dw_soft_argmin_bwd(int32_t B, int32_t D, int64_t N, const float* cost,
                                  const float* g_disp, float* g_cost, void* stream) {
  if (!cost || !g_disp || !g_cost) return DW_ERR_NULL;
  if (B < 1 || B > 65535 || D < 1 || D > SA_NW * SA_DPW || N < 1 || (long)D * N >= (1L << 30)) return DW_ERR_SHAPE;
  return launch_soft_argmin<true>(B, D, N, cost, g_disp, nullptr, g_cost, (hipStream_t)stream);
}
