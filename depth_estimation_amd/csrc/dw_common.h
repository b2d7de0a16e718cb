// Shared device helpers for libdepthwarp (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "depthwarp.h"

namespace dw {

// Constants of the forward definition, as the float32 values nearest to the
// decimal ones (written as hex floats so host and device agree to the bit).
constexpr float kNinth = 0x1.c71c72p-4f;      // fl(1/9)
constexpr float kThird = 0x1.555556p-2f;      // fl(1/3)
constexpr float kC1 = 0x1.a36e2ep-14f;        // fl(0.01^2)
constexpr float kC2 = 0x1.d7dbf4p-11f;        // fl(0.03^2)
constexpr float kWSsim = 0x1.b33334p-1f;      // fl(0.85)
constexpr float kWL1 = 0x1.333334p-3f;        // fl(0.15)
constexpr float kMinDisp = 0x1.47ae14p-7f;    // fl(1/100)
constexpr float kDispRange = 0x1.3fae14p+3f;  // fl(1/0.1 - 1/100)
constexpr float kProjEps = 0x1.ad7f2ap-24f;   // fl(1e-7)
constexpr float kMeanEps = 1e-7f;
constexpr float kWSmooth = 1e-3f;

constexpr int kWave = 64;

// Sum over the 64 lanes with DPP adds (no LDS traffic). The total is valid in
// lane 63 only. Fixed association order: bitwise reproducible.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
  const int moved = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false);
  return v + __builtin_bit_cast(float, moved);
}
__device__ __forceinline__ float wave_sum(float v) {
  v = dpp_add<0xb1, 0xf>(v);   // quad_perm:[1,0,3,2]
  v = dpp_add<0x4e, 0xf>(v);   // quad_perm:[2,3,0,1]
  v = dpp_add<0x124, 0xf>(v);  // row_ror:4
  v = dpp_add<0x128, 0xf>(v);  // row_ror:8   -> every lane holds its row's sum
  v = dpp_add<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
  v = dpp_add<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3 -> lane 63 holds the total
  return v;
}
constexpr int kSumLane = 63;

// Exchange of register halves between two values (gfx950): after swap32 `a`
// holds the lower 32 lanes of both inputs and `b` the upper 32; swap16 does the
// same with the even / odd rows of 16 lanes.  (The builtins' result is converted
// as a whole before an element is taken: element access on it is mis-compiled.)
typedef float dw_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void lane_swap32(float& a, float& b) {
  const dw_f2 r = __builtin_bit_cast(dw_f2, __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a),
                                                                             __builtin_bit_cast(unsigned, b), false, false));
  a = r.x;
  b = r.y;
}
__device__ __forceinline__ void lane_swap16(float& a, float& b) {
  const dw_f2 r = __builtin_bit_cast(dw_f2, __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, a),
                                                                             __builtin_bit_cast(unsigned, b), false, false));
  a = r.x;
  b = r.y;
}
// Wave sums of four values in one register: afterwards every lane of row 0 / 1
// / 2 / 3 (16 lanes each) holds the total of v0 / v2 / v1 / v3.  Two half
// exchanges fold the 64 lanes of each value onto one row, four DPP steps sum
// the row.  Fixed order: bitwise reproducible.
__device__ __forceinline__ float wave_sum4_fold(float v0, float v1, float v2, float v3) {
  lane_swap32(v0, v1);
  lane_swap32(v2, v3);
  float u = v0 + v1, w = v2 + v3;
  lane_swap16(u, w);
  return u + w;
}
__device__ __forceinline__ float row_sum(float t) {
  t = dpp_add<0xb1, 0xf>(t);   // quad_perm:[1,0,3,2]
  t = dpp_add<0x4e, 0xf>(t);   // quad_perm:[2,3,0,1]
  t = dpp_add<0x124, 0xf>(t);  // row_ror:4
  t = dpp_add<0x128, 0xf>(t);  // row_ror:8
  return t;
}

// Sum over the block, in a fixed order (bitwise reproducible). The result is
// valid in thread 0. `red` holds NT/64 floats. Contains two barriers.
template <int NT>
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;
  __syncthreads();
  if (lane == kSumLane) red[wid] = v;
  __syncthreads();
  float t = 0.f;
  if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 0; i < NT / kWave; ++i) t += red[i];
  }
  return t;
}

// Buffer addressing: a 128-bit resource (base, size in bytes) in scalar
// registers, a wave-uniform byte offset in a scalar register and one 32-bit
// per-lane byte offset; no 64-bit per-load address arithmetic in vector
// registers.  The range of a resource is 4 GB.
using buf_rsrc = __amdgpu_buffer_rsrc_t;
// `base` and `bytes` must be the same in every lane; they are moved to scalar
// registers explicitly (a resource the compiler believes to be per-lane costs a
// serialising loop around every load).
__device__ __forceinline__ buf_rsrc make_rsrc(const void* base, unsigned bytes) {
  const unsigned long long v = (unsigned long long)base;
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
  void* u = (void*)(((unsigned long long)hi << 32) | lo);
  return __builtin_amdgcn_make_buffer_rsrc(u, 0, __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000);
}
__device__ __forceinline__ float buf_load_f32(buf_rsrc r, unsigned lane_off, unsigned uniform_off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, lane_off, uniform_off, 0));
}

__device__ __forceinline__ int reflect_index(int i, int n) {
  i = abs(i);
  i = min(i, 2 * (n - 1) - i);  // reflected at n - 1 when beyond it
  return min(max(i, 0), n - 1);
}


struct Taps {
  int i0, i1;
  float l0, l1;
};

// Source taps of the bilinear up-sampling (align_corners = false):
// src = (dst + 0.5) * (in / out) - 0.5, clamped at 0.
__device__ __forceinline__ Taps upsample_taps(int dst, int in_size, float scale) {
  Taps t;
  float src = ((float)dst + 0.5f) * scale - 0.5f;
  src = fmaxf(src, 0.f);
  t.i0 = (int)src;
  t.i1 = t.i0 + (t.i0 < in_size - 1 ? 1 : 0);
  t.l1 = src - (float)t.i0;
  t.l0 = 1.0f - t.l1;
  return t;
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace dw
