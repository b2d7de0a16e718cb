// Velodyne scan -> image plane: projection, in-bounds filter, order-preserving
// compaction and the sparse nearest-depth map.
//
// Reference: Dataset.generate_depth_map, /root/reference/dataset/load_dataset.py
// lines 116-148 (everything between loading the scan and the Delaunay hole
// filling, which stays on the host).  NumPy evaluates these lines in float64,
// so do the kernels; every expression is one IEEE operation at a time (this
// file is compiled with -ffp-contract=off) so the results can be compared bit
// for bit.
//
//   keep rows with forward >= 0                                    (:119)
//   c = P_velo2im (3x4) . (forward, left, up, 1)                   (:122)
//   x = rint(c0/c2) - 1, y = rint(c1/c2) - 1, z = c2 (or forward)  (:123-131)
//   keep 0 <= x < W, 0 <= y < H                                    (:132-134)
//   depth[y, x] = z, the last point in scan order wins             (:137-138)
//   points sharing the key y*(W-1) + x - 1 form a group; a group of more
//   than one point writes its minimum z at the pixel of its first point
//   (:141-147).  The key is the reference's `sub2ind`, which is not the
//   pixel index: (y, W-1) and (y+1, 0) share a key.  Kept as is.
//   depth < 0 -> 0                                                  (:148)
//
// What that amounts to.  Two different pixels share a key only as the pair (y, W-1), (y+1, 0).
// A pixel of the columns 1 .. W-2 is therefore a group of its own: one point -> its z, several
// -> their minimum, i.e. the minimum either way, and with :148, max(min z, 0) = min max(z, 0).
// That is one integer atomic per point straight into the output map: the map is zero-filled, a
// point stores max(z, 0) with the sign bit set (as a signed 64-bit integer every such pattern is
// below zero -- the empty pixel -- and ordered like z), atomicMin, and the second pass over the
// points, which writes the compacted list, clears the sign bit of each kept point's pixel again
// (idempotent: every point of a pixel may do it).  No per-pixel records, no second dense array.
// Only the points of the two border columns (2 in W of them) need "last in scan order", "first of
// the group" and the group minimum: they go through three integer atomics on a small per-row
// record, and a last kernel with one thread per border pixel writes those pixels.  Every result
// is independent of the execution order.
#include "dw_common.h"

namespace dw {

constexpr int LD_NT = 256;

// per scan, border column (0: x = 0, 1: x = W - 1) and row; all-zero = no point
struct __attribute__((aligned(16))) EdgeRec {
  long long nz;  // smallest encoded z of the pixel's points (see encode_z)
  int last1;     // 1 + the largest compacted index landing on the pixel
  int nfirst;    // 0x7fffffff - the smallest one
};

struct LidarParams {
  int B, n_max, H, W, vel_depth, nblk;
  const float* points;    // (B, n_max, 4)
  const int* n_points;    // (B)
  const double* P;        // (B, 3, 4)
  double* pts_im;         // (B, n_max, 3) out: compacted (x, y, z)
  int* n_valid;           // (B) out
  double* depth;          // (B, H, W) out
  // workspace
  // kept points of block k of scan b, in scan order, at [b][k * LD_NT + 0 .. blk_count[b][k])
  int* loc;               // (B, nblk * LD_NT)  pixel index
  double* zbuf;           // (B, nblk * LD_NT)  z
  int* blk_count;         // (B, nblk)   kept points per block of LD_NT points
  EdgeRec* edge;          // (B, 2, H)
};

// max(z, 0) with the sign bit set, as a signed integer: negative for every z, increasing with z
__device__ __forceinline__ long long encode_z(double z) {
  const double zc = (z < 0.0) ? 0.0 : z;
  return __builtin_bit_cast(long long, zc) | (long long)0x8000000000000000ull;
}
__device__ __forceinline__ double decode_z(long long e) {
  return __builtin_bit_cast(double, e & 0x7fffffffffffffffLL);
}

struct Projected {
  bool keep;
  int x, y;
  double z;
};
// lines :119-134 for point i of scan b
__device__ __forceinline__ Projected project_point(const LidarParams& p, int b, int i) {
  Projected r;
  const float4 v = reinterpret_cast<const float4*>(p.points)[(long)b * p.n_max + i];
  const double X = (double)v.x, Y = (double)v.y, Z = (double)v.z;
  const double* P = p.P + (long)b * 12;
  const double c0 = ((P[0] * X + P[1] * Y) + P[2] * Z) + P[3];
  const double c1 = ((P[4] * X + P[5] * Y) + P[6] * Z) + P[7];
  const double c2 = ((P[8] * X + P[9] * Y) + P[10] * Z) + P[11];
  const double x = rint(c0 / c2) - 1.0;
  const double y = rint(c1 / c2) - 1.0;
  r.keep = (v.x >= 0.f) && (x >= 0.0) && (y >= 0.0) && (x < (double)p.W) && (y < (double)p.H);
  r.x = r.keep ? (int)x : 0;
  r.y = r.keep ? (int)y : 0;
  r.z = p.vel_depth ? X : c2;
  return r;
}

// kept points of the calling block that precede this thread, and the block total
__device__ __forceinline__ int block_rank(bool keep, int* s_w, int& total) {
  const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;
  const unsigned long long m = __ballot(keep);
  const int before = __popcll(m & ((1ull << lane) - 1ull));
  if (lane == 0) s_w[wid] = __popcll(m);
  __syncthreads();
  int off = 0;
  total = 0;
#pragma unroll
  for (int w = 0; w < LD_NT / kWave; ++w) {
    if (w < wid) off += s_w[w];
    total += s_w[w];
  }
  return off + before;
}

// Pass 1 over the points (the map is zero): project, send every kept point of the inner columns
// to its pixel, and leave the block's kept points (pixel, z), in scan order, and their number for
// pass 2, which then touches kept points only.  Also clears the border records.
__global__ __launch_bounds__(LD_NT) void lidar_project_kernel(const LidarParams p) {
  __shared__ int s_w[LD_NT / kWave];
  const int b = blockIdx.y;
  const int i = blockIdx.x * LD_NT + threadIdx.x;
  const int n = min(p.n_points[b], p.n_max);
  for (int e = i; e < 2 * p.H; e += p.nblk * LD_NT) {
    EdgeRec z;
    z.nz = 0;
    z.last1 = 0;
    z.nfirst = 0;
    p.edge[(long)b * 2 * p.H + e] = z;
  }
  Projected q;
  q.keep = false;
  if (i < n) {
    q = project_point(p, b, i);
    if (q.keep && q.x > 0 && q.x < p.W - 1)
      atomicMin(reinterpret_cast<long long*>(p.depth) + (long)b * p.H * p.W + (long)q.y * p.W + q.x, encode_z(q.z));
  }
  int total;
  const int rank = block_rank(q.keep, s_w, total);
  if (q.keep) {
    const long at = ((long)b * p.nblk + blockIdx.x) * LD_NT + rank;
    p.loc[at] = q.y * p.W + q.x;
    p.zbuf[at] = q.z;
  }
  if (threadIdx.x == 0) p.blk_count[b * p.nblk + blockIdx.x] = total;
}

// Pass 2, over the kept points block by block: the compacted list (scan order kept: a block's
// offset is the sum of the counts of the blocks before it), the sign bits of the pixels pass 1
// filled (every atomic of pass 1 has completed), and the records of the border columns, which
// need the compacted index.
// One wave per block of pass 1 (a launch of whole blocks per point block would be nothing but
// their start-up latency: a point block keeps some thirty points).
__global__ __launch_bounds__(LD_NT) void lidar_compact_kernel(const LidarParams p) {
  const int b = blockIdx.y, lane = threadIdx.x & (kWave - 1);
  const int blk = blockIdx.x * (LD_NT / kWave) + threadIdx.x / kWave;  // uniform over the wave
  if (blk >= p.nblk) return;
  // kept points of this scan in the point blocks before this one (integer sum: any order)
  int offset = 0;
  for (int k = lane; k < blk; k += kWave) offset += p.blk_count[b * p.nblk + k];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) offset += __shfl_xor(offset, off, kWave);
  const int total = p.blk_count[b * p.nblk + blk];
  for (int t = lane; t < total; t += kWave) {
    const long at = ((long)b * p.nblk + blk) * LD_NT + t;
    const int loc = p.loc[at];
    const double z = p.zbuf[at];
    const int y = (int)((unsigned)loc / (unsigned)p.W), x = loc - y * p.W;
    const int j = offset + t;
    double* o = p.pts_im + ((long)b * p.n_max + j) * 3;
    o[0] = (double)x;
    o[1] = (double)y;
    o[2] = z;
    if (x == 0 || x == p.W - 1) {
      EdgeRec* r = &p.edge[((long)b * 2 + (x == 0 ? 0 : 1)) * p.H + y];
      atomicMax(&r->last1, j + 1);
      atomicMax(&r->nfirst, 0x7fffffff - j);
      atomicMin(&r->nz, encode_z(z));
    } else {
      // (an atomic without a return value: the read-modify-write happens at the memory side and
      // the lane does not wait for it; every point of the pixel leaves the same value)
      unsigned long long* cell = reinterpret_cast<unsigned long long*>(p.depth) + (long)b * p.H * p.W + loc;
      __hip_atomic_fetch_and(cell, 0x7fffffffffffffffull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (blk == p.nblk - 1 && lane == 0) p.n_valid[b] = offset + total;
}

// One thread per pixel of the two border columns.  The group of the pixel's key
// y * (W - 1) + x - 1 is the pixel itself and the pixel that shares the key ((y, W-1) with
// (y+1, 0)); a group of more than one point has its smallest z at the pixel of its first
// point, every other pixel with points keeps the z of its last point (:137-147).
__global__ __launch_bounds__(LD_NT) void lidar_border_kernel(const LidarParams p) {
  const int b = blockIdx.y;
  const int e = blockIdx.x * LD_NT + threadIdx.x;
  if (e >= 2 * p.H) return;
  const int side = e / p.H, y = e - side * p.H;
  const EdgeRec* edge = p.edge + (long)b * 2 * p.H;
  const EdgeRec r = edge[e];
  if (r.last1 == 0) return;  // no point: the pixel keeps the zero fill
  int first = 0x7fffffff - r.nfirst, last = r.last1 - 1;
  long long nz = r.nz;
  const int my_first = first;
  // (W = 1 is refused by the host: a pixel is never in both border columns)
  const int oy = side ? y + 1 : y - 1;
  if (oy >= 0 && oy < p.H) {
    const EdgeRec o = edge[(side ? 0 : 1) * p.H + oy];
    if (o.last1 > 0) {
      first = min(first, 0x7fffffff - o.nfirst);
      last = max(last, o.last1 - 1);
      nz = min(nz, o.nz);
    }
  }
  double d;
  if (first != last && first == my_first) {
    d = decode_z(nz);
  } else {
    d = p.pts_im[((long)b * p.n_max + (r.last1 - 1)) * 3 + 2];
    d = (d < 0.0) ? 0.0 : d;
  }
  p.depth[(long)b * p.H * p.W + (long)y * p.W + (side ? p.W - 1 : 0)] = d;
}

struct LidarLayout {
  size_t loc, zbuf, blk, edge, total;
};

static LidarLayout lidar_layout(int B, int n_max, int H, int W) {
  LidarLayout L{};
  const size_t nblk = (size_t)(n_max + LD_NT - 1) / LD_NT;
  size_t o = 0;
  auto take = [&](size_t bytes) {
    const size_t at = o;
    o += align_up(bytes, 256);
    return at;
  };
  L.zbuf = take(sizeof(double) * B * nblk * LD_NT);
  L.loc = take(sizeof(int) * B * nblk * LD_NT);
  L.edge = take(sizeof(EdgeRec) * B * 2 * H);
  L.blk = take(sizeof(int) * B * nblk);
  L.total = o;
  return L;
}

static bool lidar_shape_ok(int B, int n_max, int H, int W) {
  return B >= 1 && B <= 65535 && n_max >= 1 && n_max < 0x7fffffff && H >= 1 && W >= 2 && (long)H * W < (1L << 30) &&
         (long)B * ((long)H * W + 1) < (1L << 31) && (long)B * n_max < (1L << 31);
}

}  // namespace dw

using namespace dw;

extern "C" size_t dw_lidar_workspace_bytes(int32_t B, int32_t n_max, int32_t H, int32_t W) {
  if (!lidar_shape_ok(B, n_max, H, W)) return 0;
  return lidar_layout(B, n_max, H, W).total;
}

extern "C" int dw_lidar_project(int32_t B, int32_t n_max, int32_t H, int32_t W, const float* points,
                                const int32_t* n_points, const double* P, int32_t vel_depth,
                                double* pts_im, int32_t* n_valid, double* depth, void* workspace,
                                size_t workspace_bytes, void* stream) {
  if (!points || !n_points || !P || !pts_im || !n_valid || !depth || !workspace) return DW_ERR_NULL;
  if (!lidar_shape_ok(B, n_max, H, W)) return DW_ERR_SHAPE;
  if (((uintptr_t)points & 15) || ((uintptr_t)depth & 7)) return DW_ERR_ARG;
  const LidarLayout L = lidar_layout(B, n_max, H, W);
  if (workspace_bytes < L.total || ((uintptr_t)workspace & 255)) return DW_ERR_WORKSPACE;
  char* ws = (char*)workspace;
  LidarParams p{};
  p.B = B; p.n_max = n_max; p.H = H; p.W = W; p.vel_depth = vel_depth ? 1 : 0;
  p.nblk = (n_max + LD_NT - 1) / LD_NT;
  p.points = points; p.n_points = n_points; p.P = P;
  p.pts_im = pts_im; p.n_valid = n_valid; p.depth = depth;
  p.loc = (int*)(ws + L.loc);
  p.zbuf = (double*)(ws + L.zbuf);
  p.blk_count = (int*)(ws + L.blk);
  p.edge = (EdgeRec*)(ws + L.edge);
  hipStream_t st = (hipStream_t)stream;
  const long hw = (long)H * W;
  const dim3 gpts(p.nblk, B);
  if (hipMemsetAsync(depth, 0, sizeof(double) * (size_t)B * hw, st) != hipSuccess) return DW_ERR_LAUNCH;
  lidar_project_kernel<<<gpts, LD_NT, 0, st>>>(p);
  lidar_compact_kernel<<<dim3((p.nblk + LD_NT / kWave - 1) / (LD_NT / kWave), B), LD_NT, 0, st>>>(p);
  lidar_border_kernel<<<dim3((2 * H + LD_NT - 1) / LD_NT, B), LD_NT, 0, st>>>(p);
  return hipGetLastError() == hipSuccess ? DW_OK : DW_ERR_LAUNCH;
}
