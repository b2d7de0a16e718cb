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
// "Last wins", "first of the group" and "minimum of the group" are computed
// with integer max / min atomics on the compacted point index and on an
// order-preserving integer image of z: independent of execution order.  They
// are kept per pixel, in one 16-byte record (one cache line per point): a group is a
// pixel, or the two pixels (y, W-1) and (y+1, 0), whose records are combined when the map
// is written.  The map itself is written sparsely: zero fill, then one thread per kept
// point decides the pixel it is the last point of.
#include "dw_common.h"

namespace dw {

constexpr int LD_NT = 256;

struct LidarParams {
  int B, n_max, H, W, vel_depth, nblk;
  const float* points;    // (B, n_max, 4)
  const int* n_points;    // (B)
  const double* P;        // (B, 3, 4)
  double* pts_im;         // (B, n_max, 3) out: compacted (x, y, z)
  int* n_valid;           // (B) out
  double* depth;          // (B, H, W) out
  // workspace
  int* loc;               // (B, n_max)  pixel index of point i, -1 if dropped
  double* zbuf;           // (B, n_max)  z of point i
  int* blk_count;         // (B, nblk)   kept points per block, then exclusive offsets
  struct PixRec* rec;     // (B, H*W)
};

struct __attribute__((aligned(16))) PixRec {
  long long minz;  // smallest ordered(z) of the pixel's points
  int first;       // smallest compacted index landing on the pixel
  int last;        // largest one (-1: none)
};

// order-preserving map double -> signed 64-bit integer (and back)
__device__ __forceinline__ long long ordered(double z) {
  const long long b = __builtin_bit_cast(long long, z);
  return b >= 0 ? b : (b ^ 0x7fffffffffffffffLL);
}
__device__ __forceinline__ double unordered(long long k) {
  return __builtin_bit_cast(double, k >= 0 ? k : (k ^ 0x7fffffffffffffffLL));
}

__global__ __launch_bounds__(LD_NT) void lidar_init_kernel(const LidarParams p) {
  const int b = blockIdx.y;
  const long hw = (long)p.H * p.W;
  const long i = (long)blockIdx.x * LD_NT + threadIdx.x;
  if (i < hw) {
    PixRec r;
    r.minz = 0x7fffffffffffffffLL;
    r.first = 0x7fffffff;
    r.last = -1;
    p.rec[b * hw + i] = r;  // one 16-byte store
  }
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

__global__ __launch_bounds__(LD_NT) void lidar_project_kernel(const LidarParams p) {
  __shared__ int s_w[LD_NT / kWave];
  const int b = blockIdx.y;
  const int i = blockIdx.x * LD_NT + threadIdx.x;
  const int n = min(p.n_points[b], p.n_max);
  bool keep = false;
  if (i < n) {
    const float4 v = reinterpret_cast<const float4*>(p.points)[(long)b * p.n_max + i];
    const double X = (double)v.x, Y = (double)v.y, Z = (double)v.z;
    const double* P = p.P + (long)b * 12;
    const double c0 = ((P[0] * X + P[1] * Y) + P[2] * Z) + P[3];
    const double c1 = ((P[4] * X + P[5] * Y) + P[6] * Z) + P[7];
    const double c2 = ((P[8] * X + P[9] * Y) + P[10] * Z) + P[11];
    const double x = rint(c0 / c2) - 1.0;
    const double y = rint(c1 / c2) - 1.0;
    keep = (v.x >= 0.f) && (x >= 0.0) && (y >= 0.0) && (x < (double)p.W) && (y < (double)p.H);
    p.loc[(long)b * p.n_max + i] = keep ? (int)y * p.W + (int)x : -1;
    p.zbuf[(long)b * p.n_max + i] = p.vel_depth ? X : c2;
  }
  int total;
  block_rank(keep, s_w, total);
  if (threadIdx.x == 0) p.blk_count[b * p.nblk + blockIdx.x] = total;
}

// exclusive scan of the per-block counts of one scan (one block per scan)
__global__ __launch_bounds__(LD_NT) void lidar_scan_kernel(const LidarParams p) {
  __shared__ int s_w[LD_NT / kWave];
  __shared__ int s_carry;
  const int b = blockIdx.x, tid = threadIdx.x;
  const int lane = tid & (kWave - 1), wid = tid / kWave;
  if (tid == 0) s_carry = 0;
  __syncthreads();
  for (int base = 0; base < p.nblk; base += LD_NT) {
    const int k = base + tid;
    const int v = (k < p.nblk) ? p.blk_count[b * p.nblk + k] : 0;
    int incl = v;  // inclusive scan within the wave
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
      const int t = __shfl_up(incl, off, kWave);
      if (lane >= off) incl += t;
    }
    if (lane == kWave - 1) s_w[wid] = incl;
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wid; ++w) woff += s_w[w];
    const int carry = s_carry;
    if (k < p.nblk) p.blk_count[b * p.nblk + k] = carry + woff + incl - v;
    __syncthreads();
    if (tid == LD_NT - 1) s_carry = carry + woff + incl;
    __syncthreads();
  }
  if (tid == 0) p.n_valid[b] = s_carry;
}

__global__ __launch_bounds__(LD_NT) void lidar_scatter_kernel(const LidarParams p) {
  __shared__ int s_w[LD_NT / kWave];
  const int b = blockIdx.y;
  const int i = blockIdx.x * LD_NT + threadIdx.x;
  const int n = min(p.n_points[b], p.n_max);
  const long hw = (long)p.H * p.W;
  int loc = -1;
  if (i < n) loc = p.loc[(long)b * p.n_max + i];
  int total;
  const int j = p.blk_count[b * p.nblk + blockIdx.x] + block_rank(loc >= 0, s_w, total);
  if (loc >= 0) {
    const int y = loc / p.W, x = loc - y * p.W;
    const double z = p.zbuf[(long)b * p.n_max + i];
    double* o = p.pts_im + ((long)b * p.n_max + j) * 3;
    o[0] = (double)x;
    o[1] = (double)y;
    o[2] = z;
    PixRec* r = &p.rec[b * hw + loc];
    atomicMax(&r->last, j);
    atomicMin(&r->first, j);
    atomicMin(&r->minz, ordered(z));
  }
}

// One thread per kept point j: the last point of a pixel writes the pixel.  (Pixels
// without a point keep the zero fill.)
__global__ __launch_bounds__(LD_NT) void lidar_depth_kernel(const LidarParams p) {
  const int b = blockIdx.y;
  const long hw = (long)p.H * p.W;
  const int j = blockIdx.x * LD_NT + threadIdx.x;
  if (j >= p.n_valid[b]) return;
  const double* pts = p.pts_im + (long)b * p.n_max * 3;
  const int x = (int)pts[(long)j * 3 + 0], y = (int)pts[(long)j * 3 + 1];
  const long loc = (long)y * p.W + x;
  const PixRec* rec = p.rec + b * hw;
  const PixRec r = rec[loc];
  if (r.last != j) return;
  double d = pts[(long)j * 3 + 2];
  // The group of this pixel's key y * (W - 1) + x - 1: the pixel itself and, at the ends of a
  // row, the pixel that shares the key ((y, W-1) with (y+1, 0)).  A group of more than one
  // point writes its smallest z at the pixel of its first point.
  int first = r.first, last = r.last;
  long long minz = r.minz;
  long other = -1;
  if (x == p.W - 1 && y + 1 < p.H) other = loc + 1;
  if (x == 0 && y > 0) other = loc - 1;
  if (other >= 0) {
    const PixRec o = rec[other];
    if (o.last >= 0) {
      first = min(first, o.first);
      last = max(last, o.last);
      minz = min(minz, o.minz);
    }
  }
  if (first != last && first == r.first) d = unordered(minz);
  p.depth[b * hw + loc] = (d < 0.0) ? 0.0 : d;
}

struct LidarLayout {
  size_t loc, zbuf, blk, rec, total;
};

static LidarLayout lidar_layout(int B, int n_max, int H, int W) {
  LidarLayout L{};
  const size_t hw = (size_t)H * W, nblk = (size_t)(n_max + LD_NT - 1) / LD_NT;
  size_t o = 0;
  auto take = [&](size_t bytes) {
    const size_t at = o;
    o += align_up(bytes, 256);
    return at;
  };
  L.zbuf = take(sizeof(double) * B * n_max);
  L.rec = take(sizeof(PixRec) * B * hw);
  L.loc = take(sizeof(int) * B * n_max);
  L.blk = take(sizeof(int) * B * nblk);
  L.total = o;
  return L;
}

static bool lidar_shape_ok(int B, int n_max, int H, int W) {
  return B >= 1 && B <= 65535 && n_max >= 1 && H >= 1 && W >= 2 && (long)H * W < (1L << 30) &&
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
  if (((uintptr_t)points & 15)) return DW_ERR_ARG;
  const LidarLayout L = lidar_layout(B, n_max, H, W);
  if (workspace_bytes < L.total || ((uintptr_t)workspace & 255)) return DW_ERR_WORKSPACE;
  char* ws = (char*)workspace;
  LidarParams p{};
  p.B = B; p.n_max = n_max; p.H = H; p.W = W; p.vel_depth = vel_depth ? 1 : 0;
  p.nblk = (n_max + LD_NT - 1) / LD_NT;
  p.points = points; p.n_points = n_points; p.P = P;
  p.pts_im = pts_im; p.n_valid = n_valid; p.depth = depth;
  p.loc = (int*)(ws + L.loc); p.zbuf = (double*)(ws + L.zbuf); p.blk_count = (int*)(ws + L.blk);
  p.rec = (PixRec*)(ws + L.rec);
  hipStream_t st = (hipStream_t)stream;
  const long hw = (long)H * W;
  const dim3 gpix((unsigned)((hw + LD_NT - 1) / LD_NT), B), gpts(p.nblk, B);
  lidar_init_kernel<<<gpix, LD_NT, 0, st>>>(p);
  lidar_project_kernel<<<gpts, LD_NT, 0, st>>>(p);
  lidar_scan_kernel<<<B, LD_NT, 0, st>>>(p);
  lidar_scatter_kernel<<<gpts, LD_NT, 0, st>>>(p);
  if (hipMemsetAsync(depth, 0, sizeof(double) * (size_t)B * hw, st) != hipSuccess) return DW_ERR_LAUNCH;
  lidar_depth_kernel<<<gpts, LD_NT, 0, st>>>(p);
  return hipGetLastError() == hipSuccess ? DW_OK : DW_ERR_LAUNCH;
}
