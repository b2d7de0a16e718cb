// Streaming read bandwidth calibration: sum-reduce a large buffer with different access widths.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <typename T> __global__ void rd(const T* __restrict__ p, size_t n, float* out) {
  float a = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    T v = p[i];
    const float* f = reinterpret_cast<const float*>(&v);
    for (int k = 0; k < (int)(sizeof(T) / 4); ++k) a += f[k];
  }
  if (a == 12345.678f) out[0] = a;
}
// strided planes: each wave reads 16 rows `stride` floats apart (like a cost volume walked along d)
template <typename T> __global__ void rd_planes(const T* __restrict__ p, size_t planeT, int D, float* out) {
  float a = 0.f;
  const size_t n0 = (size_t)blockIdx.x * 64 + (threadIdx.x & 63);
  const int w = threadIdx.x >> 6, nw = blockDim.x >> 6, per = D / nw;
  if (n0 < planeT) {
#pragma unroll 16
    for (int d = w * per; d < (w + 1) * per; ++d) {
      T v = p[(size_t)d * planeT + n0];
      const float* f = reinterpret_cast<const float*>(&v);
      for (int k = 0; k < (int)(sizeof(T) / 4); ++k) a += f[k];
    }
  }
  if (a == 12345.678f) out[0] = a;
}
int main() {
  const size_t N = 192 * 640, D = 256, B = 4, total = N * D * B;
  float *buf, *out;
  hipMalloc(&buf, total * 4); hipMalloc(&out, 4);
  hipMemset(buf, 0, total * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto time = [&](auto fn, const char* name) {
    for (int i = 0; i < 3; ++i) fn();
    hipEventRecord(e0); for (int i = 0; i < 20; ++i) fn(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 20;
    printf("%-28s %.3f ms  %.0f GB/s\n", name, ms, total * 4 / ms / 1e6);
  };
  time([&] { rd<float4><<<2048, 256>>>((const float4*)buf, total / 4, out); }, "linear float4 2048x256");
  time([&] { rd<float4><<<8192, 256>>>((const float4*)buf, total / 4, out); }, "linear float4 8192x256");
  time([&] { rd<float2><<<4096, 256>>>((const float2*)buf, total / 2, out); }, "linear float2 4096x256");
  time([&] { rd<float><<<4096, 256>>>(buf, total, out); }, "linear float 4096x256");
  for (int b = 0; b < 1; ++b) {
    time([&] { for (size_t s = 0; s < B; ++s) rd_planes<float><<<N / 64, 1024>>>(buf + s * N * D, N, D, out); }, "planes float  16w x 16d");
    time([&] { for (size_t s = 0; s < B; ++s) rd_planes<float2><<<N / 128, 1024>>>((const float2*)(buf + s * N * D), N / 2, D, out); }, "planes float2 16w x 16d");
    time([&] { for (size_t s = 0; s < B; ++s) rd_planes<float4><<<N / 256, 1024>>>((const float4*)(buf + s * N * D), N / 4, D, out); }, "planes float4 16w x 16d");
    time([&] { for (size_t s = 0; s < B; ++s) rd_planes<float4><<<N / 256, 512>>>((const float4*)(buf + s * N * D), N / 4, D, out); }, "planes float4  8w x 32d");
    time([&] { for (size_t s = 0; s < B; ++s) rd_planes<float4><<<N / 256, 256>>>((const float4*)(buf + s * N * D), N / 4, D, out); }, "planes float4  4w x 64d");
  }
  return 0;
}
