// HBM streaming ceilings of the box the bench runs on: read-only (sum), write-only (fill) and copy kernels over a 554 MB
// buffer (the largest activation of the step), float4 per lane, swept over workgroup count and loads in flight per lane.
// Calibrates what the normalisation passes can reach:   hipcc -O3 -w --offload-arch=gfx950 tools/membench.hip -o /tmp/membench && /tmp/membench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st_nt(float4* p, const float4& v) {
  __builtin_nontemporal_store(v4f{v.x, v.y, v.z, v.w}, reinterpret_cast<v4f*>(p));
}

template <int U>
__global__ __launch_bounds__(512) void k_read(const float4* __restrict__ x, size_t n4, float* __restrict__ out) {
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
  size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  for (; i + (U - 1) * stride < n4; i += U * stride) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = x[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
  }
  for (; i < n4; i += stride) { const float4 v = x[i]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;      // keep the loads alive
}

template <int U>
__global__ __launch_bounds__(512) void k_write(float4* __restrict__ y, size_t n4) {
  const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
  const float4 v = make_float4(1.f, 2.f, 3.f, 4.f);
  for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n4; i += stride) st_nt(y + i, v);
}

template <int U>
__global__ __launch_bounds__(512) void k_copy(const float4* __restrict__ x, float4* __restrict__ y, size_t n4) {
  const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
  size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  for (; i + (U - 1) * stride < n4; i += U * stride) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = x[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; ++u) st_nt(y + i + u * stride, v[u]);
  }
  for (; i < n4; i += stride) y[i] = x[i];
}

template <typename F>
static float time_ms(F launch, int iters) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) launch();
  hipEventRecord(a);
  for (int i = 0; i < iters; ++i) launch();
  hipEventRecord(b); hipEventSynchronize(b);
  float ms = 0.f; hipEventElapsedTime(&ms, a, b);
  return ms / iters;
}

int main() {
  const size_t bytes = 48ull * 256 * 64 * 176 * 4;       // 554 MB
  const size_t n4 = bytes / 16;
  float4 *x, *y; float* out;
  hipMalloc(&x, bytes); hipMalloc(&y, bytes); hipMalloc(&out, 64);
  hipMemset(x, 0, bytes); hipMemset(y, 0, bytes);
  printf("buffer %.0f MB; GB/s by workgroups (512 threads) x loads in flight per lane\n", bytes / 1e6);
  for (int blocks : {256, 512, 1024, 2048, 4096, 8192}) {
    float r1 = time_ms([&] { hipLaunchKernelGGL(k_read<1>, dim3(blocks), dim3(512), 0, 0, x, n4, out); }, 20);
    float r4 = time_ms([&] { hipLaunchKernelGGL(k_read<4>, dim3(blocks), dim3(512), 0, 0, x, n4, out); }, 20);
    float r8 = time_ms([&] { hipLaunchKernelGGL(k_read<8>, dim3(blocks), dim3(512), 0, 0, x, n4, out); }, 20);
    float w = time_ms([&] { hipLaunchKernelGGL(k_write<1>, dim3(blocks), dim3(512), 0, 0, y, n4); }, 20);
    float c1 = time_ms([&] { hipLaunchKernelGGL(k_copy<1>, dim3(blocks), dim3(512), 0, 0, x, y, n4); }, 20);
    float c4 = time_ms([&] { hipLaunchKernelGGL(k_copy<4>, dim3(blocks), dim3(512), 0, 0, x, y, n4); }, 20);
    printf("blocks %5d  read U1 %6.0f U4 %6.0f U8 %6.0f | write %6.0f | copy(r+w) U1 %6.0f U4 %6.0f\n", blocks, bytes / r1 / 1e6, bytes / r4 / 1e6,
           bytes / r8 / 1e6, bytes / w / 1e6, 2 * bytes / c1 / 1e6, 2 * bytes / c4 / 1e6);
  }
  return 0;
}
