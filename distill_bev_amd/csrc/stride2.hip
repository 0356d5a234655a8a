// Stride-2 1x1 convolutions (the `downsample` branch of a stage-first ResNet bottleneck, mmdet ResNet / mmdet3d
// bricks/res_block.py:102-230: nn.Conv2d(k = 1, stride = 2)) as a pixel subsample + the 1x1 GEMM kernels:
//
//   forward        xs = x[:, ::2, ::2, :] (channels-last)  ->  y = xs W^T                     (dbev_subsample2_nhwc + the bf16x6 GEMM)
//   data gradient  gxs = gy W                               ->  gx[:, ::2, ::2, :] = gxs, 0 elsewhere   (GEMM + dbev_upsample2_zero_nhwc)
//   weight grad    dW = gy^T xs                                                                (the GEMM's weight-gradient kernel on xs)
//
// The library runs these layers as strided implicit GEMMs at 65-100 TFLOP/s plus a statistics pass for the norm behind them
// (3.7 ms of the round-4 step); ATen's strided copy / index_put made the same decomposition slower than the library (round 4,
// +1.3 ms).  Both kernels here are plain HBM-bound streams: a lane moves one float4 of a pixel's channel row, a workgroup a run of
// output pixels; the zero-fill of the three skipped neighbours of every pixel is part of the same pass (no memset of the 4 x larger
// gradient).  H and W even.
#include "common.h"

namespace {

// y[n, ho, wo, :] = x[n, 2 ho, 2 wo, :];  C4 = C / 4 float4 per pixel; thread = (output pixel, float4 column)
__global__ __launch_bounds__(256) void subsample2(const float4* __restrict__ x, float4* __restrict__ y, long long Mo, int Wo, int C4) {
  const long long total = Mo * C4;
  for (long long i = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x; i < total; i += static_cast<long long>(gridDim.x) * 256) {
    const long long m = i / C4;
    const int q = static_cast<int>(i - m * C4);
    const long long R = m / Wo;                                   // global output row n * Ho + ho; input row = 2 R (H = 2 Ho)
    const int wo = static_cast<int>(m - R * Wo);
    const long long pin = (2 * R) * (2LL * Wo) + 2 * wo;          // input pixel index
    y[i] = x[pin * C4 + q];
  }
}

// gx[n, 2 ho + a, 2 wo + b, :] = (a == 0 && b == 0) ? g[n, ho, wo, :] : 0;  thread = (INPUT-resolution pixel, float4 column): every
// element of gx is written exactly once, rows of gx are written contiguously
__global__ __launch_bounds__(256) void upsample2_zero(const float4* __restrict__ g, float4* __restrict__ gx, long long Mi, int Wi, int C4) {
  const long long total = Mi * C4;
  const int Wo = Wi / 2;
  for (long long i = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x; i < total; i += static_cast<long long>(gridDim.x) * 256) {
    const long long p = i / C4;
    const int q = static_cast<int>(i - p * C4);
    const long long r = p / Wi;                                   // global input row n * H + h
    const int wi = static_cast<int>(p - r * Wi);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (((r | wi) & 1) == 0) v = g[((r >> 1) * Wo + (wi >> 1)) * C4 + q];
    st_nt(gx + i, v);                                             // written once, read by the next backward kernel from HBM anyway
  }
}

bool s2_ok(int N, int H, int W, int C) {
  return N > 0 && H > 0 && W > 0 && C > 0 && (H % 2) == 0 && (W % 2) == 0 && (C % 4) == 0;
}

}  // namespace

extern "C" int dbev_subsample2_nhwc(const float* x, float* y, int N, int H, int W, int C, dbevStream_t stream) {
  if (!s2_ok(N, H, W, C) || x == nullptr || y == nullptr) return DBEV_EINVAL;
  const long long Mo = static_cast<long long>(N) * (H / 2) * (W / 2);
  const long long total = Mo * (C / 4);
  const int grid = static_cast<int>(total / 256 + 1 < DBEV_MAX_GRID * 4LL ? total / 256 + 1 : DBEV_MAX_GRID * 4LL);
  hipLaunchKernelGGL(subsample2, dim3(grid), dim3(256), 0, dbev_stream(stream), reinterpret_cast<const float4*>(x),
                     reinterpret_cast<float4*>(y), Mo, W / 2, C / 4);
  DBEV_LAUNCH_CHECK();
  return 0;
}

extern "C" int dbev_upsample2_zero_nhwc(const float* g, float* gx, int N, int H, int W, int C, dbevStream_t stream) {
  if (!s2_ok(N, H, W, C) || g == nullptr || gx == nullptr) return DBEV_EINVAL;
  const long long Mi = static_cast<long long>(N) * H * W;
  const long long total = Mi * (C / 4);
  const int grid = static_cast<int>(total / 256 + 1 < DBEV_MAX_GRID * 4LL ? total / 256 + 1 : DBEV_MAX_GRID * 4LL);
  hipLaunchKernelGGL(upsample2_zero, dim3(grid), dim3(256), 0, dbev_stream(stream), reinterpret_cast<const float4*>(g),
                     reinterpret_cast<float4*>(gx), Mi, W, C / 4);
  DBEV_LAUNCH_CHECK();
  return 0;
}
