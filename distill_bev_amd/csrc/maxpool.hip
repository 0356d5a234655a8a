// MaxPool2d(kernel 3, stride 2, padding 1) of the ResNet stem on channels-last tensors, forward + backward.
//
// Replaces ATen's max_pool2d_with_indices / max_pool2d_with_indices_backward (mmdet ResNet stem `self.maxpool`, called from
// mmdet3d/models/detectors/bevdet_distill_more.py's image_encoder -> self.img_backbone): same values, the same winner on ties
// (the FIRST maximum in (row, column) scan order of the window, a NaN wins) -- the stem's ReLU output is full of tied zeros.
// Forward: one float4 of channels per lane, the nine taps of a window are contiguous 16-byte reads; it also stores the winning tap
// (0..8, one byte per output element).  Backward as a GATHER: an input pixel is covered by at most 2 x 2 windows; its gradient is
// the sum (window row, then window column ascending) of the output gradients of the windows whose winner it is -- no atomics, the
// 554 MB input gradient is written once (ATen's NHWC backward: 461 us for the step's stem; this: one pass at the streaming rate).
#include "common.h"

namespace {

struct MpDims { int N, H, W, C4, Ho, Wo; };

// AFFINE (round 5): the pooled tensor is max over the window of relu(x * scale + shift) -- the stem's norm -> ReLU -> pooling in one
// pass over the convolution's output (coef = scale | shift per channel, the fused norm's coefficient row): the normalised,
// rectified 554 MB map is neither written nor read back.  Same winners as the three-module sequence (the tie rule sees the same values).
template <bool AFFINE>
__global__ __launch_bounds__(256) void maxpool3x3s2_fwd(const float4* __restrict__ x, float4* __restrict__ y,
                                                        uchar4* __restrict__ tap, MpDims d, long long total,
                                                        const float* __restrict__ coef) {
  const long long t = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const int c = static_cast<int>(t % d.C4);
  long long p = t / d.C4;
  const int ow = static_cast<int>(p % d.Wo); p /= d.Wo;
  const int oh = static_cast<int>(p % d.Ho);
  const int n = static_cast<int>(p / d.Ho);
  float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
  uchar4 w = make_uchar4(0, 0, 0, 0);
  bool first = true;
  float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
  if (AFFINE) {
    sc = reinterpret_cast<const float4*>(coef)[c];
    sh = reinterpret_cast<const float4*>(coef + 4 * d.C4)[c];
  }
#pragma unroll
  for (int kh = 0; kh < 3; ++kh) {
    const int h = 2 * oh - 1 + kh;
    if (h < 0 || h >= d.H) continue;
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const int ww = 2 * ow - 1 + kw;
      if (ww < 0 || ww >= d.W) continue;
      float4 v = x[(static_cast<size_t>(n) * d.H + h) * d.W * d.C4 + static_cast<size_t>(ww) * d.C4 + c];
      if (AFFINE) {                                   // fmaxf(NaN, 0) = 0 as in ATen's relu? no: torch.relu keeps NaN -- keep it here too
        const float ax = fmaf(v.x, sc.x, sh.x), ay = fmaf(v.y, sc.y, sh.y), az = fmaf(v.z, sc.z, sh.z), aw = fmaf(v.w, sc.w, sh.w);
        v.x = ax != ax ? ax : fmaxf(ax, 0.f); v.y = ay != ay ? ay : fmaxf(ay, 0.f);
        v.z = az != az ? az : fmaxf(az, 0.f); v.w = aw != aw ? aw : fmaxf(aw, 0.f);
      }
      const unsigned char k = static_cast<unsigned char>(3 * kh + kw);
      // ATen: (val > maxval) || isnan(val), maxval starts at -inf with the window's first element as its index
      if (first || v.x > m.x || v.x != v.x) { m.x = v.x; w.x = k; }
      if (first || v.y > m.y || v.y != v.y) { m.y = v.y; w.y = k; }
      if (first || v.z > m.z || v.z != v.z) { m.z = v.z; w.z = k; }
      if (first || v.w > m.w || v.w != v.w) { m.w = v.w; w.w = k; }
      first = false;
    }
  }
  y[t] = m;
  tap[t] = w;
}

// one lane = one float4 of channels of a 2 x 2 block of input pixels (rows 2i, 2i + 1, columns 2j, 2j + 1): the block is covered by the
// windows (i, i + 1) x (j, j + 1) only, so the four (gradient, winner) pairs are read once for four outputs (a lane per pixel read
// them 2.25 times on average: FETCH 3.4 x the two tensors).  Sums in (window row, window column) order as before.
__global__ __launch_bounds__(256) void maxpool3x3s2_bwd(const float4* __restrict__ gy, const uchar4* __restrict__ tap,
                                                        float4* __restrict__ gx, MpDims d, int Hb, int Wb, long long total) {
  const long long t = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const int c = static_cast<int>(t % d.C4);
  long long p = t / d.C4;
  const int j = static_cast<int>(p % Wb); p /= Wb;
  const int i = static_cast<int>(p % Hb);
  const int n = static_cast<int>(p / Hb);
  float4 v[2][2];
  uchar4 s[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int oh = i + a, ow = j + b;
      if (oh < d.Ho && ow < d.Wo) {
        const size_t o = ((static_cast<size_t>(n) * d.Ho + oh) * d.Wo + ow) * d.C4 + c;
        v[a][b] = gy[o];
        s[a][b] = tap[o];
      } else {
        v[a][b] = make_float4(0.f, 0.f, 0.f, 0.f);
        s[a][b] = make_uchar4(255, 255, 255, 255);
      }
    }
#pragma unroll
  for (int dh = 0; dh < 2; ++dh) {
    const int h = 2 * i + dh;
    if (h >= d.H) continue;
#pragma unroll
    for (int dw = 0; dw < 2; ++dw) {
      const int w = 2 * j + dw;
      if (w >= d.W) continue;
      float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
      // windows oh with 2 oh - 1 <= h <= 2 oh + 1: an even row belongs to window i only, an odd row to i and i + 1 (same for columns)
#pragma unroll
      for (int a = 0; a <= dh; ++a)
#pragma unroll
        for (int b = 0; b <= dw; ++b) {
          const int k = 3 * (h - (2 * (i + a) - 1)) + (w - (2 * (j + b) - 1));
          if (s[a][b].x == k) g.x += v[a][b].x;
          if (s[a][b].y == k) g.y += v[a][b].y;
          if (s[a][b].z == k) g.z += v[a][b].z;
          if (s[a][b].w == k) g.w += v[a][b].w;
        }
      st_nt(&gx[((static_cast<size_t>(n) * d.H + h) * d.W + w) * d.C4 + c], g);
    }
  }
}

bool mp_dims(int N, int H, int W, int C, MpDims* d) {
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3)) return false;
  d->N = N; d->H = H; d->W = W; d->C4 = C >> 2;
  d->Ho = (H + 2 - 3) / 2 + 1;
  d->Wo = (W + 2 - 3) / 2 + 1;
  return true;
}

}  // namespace

extern "C" int dbev_maxpool3x3s2_forward(const float* x_nhwc, int N, int H, int W, int C, float* y_nhwc, unsigned char* winner,
                                         dbevStream_t stream) {
  MpDims d;
  if (!mp_dims(N, H, W, C, &d) || x_nhwc == nullptr || y_nhwc == nullptr || winner == nullptr) return DBEV_EINVAL;
  const long long total = static_cast<long long>(N) * d.Ho * d.Wo * d.C4;
  hipLaunchKernelGGL(maxpool3x3s2_fwd<false>, dim3(dbev_ceil_div(total, 256)), dim3(256), 0, dbev_stream(stream),
                     reinterpret_cast<const float4*>(x_nhwc), reinterpret_cast<float4*>(y_nhwc), reinterpret_cast<uchar4*>(winner), d,
                     total, static_cast<const float*>(nullptr));
  DBEV_LAUNCH_CHECK();
  return 0;
}

extern "C" int dbev_norm_relu_maxpool3x3s2_forward(const float* x_nhwc, const float* scale_shift, int N, int H, int W, int C,
                                                   float* y_nhwc, unsigned char* winner, dbevStream_t stream) {
  MpDims d;
  if (!mp_dims(N, H, W, C, &d) || x_nhwc == nullptr || scale_shift == nullptr || y_nhwc == nullptr || winner == nullptr) return DBEV_EINVAL;
  const long long total = static_cast<long long>(N) * d.Ho * d.Wo * d.C4;
  hipLaunchKernelGGL(maxpool3x3s2_fwd<true>, dim3(dbev_ceil_div(total, 256)), dim3(256), 0, dbev_stream(stream),
                     reinterpret_cast<const float4*>(x_nhwc), reinterpret_cast<float4*>(y_nhwc), reinterpret_cast<uchar4*>(winner), d,
                     total, scale_shift);
  DBEV_LAUNCH_CHECK();
  return 0;
}

extern "C" int dbev_maxpool3x3s2_backward(const float* grad_y_nhwc, const unsigned char* winner, int N, int H, int W, int C,
                                          float* grad_x_nhwc, dbevStream_t stream) {
  MpDims d;
  if (!mp_dims(N, H, W, C, &d) || grad_y_nhwc == nullptr || winner == nullptr || grad_x_nhwc == nullptr) return DBEV_EINVAL;
  const int Hb = (H + 1) / 2, Wb = (W + 1) / 2;                                 // 2 x 2 input blocks
  const long long total = static_cast<long long>(N) * Hb * Wb * d.C4;
  hipLaunchKernelGGL(maxpool3x3s2_bwd, dim3(dbev_ceil_div(total, 256)), dim3(256), 0, dbev_stream(stream),
                     reinterpret_cast<const float4*>(grad_y_nhwc), reinterpret_cast<const uchar4*>(winner),
                     reinterpret_cast<float4*>(grad_x_nhwc), d, Hb, Wb, total);
  DBEV_LAUNCH_CHECK();
  return 0;
}
