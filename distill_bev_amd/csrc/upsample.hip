// Bilinear upsampling (align_corners=True) forward/backward for gfx950 -- the non-GEMM half of the
// student adaptation layers of the FGD loss (nn.Upsample(scale_factor=4, mode='bilinear',
// align_corners=True) in front of the ThreeLayer 1x1 stacks, bevdet_distill.py:275-288).
// torch's generic kernel needs 1.3 ms for the 134 MB output of the backbone1 position; this is a
// pure streaming op: 4 gathers + 3 lerps per output element (forward), a <= (2s+1)^2-tap gather per
// input element (backward, no atomics -> deterministic).  Index arithmetic follows ATen's
// upsample_bilinear2d (area_pixel_compute_source_index with align_corners):
//   r = (in-1)/(out-1); src = r*o; i0 = (int)src; i1 = i0 + (i0 < in-1); l1 = src - i0; l0 = 1 - l1.
// Both NCHW and channels-last (NHWC) tensors are handled (CL: lanes run over channels).
#include "common.h"

namespace {

struct UpDims { int B, C, IH, IW, OH, OW; float rh, rw; };

__device__ __forceinline__ void src_index(int o, float r, int in, int& i0, int& i1, float& l0, float& l1) {
  const float s = r * static_cast<float>(o);
  i0 = static_cast<int>(s);
  i1 = i0 + (i0 < in - 1 ? 1 : 0);
  l1 = s - static_cast<float>(i0);
  l0 = 1.f - l1;
}

// Launch geometry removes every per-element division: blockIdx.y = output (forward) / input
// (backward) row, blockIdx.z = image (NHWC) or image*channel plane (NCHW); a thread owns one x
// position (NCHW) or one (x, 4-channel group) pair (NHWC, float4), so the row weights are
// wave-uniform scalars and all global accesses are coalesced.

// ---- NCHW: block (64, 4), grid (ceil(OW/64), ceil(OH/4), B*C) ----
__global__ __launch_bounds__(256) void up_fwd_nchw(const float* __restrict__ x, float* __restrict__ y, UpDims d) {
  const int ox = blockIdx.x * 64 + threadIdx.x;
  const int oy = blockIdx.y * 4 + threadIdx.y;
  if (ox >= d.OW || oy >= d.OH) return;
  const size_t plane = blockIdx.z;
  int y0, y1, x0, x1; float ly0, ly1, lx0, lx1;
  src_index(oy, d.rh, d.IH, y0, y1, ly0, ly1);
  src_index(ox, d.rw, d.IW, x0, x1, lx0, lx1);
  const float* r0 = x + (plane * d.IH + y0) * d.IW;
  const float* r1 = x + (plane * d.IH + y1) * d.IW;
  y[(plane * d.OH + oy) * d.OW + ox] = ly0 * (lx0 * r0[x0] + lx1 * r0[x1]) + ly1 * (lx0 * r1[x0] + lx1 * r1[x1]);
}

// ---- NHWC: grid (ceil(OW*C4/256), OH, B), C % 4 == 0 ----
__global__ __launch_bounds__(256) void up_fwd_nhwc(const float4* __restrict__ x, float4* __restrict__ y, UpDims d,
                                                   int C4) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= d.OW * C4) return;
  const int ox = t / C4, c4 = t - ox * C4;
  const int oy = blockIdx.y;
  const size_t b = blockIdx.z;
  int y0, y1, x0, x1; float ly0, ly1, lx0, lx1;
  src_index(oy, d.rh, d.IH, y0, y1, ly0, ly1);
  src_index(ox, d.rw, d.IW, x0, x1, lx0, lx1);
  const float4 a = x[((b * d.IH + y0) * d.IW + x0) * C4 + c4], bq = x[((b * d.IH + y0) * d.IW + x1) * C4 + c4];
  const float4 c = x[((b * d.IH + y1) * d.IW + x0) * C4 + c4], e = x[((b * d.IH + y1) * d.IW + x1) * C4 + c4];
  float4 o;
  o.x = ly0 * (lx0 * a.x + lx1 * bq.x) + ly1 * (lx0 * c.x + lx1 * e.x);
  o.y = ly0 * (lx0 * a.y + lx1 * bq.y) + ly1 * (lx0 * c.y + lx1 * e.y);
  o.z = ly0 * (lx0 * a.z + lx1 * bq.z) + ly1 * (lx0 * c.z + lx1 * e.z);
  o.w = ly0 * (lx0 * a.w + lx1 * bq.w) + ly1 * (lx0 * c.w + lx1 * e.w);
  y[((b * d.OH + oy) * d.OW + ox) * C4 + c4] = o;
}

// weight with which output index o reads input index i (sum of the i0 / i1 roles)
__device__ __forceinline__ float tap_weight(int o, int i, float r, int in) {
  int i0, i1; float l0, l1;
  src_index(o, r, in, i0, i1, l0, l1);
  float w = 0.f;
  if (i0 == i) w += l0;
  if (i1 == i) w += l1;
  return w;
}

// outputs o that can read input i: floor(r*o) in {i-1, i}  ->  a conservative window, exact test inside
__device__ __forceinline__ void tap_window(int i, float r, int out, int& lo, int& hi) {
  if (r > 0.f) {
    lo = max(0, static_cast<int>(ceilf((i - 1) / r)) - 1);
    hi = min(out - 1, static_cast<int>(floorf((i + 1) / r)) + 1);
  } else { lo = 0; hi = out - 1; }
}

// ---- NCHW backward: block (64, 4), grid (ceil(IW/64), ceil(IH/4), B*C) ----
__global__ __launch_bounds__(256) void up_bwd_nchw(const float* __restrict__ gy, float* __restrict__ gx, UpDims d) {
  const int ix = blockIdx.x * 64 + threadIdx.x;
  const int iy = blockIdx.y * 4 + threadIdx.y;
  if (ix >= d.IW || iy >= d.IH) return;
  const size_t plane = blockIdx.z;
  int ylo, yhi, xlo, xhi;
  tap_window(iy, d.rh, d.OH, ylo, yhi);
  tap_window(ix, d.rw, d.OW, xlo, xhi);
  float acc = 0.f;
  for (int oy = ylo; oy <= yhi; ++oy) {
    const float wy = tap_weight(oy, iy, d.rh, d.IH);
    if (wy == 0.f) continue;
    const float* row = gy + (plane * d.OH + oy) * d.OW;
    for (int ox = xlo; ox <= xhi; ++ox) {
      const float wx = tap_weight(ox, ix, d.rw, d.IW);
      if (wx != 0.f) acc += (wy * wx) * row[ox];
    }
  }
  gx[(plane * d.IH + iy) * d.IW + ix] = acc;
}

// ---- NHWC backward: grid (ceil(IW*C4/256), IH, B) ----
__global__ __launch_bounds__(256) void up_bwd_nhwc(const float4* __restrict__ gy, float4* __restrict__ gx, UpDims d,
                                                   int C4) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= d.IW * C4) return;
  const int ix = t / C4, c4 = t - ix * C4;
  const int iy = blockIdx.y;
  const size_t b = blockIdx.z;
  int ylo, yhi, xlo, xhi;
  tap_window(iy, d.rh, d.OH, ylo, yhi);
  tap_window(ix, d.rw, d.OW, xlo, xhi);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int oy = ylo; oy <= yhi; ++oy) {
    const float wy = tap_weight(oy, iy, d.rh, d.IH);
    if (wy == 0.f) continue;
    for (int ox = xlo; ox <= xhi; ++ox) {
      const float wx = tap_weight(ox, ix, d.rw, d.IW);
      if (wx == 0.f) continue;
      const float4 g = gy[((b * d.OH + oy) * d.OW + ox) * C4 + c4];
      const float w = wy * wx;
      acc.x += w * g.x; acc.y += w * g.y; acc.z += w * g.z; acc.w += w * g.w;
    }
  }
  gx[((b * d.IH + iy) * d.IW + ix) * C4 + c4] = acc;
}

bool mk(int B, int C, int IH, int IW, int OH, int OW, UpDims* d) {
  if (B <= 0 || C <= 0 || IH <= 0 || IW <= 0 || OH <= 0 || OW <= 0) return false;
  d->B = B; d->C = C; d->IH = IH; d->IW = IW; d->OH = OH; d->OW = OW;
  d->rh = OH > 1 ? static_cast<float>(IH - 1) / static_cast<float>(OH - 1) : 0.f;
  d->rw = OW > 1 ? static_cast<float>(IW - 1) / static_cast<float>(OW - 1) : 0.f;
  return true;
}

// ---- F.grid_sample(bilinear, padding_mode='zeros', align_corners=True), channels-last, forward ---------------------------------
// (shift_feature of BEVDepth4D warps the adjacent frame's BEV map into the current ego frame, bevdet_distill_more.py:41-94; the
// adjacent frame is detached, so the step only needs the forward.)  ATen's arithmetic (GridSampler.cuh): source index
// ((g + 1) / 2) * (size - 1), corner weights as products of the distances to the opposite corner, out-of-map corners skipped,
// accumulation order nw, ne, sw, se.  One float4 of channels per lane: ATen's kernel walks the channels of a pixel serially
// (220 us for the step's 8 x 80 x 128 x 128 map; this: one pass).
__global__ __launch_bounds__(256) void grid_sample_bilinear_nhwc(const float4* __restrict__ x, const float2* __restrict__ grid,
                                                                 float4* __restrict__ y, int H, int W, int C4, long long npix_out,
                                                                 int HoWo) {
  const long long t = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= npix_out * C4) return;
  const long long p = t / C4;
  const int c = static_cast<int>(t - p * C4);
  const long long n = p / HoWo;
  const float2 g = grid[p];
  float ix, iy, fx, fy, nw, ne, sw, se;
  {
    // ATen rounds the source index before it takes the corner distances: contracted into an FMA, (x_se - ix) uses the unrounded
    // product and the weights move by an ulp of ix (1e-5 of the output at index ~100)
#pragma clang fp contract(off)
    ix = ((g.x + 1.f) / 2.f) * static_cast<float>(W - 1);
    iy = ((g.y + 1.f) / 2.f) * static_cast<float>(H - 1);
    fx = floorf(ix); fy = floorf(iy);
    const float xs = fx + 1.f, ys = fy + 1.f;                                 // south-east corner
    nw = (xs - ix) * (ys - iy); ne = (ix - fx) * (ys - iy); sw = (xs - ix) * (iy - fy); se = (ix - fx) * (iy - fy);
  }
  const int x0 = static_cast<int>(fx), y0 = static_cast<int>(fy);
  const float4* __restrict__ xb = x + static_cast<size_t>(n) * H * W * C4 + c;
  float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
  const bool xin0 = x0 >= 0 && x0 < W, xin1 = x0 + 1 >= 0 && x0 + 1 < W, yin0 = y0 >= 0 && y0 < H, yin1 = y0 + 1 >= 0 && y0 + 1 < H;
  if (yin0 && xin0) { const float4 v = xb[(static_cast<size_t>(y0) * W + x0) * C4]; o.x += v.x * nw; o.y += v.y * nw; o.z += v.z * nw; o.w += v.w * nw; }
  if (yin0 && xin1) { const float4 v = xb[(static_cast<size_t>(y0) * W + x0 + 1) * C4]; o.x += v.x * ne; o.y += v.y * ne; o.z += v.z * ne; o.w += v.w * ne; }
  if (yin1 && xin0) { const float4 v = xb[(static_cast<size_t>(y0 + 1) * W + x0) * C4]; o.x += v.x * sw; o.y += v.y * sw; o.z += v.z * sw; o.w += v.w * sw; }
  if (yin1 && xin1) { const float4 v = xb[(static_cast<size_t>(y0 + 1) * W + x0 + 1) * C4]; o.x += v.x * se; o.y += v.y * se; o.z += v.z * se; o.w += v.w * se; }
  y[t] = o;
}

}  // namespace

extern "C" int dbev_upsample_bilinear_ac_forward(const float* x, float* y, int B, int C, int IH, int IW, int OH,
                                                 int OW, int channels_last, dbevStream_t stream) {
  UpDims d;
  if (!mk(B, C, IH, IW, OH, OW, &d)) return DBEV_EINVAL;
  hipStream_t s = dbev_stream(stream);
  if (channels_last) {
    if (C & 3) return DBEV_EINVAL;
    const int C4 = C >> 2;
    hipLaunchKernelGGL(up_fwd_nhwc, dim3(dbev_ceil_div(static_cast<long long>(OW) * C4, 256), OH, B), dim3(256), 0, s,
                       reinterpret_cast<const float4*>(x), reinterpret_cast<float4*>(y), d, C4);
  } else {
    if (static_cast<long long>(B) * C > 65535) return DBEV_EINVAL;
    hipLaunchKernelGGL(up_fwd_nchw, dim3(dbev_ceil_div(OW, 64), dbev_ceil_div(OH, 4), B * C), dim3(64, 4), 0, s, x, y, d);
  }
  DBEV_LAUNCH_CHECK();
  return 0;
}

extern "C" int dbev_upsample_bilinear_ac_backward(const float* grad_y, float* grad_x, int B, int C, int IH,
                                                  int IW, int OH, int OW, int channels_last,
                                                  dbevStream_t stream) {
  UpDims d;
  if (!mk(B, C, IH, IW, OH, OW, &d)) return DBEV_EINVAL;
  hipStream_t s = dbev_stream(stream);
  if (channels_last) {
    if (C & 3) return DBEV_EINVAL;
    const int C4 = C >> 2;
    hipLaunchKernelGGL(up_bwd_nhwc, dim3(dbev_ceil_div(static_cast<long long>(IW) * C4, 256), IH, B), dim3(256), 0, s,
                       reinterpret_cast<const float4*>(grad_y), reinterpret_cast<float4*>(grad_x), d, C4);
  } else {
    if (static_cast<long long>(B) * C > 65535) return DBEV_EINVAL;
    hipLaunchKernelGGL(up_bwd_nchw, dim3(dbev_ceil_div(IW, 64), dbev_ceil_div(IH, 4), B * C), dim3(64, 4), 0, s, grad_y, grad_x, d);
  }
  DBEV_LAUNCH_CHECK();
  return 0;
}

extern "C" int dbev_grid_sample_bilinear_nhwc(const float* x_nhwc, const float* grid_xy, int N, int C, int H, int W, int Ho, int Wo,
                                              float* y_nhwc, dbevStream_t stream) {
  if (N <= 0 || C <= 0 || (C & 3) || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0 || x_nhwc == nullptr || grid_xy == nullptr ||
      y_nhwc == nullptr)
    return DBEV_EINVAL;
  const long long npix = static_cast<long long>(N) * Ho * Wo;
  const int C4 = C >> 2;
  hipLaunchKernelGGL(grid_sample_bilinear_nhwc, dim3(dbev_ceil_div(npix * C4, 256)), dim3(256), 0, dbev_stream(stream),
                     reinterpret_cast<const float4*>(x_nhwc), reinterpret_cast<const float2*>(grid_xy), reinterpret_cast<float4*>(y_nhwc),
                     H, W, C4, npix, Ho * Wo);
  DBEV_LAUNCH_CHECK();
  return 0;
}
