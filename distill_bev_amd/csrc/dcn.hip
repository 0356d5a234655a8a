// Modulated deformable convolution (DCNv2) sampling stage for gfx950, channels-last.
//
// Replaces mmcv-full 1.6.0 `modulated_deform_conv` (un-vendored dependency of the reference; call
// site mmdet3d/models/necks/view_transformer_mine.py:298-306,325-329: 3x3, 256 -> 256 channels,
// deform_groups = 1 on the [B*6, 256, 16, 44] depth feature).  Algorithm restated from the published
// mmcv kernels (modulated_deform_conv_cuda_kernel.cuh: dmcn_im2col_bilinear,
// modulated_deformable_im2col / col2im / col2im_coord):
//   p_k   = (ho*stride - pad + i*dil + dy_k, wo*stride - pad + j*dil + dx_k)       k = i*kw + j
//   val_k = bilinear(x[n, :, :, c], p_k) with zero padding, 0 unless -1 < p < (H, W)
//   col[n, ho, wo, k, c] = sigmoid(logit_k) * val_k
// and the contraction of (k, c) with the weight is left to the MFMA GEMM (MIOpen 1x1 conv on the
// channels-last view of `cols`) -- the sampling is HBM/L2-bound gather work, the GEMM is not.
//
// Layout: x f32[N,H,W,C] (NHWC), om f32[N,Ho,Wo,3K] = the raw output of the reference's conv_offset
// (channels 2k, 2k+1 = dy_k, dx_k -- mmcv's cat(o1, o2) is the identity on the first 2K channels --
// and channels 2K+k = mask logits; the sigmoid is fused here), cols f32[N*Ho*Wo, K*C] (k-major), i.e.
// the NHWC image of a [N, K*C, Ho, Wo] tensor.  One lane owns 4 channels (float4) of one output pixel
// and walks the K taps: every corner fetch / column store of a lane group is one contiguous C*4-byte row.
#include "prims.h"

namespace {

struct DcnDims {
  int N, C4, H, W, Ho, Wo, kh, kw, stride, pad, dil;
};

struct Tap {
  float w1, w2, w3, w4;      // bilinear weights of (lo,lo) (lo,hi) (hi,lo) (hi,hi)
  float hh, hw, lh, lw;
  int o1, o2, o3, o4;        // pixel offsets (h*W + w) of the corners, -1 = outside the image
};

__device__ __forceinline__ Tap make_tap(float h, float w, int H, int W) {
  Tap t;
  t.o1 = t.o2 = t.o3 = t.o4 = -1;
  t.w1 = t.w2 = t.w3 = t.w4 = 0.f;
  t.hh = t.hw = t.lh = t.lw = 0.f;
  if (h > -1.f && w > -1.f && h < static_cast<float>(H) && w < static_cast<float>(W)) {
    const float hf = floorf(h), wf = floorf(w);
    const int hl = static_cast<int>(hf), wl = static_cast<int>(wf);
    const int hi = hl + 1, wi = wl + 1;
    t.lh = h - hf;  t.lw = w - wf;
    t.hh = 1.f - t.lh;  t.hw = 1.f - t.lw;
    t.w1 = t.hh * t.hw;  t.w2 = t.hh * t.lw;  t.w3 = t.lh * t.hw;  t.w4 = t.lh * t.lw;
    if (hl >= 0 && wl >= 0) t.o1 = hl * W + wl;
    if (hl >= 0 && wi <= W - 1) t.o2 = hl * W + wi;
    if (hi <= H - 1 && wl >= 0) t.o3 = hi * W + wl;
    if (hi <= H - 1 && wi <= W - 1) t.o4 = hi * W + wi;
  }
  return t;
}

__device__ __forceinline__ float4 ld4(const float4* __restrict__ img, int off, int C4, int q) {
  return off >= 0 ? img[static_cast<size_t>(off) * C4 + q] : make_float4(0.f, 0.f, 0.f, 0.f);
}

__device__ __forceinline__ float sigmoidf(float z) { return 1.f / (1.f + expf(-z)); }

__device__ __forceinline__ float dot4(const float4& a, const float4& b) {
  return fmaf(a.x, b.x, fmaf(a.y, b.y, fmaf(a.z, b.z, a.w * b.w)));
}

// G = lanes per output pixel (min(C4, 64)); lane q walks channels q, q+G, ...
__global__ __launch_bounds__(256) void dcn_im2col_nhwc(const float4* __restrict__ x, const float* __restrict__ om,
                                                       float4* __restrict__ cols, DcnDims d, int G, int rows) {
  const long long idx = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  const int row = static_cast<int>(idx / G);
  const int q0 = static_cast<int>(idx - static_cast<long long>(row) * G);
  if (row >= rows) return;
  const int wo = row % d.Wo;
  const int t = row / d.Wo;
  const int ho = t % d.Ho;
  const int n = t / d.Ho;
  const int K = d.kh * d.kw;
  const float* omr = om + static_cast<size_t>(row) * 3 * K;
  const float4* img = x + static_cast<size_t>(n) * d.H * d.W * d.C4;
  float4* out = cols + static_cast<size_t>(row) * K * d.C4;
  const float hb = static_cast<float>(ho * d.stride - d.pad), wb = static_cast<float>(wo * d.stride - d.pad);
#pragma unroll 3
  for (int k = 0; k < K; ++k) {
    const int i = k / d.kw, j = k - i * d.kw;
    const Tap tp = make_tap(hb + static_cast<float>(i * d.dil) + omr[2 * k],
                            wb + static_cast<float>(j * d.dil) + omr[2 * k + 1], d.H, d.W);
    const float m = sigmoidf(omr[2 * K + k]);
    for (int q = q0; q < d.C4; q += G) {
      const float4 v1 = ld4(img, tp.o1, d.C4, q), v2 = ld4(img, tp.o2, d.C4, q);
      const float4 v3 = ld4(img, tp.o3, d.C4, q), v4 = ld4(img, tp.o4, d.C4, q);
      float4 r;
      r.x = m * (tp.w1 * v1.x + tp.w2 * v2.x + tp.w3 * v3.x + tp.w4 * v4.x);
      r.y = m * (tp.w1 * v1.y + tp.w2 * v2.y + tp.w3 * v3.y + tp.w4 * v4.y);
      r.z = m * (tp.w1 * v1.z + tp.w2 * v2.z + tp.w3 * v3.z + tp.w4 * v4.z);
      r.w = m * (tp.w1 * v1.w + tp.w2 * v2.w + tp.w3 * v3.w + tp.w4 * v4.w);
      st_nt(out + static_cast<size_t>(k) * d.C4 + q, r);
    }
  }
}

__device__ __forceinline__ void atomic_add4(float4* __restrict__ img, int off, int C4, int q, const float4& g, float s) {
  if (off < 0) return;
  float* p = reinterpret_cast<float*>(img + static_cast<size_t>(off) * C4 + q);
  unsafeAtomicAdd(p + 0, g.x * s);   // global_atomic_add_f32, no return value
  unsafeAtomicAdd(p + 1, g.y * s);
  unsafeAtomicAdd(p + 2, g.z * s);
  unsafeAtomicAdd(p + 3, g.w * s);
}

// grad_om[row, 2k | 2k+1 | 2K+k] = d/d(dy_k), d/d(dx_k), d/d(logit_k), reduced over the channels with a
// lane-group shuffle tree.  SCATTER_GX (fallback for images whose channel slice does not fit the LDS of
// dcn_col2im_gx_lds): grad_x (pre-zeroed) += scatter of grad_cols through the bilinear weights with global
// float atomics, like the mmcv kernel (several output pixels hit the same input pixel) -- 86 G atomics/s,
// 3.7 ms at the depth-head shape, which is why the LDS kernel below exists.
template <bool SCATTER_GX>
__global__ __launch_bounds__(256) void dcn_col2im_nhwc(const float4* __restrict__ gcols, const float4* __restrict__ x,
                                                       const float* __restrict__ om, float4* __restrict__ gx,
                                                       float* __restrict__ gom, DcnDims d, int G, int rows) {
  const long long idx = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  const int row_raw = static_cast<int>(idx / G);
  const int q0 = static_cast<int>(idx - static_cast<long long>(row_raw) * G);
  const bool active = row_raw < rows;
  const int row = active ? row_raw : 0;
  const int wo = row % d.Wo;
  const int t = row / d.Wo;
  const int ho = t % d.Ho;
  const int n = t / d.Ho;
  const int K = d.kh * d.kw;
  const float* omr = om + static_cast<size_t>(row) * 3 * K;
  const float4* img = x + static_cast<size_t>(n) * d.H * d.W * d.C4;
  float4* gimg = gx + static_cast<size_t>(n) * d.H * d.W * d.C4;
  const float4* gin = gcols + static_cast<size_t>(row) * K * d.C4;
  float* gout = gom + static_cast<size_t>(row) * 3 * K;
  const float hb = static_cast<float>(ho * d.stride - d.pad), wb = static_cast<float>(wo * d.stride - d.pad);

  for (int k = 0; k < K; ++k) {
    const int i = k / d.kw, j = k - i * d.kw;
    const Tap tp = make_tap(hb + static_cast<float>(i * d.dil) + omr[2 * k],
                            wb + static_cast<float>(j * d.dil) + omr[2 * k + 1], d.H, d.W);
    const float m = sigmoidf(omr[2 * K + k]);
    float gm = 0.f, gh = 0.f, gw = 0.f;
    if (active) {
      for (int q = q0; q < d.C4; q += G) {
        const float4 g = gin[static_cast<size_t>(k) * d.C4 + q];
        const float4 v1 = ld4(img, tp.o1, d.C4, q), v2 = ld4(img, tp.o2, d.C4, q);
        const float4 v3 = ld4(img, tp.o3, d.C4, q), v4 = ld4(img, tp.o4, d.C4, q);
        const float d1 = dot4(g, v1), d2 = dot4(g, v2), d3 = dot4(g, v3), d4 = dot4(g, v4);
        gm += tp.w1 * d1 + tp.w2 * d2 + tp.w3 * d3 + tp.w4 * d4;
        gh += tp.hw * (d3 - d1) + tp.lw * (d4 - d2);       // d val / d h
        gw += tp.hh * (d2 - d1) + tp.lh * (d4 - d3);       // d val / d w
        if (SCATTER_GX) {
          atomic_add4(gimg, tp.o1, d.C4, q, g, m * tp.w1);
          atomic_add4(gimg, tp.o2, d.C4, q, g, m * tp.w2);
          atomic_add4(gimg, tp.o3, d.C4, q, g, m * tp.w3);
          atomic_add4(gimg, tp.o4, d.C4, q, g, m * tp.w4);
        }
      }
    }
    for (int o = G >> 1; o > 0; o >>= 1) {
      gm += __shfl_xor(gm, o);
      gh += __shfl_xor(gh, o);
      gw += __shfl_xor(gw, o);
    }
    if (active && q0 == 0) {
      gout[2 * k] = gh * m;
      gout[2 * k + 1] = gw * m;
      gout[2 * K + k] = gm * m * (1.f - m);
    }
  }
}

// grad_x without global atomics: one workgroup owns (image n, slice of 4*SC4 channels), keeps that slice of
// the whole H x W image in LDS (16x44 px x 16 ch = 45 KB), walks all Ho*Wo*K taps of the image adding
// g * mask * w_corner with LDS atomics (ds_add_f32), then writes the slice once -- every grad_x element is
// written exactly once, no zero-fill needed.  Needs no x values (the bilinear weights depend on om only).
__device__ __forceinline__ void lds_add4(float* __restrict__ acc, int off, int SC4, int ql, const float4& g, float s) {
  if (off < 0) return;
  float* p = acc + (static_cast<size_t>(off) * SC4 + ql) * 4;
  unsafeAtomicAdd(p + 0, g.x * s);
  unsafeAtomicAdd(p + 1, g.y * s);
  unsafeAtomicAdd(p + 2, g.z * s);
  unsafeAtomicAdd(p + 3, g.w * s);
}

__global__ __launch_bounds__(512) void dcn_col2im_gx_lds(const float4* __restrict__ gcols, const float* __restrict__ om,
                                                         float4* __restrict__ gx, DcnDims d, int SC4) {
  extern __shared__ __attribute__((aligned(16))) float acc[];     // [H*W][SC4] float4
  const int nslices = d.C4 / SC4;
  const int n = blockIdx.x / nslices;
  const int qb = (blockIdx.x - n * nslices) * SC4;
  const int npix = d.H * d.W;
  float4* acc4 = reinterpret_cast<float4*>(acc);
  for (int i = threadIdx.x; i < npix * SC4; i += blockDim.x) acc4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  __syncthreads();
  const int K = d.kh * d.kw;
  const int ntaps = d.Ho * d.Wo * K;
  const int tpi = blockDim.x / SC4;                                // taps per block iteration
  const int ql = threadIdx.x % SC4;
#pragma unroll 2
  for (int t = threadIdx.x / SC4; t < ntaps; t += tpi) {
    const int r = t / K, k = t - r * K;
    const int ho = r / d.Wo, wo = r - ho * d.Wo;
    const int row = n * d.Ho * d.Wo + r;
    const float* omr = om + static_cast<size_t>(row) * 3 * K;
    const float4 g = gcols[(static_cast<size_t>(row) * K + k) * d.C4 + qb + ql];
    const int i = k / d.kw, j = k - i * d.kw;
    const Tap tp = make_tap(static_cast<float>(ho * d.stride - d.pad + i * d.dil) + omr[2 * k],
                            static_cast<float>(wo * d.stride - d.pad + j * d.dil) + omr[2 * k + 1], d.H, d.W);
    const float m = sigmoidf(omr[2 * K + k]);
    lds_add4(acc, tp.o1, SC4, ql, g, m * tp.w1);
    lds_add4(acc, tp.o2, SC4, ql, g, m * tp.w2);
    lds_add4(acc, tp.o3, SC4, ql, g, m * tp.w3);
    lds_add4(acc, tp.o4, SC4, ql, g, m * tp.w4);
  }
  __syncthreads();
  float4* gimg = gx + static_cast<size_t>(n) * npix * d.C4;
  for (int i = threadIdx.x; i < npix * SC4; i += blockDim.x) {
    const int pix = i / SC4, q = i - pix * SC4;
    gimg[static_cast<size_t>(pix) * d.C4 + qb + q] = acc4[i];
  }
}

// ---- grad_x as a deterministic gather (C/4 a power of two in [8, 64]) ------------------------------------
// Every (output pixel, tap) touches up to 4 input pixels.  Group those "tap corners" by input pixel with the
// library's CSR primitive (int histogram -> scan -> fill -> per-pixel sort by corner id), then one lane group
// per input pixel adds  coef * grad_cols[tap, :]  over its list in ascending id order: no float atomics, a fixed
// summation order (bit-reproducible), and every grad_x element written exactly once.  The coefficient
// mask * w_corner is recomputed from the offsets by the lane that owns the list entry (one make_tap per entry).
// corner id e = ((row * K + k) << 2) | corner.
size_t align_up(size_t x) { return (x + 255) & ~static_cast<size_t>(255); }

struct GxLayout { size_t count, start, list, sorted, scanws, sortws, total; };
GxLayout gx_layout(long long npix, long long nent) {
  GxLayout L;
  size_t o = 0;
  L.count = o;  o += align_up(sizeof(int) * npix);
  L.start = o;  o += align_up(sizeof(int) * (npix + 1));
  L.list = o;   o += align_up(sizeof(int) * nent);
  L.sorted = o; o += align_up(sizeof(int) * nent);
  L.scanws = o; o += align_up(sizeof(int) * dbev::scan_workspace_ints(npix));
  L.sortws = o; o += align_up(sizeof(int) * dbev::segment_sort_workspace_ints(nent));
  L.total = o;
  return L;
}

__device__ __forceinline__ Tap tap_of(const float* __restrict__ om, const DcnDims& d, int row, int k) {
  const int K = d.kh * d.kw;
  const int wo = row % d.Wo;
  const int ho = (row / d.Wo) % d.Ho;
  const int i = k / d.kw, j = k - i * d.kw;
  const float* omr = om + static_cast<size_t>(row) * 3 * K;
  return make_tap(static_cast<float>(ho * d.stride - d.pad + i * d.dil) + omr[2 * k],
                  static_cast<float>(wo * d.stride - d.pad + j * d.dil) + omr[2 * k + 1], d.H, d.W);
}

template <bool FILL>
__global__ __launch_bounds__(256) void dcn_corner_bin(const float* __restrict__ om, DcnDims d, int ntaps /* rows*K */,
                                                      const int* __restrict__ start, int* __restrict__ count,
                                                      unsigned* __restrict__ list) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= ntaps) return;
  const int K = d.kh * d.kw;
  const int row = t / K, k = t - row * K;
  const int n = row / (d.Ho * d.Wo);
  const Tap tp = tap_of(om, d, row, k);
  const int base = n * d.H * d.W;
  const int offs[4] = {tp.o1, tp.o2, tp.o3, tp.o4};
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    if (offs[c] < 0) continue;
    const int pix = base + offs[c];
    if (FILL) list[start[pix] + atomicSub(&count[pix], 1) - 1] = (static_cast<unsigned>(t) << 2) | c;
    else atomicAdd(&count[pix], 1);
  }
}

// G = C4 lanes per input pixel, 64 / G pixels per wave.
__global__ __launch_bounds__(256) void dcn_gx_gather(const float4* __restrict__ gcols, const float* __restrict__ om,
                                                     const int* __restrict__ start, const unsigned* __restrict__ ents,
                                                     float4* __restrict__ gx, DcnDims d, int npix) {
  constexpr int DU = 8;
  const int G = d.C4;
  const int lane = threadIdx.x & 63;
  const int g0 = lane & ~(G - 1), q = lane & (G - 1);
  const int pix = static_cast<int>((static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x) / G);
  if (pix >= npix) return;                                  // whole groups leave together
  const int K = d.kh * d.kw;
  const int st = start[pix];
  const int L = start[pix + 1] - st;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int j0 = 0; j0 < L; j0 += G) {
    const int nb = min(G, L - j0);
    unsigned mytap = 0u;
    float mycoef = 0.f;
    if (q < nb) {
      const unsigned e = ents[st + j0 + q];
      mytap = e >> 2;
      const int row = static_cast<int>(mytap / K), k = static_cast<int>(mytap - static_cast<unsigned>(row) * K);
      const Tap tp = tap_of(om, d, row, k);
      const int c = e & 3;
      const float w = c == 0 ? tp.w1 : (c == 1 ? tp.w2 : (c == 2 ? tp.w3 : tp.w4));
      mycoef = sigmoidf(om[static_cast<size_t>(row) * 3 * K + 2 * K + k]) * w;
    }
    for (int h = 0; h < nb; h += DU) {                      // uniform inside the group
      float4 v[DU];
#pragma unroll
      for (int u = 0; u < DU; ++u) {
        const unsigned t = __shfl(mytap, g0 | ((h + u) & (G - 1)));
        v[u] = (h + u) < nb ? gcols[static_cast<size_t>(t) * G + q] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < DU; ++u) {
        const float c = __shfl(mycoef, g0 | ((h + u) & (G - 1)));
        if ((h + u) < nb) { acc.x = fmaf(c, v[u].x, acc.x); acc.y = fmaf(c, v[u].y, acc.y);
                            acc.z = fmaf(c, v[u].z, acc.z); acc.w = fmaf(c, v[u].w, acc.w); }
      }
    }
  }
  gx[static_cast<size_t>(pix) * G + q] = acc;
}

bool gather_ok(int C4) { return C4 >= 8 && C4 <= 64 && (C4 & (C4 - 1)) == 0; }

constexpr int DCN_LDS_BYTES = 64 * 1024;   // dynamic LDS budget of the slice kernel (3 workgroups per CU at 45 KB)

bool dims_ok(int N, int C, int H, int W, int Ho, int Wo, int kh, int kw, int stride, int pad, int dil, int* G) {
  if (N <= 0 || C <= 0 || (C & 3) || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0 || kh <= 0 || kw <= 0 || stride <= 0 ||
      pad < 0 || dil <= 0)
    return false;
  const int C4 = C >> 2;
  if (C4 <= 64) {
    if (C4 & (C4 - 1)) return false;     // lane groups are powers of two
    *G = C4;
  } else {
    if (C4 & 63) return false;
    *G = 64;
  }
  if (static_cast<long long>(N) * Ho * Wo > 0x7fffffffLL / 64) return false;
  return (H + 2 * pad - dil * (kh - 1) - 1) / stride + 1 == Ho && (W + 2 * pad - dil * (kw - 1) - 1) / stride + 1 == Wo;
}

}  // namespace

extern "C" int dbev_dcnv2_im2col(const float* x_nhwc, const float* offset_mask_nhwc, float* cols, int N, int C,
                                 int H, int W, int Ho, int Wo, int kh, int kw, int stride, int pad, int dil,
                                 dbevStream_t stream) {
  int G = 0;
  if (!dims_ok(N, C, H, W, Ho, Wo, kh, kw, stride, pad, dil, &G)) return DBEV_EINVAL;
  if (x_nhwc == nullptr || offset_mask_nhwc == nullptr || cols == nullptr) return DBEV_EINVAL;
  const DcnDims d{N, C >> 2, H, W, Ho, Wo, kh, kw, stride, pad, dil};
  const int rows = N * Ho * Wo;
  const long long threads = static_cast<long long>(rows) * G;
  hipLaunchKernelGGL(dcn_im2col_nhwc, dim3(dbev_ceil_div(threads, 256)), dim3(256), 0, dbev_stream(stream),
                     reinterpret_cast<const float4*>(x_nhwc), offset_mask_nhwc, reinterpret_cast<float4*>(cols), d, G,
                     rows);
  DBEV_LAUNCH_CHECK();
  return 0;
}

extern "C" size_t dbev_dcnv2_col2im_workspace_bytes(int N, int C, int H, int W, int Ho, int Wo, int kh, int kw) {
  if (N <= 0 || C <= 0 || (C & 3) || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0 || kh <= 0 || kw <= 0) return 0;
  if (!gather_ok(C >> 2)) return 256;     // LDS-slice / atomic paths need no scratch
  const long long npix = static_cast<long long>(N) * H * W, nent = static_cast<long long>(N) * Ho * Wo * kh * kw * 4;
  if (nent > 0x3fffffffLL) return 256;
  return gx_layout(npix, nent).total;
}

extern "C" int dbev_dcnv2_col2im(const float* grad_cols, const float* x_nhwc, const float* offset_mask_nhwc,
                                 float* grad_x_nhwc, float* grad_offset_mask_nhwc, int N, int C, int H, int W, int Ho,
                                 int Wo, int kh, int kw, int stride, int pad, int dil, void* workspace,
                                 size_t workspace_bytes, dbevStream_t stream) {
  int G = 0;
  if (!dims_ok(N, C, H, W, Ho, Wo, kh, kw, stride, pad, dil, &G)) return DBEV_EINVAL;
  if (grad_cols == nullptr || x_nhwc == nullptr || offset_mask_nhwc == nullptr || grad_x_nhwc == nullptr ||
      grad_offset_mask_nhwc == nullptr)
    return DBEV_EINVAL;
  const DcnDims d{N, C >> 2, H, W, Ho, Wo, kh, kw, stride, pad, dil};
  const int rows = N * Ho * Wo;
  const long long threads = static_cast<long long>(rows) * G;
  hipStream_t s = dbev_stream(stream);
  const long long npix = static_cast<long long>(N) * H * W, nent = static_cast<long long>(rows) * kh * kw * 4;
  if (gather_ok(d.C4) && nent <= 0x3fffffffLL) {
    const GxLayout Lw = gx_layout(npix, nent);
    if (workspace == nullptr || workspace_bytes < Lw.total) return DBEV_EINVAL;
    char* ws = static_cast<char*>(workspace);
    int* count = reinterpret_cast<int*>(ws + Lw.count);
    int* start = reinterpret_cast<int*>(ws + Lw.start);
    unsigned* list = reinterpret_cast<unsigned*>(ws + Lw.list);
    unsigned* sorted = reinterpret_cast<unsigned*>(ws + Lw.sorted);
    const int ntaps = rows * kh * kw;
    hipLaunchKernelGGL((dcn_col2im_nhwc<false>), dim3(dbev_ceil_div(threads, 256)), dim3(256), 0, s,
                       reinterpret_cast<const float4*>(grad_cols), reinterpret_cast<const float4*>(x_nhwc),
                       offset_mask_nhwc, reinterpret_cast<float4*>(grad_x_nhwc), grad_offset_mask_nhwc, d, G, rows);
    DBEV_HIP_TRY(hipMemsetAsync(count, 0, sizeof(int) * npix, s));
    hipLaunchKernelGGL((dcn_corner_bin<false>), dim3(dbev_ceil_div(ntaps, 256)), dim3(256), 0, s, offset_mask_nhwc, d,
                       ntaps, start, count, list);
    int rc = dbev::exclusive_scan_i32(count, start, npix, false, nullptr, reinterpret_cast<int*>(ws + Lw.scanws), s);
    if (rc) return rc;
    hipLaunchKernelGGL((dcn_corner_bin<true>), dim3(dbev_ceil_div(ntaps, 256)), dim3(256), 0, s, offset_mask_nhwc, d,
                       ntaps, start, count, list);
    rc = dbev::segment_sort_u32(start, list, sorted, static_cast<int>(npix), reinterpret_cast<int*>(ws + Lw.sortws), s);
    if (rc) return rc;
    hipLaunchKernelGGL(dcn_gx_gather, dim3(dbev_ceil_div(npix * d.C4, 256)), dim3(256), 0, s,
                       reinterpret_cast<const float4*>(grad_cols), offset_mask_nhwc, start, sorted,
                       reinterpret_cast<float4*>(grad_x_nhwc), d, static_cast<int>(npix));
    DBEV_LAUNCH_CHECK();
    return 0;
  }
  int SC4 = 0;                                   // float4 columns per workgroup slice: largest of 4, 2, 1 that fits
  for (int c = 4; c >= 1; c >>= 1)
    if (d.C4 % c == 0 && static_cast<long long>(H) * W * c * 16 <= DCN_LDS_BYTES) { SC4 = c; break; }
  if (SC4 > 0) {
    hipLaunchKernelGGL((dcn_col2im_nhwc<false>), dim3(dbev_ceil_div(threads, 256)), dim3(256), 0, s,
                       reinterpret_cast<const float4*>(grad_cols), reinterpret_cast<const float4*>(x_nhwc),
                       offset_mask_nhwc, reinterpret_cast<float4*>(grad_x_nhwc), grad_offset_mask_nhwc, d, G, rows);
    hipLaunchKernelGGL(dcn_col2im_gx_lds, dim3(N * (d.C4 / SC4)), dim3(512), static_cast<size_t>(H) * W * SC4 * 16, s,
                       reinterpret_cast<const float4*>(grad_cols), offset_mask_nhwc,
                       reinterpret_cast<float4*>(grad_x_nhwc), d, SC4);
  } else {
    DBEV_HIP_TRY(hipMemsetAsync(grad_x_nhwc, 0, sizeof(float) * static_cast<size_t>(N) * H * W * C, s));
    hipLaunchKernelGGL((dcn_col2im_nhwc<true>), dim3(dbev_ceil_div(threads, 256)), dim3(256), 0, s,
                       reinterpret_cast<const float4*>(grad_cols), reinterpret_cast<const float4*>(x_nhwc),
                       offset_mask_nhwc, reinterpret_cast<float4*>(grad_x_nhwc), grad_offset_mask_nhwc, d, G, rows);
  }
  DBEV_LAUNCH_CHECK();
  return 0;
}
