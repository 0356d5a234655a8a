// Foreground-masked BEV feature distillation (FGD) kernels for gfx950.
//
// Reference: mmdet3d/models/detectors/bevdet_distill.py
//   foreground_scale_mask :755-843   host numpy + numba points_in_rbbox over H*W cell points x M
//                                    boxes per sample per distill position, tensor->numpy->
//                                    tensor->.to(device) round trip inside the step
//   fgd_distill_loss      :1084-1108 attention maps (mean_c|f|, mean_hw|f|)
//                         :1253-1262,1282-1287  (S-T)^2 recomputed 3x, each times a broadcast
//                                    mask and reduced
// Here:
//   dbev_fg_scale_mask      : device rasteriser.  The 6 face planes of each (flattened) box are
//                             tiny host numpy work (bit-identical to box_np_ops); the
//                             H*W x M x 6 sign tests run on the GPU with the reference's exact
//                             fp32 expression ((px*nx + py*ny) + pz*nz) + d  (no FMA), so the
//                             masks are bit-exact.  No D2H/H2D of masks.
//   dbev_abs_mean_maps      : ONE pass over a feature map -> per-pixel mean_c|x| and
//                             per-channel mean_hw|x| (inputs of both attention softmaxes).
//   dbev_fgd_masked_mse_*   : ONE pass over (S, T) produces all three weighted sums
//                             sum (S-T)^2 W_fg, sum (S-T)^2 W_bg, sum (S-T)^2 W_fp c_att[c]
//                             (and one pass for dS); no (S-T)^2 tensor is materialised.
//                             Two-stage fixed-order reduction -> deterministic.
// All are HBM-streaming kernels: algorithmic bytes 4*HW*(Cs+Ct) + 3*4*HW per sample.
#include "common.h"

namespace {

constexpr int MAX_BOX_TILE = 64;  // boxes staged in LDS per round (64*6*4 floats = 6 KiB)

__global__ __launch_bounds__(256) void fg_mask_kernel(const float4* __restrict__ planes,
                                                      const float* __restrict__ box_scale,
                                                      const int* __restrict__ box_off,
                                                      const float* __restrict__ xs,
                                                      const float* __restrict__ ys, int H, int W,
                                                      float* __restrict__ fg, float* __restrict__ fg_scale,
                                                      int* __restrict__ fg_count) {
#pragma clang fp contract(off)
  __shared__ float4 sp[MAX_BOX_TILE * 6];
  __shared__ float ssc[MAX_BOX_TILE];
  __shared__ int scount[4];
  const int b = blockIdx.y;
  const int cell = blockIdx.x * blockDim.x + threadIdx.x;   // iy * W + ix
  const bool valid = cell < H * W;
  const int iy = valid ? cell / W : 0;
  const int ix = valid ? cell - iy * W : 0;
  const float px = xs[ix], py = ys[iy], pz = 0.5f;
  const int j0 = box_off[b], j1 = box_off[b + 1];
  bool hit = false;
  float sc = 0.f;
  for (int base = j0; base < j1; base += MAX_BOX_TILE) {
    const int nb = min(MAX_BOX_TILE, j1 - base);
    __syncthreads();
    for (int i = threadIdx.x; i < nb * 6; i += blockDim.x) sp[i] = planes[static_cast<size_t>(base) * 6 + i];
    for (int i = threadIdx.x; i < nb; i += blockDim.x) ssc[i] = box_scale[base + i];
    __syncthreads();
    if (!hit) {
      for (int j = 0; j < nb; ++j) {
        bool inside = true;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          const float4 pl = sp[j * 6 + k];
          const float sign = ((px * pl.x + py * pl.y) + pz * pl.z) + pl.w;
          inside = inside && (sign < 0.f);       // sign >= 0 (or NaN) -> outside (:750-752)
        }
        if (inside) { hit = true; sc = ssc[j]; break; }   // lowest box index wins (:798-801)
      }
    }
  }
  if (valid) {
    fg[static_cast<size_t>(b) * H * W + cell] = hit ? 1.f : 0.f;
    fg_scale[static_cast<size_t>(b) * H * W + cell] = hit ? sc : 0.f;
  }
  const unsigned long long m = __ballot(valid && hit);
  if ((threadIdx.x & 63) == 0) scount[threadIdx.x >> 6] = __popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(&fg_count[b], scount[0] + scount[1] + scount[2] + scount[3]);
}

__global__ __launch_bounds__(256) void bg_scale_kernel(const int* __restrict__ fg_count, int HW,
                                                       float* __restrict__ bg_scale) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= HW) return;
  // 1.0 / (H*W - n_fg) in float64, stored as float32 (:823-825, :839)
  const double v = 1.0 / static_cast<double>(HW - fg_count[b]);
  bg_scale[static_cast<size_t>(b) * HW + i] = static_cast<float>(v);
}

// ---- |x| means --------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// grid (ceil(HW/1024), ceil(C/64), B); 256 threads x 4 consecutive pixels (float4 when HW % 4 == 0).
// A block covers 1024 pixels x 64 channels: per-channel sums over its pixels (one wave reduction
// per channel per 256 pixels) and per-pixel partial sums over its channels; two tiny fixed-order
// kernels finish both means.  dynamic LDS 4*64 floats.
constexpr int AM_PX = 1024;
constexpr int AM_CH = 64;

__global__ __launch_bounds__(256) void abs_mean_kernel(const float* __restrict__ x, int C, int HW,
                                                       float* __restrict__ pixpart,
                                                       float* __restrict__ chpart) {
  __shared__ float chs[4][AM_CH];
  const int b = blockIdx.z, tile = blockIdx.x, ntile = gridDim.x, chunk = blockIdx.y, nchunk = gridDim.y;
  const int c0 = chunk * AM_CH, c1 = min(C, c0 + AM_CH);
  const int p0 = tile * AM_PX + threadIdx.x * 4;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const bool vec = (HW & 3) == 0 && p0 + 3 < HW;
  const float* xb = x + static_cast<size_t>(b) * C * HW + p0;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int c = c0; c < c1; ++c) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    const float* xc = xb + static_cast<size_t>(c) * HW;
    if (vec) {
      v = *reinterpret_cast<const float4*>(xc);
    } else {
      if (p0 < HW) v.x = xc[0];
      if (p0 + 1 < HW) v.y = xc[1];
      if (p0 + 2 < HW) v.z = xc[2];
      if (p0 + 3 < HW) v.w = xc[3];
    }
    v.x = fabsf(v.x); v.y = fabsf(v.y); v.z = fabsf(v.z); v.w = fabsf(v.w);
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    const float s = wave_sum((v.x + v.y) + (v.z + v.w));
    if (lane == 0) chs[w][c - c0] = s;
  }
  float* po = pixpart + (static_cast<size_t>(b) * nchunk + chunk) * HW + p0;
  if (vec) {
    *reinterpret_cast<float4*>(po) = acc;
  } else {
    if (p0 < HW) po[0] = acc.x;
    if (p0 + 1 < HW) po[1] = acc.y;
    if (p0 + 2 < HW) po[2] = acc.z;
    if (p0 + 3 < HW) po[3] = acc.w;
  }
  __syncthreads();
  if (threadIdx.x < c1 - c0)
    chpart[(static_cast<size_t>(b) * ntile + tile) * C + c0 + threadIdx.x] =
        (chs[0][threadIdx.x] + chs[1][threadIdx.x]) + (chs[2][threadIdx.x] + chs[3][threadIdx.x]);
}

__global__ __launch_bounds__(256) void abs_mean_pix_final(const float* __restrict__ pixpart, int C, int HW,
                                                          int nchunk, float* __restrict__ pix) {
  const int b = blockIdx.y;
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= HW) return;
  float s = 0.f;
  for (int k = 0; k < nchunk; ++k) s += pixpart[(static_cast<size_t>(b) * nchunk + k) * HW + p];
  pix[static_cast<size_t>(b) * HW + p] = s / static_cast<float>(C);
}

// 32 channels x 8 tile phases per workgroup (the NHWC path has 256 row tiles per sample: one thread per channel
// walking them serially was a 30 us latency chain); fixed summation order.  grid (ceil(C/32), B)
__global__ __launch_bounds__(256) void abs_mean_ch_final(const float* __restrict__ chpart, int C, int HW,
                                                         int ntile, float* __restrict__ ch) {
  __shared__ float red[8][32];
  const int b = blockIdx.y;
  const int cl = threadIdx.x & 31, ph = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  float s = 0.f;
  if (c < C)
    for (int t = ph; t < ntile; t += 8) s += chpart[(static_cast<size_t>(b) * ntile + t) * C + c];
  red[ph][cl] = s;
  __syncthreads();
  if (ph == 0 && c < C) {
    for (int p = 1; p < 8; ++p) s += red[p][cl];
    ch[b * C + c] = s / static_cast<float>(HW);
  }
}

// ---- masked MSE -------------------------------------------------------------------------
constexpr int MSE_CCHUNK = 16;   // channels per block

// grid (HW/1024, ceil(C/16), B)
__global__ __launch_bounds__(256) void masked_mse_fwd(const float4* __restrict__ S,
                                                      const float4* __restrict__ T,
                                                      const float4* __restrict__ Wfg,
                                                      const float4* __restrict__ Wbg,
                                                      const float4* __restrict__ Wfp,
                                                      const float* __restrict__ Cc, int C, int HW4,
                                                      float* __restrict__ partial) {
  __shared__ float red[3][4];
  const int b = blockIdx.z;
  const int p4 = blockIdx.x * 256 + threadIdx.x;
  const int c0 = blockIdx.y * MSE_CCHUNK;
  const int c1 = min(C, c0 + MSE_CCHUNK);
  float a_fg = 0.f, a_bg = 0.f, a_fp = 0.f;
  if (p4 < HW4) {
    const size_t wo = static_cast<size_t>(b) * HW4 + p4;
    const float4 wf = Wfg[wo], wb = Wbg[wo];
    const float4 wp = Wfp ? Wfp[wo] : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int c = c0; c < c1; ++c) {
      const size_t o = (static_cast<size_t>(b) * C + c) * HW4 + p4;
      const float4 s = S[o], t = T[o];
      const float dx = s.x - t.x, dy = s.y - t.y, dz = s.z - t.z, dw = s.w - t.w;
      const float qx = dx * dx, qy = dy * dy, qz = dz * dz, qw = dw * dw;
      a_fg += (qx * wf.x + qy * wf.y) + (qz * wf.z + qw * wf.w);
      a_bg += (qx * wb.x + qy * wb.y) + (qz * wb.z + qw * wb.w);
      if (Wfp) {
        const float cc = Cc ? Cc[b * C + c] : 1.f;
        a_fp += cc * ((qx * wp.x + qy * wp.y) + (qz * wp.z + qw * wp.w));
      }
    }
  }
  a_fg = wave_sum(a_fg); a_bg = wave_sum(a_bg); a_fp = wave_sum(a_fp);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) { red[0][w] = a_fg; red[1][w] = a_bg; red[2][w] = a_fp; }
  __syncthreads();
  if (threadIdx.x < 3) {
    const int k = threadIdx.x;
    const size_t blk = (static_cast<size_t>(blockIdx.z) * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    partial[blk * 3 + k] = (red[k][0] + red[k][1]) + (red[k][2] + red[k][3]);
  }
}

// single block: fixed-order fp64 sum of the per-block partials -> out[3]
__global__ __launch_bounds__(256) void masked_mse_final(const float* __restrict__ partial, int nblk,
                                                        float* __restrict__ out) {
  __shared__ double red[3][256];
  double a[3] = {0.0, 0.0, 0.0};
  for (int i = threadIdx.x; i < nblk; i += 256) {
    a[0] += partial[i * 3 + 0]; a[1] += partial[i * 3 + 1]; a[2] += partial[i * 3 + 2];
  }
  for (int k = 0; k < 3; ++k) red[k][threadIdx.x] = a[k];
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o)
      for (int k = 0; k < 3; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x < 3) out[threadIdx.x] = static_cast<float>(red[threadIdx.x][0]);
}

// dS = 2 (S-T) (g0 Wfg + g1 Wbg + g2 Cc[c] Wfp)
__global__ __launch_bounds__(256) void masked_mse_bwd(const float4* __restrict__ S,
                                                      const float4* __restrict__ T,
                                                      const float4* __restrict__ Wfg,
                                                      const float4* __restrict__ Wbg,
                                                      const float4* __restrict__ Wfp,
                                                      const float* __restrict__ Cc,
                                                      const float* __restrict__ gsc, int C, int HW4,
                                                      float4* __restrict__ dS) {
  const int b = blockIdx.z;
  const int p4 = blockIdx.x * 256 + threadIdx.x;
  if (p4 >= HW4) return;
  const int c0 = blockIdx.y * MSE_CCHUNK;
  const int c1 = min(C, c0 + MSE_CCHUNK);
  const float g0 = gsc[0], g1 = gsc[1], g2 = gsc[2];
  const size_t wo = static_cast<size_t>(b) * HW4 + p4;
  const float4 wf = Wfg[wo], wb = Wbg[wo];
  const float4 wp = Wfp ? Wfp[wo] : make_float4(0.f, 0.f, 0.f, 0.f);
  float4 base;
  base.x = g0 * wf.x + g1 * wb.x; base.y = g0 * wf.y + g1 * wb.y;
  base.z = g0 * wf.z + g1 * wb.z; base.w = g0 * wf.w + g1 * wb.w;
  for (int c = c0; c < c1; ++c) {
    const size_t o = (static_cast<size_t>(b) * C + c) * HW4 + p4;
    const float4 s = S[o], t = T[o];
    float k = 0.f;
    if (Wfp) k = g2 * (Cc ? Cc[b * C + c] : 1.f);
    float4 r;
    r.x = 2.f * (s.x - t.x) * (base.x + k * wp.x);
    r.y = 2.f * (s.y - t.y) * (base.y + k * wp.y);
    r.z = 2.f * (s.z - t.z) * (base.z + k * wp.z);
    r.w = 2.f * (s.w - t.w) * (base.w + k * wp.w);
    dS[o] = r;
  }
}


// ---- channels-last (NHWC) variants ---------------------------------------------------------
// x, S, T f32[B, HW, C]: a pixel's C channels are contiguous.  One WAVE per pixel row (lane = float4 column,
// NQ = ceil(C/256) columns per lane), 4 waves per workgroup walking one contiguous range of rows: the
// per-pixel weights are wave-uniform scalars, the per-channel factors / channel sums live in registers.
constexpr int NHWC_ROWS_PER_BLOCK = 64;

template <int NQ>
__global__ __launch_bounds__(256) void abs_mean_nhwc(const float4* __restrict__ x, int C4, int HW,
                                                     float* __restrict__ pix, float* __restrict__ chpart,
                                                     float* __restrict__ pool /* mean_c x per pixel, may be null */) {
  __shared__ float4 red[4][64 * NQ];
  const int b = blockIdx.y, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int r0 = blockIdx.x * NHWC_ROWS_PER_BLOCK, r1 = min(HW, r0 + NHWC_ROWS_PER_BLOCK);
  const float invC = 1.f / static_cast<float>(C4 * 4);
  float4 acc[NQ];
#pragma unroll
  for (int j = 0; j < NQ; ++j) acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int r = r0 + w; r < r1; r += 8) {
    float ps[2] = {0.f, 0.f}, sg[2] = {0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int rr = r + 4 * u;
      if (rr < r1) {
        const float4* row = x + (static_cast<size_t>(b) * HW + rr) * C4;
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
          const int q = lane + 64 * j;
          if (q < C4) {
            float4 v = row[q];
            sg[u] += (v.x + v.y) + (v.z + v.w);
            v.x = fabsf(v.x); v.y = fabsf(v.y); v.z = fabsf(v.z); v.w = fabsf(v.w);
            acc[j].x += v.x; acc[j].y += v.y; acc[j].z += v.z; acc[j].w += v.w;
            ps[u] += (v.x + v.y) + (v.z + v.w);
          }
        }
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int rr = r + 4 * u;
      const float t = wave_sum(ps[u]);
      if (lane == 0 && rr < r1) pix[static_cast<size_t>(b) * HW + rr] = t * invC;
      if (pool != nullptr) {                               // uniform
        const float t2 = wave_sum(sg[u]);
        if (lane == 0 && rr < r1) pool[static_cast<size_t>(b) * HW + rr] = t2 * invC;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < NQ; ++j) red[w][lane + 64 * j] = acc[j];
  __syncthreads();
  for (int q = threadIdx.x; q < C4; q += 256) {
    float4 t = red[0][q];
    for (int k = 1; k < 4; ++k) { t.x += red[k][q].x; t.y += red[k][q].y; t.z += red[k][q].z; t.w += red[k][q].w; }
    reinterpret_cast<float4*>(chpart + (static_cast<size_t>(b) * gridDim.x + blockIdx.x) * C4 * 4)[q] = t;
  }
}

template <int NQ>
__global__ __launch_bounds__(256) void masked_mse_fwd_nhwc(const float4* __restrict__ S, const float4* __restrict__ T,
                                                           const float* __restrict__ Wfg, const float* __restrict__ Wbg,
                                                           const float* __restrict__ Wfp, const float* __restrict__ Cc,
                                                           int C4, int HW, float* __restrict__ partial) {
  __shared__ float red[3][4];
  const int b = blockIdx.y, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int r0 = blockIdx.x * NHWC_ROWS_PER_BLOCK, r1 = min(HW, r0 + NHWC_ROWS_PER_BLOCK);
  float4 cc[NQ];
#pragma unroll
  for (int j = 0; j < NQ; ++j) {
    const int q = lane + 64 * j;
    cc[j] = (Cc != nullptr && q < C4) ? reinterpret_cast<const float4*>(Cc + static_cast<size_t>(b) * C4 * 4)[q]
                                      : make_float4(1.f, 1.f, 1.f, 1.f);
  }
  float a_fg = 0.f, a_bg = 0.f, a_fp = 0.f;
  for (int r = r0 + w; r < r1; r += 4) {
    const size_t pr = static_cast<size_t>(b) * HW + r;
    const float wf = Wfg[pr], wb = Wbg[pr], wp = Wfp ? Wfp[pr] : 0.f;
    float qs = 0.f, qp = 0.f;
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
      const int q = lane + 64 * j;
      if (q < C4) {
        const float4 s = S[pr * C4 + q], t = T[pr * C4 + q];
        const float dx = s.x - t.x, dy = s.y - t.y, dz = s.z - t.z, dw = s.w - t.w;
        const float qx = dx * dx, qy = dy * dy, qz = dz * dz, qw = dw * dw;
        qs += (qx + qy) + (qz + qw);
        qp += (qx * cc[j].x + qy * cc[j].y) + (qz * cc[j].z + qw * cc[j].w);
      }
    }
    a_fg = fmaf(qs, wf, a_fg);
    a_bg = fmaf(qs, wb, a_bg);
    a_fp = fmaf(qp, wp, a_fp);
  }
  a_fg = wave_sum(a_fg); a_bg = wave_sum(a_bg); a_fp = wave_sum(a_fp);
  if (lane == 0) { red[0][w] = a_fg; red[1][w] = a_bg; red[2][w] = a_fp; }
  __syncthreads();
  if (threadIdx.x < 3) {
    const int k = threadIdx.x;
    const size_t blk = static_cast<size_t>(blockIdx.y) * gridDim.x + blockIdx.x;
    partial[blk * 3 + k] = (red[k][0] + red[k][1]) + (red[k][2] + red[k][3]);
  }
}

template <int NQ>
__global__ __launch_bounds__(256) void masked_mse_bwd_nhwc(const float4* __restrict__ S, const float4* __restrict__ T,
                                                           const float* __restrict__ Wfg, const float* __restrict__ Wbg,
                                                           const float* __restrict__ Wfp, const float* __restrict__ Cc,
                                                           const float* __restrict__ gsc,
                                                           const float* __restrict__ gpool /* d/d mean_c S, may be null */,
                                                           int C4, int HW, float4* __restrict__ dS) {
  const int b = blockIdx.y, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const float invC = 1.f / static_cast<float>(C4 * 4);
  const int r0 = blockIdx.x * NHWC_ROWS_PER_BLOCK, r1 = min(HW, r0 + NHWC_ROWS_PER_BLOCK);
  const float g0 = gsc[0], g1 = gsc[1], g2 = gsc[2];
  float4 cc[NQ];
#pragma unroll
  for (int j = 0; j < NQ; ++j) {
    const int q = lane + 64 * j;
    cc[j] = (Cc != nullptr && q < C4) ? reinterpret_cast<const float4*>(Cc + static_cast<size_t>(b) * C4 * 4)[q]
                                      : make_float4(1.f, 1.f, 1.f, 1.f);
  }
  for (int r = r0 + w; r < r1; r += 4) {
    const size_t pr = static_cast<size_t>(b) * HW + r;
    const float base = g0 * Wfg[pr] + g1 * Wbg[pr];
    const float kp = Wfp ? g2 * Wfp[pr] : 0.f;
    const float gp = gpool ? gpool[pr] * invC : 0.f;
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
      const int q = lane + 64 * j;
      if (q < C4) {
        const float4 s = S[pr * C4 + q], t = T[pr * C4 + q];
        float4 o;
        o.x = 2.f * (s.x - t.x) * (base + kp * cc[j].x) + gp;
        o.y = 2.f * (s.y - t.y) * (base + kp * cc[j].y) + gp;
        o.z = 2.f * (s.z - t.z) * (base + kp * cc[j].z) + gp;
        o.w = 2.f * (s.w - t.w) * (base + kp * cc[j].w) + gp;
        dS[pr * C4 + q] = o;
      }
    }
  }
}

#define DBEV_NQ_DISPATCH(NQV, KERNEL, ...)                                                     \
  switch (NQV) {                                                                               \
    case 1: hipLaunchKernelGGL((KERNEL<1>), __VA_ARGS__); break;                               \
    case 2: hipLaunchKernelGGL((KERNEL<2>), __VA_ARGS__); break;                               \
    case 3: hipLaunchKernelGGL((KERNEL<3>), __VA_ARGS__); break;                               \
    default: hipLaunchKernelGGL((KERNEL<4>), __VA_ARGS__); break;                              \
  }

bool nhwc_ok(int B, int C, int HW) { return B > 0 && HW > 0 && C > 0 && (C & 3) == 0 && C <= 1024; }

}  // namespace

extern "C" size_t dbev_abs_mean_maps_nhwc_workspace_bytes(int B, int C, int HW) {
  if (!nhwc_ok(B, C, HW)) return 0;
  return sizeof(float) * static_cast<size_t>(B) * dbev_ceil_div(HW, NHWC_ROWS_PER_BLOCK) * C + 256;
}

extern "C" int dbev_abs_mean_maps_nhwc(const float* x_nhwc, int B, int C, int HW, float* pix_mean, float* ch_mean,
                                       float* pix_signed_mean, void* workspace, size_t workspace_bytes,
                                       dbevStream_t stream) {
  if (!nhwc_ok(B, C, HW)) return DBEV_EINVAL;
  if (workspace == nullptr || workspace_bytes < dbev_abs_mean_maps_nhwc_workspace_bytes(B, C, HW)) return DBEV_EINVAL;
  hipStream_t s = dbev_stream(stream);
  const int nbx = dbev_ceil_div(HW, NHWC_ROWS_PER_BLOCK), C4 = C >> 2;
  float* chpart = static_cast<float*>(workspace);
  DBEV_NQ_DISPATCH((C4 + 63) / 64, abs_mean_nhwc, dim3(nbx, B), dim3(256), 0, s,
                   reinterpret_cast<const float4*>(x_nhwc), C4, HW, pix_mean, chpart, pix_signed_mean);
  hipLaunchKernelGGL(abs_mean_ch_final, dim3(dbev_ceil_div(C, 32), B), dim3(256), 0, s, chpart, C, HW, nbx, ch_mean);
  DBEV_LAUNCH_CHECK();
  return 0;
}

extern "C" size_t dbev_fgd_masked_mse_nhwc_workspace_bytes(int B, int C, int HW) {
  if (!nhwc_ok(B, C, HW)) return 0;
  return sizeof(float) * 3 * static_cast<size_t>(B) * dbev_ceil_div(HW, NHWC_ROWS_PER_BLOCK);
}

extern "C" int dbev_fgd_masked_mse_forward_nhwc(const float* S, const float* T, const float* Wfg, const float* Wbg,
                                                const float* Wfp, const float* Cc, int B, int C, int HW, float* out3,
                                                void* workspace, size_t workspace_bytes, dbevStream_t stream) {
  if (!nhwc_ok(B, C, HW)) return DBEV_EINVAL;
  if (workspace == nullptr || workspace_bytes < dbev_fgd_masked_mse_nhwc_workspace_bytes(B, C, HW)) return DBEV_EINVAL;
  hipStream_t s = dbev_stream(stream);
  const int nbx = dbev_ceil_div(HW, NHWC_ROWS_PER_BLOCK), C4 = C >> 2;
  float* partial = static_cast<float*>(workspace);
  DBEV_NQ_DISPATCH((C4 + 63) / 64, masked_mse_fwd_nhwc, dim3(nbx, B), dim3(256), 0, s,
                   reinterpret_cast<const float4*>(S), reinterpret_cast<const float4*>(T), Wfg, Wbg, Wfp, Cc, C4, HW,
                   partial);
  hipLaunchKernelGGL(masked_mse_final, dim3(1), dim3(256), 0, s, partial, nbx * B, out3);
  DBEV_LAUNCH_CHECK();
  return 0;
}

extern "C" int dbev_fgd_masked_mse_backward_nhwc(const float* S, const float* T, const float* Wfg, const float* Wbg,
                                                 const float* Wfp, const float* Cc, const float* grad_scale3,
                                                 const float* grad_pixel_mean, int B, int C, int HW, float* dS,
                                                 dbevStream_t stream) {
  if (!nhwc_ok(B, C, HW)) return DBEV_EINVAL;
  const int nbx = dbev_ceil_div(HW, NHWC_ROWS_PER_BLOCK), C4 = C >> 2;
  DBEV_NQ_DISPATCH((C4 + 63) / 64, masked_mse_bwd_nhwc, dim3(nbx, B), dim3(256), 0, dbev_stream(stream),
                   reinterpret_cast<const float4*>(S), reinterpret_cast<const float4*>(T), Wfg, Wbg, Wfp, Cc,
                   grad_scale3, grad_pixel_mean, C4, HW, reinterpret_cast<float4*>(dS));
  DBEV_LAUNCH_CHECK();
  return 0;
}

namespace {
}  // namespace

extern "C" int dbev_fg_scale_mask(const float* planes, const float* box_scale, const int32_t* box_offsets,
                                  const float* xs, const float* ys, int B, int H, int W, float* fg,
                                  float* fg_scale, float* bg_scale, int32_t* fg_count,
                                  dbevStream_t stream) {
  if (B <= 0 || H <= 0 || W <= 0) return DBEV_EINVAL;
  hipStream_t s = dbev_stream(stream);
  DBEV_HIP_TRY(hipMemsetAsync(fg_count, 0, sizeof(int) * B, s));
  const dim3 grid(dbev_ceil_div(H * W, 256), B);
  hipLaunchKernelGGL(fg_mask_kernel, grid, dim3(256), 0, s, reinterpret_cast<const float4*>(planes),
                     box_scale, box_offsets, xs, ys, H, W, fg, fg_scale, fg_count);
  hipLaunchKernelGGL(bg_scale_kernel, grid, dim3(256), 0, s, fg_count, H * W, bg_scale);
  DBEV_LAUNCH_CHECK();
  return 0;
}

extern "C" size_t dbev_abs_mean_maps_workspace_bytes(int B, int C, int HW) {
  if (B <= 0 || C <= 0 || HW <= 0) return 0;
  const size_t chpart = static_cast<size_t>(B) * dbev_ceil_div(HW, AM_PX) * C;
  const size_t pixpart = static_cast<size_t>(B) * dbev_ceil_div(C, AM_CH) * HW;
  return sizeof(float) * (chpart + pixpart) + 256;
}

extern "C" int dbev_abs_mean_maps(const float* x, int B, int C, int HW, float* pix_mean, float* ch_mean,
                                  void* workspace, size_t workspace_bytes, dbevStream_t stream) {
  if (B <= 0 || C <= 0 || HW <= 0) return DBEV_EINVAL;
  if (workspace == nullptr || workspace_bytes < dbev_abs_mean_maps_workspace_bytes(B, C, HW)) return DBEV_EINVAL;
  hipStream_t s = dbev_stream(stream);
  const int ntile = dbev_ceil_div(HW, AM_PX), nchunk = dbev_ceil_div(C, AM_CH);
  float* chpart = static_cast<float*>(workspace);
  float* pixpart = chpart + ((static_cast<size_t>(B) * ntile * C + 63) & ~static_cast<size_t>(63));
  hipLaunchKernelGGL(abs_mean_kernel, dim3(ntile, nchunk, B), dim3(256), 0, s, x, C, HW, pixpart, chpart);
  hipLaunchKernelGGL(abs_mean_pix_final, dim3(dbev_ceil_div(HW, 256), B), dim3(256), 0, s, pixpart, C, HW, nchunk,
                     pix_mean);
  hipLaunchKernelGGL(abs_mean_ch_final, dim3(dbev_ceil_div(C, 32), B), dim3(256), 0, s, chpart, C, HW, ntile,
                     ch_mean);
  DBEV_LAUNCH_CHECK();
  return 0;
}

static inline void mse_grid(int B, int C, int HW, dim3* g) {
  *g = dim3(dbev_ceil_div(HW / 4, 256), dbev_ceil_div(C, MSE_CCHUNK), B);
}

extern "C" size_t dbev_fgd_masked_mse_workspace_bytes(int B, int C, int HW) {
  if (B <= 0 || C <= 0 || HW <= 0 || (HW & 3)) return 0;
  dim3 g;
  mse_grid(B, C, HW, &g);
  return sizeof(float) * 3 * static_cast<size_t>(g.x) * g.y * g.z;
}

extern "C" int dbev_fgd_masked_mse_forward(const float* S, const float* T, const float* Wfg,
                                           const float* Wbg, const float* Wfp, const float* Cc, int B,
                                           int C, int HW, float* out3, void* workspace,
                                           size_t workspace_bytes, dbevStream_t stream) {
  if (B <= 0 || C <= 0 || HW <= 0 || (HW & 3)) return DBEV_EINVAL;
  const size_t need = dbev_fgd_masked_mse_workspace_bytes(B, C, HW);
  if (workspace == nullptr || workspace_bytes < need) return DBEV_EINVAL;
  hipStream_t s = dbev_stream(stream);
  dim3 g;
  mse_grid(B, C, HW, &g);
  float* partial = static_cast<float*>(workspace);
  hipLaunchKernelGGL(masked_mse_fwd, g, dim3(256), 0, s, reinterpret_cast<const float4*>(S),
                     reinterpret_cast<const float4*>(T), reinterpret_cast<const float4*>(Wfg),
                     reinterpret_cast<const float4*>(Wbg), reinterpret_cast<const float4*>(Wfp), Cc, C, HW / 4,
                     partial);
  hipLaunchKernelGGL(masked_mse_final, dim3(1), dim3(256), 0, s, partial, static_cast<int>(g.x * g.y * g.z), out3);
  DBEV_LAUNCH_CHECK();
  return 0;
}

extern "C" int dbev_fgd_masked_mse_backward(const float* S, const float* T, const float* Wfg,
                                            const float* Wbg, const float* Wfp, const float* Cc,
                                            const float* grad_scale3, int B, int C, int HW, float* dS,
                                            dbevStream_t stream) {
  if (B <= 0 || C <= 0 || HW <= 0 || (HW & 3)) return DBEV_EINVAL;
  dim3 g;
  mse_grid(B, C, HW, &g);
  hipLaunchKernelGGL(masked_mse_bwd, g, dim3(256), 0, dbev_stream(stream), reinterpret_cast<const float4*>(S),
                     reinterpret_cast<const float4*>(T), reinterpret_cast<const float4*>(Wfg),
                     reinterpret_cast<const float4*>(Wbg), reinterpret_cast<const float4*>(Wfp), Cc, grad_scale3,
                     C, HW / 4, reinterpret_cast<float4*>(dS));
  DBEV_LAUNCH_CHECK();
  return 0;
}
