// 3x3 convolution (stride 1, padding 1) with a SKINNY output: Cout <= 3, channels-last fp32.
//
// The final layer of every CenterHead branch (mmdet3d/models/dense_heads/centerpoint_head.py:17-130 SeparateHead:
// reg 2, height 1, dim 3, rot 2, vel 2, heatmap 1-2 channels from a 64-channel hidden map; 36 such convolutions in the
// student head, 36 in the teacher's) has 0.15-0.45 GFLOP and reads a 33.5 MB input: it is HBM-bound streaming work,
// not GEMM-shaped.  MIOpen's implicit-GEMM tiles (64x16 ... with N padded from 1-3 to 16) need 57 us forward, 18 us
// data-gradient, 58 us weight-gradient plus ~6 SetTensor / bias / bias-gradient launches per convolution -- 0.6 TB/s.
// Here: output-stationary forward and data-gradient, input-stationary weight-gradient, one lane group of Cin/4 lanes
// per pixel (float4 = 4 input channels per lane), weights in LDS, fixed-order two-stage reductions (no float atomics).
//
//   x    f32[N, H, W, Cin]     (NHWC),  Cin/4 in {8, 16, 32, 64}
//   wp   f32[Cout, 3, 3, Cin]  = weight.permute(0, 2, 3, 1) of the torch [Cout, Cin, 3, 3] parameter
//   y    f32[N, H, W, Cout]    (NHWC image of [N, Cout, H, W])
#include "common.h"

#include <stdlib.h>

namespace {

constexpr int SK_MAX_CO = 3;

struct SkDims { int N, H, W, C4; int XP4, DP4; };   // XP4 / DP4: pixel pitch of x / dx in float4 (= C4 for a dense map; larger for a
                                                       // channel slice of a wider NHWC tensor: the batched head branches)

__device__ __forceinline__ float dot4f(const float4& a, const float4& b) {
  return fmaf(a.x, b.x, fmaf(a.y, b.y, fmaf(a.z, b.z, a.w * b.w)));
}

constexpr int SK_PERSIST_BLOCKS = 2048;   // forward / data-gradient: lane groups stride over the pixels, weights stay in registers

template <int CO>
__global__ __launch_bounds__(256) void sk_fwd(const float4* __restrict__ x, const float4* __restrict__ wp,
                                              const float* __restrict__ bias, float* __restrict__ y, SkDims d) {
  const int G = d.C4;
  const int q = threadIdx.x & (G - 1);
  float4 wr[CO * 9];                                 // this lane's 4 input channels of every (co, tap)
#pragma unroll
  for (int i = 0; i < CO * 9; ++i) wr[i] = wp[i * G + q];
  float br[CO];
#pragma unroll
  for (int co = 0; co < CO; ++co) br[co] = bias ? bias[co] : 0.f;
  const long long npix = static_cast<long long>(d.N) * d.H * d.W;
  const long long ngrp = static_cast<long long>(gridDim.x) * blockDim.x / G;
  // all lanes of a wave run the same number of iterations (shuffles below): bound by the wave's first group
  const long long g0 = (static_cast<long long>(blockIdx.x) * blockDim.x + (threadIdx.x & ~63)) / G;
  const long long gme = (static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x) / G;
  for (long long pb = g0; pb < npix; pb += ngrp) {
    const long long p = pb + (gme - g0);
    const bool live = p < npix;
    const long long pp = live ? p : 0;
    const unsigned pu = static_cast<unsigned>(pp);            // N*H*W < 2^31 (sk_ok): 32-bit divisions
    const unsigned rowi = pu / static_cast<unsigned>(d.W);
    const int w = static_cast<int>(pu - rowi * d.W);
    const long long n = rowi / static_cast<unsigned>(d.H);
    const int h = static_cast<int>(rowi - static_cast<unsigned>(n) * d.H);
    float4 v[9];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int hh = h + tap / 3 - 1, ww = w + tap % 3 - 1;
      v[tap] = (live && hh >= 0 && hh < d.H && ww >= 0 && ww < d.W) ? x[((n * d.H + hh) * d.W + ww) * d.XP4 + q]
                                                                  : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int co = 0; co < CO; ++co) {
      float s = 0.f;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) s += dot4f(v[tap], wr[co * 9 + tap]);
      for (int o = G >> 1; o > 0; o >>= 1) s += __shfl_xor(s, o);
      if (live && q == 0) y[p * CO + co] = s + br[co];
    }
  }
}

// dx[n,h,w,ci] = sum_{tap,co} dy[n, h-(kh-1), w-(kw-1), co] * W[co,ci,kh,kw]
template <int CO>
__global__ __launch_bounds__(256) void sk_bwd_data(const float* __restrict__ dy, const float4* __restrict__ wp,
                                                   float4* __restrict__ dx, SkDims d) {
  const int G = d.C4;
  const int q = threadIdx.x & (G - 1);
  float4 wr[CO * 9];
#pragma unroll
  for (int i = 0; i < CO * 9; ++i) wr[i] = wp[i * G + q];
  const long long npix = static_cast<long long>(d.N) * d.H * d.W;
  const long long ngrp = static_cast<long long>(gridDim.x) * blockDim.x / G;
  for (long long p = (static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x) / G; p < npix; p += ngrp) {
    const unsigned pu = static_cast<unsigned>(p);
    const unsigned rowi = pu / static_cast<unsigned>(d.W);
    const int w = static_cast<int>(pu - rowi * d.W);
    const long long n = rowi / static_cast<unsigned>(d.H);
    const int h = static_cast<int>(rowi - static_cast<unsigned>(n) * d.H);
    float gv[9][CO];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int hh = h - (tap / 3 - 1), ww = w - (tap % 3 - 1);
      const bool in = hh >= 0 && hh < d.H && ww >= 0 && ww < d.W;
      const float* g = dy + ((n * d.H + hh) * d.W + ww) * CO;
#pragma unroll
      for (int co = 0; co < CO; ++co) gv[tap][co] = in ? g[co] : 0.f;
    }
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int co = 0; co < CO; ++co) {
        const float4 wv = wr[co * 9 + tap];
        acc.x = fmaf(gv[tap][co], wv.x, acc.x); acc.y = fmaf(gv[tap][co], wv.y, acc.y);
        acc.z = fmaf(gv[tap][co], wv.z, acc.z); acc.w = fmaf(gv[tap][co], wv.w, acc.w);
      }
    st_nt(dx + p * d.DP4 + q, acc);
  }
}

// input-stationary weight gradient: every lane keeps dWp[co][tap] for its 4 input channels (CO*9 float4 registers),
// walks a contiguous range of pixels reading x ONCE and the 9*CO neighbouring dy scalars, then the lane groups of the
// workgroup are merged through LDS and written as one partial per workgroup.  dbias in the same pass.
template <int CO>
__global__ __launch_bounds__(256) void sk_bwd_weight(const float4* __restrict__ x, const float* __restrict__ dy,
                                                     float4* __restrict__ part_w /* [blocks][CO*9][C4] */,
                                                     float* __restrict__ part_b /* [blocks][CO] */, SkDims d) {
  __shared__ float4 red[256];
  const int G = d.C4;
  const int grp = threadIdx.x / G, q = threadIdx.x & (G - 1), ngrp = blockDim.x / G;
  const long long npix = static_cast<long long>(d.N) * d.H * d.W;
  const long long per = (npix + gridDim.x - 1) / gridDim.x;
  const long long p0 = static_cast<long long>(blockIdx.x) * per;
  const long long p1 = p0 + per < npix ? p0 + per : npix;
  float4 acc[CO * 9];
#pragma unroll
  for (int i = 0; i < CO * 9; ++i) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  float bsum[CO];
#pragma unroll
  for (int co = 0; co < CO; ++co) bsum[co] = 0.f;
  constexpr int U = CO == 3 ? 1 : 2;      // CO = 3 already holds 27 float4 accumulators per lane
  for (long long pa = p0 + grp; pa < p1; pa += U * ngrp) {
    // U pixels per iteration: every load is issued before the FMAs (the loop is latency-bound otherwise)
    float4 v[U];
    float gv[U][9][CO];
    float gc[U][CO];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long p = pa + u * ngrp;
      const bool live = p < p1;
      const long long pp = live ? p : p0;
      const unsigned pu = static_cast<unsigned>(pp);
      const unsigned rowi = pu / static_cast<unsigned>(d.W);
      const int w = static_cast<int>(pu - rowi * d.W);
      const long long n = rowi / static_cast<unsigned>(d.H);
      const int h = static_cast<int>(rowi - static_cast<unsigned>(n) * d.H);
      v[u] = live ? x[pp * d.XP4 + q] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int co = 0; co < CO; ++co) gc[u][co] = live ? dy[pp * CO + co] : 0.f;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        // x[p] is the (kh, kw) neighbour of the output pixel at (h - (kh-1), w - (kw-1))
        const int hh = h - (tap / 3 - 1), ww = w - (tap % 3 - 1);
        const bool in = live && hh >= 0 && hh < d.H && ww >= 0 && ww < d.W;
        const float* g = dy + ((n * d.H + hh) * d.W + ww) * CO;
#pragma unroll
        for (int co = 0; co < CO; ++co) gv[u][tap][co] = in ? g[co] : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int co = 0; co < CO; ++co) bsum[co] += gc[u][co];             // same value in every lane of the group
#pragma unroll
      for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int co = 0; co < CO; ++co) {
          float4& a = acc[co * 9 + tap];
          const float g1 = gv[u][tap][co];
          a.x = fmaf(g1, v[u].x, a.x); a.y = fmaf(g1, v[u].y, a.y); a.z = fmaf(g1, v[u].z, a.z); a.w = fmaf(g1, v[u].w, a.w);
        }
    }
  }
  // merge the lane groups (fixed order: group 0, 1, 2, ...)
#pragma unroll
  for (int i = 0; i < CO * 9; ++i) {
    __syncthreads();
    red[threadIdx.x] = acc[i];
    __syncthreads();
    if (grp == 0) {
      float4 t = red[q];
      for (int k = 1; k < ngrp; ++k) {
        const float4 u = red[k * G + q];
        t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
      }
      part_w[(static_cast<size_t>(blockIdx.x) * CO * 9 + i) * G + q] = t;
    }
  }
  __syncthreads();
  if (q == 0) {
#pragma unroll
    for (int co = 0; co < CO; ++co) reinterpret_cast<float*>(red)[grp * CO + co] = bsum[co];
  }
  __syncthreads();
  if (threadIdx.x < CO) {
    float t = 0.f;
    for (int k = 0; k < ngrp; ++k) t += reinterpret_cast<float*>(red)[k * CO + threadIdx.x];
    part_b[static_cast<size_t>(blockIdx.x) * CO + threadIdx.x] = t;
  }
}

// fixed-order fp64 merge of the workgroup partials: 16 outputs x 16 partial phases per workgroup
__global__ __launch_bounds__(256) void sk_final(const float* __restrict__ part, int nblk, int n_out, float* __restrict__ out) {
  __shared__ double red[16][16];
  const int ol = threadIdx.x & 15, ph = threadIdx.x >> 4;
  const int i = blockIdx.x * 16 + ol;
  double s = 0.0;
  if (i < n_out)
    for (int b = ph; b < nblk; b += 16) s += static_cast<double>(part[static_cast<size_t>(b) * n_out + i]);
  red[ph][ol] = s;
  __syncthreads();
  if (ph == 0 && i < n_out) {
    for (int p = 1; p < 16; ++p) s += red[p][ol];
    out[i] = static_cast<float>(s);
  }
}

constexpr int SK_WG_BLOCKS = 512;    // swept 256..1024: fewer, longer workgroups amortise the 9*Cout LDS merge rounds

bool sk_ok(int N, int Cin, int H, int W, int Cout, long long x_pitch, long long dx_pitch, SkDims* d) {
  if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || (Cin & 3) || Cout < 1 || Cout > SK_MAX_CO) return false;
  const int C4 = Cin >> 2;
  if (C4 < 8 || C4 > 64 || (C4 & (C4 - 1))) return false;
  if (x_pitch < Cin || dx_pitch < Cin || (x_pitch & 3) || (dx_pitch & 3) || x_pitch > (1 << 20) || dx_pitch > (1 << 20)) return false;
  if (static_cast<long long>(N) * H * W > 0x7fffffffLL) return false;
  *d = SkDims{N, H, W, C4, static_cast<int>(x_pitch >> 2), static_cast<int>(dx_pitch >> 2)};
  return true;
}

// ---- all branches of a group in ONE launch ---------------------------------------------------------------------------------
// The 36 final convolutions of the batched CenterHead read 64-channel slices of the same wide map; launched one by one each is a
// 33 MB pass that lasts 10-25 us (1.3 TB/s: ramp-up and tail, not bandwidth).  grid.y = branch: every branch is padded to 3 output
// channels in a packed weight buffer (zero rows cost nothing on an HBM-bound kernel), only its real channels are stored.
constexpr int SK_MAX_BR = 48;
struct SkMulti {
  float* y[SK_MAX_BR];            // forward: outputs f32[N,H,W,co[b]];  backward: unused
  const float* gy[SK_MAX_BR];     // backward: output gradients f32[N,H,W,co[b]]
  int co[SK_MAX_BR];
};

// The backward kernels below walk STRIPS of SK_SW consecutive pixels of a row with a sliding window of the output gradients (9 new
// scalar loads per pixel instead of 27: 494 -> 343 us weight gradient, 410 -> 257 us data gradient for the student head).  The
// forward was tried the same way (3 new taps per pixel instead of 9, packed FMAs, a reduce-scatter of the strip's 24 sums over the
// 16 lanes instead of 4 shuffles per value): 490 us against 496 -- every input row is still fetched by three row strips through
// L2 (3.6 GB), and the window + 27 weight registers leave two waves per SIMD; it keeps the simple form.
constexpr int SK_SW = 8;

__global__ __launch_bounds__(256) void sk_fwd_multi(const float4* __restrict__ xbase, const float4* __restrict__ wpk,
                                                    const float* __restrict__ bpk, SkMulti m, SkDims d) {
  constexpr int CO = 3;
  const int br = blockIdx.y, con = m.co[br];
  const int G = d.C4;
  const int q = threadIdx.x & (G - 1);
  const float4* __restrict__ x = xbase + static_cast<size_t>(br) * G;
  const float4* __restrict__ wp = wpk + static_cast<size_t>(br) * CO * 9 * G;
  float* __restrict__ y = m.y[br];
  float4 wr[CO * 9];
#pragma unroll
  for (int i = 0; i < CO * 9; ++i) wr[i] = wp[i * G + q];
  float br_[CO];
#pragma unroll
  for (int co = 0; co < CO; ++co) br_[co] = bpk[br * CO + co];
  const long long npix = static_cast<long long>(d.N) * d.H * d.W;
  const long long ngrp = static_cast<long long>(gridDim.x) * blockDim.x / G;
  const long long g0 = (static_cast<long long>(blockIdx.x) * blockDim.x + (threadIdx.x & ~63)) / G;
  const long long gme = (static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x) / G;
  for (long long pb = g0; pb < npix; pb += ngrp) {
    const long long p = pb + (gme - g0);
    const bool live = p < npix;
    const long long pp = live ? p : 0;
    const unsigned pu = static_cast<unsigned>(pp);
    const unsigned rowi = pu / static_cast<unsigned>(d.W);
    const int w = static_cast<int>(pu - rowi * d.W);
    const long long n = rowi / static_cast<unsigned>(d.H);
    const int h = static_cast<int>(rowi - static_cast<unsigned>(n) * d.H);
    float4 v[9];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int hh = h + tap / 3 - 1, ww = w + tap % 3 - 1;
      v[tap] = (live && hh >= 0 && hh < d.H && ww >= 0 && ww < d.W) ? x[((n * d.H + hh) * d.W + ww) * d.XP4 + q]
                                                                  : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int co = 0; co < CO; ++co) {
      float s = 0.f;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) s += dot4f(v[tap], wr[co * 9 + tap]);
      for (int o = G >> 1; o > 0; o >>= 1) s += __shfl_xor(s, o);
      if (live && q == 0 && co < con) y[p * con + co] = s + br_[co];
    }
  }
}

// Forward for 64-channel maps (16 lanes per pixel) through an LDS tile: a workgroup stages the (4 + 2) x (32 + 2) pixel halo tile of
// its branch once (52 KB, whole 256-byte pixel rows), every lane group then walks 8 pixels of one tile row with a sliding window
// of ds_read_b128 taps -- each input element crosses HBM / L2 1.3 times instead of being fetched 9 times through L1 (the kernel above)
// or 3 times through L2 (a register strip walk).  Per-lane partial sums of the 8 pixels x 3 outputs are reduce-scattered over the
// group's 16 lanes (30 shuffles instead of 96) and leave from all lanes.
constexpr int SKT_H = 4, SKT_W = 32, SKT_G = 16;
constexpr int SKT_PIX = (SKT_H + 2) * (SKT_W + 2);

__device__ __forceinline__ void fma4v(float4& a, const float4& v, const float4& w) {
  a.x = fmaf(v.x, w.x, a.x); a.y = fmaf(v.y, w.y, a.y); a.z = fmaf(v.z, w.z, a.z); a.w = fmaf(v.w, w.w, a.w);
}

__global__ __launch_bounds__(256, 2) void sk_fwd_multi_lds(const float4* __restrict__ xbase, const float4* __restrict__ wpk,
                                                           const float* __restrict__ bpk, SkMulti m, SkDims d, int tiles_h, int tiles_w) {
  constexpr int CO = 3, G = SKT_G;
  __shared__ float4 tile[SKT_PIX * G];                                        // [halo pixel][q]
  const int br = blockIdx.y, con = m.co[br];
  const int q = threadIdx.x & (G - 1), grp = threadIdx.x / G;                 // 16 groups: tile row grp / 4, columns 8 (grp % 4) ..
  const float4* __restrict__ x = xbase + static_cast<size_t>(br) * G;
  const float4* __restrict__ wp = wpk + static_cast<size_t>(br) * CO * 9 * G;
  float* __restrict__ y = m.y[br];
  float4 wr[CO * 9];
#pragma unroll
  for (int i = 0; i < CO * 9; ++i) wr[i] = wp[i * G + q];
  const float b0v = bpk[br * CO + 0], b1v = bpk[br * CO + 1], b2v = bpk[br * CO + 2];
  const int ntile = d.N * tiles_h * tiles_w;
  const bool b3 = (q & 8) != 0, b2 = (q & 4) != 0, b1 = (q & 2) != 0, b0 = (q & 1) != 0;
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int t = blockIdx.x; t < ntile; t += gridDim.x) {
    const int tw = t % tiles_w, th = (t / tiles_w) % tiles_h, n = t / (tiles_w * tiles_h);
    const int h0 = th * SKT_H, w0 = tw * SKT_W;
    __syncthreads();                                                          // the previous tile is consumed
    for (int i = threadIdx.x; i < SKT_PIX * G; i += 256) {
      const int pix = i / G, qq = i - pix * G;
      const int hh = h0 - 1 + pix / (SKT_W + 2), ww = w0 - 1 + pix % (SKT_W + 2);
      tile[i] = (hh >= 0 && hh < d.H && ww >= 0 && ww < d.W)
                    ? x[((static_cast<size_t>(n) * d.H + hh) * d.W + ww) * d.XP4 + qq] : z;
    }
    __syncthreads();
    const int tr = grp >> 2, tc = (grp & 3) * 8;                               // this group's output row / first column inside the tile
    const float4* __restrict__ r0 = tile + (static_cast<size_t>(tr) * (SKT_W + 2) + tc) * G + q;   // halo row tr = image row h - 1
    const float4* __restrict__ r1 = r0 + (SKT_W + 2) * G;
    const float4* __restrict__ r2 = r1 + (SKT_W + 2) * G;
    float4 c0[3] = {r0[0], r1[0], r2[0]}, c1[3] = {r0[G], r1[G], r2[G]}, c2[3];
    float ps[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      c2[0] = r0[(i + 2) * G]; c2[1] = r1[(i + 2) * G]; c2[2] = r2[(i + 2) * G];
#pragma unroll
      for (int co = 0; co < CO; ++co) {
        float4 a4 = z;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          fma4v(a4, c0[r], wr[co * 9 + 3 * r + 0]);
          fma4v(a4, c1[r], wr[co * 9 + 3 * r + 1]);
          fma4v(a4, c2[r], wr[co * 9 + 3 * r + 2]);
        }
        ps[i][co] = (a4.x + a4.y) + (a4.z + a4.w);
      }
      ps[i][3] = 0.f;
#pragma unroll
      for (int r = 0; r < 3; ++r) { c0[r] = c1[r]; c1[r] = c2[r]; }
    }
    // reduce-scatter of the 32 slots 4 i + co: lane (b3 b2 b1 b0) ends up with pixel 4 b3 + 2 b2 + b1, outputs 2 b0 + (0, 1)
    float v16[16], v8[8], v4[4], v2[2];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const float lo = ps[k >> 2][k & 3], hi = ps[4 + (k >> 2)][k & 3];
      v16[k] = (b3 ? hi : lo) + __shfl_xor(b3 ? lo : hi, 8);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) v8[k] = (b2 ? v16[8 + k] : v16[k]) + __shfl_xor(b2 ? v16[k] : v16[8 + k], 4);
#pragma unroll
    for (int k = 0; k < 4; ++k) v4[k] = (b1 ? v8[4 + k] : v8[k]) + __shfl_xor(b1 ? v8[k] : v8[4 + k], 2);
#pragma unroll
    for (int k = 0; k < 2; ++k) v2[k] = (b0 ? v4[2 + k] : v4[k]) + __shfl_xor(b0 ? v4[k] : v4[2 + k], 1);
    const int h = h0 + tr, w = w0 + tc + (b3 ? 4 : 0) + (b2 ? 2 : 0) + (b1 ? 1 : 0);
    if (h < d.H && w < d.W) {
      float* __restrict__ yo = y + ((static_cast<size_t>(n) * d.H + h) * d.W + w) * con;
      if (!b0) {
        yo[0] = v2[0] + b0v;
        if (con > 1) yo[1] = v2[1] + b1v;
      } else if (con > 2) {
        yo[2] = v2[0] + b2v;
      }
    }
  }
}

// dy window of a strip walk: the <= 3 output-gradient values of the three rows at one column (zeros outside the map)
template <int CO>
__device__ __forceinline__ void sk_dy_col(const float* __restrict__ dy, size_t rowbase[3], const bool rok[3], int w, int W, int con,
                                          float (&g)[3][CO]) {
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const bool in = rok[r] && w >= 0 && w < W;
    const float* __restrict__ gp = dy + (rowbase[r] + (in ? w : 0)) * con;
#pragma unroll
    for (int co = 0; co < CO; ++co) g[r][co] = (in && co < con) ? gp[co] : 0.f;
  }
}

__global__ __launch_bounds__(256) void sk_bwd_data_multi(const float4* __restrict__ wpk, float4* __restrict__ dxbase, SkMulti m, SkDims d) {
  constexpr int CO = 3;
  const int br = blockIdx.y, con = m.co[br];
  const int G = d.C4;
  const int q = threadIdx.x & (G - 1);
  const float4* __restrict__ wp = wpk + static_cast<size_t>(br) * CO * 9 * G;
  const float* __restrict__ dy = m.gy[br];
  float4* __restrict__ dx = dxbase + static_cast<size_t>(br) * G;
  float4 wr[CO * 9];
#pragma unroll
  for (int i = 0; i < CO * 9; ++i) wr[i] = wp[i * G + q];
  const int spr = (d.W + SK_SW - 1) / SK_SW;
  const long long nstrip = static_cast<long long>(d.N) * d.H * spr;
  const long long ngrp = static_cast<long long>(gridDim.x) * blockDim.x / G;
  for (long long si = (static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x) / G; si < nstrip; si += ngrp) {
    const unsigned su = static_cast<unsigned>(si);
    const unsigned rowi = su / static_cast<unsigned>(spr);
    const int w0 = static_cast<int>(su - rowi * spr) * SK_SW;
    const unsigned n = rowi / static_cast<unsigned>(d.H);
    const int h = static_cast<int>(rowi - n * d.H);
    // dx[p] = sum_tap sum_co dy[p - (tap offset)][co] * w[co][tap]: tap (kr, kc) reads dy at (h - kr + 1, w - kc + 1)
    size_t rowbase[3];                                                        // slot r = output row h + 1 - r  (tap row kr = r)
    bool rok[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int hh = h + 1 - r;
      rok[r] = hh >= 0 && hh < d.H;
      rowbase[r] = (static_cast<size_t>(n) * d.H + (rok[r] ? hh : h)) * d.W;
    }
    float ga[3][CO], gb[3][CO], gc[3][CO];                                    // output columns w + 1, w, w - 1  (tap columns 0, 1, 2)
    sk_dy_col<CO>(dy, rowbase, rok, w0 - 1, d.W, con, gc);
    sk_dy_col<CO>(dy, rowbase, rok, w0, d.W, con, gb);
#pragma unroll
    for (int i = 0; i < SK_SW; ++i) {
      const int w = w0 + i;
      sk_dy_col<CO>(dy, rowbase, rok, w + 1, d.W, con, ga);
      if (w < d.W) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
          for (int kc = 0; kc < 3; ++kc)
#pragma unroll
            for (int co = 0; co < CO; ++co) {                                  // tap = 3 r + kc, then co: the old order
              const float g1 = kc == 0 ? ga[r][co] : (kc == 1 ? gb[r][co] : gc[r][co]);
              const float4 wv = wr[co * 9 + 3 * r + kc];
              acc.x = fmaf(g1, wv.x, acc.x); acc.y = fmaf(g1, wv.y, acc.y);
              acc.z = fmaf(g1, wv.z, acc.z); acc.w = fmaf(g1, wv.w, acc.w);
            }
        st_nt(dx + ((static_cast<size_t>(n) * d.H + h) * d.W + w) * d.DP4 + q, acc);
      }
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int co = 0; co < CO; ++co) { gc[r][co] = gb[r][co]; gb[r][co] = ga[r][co]; }
    }
  }
}

// weight / bias gradient partials of every branch: part_w [branch][gridDim.x][27][C4], part_b [branch][gridDim.x][3]
__global__ __launch_bounds__(256) void sk_bwd_weight_multi(const float4* __restrict__ xbase, float4* __restrict__ part_w,
                                                           float* __restrict__ part_b, SkMulti m, SkDims d) {
  constexpr int CO = 3;
  __shared__ float4 red[256];
  const int br = blockIdx.y, con = m.co[br];
  const int G = d.C4;
  const float4* __restrict__ x = xbase + static_cast<size_t>(br) * G;
  const float* __restrict__ dy = m.gy[br];
  const int grp = threadIdx.x / G, q = threadIdx.x & (G - 1), ngrp = blockDim.x / G;
  // a workgroup owns a contiguous range of strips (SK_SW pixels of a row), its lane groups walk them with a sliding window of the
  // output gradients: 9 scalar loads per pixel instead of 27
  const int spr = (d.W + SK_SW - 1) / SK_SW;
  const long long nstrip = static_cast<long long>(d.N) * d.H * spr;
  const long long per = (nstrip + gridDim.x - 1) / gridDim.x;
  const long long s0 = static_cast<long long>(blockIdx.x) * per;
  const long long s1 = s0 + per < nstrip ? s0 + per : nstrip;
  float4 acc[CO * 9];
#pragma unroll
  for (int i = 0; i < CO * 9; ++i) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  float bsum[CO];
#pragma unroll
  for (int co = 0; co < CO; ++co) bsum[co] = 0.f;
  for (long long si = s0 + grp; si < s1; si += ngrp) {
    const unsigned su = static_cast<unsigned>(si);
    const unsigned rowi = su / static_cast<unsigned>(spr);
    const int w0 = static_cast<int>(su - rowi * spr) * SK_SW;
    const unsigned n = rowi / static_cast<unsigned>(d.H);
    const int h = static_cast<int>(rowi - n * d.H);
    size_t rowbase[3];                                                        // slot r = output row h + 1 - r  (tap row kr = r)
    bool rok[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int hh = h + 1 - r;
      rok[r] = hh >= 0 && hh < d.H;
      rowbase[r] = (static_cast<size_t>(n) * d.H + (rok[r] ? hh : h)) * d.W;
    }
    const float4* __restrict__ xr = x + (static_cast<size_t>(n) * d.H + h) * d.W * d.XP4 + q;
    float ga[3][CO], gb[3][CO], gc[3][CO];                                    // output columns w + 1, w, w - 1  (tap columns 0, 1, 2)
    sk_dy_col<CO>(dy, rowbase, rok, w0 - 1, d.W, con, gc);
    sk_dy_col<CO>(dy, rowbase, rok, w0, d.W, con, gb);
#pragma unroll
    for (int i = 0; i < SK_SW; ++i) {
      const int w = w0 + i;
      sk_dy_col<CO>(dy, rowbase, rok, w + 1, d.W, con, ga);
      if (w < d.W) {
        const float4 v = xr[static_cast<size_t>(w) * d.XP4];
#pragma unroll
        for (int co = 0; co < CO; ++co) bsum[co] += gb[1][co];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
          for (int kc = 0; kc < 3; ++kc)
#pragma unroll
            for (int co = 0; co < CO; ++co) {
              float4& a = acc[co * 9 + 3 * r + kc];
              const float g1 = kc == 0 ? ga[r][co] : (kc == 1 ? gb[r][co] : gc[r][co]);
              a.x = fmaf(g1, v.x, a.x); a.y = fmaf(g1, v.y, a.y); a.z = fmaf(g1, v.z, a.z); a.w = fmaf(g1, v.w, a.w);
            }
      }
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int co = 0; co < CO; ++co) { gc[r][co] = gb[r][co]; gb[r][co] = ga[r][co]; }
    }
  }
  const size_t blk = static_cast<size_t>(br) * gridDim.x + blockIdx.x;
#pragma unroll
  for (int i = 0; i < CO * 9; ++i) {
    __syncthreads();
    red[threadIdx.x] = acc[i];
    __syncthreads();
    if (grp == 0) {
      float4 t = red[q];
      for (int k = 1; k < ngrp; ++k) {
        const float4 u = red[k * G + q];
        t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
      }
      part_w[(blk * CO * 9 + i) * G + q] = t;
    }
  }
  __syncthreads();
  if (q == 0) {
#pragma unroll
    for (int co = 0; co < CO; ++co) reinterpret_cast<float*>(red)[grp * CO + co] = bsum[co];
  }
  __syncthreads();
  if (threadIdx.x < CO) {
    float t = 0.f;
    for (int k = 0; k < ngrp; ++k) t += reinterpret_cast<float*>(red)[k * CO + threadIdx.x];
    part_b[blk * CO + threadIdx.x] = t;
  }
}

// out[b][i] = fp64 sum over the workgroup partials part[b][blk][i]   (grid (ceil(n_out / 16), branches))
__global__ __launch_bounds__(256) void sk_final_multi(const float* __restrict__ part, int nblk, int n_out, float* __restrict__ out) {
  __shared__ double red[16][16];
  const int ol = threadIdx.x & 15, ph = threadIdx.x >> 4;
  const int i = blockIdx.x * 16 + ol;
  const float* __restrict__ pb = part + static_cast<size_t>(blockIdx.y) * nblk * n_out;
  double s = 0.0;
  if (i < n_out)
    for (int b = ph; b < nblk; b += 16) s += static_cast<double>(pb[static_cast<size_t>(b) * n_out + i]);
  red[ph][ol] = s;
  __syncthreads();
  if (ph == 0 && i < n_out) {
    for (int p = 1; p < 16; ++p) s += red[p][ol];
    out[static_cast<size_t>(blockIdx.y) * n_out + i] = static_cast<float>(s);
  }
}

constexpr int SK_MULTI_WG_BLOCKS = 128;   // weight-gradient workgroups per branch (x up to 48 branches)

#define SK_DISPATCH(CO, KERNEL, ...)                            \
  switch (CO) {                                                 \
    case 1: hipLaunchKernelGGL((KERNEL<1>), __VA_ARGS__); break; \
    case 2: hipLaunchKernelGGL((KERNEL<2>), __VA_ARGS__); break; \
    default: hipLaunchKernelGGL((KERNEL<3>), __VA_ARGS__); break; \
  }

}  // namespace

extern "C" size_t dbev_skinny_conv3x3_workspace_bytes(int Cin, int Cout) {
  if (Cin <= 0 || (Cin & 3) || Cout < 1 || Cout > SK_MAX_CO) return 0;
  return sizeof(float) * static_cast<size_t>(SK_WG_BLOCKS) * (static_cast<size_t>(Cout) * 9 * Cin + Cout) + 256;
}

static int sk_forward_pitched(const float* x_nhwc, long long x_pitch, const float* weight_ohwi,
                                                   const float* bias, float* y_nhwc, int N, int Cin, int H, int W, int Cout,
                                                   dbevStream_t stream) {
  SkDims d;
  if (!sk_ok(N, Cin, H, W, Cout, x_pitch, Cin, &d) || x_nhwc == nullptr || weight_ohwi == nullptr || y_nhwc == nullptr ||
      (reinterpret_cast<uintptr_t>(x_nhwc) & 15))
    return DBEV_EINVAL;
  const long long threads = static_cast<long long>(N) * H * W * d.C4;
  long long blocks = (threads + 255) / 256;
  if (blocks > SK_PERSIST_BLOCKS) blocks = SK_PERSIST_BLOCKS;
  SK_DISPATCH(Cout, sk_fwd, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, dbev_stream(stream),
              reinterpret_cast<const float4*>(x_nhwc), reinterpret_cast<const float4*>(weight_ohwi), bias, y_nhwc, d);
  DBEV_LAUNCH_CHECK();
  return 0;
}

extern "C" int dbev_skinny_conv3x3_forward(const float* x_nhwc, const float* weight_ohwi, const float* bias, float* y_nhwc,
                                           int N, int Cin, int H, int W, int Cout, dbevStream_t stream) {
  return sk_forward_pitched(x_nhwc, Cin, weight_ohwi, bias, y_nhwc, N, Cin, H, W, Cout, stream);
}

static int sk_backward_pitched(const float* grad_y_nhwc, const float* x_nhwc, long long x_pitch,
                                                    const float* weight_ohwi, float* grad_x_nhwc, long long grad_x_pitch,
                                                    float* grad_weight_ohwi, float* grad_bias, int N, int Cin, int H, int W,
                                                    int Cout, void* workspace, size_t workspace_bytes, dbevStream_t stream) {
  SkDims d;
  if (!sk_ok(N, Cin, H, W, Cout, x_pitch, grad_x_nhwc != nullptr ? grad_x_pitch : Cin, &d) || grad_y_nhwc == nullptr ||
      x_nhwc == nullptr || weight_ohwi == nullptr || (reinterpret_cast<uintptr_t>(x_nhwc) & 15) ||
      (reinterpret_cast<uintptr_t>(grad_x_nhwc) & 15))
    return DBEV_EINVAL;
  hipStream_t s = dbev_stream(stream);
  const long long threads = static_cast<long long>(N) * H * W * d.C4;
  long long pblocks = (threads + 255) / 256;
  if (pblocks > SK_PERSIST_BLOCKS) pblocks = SK_PERSIST_BLOCKS;
  if (grad_x_nhwc != nullptr)
    SK_DISPATCH(Cout, sk_bwd_data, dim3(static_cast<unsigned>(pblocks)), dim3(256), 0, s, grad_y_nhwc,
                reinterpret_cast<const float4*>(weight_ohwi), reinterpret_cast<float4*>(grad_x_nhwc), d);
  if (grad_weight_ohwi != nullptr) {
    if (grad_bias == nullptr || workspace == nullptr || workspace_bytes < dbev_skinny_conv3x3_workspace_bytes(Cin, Cout))
      return DBEV_EINVAL;
    const long long npix = static_cast<long long>(N) * H * W;
    const int rows_per_iter = 256 / d.C4;
    long long blocks = (npix + rows_per_iter * 8 - 1) / (rows_per_iter * 8);
    if (blocks > SK_WG_BLOCKS) blocks = SK_WG_BLOCKS;
    if (blocks < 1) blocks = 1;
    float* part_w = static_cast<float*>(workspace);
    float* part_b = part_w + static_cast<size_t>(SK_WG_BLOCKS) * Cout * 9 * Cin;
    SK_DISPATCH(Cout, sk_bwd_weight, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, s,
                reinterpret_cast<const float4*>(x_nhwc), grad_y_nhwc, reinterpret_cast<float4*>(part_w), part_b, d);
    const int nw = Cout * 9 * Cin;
    hipLaunchKernelGGL(sk_final, dim3(dbev_ceil_div(nw, 16)), dim3(256), 0, s, part_w, static_cast<int>(blocks), nw,
                       grad_weight_ohwi);
    hipLaunchKernelGGL(sk_final, dim3(1), dim3(256), 0, s, part_b, static_cast<int>(blocks), Cout, grad_bias);
  }
  DBEV_LAUNCH_CHECK();
  return 0;
}

extern "C" int dbev_skinny_conv3x3_backward(const float* grad_y_nhwc, const float* x_nhwc, const float* weight_ohwi,
                                            float* grad_x_nhwc, float* grad_weight_ohwi, float* grad_bias, int N, int Cin,
                                            int H, int W, int Cout, void* workspace, size_t workspace_bytes,
                                            dbevStream_t stream) {
  return sk_backward_pitched(grad_y_nhwc, x_nhwc, Cin, weight_ohwi, grad_x_nhwc, Cin, grad_weight_ohwi, grad_bias,
                                              N, Cin, H, W, Cout, workspace, workspace_bytes, stream);
}

extern "C" size_t dbev_skinny_conv3x3_multi_workspace_bytes(int Cin, int n_branch) {
  if (Cin <= 0 || (Cin & 3) || n_branch < 1 || n_branch > SK_MAX_BR) return 0;
  return sizeof(float) * static_cast<size_t>(n_branch) * SK_MULTI_WG_BLOCKS * (27 * static_cast<size_t>(Cin) + 3) + 256;
}

static bool sk_multi_args(SkMulti* m, float* const* y, const float* const* gy, const int32_t* cout, int n_branch) {
  if (n_branch < 1 || n_branch > SK_MAX_BR || cout == nullptr) return false;
  for (int b = 0; b < n_branch; ++b) {
    if (cout[b] < 1 || cout[b] > SK_MAX_CO) return false;
    m->co[b] = cout[b];
    m->y[b] = y ? y[b] : nullptr;
    m->gy[b] = gy ? gy[b] : nullptr;
    if ((y && !y[b]) || (gy && !gy[b])) return false;
  }
  return true;
}

extern "C" int dbev_skinny_conv3x3_multi_forward(const float* x_nhwc, long long x_pitch, const float* weights_packed,
                                                 const float* bias_packed, float* const* y_nhwc, const int32_t* cout,
                                                 int n_branch, int N, int Cin, int H, int W, dbevStream_t stream) {
  SkDims d;
  SkMulti m;
  if (!sk_ok(N, Cin, H, W, 3, x_pitch, Cin, &d) || x_nhwc == nullptr || weights_packed == nullptr || bias_packed == nullptr ||
      (reinterpret_cast<uintptr_t>(x_nhwc) & 15) || !sk_multi_args(&m, y_nhwc, nullptr, cout, n_branch) ||
      x_pitch < static_cast<long long>(n_branch) * Cin)
    return DBEV_EINVAL;
  static const bool lds_ok = !(getenv("DBEV_SKINNY_LDS") && atoi(getenv("DBEV_SKINNY_LDS")) == 0);
  if (d.C4 == SKT_G && lds_ok) {                   // 64-channel maps (the CenterHead branches): LDS-tiled kernel
    const int th = (H + SKT_H - 1) / SKT_H, tw = (W + SKT_W - 1) / SKT_W;
    const long long ntile = static_cast<long long>(N) * th * tw;
    if (ntile > 0x7fffffffLL) return DBEV_EINVAL;
    long long blocks = ntile;
    const long long cap = DBEV_NUM_CU * 12 / n_branch > 32 ? DBEV_NUM_CU * 12 / n_branch : 32;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(sk_fwd_multi_lds, dim3(static_cast<unsigned>(blocks), n_branch), dim3(256), 0, dbev_stream(stream),
                       reinterpret_cast<const float4*>(x_nhwc), reinterpret_cast<const float4*>(weights_packed), bias_packed, m, d, th, tw);
    DBEV_LAUNCH_CHECK();
    return 0;
  }
  const long long threads = static_cast<long long>(N) * H * W * d.C4;
  long long blocks = (threads + 255) / 256;
  const long long cap = SK_PERSIST_BLOCKS * 2 / n_branch > 64 ? SK_PERSIST_BLOCKS * 2 / n_branch : 64;
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(sk_fwd_multi, dim3(static_cast<unsigned>(blocks), n_branch), dim3(256), 0, dbev_stream(stream),
                     reinterpret_cast<const float4*>(x_nhwc), reinterpret_cast<const float4*>(weights_packed), bias_packed, m, d);
  DBEV_LAUNCH_CHECK();
  return 0;
}

extern "C" int dbev_skinny_conv3x3_multi_backward(const float* const* grad_y_nhwc, const float* x_nhwc, long long x_pitch,
                                                  const float* weights_packed, float* grad_x_nhwc, long long grad_x_pitch,
                                                  float* grad_weights_packed, float* grad_bias_packed, const int32_t* cout,
                                                  int n_branch, int N, int Cin, int H, int W, void* workspace,
                                                  size_t workspace_bytes, dbevStream_t stream) {
  SkDims d;
  SkMulti m;
  if (!sk_ok(N, Cin, H, W, 3, x_pitch, grad_x_nhwc != nullptr ? grad_x_pitch : Cin, &d) || x_nhwc == nullptr ||
      weights_packed == nullptr || (reinterpret_cast<uintptr_t>(x_nhwc) & 15) || (reinterpret_cast<uintptr_t>(grad_x_nhwc) & 15) ||
      !sk_multi_args(&m, nullptr, grad_y_nhwc, cout, n_branch) || x_pitch < static_cast<long long>(n_branch) * Cin ||
      (grad_x_nhwc != nullptr && grad_x_pitch < static_cast<long long>(n_branch) * Cin))
    return DBEV_EINVAL;
  hipStream_t s = dbev_stream(stream);
  const long long threads = static_cast<long long>(N) * H * W * d.C4;
  long long pblocks = (threads + 255) / 256;
  const long long cap = SK_PERSIST_BLOCKS * 2 / n_branch > 64 ? SK_PERSIST_BLOCKS * 2 / n_branch : 64;
  if (pblocks > cap) pblocks = cap;
  if (grad_x_nhwc != nullptr)
    hipLaunchKernelGGL(sk_bwd_data_multi, dim3(static_cast<unsigned>(pblocks), n_branch), dim3(256), 0, s,
                       reinterpret_cast<const float4*>(weights_packed), reinterpret_cast<float4*>(grad_x_nhwc), m, d);
  if (grad_weights_packed != nullptr) {
    if (grad_bias_packed == nullptr || workspace == nullptr ||
        workspace_bytes < dbev_skinny_conv3x3_multi_workspace_bytes(Cin, n_branch))
      return DBEV_EINVAL;
    const long long npix = static_cast<long long>(N) * H * W;
    const int rows_per_iter = 256 / d.C4;
    long long blocks = (npix + rows_per_iter * 8 - 1) / (rows_per_iter * 8);
    if (blocks > SK_MULTI_WG_BLOCKS) blocks = SK_MULTI_WG_BLOCKS;
    if (blocks < 1) blocks = 1;
    float* part_w = static_cast<float*>(workspace);
    float* part_b = part_w + static_cast<size_t>(n_branch) * SK_MULTI_WG_BLOCKS * 27 * Cin;
    hipLaunchKernelGGL(sk_bwd_weight_multi, dim3(static_cast<unsigned>(blocks), n_branch), dim3(256), 0, s,
                       reinterpret_cast<const float4*>(x_nhwc), reinterpret_cast<float4*>(part_w), part_b, m, d);
    const int nw = 27 * Cin;
    hipLaunchKernelGGL(sk_final_multi, dim3(dbev_ceil_div(nw, 16), n_branch), dim3(256), 0, s, part_w, static_cast<int>(blocks), nw,
                       grad_weights_packed);
    hipLaunchKernelGGL(sk_final_multi, dim3(1, n_branch), dim3(256), 0, s, part_b, static_cast<int>(blocks), 3, grad_bias_packed);
  }
  DBEV_LAUNCH_CHECK();
  return 0;
}
