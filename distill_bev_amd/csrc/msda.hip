// Multi-scale deformable attention, forward + backward.
//
// Replaces mmcv-full 1.6.0 `_ext.ms_deform_attn_forward / ms_deform_attn_backward` (un-vendored third party; call site
// mmdet3d/models/transformer_modules/multi_scale_deformable_attn_function.py:10-12,42-49,70-82 and the BEVFormer
// attention modules spatial_cross_attention.py / temporal_self_attention.py):
//   out[b, q, h, :] = sum_{l, p} attn[b, q, h, l, p] * bilinear(value_l[b, :, h, :], loc[b, q, h, l, p] * (W_l, H_l) - 0.5)
// with zero padding (every out-of-map corner contributes 0), value [B, S, NH, D] (S = sum_l H_l W_l), loc in [0, 1]
// (x, y), i.e. F.grid_sample(align_corners=False, padding_mode='zeros') per level -- the published definition of
// Deformable-DETR's op that mmcv implements.
//
// Mapping (wave64): one (b, q, h) row of D channels is D/4 lanes of float4; a lane group walks the L*P samples of its
// row, the 4 corner rows of a sample are contiguous D*4-byte gathers (value is head-major per key), a batch of P samples
// (4P gathers) is in flight per lane.  With NH*D/4 = 64 (8 heads x 32 channels, BEVFormer) a wave is exactly one query
// and writes 1 KB of contiguous output.
// Backward: d attn and d loc are per-sample reductions over the group's lanes (shuffle tree, no atomics);
// d value is a DETERMINISTIC gather instead of mmcv's float atomicAdd scatter: the samples are grouped by their anchor
// cell (top-left corner, one integer atomic per sample) with the CSR primitive (integer histogram -> scan -> fill ->
// per-bin sort by sample id); one lane group per value row walks the four anchor bins it is a corner of and adds
// coef * grad_out rows in (corner, ascending sample) order -- bit-reproducible.
#include "common.h"
#include "prims.h"

namespace {

constexpr int MSDA_MAX_LEVELS = 8;

struct MsdaDims {
  int B, S, NH, D4, Q, L, P;
  int h[MSDA_MAX_LEVELS], w[MSDA_MAX_LEVELS], start[MSDA_MAX_LEVELS];
  int astart[MSDA_MAX_LEVELS], A;     // anchor cells (h_low + 1, w_low + 1) of a level: (H + 1) x (W + 1), A = their total
};

struct Bil {
  int o1, o2, o3, o4;        // key offsets inside the level (y * W + x) or -1
  float w1, w2, w3, w4;      // hh*hw, hh*lw, lh*hw, lh*lw
  float hh, hw, lh, lw;
  int ay, ax;                // anchor cell (h_low + 1, w_low + 1) in the (H + 1) x (W + 1) anchor grid
  bool any;
};

__device__ __forceinline__ Bil bil_of(float lx, float ly, int H, int W) {
  Bil t;
  const float h_im = ly * static_cast<float>(H) - 0.5f, w_im = lx * static_cast<float>(W) - 0.5f;
  t.any = h_im > -1.f && w_im > -1.f && h_im < static_cast<float>(H) && w_im < static_cast<float>(W);
  const int h_low = static_cast<int>(floorf(h_im)), w_low = static_cast<int>(floorf(w_im));
  const int h_high = h_low + 1, w_high = w_low + 1;
  t.ay = h_high; t.ax = w_high;
  t.lh = h_im - static_cast<float>(h_low);
  t.lw = w_im - static_cast<float>(w_low);
  t.hh = 1.f - t.lh;
  t.hw = 1.f - t.lw;
  t.w1 = t.hh * t.hw; t.w2 = t.hh * t.lw; t.w3 = t.lh * t.hw; t.w4 = t.lh * t.lw;
  t.o1 = (t.any && h_low >= 0 && w_low >= 0) ? h_low * W + w_low : -1;
  t.o2 = (t.any && h_low >= 0 && w_high <= W - 1) ? h_low * W + w_high : -1;
  t.o3 = (t.any && h_high <= H - 1 && w_low >= 0) ? h_high * W + w_low : -1;
  t.o4 = (t.any && h_high <= H - 1 && w_high <= W - 1) ? h_high * W + w_high : -1;
  return t;
}

__device__ __forceinline__ float4 ld_row(const float4* __restrict__ value, const MsdaDims& d, int b, int key, int head,
                                         int q4) {
  return value[((static_cast<size_t>(b) * d.S + key) * d.NH + head) * d.D4 + q4];
}

__device__ __forceinline__ void fma4s(float4& a, float c, const float4& v) {
  a.x = fmaf(c, v.x, a.x); a.y = fmaf(c, v.y, a.y); a.z = fmaf(c, v.z, a.z); a.w = fmaf(c, v.w, a.w);
}
__device__ __forceinline__ float dot4(const float4& a, const float4& b) {
  return fmaf(a.x, b.x, fmaf(a.y, b.y, fmaf(a.z, b.z, a.w * b.w)));
}

// rows = B * Q * NH groups of G = D4 lanes
template <int PB>
__global__ __launch_bounds__(256) void msda_fwd(const float4* __restrict__ value, const float* __restrict__ loc,
                                                const float* __restrict__ attn, float4* __restrict__ out, MsdaDims d,
                                                long long rows) {
  const int G = d.D4;
  const long long gid = (static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x) / G;
  if (gid >= rows) return;                                  // whole groups leave together
  const int q4 = threadIdx.x & (G - 1);
  const int head = static_cast<int>(gid % d.NH);
  const int b = static_cast<int>(gid / (static_cast<long long>(d.NH) * d.Q));
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  const size_t s0 = static_cast<size_t>(gid) * d.L * d.P;
  for (int l = 0; l < d.L; ++l) {
    const int H = d.h[l], W = d.w[l], st = d.start[l];
    for (int p0 = 0; p0 < d.P; p0 += PB) {
      float4 v[PB][4];
      float c[PB][4];
#pragma unroll
      for (int u = 0; u < PB; ++u) {
        const int p = p0 + u;
        const bool on = p < d.P;
        const size_t si = s0 + static_cast<size_t>(l) * d.P + (on ? p : 0);
        const float a = on ? attn[si] : 0.f;
        const Bil t = bil_of(loc[2 * si], loc[2 * si + 1], H, W);
        const int o[4] = {t.o1, t.o2, t.o3, t.o4};
        const float w[4] = {t.w1, t.w2, t.w3, t.w4};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const bool ok = on && o[k] >= 0;
          c[u][k] = ok ? a * w[k] : 0.f;
          v[u][k] = ok ? ld_row(value, d, b, st + o[k], head, q4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
#pragma unroll
      for (int u = 0; u < PB; ++u) {
        // mmcv order: val = w1 v1 + w2 v2 + w3 v3 + w4 v4, then col += attn * val; here attn is folded into the
        // corner coefficients (one rounding more per corner, inside the 1e-5 tolerance of the op's tests)
#pragma unroll
        for (int k = 0; k < 4; ++k) fma4s(acc, c[u][k], v[u][k]);
      }
    }
  }
  out[static_cast<size_t>(gid) * G + q4] = acc;
}

// sum over the G lanes of a group (every lane gets the total), in a fixed order.  G = 8 (D = 32, BEVFormer): three DPP
// row operations at VALU rate -- quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror -- instead of LDS-path permutes.
template <int GT>
__device__ __forceinline__ float group_sum(float v, int G) {
  if (GT == 8) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, false));
    return v;
  }
  for (int m = 1; m < G; m <<= 1) v += __shfl_xor(v, m);
  return v;
}

// per-sample gradients wrt attention weight and sampling location; PB samples (4 PB corner gathers) in flight per lane.
// BIN: the lane that stores a sample's gradients also counts the sample into its anchor bin (first half of the CSR build of the
// d value pass, see msda_anchor_bin: the integer atomics ride under this kernel's gathers instead of a pass of their own).
template <int GT, int PB, bool BIN>
__global__ __launch_bounds__(256) void msda_bwd_sample(const float4* __restrict__ value, const float* __restrict__ loc,
                                                       const float* __restrict__ attn, const float4* __restrict__ gout,
                                                       float* __restrict__ gloc, float* __restrict__ gattn, MsdaDims d,
                                                       long long rows, int* __restrict__ count, int* __restrict__ rank) {
  const int G = GT ? GT : d.D4;
  const long long gid = (static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x) / G;
  if (gid >= rows) return;
  const int q4 = threadIdx.x & (G - 1);
  const int head = static_cast<int>(gid % d.NH);
  const int b = static_cast<int>(gid / (static_cast<long long>(d.NH) * d.Q));
  const float4 go = gout[static_cast<size_t>(gid) * G + q4];
  const size_t s0 = static_cast<size_t>(gid) * d.L * d.P;
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  int prank[PB];                                               // BIN: the previous batch's arrival ranks, stored one batch late so
  long long psi[PB];                                           // that the atomics' round trip hides under this batch's gathers
#pragma unroll
  for (int u = 0; u < PB; ++u) { prank[u] = 0; psi[u] = -1; }
  for (int l = 0; l < d.L; ++l) {
    const int H = d.h[l], W = d.w[l], st = d.start[l];
    for (int p0 = 0; p0 < d.P; p0 += PB) {
      Bil t[PB];
      float4 v[PB][4];
#pragma unroll
      for (int u = 0; u < PB; ++u) {
        const size_t si = s0 + static_cast<size_t>(l) * d.P + min(p0 + u, d.P - 1);
        t[u] = bil_of(loc[2 * si], loc[2 * si + 1], H, W);
        v[u][0] = t[u].o1 >= 0 ? ld_row(value, d, b, st + t[u].o1, head, q4) : z;
        v[u][1] = t[u].o2 >= 0 ? ld_row(value, d, b, st + t[u].o2, head, q4) : z;
        v[u][2] = t[u].o3 >= 0 ? ld_row(value, d, b, st + t[u].o3, head, q4) : z;
        v[u][3] = t[u].o4 >= 0 ? ld_row(value, d, b, st + t[u].o4, head, q4) : z;
      }
      if (BIN && q4 == 0) {
#pragma unroll
        for (int u = 0; u < PB; ++u)
          if (psi[u] >= 0) rank[psi[u]] = prank[u];
      }
#pragma unroll
      for (int u = 0; u < PB; ++u) {
        if (p0 + u >= d.P) break;                              // uniform
        const size_t si = s0 + static_cast<size_t>(l) * d.P + p0 + u;
        // per-lane partial dot products with grad_out over this lane's 4 channels
        const float d1 = dot4(go, v[u][0]), d2 = dot4(go, v[u][1]), d3 = dot4(go, v[u][2]), d4 = dot4(go, v[u][3]);
        float ga = t[u].w1 * d1 + t[u].w2 * d2 + t[u].w3 * d3 + t[u].w4 * d4;
        float gw = t[u].hh * (d2 - d1) + t[u].lh * (d4 - d3);   // d / d w_im
        float gh = t[u].hw * (d3 - d1) + t[u].lw * (d4 - d2);   // d / d h_im
        ga = group_sum<GT>(ga, G);
        gw = group_sum<GT>(gw, G);
        gh = group_sum<GT>(gh, G);
        if (q4 == 0) {
          const float a = attn[si];
          gattn[si] = t[u].any ? ga : 0.f;
          gloc[2 * si] = t[u].any ? static_cast<float>(W) * a * gw : 0.f;
          gloc[2 * si + 1] = t[u].any ? static_cast<float>(H) * a * gh : 0.f;
          if (BIN) {
            psi[u] = t[u].any ? static_cast<long long>(si) : -1;
            if (t[u].any) prank[u] = atomicAdd(&count[(b * d.A + d.astart[l] + t[u].ay * (W + 1) + t[u].ax) * d.NH + head], 1);
          }
        }
      }
    }
  }
  if (BIN && q4 == 0) {
#pragma unroll
    for (int u = 0; u < PB; ++u)
      if (psi[u] >= 0) rank[psi[u]] = prank[u];
  }
}

// The d value gather groups the SAMPLES by their anchor cell (h_low, w_low) in [-1, H-1] x [-1, W-1] -- one integer atomic per
// sample, not one per corner.  A value row (y, x) is corner k of exactly the samples anchored at
//   k = 0: (y, x)   k = 1: (y, x - 1)   k = 2: (y - 1, x)   k = 3: (y - 1, x - 1)
// so its gradient is the sum over those four anchor bins; bin index ((b * A + astart[l] + (h_low+1) * (W+1) + (w_low+1)) * NH + head.
template <bool FILL>
__global__ __launch_bounds__(256) void msda_anchor_bin(const float* __restrict__ loc, MsdaDims d, long long nsamples,
                                                       const int* __restrict__ start, int* __restrict__ count,
                                                       int* __restrict__ rank, unsigned* __restrict__ list) {
  const long long s = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (s >= nsamples) return;
  const int LP = d.L * d.P;
  const long long gid = s / LP;
  const int lp = static_cast<int>(s - gid * LP), l = lp / d.P;
  const int head = static_cast<int>(gid % d.NH);
  const int b = static_cast<int>(gid / (static_cast<long long>(d.NH) * d.Q));
  const int H = d.h[l], W = d.w[l];
  const float h_im = loc[2 * s + 1] * static_cast<float>(H) - 0.5f, w_im = loc[2 * s] * static_cast<float>(W) - 0.5f;
  if (!(h_im > -1.f && w_im > -1.f && h_im < static_cast<float>(H) && w_im < static_cast<float>(W))) return;
  const int ay = static_cast<int>(floorf(h_im)) + 1, ax = static_cast<int>(floorf(w_im)) + 1;
  const int bin = (b * d.A + d.astart[l] + ay * (W + 1) + ax) * d.NH + head;
  // counting pass: the arrival rank inside the bin is kept per sample, so the fill pass needs no second atomic (the order
  // inside a bin is arbitrary here and fixed by the sort that follows)
  if (FILL) list[start[bin] + rank[s]] = static_cast<unsigned>(s);
  else rank[s] = atomicAdd(&count[bin], 1);
}

// After the sort: one record per binned sample, in bin order -- the (b, q, head) row of grad_out it reads and its four
// corner coefficients attn * w_k.  The gather then streams its bins' records (coalesced) instead of chasing
// entry -> location / weight -> bilinear weights once per corner.
__global__ __launch_bounds__(256) void msda_expand(const unsigned* __restrict__ sorted, const int* __restrict__ start,
                                                   long long bins, const float* __restrict__ loc,
                                                   const float* __restrict__ attn, MsdaDims d, unsigned* __restrict__ rec_row,
                                                   float4* __restrict__ rec_w) {
  const long long e = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (e >= start[bins]) return;
  const unsigned s = sorted[e];
  const unsigned LP = static_cast<unsigned>(d.L * d.P);
  const unsigned gid = s / LP;
  const int l = static_cast<int>(s - gid * LP) / d.P;
  const Bil t = bil_of(loc[2 * static_cast<size_t>(s)], loc[2 * static_cast<size_t>(s) + 1], d.h[l], d.w[l]);
  const float a = attn[s];
  rec_row[e] = gid;
  rec_w[e] = make_float4(a * t.w1, a * t.w2, a * t.w3, a * t.w4);
}

__global__ __launch_bounds__(256) void msda_gv_gather(const float4* __restrict__ gout, const int* __restrict__ start,
                                                      const unsigned* __restrict__ rec_row,
                                                      const float4* __restrict__ rec_w, float4* __restrict__ gvalue,
                                                      MsdaDims d, long long vrows) {
  constexpr int DU = 8;
  const int G = d.D4;
  const int lane = threadIdx.x & 63;
  const int g0 = lane & ~(G - 1), q4 = lane & (G - 1);
  const long long row = (static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x) / G;
  if (row >= vrows) return;
  const int head = static_cast<int>(row % d.NH);
  const long long bk = row / d.NH;
  const int b = static_cast<int>(bk / d.S), key = static_cast<int>(bk - static_cast<long long>(b) * d.S);
  int l = 0;
  while (l + 1 < d.L && key >= d.start[l + 1]) ++l;
  const int H = d.h[l], W = d.w[l], cell = key - d.start[l];
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (cell < H * W) {                                          // (rows past the last level's cells get a zero gradient)
    const int y = cell / W, x = cell - y * W;
    const int base = b * d.A + d.astart[l];
#pragma unroll 1
    for (int k = 0; k < 4; ++k) {                              // fixed order: corner, then ascending sample id
      const int ay = y + 1 - (k >> 1), ax = x + 1 - (k & 1);
      const int bin = (base + ay * (W + 1) + ax) * d.NH + head;
      const int st = start[bin];
      const int n = start[bin + 1] - st;
      for (int j0 = 0; j0 < n; j0 += G) {
        const int nb = min(G, n - j0);
        unsigned mygid = 0u;
        float mycoef = 0.f;
        if (q4 < nb) {
          mygid = rec_row[st + j0 + q4];
          const float4 w = rec_w[st + j0 + q4];
          mycoef = k == 0 ? w.x : (k == 1 ? w.y : (k == 2 ? w.z : w.w));
        }
        for (int h = 0; h < nb; h += DU) {                      // uniform inside the group
          float4 v[DU];
#pragma unroll
          for (int u = 0; u < DU; ++u) {
            const unsigned gi = __shfl(mygid, g0 | ((h + u) & (G - 1)));
            v[u] = (h + u) < nb ? gout[static_cast<size_t>(gi) * G + q4] : make_float4(0.f, 0.f, 0.f, 0.f);
          }
#pragma unroll
          for (int u = 0; u < DU; ++u) {
            const float c = __shfl(mycoef, g0 | ((h + u) & (G - 1)));
            if ((h + u) < nb) fma4s(acc, c, v[u]);
          }
        }
      }
    }
  }
  gvalue[static_cast<size_t>(row) * G + q4] = acc;
}

// d value, D = 32 (8 lanes per row): one workgroup per 8 x 8 tile of value rows of one (sample, level, head).  Its 32 lane groups
// walk the 9 x 9 anchor bins that touch the tile, each bin ONCE for all four corners it feeds (msda_gv_gather walks every bin four
// times, once from each corner's value row: 4 x the grad_out gathers and 4 x the record reads); a sample's coefficients are
// rebuilt from its location / weight here, so the record pass (msda_expand) is gone as well.  Every (corner, value row) of the
// tile is produced by exactly one anchor: the four per-corner partial tiles sit in LDS, are written once each -- no atomics, no
// ordering between groups -- and summed in corner order at the end: bit-reproducible.  Anchors on a tile edge are walked by the
// neighbouring tile too (81 bins per 64 rows: 1.27 x instead of 4 x).
constexpr int MSDA_MT = 8;
__global__ __launch_bounds__(256) void msda_gv_tile(const float4* __restrict__ gout, const int* __restrict__ start,
                                                    const unsigned* __restrict__ sorted, const float* __restrict__ loc,
                                                    const float* __restrict__ attn, float4* __restrict__ gvalue, MsdaDims d,
                                                    int tiles_per_bh, int nblocks) {
  constexpr int G = 8, MT = MSDA_MT, NA = (MT + 1) * (MT + 1);
  __shared__ float4 part[4][MT * MT][G];                       // 32 KB
  // one XCD walks whole (sample, head) slices: the grad_out rows, locations and weights such a slice touches (Q x 128 + 256 + 128
  // bytes) stay in ITS L2 instead of being fetched line by line into all eight
  const int lb = xcd_block();
  if (lb >= nblocks) return;
  const int bh = lb / tiles_per_bh;
  int t = lb - bh * tiles_per_bh;
  const int b = bh / d.NH, head = bh - b * d.NH;
  int l = 0;
  for (;; ++l) {
    const int nt = ((d.h[l] + MT - 1) / MT) * ((d.w[l] + MT - 1) / MT);
    if (t < nt) break;
    t -= nt;
  }
  const int H = d.h[l], W = d.w[l];
  const int tw = (W + MT - 1) / MT;
  const int ty = t / tw, tx = t - ty * tw;
  const int lane = threadIdx.x & 63;
  const int g0 = lane & ~(G - 1), q4 = lane & (G - 1);
  const int grp = threadIdx.x / G;                             // 32 groups
  const int base = b * d.A + d.astart[l];
  const unsigned LP = static_cast<unsigned>(d.L * d.P);
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 1
  for (int a = grp; a < NA; a += 32) {
    const int aty = a / (MT + 1), atx = a - aty * (MT + 1);
    const int ay = ty * MT + aty, ax = tx * MT + atx;
    float4 acc[4] = {z, z, z, z};
    if (ay <= H && ax <= W) {
      const int bin = (base + ay * (W + 1) + ax) * d.NH + head;
      const int st = start[bin];
      const int n = start[bin + 1] - st;
      // three-deep pipeline over batches of 8 samples: sample ids two batches ahead, location / weight one batch ahead, the
      // grad_out rows of the current batch -- one L2 round trip per batch instead of three dependent ones
      unsigned sm_n = q4 < n ? sorted[st + q4] : 0u;                       // batch 0
      float2 xy = *reinterpret_cast<const float2*>(loc + 2 * static_cast<size_t>(sm_n));
      float aw = q4 < n ? attn[sm_n] : 0.f;
      unsigned sm = sm_n;
      sm_n = G + q4 < n ? sorted[st + G + q4] : 0u;                        // batch 1
      for (int j0 = 0; j0 < n; j0 += G) {
        const int nb = min(G, n - j0);
        const unsigned mygid = sm / LP;
        const float h_im = xy.y * static_cast<float>(H) - 0.5f, w_im = xy.x * static_cast<float>(W) - 0.5f;
        const float lh = h_im - static_cast<float>(ay - 1), lw = w_im - static_cast<float>(ax - 1);
        const float hh = 1.f - lh, hw = 1.f - lw;
        const float c0 = aw * (hh * hw), c1 = aw * (hh * lw), c2 = aw * (lh * hw), c3 = aw * (lh * lw);   // aw = 0 past the bin's end
        float4 v[G];
#pragma unroll
        for (int u = 0; u < G; ++u) {
          const unsigned gi = __shfl(mygid, g0 | u);
          v[u] = u < nb ? gout[static_cast<size_t>(gi) * G + q4] : z;
        }
        sm = sm_n;                                                          // next batch's location / weight, the one after's ids
        xy = *reinterpret_cast<const float2*>(loc + 2 * static_cast<size_t>(sm));
        aw = j0 + G + q4 < n ? attn[sm] : 0.f;
        sm_n = j0 + 2 * G + q4 < n ? sorted[st + j0 + 2 * G + q4] : 0u;
#pragma unroll
        for (int u = 0; u < G; ++u) {                          // ascending sample id inside each corner's sum
          fma4s(acc[0], __shfl(c0, g0 | u), v[u]);              // (lanes past nb hold zero coefficients and zero rows)
          fma4s(acc[1], __shfl(c1, g0 | u), v[u]);
          fma4s(acc[2], __shfl(c2, g0 | u), v[u]);
          fma4s(acc[3], __shfl(c3, g0 | u), v[u]);
        }
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {                              // corner k of anchor (ay, ax) is value row (ay - 1 + k/2, ax - 1 + k%2)
      const int ly = aty - 1 + (k >> 1), lx = atx - 1 + (k & 1);
      if (ly >= 0 && ly < MT && lx >= 0 && lx < MT) part[k][ly * MT + lx][q4] = acc[k];
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < MT * MT * G; i += 256) {
    const int r = i / G, q = i - r * G;
    const int y = ty * MT + r / MT, x = tx * MT + (r & (MT - 1));
    if (y < H && x < W) {
      float4 s0 = part[0][r][q];
      const float4 s1 = part[1][r][q], s2 = part[2][r][q], s3 = part[3][r][q];
      s0.x = ((s0.x + s1.x) + s2.x) + s3.x; s0.y = ((s0.y + s1.y) + s2.y) + s3.y;
      s0.z = ((s0.z + s1.z) + s2.z) + s3.z; s0.w = ((s0.w + s1.w) + s2.w) + s3.w;
      st_nt(&gvalue[((static_cast<size_t>(b) * d.S + d.start[l] + y * W + x) * d.NH + head) * G + q], s0);
    }
  }
}

size_t align_up256(size_t b) { return (b + 255) & ~static_cast<size_t>(255); }

struct MsdaWs { size_t count, start, list, sorted, recw, scanws, sortws, total; };
MsdaWs msda_ws(long long vrows /* anchor bins */, long long nent) {
  MsdaWs L;
  size_t o = 0;
  L.count = o;  o += align_up256(sizeof(int) * vrows);
  L.start = o;  o += align_up256(sizeof(int) * (vrows + 1));
  L.list = o;   o += align_up256(sizeof(int) * nent);
  L.sorted = o; o += align_up256(sizeof(int) * nent);
  L.recw = o;   o += align_up256(sizeof(float) * 4 * nent);
  L.scanws = o; o += align_up256(sizeof(int) * dbev::scan_workspace_ints(vrows));
  L.sortws = o; o += align_up256(sizeof(int) * dbev::segment_sort_workspace_ints(nent));
  L.total = o;
  return L;
}

// (H + 1)(W + 1) <= 2 H W + 2 for H, W >= 1: bound of the anchor-bin count that needs no shapes (workspace query)
long long msda_max_anchors(int S, int L) { return 2LL * S + 2LL * L; }

bool msda_dims(int B, int S, int NH, int D, int Q, int L, int P, const int32_t* shapes_hw, const int32_t* level_start,
               MsdaDims* d) {
  if (B <= 0 || S <= 0 || NH <= 0 || D <= 0 || (D & 3) || Q <= 0 || L <= 0 || L > MSDA_MAX_LEVELS || P <= 0 ||
      shapes_hw == nullptr || level_start == nullptr)
    return false;
  const int D4 = D >> 2;
  if (D4 > 64 || (D4 & (D4 - 1))) return false;              // lane groups are powers of two within a wave
  d->B = B; d->S = S; d->NH = NH; d->D4 = D4; d->Q = Q; d->L = L; d->P = P;
  long long tot = 0, anchors = 0;
  for (int l = 0; l < L; ++l) {
    d->h[l] = shapes_hw[2 * l];
    d->w[l] = shapes_hw[2 * l + 1];
    d->start[l] = level_start[l];
    if (d->h[l] <= 0 || d->w[l] <= 0 || d->start[l] < 0) return false;
    if (l > 0 && d->start[l] < d->start[l - 1] + d->h[l - 1] * d->w[l - 1]) return false;   // levels in ascending, disjoint order
    tot = d->start[l] + static_cast<long long>(d->h[l]) * d->w[l];
    if (tot > S) return false;
    d->astart[l] = static_cast<int>(anchors);
    anchors += static_cast<long long>(d->h[l] + 1) * (d->w[l] + 1);
  }
  if (anchors > msda_max_anchors(S, L)) return false;
  d->A = static_cast<int>(anchors);
  const long long nsamples = static_cast<long long>(B) * Q * NH * L * P;
  return nsamples < (1LL << 32) - 1 && static_cast<long long>(B) * anchors * NH < 0x7fffffffLL &&
         static_cast<long long>(B) * S * NH < 0x7fffffffLL;
}

}  // namespace

extern "C" size_t dbev_msda_backward_workspace_bytes(int B, int S, int NH, int Q, int L, int P) {
  if (B <= 0 || S <= 0 || NH <= 0 || Q <= 0 || L <= 0 || P <= 0) return 0;
  const long long nsamples = static_cast<long long>(B) * Q * NH * L * P;
  const long long bins = static_cast<long long>(B) * msda_max_anchors(S, L) * NH;
  if (nsamples >= (1LL << 32) - 1 || bins >= 0x7fffffffLL) return 0;
  return msda_ws(bins, nsamples).total;
}

extern "C" int dbev_msda_forward(const float* value, const int32_t* spatial_shapes_hw_host,
                                 const int32_t* level_start_host, const float* sampling_loc, const float* attn_weight,
                                 int B, int S, int NH, int D, int Q, int L, int P, float* out, dbevStream_t stream) {
  MsdaDims d;
  if (!msda_dims(B, S, NH, D, Q, L, P, spatial_shapes_hw_host, level_start_host, &d)) return DBEV_EINVAL;
  if (value == nullptr || sampling_loc == nullptr || attn_weight == nullptr || out == nullptr) return DBEV_EINVAL;
  const long long rows = static_cast<long long>(B) * Q * NH;
  const long long threads = rows * d.D4;
  const dim3 grid(static_cast<unsigned>((threads + 255) / 256));
  // algorithmic bytes: value read once, locations + weights, output (the 4 x L x P corner gathers hit L2 / MALL)
  DbevKt kt(DBEV_K_MSDA_FWD, 4LL * B * S * NH * D + 12LL * rows * L * P + 4LL * rows * D, dbev_stream(stream));
  if (P >= 8)
    hipLaunchKernelGGL((msda_fwd<8>), grid, dim3(256), 0, dbev_stream(stream), reinterpret_cast<const float4*>(value),
                       sampling_loc, attn_weight, reinterpret_cast<float4*>(out), d, rows);
  else
    hipLaunchKernelGGL((msda_fwd<4>), grid, dim3(256), 0, dbev_stream(stream), reinterpret_cast<const float4*>(value),
                       sampling_loc, attn_weight, reinterpret_cast<float4*>(out), d, rows);
  DBEV_LAUNCH_CHECK();
  return 0;
}

extern "C" int dbev_msda_backward(const float* value, const int32_t* spatial_shapes_hw_host,
                                  const int32_t* level_start_host, const float* sampling_loc, const float* attn_weight,
                                  const float* grad_out, int B, int S, int NH, int D, int Q, int L, int P,
                                  float* grad_value, float* grad_sampling_loc, float* grad_attn_weight, void* workspace,
                                  size_t workspace_bytes, dbevStream_t stream) {
  MsdaDims d;
  if (!msda_dims(B, S, NH, D, Q, L, P, spatial_shapes_hw_host, level_start_host, &d)) return DBEV_EINVAL;
  if (value == nullptr || sampling_loc == nullptr || attn_weight == nullptr || grad_out == nullptr ||
      grad_value == nullptr || grad_sampling_loc == nullptr || grad_attn_weight == nullptr || workspace == nullptr)
    return DBEV_EINVAL;
  const long long rows = static_cast<long long>(B) * Q * NH, vrows = static_cast<long long>(B) * S * NH;
  const long long nsamples = rows * L * P;
  const long long bins = static_cast<long long>(B) * d.A * NH;
  const MsdaWs Lw = msda_ws(static_cast<long long>(B) * msda_max_anchors(S, L) * NH, nsamples);     // layout of the workspace query
  if (workspace_bytes < Lw.total) return DBEV_EINVAL;
  hipStream_t s = dbev_stream(stream);
  char* ws = static_cast<char*>(workspace);
  int* count = reinterpret_cast<int*>(ws + Lw.count);
  int* start = reinterpret_cast<int*>(ws + Lw.start);
  unsigned* list = reinterpret_cast<unsigned*>(ws + Lw.list);
  unsigned* sorted = reinterpret_cast<unsigned*>(ws + Lw.sorted);
  const float4* v4 = reinterpret_cast<const float4*>(value);
  const float4* g4 = reinterpret_cast<const float4*>(grad_out);
  int* rank = reinterpret_cast<int*>(sorted);                 // dead before the sort writes `sorted`
  static const bool fused_count = !(getenv("DBEV_MSDA_FUSED_COUNT") && atoi(getenv("DBEV_MSDA_FUSED_COUNT")) == 0);
  static const bool tile_ok = !(getenv("DBEV_MSDA_TILE") && atoi(getenv("DBEV_MSDA_TILE")) == 0);
  DBEV_HIP_TRY(hipMemsetAsync(count, 0, sizeof(int) * bins, s));
  { DbevKt kt(DBEV_K_MSDA_BWD_SAMPLE, 4LL * B * S * NH * D + 24LL * nsamples + 4LL * rows * D, s);
  const dim3 bgrid(static_cast<unsigned>((rows * d.D4 + 255) / 256));
  if (d.D4 == 8 && fused_count)
    hipLaunchKernelGGL((msda_bwd_sample<8, 4, true>), bgrid, dim3(256), 0, s, v4, sampling_loc, attn_weight, g4, grad_sampling_loc,
                       grad_attn_weight, d, rows, count, rank);
  else if (d.D4 == 8)
    hipLaunchKernelGGL((msda_bwd_sample<8, 4, false>), bgrid, dim3(256), 0, s, v4, sampling_loc, attn_weight, g4, grad_sampling_loc,
                       grad_attn_weight, d, rows, count, rank);
  else if (fused_count)
    hipLaunchKernelGGL((msda_bwd_sample<0, 4, true>), bgrid, dim3(256), 0, s, v4, sampling_loc, attn_weight, g4, grad_sampling_loc,
                       grad_attn_weight, d, rows, count, rank);
  else
    hipLaunchKernelGGL((msda_bwd_sample<0, 4, false>), bgrid, dim3(256), 0, s, v4, sampling_loc, attn_weight, g4, grad_sampling_loc,
                       grad_attn_weight, d, rows, count, rank); }
  const dim3 sgrid(static_cast<unsigned>((nsamples + 255) / 256));
  if (!fused_count)
    hipLaunchKernelGGL((msda_anchor_bin<false>), sgrid, dim3(256), 0, s, sampling_loc, d, nsamples, start, count, rank, list);
  int rc = dbev::exclusive_scan_i32(count, start, bins, false, nullptr, reinterpret_cast<int*>(ws + Lw.scanws), s);
  if (rc) return rc;
  hipLaunchKernelGGL((msda_anchor_bin<true>), sgrid, dim3(256), 0, s, sampling_loc, d, nsamples, start, count, rank, list);
  rc = dbev::segment_sort_u32(start, list, sorted, static_cast<int>(bins), reinterpret_cast<int*>(ws + Lw.sortws), s);
  if (rc) return rc;
  DbevKt kt(DBEV_K_MSDA_GV_GATHER, 4LL * B * S * NH * D + 20LL * nsamples + 4LL * rows * D, s);
  if (d.D4 == 8 && tile_ok) {
    long long cells = 0, tiles = 0;
    bool dense = true;                                          // the levels tile [0, S) without gaps: every row is written
    for (int l = 0; l < L; ++l) {
      dense = dense && d.start[l] == cells;
      cells += static_cast<long long>(d.h[l]) * d.w[l];
      tiles += static_cast<long long>((d.h[l] + MSDA_MT - 1) / MSDA_MT) * ((d.w[l] + MSDA_MT - 1) / MSDA_MT);
    }
    dense = dense && cells == S;
    if (tiles * B * NH >= 0x7fffffffLL) return DBEV_EINVAL;
    if (!dense) DBEV_HIP_TRY(hipMemsetAsync(grad_value, 0, sizeof(float) * vrows * D, s));
    const int nblocks = static_cast<int>(tiles * B * NH);
    hipLaunchKernelGGL(msda_gv_tile, dim3(dbev_round_xcd(nblocks)), dim3(256), 0, s, g4, start, sorted, sampling_loc,
                       attn_weight, reinterpret_cast<float4*>(grad_value), d, static_cast<int>(tiles), nblocks);
  } else {
    float4* rec_w = reinterpret_cast<float4*>(ws + Lw.recw);
    hipLaunchKernelGGL(msda_expand, sgrid, dim3(256), 0, s, sorted, start, bins, sampling_loc, attn_weight, d, list /* reused */,
                       rec_w);
    hipLaunchKernelGGL(msda_gv_gather, dim3(static_cast<unsigned>((vrows * d.D4 + 255) / 256)), dim3(256), 0, s, g4, start, list,
                       rec_w, reinterpret_cast<float4*>(grad_value), d, vrows);
  }
  DBEV_LAUNCH_CHECK();
  return 0;
}
