// Sparse 3-D convolution for the voxel teachers: rulebook build + gather-GEMM on the fp32 matrix cores.
//
// Replaces the spconv v1.x extension bundled with the reference (mmdet3d/ops/spconv: rulebook kernels
// include/spconv/indice.cu.h:24-215, gather / GEMM / scatter-add include/spconv/reordering.cu.h:22-128 and
// spconv_ops.h:302-348, Python surface ops.py:46-126, conv.py:60-225) -- CUDA only there (src/all.cc:15).
//
// Reference data flow per layer: for each of the K kernel offsets: gather the input rows of the offset's (in, out) pairs
// into a buffer, one GEMM with W[k], scatter-add the result rows into the output: 3 K kernel launches, 2 K passes over
// temporary buffers, the output read-modify-written K times.
// Here the rulebook is OUTPUT-STATIONARY: nbr[o, k] = input row that offset k pairs with output o (or -1).  One kernel
// per layer: a wave owns 16 output sites, walks k = 0..K-1 (the reference's accumulation order), gathers the neighbour
// rows straight into MFMA B operands, takes W[k] from LDS as the A operand and keeps the Cout x 16 accumulator tile in
// registers (v_mfma_f32_16x16x4_f32, exact fp32); offsets no site of the workgroup uses are skipped.  The output is
// written once.  No float atomics -> bit-reproducible.
//
// Rulebook: the occupied cells of a grid are a BITMAP + a popcount prefix per 32-cell word (a rank structure): the row of a
// cell is rank -> row[start[word] + popc(bits below)], two adjacent-word reads that the 27 neighbours of a site and the sites
// of a wave share (round 2 probed an open-addressing hash table: one random 64-byte sector per probe, 9 of the voxel teacher's
// 100 ms).  Output sites of a strided convolution = set bits of the bitmap over the output grid, enumerated in ascending cell
// order by the same popcount scan (the reference sorts the unique cell ids: same order).  The reference's own (K, 2, N) pair
// lists are derived from the neighbour table for API compatibility (ops.get_indice_pairs).
#include "common.h"
#include "prims.h"

#include <stdlib.h>

namespace {

struct SpGeom {
  int B, in_d[3], out_d[3], ks[3], st[3], pd[3], dl[3], K;
  int tr;      // transposed convolution (conv.py SparseConvTranspose*, indice.h getValidOutPosTranspose): out = in * stride - pad + k * dil
};

// occupied input cells: one bit per cell of the [B, D, H, W] grid (rows with out-of-range coordinates are ignored)
__device__ __forceinline__ long long sp_in_cell(const int4& c, const SpGeom& g) {
  if (c.x < 0 || c.x >= g.B || c.y < 0 || c.y >= g.in_d[0] || c.z < 0 || c.z >= g.in_d[1] || c.w < 0 || c.w >= g.in_d[2])
    return -1;
  return ((static_cast<long long>(c.x) * g.in_d[0] + c.y) * g.in_d[1] + c.z) * g.in_d[2] + c.w;
}

__global__ __launch_bounds__(256) void sp_mark_inputs(const int* __restrict__ idx, int n, SpGeom g, unsigned* __restrict__ bits) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const long long lin = sp_in_cell(reinterpret_cast<const int4*>(idx)[i], g);
  if (lin >= 0) atomicOr(&bits[lin >> 5], 1u << (lin & 31));
}

// rank -> row: duplicate coordinates keep the lowest row (row is pre-set to INT_MAX)
__global__ __launch_bounds__(256) void sp_rank_rows(const int* __restrict__ idx, int n, SpGeom g, const unsigned* __restrict__ bits,
                                                    const int* __restrict__ start, int* __restrict__ row) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const long long lin = sp_in_cell(reinterpret_cast<const int4*>(idx)[i], g);
  if (lin < 0) return;
  const unsigned w = bits[lin >> 5];
  atomicMin(&row[start[lin >> 5] + __popc(w & ((1u << (lin & 31)) - 1u))], i);
}

// strided / padded convolution: mark every output cell some (input, offset) pair reaches.  One thread per input site: per axis the
// kernel taps whose output coordinate is integral and in range (at stride 2, kernel 3: one or two of the three), then their product
// (<= 8 of the 27 offsets) -- a thread per (input, offset) pair spent 19 of 27 threads on the parity test alone.
__global__ __launch_bounds__(256) void sp_mark_outputs(const int* __restrict__ idx, int n, SpGeom g,
                                                       unsigned* __restrict__ bits) {
  constexpr int MAXT = 8;                                  // taps kept per axis (sp_geom admits kernel sizes <= 8 per axis)
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int4 c = reinterpret_cast<const int4*>(idx)[i];
  if (c.x < 0 || c.x >= g.B) return;
  const int in[3] = {c.y, c.z, c.w};
  int o[3][MAXT], cnt[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    cnt[a] = 0;
    for (int kk = 0; kk < g.ks[a]; ++kk) {
      int oo;
      if (g.tr) {
        oo = in[a] * g.st[a] - g.pd[a] + kk * g.dl[a];
        if (oo < 0) continue;
      } else {
        const int num = in[a] + g.pd[a] - kk * g.dl[a];
        if (num < 0 || num % g.st[a]) continue;
        oo = num / g.st[a];
      }
      if (oo >= g.out_d[a] || cnt[a] >= MAXT) continue;
      o[a][cnt[a]++] = oo;
    }
  }
  for (int a0 = 0; a0 < cnt[0]; ++a0)
    for (int a1 = 0; a1 < cnt[1]; ++a1) {
      const long long rowl = (static_cast<long long>(c.x) * g.out_d[0] + o[0][a0]) * g.out_d[1] + o[1][a1];
      for (int a2 = 0; a2 < cnt[2]; ++a2) {
        const long long lin = rowl * g.out_d[2] + o[2][a2];
        atomicOr(&bits[lin >> 5], 1u << (lin & 31));
      }
    }
}

__global__ __launch_bounds__(256) void sp_popcount(const unsigned* __restrict__ bits, int nwords, int* __restrict__ cnt) {
  const int w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w < nwords) cnt[w] = __popc(bits[w]);
}

__global__ __launch_bounds__(256) void sp_emit_outputs(const unsigned* __restrict__ bits, const int* __restrict__ start,
                                                       int nwords, SpGeom g, int* __restrict__ out_idx) {
  const int w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= nwords) return;
  unsigned b = bits[w];
  int pos = start[w];
  while (b) {
    const int bit = __ffs(b) - 1;
    b &= b - 1;
    long long lin = (static_cast<long long>(w) << 5) + bit;
    const int x = static_cast<int>(lin % g.out_d[2]); lin /= g.out_d[2];
    const int y = static_cast<int>(lin % g.out_d[1]); lin /= g.out_d[1];
    const int z = static_cast<int>(lin % g.out_d[0]); lin /= g.out_d[0];
    reinterpret_cast<int4*>(out_idx)[pos++] = make_int4(static_cast<int>(lin), z, y, x);
  }
}

// nbr[o, k] = input row at  o * stride - pad + k * dilation  (cross-correlation, like the dense convolution)
__global__ __launch_bounds__(256) void sp_neighbors(const int* __restrict__ out_idx, int m, SpGeom g,
                                                    const unsigned* __restrict__ bits, const int* __restrict__ start,
                                                    const int* __restrict__ row, int* __restrict__ nbr,
                                                    int* __restrict__ inv /* [N, K] or null */) {
  const long long t = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= static_cast<long long>(m) * g.K) return;
  const int o = static_cast<int>(t / g.K), k = static_cast<int>(t - static_cast<long long>(o) * g.K);
  const int4 c = reinterpret_cast<const int4*>(out_idx)[o];
  const int kz = k / (g.ks[1] * g.ks[2]), ky = (k / g.ks[2]) % g.ks[1], kx = k % g.ks[2];
  int z, y, x;
  bool ok = true;
  if (g.tr) {                                 // in = (out + pad - k * dil) / stride where that is integral
    const int nz = c.y + g.pd[0] - kz * g.dl[0], ny = c.z + g.pd[1] - ky * g.dl[1], nx = c.w + g.pd[2] - kx * g.dl[2];
    ok = nz >= 0 && ny >= 0 && nx >= 0 && nz % g.st[0] == 0 && ny % g.st[1] == 0 && nx % g.st[2] == 0;
    z = nz / g.st[0]; y = ny / g.st[1]; x = nx / g.st[2];
  } else {
    z = c.y * g.st[0] - g.pd[0] + kz * g.dl[0]; y = c.z * g.st[1] - g.pd[1] + ky * g.dl[1];
    x = c.w * g.st[2] - g.pd[2] + kx * g.dl[2];
  }
  int r = -1;
  if (ok && z >= 0 && z < g.in_d[0] && y >= 0 && y < g.in_d[1] && x >= 0 && x < g.in_d[2] && c.x >= 0 && c.x < g.B) {
    const long long lin = ((static_cast<long long>(c.x) * g.in_d[0] + z) * g.in_d[1] + y) * g.in_d[2] + x;
    const unsigned w = bits[lin >> 5], bit = 1u << (lin & 31);
    if (w & bit) r = row[start[lin >> 5] + __popc(w & (bit - 1u))];
  }
  nbr[t] = r;
  if (inv != nullptr && r >= 0) inv[static_cast<size_t>(r) * g.K + k] = o;     // unique writer per (input, offset)
}

// inverse table from the neighbour table: inv[r, k] = output row that offset k pairs with input row r (unique writer per entry)
__global__ __launch_bounds__(256) void sp_inverse(const int* __restrict__ nbr, long long nt, int K, int* __restrict__ inv) {
  const long long t = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= nt) return;
  const int r = nbr[t];
  if (r >= 0) inv[static_cast<size_t>(r) * K + t % K] = static_cast<int>(t / K);
}

// Sparse max pooling (pool.py:21-74, pool_ops.h:26-58, maxpool_cuda.cu:28-160): the reference starts from a ZERO output and raises it
// with every paired input, i.e. out[o, c] = max(0, max_k in[nbr[o, k], c]).  One float4 of channels per lane, offsets in order.
__global__ __launch_bounds__(256) void sp_maxpool_fwd(const float4* __restrict__ in, const int* __restrict__ nbr, long long total,
                                                      int K, int C4, float4* __restrict__ out) {
  const long long t = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const long long o = t / C4;
  const int c = static_cast<int>(t - o * C4);
  float4 m = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int k = 0; k < K; ++k) {
    const int r = nbr[o * K + k];
    if (r < 0) continue;
    const float4 v = in[static_cast<size_t>(r) * C4 + c];
    m.x = v.x > m.x ? v.x : m.x; m.y = v.y > m.y ? v.y : m.y; m.z = v.z > m.z ? v.z : m.z; m.w = v.w > m.w ? v.w : m.w;
  }
  out[t] = m;
}

// din[r, c] = sum over the pairs (r, o) with in[r, c] == out[o, c] of dout[o, c]  (maxpool_cuda.cu:165-230: every tie gets the
// gradient); gathered through the inverse table in offset order -- the order the reference's per-offset launches add in, no atomics
__global__ __launch_bounds__(256) void sp_maxpool_bwd(const float4* __restrict__ in, const float4* __restrict__ out,
                                                      const float4* __restrict__ dout, const int* __restrict__ inv,
                                                      long long total, int K, int C4, float4* __restrict__ din) {
  const long long t = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const long long r = t / C4;
  const int c = static_cast<int>(t - r * C4);
  const float4 v = in[t];
  float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int k = 0; k < K; ++k) {
    const int o = inv[r * K + k];
    if (o < 0) continue;
    const float4 y = out[static_cast<size_t>(o) * C4 + c], d = dout[static_cast<size_t>(o) * C4 + c];
    if (v.x == y.x) g.x += d.x;
    if (v.y == y.y) g.y += d.y;
    if (v.z == y.z) g.z += d.z;
    if (v.w == y.w) g.w += d.w;
  }
  din[t] = g;
}

// reference-format pair lists from the table: flags laid out [K, M]
__global__ __launch_bounds__(256) void sp_pair_flags(const int* __restrict__ nbr, int m, int K, int* __restrict__ flags) {
  const long long t = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= static_cast<long long>(m) * K) return;
  const int k = static_cast<int>(t / m), o = static_cast<int>(t - static_cast<long long>(k) * m);
  flags[t] = nbr[static_cast<size_t>(o) * K + k] >= 0 ? 1 : 0;
}

__global__ __launch_bounds__(256) void sp_pair_fill(const int* __restrict__ nbr, const int* __restrict__ pos, int m, int K,
                                                    int n_in, int* __restrict__ pairs /* [K, 2, n_in] */,
                                                    int* __restrict__ pair_num) {
  const long long t = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= static_cast<long long>(m) * K) return;
  const int k = static_cast<int>(t / m), o = static_cast<int>(t - static_cast<long long>(k) * m);
  const int base = pos[static_cast<size_t>(k) * m];
  if (o == 0) pair_num[k] = pos[static_cast<size_t>(k + 1) * m] - base;      // pos has K*M + 1 entries
  const int r = nbr[static_cast<size_t>(o) * K + k];
  if (r >= 0) {
    const int p = pos[t] - base;
    pairs[(static_cast<size_t>(k) * 2 + 0) * n_in + p] = r;
    pairs[(static_cast<size_t>(k) * 2 + 1) * n_in + p] = o;
  }
}

typedef float floatx4 __attribute__((ext_vector_type(4)));

// out[o, :] = bias + sum_k in[nbr[o, k], :] @ W[k]        W [K, Cin, Cout], Cin % 16 == 0, Cout = 16 * CT
// A wave owns NS = 2 sub-tiles of 16 output sites (a workgroup 128 sites): every A operand read from LDS (a weight
// column) feeds two MFMAs, and the weight slice W[k] is staged once per 128 sites.
constexpr int SP_NS = 2;
constexpr int SP_SITES = 64 * SP_NS;        // output sites per workgroup

// NCH > 1: the weight slice is staged in NCH chunks of Cin / NCH input channels (128 -> 128: 34 KB instead of 68 KB of LDS per
// workgroup, three resident workgroups per CU instead of two, for two more barriers per offset).
template <int CT, int NCH = 1>
__global__ __launch_bounds__(256) void sp_conv_fwd(const float* __restrict__ in, const float* __restrict__ W,
                                                   const float* __restrict__ bias, const int* __restrict__ nbr,
                                                   float* __restrict__ out, int M, int K, int Cin,
                                                   const float* __restrict__ scale, const float* __restrict__ residual,
                                                   int relu) {
  extern __shared__ float sW[];                     // [Cin][COUT + 4]
  constexpr int COUT = 16 * CT, STR = COUT + 4;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int j = lane & 15, kk = lane >> 4;
  int o[SP_NS];
#pragma unroll
  for (int s = 0; s < SP_NS; ++s) o[s] = blockIdx.x * SP_SITES + 16 * SP_NS * wv + 16 * s + j;
  floatx4 acc[SP_NS][CT];
#pragma unroll
  for (int s = 0; s < SP_NS; ++s)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) acc[s][ct] = floatx4{0.f, 0.f, 0.f, 0.f};
  const float4* W4 = reinterpret_cast<const float4*>(W);
  const int G16 = Cin / 16;
  int nb_next[SP_NS];
#pragma unroll
  for (int s = 0; s < SP_NS; ++s) nb_next[s] = o[s] < M ? nbr[static_cast<size_t>(o[s]) * K] : -1;
  for (int k = 0; k < K; ++k) {
    int nb[SP_NS];
    bool mine = false;
#pragma unroll
    for (int s = 0; s < SP_NS; ++s) {
      nb[s] = nb_next[s];
      if (k + 1 < K) nb_next[s] = o[s] < M ? nbr[static_cast<size_t>(o[s]) * K + k + 1] : -1;   // next offset's row ids in flight
      mine = mine || nb[s] >= 0;
    }
    const bool wave_any = __any(mine);
    if (!__syncthreads_or(wave_any)) continue;      // also the barrier that lets sW be overwritten
    const int CR = Cin / NCH, GR = G16 / NCH;       // input channels / 16-channel groups per staged chunk
    const float* row[SP_NS];
#pragma unroll
    for (int s = 0; s < SP_NS; ++s) row[s] = in + static_cast<size_t>(nb[s] >= 0 ? nb[s] : 0) * Cin + 4 * kk;
#pragma unroll 1
    for (int h = 0; h < NCH; ++h) {
      if (h > 0) __syncthreads();                   // every wave is done with the previous chunk
      for (int i = tid; i < CR * (COUT / 4); i += 256) {
        const int r = i / (COUT / 4), c4 = i - r * (COUT / 4);
        *reinterpret_cast<float4*>(&sW[r * STR + 4 * c4]) = W4[(static_cast<size_t>(k) * Cin + h * CR + r) * (COUT / 4) + c4];
      }
      // this lane's slices of the two neighbour rows (cin = 16 g + 4 kk + t), 2 groups (32 input channels) at a time: requested
      // before the barrier so that the gathers overlap the weight staging, the next pair is fetched under the MFMAs
      const int gb = h * GR, ge = gb + GR;
      float4 bv[SP_NS][2];
#pragma unroll
      for (int s = 0; s < SP_NS; ++s) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
          bv[s][u] = (nb[s] >= 0 && gb + u < ge) ? *reinterpret_cast<const float4*>(row[s] + 16 * (gb + u))
                                                 : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      __syncthreads();
      if (wave_any) {
        for (int g0 = gb; g0 < ge; g0 += 2) {
          float4 nx[SP_NS][2];
#pragma unroll
          for (int s = 0; s < SP_NS; ++s)
#pragma unroll
            for (int u = 0; u < 2; ++u)
              nx[s][u] = (nb[s] >= 0 && g0 + 2 + u < ge) ? *reinterpret_cast<const float4*>(row[s] + 16 * (g0 + 2 + u))
                                                         : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            if (g0 + u < ge) {
              const float b0[4] = {bv[0][u].x, bv[0][u].y, bv[0][u].z, bv[0][u].w};
              const float b1[4] = {bv[1][u].x, bv[1][u].y, bv[1][u].z, bv[1][u].w};
#pragma unroll
              for (int t = 0; t < 4; ++t) {
                const float* wr = &sW[(16 * (g0 + u - gb) + 4 * kk + t) * STR + j];
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                  const float a = wr[16 * ct];
                  acc[0][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b0[t], acc[0][ct], 0, 0, 0);
                  acc[1][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b1[t], acc[1][ct], 0, 0, 0);
                }
              }
            }
          }
#pragma unroll
          for (int s = 0; s < SP_NS; ++s)
#pragma unroll
            for (int u = 0; u < 2; ++u) bv[s][u] = nx[s][u];
        }
      }
    }
  }
#pragma unroll
  for (int s = 0; s < SP_NS; ++s) {
    if (o[s] < M) {
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {             // accumulator rows = output channels 16 ct + 4 kk + (0..3)
        float4 v = make_float4(acc[s][ct][0], acc[s][ct][1], acc[s][ct][2], acc[s][ct][3]);
        if (scale != nullptr) {                     // folded eval-mode BatchNorm1d: y = conv * scale + shift (shift in `bias`)
          const float4 sc = *reinterpret_cast<const float4*>(scale + 16 * ct + 4 * kk);
          v.x *= sc.x; v.y *= sc.y; v.z *= sc.z; v.w *= sc.w;
        }
        if (bias != nullptr) {
          const float4 bi = *reinterpret_cast<const float4*>(bias + 16 * ct + 4 * kk);
          v.x += bi.x; v.y += bi.y; v.z += bi.z; v.w += bi.w;
        }
        if (residual != nullptr) {
          const float4 rs = *reinterpret_cast<const float4*>(residual + static_cast<size_t>(o[s]) * COUT + 16 * ct + 4 * kk);
          v.x += rs.x; v.y += rs.y; v.z += rs.z; v.w += rs.w;
        }
        if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        *reinterpret_cast<float4*>(out + static_cast<size_t>(o[s]) * COUT + 16 * ct + 4 * kk) = v;
      }
    }
  }
}

// The same convolution for the SPARSE early stages (16 / 32 channels, 1 ... 6 of 27 neighbours per site -- the output-stationary kernel
// above multiplies an offset for all 16 sites of a tile when any of them has that neighbour: 4 ... 25 x the valid pairs there).
// Here a wave owns 64 output sites and, per kernel offset, COMPACTS its valid (site, input row) pairs (ballot + prefix count into a
// wave-private LDS list), multiplies them 16 pairs per MFMA tile and adds the 16 x Cout results into a wave-private LDS
// accumulator [64 sites][Cout] (read-modify-write, no atomics: a site appears once per offset, offsets are walked in order and a
// wave's LDS operations execute in order -> the reference's accumulation order per site, bit-reproducible).  One tile per offset
// holds ~15 of 16 valid pairs at 6 neighbours per site instead of 4 of 16.
constexpr int SPC_SITES = 256;          // output sites per workgroup (64 per wave)
constexpr int SPC_MAXK = 27;

// The wave's 64 x K slice of the neighbour table is one contiguous block of memory and comes in with coalesced loads once (the
// per-offset strided read of one int per lane cost more than the gathers).  Weights: WLDS = the workgroup stages W[k] in LDS per
// offset (two barriers per offset, every A operand an LDS read); !WLDS (16 -> 16 channels: 4 values per lane) = each lane loads
// its A operands of W[k] straight into registers (L2 hits) and the four waves never synchronise.
template <int CT, int G16, bool WLDS>
__global__ __launch_bounds__(256) void sp_conv_fwd_cmp(const float* __restrict__ in, const float* __restrict__ W,
                                                       const float* __restrict__ bias, const int* __restrict__ nbr,
                                                       float* __restrict__ out, int M, int K,
                                                       const float* __restrict__ scale, const float* __restrict__ residual, int relu) {
  extern __shared__ float lds[];
  constexpr int COUT = 16 * CT, CIN = 16 * G16, STR = COUT + 4;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int j = lane & 15, kk = lane >> 4;
  float* wacc = lds + 64 * wv * STR;                                        // [64 sites][STR]   (wave-private)
  int2* wl = reinterpret_cast<int2*>(lds + SPC_SITES * STR) + 64 * wv;      // [64] (site in wave, input row)
  float* sW = lds + SPC_SITES * STR + 2 * SPC_SITES;                        // WLDS: [CIN][STR]
  int* wnb = reinterpret_cast<int*>(lds + SPC_SITES * STR + 2 * SPC_SITES) + 64 * K * wv;   // !WLDS: [64 sites][K]
  const int base = blockIdx.x * SPC_SITES + 64 * wv;
  if (!WLDS && base >= M) return;
  for (int i = lane; i < 64 * STR / 4; i += 64) reinterpret_cast<float4*>(wacc)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  const int o = base + lane;
  int nb_next = -1;
  if (WLDS) {                                          // (staging the table costs the third resident workgroup here: strided reads, one offset ahead)
    nb_next = o < M ? nbr[static_cast<size_t>(o) * K] : -1;
  } else {
    const long long nbr_end = static_cast<long long>(M) * K;
    for (int i = lane; i < 64 * K; i += 64) {
      const long long gi = static_cast<long long>(base) * K + i;
      wnb[i] = gi < nbr_end ? nbr[gi] : -1;
    }
    __builtin_amdgcn_wave_barrier();
  }
  const float4* W4 = reinterpret_cast<const float4*>(W);
  for (int k = 0; k < K; ++k) {
    int nb;
    if (WLDS) {
      nb = nb_next;
      if (k + 1 < K) nb_next = o < M ? nbr[static_cast<size_t>(o) * K + k + 1] : -1;
    } else {
      nb = wnb[lane * K + k];
    }
    const unsigned long long mask = __ballot(nb >= 0);
    const int n = __popcll(mask);
    float a[G16][4][CT];                                                     // !WLDS: W[k] in the A-operand layout of this lane
    if (WLDS) {
      if (!__syncthreads_or(n > 0)) continue;                                // also the barrier that lets sW be overwritten
      for (int i = tid; i < CIN * (COUT / 4); i += 256) {
        const int row = i / (COUT / 4), c4 = i - row * (COUT / 4);
        *reinterpret_cast<float4*>(&sW[row * STR + 4 * c4]) = W4[(static_cast<size_t>(k) * CIN + row) * (COUT / 4) + c4];
      }
    } else {
      if (n == 0) continue;
#pragma unroll
      for (int g = 0; g < G16; ++g)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int ct = 0; ct < CT; ++ct) a[g][t][ct] = W[(static_cast<size_t>(k) * CIN + 16 * g + 4 * kk + t) * COUT + 16 * ct + j];
    }
    if (nb >= 0) wl[__popcll(mask & ((1ull << lane) - 1ull))] = make_int2(lane, nb);
    if (WLDS) __syncthreads(); else __builtin_amdgcn_wave_barrier();
    for (int t0 = 0; t0 < n; t0 += 16) {
      const int p = t0 + j;
      const bool valid = p < n;
      const int2 e = valid ? wl[p] : make_int2(0, 0);
      const float* row = in + static_cast<size_t>(e.y) * CIN + 4 * kk;
      float4 bv[G16];
#pragma unroll
      for (int g = 0; g < G16; ++g) bv[g] = valid ? *reinterpret_cast<const float4*>(row + 16 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
      floatx4 acc[CT];
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) acc[ct] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int g = 0; g < G16; ++g) {
        const float b[4] = {bv[g].x, bv[g].y, bv[g].z, bv[g].w};
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int ct = 0; ct < CT; ++ct)
            acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(WLDS ? sW[(16 * g + 4 * kk + t) * STR + 16 * ct + j] : a[g][t][ct], b[t], acc[ct], 0, 0, 0);
      }
      if (valid) {                                     // accumulator rows = output channels 16 ct + 4 kk + (0..3) of pair j
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
          float4* d = reinterpret_cast<float4*>(wacc + e.x * STR + 16 * ct + 4 * kk);
          float4 v = *d;
          v.x += acc[ct][0]; v.y += acc[ct][1]; v.z += acc[ct][2]; v.w += acc[ct][3];
          *d = v;
        }
      }
    }
    __builtin_amdgcn_wave_barrier();                   // the list is rewritten by the next offset
  }
  for (int i = lane; i < 64 * (COUT / 4); i += 64) {
    const int site = i / (COUT / 4), c4 = i - site * (COUT / 4);
    const int o2 = base + site;
    if (o2 >= M) continue;
    float4 v = *reinterpret_cast<const float4*>(wacc + site * STR + 4 * c4);
    if (scale != nullptr) {
      const float4 sc = *reinterpret_cast<const float4*>(scale + 4 * c4);
      v.x *= sc.x; v.y *= sc.y; v.z *= sc.z; v.w *= sc.w;
    }
    if (bias != nullptr) {
      const float4 bi = *reinterpret_cast<const float4*>(bias + 4 * c4);
      v.x += bi.x; v.y += bi.y; v.z += bi.z; v.w += bi.w;
    }
    if (residual != nullptr) {
      const float4 rs = *reinterpret_cast<const float4*>(residual + static_cast<size_t>(o2) * COUT + 4 * c4);
      v.x += rs.x; v.y += rs.y; v.z += rs.z; v.w += rs.w;
    }
    if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    *reinterpret_cast<float4*>(out + static_cast<size_t>(o2) * COUT + 4 * c4) = v;
  }
}

// ---- backward (spconv_ops.h:352-420 indice_conv_backward: per offset k a gather, two GEMMs and a scatter-add with float atomics) ----
// Data gradient: din[i, :] = sum_k dout[inv[i, k], :] @ W[k]^T is the FORWARD gather-GEMM on the inverse table with the per-offset
// transposed weights (sp_transpose_w + sp_conv_fwd): input-stationary, written once, no atomics.
// Weight gradient: dW[k] = sum over the offset's (in, out) pairs of in[i, :]^T (x) dout[o, :] -- a [Cin x Cout] GEMM per offset whose
// reduction dimension is the COMPACTED pair list of that offset (the reference's indice_pairs[k], ascending output row), so the
// matrix cores only see valid pairs.  Workgroup (part, k) walks its contiguous slice of the list in chunks of 64 pairs: both
// gathered row sets are staged in LDS (float4 row gathers), every wave keeps TR x TC accumulator tiles of 16 x 16 in registers
// (v_mfma_f32_16x16x4_f32: A = 4 pairs x 16 input channels, B = 4 pairs x 16 output channels) and writes one partial
// matrix; sp_wgrad_reduce adds the partials of an offset in slice order -> bit-reproducible.
__global__ __launch_bounds__(256) void sp_transpose_w(const float* __restrict__ W, float* __restrict__ Wt, int K, int Cin, int Cout) {
  const long long t = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= static_cast<long long>(K) * Cin * Cout) return;
  const int ci = static_cast<int>(t % Cin);
  const long long r = t / Cin;
  const int co = static_cast<int>(r % Cout), k = static_cast<int>(r / Cout);
  Wt[t] = W[(static_cast<size_t>(k) * Cin + ci) * Cout + co];            // Wt [K, Cout, Cin]
}

constexpr int SPW_CHUNK = 64;          // pairs staged per trip
// row padding (floats) of the two LDS stages: ds_read_b32 banks = (dword address) mod 32 inside each 32-lane half, and lanes l / l + 16
// read consecutive rows -> the row stride must be 16 mod 32 (C = 16: none, C a multiple of 32: 16)
__host__ __device__ constexpr int spw_pad(int C) { return (C & 31) == 0 ? 16 : 0; }

template <int TR, int TC>
__global__ __launch_bounds__(256) void sp_conv_wgrad(const float* __restrict__ in, const float* __restrict__ gout,
                                                     const int* __restrict__ pairs, const int* __restrict__ pair_num,
                                                     int pair_stride, int src, int slice, int Cin, int Cout, int WR, int WC,
                                                     float* __restrict__ partial) {
  extern __shared__ float lds[];
  const int SA = Cin + spw_pad(Cin), SG = Cout + spw_pad(Cout);
  float* As = lds;                                   // [SPW_CHUNK][SA]
  float* Gs = lds + SPW_CHUNK * SA;                  // [SPW_CHUNK][SG]
  int* rows = reinterpret_cast<int*>(Gs + SPW_CHUNK * SG);   // [2][SPW_CHUNK]: input row, output row (-1: past the list)
  const int part = blockIdx.x, k = blockIdx.y, nparts = gridDim.x;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int j = lane & 15, kk = lane >> 4;
  const int wr = wv / WC, wc = wv - wr * WC;         // this wave's block of tiles: rows wr*TR.., columns wc*TC..
  const bool active = wv < WR * WC;
  floatx4 acc[TR][TC];
#pragma unroll
  for (int a = 0; a < TR; ++a)
#pragma unroll
    for (int b = 0; b < TC; ++b) acc[a][b] = floatx4{0.f, 0.f, 0.f, 0.f};
  const int nk = pair_num[k];
  const int p0 = part * slice, p1 = min(nk, p0 + slice);
  const int* pin = pairs + (static_cast<size_t>(k) * 2 + src) * pair_stride;
  const int* pout = pairs + (static_cast<size_t>(k) * 2 + (1 - src)) * pair_stride;
  const int A4 = Cin >> 2, G4 = Cout >> 2;
  for (int c0 = p0; c0 < p1; c0 += SPW_CHUNK) {
    __syncthreads();                                 // the previous trip's tiles are consumed
    if (tid < SPW_CHUNK) {
      const int p = c0 + tid;
      rows[tid] = p < p1 ? pin[p] : -1;
      rows[SPW_CHUNK + tid] = p < p1 ? pout[p] : -1;
    }
    __syncthreads();
    for (int i = tid; i < SPW_CHUNK * A4; i += 256) {
      const int s = i / A4, c4 = i - s * A4, r = rows[s];
      const float4 v = r >= 0 ? reinterpret_cast<const float4*>(in)[static_cast<size_t>(r) * A4 + c4] : make_float4(0.f, 0.f, 0.f, 0.f);
      *reinterpret_cast<float4*>(&As[s * SA + 4 * c4]) = v;
    }
    for (int i = tid; i < SPW_CHUNK * G4; i += 256) {
      const int s = i / G4, c4 = i - s * G4, r = rows[SPW_CHUNK + s];
      const float4 v = r >= 0 ? reinterpret_cast<const float4*>(gout)[static_cast<size_t>(r) * G4 + c4] : make_float4(0.f, 0.f, 0.f, 0.f);
      *reinterpret_cast<float4*>(&Gs[s * SG + 4 * c4]) = v;
    }
    __syncthreads();
    if (active) {
      const int ng = (min(p1 - c0, SPW_CHUNK) + 3) >> 2;       // groups of 4 pairs that hold any
      for (int g = 0; g < ng; ++g) {
        float a[TR], b[TC];
#pragma unroll
        for (int t = 0; t < TR; ++t) a[t] = As[(4 * g + kk) * SA + 16 * (wr * TR + t) + j];
#pragma unroll
        for (int t = 0; t < TC; ++t) b[t] = Gs[(4 * g + kk) * SG + 16 * (wc * TC + t) + j];
#pragma unroll
        for (int ta = 0; ta < TR; ++ta)
#pragma unroll
          for (int tb = 0; tb < TC; ++tb) acc[ta][tb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ta], b[tb], acc[ta][tb], 0, 0, 0);
      }
    }
  }
  if (active) {                                      // partial [K][nparts][Cin][Cout]; a lane holds rows 4 kk + (0..3), column j of a tile
    float* P = partial + (static_cast<size_t>(k) * nparts + part) * Cin * Cout;
#pragma unroll
    for (int ta = 0; ta < TR; ++ta)
#pragma unroll
      for (int tb = 0; tb < TC; ++tb)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          P[static_cast<size_t>(16 * (wr * TR + ta) + 4 * kk + r) * Cout + 16 * (wc * TC + tb) + j] = acc[ta][tb][r];
  }
}

__global__ __launch_bounds__(256) void sp_wgrad_reduce(const float* __restrict__ partial, int nparts, long long per_k, int K,
                                                       float* __restrict__ gw) {
  const long long t = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= per_k * K) return;
  const long long k = t / per_k, e = t - k * per_k;
  const float* p = partial + k * nparts * per_k + e;
  float s = 0.f;
  for (int i = 0; i < nparts; ++i) s += p[static_cast<size_t>(i) * per_k];     // slice order: fixed
  gw[t] = s;
}

struct SpwPlan { int WR, WC, TR, TC, nparts, slice; size_t lds; };

bool spw_plan(int K, int Cin, int Cout, int n_pairs_max, SpwPlan* p) {
  const int ci = Cin / 16, co = Cout / 16;
  if (Cin <= 0 || Cout <= 0 || (Cin & 15) || (Cout & 15) || ci > 8 || co > 8 || (ci & (ci - 1)) || (co & (co - 1)) || K <= 0 || n_pairs_max < 0)
    return false;
  p->WR = ci >= 4 ? 4 : ci;
  p->WC = 4 / p->WR < co ? 4 / p->WR : co;
  p->TR = ci / p->WR;
  p->TC = co / p->WC;
  int nparts = 1024 / K;                             // ~4 workgroups per CU over all offsets
  if (nparts < 1) nparts = 1;
  if (nparts > 64) nparts = 64;
  int slice = (n_pairs_max + nparts - 1) / nparts;
  slice = (slice + SPW_CHUNK - 1) / SPW_CHUNK * SPW_CHUNK;
  if (slice < SPW_CHUNK) slice = SPW_CHUNK;
  p->nparts = (n_pairs_max + slice - 1) / slice;
  if (p->nparts < 1) p->nparts = 1;
  p->slice = slice;
  p->lds = sizeof(float) * SPW_CHUNK * (static_cast<size_t>(Cin) + Cout + spw_pad(Cin) + spw_pad(Cout)) + sizeof(int) * 2 * SPW_CHUNK;
  return true;
}

// dense() of a sparse tensor in the channels-first layout the encoders return: canvas [B, C, D, H, W] (pre-zeroed)
__global__ __launch_bounds__(256) void sp_to_dense(const float* __restrict__ feats, const int* __restrict__ idx, int n, int C,
                                                   int D, int H, int Wd, float* __restrict__ canvas) {
  const long long t = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= static_cast<long long>(n) * C) return;
  const int i = static_cast<int>(t / C), c = static_cast<int>(t - static_cast<long long>(i) * C);
  const int4 p = reinterpret_cast<const int4*>(idx)[i];
  canvas[(((static_cast<size_t>(p.x) * C + c) * D + p.y) * H + p.z) * Wd + p.w] = feats[t];
}

bool sp_geom(int B, const int32_t* in_d, const int32_t* out_d, const int32_t* ks, const int32_t* st, const int32_t* pd,
             const int32_t* dl, int transposed, SpGeom* g) {
  if (B <= 0 || !in_d || !out_d || !ks || !st || !pd || !dl) return false;
  g->B = B;
  g->tr = transposed ? 1 : 0;
  long long vin = B, vout = B;
  g->K = 1;
  for (int a = 0; a < 3; ++a) {
    g->in_d[a] = in_d[a]; g->out_d[a] = out_d[a]; g->ks[a] = ks[a]; g->st[a] = st[a]; g->pd[a] = pd[a]; g->dl[a] = dl[a];
    if (in_d[a] <= 0 || out_d[a] <= 0 || ks[a] <= 0 || st[a] <= 0 || pd[a] < 0 || dl[a] <= 0) return false;
    vin *= in_d[a]; vout *= out_d[a];
    g->K *= ks[a];
  }
  return vin < (1LL << 36) && vout < (1LL << 36) && g->K <= 125 && ks[0] <= 8 && ks[1] <= 8 && ks[2] <= 8;
}

size_t sp_align(size_t b) { return (b + 255) & ~static_cast<size_t>(255); }

// one workspace serves dbev_spconv_outputs and dbev_spconv_neighbors
struct SpWs { size_t obits, ocnt, ostart, ibits, icnt, istart, irow, flags, pos, scan, total; long long nwo, nwi; };
SpWs sp_ws(int n_in, int B, const int32_t* in_d, const int32_t* out_d, int K, int max_out) {
  SpWs w;
  const long long vin = static_cast<long long>(B) * in_d[0] * in_d[1] * in_d[2];
  const long long vout = static_cast<long long>(B) * out_d[0] * out_d[1] * out_d[2];
  w.nwi = (vin + 31) / 32;
  w.nwo = (vout + 31) / 32;
  const long long nflag = static_cast<long long>(K) * (max_out > 0 ? max_out : 1) + 1;
  long long nscan = w.nwo > nflag ? w.nwo : nflag;
  if (w.nwi > nscan) nscan = w.nwi;
  size_t o = 0;
  w.obits = o;  o += sp_align(sizeof(int) * w.nwo);
  w.ocnt = o;   o += sp_align(sizeof(int) * w.nwo);
  w.ostart = o; o += sp_align(sizeof(int) * (w.nwo + 1));
  w.ibits = o;  o += sp_align(sizeof(int) * w.nwi);
  w.icnt = o;   o += sp_align(sizeof(int) * w.nwi);
  w.istart = o; o += sp_align(sizeof(int) * (w.nwi + 1));
  w.irow = o;   o += sp_align(sizeof(int) * (n_in > 0 ? n_in : 1));
  w.flags = o;  o += sp_align(sizeof(int) * nflag);
  w.pos = o;    o += sp_align(sizeof(int) * nflag);
  w.scan = o;   o += sp_align(sizeof(int) * dbev::scan_workspace_ints(nscan));
  w.total = o + 4096;
  return w;
}

bool sp_dims_ok(int B, const int32_t* in_d, const int32_t* out_d) {
  if (B <= 0 || in_d == nullptr || out_d == nullptr) return false;
  for (int a = 0; a < 3; ++a)
    if (in_d[a] <= 0 || out_d[a] <= 0) return false;
  return static_cast<long long>(B) * in_d[0] * in_d[1] * in_d[2] < (1LL << 36) &&
         static_cast<long long>(B) * out_d[0] * out_d[1] * out_d[2] < (1LL << 36);
}

}  // namespace

// ---- rulebook --------------------------------------------------------------------------------------------------------
// workspace: bitmaps + per-word counts / starts over the input and the output grid, rank -> row table, pair-list flags, scan scratch
extern "C" size_t dbev_spconv_build_workspace_bytes(int n_in, int B, const int32_t* in_dims_host, const int32_t* out_dims_host, int K,
                                                    int max_out) {
  if (n_in < 0 || K <= 0 || !sp_dims_ok(B, in_dims_host, out_dims_host)) return 0;
  return sp_ws(n_in, B, in_dims_host, out_dims_host, K, max_out).total;
}

// Step 1 (strided / padded convolutions only; submanifold convolutions keep the input sites): enumerate the output sites.
// -> *n_out_device, out_indices [<= max_out, 4] in ascending cell order.  The caller reads n_out back (the reference
// does the same through num_act_out) and calls dbev_spconv_neighbors with it.
extern "C" int dbev_spconv_outputs(const int32_t* indices, int n_in, int B, const int32_t* in_dims_host,
                                   const int32_t* out_dims_host, const int32_t* ksize_host, const int32_t* stride_host,
                                   const int32_t* padding_host, const int32_t* dilation_host, int transposed,
                                   int32_t* out_indices, int max_out, int32_t* n_out_device, void* workspace,
                                   size_t workspace_bytes, dbevStream_t stream) {
  SpGeom g;
  if (!sp_geom(B, in_dims_host, out_dims_host, ksize_host, stride_host, padding_host, dilation_host, transposed, &g))
    return DBEV_EINVAL;
  if (n_in < 0 || (n_in > 0 && indices == nullptr) || out_indices == nullptr || n_out_device == nullptr || workspace == nullptr)
    return DBEV_EINVAL;
  const SpWs L = sp_ws(n_in, B, in_dims_host, out_dims_host, g.K, max_out);
  if (workspace_bytes < L.total) return DBEV_EINVAL;
  hipStream_t s = dbev_stream(stream);
  const int nwords = static_cast<int>(L.nwo);
  char* ws = static_cast<char*>(workspace);
  unsigned* bits = reinterpret_cast<unsigned*>(ws + L.obits);
  int* cnt = reinterpret_cast<int*>(ws + L.ocnt);
  int* start = reinterpret_cast<int*>(ws + L.ostart);
  int* scanws = reinterpret_cast<int*>(ws + L.scan);
  DBEV_HIP_TRY(hipMemsetAsync(bits, 0, sizeof(int) * nwords, s));
  if (n_in > 0)
    hipLaunchKernelGGL(sp_mark_outputs, dim3(dbev_ceil_div(n_in, 256)), dim3(256), 0, s, indices, n_in, g, bits);
  hipLaunchKernelGGL(sp_popcount, dim3(dbev_ceil_div(nwords, 256)), dim3(256), 0, s, bits, nwords, cnt);
  int rc = dbev::exclusive_scan_i32(cnt, start, nwords, false, n_out_device, scanws, s);
  if (rc) return rc;
  // an output set larger than max_out cannot happen for max_out >= min(n_in * K, grid volume) -- the Python side sizes it so
  hipLaunchKernelGGL(sp_emit_outputs, dim3(dbev_ceil_div(nwords, 256)), dim3(256), 0, s, bits, start, nwords, g, out_indices);
  DBEV_LAUNCH_CHECK();
  return 0;
}

// Step 2: neighbour table nbr [n_out, K] (+ the inverse table inv [n_in, K] for SparseInverseConv, + the reference's pair
// lists indice_pairs [K, 2, n_in] / indice_pair_num [K]; either may be NULL).
extern "C" int dbev_spconv_neighbors(const int32_t* indices, int n_in, const int32_t* out_indices, int n_out, int B,
                                     const int32_t* in_dims_host, const int32_t* out_dims_host, const int32_t* ksize_host,
                                     const int32_t* stride_host, const int32_t* padding_host,
                                     const int32_t* dilation_host, int transposed, int32_t* nbr, int32_t* inv,
                                     int32_t* indice_pairs, int32_t* indice_pair_num, void* workspace, size_t workspace_bytes,
                                     dbevStream_t stream) {
  SpGeom g;
  if (!sp_geom(B, in_dims_host, out_dims_host, ksize_host, stride_host, padding_host, dilation_host, transposed, &g))
    return DBEV_EINVAL;
  if (n_in < 0 || n_out < 0 || (n_in > 0 && indices == nullptr) || (n_out > 0 && (out_indices == nullptr || nbr == nullptr)) ||
      workspace == nullptr)
    return DBEV_EINVAL;
  const SpWs L = sp_ws(n_in, B, in_dims_host, out_dims_host, g.K, n_out);
  if (workspace_bytes < L.total) return DBEV_EINVAL;
  hipStream_t s = dbev_stream(stream);
  char* ws = static_cast<char*>(workspace);
  unsigned* ibits = reinterpret_cast<unsigned*>(ws + L.ibits);
  int* icnt = reinterpret_cast<int*>(ws + L.icnt);
  int* istart = reinterpret_cast<int*>(ws + L.istart);
  int* irow = reinterpret_cast<int*>(ws + L.irow);
  int* flags = reinterpret_cast<int*>(ws + L.flags);
  int* pos = reinterpret_cast<int*>(ws + L.pos);
  int* scanws = reinterpret_cast<int*>(ws + L.scan);
  const int nwi = static_cast<int>(L.nwi);
  // rank structure of the occupied input cells
  DBEV_HIP_TRY(hipMemsetAsync(ibits, 0, sizeof(int) * nwi, s));
  if (n_in > 0) {
    hipLaunchKernelGGL(sp_mark_inputs, dim3(dbev_ceil_div(n_in, 256)), dim3(256), 0, s, indices, n_in, g, ibits);
    DBEV_HIP_TRY(hipMemsetAsync(irow, 0x7f, sizeof(int) * n_in, s));
  }
  hipLaunchKernelGGL(sp_popcount, dim3(dbev_ceil_div(nwi, 256)), dim3(256), 0, s, ibits, nwi, icnt);
  int rc = dbev::exclusive_scan_i32(icnt, istart, nwi, false, nullptr, scanws, s);
  if (rc) return rc;
  if (n_in > 0)
    hipLaunchKernelGGL(sp_rank_rows, dim3(dbev_ceil_div(n_in, 256)), dim3(256), 0, s, indices, n_in, g, ibits, istart, irow);
  if (inv != nullptr && n_in > 0) DBEV_HIP_TRY(hipMemsetAsync(inv, 0xff, sizeof(int) * static_cast<size_t>(n_in) * g.K, s));
  if (n_out > 0) {
    const long long nt = static_cast<long long>(n_out) * g.K;
    hipLaunchKernelGGL(sp_neighbors, dim3(dbev_ceil_div(nt, 256)), dim3(256), 0, s, out_indices, n_out, g, ibits, istart, irow,
                       nbr, inv);
    if (indice_pairs != nullptr && indice_pair_num != nullptr && n_in > 0) {
      DBEV_HIP_TRY(hipMemsetAsync(indice_pairs, 0xff, sizeof(int) * static_cast<size_t>(g.K) * 2 * n_in, s));
      hipLaunchKernelGGL(sp_pair_flags, dim3(dbev_ceil_div(nt, 256)), dim3(256), 0, s, nbr, n_out, g.K, flags);
      rc = dbev::exclusive_scan_i32(flags, pos, nt, false, nullptr, scanws, s);
      if (rc) return rc;
      hipLaunchKernelGGL(sp_pair_fill, dim3(dbev_ceil_div(nt, 256)), dim3(256), 0, s, nbr, pos, n_out, g.K, n_in,
                         indice_pairs, indice_pair_num);
    }
  } else if (indice_pair_num != nullptr) {
    DBEV_HIP_TRY(hipMemsetAsync(indice_pair_num, 0, sizeof(int) * g.K, s));
  }
  DBEV_LAUNCH_CHECK();
  return 0;
}

// Step 2b (on demand): the reference's pair lists from an existing neighbour table. The training backward and the
// get_indice_pairs() API read them; an inference forward never does, so the rulebook builds them lazily.
extern "C" size_t dbev_spconv_pair_lists_workspace_bytes(int n_out, int K) {
  const long long nflag = static_cast<long long>(K) * (n_out > 0 ? n_out : 1) + 1;
  return sp_align(sizeof(int) * nflag) * 2 + sp_align(sizeof(int) * dbev::scan_workspace_ints(nflag)) + 256;
}

extern "C" int dbev_spconv_pair_lists(const int32_t* nbr, int n_out, int K, int n_in, int32_t* indice_pairs,
                                      int32_t* indice_pair_num, void* workspace, size_t workspace_bytes,
                                      dbevStream_t stream) {
  if (n_out < 0 || n_in < 0 || K <= 0 || indice_pairs == nullptr || indice_pair_num == nullptr || workspace == nullptr ||
      workspace_bytes < dbev_spconv_pair_lists_workspace_bytes(n_out, K) || (n_out > 0 && nbr == nullptr))
    return DBEV_EINVAL;
  hipStream_t s = dbev_stream(stream);
  DBEV_HIP_TRY(hipMemsetAsync(indice_pair_num, 0, sizeof(int) * K, s));
  if (n_out == 0 || n_in == 0) return 0;
  const long long nt = static_cast<long long>(n_out) * K;
  char* ws = static_cast<char*>(workspace);
  size_t o = 0;
  int* flags = reinterpret_cast<int*>(ws + o); o += sp_align(sizeof(int) * (nt + 1));
  int* pos = reinterpret_cast<int*>(ws + o);   o += sp_align(sizeof(int) * (nt + 1));
  int* scanws = reinterpret_cast<int*>(ws + o);
  DBEV_HIP_TRY(hipMemsetAsync(indice_pairs, 0xff, sizeof(int) * static_cast<size_t>(K) * 2 * n_in, s));
  hipLaunchKernelGGL(sp_pair_flags, dim3(dbev_ceil_div(nt, 256)), dim3(256), 0, s, nbr, n_out, K, flags);
  int rc = dbev::exclusive_scan_i32(flags, pos, nt, false, nullptr, scanws, s);
  if (rc) return rc;
  hipLaunchKernelGGL(sp_pair_fill, dim3(dbev_ceil_div(nt, 256)), dim3(256), 0, s, nbr, pos, n_out, K, n_in, indice_pairs,
                     indice_pair_num);
  DBEV_LAUNCH_CHECK();
  return 0;
}

// Step 2c (on demand): the inverse table alone, from an existing neighbour table (what dbev_spconv_neighbors writes when
// inv != NULL).  SparseInverseConv3d and the data gradient read it; an inference forward of the encoder never does.
extern "C" int dbev_spconv_inverse_table(const int32_t* nbr, int n_out, int K, int n_in, int32_t* inv, dbevStream_t stream) {
  if (n_out < 0 || n_in < 0 || K <= 0 || (n_in > 0 && inv == nullptr) || (n_out > 0 && nbr == nullptr)) return DBEV_EINVAL;
  hipStream_t s = dbev_stream(stream);
  if (n_in == 0) return 0;
  DBEV_HIP_TRY(hipMemsetAsync(inv, 0xff, sizeof(int) * static_cast<size_t>(n_in) * K, s));
  if (n_out > 0) {
    const long long nt = static_cast<long long>(n_out) * K;
    hipLaunchKernelGGL(sp_inverse, dim3(dbev_ceil_div(nt, 256)), dim3(256), 0, s, nbr, nt, K, inv);
  }
  DBEV_LAUNCH_CHECK();
  return 0;
}

// ---- max pooling -----------------------------------------------------------------------------------------------------
extern "C" int dbev_spconv_maxpool_forward(const float* features, const int32_t* nbr, int n_out, int K, int C, float* out_features,
                                           dbevStream_t stream) {
  if (n_out < 0 || K <= 0 || C <= 0 || (C & 3)) return DBEV_EINVAL;
  if (n_out == 0) return 0;
  if (features == nullptr || nbr == nullptr || out_features == nullptr) return DBEV_EINVAL;
  const long long total = static_cast<long long>(n_out) * (C / 4);
  hipLaunchKernelGGL(sp_maxpool_fwd, dim3(dbev_ceil_div(total, 256)), dim3(256), 0, dbev_stream(stream),
                     reinterpret_cast<const float4*>(features), nbr, total, K, C / 4, reinterpret_cast<float4*>(out_features));
  DBEV_LAUNCH_CHECK();
  return 0;
}

extern "C" int dbev_spconv_maxpool_backward(const float* features, const float* out_features, const float* grad_out,
                                            const int32_t* inv, int n_in, int K, int C, float* grad_in, dbevStream_t stream) {
  if (n_in < 0 || K <= 0 || C <= 0 || (C & 3)) return DBEV_EINVAL;
  if (n_in == 0) return 0;
  if (features == nullptr || out_features == nullptr || grad_out == nullptr || inv == nullptr || grad_in == nullptr)
    return DBEV_EINVAL;
  const long long total = static_cast<long long>(n_in) * (C / 4);
  hipLaunchKernelGGL(sp_maxpool_bwd, dim3(dbev_ceil_div(total, 256)), dim3(256), 0, dbev_stream(stream),
                     reinterpret_cast<const float4*>(features), reinterpret_cast<const float4*>(out_features),
                     reinterpret_cast<const float4*>(grad_out), inv, total, K, C / 4, reinterpret_cast<float4*>(grad_in));
  DBEV_LAUNCH_CHECK();
  return 0;
}

// ---- convolution -----------------------------------------------------------------------------------------------------
extern "C" int dbev_spconv_forward(const float* features, const float* weight, const float* bias, const int32_t* nbr,
                                   int n_out, int K, int Cin, int Cout, float* out_features, dbevStream_t stream) {
  return dbev_spconv_forward_fused(features, weight, nullptr, bias, nullptr, 0, nbr, n_out, K, Cin, Cout, out_features, stream);
}

extern "C" int dbev_spconv_forward_fused(const float* features, const float* weight, const float* scale, const float* shift,
                                         const float* residual, int relu, const int32_t* nbr, int n_out, int K, int Cin,
                                         int Cout, float* out_features, dbevStream_t stream) {
  const float* bias = shift;
  if (n_out < 0 || K <= 0 || Cin <= 0 || (Cin & 15) || Cout <= 0 || (Cout & 15) || Cout > 128 || Cin > 256) return DBEV_EINVAL;
  if (n_out == 0) return 0;
  if (features == nullptr || weight == nullptr || nbr == nullptr || out_features == nullptr) return DBEV_EINVAL;
  hipStream_t s = dbev_stream(stream);
  // the sparse early stages (<= 32 input channels: 1 ... 6 of 27 neighbours per site in the voxel teachers) run on the pair-compacting
  // kernel, the dense ones on the output-stationary kernel; DBEV_SPCONV_COMPACT=0 / 1 forces one of them (A/B runs)
  static const int force_cmp = getenv("DBEV_SPCONV_COMPACT") ? atoi(getenv("DBEV_SPCONV_COMPACT")) : -1;
  if ((force_cmp == 1 || force_cmp < 0) && Cin <= 32 && Cout <= 64 && K <= SPC_MAXK) {
    const bool wlds = !(Cin == 16 && Cout == 16);
    const size_t ldc = sizeof(float) * SPC_SITES * (Cout + 4) + sizeof(int2) * SPC_SITES +
                       (wlds ? sizeof(float) * Cin * (Cout + 4) : sizeof(int) * SPC_SITES * K);
    const dim3 gridc(dbev_ceil_div(n_out, SPC_SITES));
#define SPC_GO(CTV, GV, WL)                                                                                          \
  do {                                                                                                               \
    if (ldc > 64 * 1024)                                                                                             \
      DBEV_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(sp_conv_fwd_cmp<CTV, GV, WL>),                  \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(ldc)));          \
    hipLaunchKernelGGL((sp_conv_fwd_cmp<CTV, GV, WL>), gridc, dim3(256), ldc, s, features, weight, bias, nbr, out_features, n_out, \
                       K, scale, residual, relu);                                                                    \
  } while (0)
#define SPC_LAUNCH(CTV)                                                        \
  do {                                                                         \
    if (!wlds) SPC_GO(1, 1, false);                                            \
    else if (Cin == 16) SPC_GO(CTV, 1, true);                                  \
    else SPC_GO(CTV, 2, true);                                                 \
  } while (0)
    DbevKt kt(DBEV_K_SPCONV_FWD, 4LL * n_out * (Cout + K) + 4LL * K * Cin * Cout, s);
    switch (Cout / 16) {
      case 1: SPC_LAUNCH(1); break;
      case 2: SPC_LAUNCH(2); break;
      case 3: SPC_LAUNCH(3); break;
      default: SPC_LAUNCH(4); break;
    }
#undef SPC_LAUNCH
#undef SPC_GO
    DBEV_LAUNCH_CHECK();
    return 0;
  }
  static const int nch_env = getenv("DBEV_SPCONV_CHUNKS") ? atoi(getenv("DBEV_SPCONV_CHUNKS")) : 0;
  // two chunks when a whole slice would leave two workgroups per CU (68 KB at 128 -> 128)
  int nch = nch_env > 0 ? nch_env : (sizeof(float) * static_cast<size_t>(Cin) * (Cout + 4) > 48 * 1024 ? 2 : 1);
  if ((nch != 1 && nch != 2 && nch != 4) || Cin % (16 * nch) != 0 || Cout != 128) nch = 1;
  const size_t lds = sizeof(float) * static_cast<size_t>(Cin / nch) * (Cout + 4);
  const dim3 grid(dbev_ceil_div(n_out, SP_SITES));
#define SP_LAUNCH2(CTV, NCHV)                                                                                       \
  do {                                                                                                              \
    if (lds > 64 * 1024)                                                                                            \
      DBEV_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(sp_conv_fwd<CTV, NCHV>),                       \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));         \
    hipLaunchKernelGGL((sp_conv_fwd<CTV, NCHV>), grid, dim3(256), lds, s, features, weight, bias, nbr, out_features, n_out, K,  \
                       Cin, scale, residual, relu);                                                                 \
  } while (0)
#define SP_LAUNCH(CTV) SP_LAUNCH2(CTV, 1)
  // log entry: output rows + neighbour table + weights (the gathered input rows depend on the rulebook: added by the caller)
  DbevKt kt(DBEV_K_SPCONV_FWD, 4LL * n_out * (Cout + K) + 4LL * K * Cin * Cout, s);
  switch (Cout / 16) {
    case 1: SP_LAUNCH(1); break;
    case 2: SP_LAUNCH(2); break;
    case 3: SP_LAUNCH(3); break;
    case 4: SP_LAUNCH(4); break;
    case 5: SP_LAUNCH(5); break;
    case 6: SP_LAUNCH(6); break;
    case 7: SP_LAUNCH(7); break;
    default:
      if (nch == 2) SP_LAUNCH2(8, 2);
      else if (nch == 4) SP_LAUNCH2(8, 4);
      else SP_LAUNCH(8);
      break;
  }
#undef SP_LAUNCH
#undef SP_LAUNCH2
  DBEV_LAUNCH_CHECK();
  return 0;
}

extern "C" int dbev_sparse_to_dense(const float* features, const int32_t* indices, int n, int C, int B, int D, int H, int W,
                                    float* canvas_ncdhw, dbevStream_t stream) {
  if (n < 0 || C <= 0 || B <= 0 || D <= 0 || H <= 0 || W <= 0 || canvas_ncdhw == nullptr) return DBEV_EINVAL;
  hipStream_t s = dbev_stream(stream);
  DBEV_HIP_TRY(hipMemsetAsync(canvas_ncdhw, 0, sizeof(float) * static_cast<size_t>(B) * C * D * H * W, s));
  if (n > 0)
    hipLaunchKernelGGL(sp_to_dense, dim3(dbev_ceil_div(static_cast<long long>(n) * C, 256)), dim3(256), 0, s, features,
                       indices, n, C, D, H, W, canvas_ncdhw);
  DBEV_LAUNCH_CHECK();
  return 0;
}

// ---- backward --------------------------------------------------------------------------------------------------------
extern "C" int dbev_spconv_backward_data(const float* grad_out, const float* weight, const int32_t* inverse_table, int n_in,
                                         int K, int Cin, int Cout, float* grad_features, void* workspace,
                                         size_t workspace_bytes, dbevStream_t stream) {
  if (n_in < 0 || K <= 0 || Cin <= 0 || Cout <= 0 || (Cin & 15) || (Cout & 15) || Cin > 128 || Cout > 256) return DBEV_EINVAL;
  if (n_in == 0) return 0;
  const size_t need = sizeof(float) * static_cast<size_t>(K) * Cin * Cout;
  if (grad_out == nullptr || weight == nullptr || inverse_table == nullptr || grad_features == nullptr || workspace == nullptr ||
      workspace_bytes < need)
    return DBEV_EINVAL;
  hipStream_t s = dbev_stream(stream);
  float* Wt = static_cast<float*>(workspace);
  hipLaunchKernelGGL(sp_transpose_w, dim3(dbev_ceil_div(static_cast<long long>(K) * Cin * Cout, 256)), dim3(256), 0, s, weight, Wt,
                     K, Cin, Cout);
  DBEV_LAUNCH_CHECK();
  // the forward kernel with the roles exchanged: "input" rows = dout [*, Cout], "output" rows = din [n_in, Cin]
  return dbev_spconv_forward_fused(grad_out, Wt, nullptr, nullptr, nullptr, 0, inverse_table, n_in, K, Cout, Cin, grad_features, stream);
}

extern "C" size_t dbev_spconv_backward_weight_workspace_bytes(int K, int Cin, int Cout, int n_pairs_max) {
  SpwPlan p;
  if (!spw_plan(K, Cin, Cout, n_pairs_max, &p)) return 0;
  return sizeof(float) * static_cast<size_t>(K) * p.nparts * Cin * Cout + 256;
}

extern "C" int dbev_spconv_backward_weight(const float* features, const float* grad_out, const int32_t* indice_pairs,
                                           const int32_t* indice_pair_num, int pair_stride, int n_pairs_max, int inverse, int K,
                                           int Cin, int Cout, float* grad_weight, void* workspace, size_t workspace_bytes,
                                           dbevStream_t stream) {
  SpwPlan p;
  if (!spw_plan(K, Cin, Cout, n_pairs_max, &p) || pair_stride < n_pairs_max) return DBEV_EINVAL;
  if (grad_weight == nullptr || workspace == nullptr ||
      workspace_bytes < dbev_spconv_backward_weight_workspace_bytes(K, Cin, Cout, n_pairs_max))
    return DBEV_EINVAL;
  hipStream_t s = dbev_stream(stream);
  const long long per_k = static_cast<long long>(Cin) * Cout;
  if (n_pairs_max == 0) {
    DBEV_HIP_TRY(hipMemsetAsync(grad_weight, 0, sizeof(float) * per_k * K, s));
    return 0;
  }
  if (features == nullptr || grad_out == nullptr || indice_pairs == nullptr || indice_pair_num == nullptr) return DBEV_EINVAL;
  float* partial = static_cast<float*>(workspace);
  const dim3 grid(p.nparts, K);
  const int src = inverse ? 1 : 0;       // inverse convolution: the list's output rows are this layer's inputs
#define SPW_LAUNCH(TRV, TCV)                                                                                               \
  do {                                                                                                                     \
    if (p.lds > 64 * 1024)                                                                                                 \
      DBEV_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(sp_conv_wgrad<TRV, TCV>),                             \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(p.lds)));             \
    hipLaunchKernelGGL((sp_conv_wgrad<TRV, TCV>), grid, dim3(256), p.lds, s, features, grad_out, indice_pairs, indice_pair_num, \
                       pair_stride, src, p.slice, Cin, Cout, p.WR, p.WC, partial);                                         \
  } while (0)
  if (p.TR == 1) {
    if (p.TC == 1) SPW_LAUNCH(1, 1); else if (p.TC == 2) SPW_LAUNCH(1, 2); else if (p.TC == 4) SPW_LAUNCH(1, 4); else SPW_LAUNCH(1, 8);
  } else {
    if (p.TC == 1) SPW_LAUNCH(2, 1); else if (p.TC == 2) SPW_LAUNCH(2, 2); else if (p.TC == 4) SPW_LAUNCH(2, 4); else SPW_LAUNCH(2, 8);
  }
#undef SPW_LAUNCH
  hipLaunchKernelGGL(sp_wgrad_reduce, dim3(dbev_ceil_div(per_k * K, 256)), dim3(256), 0, s, partial, p.nparts, per_k, K, grad_weight);
  DBEV_LAUNCH_CHECK();
  return 0;
}
