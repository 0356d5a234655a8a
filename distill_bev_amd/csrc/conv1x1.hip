// 1x1 convolution (a plain GEMM in channels-last) on the fp32 matrix cores with the BatchNorm batch statistics of its OUTPUT in the
// epilogue -- the `conv1 / conv3 / downsample` layers of the ResNet bottlenecks (mmdet3d/models/bricks/res_block.py:102-230:
// conv -> norm -> act), whose output the reference's BatchNorm then re-reads in full for its mean / variance pass.
//
//   Y[m, n] = sum_k X[m, k] * W[n, k]            X [M, K] = NHWC activations (M = N*H*W pixels), W [Cout, Cin] = OIHW 1x1 weight
//   partial[g][0][n] = sum over the rows workgroup g owned of Y[m, n],   partial[g][1][n] = sum of Y[m, n]^2
//
// so that dbev_bn_act's finalize kernel can merge the partial rows exactly as it merges bn_stats' (same layout) and the
// statistics pass over the 69 ... 554 MB output never runs.  v_mfma_f32_32x32x2_f32: exact fp32 FMA chains, the precision of the
// reference's fp32 convolution.
//
// Mapping: a workgroup of 4 waves owns 128 pixels x NT*32 output channels per work item and walks a strided list of items
// (persistent: <= 2 workgroups per CU, the batch statistics of everything a workgroup computed stay in registers -> a few hundred
// partial rows however large M is).  MFMA rows = pixels (A operand = activation), columns = channels (B operand = weight): a lane
// holds ONE output channel and 16 pixel rows per 32x32 tile, so the per-channel sums are in-register adds and the output leaves the
// accumulators directly as 128-byte row segments (32 consecutive channels of a pixel per half-wave store).  K is walked in chunks
// of 32 through a double-buffered k-major LDS stage (conflict-free ds_read_b32 for both operands); the global loads of chunk c+1 --
// and of the next work item's first chunk during the epilogue -- are in flight while chunk c is multiplied.  Work items are ordered
// (pixel tile, channel slice) with the slice fastest and XCD-aware, so the slices of a tile re-read its activation rows from ONE
// XCD's L2, back to back.
#include "common.h"

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int C1_PXB = 128;           // pixels per work item (4 waves x 32)
constexpr int C1_KC = 32;             // K chunk
constexpr int C1_XSTR = C1_PXB + 1;   // k-major stage rows, pad 1: the transposing scalar writes of the loader spread over the banks

template <int NT, bool STATS>
__global__ __launch_bounds__(256, 2) void c1x1_fwd(const float* __restrict__ X, const float* __restrict__ Wt, float* __restrict__ Y,
                                                   float* __restrict__ partial, int M, int K, int N, int x_row_stride) {
  constexpr int NCH = NT * 32;
  constexpr int WSTR = NCH + 1;
  __shared__ float smem[2 * C1_KC * (C1_XSTR + WSTR)];
  float (*sX)[C1_KC][C1_XSTR] = reinterpret_cast<float (*)[C1_KC][C1_XSTR]>(smem);
  float (*sW)[C1_KC][WSTR] = reinterpret_cast<float (*)[C1_KC][WSTR]>(smem + 2 * C1_KC * C1_XSTR);
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int nsl = N / NCH;                                    // channel slices per pixel tile
  const int ntile = (M + C1_PXB - 1) / C1_PXB;
  const int nitem = ntile * nsl;
  const int G = gridDim.x;                                    // a multiple of 8 (XCDs) and of nsl: a workgroup keeps ONE slice
  const int L = xcd_block();
  const int slice = L % nsl, n0 = slice * NCH;
  const int nchunk = K / C1_KC;

  constexpr int XV = C1_PXB * C1_KC / 4 / 256;                // float4 loads per thread per chunk (8 lanes = one row's 128 bytes)
  constexpr int WV = NCH * C1_KC / 4 / 256;
  float4 rx[XV], rw[WV];
  auto load_chunk = [&](int m0, int kc) {
#pragma unroll
    for (int i = 0; i < XV; ++i) {
      const int f = tid + 256 * i, px = f >> 3, kq = f & 7;
      const int m = m0 + px;
      rx[i] = m < M ? *reinterpret_cast<const float4*>(X + static_cast<size_t>(m) * x_row_stride + kc * C1_KC + 4 * kq)
                    : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < WV; ++i) {
      const int f = tid + 256 * i, ch = f >> 3, kq = f & 7;
      rw[i] = *reinterpret_cast<const float4*>(Wt + static_cast<size_t>(n0 + ch) * K + kc * C1_KC + 4 * kq);
    }
  };
  auto store_chunk = [&](int buf) {
#pragma unroll
    for (int i = 0; i < XV; ++i) {
      const int f = tid + 256 * i, px = f >> 3, kq = f & 7;
      sX[buf][4 * kq + 0][px] = rx[i].x; sX[buf][4 * kq + 1][px] = rx[i].y;
      sX[buf][4 * kq + 2][px] = rx[i].z; sX[buf][4 * kq + 3][px] = rx[i].w;
    }
#pragma unroll
    for (int i = 0; i < WV; ++i) {
      const int f = tid + 256 * i, ch = f >> 3, kq = f & 7;
      sW[buf][4 * kq + 0][ch] = rw[i].x; sW[buf][4 * kq + 1][ch] = rw[i].y;
      sW[buf][4 * kq + 2][ch] = rw[i].z; sW[buf][4 * kq + 3][ch] = rw[i].w;
    }
  };

  float s1[NT], s2[NT];                                        // this lane's channel (32 t + l31): sum y, sum y^2 over its pixel rows
#pragma unroll
  for (int t = 0; t < NT; ++t) { s1[t] = 0.f; s2[t] = 0.f; }

  int item = L;
  if (item < nitem) load_chunk((item / nsl) * C1_PXB, 0);
  int buf = 0;
  for (; item < nitem; item += G) {
    const int m0 = (item / nsl) * C1_PXB;
    floatx16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    store_chunk(buf);                                          // chunk 0 of this item (its loads were issued one stage earlier)
    __syncthreads();
    for (int c = 0; c < nchunk; ++c) {
      if (c + 1 < nchunk) load_chunk(m0, c + 1);               // in flight during the MFMAs below
      else if (item + G < nitem) load_chunk(((item + G) / nsl) * C1_PXB, 0);   // next item's first chunk: under the last MFMAs + epilogue
#pragma unroll
      for (int kk = 0; kk < C1_KC / 2; ++kk) {
        const int k = 2 * kk + half;
        const float a = sX[buf][k][32 * w + l31];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const float b = sW[buf][k][32 * t + l31];
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
        }
      }
      if (c + 1 < nchunk) {
        store_chunk(buf ^ 1);                                  // the other buffer: its readers passed the previous barrier
        __syncthreads();
      }
      buf ^= 1;
    }
    // epilogue: accumulator register 4q + r of tile t = pixel m0 + 32 w + 8 q + 4 half + r, channel n0 + 32 t + l31
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      float a1 = 0.f, a2 = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = acc[t][4 * q + r];
          const int m = m0 + 32 * w + 8 * q + 4 * half + r;
          if (m < M) Y[static_cast<size_t>(m) * N + n0 + 32 * t + l31] = v;
          if (STATS) { a1 += v; a2 = fmaf(v, v, a2); }          // rows past M hold exact zeros (zero-filled activations)
        }
      }
      if (STATS) { s1[t] += a1; s2[t] += a2; }
    }
    // no barrier here: the next item's store_chunk(buf) goes to the buffer the LAST chunk did not read (its last readers passed an
    // earlier barrier), so a wave may start staging the next item while the others finish this one
  }
  if (STATS) {
    // 2 halves x 4 waves -> one value per channel, fixed order; row g = L / nsl of the slice's partial table
    float* red = smem;                                         // [2][4][NCH]
    __syncthreads();
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const float b1 = s1[t] + __shfl_xor(s1[t], 32), b2 = s2[t] + __shfl_xor(s2[t], 32);
      if (half == 0) {
        red[(0 * 4 + w) * NCH + 32 * t + l31] = b1;
        red[(1 * 4 + w) * NCH + 32 * t + l31] = b2;
      }
    }
    __syncthreads();
    for (int i = tid; i < 2 * NCH; i += 256) {
      const int which = i / NCH, c = i - which * NCH;
      const float v = ((red[(which * 4 + 0) * NCH + c] + red[(which * 4 + 1) * NCH + c]) + red[(which * 4 + 2) * NCH + c]) +
                      red[(which * 4 + 3) * NCH + c];
      partial[(static_cast<size_t>(L / nsl) * 2 + which) * N + n0 + c] = v;
    }
  }
}

int c1_pick_nt(int N) {
  if (N % 32) return 0;
  const int t = N / 32;
  for (int nt : {4, 2, 1})
    if (t % nt == 0) return nt;
  return 0;
}

struct C1Plan { int nt, nsl, grid, rows; };

bool c1_plan(long long M, int K, int N, C1Plan* p) {
  p->nt = c1_pick_nt(N);
  if (M <= 0 || M > 0x3fffffffLL || p->nt == 0 || K <= 0 || (K % C1_KC)) return false;
  p->nsl = N / (32 * p->nt);
  if (p->nsl > 64) return false;
  const long long nitem = (M + C1_PXB - 1) / C1_PXB * p->nsl;
  // persistent: two workgroups per CU; a multiple of 8 XCDs x nsl so that a workgroup keeps one channel slice
  long long g = 2LL * DBEV_NUM_CU;
  const long long unit = static_cast<long long>(DBEV_NUM_XCD) * p->nsl;
  g = g / unit * unit;
  if (g < unit) g = unit;
  const long long need = (nitem + unit - 1) / unit * unit;
  if (need < g) g = need;
  p->grid = static_cast<int>(g);
  p->rows = p->grid / p->nsl;
  return true;
}

}  // namespace

extern "C" int dbev_conv1x1_stats_rows(long long M, int Cin, int Cout) {
  C1Plan p;
  return c1_plan(M, Cin, Cout, &p) ? p.rows : 0;
}

extern "C" int dbev_conv1x1_forward(const float* x_nhwc, const float* weight, float* y_nhwc, float* stats_partial, long long M,
                                    int Cin, int Cout, int x_row_stride, dbevStream_t stream) {
  C1Plan p;
  if (!c1_plan(M, Cin, Cout, &p) || x_nhwc == nullptr || weight == nullptr || y_nhwc == nullptr || x_row_stride < Cin ||
      (x_row_stride & 3))
    return DBEV_EINVAL;
  hipStream_t s = dbev_stream(stream);
  const int m = static_cast<int>(M);
#define C1_LAUNCH(NTV)                                                                                                          \
  do {                                                                                                                          \
    if (stats_partial != nullptr)                                                                                               \
      hipLaunchKernelGGL((c1x1_fwd<NTV, true>), dim3(p.grid), dim3(256), 0, s, x_nhwc, weight, y_nhwc, stats_partial, m, Cin, Cout, \
                         x_row_stride);                                                                                         \
    else                                                                                                                        \
      hipLaunchKernelGGL((c1x1_fwd<NTV, false>), dim3(p.grid), dim3(256), 0, s, x_nhwc, weight, y_nhwc, stats_partial, m, Cin, Cout, \
                         x_row_stride);                                                                                         \
  } while (0)
  DbevKt kt(DBEV_K_CONV1X1_FWD, 4LL * M * (Cin + Cout), s);
  switch (p.nt) {
    case 4: C1_LAUNCH(4); break;
    case 2: C1_LAUNCH(2); break;
    default: C1_LAUNCH(1); break;
  }
#undef C1_LAUNCH
  DBEV_LAUNCH_CHECK();
  return 0;
}
