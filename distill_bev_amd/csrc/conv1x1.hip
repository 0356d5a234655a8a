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
// accumulators directly as 128-byte row segments (32 consecutive channels of a pixel per half-wave store).
// Operand staging: both tiles sit in LDS ROW-major ([pixel][k], [channel][k], row stride KC + 4 floats) exactly as the 16-byte global
// loads deliver them (one ds_write_b128 per load, no transposing scalar writes).  The reduction index is consumed in a permuted
// order -- MFMA step 4j + e takes k = 8j + 4*half + e -- so that the four steps' operand values of a lane are ONE ds_read_b128
// (the stride's 36 = 4 * 9 puts the 16 lanes of every b128 service group on 16 distinct bank quads: conflict-free); A and B use
// the same permutation, the sum is unchanged.  K is walked in chunks of 32, double-buffered; the global loads of chunk c+1 -- and
// of the next work item's first chunk during the epilogue -- are in flight while chunk c is multiplied.  With WRES (K <= 64) the
// workgroup's weight slice stays resident in LDS for its whole life and only activations are staged.  Work items are ordered
// (pixel tile, channel slice) with the slice fastest and XCD-aware, so the slices of a tile re-read its activation rows from ONE
// XCD's L2, back to back.
#include "common.h"

#include <stdlib.h>

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int C1_PXB = 128;           // pixels per work item (4 waves x 32)
constexpr int C1_KC = 32;             // K chunk
constexpr int C1_STR = C1_KC + 4;     // LDS row stride (floats) of a chunk tile

// V float4 per thread = a [32 * V rows][32 floats] chunk tile: 8 consecutive lanes take one row's 128 bytes
template <int V>
__device__ __forceinline__ void c1_load_rows(float4 (&r)[V], const float* __restrict__ base, int row0, int row_end, int row_stride,
                                             int kc, int tid) {
#pragma unroll
  for (int i = 0; i < V; ++i) {
    const int f = tid + 256 * i, row = row0 + (f >> 3), kq = f & 7;
    r[i] = row < row_end ? *reinterpret_cast<const float4*>(base + static_cast<size_t>(row) * row_stride + kc * C1_KC + 4 * kq)
                         : make_float4(0.f, 0.f, 0.f, 0.f);
  }
}
template <int V>
__device__ __forceinline__ void c1_store_rows(const float4 (&r)[V], float* __restrict__ tile, int tid) {
#pragma unroll
  for (int i = 0; i < V; ++i) {
    const int f = tid + 256 * i;
    *reinterpret_cast<float4*>(tile + (f >> 3) * C1_STR + 4 * (f & 7)) = r[i];
  }
}

template <int NT, bool STATS, bool WRES>
__global__ __launch_bounds__(256, 2) void c1x1_fwd(const float* __restrict__ X, const float* __restrict__ Wt, float* __restrict__ Y,
                                                   float* __restrict__ partial, int M, int K, int N, int x_row_stride, int dbg) {
  constexpr int NCH = NT * 32;
  constexpr int WBUF = WRES ? 2 : 2;                          // WRES: the two K chunks of the resident slice (K <= 64)
  __shared__ __attribute__((aligned(16))) float smem[2 * C1_PXB * C1_STR + WBUF * NCH * C1_STR];
  float (*sX)[C1_PXB][C1_STR] = reinterpret_cast<float (*)[C1_PXB][C1_STR]>(smem);
  float (*sW)[NCH][C1_STR] = reinterpret_cast<float (*)[NCH][C1_STR]>(smem + 2 * C1_PXB * C1_STR);
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
#define load_x(m0_, kc_) c1_load_rows<XV>(rx, X, (m0_), M, x_row_stride, (kc_), tid)
#define load_w(kc_) c1_load_rows<WV>(rw, Wt, n0, N, K, (kc_), tid)
#define store_x(buf_) c1_store_rows<XV>(rx, &sX[(buf_)][0][0], tid)
#define store_w(buf_) c1_store_rows<WV>(rw, &sW[(buf_)][0][0], tid)

  float s1[NT], s2[NT];                                        // this lane's channel (32 t + l31): sum y, sum y^2 over its pixel rows
#pragma unroll
  for (int t = 0; t < NT; ++t) { s1[t] = 0.f; s2[t] = 0.f; }

  if (WRES) {                                                  // the whole weight slice, once (K <= 64: at most two chunks)
    load_w(0); store_w(0);
    if (nchunk > 1) { load_w(1); store_w(1); }
  }
  int item = L;
  int buf = 0;
  if (item < nitem) {                                          // chunk 0 of the first item
    load_x((item / nsl) * C1_PXB, 0);
    if (!WRES) load_w(0);
    store_x(0);
    if (!WRES) store_w(0);
  }
  // Software pipeline over the work items: the OUTPUT of item i (16 * NT values per lane) is stored -- and its statistics taken --
  // under the MFMAs of item i+1's first chunk, ONE value behind each MFMA (a store, two statistics FMAs and an address add fit in
  // the 64-cycle shadow of a v_mfma_f32_32x32x2; a burst of them between MFMA groups left the matrix pipe idle a fifth of the
  // time).  Two accumulator sets alternate (item loop unrolled by two): no copy between them.
  // register 4q + r of tile t = pixel row0 + 8 q + r (row0 = m0 + 32 w + 4 half), channel n0 + 32 t + l31
#define C1_EMIT1(PRV, pm0_, full_, q_, t_, r_)                                                                 \
  do {                                                                                                         \
    const float v_ = PRV[t_][4 * (q_) + (r_)];                                                                 \
    const int m_ = (pm0_) + 32 * w + 8 * (q_) + 4 * half + (r_);                                               \
    if (((full_) || m_ < M) && !(dbg & 1)) Y[static_cast<size_t>(m_) * N + n0 + 32 * (t_) + l31] = v_;         \
    if (STATS) { s1[t_] += v_; s2[t_] = fmaf(v_, v_, s2[t_]); }   /* rows past M hold exact zeros */            \
  } while (0)
#define C1_STEP(ACC, PRV, comp, r_)                                                                            \
  _Pragma("unroll") for (int t = 0; t < NT; ++t) {                                                             \
    ACC[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.comp, bv[t].comp, ACC[t], 0, 0, 0);                       \
    if (emit) C1_EMIT1(PRV, prev_m0, prev_full, j, t, r_);                                                     \
  }
#define C1_ITEM(ACC, PRV, item_)                                                                               \
  do {                                                                                                         \
    const int it_ = (item_);                                                                                   \
    const int m0 = (it_ / nsl) * C1_PXB;                                                                       \
    _Pragma("unroll") for (int t = 0; t < NT; ++t)                                                             \
      _Pragma("unroll") for (int r = 0; r < 16; ++r) ACC[t][r] = 0.f;                                          \
    __syncthreads();               /* chunk 0 of this item is in sX[buf] (staged one stage earlier) */          \
    for (int c = 0; c < nchunk; ++c) {                                                                         \
      /* prefetch (in flight during the MFMAs below): the next chunk of this item, or the first chunk of the next item */ \
      const bool last = c + 1 >= nchunk;                                                                       \
      const bool pf = !last || it_ + G < nitem;                                                                \
      const int pm0 = last ? ((it_ + G) / nsl) * C1_PXB : m0, pkc = last ? 0 : c + 1;                          \
      if (pf && !(dbg & 2)) {                                                                                  \
        load_x(pm0, pkc);                                                                                      \
        if (!WRES) load_w(pkc);                                                                                \
      }                                                                                                        \
      const int wb = WRES ? c : buf;                                                                           \
      const bool emit = c == 0 && prev_m0 >= 0 && !(dbg & 16);                                                 \
      const float* xrow = &sX[buf][32 * w + l31][4 * half];                                                    \
      const float* wrow = &sW[wb][l31][4 * half];                                                              \
      if (!(dbg & 4)) {                                                                                        \
        _Pragma("unroll") for (int j = 0; j < C1_KC / 8; ++j) {                                                \
          /* the prefetched global chunk goes to the other LDS buffer before the LAST step group: its write latency and the  \
             barrier that follows sit under that group's MFMAs */                                               \
          if (j + 1 == C1_KC / 8 && pf && !(dbg & 8)) {                                                        \
            store_x(buf ^ 1);      /* the other buffer: its last readers passed an earlier barrier */          \
            if (!WRES) store_w(buf ^ 1);                                                                       \
          }                                                                                                    \
          const float4 av = *reinterpret_cast<const float4*>(xrow + 8 * j);                                    \
          float4 bv[NT];                                                                                       \
          _Pragma("unroll") for (int t = 0; t < NT; ++t) bv[t] = *reinterpret_cast<const float4*>(wrow + 32 * t * C1_STR + 8 * j); \
          C1_STEP(ACC, PRV, x, 0)                                                                              \
          C1_STEP(ACC, PRV, y, 1)                                                                              \
          C1_STEP(ACC, PRV, z, 2)                                                                              \
          C1_STEP(ACC, PRV, w, 3)                                                                              \
        }                                                                                                      \
      } else if (pf) {                                                                                         \
        store_x(buf ^ 1);                                                                                      \
        if (!WRES) store_w(buf ^ 1);                                                                           \
      }                                                                                                        \
      if (!last) __syncthreads();                                                                              \
      buf ^= 1;                                                                                                \
    }                                                                                                          \
    prev_m0 = m0;                                                                                              \
    prev_full = m0 + C1_PXB <= M;                                                                              \
  } while (0)

  floatx16 accA[NT], accB[NT];
  int prev_m0 = -1;                                            // < 0: nothing pending
  bool prev_full = false, last_in_a = false;
  for (; item < nitem; item += 2 * G) {
    C1_ITEM(accA, accB, item);
    last_in_a = true;
    if (item + G < nitem) {
      C1_ITEM(accB, accA, item + G);
      last_in_a = false;
    }
  }
  if (prev_m0 >= 0) {                                          // the last item's output
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (last_in_a) C1_EMIT1(accA, prev_m0, prev_full, q, t, r); else C1_EMIT1(accB, prev_m0, prev_full, q, t, r);
        }
  }
#undef C1_ITEM
#undef C1_STEP
#undef C1_EMIT1
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

int c1_dbg() { static const int v = getenv("DBEV_C1_DBG") ? atoi(getenv("DBEV_C1_DBG")) : 0; return v; }

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

#undef load_x
#undef load_w
#undef store_x
#undef store_w

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
  const bool wres = Cin <= 2 * C1_KC;       // the weight slice fits the two chunk buffers: staged once per workgroup
#define C1_GO(NTV, ST, WR)                                                                                                      \
  hipLaunchKernelGGL((c1x1_fwd<NTV, ST, WR>), dim3(p.grid), dim3(256), 0, s, x_nhwc, weight, y_nhwc, stats_partial, m, Cin, Cout, \
                     x_row_stride, c1_dbg())
#define C1_LAUNCH(NTV)                                                                                                          \
  do {                                                                                                                          \
    if (stats_partial != nullptr) { if (wres) C1_GO(NTV, true, true); else C1_GO(NTV, true, false); }                           \
    else { if (wres) C1_GO(NTV, false, true); else C1_GO(NTV, false, false); }                                                  \
  } while (0)
  DbevKt kt(DBEV_K_CONV1X1_FWD, 4LL * M * (Cin + Cout), s);
  switch (p.nt) {
    case 4: C1_LAUNCH(4); break;
    case 2: C1_LAUNCH(2); break;
    default: C1_LAUNCH(1); break;
  }
#undef C1_GO
#undef C1_LAUNCH
  DBEV_LAUNCH_CHECK();
  return 0;
}
