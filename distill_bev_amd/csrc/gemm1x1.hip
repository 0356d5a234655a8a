// 1x1 convolutions (plain GEMMs in channels-last) of the dense stack on the fp32 matrix cores -- conv1 / conv3 / downsample of the
// ResNet bottlenecks (mmdet3d/models/bricks/res_block.py:102-230), the lateral / output convolutions of the necks
// (necks/fpn.py:10-204, necks/lss_fpn.py:10-72): forward, data gradient (the same kernel on grad_y with the transposed weight) and
// weight gradient, built on what the Winograd kernels' measurements showed (csrc/wino.hip, tools/mfma_filler_bench.hip): on gfx950 a
// VALU instruction issued by a wave is NOT hidden under its v_mfma_f32_32x32x2_f32 (~7 cycles of matrix time each), LDS reads, SALU
// and LDS-DMA issue are.  So the main loops contain no VALU instruction at all:
//   * both operand tiles go global -> LDS by LDS-DMA in the scalar-base form (SGPR address advanced by SALU, ONE constant VGPR offset
//     per operand and wave); the LDS image is the lane order of the DMA, bank conflicts of the row-major [row][32 k] image are removed
//     by permuting the 16-byte quads on the SOURCE side (slot s of row r holds quad s ^ ((r >> 1) & 7): the 16 lanes of every
//     ds_read_b128 service group then hit 16 distinct bank quads);
//   * operands are ds_read_b128 at precomputed offsets (reduction index consumed in the permuted order of conv1x1.hip: step 4 j + e
//     takes k = 8 j + 4 half + e), 64 MFMAs per 32-deep chunk and wave;
//   * two workgroups of 64 KB LDS and <= 128 registers per lane share a CU: one's barrier / epilogue bubbles are the other's matrix time.
// The forward's epilogue can take the per-channel sums of y and y^2 (the BatchNorm statistics of its output, `partial` rows in
// bn_finalize's layout) -- two VALU instructions per output value, a few per cent of a tile's matrix time for K >= 256.
#include "common.h"

#include <stdlib.h>

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

constexpr int G1_KC = 32;                        // reduction chunk
constexpr int G1_TM = 128;                       // rows (pixels) of a workgroup tile

__device__ __forceinline__ unsigned long long g1_uniform64(unsigned long long v) {
  const unsigned lo = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(v)), hi = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(v >> 32));
  return (static_cast<unsigned long long>(hi) << 32) | lo;
}

#define G1_DMA(voff_, sbase_, ldsaddr_)                                                                              \
  do {                                                                                                               \
    unsigned keep_;                                                                                                  \
    const unsigned m0v_ = __builtin_amdgcn_readfirstlane(ldsaddr_);                                                  \
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"  \
                 : "=&s"(keep_) : "v"(voff_), "s"(m0v_), "s"(sbase_) : "memory");                                    \
  } while (0)

// ---- Y[M, N] = X[M, K] * Wt[N, K]^T ---------------------------------------------------------------------------------------------
// X rows of x_stride floats (>= K), Wt row-major [N, K], Y row-major [M, N].  M % 128 == 0, K % 32 == 0, N % (64 NT) == 0.
// Workgroup = 4 waves (wm, wn): tile 128 rows x 64 NT columns, wave tile 64 x 32 NT (2 x NT accumulator tiles of 32 x 32).
// Persistent along M for one column block: workgroup (mb, nb) walks row tiles mb, mb + Gm, ... and keeps the column sums of
// everything it wrote in registers -> partial[mb][2][N].
template <int NT, bool STATS>
__global__ __launch_bounds__(256, 2) void g1_fwd(const float* __restrict__ X, const float* __restrict__ Wt, float* __restrict__ Y,
                                                 float* __restrict__ partial, int M, int K, int N, int xs, int Gm) {
  constexpr int BN = 64 * NT;                                  // columns of the workgroup tile
  constexpr int ABUF = G1_TM * G1_KC, BBUF = BN * G1_KC;       // floats per operand buffer
  __shared__ __attribute__((aligned(16))) float smem[2 * ABUF + 2 * BBUF];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int wu = __builtin_amdgcn_readfirstlane(w);
  const int wm = wu & 1, wn = wu >> 1;
  const int nblk = N / BN;
  const int L = xcd_block();
  const int nb = L % nblk, mb = L / nblk;
  if (mb >= Gm) return;
  const int n0 = nb * BN;
  const int mtiles = M / G1_TM, nchunk = K / G1_KC;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(reinterpret_cast<size_t>((lds_ptr_t)smem)));

  // DMA lanes: piece = 8 rows x 128 bytes; lane l = row (l >> 3) of the piece, LDS slot l & 7 <- source quad (l & 7) ^ ((row >> 1) & 7).
  // Wave w issues pieces w, w + 4, ...: their first rows are 8 w (mod 32), so ((row >> 1) & 7) = (4 w + (l >> 4)) & 7 for all of them.
  const unsigned r8 = lane >> 3, swz = (lane & 7) ^ ((4 * wu + (r8 >> 1)) & 7);
  const unsigned avoff = (r8 * static_cast<unsigned>(xs)) * 4u + swz * 16u;
  const unsigned bvoff = (r8 * static_cast<unsigned>(K)) * 4u + swz * 16u;
  // chunk (row tile mt, k chunk kc) -> sA[buf], sB[buf]
#define G1_LOAD(mt_, kc_, buf_)                                                                                      \
  do {                                                                                                               \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                                  \
      const int piece = wu + 4 * i;                                                                                  \
      const unsigned long long sb_ = g1_uniform64(reinterpret_cast<unsigned long long>(                              \
          X + (static_cast<size_t>(mt_) * G1_TM + 8 * piece) * xs + (kc_) * G1_KC));                                 \
      G1_DMA(avoff, sb_, lds0 + static_cast<unsigned>(((buf_) * ABUF + piece * 256) * 4));                           \
    }                                                                                                                \
    _Pragma("unroll") for (int i = 0; i < 2 * NT; ++i) {                                                             \
      const int piece = wu + 4 * i;                                                                                  \
      const unsigned long long sb_ = g1_uniform64(reinterpret_cast<unsigned long long>(                              \
          Wt + (static_cast<size_t>(n0) + 8 * piece) * K + (kc_) * G1_KC));                                          \
      G1_DMA(bvoff, sb_, lds0 + static_cast<unsigned>((2 * ABUF + (buf_) * BBUF + piece * 256) * 4));                \
    }                                                                                                                \
  } while (0)

  // operand reads: row (64 wm + 32 t + l31) of sA / row (32 NT wn + 32 u + l31) of sB, slot of step group j = (2 j + half) ^ ((l31 >> 1) & 7)
  int slot[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) slot[j] = ((2 * j + half) ^ ((l31 >> 1) & 7)) * 4;
  const int arow = (64 * wm + l31) * G1_KC, brow = (32 * NT * wn + l31) * G1_KC;

  floatx16 acc[2][NT];
  float s1[NT], s2[NT];
#pragma unroll
  for (int u = 0; u < NT; ++u) { s1[u] = 0.f; s2[u] = 0.f; }

  // output offsets of the 16 accumulator registers (rows (r & 3) + 8 (r >> 2) + 4 half, column l31)
  const int colbase = n0 + 32 * NT * wn + l31;

  int mt = mb;
  int buf = 0;
  if (mt < mtiles) G1_LOAD(mt, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (; mt < mtiles; mt += Gm) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int u = 0; u < NT; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;
    for (int kc = 0; kc < nchunk; ++kc) {
      // the next chunk of this tile, or the first chunk of the workgroup's next tile (clamped at the end: a redundant load)
      const bool last = kc + 1 == nchunk;
      const int nmt = last ? (mt + Gm < mtiles ? mt + Gm : mt) : mt, nkc = last ? 0 : kc + 1;
      G1_LOAD(nmt, nkc, buf ^ 1);
      const float* sa = smem + buf * ABUF + arow;
      const float* sb = smem + 2 * ABUF + buf * BBUF + brow;
      float4 a[2][2], b[2][NT];                               // [step-group parity][tile]
#pragma unroll
      for (int t = 0; t < 2; ++t) a[0][t] = *reinterpret_cast<const float4*>(sa + t * 32 * G1_KC + slot[0]);
#pragma unroll
      for (int u = 0; u < NT; ++u) b[0][u] = *reinterpret_cast<const float4*>(sb + u * 32 * G1_KC + slot[0]);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (j < 3) {
#pragma unroll
          for (int t = 0; t < 2; ++t) a[(j + 1) & 1][t] = *reinterpret_cast<const float4*>(sa + t * 32 * G1_KC + slot[j + 1]);
#pragma unroll
          for (int u = 0; u < NT; ++u) b[(j + 1) & 1][u] = *reinterpret_cast<const float4*>(sb + u * 32 * G1_KC + slot[j + 1]);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int u = 0; u < NT; ++u) {
              const float4 av = a[j & 1][t], bv = b[j & 1][u];
              acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(e == 0 ? av.x : e == 1 ? av.y : e == 2 ? av.z : av.w,
                                                               e == 0 ? bv.x : e == 1 ? bv.y : e == 2 ? bv.z : bv.w, acc[t][u], 0, 0, 0);
            }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      buf ^= 1;
    }
    // epilogue: the tile's outputs (and their column sums)
    float* yb = Y + (static_cast<size_t>(mt) * G1_TM + 64 * wm) * N;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int u = 0; u < NT; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = acc[t][u][r];
          yb[static_cast<size_t>(32 * t + (r & 3) + 8 * (r >> 2) + 4 * half) * N + colbase + 32 * u] = v;
          if (STATS) { s1[u] += v; s2[u] = fmaf(v, v, s2[u]); }
        }
  }
#undef G1_LOAD
  if (STATS) {
    float* red = smem;                                         // [2 which][2 wm][BN]
    __syncthreads();
#pragma unroll
    for (int u = 0; u < NT; ++u) {
      const float b1 = s1[u] + __shfl_xor(s1[u], 32), b2 = s2[u] + __shfl_xor(s2[u], 32);
      if (half == 0) {
        red[(0 * 2 + wm) * BN + 32 * NT * wn + 32 * u + l31] = b1;
        red[(1 * 2 + wm) * BN + 32 * NT * wn + 32 * u + l31] = b2;
      }
    }
    __syncthreads();
    for (int i = tid; i < 2 * BN; i += 256) {
      const int which = i / BN, c = i - which * BN;
      partial[(static_cast<size_t>(mb) * 2 + which) * N + n0 + c] = red[(which * 2 + 0) * BN + c] + red[(which * 2 + 1) * BN + c];
    }
  }
}

// ---- GW[N, K] (+)= GY[M, N]^T * X[M, K]  (one share of the rows per workgroup) --------------------------------------------------------
// Reduction over the rows: MFMA rows = output channels n, columns = input channels k, a step = two rows of the operands (lanes 0-31 /
// 32-63).  Operand chunks of 32 rows: [row][64 TN] and [row][64 TK] floats exactly as they lie in memory (LDS-DMA, 2 rows per piece
// at 128 channels); a lane reads its channel of a row with ds_read_b32 (32 consecutive channels per half wave: conflict-free).
// part[split][N][K].
template <int TN, int TK>
__global__ __launch_bounds__(256, 2) void g1_wgrad(const float* __restrict__ GY, const float* __restrict__ X, float* __restrict__ part,
                                                   int M, int K, int N, int xs, int nsplit, int rows_per_split) {
  constexpr int BN = 64 * TN, BK = 64 * TK;
  constexpr int ABUF = G1_KC * BN, BBUF = G1_KC * BK;
  __shared__ __attribute__((aligned(16))) float smem[2 * ABUF + 2 * BBUF];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int wu = __builtin_amdgcn_readfirstlane(w);
  const int wn = wu & 1, wk = wu >> 1;
  const int nbn = N / BN, nbk = K / BK, nblk = nbn * nbk;
  const int L = xcd_block();
  const int blk = L % nblk, split = L / nblk;
  if (split >= nsplit) return;
  const int n0 = (blk % nbn) * BN, k0 = (blk / nbn) * BK;
  const int m_begin = split * rows_per_split, m_end = min(M, m_begin + rows_per_split);
  const int nch = (m_end - m_begin) / G1_KC;                   // rows_per_split and M are multiples of 32
  const unsigned lds0 = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(reinterpret_cast<size_t>((lds_ptr_t)smem)));

  // DMA: a piece (1 KB) = 256 / BN rows of the GY chunk (resp. 256 / BK rows of X); lane l = row l / (BN / 4), quad l % (BN / 4)
  constexpr int APR = 256 / BN, BPR = 256 / BK;                // rows per piece
  const unsigned avoff = ((lane / (BN / 4)) * static_cast<unsigned>(N)) * 4u + (lane % (BN / 4)) * 16u;
  const unsigned bvoff = ((lane / (BK / 4)) * static_cast<unsigned>(xs)) * 4u + (lane % (BK / 4)) * 16u;
#define G1W_LOAD(ch_, buf_)                                                                                          \
  do {                                                                                                               \
    const size_t row0_ = static_cast<size_t>(m_begin) + static_cast<size_t>(ch_) * G1_KC;                            \
    _Pragma("unroll") for (int i = 0; i < G1_KC / APR / 4; ++i) {                                                    \
      const int piece = wu + 4 * i;                                                                                  \
      const unsigned long long sb_ = g1_uniform64(reinterpret_cast<unsigned long long>(GY + (row0_ + piece * APR) * N + n0)); \
      G1_DMA(avoff, sb_, lds0 + static_cast<unsigned>(((buf_) * ABUF + piece * 256) * 4));                           \
    }                                                                                                                \
    _Pragma("unroll") for (int i = 0; i < G1_KC / BPR / 4; ++i) {                                                    \
      const int piece = wu + 4 * i;                                                                                  \
      const unsigned long long sb_ = g1_uniform64(reinterpret_cast<unsigned long long>(X + (row0_ + piece * BPR) * xs + k0)); \
      G1_DMA(bvoff, sb_, lds0 + static_cast<unsigned>((2 * ABUF + (buf_) * BBUF + piece * 256) * 4));                \
    }                                                                                                                \
  } while (0)

  floatx16 acc[TN][TK];
#pragma unroll
  for (int t = 0; t < TN; ++t)
#pragma unroll
    for (int u = 0; u < TK; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;
  const int aoff = half * BN + 32 * TN * wn + l31, boff = half * BK + 32 * TK * wk + l31;

  if (nch > 0) G1W_LOAD(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int buf = 0;
  for (int ch = 0; ch < nch; ++ch) {
    G1W_LOAD(ch + 1 < nch ? ch + 1 : ch, buf ^ 1);
    const float* sa = smem + buf * ABUF + aoff;
    const float* sb = smem + 2 * ABUF + buf * BBUF + boff;
    float a[2][TN], b[2][TK];
#pragma unroll
    for (int t = 0; t < TN; ++t) a[0][t] = sa[32 * t];
#pragma unroll
    for (int u = 0; u < TK; ++u) b[0][u] = sb[32 * u];
#pragma unroll
    for (int s = 0; s < G1_KC / 2; ++s) {
      if (s + 1 < G1_KC / 2) {
#pragma unroll
        for (int t = 0; t < TN; ++t) a[(s + 1) & 1][t] = sa[(2 * (s + 1)) * BN + 32 * t];
#pragma unroll
        for (int u = 0; u < TK; ++u) b[(s + 1) & 1][u] = sb[(2 * (s + 1)) * BK + 32 * u];
      }
#pragma unroll
      for (int t = 0; t < TN; ++t)
#pragma unroll
        for (int u = 0; u < TK; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s & 1][t], b[s & 1][u], acc[t][u], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    buf ^= 1;
  }
#undef G1W_LOAD
  float* out = part + (static_cast<size_t>(split) * N + n0 + 32 * TN * wn) * K + k0 + 32 * TK * wk + l31;
#pragma unroll
  for (int t = 0; t < TN; ++t)
#pragma unroll
    for (int u = 0; u < TK; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        out[static_cast<size_t>(32 * t + (r & 3) + 8 * (r >> 2) + 4 * half) * K + 32 * u] = acc[t][u][r];
}

// part[nsplit][count] -> out[count]: shares added in a fixed order (four interleaved chains, then pairwise)
__global__ __launch_bounds__(256) void g1_sum(const float* __restrict__ part, float* __restrict__ out, int nsplit, long long count) {
  const long long idx = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x;
  if (idx >= count) return;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  const float* ps = part + idx;
  int s = 0;
  for (; s + 4 <= nsplit; s += 4) {
    s0 += ps[(s + 0) * count];
    s1 += ps[(s + 1) * count];
    s2 += ps[(s + 2) * count];
    s3 += ps[(s + 3) * count];
  }
  for (; s < nsplit; ++s) s0 += ps[s * count];
  out[idx] = (s0 + s1) + (s2 + s3);
}

struct G1Plan { int nt, nblk, gm, grid; };

bool g1_plan(long long M, int K, int N, int xs, G1Plan* p) {
  if (M <= 0 || M > 0x3fffffffLL || (M % G1_TM) || K <= 0 || (K % G1_KC) || N <= 0 || (N % 64) || xs < K || (xs & 3)) return false;
  if (static_cast<long long>(8) * xs * 4 + 128 >= 0x7fffffffLL) return false;
  p->nt = (N % 128) == 0 ? 2 : 1;
  p->nblk = N / (64 * p->nt);
  const long long mtiles = M / G1_TM;
  // Gm workgroups walk the row tiles of one column block.  One tile each (the dispatcher balances; two workgroups share a CU, so
  // the turnover is covered) unless that would leave more than ~1000 partial statistics rows: then an even split of the tiles
  long long gm = mtiles;
  if (gm > 1024) {
    const long long per = (mtiles + 1023) / 1024;
    gm = (mtiles + per - 1) / per;
  }
  p->gm = static_cast<int>(gm);
  p->grid = dbev_round_xcd(p->gm * p->nblk);
  return true;
}

struct G1WPlan { int tn, tk, nblk, nsplit, rows, grid; };

bool g1w_plan(long long M, int K, int N, int xs, G1WPlan* p) {
  if (M <= 0 || M > 0x3fffffffLL || (M % G1_KC) || K <= 0 || (K % 64) || N <= 0 || (N % 64) || xs < K || (xs & 3)) return false;
  p->tn = (N % 128) == 0 ? 2 : 1;
  p->tk = (K % 128) == 0 ? 2 : 1;
  p->nblk = (N / (64 * p->tn)) * (K / (64 * p->tk));
  long long ns = (2LL * DBEV_NUM_CU) / p->nblk;              // two workgroups per CU, one round
  if (ns < 1) ns = 1;
  const long long chunks = M / G1_KC;
  if (ns > chunks) ns = chunks;
  const long long per = (chunks + ns - 1) / ns;
  p->rows = static_cast<int>(per * G1_KC);
  p->nsplit = static_cast<int>((chunks + per - 1) / per);
  p->grid = dbev_round_xcd(p->nblk * p->nsplit);
  return true;
}

}  // namespace

extern "C" int dbev_gemm1x1_stats_rows(long long M, int Cin, int Cout, int x_row_stride) {
  G1Plan p;
  return g1_plan(M, Cin, Cout, x_row_stride, &p) ? p.gm : 0;
}

extern "C" int dbev_gemm1x1_forward(const float* x_nhwc, const float* weight, float* y_nhwc, float* stats_partial, long long M, int Cin,
                                    int Cout, int x_row_stride, dbevStream_t stream) {
  G1Plan p;
  if (!g1_plan(M, Cin, Cout, x_row_stride, &p) || x_nhwc == nullptr || weight == nullptr || y_nhwc == nullptr) return DBEV_EINVAL;
  hipStream_t s = dbev_stream(stream);
  const int m = static_cast<int>(M);
  DbevKt kt(DBEV_K_GEMM1X1_FWD, 2LL * M * Cin * Cout, s);      // the log's work field: FLOPs
#define G1_GO(NTV, ST)                                                                                                            \
  hipLaunchKernelGGL((g1_fwd<NTV, ST>), dim3(p.grid), dim3(256), 0, s, x_nhwc, weight, y_nhwc, stats_partial, m, Cin, Cout,       \
                     x_row_stride, p.gm)
  if (p.nt == 2) { if (stats_partial != nullptr) G1_GO(2, true); else G1_GO(2, false); }
  else { if (stats_partial != nullptr) G1_GO(1, true); else G1_GO(1, false); }
#undef G1_GO
  DBEV_LAUNCH_CHECK();
  return 0;
}

extern "C" size_t dbev_gemm1x1_backward_weight_workspace_bytes(long long M, int Cin, int Cout, int x_row_stride) {
  G1WPlan p;
  if (!g1w_plan(M, Cin, Cout, x_row_stride, &p)) return 0;
  return static_cast<size_t>(p.nsplit) * Cin * Cout * sizeof(float);
}

extern "C" int dbev_gemm1x1_backward_weight(const float* x_nhwc, const float* grad_y_nhwc, float* grad_weight, long long M, int Cin,
                                            int Cout, int x_row_stride, void* workspace, size_t workspace_bytes, dbevStream_t stream) {
  G1WPlan p;
  if (!g1w_plan(M, Cin, Cout, x_row_stride, &p) || x_nhwc == nullptr || grad_y_nhwc == nullptr || grad_weight == nullptr ||
      workspace == nullptr || workspace_bytes < static_cast<size_t>(p.nsplit) * Cin * Cout * sizeof(float))
    return DBEV_EINVAL;
  hipStream_t s = dbev_stream(stream);
  float* part = static_cast<float*>(workspace);
  const int m = static_cast<int>(M);
  DbevKt kt(DBEV_K_GEMM1X1_WGRAD, 2LL * M * Cin * Cout, s);
#define G1W_GO(TNV, TKV)                                                                                                          \
  hipLaunchKernelGGL((g1_wgrad<TNV, TKV>), dim3(p.grid), dim3(256), 0, s, grad_y_nhwc, x_nhwc, part, m, Cin, Cout, x_row_stride,  \
                     p.nsplit, p.rows)
  if (p.tn == 2) { if (p.tk == 2) G1W_GO(2, 2); else G1W_GO(2, 1); }
  else { if (p.tk == 2) G1W_GO(1, 2); else G1W_GO(1, 1); }
#undef G1W_GO
  DBEV_LAUNCH_CHECK();
  const long long count = static_cast<long long>(Cin) * Cout;
  hipLaunchKernelGGL(g1_sum, dim3(dbev_ceil_div(count, 256)), dim3(256), 0, s, part, grad_weight, p.nsplit, count);
  DBEV_LAUNCH_CHECK();
  return 0;
}
