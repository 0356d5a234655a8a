// The ResNet stem convolution of the image backbone -- nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False) in front of
// norm1 / relu / maxpool (mmdet ResNet._make_stem_layer / ResNet.forward; reached from mmdet3d/models/detectors/bevdet.py:
// image_encoder) -- on the fp32 matrix cores (v_mfma_f32_32x32x2_f32), forward and weight gradient.  The image has no gradient.
//
// Why a kernel of its own: K = 7 * 7 * 3 = 147 and a 3-channel NHWC input (12 bytes per pixel) fit no tiling of a general implicit
// GEMM well -- the library runs the two forward calls of a step at 68 TFLOP/s (0.60 ms each for 48 x 256 x 704 images) and the weight
// gradient at 59 (0.69 ms), and a statistics pass over the 554 MB output follows each forward.  Here
//   * the patch of an output pixel is 7 runs of 21 CONTIGUOUS floats (7 pixels x 3 channels) of the staged input rows, and the
//     weight's channels-last memory [64][7][7][3] has the same (ky, kx, c) order: a K index is `ky * ROW + r` in LDS, a compile-time
//     immediate of the fully unrolled K loop.  The main loops hold MFMAs and ds_read_b32 only -- no VALU (docs/design/03_kernels.md:
//     a VALU instruction beside an fp32 MFMA is never hidden);
//   * the forward's epilogue leaves the per-channel sums of y and y^2 of its pixels (one partial row per workgroup, bn_finalize's
//     layout), so the statistics pass of norm1 does not run;
//   * the weight gradient is a persistent kernel: a workgroup keeps its 64 x 160 accumulator tile in registers over all its pixel
//     tiles and writes ONE partial; a fixed-order reduction sums the partials (bit-reproducible, no atomics).
// Products are exact fp32, accumulation is fp32 in an order of its own (as every matrix-core kernel, the library's included).
#include "common.h"

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx2 __attribute__((ext_vector_type(2)));

constexpr int SK = 147;                      // 7 * 7 * 3
constexpr int SKK = 74;                      // MFMA K steps (K padded to 148)
constexpr int SKP = 2 * SKK;
constexpr int S_TW = 32;                     // output pixels of a tile row = the M of one MFMA
constexpr int S_ROWF = (2 * S_TW + 5) * 3;   // 207 floats: the input pixels under one tile row, 3 channels each
constexpr int SF_TH = 8, SF_ROWS = 2 * SF_TH + 5;   // forward tile: 8 output rows <- 21 input rows
constexpr int SG_TH = 4, SG_ROWS = 2 * SG_TH + 5;   // weight-gradient tile: 4 output rows <- 13 input rows
constexpr int SG_KB = 5, SG_NP = SG_KB * 32;        // 5 column blocks of the gradient tile (160 >= 148)

__device__ __forceinline__ floatx16 mfma2(float a, float b, floatx16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }

// wpk[kk][co][h] = w[co][2 kk + h] (0 for the pad column): the B operand of K step kk, conflict-free for the 64 lanes of a read
__global__ __launch_bounds__(256) void stem_pack(const float* __restrict__ w, float* __restrict__ wpk) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= SKK * 128) return;
  const int kk = i >> 7, co = (i >> 1) & 63, h = i & 1;
  const int k = 2 * kk + h;
  wpk[i] = k < SK ? w[co * SK + k] : 0.f;
}

// rows [iy0, iy0 + ROWS) x floats [if0, if0 + S_ROWF) of image n (zeros outside the image) -> registers, thread = one float column;
// issued a tile AHEAD of its use (the loads fly under the MFMAs of the current tile), written to LDS by STEM_COMMIT_ROWS.
// (macros: the same code as helper templates taking the array by reference left the array in scratch memory)
#define STEM_FETCH_ROWS(ROWS, xn, pre, iy0, if0)                                                   \
  {                                                                                               \
    const int f_ = (if0) + tid;                                                                   \
    const bool fok_ = tid < S_ROWF && f_ >= 0 && f_ < W3;                                         \
    _Pragma("unroll") for (int r_ = 0; r_ < ROWS; ++r_) {                                         \
      const int iy_ = (iy0) + r_;                                                                 \
      const bool ok_ = fok_ && iy_ >= 0 && iy_ < H;                                               \
      const float v_ = (xn)[ok_ ? static_cast<size_t>(iy_) * W3 + f_ : 0];                       \
      pre[r_] = ok_ ? v_ : 0.f;                                                                   \
    }                                                                                             \
  }
#define STEM_COMMIT_ROWS(ROWS, dst, pre)                                                          \
  if (tid < S_ROWF) {                                                                             \
    _Pragma("unroll") for (int r_ = 0; r_ < ROWS; ++r_)(dst)[r_ * S_ROWF + tid] = pre[r_];        \
  }
// workgroup barrier that orders LDS traffic only: __syncthreads() also drains the wave's global stores (vmcnt(0)), i.e. puts the
// write latency of a tile's output between two tiles
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ unsigned long long uniform64(unsigned long long v) {
  const unsigned lo = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(v)), hi = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(v >> 32));
  return (static_cast<unsigned long long>(hi) << 32) | lo;
}
// The forward's input rows go global -> LDS by LDS-DMA (global_load_lds_dword: lane l of the wave writes the dword at M0 + 4 l), a
// tile ahead, with no registers in between: as register prefetch the compiler parked every loaded value in an AGPR at once, i.e.
// waited for each of the 21 loads in turn.  ROWS rows x 207 float columns (thread = column) of image rows iy0.., float columns if0..;
// what lies outside the image is written as zeros by ds_write (a row: uniform branch; border columns: the masked lanes).
// `ldsdst`: LDS byte address of the buffer's first float.  The caller waits (vmcnt(0)) before the barrier that publishes the tile.
template <int ROWS>
__device__ __forceinline__ void dma_rows(const float* __restrict__ xn, float* dst, unsigned ldsdst, int iy0, int if0, int H, int W3, int tid) {
  const int wv = tid >> 6, lane = tid & 63;
  const int f = if0 + tid;
  const bool col = tid < S_ROWF, fok = col && f >= 0 && f < W3;
  const unsigned long long mask = __ballot(fok);
  const unsigned voff = lane * 4;
  const unsigned ldsw = __builtin_amdgcn_readfirstlane(ldsdst + wv * 256);
  const bool border = __any(col && !fok);
#pragma unroll
  for (int r = 0; r < ROWS; ++r) {
    const int iy = iy0 + r;
    if (iy >= 0 && iy < H) {
      if (mask != 0) {
        const unsigned long long sb = uniform64(reinterpret_cast<unsigned long long>(xn + static_cast<long long>(iy) * W3 + if0 + wv * 64));
        const unsigned la = ldsw + r * (S_ROWF * 4);
        unsigned keep;
        unsigned long long ex;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_mov_b64 %1, exec\n\ts_mov_b64 exec, %5\n\t"
                     "global_load_lds_dword %2, %4\n\ts_mov_b64 exec, %1\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep), "=&s"(ex) : "v"(voff), "s"(la), "s"(sb), "s"(mask) : "memory");
      }
      if (border && col && !fok) dst[r * S_ROWF + tid] = 0.f;
    } else if (col) {
      dst[r * S_ROWF + tid] = 0.f;
    }
  }
}

// workgroup = (image n, band of 8 output rows, x split): walks the 32-pixel tiles of its band; wave w: output rows 2 w, 2 w + 1 of the
// band x 64 channels = 2 x 2 MFMA tiles
template <bool STATS>
__global__ __launch_bounds__(256, 2) void stem_fwd(const float* __restrict__ x, const float* __restrict__ wpk, float* __restrict__ z,
                                                   float* __restrict__ part, int H, int W, int Ho, int Wo, int tilesY, int tilesX, int XS) {
  constexpr int SIN = (SF_ROWS + 1) * S_ROWF;
  __shared__ float sW[SKK * 128];
  __shared__ float sIn2[2 * SIN];                                    // two tiles' input rows: tile t + 1 lands while t is multiplied
  __shared__ float sSt[4][2][64];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, hi = lane >> 5, l32 = lane & 31;
  int b = blockIdx.x;
  const int xs = b % XS;
  b /= XS;
  const int ty = b % tilesY, n = b / tilesY;
  for (int i = tid; i < SKK * 128 / 4; i += 256) reinterpret_cast<float4*>(sW)[i] = reinterpret_cast<const float4*>(wpk)[i];
  if (tid < S_ROWF) {                                       // spare rows: what the (zero-weight) pad column reads must be finite
    sIn2[SF_ROWS * S_ROWF + tid] = 0.f;
    sIn2[SIN + SF_ROWS * S_ROWF + tid] = 0.f;
  }
  const int oy0 = ty * SF_TH, W3 = W * 3;
  const float* xn = x + static_cast<size_t>(n) * H * W3;
  float st_s[2] = {0.f, 0.f}, st_q[2] = {0.f, 0.f};
  floatx2 st_s2[2] = {{0.f, 0.f}, {0.f, 0.f}}, st_q2[2] = {{0.f, 0.f}, {0.f, 0.f}};
  const float* bp = sW + l32 * 2 + hi;
  typedef __attribute__((address_space(3))) unsigned char* lds_p;
  const unsigned ldsIn = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(reinterpret_cast<size_t>((lds_p)sIn2)));
  dma_rows<SF_ROWS>(xn, sIn2, ldsIn, 2 * oy0 - 3, (2 * xs * S_TW - 3) * 3, H, W3, tid);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  lds_barrier();
  int cur = 0;
  for (int tx = xs; tx < tilesX; tx += XS) {
    const int ox0 = tx * S_TW;
    const bool more = tx + XS < tilesX;
    if (more) dma_rows<SF_ROWS>(xn, sIn2 + (cur ^ 1) * SIN, ldsIn + (cur ^ 1) * (SIN * 4), 2 * oy0 - 3, (2 * (tx + XS) * S_TW - 3) * 3, H, W3, tid);
    const float* sIn = sIn2 + cur * SIN;
    const float* aN = sIn + (4 * wv) * S_ROWF + 6 * l32 + hi;          // K pair inside one patch row: the upper half-wave reads r + 1
    const float* aS = sIn + (4 * wv) * S_ROWF + 6 * l32 + hi * (S_ROWF - 20);   // pair (r = 20, next row's r = 0)
    floatx16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc[0][0][i] = 0.f; acc[0][1][i] = 0.f; acc[1][0][i] = 0.f; acc[1][1][i] = 0.f; }
#pragma unroll
    for (int kk = 0; kk < SKK; ++kk) {
      const int k0 = 2 * kk, ky = k0 / 21, r = k0 - 21 * ky;
      const int off = ky * S_ROWF + r;
      const float* ap = (r == 20 && kk != SKK - 1) ? aS : aN;
      const float a0 = ap[off], a1 = ap[off + 2 * S_ROWF];
      const float b0 = bp[kk * 128], b1 = bp[kk * 128 + 64];
      acc[0][0] = mfma2(a0, b0, acc[0][0]);
      acc[0][1] = mfma2(a0, b1, acc[0][1]);
      acc[1][0] = mfma2(a1, b0, acc[1][0]);
      acc[1][1] = mfma2(a1, b1, acc[1][1]);
    }
    // the next tile's rows have landed long ago; waited for BEFORE this tile's stores are issued: gfx9 counts loads and stores in one
    // vmcnt, so a wait behind 64 fresh stores is a wait for the stores' write latency
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      const int oy = oy0 + 2 * wv + rb;
      if (oy >= Ho) continue;
      float* zr = z + (static_cast<size_t>(n) * Ho + oy) * Wo * 64 + l32;
      if (STATS && ox0 + S_TW <= Wo) {                      // whole tile row inside the map: the sums as packed pairs (v_pk_add / v_pk_fma)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
          for (int i = 0; i < 16; i += 2) {
            const floatx2 v2 = {acc[rb][cb][i], acc[rb][cb][i + 1]};
            st_s2[cb] += v2;
            st_q2[cb] = __builtin_elementwise_fma(v2, v2, st_q2[cb]);
          }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int ox = ox0 + (i >> 2) * 8 + hi * 4 + (i & 3);
          zr[static_cast<size_t>(ox) * 64] = acc[rb][0][i];
          zr[static_cast<size_t>(ox) * 64 + 32] = acc[rb][1][i];
        }
        continue;
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int ox = ox0 + (i >> 2) * 8 + hi * 4 + (i & 3);
        if (ox < Wo) {
#pragma unroll
          for (int cb = 0; cb < 2; ++cb) {
            const float v = acc[rb][cb][i];
            zr[static_cast<size_t>(ox) * 64 + cb * 32] = v;
            if (STATS) { st_s[cb] += v; st_q[cb] = fmaf(v, v, st_q[cb]); }
          }
        }
      }
    }
    lds_barrier();                                            // tile t + 1 is in LDS, and nobody reads tile t any more
    cur ^= 1;
  }
  if (STATS) {
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      const float ls = st_s[cb] + (st_s2[cb][0] + st_s2[cb][1]), lq = st_q[cb] + (st_q2[cb][0] + st_q2[cb][1]);
      const float s = ls + __shfl_xor(ls, 32), q = lq + __shfl_xor(lq, 32);
      if (hi == 0) { sSt[wv][0][cb * 32 + l32] = s; sSt[wv][1][cb * 32 + l32] = q; }
    }
    __syncthreads();
    if (tid < 128) {
      const int j = tid >> 6, c = tid & 63;
      part[(static_cast<size_t>(blockIdx.x) * 2 + j) * 64 + c] = ((sSt[0][j][c] + sSt[1][j][c]) + sSt[2][j][c]) + sSt[3][j][c];
    }
  }
}

// dW[co][k] = sum over pixels gy[p][co] * patch[p][k].  Persistent workgroups over tiles of 4 output rows x 32 pixels; wave w: row w of
// the tile = 16 pixel pairs = 16 K steps of 2 x 5 MFMAs (A = gy^T, B = the patches); partial f32[gridDim.x][64][148]
__global__ __launch_bounds__(256, 2) void stem_wgrad(const float* __restrict__ x, const float* __restrict__ gy, float* __restrict__ partial,
                                                     int H, int W, int Ho, int Wo, int tilesY, int tilesX, int T) {
  __shared__ float smem[(SG_ROWS + 1) * S_ROWF + 128 * 64];          // input rows | gy tile [pair 64][co 64][h 2]; reused for the merge
  float* sIn = smem;
  float* sG = smem + (SG_ROWS + 1) * S_ROWF;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, hi = lane >> 5, l32 = lane & 31;
  const int W3 = W * 3;
  if (tid < S_ROWF) sIn[SG_ROWS * S_ROWF + tid] = 0.f;
  int boff[SG_KB];                                                   // this lane's patch column k = kb * 32 + l32 as an LDS offset
#pragma unroll
  for (int kb = 0; kb < SG_KB; ++kb) {
    int k = kb * 32 + l32;
    k = k < SK ? k : SK - 1;                                         // columns past 147 are computed on valid data and dropped
    const int ky = k / 21;
    boff[kb] = (2 * wv) * S_ROWF + ky * S_ROWF + (k - 21 * ky) + 6 * hi;
  }
  const float* ga = sG + (wv * 16 * 64 + l32) * 2 + hi;
  floatx16 acc[2][SG_KB];
#pragma unroll
  for (int cb = 0; cb < 2; ++cb)
#pragma unroll
    for (int kb = 0; kb < SG_KB; ++kb)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[cb][kb][i] = 0.f;
  float pre[SG_ROWS];
  float4 pg[8];
  auto fetch = [&](int t) {                                          // input rows + 128 pixels x 16 float4 of gy (zeros past the edges)
    const int tx = t % tilesX;
    const int r2 = t / tilesX;
    const int ty = r2 % tilesY, n = r2 / tilesY;
    const int oy0 = ty * SG_TH, ox0 = tx * S_TW;
    const float* xn = x + static_cast<size_t>(n) * H * W3;
    STEM_FETCH_ROWS(SG_ROWS, xn, pre, 2 * oy0 - 3, (2 * ox0 - 3) * 3);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int i = tid + 256 * j;
      const int px = i >> 4, q = i & 15;
      const int oy = oy0 + (px >> 5), ox = ox0 + (px & 31);
      pg[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (oy < Ho && ox < Wo) pg[j] = reinterpret_cast<const float4*>(gy)[((static_cast<size_t>(n) * Ho + oy) * Wo + ox) * 16 + q];
    }
  };
  for (int t = blockIdx.x; t < T; t += gridDim.x) {
    if (t == static_cast<int>(blockIdx.x)) {                           // first tile: nothing was fetched ahead
      fetch(t);
    }
    lds_barrier();                                                   // every wave is done with the previous tile's buffers
    STEM_COMMIT_ROWS(SG_ROWS, sIn, pre);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int i = tid + 256 * j;
      const int px = i >> 4, q = i & 15;
      const int oyl = px >> 5, oxl = px & 31;
      float* d = sG + ((oyl * 16 + (oxl >> 1)) * 64 + 4 * q) * 2 + (oxl & 1);
      d[0] = pg[j].x; d[2] = pg[j].y; d[4] = pg[j].z; d[6] = pg[j].w;
    }
    lds_barrier();
    if (t + static_cast<int>(gridDim.x) < T) fetch(t + gridDim.x);   // the next tile's loads fly under this tile's MFMAs
#pragma unroll
    for (int pp = 0; pp < 16; ++pp) {
      const float a0 = ga[pp * 128], a1 = ga[pp * 128 + 64];
      float bv[SG_KB];
#pragma unroll
      for (int kb = 0; kb < SG_KB; ++kb) bv[kb] = sIn[boff[kb] + pp * 12];
#pragma unroll
      for (int kb = 0; kb < SG_KB; ++kb) {
        acc[0][kb] = mfma2(a0, bv[kb], acc[0][kb]);
        acc[1][kb] = mfma2(a1, bv[kb], acc[1][kb]);
      }
    }
  }
  // the four waves' tiles, summed in wave order through LDS, then one coalesced partial per workgroup
  float* sRed = smem;                                                // 64 x 160 floats = 40,960 B <= the staging buffers
  for (int w = 0; w < 4; ++w) {
    __syncthreads();
    if (wv == w) {
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int kb = 0; kb < SG_KB; ++kb)
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int co = cb * 32 + (i >> 2) * 8 + hi * 4 + (i & 3);
            float* d = sRed + co * SG_NP + kb * 32 + l32;
            *d = (w == 0) ? acc[cb][kb][i] : *d + acc[cb][kb][i];
          }
    }
  }
  __syncthreads();
  float* po = partial + static_cast<size_t>(blockIdx.x) * (64 * SKP);
  for (int i = tid; i < 64 * SKP; i += 256) {
    const int co = i / SKP, k = i - co * SKP;
    po[i] = sRed[co * SG_NP + k];
  }
}

// dW[co][k] = sum over the P partials, fixed order: 16 outputs x 16 phases per workgroup, fp64 across partials
__global__ __launch_bounds__(256) void stem_wgrad_reduce(const float* __restrict__ partial, int P, float* __restrict__ gw) {
  __shared__ double sm[16][16];
  const int o = threadIdx.x & 15, ph = threadIdx.x >> 4;
  const int idx = blockIdx.x * 16 + o;                               // < 64 * SKP by the grid
  double s = 0.0;
  int p = ph;
  for (; p + 7 * 16 < P; p += 8 * 16) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = partial[static_cast<size_t>(p + 16 * u) * (64 * SKP) + idx];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += static_cast<double>(v[u]);
  }
  for (; p < P; p += 16) s += static_cast<double>(partial[static_cast<size_t>(p) * (64 * SKP) + idx]);
  sm[ph][o] = s;
  __syncthreads();
  if (ph == 0) {
    double tot = 0.0;
#pragma unroll
    for (int j = 0; j < 16; ++j) tot += sm[j][o];
    const int co = idx / SKP, k = idx - co * SKP;
    if (k < SK) gw[co * SK + k] = static_cast<float>(tot);
  }
}

struct StemGeom { int Ho, Wo, tilesX, tilesYf, tilesYg, XS; };
bool stem_geom(int N, int H, int W, StemGeom* g) {
  if (N <= 0 || H < 7 || W < 7 || static_cast<long long>(N) * H * W * 3 >= (1LL << 31)) return false;
  g->Ho = (H - 1) / 2 + 1;
  g->Wo = (W - 1) / 2 + 1;
  if (static_cast<long long>(N) * g->Ho * g->Wo * 64 >= (1LL << 33)) return false;
  g->tilesX = dbev_ceil_div(g->Wo, S_TW);
  g->tilesYf = dbev_ceil_div(g->Ho, SF_TH);
  g->tilesYg = dbev_ceil_div(g->Ho, SG_TH);
  const long long bands = static_cast<long long>(N) * g->tilesYf;
  long long xs = (1536 + bands - 1) / bands;                         // ~3 resident rounds of 2 x 256 workgroups
  g->XS = static_cast<int>(xs < 1 ? 1 : (xs > g->tilesX ? g->tilesX : xs));
  return bands * g->XS < (1LL << 30) && static_cast<long long>(N) * g->tilesYg * g->tilesX < (1LL << 30);
}
constexpr int SG_GRID = 512;                                         // persistent weight-gradient workgroups (2 per CU)

}  // namespace

extern "C" int dbev_stem7x7s2_stats_rows(int N, int H, int W) {
  StemGeom g;
  return stem_geom(N, H, W, &g) ? N * g.tilesYf * g.XS : 0;
}

extern "C" long long dbev_stem7x7s2_workspace_bytes(int N, int H, int W) {
  StemGeom g;
  if (!stem_geom(N, H, W, &g)) return 0;
  return static_cast<long long>(SG_GRID) * 64 * SKP * 4;            // the weight gradient's partials (the forward needs 37,888 B of it)
}

extern "C" int dbev_stem7x7s2_forward(const float* x_nhwc, const float* weight, int N, int H, int W, float* z_nhwc, float* stats_partial,
                                      void* workspace, size_t workspace_bytes, dbevStream_t stream) {
  StemGeom g;
  if (!stem_geom(N, H, W, &g) || x_nhwc == nullptr || weight == nullptr || z_nhwc == nullptr || workspace == nullptr) return DBEV_EINVAL;
  if (workspace_bytes < static_cast<size_t>(SKK) * 128 * 4) return DBEV_EINVAL;
  hipStream_t s = dbev_stream(stream);
  float* wpk = static_cast<float*>(workspace);
  hipLaunchKernelGGL(stem_pack, dim3(dbev_ceil_div(SKK * 128, 256)), dim3(256), 0, s, weight, wpk);
  const unsigned grid = static_cast<unsigned>(N * g.tilesYf * g.XS);
  const long long M = static_cast<long long>(N) * g.Ho * g.Wo;
  {
    DbevKt kt(DBEV_K_STEM_FWD, 2LL * M * 64 * SK, s);
    if (stats_partial != nullptr)
      hipLaunchKernelGGL((stem_fwd<true>), dim3(grid), dim3(256), 0, s, x_nhwc, wpk, z_nhwc, stats_partial, H, W, g.Ho, g.Wo, g.tilesYf, g.tilesX, g.XS);
    else
      hipLaunchKernelGGL((stem_fwd<false>), dim3(grid), dim3(256), 0, s, x_nhwc, wpk, z_nhwc, stats_partial, H, W, g.Ho, g.Wo, g.tilesYf, g.tilesX, g.XS);
  }
  DBEV_LAUNCH_CHECK();
  return 0;
}

extern "C" int dbev_stem7x7s2_backward_weight(const float* x_nhwc, const float* grad_z_nhwc, int N, int H, int W, float* grad_weight,
                                              void* workspace, size_t workspace_bytes, dbevStream_t stream) {
  StemGeom g;
  if (!stem_geom(N, H, W, &g) || x_nhwc == nullptr || grad_z_nhwc == nullptr || grad_weight == nullptr || workspace == nullptr) return DBEV_EINVAL;
  if (workspace_bytes < static_cast<size_t>(dbev_stem7x7s2_workspace_bytes(N, H, W))) return DBEV_EINVAL;
  hipStream_t s = dbev_stream(stream);
  const int T = N * g.tilesYg * g.tilesX;
  const int P = T < SG_GRID ? T : SG_GRID;
  float* partial = static_cast<float*>(workspace);
  const long long M = static_cast<long long>(N) * g.Ho * g.Wo;
  {
    DbevKt kt(DBEV_K_STEM_WGRAD, 2LL * M * 64 * SK, s);
    hipLaunchKernelGGL(stem_wgrad, dim3(P), dim3(256), 0, s, x_nhwc, grad_z_nhwc, partial, H, W, g.Ho, g.Wo, g.tilesYg, g.tilesX, T);
  }
  hipLaunchKernelGGL(stem_wgrad_reduce, dim3(64 * SKP / 16), dim3(256), 0, s, partial, P, grad_weight);
  DBEV_LAUNCH_CHECK();
  return 0;
}
