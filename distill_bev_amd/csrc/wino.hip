// 3x3 / stride 1 / pad 1 convolutions of the dense stack (the 3x3 layer of every ResNet bottleneck, the BasicBlocks of the depth net /
// BEV encoder, FPN_LSS, the CenterHead branch stacks, SECOND: mmdet3d/models/bricks/res_block.py:11-230, necks/lss_fpn.py:30-60,
// dense_heads/centerpoint_head.py:17-130, backbones/second.py:60-78 -- ~60 % of the step's convolution FLOPs) as Winograd F(2x2, 3x3)
// on the fp32 matrix cores:
//
//     Y = A^T [ (G g G^T) .* (B^T d B) ] A        d = 4x4 input tile (stride 2), g = 3x3 filter, Y = 2x2 output tile
//
// 16 multiplications per 4 outputs and channel pair instead of 36: the direct algorithm's MFMA work divided by 2.25, at fp32
// (v_mfma_f32_32x32x2_f32: exact fp32 FMA chains; the transforms are additions and multiplications by 1/2).  Per Winograd position
// p = (i, j) the channel contraction is an ordinary GEMM  M_p[tile, co] = sum_c V_p[tile, c] * U_p[c, co].
//
// Mapping.  A workgroup (4 waves, ONE wave per SIMD) owns BH x BW = 64 tiles of one image x 64 output channels; wave (mi, ni) owns
// the 32 tiles x 32 channels sub-block for ALL 16 positions: 16 accumulator tiles of 32x32 = 256 registers per lane.  That choice
// makes both transforms register-local: the lane that feeds tile t into the MFMA A operand reads t's 4x4 input pixels from the LDS
// patch and transforms them itself (no transformed-input tensor anywhere), and since the C layout of the 16 accumulators is the same
// lane/register for the same (tile, channel), the output transform is 24 in-register adds per tile -- no exchange, and the BatchNorm
// statistics of the output (sum y, sum y^2 per channel, `partial` rows in bn_finalize's layout) are in-register adds as in conv1x1.hip.
// Operands: the input patch ((2 BH + 2) x (2 BW + 2) pixels, zero padded) is staged 16 channels at a time, double-buffered; the
// transformed filters U are PRE-PACKED by wino_filter_pack in exactly the order the lanes consume them ([co block][k group of 8]
// [position][ni][lane][4 steps]), so a k group's 32 KB is one contiguous run that goes global -> LDS by LDS-DMA (global_load_lds_dwordx4:
// no staging registers, which the 256 accumulators leave no room for) and the B operand of 4 MFMA steps is ONE conflict-free
// ds_read_b128.  The reduction index is consumed in the permuted order of conv1x1.hip (step 4j+e takes k = 8j + 4*half + e).
// Per k group a wave issues 64 MFMAs (4096 cycles) against 16 + 16 LDS reads, ~130 VALU and one barrier.
// The data gradient of such a layer is the same kernel on dy with the filters rotated and transposed (mode 1 of the pack kernel).
#include "common.h"

#include <stdlib.h>

#include <atomic>

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));
// Ablation switches (DBEV_WINO_DBG bits, dev builds with -DDBEV_WINO_ABLATE only: the product kernels carry no tests of them in their loops)
#ifdef DBEV_WINO_ABLATE
#define WN_DBG(bit_) (dbg & (bit_))
#else
#define WN_DBG(bit_) 0
#endif
typedef __attribute__((address_space(3))) void* lds_ptr_t;

constexpr int WN_UCHUNK = 16 * 2 * 64 * 4;      // floats of one (channel block, k group) of packed filters = 32 KB
constexpr int W3_UCHUNK = 16 * 2 * 64 * 2;      // floats of one (channel block, k group of 4) of packed filters = 16 KB
constexpr int W3_VBUF = 16 * 32 * 4;            // floats of one transformed-input buffer of wino_fwd3 = 8 KB

// ---- filter transform + packing -------------------------------------------------------------------------------------------------
// weight element (co, c, a, b) at co*so + c*sc + a*sa + b*sb (any strides: OIHW or channels-last).
// mode 0 (forward):        reduction index k = c,  output index j = co, taps g[a][b] = w[co][c][a][b]
// mode 1 (data gradient):  k = co, j = c, taps g[a][b] = w[co][c][2-a][2-b]
// U[((jb * (K/8) + kg) * 16 + p) * 2 + ni][lane][e] = (G g G^T)[p] for k = 8 kg + 4 (lane >> 5) + e, j = 64 jb + 32 ni + (lane & 31)
__device__ __forceinline__ void wino_pack_elem(long long idx, const float* __restrict__ w, long long so, long long sc, long long sa,
                                               long long sb, int K, int J, int mode, float* __restrict__ U) {
  const int nkg = K / 8;
  const long long total = static_cast<long long>(J / 64) * nkg * 512;
  if (idx >= total) return;
  const int e = idx & 3, lane = (idx >> 2) & 63, ni = (idx >> 8) & 1;
  const long long rest = idx >> 9;
  const int kg = static_cast<int>(rest % nkg), jb = static_cast<int>(rest / nkg);
  const int k = 8 * kg + 4 * (lane >> 5) + e, j = 64 * jb + 32 * ni + (lane & 31);
  const int co = mode ? k : j, c = mode ? j : k;
  float g[3][3];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      const int aa = mode ? 2 - a : a, bb = mode ? 2 - b : b;
      g[a][b] = w[co * so + c * sc + aa * sa + bb * sb];
    }
  float t[4][3];
#pragma unroll
  for (int b = 0; b < 3; ++b) {
    t[0][b] = g[0][b];
    t[1][b] = 0.5f * ((g[0][b] + g[2][b]) + g[1][b]);
    t[2][b] = 0.5f * ((g[0][b] + g[2][b]) - g[1][b]);
    t[3][b] = g[2][b];
  }
  float* out = U + ((static_cast<long long>(jb) * nkg + kg) * 16) * 512 + ni * 256 + lane * 4 + e;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float u0 = t[i][0], u1 = 0.5f * ((t[i][0] + t[i][2]) + t[i][1]), u2 = 0.5f * ((t[i][0] + t[i][2]) - t[i][1]), u3 = t[i][2];
    out[(4 * i + 0) * 512] = u0;
    out[(4 * i + 1) * 512] = u1;
    out[(4 * i + 2) * 512] = u2;
    out[(4 * i + 3) * 512] = u3;
  }
}

__global__ __launch_bounds__(256) void wino_filter_pack(const float* __restrict__ w, long long so, long long sc, long long sa,
                                                        long long sb, int K, int J, int mode, float* __restrict__ U) {
  wino_pack_elem(static_cast<long long>(blockIdx.x) * 256 + threadIdx.x, w, so, sc, sa, sb, K, J, mode, U);
}

// two fp32 additions / subtractions in ONE VALU instruction (hipcc splits a float2 expression into two v_add_f32; every VALU
// instruction costs matrix time here, see below)
typedef float floatx2 __attribute__((ext_vector_type(2)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ floatx2 pk_add(floatx2 a, floatx2 b) {
  floatx2 r;
  asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ floatx2 pk_sub(floatx2 a, floatx2 b) {
  floatx2 r;
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ floatx2 pk_fma(floatx2 a, floatx2 b, floatx2 c) {       // a * b + c
  floatx2 r;
  asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

// a wave-uniform 64-bit value the compiler cannot prove uniform -> SGPR pair (the scalar base of an LDS-DMA)
__device__ __forceinline__ unsigned long long uniform64(unsigned long long v) {
  const unsigned lo = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(v)), hi = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(v >> 32));
  return (static_cast<unsigned long long>(hi) << 32) | lo;
}

// ---- the convolution ------------------------------------------------------------------------------------------------------------
// X [N, H, W, C] channels-last, U packed by wino_filter_pack (K = C, J = Co), Y [N, H, W, Co]; H, W even, C % 8 == 0, Co % 64 == 0,
// H * W * C * 4 < 2^31.  partial (STATS): f32[ntb, 2, Co] per tile block (sum y, sum y^2) per channel.
//
// What shapes the main loop (tools/mfma_filler_bench.hip, profiles/r04_mfma_fillers.txt): on gfx950 the fp32 MFMA runs on the SIMD's
// fp32 lanes -- a VALU instruction issued by the same wave is NOT hidden in the shadow of a v_mfma_f32_32x32x2_f32, it costs ~7 cycles
// of matrix time (11 % of an MFMA) each, while LDS reads / writes, SALU and LDS-DMA issue are free.  With one wave per SIMD the first
// version of this kernel (every lane transforming its own tile: 128 packed adds + ~100 address / select instructions per 64 MFMAs) ran
// the matrix pipe at 61 %.  So:
//  * the input transform of a k group (64 tiles x 8 channels -> 16 positions) is computed ONCE per workgroup, 1/4 by each wave
//    (thread = tile x channel pair: 16 ds_read_b64, 32 v_pk_add_f32, 16 ds_write_b64), and handed to all four waves through LDS in the
//    MFMA A-operand order -- 32 VALU per 64 MFMAs instead of 128, the two waves that share a tile half no longer repeat each other;
//  * nothing else in the loop is a VALU instruction: the patch stage (8 channels) and the packed filters go global -> LDS by LDS-DMA in
//    the scalar-base form (SGPR address advanced by SALU, one constant VGPR offset per piece); patch pixels outside the image are
//    lanes masked out of the DMA whose LDS slots were zeroed once; operands are fetched by ds_read_b128 at immediate offsets.
// Per k group g (one barrier): MFMAs on V(g), U(g); transform of patch stage g+1 -> V(g+1); DMA of U(g+1) and of patch stage g+2.
template <int BH, int BW, bool STATS>
__global__ __launch_bounds__(256, 1) void wino_fwd(const float* __restrict__ X, const float* __restrict__ U, const float* __restrict__ bias,
                                                   float* __restrict__ Y, float* __restrict__ partial, int N, int H, int W, int C,
                                                   int Co, int ntb, int dbg) {
  constexpr int PW = 2 * BW + 2, PH = 2 * BH + 2, NPIX = PW * PH;
  constexpr int NPC = (NPIX + 31) / 32;                   // patch DMA pieces (32 pixels x 32 bytes = 1 KB each)
  constexpr int PBUF = NPC * 256;                         // floats per patch stage buffer
  constexpr int VBUF = 16 * 64 * 8;                       // floats per transformed-input buffer
  __shared__ __attribute__((aligned(16))) float smem[2 * PBUF + 2 * VBUF + 2 * WN_UCHUNK];
  float* sP = smem;
  float* sV = smem + 2 * PBUF;
  float* sU = sV + 2 * VBUF;
  const bool relu = (dbg >> 16) & 1;                             // launch flag: ReLU on the output (the folded eval-mode norm + ReLU)
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int mi = w & 1, ni = w >> 1, half = lane >> 5, l31 = lane & 31;
  const int ncb = Co / 64;
  const int L = xcd_block();
  const int cb = L % ncb, tb = L / ncb;
  if (tb >= ntb) return;
  const int TH = H >> 1, TW = W >> 1;
  const int NBW = (TW + BW - 1) / BW, NBH = (TH + BH - 1) / BH;
  const int bw = tb % NBW, bh = (tb / NBW) % NBH, n = tb / (NBW * NBH);
  const int th0 = bh * BH, tw0 = bw * BW;
  const int nkg = C / 8;
  const int wu = __builtin_amdgcn_readfirstlane(w);

  // ---- patch DMA: piece j = pixels 32 j .. 32 j + 31, lane l = pixel 32 j + (l >> 1), channels 4 (l & 1) .. + 3 of the stage.
  // Wave w issues pieces w, w + 4, w + 8.  voff = byte offset of the lane's pixel inside the image; lanes of pixels outside the
  // image (or past the patch) are masked out: their LDS slots keep the zeros written below.
  unsigned pvoff[3];
  bool pok[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int pix = 32 * (wu + 4 * i) + (lane >> 1);
    const int pr = pix / PW, pc = pix - pr * PW;
    const int h = 2 * th0 - 1 + pr, x = 2 * tw0 - 1 + pc;
    pok[i] = (wu + 4 * i) < NPC && pix < NPIX && h >= 0 && h < H && x >= 0 && x < W;
    pvoff[i] = pok[i] ? static_cast<unsigned>(((h * W + x) * C + 4 * (lane & 1)) * 4) : 0u;
  }
  const unsigned long long ximg = reinterpret_cast<unsigned long long>(X + static_cast<size_t>(n) * H * W * C);
  const unsigned long long ucb = reinterpret_cast<unsigned long long>(U + static_cast<size_t>(cb) * nkg * WN_UCHUNK);
  const unsigned lds0 = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(reinterpret_cast<size_t>((lds_ptr_t)smem)));
  const unsigned uvoff = lane * 16;
#define WN_DMA(voff_, sbase_, ldsaddr_)                                                                              \
  do {                                                                                                               \
    unsigned keep_;                                                                                                  \
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"  \
                 : "=&s"(keep_) : "v"(voff_), "s"(ldsaddr_), "s"(sbase_) : "memory");                                \
  } while (0)
  // patch stage st_ -> sP[buf_]
#define WN_DMA_PATCH(i_, st_, buf_)                                                                                  \
  do {                                                                                                               \
    if (pok[i_]) {                                                                                                   \
      const unsigned long long sb_ = ximg + static_cast<unsigned long long>(st_) * 32ull;                            \
      const unsigned la_ = lds0 + static_cast<unsigned>(((buf_) * PBUF + (wu + 4 * (i_)) * 256) * 4);                \
      WN_DMA(pvoff[i_], sb_, la_);                                                                                   \
    }                                                                                                                \
  } while (0)
  // piece 8 w + i_ of the packed filters of k group kg_ -> sU[buf_]
#define WN_DMA_U(i_, kg_, buf_)                                                                                      \
  do {                                                                                                               \
    const unsigned long long sb_ = ucb + (static_cast<unsigned long long>(kg_) * WN_UCHUNK + (8 * wu + (i_)) * 256) * 4ull; \
    const unsigned la_ = lds0 + static_cast<unsigned>((2 * PBUF + 2 * VBUF + (buf_) * WN_UCHUNK + (8 * wu + (i_)) * 256) * 4); \
    WN_DMA(uvoff, sb_, la_);                                                                                         \
  } while (0)
#define WN_WAIT_VM() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")

  // ---- transform task of this thread: tile tt = tid >> 2 of the block, channel pair kq = tid & 3 of the stage
  const int tt = tid >> 2, kq = tid & 3;
  const int ttr = tt / BW, ttc = tt - ttr * BW;
  const int tsrc = ((2 * ttr) * PW + 2 * ttc) * 8 + 2 * kq;      // float index of (pixel (2 tr, 2 tc), channel 2 kq) in a patch buffer
  const int tdst = tt * 8 + 2 * kq;                              // float index in a V buffer; position p adds 512
  // ---- MFMA operands of this lane: A = V[p][32 mi + l31][4 half ..], B = U[p][ni][lane]
  const int aoff = (32 * mi + l31) * 8 + 4 * half;
  const int boff = (ni * 64 + lane) * 4;

  floatx16 acc[16];                                              // zeroed in the prologue, under the first DMA round trip

  floatx2 d[4][4], T[4][4];
#define WN_DREAD(ptr_, a_, b_) d[a_][b_] = *reinterpret_cast<const floatx2*>((ptr_) + ((a_) * PW + (b_)) * 8)
  // row i of B^T d: d0 - d2, d1 + d2, d2 - d1, d1 - d3 (per column b)
#define WN_TOP(i_, b_) T[i_][b_] = (i_) == 0 ? pk_sub(d[0][b_], d[2][b_]) : (i_) == 1 ? pk_add(d[1][b_], d[2][b_]) : (i_) == 2 ? pk_sub(d[2][b_], d[1][b_]) : pk_sub(d[1][b_], d[3][b_])
  // V[i][j] = (row i of B^T d) B: t0 - t2, t1 + t2, t2 - t1, t1 - t3
#define WN_VCOL(i_, j_) \
  ((j_) == 0 ? pk_sub(T[i_][0], T[i_][2]) : (j_) == 1 ? pk_add(T[i_][1], T[i_][2]) : (j_) == 2 ? pk_sub(T[i_][2], T[i_][1]) : pk_sub(T[i_][1], T[i_][3]))
#define WN_VSTORE(dst_, i_, j_, v_) *reinterpret_cast<floatx2*>((dst_) + (4 * (i_) + (j_)) * 512) = (v_)
#define WN_VOUT(dst_, i_, j_) WN_VSTORE(dst_, i_, j_, WN_VCOL(i_, j_))
  floatx2 vo[8];

  // prologue: zero the patch buffers (the masked-out slots stay zero for the whole kernel), then stage 0, 1 and the first filters
  for (int i = tid; i < 2 * PBUF / 4; i += 256) reinterpret_cast<float4*>(sP)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 3; ++i) WN_DMA_PATCH(i, 0, 0);
#pragma unroll
  for (int i = 0; i < 3; ++i) WN_DMA_PATCH(i, nkg > 1 ? 1 : 0, 1);
#pragma unroll
  for (int i = 0; i < 8; ++i) WN_DMA_U(i, 0, 0);
  __builtin_amdgcn_sched_barrier(0);                             // the 256 accumulator writes go HERE: behind the DMA issue, under its latency
#pragma unroll
  for (int p = 0; p < 16; ++p)
#pragma unroll
    for (int r = 0; r < 16; ++r) {                                // (as asm: plain assignments are sunk to the first use, behind the wait)
      float z;
      asm volatile("v_accvgpr_write_b32 %0, 0" : "=a"(z));
      acc[p][r] = z;
    }
  __builtin_amdgcn_sched_barrier(0);
  WN_WAIT_VM();
  __syncthreads();
  {
    const float* ps = sP + tsrc;
    float* vd = sV + tdst;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b2 = 0; b2 < 4; ++b2) WN_DREAD(ps, a, b2);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int b2 = 0; b2 < 4; ++b2) WN_TOP(i, b2);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) WN_VOUT(vd, i, j);
  }
  __syncthreads();

  const int nkg_run = WN_DBG(2) ? 1 : nkg;
  for (int kg = 0; kg < nkg_run; ++kg) {
    const int cur = kg & 1, nxt = cur ^ 1;
    // branch-free body: past the end the DMAs repeat the last chunk / stage into buffers nobody reads any more
    const int kgn = min(kg + 1, nkg - 1), stn = min(kg + 2, nkg - 1);
    const float* ps = sP + nxt * PBUF + tsrc;                    // patch stage kg + 1 (landed before the previous barrier)
    float* vd = sV + nxt * VBUF + tdst;                          // V(kg + 1)
    const float4* ap = reinterpret_cast<const float4*>(sV + cur * VBUF + aoff);
    const float4* bp = reinterpret_cast<const float4*>(sU + cur * WN_UCHUNK + boff);
    float4 av = ap[0], bv = bp[0], an = av, bn = bv;
#pragma unroll
    for (int p = 0; p < 16; ++p) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int sl = 4 * p + e;
        const float a_ = e == 0 ? av.x : e == 1 ? av.y : e == 2 ? av.z : av.w;
        const float b_ = e == 0 ? bv.x : e == 1 ? bv.y : e == 2 ? bv.z : bv.w;
        acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_, b_, acc[p], 0, 0, 0);
        if (e == 0 && p < 15 && !WN_DBG(512)) { an = ap[(p + 1) * 128]; bn = bp[(p + 1) * 128]; }
        if (!WN_DBG(1024)) {
        // the transform's VALU work in three groups (a lone VALU instruction between two MFMAs costs ~17 cycles of matrix time, one
        // of a group of 8-16 ~6: profiles/r04_mfma_fillers.txt); the LDS reads / writes around them one per slot
        if (sl >= 1 && sl < 17) { const int q = sl - 1; WN_DREAD(ps, q >> 2, q & 3); }
        if (sl == 20) {
#pragma unroll
          for (int q = 0; q < 16; ++q) WN_TOP(q >> 2, q & 3);
        }
        if (sl == 24 || sl == 34) {
#pragma unroll
          for (int q = 0; q < 8; ++q) vo[q] = WN_VCOL((sl == 24 ? 0 : 2) + (q >> 2), q & 3);
        }
        if (sl >= 25 && sl < 33) { const int q = sl - 25; WN_VSTORE(vd, q >> 2, q & 3, vo[q]); }
        if (sl >= 35 && sl < 43) { const int q = sl - 35; WN_VSTORE(vd, 2 + (q >> 2), q & 3, vo[q]); }
        }
        if (sl >= 2 && sl < 10 && !WN_DBG(16)) WN_DMA_U(sl - 2, kgn, nxt);
        if (sl >= 10 && sl < 13 && !WN_DBG(8)) WN_DMA_PATCH(sl - 10, stn, cur);     // stage kg + 2 over stage kg (read during kg - 1)
        __builtin_amdgcn_sched_barrier(0);
      }
      av = an; bv = bn;
    }
    if (!WN_DBG(4)) {
      WN_WAIT_VM();
      __syncthreads();
    }
  }
#undef WN_DREAD
#undef WN_TOP
#undef WN_VOUT
#undef WN_VCOL
#undef WN_VSTORE
#undef WN_DMA
#undef WN_DMA_PATCH
#undef WN_DMA_U
#undef WN_WAIT_VM

  // output transform + store: register r of every accumulator = tile ti = (r & 3) + 8 (r >> 2) + 4 half of this wave's 32, channel l31
  const int co = 64 * cb + 32 * ni + l31;
  const float bco = bias != nullptr ? bias[co] : 0.f;
  float s1 = 0.f, s2 = 0.f;
  if (th0 + BH <= TH && tw0 + BW <= TW && !WN_DBG(1)) {
    // The block lies inside the image (uniform test): no per-tile bounds tests.  Nothing overlaps this part -- every VALU instruction
    // is paid in full -- so registers r, r + 1 (two tiles side by side) go through the transform as ONE packed operation each, and the
    // four pixels of a tile are stored at ONE 32-bit offset from four scalar bases (pixel (0,0), (0,1), (1,0), (1,1) of the image).
    static_assert(BW == 8 || BW == 16, "tile (row, column) of register r below");
    unsigned long long yb[8];
    yb[0] = uniform64(reinterpret_cast<unsigned long long>(Y + static_cast<size_t>(n) * H * W * Co));
    yb[1] = yb[0] + static_cast<unsigned long long>(Co) * 4ull;
    yb[2] = yb[0] + static_cast<unsigned long long>(W) * Co * 4ull;
    yb[3] = yb[2] + static_cast<unsigned long long>(Co) * 4ull;
#pragma unroll
    for (int i = 0; i < 4; ++i) yb[4 + i] = yb[i] + static_cast<unsigned long long>(Co) * 8ull;
    // scalar-base store: the compiler turns base + offset arithmetic on pointers into 64-bit VALU additions per store
#define WN_ST(voff_, val_, sbase_) asm volatile("global_store_dword %0, %1, %2" :: "v"(voff_), "v"(val_), "s"(sbase_) : "memory")
    // tile of register r: BW = 8: row 4 mi + (r >> 2), column (r & 3) + 4 half; BW = 16: row 2 mi + (r >> 3), column (r & 3) + 4 half + 8 ((r >> 2) & 1)
    const int tr0 = (BW == 8 ? 4 : 2) * mi, tc0 = 4 * half;
    const unsigned voff0 = static_cast<unsigned>(((2 * (th0 + tr0) * W + 2 * (tw0 + tc0)) * Co + co) * 4);
    const unsigned rowstep = static_cast<unsigned>(2 * W * Co * 4), colstep = static_cast<unsigned>(2 * Co * 4);
    const floatx2 b2 = {bco, bco};
    floatx2 t1 = {0.f, 0.f}, t2 = {0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      const int dr = BW == 8 ? (r >> 2) : (r >> 3), dc = BW == 8 ? (r & 3) : (r & 3) + 8 * ((r >> 2) & 1);
      floatx2 s0[4], s1r[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const floatx2 a0 = {acc[0 + j][r], acc[0 + j][r + 1]}, a1 = {acc[4 + j][r], acc[4 + j][r + 1]};
        const floatx2 a2 = {acc[8 + j][r], acc[8 + j][r + 1]}, a3 = {acc[12 + j][r], acc[12 + j][r + 1]};
        s0[j] = pk_add(pk_add(a0, a1), a2);
        s1r[j] = pk_sub(pk_sub(a1, a2), a3);
      }
      floatx2 y00 = pk_add(pk_add(pk_add(s0[0], s0[1]), s0[2]), b2), y01 = pk_add(pk_sub(pk_sub(s0[1], s0[2]), s0[3]), b2);
      floatx2 y10 = pk_add(pk_add(pk_add(s1r[0], s1r[1]), s1r[2]), b2), y11 = pk_add(pk_sub(pk_sub(s1r[1], s1r[2]), s1r[3]), b2);
      if (relu) {
        y00.x = fmaxf(y00.x, 0.f); y00.y = fmaxf(y00.y, 0.f); y01.x = fmaxf(y01.x, 0.f); y01.y = fmaxf(y01.y, 0.f);
        y10.x = fmaxf(y10.x, 0.f); y10.y = fmaxf(y10.y, 0.f); y11.x = fmaxf(y11.x, 0.f); y11.y = fmaxf(y11.y, 0.f);
      }
      const unsigned vo = voff0 + dr * rowstep + dc * colstep;
      WN_ST(vo, y00.x, yb[0]); WN_ST(vo, y00.y, yb[4]);              // bases 4..7: the tile one column to the right
      WN_ST(vo, y01.x, yb[1]); WN_ST(vo, y01.y, yb[5]);
      WN_ST(vo, y10.x, yb[2]); WN_ST(vo, y10.y, yb[6]);
      WN_ST(vo, y11.x, yb[3]); WN_ST(vo, y11.y, yb[7]);
      if (STATS) {
        t1 = pk_add(t1, pk_add(pk_add(y00, y01), pk_add(y10, y11)));
        t2 = pk_fma(y00, y00, t2); t2 = pk_fma(y01, y01, t2); t2 = pk_fma(y10, y10, t2); t2 = pk_fma(y11, y11, t2);
      }
    }
    s1 = t1.x + t1.y; s2 = t2.x + t2.y;
#undef WN_ST
  } else {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int t = 32 * mi + (r & 3) + 8 * (r >> 2) + 4 * half;
      const int tr = t / BW, tc = t - tr * BW;
      const int th = th0 + tr, tw = tw0 + tc;
      float s0[4], s1r[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        s0[j] = (acc[0 + j][r] + acc[4 + j][r]) + acc[8 + j][r];
        s1r[j] = (acc[4 + j][r] - acc[8 + j][r]) - acc[12 + j][r];
      }
      float y00 = (s0[0] + s0[1]) + s0[2] + bco, y01 = (s0[1] - s0[2]) - s0[3] + bco;
      float y10 = (s1r[0] + s1r[1]) + s1r[2] + bco, y11 = (s1r[1] - s1r[2]) - s1r[3] + bco;
      if (relu) { y00 = fmaxf(y00, 0.f); y01 = fmaxf(y01, 0.f); y10 = fmaxf(y10, 0.f); y11 = fmaxf(y11, 0.f); }
      if (th < TH && tw < TW && !WN_DBG(1)) {
        float* yp = Y + (static_cast<size_t>(n * H + 2 * th) * W + 2 * tw) * Co + co;
        yp[0] = y00;
        yp[Co] = y01;
        yp[static_cast<size_t>(W) * Co] = y10;
        yp[static_cast<size_t>(W) * Co + Co] = y11;
        if (STATS) {
          s1 += (y00 + y01) + (y10 + y11);
          s2 = fmaf(y00, y00, s2); s2 = fmaf(y01, y01, s2); s2 = fmaf(y10, y10, s2); s2 = fmaf(y11, y11, s2);
        }
      }
    }
  }
  if (STATS) {
    float* red = smem;                                           // [2 which][2 mi][64 channels of the block]
    s1 += __shfl_xor(s1, 32);
    s2 += __shfl_xor(s2, 32);
    __syncthreads();
    if (half == 0) {
      red[(0 * 2 + mi) * 64 + 32 * ni + l31] = s1;
      red[(1 * 2 + mi) * 64 + 32 * ni + l31] = s2;
    }
    __syncthreads();
    if (tid < 128) {
      const int which = tid >> 6, c = tid & 63;
      partial[(static_cast<size_t>(tb) * 2 + which) * Co + 64 * cb + c] = red[(which * 2 + 0) * 64 + c] + red[(which * 2 + 1) * 64 + c];
    }
  }
}

// ---- the convolution, persistent workgroups (round 6) ------------------------------------------------------------------------------
// Counters of wino_fwd on the image backbone's 48-image shapes (profiles/r06_pmc_wino.txt; clock 2.4-2.5 GHz by GRBM_GUI_ACTIVE -- no
// throttling in the spaced runs): the matrix pipe is busy 0.84 / 0.78 / 0.65 of the cycles a CU has a workgroup (256 / 128 / 64 input
// channels: ~7 us per work item are prologue -- zeroing, the first DMA round trip, the first transform -- and epilogue, whatever the
// item's length), and a CU HAS a workgroup only 0.74-0.79 of the launch (turnover between workgroups, the partly filled last round).
// Same main loop, but ONE workgroup per CU that walks work items (tile block x 64-channel block):
//  * items come from eight queues, one per XCD (the items of an XCD's queue are neighbours: the four channel blocks of a tile block
//    fetch the same patch, from ONE L2), a device counter each; the first item of a workgroup is its own by position, the counter
//    for item i + 1 is bumped under item i's first k group; a workgroup whose queue is empty finishes its item, then takes from
//    the queues that still have work, so the last round is shared by all 256 CUs whatever the split;
//  * behind the last k group of an item the patch stages 0 / 1 and the first filter chunk of the NEXT item are requested, then the
//    output transform and the stores of this item run under that round trip -- the prologue's latency is gone;
//  * patch slots outside the image are zeroed per item by the lanes the DMA skips (the item's edge mask is constant over its stages);
//  * the wait for the next item's first stages leaves this item's 64 stores in flight (vmcnt counts loads and stores in issue order,
//    the stores are the youngest), and the barriers between the stores and the next k group do not drain it.
// The last workgroup out resets the counters (a set of counters per launch in flight: g_wino_queue).  Results are bit-identical to
// wino_fwd's: same per-item arithmetic in the same order.
// dev builds (-DDBEV_WINO_ABLATE): shader-clock stamps of workgroup 0 between the phases of an item, summed over its items
#ifdef DBEV_WINO_ABLATE
__device__ unsigned long long g_wp_prof[16];
#define WP_T0() unsigned long long wp_t_ = __builtin_amdgcn_s_memtime(), wp_acc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}; \
  const unsigned long long wp_c0_ = wp_t_, wp_r0_ = __builtin_amdgcn_s_memrealtime()
#define WP_STAMP(i_) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); wp_acc_[i_] += n_ - wp_t_; wp_t_ = n_; if ((i_) == 6) ++wp_acc_[7]; } while (0)
#define WP_DUMP() do { if (blockIdx.x == 0 && tid == 0) { for (int i_ = 0; i_ < 8; ++i_) g_wp_prof[i_] = wp_acc_[i_];          \
    g_wp_prof[8] += __builtin_amdgcn_s_memtime() - wp_c0_; g_wp_prof[9] += __builtin_amdgcn_s_memrealtime() - wp_r0_;            \
    g_wp_prof[10] += wp_acc_[7]; g_wp_prof[11] += 1; } } while (0)    /* [8..11]: summed over launches: shader cycles, 100 MHz ticks, items, launches */
#else
#define WP_T0()
#define WP_STAMP(i_)
#define WP_DUMP()
#endif
constexpr int WQ_SETS = 64, WQ_INTS = 16;                  // [8 queue counters, 1 exit counter, padding]
__device__ int g_wino_queue[WQ_SETS * WQ_INTS];

template <int BH, int BW, bool STATS>
__global__ __launch_bounds__(256, 1) void wino_fwdp(const float* __restrict__ X, const float* __restrict__ U, const float* __restrict__ bias,
                                                    float* __restrict__ Y, float* __restrict__ partial, int N, int H, int W, int C,
                                                    int Co, int ntb, int dbg, int qset) {
  constexpr int PW = 2 * BW + 2, PH = 2 * BH + 2, NPIX = PW * PH;
  constexpr int NPC = (NPIX + 31) / 32;                   // patch DMA pieces (32 pixels x 32 bytes = 1 KB each)
  constexpr int PBUF = NPC * 256;                         // floats per patch stage buffer
  constexpr int VBUF = 16 * 64 * 8;                       // floats per transformed-input buffer
  __shared__ __attribute__((aligned(16))) float smem[2 * PBUF + 2 * VBUF + 2 * WN_UCHUNK];
  __shared__ int s_item;
  float* sP = smem;
  float* sV = smem + 2 * PBUF;
  float* sU = sV + 2 * VBUF;
  const bool relu = (dbg >> 16) & 1;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int mi = w & 1, ni = w >> 1, half = lane >> 5, l31 = lane & 31;
  const int ncb = Co / 64;
  const int TH = H >> 1, TW = W >> 1;
  const int NBW = (TW + BW - 1) / BW, NBH = (TH + BH - 1) / BH;
  const int nkg = C / 8;
  const int wu = __builtin_amdgcn_readfirstlane(w);
  const int nitems = ntb * ncb;
  const int qper = (nitems + DBEV_NUM_XCD - 1) / DBEV_NUM_XCD;
  const int q0 = blockIdx.x % DBEV_NUM_XCD;
  int* const ctr = g_wino_queue + qset * WQ_INTS;
  // Queue q = items [q qper, (q + 1) qper).  The workgroups of XCD q (blockIdx % 8 == q: placement is a speed assumption only) take their
  // FIRST item without asking: workgroup j of the queue (j = blockIdx / 8) owns local index j; the head counter hands out the local
  // indices from nown = (workgroups of the queue) on.  A workgroup whose queue is empty looks at the other heads (one 64-byte read)
  // and bumps the first queue that still has items.
  const int nown = (static_cast<int>(gridDim.x) - q0 + DBEV_NUM_XCD - 1) / DBEV_NUM_XCD;
  const int myend = min(nitems, (q0 + 1) * qper) - q0 * qper;         // local indices of my queue: [0, myend)
  auto steal = [&]() -> int {
    for (int k = 1; k < DBEV_NUM_XCD; ++k) {
      const int q = (q0 + k) % DBEV_NUM_XCD;
      const int beg = q * qper, cnt = min(nitems, beg + qper) - beg;
      const int wgs = (static_cast<int>(gridDim.x) - q + DBEV_NUM_XCD - 1) / DBEV_NUM_XCD;
      if (cnt <= wgs) continue;
      if (__hip_atomic_load(ctr + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + wgs >= cnt) continue;
      const int v = atomicAdd(ctr + q, 1) + wgs;
      if (v < cnt) return beg + v;
    }
    return -1;
  };
  const unsigned lds0 = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(reinterpret_cast<size_t>((lds_ptr_t)smem)));
  const unsigned uvoff = lane * 16;
#define WN_DMA(voff_, sbase_, ldsaddr_)                                                                              \
  do {                                                                                                               \
    unsigned keep_;                                                                                                  \
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"  \
                 : "=&s"(keep_) : "v"(voff_), "s"(ldsaddr_), "s"(sbase_) : "memory");                                \
  } while (0)
#define WN_DMA_PATCH(i_, st_, buf_)                                                                                  \
  do {                                                                                                               \
    if (pok[i_]) {                                                                                                   \
      const unsigned long long sb_ = ximg + static_cast<unsigned long long>(st_) * 32ull;                            \
      const unsigned la_ = lds0 + static_cast<unsigned>(((buf_) * PBUF + (wu + 4 * (i_)) * 256) * 4);                \
      WN_DMA(pvoff[i_], sb_, la_);                                                                                   \
    }                                                                                                                \
  } while (0)
#define WN_DMA_U(i_, kg_, buf_)                                                                                      \
  do {                                                                                                               \
    const unsigned long long sb_ = ucb + (static_cast<unsigned long long>(kg_) * WN_UCHUNK + (8 * wu + (i_)) * 256) * 4ull; \
    const unsigned la_ = lds0 + static_cast<unsigned>((2 * PBUF + 2 * VBUF + (buf_) * WN_UCHUNK + (8 * wu + (i_)) * 256) * 4); \
    WN_DMA(uvoff, sb_, la_);                                                                                         \
  } while (0)
#define WN_WAIT_VM() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
  // gfx9 counts loads AND stores in vmcnt, in issue order, and __syncthreads() drains it: between an item's output stores and the next
  // item's first k group the barriers are bare (LDS traffic only) and the wait for the next item's first stages -- requested BEFORE the
  // stores -- leaves the youngest 63 operations (the stores) in flight
#define WP_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
  // geometry of item it_: channel block, tile block -> image, first tile; the DMA lanes' pixel offsets and edge mask; then the zeros
  // of the masked slots (both stage buffers) and the requests for stages 0, 1 and the first filter chunk
  unsigned pvoff[3];
  bool pok[3];
  unsigned long long ximg = 0, ucb = 0;
  int cb = 0, tb = 0, n = 0, th0 = 0, tw0 = 0;
#define WP_START(it_)                                                                                                \
  do {                                                                                                               \
    cb = (it_) % ncb; tb = (it_) / ncb;                                                                              \
    const int bw_ = tb % NBW, bh_ = (tb / NBW) % NBH;                                                                \
    n = tb / (NBW * NBH); th0 = bh_ * BH; tw0 = bw_ * BW;                                                            \
    _Pragma("unroll") for (int i = 0; i < 3; ++i) {                                                                  \
      const int pix = 32 * (wu + 4 * i) + (lane >> 1);                                                               \
      const int pr = pix / PW, pc = pix - pr * PW;                                                                   \
      const int h = 2 * th0 - 1 + pr, x = 2 * tw0 - 1 + pc;                                                          \
      const bool in_patch = (wu + 4 * i) < NPC && pix < NPIX;                                                        \
      pok[i] = in_patch && h >= 0 && h < H && x >= 0 && x < W;                                                       \
      pvoff[i] = pok[i] ? static_cast<unsigned>(((h * W + x) * C + 4 * (lane & 1)) * 4) : 0u;                        \
      if (in_patch && !pok[i]) {                                                                                     \
        float4* z_ = reinterpret_cast<float4*>(sP + (wu + 4 * i) * 256 + lane * 4);                                  \
        z_[0] = make_float4(0.f, 0.f, 0.f, 0.f);                                                                     \
        z_[PBUF / 4] = make_float4(0.f, 0.f, 0.f, 0.f);                                                              \
      }                                                                                                              \
    }                                                                                                                \
    ximg = uniform64(reinterpret_cast<unsigned long long>(X + static_cast<size_t>(n) * H * W * C));                  \
    ucb = uniform64(reinterpret_cast<unsigned long long>(U + static_cast<size_t>(cb) * nkg * WN_UCHUNK));            \
    _Pragma("unroll") for (int i = 0; i < 3; ++i) WN_DMA_PATCH(i, 0, 0);                                             \
    _Pragma("unroll") for (int i = 0; i < 3; ++i) WN_DMA_PATCH(i, nkg > 1 ? 1 : 0, 1);                               \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) WN_DMA_U(i, 0, 0);                                                 \
  } while (0)

  const int tt = tid >> 2, kq = tid & 3;
  const int ttr = tt / BW, ttc = tt - ttr * BW;
  const int tsrc = ((2 * ttr) * PW + 2 * ttc) * 8 + 2 * kq;
  const int tdst = tt * 8 + 2 * kq;
  const int aoff = (32 * mi + l31) * 8 + 4 * half;
  const int boff = (ni * 64 + lane) * 4;

  const floatx16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  floatx2 d[4][4], T[4][4];
#define WN_DREAD(ptr_, a_, b_) d[a_][b_] = *reinterpret_cast<const floatx2*>((ptr_) + ((a_) * PW + (b_)) * 8)
#define WN_TOP(i_, b_) T[i_][b_] = (i_) == 0 ? pk_sub(d[0][b_], d[2][b_]) : (i_) == 1 ? pk_add(d[1][b_], d[2][b_]) : (i_) == 2 ? pk_sub(d[2][b_], d[1][b_]) : pk_sub(d[1][b_], d[3][b_])
#define WN_VCOL(i_, j_) \
  ((j_) == 0 ? pk_sub(T[i_][0], T[i_][2]) : (j_) == 1 ? pk_add(T[i_][1], T[i_][2]) : (j_) == 2 ? pk_sub(T[i_][2], T[i_][1]) : pk_sub(T[i_][1], T[i_][3]))
#define WN_VSTORE(dst_, i_, j_, v_) *reinterpret_cast<floatx2*>((dst_) + (4 * (i_) + (j_)) * 512) = (v_)
#define WN_VOUT(dst_, i_, j_) WN_VSTORE(dst_, i_, j_, WN_VCOL(i_, j_))
  floatx2 vo[8];

  // prologue of the workgroup: zero the patch buffers once, take the first item
  for (int i = tid; i < 2 * PBUF / 4; i += 256) reinterpret_cast<float4*>(sP)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  __syncthreads();
  int item = static_cast<int>(blockIdx.x) / DBEV_NUM_XCD < myend ? q0 * qper + static_cast<int>(blockIdx.x) / DBEV_NUM_XCD : -1;
  if (item < 0) {                                                 // (more workgroups than items in my queue: help elsewhere)
    if (tid == 0) s_item = steal();
    __syncthreads();
    item = __builtin_amdgcn_readfirstlane(s_item);
  }
  if (item >= 0) WP_START(item);

  bool fast_prev = false;
  WP_T0();
  while (item >= 0) {
    // stages 0 / 1 and the first filters of `item` are in flight (requested by WP_START: above, or behind the previous item's loop)
    if (fast_prev) asm volatile("s_waitcnt vmcnt(63)" ::: "memory");   // the 64 stores of the previous item's epilogue are the youngest
    else WN_WAIT_VM();
    WP_BARRIER();
    WP_STAMP(0);                                                  // waited for this item's first stages
    const int co = 64 * cb + 32 * ni + l31;
    float bco = bias != nullptr ? bias[co] : 0.f;                 // (its wait sits behind the first k group)
    // the bump of my queue's head for the item after this one, by lane 0 of wave 0, as ONE instruction whose result nobody looks at
    // before the wait at the end of the first k group (a compiler-visible atomic is waited for on the spot: ~1 us in front of the barrier)
    int vnext = 0;
    if (wu == 0) {
      const unsigned long long cq_ = reinterpret_cast<unsigned long long>(ctr + q0);
      asm volatile("s_mov_b64 s[2:3], exec\n\ts_mov_b64 exec, 1\n\tv_mov_b32 %0, 1\n\tglobal_atomic_add %0, %1, %0, %2 sc0\n\ts_mov_b64 exec, s[2:3]"
                   : "=&v"(vnext) : "v"(0), "s"(cq_) : "s2", "s3", "memory");
    }
    {
      const float* ps = sP + tsrc;
      float* vd = sV + tdst;
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b2 = 0; b2 < 4; ++b2) WN_DREAD(ps, a, b2);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int b2 = 0; b2 < 4; ++b2) WN_TOP(i, b2);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) WN_VOUT(vd, i, j);
    }
    WP_BARRIER();
    WP_STAMP(1);                                                  // first transform

    // one k group; FIRST_: the item's first -- its MFMAs start the 16 accumulators from the constant 0 (no zeroing pass, and no
    // accumulator value is carried from one item to the next: the compiler shuffled and spilled the tuples when one was)
#define WP_KGROUP(FIRST_, kg_)                                                                                        \
    do {                                                                                                             \
      const int cur = (kg_) & 1, nxt = cur ^ 1;                                                                      \
      const int kgn = min((kg_) + 1, nkg - 1), stn = min((kg_) + 2, nkg - 1);                                        \
      const float* ps = sP + nxt * PBUF + tsrc;                                                                      \
      float* vd = sV + nxt * VBUF + tdst;                                                                            \
      const float4* ap = reinterpret_cast<const float4*>(sV + cur * VBUF + aoff);                                    \
      const float4* bp = reinterpret_cast<const float4*>(sU + cur * WN_UCHUNK + boff);                               \
      float4 av = ap[0], bv = bp[0], an = av, bn = bv;                                                               \
      _Pragma("unroll") for (int p = 0; p < 16; ++p) {                                                               \
        _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                              \
          const int sl = 4 * p + e;                                                                                  \
          const float a_ = e == 0 ? av.x : e == 1 ? av.y : e == 2 ? av.z : av.w;                                     \
          const float b_ = e == 0 ? bv.x : e == 1 ? bv.y : e == 2 ? bv.z : bv.w;                                     \
          if ((FIRST_) && e == 0) acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_, b_, zero16, 0, 0, 0);            \
          else acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_, b_, acc[p], 0, 0, 0);                               \
          if (e == 0 && p < 15) { an = ap[(p + 1) * 128]; bn = bp[(p + 1) * 128]; }                                  \
          if (sl >= 1 && sl < 17) { const int q = sl - 1; WN_DREAD(ps, q >> 2, q & 3); }                             \
          if (sl == 20) {                                                                                            \
            _Pragma("unroll") for (int q = 0; q < 16; ++q) WN_TOP(q >> 2, q & 3);                                    \
          }                                                                                                          \
          if (sl == 24 || sl == 34) {                                                                                \
            _Pragma("unroll") for (int q = 0; q < 8; ++q) vo[q] = WN_VCOL((sl == 24 ? 0 : 2) + (q >> 2), q & 3);     \
          }                                                                                                          \
          if (sl >= 25 && sl < 33) { const int q = sl - 25; WN_VSTORE(vd, q >> 2, q & 3, vo[q]); }                   \
          if (sl >= 35 && sl < 43) { const int q = sl - 35; WN_VSTORE(vd, 2 + (q >> 2), q & 3, vo[q]); }             \
          if (sl >= 2 && sl < 10) WN_DMA_U(sl - 2, kgn, nxt);                                                        \
          if (sl >= 10 && sl < 13) WN_DMA_PATCH(sl - 10, stn, cur);                                                  \
          __builtin_amdgcn_sched_barrier(0);                                                                         \
        }                                                                                                            \
        av = an; bv = bn;                                                                                            \
      }                                                                                                              \
      WN_WAIT_VM();                                                                                                  \
      __syncthreads();                                                                                               \
    } while (0)
    // an accumulator element where the output transform consumes it: read from its AGPR THERE (left to the compiler, all 256 are copied
    // to VGPRs right behind the main loop and everything else that lives across the item loop is spilled; the last MFMA is thousands
    // of cycles back: no hazard for the hand-written read)
#define WP_AR(x_) ({ float v_; asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v_) : "a"(x_)); v_; })
    floatx16 acc[16];
    WP_KGROUP(true, 0);
    asm volatile("" : "+v"(vnext), "+v"(bco));                    // (both have landed: the k group ended with vmcnt(0))
    WP_STAMP(2);                                                  // first k group
    for (int kg = 1; kg < nkg; ++kg) WP_KGROUP(false, kg);
    WP_STAMP(3);                                                  // the other k groups
#undef WP_KGROUP

    // this item's geometry for the epilogue; the next item (its requests go out before the epilogue's stores)
    const int e_cb = cb, e_tb = tb, e_n = n, e_th0 = th0, e_tw0 = tw0;
    // the next item: my queue's, known already -- its first stages are requested before this item's stores --, or, once my queue is
    // empty, somebody else's: looked for AFTER this item's output is on its way
    if (tid == 0) s_item = vnext + nown < myend ? q0 * qper + vnext + nown : -1;
    __syncthreads();
    int next = __builtin_amdgcn_readfirstlane(s_item);
    const bool late = next < 0;
    WP_STAMP(4);                                                  // next item index
    if (!late) WP_START(next);
    __builtin_amdgcn_sched_barrier(0);
    WP_STAMP(5);                                                  // next item's geometry and requests

    // output transform + store: register r of every accumulator = tile ti = (r & 3) + 8 (r >> 2) + 4 half of this wave's 32, channel l31
    float s1 = 0.f, s2 = 0.f;
    fast_prev = e_th0 + BH <= TH && e_tw0 + BW <= TW;
    if (fast_prev) {
      static_assert(BW == 8 || BW == 16, "tile (row, column) of register r below");
      unsigned long long yb[8];
      yb[0] = uniform64(reinterpret_cast<unsigned long long>(Y + static_cast<size_t>(e_n) * H * W * Co));
      yb[1] = yb[0] + static_cast<unsigned long long>(Co) * 4ull;
      yb[2] = yb[0] + static_cast<unsigned long long>(W) * Co * 4ull;
      yb[3] = yb[2] + static_cast<unsigned long long>(Co) * 4ull;
#pragma unroll
      for (int i = 0; i < 4; ++i) yb[4 + i] = yb[i] + static_cast<unsigned long long>(Co) * 8ull;
#define WN_ST(voff_, val_, sbase_) asm volatile("global_store_dword %0, %1, %2" :: "v"(voff_), "v"(val_), "s"(sbase_) : "memory")
      // (opaque copies: everything below that depends only on the lane and the layer would otherwise be hoisted out of the item loop and
      // kept -- spilled -- across the main loop, its reloads then waiting between the stores)
      int Wq = W, Coq = Co;
      asm volatile("" : "+s"(Wq), "+s"(Coq));
      const int tr0 = (BW == 8 ? 4 : 2) * mi, tc0 = 4 * half;
      const unsigned voff0 = static_cast<unsigned>(((2 * (e_th0 + tr0) * Wq + 2 * (e_tw0 + tc0)) * Coq + co) * 4);
      const unsigned rowstep = static_cast<unsigned>(2 * Wq * Coq * 4), colstep = static_cast<unsigned>(2 * Coq * 4);
      const floatx2 b2 = {bco, bco};
      floatx2 t1 = {0.f, 0.f}, t2 = {0.f, 0.f};
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const int dr = BW == 8 ? (r >> 2) : (r >> 3), dc = BW == 8 ? (r & 3) : (r & 3) + 8 * ((r >> 2) & 1);
        floatx2 s0[4], s1r[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const floatx2 a0 = {WP_AR(acc[0 + j][r]), WP_AR(acc[0 + j][r + 1])}, a1 = {WP_AR(acc[4 + j][r]), WP_AR(acc[4 + j][r + 1])};
          const floatx2 a2 = {WP_AR(acc[8 + j][r]), WP_AR(acc[8 + j][r + 1])}, a3 = {WP_AR(acc[12 + j][r]), WP_AR(acc[12 + j][r + 1])};
          s0[j] = pk_add(pk_add(a0, a1), a2);
          s1r[j] = pk_sub(pk_sub(a1, a2), a3);
        }
        floatx2 y00 = pk_add(pk_add(pk_add(s0[0], s0[1]), s0[2]), b2), y01 = pk_add(pk_sub(pk_sub(s0[1], s0[2]), s0[3]), b2);
        floatx2 y10 = pk_add(pk_add(pk_add(s1r[0], s1r[1]), s1r[2]), b2), y11 = pk_add(pk_sub(pk_sub(s1r[1], s1r[2]), s1r[3]), b2);
        if (relu) {
          y00.x = fmaxf(y00.x, 0.f); y00.y = fmaxf(y00.y, 0.f); y01.x = fmaxf(y01.x, 0.f); y01.y = fmaxf(y01.y, 0.f);
          y10.x = fmaxf(y10.x, 0.f); y10.y = fmaxf(y10.y, 0.f); y11.x = fmaxf(y11.x, 0.f); y11.y = fmaxf(y11.y, 0.f);
        }
        const unsigned vo_ = voff0 + dr * rowstep + dc * colstep;
        WN_ST(vo_, y00.x, yb[0]); WN_ST(vo_, y00.y, yb[4]);
        WN_ST(vo_, y01.x, yb[1]); WN_ST(vo_, y01.y, yb[5]);
        WN_ST(vo_, y10.x, yb[2]); WN_ST(vo_, y10.y, yb[6]);
        WN_ST(vo_, y11.x, yb[3]); WN_ST(vo_, y11.y, yb[7]);
        if (STATS) {
          t1 = pk_add(t1, pk_add(pk_add(y00, y01), pk_add(y10, y11)));
          t2 = pk_fma(y00, y00, t2); t2 = pk_fma(y01, y01, t2); t2 = pk_fma(y10, y10, t2); t2 = pk_fma(y11, y11, t2);
        }
        __builtin_amdgcn_sched_barrier(0);                         // (the accumulator reads of the next pair stay here: register pressure)
      }
      s1 = t1.x + t1.y; s2 = t2.x + t2.y;
#undef WN_ST
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int t = 32 * mi + (r & 3) + 8 * (r >> 2) + 4 * half;
        const int tr = t / BW, tc = t - tr * BW;
        const int th = e_th0 + tr, tw = e_tw0 + tc;
        float s0[4], s1r[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float m0 = WP_AR(acc[0 + j][r]), m1 = WP_AR(acc[4 + j][r]), m2 = WP_AR(acc[8 + j][r]), m3 = WP_AR(acc[12 + j][r]);
          s0[j] = (m0 + m1) + m2;
          s1r[j] = (m1 - m2) - m3;
        }
        float y00 = (s0[0] + s0[1]) + s0[2] + bco, y01 = (s0[1] - s0[2]) - s0[3] + bco;
        float y10 = (s1r[0] + s1r[1]) + s1r[2] + bco, y11 = (s1r[1] - s1r[2]) - s1r[3] + bco;
        if (relu) { y00 = fmaxf(y00, 0.f); y01 = fmaxf(y01, 0.f); y10 = fmaxf(y10, 0.f); y11 = fmaxf(y11, 0.f); }
        __builtin_amdgcn_sched_barrier(0);
        if (th < TH && tw < TW) {
          float* yp = Y + (static_cast<size_t>(e_n * H + 2 * th) * W + 2 * tw) * Co + co;
          yp[0] = y00;
          yp[Co] = y01;
          yp[static_cast<size_t>(W) * Co] = y10;
          yp[static_cast<size_t>(W) * Co + Co] = y11;
          if (STATS) {
            s1 += (y00 + y01) + (y10 + y11);
            s2 = fmaf(y00, y00, s2); s2 = fmaf(y01, y01, s2); s2 = fmaf(y10, y10, s2); s2 = fmaf(y11, y11, s2);
          }
        }
      }
    }
    if (STATS) {
      float* red = sV + VBUF;                                      // [2 which][2 mi][64 channels]: V buffer 1 is idle until the next item's first k group
      s1 += __shfl_xor(s1, 32);
      s2 += __shfl_xor(s2, 32);
      if (half == 0) {
        red[(0 * 2 + mi) * 64 + 32 * ni + l31] = s1;
        red[(1 * 2 + mi) * 64 + 32 * ni + l31] = s2;
      }
      WP_BARRIER();
      if (tid < 128) {
        const int which = tid >> 6, c = tid & 63;
        partial[(static_cast<size_t>(e_tb) * 2 + which) * Co + 64 * e_cb + c] = red[(which * 2 + 0) * 64 + c] + red[(which * 2 + 1) * 64 + c];
      }
    }
    WP_STAMP(6);                                                  // output transform, stores, statistics
    if (late) {
      if (tid == 0) s_item = steal();
      __syncthreads();
      next = __builtin_amdgcn_readfirstlane(s_item);
      if (next >= 0) WP_START(next);
      fast_prev = false;                                          // (requests behind the stores: wait for everything)
    }
    item = next;
  }
  WP_DUMP();
#undef WP_BARRIER
#undef WP_AR
#undef WN_DREAD
#undef WN_TOP
#undef WN_VOUT
#undef WN_VCOL
#undef WN_VSTORE
#undef WN_DMA
#undef WN_DMA_PATCH
#undef WN_DMA_U
#undef WN_WAIT_VM
#undef WP_START
  // the last workgroup out leaves the counters at zero for the next launch that gets this set
  if (tid == 0) {
    __threadfence();
    if (atomicAdd(ctr + DBEV_NUM_XCD, 1) == static_cast<int>(gridDim.x) - 1) {
#pragma unroll
      for (int q = 0; q <= DBEV_NUM_XCD; ++q) atomicExch(ctr + q, 0);
    }
  }
}

// ---- the convolution, third version: two workgroups per CU ------------------------------------------------------------------------
// wino_fwd keeps 256 accumulators per lane: one wave per SIMD, one workgroup per CU -- whatever a workgroup does besides MFMAs (the
// barrier of every k group and the restart behind it: 10 % of its time; prologue, output transform and stores: ~10 us per workgroup;
// the dispatcher's turnover between workgroups: 11 % of a launch; the last, partly filled round) leaves the matrix pipe idle.  Here a
// wave keeps 8 of the 16 positions (rows 2 ph, 2 ph + 1; 128 accumulators) of 32 tiles x 32 channels, a workgroup (4 waves: 2 channel
// halves x 2 position halves) owns 32 tiles x 64 channels with 57 KB of LDS, and TWO workgroups share a CU: each SIMD holds two waves
// of different workgroups, and one's bubbles are the other's matrix time.  Per k group of 4 channels: 16 MFMAs per wave, the transform
// of the next group's 32 x 4 patch values as 256 tasks (tile, channel pair, V row: 8 ds_read_b64, 8 packed VALU, 4 ds_write_b64), the
// 16 KB of packed filters and the 4-channel patch stage by LDS-DMA.  The output transform needs both position halves: the ph = 1 wave
// hands its partial 2x2 outputs to its ph = 0 partner through LDS.
// ---- filter transform + packing for wino_fwd3 ---------------------------------------------------------------------------------------
// weight element (co, c, a, b) at co*so + c*sc + a*sa + b*sb (any strides: OIHW or channels-last).
// mode 0 (forward):        reduction index k = c,  output index j = co, taps g[a][b] = w[co][c][a][b]
// mode 1 (data gradient):  k = co, j = c, taps g[a][b] = w[co][c][2-a][2-b]
// U[((jb * (K/4) + kg) * 16 + p) * 2 + ni][lane][e] = (G g G^T)[p] for k = 4 kg + 2 (lane >> 5) + e, j = 64 jb + 32 ni + (lane & 31):
// the B operand of position p, channel half ni, for two MFMA steps of a lane, is one ds_read_b64 of an image that LDS-DMA copies linearly

__device__ __forceinline__ void wino_pack3_elem(long long idx, const float* __restrict__ w, long long so, long long sc, long long sa,
                                                long long sb, int K, int J, int mode, float* __restrict__ U) {
  const int nkg = K / 4;
  const long long total = static_cast<long long>(J / 64) * nkg * 256;
  if (idx >= total) return;
  const int e = idx & 1, lane = (idx >> 1) & 63, ni = (idx >> 7) & 1;
  const long long rest = idx >> 8;
  const int kg = static_cast<int>(rest % nkg), jb = static_cast<int>(rest / nkg);
  const int k = 4 * kg + 2 * (lane >> 5) + e, j = 64 * jb + 32 * ni + (lane & 31);
  const int co = mode ? k : j, c = mode ? j : k;
  float g[3][3];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      const int aa = mode ? 2 - a : a, bb = mode ? 2 - b : b;
      g[a][b] = w[co * so + c * sc + aa * sa + bb * sb];
    }
  float t[4][3];
#pragma unroll
  for (int b = 0; b < 3; ++b) {
    t[0][b] = g[0][b];
    t[1][b] = 0.5f * ((g[0][b] + g[2][b]) + g[1][b]);
    t[2][b] = 0.5f * ((g[0][b] + g[2][b]) - g[1][b]);
    t[3][b] = g[2][b];
  }
  float* out = U + ((static_cast<long long>(jb) * nkg + kg) * 16) * 256 + ni * 128 + lane * 2 + e;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float u0 = t[i][0], u1 = 0.5f * ((t[i][0] + t[i][2]) + t[i][1]), u2 = 0.5f * ((t[i][0] + t[i][2]) - t[i][1]), u3 = t[i][2];
    out[(4 * i + 0) * 256] = u0;
    out[(4 * i + 1) * 256] = u1;
    out[(4 * i + 2) * 256] = u2;
    out[(4 * i + 3) * 256] = u3;
  }
}

__global__ __launch_bounds__(256) void wino_filter_pack3(const float* __restrict__ w, long long so, long long sc, long long sa,
                                                         long long sb, int K, int J, int mode, float* __restrict__ U) {
  wino_pack3_elem(static_cast<long long>(blockIdx.x) * 256 + threadIdx.x, w, so, sc, sa, sb, K, J, mode, U);
}

// both directions of a layer in ONE launch (blockIdx.y = 0: forward filters, 1: data-gradient filters), each in the format of the
// forward kernel that direction gets (fmt 2: wino_fwd, second half of its buffer; 3: wino_fwd3, first half; 0: not wanted)
__global__ __launch_bounds__(256) void wino_filter_pack_pair(const float* __restrict__ w, long long so, long long sc, long long sa,
                                                             long long sb, int Cout, int Cin, int fmt_fwd, int fmt_dgrad,
                                                             float* __restrict__ Uf, float* __restrict__ Ud) {
  const int mode = blockIdx.y;
  const int fmt = mode ? fmt_dgrad : fmt_fwd;
  if (fmt == 0) return;
  const int K = mode ? Cout : Cin, J = mode ? Cin : Cout;
  float* U = mode ? Ud : Uf;
  const long long idx = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x;
  if (fmt == 2 || fmt == 6) wino_pack_elem(idx, w, so, sc, sa, sb, K, J, mode, U + 16LL * K * J);
  if (fmt == 3 || fmt == 6) wino_pack3_elem(idx, w, so, sc, sa, sb, K, J, mode, U);
}


template <int BH, int BW, bool STATS>
__global__ __launch_bounds__(256, 2) void wino_fwd3(const float* __restrict__ X, const float* __restrict__ U, const float* __restrict__ bias,
                                                    float* __restrict__ Y, float* __restrict__ partial, int N, int H, int W, int C,
                                                    int Co, int ntb, int dbg, int tb_first) {
  static_assert(BH * BW == 32, "32 tiles per workgroup");
  constexpr int PW = 2 * BW + 2, PH = 2 * BH + 2, NPIX = PW * PH;
  constexpr int NPC = (NPIX + 63) / 64;                    // patch DMA pieces: 64 pixels x 16 bytes (4 channels) = 1 KB
  constexpr int PBUF = NPC * 256;
  __shared__ __attribute__((aligned(16))) float smem[2 * PBUF + 2 * W3_VBUF + 2 * W3_UCHUNK];
  float* sP = smem;
  float* sV = smem + 2 * PBUF;
  float* sU = sV + 2 * W3_VBUF;
  const bool relu = (dbg >> 16) & 1;                             // launch flag: ReLU on the output (the folded eval-mode norm + ReLU)
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int ni = w & 1, ph = w >> 1, half = lane >> 5, l31 = lane & 31;
  const int ncb = Co / 64;
  const int L = xcd_block();
  const int cb = L % ncb, tb = L / ncb + tb_first;            // tb_first > 0: the tail of a layer whose full rounds wino_fwd ran
  if (tb >= ntb) return;
  const int TH = H >> 1, TW = W >> 1;
  const int NBW = (TW + BW - 1) / BW, NBH = (TH + BH - 1) / BH;
  const int bw = tb % NBW, bh = (tb / NBW) % NBH, n = tb / (NBW * NBH);
  const int th0 = bh * BH, tw0 = bw * BW;
  const int nkg = C / 4;
  const int wu = __builtin_amdgcn_readfirstlane(w);

  // patch DMA: piece j = pixels 64 j .. 64 j + 63, lane = pixel; wave w issues piece w (w < NPC).  Masked-out lanes (outside the
  // image / past the patch) leave the zeros written below in their LDS slots.
  bool pok;
  unsigned pvoff;
  {
    const int pix = 64 * wu + lane;
    const int pr = pix / PW, pc = pix - pr * PW;
    const int h = 2 * th0 - 1 + pr, x = 2 * tw0 - 1 + pc;
    pok = wu < NPC && pix < NPIX && h >= 0 && h < H && x >= 0 && x < W;
    pvoff = pok ? static_cast<unsigned>(((h * W + x) * C) * 4) : 0u;
  }
  const unsigned long long ximg = reinterpret_cast<unsigned long long>(X + static_cast<size_t>(n) * H * W * C);
  const unsigned long long ucb = reinterpret_cast<unsigned long long>(U + static_cast<size_t>(cb) * nkg * W3_UCHUNK);
  const unsigned lds0 = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(reinterpret_cast<size_t>((lds_ptr_t)smem)));
  const unsigned uvoff = lane * 16;
#define W3_DMA(voff_, sbase_, ldsaddr_)                                                                              \
  do {                                                                                                               \
    unsigned keep_;                                                                                                  \
    const unsigned m0v_ = __builtin_amdgcn_readfirstlane(ldsaddr_);                                                  \
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"  \
                 : "=&s"(keep_) : "v"(voff_), "s"(m0v_), "s"(sbase_) : "memory");                                    \
  } while (0)
#define W3_DMA_PATCH(st_, buf_)                                                                                      \
  do {                                                                                                               \
    if (pok) {                                                                                                       \
      const unsigned long long sb_ = ximg + static_cast<unsigned long long>(st_) * 16ull;                            \
      const unsigned la_ = lds0 + static_cast<unsigned>(((buf_) * PBUF + wu * 256) * 4);                             \
      W3_DMA(pvoff, sb_, la_);                                                                                       \
    }                                                                                                                \
  } while (0)
  // piece 4 w + i_ of the packed filters of k group kg_ -> sU[buf_]
#define W3_DMA_U(i_, kg_, buf_)                                                                                      \
  do {                                                                                                               \
    const unsigned long long sb_ = ucb + (static_cast<unsigned long long>(kg_) * W3_UCHUNK + (4 * wu + (i_)) * 256) * 4ull; \
    const unsigned la_ = lds0 + static_cast<unsigned>((2 * PBUF + 2 * W3_VBUF + (buf_) * W3_UCHUNK + (4 * wu + (i_)) * 256) * 4); \
    W3_DMA(uvoff, sb_, la_);                                                                                         \
  } while (0)

  // transform task: tile tt = tid >> 3, channel pair kq = (tid >> 2) & 1 of the stage, V row vi = tid & 3.
  // Row vi of B^T d = d[ra] + sg * d[rb] with (ra, rb, sg) = (0, 2, -), (1, 2, +), (2, 1, -), (1, 3, -)
  const int tt = tid >> 3, kq = (tid >> 2) & 1, vi = tid & 3;
  const int ttr = tt / BW, ttc = tt - ttr * BW;
  const int ra = vi == 0 ? 0 : vi == 2 ? 2 : 1, rb = vi == 0 ? 2 : vi == 1 ? 2 : vi == 2 ? 1 : 3;
  const float sgf = vi == 1 ? 1.f : -1.f;
  const floatx2 sg = {sgf, sgf};
  const int tsa = ((2 * ttr + ra) * PW + 2 * ttc) * 4 + 2 * kq, tsb = ((2 * ttr + rb) * PW + 2 * ttc) * 4 + 2 * kq;
  const int tdst = ((4 * vi) * 32 + tt) * 4 + 2 * kq;            // V[4 vi + j][tt][2 kq ..]: j adds 32 * 4
  // MFMA operands: A = V[p][l31][2 half ..], B = U[p][ni][lane][..], p = 8 ph + q
  const int aoff = ((8 * ph) * 32 + l31) * 4 + 2 * half;
  const int boff = ((8 * ph) * 2 + ni) * 128 + lane * 2;

  floatx16 acc[8];                                        // zeroed in the prologue, under the first DMA round trip

  floatx2 da[4], db[4], T[4];
#define W3_TRANSFORM_READ(ps_, b_) do { da[b_] = *reinterpret_cast<const floatx2*>((ps_) + tsa + (b_) * 4); db[b_] = *reinterpret_cast<const floatx2*>((ps_) + tsb + (b_) * 4); } while (0)
#define W3_VCOL(j_) ((j_) == 0 ? pk_sub(T[0], T[2]) : (j_) == 1 ? pk_add(T[1], T[2]) : (j_) == 2 ? pk_sub(T[2], T[1]) : pk_sub(T[1], T[3]))
#define W3_VSTORE(vd_, j_, v_) *reinterpret_cast<floatx2*>((vd_) + tdst + (j_) * 128) = (v_)
#define W3_VOUT(vd_, j_) W3_VSTORE(vd_, j_, W3_VCOL(j_))
  floatx2 vo[4];

  for (int i = tid; i < 2 * PBUF / 4; i += 256) reinterpret_cast<float4*>(sP)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  __syncthreads();
  W3_DMA_PATCH(0, 0);
  W3_DMA_PATCH(nkg > 1 ? 1 : 0, 1);
#pragma unroll
  for (int i = 0; i < 4; ++i) W3_DMA_U(i, 0, 0);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int p = 0; p < 8; ++p)
#pragma unroll
    for (int r = 0; r < 16; ++r) {                                // (as asm: plain assignments are sunk to the first use, behind the wait)
      float z;
      asm volatile("v_accvgpr_write_b32 %0, 0" : "=a"(z));
      acc[p][r] = z;
    }
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  {
#pragma unroll
    for (int b2 = 0; b2 < 4; ++b2) W3_TRANSFORM_READ(sP, b2);
#pragma unroll
    for (int b2 = 0; b2 < 4; ++b2) T[b2] = pk_fma(db[b2], sg, da[b2]);
#pragma unroll
    for (int j = 0; j < 4; ++j) W3_VOUT(sV, j);
  }
  __syncthreads();

  const int nkg_run = WN_DBG(2) ? 1 : nkg;
  for (int kg = 0; kg < nkg_run; ++kg) {
    const int cur = kg & 1, nxt = cur ^ 1;
    const int kgn = min(kg + 1, nkg - 1), stn = min(kg + 2, nkg - 1);       // past the end: redundant loads nobody reads
    const float* ps = sP + nxt * PBUF;                                     // patch stage kg + 1
    float* vd = sV + nxt * W3_VBUF;                                        // V(kg + 1)
    const floatx2* ap = reinterpret_cast<const floatx2*>(sV + cur * W3_VBUF + aoff);
    const floatx2* bp = reinterpret_cast<const floatx2*>(sU + cur * W3_UCHUNK + boff);
    floatx2 a2[2], b2v[2];
    a2[0] = ap[0]; b2v[0] = bp[0];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int sl = 2 * q + e;
        acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(e ? a2[q & 1].y : a2[q & 1].x, e ? b2v[q & 1].y : b2v[q & 1].x, acc[q], 0, 0, 0);
        if (e == 0 && q < 7) { a2[(q + 1) & 1] = ap[(q + 1) * 64]; b2v[(q + 1) & 1] = bp[(q + 1) * 128]; }
        if (sl < 4) W3_TRANSFORM_READ(ps, sl);
        if (sl == 6) {                                     // VALU work in two groups of 4, not one per slot (see wino_fwd)
#pragma unroll
          for (int q = 0; q < 4; ++q) T[q] = pk_fma(db[q], sg, da[q]);
        }
        if (sl == 8) {
#pragma unroll
          for (int q = 0; q < 4; ++q) vo[q] = W3_VCOL(q);
        }
        if (sl >= 9 && sl < 13) W3_VSTORE(vd, sl - 9, vo[sl - 9]);
        if (sl >= 1 && sl < 5) W3_DMA_U(sl - 1, kgn, nxt);
        if (sl == 5) W3_DMA_PATCH(stn, cur);                       // stage kg + 2 over stage kg (transformed during kg - 1)
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
#undef W3_DMA
#undef W3_DMA_PATCH
#undef W3_DMA_U
#undef W3_TRANSFORM_READ
#undef W3_VOUT
#undef W3_VCOL
#undef W3_VSTORE

  // ---- output transform.  This wave holds M[i][j] for i = 2 ph, 2 ph + 1.  Column sums of A^T M: s0[j] = m0j + m1j + m2j,
  // s1[j] = m1j - m2j - m3j: the ph = 0 wave contributes (m0j + m1j, m1j), the ph = 1 wave (m2j, -m2j - m3j); the row transform is
  // linear, so each wave forms its partial 2x2 outputs and the ph = 1 wave passes them to its partner through LDS.
  const int co = 64 * cb + 32 * ni + l31;
  float* xch = sU + ni * (16 * 4 * 64);                          // [r][4][64 lanes]: the filter buffers are idle now
  if (ph == 1) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float s0[4], s1[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) { s0[j] = acc[j][r]; s1[j] = -acc[j][r] - acc[4 + j][r]; }
      xch[(r * 4 + 0) * 64 + lane] = (s0[0] + s0[1]) + s0[2];
      xch[(r * 4 + 1) * 64 + lane] = (s0[1] - s0[2]) - s0[3];
      xch[(r * 4 + 2) * 64 + lane] = (s1[0] + s1[1]) + s1[2];
      xch[(r * 4 + 3) * 64 + lane] = (s1[1] - s1[2]) - s1[3];
    }
  }
  __syncthreads();
  float s1s = 0.f, s2s = 0.f;
  if (ph == 0) {
    const float bco = bias != nullptr ? bias[co] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int t = (r & 3) + 8 * (r >> 2) + 4 * half;
      const int tr = t / BW, tc = t - tr * BW;
      const int th = th0 + tr, tw = tw0 + tc;
      float s0[4], s1[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) { s0[j] = acc[j][r] + acc[4 + j][r]; s1[j] = acc[4 + j][r]; }
      float y00 = ((s0[0] + s0[1]) + s0[2]) + xch[(r * 4 + 0) * 64 + lane] + bco;
      float y01 = ((s0[1] - s0[2]) - s0[3]) + xch[(r * 4 + 1) * 64 + lane] + bco;
      float y10 = ((s1[0] + s1[1]) + s1[2]) + xch[(r * 4 + 2) * 64 + lane] + bco;
      float y11 = ((s1[1] - s1[2]) - s1[3]) + xch[(r * 4 + 3) * 64 + lane] + bco;
      if (relu) { y00 = fmaxf(y00, 0.f); y01 = fmaxf(y01, 0.f); y10 = fmaxf(y10, 0.f); y11 = fmaxf(y11, 0.f); }
      if (th < TH && tw < TW && !WN_DBG(1)) {
        float* yp = Y + (static_cast<size_t>(n * H + 2 * th) * W + 2 * tw) * Co + co;
        yp[0] = y00;
        yp[Co] = y01;
        yp[static_cast<size_t>(W) * Co] = y10;
        yp[static_cast<size_t>(W) * Co + Co] = y11;
        if (STATS) {
          s1s += (y00 + y01) + (y10 + y11);
          s2s = fmaf(y00, y00, s2s); s2s = fmaf(y01, y01, s2s); s2s = fmaf(y10, y10, s2s); s2s = fmaf(y11, y11, s2s);
        }
      }
    }
  }
  if (STATS) {
    float* red = sV;                                             // [2 which][64 channels of the block]
    s1s += __shfl_xor(s1s, 32);
    s2s += __shfl_xor(s2s, 32);
    if (ph == 0 && half == 0) {
      red[0 * 64 + 32 * ni + l31] = s1s;
      red[1 * 64 + 32 * ni + l31] = s2s;
    }
    __syncthreads();
    if (tid < 128) {
      const int which = tid >> 6, c = tid & 63;
      partial[(static_cast<size_t>(tb - tb_first) * 2 + which) * Co + 64 * cb + c] = red[which * 64 + c];
    }
  }
}

// ---- weight gradient ------------------------------------------------------------------------------------------------------------
// dU_p[c, co] = sum over tiles of V_p[tile, c] * Z_p[tile, co],  V = B^T d B (the forward's input transform), Z = A dY A^T (2x2 output
// gradient tile -> 4x4), then grad g = G^T dU G per (c, co): the adjoint of the forward in the filter, 16 multiplications per tile and
// channel pair instead of 36.  Per position a GEMM with the TILES as the reduction index: MFMA rows = input channels, columns = output
// channels, one step = two tiles (lanes 0-31 / 32-63).  A workgroup (4 waves, one per SIMD) owns 64 x 64 channels for all 16 positions
// (256 accumulators per lane, as in the forward) and a contiguous share of the image tiles, staged 2 x 8 tiles at a time (6 x 18 input
// pixels + 4 x 16 gradient pixels x 64 channels, double-buffered); both transforms are per-lane register arithmetic on plain
// ds_read_b32 rows (32 consecutive channels per half wave: conflict-free).  The per-share partial dU go to a workspace
// [share][position][co][c] and wino_wgrad_reduce adds them in share order (no atomics: bit-reproducible) and applies G^T . G.
constexpr int WG_SBH = 2, WG_SBW = 8;                                   // tiles per stage
constexpr int WG_PW = 2 * WG_SBW + 2, WG_PH = 2 * WG_SBH + 2;          // 18 x 6 input pixels
constexpr int WG_NPX = WG_PW * WG_PH;                                   // 108
constexpr int WG_NDY = (2 * WG_SBH) * (2 * WG_SBW);                     // 64 gradient pixels
constexpr int WG_STAGE = (WG_NPX + WG_NDY) * 64;                        // floats per stage buffer

__global__ __launch_bounds__(256, 1) void wino_wgrad(const float* __restrict__ X, const float* __restrict__ DY, float* __restrict__ part,
                                                     int N, int H, int W, int C, int Co, int nsb, int per, int nblk, int nsplit) {
  __shared__ __attribute__((aligned(16))) float smem[2 * WG_STAGE];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int ci = w & 1, coi = w >> 1, half = lane >> 5, l31 = lane & 31;
  const int L = xcd_block();
  const int blk = L % nblk, split = L / nblk;
  if (split >= nsplit) return;
  const int ncb = C / 64;
  const int c0 = (blk % ncb) * 64, o0 = (blk / ncb) * 64;
  const int TH = H >> 1, TW = W >> 1;
  const int NSW = (TW + WG_SBW - 1) / WG_SBW, NSH = (TH + WG_SBH - 1) / WG_SBH;
  const int sb_begin = split * per, sb_end = min(nsb, sb_begin + per);

  constexpr int NLX = (WG_NPX * 16 + 255) / 256, NLY = WG_NDY * 16 / 256;      // float4 per thread per stage: 7 + 4
  float4 rx[NLX], ry[NLY];
#define WG_LOAD(sb_)                                                                                                 \
  do {                                                                                                               \
    const int sbw_ = (sb_) % NSW, sbh_ = ((sb_) / NSW) % NSH, n_ = (sb_) / (NSW * NSH);                              \
    const int h0_ = 2 * WG_SBH * sbh_, w0_ = 2 * WG_SBW * sbw_;                                                      \
    _Pragma("unroll") for (int i = 0; i < NLX; ++i) {                                                                \
      const int f = tid + 256 * i, pix = f >> 4, q = f & 15;                                                         \
      const int pr = pix / WG_PW, pc = pix - pr * WG_PW;                                                             \
      const int h = h0_ - 1 + pr, x = w0_ - 1 + pc;                                                                  \
      const bool ok = pix < WG_NPX && h >= 0 && h < H && x >= 0 && x < W;                                            \
      rx[i] = ok ? *reinterpret_cast<const float4*>(X + (static_cast<size_t>(n_ * H + h) * W + x) * C + c0 + 4 * q)  \
                 : make_float4(0.f, 0.f, 0.f, 0.f);                                                                  \
    }                                                                                                                \
    _Pragma("unroll") for (int i = 0; i < NLY; ++i) {                                                                \
      const int f = tid + 256 * i, pix = f >> 4, q = f & 15;                                                         \
      const int h = h0_ + pix / (2 * WG_SBW), x = w0_ + pix % (2 * WG_SBW);                                          \
      ry[i] = (h < H && x < W) ? *reinterpret_cast<const float4*>(DY + (static_cast<size_t>(n_ * H + h) * W + x) * Co + o0 + 4 * q) \
                               : make_float4(0.f, 0.f, 0.f, 0.f);                                                    \
    }                                                                                                                \
  } while (0)
#define WG_STORE(buf_)                                                                                               \
  do {                                                                                                               \
    float* sx_ = smem + (buf_) * WG_STAGE;                                                                           \
    float* sy_ = sx_ + WG_NPX * 64;                                                                                  \
    _Pragma("unroll") for (int i = 0; i < NLX; ++i) {                                                                \
      const int f = tid + 256 * i;                                                                                   \
      if (f < WG_NPX * 16) *reinterpret_cast<float4*>(sx_ + 4 * f) = rx[i];                                          \
    }                                                                                                                \
    _Pragma("unroll") for (int i = 0; i < NLY; ++i) *reinterpret_cast<float4*>(sy_ + 4 * (tid + 256 * i)) = ry[i];   \
  } while (0)

  floatx16 acc[16];
#pragma unroll
  for (int p = 0; p < 16; ++p)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;

  // operands of one MFMA step (tiles 2 s + half of the stage): 16 input pixels, 4 gradient pixels, this lane's channel
  float d[4][4], g[2][2], V[16], Z[16], Vn[16], Zn[16];
#define WG_READ(sx_, sy_, s_)                                                                                        \
  do {                                                                                                               \
    const int t_ = 2 * (s_) + half, tr_ = t_ / WG_SBW, tc_ = t_ % WG_SBW;                                            \
    const float* px_ = (sx_) + ((2 * tr_) * WG_PW + 2 * tc_) * 64 + 32 * ci + l31;                                   \
    const float* py_ = (sy_) + ((2 * tr_) * (2 * WG_SBW) + 2 * tc_) * 64 + 32 * coi + l31;                           \
    _Pragma("unroll") for (int a = 0; a < 4; ++a)                                                                    \
      _Pragma("unroll") for (int b = 0; b < 4; ++b) d[a][b] = px_[(a * WG_PW + b) * 64];                             \
    _Pragma("unroll") for (int a = 0; a < 2; ++a)                                                                    \
      _Pragma("unroll") for (int b = 0; b < 2; ++b) g[a][b] = py_[(a * (2 * WG_SBW) + b) * 64];                      \
  } while (0)
#define WG_VROWS(DST, i0_, i1_)                                                                                      \
  _Pragma("unroll") for (int i = (i0_); i < (i1_); ++i) {                                                            \
    float t[4];                                                                                                      \
    _Pragma("unroll") for (int b = 0; b < 4; ++b)                                                                    \
      t[b] = i == 0 ? d[0][b] - d[2][b] : i == 1 ? d[1][b] + d[2][b] : i == 2 ? d[2][b] - d[1][b] : d[1][b] - d[3][b]; \
    DST[4 * i + 0] = t[0] - t[2]; DST[4 * i + 1] = t[1] + t[2]; DST[4 * i + 2] = t[2] - t[1]; DST[4 * i + 3] = t[1] - t[3]; \
  }
  // Z = A dY A^T, A = [[1,0],[1,1],[1,-1],[0,-1]]
#define WG_Z(DST)                                                                                                    \
  do {                                                                                                               \
    const float w_[4][2] = {{g[0][0], g[0][1]}, {g[0][0] + g[1][0], g[0][1] + g[1][1]},                              \
                            {g[0][0] - g[1][0], g[0][1] - g[1][1]}, {-g[1][0], -g[1][1]}};                          \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                                  \
      DST[4 * i + 0] = w_[i][0]; DST[4 * i + 1] = w_[i][0] + w_[i][1]; DST[4 * i + 2] = w_[i][0] - w_[i][1]; DST[4 * i + 3] = -w_[i][1]; \
    }                                                                                                                \
  } while (0)

  if (sb_begin < sb_end) {
    WG_LOAD(sb_begin);
    WG_STORE(0);
  }
  __syncthreads();
  int buf = 0;
  for (int sb = sb_begin; sb < sb_end; ++sb) {
    const bool more = sb + 1 < sb_end;
    if (more) WG_LOAD(sb + 1);
    const float* sx = smem + buf * WG_STAGE;
    const float* sy = sx + WG_NPX * 64;
    WG_READ(sx, sy, 0);
    WG_VROWS(V, 0, 4);
    WG_Z(Z);
#pragma unroll
    for (int s = 0; s < WG_SBH * WG_SBW / 2; ++s) {
      const bool nxt = s + 1 < WG_SBH * WG_SBW / 2;
#pragma unroll
      for (int q = 0; q < 4; ++q) {                       // the next step's operands are read / transformed under this step's MFMAs
        if (nxt) {
          if (q == 0) WG_READ(sx, sy, s + 1);
          if (q == 1) WG_VROWS(Vn, 0, 2);
          if (q == 2) WG_VROWS(Vn, 2, 4);
          if (q == 3) WG_Z(Zn);
        }
#pragma unroll
        for (int p = 4 * q; p < 4 * q + 4; ++p) acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[p], Z[p], acc[p], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (nxt) {
#pragma unroll
        for (int p = 0; p < 16; ++p) { V[p] = Vn[p]; Z[p] = Zn[p]; }
      }
    }
    if (more) WG_STORE(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }
#undef WG_LOAD
#undef WG_STORE
#undef WG_READ
#undef WG_VROWS
#undef WG_Z
  // accumulator register 4 q + r of position p = input channel c0 + 32 ci + 8 q + 4 half + r, output channel o0 + 32 coi + l31
  float* out = part + (static_cast<size_t>(split) * 16 * Co + o0 + 32 * coi + l31) * C + c0 + 32 * ci + 4 * half;
#pragma unroll
  for (int p = 0; p < 16; ++p)
#pragma unroll
    for (int q = 0; q < 4; ++q)
      *reinterpret_cast<float4*>(out + static_cast<size_t>(p) * Co * C + 8 * q) =
          make_float4(acc[p][4 * q + 0], acc[p][4 * q + 1], acc[p][4 * q + 2], acc[p][4 * q + 3]);
}

// ---- weight gradient, second version ---------------------------------------------------------------------------------------------
// Same GEMMs as wino_wgrad, built on what the forward kernel's measurements showed (a VALU instruction costs ~7 cycles of fp32 matrix
// time, LDS traffic / SALU / LDS-DMA issue cost none).  wino_wgrad spends ~100 VALU instructions per 16 MFMAs (every wave transforms
// its own operands, scalar arithmetic, register copies): 60 % matrix utilisation.  Here, per stage of 8 tiles (one tile row x 8):
//  * the input transform V is computed ONCE per workgroup (thread = channel pair x tile: 16 ds_read_b64, 32 v_pk_add_f32, 16
//    ds_write_b64) and handed to the waves through LDS as [position][tile][64 channels]; the A operand of a step is a ds_read_b32;
//  * the gradient transform Z = A dY A^T (cheap: 32 packed operations for 4 tiles) stays per lane, a row of positions at a time, the
//    tiles of two steps paired in one v_pk_add_f32;
//  * raw input rows (4 x 18 pixels x 64 channels) and gradient rows (2 x 16 x 64) arrive by scalar-base LDS-DMA two stages ahead;
//    pixels outside the image (and, in the last column block, which is clamped to the image and overlaps its neighbour, the gradient
//    columns that neighbour already covered) are replaced by zeros stored by the lanes the DMA skips.
// K order: step e of a stage multiplies tile e (lanes 0-31) and tile 4 + e (lanes 32-63).
constexpr int W2_T = 8;                                  // tiles per stage
constexpr int W2_PW = 2 * W2_T + 2;                      // 18 input columns, 4 rows
constexpr int W2_PPC = (4 * W2_PW + 3) / 4;              // patch DMA pieces of 4 pixels (1 KB): 18
constexpr int W2_PBUF = W2_PPC * 256;                    // floats per raw patch buffer
constexpr int W2_DPC = 2 * 2 * W2_T / 4;                 // gradient DMA pieces: 8
constexpr int W2_DBUF = W2_DPC * 256;                    // floats per raw gradient buffer
constexpr int W2_VBUF = 16 * W2_T * 64;                  // floats per transformed-input buffer

template <int ABL>
__global__ __launch_bounds__(256, 1) void wino_wgrad2(const float* __restrict__ X, const float* __restrict__ DY, float* __restrict__ part,
                                                      int N, int H, int W, int C, int Co, int nsb, int per, int nblk, int nsplit, int dbg) {
  __shared__ __attribute__((aligned(16))) float smem[2 * W2_PBUF + 3 * W2_DBUF + 2 * W2_VBUF];
  float* sP = smem;
  float* sD = smem + 2 * W2_PBUF;
  float* sV = sD + 3 * W2_DBUF;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int ci = w & 1, coi = w >> 1, half = lane >> 5, l31 = lane & 31;
  const int L = xcd_block();
  const int blk = L % nblk, split = L / nblk;
  if (split >= nsplit) return;
  const int ncb = C / 64;
  const int c0 = (blk % ncb) * 64, o0 = (blk / ncb) * 64;
  const int TH = H >> 1, TW = W >> 1;
  const int NSW = (TW + W2_T - 1) / W2_T;                  // column blocks per tile row (the last one clamped to TW - 8)
  const int ovl = NSW * W2_T - TW;                         // tiles of the last block that its left neighbour already covered
  const int sb_begin = split * per, sb_end = min(nsb, sb_begin + per);
  const int nst = sb_end - sb_begin;
  const int wu = __builtin_amdgcn_readfirstlane(w);
  const unsigned lds0 = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(reinterpret_cast<size_t>((lds_ptr_t)smem)));

  // ---- DMA lanes.  Patch piece j = pixels 4 j .. 4 j + 3 (row-major in the 4 x 18 rows), lane l = pixel 4 j + (l >> 4), channels
  // 4 (l & 15) .. + 3; wave w issues pieces w, w + 4, ...  flags: 1 top row, 2 bottom row, 4 left column, 8 right column -- the lane is
  // skipped (and stores zeros) when the stage's edge mask has one of its flags.
  unsigned pvoff[5], pflag[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const int pix = 4 * (wu + 4 * i) + (lane >> 4);
    const int pr = pix / W2_PW, pc = pix - pr * W2_PW;
    pvoff[i] = static_cast<unsigned>(((pr * W + pc) * C + 4 * (lane & 15)) * 4);
    pflag[i] = (pr == 0 ? 1u : 0u) | (pr == 3 ? 2u : 0u) | (pc == 0 ? 4u : 0u) | (pc == W2_PW - 1 ? 8u : 0u) |
               ((wu + 4 * i) >= W2_PPC || pix >= 4 * W2_PW ? 16u : 0u);
  }
  unsigned dvoff[2], dflag[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int pix = 4 * (wu + 4 * i) + (lane >> 4);
    const int pr = pix / (2 * W2_T), pc = pix - pr * (2 * W2_T);
    dvoff[i] = static_cast<unsigned>(((pr * W + pc) * Co + 4 * (lane & 15)) * 4);
    dflag[i] = pc < 2 * ovl ? 1u : 0u;
  }
  // stage geometry (scalar): n, th, column block cbk -> bases and edge masks.  Kept as counters, advanced without divisions.
#define W2_GEOM(n_, th_, cbk_, xb_, db_, pe_, de_)                                                                   \
  do {                                                                                                               \
    const int tw0_ = (cbk_) == NSW - 1 ? TW - W2_T : (cbk_) * W2_T;                                                  \
    xb_ = uniform64(reinterpret_cast<unsigned long long>(                                                            \
        X + (static_cast<long long>((n_) * H + 2 * (th_) - 1) * W + 2 * tw0_ - 1) * C + c0));                        \
    db_ = uniform64(reinterpret_cast<unsigned long long>(DY + (static_cast<long long>((n_) * H + 2 * (th_)) * W + 2 * tw0_) * Co + o0)); \
    pe_ = 16u | ((th_) == 0 ? 1u : 0u) | ((th_) == TH - 1 ? 2u : 0u) | (tw0_ == 0 ? 4u : 0u) | (tw0_ + W2_T == TW ? 8u : 0u); \
    de_ = ((cbk_) == NSW - 1 && ovl > 0) ? 1u : 0u;                                                                  \
  } while (0)
#define W2_ADVANCE(n_, th_, cbk_)                                                                                    \
  do {                                                                                                               \
    if (++(cbk_) == NSW) { (cbk_) = 0; if (++(th_) == TH) { (th_) = 0; ++(n_); } }                                   \
  } while (0)
  // patch piece i_ of the stage (base xb_, zero-lane masks pm[] from W2_MASKS) -> sP[buf_]; gradient piece -> sD[buf_].  Branch-free and
  // without VALU work at the issue site (a lone VALU instruction between two MFMAs costs ~17 cycles of matrix time): the lanes of
  // mask_ are taken out of EXEC for the load and store zeros instead; the masks of all seven pieces are formed in one group per stage.
  floatx4 zero4 = {0.f, 0.f, 0.f, 0.f};
  asm volatile("" : "+v"(zero4));                         // opaque: kept in four registers instead of being re-formed at every use
  const unsigned lanebase = lds0 + static_cast<unsigned>((w * 256 + lane * 4) * 4);
#define W2_DMA_M(voff_, sbase_, ldsaddr_, mask_, zaddr_, zoff_)                                                      \
  do {                                                                                                               \
    unsigned keep_;                                                                                                  \
    const unsigned m0v_ = __builtin_amdgcn_readfirstlane(ldsaddr_);                                                  \
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_andn2_b64 exec, exec, %4\n\tglobal_load_lds_dwordx4 %1, %3\n\t" \
                 "s_mov_b64 exec, %4\n\tds_write_b128 %5, %6 offset:%7\n\ts_mov_b64 exec, -1\n\ts_mov_b32 m0, %0"     \
                 : "=&s"(keep_) : "v"(voff_), "s"(m0v_), "s"(sbase_), "s"(mask_), "v"(zaddr_), "v"(zero4), "n"(zoff_) : "memory"); \
  } while (0)
#define W2_MASKS(pe_, de_, pm_, dm_)                                                                                 \
  do {                                                                                                               \
    _Pragma("unroll") for (int i = 0; i < 5; ++i) pm_[i] = __builtin_amdgcn_ballot_w64((pflag[i] & (pe_) & 15u) != 0u); \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) dm_[i] = __builtin_amdgcn_ballot_w64((dflag[i] & (de_)) != 0u);     \
  } while (0)
#define W2_DMA_PATCH(i_, xb_, pm_, buf_, zaddr_)                                                                     \
  do {                                                                                                               \
    if ((i_) < 4 || wu + 4 * (i_) < W2_PPC) {                                                                        \
      const unsigned la_ = lds0 + static_cast<unsigned>(((buf_) * W2_PBUF + (wu + 4 * (i_)) * 256) * 4);             \
      W2_DMA_M(pvoff[i_], xb_, la_, pm_[i_], zaddr_, (i_) * 4096);                                                   \
    }                                                                                                                \
  } while (0)
#define W2_DMA_DY(i_, db_, dm_, buf_, zaddr_)                                                                        \
  do {                                                                                                               \
    const unsigned la_ = lds0 + static_cast<unsigned>((2 * W2_PBUF + (buf_) * W2_DBUF + (wu + 4 * (i_)) * 256) * 4); \
    W2_DMA_M(dvoff[i_], db_, la_, dm_[i_], zaddr_, (i_) * 4096);                                                     \
  } while (0)
#define W2_ZP(buf_) (lanebase + static_cast<unsigned>((buf_) * W2_PBUF * 4))
#define W2_ZD(buf_) (lanebase + static_cast<unsigned>((2 * W2_PBUF + (buf_) * W2_DBUF) * 4))

  // ---- transform task: channel pair cp = tid & 31, tile tl = tid >> 5 of the stage
  const int cp = tid & 31, tl = tid >> 5;
  const int tsrc = (2 * tl) * 64 + 2 * cp;                 // float index of (row 0, column 2 tl, channel 2 cp) in a patch buffer
  const int tdst = tl * 64 + 2 * cp;                       // float index in a V buffer; position p adds 8 * 64
  // ---- MFMA operands: A = V[p][4 half + e][32 ci + l31]; Z from gradient pixels (rows 0 / 1, columns 2 tile, 2 tile + 1) of tiles
  // 4 half + e at channel 32 coi + l31
  const int aoff = (4 * half) * 64 + 32 * ci + l31;
  const int yoff = (2 * 4 * half) * 64 + 32 * coi + l31;

  floatx16 acc[16];                                        // zeroed in the prologue, under the first DMA round trip
  const unsigned long long t0c = WN_DBG(256) ? __builtin_amdgcn_s_memtime() : 0ull, t0r = WN_DBG(256) ? __builtin_amdgcn_s_memrealtime() : 0ull;

  floatx2 d[4][4], T[4][4];
#define W2_DREAD(ptr_, a_, b_) d[a_][b_] = *reinterpret_cast<const floatx2*>((ptr_) + ((a_) * W2_PW + (b_)) * 64)
#define W2_TOP(i_, b_) T[i_][b_] = (i_) == 0 ? pk_sub(d[0][b_], d[2][b_]) : (i_) == 1 ? pk_add(d[1][b_], d[2][b_]) : (i_) == 2 ? pk_sub(d[2][b_], d[1][b_]) : pk_sub(d[1][b_], d[3][b_])
#define W2_VCOL(i_, j_) \
  ((j_) == 0 ? pk_sub(T[i_][0], T[i_][2]) : (j_) == 1 ? pk_add(T[i_][1], T[i_][2]) : (j_) == 2 ? pk_sub(T[i_][2], T[i_][1]) : pk_sub(T[i_][1], T[i_][3]))
#define W2_VSTORE(dst_, i_, j_, v_) *reinterpret_cast<floatx2*>((dst_) + (4 * (i_) + (j_)) * (W2_T * 64)) = (v_)
#define W2_VOUT(dst_, i_, j_) W2_VSTORE(dst_, i_, j_, W2_VCOL(i_, j_))
  floatx2 vo[8];
  // gradient pixels of this lane's 4 tiles as two tile pairs q = 0, 1 (tiles 4 half + 2 q, + 1): y[a][b][q]
  floatx2 y[2][2][2], w1[2][2], w2[2][2];
  floatx2 z[2][4][2];                                       // rows of positions alternate between the two sets: z[row & 1][j][q]
#define W2_YREAD(ptr_)                                                                                               \
  _Pragma("unroll") for (int a = 0; a < 2; ++a)                                                                      \
    _Pragma("unroll") for (int b2 = 0; b2 < 2; ++b2)                                                                 \
      _Pragma("unroll") for (int q = 0; q < 2; ++q) {                                                                \
        y[a][b2][q].x = (ptr_)[((a * 2 * W2_T) + 2 * (2 * q) + b2) * 64];                                            \
        y[a][b2][q].y = (ptr_)[((a * 2 * W2_T) + 2 * (2 * q + 1) + b2) * 64];                                        \
      }
  // rows of A dY: w0 = y0, w1 = y0 + y1, w2 = y0 - y1, w3 = -y1; row i of Z: (a, a + b, a - b, -b) of (w_i[0], w_i[1]).  The two
  // negations are left out here (Z~[i][j] = s_i s_j Z[i][j], s_3 = -1): the reduce kernel flips the sign of those positions' sums
#define W2_ZROW(dst_, i_, q_)                                                                                        \
  do {                                                                                                               \
    const floatx2 a_ = (i_) == 0 ? y[0][0][q_] : (i_) == 1 ? w1[0][q_] : (i_) == 2 ? w2[0][q_] : y[1][0][q_];        \
    const floatx2 b_ = (i_) == 0 ? y[0][1][q_] : (i_) == 1 ? w1[1][q_] : (i_) == 2 ? w2[1][q_] : y[1][1][q_];        \
    dst_[0][q_] = a_; dst_[1][q_] = pk_add(a_, b_); dst_[2][q_] = pk_sub(a_, b_); dst_[3][q_] = b_;                  \
  } while (0)

  // ---- prologue: stages 0 and 1 of this share
  int n_a, th_a, cb_a;                                      // geometry counters of the next stage to FETCH
  {
    const int rows = TH * NSW;
    n_a = sb_begin / rows;
    const int r_ = sb_begin - n_a * rows;
    th_a = r_ / NSW; cb_a = r_ - th_a * NSW;
  }
  unsigned long long xb, db;
  unsigned pe, de;
  unsigned long long pm[5], dm[2];
  W2_GEOM(n_a, th_a, cb_a, xb, db, pe, de);
  W2_MASKS(pe, de, pm, dm);
#pragma unroll
  for (int i = 0; i < 5; ++i) W2_DMA_PATCH(i, xb, pm, 0, W2_ZP(0));
#pragma unroll
  for (int i = 0; i < 2; ++i) W2_DMA_DY(i, db, dm, 0, W2_ZD(0));
  if (nst > 1) W2_ADVANCE(n_a, th_a, cb_a);
  W2_GEOM(n_a, th_a, cb_a, xb, db, pe, de);
  W2_MASKS(pe, de, pm, dm);
#pragma unroll
  for (int i = 0; i < 5; ++i) W2_DMA_PATCH(i, xb, pm, 1, W2_ZP(1));
#pragma unroll
  for (int i = 0; i < 2; ++i) W2_DMA_DY(i, db, dm, 1, W2_ZD(1));
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int p = 0; p < 16; ++p)
#pragma unroll
    for (int r = 0; r < 16; ++r) {                                // (as asm: plain assignments are sunk to the first use, behind the wait)
      float z;
      asm volatile("v_accvgpr_write_b32 %0, 0" : "=a"(z));
      acc[p][r] = z;
    }
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  {
    const float* ps = sP + tsrc;
    float* vd = sV + tdst;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b2 = 0; b2 < 4; ++b2) W2_DREAD(ps, a, b2);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int b2 = 0; b2 < 4; ++b2) W2_TOP(i, b2);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) W2_VOUT(vd, i, j);
    const float* py = sD + yoff;
    W2_YREAD(py);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
#pragma unroll
      for (int b2 = 0; b2 < 2; ++b2) { w1[b2][q] = pk_add(y[0][b2][q], y[1][b2][q]); w2[b2][q] = pk_sub(y[0][b2][q], y[1][b2][q]); }
      W2_ZROW(z[0], 0, q);
    }
  }
  __syncthreads();

  for (int k = 0; k < nst; ++k) {
    const int cur = k & 1, nxt = cur ^ 1;
    // fetch counters: stage k + 2 (clamped to the share's last stage: a redundant load into buffers nobody reads any more)
    if (k + 2 < nst) W2_ADVANCE(n_a, th_a, cb_a);
    W2_GEOM(n_a, th_a, cb_a, xb, db, pe, de);
    W2_MASKS(pe, de, pm, dm);
    const int dnx = (k + 1) % 3, dft = (k + 2) % 3;
    const unsigned zp = W2_ZP(cur), zd = W2_ZD(dft);
    const float* ps = sP + nxt * W2_PBUF + tsrc;           // raw patch of stage k + 1
    float* vd = sV + nxt * W2_VBUF + tdst;                 // V(k + 1)
    const float* ap = sV + cur * W2_VBUF + aoff;
    const float* pyn = sD + dnx * W2_DBUF + yoff;          // gradient rows of stage k + 1
    float a4[2][4];                                        // A operands of a position; positions alternate between the two sets
#pragma unroll
    for (int e = 0; e < 4; ++e) a4[0][e] = ap[e * 64];
#pragma unroll
    for (int p = 0; p < 16; ++p) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int sl = 4 * p + e;
        const floatx2 zq = z[(p >> 2) & 1][p & 3][e >> 1];
        acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[p & 1][e], (e & 1) ? zq.y : zq.x, acc[p], 0, 0, 0);
        if (p < 15 && !(ABL & 1) && (e & 1) == 0) {        // two operands per LDS instruction (ds_read2st64_b32)
          a4[(p + 1) & 1][e] = ap[((p + 1) * W2_T + e) * 64];
          a4[(p + 1) & 1][e + 1] = ap[((p + 1) * W2_T + e + 1) * 64];
        }
        // VALU work in groups (profiles/r04_mfma_fillers.txt: a lone VALU instruction between two MFMAs costs ~17 cycles of matrix
        // time, one of a group of 8-16 costs ~6): raw reads one per slot, then all 16 row combinations at once, the column
        // combinations 8 at a time with their LDS stores spread over the following slots
        if (!(ABL & 2)) {
          if (sl >= 1 && sl < 17) { const int q = sl - 1; W2_DREAD(ps, q >> 2, q & 3); }
          if (sl == 20) {
#pragma unroll
            for (int q = 0; q < 16; ++q) W2_TOP(q >> 2, q & 3);
          }
          if (sl == 24 || sl == 34) {
#pragma unroll
            for (int q = 0; q < 8; ++q) vo[q] = W2_VCOL((sl == 24 ? 0 : 2) + (q >> 2), q & 3);
          }
          if (sl >= 25 && sl < 33) { const int q = sl - 25; W2_VSTORE(vd, q >> 2, q & 3, vo[q]); }
          if (sl >= 35 && sl < 43) { const int q = sl - 35; W2_VSTORE(vd, 2 + (q >> 2), q & 3, vo[q]); }
        }
        // the next row of positions: rows 1..3 of this stage under rows 0..2; row 0 of the next stage, from its gradient pixels,
        // under row 3
        if (!(ABL & 4)) {
          if ((p & 3) == 0 && p < 12 && e == 0) { W2_ZROW(z[((p >> 2) + 1) & 1], (p >> 2) + 1, 0); W2_ZROW(z[((p >> 2) + 1) & 1], (p >> 2) + 1, 1); }
          if (p == 12 && e == 0) { W2_YREAD(pyn); }
          if (p == 13 && e == 2) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
#pragma unroll
              for (int b2 = 0; b2 < 2; ++b2) { w1[b2][q] = pk_add(y[0][b2][q], y[1][b2][q]); w2[b2][q] = pk_sub(y[0][b2][q], y[1][b2][q]); }
              W2_ZROW(z[0], 0, q);
            }
          }
        }
        if (sl >= 2 && sl < 7 && !WN_DBG(32)) W2_DMA_PATCH(sl - 2, xb, pm, cur, zp);     // raw patch of stage k + 2 over the one transformed during k - 1
        if (sl >= 7 && sl < 9 && !WN_DBG(32)) W2_DMA_DY(sl - 7, db, dm, dft, zd);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (!WN_DBG(64)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (!WN_DBG(128)) __syncthreads();
  }
#undef W2_DMA_M
#undef W2_MASKS
#undef W2_ZP
#undef W2_ZD
#undef W2_GEOM
#undef W2_ADVANCE
#undef W2_DMA_PATCH
#undef W2_DMA_DY
#undef W2_DREAD
#undef W2_TOP
#undef W2_VOUT
#undef W2_VCOL
#undef W2_VSTORE
#undef W2_YREAD
#undef W2_ZROW
  // accumulator register 4 q + r of position p = input channel c0 + 32 ci + 8 q + 4 half + r, output channel o0 + 32 coi + l31
  float* out = part + (static_cast<size_t>(split) * 16 * Co + o0 + 32 * coi + l31) * C + c0 + 32 * ci + 4 * half;
#pragma unroll
  for (int p = 0; p < 16; ++p)
#pragma unroll
    for (int q = 0; q < 4; ++q)
      *reinterpret_cast<float4*>(out + static_cast<size_t>(p) * Co * C + 8 * q) =
          make_float4(acc[p][4 * q + 0], acc[p][4 * q + 1], acc[p][4 * q + 2], acc[p][4 * q + 3]);
  if (WN_DBG(256) && blk == 0 && split == nsplit - 1 && tid == 0) {     // ablation only: shader clocks / 100 MHz ticks of this workgroup
    const unsigned long long t1c = __builtin_amdgcn_s_memtime(), t1r = __builtin_amdgcn_s_memrealtime();
    unsigned* o = reinterpret_cast<unsigned*>(part + static_cast<size_t>(split) * 16 * Co * C);
    o[0] = static_cast<unsigned>(t1c - t0c); o[1] = static_cast<unsigned>(t1r - t0r); o[2] = static_cast<unsigned>(nst);
  }
}

// part [nsplit][16][Co][C]: the shares of every element are added in a fixed order (four interleaved chains, then pairwise) into
// share 0's slot -- one thread per (position, co, c), so that a 64 x 64 layer with hundreds of shares still fills the chip ...
// (round 5: folding this sum into wino_wgrad_reduce -- one thread per (co, c) walking 16 positions x nsplit shares -- was measured and
// reverted: 88 us instead of 14 + 5 us per layer; the sum needs the 16 x more threads of its own launch)
__global__ __launch_bounds__(256) void wino_wgrad_sum(float* __restrict__ part, int nsplit, long long plane16) {
  const long long idx = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x;
  if (idx >= plane16) return;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  const float* ps = part + idx;
  int s = 0;
  for (; s + 4 <= nsplit; s += 4) {
    s0 += ps[(s + 0) * plane16];
    s1 += ps[(s + 1) * plane16];
    s2 += ps[(s + 2) * plane16];
    s3 += ps[(s + 3) * plane16];
  }
  for (; s < nsplit; ++s) s0 += ps[s * plane16];
  part[idx] = (s0 + s1) + (s2 + s3);
}

// ... then grad_w (co, c, a, b) at co*so + c*sc + a*sa + b*sb = G^T dU G per (co, c), dU = share 0's slot [16][Co][C]
__global__ __launch_bounds__(256) void wino_wgrad_reduce(const float* __restrict__ part, int C, int Co, float* __restrict__ gw,
                                                         long long so, long long sc, long long sa, long long sb, int unsigned_z) {
  const long long idx = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x;
  if (idx >= static_cast<long long>(C) * Co) return;
  const int c = static_cast<int>(idx % C), co = static_cast<int>(idx / C);
  float u[16];
  const size_t plane = static_cast<size_t>(Co) * C;
  const float* ps = part + static_cast<size_t>(co) * C + c;
#pragma unroll
  for (int p = 0; p < 16; ++p) u[p] = ps[p * plane];
  if (unsigned_z) {                        // wino_wgrad2 accumulates s_i s_j dU (s_3 = -1): positions (i, 3) and (3, j), i, j < 3
    u[3] = -u[3]; u[7] = -u[7]; u[11] = -u[11]; u[12] = -u[12]; u[13] = -u[13]; u[14] = -u[14];
  }
  float t[3][4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float hs = 0.5f * (u[4 + j] + u[8 + j]), hd = 0.5f * (u[4 + j] - u[8 + j]);
    t[0][j] = u[j] + hs;
    t[1][j] = hd;
    t[2][j] = hs + u[12 + j];
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float hs = 0.5f * (t[a][1] + t[a][2]), hd = 0.5f * (t[a][1] - t[a][2]);
    float* o = gw + co * so + c * sc + a * sa;
    o[0] = t[a][0] + hs;
    o[sb] = hd;
    o[2 * sb] = hs + t[a][3];
  }
}

struct WinoWgPlan { int nsb, nblk, nsplit, per, grid, v2; };

int wino_wg_version() { static const int v = getenv("DBEV_WINO_WGRAD_V") ? atoi(getenv("DBEV_WINO_WGRAD_V")) : 2; return v; }

bool wino_wg_plan(int N, int H, int W, int C, int Co, WinoWgPlan* p) {
  if (N <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1) || C <= 0 || (C % 64) || Co <= 0 || (Co % 64)) return false;
  if (static_cast<long long>(N) * H * W * (C > Co ? C : Co) >= 0x7fffffffLL) return false;
  const int TH = H / 2, TW = W / 2;
  // second version: stages of 1 x 8 tiles, byte offsets inside an image as 32-bit LDS-DMA offsets
  p->v2 = wino_wg_version() >= 2 && TW >= W2_T && static_cast<long long>(H + 2) * W * (C > Co ? C : Co) * 4 < 0x7fffffffLL;
  const long long nsb = p->v2 ? static_cast<long long>(N) * TH * ((TW + W2_T - 1) / W2_T)
                              : static_cast<long long>(N) * ((TH + WG_SBH - 1) / WG_SBH) * ((TW + WG_SBW - 1) / WG_SBW);
  if (nsb > 0x3fffffffLL) return false;
  p->nsb = static_cast<int>(nsb);
  p->nblk = (C / 64) * (Co / 64);
  // shares of the tile range: ONE round of the 256 CUs (rounded down: never a sliver of a second round); every share costs a
  // 16 x Cin x Cout partial written and read back
  long long ns = DBEV_NUM_CU / p->nblk;
  if (ns > nsb) ns = nsb;
  if (ns < 1) ns = 1;
  p->per = static_cast<int>((nsb + ns - 1) / ns);
  p->nsplit = static_cast<int>((nsb + p->per - 1) / p->per);
  p->grid = dbev_round_xcd(p->nblk * p->nsplit);
  return true;
}

std::atomic<unsigned> g_wq_seq{0};            // launches of wino_fwdp take the counter sets of g_wino_queue in turn

int wino_dbg() { static const int v = getenv("DBEV_WINO_DBG") ? atoi(getenv("DBEV_WINO_DBG")) : 0; return v; }

struct WinoPlan {
  int bh, bw, ntb, ncb, grid, v3;
  int persist;                                 // wino_fwdp: one workgroup per CU walking all (tile block, channel block) items
  int n2;                                      // > 0: hybrid -- wino_fwd runs tile blocks [0, n2) (whole rounds), wino_fwd3 the rest
  int bh3, bw3, tb3_first, ntb3, grid3;        // the wino_fwd3 part of a hybrid launch (its 32-tile blocks from tb3_first on)
};

// DBEV_WINO_FWD_V = 2 / 3 forces one forward kernel (A/B runs); default 0: chosen per layer
int wino_fwd_version() { static const int v = getenv("DBEV_WINO_FWD_V") ? atoi(getenv("DBEV_WINO_FWD_V")) : 0; return v; }

bool wino_plan(int N, int H, int W, int C, int Co, WinoPlan* p) {
  if (N <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1) || C <= 0 || (C % 4) || Co <= 0 || (Co % 64)) return false;
  if (static_cast<long long>(N) * H * W * (C > Co ? C : Co) >= 0x7fffffffLL) return false;      // 32-bit element offsets
  // 32-bit BYTE offsets inside an image: the input side (LDS-DMA) and the output side (store offsets of the fast path, and the
  // data-gradient launch where the roles of C and Co swap) -- mirrors wino.eligible() for direct C-ABI callers
  if (static_cast<long long>(H) * W * (C > Co ? C : Co) * 4 >= 0x7fffffffLL) return false;
  const int TH = H / 2, TW = W / 2;
  p->ncb = Co / 64;
  // 64-tile blocks of wino_fwd: 8 x 8 or 4 x 16, whichever wastes fewer tile slots at the image edges
  const long long w88 = static_cast<long long>((TH + 7) / 8) * ((TW + 7) / 8), w416 = static_cast<long long>((TH + 3) / 4) * ((TW + 15) / 16);
  const long long nb64 = (w416 < w88 ? w416 : w88) * N;
  // Which kernel (measured, profiles/r04_wino_vs_miopen.txt): wino_fwd (64-tile items, one per CU) is 10-17 % faster per tile once the
  // chip is full; wino_fwd3 (32-tile items, two per CU) wins where wino_fwd would leave CUs idle: under ~200 items (64 items: 98 vs
  // 129 us, 128: 55 vs 69), and when a second round would be at most half full (384 items: 247 vs 264 us)
  const int ver = wino_fwd_version();                      // 2 / 3 / 4 force wino_fwd / wino_fwd3 / wino_fwdp (tests, A/B runs)
  const long long items = nb64 * p->ncb;
  p->v3 = ver == 3 || (C % 8) != 0 || (ver != 2 && ver != 4 && (items <= 192 || (items > 256 && items <= 384)));
  if (p->v3) {                                   // 32-tile blocks: 4 x 8 or 2 x 16
    const long long w48 = static_cast<long long>((TH + 3) / 4) * ((TW + 7) / 8), w216 = static_cast<long long>((TH + 1) / 2) * ((TW + 15) / 16);
    if (w216 < w48) { p->bh = 2; p->bw = 16; p->ntb = static_cast<int>(w216) * N; }
    else { p->bh = 4; p->bw = 8; p->ntb = static_cast<int>(w48) * N; }
  } else if (w416 < w88) { p->bh = 4; p->bw = 16; p->ntb = static_cast<int>(w416) * N; }
  else { p->bh = 8; p->bw = 8; p->ntb = static_cast<int>(w88) * N; }
  const long long g = static_cast<long long>(p->ntb) * p->ncb;
  if (g > 0x3fffffffLL) return false;
  p->grid = dbev_round_xcd(static_cast<int>(g));
  // Hybrid: a layer whose last round of 64-tile items would be at most half full (2.25 rounds: the 48-image ResNet stages) pays a whole
  // round for it.  wino_fwd then runs only the whole rounds (tile blocks [0, n2), n2 a multiple of a block row), and the remaining tile
  // rows go to wino_fwd3 as 32-tile items, two per CU: <= 256 of them take ~0.58 of a round, <= 512 ~1.16 (profiles/r04_wino_vs_miopen.txt).
  p->n2 = 0;
  // Persistent workgroups (round 6, wino_fwdp): no workgroup turnover, the next item's first stages requested under this item's output
  // transform, the last round shared by all CUs.  MEASURED NEUTRAL TO SLIGHTLY SLOWER (profiles/r06_wino_persistent.txt: per layer
  // -3 ... +10 %, the step 101.0 -> 101.4 ms): what it removes (turnover, the prologue's round trip) it pays back in item hand-over
  // (~12 k shader cycles per item against ~10 k of prologue + epilogue + turnover before), and the kernel runs at the clock the power
  // limit leaves (2.15 GHz in the step, measured by this kernel's own stamps) either way.  Off by default; DBEV_WINO_PERSIST=1 /
  // DBEV_WINO_FWD_V=4 select it (tests, the clock probe tools/wino_clock_in_step.py).
  static const int persist_on = getenv("DBEV_WINO_PERSIST") ? atoi(getenv("DBEV_WINO_PERSIST")) : 0;
  p->persist = !p->v3 && ((ver == 0 && persist_on && items > DBEV_NUM_CU) || ver == 4);
  if (p->persist) {
    p->grid = DBEV_NUM_CU;
    return true;
  }
  static const int hybrid_on = getenv("DBEV_WINO_HYBRID") ? atoi(getenv("DBEV_WINO_HYBRID")) : 1;
  if (!p->v3 && ver == 0 && hybrid_on && (TH % p->bh) == 0) {
    const int nbw = (TW + p->bw - 1) / p->bw;
    const long long n2 = (items / DBEV_NUM_CU * DBEV_NUM_CU / p->ncb) / nbw * nbw;
    const long long rem = nb64 - n2, halves = 2 * rem * p->ncb;
    if (n2 > 0 && rem > 0 && halves <= 2 * DBEV_NUM_CU) {
      const double t_v2 = static_cast<double>((items + DBEV_NUM_CU - 1) / DBEV_NUM_CU);
      const double t_hy = static_cast<double>((n2 * p->ncb + DBEV_NUM_CU - 1) / DBEV_NUM_CU) + (halves <= DBEV_NUM_CU ? 0.58 : 1.16) + 0.06;
      if (t_hy < 0.92 * t_v2) {              // measured: the model's 12 % at 2.25 rounds is 4 % in the layer, its 4 % at 8.25 rounds nothing
        p->n2 = static_cast<int>(n2);
        p->bh3 = p->bh / 2; p->bw3 = p->bw;
        p->tb3_first = 2 * p->n2;                                  // block rows double, the row-major order of the blocks is kept
        p->ntb3 = (TH / p->bh3) * nbw * N;
        p->grid3 = dbev_round_xcd(static_cast<int>((p->ntb3 - p->tb3_first) * p->ncb));
        p->grid = dbev_round_xcd(p->n2 * p->ncb);
        p->ntb = p->n2;                                            // wino_fwd's bound: its blocks only
      }
    }
  }
  return true;
}

}  // namespace

#ifdef DBEV_WINO_ABLATE
extern "C" int dbev_wino_prof_read(unsigned long long* out16) {      // dev builds only: not part of the ABI
  return hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_wp_prof), sizeof(unsigned long long) * 16) == hipSuccess ? 0 : -1;
}
#endif

extern "C" long long dbev_wino_filter_floats(int K, int J) {
  if (K <= 0 || J <= 0 || (K % 4) || (J % 64)) return 0;
  // both forward kernels' formats, one after the other (the second only when wino_fwd can take the layer: K % 8 == 0)
  return static_cast<long long>(J / 64) * (K / 4) * W3_UCHUNK * ((K % 8) == 0 ? 2 : 1);
}

extern "C" int dbev_wino_filter_pack(const float* weight, long long so, long long sc, long long sa, long long sb, int Cout, int Cin,
                                     int flags, float* packed, dbevStream_t stream) {
  const int data_gradient = flags & 1;
  const int K = data_gradient ? Cout : Cin, J = data_gradient ? Cin : Cout;
  const long long n = dbev_wino_filter_floats(K, J);
  if (n == 0 || weight == nullptr || packed == nullptr) return DBEV_EINVAL;
  const long long one = 16LL * K * J;                       // floats of one format
  const int only = (flags >> 1) & 3;                        // 0: both formats, 1: wino_fwd3's only, 2: wino_fwd's only
  if (only != 2)
    hipLaunchKernelGGL(wino_filter_pack3, dim3(dbev_ceil_div(one / 16, 256)), dim3(256), 0, dbev_stream(stream), weight, so, sc, sa, sb,
                       K, J, data_gradient, packed);
  if ((K % 8) == 0 && only != 1)
    hipLaunchKernelGGL(wino_filter_pack, dim3(dbev_ceil_div(one / 16, 256)), dim3(256), 0, dbev_stream(stream), weight, so, sc, sa, sb, K,
                       J, data_gradient, packed + one);
  DBEV_LAUNCH_CHECK();
  return 0;
}

extern "C" int dbev_wino_filter_pack_pair(const float* weight, long long so, long long sc, long long sa, long long sb, int Cout,
                                          int Cin, int fwd_kernel, int dgrad_kernel, float* packed_fwd, float* packed_dgrad,
                                          dbevStream_t stream) {
  // fwd_kernel / dgrad_kernel: what dbev_wino_conv3x3_forward_kernel returns for the layer each pack will be applied to (2 / 3), 0: skip
  const auto known = [](int k) { return k == 2 || k == 3 || k == 6; };             // 6: both formats (a hybrid launch)
  const bool okf = fwd_kernel == 0 || (known(fwd_kernel) && packed_fwd != nullptr && dbev_wino_filter_floats(Cin, Cout) > 0 &&
                                       (fwd_kernel == 3 || (Cin % 8) == 0));
  const bool okd = dgrad_kernel == 0 || (known(dgrad_kernel) && packed_dgrad != nullptr && dbev_wino_filter_floats(Cout, Cin) > 0 &&
                                         (dgrad_kernel == 3 || (Cout % 8) == 0));
  if (weight == nullptr || !okf || !okd || (fwd_kernel == 0 && dgrad_kernel == 0)) return DBEV_EINVAL;
  const long long threads = static_cast<long long>(Cin) * Cout;                       // one per (reduction, output) channel pair
  hipLaunchKernelGGL(wino_filter_pack_pair, dim3(dbev_ceil_div(threads, 256), 2), dim3(256), 0, dbev_stream(stream), weight, so, sc, sa,
                     sb, Cout, Cin, fwd_kernel, dgrad_kernel, packed_fwd, packed_dgrad);
  DBEV_LAUNCH_CHECK();
  return 0;
}

// every trainable layer's filters in ONE launch (round 5): blockIdx.z = job of a device table of dbevPackJob (kind_a / kind_b = the
// forward / data-gradient kernel codes of dbev_wino_filter_pack_pair, 0: skip; out_a / out_b = its two pack buffers)
__global__ __launch_bounds__(256) void wino_filter_pack_multi(const dbevPackJob* __restrict__ jobs) {
  const dbevPackJob j = jobs[blockIdx.z];
  const int mode = blockIdx.y;
  const int fmt = mode ? j.kind_b : j.kind_a;
  if (fmt == 0) return;
  const int K = mode ? j.Cout : j.Cin, J = mode ? j.Cin : j.Cout;
  float* U = static_cast<float*>(mode ? j.out_b : j.out_a);
  const long long idx = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x;      // (the element kernels return past their own range)
  if (fmt == 2 || fmt == 6) wino_pack_elem(idx, j.weight, j.so, j.sc, j.sa, j.sb, K, J, mode, U + 16LL * K * J);
  if (fmt == 3 || fmt == 6) wino_pack3_elem(idx, j.weight, j.so, j.sc, j.sa, j.sb, K, J, mode, U);
}

extern "C" int dbev_wino_filter_pack_multi(const dbevPackJob* jobs_device, int n_jobs, long long max_pairs, dbevStream_t stream) {
  if (jobs_device == nullptr || n_jobs <= 0 || n_jobs > 65535 || max_pairs <= 0) return DBEV_EINVAL;
  hipLaunchKernelGGL(wino_filter_pack_multi, dim3(dbev_ceil_div(max_pairs, 256), 2, n_jobs), dim3(256), 0, dbev_stream(stream), jobs_device);
  DBEV_LAUNCH_CHECK();
  return 0;
}

extern "C" int dbev_wino_conv3x3_forward_kernel(int N, int H, int W, int Cin, int Cout) {
  WinoPlan p;
  return wino_plan(N, H, W, Cin, Cout, &p) ? (p.v3 ? 3 : p.n2 > 0 ? 6 : 2) : 0;
}

extern "C" int dbev_wino_conv3x3_stats_rows(int N, int H, int W, int Cin, int Cout) {
  WinoPlan p;
  return wino_plan(N, H, W, Cin, Cout, &p) ? (p.n2 > 0 ? p.n2 + (p.ntb3 - p.tb3_first) : p.ntb) : 0;
}

extern "C" int dbev_wino_conv3x3_forward_act(const float* x_nhwc, const float* packed, const float* bias, float* y_nhwc,
                                             float* stats_partial, int N, int H, int W, int Cin, int Cout, int relu,
                                             dbevStream_t stream) {
  WinoPlan p;
  if (!wino_plan(N, H, W, Cin, Cout, &p) || x_nhwc == nullptr || packed == nullptr || y_nhwc == nullptr) return DBEV_EINVAL;
  hipStream_t s = dbev_stream(stream);
  // the kernel log's "bytes" field carries the launch's algorithmic FLOPs here: the Winograd-domain products, 2 * 16 per tile and
  // channel pair (the direct convolution's 2 * 36 per tile are what it replaces)
  DbevKt kt(DBEV_K_WINO_FWD, 32LL * N * (H / 2) * (W / 2) * Cin * Cout, s);
#define WN_GO(BHV, BWV, ST)                                                                                                       \
  hipLaunchKernelGGL((wino_fwd<BHV, BWV, ST>), dim3(p.grid), dim3(256), 0, s, x_nhwc, packed, bias, y_nhwc, stats_partial, N, H, W, \
                     Cin, Cout, p.ntb, wino_dbg() | (relu ? 1 << 16 : 0))
#define WN_GO3(BHV, BWV, ST)                                                                                                      \
  hipLaunchKernelGGL((wino_fwd3<BHV, BWV, ST>), dim3(p.grid), dim3(256), 0, s, x_nhwc, packed, bias, y_nhwc, stats_partial, N, H, W, \
                     Cin, Cout, p.ntb, wino_dbg() | (relu ? 1 << 16 : 0), 0)
#define WN_GOP(BHV, BWV, ST)                                                                                                      \
  hipLaunchKernelGGL((wino_fwdp<BHV, BWV, ST>), dim3(p.grid), dim3(256), 0, s, x_nhwc, packed, bias, y_nhwc, stats_partial, N, H, W, \
                     Cin, Cout, p.ntb, wino_dbg() | (relu ? 1 << 16 : 0), static_cast<int>(g_wq_seq.fetch_add(1) % WQ_SETS))
#define WN_GO3T(BHV, BWV, ST)                     /* the tail of a hybrid launch: blocks tb3_first.. , statistics rows behind wino_fwd's */ \
  hipLaunchKernelGGL((wino_fwd3<BHV, BWV, ST>), dim3(p.grid3), dim3(256), 0, s, x_nhwc, packed, bias, y_nhwc,                        \
                     stats_partial != nullptr ? stats_partial + static_cast<size_t>(p.n2) * 2 * Cout : nullptr, N, H, W, Cin, Cout,  \
                     p.ntb3, wino_dbg() | (relu ? 1 << 16 : 0), p.tb3_first)
  if (p.v3) {
    if (p.bw == 8) { if (stats_partial != nullptr) WN_GO3(4, 8, true); else WN_GO3(4, 8, false); }
    else { if (stats_partial != nullptr) WN_GO3(2, 16, true); else WN_GO3(2, 16, false); }
  } else if (p.persist) {
    packed += 16LL * Cin * Cout;                            // wino_fwd's format: the second half of the packed buffer
    if (p.bw == 8) { if (stats_partial != nullptr) WN_GOP(8, 8, true); else WN_GOP(8, 8, false); }
    else { if (stats_partial != nullptr) WN_GOP(4, 16, true); else WN_GOP(4, 16, false); }
  } else {
    if (p.n2 > 0) {                                         // hybrid: the rows wino_fwd leaves (launched first: it owns fewer CUs per item)
      if (p.bw3 == 8) { if (stats_partial != nullptr) WN_GO3T(4, 8, true); else WN_GO3T(4, 8, false); }
      else { if (stats_partial != nullptr) WN_GO3T(2, 16, true); else WN_GO3T(2, 16, false); }
    }
    packed += 16LL * Cin * Cout;                            // the second format of the packed buffer
    if (p.bw == 8) { if (stats_partial != nullptr) WN_GO(8, 8, true); else WN_GO(8, 8, false); }
    else { if (stats_partial != nullptr) WN_GO(4, 16, true); else WN_GO(4, 16, false); }
  }
#undef WN_GO
#undef WN_GO3
#undef WN_GO3T
#undef WN_GOP
  DBEV_LAUNCH_CHECK();
  return 0;
}

extern "C" int dbev_wino_conv3x3_forward(const float* x_nhwc, const float* packed, const float* bias, float* y_nhwc,
                                         float* stats_partial, int N, int H, int W, int Cin, int Cout, dbevStream_t stream) {
  return dbev_wino_conv3x3_forward_act(x_nhwc, packed, bias, y_nhwc, stats_partial, N, H, W, Cin, Cout, 0, stream);
}

extern "C" size_t dbev_wino_conv3x3_backward_weight_workspace_bytes(int N, int H, int W, int Cin, int Cout) {
  WinoWgPlan p;
  if (!wino_wg_plan(N, H, W, Cin, Cout, &p)) return 0;
  return static_cast<size_t>(p.nsplit) * 16 * Cin * Cout * sizeof(float);
}

extern "C" int dbev_wino_conv3x3_backward_weight(const float* x_nhwc, const float* grad_y_nhwc, float* grad_weight, long long so,
                                                 long long sc, long long sa, long long sb, int N, int H, int W, int Cin, int Cout,
                                                 void* workspace, size_t workspace_bytes, dbevStream_t stream) {
  WinoWgPlan p;
  if (!wino_wg_plan(N, H, W, Cin, Cout, &p) || x_nhwc == nullptr || grad_y_nhwc == nullptr || grad_weight == nullptr ||
      workspace == nullptr || workspace_bytes < static_cast<size_t>(p.nsplit) * 16 * Cin * Cout * sizeof(float))
    return DBEV_EINVAL;
  hipStream_t s = dbev_stream(stream);
  float* part = static_cast<float*>(workspace);
  DbevKt kt(DBEV_K_WINO_WGRAD, 32LL * N * (H / 2) * (W / 2) * Cin * Cout, s);      // whole entry (GEMMs + share sum + G^T . G)
  if (p.v2) {
#define W2_GO(A_) hipLaunchKernelGGL(wino_wgrad2<A_>, dim3(p.grid), dim3(256), 0, s, x_nhwc, grad_y_nhwc, part, N, H, W, Cin, Cout, p.nsb, p.per, p.nblk, p.nsplit, wino_dbg())
#ifdef DBEV_WINO_ABLATE                                     // dev builds: DBEV_WINO_DBG bits 9-11 drop the operand reads / the transform / Z
    switch ((wino_dbg() >> 9) & 7) {
      case 1: W2_GO(1); break; case 2: W2_GO(2); break; case 4: W2_GO(4); break; case 7: W2_GO(7); break; case 6: W2_GO(6); break;
      default: W2_GO(0);
    }
#else
    W2_GO(0);
#endif
#undef W2_GO
  } else
    hipLaunchKernelGGL(wino_wgrad, dim3(p.grid), dim3(256), 0, s, x_nhwc, grad_y_nhwc, part, N, H, W, Cin, Cout, p.nsb, p.per, p.nblk,
                       p.nsplit);
  DBEV_LAUNCH_CHECK();
  const long long plane16 = 16LL * Cin * Cout;
  if (p.nsplit > 1) {
    hipLaunchKernelGGL(wino_wgrad_sum, dim3(dbev_ceil_div(plane16, 256)), dim3(256), 0, s, part, p.nsplit, plane16);
    DBEV_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(wino_wgrad_reduce, dim3(dbev_ceil_div(static_cast<long long>(Cin) * Cout, 256)), dim3(256), 0, s, part, Cin, Cout,
                     grad_weight, so, sc, sa, sb, p.v2);
  DBEV_LAUNCH_CHECK();
  return 0;
}
