// fp32 GEMMs of the 1x1 convolutions on the BF16 matrix cores, at fp32 accuracy ("bf16x6").
//
// Every fp32 operand is split into three bf16 values, x = x0 + x1 + x2 (x0 = the upper 16 bits of x, x1 those of the exact remainder
// x - x0, x2 those of x - x0 - x1: 24 mantissa bits in three 8-bit pieces), and a product a * b is assembled from the six partial products
// a_i * b_j with i + j <= 2 -- each one EXACT in fp32 (8 x 8 bits), accumulated in fp32 by v_mfma_f32_32x32x16_bf16; the three dropped
// products are below 2^-23 |a b|.  Measured (tools/bf16x6_error.py, tests/test_gpu_gemm_bf6.py): max |y - fp64| / max |y| = 1.5-3e-7, at
// or below a plain fp32 GEMM's 2-6e-7.  Why: on gfx950 the bf16 matrix pipe sustains 2 457 TFLOP/s against 155.6 for
// v_mfma_f32_32x32x2_f32 (tools/mfma_bf16_bench.hip: 32.8 vs 64.7 cycles per instruction of 16 vs 2 reduction steps), so six bf16
// instructions do the work of eight fp32 ones in 0.38 of the time, and -- unlike under the fp32 instruction, which occupies the SIMD's
// fp32 lanes -- VALU work issued between bf16 MFMAs costs ~2 cycles per instruction, so the split of the activation operand can be done
// on the fly (4.5 VALU operations per element, amortised over the tile's columns).  The weights are split once per step by b6_pack.
//
//   Y[M, N] = X[M, K] * W[N, K]^T     X rows of x_stride floats, Y row-major, M % 128 == 0, K % 64 == 0, N % 64 == 0
//
// Forward / data gradient (b6_fwd): workgroup = 4 waves, tile 128 rows x BN columns (BN = 128: waves 2 x 2, wave tile 64 x 64; BN = 64:
// waves 4 x 1, wave tile 32 x 64), two workgroups per CU; reduction in chunks of 16 (ONE MFMA reduction step), double-buffered in LDS as
// three bf16 planes per operand ([plane][row][16 k], the two 16-byte units of a row swapped for rows with bit 3 set: the 16 lanes of a
// ds_read_b128 service group hit 16 distinct bank quads, SQ_LDS_BANK_CONFLICT = 0).  Per chunk and wave: 12 (9) ds_read_b128, 24 (12)
// MFMAs -- consecutive ones on different accumulator tiles --, the activations of two chunks ahead fetched into registers, those of
// the next chunk split and written to the other buffer, one barrier.  Weight gradient (b6_wgrad): see there.
#include "common.h"

#include <stdlib.h>

namespace {

// ablation switches (dev builds with -DDBEV_BF6_ABLATE, DBEV_BF6_DBG bits: 1 no global fetch, 2 no split / LDS staging, 4 no barrier, 8 no stores, 16 clock probe)
#ifdef DBEV_BF6_ABLATE
#define B6_DBG(bit_) (dbg & (bit_))
#else
#define B6_DBG(bit_) 0
#endif

#ifndef B6_NT
#define B6_NT 0
#endif

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int B6_BM = 128, B6_KC = 16;

__device__ __forceinline__ unsigned b6_rne(float x) {                    // bf16(x), round to nearest even, as the upper half of a dword
  const unsigned u = __float_as_uint(x);
  const unsigned r = (u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u;
  // a finite x above the largest bf16 (|x| > 3.3895e38) would round to inf and leave a NaN remainder: keep the truncated piece
  // there (the later pieces absorb the remainder exactly, as for every other x).  inf / NaN inputs pass through unchanged.
  return ((r & 0x7f800000u) == 0x7f800000u && (u & 0x7f800000u) != 0x7f800000u) ? (u & 0xffff0000u) : r;
}

// ---- weight split + packing: W element (n, k) at w[n * sn + k * sk] -> planes in the LDS image order -----------------------------
// packed (bf16 elements): ((((nb * nkc + kc) * 3 + plane) * BN + row) * 16) + (u ^ ((row >> 3) & 1)) * 8 + i,  n = nb BN + row,
// k = 16 kc + 8 u + i
__device__ __forceinline__ void b6_pack_elem(long long idx, const float* __restrict__ w, long long sn, long long sk, int N, int K, int BN,
                                             unsigned short* __restrict__ out) {
  const int nk8 = K / 8;                                                              // idx = (n, k unit of 8)
  if (idx >= static_cast<long long>(N) * nk8) return;
  const int n = static_cast<int>(idx / nk8), k8 = static_cast<int>(idx % nk8);
  const int nb = n / BN, row = n % BN, kc = k8 / 2, u = k8 % 2, nkc = K / B6_KC;
  unsigned pk[3][4];                                                                  // two bf16 per dword, fully unrolled
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float x = w[n * sn + static_cast<long long>(k8 * 8 + i) * sk];
    const unsigned h0 = b6_rne(x);
    const float r1 = x - __uint_as_float(h0);
    const unsigned h1 = b6_rne(r1);
    const float r2 = r1 - __uint_as_float(h1);
    const unsigned h2 = b6_rne(r2);
    if (i & 1) { pk[0][i >> 1] |= h0; pk[1][i >> 1] |= h1; pk[2][i >> 1] |= h2; }
    else { pk[0][i >> 1] = h0 >> 16; pk[1][i >> 1] = h1 >> 16; pk[2][i >> 1] = h2 >> 16; }
  }
#pragma unroll
  for (int pl = 0; pl < 3; ++pl) {
    unsigned short* o = out + ((((static_cast<long long>(nb) * nkc + kc) * 3 + pl) * BN + row) * 16) + ((u ^ ((row >> 3) & 1)) * 8);
    *reinterpret_cast<uint4*>(o) = make_uint4(pk[pl][0], pk[pl][1], pk[pl][2], pk[pl][3]);
  }
}

__global__ __launch_bounds__(256) void b6_pack(const float* __restrict__ w, long long sn, long long sk, int N, int K, int BN,
                                               unsigned short* __restrict__ out) {
  b6_pack_elem(static_cast<long long>(blockIdx.x) * 256 + threadIdx.x, w, sn, sk, N, K, BN, out);
}

// both orientations of a [Co, Ci] filter in one launch: blockIdx.y = 0 the forward planes (rows = output channels), 1 the data
// gradient's (rows = input channels, reduction over the output channels)
__global__ __launch_bounds__(256) void b6_pack_pair(const float* __restrict__ w, long long so, long long sc, int Co, int Ci, int bn_f,
                                                    unsigned short* __restrict__ out_f, int bn_t, unsigned short* __restrict__ out_t) {
  const long long idx = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x;
  if (blockIdx.y == 0) b6_pack_elem(idx, w, so, sc, Co, Ci, bn_f, out_f);
  else b6_pack_elem(idx, w, sc, so, Ci, Co, bn_t, out_t);
}

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float floatx2 __attribute__((ext_vector_type(2)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef unsigned uintx4 __attribute__((ext_vector_type(4)));

// (a, b) -> three dwords of two bf16 each (round to nearest even, v_cvt_pk_bf16_f32): a = p0.lo + p1.lo + p2.lo up to 2^-25 |a|
__device__ __forceinline__ void b6_split2(float a, float b, unsigned& p0, unsigned& p1, unsigned& p2) {
  floatx2 v = {a, b};
  bf16x2 h = __builtin_convertvector(v, bf16x2);
  p0 = *reinterpret_cast<unsigned*>(&h);
  // a FINITE value above the largest bf16 (|x| > 3.3895e38) rounds to inf and would leave a NaN remainder where the library's fp32
  // GEMM returns a finite result: keep the truncated leading piece there (the later pieces absorb the remainder exactly).  One
  // max + compare per pair on the common path; inf / NaN inputs are left to propagate.
  if (__builtin_expect(fmaxf(fabsf(a), fabsf(b)) >= 3.3895314e38f, 0)) {
    const unsigned ua = __float_as_uint(a), ub = __float_as_uint(b);
    if ((ua & 0x7f800000u) != 0x7f800000u && (p0 & 0x00007f80u) == 0x00007f80u) p0 = (p0 & 0xffff0000u) | (ua >> 16);
    if ((ub & 0x7f800000u) != 0x7f800000u && (p0 & 0x7f800000u) == 0x7f800000u) p0 = (p0 & 0x0000ffffu) | (ub & 0xffff0000u);
  }
  floatx2 f = {__uint_as_float(p0 << 16), __uint_as_float(p0 & 0xffff0000u)};
  v = v - f;                                                           // exact
  h = __builtin_convertvector(v, bf16x2);
  p1 = *reinterpret_cast<unsigned*>(&h);
  f = floatx2{__uint_as_float(p1 << 16), __uint_as_float(p1 & 0xffff0000u)};
  v = v - f;
  h = __builtin_convertvector(v, bf16x2);
  p2 = *reinterpret_cast<unsigned*>(&h);
}

__device__ __forceinline__ unsigned long long b6_uniform64(unsigned long long v) {
  const unsigned lo = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(v)), hi = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(v >> 32));
  return (static_cast<unsigned long long>(hi) << 32) | lo;
}

// LDS image of a chunk (16 reduction steps): [plane 3][row][2 units of 8 bf16], unit u of row r in slot u ^ ((r >> 3) & 1)
// The reduction runs in groups of FOUR chunks (K % 64 == 0): the matrix pipe accumulates a group from zero (the first instruction of
// a tile takes the constant 0 as its addend), then the group's sum is added to `tot` by the VALU in round-to-nearest -- the error of a
// long chain of matrix instructions on one accumulator grows with its length (1.8e-6 of the output scale at K = 2048 against 4.8e-7
// for the fp32 kernels), with 24-instruction chains it is 2-3e-7 at every K.
// STATS: the per-channel sums of y and y^2 of the tile (the BatchNorm statistics of the output: `partial` f32[M / 128][2][N], rows in
// bn_finalize's layout, merged there in fp64) leave the accumulators in the epilogue -- no statistics pass over y
template <int BN, int OCC, bool STATS>
__global__ __launch_bounds__(256, OCC) void b6_fwd(const float* __restrict__ X, const unsigned short* __restrict__ Wp,
                                                   float* __restrict__ Y, float* __restrict__ partial, int M, int K, int N, int xs, int dbg) {
  constexpr int WN = BN / 64, WM = 4 / WN, TM = B6_BM / WM / 32;            // waves along N / M, 32-row tiles per wave
  constexpr int APL = B6_BM * 32, BPL = BN * 32;                            // bytes of one plane of a chunk
  constexpr int ABUF = 3 * APL, BBUF = 3 * BPL;
  constexpr int BV = BBUF / 16;                                             // 16-byte units of a packed weight chunk: 768 / 384
  static_assert(BV == 768 || BV == 384, "weight chunk rounds below");
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * ABUF + 2 * BBUF];
  unsigned char* sA = smem;
  unsigned char* sB = smem + 2 * ABUF;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int wm = w % WM, wn = w / WM;
  const int nblk = N / BN;
  const int L = xcd_block();
  const int nb = L % nblk, mb = L / nblk;
  if (mb >= M / B6_BM) return;
  const int m0 = mb * B6_BM, n0 = nb * BN;
  const int nkc = K / B6_KC;

  // activation fetch: thread = (row tid >> 2 and that row + 64, floats 4 (tid & 3) .. + 3 of the chunk); scalar base + 32-bit offsets
  const int row0 = tid >> 2, row1 = 64 + (tid >> 2), c4 = tid & 3;
  typedef const __attribute__((address_space(1))) float* gfloat_p;      // (a pointer rebuilt from integers is generic: flat loads)
  typedef const __attribute__((address_space(1))) uintx4* guint4_p;
  const gfloat_p xb = reinterpret_cast<gfloat_p>(b6_uniform64(reinterpret_cast<unsigned long long>(X + static_cast<size_t>(m0) * xs)));
  const unsigned xo0 = static_cast<unsigned>(row0 * xs + 4 * c4), xo1 = static_cast<unsigned>(row1 * xs + 4 * c4);
  const int aoff0 = row0 * 32 + (((c4 >> 1) ^ ((row0 >> 3) & 1)) * 16) + (c4 & 1) * 8;
  const int aoff1 = row1 * 32 + (((c4 >> 1) ^ ((row1 >> 3) & 1)) * 16) + (c4 & 1) * 8;
  const guint4_p wsrc = reinterpret_cast<guint4_p>(b6_uniform64(reinterpret_cast<unsigned long long>(
      reinterpret_cast<const uintx4*>(Wp) + static_cast<size_t>(nb) * nkc * BV)));
  // two chunks in flight in registers (set q = chunk & 1); named scalars: indexed arrays of these end up in scratch memory
  floatx4 xa0_0, xa0_1, xa1_0, xa1_1;
  uintx4 wb0_0, wb0_1 = {0, 0, 0, 0}, wb0_2 = {0, 0, 0, 0}, wb1_0, wb1_1 = {0, 0, 0, 0}, wb1_2 = {0, 0, 0, 0};
#define B6_FETCH(q_, kc_)                                                                                            \
  do {                                                                                                               \
    const gfloat_p xc_ = xb + (kc_) * B6_KC;                                                                         \
    xa##q_##_0 = *reinterpret_cast<const __attribute__((address_space(1))) floatx4*>(xc_ + xo0);                      \
    xa##q_##_1 = *reinterpret_cast<const __attribute__((address_space(1))) floatx4*>(xc_ + xo1);                      \
    const guint4_p ws_ = wsrc + static_cast<size_t>(kc_) * BV;                                                       \
    wb##q_##_0 = ws_[tid];                                                                                           \
    if (BV == 768 || tid < 128) wb##q_##_1 = ws_[tid + 256];                                                         \
    if (BV == 768) wb##q_##_2 = ws_[tid + 512];                                                                      \
  } while (0)
#define B6_SPLIT_STORE(v_, off_)                                                                                     \
  do {                                                                                                               \
    unsigned p0a, p1a, p2a, p0b, p1b, p2b;                                                                           \
    b6_split2((v_).x, (v_).y, p0a, p1a, p2a);                                                                        \
    b6_split2((v_).z, (v_).w, p0b, p1b, p2b);                                                                        \
    *reinterpret_cast<uint2*>(a_ + (off_)) = make_uint2(p0a, p0b);                                                   \
    *reinterpret_cast<uint2*>(a_ + APL + (off_)) = make_uint2(p1a, p1b);                                             \
    *reinterpret_cast<uint2*>(a_ + 2 * APL + (off_)) = make_uint2(p2a, p2b);                                         \
  } while (0)
#define B6_STAGE(q_, buf_)                                                                                           \
  do {                                                                                                               \
    unsigned char* a_ = sA + (buf_) * ABUF;                                                                          \
    B6_SPLIT_STORE(xa##q_##_0, aoff0);                                                                               \
    B6_SPLIT_STORE(xa##q_##_1, aoff1);                                                                               \
    uintx4* b_ = reinterpret_cast<uintx4*>(sB + (buf_) * BBUF) + tid;                                                \
    b_[0] = wb##q_##_0;                                                                                              \
    if (BV == 768 || tid < 128) b_[256] = wb##q_##_1;                                                                \
    if (BV == 768) b_[512] = wb##q_##_2;                                                                             \
  } while (0)

  floatx16 acc[TM][2], tot[TM][2];
  const floatx16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) tot[a][b] = zero16;

  const unsigned long long t0c = B6_DBG(16) ? __builtin_amdgcn_s_memtime() : 0ull, t0r = B6_DBG(16) ? __builtin_amdgcn_s_memrealtime() : 0ull;
  B6_FETCH(0, 0);
  B6_FETCH(1, 1);                                                        // nkc >= 4
  B6_STAGE(0, 0);
  __syncthreads();
  // operand addresses of this lane (the chunk is ONE MFMA reduction step: unit = half)
  int aaddr[TM], baddr[2];
#pragma unroll
  for (int a = 0; a < TM; ++a) {
    const int row = (wm * TM + a) * 32 + l31;
    aaddr[a] = row * 32 + ((half ^ ((row >> 3) & 1)) * 16);
  }
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int row = (wn * 2 + b) * 32 + l31;
    baddr[b] = row * 32 + ((half ^ ((row >> 3) & 1)) * 16);
  }

  // chunk j of a group: registers of set j & 1 hold chunk kc + 1's operands when it starts ... (q_ = j & 1, cur_ = LDS buffer j & 1)
#define B6_CHUNK(j_, first_)                                                                                         \
  do {                                                                                                               \
    const int kc = g + (j_);                                                                                         \
    const unsigned char* a_ = sA + ((j_) & 1) * ABUF;                                                                \
    const unsigned char* b_ = sB + ((j_) & 1) * BBUF;                                                                \
    bf16x8 af[TM][3], bf[2][3];                                                                                      \
    _Pragma("unroll") for (int a = 0; a < TM; ++a)                                                                    \
      _Pragma("unroll") for (int p = 0; p < 3; ++p) af[a][p] = *reinterpret_cast<const bf16x8*>(a_ + p * APL + aaddr[a]); \
    _Pragma("unroll") for (int b = 0; b < 2; ++b)                                                                     \
      _Pragma("unroll") for (int p = 0; p < 3; ++p) bf[b][p] = *reinterpret_cast<const bf16x8*>(b_ + p * BPL + baddr[b]); \
    /* chunk kc + 1 (registers, other set) -> the other LDS buffer; chunk kc + 2 -> registers, this set */           \
    if (kc + 1 < nkc && !B6_DBG(2)) { if (((j_) & 1) == 0) B6_STAGE(1, 1); else B6_STAGE(0, 0); }                    \
    if (kc + 2 < nkc && !B6_DBG(1)) { if (((j_) & 1) == 0) B6_FETCH(0, kc + 2); else B6_FETCH(1, kc + 2); }          \
    /* the six partial products, smallest first; consecutive instructions go to DIFFERENT accumulator tiles (a dependent    \
       instruction waits for the whole pass count of its predecessor) */                                             \
    _Pragma("unroll") for (int t = 0; t < 6; ++t) {                                                                   \
      const int pa = t == 0 ? 2 : (t == 1 || t == 3) ? 1 : 0, pb = t == 0 || t == 3 || t == 5 ? 0 : (t == 1 || t == 4) ? 1 : 2; \
      _Pragma("unroll") for (int a = 0; a < TM; ++a)                                                                  \
        _Pragma("unroll") for (int b = 0; b < 2; ++b)                                                                 \
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][pa], bf[b][pb], (first_) && t == 0 ? zero16 : acc[a][b], 0, 0, 0); \
    }                                                                                                                \
    if (!B6_DBG(4)) __syncthreads();                                                                                 \
  } while (0)
  for (int g = 0; g < nkc; g += 4) {
    B6_CHUNK(0, true);
    B6_CHUNK(1, false);
    B6_CHUNK(2, false);
    B6_CHUNK(3, false);
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) tot[a][b] += acc[a][b];
  }
#undef B6_CHUNK
#undef B6_FETCH
#undef B6_STAGE
#undef B6_SPLIT_STORE
  // accumulator register r = row (r & 3) + 8 (r >> 2) + 4 half of the 32 x 32 tile, column l31
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      float* y = Y + static_cast<size_t>(m0 + (wm * TM + a) * 32 + 4 * half) * N + n0 + (wn * 2 + b) * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (!B6_DBG(8) || tot[a][b][r] == 12345.678f) {
          if (B6_NT) __builtin_nontemporal_store(tot[a][b][r], y + static_cast<size_t>((r & 3) + 8 * (r >> 2)) * N);
          else y[static_cast<size_t>((r & 3) + 8 * (r >> 2)) * N] = tot[a][b][r];
        }
    }
  if (STATS) {
    float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) { const float v = tot[a][b][r]; s1[b] += v; s2[b] = fmaf(v, v, s2[b]); }
#pragma unroll
    for (int b = 0; b < 2; ++b) { s1[b] += __shfl_xor(s1[b], 32); s2[b] += __shfl_xor(s2[b], 32); }
    float* red = reinterpret_cast<float*>(smem);                             // [which 2][wm][BN]; the loop ended on a barrier
    if (half == 0) {
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        red[(0 * WM + wm) * BN + (wn * 2 + b) * 32 + l31] = s1[b];
        red[(1 * WM + wm) * BN + (wn * 2 + b) * 32 + l31] = s2[b];
      }
    }
    __syncthreads();
    if (tid < 2 * BN) {
      const int which = tid / BN, c = tid % BN;
      float s = red[(which * WM) * BN + c];
#pragma unroll
      for (int j = 1; j < WM; ++j) s += red[(which * WM + j) * BN + c];      // fixed order
      partial[(static_cast<size_t>(mb) * 2 + which) * N + n0 + c] = s;
    }
  }
  if (B6_DBG(16) && blockIdx.x == 0 && tid == 0) {          // ablation only: shader clocks / 100 MHz ticks of this workgroup -> y[0..1]
    const unsigned long long t1c = __builtin_amdgcn_s_memtime(), t1r = __builtin_amdgcn_s_memrealtime();
    reinterpret_cast<unsigned*>(Y)[0] = static_cast<unsigned>(t1c - t0c);
    reinterpret_cast<unsigned*>(Y)[1] = static_cast<unsigned>(t1r - t0r);
  }
}

// ---- b6_fwd2 (round 5): the same tile, LDS image and accumulation scheme as b6_fwd, software-pipelined -----------------------------
// What b6_fwd loses (rocprof + ISA, round 5): the compiler runs a chunk's 24 MFMAs back to back and the staging of the next chunk
// (split, LDS writes, fetch) AFTER them, so a wave alternates ~770 matrix cycles with ~700 cycles of everything else and the two
// waves of a SIMD cover each other to 54 %; the split's exact subtractions and the group sums came out as v_pk_add_f32, which costs
// ~13 extra cycles each beside bf16 MFMAs (MI355X_MICROARCH.md, "anti-lever"); and the weight planes made a round trip through 12
// VGPRs.  Here: (1) the chunk body is ONE basic block (out-of-range prefetches are clamped, not branched around) whose instruction
// order is prescribed by sched_group_barrier: operand reads first, then one MFMA / a few VALU / now and then an LDS write or a fetch;
// (2) scalar v_sub_f32 / v_add_f32 (scalar source + -fno-slp-vectorize for this file); (3) the packed weight chunk goes global -> LDS by
// LDS-DMA (global_load_lds_dwordx4), two chunks ahead into a ring of four buffers -- no registers, no ds_write, and its landing is
// implied by the wait the activation split needs anyway; (4) a finite |x| above the largest bf16 is clamped for the LEADING piece only
// (v_med3_f32; the remainder pieces absorb the difference exactly), inf / NaN still propagate.
// (plain scalar C++ -- the file is compiled with -fno-slp-vectorize so that adjacent scalar operations stay v_sub_f32 / v_add_f32.
// NOT inline asm: the hazard recogniser does not see inline asm as a VALU instruction and would leave out the wait states a VALU
// read of an MFMA result needs -- measured: the group sums read the accumulators before the matrix pipe had written them.)
__device__ __forceinline__ float b6_sub(float a, float b) { return a - b; }
__device__ __forceinline__ float b6_add(float a, float b) { return a + b; }
constexpr float B6_BF16_MAX = 3.3895313892515355e38f;               // 0x7f7f0000

__device__ __forceinline__ void b6_split2s(float a, float b, unsigned& p0, unsigned& p1, unsigned& p2) {
  floatx2 v = {__builtin_amdgcn_fmed3f(a, B6_BF16_MAX, -B6_BF16_MAX), __builtin_amdgcn_fmed3f(b, B6_BF16_MAX, -B6_BF16_MAX)};
  bf16x2 h = __builtin_convertvector(v, bf16x2);
  p0 = *reinterpret_cast<unsigned*>(&h);
  float ra = b6_sub(a, __uint_as_float(p0 << 16)), rb = b6_sub(b, __uint_as_float(p0 & 0xffff0000u));      // exact
  v = floatx2{ra, rb};
  h = __builtin_convertvector(v, bf16x2);
  p1 = *reinterpret_cast<unsigned*>(&h);
  ra = b6_sub(ra, __uint_as_float(p1 << 16));
  rb = b6_sub(rb, __uint_as_float(p1 & 0xffff0000u));
  v = floatx2{ra, rb};
  h = __builtin_convertvector(v, bf16x2);
  p2 = *reinterpret_cast<unsigned*>(&h);
}

#define B6_SGB(mask_, n_) __builtin_amdgcn_sched_group_barrier((mask_), (n_), 0)
// schedule knobs of b6_fwd2's chunk (dev builds override them: tools/build_variant.sh): VALU instructions per MFMA round, the rounds
// after which the plane stores of the first / second float4 are placed
// (measured over the step's twelve 1x1 shapes, tools/kbench_bf6_fwd.py, sum of the forward times: 3 / 9 / 18 -> 1.66 ms, 5 / 6 / 11 -> 1.62 ms,
// 6 / 5 / 9 -> 1.65, 8 / 4 / 7 and 12 / 3 / 5 -> 1.67: the split a little ahead of the MFMAs, not all of it up front)
// ... and of b6_wgrad2's (128 x 128 tiles): VALU per round, the round around which the first channel's three plane stores sit, the
// round from which the second channel's follow
// weight chunks in flight ahead of the one being consumed (ring of four buffers: 2 or 3)
#ifndef B6_AHEAD
#define B6_AHEAD 2
#endif
#ifndef B6W_VG
#define B6W_VG 5
#endif
#ifndef B6W_H
#define B6W_H 12
#endif
#ifndef B6W_T
#define B6W_T 20
#endif
#ifndef B6_VG
#define B6_VG 5
#endif
#ifndef B6_W0
#define B6_W0 6
#endif
#ifndef B6_W1
#define B6_W1 11
#endif

// CONV (round 5): the same GEMM as an IMPLICIT one for nn.Conv2d(C, Co, 3, stride 2, padding 1) -- row m is the output pixel (n, i, j),
// reduction index k = (ky * 3 + kx) * C + c (the channels-last filter's memory order, so the packed planes come from the ordinary
// pack of the [Co][9 C] matrix), a 16-wide chunk lies inside one tap: its activation fragment is 16 consecutive channels of input
// pixel (2 i + ky - 1, 2 j + kx - 1), or zeros outside the map (the load goes to a valid offset and the value is discarded: no
// branch).  Everything behind the activation fetch -- split, LDS image, weight DMA, schedule, epilogues -- is the 1x1 kernel's.
struct B6Conv { int H, W, C, Ho, Wo, cshift; };                          // cshift = log2(C / 16) (C a power of two >= 64)
// (Round 5 also carried an AFF variant -- relu(x * scale[k] + shift[k]) applied as the operand is fetched, the norm in front of the
// convolution never written -- for the gradient-free frame: bit-equal, measured neutral (the eight extra VALU instructions per fetched
// float4 cost the VALU-co-limited GEMM what the pass cost the norm); retired in round 6 with its ABI entry.)
typedef unsigned uintx4_t __attribute__((ext_vector_type(4)));
template <int BN, bool STATS, bool CONV = false>
__global__ __launch_bounds__(256, 2) void b6_fwd2(const float* __restrict__ X, const unsigned short* __restrict__ Wp,
                                                  float* __restrict__ Y, float* __restrict__ partial, int M, int K, int N, int xs,
                                                  B6Conv cv = B6Conv{}, const float* __restrict__ bias = nullptr) {
  constexpr int WN = BN / 64, WM = 4 / WN, TM = B6_BM / WM / 32;            // waves along N / M, 32-row tiles per wave
  constexpr int APL = B6_BM * 32, BPL = BN * 32;                            // bytes of one plane of a chunk
  constexpr int ABUF = 3 * APL, BBUF = 3 * BPL;                             // 12288, 12288 / 6144
  constexpr int NDMA = BBUF / 1024;                                          // 64-lane x 16-byte DMA instructions per chunk: 12 / 6
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];      // sA[2][ABUF] | sB[4][BBUF]
  unsigned char* sA = smem;
  unsigned char* sB = smem + 2 * ABUF;
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int wm = w % WM, wn = w / WM;
  const int nblk = N / BN;
  const int L = xcd_block();
  const int nb = L % nblk, mb = L / nblk;
  if (mb >= (M + B6_BM - 1) / B6_BM) return;
  const int m0 = mb * B6_BM, n0 = nb * BN;
  const int nkc = K / B6_KC;
  // Round 6: M need not be a multiple of 128 (the BEVFormer recipe's 928 x 1600 images give 58 x 100 and 29 x 50 maps).  The 1x1 path
  // reads X and writes Y through BUFFER descriptors of this row block's valid rows: a row past M loads zeros (so its outputs and its
  // share of the statistics are exact zeros) and its stores are dropped by the bounds check -- no instruction added anywhere.
  const int vrows = min(B6_BM, M - m0);

  const int row0 = tid >> 2, row1 = 64 + (tid >> 2), c4 = tid & 3;
  typedef const __attribute__((address_space(1))) float* gfloat_p;
  const gfloat_p xb = reinterpret_cast<gfloat_p>(b6_uniform64(reinterpret_cast<unsigned long long>(X + (CONV ? 0 : static_cast<size_t>(m0) * xs))));
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(reinterpret_cast<const float*>(b6_uniform64(reinterpret_cast<unsigned long long>(X + static_cast<size_t>(m0) * xs)))), 0,
      vrows * xs * 4, 0x00020000);
  const unsigned xo0 = static_cast<unsigned>(row0 * xs + 4 * c4), xo1 = static_cast<unsigned>(row1 * xs + 4 * c4);
  int pb0 = 0, pb1 = 0;                                                     // CONV: float offset of tap (0, 0) of the row's pixel (may be < 0)
  unsigned vm0 = 0, vm1 = 0;                                                //       bit t: tap t lies inside the map
  if (CONV) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int m = m0 + (q ? row1 : row0);
      const int j = m % cv.Wo, t = m / cv.Wo;
      const int i = t % cv.Ho, n = t / cv.Ho;
      const int pb = ((n * cv.H + 2 * i - 1) * cv.W + 2 * j - 1) * cv.C + 4 * c4;
      unsigned vm = 0;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int yy = 2 * i + ky - 1, xx = 2 * j + kx - 1;
          if (yy >= 0 && yy < cv.H && xx >= 0 && xx < cv.W) vm |= 1u << (ky * 3 + kx);
        }
      if (m >= M) vm = 0;                                                    // (round 6: a row past M has no tap inside the map: zeros)
      if (q) { pb1 = pb; vm1 = vm; } else { pb0 = pb; vm0 = vm; }
    }
  }
  const int aoff0 = row0 * 32 + (((c4 >> 1) ^ ((row0 >> 3) & 1)) * 16) + (c4 & 1) * 8;
  const int aoff1 = row1 * 32 + (((c4 >> 1) ^ ((row1 >> 3) & 1)) * 16) + (c4 & 1) * 8;
  // weight chunks of this column block: LDS-DMA, wave w copies the 1 KB pieces w, w + 4, w + 8 (BN = 128) / piece w and a quarter
  // of the last two (BN = 64: 32 lanes) of the 12 KB / 6 KB chunk image
  const unsigned long long wsrc = b6_uniform64(reinterpret_cast<unsigned long long>(Wp) + static_cast<unsigned long long>(nb) * nkc * BBUF);
  typedef __attribute__((address_space(3))) unsigned char* lds_p;
  const unsigned ldsB = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(reinterpret_cast<size_t>((lds_p)sB)));
  const unsigned dvoff = lane * 16;
#define B6_DMA(sbase_, ldsaddr_)                                                                                     \
  do {                                                                                                               \
    unsigned keep_;                                                                                                  \
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0" \
                 : "=&s"(keep_) : "v"(dvoff), "s"(ldsaddr_), "s"(sbase_) : "memory");                                \
  } while (0)
#define B6_DMA_HALF(sbase_, ldsaddr_)                                                                                \
  do {                                                                                                               \
    unsigned keep_;                                                                                                  \
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_mov_b64 exec, 0xffffffff\n\tglobal_load_lds_dwordx4 %1, %3\n\t" \
                 "s_mov_b64 exec, -1\n\ts_mov_b32 m0, %0"                                                            \
                 : "=&s"(keep_) : "v"(dvoff), "s"(ldsaddr_), "s"(sbase_) : "memory");                                \
  } while (0)
  // chunk kc_ (clamped) -> ring slot slot_
#define B6_DMA_CHUNK(kc_, slot_)                                                                                     \
  do {                                                                                                               \
    const int kq_ = (kc_) < nkc ? (kc_) : nkc - 1;                                                                   \
    const unsigned long long sb_ = wsrc + static_cast<unsigned long long>(kq_) * BBUF;                               \
    const unsigned la_ = ldsB + (slot_) * BBUF;                                                                      \
    if (NDMA == 12) {                                                                                                \
      B6_DMA(sb_ + w * 1024, la_ + w * 1024);                                                                        \
      B6_DMA(sb_ + (w + 4) * 1024, la_ + (w + 4) * 1024);                                                            \
      B6_DMA(sb_ + (w + 8) * 1024, la_ + (w + 8) * 1024);                                                            \
    } else {                                                                                                         \
      B6_DMA(sb_ + w * 1024, la_ + w * 1024);                                                                        \
      B6_DMA_HALF(sb_ + 4096 + w * 512, la_ + 4096 + w * 512);                                                      \
    }                                                                                                                \
  } while (0)
  floatx4 xa0_0, xa0_1, xa1_0, xa1_1;                                    // raw activations of two chunks in flight
  bool xf0_0 = true, xf0_1 = true, xf1_0 = true, xf1_1 = true;           // CONV: ... and whether their tap lies inside the map
#define B6_LOADA(q_, kc_)                                                                                            \
  do {                                                                                                               \
    const int kq_ = (kc_) < nkc ? (kc_) : nkc - 1;                                                                   \
    if (CONV) {                                                                                                      \
      const int tap_ = kq_ >> cv.cshift;                                                                             \
      const int ky_ = (tap_ * 11) >> 5;                                                                              \
      const int toff_ = (ky_ * cv.W + (tap_ - 3 * ky_)) * cv.C + ((kq_ - (tap_ << cv.cshift)) << 4);                 \
      xf##q_##_0 = (vm0 >> tap_) & 1u;                                                                               \
      xf##q_##_1 = (vm1 >> tap_) & 1u;                                                                               \
      const unsigned o0_ = xf##q_##_0 ? static_cast<unsigned>(pb0 + toff_) : static_cast<unsigned>(4 * c4);         \
      const unsigned o1_ = xf##q_##_1 ? static_cast<unsigned>(pb1 + toff_) : static_cast<unsigned>(4 * c4);         \
      xa##q_##_0 = *reinterpret_cast<const __attribute__((address_space(1))) floatx4*>(xb + o0_);                     \
      xa##q_##_1 = *reinterpret_cast<const __attribute__((address_space(1))) floatx4*>(xb + o1_);                     \
    } else {                                                                                                         \
      xa##q_##_0 = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(xrs, 4 * xo0, kq_ * (B6_KC * 4), 0)); \
      xa##q_##_1 = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(xrs, 4 * xo1, kq_ * (B6_KC * 4), 0)); \
    }                                                                                                                \
  } while (0)
#define B6_SPLIT_STORE2(v_, off_)                                                                                    \
  do {                                                                                                               \
    unsigned p0a, p1a, p2a, p0b, p1b, p2b;                                                                           \
    b6_split2s((v_).x, (v_).y, p0a, p1a, p2a);                                                                       \
    b6_split2s((v_).z, (v_).w, p0b, p1b, p2b);                                                                       \
    *reinterpret_cast<uint2*>(a_ + (off_)) = make_uint2(p0a, p0b);                                                   \
    *reinterpret_cast<uint2*>(a_ + APL + (off_)) = make_uint2(p1a, p1b);                                             \
    *reinterpret_cast<uint2*>(a_ + 2 * APL + (off_)) = make_uint2(p2a, p2b);                                         \
  } while (0)
#define B6_STAGEA(q_, buf_)                                                                                          \
  do {                                                                                                               \
    unsigned char* a_ = sA + (buf_) * ABUF;                                                                          \
    if (CONV) {                                                                                                      \
      const floatx4 z4_ = {0.f, 0.f, 0.f, 0.f};                                                                      \
      xa##q_##_0 = xf##q_##_0 ? xa##q_##_0 : z4_;                                                                    \
      xa##q_##_1 = xf##q_##_1 ? xa##q_##_1 : z4_;                                                                    \
    }                                                                                                                \
    B6_SPLIT_STORE2(xa##q_##_0, aoff0);                                                                              \
    B6_SPLIT_STORE2(xa##q_##_1, aoff1);                                                                              \
  } while (0)

  floatx16 acc[TM][2];
  float tot[TM][2][16];
  const floatx16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) tot[a][b][r] = 0.f;

  // prologue: weights of chunks 0, 1 on their way, activations of chunks 0 (staged), 1, 2 (registers)
  B6_DMA_CHUNK(0, 0);
  B6_DMA_CHUNK(1, 1);
  if (B6_AHEAD == 3) B6_DMA_CHUNK(2, 2);
  B6_LOADA(0, 0);
  B6_LOADA(1, 1);
  B6_STAGEA(0, 0);
  B6_LOADA(0, 2);
  asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                       // the four activation loads may stay in flight; both DMAs landed
  __syncthreads();
  int aaddr[TM], baddr[2];
#pragma unroll
  for (int a = 0; a < TM; ++a) {
    const int row = (wm * TM + a) * 32 + l31;
    aaddr[a] = row * 32 + ((half ^ ((row >> 3) & 1)) * 16);
  }
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int row = (wn * 2 + b) * 32 + l31;
    baddr[b] = row * 32 + ((half ^ ((row >> 3) & 1)) * 16);
  }

  // chunk j_ of a group (kc = g + j_): LDS buffers A[j_ & 1], B[j_]; raw activations: set (j_ + 1) & 1 = chunk kc + 1 (staged here,
  // then refilled with chunk kc + 3), the other set = chunk kc + 2
#define B6_CHUNK2(j_, first_, last_)                                                                                 \
  do {                                                                                                               \
    const int kc = g + (j_);                                                                                         \
    const unsigned char* ar_ = sA + ((j_) & 1) * ABUF;                                                               \
    const unsigned char* br_ = sB + (j_) * BBUF;                                                                     \
    bf16x8 af[TM][3], bf[2][3];                                                                                      \
    _Pragma("unroll") for (int p = 2; p >= 0; --p) {                                                                  \
      _Pragma("unroll") for (int a = 0; a < TM; ++a) af[a][p] = *reinterpret_cast<const bf16x8*>(ar_ + p * APL + aaddr[a]); \
      _Pragma("unroll") for (int b = 0; b < 2; ++b) bf[b][p] = *reinterpret_cast<const bf16x8*>(br_ + p * BPL + baddr[b]); \
    }                                                                                                                \
    if (((j_) & 1) == 0) B6_STAGEA(1, 1); else B6_STAGEA(0, 0);                                                      \
    B6_DMA_CHUNK(kc + B6_AHEAD, ((j_) + B6_AHEAD) & 3);                                                              \
    if (((j_) & 1) == 0) B6_LOADA(1, kc + 3); else B6_LOADA(0, kc + 3);                                              \
    _Pragma("unroll") for (int t = 0; t < 6; ++t) {                                                                   \
      const int pa = t == 0 ? 2 : (t == 1 || t == 3) ? 1 : 0, pb = t == 0 || t == 3 || t == 5 ? 0 : (t == 1 || t == 4) ? 1 : 2; \
      _Pragma("unroll") for (int a = 0; a < TM; ++a)                                                                  \
        _Pragma("unroll") for (int b = 0; b < 2; ++b)                                                                 \
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][pa], bf[b][pb], (first_) && t == 0 ? zero16 : acc[a][b], 0, 0, 0); \
    }                                                                                                                \
    if (last_) {                                                                                                     \
      _Pragma("unroll") for (int a = 0; a < TM; ++a)                                                                  \
        _Pragma("unroll") for (int b = 0; b < 2; ++b)                                                                 \
          _Pragma("unroll") for (int r = 0; r < 16; ++r) tot[a][b][r] = b6_add(tot[a][b][r], acc[a][b][r]);           \
    }                                                                                                                \
    /* prescribed order: the operand reads, then MFMA / VALU / (LDS write | fetch) rounds */                        \
    B6_SGB(0x100, 3 * TM + 6);                                                                                       \
    _Pragma("unroll") for (int i = 0; i < 12 * TM; ++i) {                                                             \
      B6_SGB(0x008, 1);                                                                                              \
      B6_SGB(0x002, TM == 2 ? B6_VG : 6);                                                                            \
      /* the plane stores follow their split (26 / 52 VALU instructions in), the refill of the register set follows its last reader */ \
      if (TM == 2 ? (i == B6_W0 || i == B6_W0 + 2 || i == B6_W1 || i == B6_W1 + 2) : (i == 4 || i == 5 || i == 9 || i == 10)) B6_SGB(0x200, 1); \
      if (i == 12 * TM - 3 || i == 12 * TM - 2) B6_SGB(0x020, 1);                                                    \
    }                                                                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                               \
    __syncthreads();                                                                                                 \
    __builtin_amdgcn_sched_barrier(0);            /* nothing moves across a chunk boundary (register-only work would: the next chunk's split) */ \
  } while (0)
  for (int g = 0; g < nkc; g += 4) {
    B6_CHUNK2(0, true, false);
    B6_CHUNK2(1, false, false);
    B6_CHUNK2(2, false, false);
    B6_CHUNK2(3, false, true);
  }
#undef B6_CHUNK2
#undef B6_STAGEA
#undef B6_SPLIT_STORE2
#undef B6_LOADA
#undef B6_DMA_CHUNK
#undef B6_DMA_HALF
#undef B6_DMA
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the clamped prefetches of the last chunks: no DMA may outlive the workgroup's LDS
  __syncthreads();                                           // ... nor land in the LDS another wave re-uses for its output tile below
  if (bias != nullptr) {                                     // round 6: y = x W^T + bias (nn.Linear / a biased 1x1 convolution) without a pass of its own
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const float bv = bias[n0 + (wn * 2 + b) * 32 + l31];
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) tot[a][b][r] = b6_add(tot[a][b][r], bv);
    }
  }
  if (STATS) {
    float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) { const float v = tot[a][b][r]; s1[b] += v; s2[b] = fmaf(v, v, s2[b]); }
#pragma unroll
    for (int b = 0; b < 2; ++b) { s1[b] += __shfl_xor(s1[b], 32); s2[b] += __shfl_xor(s2[b], 32); }
    float* red = reinterpret_cast<float*>(smem);                             // [which 2][wm][BN]; the loop ended on a barrier
    if (half == 0) {
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        red[(0 * WM + wm) * BN + (wn * 2 + b) * 32 + l31] = s1[b];
        red[(1 * WM + wm) * BN + (wn * 2 + b) * 32 + l31] = s2[b];
      }
    }
    __syncthreads();
    if (tid < 2 * BN) {
      const int which = tid / BN, c = tid % BN;
      float s = red[(which * WM) * BN + c];
#pragma unroll
      for (int j = 1; j < WM; ++j) s += red[(which * WM + j) * BN + c];      // fixed order
      partial[(static_cast<size_t>(mb) * 2 + which) * N + n0 + c] = s;
    }
  }
  // output: the accumulator layout (register r = row (r & 3) + 8 (r >> 2) + 4 half of a 32 x 32 tile, column l31) is transposed through
  // the wave's own 16 KB (8 KB) of LDS so that a lane stores 16 consecutive bytes and an instruction covers 256-byte row segments:
  // 16 (8) global_store_dwordx4 per lane instead of 64 (32) global_store_dword -- the scalar stores were issue-bound (rocprof: the
  // K = 64 layers of the first stage wrote at 3.5 TB/s)
  {
    if (STATS) __syncthreads();                                              // `red` shares the front of the LDS
    float* tw = reinterpret_cast<float*>(smem) + w * (TM * 32 * 64);         // [TM * 32 rows][64 columns] of this wave
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) tw[(a * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * 64 + b * 32 + l31] = tot[a][b][r];
    // (the wave reads only what it wrote itself: the compiler's lgkmcnt wait orders the two phases)
    const int rr = lane >> 4, c4q = lane & 15;                               // 4 rows per instruction, 16 lanes x float4 per row
    // (a buffer descriptor of the block's VALID rows: the stores of rows past M are dropped by the hardware's bounds check)
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<float*>(b6_uniform64(reinterpret_cast<unsigned long long>(Y + static_cast<size_t>(m0) * N))), 0, vrows * N * 4, 0x00020000);
    const unsigned yo = static_cast<unsigned>(((wm * TM * 32 + rr) * N + n0 + wn * 64 + 4 * c4q) * 4);
#pragma unroll
    for (int i = 0; i < TM * 8; ++i) {
      const int row = 4 * i + rr;
      const floatx4 v = *reinterpret_cast<const floatx4*>(tw + row * 64 + 4 * c4q);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uintx4_t, v), yrs, yo + static_cast<unsigned>(4 * i * N * 4), 0, 0);
    }
  }
}

// ---- weight gradient: dW[Co, Ci] = sum_m GY[m, Co] * X[m, Ci] --------------------------------------------------------------------
// Both operands are activations: both are split on the fly.  The reduction index is the pixel index m, which is the SLOW index of both
// channels-last tensors, while an MFMA lane wants 8 consecutive reduction steps of one row -- so a thread fetches an 8 (pixels) x 4
// (channels) block (8 float4, coalesced along the channels), and each of its 4 channels' 8 values is exactly one 16-byte fragment
// piece per plane: the transposition happens in registers, the LDS image [plane][channel row][32 m] is written by ds_write_b128.
// Workgroup = 4 waves (2 x 2), tile 128 (or 64) Co x 128 (or 64) Ci, chunks of 32 pixels (two reduction steps), one LDS buffer (48 KB: two
// workgroups per CU), threads 0-127 stage the gradient operand, 128-255 the input operand.  The pixel range is cut into `nsplit` shares
// (one workgroup each per tile), partial sums [nsplit][Co][Ci] are merged by b6_wsum in a fixed order: bit-reproducible, no atomics,
// no zero-fill launch.
constexpr int B6W_KC = 32;

// TA / TB: rows of the gradient-side / input-side tile (128 or 64 output / input channels); wave tile TA/2 x TB/2
template <int TA, int TB>
__global__ __launch_bounds__(256, 2) void b6_wgrad(const float* __restrict__ GY, const float* __restrict__ X, float* __restrict__ part,
                                                   int M, int Co, int Ci, int xs, int rows_per_split, int nsplit) {
  constexpr int PLA = TA * 64, PLB = TB * 64;                               // bytes of one plane of each operand
  constexpr int NA = TA / 64, NB = TB / 64;                                 // 32-row tiles per wave
  __shared__ __attribute__((aligned(16))) unsigned char smem[3 * PLA + 3 * PLB];
  unsigned char* sA = smem;                                                 // gradient operand: rows = output channels
  unsigned char* sB = smem + 3 * PLA;                                       // input operand: rows = input channels
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int wm = w & 1, wn = w >> 1;
  const int tiles_n = Ci / TB, tiles = (Co / TA) * tiles_n;
  const int L = xcd_block();
  const int tile = L % tiles, split = L / tiles;
  if (split >= nsplit) return;
  const int co0 = (tile / tiles_n) * TA, ci0 = (tile % tiles_n) * TB;
  const int mbeg = split * rows_per_split, mend = min(M, mbeg + rows_per_split);
  const int nch = (mend - mbeg) / B6W_KC;

  // staging role of this thread
  const bool isB = tid >= 128;
  const int st = tid & 127;
  const int tw = isB ? TB : TA;                                             // a 64-wide operand is staged by 64 of the 128 threads
  const bool stage = st < tw;
  const int mg = tw == 128 ? st >> 5 : (st >> 4) & 3, cq = tw == 128 ? st & 31 : st & 15;   // pixels 8 mg .. + 7 of the chunk, channels 4 cq .. + 3
  const int PL = isB ? PLB : PLA;
  const float* src = isB ? X + static_cast<size_t>(mbeg + 8 * mg) * xs + ci0 + 4 * cq
                         : GY + static_cast<size_t>(mbeg + 8 * mg) * Co + co0 + 4 * cq;
  const int sstride = isB ? xs : Co;
  unsigned char* sdst = (isB ? sB : sA);
  int soff[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int row = 4 * cq + c;
    soff[c] = row * 64 + ((mg ^ ((row >> 2) & 3)) * 16);
  }
  floatx4 v0, v1, v2, v3, v4, v5, v6, v7;
#define B6W_FETCH(ch_)                                                                                               \
  do {                                                                                                               \
    if (!stage) break;                                                                                               \
    const float* p_ = src + static_cast<size_t>(ch_) * B6W_KC * sstride;                                             \
    v0 = *reinterpret_cast<const floatx4*>(p_);                                                                      \
    v1 = *reinterpret_cast<const floatx4*>(p_ + sstride);                                                            \
    v2 = *reinterpret_cast<const floatx4*>(p_ + 2 * sstride);                                                        \
    v3 = *reinterpret_cast<const floatx4*>(p_ + 3 * sstride);                                                        \
    v4 = *reinterpret_cast<const floatx4*>(p_ + 4 * sstride);                                                        \
    v5 = *reinterpret_cast<const floatx4*>(p_ + 5 * sstride);                                                        \
    v6 = *reinterpret_cast<const floatx4*>(p_ + 6 * sstride);                                                        \
    v7 = *reinterpret_cast<const floatx4*>(p_ + 7 * sstride);                                                        \
  } while (0)
  // channel c_ of the block: 8 pixel values -> one 16-byte piece per plane
#define B6W_STAGE1(c_)                                                                                               \
  do {                                                                                                               \
    if (!stage) break;                                                                                               \
    unsigned a0, a1, a2, b0, b1, b2, c0, c1, c2, d0, d1, d2;                                                         \
    b6_split2(v0[c_], v1[c_], a0, a1, a2);                                                                           \
    b6_split2(v2[c_], v3[c_], b0, b1, b2);                                                                           \
    b6_split2(v4[c_], v5[c_], c0, c1, c2);                                                                           \
    b6_split2(v6[c_], v7[c_], d0, d1, d2);                                                                           \
    *reinterpret_cast<uintx4*>(sdst + soff[c_]) = uintx4{a0, b0, c0, d0};                                            \
    *reinterpret_cast<uintx4*>(sdst + PL + soff[c_]) = uintx4{a1, b1, c1, d1};                                       \
    *reinterpret_cast<uintx4*>(sdst + 2 * PL + soff[c_]) = uintx4{a2, b2, c2, d2};                                   \
  } while (0)

  floatx16 acc[NA][NB], tot[NA][NB];
  const floatx16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int a = 0; a < NA; ++a)
#pragma unroll
    for (int b = 0; b < NB; ++b) { tot[a][b] = zero16; acc[a][b] = zero16; }
  int aaddr[NA][2], baddr[NB][2];                                           // [tile][reduction step]
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2) {
    const int u = 2 * s2 + half;
#pragma unroll
    for (int a = 0; a < NA; ++a) {
      const int ra = (wm * NA + a) * 32 + l31;
      aaddr[a][s2] = ra * 64 + ((u ^ ((ra >> 2) & 3)) * 16);
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int rb = (wn * NB + b) * 32 + l31;
      baddr[b][s2] = rb * 64 + ((u ^ ((rb >> 2) & 3)) * 16);
    }
  }

  if (nch > 0) B6W_FETCH(0);
  for (int ch = 0; ch < nch; ++ch) {
    __syncthreads();                                        // everybody has read the previous chunk's fragments
    B6W_STAGE1(0); B6W_STAGE1(1); B6W_STAGE1(2); B6W_STAGE1(3);
    __syncthreads();
    if (ch + 1 < nch) B6W_FETCH(ch + 1);
    const bool first = (ch & 3) == 0;                       // groups of four chunks = 48 chained instructions per accumulator
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      bf16x8 af[NA][3], bf[NB][3];
#pragma unroll
      for (int p = 0; p < 3; ++p) {
#pragma unroll
        for (int a = 0; a < NA; ++a) af[a][p] = *reinterpret_cast<const bf16x8*>(sA + p * PLA + aaddr[a][s2]);
#pragma unroll
        for (int b = 0; b < NB; ++b) bf[b][p] = *reinterpret_cast<const bf16x8*>(sB + p * PLB + baddr[b][s2]);
      }
#pragma unroll
      for (int t = 0; t < 6; ++t) {
        const int pa = t == 0 ? 2 : (t == 1 || t == 3) ? 1 : 0, pb = t == 0 || t == 3 || t == 5 ? 0 : (t == 1 || t == 4) ? 1 : 2;
#pragma unroll
        for (int a = 0; a < NA; ++a)
#pragma unroll
          for (int b = 0; b < NB; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][pa], bf[b][pb], acc[a][b], 0, 0, 0);
      }
    }
    if ((ch & 3) == 3 || ch + 1 == nch) {
#pragma unroll
      for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b) { tot[a][b] += acc[a][b]; acc[a][b] = zero16; }
    }
    (void)first;
  }
#undef B6W_FETCH
#undef B6W_STAGE1
#pragma unroll
  for (int a = 0; a < NA; ++a)
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      float* o = part + (static_cast<size_t>(split) * Co + co0 + (wm * NA + a) * 32 + 4 * half) * Ci + ci0 + (wn * NB + b) * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) o[static_cast<size_t>((r & 3) + 8 * (r >> 2)) * Ci] = tot[a][b][r];
    }
}

// ---- b6_wgrad2 (round 5): the weight gradient software-pipelined like b6_fwd2 ------------------------------------------------------
// b6_wgrad stages a 32-pixel chunk into ONE LDS buffer between two barriers: the split of both operands (208 VALU instructions per
// thread, v_pk_add_f32 among them) and the 48 MFMAs of a wave never overlap inside a workgroup.  Here: chunks of 16 pixels (one
// reduction step), two LDS buffers, one barrier per chunk; every thread splits 8 pixels x 2 channels of one operand (104 VALU, scalar)
// UNDER the chunk's 24 MFMAs in an order prescribed by sched_group_barrier; raw values fetched two chunks ahead (8 x 8-byte loads per
// thread, a wave covers 512-byte row segments).  Groups of 8 chunks = 48 chained MFMAs per accumulator, as before.  M % 128 == 0.
template <int TA, int TB>
__global__ __launch_bounds__(256, 2) void b6_wgrad2(const float* __restrict__ GY, const float* __restrict__ X, float* __restrict__ part,
                                                    int M, int Co, int Ci, int xs, int rows_per_split, int nsplit) {
  constexpr int PLA = TA * 32, PLB = TB * 32;                               // bytes of one plane of each operand (16 pixels)
  constexpr int NA = TA / 64, NB = TB / 64;                                 // 32-row tiles per wave
  constexpr int BUF = 3 * PLA + 3 * PLB;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * BUF];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int wm = w & 1, wn = w >> 1;
  const int tiles_n = Ci / TB, tiles = (Co / TA) * tiles_n;
  const int L = xcd_block();
  const int tile = L % tiles, split = L / tiles;
  if (split >= nsplit) return;
  const int co0 = (tile / tiles_n) * TA, ci0 = (tile % tiles_n) * TB;
  const int mbeg = split * rows_per_split, mend = min(M, mbeg + rows_per_split);
  // Round 6: M (and so the last share) need not be a multiple of 128 pixels -- the operands are read through BUFFER descriptors of the
  // share's valid rows, a pixel past the end loads zeros and adds nothing; whole groups of 8 chunks as before
  const int nrows = mend - mbeg;
  const int nch = (nrows + 8 * B6_KC - 1) / (8 * B6_KC) * 8;

  // staging role: operand (gradient rows = output channels | input rows = input channels), pixel group pg (8 pixels), channel pair cp
  const bool isB = tid >= TA;
  const int st = isB ? tid - TA : tid;
  const bool stage = (TA + TB == 256) || st < (isB ? TB : TA);      // 128 x 128 tiles: every thread stages (no branch in the chunk body)
  const int pg = st & 1, cp = st >> 1;                                      // (consecutive lanes: the two pixel groups of one channel pair)
  const int sstride = isB ? xs : Co;
  const float* src0 = (isB ? X + ci0 : GY + co0) + static_cast<size_t>(mbeg) * sstride;     // (uniform per wave: TA is a multiple of 64)
  const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(reinterpret_cast<const float*>(b6_uniform64(reinterpret_cast<unsigned long long>(src0)))), 0,
      __builtin_amdgcn_readfirstlane(nrows * sstride * 4), 0x00020000);
  const unsigned svoff = static_cast<unsigned>((8 * pg * sstride + 2 * cp) * 4);              // byte offset of (pixel 8 pg, channel 2 cp)
  const unsigned srow = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(sstride * 4));
  unsigned char* sbase = smem + (isB ? 3 * PLA : 0);
  const int PL = isB ? PLB : PLA;
  int soff[2];
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const int row = 2 * cp + c;
    soff[c] = row * 32 + ((pg ^ ((row >> 3) & 1)) * 16);
  }
  floatx2 ra0, ra1, ra2, ra3, ra4, ra5, ra6, ra7, rb0, rb1, rb2, rb3, rb4, rb5, rb6, rb7;      // two chunks in flight
#define B6W2_LD1(k_) __builtin_bit_cast(floatx2, __builtin_amdgcn_raw_buffer_load_b64(srs, svoff + (k_) * srow, so_, 0))
#define B6W2_LOAD(q_, ch_)                                                                                           \
  do {                                                                                                               \
    const int cq_ = (ch_) < nch ? (ch_) : nch - 1;                                                                   \
    const unsigned so_ = static_cast<unsigned>(cq_) * B6_KC * srow;                                                  \
    if (stage) {                                                                                                     \
      r##q_##0 = B6W2_LD1(0);                                                                                        \
      r##q_##1 = B6W2_LD1(1);                                                                                        \
      r##q_##2 = B6W2_LD1(2);                                                                                        \
      r##q_##3 = B6W2_LD1(3);                                                                                        \
      r##q_##4 = B6W2_LD1(4);                                                                                        \
      r##q_##5 = B6W2_LD1(5);                                                                                        \
      r##q_##6 = B6W2_LD1(6);                                                                                        \
      r##q_##7 = B6W2_LD1(7);                                                                                        \
    }                                                                                                                \
  } while (0)
  // channel c_ of the pair: its 8 pixel values -> one 16-byte piece per plane
#define B6W2_STAGE1(q_, c_, dst_)                                                                                    \
  do {                                                                                                               \
    unsigned a0, a1, a2, b0, b1, b2, c0, c1, c2, d0, d1, d2;                                                         \
    b6_split2s(r##q_##0[c_], r##q_##1[c_], a0, a1, a2);                                                              \
    b6_split2s(r##q_##2[c_], r##q_##3[c_], b0, b1, b2);                                                              \
    b6_split2s(r##q_##4[c_], r##q_##5[c_], c0, c1, c2);                                                              \
    b6_split2s(r##q_##6[c_], r##q_##7[c_], d0, d1, d2);                                                              \
    *reinterpret_cast<uintx4*>((dst_) + soff[c_]) = uintx4{a0, b0, c0, d0};                                          \
    *reinterpret_cast<uintx4*>((dst_) + PL + soff[c_]) = uintx4{a1, b1, c1, d1};                                     \
    *reinterpret_cast<uintx4*>((dst_) + 2 * PL + soff[c_]) = uintx4{a2, b2, c2, d2};                                 \
  } while (0)
#define B6W2_STAGE(q_, buf_)                                                                                         \
  do {                                                                                                               \
    if (stage) {                                                                                                     \
      unsigned char* d_ = sbase + (buf_) * BUF;                                                                      \
      B6W2_STAGE1(q_, 0, d_);                                                                                        \
      B6W2_STAGE1(q_, 1, d_);                                                                                        \
    }                                                                                                                \
  } while (0)

  floatx16 acc[NA][NB];
  float tot[NA][NB][16];
  const floatx16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int a = 0; a < NA; ++a)
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) tot[a][b][r] = 0.f;
  int aaddr[NA], baddr[NB];
#pragma unroll
  for (int a = 0; a < NA; ++a) {
    const int ra = (wm * NA + a) * 32 + l31;
    aaddr[a] = ra * 32 + ((half ^ ((ra >> 3) & 1)) * 16);
  }
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const int rb = (wn * NB + b) * 32 + l31;
    baddr[b] = 3 * PLA + rb * 32 + ((half ^ ((rb >> 3) & 1)) * 16);
  }

  B6W2_LOAD(a, 0);
  B6W2_LOAD(b, 1);
  B6W2_STAGE(a, 0);
  B6W2_LOAD(a, 2);
  __syncthreads();
  __builtin_amdgcn_sched_barrier(0);
  // chunk j_ of a group (ch = g + j_): LDS buffer j_ & 1; raw set (j_ + 1) & 1 = chunk ch + 1 (staged here, then refilled with ch + 3)
#define B6W2_CHUNK(j_, first_, last_)                                                                                \
  do {                                                                                                               \
    const int ch = g + (j_);                                                                                         \
    const unsigned char* rd_ = smem + ((j_) & 1) * BUF;                                                              \
    bf16x8 af[NA][3], bf[NB][3];                                                                                     \
    _Pragma("unroll") for (int p = 2; p >= 0; --p) {                                                                  \
      _Pragma("unroll") for (int a = 0; a < NA; ++a) af[a][p] = *reinterpret_cast<const bf16x8*>(rd_ + p * PLA + aaddr[a]); \
      _Pragma("unroll") for (int b = 0; b < NB; ++b) bf[b][p] = *reinterpret_cast<const bf16x8*>(rd_ + p * PLB + baddr[b]); \
    }                                                                                                                \
    if (((j_) & 1) == 0) { B6W2_STAGE(b, 1); B6W2_LOAD(b, ch + 3); } else { B6W2_STAGE(a, 0); B6W2_LOAD(a, ch + 3); } \
    _Pragma("unroll") for (int t = 0; t < 6; ++t) {                                                                   \
      const int pa = t == 0 ? 2 : (t == 1 || t == 3) ? 1 : 0, pb = t == 0 || t == 3 || t == 5 ? 0 : (t == 1 || t == 4) ? 1 : 2; \
      _Pragma("unroll") for (int a = 0; a < NA; ++a)                                                                  \
        _Pragma("unroll") for (int b = 0; b < NB; ++b)                                                                \
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][pa], bf[b][pb], (first_) && t == 0 ? zero16 : acc[a][b], 0, 0, 0); \
    }                                                                                                                \
    if (last_) {                                                                                                     \
      _Pragma("unroll") for (int a = 0; a < NA; ++a)                                                                  \
        _Pragma("unroll") for (int b = 0; b < NB; ++b)                                                                \
          _Pragma("unroll") for (int r = 0; r < 16; ++r) tot[a][b][r] = b6_add(tot[a][b][r], acc[a][b][r]);           \
    }                                                                                                                \
    B6_SGB(0x100, 3 * (NA + NB));                                                                                    \
    _Pragma("unroll") for (int i = 0; i < 6 * NA * NB; ++i) {                                                         \
      B6_SGB(0x008, 1);                                                                                              \
      B6_SGB(0x002, NA * NB == 4 ? B6W_VG : (NA * NB == 2 ? 10 : 20));                                               \
      if (NA * NB == 4 ? (i == B6W_H - 2 || i == B6W_H || i == B6W_H + 2) : (i == 6 * NA * NB / 2 - 2 || i == 6 * NA * NB / 2 || i == 6 * NA * NB / 2 + 2)) B6_SGB(0x200, 1); \
      if (NA * NB == 4 ? (i >= B6W_T && i < B6W_T + 3) : (i >= 6 * NA * NB - 4 && i < 6 * NA * NB - 1)) B6_SGB(0x200, 1); \
      if (i == 6 * NA * NB - 2 || i == 6 * NA * NB - 1) B6_SGB(0x020, 4);                                            \
    }                                                                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                               \
    __syncthreads();                                                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                                               \
  } while (0)
  for (int g = 0; g < nch; g += 8) {
    B6W2_CHUNK(0, true, false);
    B6W2_CHUNK(1, false, false);
    B6W2_CHUNK(2, false, false);
    B6W2_CHUNK(3, false, false);
    B6W2_CHUNK(4, false, false);
    B6W2_CHUNK(5, false, false);
    B6W2_CHUNK(6, false, false);
    B6W2_CHUNK(7, false, true);
  }
#undef B6W2_CHUNK
#undef B6W2_STAGE
#undef B6W2_STAGE1
#undef B6W2_LOAD
#undef B6W2_LD1
#pragma unroll
  for (int a = 0; a < NA; ++a)
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      float* o = part + (static_cast<size_t>(split) * Co + co0 + (wm * NA + a) * 32 + 4 * half) * Ci + ci0 + (wn * NB + b) * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) o[static_cast<size_t>((r & 3) + 8 * (r >> 2)) * Ci] = tot[a][b][r];
    }
}

// dW = sum of the shares, four interleaved chains then pairwise (fixed order)
__global__ __launch_bounds__(256) void b6_wsum(const float* __restrict__ part, int nsplit, long long plane, float* __restrict__ out) {
  const long long idx = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x;
  if (idx >= plane) return;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  const float* ps = part + idx;
  int s = 0;
  for (; s + 4 <= nsplit; s += 4) {
    s0 += ps[(s + 0) * plane];
    s1 += ps[(s + 1) * plane];
    s2 += ps[(s + 2) * plane];
    s3 += ps[(s + 3) * plane];
  }
  for (; s < nsplit; ++s) s0 += ps[s * plane];
  out[idx] = (s0 + s1) + (s2 + s3);
}

struct B6WPlan { int nsplit, rows, grid, ta, tb; };

bool b6w_plan(long long M, int Ci, int Co, int xs, B6WPlan* p) {
  if (M <= 0 || M > 0x7fffffffLL - 128 || M * static_cast<long long>(xs > Co ? xs : Co) * 4 >= (1LL << 32) || Ci <= 0 || (Ci % 64) || Co <= 0 || (Co % 64) || xs < Ci || (xs % 4)) return false;
  p->ta = (Co % 128) == 0 ? 128 : 64;
  p->tb = (Ci % 128) == 0 ? 128 : 64;
  const int tiles = (Co / p->ta) * (Ci / p->tb);
  long long ns = (2 * DBEV_NUM_CU) / tiles;                  // two workgroups per CU, one round
  const long long chunks = (M + B6W_KC - 1) / B6W_KC;         // (round 6: the last chunk may be partly past M -- zeros)
  if (ns > chunks / 8) ns = chunks / 8;                      // a share reduces at least 8 chunks
  if (ns < 1) ns = 1;
  long long per = (chunks + ns - 1) / ns;
  per = (per + 3) / 4 * 4;                                   // whole groups of four chunks
  ns = (chunks + per - 1) / per;
  p->nsplit = static_cast<int>(ns);
  p->rows = static_cast<int>(per * B6W_KC);
  p->grid = dbev_round_xcd(tiles * p->nsplit);
  return true;
}

// tile_n: 128 / 64 columns per workgroup tile; 0 = 128 when N allows it (64 gives twice the workgroups: layers with few rows)
int b6_bn(int N, int tile_n) { return (tile_n == 64 || (N % 128) != 0) ? 64 : 128; }

bool b6_ok(long long M, int K, int N, int xs) {
  return M > 0 && M <= 0x7fffffffLL - B6_BM && K > 0 && (K % 64) == 0 && N > 0 && (N % 64) == 0 && xs >= K && (xs % 4) == 0 &&
         M * static_cast<long long>(xs > N ? xs : N) < (1LL << 40);
}

}  // namespace

extern "C" long long dbev_gemm_bf16x6_packed_bytes(int N, int K) {
  if (N <= 0 || K <= 0 || (N % 64) || (K % 64)) return 0;
  return 3LL * N * K * 2;
}

extern "C" int dbev_gemm_bf16x6_pack(const float* weight, long long stride_n, long long stride_k, int N, int K, int tile_n, void* packed,
                                     dbevStream_t stream) {
  if (dbev_gemm_bf16x6_packed_bytes(N, K) == 0 || weight == nullptr || packed == nullptr || (tile_n != 0 && tile_n != 64 && tile_n != 128) ||
      (tile_n == 128 && (N % 128) != 0))
    return DBEV_EINVAL;
  const long long threads = static_cast<long long>(N) * (K / 8);
  hipLaunchKernelGGL(b6_pack, dim3(dbev_ceil_div(threads, 256)), dim3(256), 0, dbev_stream(stream), weight, stride_n, stride_k, N, K,
                     b6_bn(N, tile_n), static_cast<unsigned short*>(packed));
  DBEV_LAUNCH_CHECK();
  return 0;
}

extern "C" int dbev_gemm_bf16x6_pack_pair(const float* weight, long long stride_o, long long stride_c, int Cout, int Cin, int tile_fwd,
                                          void* packed_fwd, int tile_dgrad, void* packed_dgrad, dbevStream_t stream) {
  const auto okt = [](int t, int n) { return t == 0 || t == 64 || (t == 128 && (n % 128) == 0); };
  if (dbev_gemm_bf16x6_packed_bytes(Cout, Cin) == 0 || dbev_gemm_bf16x6_packed_bytes(Cin, Cout) == 0 || weight == nullptr ||
      packed_fwd == nullptr || packed_dgrad == nullptr || !okt(tile_fwd, Cout) || !okt(tile_dgrad, Cin))
    return DBEV_EINVAL;
  const long long threads = static_cast<long long>(Cout) * Cin / 8;                  // the same count for both orientations
  hipLaunchKernelGGL(b6_pack_pair, dim3(dbev_ceil_div(threads, 256), 2), dim3(256), 0, dbev_stream(stream), weight, stride_o, stride_c, Cout,
                     Cin, b6_bn(Cout, tile_fwd), static_cast<unsigned short*>(packed_fwd), b6_bn(Cin, tile_dgrad),
                     static_cast<unsigned short*>(packed_dgrad));
  DBEV_LAUNCH_CHECK();
  return 0;
}

// every trainable 1x1 filter's planes in ONE launch (round 5): blockIdx.z = job; so / sc = strides of the [Cout, Cin] view,
// kind_a / kind_b = tile width (64 / 128) of the forward / data-gradient planes, 0: skip
__global__ __launch_bounds__(256) void b6_pack_multi(const dbevPackJob* __restrict__ jobs) {
  const dbevPackJob j = jobs[blockIdx.z];
  const long long idx = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x;
  if (blockIdx.y == 0) { if (j.kind_a) b6_pack_elem(idx, j.weight, j.so, j.sc, j.Cout, j.Cin, j.kind_a, static_cast<unsigned short*>(j.out_a)); }
  else if (j.kind_b) b6_pack_elem(idx, j.weight, j.sc, j.so, j.Cin, j.Cout, j.kind_b, static_cast<unsigned short*>(j.out_b));
}

extern "C" int dbev_gemm_bf16x6_pack_multi(const dbevPackJob* jobs_device, int n_jobs, long long max_units, dbevStream_t stream) {
  if (jobs_device == nullptr || n_jobs <= 0 || n_jobs > 65535 || max_units <= 0) return DBEV_EINVAL;
  hipLaunchKernelGGL(b6_pack_multi, dim3(dbev_ceil_div(max_units, 256), 2, n_jobs), dim3(256), 0, dbev_stream(stream), jobs_device);
  DBEV_LAUNCH_CHECK();
  return 0;
}

extern "C" int dbev_gemm_bf16x6_stats_rows(long long M) { return (M > 0 && M <= 0x7fffffffLL - B6_BM) ? static_cast<int>((M + B6_BM - 1) / B6_BM) : 0; }

namespace {
int b6_forward(const float* x, const void* packed, const float* bias, float* y, float* stats_partial, long long M, int K, int N,
               int x_row_stride, int tile_n, dbevStream_t stream);
}

extern "C" int dbev_gemm_bf16x6_forward_stats(const float* x, const void* packed, float* y, float* stats_partial, long long M, int K, int N,
                                              int x_row_stride, int tile_n, dbevStream_t stream) {
  return b6_forward(x, packed, nullptr, y, stats_partial, M, K, N, x_row_stride, tile_n, stream);
}

extern "C" int dbev_gemm_bf16x6_forward_bias(const float* x, const void* packed, const float* bias, float* y, long long M, int K, int N,
                                             int x_row_stride, int tile_n, dbevStream_t stream) {
  return b6_forward(x, packed, bias, y, nullptr, M, K, N, x_row_stride, tile_n, stream);
}

namespace {
int b6_forward(const float* x, const void* packed, const float* bias, float* y, float* stats_partial, long long M, int K, int N,
               int x_row_stride, int tile_n, dbevStream_t stream) {
  if (!b6_ok(M, K, N, x_row_stride) || x == nullptr || packed == nullptr || y == nullptr || (tile_n != 0 && tile_n != 64 && tile_n != 128) ||
      (tile_n == 128 && (N % 128) != 0))
    return DBEV_EINVAL;
  hipStream_t s = dbev_stream(stream);
  const int m = static_cast<int>(M);
  const int bn = b6_bn(N, tile_n);
  const int grid = dbev_round_xcd(((m + B6_BM - 1) / B6_BM) * (N / bn));
  DbevKt kt(DBEV_K_B6_FWD, 2LL * M * K * N, s);                  // the log's work field: fp32-equivalent FLOPs
  static const int dbg = getenv("DBEV_BF6_DBG") ? atoi(getenv("DBEV_BF6_DBG")) : 0;
  const unsigned short* pw = static_cast<const unsigned short*>(packed);
  static const int ver = getenv("DBEV_BF6_V") ? atoi(getenv("DBEV_BF6_V")) : 2;          // 1: the round-4 kernel (A/B runs)
  if (ver != 2 && ((M % B6_BM) != 0 || bias != nullptr)) return DBEV_EINVAL;   // (the round-4 kernel: whole 128-row blocks, no bias)
  if (ver == 2) {
#define B6_GO2(BNV, ST)                                                                                                            \
  do {                                                                                                                             \
    constexpr int lds_ = 2 * 3 * B6_BM * 32 + 4 * 3 * BNV * 32;                                                                    \
    static bool once_ = false;                                                                                                     \
    if (!once_) {                                                                                                                  \
      DBEV_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(b6_fwd2<BNV, ST>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_)); \
      once_ = true;                                                                                                                \
    }                                                                                                                              \
    hipLaunchKernelGGL((b6_fwd2<BNV, ST>), dim3(grid), dim3(256), lds_, s, x, pw, y, stats_partial, m, K, N, x_row_stride,         \
                       B6Conv{}, bias);                                                                                           \
  } while (0)
    if (bn == 128) { if (stats_partial != nullptr) B6_GO2(128, true); else B6_GO2(128, false); }
    else { if (stats_partial != nullptr) B6_GO2(64, true); else B6_GO2(64, false); }
#undef B6_GO2
    DBEV_LAUNCH_CHECK();
    return 0;
  }
#define B6_GO(BNV, ST) hipLaunchKernelGGL((b6_fwd<BNV, 2, ST>), dim3(grid), dim3(256), 0, s, x, pw, y, stats_partial, m, K, N, x_row_stride, dbg)
  if (bn == 128) { if (stats_partial != nullptr) B6_GO(128, true); else B6_GO(128, false); }
  else { if (stats_partial != nullptr) B6_GO(64, true); else B6_GO(64, false); }
#undef B6_GO
  DBEV_LAUNCH_CHECK();
  return 0;
}
}  // namespace

extern "C" int dbev_gemm_bf16x6_forward(const float* x, const void* packed, float* y, long long M, int K, int N, int x_row_stride,
                                        int tile_n, dbevStream_t stream) {
  return dbev_gemm_bf16x6_forward_stats(x, packed, y, nullptr, M, K, N, x_row_stride, tile_n, stream);
}

namespace {
bool c3s2_ok(int N, int H, int W, int C, int Co) {
  if (N <= 0 || H < 2 || W < 2 || (H & 1) || (W & 1) || C < 64 || (C & (C - 1)) != 0 || Co <= 0 || Co % 64) return false;   // C = 64, 128, 256 ...
  const long long M = static_cast<long long>(N) * (H / 2) * (W / 2);
  return static_cast<long long>(N) * H * W * C < (1LL << 31) && M * Co < (1LL << 31) && b6_ok(M, 9 * C, Co, 9 * C);
}
}  // namespace

extern "C" int dbev_conv3x3s2_bf16x6_ok(int N, int H, int W, int C, int Co) { return c3s2_ok(N, H, W, C, Co) ? 1 : 0; }

extern "C" int dbev_conv3x3s2_bf16x6_forward_stats(const float* x_nhwc, const void* packed, float* y_nhwc, float* stats_partial, int N, int H,
                                                   int W, int C, int Co, int tile_n, dbevStream_t stream) {
  if (!c3s2_ok(N, H, W, C, Co) || x_nhwc == nullptr || packed == nullptr || y_nhwc == nullptr ||
      (tile_n != 0 && tile_n != 64 && tile_n != 128) || (tile_n == 128 && (Co % 128) != 0))
    return DBEV_EINVAL;
  hipStream_t s = dbev_stream(stream);
  B6Conv cv{H, W, C, H / 2, W / 2, 0};
  while ((16 << cv.cshift) < C) ++cv.cshift;
  const int m = N * cv.Ho * cv.Wo, K = 9 * C;
  const int bn = b6_bn(Co, tile_n);
  const int grid = dbev_round_xcd(((m + B6_BM - 1) / B6_BM) * (Co / bn));
  DbevKt kt(DBEV_K_B6_FWD, 2LL * m * K * Co, s);
  const unsigned short* pw = static_cast<const unsigned short*>(packed);
#define B6_GOC(BNV, ST)                                                                                                            \
  do {                                                                                                                             \
    constexpr int lds_ = 2 * 3 * B6_BM * 32 + 4 * 3 * BNV * 32;                                                                    \
    static bool once_ = false;                                                                                                     \
    if (!once_) {                                                                                                                  \
      DBEV_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(b6_fwd2<BNV, ST, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_)); \
      once_ = true;                                                                                                                \
    }                                                                                                                              \
    hipLaunchKernelGGL((b6_fwd2<BNV, ST, true>), dim3(grid), dim3(256), lds_, s, x_nhwc, pw, y_nhwc, stats_partial, m, K, Co, K, cv); \
  } while (0)
  if (bn == 128) { if (stats_partial != nullptr) B6_GOC(128, true); else B6_GOC(128, false); }
  else { if (stats_partial != nullptr) B6_GOC(64, true); else B6_GOC(64, false); }
#undef B6_GOC
  DBEV_LAUNCH_CHECK();
  return 0;
}

extern "C" size_t dbev_gemm_bf16x6_backward_weight_workspace_bytes(long long M, int Cin, int Cout, int x_row_stride) {
  B6WPlan p;
  if (!b6w_plan(M, Cin, Cout, x_row_stride, &p)) return 0;
  return static_cast<size_t>(p.nsplit) * Cin * Cout * sizeof(float);
}

extern "C" int dbev_gemm_bf16x6_backward_weight(const float* x, const float* grad_y, float* grad_weight, long long M, int Cin, int Cout,
                                                int x_row_stride, void* workspace, size_t workspace_bytes, dbevStream_t stream) {
  B6WPlan p;
  if (!b6w_plan(M, Cin, Cout, x_row_stride, &p) || x == nullptr || grad_y == nullptr || grad_weight == nullptr || workspace == nullptr ||
      workspace_bytes < static_cast<size_t>(p.nsplit) * Cin * Cout * sizeof(float))
    return DBEV_EINVAL;
  hipStream_t s = dbev_stream(stream);
  float* part = static_cast<float*>(workspace);
  DbevKt kt(DBEV_K_B6_WGRAD, 2LL * M * Cin * Cout, s);
  static const int ver = getenv("DBEV_BF6_WV") ? atoi(getenv("DBEV_BF6_WV")) : 2;         // 1: the round-4 kernel (A/B runs)
  if (ver != 2 && (M % 128) != 0) return DBEV_EINVAL;                                    // (the round-4 kernel has no partial chunks)
  const bool v2 = ver == 2;
#define B6W_GO(TAV, TBV)                                                                                                          \
  do {                                                                                                                            \
    if (v2) hipLaunchKernelGGL((b6_wgrad2<TAV, TBV>), dim3(p.grid), dim3(256), 0, s, grad_y, x, p.nsplit > 1 ? part : grad_weight, \
                               static_cast<int>(M), Cout, Cin, x_row_stride, p.rows, p.nsplit);                                   \
    else hipLaunchKernelGGL((b6_wgrad<TAV, TBV>), dim3(p.grid), dim3(256), 0, s, grad_y, x, p.nsplit > 1 ? part : grad_weight,    \
                            static_cast<int>(M), Cout, Cin, x_row_stride, p.rows, p.nsplit);                                      \
  } while (0)
  if (p.ta == 128) { if (p.tb == 128) B6W_GO(128, 128); else B6W_GO(128, 64); }
  else { if (p.tb == 128) B6W_GO(64, 128); else B6W_GO(64, 64); }
#undef B6W_GO
  DBEV_LAUNCH_CHECK();
  if (p.nsplit > 1) {
    const long long plane = static_cast<long long>(Cin) * Cout;
    hipLaunchKernelGGL(b6_wsum, dim3(dbev_ceil_div(plane, 256)), dim3(256), 0, s, part, p.nsplit, plane, grad_weight);
    DBEV_LAUNCH_CHECK();
  }
  return 0;
}
