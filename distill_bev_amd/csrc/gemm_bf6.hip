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
// on the fly (5.5 VALU operations per element, amortised over the tile's columns).  The weights are split once per step by b6_pack.
//
//   Y[M, N] = X[M, K] * W[N, K]^T     X rows of x_stride floats, Y row-major, M % 256 == 0, K % 32 == 0, N % 64 == 0
//
// Workgroup = 8 waves (two per SIMD), tile 256 rows x BN columns (BN = 128: waves 4 x 2, wave tile 64 x 64; BN = 64: waves 8 x 1, wave
// tile 32 x 64), reduction in chunks of 32, double-buffered in LDS as three bf16 planes per operand ([plane][row][32 k], the four 16-byte
// units of a row XOR-swizzled by (row >> 2) & 3: the 16 lanes of a ds_read_b128 service group hit 16 distinct bank quads).  Per chunk and
// wave: 12 (BN = 64: 9) ds_read_b128 per reduction step of 16, 48 (24) MFMAs, the next chunk's activations fetched into registers
// under them, split and written to the other buffer behind them, one barrier.
#include "common.h"

#include <stdlib.h>

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int B6_BM = 128, B6_KC = 16;

__device__ __forceinline__ unsigned b6_rne(float x) {                    // bf16(x), round to nearest even, as the upper half of a dword
  const unsigned u = __float_as_uint(x);
  return (u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u;
}

// ---- weight split + packing: W element (n, k) at w[n * sn + k * sk] -> planes in the LDS image order -----------------------------
// packed (bf16 elements): ((((nb * nkc + kc) * 3 + plane) * BN + row) * 16) + (u ^ ((row >> 3) & 1)) * 8 + i,  n = nb BN + row,
// k = 16 kc + 8 u + i
__global__ __launch_bounds__(256) void b6_pack(const float* __restrict__ w, long long sn, long long sk, int N, int K, int BN,
                                               unsigned short* __restrict__ out) {
  const long long idx = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x;      // (n, k unit of 8)
  const int nk8 = K / 8;
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

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float floatx2 __attribute__((ext_vector_type(2)));

// (a, b) -> three dwords of two bf16 each (round to nearest even, v_cvt_pk_bf16_f32): a = p0.lo + p1.lo + p2.lo up to 2^-25 |a|
__device__ __forceinline__ void b6_split2(float a, float b, unsigned& p0, unsigned& p1, unsigned& p2) {
  floatx2 v = {a, b};
  bf16x2 h = __builtin_convertvector(v, bf16x2);
  p0 = *reinterpret_cast<unsigned*>(&h);
  floatx2 f = {__uint_as_float(p0 << 16), __uint_as_float(p0 & 0xffff0000u)};
  v = v - f;                                                           // exact
  h = __builtin_convertvector(v, bf16x2);
  p1 = *reinterpret_cast<unsigned*>(&h);
  f = floatx2{__uint_as_float(p1 << 16), __uint_as_float(p1 & 0xffff0000u)};
  v = v - f;
  h = __builtin_convertvector(v, bf16x2);
  p2 = *reinterpret_cast<unsigned*>(&h);
}

// LDS image of a chunk (16 reduction steps): [plane 3][row][2 units of 8 bf16], unit u of row r in slot u ^ ((r >> 3) & 1)
template <int BN, int OCC, int FL>
__global__ __launch_bounds__(256, OCC) void b6_fwd(const float* __restrict__ X, const unsigned short* __restrict__ Wp,
                                                   float* __restrict__ Y, int M, int K, int N, int xs) {
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

  // activation fetch: float4 f = tid + 256 i of the chunk (128 rows x 4 float4): row f / 4, floats 4 (f % 4) .. + 3
  const int row0 = tid >> 2, row1 = 64 + (tid >> 2), c4 = tid & 3;
  const float* xrow0 = X + static_cast<size_t>(m0 + row0) * xs + 4 * c4;
  const float* xrow1 = X + static_cast<size_t>(m0 + row1) * xs + 4 * c4;
  const int aoff0 = row0 * 32 + (((c4 >> 1) ^ ((row0 >> 3) & 1)) * 16) + (c4 & 1) * 8;
  const int aoff1 = row1 * 32 + (((c4 >> 1) ^ ((row1 >> 3) & 1)) * 16) + (c4 & 1) * 8;
  const uint4* wsrc = reinterpret_cast<const uint4*>(Wp) + static_cast<size_t>(nb) * nkc * BV + tid;
  // two chunks in flight in registers (set q = chunk & 1); named scalars: indexed arrays of these end up in scratch memory
  float4 xa0_0, xa0_1, xa1_0, xa1_1;
  uint4 wb0_0, wb0_1 = make_uint4(0, 0, 0, 0), wb0_2 = make_uint4(0, 0, 0, 0), wb1_0, wb1_1 = make_uint4(0, 0, 0, 0), wb1_2 = make_uint4(0, 0, 0, 0);
#define B6_FETCH(q_, kc_)                                                                                            \
  do {                                                                                                               \
    xa##q_##_0 = *reinterpret_cast<const float4*>(xrow0 + (kc_) * B6_KC);                                            \
    xa##q_##_1 = *reinterpret_cast<const float4*>(xrow1 + (kc_) * B6_KC);                                            \
    const uint4* ws_ = wsrc + static_cast<size_t>(kc_) * BV;                                                         \
    wb##q_##_0 = ws_[0];                                                                                             \
    if (BV == 768 || tid < 128) wb##q_##_1 = ws_[256];                                                               \
    if (BV == 768) wb##q_##_2 = ws_[512];                                                                            \
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
    uint4* b_ = reinterpret_cast<uint4*>(sB + (buf_) * BBUF) + tid;                                                  \
    b_[0] = wb##q_##_0;                                                                                              \
    if (BV == 768 || tid < 128) b_[256] = wb##q_##_1;                                                                \
    if (BV == 768) b_[512] = wb##q_##_2;                                                                             \
  } while (0)

  // acc: what the MFMAs accumulate into; every FL chunks it is added to `tot` by the VALU (round to nearest) and restarted from zero:
  // the matrix pipe's accumulation error grows with the number of instructions chained on one accumulator (measured 1.8e-6 of the
  // output scale at K = 2048 against 4.8e-7 for the fp32 kernels; with FL = 8, i.e. 48 chained instructions, 2-5e-7 at every K)
  floatx16 acc[TM][2], tot[FL > 0 ? TM : 1][2];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[a][b][r] = 0.f; if (FL > 0) tot[a][b][r] = 0.f; }

  B6_FETCH(0, 0);
  if (nkc > 1) B6_FETCH(1, 1);
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

#define B6_CHUNK(q_)                                                                                                 \
  do {                                                                                                               \
    const int cur = kc & 1;                                                                                          \
    const unsigned char* a_ = sA + cur * ABUF;                                                                       \
    const unsigned char* b_ = sB + cur * BBUF;                                                                       \
    bf16x8 af[TM][3], bf[2][3];                                                                                      \
    _Pragma("unroll") for (int a = 0; a < TM; ++a)                                                                    \
      _Pragma("unroll") for (int p = 0; p < 3; ++p) af[a][p] = *reinterpret_cast<const bf16x8*>(a_ + p * APL + aaddr[a]); \
    _Pragma("unroll") for (int b = 0; b < 2; ++b)                                                                     \
      _Pragma("unroll") for (int p = 0; p < 3; ++p) bf[b][p] = *reinterpret_cast<const bf16x8*>(b_ + p * BPL + baddr[b]); \
    /* chunk kc + 1 (registers, set 1 - q_) -> the other LDS buffer under the MFMAs; chunk kc + 2 -> registers, set q_ */ \
    if (kc + 1 < nkc) { if ((q_) == 0) B6_STAGE(1, cur ^ 1); else B6_STAGE(0, cur ^ 1); }                            \
    if (kc + 2 < nkc) { if ((q_) == 0) B6_FETCH(0, kc + 2); else B6_FETCH(1, kc + 2); }                              \
    _Pragma("unroll") for (int a = 0; a < TM; ++a)                                                                    \
      _Pragma("unroll") for (int b = 0; b < 2; ++b) {                                                                 \
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][2], bf[b][0], acc[a][b], 0, 0, 0);                 \
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][1], bf[b][1], acc[a][b], 0, 0, 0);                 \
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][0], bf[b][2], acc[a][b], 0, 0, 0);                 \
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][1], bf[b][0], acc[a][b], 0, 0, 0);                 \
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][0], bf[b][1], acc[a][b], 0, 0, 0);                 \
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][0], bf[b][0], acc[a][b], 0, 0, 0);                 \
      }                                                                                                              \
    if (FL > 0 && (kc % FL) == FL - 1) {                                                                             \
      _Pragma("unroll") for (int a = 0; a < TM; ++a)                                                                  \
        _Pragma("unroll") for (int b = 0; b < 2; ++b)                                                                 \
          _Pragma("unroll") for (int r = 0; r < 16; ++r) { tot[a][b][r] += acc[a][b][r]; acc[a][b][r] = 0.f; }       \
    }                                                                                                                \
    __syncthreads();                                                                                                 \
  } while (0)
  // chunk kc's registers were fetched into set kc & 1; the loop body is written for both parities so that every register
  // array index is a compile-time constant
  for (int kc = 0; kc < nkc; kc += 2) {
    B6_CHUNK(0);
    ++kc;
    if (kc < nkc) B6_CHUNK(1);
    --kc;
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
      for (int r = 0; r < 16; ++r) y[static_cast<size_t>((r & 3) + 8 * (r >> 2)) * N] = FL > 0 ? tot[a][b][r] + acc[a][b][r] : acc[a][b][r];
    }
}

int b6_bn(int N) { return (N % 128) == 0 ? 128 : 64; }

bool b6_ok(long long M, int K, int N, int xs) {
  return M > 0 && (M % B6_BM) == 0 && M <= 0x7fffffffLL && K > 0 && (K % B6_KC) == 0 && N > 0 && (N % 64) == 0 && xs >= K && (xs % 4) == 0 &&
         M * static_cast<long long>(xs > N ? xs : N) < (1LL << 40);
}

}  // namespace

extern "C" long long dbev_gemm_bf16x6_packed_bytes(int N, int K) {
  if (N <= 0 || K <= 0 || (N % 64) || (K % B6_KC)) return 0;
  return 3LL * N * K * 2;
}

extern "C" int dbev_gemm_bf16x6_pack(const float* weight, long long stride_n, long long stride_k, int N, int K, void* packed,
                                     dbevStream_t stream) {
  if (dbev_gemm_bf16x6_packed_bytes(N, K) == 0 || weight == nullptr || packed == nullptr) return DBEV_EINVAL;
  const long long threads = static_cast<long long>(N) * (K / 8);
  hipLaunchKernelGGL(b6_pack, dim3(dbev_ceil_div(threads, 256)), dim3(256), 0, dbev_stream(stream), weight, stride_n, stride_k, N, K,
                     b6_bn(N), static_cast<unsigned short*>(packed));
  DBEV_LAUNCH_CHECK();
  return 0;
}

extern "C" int dbev_gemm_bf16x6_forward(const float* x, const void* packed, float* y, long long M, int K, int N, int x_row_stride,
                                        dbevStream_t stream) {
  if (!b6_ok(M, K, N, x_row_stride) || x == nullptr || packed == nullptr || y == nullptr) return DBEV_EINVAL;
  hipStream_t s = dbev_stream(stream);
  const int m = static_cast<int>(M);
  const int bn = b6_bn(N);
  const int grid = dbev_round_xcd((m / B6_BM) * (N / bn));
  DbevKt kt(DBEV_K_GEMM1X1_FWD, 2LL * M * K * N, s);
  static const int occ = getenv("DBEV_BF6_OCC") ? atoi(getenv("DBEV_BF6_OCC")) : 2;
  const unsigned short* pw = static_cast<const unsigned short*>(packed);
  static const int fl = getenv("DBEV_BF6_FLUSH") ? atoi(getenv("DBEV_BF6_FLUSH")) : 8;
#define B6_GO(BNV, OV, FV) hipLaunchKernelGGL((b6_fwd<BNV, OV, FV>), dim3(grid), dim3(256), 0, s, x, pw, y, m, K, N, x_row_stride)
  if (bn == 128) { if (fl == 0) B6_GO(128, 2, 0); else if (fl == 4) B6_GO(128, 2, 4); else B6_GO(128, 2, 8); }
  else { if (fl == 0) B6_GO(64, 2, 0); else if (fl == 4) B6_GO(64, 2, 4); else B6_GO(64, 2, 8); }
  (void)occ;
#undef B6_GO
  DBEV_LAUNCH_CHECK();
  return 0;
}
