// Fused student adaptation (1x1 convolution) + foreground-masked MSE of the FGD distillation loss, 'head' position.
//
// Reference sequence (mmdet3d/models/detectors/bevdet_distill.py):
//   student_feat = self.channel_wise_adaptations[index](student_feat)            :1003-1004  (nn.Conv2d(256, 384, 1) :232-234)
//   s_attention  = softmax(mean_c |student_feat| / T) * HW                       :1089-1092
//   kd_fg / kd_bg / kd_fp = sum((student_feat - teacher_feat)^2 * mask_k) * w_k  :1253-1262, :1282-1287
//   kd_spatial   = L1(mean_c teacher, conv3x3(mean_c student_feat))              :1272-1278
// = one 25.8 GFLOP GEMM (bs 8) whose 201 MB output is written once and read three more times.
//
// Every mask of the reference factors into (per-pixel weight) x (optional per-channel weight Cc of the fp term), so the
// three sums are  sum_p w_k(p) * E(p)  with  E(p) = sum_c d^2,  Efp(p) = sum_c Cc[c] d^2,  d = s - t.  This kernel computes
// the GEMM on the fp32 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32 FMA chains, the reference's precision) and
// reduces E, Efp, mean_c |s| (the attention input) and mean_c s (the spatial-term input) per pixel in the epilogue.
// The adapted student tensor is never written; the difference d is (it is all the backward needs).
//
// Mapping (wave64, gfx950): a workgroup of 4 waves owns 128 pixels x NT*32 output channels; MFMA rows = channels
// (A operand = weight tile), MFMA columns = pixels (B operand = activation tile), so a lane holds ONE pixel column and
// 16 channel rows per 32x32 tile: the per-pixel reductions are in-register sums plus one cross-half shuffle, and
// teacher / difference rows move as float4 (4 consecutive channels = accumulator registers 4q..4q+3).
// K is walked in chunks of 32 through a double-buffered, k-major LDS stage (conflict-free ds_read_b32 for both
// operands, one barrier per chunk); global loads of chunk c+1 are in flight while chunk c is multiplied; the loader
// reads whole 128-byte row chunks per 8 lanes (a 16-byte-per-row mapping thrashed the 32 KB L1: 38 % of MFMA peak).
// Teacher and difference tiles go through LDS so that both move as whole rows.
#include "common.h"

#include <stdlib.h>

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int AM_PXB = 128;   // pixels per workgroup (4 waves x 32)
constexpr int AM_KC = 32;     // K chunk

constexpr int AM_STR = AM_PXB + 1;   // LDS row stride of the k-major operand stages (pad 1: the coalesced loader's writes spread over banks)

template <int NT, int KC = AM_KC>
__global__ __launch_bounds__(256, (NT >= 3 ? 2 : (KC <= 16 ? 4 : 3))) void adapt_mse_fwd(const float* __restrict__ X, const float* __restrict__ Wt,
                                                         const float* __restrict__ bias, const float* __restrict__ T,
                                                         const float* __restrict__ Cc, float* __restrict__ D,
                                                         float* __restrict__ maps, int M, int K, int N, int HW) {
  constexpr int NCH = NT * 32;
  constexpr int WSTR = NCH + 1;
  constexpr int TSTR = NCH + 4;                               // epilogue tile [pixel][channel], 16-byte aligned rows
  constexpr int OPER = 2 * KC * (AM_STR + WSTR);           // floats of the two double-buffered operand stages
  constexpr int TILE = AM_PXB * TSTR;
  __shared__ float smem[OPER > TILE ? OPER : TILE];
  float (*sX)[KC][AM_STR] = reinterpret_cast<float (*)[KC][AM_STR]>(smem);
  float (*sW)[KC][WSTR] = reinterpret_cast<float (*)[KC][WSTR]>(smem + 2 * KC * AM_STR);
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  // XCD-aware order: the N / NCH channel slices of one pixel tile re-read the same 128 x K activation rows; logical block id L walks
  // (tile, slice) with the slice fastest and xcd_block() keeps consecutive L on ONE XCD back to back, so the second and third read
  // of a tile come out of that XCD's L2 instead of HBM (a (tiles, slices) grid put them ~1000 workgroups apart: 1.78x the bytes)
  const int nsl = N / NCH;
  const int L = xcd_block();
  const int tile_id = L / nsl, slice_id = L - tile_id * nsl;
  if (tile_id * AM_PXB >= M) return;                          // grid rounded up to a multiple of 8 XCDs
  const int m0 = tile_id * AM_PXB, n0 = slice_id * NCH;
  const int half = lane >> 5, l31 = lane & 31;

  // loader: 8 consecutive lanes read one row's 128-byte chunk (whole cache lines per wave instruction)
  constexpr int KQ = KC / 4;                                  // float4 per row chunk
  constexpr int XV = AM_PXB * KC / 4 / 256;                // float4 loads per thread per chunk
  constexpr int WV = (NCH * KC / 4 + 255) / 256;
  float4 rx[XV], rw[WV];
  auto load_chunk = [&](int kc) {
#pragma unroll
    for (int i = 0; i < XV; ++i) {
      const int f = tid + 256 * i, px = f / KQ, kq = f % KQ;
      const int m = m0 + px;
      rx[i] = m < M ? *reinterpret_cast<const float4*>(X + static_cast<size_t>(m) * K + kc * KC + 4 * kq)
                    : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < WV; ++i) {
      const int f = tid + 256 * i;
      if (f < NCH * KC / 4) {
        const int ch = f / KQ, kq = f % KQ;
        rw[i] = *reinterpret_cast<const float4*>(Wt + static_cast<size_t>(n0 + ch) * K + kc * KC + 4 * kq);
      }
    }
  };
  auto store_chunk = [&](int buf) {
#pragma unroll
    for (int i = 0; i < XV; ++i) {
      const int f = tid + 256 * i, px = f / KQ, kq = f % KQ;
      sX[buf][4 * kq + 0][px] = rx[i].x; sX[buf][4 * kq + 1][px] = rx[i].y;
      sX[buf][4 * kq + 2][px] = rx[i].z; sX[buf][4 * kq + 3][px] = rx[i].w;
    }
#pragma unroll
    for (int i = 0; i < WV; ++i) {
      const int f = tid + 256 * i;
      if (f < NCH * KC / 4) {
        const int ch = f / KQ, kq = f % KQ;
        sW[buf][4 * kq + 0][ch] = rw[i].x; sW[buf][4 * kq + 1][ch] = rw[i].y;
        sW[buf][4 * kq + 2][ch] = rw[i].z; sW[buf][4 * kq + 3][ch] = rw[i].w;
      }
    }
  };

  floatx16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  const int nchunk = K / KC;
  load_chunk(0);
  store_chunk(0);
  __syncthreads();
  for (int c = 0; c < nchunk; ++c) {
    const int buf = c & 1;
    if (c + 1 < nchunk) load_chunk(c + 1);                   // in flight during the MFMAs below
#pragma unroll
    for (int kk = 0; kk < KC / 2; ++kk) {
      const int k = 2 * kk + half;
      const float b = sX[buf][k][32 * w + l31];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const float a = sW[buf][k][32 * t + l31];
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
      }
    }
    if (c + 1 < nchunk) store_chunk(buf ^ 1);                // the other buffer: its readers passed the previous barrier
    __syncthreads();
  }

  // ---- epilogue ----------------------------------------------------------------------------------------------------
  // 1. the teacher tile [128 pixels][NCH channels] comes in with whole-row coalesced float4 loads into LDS (the operand
  //    stages are free now); 2. every lane (= one pixel column of the accumulators) reads its teacher values from LDS,
  //    forms d = s - t and the per-pixel sums in registers and puts d back; 3. the d tile leaves with coalesced stores.
  float* tile = smem;
  constexpr int C4 = NCH / 4;                                 // float4 per row of the slice
  for (int f = tid; f < AM_PXB * C4; f += 256) {
    const int px = f / C4, c4 = f - px * C4;
    const int m = m0 + px;
    const float4 tv = m < M ? *reinterpret_cast<const float4*>(T + static_cast<size_t>(m) * N + n0 + 4 * c4)
                            : make_float4(0.f, 0.f, 0.f, 0.f);
    *reinterpret_cast<float4*>(&tile[px * TSTR + 4 * c4]) = tv;
  }
  __syncthreads();
  const int pl = 32 * w + l31;                                // pixel of this lane inside the tile
  const int p = m0 + pl;
  const bool live = p < M;
  const int bidx = live ? p / HW : 0;
  float e = 0.f, efp = 0.f, sa = 0.f, sp = 0.f;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int cl = 32 * t + 8 * q + 4 * half;               // accumulator rows 4q..4q+3 = channels n0 + cl + (0..3)
      const float4 bi = *reinterpret_cast<const float4*>(bias + n0 + cl);
      float4 cw = make_float4(1.f, 1.f, 1.f, 1.f);
      if (Cc != nullptr) cw = *reinterpret_cast<const float4*>(Cc + static_cast<size_t>(bidx) * N + n0 + cl);
      float4* tp = reinterpret_cast<float4*>(&tile[pl * TSTR + cl]);
      const float4 tv = *tp;
      const float s0 = acc[t][4 * q + 0] + bi.x, s1 = acc[t][4 * q + 1] + bi.y;
      const float s2 = acc[t][4 * q + 2] + bi.z, s3 = acc[t][4 * q + 3] + bi.w;
      float4 d;
      d.x = s0 - tv.x; d.y = s1 - tv.y; d.z = s2 - tv.z; d.w = s3 - tv.w;
      *tp = d;
      const float q0 = d.x * d.x, q1 = d.y * d.y, q2 = d.z * d.z, q3 = d.w * d.w;
      e += (q0 + q1) + (q2 + q3);
      efp += (cw.x * q0 + cw.y * q1) + (cw.z * q2 + cw.w * q3);
      sa += (fabsf(s0) + fabsf(s1)) + (fabsf(s2) + fabsf(s3));
      sp += (s0 + s1) + (s2 + s3);
    }
  }
  // the other half-wave holds the other 16 rows of every tile for the same pixel
  e += __shfl_xor(e, 32); efp += __shfl_xor(efp, 32); sa += __shfl_xor(sa, 32); sp += __shfl_xor(sp, 32);
  if (live && half == 0) {
    float* mp = maps + static_cast<size_t>(slice_id) * 4 * M;
    mp[p] = e; mp[static_cast<size_t>(M) + p] = efp; mp[2 * static_cast<size_t>(M) + p] = sa; mp[3 * static_cast<size_t>(M) + p] = sp;
  }
  __syncthreads();
  for (int f = tid; f < AM_PXB * C4; f += 256) {
    const int px = f / C4, c4 = f - px * C4;
    const int m = m0 + px;
    if (m < M)
      *reinterpret_cast<float4*>(D + static_cast<size_t>(m) * N + n0 + 4 * c4) = *reinterpret_cast<const float4*>(&tile[px * TSTR + 4 * c4]);
  }
}

// dS[p, c] = 2 d[p, c] (gE[p] + gEfp[p] Cc[b, c]) + gP[p] / N   (gradient of the per-pixel maps wrt the adapted student)
__global__ __launch_bounds__(256) void adapt_mse_ds(const float4* __restrict__ D, const float* __restrict__ gE,
                                                    const float* __restrict__ gEfp, const float* __restrict__ gP,
                                                    const float* __restrict__ Cc, int M, int N4, int HW,
                                                    float4* __restrict__ dS) {
  const float invN = 1.f / static_cast<float>(N4 * 4);
  const size_t total = static_cast<size_t>(M) * N4;
  for (size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x; i < total; i += static_cast<size_t>(gridDim.x) * 256) {
    const int p = static_cast<int>(i / N4), q = static_cast<int>(i % N4);
    const float a = 2.f * gE[p];
    const float bb = gEfp != nullptr ? 2.f * gEfp[p] : 0.f;
    const float g = gP != nullptr ? gP[p] * invN : 0.f;
    float4 cw = make_float4(0.f, 0.f, 0.f, 0.f);
    if (gEfp != nullptr && Cc != nullptr) cw = reinterpret_cast<const float4*>(Cc + static_cast<size_t>(p / HW) * N4 * 4)[q];
    const float4 d = D[i];
    float4 o;
    o.x = fmaf(d.x, fmaf(bb, cw.x, a), g); o.y = fmaf(d.y, fmaf(bb, cw.y, a), g);
    o.z = fmaf(d.z, fmaf(bb, cw.z, a), g); o.w = fmaf(d.w, fmaf(bb, cw.w, a), g);
    dS[i] = o;
  }
}

int pick_nt(int N) {
  const int t = N / 32;
  if (N % 32) return 0;
  static const int force = getenv("DBEV_ADAPT_NT") ? atoi(getenv("DBEV_ADAPT_NT")) : 0;     // A/B runs
  if (force >= 1 && force <= 4 && t % force == 0) return force;
  // 2 tiles = 64 channels per workgroup: 114 VGPRs and 50 KB of LDS -> three workgroups per CU, whose epilogues (teacher tile in,
  // difference tile out) hide under each other's MFMAs; 4 tiles (210 VGPRs, 68 KB, two per CU) measured 7 % slower, 1 tile 11 %
  for (int nt : {2, 3, 4, 1})
    if (t % nt == 0) return nt;
  return 0;
}

}  // namespace

extern "C" int dbev_adapt_mse_map_slices(int Cs, int Ct) {
  const int nt = pick_nt(Ct);
  if (nt == 0 || Cs <= 0 || (Cs % AM_KC)) return 0;
  return Ct / (32 * nt);
}

extern "C" int dbev_adapt_mse_forward(const float* x_nhwc, const float* weight, const float* bias,
                                      const float* teacher_nhwc, const float* channel_weight, int B, int HW, int Cs,
                                      int Ct, float* diff_nhwc, float* maps, dbevStream_t stream) {
  const int nt = pick_nt(Ct);
  if (B <= 0 || HW <= 0 || nt == 0 || Cs <= 0 || (Cs % AM_KC) || x_nhwc == nullptr || weight == nullptr ||
      bias == nullptr || teacher_nhwc == nullptr || diff_nhwc == nullptr || maps == nullptr)
    return DBEV_EINVAL;
  const long long M = static_cast<long long>(B) * HW;
  if (M > 0x3fffffffLL) return DBEV_EINVAL;
  hipStream_t s = dbev_stream(stream);
  const dim3 grid(dbev_round_xcd(dbev_ceil_div(M, AM_PXB) * (Ct / (32 * nt))));
#define AM_LAUNCH(NTV) hipLaunchKernelGGL((adapt_mse_fwd<NTV>), grid, dim3(256), 0, s, x_nhwc, weight, bias, teacher_nhwc, \
                                         channel_weight, diff_nhwc, maps, static_cast<int>(M), Cs, Ct, HW)
  // 64-channel slices walk K in chunks of 16: 35 KB of LDS (the epilogue tile) and four workgroups per CU; chunks of 32 (50 KB, three
  // per CU) measured 3 % slower (DBEV_ADAPT_KC=32 for A/B runs)
  static const int kc_env = getenv("DBEV_ADAPT_KC") ? atoi(getenv("DBEV_ADAPT_KC")) : 16;
  if (nt == 2 && kc_env == 16 && Cs % 16 == 0) {
    hipLaunchKernelGGL((adapt_mse_fwd<2, 16>), grid, dim3(256), 0, s, x_nhwc, weight, bias, teacher_nhwc, channel_weight, diff_nhwc,
                       maps, static_cast<int>(M), Cs, Ct, HW);
    DBEV_LAUNCH_CHECK();
    return 0;
  }
  DbevKt kt(DBEV_K_ADAPT_MSE_FWD, 4LL * M * (Cs + 2LL * Ct), s);      // x + teacher read, difference written
  switch (nt) {
    case 4: AM_LAUNCH(4); break;
    case 3: AM_LAUNCH(3); break;
    case 2: AM_LAUNCH(2); break;
    default: AM_LAUNCH(1); break;
  }
#undef AM_LAUNCH
  DBEV_LAUNCH_CHECK();
  return 0;
}

extern "C" int dbev_adapt_mse_backward_ds(const float* diff_nhwc, const float* grad_e, const float* grad_efp,
                                          const float* grad_pool, const float* channel_weight, int B, int HW, int Ct,
                                          float* ds_nhwc, dbevStream_t stream) {
  if (B <= 0 || HW <= 0 || Ct <= 0 || (Ct & 3) || diff_nhwc == nullptr || grad_e == nullptr || ds_nhwc == nullptr)
    return DBEV_EINVAL;
  const long long M = static_cast<long long>(B) * HW;
  const long long total = M * (Ct / 4);
  const int blocks = static_cast<int>(total / 256 / 4 < DBEV_MAX_GRID * 2 ? (total + 1023) / 1024 : DBEV_MAX_GRID * 2);
  hipLaunchKernelGGL(adapt_mse_ds, dim3(blocks < 1 ? 1 : blocks), dim3(256), 0, dbev_stream(stream),
                     reinterpret_cast<const float4*>(diff_nhwc), grad_e, grad_efp, grad_pool, channel_weight,
                     static_cast<int>(M), Ct / 4, HW, reinterpret_cast<float4*>(ds_nhwc));
  DBEV_LAUNCH_CHECK();
  return 0;
}
