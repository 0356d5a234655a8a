// Depth head of the BEVDepth view transformer, forward: normalise -> 1x1 convolution -> softmax in one pass.
//
// Reference sequence (mmdet3d/models/necks/view_transformer_mine.py:300-309, 325-328; bevdet_distill_more.py:398-416):
//   depth_feat  = self.dcn(depth_feat)              nn.Sequential(DCNv2, nn.BatchNorm2d(c))          -> [BN, c, H, W]
//   depth_digit = self.depthnet(depth_feat)         nn.Conv2d(c, D, kernel_size=1)                   -> [BN, D, H, W]
//   depth_prob  = self.get_depth_dist(depth_digit)  softmax(dim=1)                                   -> lift
// = three launches (norm apply, MIOpen 1x1, softmax) and the normalised c-channel map written and read back.  Here, after the
// statistics of the deformable convolution's output are known (dbev_bn_act_train_forward_pre with y = NULL, or the running
// statistics in eval mode), one kernel reads a 64-pixel tile of that output, applies scale / shift on the way into LDS, multiplies
// it with the D x c weight on the fp32 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 accumulation), adds the bias,
// takes the softmax over the D depth bins of each pixel and writes depth_digit (the depth loss reads it) and depth_prob (the lift
// reads it).  The normalised map is only written when the caller asks for it (the weight gradient of the 1x1 needs it).
//
// Mapping (wave64): a workgroup of 4 waves owns 64 pixels x 64 output columns (D <= 64, missing columns are zero weights); wave
// (wr, wc) multiplies the 32-pixel x 32-column block; A operand = pixel rows from LDS, B operand = weight columns from LDS (both
// tiles padded to odd-ish strides: conflict-free reads); the 64 x 64 logits go back through LDS so that 4 lanes own one pixel's D
// values for the softmax (two shuffles per reduction).  Workgroups are persistent and keep the transposed weight in LDS.
#include "common.h"

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int DH_PX = 64;     // pixels per tile
constexpr int DH_N = 64;      // padded output columns
constexpr int DH_WSTR = DH_N + 1;

__global__ __launch_bounds__(256) void depth_head_fwd(const float* __restrict__ X, const float* __restrict__ scale_shift,
                                                      const float* __restrict__ Wt, const float* __restrict__ bias,
                                                      float* __restrict__ digit, float* __restrict__ prob,
                                                      float* __restrict__ xn, long long M, int C, int N, int tiles) {
  extern __shared__ float smem[];
  const int XSTR = C + 4;                                      // 16-byte aligned rows; the A-operand read is 2-way conflicted at worst
  float* sW = smem;                                            // [C][DH_WSTR]   transposed weight, columns >= N zero
  float* sX = smem + static_cast<size_t>(C) * DH_WSTR;         // [DH_PX][XSTR]  normalised pixels, later the logits [DH_PX][DH_WSTR]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5, wr = w >> 1, wc = w & 1;
  for (int i = tid; i < C * DH_N; i += 256) {                  // weight [N][C] -> sW[c][n]
    const int c = i / DH_N, n = i - c * DH_N;
    sW[c * DH_WSTR + n] = n < N ? Wt[static_cast<size_t>(n) * C + c] : 0.f;
  }
  const int C4 = C >> 2;
  for (int t = blockIdx.x; t < tiles; t += gridDim.x) {
    const long long m0 = static_cast<long long>(t) * DH_PX;
    __syncthreads();                                           // the previous tile's logits are consumed (first tile: sW is complete)
    for (int i = tid; i < DH_PX * C4; i += 256) {
      const int px = i / C4, c4 = i - px * C4;
      const long long m = m0 + px;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (m < M) {
        v = *reinterpret_cast<const float4*>(X + m * C + 4 * c4);
        const float4 sc = *reinterpret_cast<const float4*>(scale_shift + 4 * c4);
        const float4 sh = *reinterpret_cast<const float4*>(scale_shift + C + 4 * c4);
        v.x = fmaf(v.x, sc.x, sh.x); v.y = fmaf(v.y, sc.y, sh.y); v.z = fmaf(v.z, sc.z, sh.z); v.w = fmaf(v.w, sc.w, sh.w);
        if (xn != nullptr) *reinterpret_cast<float4*>(xn + m * C + 4 * c4) = v;
      }
      *reinterpret_cast<float4*>(&sX[px * XSTR + 4 * c4]) = v;
    }
    __syncthreads();
    floatx16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const float* ap = sX + (32 * wr + l31) * XSTR + half;
    const float* bp = sW + half * DH_WSTR + 32 * wc + l31;
    for (int k = 0; k < C; k += 4) {                           // C % 4 == 0
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[k], bp[k * DH_WSTR], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[k + 2], bp[(k + 2) * DH_WSTR], acc, 0, 0, 0);
    }
    __syncthreads();                                           // every wave is done with the pixel tile: reuse it for the logits
    const int col = 32 * wc + l31;
    const float bv = col < N ? bias[col] : 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int j = 0; j < 4; ++j)                              // accumulator 4q + j = pixel row 8q + 4 half + j of the wave's block
        sX[(32 * wr + 8 * q + 4 * half + j) * DH_WSTR + col] = acc[4 * q + j] + bv;
    __syncthreads();
    // softmax over the N columns of a pixel: 4 lanes per pixel, 16 columns each
    const int px = tid >> 2, part = tid & 3;
    const float* row = sX + px * DH_WSTR + 16 * part;
    float v[16];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      v[i] = row[i];
      if (16 * part + i < N) mx = fmaxf(mx, v[i]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 1));
    mx = fmaxf(mx, __shfl_xor(mx, 2));
    float e[16], sum = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      e[i] = 16 * part + i < N ? expf(v[i] - mx) : 0.f;
      sum += e[i];
    }
    sum += __shfl_xor(sum, 1);
    sum += __shfl_xor(sum, 2);
    const float inv = 1.f / sum;
    const long long m = m0 + px;
    if (m < M) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int n = 16 * part + i;
        if (n < N) {
          digit[m * N + n] = v[i];
          prob[m * N + n] = e[i] * inv;
        }
      }
    }
  }
}

}  // namespace

extern "C" int dbev_depth_head_forward(const float* x_nhwc, const float* scale_shift, const float* weight, const float* bias,
                                       long long M, int C, int N, float* depth_digit_nhwc, float* depth_prob_nhwc,
                                       float* normalised_nhwc, dbevStream_t stream) {
  if (M <= 0 || C <= 0 || (C & 3) || C > 256 || N <= 0 || N > DH_N || x_nhwc == nullptr || scale_shift == nullptr ||
      weight == nullptr || bias == nullptr || depth_digit_nhwc == nullptr || depth_prob_nhwc == nullptr)
    return DBEV_EINVAL;
  const long long tiles = (M + DH_PX - 1) / DH_PX;
  if (tiles > 0x7fffffffLL) return DBEV_EINVAL;
  hipStream_t s = dbev_stream(stream);
  const size_t tile = static_cast<size_t>(DH_PX) * (C + 4 > DH_WSTR ? C + 4 : DH_WSTR);      // pixel tile, then the logits
  const size_t lds = sizeof(float) * (static_cast<size_t>(C) * DH_WSTR + tile);
  static bool attr_set = false;
  if (!attr_set) {
    DBEV_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(depth_head_fwd), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     160 * 1024));
    attr_set = true;
  }
  const int grid = static_cast<int>(tiles < DBEV_NUM_CU ? tiles : DBEV_NUM_CU);
  hipLaunchKernelGGL(depth_head_fwd, dim3(grid), dim3(256), lds, s, x_nhwc, scale_shift, weight, bias, depth_digit_nhwc,
                     depth_prob_nhwc, normalised_nhwc, M, C, N, static_cast<int>(tiles));
  DBEV_LAUNCH_CHECK();
  return 0;
}
