// What bounds an fp32-MFMA GEMM inner loop on this box?  v_mfma_f32_32x32x2_f32 fed (a) from registers only, (b) from LDS with the
// ds_read_b128 operand pattern of csrc/conv1x1.hip (one A + NT B reads per 4 * NT MFMAs), (c) as (b) plus one workgroup barrier per
// 16 * NT MFMAs, at 1 / 2 workgroups (of 4 waves) per CU.   hipcc -O3 -w --offload-arch=gfx950 tools/mfma_bench.hip -o /tmp/mfma_bench && /tmp/mfma_bench
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float floatx16 __attribute__((ext_vector_type(16)));
constexpr int STR = 36;

template <int NT, int MODE>
__global__ __launch_bounds__(256, 2) void k(float* __restrict__ out, int iters) {
  __shared__ __attribute__((aligned(16))) float sX[2][128][STR];
  __shared__ __attribute__((aligned(16))) float sW[2][NT * 32][STR];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, half = lane >> 5, l31 = lane & 31;
  for (int i = tid; i < 2 * 128 * STR; i += 256) (&sX[0][0][0])[i] = 1e-3f * (i & 63);
  for (int i = tid; i < 2 * NT * 32 * STR; i += 256) (&sW[0][0][0])[i] = 1e-3f * (i & 31);
  __syncthreads();
  floatx16 acc[NT];
  for (int t = 0; t < NT; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float4 av = make_float4(1.f, 2.f, 3.f, 4.f), bv[NT];
  for (int t = 0; t < NT; ++t) bv[t] = make_float4(1.f + t, 2.f, 3.f, 4.f);
  int buf = 0;
  for (int it = 0; it < iters; ++it) {                 // one "chunk": 4 groups of 4 * NT MFMAs
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (MODE >= 1) {
        av = *reinterpret_cast<const float4*>(&sX[buf][32 * w + l31][8 * j + 4 * half]);
#pragma unroll
        for (int t = 0; t < NT; ++t) bv[t] = *reinterpret_cast<const float4*>(&sW[buf][32 * t + l31][8 * j + 4 * half]);
      }
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv[t].x, acc[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv[t].y, acc[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv[t].z, acc[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv[t].w, acc[t], 0, 0, 0);
    }
    if (MODE >= 2) { __syncthreads(); buf ^= 1; }
  }
  float s = 0.f;
  for (int t = 0; t < NT; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
  if (s == 12345.678f) out[0] = s;
}

template <int NT, int MODE>
void run(int wg_per_cu, const char* name) {
  float* out;
  hipMalloc(&out, 4);
  const int iters = 2000, grid = 256 * wg_per_cu;
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  k<NT, MODE><<<grid, 256>>>(out, 10);
  hipDeviceSynchronize();
  hipEventRecord(a);
  k<NT, MODE><<<grid, 256>>>(out, iters);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  const double flop = 2.0 * 32 * 32 * 2 * (16.0 * NT) * iters * 4.0 * grid;
  printf("NT=%d %-28s %d WG/CU: %7.3f ms  %6.1f TFLOP/s\n", NT, name, wg_per_cu, ms, flop / ms / 1e9);
  hipFree(out);
}

template <int NT, int MODE>
void run_short(int iters, const char* name) {      // many short launches, like a 160 us convolution kernel
  float* out;
  hipMalloc(&out, 4);
  const int grid = 512, reps = 20;
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) k<NT, MODE><<<grid, 256>>>(out, iters);
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int i = 0; i < reps; ++i) k<NT, MODE><<<grid, 256>>>(out, iters);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  const double flop = 2.0 * 32 * 32 * 2 * (16.0 * NT) * iters * 4.0 * grid * reps;
  printf("NT=%d %-28s short x%d iters=%d: %7.1f us/launch  %6.1f TFLOP/s\n", NT, name, reps, iters, ms * 1e3 / reps, flop / ms / 1e9);
  hipFree(out);
}

int main() {
  run_short<4, 2>(33, "LDS operands + barrier");
  run_short<4, 2>(66, "LDS operands + barrier");
  run_short<4, 2>(330, "LDS operands + barrier");
  run_short<2, 2>(66, "LDS operands + barrier");
  for (int wg = 1; wg <= 2; ++wg) {
    run<4, 0>(wg, "registers only");
    run<4, 1>(wg, "LDS b128 operands");
    run<4, 2>(wg, "LDS operands + barrier");
    run<2, 0>(wg, "registers only");
    run<2, 1>(wg, "LDS b128 operands");
    run<2, 2>(wg, "LDS operands + barrier");
  }
  return 0;
}
