// fp32 products on the bf16 matrix cores?  Sustained rate of v_mfma_f32_32x32x16_bf16 against v_mfma_f32_32x32x2_f32 on this box (registers
// only, 1 workgroup of 4 waves per CU), alone and with the VALU work of an on-the-fly 3-way bf16 split of the operands between the MFMAs
// (a = a0 + a1 + a2, six products a_i b_j with i + j <= 2 reproduce an fp32 product to ~2^-23: tools/bf16x6_error.py).
//   hipcc -O3 -w --offload-arch=gfx950 tools/mfma_bf16_bench.hip -o /tmp/mfma_bf16_bench && /tmp/mfma_bf16_bench
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short shortx8 __attribute__((ext_vector_type(8)));

// MODE 0: fp32 32x32x2, 1: bf16 32x32x16, 2: bf16 32x32x16 with VALU split work (NV packed ops per 6 MFMAs)
template <int MODE, int NV>
__global__ __launch_bounds__(256, 1) void k(float* __restrict__ out, const float* __restrict__ in, int iters) {
  const int lane = threadIdx.x & 63;
  floatx16 acc[8];
  for (int t = 0; t < 8; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float fa = in[lane], fb = in[lane + 64];
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = in[lane + 8 * i];
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(fa + i); b[i] = (__bf16)(fb - i); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      if (MODE == 0) {
#pragma unroll
        for (int j = 0; j < 6; ++j) acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[g], 0, 0, 0);
      } else {
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          acc[(g + j) & 7] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[(g + j) & 7], 0, 0, 0);
        }
        if (MODE == 2) {
          // the split of fresh fp32 operands: hi = x & 0xffff0000 (truncation keeps the residual exact), r = x - hi, ...
#pragma unroll
          for (int q = 0; q < NV; ++q) {
            const int i = q & 7;
            const float hi = __uint_as_float(__float_as_uint(v[i]) & 0xffff0000u);
            v[i] = (v[i] - hi) * 1.0001f + hi * 0.5f;
          }
          a[0] = (__bf16)v[0]; b[0] = (__bf16)v[1];
        }
      }
    }
  }
  float s = 0.f;
  for (int t = 0; t < 8; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
  for (int i = 0; i < 8; ++i) s += v[i];
  if (s == 12345.678f) out[0] = s;
}

template <int MODE, int NV>
void run(const char* name, int iters, int reps) {
  float *out, *in;
  hipMalloc(&out, 4); hipMalloc(&in, 4096); hipMemset(in, 0, 4096);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE, NV><<<256, 256>>>(out, in, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < reps; ++r) k<MODE, NV><<<256, 256>>>(out, in, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double per = MODE == 0 ? 2.0 * 32 * 32 * 2 : 2.0 * 32 * 32 * 16;
  const double flop = per * 48.0 * iters * 4.0 * 256 * reps;
  printf("%-58s iters %5d x %2d: %8.3f ms  %7.1f TFLOP/s  (%.1f cycles per MFMA at 2.4 GHz)\n", name, iters, reps, ms, flop / ms / 1e9,
         ms * 1e-3 * 2.4e9 / (48.0 * iters * reps));
  hipFree(out); hipFree(in);
}

int main() {
  for (int pass = 0; pass < 2; ++pass) {
    const int it = pass == 0 ? 200 : 4000, reps = pass == 0 ? 1 : 10;          // a ~100 us launch / ~2 s of sustained load
    run<0, 0>("fp32 32x32x2", it, reps);
    run<1, 0>("bf16 32x32x16", it * 2, reps);
    run<2, 4>("bf16 32x32x16 + 4 split groups (~5 VALU each) per 6 MFMAs", it * 2, reps);
    run<2, 8>("bf16 32x32x16 + 8 split groups per 6 MFMAs", it * 2, reps);
    run<2, 16>("bf16 32x32x16 + 16 split groups per 6 MFMAs", it * 2, reps);
  }
  return 0;
}
