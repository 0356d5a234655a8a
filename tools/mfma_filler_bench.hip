// How many side instructions does ONE wave per SIMD hide in the shadow of a v_mfma_f32_32x32x2_f32 (64 cycles)?  The Winograd kernels
// (csrc/wino.hip) keep 256 accumulators per lane, i.e. one wave per SIMD, and interleave the input transform's VALU work and LDS reads
// with the MFMAs.  Loop body = 64 MFMAs over 16 accumulators; after each MFMA F fillers of a given kind.
//   hipcc -O3 -w --offload-arch=gfx950 tools/mfma_filler_bench.hip -o /tmp/mfb && /tmp/mfb
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx2 __attribute__((ext_vector_type(2)));

// KIND 0: v_add_f32, 1: v_pk_add_f32, 2: ds_read_b128 every 4th MFMA + F v_add, 3: operand produced by a VALU right before its MFMA
// ORDER 0: 4 dependent MFMAs per accumulator in a row, 1: consecutive MFMAs on different accumulators
template <int F, int KIND, int ORDER>
__global__ __launch_bounds__(256, 1) void k(float* __restrict__ out, int iters, const float* __restrict__ gbuf) {
  __shared__ __attribute__((aligned(16))) float sm[4096];
  const int tid = threadIdx.x;
  for (int i = tid; i < 4096; i += 256) sm[i] = 1e-3f * (i & 63);
  __syncthreads();
  floatx16 acc[16];
  for (int p = 0; p < 16; ++p) for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;
  float a = 1.f + tid * 1e-6f, b = 2.f;
  float f0 = tid, f1 = 1.f, f2 = 2.f, f3 = 3.f;
  floatx2 g0 = {1.f, 2.f}, g1 = {3.f, 4.f};
  float4 lv = make_float4(0.f, 0.f, 0.f, 0.f);
  unsigned sc = iters;
  const unsigned voff = (tid & 63) * 16, ldsb = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)sm + (tid >> 6) * 1024);
  const unsigned long long gsrc = (unsigned long long)(gbuf + __builtin_amdgcn_readfirstlane(tid >> 6) * 256);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 64; ++s) {
      const int p = ORDER == 0 ? (s >> 2) : (s & 15);
      float aa = a;
      if (KIND == 3) { asm volatile("v_add_f32 %0, %1, %2" : "=v"(aa) : "v"(f0), "v"(f1)); }
      acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(aa, b, acc[p], 0, 0, 0);
      if (KIND == 2 && (s & 3) == 0) { lv = *reinterpret_cast<const float4*>(&sm[((tid * 4 + s * 16) & 4092)]); asm volatile("" :: "v"(lv.x), "v"(lv.y), "v"(lv.z), "v"(lv.w)); }
#pragma unroll
      for (int i = 0; i < (KIND >= 4 ? 0 : F); ++i) {
        if (KIND == 1) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(g0) : "v"(g1));
        else if (i & 1) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f2) : "v"(f3));
        else asm volatile("v_add_f32 %0, %0, %1" : "+v"(f0) : "v"(f1));
      }
      if (KIND == 4) {
#pragma unroll
        for (int i = 0; i < F; ++i) { *reinterpret_cast<float2*>(&sm[(tid * 2 + 512 * i) & 4094]) = make_float2(f1, f3); }
      }
      if (KIND == 5) {                                   // 8-way conflicted ds_read_b64 (64-byte lane stride)
#pragma unroll
        for (int i = 0; i < F; ++i) { float2 t = *reinterpret_cast<const float2*>(&sm[((tid & 63) * 16 + 2 * i) & 4094]); asm volatile("" :: "v"(t.x), "v"(t.y)); }
      }
      if (KIND == 6) {
#pragma unroll
        for (int i = 0; i < F; ++i) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sc));
      }
      if (KIND == 7 && (s & 7) == 0) {
        unsigned keep_;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep_) : "v"(voff), "s"(ldsb), "s"(gsrc) : "memory");
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (KIND == 7) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  if (KIND == 6 && sc == 12345) out[1] = 1.f;
  float sacc = f0 + f2 + g0.x + lv.x;
  for (int p = 0; p < 16; ++p) for (int r = 0; r < 16; ++r) sacc += acc[p][r];
  if (sacc == 12345.678f) out[0] = sacc;
}

template <int F, int KIND, int ORDER>
void run(const char* name) {
  float* out;
  hipMalloc(&out, 8);
  float* gbuf; hipMalloc(&gbuf, 1 << 20); hipMemset(gbuf, 0, 1 << 20);
  const int iters = 300, grid = 256;
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  k<F, KIND, ORDER><<<grid, 256>>>(out, 10, gbuf);
  hipDeviceSynchronize();
  hipEventRecord(a);
  k<F, KIND, ORDER><<<grid, 256>>>(out, iters, gbuf);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  const double flop = 2.0 * 32 * 32 * 2 * 64.0 * iters * 4.0 * grid;
  printf("%-44s F=%2d order=%d: %7.3f ms  %6.1f TFLOP/s\n", name, F, ORDER, ms, flop / ms / 1e9);
  hipFree(out);
}


// Two waves per SIMD: does the VALU work of one wave hide under the MFMAs of the OTHER wave of the same SIMD?  NW waves per workgroup
// (4: one per SIMD, 8: two per SIMD), 8 accumulators (128 registers) per lane either way, F v_pk_add_f32 after each MFMA.
template <int F, int NW>
__global__ __launch_bounds__(64 * NW, 1) void k2(float* __restrict__ out, int iters) {
  const int tid = threadIdx.x;
  floatx16 acc[8];
  for (int p = 0; p < 8; ++p) for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;
  float a = 1.f + tid * 1e-6f, b = 2.f;
  floatx2 g0 = {1.f, 2.f}, g1 = {3.f, 4.f}, g2 = {1.f, 2.f}, g3 = {3.f, 4.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 32; ++s) {
      acc[s >> 2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[s >> 2], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < F; ++i) {
        if (i & 1) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(g0) : "v"(g1));
        else asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(g2) : "v"(g3));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float sacc = g0.x + g2.x;
  for (int p = 0; p < 8; ++p) for (int r = 0; r < 16; ++r) sacc += acc[p][r];
  if (sacc == 12345.678f) out[0] = sacc;
}

template <int F, int NW>
void run2(const char* name) {
  float* out;
  hipMalloc(&out, 8);
  const int iters = 600, grid = 256;
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  k2<F, NW><<<grid, 64 * NW>>>(out, 10);
  hipDeviceSynchronize();
  hipEventRecord(a);
  k2<F, NW><<<grid, 64 * NW>>>(out, iters);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  const double flop = 2.0 * 32 * 32 * 2 * 32.0 * iters * NW * grid;
  printf("%-44s F=%2d waves/SIMD=%d: %7.3f ms  %6.1f TFLOP/s\n", name, F, NW / 4, ms, flop / ms / 1e9);
  hipFree(out);
}

// Clustering: the same number of v_pk_add_f32 per MFMA on average (F), issued as one group of F * EVERY after every EVERY-th MFMA
// (8 independent register chains).
template <int F, int EVERY>
__global__ __launch_bounds__(256, 1) void k3(float* __restrict__ out, int iters) {
  const int tid = threadIdx.x;
  floatx16 acc[16];
  for (int p = 0; p < 16; ++p) for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;
  float a = 1.f + tid * 1e-6f, b = 2.f;
  floatx2 g[8], h = {3.f, 4.f};
  for (int i = 0; i < 8; ++i) g[i] = floatx2{1.f * i, 2.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 64; ++s) {
      acc[s >> 2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[s >> 2], 0, 0, 0);
      if (s % EVERY == EVERY - 1) {
#pragma unroll
        for (int i = 0; i < F * EVERY; ++i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(g[i & 7]) : "v"(h));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float sacc = 0.f;
  for (int i = 0; i < 8; ++i) sacc += g[i].x;
  for (int p = 0; p < 16; ++p) for (int r = 0; r < 16; ++r) sacc += acc[p][r];
  if (sacc == 12345.678f) out[0] = sacc;
}

template <int F, int EVERY>
void run3() {
  float* out;
  hipMalloc(&out, 8);
  const int iters = 300, grid = 256;
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  k3<F, EVERY><<<grid, 256>>>(out, 10);
  hipDeviceSynchronize();
  hipEventRecord(a);
  k3<F, EVERY><<<grid, 256>>>(out, iters);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  const double flop = 2.0 * 32 * 32 * 2 * 64.0 * iters * 4.0 * grid;
  printf("k3: %d v_pk_add per MFMA on average, in groups of %3d every %2d MFMAs: %7.3f ms  %6.1f TFLOP/s\n", F, F * EVERY, EVERY, ms, flop / ms / 1e9);
  hipFree(out);
}

int main() {
  run3<1, 1>(); run3<1, 2>(); run3<1, 4>(); run3<1, 8>(); run3<1, 16>(); run3<1, 64>();
  run3<2, 1>(); run3<2, 2>(); run3<2, 4>(); run3<2, 8>(); run3<2, 16>(); run3<2, 64>();
  run2<0, 4>("k2 bare"); run2<0, 8>("k2 bare");
  run2<1, 4>("k2 v_pk_add fillers"); run2<1, 8>("k2 v_pk_add fillers");
  run2<2, 4>("k2 v_pk_add fillers"); run2<2, 8>("k2 v_pk_add fillers");
  run2<4, 4>("k2 v_pk_add fillers"); run2<4, 8>("k2 v_pk_add fillers");
  run2<8, 4>("k2 v_pk_add fillers"); run2<8, 8>("k2 v_pk_add fillers");
  run<0, 0, 0>("bare MFMAs");
  run<0, 0, 1>("bare MFMAs");
  run<2, 0, 0>("v_add_f32 fillers"); run<4, 0, 0>("v_add_f32 fillers"); run<6, 0, 0>("v_add_f32 fillers");
  run<8, 0, 0>("v_add_f32 fillers"); run<12, 0, 0>("v_add_f32 fillers"); run<16, 0, 0>("v_add_f32 fillers");
  run<4, 0, 1>("v_add_f32 fillers"); run<8, 0, 1>("v_add_f32 fillers");
  run<1, 1, 0>("v_pk_add_f32 fillers"); run<2, 1, 0>("v_pk_add_f32 fillers"); run<4, 1, 0>("v_pk_add_f32 fillers"); run<8, 1, 0>("v_pk_add_f32 fillers");
  run<0, 2, 0>("ds_read_b128 per 4 MFMAs"); run<2, 2, 0>("ds_read_b128 per 4 MFMAs + v_add"); run<4, 2, 0>("ds_read_b128 per 4 MFMAs + v_add");
  run<0, 3, 0>("A operand from a VALU right before"); run<2, 3, 0>("A operand from a VALU right before + v_add");
  run<1, 4, 0>("ds_write_b64 fillers"); run<2, 4, 0>("ds_write_b64 fillers"); run<4, 4, 0>("ds_write_b64 fillers");
  run<1, 5, 0>("conflicted ds_read_b64 fillers"); run<2, 5, 0>("conflicted ds_read_b64 fillers"); run<4, 5, 0>("conflicted ds_read_b64 fillers");
  run<4, 6, 0>("s_add_u32 fillers"); run<8, 6, 0>("s_add_u32 fillers");
  run<0, 7, 0>("LDS-DMA 1 KB per 8 MFMAs (saddr form)");
  return 0;
}
