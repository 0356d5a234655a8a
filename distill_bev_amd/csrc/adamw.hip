// AdamW over all parameters of a group in ONE launch, with the gradient clipping factor applied on the fly.
//
// The reference trains with `optimizer = dict(type='AdamW', lr=2e-4, weight_decay=0.01)` and `grad_clip = dict(max_norm=35)`
// (the recipes under configs/: mmcv's OptimizerHook -> torch.nn.utils.clip_grad_norm_ -> torch.optim.AdamW.step).  torch's fused AdamW walks the ~320
// parameter tensors of the student in 14 multi_tensor_apply launches at 2.2 TB/s (0.69 ms per step for 54 M parameters), behind a
// separate pass that multiplies every gradient by the clipping factor (0.14 ms).  Here: one launch over a chunk map (chunk -> tensor,
// offset; built once) and a per-step pointer table (the gradient tensors are new every step), the factor read from device memory
// and applied to the gradient value as it is loaded -- the same rounded product `g * c` the in-place clip would have stored.
// Arithmetic: torch's fused kernel's (ATen fused_adam_utils: decoupled weight decay, the moment updates and the decay evaluated in
// double and rounded to fp32 once, step size and denominator in fp32); tests/test_gpu_adamw.py compares several steps with
// torch.optim.AdamW(fused=True).
#include "common.h"

namespace {

struct AdamTensor { float* p; const float* g; float* m; float* v; long long n; };   // 40 bytes (dbevAdamTensor of the header)
constexpr int AD_CHUNK = 4096;

struct AdamHyper { double lr, beta1, beta2, eps, wd; float bc1, bc2_sqrt; };

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, const AdamHyper& h, float step_size) {
  if (h.wd != 0.0) p = static_cast<float>(static_cast<double>(p) - h.lr * h.wd * static_cast<double>(p));
  m = static_cast<float>(h.beta1 * static_cast<double>(m) + (1.0 - h.beta1) * static_cast<double>(g));
  v = static_cast<float>(h.beta2 * static_cast<double>(v) + (1.0 - h.beta2) * static_cast<double>(g) * static_cast<double>(g));
  const float denom = static_cast<float>(static_cast<double>(sqrtf(v) / h.bc2_sqrt) + h.eps);
  p -= step_size * m / denom;
}

__global__ __launch_bounds__(256) void adamw_multi(const AdamTensor* __restrict__ tensors, const int2* __restrict__ chunks,
                                                   const float* __restrict__ gscale, AdamHyper h) {
  const int2 ck = chunks[blockIdx.x];
  const AdamTensor t = tensors[ck.x];
  const long long off = static_cast<long long>(ck.y) * AD_CHUNK;
  const long long left = t.n - off;
  const int cnt = left < AD_CHUNK ? static_cast<int>(left) : AD_CHUNK;
  const float c = gscale != nullptr ? *gscale : 1.f;
  const float step_size = static_cast<float>(h.lr / static_cast<double>(h.bc1));
  float* p = t.p + off;
  const float* g = t.g + off;
  float* m = t.m + off;
  float* v = t.v + off;
  const bool vec = ((reinterpret_cast<size_t>(p) | reinterpret_cast<size_t>(g) | reinterpret_cast<size_t>(m) | reinterpret_cast<size_t>(v)) & 15) == 0;
  if (vec) {
    const int n4 = cnt >> 2;
#pragma unroll
    for (int it = 0; it < AD_CHUNK / 1024; ++it) {
      const int i = it * 256 + threadIdx.x;
      if (i < n4) {
        float4 pp = reinterpret_cast<float4*>(p)[i], mm = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
        const float4 gg = reinterpret_cast<const float4*>(g)[i];
        adam_one(pp.x, gscale != nullptr ? gg.x * c : gg.x, mm.x, vv.x, h, step_size);
        adam_one(pp.y, gscale != nullptr ? gg.y * c : gg.y, mm.y, vv.y, h, step_size);
        adam_one(pp.z, gscale != nullptr ? gg.z * c : gg.z, mm.z, vv.z, h, step_size);
        adam_one(pp.w, gscale != nullptr ? gg.w * c : gg.w, mm.w, vv.w, h, step_size);
        reinterpret_cast<float4*>(p)[i] = pp;
        reinterpret_cast<float4*>(m)[i] = mm;
        reinterpret_cast<float4*>(v)[i] = vv;
      }
    }
    for (int i = (n4 << 2) + threadIdx.x; i < cnt; i += 256) {
      float pp = p[i], mm = m[i], vv = v[i];
      adam_one(pp, gscale != nullptr ? g[i] * c : g[i], mm, vv, h, step_size);
      p[i] = pp; m[i] = mm; v[i] = vv;
    }
  } else {
    for (int i = threadIdx.x; i < cnt; i += 256) {
      float pp = p[i], mm = m[i], vv = v[i];
      adam_one(pp, gscale != nullptr ? g[i] * c : g[i], mm, vv, h, step_size);
      p[i] = pp; m[i] = mm; v[i] = vv;
    }
  }
}

}  // namespace

extern "C" int dbev_adamw_chunk_elems(void) { return AD_CHUNK; }

extern "C" int dbev_adamw_multi(const void* tensors, const void* chunks, int n_chunks, const float* grad_scale, double lr, double beta1,
                                double beta2, double eps, double weight_decay, float bias_correction1, float bias_correction2_sqrt,
                                dbevStream_t stream) {
  if (n_chunks < 0 || (n_chunks > 0 && (tensors == nullptr || chunks == nullptr)) || !(bias_correction1 > 0.f) ||
      !(bias_correction2_sqrt > 0.f))
    return DBEV_EINVAL;
  if (n_chunks == 0) return 0;
  AdamHyper h{lr, beta1, beta2, eps, weight_decay, bias_correction1, bias_correction2_sqrt};
  hipLaunchKernelGGL(adamw_multi, dim3(n_chunks), dim3(256), 0, dbev_stream(stream), static_cast<const AdamTensor*>(tensors),
                     static_cast<const int2*>(chunks), grad_scale, h);
  DBEV_LAUNCH_CHECK();
  return 0;
}
