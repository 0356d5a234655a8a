// Library identity entry points of libdbev_hip.so and the per-kernel timing log (include/dbev_hip.h).
#include "common.h"

#include <atomic>
#include <mutex>
#include <vector>

extern "C" int dbev_abi_version(void) { return DBEV_ABI_VERSION; }
extern "C" const char* dbev_target_arch(void) { return "gfx950"; }

namespace {
struct KtRec { hipEvent_t a, b; int kid; long long bytes; };
std::vector<KtRec> g_log;
std::mutex g_mu;
}  // namespace

unsigned g_dbev_kt_mask = 0;   // bit k set: kernel id k is logged

// ---- fallback ledger: the host-side mirrors (bn_act, SkinnyConv2d, fused adaptation, fused pillar path, batched head) note here whenever a DEVICE tensor they were built for takes the stock torch path instead of the kernels of this
// library (ineligible layout / channel count / mode), so that a layout regression shows up as a number, not as a slower step.
namespace {
std::atomic<long long> g_fallbacks[DBEV_FB_SITES];
}
extern "C" int dbev_fallback_note(int site) {
  if (site < 0 || site >= DBEV_FB_SITES) return DBEV_EINVAL;
  g_fallbacks[site].fetch_add(1, std::memory_order_relaxed);
  return 0;
}
extern "C" long long dbev_fallback_count(int site) {
  if (site >= DBEV_FB_SITES) return -1;
  if (site >= 0) return g_fallbacks[site].load(std::memory_order_relaxed);
  long long t = 0;
  for (int i = 0; i < DBEV_FB_SITES; ++i) t += g_fallbacks[i].load(std::memory_order_relaxed);
  return t;
}
extern "C" int dbev_fallback_reset(void) {
  for (int i = 0; i < DBEV_FB_SITES; ++i) g_fallbacks[i].store(0, std::memory_order_relaxed);
  return 0;
}

void dbev_kt_push(hipEvent_t a, hipEvent_t b, int kid, long long bytes) {
  std::lock_guard<std::mutex> lk(g_mu);
  g_log.push_back(KtRec{a, b, kid, bytes});
}

extern "C" int dbev_kernel_timing_enable(int mask) {
  g_dbev_kt_mask = mask < 0 ? 0xffffffffu : static_cast<unsigned>(mask);
  return 0;
}

extern "C" int dbev_kernel_timing_read(int* kernel_id, float* ms, long long* algorithmic_bytes, int cap) {
  std::lock_guard<std::mutex> lk(g_mu);
  const int n = static_cast<int>(g_log.size());
  for (int i = 0; i < n; ++i) {
    KtRec& r = g_log[i];
    float t = 0.f;
    if (hipEventSynchronize(r.b) == hipSuccess) (void)hipEventElapsedTime(&t, r.a, r.b);
    if (i < cap && kernel_id != nullptr && ms != nullptr && algorithmic_bytes != nullptr) {
      kernel_id[i] = r.kid;
      ms[i] = t;
      algorithmic_bytes[i] = r.bytes;
    }
    (void)hipEventDestroy(r.a);
    (void)hipEventDestroy(r.b);
  }
  g_log.clear();
  return n;
}

extern "C" const char* dbev_kernel_name(int kid) {
  switch (kid) {
    case DBEV_K_BN_STATS: return "bn_stats";
    case DBEV_K_BN_FINALIZE: return "bn_finalize";
    case DBEV_K_BN_APPLY: return "bn_apply<false,*>";
    case DBEV_K_BN_APPLY_RES: return "bn_apply<true,*>";
    case DBEV_K_BN_BWD_REDUCE: return "bn_bwd_reduce<0|1>";
    case DBEV_K_BN_BWD_REDUCE_Y: return "bn_bwd_reduce<2>";
    case DBEV_K_BN_BWD_FINALIZE: return "bn_bwd_finalize";
    case DBEV_K_BN_BWD_DX: return "bn_bwd_dx<*,false>";
    case DBEV_K_BN_BWD_DX_RES: return "bn_bwd_dx<*,true>";
    case DBEV_K_SPCONV_FWD: return "sp_conv_fwd";
    case DBEV_K_MSDA_FWD: return "msda_fwd";
    case DBEV_K_MSDA_BWD_SAMPLE: return "msda_bwd_sample";
    case DBEV_K_MSDA_GV_GATHER: return "msda_gv_gather";
    case DBEV_K_ADAPT_MSE_FWD: return "adapt_mse_fwd";
    case DBEV_K_CONV1X1_FWD: return "c1x1_fwd";
    case DBEV_K_WINO_FWD: return "wino_fwd";
    case DBEV_K_WINO_WGRAD: return "wino_wgrad";
    case DBEV_K_GEMM1X1_FWD: return "g1_fwd";
    case DBEV_K_GEMM1X1_WGRAD: return "g1_wgrad";
    case DBEV_K_B6_FWD: return "b6_fwd";
    case DBEV_K_B6_WGRAD: return "b6_wgrad";
    case DBEV_K_STEM_FWD: return "stem_fwd";
    case DBEV_K_STEM_WGRAD: return "stem_wgrad";
    default: return "?";
  }
}
