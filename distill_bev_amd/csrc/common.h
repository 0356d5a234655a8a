// Shared helpers for the gfx950 kernels of libdbev_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dbev_hip.h"

#define DBEV_WAVE 64

#define DBEV_LAUNCH_CHECK()                         \
  do {                                              \
    hipError_t _e = hipGetLastError();              \
    if (_e != hipSuccess) return static_cast<int>(_e); \
  } while (0)

#define DBEV_HIP_TRY(expr)                          \
  do {                                              \
    hipError_t _e = (expr);                         \
    if (_e != hipSuccess) return static_cast<int>(_e); \
  } while (0)

static inline int dbev_ceil_div(long long a, long long b) { return static_cast<int>((a + b - 1) / b); }

// MI355X: 256 CUs; memory-bound grid-stride kernels are capped at 8 blocks of 256 per CU.
static constexpr int DBEV_NUM_CU = 256;
static constexpr int DBEV_MAX_GRID = DBEV_NUM_CU * 8;

static inline hipStream_t dbev_stream(dbevStream_t s) { return reinterpret_cast<hipStream_t>(s); }

// XCD-aware block order.  The dispatcher deals workgroups round-robin over the 8 XCDs (block i runs
// on XCD i % 8) and every XCD has a private 4 MB L2.  Launch a grid that is a multiple of 8 and use
// logical block xcd_block(): XCD k then walks the k-th contiguous eighth of the index space, so data
// that is shared between neighbouring blocks (one sample-frame's feature map) lives in ONE L2 instead
// of being replicated in all eight.
static constexpr int DBEV_NUM_XCD = 8;
static inline int dbev_round_xcd(int blocks) { return (blocks + DBEV_NUM_XCD - 1) / DBEV_NUM_XCD * DBEV_NUM_XCD; }
__device__ __forceinline__ int xcd_block() {
  return (blockIdx.x % DBEV_NUM_XCD) * (gridDim.x / DBEV_NUM_XCD) + blockIdx.x / DBEV_NUM_XCD;
}

// streaming 16-byte store (global_store_dwordx4 ... nt): output that is written once and not re-read by this
// kernel should not displace the L2 lines the gathers need; measured +7 % on the canvas write.
__device__ __forceinline__ void st_nt(float4* p, const float4& v) {
  typedef float v4f __attribute__((ext_vector_type(4)));
  const v4f t = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(t, reinterpret_cast<v4f*>(p));
}

// per-kernel timing log (abi.hip): DbevKt brackets the launches of its scope with an event pair when enabled
extern unsigned g_dbev_kt_mask;
void dbev_kt_push(hipEvent_t a, hipEvent_t b, int kid, long long bytes);
struct DbevKt {
  hipEvent_t a = nullptr, b = nullptr;
  hipStream_t s;
  int kid;
  long long bytes;
  DbevKt(int kid_, long long bytes_, hipStream_t s_) : s(s_), kid(kid_), bytes(bytes_) {
    if (((g_dbev_kt_mask >> kid_) & 1u) && hipEventCreate(&a) == hipSuccess && hipEventCreate(&b) == hipSuccess) (void)hipEventRecord(a, s);
    else a = nullptr;
  }
  ~DbevKt() {
    if (a != nullptr) {
      (void)hipEventRecord(b, s);
      dbev_kt_push(a, b, kid, bytes);
    }
  }
};
