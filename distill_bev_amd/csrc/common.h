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
