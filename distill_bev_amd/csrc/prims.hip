// exclusive scan + segment sort (see prims.h).
#include "prims.h"

namespace dbev {
namespace {

constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 16;                       // consecutive items per thread
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;  // 4096 items per block

__device__ __forceinline__ int wave_incl_scan(int v, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(v, o);
    if (lane >= o) v += t;
  }
  return v;
}

// exclusive scan of one int per thread across a 256-thread block; returns exclusive prefix,
// block total in *total (valid for all threads).
__device__ __forceinline__ int block_excl_scan(int v, int* total, int* lds /*>=5 ints*/) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int incl = wave_incl_scan(v, lane);
  if (lane == 63) lds[w] = incl;
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < SCAN_THREADS / 64; ++i) {
    const int x = lds[i];
    if (i < w) base += x;
    tot += x;
  }
  __syncthreads();
  *total = tot;
  return base + incl - v;
}

__device__ __forceinline__ int f_of(int x, bool as_flags) { return as_flags ? (x > 0 ? 1 : 0) : x; }

// a thread's SCAN_ITEMS consecutive items: four 16-byte loads when the run is whole and aligned (one int per load left 64 lanes
// striding 64 bytes: 16 instructions x 64 cache lines per wave)
__device__ __forceinline__ void load_items(const int* __restrict__ in, long long base, long long n, bool as_flags,
                                           int (&v)[SCAN_ITEMS]) {
  if (base + SCAN_ITEMS <= n && (reinterpret_cast<size_t>(in) & 15) == 0) {
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k += 4) {
      const int4 t = *reinterpret_cast<const int4*>(in + base + k);
      v[k] = f_of(t.x, as_flags); v[k + 1] = f_of(t.y, as_flags); v[k + 2] = f_of(t.z, as_flags); v[k + 3] = f_of(t.w, as_flags);
    }
  } else {
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
      const long long i = base + k;
      v[k] = i < n ? f_of(in[i], as_flags) : 0;
    }
  }
}

__global__ __launch_bounds__(SCAN_THREADS) void scan_tile_sums(const int* __restrict__ in,
                                                               int* __restrict__ tile_sums,
                                                               long long n, bool as_flags) {
  __shared__ int lds[8];
  const long long base = static_cast<long long>(blockIdx.x) * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
  int v[SCAN_ITEMS];
  load_items(in, base, n, as_flags, v);
  int s = 0;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k) s += v[k];
  int tot;
  block_excl_scan(s, &tot, lds);
  if (threadIdx.x == 0) tile_sums[blockIdx.x] = tot;
}

// single block: exclusive scan of the tile sums in place; writes grand total to tile_sums[nt]
__global__ __launch_bounds__(SCAN_THREADS) void scan_tile_offsets(int* __restrict__ tile_sums, int nt) {
  __shared__ int lds[8];
  int carry = 0;
  for (int b = 0; b < nt; b += SCAN_THREADS) {
    const int i = b + threadIdx.x;
    const int v = i < nt ? tile_sums[i] : 0;
    int tot;
    const int ex = block_excl_scan(v, &tot, lds);
    if (i < nt) tile_sums[i] = carry + ex;
    carry += tot;
  }
  if (threadIdx.x == 0) tile_sums[nt] = carry;
}

__global__ __launch_bounds__(SCAN_THREADS) void scan_downsweep(const int* __restrict__ in,
                                                               int* __restrict__ out,
                                                               const int* __restrict__ tile_offs,
                                                               long long n, int nt, bool as_flags,
                                                               int* __restrict__ total_out) {
  __shared__ int lds[8];
  const long long base = static_cast<long long>(blockIdx.x) * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
  int v[SCAN_ITEMS];
  load_items(in, base, n, as_flags, v);
  int s = 0;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k) s += v[k];
  int tot;
  int ex = block_excl_scan(s, &tot, lds) + tile_offs[blockIdx.x];
  if (base + SCAN_ITEMS <= n && (reinterpret_cast<size_t>(out) & 15) == 0) {
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k += 4) {
      int4 o4;
      o4.x = ex; ex += v[k];
      o4.y = ex; ex += v[k + 1];
      o4.z = ex; ex += v[k + 2];
      o4.w = ex; ex += v[k + 3];
      *reinterpret_cast<int4*>(out + base + k) = o4;
    }
  } else {
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
      const long long i = base + k;
      if (i < n) out[i] = ex;
      ex += v[k];
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    const int grand = tile_offs[nt];
    out[n] = grand;
    if (total_out) *total_out = grand;
  }
}

// Segment sort (values distinct).  Three size classes:
//   <= 16 values : 16-lane bitonic network, four consecutive segments per wave (78 % of the occupied BEV
//                  cells and virtually every LiDAR pillar);
//   17..64       : 64-lane bitonic network in registers, the wave visits those segments one by one;
//   > 64         : (dense BEV cells next to the cameras: 2 % of the cells, 12 % of the points of a
//                  6-camera frame, and adjacent to each other) are queued on a device work list and
//                  sorted by a second kernel, a whole workgroup per segment: LDS slab + counting rank against 4 broadcast LDS values per read.  Leaving
//                  them to the wave that owns them made 4 x 600-value segments in a row the critical path
//                  of the launch (105 of 130 us).
constexpr int SORT_LDS_VALUES = 2048;   // slab of the long-segment kernel (8 KiB)

__device__ __forceinline__ void sort_segment_wave64(const unsigned* __restrict__ src, unsigned* __restrict__ dst,
                                                    int st, int L, int lane) {
  unsigned v = lane < L ? src[st + lane] : 0xFFFFFFFFu;
#pragma unroll
  for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
    for (int j = k >> 1; j > 0; j >>= 1) {
      const unsigned o = __shfl_xor(v, j);
      const bool up = (lane & k) == 0;
      const bool lower = (lane & j) == 0;
      v = (lower == up) ? (v < o ? v : o) : (v > o ? v : o);
    }
  }
  if (lane < L) dst[st + lane] = v;
}

__global__ __launch_bounds__(256) void segment_sort_kernel(const int* __restrict__ starts,
                                                           const unsigned* __restrict__ src,
                                                           unsigned* __restrict__ dst, int n_seg,
                                                           int* __restrict__ long_list /* [0] = count */) {
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  const int grp = lane >> 4, lig = lane & 15;
  const int seg = wave * 4 + grp;
  int st = 0, L = 0;
  if (seg < n_seg) { st = starts[seg]; L = starts[seg + 1] - st; }
  const bool small = L <= 16;
  if (L > 64 && lig == 0) long_list[1 + atomicAdd(long_list, 1)] = seg;   // order irrelevant
  if (__any(small && L > 0)) {
    unsigned v = (small && lig < L) ? src[st + lig] : 0xFFFFFFFFu;
    if (__any(small && L > 1)) {
#pragma unroll
      for (int k = 2; k <= 16; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
          const unsigned o = __shfl_xor(v, j);
          const bool up = (lig & k) == 0;
          const bool lower = (lig & j) == 0;
          v = (lower == up) ? (v < o ? v : o) : (v > o ? v : o);
        }
      }
    }
    if (small && lig < L) dst[st + lig] = v;
  }
  if (__any(!small && L <= 64)) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int Lg = __shfl(L, g * 16);
      const int sg = __shfl(st, g * 16);
      if (Lg > 16 && Lg <= 64) sort_segment_wave64(src, dst, sg, Lg, lane);
    }
  }
}

__global__ __launch_bounds__(256) void segment_sort_long_kernel(const int* __restrict__ starts,
                                                                const unsigned* __restrict__ src,
                                                                unsigned* __restrict__ dst,
                                                                const int* __restrict__ long_list) {
  __shared__ __attribute__((aligned(16))) unsigned slab[SORT_LDS_VALUES];
  const int tid = threadIdx.x;
  const int nl = long_list[0];
  for (int k = blockIdx.x; k < nl; k += gridDim.x) {
    const int seg = long_list[1 + k];
    const int st = starts[seg];
    const int L = starts[seg + 1] - st;
    const unsigned* p = src + st;
    if (L <= SORT_LDS_VALUES) {
      const int Lp = (L + 3) & ~3;
      for (int i = tid; i < Lp; i += 256) slab[i] = i < L ? p[i] : 0xFFFFFFFFu;   // pad: never < v
      __syncthreads();
      const uint4* s4 = reinterpret_cast<const uint4*>(slab);
      for (int i = tid; i < L; i += 256) {
        const unsigned v = slab[i];
        int rank = 0;
#pragma unroll 4
        for (int j = 0; j < (Lp >> 2); ++j) {
          const uint4 q = s4[j];                                                   // same address in every lane
          rank += (q.x < v) + (q.y < v) + (q.z < v) + (q.w < v);
        }
        dst[st + rank] = v;
      }
      __syncthreads();
    } else {
      // degenerate geometry (thousands of points in one cell): same ranking straight from L2
      for (int i = tid; i < L; i += 256) {
        const unsigned v = p[i];
        int rank = 0;
        for (int j = 0; j < L; ++j) rank += p[j] < v ? 1 : 0;
        dst[st + rank] = v;
      }
    }
  }
}

}  // namespace

size_t scan_workspace_ints(long long n) { return static_cast<size_t>((n + SCAN_TILE - 1) / SCAN_TILE) + 2; }

int exclusive_scan_i32(const int* in, int* out, long long n, bool as_flags, int* total_out, int* ws,
                       hipStream_t s) {
  if (n < 0) return DBEV_EINVAL;
  const int nt = static_cast<int>((n + SCAN_TILE - 1) / SCAN_TILE);
  if (nt == 0) {
    DBEV_HIP_TRY(hipMemsetAsync(out, 0, sizeof(int), s));
    if (total_out) DBEV_HIP_TRY(hipMemsetAsync(total_out, 0, sizeof(int), s));
    return 0;
  }
  hipLaunchKernelGGL(scan_tile_sums, dim3(nt), dim3(SCAN_THREADS), 0, s, in, ws, n, as_flags);
  hipLaunchKernelGGL(scan_tile_offsets, dim3(1), dim3(SCAN_THREADS), 0, s, ws, nt);
  hipLaunchKernelGGL(scan_downsweep, dim3(nt), dim3(SCAN_THREADS), 0, s, in, out, ws, n, nt, as_flags,
                     total_out);
  DBEV_LAUNCH_CHECK();
  return 0;
}

size_t segment_sort_workspace_ints(long long n_values) { return static_cast<size_t>(n_values / 65) + 2; }

int segment_sort_u32(const int* starts, const unsigned* src, unsigned* dst, int n_seg, int* ws, hipStream_t s) {
  if (n_seg <= 0) return 0;
  if (ws == nullptr) return DBEV_EINVAL;
  DBEV_HIP_TRY(hipMemsetAsync(ws, 0, sizeof(int), s));
  hipLaunchKernelGGL(segment_sort_kernel, dim3(dbev_ceil_div(n_seg, 16)), dim3(256), 0, s, starts, src,
                     dst, n_seg, ws);
  hipLaunchKernelGGL(segment_sort_long_kernel, dim3(DBEV_NUM_CU * 4), dim3(256), 0, s, starts, src, dst, ws);
  DBEV_LAUNCH_CHECK();
  return 0;
}

}  // namespace dbev
