// Voxelization for gfx950: dynamic (per-point cell coordinate) and hard (first-come,
// max_points / max_voxels) -- replaces mmdet3d/ops/voxel/src/voxelization_cuda.cu
// (dynamic_voxelize_kernel :24-61, hard_voxelize_gpu :231-373) with the results of the
// reference's *CPU* implementation (voxelization_cpu.cpp:7-101), bit for bit.
//
// The reference's deterministic GPU path ranks duplicates with an O(N^2) scan
// (point_to_voxelidx_kernel :105-147) and a <<<1,1>>> serial kernel (determin_voxel_num
// :149-180) fenced by three device syncs.  Here the first-come order is recovered in
// O(N) parallel work with no sync:
//   cell id per point + int histogram -> exclusive scan -> fill -> per-cell sort by point id
//   => point lists per cell in input order; the head of each list is the cell's first point;
//   a scan over "is head" flags in POINT order numbers the voxels exactly as the serial
//   CPU loop does (new id at first sight of a cell), and the first max_points list entries
//   are the points the CPU loop keeps.
#include "prims.h"

namespace {

struct VoxParams {
  float vs[3];
  float rmin[3];
  int grid[3];  // x, y, z cells
};

// voxelization_cpu.cpp:22-32: c = floor((p - range_min) / voxel_size); reject c<0 || c>=grid.
// fp32 subtract + IEEE divide (no reciprocal, no contraction) -> same bits as the CPU code.
// NaN / +-inf coordinates are rejected (x86 int conversion yields INT_MIN there).
__device__ __forceinline__ bool cell_of_point(const float* __restrict__ p, const VoxParams& P,
                                              int& cx, int& cy, int& cz) {
#pragma clang fp contract(off)
  int c[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const float q = (p[j] - P.rmin[j]) / P.vs[j];
    const float fl = floorf(q);
    if (!(fl >= 0.f && fl < static_cast<float>(P.grid[j]))) return false;
    c[j] = static_cast<int>(fl);
  }
  cx = c[0]; cy = c[1]; cz = c[2];
  return true;
}

__global__ __launch_bounds__(256) void dynamic_voxelize_kernel(const float* __restrict__ points,
                                                               int* __restrict__ coors, int n, int nf,
                                                               VoxParams P) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int cx, cy, cz;
  const bool ok = cell_of_point(points + static_cast<size_t>(i) * nf, P, cx, cy, cz);
  int* o = coors + static_cast<size_t>(i) * 3;
  o[0] = ok ? cz : -1;
  o[1] = ok ? cy : -1;
  o[2] = ok ? cx : -1;
}

// ---- hard voxelization -------------------------------------------------------------------
__global__ __launch_bounds__(256) void hv_cell_count(const float* __restrict__ points, int n, int nf,
                                                     VoxParams P, int* __restrict__ cell,
                                                     int* __restrict__ count) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int cx, cy, cz;
  const bool ok = cell_of_point(points + static_cast<size_t>(i) * nf, P, cx, cy, cz);
  const int lin = ok ? (cz * P.grid[1] + cy) * P.grid[0] + cx : -1;
  cell[i] = lin;
  if (ok) atomicAdd(&count[lin], 1);
}

// consumes `count` as a countdown cursor: slot order inside a cell is arbitrary (sorted next)
__global__ __launch_bounds__(256) void hv_fill(const int* __restrict__ cell, int n,
                                               const int* __restrict__ start, int* __restrict__ count,
                                               unsigned* __restrict__ list) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int c = cell[i];
  if (c < 0) return;
  const int pos = atomicSub(&count[c], 1) - 1;
  list[start[c] + pos] = static_cast<unsigned>(i);
}

__global__ __launch_bounds__(256) void hv_head_flags(const int* __restrict__ cell, int n,
                                                     const int* __restrict__ start,
                                                     const unsigned* __restrict__ sorted,
                                                     int* __restrict__ flag) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int c = cell[i];
  flag[i] = (c >= 0 && sorted[start[c]] == static_cast<unsigned>(i)) ? 1 : 0;
}

// one thread per sorted list slot j (slots are grouped by cell, ascending point id inside)
__global__ __launch_bounds__(256) void hv_write(const float* __restrict__ points, int nf,
                                                const int* __restrict__ cell,
                                                const int* __restrict__ start,
                                                const unsigned* __restrict__ sorted, int n_valid_ptr_off,
                                                const int* __restrict__ vid_of_point,
                                                float* __restrict__ voxels, int* __restrict__ coors,
                                                int* __restrict__ num_points, int max_points,
                                                int max_voxels, VoxParams P, const int* __restrict__ n_valid) {
  (void)n_valid_ptr_off;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= *n_valid) return;
  const unsigned pid = sorted[j];
  const int c = cell[pid];
  const int s = start[c];
  const int r = j - s;
  if (r >= max_points) return;
  const int vid = vid_of_point[sorted[s]];
  if (vid >= max_voxels) return;  // cell first seen after the voxel budget was spent (:83)
  const float* src = points + static_cast<size_t>(pid) * nf;
  float* dst = voxels + (static_cast<size_t>(vid) * max_points + r) * nf;
  for (int k = 0; k < nf; ++k) dst[k] = src[k];
  if (r == 0) {
    const int L = start[c + 1] - s;
    num_points[vid] = L < max_points ? L : max_points;
    const int x = c % P.grid[0];
    const int y = (c / P.grid[0]) % P.grid[1];
    const int z = c / (P.grid[0] * P.grid[1]);
    coors[vid * 3 + 0] = z;
    coors[vid * 3 + 1] = y;
    coors[vid * 3 + 2] = x;
  }
}

__global__ void hv_voxel_num(const int* __restrict__ n_heads, int max_voxels, int* __restrict__ out) {
  const int h = *n_heads;
  *out = h < max_voxels ? h : max_voxels;
}

bool make_params(const float* voxel_size, const float* coors_range, VoxParams* P) {
  for (int i = 0; i < 3; ++i) {
    if (!(voxel_size[i] > 0.f)) return false;
    P->vs[i] = voxel_size[i];
    P->rmin[i] = coors_range[i];
    // voxelization_cpu.cpp:157-160
    P->grid[i] = static_cast<int>(round((coors_range[3 + i] - coors_range[i]) / voxel_size[i]));
    if (P->grid[i] <= 0) return false;
  }
  return true;
}

size_t align_up(size_t x) { return (x + 255) & ~static_cast<size_t>(255); }

struct HvLayout {
  size_t cell, count, start, list, sorted, flag, vidscan, scanws, sortws, nheads, total;
};

HvLayout hv_layout(long long n, long long ncell) {
  HvLayout L;
  size_t o = 0;
  L.cell = o;    o += align_up(sizeof(int) * n);
  L.count = o;   o += align_up(sizeof(int) * ncell);
  L.start = o;   o += align_up(sizeof(int) * (ncell + 1));
  L.list = o;    o += align_up(sizeof(int) * n);
  L.sorted = o;  o += align_up(sizeof(int) * n);
  L.flag = o;    o += align_up(sizeof(int) * n);
  L.vidscan = o; o += align_up(sizeof(int) * (n + 1));
  const size_t sw = dbev::scan_workspace_ints(ncell > n ? ncell : n);
  L.scanws = o;  o += align_up(sizeof(int) * sw);
  L.sortws = o;  o += align_up(sizeof(int) * dbev::segment_sort_workspace_ints(n));
  L.nheads = o;  o += align_up(sizeof(int) * 4);
  L.total = o;
  return L;
}

}  // namespace

extern "C" int dbev_dynamic_voxelize(const float* points, int32_t* coors, int num_points,
                                     int num_features, const float* voxel_size_host,
                                     const float* coors_range_host, int ndim, dbevStream_t stream) {
  if (ndim != 3 || num_features < 3 || num_points < 0) return DBEV_EINVAL;
  VoxParams P;
  if (!make_params(voxel_size_host, coors_range_host, &P)) return DBEV_EINVAL;
  if (num_points == 0) return 0;
  hipLaunchKernelGGL(dynamic_voxelize_kernel, dim3(dbev_ceil_div(num_points, 256)), dim3(256), 0,
                     dbev_stream(stream), points, coors, num_points, num_features, P);
  DBEV_LAUNCH_CHECK();
  return 0;
}

extern "C" size_t dbev_hard_voxelize_workspace_bytes(int num_points, const float* voxel_size_host,
                                                      const float* coors_range_host) {
  VoxParams P;
  if (num_points < 0 || !make_params(voxel_size_host, coors_range_host, &P)) return 0;
  const long long ncell = static_cast<long long>(P.grid[0]) * P.grid[1] * P.grid[2];
  return hv_layout(num_points, ncell).total;
}

extern "C" int dbev_hard_voxelize(const float* points, float* voxels, int32_t* coors,
                                  int32_t* num_points_per_voxel, int32_t* voxel_num_out,
                                  int num_points, int num_features, const float* voxel_size_host,
                                  const float* coors_range_host, int max_points, int max_voxels,
                                  int ndim, void* workspace, size_t workspace_bytes,
                                  dbevStream_t stream) {
  if (ndim != 3 || num_features < 3 || num_points < 0 || max_points <= 0 || max_voxels <= 0)
    return DBEV_EINVAL;
  VoxParams P;
  if (!make_params(voxel_size_host, coors_range_host, &P)) return DBEV_EINVAL;
  const long long ncell = static_cast<long long>(P.grid[0]) * P.grid[1] * P.grid[2];
  if (ncell > 0x7fffffffLL) return DBEV_EINVAL;
  const HvLayout L = hv_layout(num_points, ncell);
  if (workspace_bytes < L.total || workspace == nullptr) return DBEV_EINVAL;
  hipStream_t s = dbev_stream(stream);
  char* ws = static_cast<char*>(workspace);
  int* cell = reinterpret_cast<int*>(ws + L.cell);
  int* count = reinterpret_cast<int*>(ws + L.count);
  int* start = reinterpret_cast<int*>(ws + L.start);
  unsigned* list = reinterpret_cast<unsigned*>(ws + L.list);
  unsigned* sorted = reinterpret_cast<unsigned*>(ws + L.sorted);
  int* flag = reinterpret_cast<int*>(ws + L.flag);
  int* vidscan = reinterpret_cast<int*>(ws + L.vidscan);
  int* scanws = reinterpret_cast<int*>(ws + L.scanws);
  int* sortws = reinterpret_cast<int*>(ws + L.sortws);
  int* nheads = reinterpret_cast<int*>(ws + L.nheads);

  // outputs: the reference's caller hands in zeros (voxelize.py:57-62); do not rely on it
  DBEV_HIP_TRY(hipMemsetAsync(voxels, 0, sizeof(float) * static_cast<size_t>(max_voxels) * max_points * num_features, s));
  DBEV_HIP_TRY(hipMemsetAsync(coors, 0, sizeof(int) * static_cast<size_t>(max_voxels) * 3, s));
  DBEV_HIP_TRY(hipMemsetAsync(num_points_per_voxel, 0, sizeof(int) * static_cast<size_t>(max_voxels), s));
  DBEV_HIP_TRY(hipMemsetAsync(voxel_num_out, 0, sizeof(int), s));
  if (num_points == 0) return 0;
  DBEV_HIP_TRY(hipMemsetAsync(count, 0, sizeof(int) * ncell, s));
  const int nb = dbev_ceil_div(num_points, 256);
  hipLaunchKernelGGL(hv_cell_count, dim3(nb), dim3(256), 0, s, points, num_points, num_features, P, cell, count);
  int rc = dbev::exclusive_scan_i32(count, start, ncell, false, nheads + 1 /* n_valid */, scanws, s);
  if (rc) return rc;
  hipLaunchKernelGGL(hv_fill, dim3(nb), dim3(256), 0, s, cell, num_points, start, count, list);
  rc = dbev::segment_sort_u32(start, list, sorted, static_cast<int>(ncell), sortws, s);
  if (rc) return rc;
  hipLaunchKernelGGL(hv_head_flags, dim3(nb), dim3(256), 0, s, cell, num_points, start, sorted, flag);
  rc = dbev::exclusive_scan_i32(flag, vidscan, num_points, false, nheads, scanws, s);
  if (rc) return rc;
  hipLaunchKernelGGL(hv_write, dim3(nb), dim3(256), 0, s, points, num_features, cell, start, sorted, 0,
                     vidscan, voxels, coors, num_points_per_voxel, max_points, max_voxels, P,
                     nheads + 1);
  hipLaunchKernelGGL(hv_voxel_num, dim3(1), dim3(1), 0, s, nheads, max_voxels, voxel_num_out);
  DBEV_LAUNCH_CHECK();
  return 0;
}
