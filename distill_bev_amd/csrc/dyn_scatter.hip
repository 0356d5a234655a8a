// dynamic_scatter (points -> voxels reduce) for gfx950 -- replaces
// mmdet3d/ops/voxel/src/scatter_points_cuda.cu (dynamic_point_to_voxel_forward_gpu :183-239,
// feats_reduce_kernel :80-103, backward :241-308).
//
// The reference sorts the N coordinate rows (at::unique_dim) and then issues one float
// atomic (CAS loop for max) per (point, feature).  Here:
//   * the sorted-unique row order is the ascending linear cell id on the voxel grid, so
//     "unique" is an occupancy histogram + exclusive scan over the dense grid (no sort);
//   * points are grouped per voxel in ascending point id (histogram -> scan -> fill ->
//     segment sort), and ONE wavefront reduces one voxel with lanes = feature channels:
//     coalesced 4C-byte row reads, no float atomics, and a fixed summation order
//     (sum/mean are run-to-run deterministic and equal to the sequential CPU order;
//     the reference's atomicAdd order is not).
#include <math.h>

#include "prims.h"

namespace {

size_t align_up(size_t x) { return (x + 255) & ~static_cast<size_t>(255); }

struct DsLayout { size_t cell, count, vid, cursor, tmp, scanws, sortws, total; };

DsLayout ds_layout(long long n, long long ncell) {
  DsLayout L;
  size_t o = 0;
  L.cell = o;   o += align_up(sizeof(int) * n);
  L.count = o;  o += align_up(sizeof(int) * ncell);
  L.vid = o;    o += align_up(sizeof(int) * (ncell + 1));
  L.cursor = o; o += align_up(sizeof(int) * n);
  L.tmp = o;    o += align_up(sizeof(int) * n);
  L.scanws = o; o += align_up(sizeof(int) * dbev::scan_workspace_ints(ncell > n ? ncell : n));
  L.sortws = o; o += align_up(sizeof(int) * dbev::segment_sort_workspace_ints(n));
  L.total = o;
  return L;
}

// scatter_points_cuda.cu:199: a row with ANY negative entry is invalid.
__global__ __launch_bounds__(256) void ds_cell_count(const int* __restrict__ coors, int n, int gz, int gy,
                                                     int gx, int* __restrict__ cell,
                                                     int* __restrict__ count) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int z = coors[i * 3 + 0], y = coors[i * 3 + 1], x = coors[i * 3 + 2];
  const bool ok = z >= 0 && y >= 0 && x >= 0 && z < gz && y < gy && x < gx;
  const int lin = ok ? (z * gy + y) * gx + x : -1;
  cell[i] = lin;
  if (ok) atomicAdd(&count[lin], 1);
}

__global__ __launch_bounds__(256) void ds_emit_voxels(const int* __restrict__ count,
                                                      const int* __restrict__ vid, long long ncell,
                                                      int gy, int gx, int* __restrict__ out_coors,
                                                      int* __restrict__ reduce_count,
                                                      int* __restrict__ cursor) {
  const long long c = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (c >= ncell) return;
  const int k = count[c];
  if (k <= 0) return;
  const int v = vid[c];
  out_coors[v * 3 + 0] = static_cast<int>(c / (static_cast<long long>(gy) * gx));
  out_coors[v * 3 + 1] = static_cast<int>((c / gx) % gy);
  out_coors[v * 3 + 2] = static_cast<int>(c % gx);
  reduce_count[v] = k;
  cursor[v] = k;
}

__global__ __launch_bounds__(256) void ds_map_points(const int* __restrict__ cell, int n,
                                                     const int* __restrict__ vid,
                                                     int* __restrict__ coors_map) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int c = cell[i];
  coors_map[i] = c < 0 ? -1 : vid[c];
}

__global__ __launch_bounds__(256) void ds_fill(const int* __restrict__ coors_map, int n,
                                               const int* __restrict__ vstart, int* __restrict__ cursor,
                                               unsigned* __restrict__ tmp) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int v = coors_map[i];
  if (v < 0) return;
  const int pos = atomicSub(&cursor[v], 1) - 1;
  tmp[vstart[v] + pos] = static_cast<unsigned>(i);
}

// one wave per voxel, lane = channel.  reduce_type: 0 sum, 1 mean, 2 max (voxelization.h:4)
__global__ __launch_bounds__(256) void ds_reduce(const float* __restrict__ feats,
                                                 const int* __restrict__ vstart,
                                                 const unsigned* __restrict__ vlist,
                                                 float* __restrict__ reduced, int m, int C,
                                                 int reduce_type) {
  const int v = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (v >= m) return;
  const int st = vstart[v];
  const int L = vstart[v + 1] - st;
  for (int ch = lane; ch < C; ch += 64) {
    float acc = reduce_type == 2 ? -INFINITY : 0.f;
    int j = 0;
    for (; j + 4 <= L; j += 4) {
      const float f0 = feats[static_cast<size_t>(vlist[st + j + 0]) * C + ch];
      const float f1 = feats[static_cast<size_t>(vlist[st + j + 1]) * C + ch];
      const float f2 = feats[static_cast<size_t>(vlist[st + j + 2]) * C + ch];
      const float f3 = feats[static_cast<size_t>(vlist[st + j + 3]) * C + ch];
      if (reduce_type == 2) {
        acc = fmaxf(fmaxf(fmaxf(fmaxf(acc, f0), f1), f2), f3);
      } else {
        acc += f0; acc += f1; acc += f2; acc += f3;  // sequential point-id order
      }
    }
    for (; j < L; ++j) {
      const float f = feats[static_cast<size_t>(vlist[st + j]) * C + ch];
      acc = reduce_type == 2 ? fmaxf(acc, f) : acc + f;
    }
    if (reduce_type == 1) acc = acc / static_cast<float>(L);
    reduced[static_cast<size_t>(v) * C + ch] = acc;
  }
}

// C % 4 == 0 and C/4 a power of two <= 64: a group of C/4 lanes per voxel (each lane 4 channels, one float4 row
// gather per point), 64 / (C/4) voxels per wave.  LiDAR pillars hold ~2 points: the wave-per-voxel kernel above is
// a chain of dependent loads per voxel with 1 KB of payload; several voxels per wave overlap those chains.
// Same per-channel operation order (sequential in point-id order) -> bit-identical to ds_reduce.
__global__ __launch_bounds__(256) void ds_reduce_vec4(const float4* __restrict__ feats,
                                                      const int* __restrict__ vstart,
                                                      const unsigned* __restrict__ vlist,
                                                      float4* __restrict__ reduced, int m, int C4,
                                                      int reduce_type) {
  const long long tid = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  const int v = static_cast<int>(tid / C4);
  const int q = static_cast<int>(tid - static_cast<long long>(v) * C4);
  if (v >= m) return;
  const int st = vstart[v];
  const int L = vstart[v + 1] - st;
  const float init = reduce_type == 2 ? -INFINITY : 0.f;
  float4 acc = make_float4(init, init, init, init);
  int j = 0;
  for (; j + 4 <= L; j += 4) {
    float4 f[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) f[u] = feats[static_cast<size_t>(vlist[st + j + u]) * C4 + q];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (reduce_type == 2) {
        acc.x = fmaxf(acc.x, f[u].x); acc.y = fmaxf(acc.y, f[u].y); acc.z = fmaxf(acc.z, f[u].z); acc.w = fmaxf(acc.w, f[u].w);
      } else {
        acc.x += f[u].x; acc.y += f[u].y; acc.z += f[u].z; acc.w += f[u].w;     // sequential point-id order
      }
    }
  }
  for (; j < L; ++j) {
    const float4 f = feats[static_cast<size_t>(vlist[st + j]) * C4 + q];
    if (reduce_type == 2) {
      acc.x = fmaxf(acc.x, f.x); acc.y = fmaxf(acc.y, f.y); acc.z = fmaxf(acc.z, f.z); acc.w = fmaxf(acc.w, f.w);
    } else {
      acc.x += f.x; acc.y += f.y; acc.z += f.z; acc.w += f.w;
    }
  }
  if (reduce_type == 1) {
    const float fl = static_cast<float>(L);
    acc.x = acc.x / fl; acc.y = acc.y / fl; acc.z = acc.z / fl; acc.w = acc.w / fl;
  }
  st_nt(reduced + static_cast<size_t>(v) * C4 + q, acc);
}

__global__ __launch_bounds__(256) void ds_bwd_add(float* __restrict__ grad_feats,
                                                  const float* __restrict__ grad_reduced,
                                                  const int* __restrict__ coors_map,
                                                  const int* __restrict__ reduce_count, long long total,
                                                  int C, int reduce_type) {
  for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < total;
       i += static_cast<long long>(gridDim.x) * blockDim.x) {
    const long long p = i / C;
    const int ch = static_cast<int>(i - p * C);
    const int v = coors_map[p];
    float g = 0.f;
    if (v >= 0) {
      g = grad_reduced[static_cast<size_t>(v) * C + ch];
      if (reduce_type == 1) g = g / static_cast<float>(reduce_count[v]);
    }
    grad_feats[i] = g;
  }
}

// max: the LOWEST point id whose feature equals the voxel max takes the gradient
// (atomicMin traceback, scatter_points_cuda.cu:154-157); lists are ascending -> first hit.
__global__ __launch_bounds__(256) void ds_bwd_max(float* __restrict__ grad_feats,
                                                  const float* __restrict__ grad_reduced,
                                                  const float* __restrict__ feats,
                                                  const float* __restrict__ reduced,
                                                  const int* __restrict__ vstart,
                                                  const unsigned* __restrict__ vlist, int m, int C) {
  const int v = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (v >= m) return;
  const int st = vstart[v];
  const int L = vstart[v + 1] - st;
  for (int ch = lane; ch < C; ch += 64) {
    const float r = reduced[static_cast<size_t>(v) * C + ch];
    for (int j = 0; j < L; ++j) {
      const size_t o = static_cast<size_t>(vlist[st + j]) * C + ch;
      if (feats[o] == r) {
        grad_feats[o] = grad_reduced[static_cast<size_t>(v) * C + ch];
        break;
      }
    }
  }
}

}  // namespace

extern "C" size_t dbev_dynamic_scatter_workspace_bytes(int num_points, int grid_z, int grid_y, int grid_x) {
  if (num_points < 0 || grid_z <= 0 || grid_y <= 0 || grid_x <= 0) return 0;
  return ds_layout(num_points, static_cast<long long>(grid_z) * grid_y * grid_x).total;
}

extern "C" int dbev_dynamic_scatter_prepare(const int32_t* coors, int num_points, int grid_z, int grid_y,
                                            int grid_x, int32_t* out_coors, int32_t* coors_map,
                                            int32_t* reduce_count, int32_t* voxel_point_start,
                                            int32_t* voxel_point_list, int32_t* num_voxels_out,
                                            void* workspace, size_t workspace_bytes,
                                            dbevStream_t stream) {
  if (num_points < 0 || grid_z <= 0 || grid_y <= 0 || grid_x <= 0) return DBEV_EINVAL;
  const long long ncell = static_cast<long long>(grid_z) * grid_y * grid_x;
  if (ncell > 0x7fffffffLL) return DBEV_EINVAL;
  const DsLayout L = ds_layout(num_points, ncell);
  if (workspace == nullptr || workspace_bytes < L.total) return DBEV_EINVAL;
  hipStream_t s = dbev_stream(stream);
  DBEV_HIP_TRY(hipMemsetAsync(num_voxels_out, 0, sizeof(int), s));
  DBEV_HIP_TRY(hipMemsetAsync(voxel_point_start, 0, sizeof(int) * (static_cast<size_t>(num_points) + 1), s));
  if (num_points == 0) return 0;
  char* ws = static_cast<char*>(workspace);
  int* cell = reinterpret_cast<int*>(ws + L.cell);
  int* count = reinterpret_cast<int*>(ws + L.count);
  int* vid = reinterpret_cast<int*>(ws + L.vid);
  int* cursor = reinterpret_cast<int*>(ws + L.cursor);
  unsigned* tmp = reinterpret_cast<unsigned*>(ws + L.tmp);
  int* scanws = reinterpret_cast<int*>(ws + L.scanws);
  int* sortws = reinterpret_cast<int*>(ws + L.sortws);
  DBEV_HIP_TRY(hipMemsetAsync(count, 0, sizeof(int) * ncell, s));
  DBEV_HIP_TRY(hipMemsetAsync(reduce_count, 0, sizeof(int) * static_cast<size_t>(num_points), s));
  const int nb = dbev_ceil_div(num_points, 256);
  hipLaunchKernelGGL(ds_cell_count, dim3(nb), dim3(256), 0, s, coors, num_points, grid_z, grid_y, grid_x, cell, count);
  int rc = dbev::exclusive_scan_i32(count, vid, ncell, true, num_voxels_out, scanws, s);
  if (rc) return rc;
  hipLaunchKernelGGL(ds_emit_voxels, dim3(dbev_ceil_div(ncell, 256)), dim3(256), 0, s, count, vid, ncell,
                     grid_y, grid_x, out_coors, reduce_count, cursor);
  hipLaunchKernelGGL(ds_map_points, dim3(nb), dim3(256), 0, s, cell, num_points, vid, coors_map);
  rc = dbev::exclusive_scan_i32(reduce_count, voxel_point_start, num_points, false, nullptr, scanws, s);
  if (rc) return rc;
  hipLaunchKernelGGL(ds_fill, dim3(nb), dim3(256), 0, s, coors_map, num_points, voxel_point_start, cursor, tmp);
  rc = dbev::segment_sort_u32(voxel_point_start, tmp, reinterpret_cast<unsigned*>(voxel_point_list),
                              num_points, sortws, s);
  if (rc) return rc;
  DBEV_LAUNCH_CHECK();
  return 0;
}

extern "C" int dbev_dynamic_scatter_reduce(const float* feats, const int32_t* voxel_point_start,
                                           const int32_t* voxel_point_list, float* reduced,
                                           int num_voxels, int num_feats, int reduce_type,
                                           dbevStream_t stream) {
  if (num_voxels < 0 || num_feats <= 0 || reduce_type < 0 || reduce_type > 2) return DBEV_EINVAL;
  if (num_voxels == 0) return 0;
  const int C4 = num_feats >> 2;
  if ((num_feats & 3) == 0 && C4 <= 64 && (C4 & (C4 - 1)) == 0) {
    const long long threads = static_cast<long long>(num_voxels) * C4;
    hipLaunchKernelGGL(ds_reduce_vec4, dim3(dbev_ceil_div(threads, 256)), dim3(256), 0, dbev_stream(stream),
                       reinterpret_cast<const float4*>(feats), voxel_point_start,
                       reinterpret_cast<const unsigned*>(voxel_point_list), reinterpret_cast<float4*>(reduced),
                       num_voxels, C4, reduce_type);
    DBEV_LAUNCH_CHECK();
    return 0;
  }
  hipLaunchKernelGGL(ds_reduce, dim3(dbev_ceil_div(num_voxels, 4)), dim3(256), 0, dbev_stream(stream), feats,
                     voxel_point_start, reinterpret_cast<const unsigned*>(voxel_point_list), reduced,
                     num_voxels, num_feats, reduce_type);
  DBEV_LAUNCH_CHECK();
  return 0;
}

extern "C" int dbev_dynamic_scatter_backward(float* grad_feats, const float* grad_reduced,
                                             const float* feats, const float* reduced,
                                             const int32_t* coors_map, const int32_t* reduce_count,
                                             const int32_t* voxel_point_start,
                                             const int32_t* voxel_point_list, int num_points,
                                             int num_voxels, int num_feats, int reduce_type,
                                             dbevStream_t stream) {
  if (num_points < 0 || num_voxels < 0 || num_feats <= 0 || reduce_type < 0 || reduce_type > 2)
    return DBEV_EINVAL;
  if (num_points == 0) return 0;
  hipStream_t s = dbev_stream(stream);
  const long long total = static_cast<long long>(num_points) * num_feats;
  if (reduce_type == 2 || num_voxels == 0) {
    DBEV_HIP_TRY(hipMemsetAsync(grad_feats, 0, sizeof(float) * total, s));
    if (num_voxels == 0) return 0;
    hipLaunchKernelGGL(ds_bwd_max, dim3(dbev_ceil_div(num_voxels, 4)), dim3(256), 0, s, grad_feats,
                       grad_reduced, feats, reduced, voxel_point_start,
                       reinterpret_cast<const unsigned*>(voxel_point_list), num_voxels, num_feats);
  } else {
    long long blocks = (total + 255) / 256;
    if (blocks > DBEV_MAX_GRID * 4) blocks = DBEV_MAX_GRID * 4;
    hipLaunchKernelGGL(ds_bwd_add, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, s, grad_feats,
                       grad_reduced, coors_map, reduce_count, total, num_feats, reduce_type);
  }
  DBEV_LAUNCH_CHECK();
  return 0;
}
