// Fused teacher pillar path for gfx950:  points -> voxelize -> pillar feature net -> BEV canvas.
//
// Replaces, for the frozen (eval, no_grad) CenterPoint-pillar teacher, the whole sequence
//   DynamicCenterPoint.voxelize          (dynamic_centerpoint.py:71-93, per-sample launches + pad + cat)
//   DynamicPillarFeatureNet.forward      (pillar_encoder.py:283-338: cluster_scatter(mean) -> canvas gather
//                                         -> decorate to 10 ch -> Linear(10,64,no bias)+BN1d+ReLU -> pfn_scatter(max))
//   PointPillarsScatter.forward_batch    (pillar_scatter.py:62-102)
// which in the reference is two sorted-unique passes, float atomics, a [C, 512*512*B] broadcast
// canvas, a python loop per sample and (in any op-level implementation) a device->host read of
// the data-dependent pillar count M.  Here nothing leaves the device and nothing is M-shaped on
// the host: buffers are sized by the point count, M lives in a device int.
//
//   vfe_cell_count : cell of every point (same fp32 floor((p-min)/vs) as dbev_dynamic_voxelize), int histogram
//   scan           : occupied cells -> pillar ids in (b, y, x) order  (== sorted-unique order)
//   vfe_emit       : cellmap[b,y,x] = pillar id / -1, per-pillar counts
//   scan, fill, segment sort : CSR pillar -> points, ascending point id
//   vfe_reduce     : ONE wavefront per pillar, lane = output channel: sequential xyz mean, decorate,
//                    10 FMAs against the lane's weight column, folded BN, ReLU, running max
//   launch_canvas  : output-stationary canvas write (pillar_scatter.hip)
#include <math.h>

#include "pillar_scatter.h"
#include "prims.h"

namespace {

constexpr int MAX_B = 64;
constexpr int MAX_K = 16;   // F + 5 decorated input channels

struct VfeParams {
  float vs[3], rmin[3];
  int grid[3];
  float vx, vy, x_offset, y_offset;   // pillar_encoder.py:86-90
  int sample_start[MAX_B + 1];
  int B;
};

size_t align_up(size_t x) { return (x + 255) & ~static_cast<size_t>(255); }

struct VfeLayout { size_t cell, count, vid, vcount, cursor, vstart, tmp, list, vcell, scanws, sortws, total; };

VfeLayout vfe_layout(long long n, long long ncell) {
  VfeLayout L;
  size_t o = 0;
  L.cell = o;   o += align_up(sizeof(int) * n);
  L.count = o;  o += align_up(sizeof(int) * ncell);
  L.vid = o;    o += align_up(sizeof(int) * (ncell + 1));
  L.vcount = o; o += align_up(sizeof(int) * n);
  L.cursor = o; o += align_up(sizeof(int) * n);
  L.vstart = o; o += align_up(sizeof(int) * (n + 1));
  L.tmp = o;    o += align_up(sizeof(int) * n);
  L.list = o;   o += align_up(sizeof(int) * n);
  L.vcell = o;  o += align_up(sizeof(int) * n);
  L.scanws = o; o += align_up(sizeof(int) * dbev::scan_workspace_ints(ncell > n ? ncell : n));
  L.sortws = o; o += align_up(sizeof(int) * dbev::segment_sort_workspace_ints(n));
  L.total = o;
  return L;
}

__global__ __launch_bounds__(256) void vfe_cell_count(const float* __restrict__ points, int n, int nf,
                                                      VfeParams P, int* __restrict__ cell,
                                                      int* __restrict__ count) {
#pragma clang fp contract(off)
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* p = points + static_cast<size_t>(i) * nf;
  int c[3];
  bool ok = true;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const float fl = floorf((p[j] - P.rmin[j]) / P.vs[j]);   // voxelization_cpu.cpp:22-32
    ok = ok && (fl >= 0.f) && (fl < static_cast<float>(P.grid[j]));
    c[j] = static_cast<int>(fl);
  }
  int lin = -1;
  if (ok) {
    int b = 0;
    while (b + 1 < P.B && i >= P.sample_start[b + 1]) ++b;
    lin = (b * P.grid[1] + c[1]) * P.grid[0] + c[0];
    atomicAdd(&count[lin], 1);
  }
  cell[i] = lin;
}

__global__ __launch_bounds__(256) void vfe_emit(const int* __restrict__ count, const int* __restrict__ vid,
                                                int ncell, int* __restrict__ cellmap,
                                                int* __restrict__ vcount, int* __restrict__ cursor,
                                                int* __restrict__ vcell) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= ncell) return;
  const int k = count[c];
  if (k > 0) {
    const int v = vid[c];
    vcount[v] = k;
    cursor[v] = k;
    vcell[v] = c;
    cellmap[c] = v;
  } else {
    cellmap[c] = -1;
  }
}

__global__ __launch_bounds__(256) void vfe_fill(const int* __restrict__ cell, int n, const int* __restrict__ vid,
                                                const int* __restrict__ vstart, int* __restrict__ cursor,
                                                unsigned* __restrict__ tmp) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int c = cell[i];
  if (c < 0) return;
  const int v = vid[c];
  const int pos = atomicSub(&cursor[v], 1) - 1;
  tmp[vstart[v] + pos] = static_cast<unsigned>(i);
}

// one wave per pillar; lane = output channel (Cout <= 64)
__global__ __launch_bounds__(256) void vfe_reduce(const float* __restrict__ points, int nf,
                                                  const int* __restrict__ vstart,
                                                  const unsigned* __restrict__ vlist,
                                                  const int* __restrict__ vcell,
                                                  const int* __restrict__ num_voxels,
                                                  const float* __restrict__ W,        // [Cout, K] row-major
                                                  const float* __restrict__ bn_w, const float* __restrict__ bn_b,
                                                  const float* __restrict__ bn_mean, const float* __restrict__ bn_var,
                                                  float bn_eps, int Cout, int K, VfeParams P,
                                                  float* __restrict__ voxel_feats) {
  const int lane = threadIdx.x & 63;
  // wave-uniform by construction; readfirstlane lets the compiler keep the pillar bookkeeping (list bounds,
  // point ids, point coordinates: identical for all 64 lanes) in SGPRs / scalar loads
  const int wave0 = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  const int M = *num_voxels;
  // per-lane constants, loaded ONCE per wave (the wave then walks ~M/nwaves pillars): weight column of
  // output channel `lane` and its folded BatchNorm1d (eval): (y - mean) / sqrt(var + eps) * weight + bias
  // (static register indices only: a runtime-indexed array would live in scratch)
  constexpr int MAX_F = MAX_K - 5;
  float wf[MAX_F];
  const bool act = lane < Cout;
#pragma unroll
  for (int k = 0; k < MAX_F; ++k) wf[k] = (act && k < nf) ? W[lane * K + k] : 0.f;
  const float wd0 = act ? W[lane * K + nf + 0] : 0.f, wd1 = act ? W[lane * K + nf + 1] : 0.f;
  const float wd2 = act ? W[lane * K + nf + 2] : 0.f, wd3 = act ? W[lane * K + nf + 3] : 0.f;
  const float wd4 = act ? W[lane * K + nf + 4] : 0.f;
  const float inv_std = act ? 1.f / sqrtf(bn_var[lane] + bn_eps) : 0.f;
  const float mu = act ? bn_mean[lane] : 0.f, ga = act ? bn_w[lane] : 0.f, be = act ? bn_b[lane] : 0.f;
  // next pillar's bookkeeping is fetched while the current one is reduced
  int st_n = 0, en_n = 0, c_n = 0;
  if (wave0 < M) { st_n = vstart[wave0]; en_n = vstart[wave0 + 1]; c_n = vcell[wave0]; }
  for (int v = wave0; v < M; v += nwaves) {
    const int st = st_n;
    const int L = en_n - st_n;
    const int c = c_n;
    if (v + nwaves < M) { st_n = vstart[v + nwaves]; en_n = vstart[v + nwaves + 1]; c_n = vcell[v + nwaves]; }
    const int cx = c % P.grid[0];
    const int cy = (c / P.grid[0]) % P.grid[1];
    // pillar centre: coors.type_as(features) * vx + x_offset (pillar_encoder.py:318-321), fp32
    const float pcx = static_cast<float>(cx) * P.vx + P.x_offset;
    const float pcy = static_cast<float>(cy) * P.vy + P.y_offset;
    // cluster mean of xyz: sequential fp32 sum in point order / count (DynamicScatter mean)
    float sx = 0.f, sy = 0.f, sz = 0.f;
    for (int j = 0; j < L; ++j) {
      const float* p = points + static_cast<size_t>(vlist[st + j]) * nf;
      sx += p[0]; sy += p[1]; sz += p[2];
    }
    const float fl = static_cast<float>(L);
    const float mx = sx / fl, my = sy / fl, mz = sz / fl;
    float acc = -INFINITY;
    for (int j = 0; j < L; ++j) {
      const float* p = points + static_cast<size_t>(vlist[st + j]) * nf;
      float y = 0.f;
#pragma unroll
      for (int k = 0; k < MAX_F; ++k)
        if (k < nf) y = fmaf(wf[k], p[k], y);
      y = fmaf(wd0, p[0] - mx, y);
      y = fmaf(wd1, p[1] - my, y);
      y = fmaf(wd2, p[2] - mz, y);
      y = fmaf(wd3, p[0] - pcx, y);
      y = fmaf(wd4, p[1] - pcy, y);
      y = (y - mu) * inv_std * ga + be;
      y = fmaxf(y, 0.f);
      acc = fmaxf(acc, y);
    }
    if (act) voxel_feats[static_cast<size_t>(v) * Cout + lane] = acc;
  }
}

// Cout == 64: 16 lanes per pillar (lane q owns output channels 4q..4q+3), FOUR pillars per wave.  The wave-per-
// pillar kernel above is bound by the dependent chain list entry -> point row -> 10 FMAs -> store per pillar
// (1.8 points per pillar on average: 2.7 us each); four independent chains per wave divide that by ~3.
// Every output channel sees exactly the same fp32 operation sequence as in vfe_reduce -> bit-identical features.
template <int NF>
__global__ __launch_bounds__(256) void vfe_reduce_c64(const float* __restrict__ points,
                                                      const int* __restrict__ vstart,
                                                      const unsigned* __restrict__ vlist,
                                                      const int* __restrict__ vcell,
                                                      const int* __restrict__ num_voxels,
                                                      const float* __restrict__ W,        // [64, K] row-major
                                                      const float* __restrict__ bn_w, const float* __restrict__ bn_b,
                                                      const float* __restrict__ bn_mean, const float* __restrict__ bn_var,
                                                      float bn_eps, VfeParams P, float4* __restrict__ voxel_feats) {
  constexpr int K = NF + 5;
  const int lane = threadIdx.x & 63;
  const int grp = lane >> 4, q = lane & 15;
  const int wave0 = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  const int M = *num_voxels;
  float w[4][K], inv_std[4], mu[4], ga[4], be[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int ch = q * 4 + c;
#pragma unroll
    for (int k = 0; k < K; ++k) w[c][k] = W[ch * K + k];
    inv_std[c] = 1.f / sqrtf(bn_var[ch] + bn_eps);
    mu[c] = bn_mean[ch]; ga[c] = bn_w[ch]; be[c] = bn_b[ch];
  }
  // the next pillar's bookkeeping (list bounds, cell, first point id) is fetched while the current one is reduced
  int st_n = 0, en_n = 0, c_n = 0;
  unsigned p0_n = 0u;
  {
    const int v = wave0 * 4 + grp;
    if (v < M) { st_n = vstart[v]; en_n = vstart[v + 1]; c_n = vcell[v]; p0_n = vlist[st_n]; }
  }
  for (int v0 = wave0 * 4; v0 < M; v0 += nwaves * 4) {
    const int v = v0 + grp;
    const bool live = v < M;
    const int st = st_n;
    const int L = en_n - st_n;
    const int c = c_n;
    const unsigned first = p0_n;
    {
      const int vn = v + nwaves * 4;
      st_n = en_n = c_n = 0;
      if (vn < M) { st_n = vstart[vn]; en_n = vstart[vn + 1]; c_n = vcell[vn]; p0_n = vlist[st_n]; }
    }
    const int cx = c % P.grid[0];
    const int cy = (c / P.grid[0]) % P.grid[1];
    const float pcx = static_cast<float>(cx) * P.vx + P.x_offset;
    const float pcy = static_cast<float>(cy) * P.vy + P.y_offset;
    float sx = 0.f, sy = 0.f, sz = 0.f;
    for (int j = 0; j < L; ++j) {
      const float* p = points + static_cast<size_t>(j == 0 ? first : vlist[st + j]) * NF;
      sx += p[0]; sy += p[1]; sz += p[2];
    }
    const float fl = static_cast<float>(L);
    const float mx = sx / fl, my = sy / fl, mz = sz / fl;
    float acc[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    for (int j = 0; j < L; ++j) {
      const float* p = points + static_cast<size_t>(j == 0 ? first : vlist[st + j]) * NF;
      float pv[NF];
#pragma unroll
      for (int k = 0; k < NF; ++k) pv[k] = p[k];
      const float d0 = pv[0] - mx, d1 = pv[1] - my, d2 = pv[2] - mz, d3 = pv[0] - pcx, d4 = pv[1] - pcy;
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) {
        float y = 0.f;
#pragma unroll
        for (int k = 0; k < NF; ++k) y = fmaf(w[cc][k], pv[k], y);
        y = fmaf(w[cc][NF + 0], d0, y);
        y = fmaf(w[cc][NF + 1], d1, y);
        y = fmaf(w[cc][NF + 2], d2, y);
        y = fmaf(w[cc][NF + 3], d3, y);
        y = fmaf(w[cc][NF + 4], d4, y);
        y = (y - mu[cc]) * inv_std[cc] * ga[cc] + be[cc];
        y = fmaxf(y, 0.f);
        acc[cc] = fmaxf(acc[cc], y);
      }
    }
    if (live) voxel_feats[static_cast<size_t>(v) * 16 + q] = make_float4(acc[0], acc[1], acc[2], acc[3]);
  }
}

}  // namespace

extern "C" size_t dbev_pillar_vfe_workspace_bytes(int n_points, int B, int ny, int nx) {
  if (n_points < 0 || B <= 0 || ny <= 0 || nx <= 0) return 0;
  return vfe_layout(n_points, static_cast<long long>(B) * ny * nx).total;
}

extern "C" int dbev_pillar_vfe_canvas(const float* points, int n_points, int num_features,
                                      const int32_t* sample_start_host, int B, const float* voxel_size_host,
                                      const float* coors_range_host, const float* pfn_weight,
                                      const float* bn_weight, const float* bn_bias, const float* bn_mean,
                                      const float* bn_var, float bn_eps, int out_channels, float* voxel_feats,
                                      int32_t* cellmap, int32_t* num_voxels_out, float* canvas,
                                      int channels_last, void* workspace, size_t workspace_bytes,
                                      dbevStream_t stream) {
  if (n_points < 0 || num_features < 3 || num_features + 5 > MAX_K || B <= 0 || B > MAX_B ||
      out_channels <= 0 || out_channels > 64)
    return DBEV_EINVAL;
  VfeParams P;
  for (int i = 0; i < 3; ++i) {
    if (!(voxel_size_host[i] > 0.f)) return DBEV_EINVAL;
    P.vs[i] = voxel_size_host[i];
    P.rmin[i] = coors_range_host[i];
    P.grid[i] = static_cast<int>(round((coors_range_host[3 + i] - coors_range_host[i]) / voxel_size_host[i]));
    if (P.grid[i] <= 0) return DBEV_EINVAL;
  }
  if (P.grid[2] != 1) return DBEV_EINVAL;   // pillars: one cell along z
  P.vx = voxel_size_host[0];
  P.vy = voxel_size_host[1];
  P.x_offset = static_cast<float>(static_cast<double>(voxel_size_host[0]) / 2 + coors_range_host[0]);
  P.y_offset = static_cast<float>(static_cast<double>(voxel_size_host[1]) / 2 + coors_range_host[1]);
  P.B = B;
  for (int b = 0; b <= B; ++b) P.sample_start[b] = sample_start_host[b];
  if (P.sample_start[0] != 0 || P.sample_start[B] != n_points) return DBEV_EINVAL;
  const int nx = P.grid[0], ny = P.grid[1];
  const long long ncell = static_cast<long long>(B) * ny * nx;
  if (ncell > 0x7fffffffLL) return DBEV_EINVAL;
  const VfeLayout L = vfe_layout(n_points, ncell);
  if (workspace == nullptr || workspace_bytes < L.total) return DBEV_EINVAL;
  hipStream_t s = dbev_stream(stream);
  char* ws = static_cast<char*>(workspace);
  int* cell = reinterpret_cast<int*>(ws + L.cell);
  int* count = reinterpret_cast<int*>(ws + L.count);
  int* vid = reinterpret_cast<int*>(ws + L.vid);
  int* vcount = reinterpret_cast<int*>(ws + L.vcount);
  int* cursor = reinterpret_cast<int*>(ws + L.cursor);
  int* vstart = reinterpret_cast<int*>(ws + L.vstart);
  unsigned* tmp = reinterpret_cast<unsigned*>(ws + L.tmp);
  unsigned* list = reinterpret_cast<unsigned*>(ws + L.list);
  int* vcell = reinterpret_cast<int*>(ws + L.vcell);
  int* scanws = reinterpret_cast<int*>(ws + L.scanws);
  int* sortws = reinterpret_cast<int*>(ws + L.sortws);

  DBEV_HIP_TRY(hipMemsetAsync(count, 0, sizeof(int) * ncell, s));
  DBEV_HIP_TRY(hipMemsetAsync(vcount, 0, sizeof(int) * static_cast<size_t>(n_points > 0 ? n_points : 1), s));
  const int nb = dbev_ceil_div(n_points > 0 ? n_points : 1, 256);
  if (n_points > 0)
    hipLaunchKernelGGL(vfe_cell_count, dim3(nb), dim3(256), 0, s, points, n_points, num_features, P, cell, count);
  int rc = dbev::exclusive_scan_i32(count, vid, ncell, true, num_voxels_out, scanws, s);
  if (rc) return rc;
  hipLaunchKernelGGL(vfe_emit, dim3(dbev_ceil_div(ncell, 256)), dim3(256), 0, s, count, vid,
                     static_cast<int>(ncell), cellmap, vcount, cursor, vcell);
  if (n_points > 0) {
    rc = dbev::exclusive_scan_i32(vcount, vstart, n_points, false, nullptr, scanws, s);
    if (rc) return rc;
    hipLaunchKernelGGL(vfe_fill, dim3(nb), dim3(256), 0, s, cell, n_points, vid, vstart, cursor, tmp);
    rc = dbev::segment_sort_u32(vstart, tmp, list, n_points, sortws, s);
    if (rc) return rc;
    if (out_channels == 64 && (num_features == 4 || num_features == 5)) {
      // M is only known on the device: persistent grid sized for the worst case of one pillar per point
      if (num_features == 5)
        hipLaunchKernelGGL((vfe_reduce_c64<5>), dim3(DBEV_MAX_GRID), dim3(256), 0, s, points, vstart, list, vcell,
                           num_voxels_out, pfn_weight, bn_weight, bn_bias, bn_mean, bn_var, bn_eps, P,
                           reinterpret_cast<float4*>(voxel_feats));
      else
        hipLaunchKernelGGL((vfe_reduce_c64<4>), dim3(DBEV_MAX_GRID), dim3(256), 0, s, points, vstart, list, vcell,
                           num_voxels_out, pfn_weight, bn_weight, bn_bias, bn_mean, bn_var, bn_eps, P,
                           reinterpret_cast<float4*>(voxel_feats));
    } else {
      hipLaunchKernelGGL(vfe_reduce, dim3(DBEV_MAX_GRID), dim3(256), 0, s, points, num_features,
                         vstart, list, vcell, num_voxels_out, pfn_weight, bn_weight, bn_bias, bn_mean, bn_var,
                         bn_eps, out_channels, num_features + 5, P, voxel_feats);
    }
  }
  if (canvas == nullptr) return 0;     // caller writes the canvas itself (dbev_pillars_canvas)
  return dbev::launch_canvas(voxel_feats, cellmap, canvas, out_channels, B, ny, nx, channels_last, s);
}

extern "C" int dbev_pillars_canvas(const float* voxel_feats, const int32_t* cellmap, float* canvas, int C, int B,
                                   int ny, int nx, int channels_last, dbevStream_t stream) {
  if (C <= 0 || B <= 0 || ny <= 0 || nx <= 0 || voxel_feats == nullptr || cellmap == nullptr || canvas == nullptr)
    return DBEV_EINVAL;
  return dbev::launch_canvas(voxel_feats, cellmap, canvas, C, B, ny, nx, channels_last, dbev_stream(stream));
}
