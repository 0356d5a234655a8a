// PointPillarsScatter for gfx950 -- replaces the per-sample python loop of
// mmdet3d/models/middle_encoders/pillar_scatter.py:62-102 (zero canvas, boolean-mask gather,
// index_put, stack) with one output-stationary pass over the whole batch canvas:
//   step 1: cellmap[b, y, x] = pillar row id or -1                 (4 B per BEV cell)
//   step 2: every canvas element is written exactly once -- the pillar's feature or 0 --
//           so there is no separate 4*C*ny*nx*B-byte memset and no scattered store.
// NCHW output (the reference layout) goes through an LDS transpose tile so that both the
// pillar-row reads (4C contiguous bytes) and the canvas writes (64 consecutive x) coalesce;
// channels_last output ([B, ny, nx, C] physical) is a straight row copy.
// HBM roofline: M(4C+16) + 4 C ny nx B bytes; write dominated.
#include "common.h"
#include "pillar_scatter.h"

namespace {

__global__ __launch_bounds__(256) void ps_cellmap(const int* __restrict__ coors, int m, int B, int ny,
                                                  int nx, int* __restrict__ cellmap) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= m) return;
  const int b = coors[v * 4 + 0], y = coors[v * 4 + 2], x = coors[v * 4 + 3];
  if (b < 0 || b >= B || y < 0 || y >= ny || x < 0 || x >= nx) return;
  // duplicates: the later row wins (sequential index_put semantics)
  atomicMax(&cellmap[(b * ny + y) * nx + x], v);
}

constexpr int TILE_X = 64;

// grid: (ceil(nx/64), ny, B); block 256.  dynamic LDS: TILE_X * (C + 1) floats.
__global__ __launch_bounds__(256) void ps_canvas_nchw(const float* __restrict__ feats,
                                                      const int* __restrict__ cellmap,
                                                      float* __restrict__ canvas, int C, int ny, int nx) {
  extern __shared__ __attribute__((aligned(16))) float tile[];
  __shared__ int vids[TILE_X];
  const int x0 = blockIdx.x * TILE_X, y = blockIdx.y, b = blockIdx.z;
  const int t = threadIdx.x;
  if (t < TILE_X) vids[t] = (x0 + t < nx) ? cellmap[(b * ny + y) * nx + x0 + t] : -1;
  __syncthreads();
  const int ld = C + 1;
  for (int i = t; i < TILE_X * C; i += 256) {
    const int cx = i / C, ch = i - cx * C;
    const int v = vids[cx];
    tile[cx * ld + ch] = v >= 0 ? feats[static_cast<size_t>(v) * C + ch] : 0.f;
  }
  __syncthreads();
  for (int i = t; i < TILE_X * C; i += 256) {
    const int ch = i / TILE_X, cx = i - ch * TILE_X;
    if (x0 + cx < nx)
      canvas[((static_cast<size_t>(b) * C + ch) * ny + y) * nx + x0 + cx] = tile[cx * ld + ch];
  }
}

// Wide-tile variant (C == 64, nx % 256 == 0): 256 consecutive x cells x 64 channels per 512-thread
// workgroup.  The LDS tile is channel-major ([64][256 + 4] floats): it is zero-filled with 16-byte
// LDS stores, only the OCCUPIED cells' rows are fetched (16 lanes x float4 = one coalesced 256 B
// row per lane group; pillar rows of one BEV row are consecutive in memory because dynamic scatter
// emits them in (b, y, x) order) and the output side is ds_read_b128 + one 16-byte global store per
// lane -- every channel row leaves as a contiguous 1 KiB run.  All index arithmetic is shifts
// (C and the tile width are compile-time).  Tiles without any pillar skip LDS entirely.
constexpr int WIDE_X = 256;
constexpr int WIDE_LD = WIDE_X + 4;
constexpr int WIDE_C = 64;

__global__ __launch_bounds__(512) void ps_canvas_nchw_wide(const float* __restrict__ feats,
                                                           const int* __restrict__ cellmap,
                                                           float* __restrict__ canvas, int ny, int nx) {
  __shared__ __attribute__((aligned(16))) float tile[WIDE_C * WIDE_LD];   // 66,560 B
  __shared__ int vids[WIDE_X];
  __shared__ int occ[WIDE_X];
  __shared__ int nocc;
  const int x0 = blockIdx.x * WIDE_X, y = blockIdx.y, b = blockIdx.z;
  const int t = threadIdx.x;
  if (t == 0) nocc = 0;
  int v = -1;
  if (t < WIDE_X) { v = cellmap[(b * ny + y) * nx + x0 + t]; vids[t] = v; }
  __syncthreads();
  if (v >= 0) occ[atomicAdd(&nocc, 1)] = t;          // order irrelevant: each cell owns its column
  float4* out4 = reinterpret_cast<float4*>(canvas);
  const size_t row4 = static_cast<size_t>(nx) >> 2;
  const size_t obase = (static_cast<size_t>(b) * WIDE_C * ny + y) * row4 + (x0 >> 2);
  const size_t ch_stride4 = static_cast<size_t>(ny) * row4;
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  // zero the tile (16,640 floats = 4160 float4; 512 threads)
  float4* tile4 = reinterpret_cast<float4*>(tile);
  for (int i = t; i < WIDE_C * WIDE_LD / 4; i += 512) tile4[i] = z;
  __syncthreads();
  const int n = nocc;
  if (n == 0) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int i = t + 512 * k;                       // 4096 float4 = 64 ch x 64 quads
      out4[obase + static_cast<size_t>(i >> 6) * ch_stride4 + (i & 63)] = z;
    }
    return;
  }
  // fetch occupied rows: 16 lanes per row (float4 each), 32 rows per pass
  const int q = t & 15;
  for (int r = t >> 4; r < n; r += 32) {
    const int cx = occ[r];
    const float4 f = reinterpret_cast<const float4*>(feats)[static_cast<size_t>(vids[cx]) * (WIDE_C / 4) + q];
    tile[(4 * q + 0) * WIDE_LD + cx] = f.x;
    tile[(4 * q + 1) * WIDE_LD + cx] = f.y;
    tile[(4 * q + 2) * WIDE_LD + cx] = f.z;
    tile[(4 * q + 3) * WIDE_LD + cx] = f.w;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int i = t + 512 * k;
    const int ch = i >> 6, xq = i & 63;
    out4[obase + static_cast<size_t>(ch) * ch_stride4 + xq] =
        *reinterpret_cast<const float4*>(&tile[ch * WIDE_LD + 4 * xq]);
  }
}

__global__ __launch_bounds__(256) void ps_canvas_nhwc(const float* __restrict__ feats,
                                                      const int* __restrict__ cellmap,
                                                      float* __restrict__ canvas, int C, long long ncell) {
  const long long total = ncell * C;
  for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < total;
       i += static_cast<long long>(gridDim.x) * blockDim.x) {
    const long long cell = i / C;
    const int ch = static_cast<int>(i - cell * C);
    const int v = cellmap[cell];
    canvas[i] = v >= 0 ? feats[static_cast<size_t>(v) * C + ch] : 0.f;
  }
}

// C % 4 == 0: one float4 (4 channels of one cell) per lane, 4 independent cells-rows in flight per lane.
// The canvas is one contiguous stream of ncell * C floats written exactly once; a wave covers 1 KB of it
// (64 / (C/4) consecutive cells), the cell ids are read once per lane group, the occupied rows are
// contiguous C*4-byte gathers.
__global__ __launch_bounds__(256) void ps_canvas_nhwc_vec(const float4* __restrict__ feats,
                                                          const int* __restrict__ cellmap,
                                                          float4* __restrict__ canvas, int C4, long long total4) {
  constexpr int U = 4;
  const long long stride = static_cast<long long>(gridDim.x) * blockDim.x;
  for (long long i0 = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i0 < total4; i0 += U * stride) {
    int v[U];
    int q[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long i = i0 + u * stride;
      v[u] = -1;
      q[u] = 0;
      if (i < total4) {
        const long long cell = i / C4;
        q[u] = static_cast<int>(i - cell * C4);
        v[u] = cellmap[cell];
      }
    }
    float4 r[U];
#pragma unroll
    for (int u = 0; u < U; ++u)
      r[u] = v[u] >= 0 ? feats[static_cast<size_t>(v[u]) * C4 + q[u]] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long i = i0 + u * stride;
      if (i < total4) st_nt(canvas + i, r[u]);
    }
  }
}

// backward: grad_feats[v, c] = grad_canvas[b, c, y, x] (rows that lost a duplicate race get 0)
__global__ __launch_bounds__(256) void ps_backward(const float* __restrict__ grad_canvas,
                                                   const int* __restrict__ coors,
                                                   const int* __restrict__ cellmap, int m, int C, int B,
                                                   int ny, int nx, int channels_last,
                                                   float* __restrict__ grad_feats) {
  const long long total = static_cast<long long>(m) * C;
  for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < total;
       i += static_cast<long long>(gridDim.x) * blockDim.x) {
    const int v = static_cast<int>(i / C);
    const int ch = static_cast<int>(i - static_cast<long long>(v) * C);
    const int b = coors[v * 4 + 0], y = coors[v * 4 + 2], x = coors[v * 4 + 3];
    float g = 0.f;
    if (b >= 0 && b < B && y >= 0 && y < ny && x >= 0 && x < nx &&
        cellmap[(b * ny + y) * nx + x] == v) {
      g = channels_last ? grad_canvas[((static_cast<size_t>(b) * ny + y) * nx + x) * C + ch]
                        : grad_canvas[((static_cast<size_t>(b) * C + ch) * ny + y) * nx + x];
    }
    grad_feats[i] = g;
  }
}

}  // namespace

namespace dbev {
// canvas[b, c, y, x] (or NHWC) = feats[cellmap[b, y, x], c] or 0 -- every element written once
int launch_canvas(const float* voxel_features, const int* cellmap, float* canvas, int C, int B, int ny, int nx,
                  int channels_last, hipStream_t s) {
  const long long ncell = static_cast<long long>(B) * ny * nx;
  if (channels_last && (C & 3) == 0) {
    const long long total4 = ncell * (C >> 2);
    long long blocks = (total4 + 256 * 4 - 1) / (256 * 4);
    if (blocks > DBEV_MAX_GRID * 8) blocks = DBEV_MAX_GRID * 8;
    hipLaunchKernelGGL(ps_canvas_nhwc_vec, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, s,
                       reinterpret_cast<const float4*>(voxel_features), cellmap, reinterpret_cast<float4*>(canvas),
                       C >> 2, total4);
  } else if (channels_last) {
    long long blocks = (ncell * C + 255) / 256;
    if (blocks > DBEV_MAX_GRID * 8) blocks = DBEV_MAX_GRID * 8;
    hipLaunchKernelGGL(ps_canvas_nhwc, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, s, voxel_features,
                       cellmap, canvas, C, ncell);
  } else if (nx % WIDE_X == 0 && C == WIDE_C) {
    hipLaunchKernelGGL(ps_canvas_nchw_wide, dim3(nx / WIDE_X, ny, B), dim3(512), 0, s, voxel_features, cellmap,
                       canvas, ny, nx);
  } else {
    const size_t lds = sizeof(float) * TILE_X * (C + 1);
    if (lds > 160 * 1024) return DBEV_EINVAL;
    hipLaunchKernelGGL(ps_canvas_nchw, dim3(dbev_ceil_div(nx, TILE_X), ny, B), dim3(256), lds, s,
                       voxel_features, cellmap, canvas, C, ny, nx);
  }
  DBEV_LAUNCH_CHECK();
  return 0;
}
}  // namespace dbev

extern "C" int dbev_pillars_scatter(const float* voxel_features, const int32_t* coors, int num_voxels,
                                    int C, int B, int ny, int nx, float* canvas, int channels_last,
                                    int32_t* cellmap, dbevStream_t stream) {
  if (num_voxels < 0 || C <= 0 || B <= 0 || ny <= 0 || nx <= 0 || cellmap == nullptr) return DBEV_EINVAL;
  hipStream_t s = dbev_stream(stream);
  const long long ncell = static_cast<long long>(B) * ny * nx;
  DBEV_HIP_TRY(hipMemsetAsync(cellmap, 0xff, sizeof(int) * ncell, s));  // -1
  if (num_voxels > 0)
    hipLaunchKernelGGL(ps_cellmap, dim3(dbev_ceil_div(num_voxels, 256)), dim3(256), 0, s, coors, num_voxels,
                       B, ny, nx, cellmap);
  return dbev::launch_canvas(voxel_features, cellmap, canvas, C, B, ny, nx, channels_last, s);
}

extern "C" int dbev_pillars_scatter_backward(const float* grad_canvas, const int32_t* coors,
                                             const int32_t* cellmap, int num_voxels, int C, int B, int ny,
                                             int nx, int channels_last, float* grad_feats,
                                             dbevStream_t stream) {
  if (num_voxels < 0 || C <= 0 || B <= 0 || ny <= 0 || nx <= 0) return DBEV_EINVAL;
  if (num_voxels == 0) return 0;
  long long blocks = (static_cast<long long>(num_voxels) * C + 255) / 256;
  if (blocks > DBEV_MAX_GRID * 4) blocks = DBEV_MAX_GRID * 4;
  hipLaunchKernelGGL(ps_backward, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, dbev_stream(stream),
                     grad_canvas, coors, cellmap, num_voxels, C, B, ny, nx, channels_last, grad_feats);
  DBEV_LAUNCH_CHECK();
  return 0;
}
