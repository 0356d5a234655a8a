// Fused Lift-Splat for gfx950 -- the student's view transform without the 64 MB/frame
// "volume" tensor.
//
// Reference sequence (bevdet_distill_more.py:411-421 + view_transformer_mine.py:141-181):
//   volume = depth[:,None] * feat[:,:,None]            -> f32[B,N,D,H,W,C] materialised
//   voxel_pooling: trunc-index, mask, rank, argsort, 3 gathers, cumsum, diff, index_put
// i.e. >= 6 passes over a 57-64 MB tensor per 6-camera frame plus an int64 sort.
//
// Here (algorithmic bytes 6.27 MB/frame fwd):
//   prepare : geom -> voxel index of every frustum point (bit-exact: fp32 sub, IEEE div,
//             trunc toward zero, vt_mine.py:150), int histogram over the BEV cells,
//             exclusive scan, fill, per-cell sort by point id  => CSR cell -> points.
//             Depends on camera geometry only; shared by forward and backward.
//   forward : one wavefront per BEV cell.  A feature row is C/4 float4 lanes (16 at C=64)
//             so each wave instruction gathers 4 points' rows (channels-last feat, 256 B
//             each, L2 resident) and FMAs them with the point's depth probability.
//             Every cell is written (zeros for empty ones): no memset, no atomics,
//             fixed summation order.
//   backward: pixel-stationary.  C/4 lanes own one (camera, h, w) pixel, keep its feature
//             row and its grad_feat accumulator in registers and walk the D depth bins:
//             gather the cell's grad row once, use it for grad_feat += depth * g and for
//             grad_depth = <g, feat> (lane-group shuffle reduce).  No atomics.
// The same CSR also serves voxel_pooling(geom, volume) for callers that do hold a volume.
#include "prims.h"

namespace {

struct GridParams {
  float lo[3];   // bx - dx/2, computed in fp32 on the host exactly like the reference tensor op
  float dx[3];
  int nx[3];     // X, Y, Z
};

size_t align_up(size_t x) { return (x + 255) & ~static_cast<size_t>(255); }

struct LsLayout { size_t count, list, scanws, sortws, total; };

LsLayout ls_layout(long long np, long long ncell) {
  LsLayout L;
  size_t o = 0;
  L.count = o;  o += align_up(sizeof(int) * ncell);
  L.list = o;   o += align_up(sizeof(int) * np);
  L.scanws = o; o += align_up(sizeof(int) * dbev::scan_workspace_ints(ncell));
  L.sortws = o; o += align_up(sizeof(int) * dbev::segment_sort_workspace_ints(np));
  L.total = o;
  return L;
}

// ---- block-aggregated histogram / fill --------------------------------------------------------
// Neighbouring frustum points (same camera, same depth bin, adjacent pixels) mostly share a BEV cell:
// 1024 consecutive points of a near depth bin land in ~20-300 distinct cells.  Each workgroup therefore
// counts its 1024 points in an LDS hash table (cell -> count, LDS atomics) and issues ONE global atomic
// per distinct cell instead of one per point (measured: 112 + 143 us -> see DESIGN.md for the pair).
constexpr int AGG_PTS = 1024;            // points per workgroup (256 threads x 4)
constexpr int AGG_SLOTS = 2048;          // open-addressing table, load factor <= 0.5
constexpr int AGG_EMPTY = -1;

__device__ __forceinline__ int agg_insert(int* keys, int cell) {
  unsigned h = (static_cast<unsigned>(cell) * 2654435761u) >> 21;   // 11 bits
  for (;;) {
    const int prev = atomicCAS(&keys[h], AGG_EMPTY, cell);
    if (prev == AGG_EMPTY || prev == cell) return static_cast<int>(h);
    h = (h + 1) & (AGG_SLOTS - 1);
  }
}

struct CamParams { const float* cam; const float* frustum; int DHW; };   // cam: [BN][24] = A(9) pt(3) C(9) t(3)

// MODE 0: geom tensor given.  MODE 1: ego-frame point computed in-kernel from the camera matrices with exactly
// the multiply-add order of lss._apply3x3 (no FMA) -> same bits as the torch geometry.  MODE 2: integer voxel
// coordinates (x, y, z, b) of the bev_pool surface (bev_pool.py:83-97): cell = ((b*D + z)*H + x)*W + y with
// H = G.nx[0], W = G.nx[1], D = G.nx[2], B = pts_per_batch (reused); `geom` then points at int32[np, 4].
template <int MODE>
__global__ __launch_bounds__(256) void ls_cell_count_agg(const float* __restrict__ geom, CamParams cp, int np,
                                                         int pts_per_batch, GridParams G,
                                                         int* __restrict__ point_cell,
                                                         int* __restrict__ count) {
#pragma clang fp contract(off)
  __shared__ int keys[AGG_SLOTS];
  __shared__ int cnt[AGG_SLOTS];
  for (int i = threadIdx.x; i < AGG_SLOTS; i += 256) { keys[i] = AGG_EMPTY; cnt[i] = 0; }
  __syncthreads();
  const int base = blockIdx.x * AGG_PTS;
#pragma unroll
  for (int k = 0; k < AGG_PTS / 256; ++k) {
    const int p = base + k * 256 + threadIdx.x;
    if (p >= np) continue;
    if (MODE == 2 || MODE == 3) {
      int4 c;
      if (MODE == 2) {
        c = reinterpret_cast<const int4*>(geom)[p];
      } else {                       // int64 coordinates as the reference's Python surface passes them (bev_pool.py:83-97):
        const longlong2 a = reinterpret_cast<const longlong2*>(geom)[2 * static_cast<size_t>(p)];       // no int32 copy pass
        const longlong2 b = reinterpret_cast<const longlong2*>(geom)[2 * static_cast<size_t>(p) + 1];
        const long long lim = 0x7fffffffLL;
        c.x = (a.x < 0 || a.x > lim) ? -1 : static_cast<int>(a.x);
        c.y = (a.y < 0 || a.y > lim) ? -1 : static_cast<int>(a.y);
        c.z = (b.x < 0 || b.x > lim) ? -1 : static_cast<int>(b.x);
        c.w = (b.y < 0 || b.y > lim) ? -1 : static_cast<int>(b.y);
      }
      int lin = -1;
      if (c.x >= 0 && c.x < G.nx[0] && c.y >= 0 && c.y < G.nx[1] && c.z >= 0 && c.z < G.nx[2] && c.w >= 0 &&
          c.w < pts_per_batch) {
        lin = ((c.w * G.nx[2] + c.z) * G.nx[0] + c.x) * G.nx[1] + c.y;
        atomicAdd(&cnt[agg_insert(keys, lin)], 1);
      }
      point_cell[p] = lin;
      continue;
    }
    float gx, gy, gz;
    if (MODE == 1) {
      const int bn = p / cp.DHW;
      const int r = p - bn * cp.DHW;
      const float* m = cp.cam + static_cast<size_t>(bn) * 24;
      const float* fr = cp.frustum + static_cast<size_t>(r) * 3;
      const float x = fr[0] - m[9], y = fr[1] - m[10], z = fr[2] - m[11];       // frustum - post_trans
      float qx = m[0] * x + m[1] * y + m[2] * z;                                  // inverse(post_rots) @ .
      float qy = m[3] * x + m[4] * y + m[5] * z;
      const float qz = m[6] * x + m[7] * y + m[8] * z;
      qx = qx * qz; qy = qy * qz;                                                 // (u*d, v*d, d)
      gx = (m[12] * qx + m[13] * qy + m[14] * qz) + m[21];                        // rots @ inverse(intrins) @ . + trans
      gy = (m[15] * qx + m[16] * qy + m[17] * qz) + m[22];
      gz = (m[18] * qx + m[19] * qy + m[20] * qz) + m[23];
    } else {
      const float* g = geom + static_cast<size_t>(p) * 3;
      gx = g[0]; gy = g[1]; gz = g[2];
    }
    const float gg[3] = {gx, gy, gz};
    int idx[3];
    bool ok = true;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float t = truncf((gg[a] - G.lo[a]) / G.dx[a]);
      ok = ok && (t >= 0.f) && (t < static_cast<float>(G.nx[a]));
      idx[a] = static_cast<int>(t);
    }
    int lin = -1;
    if (ok) {
      const int b = p / pts_per_batch;
      lin = ((b * G.nx[1] + idx[1]) * G.nx[0] + idx[0]) * G.nx[2] + idx[2];
      atomicAdd(&cnt[agg_insert(keys, lin)], 1);
    }
    point_cell[p] = lin;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < AGG_SLOTS; i += 256)
    if (keys[i] != AGG_EMPTY) atomicAdd(&count[keys[i]], cnt[i]);
}

__global__ __launch_bounds__(256) void ls_fill_agg(const int* __restrict__ point_cell, int np,
                                                   const int* __restrict__ cell_start,
                                                   int* __restrict__ count, unsigned* __restrict__ list) {
  __shared__ int keys[AGG_SLOTS];
  __shared__ int cnt[AGG_SLOTS];      // local count, then the range base reserved in the cell's list
  for (int i = threadIdx.x; i < AGG_SLOTS; i += 256) { keys[i] = AGG_EMPTY; cnt[i] = 0; }
  __syncthreads();
  const int base = blockIdx.x * AGG_PTS;
  int slot[AGG_PTS / 256], rank[AGG_PTS / 256], cellv[AGG_PTS / 256];
#pragma unroll
  for (int k = 0; k < AGG_PTS / 256; ++k) {
    const int p = base + k * 256 + threadIdx.x;
    cellv[k] = p < np ? point_cell[p] : -1;
    slot[k] = -1;
    if (cellv[k] >= 0) {
      slot[k] = agg_insert(keys, cellv[k]);
      rank[k] = atomicAdd(&cnt[slot[k]], 1);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < AGG_SLOTS; i += 256)
    if (keys[i] != AGG_EMPTY) {
      const int c = cnt[i];
      cnt[i] = cell_start[keys[i]] + (atomicSub(&count[keys[i]], c) - c);   // reserve [base, base + c)
    }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < AGG_PTS / 256; ++k)
    if (slot[k] >= 0) list[cnt[slot[k]] + rank[k]] = static_cast<unsigned>(base + k * 256 + threadIdx.x);
}

constexpr int HOT_CELL_POINTS = 128;   // cells with more points get a whole workgroup (see ls_forward_hot)

__device__ __forceinline__ void fma4(float4& a, float s, const float4& v) {
  a.x = fmaf(s, v.x, a.x); a.y = fmaf(s, v.y, a.y); a.z = fmaf(s, v.z, a.z); a.w = fmaf(s, v.w, a.w);
}

// LIFT=true : row(p) = depth[p] * feat[bn(p), hw(p), :]      (fused lift)
// LIFT=false: row(p) = x[p, :]                               (volume already materialised)
template <bool LIFT, int UNROLL, bool SKIP_HOT>
__global__ __launch_bounds__(256) void ls_forward(const float* __restrict__ depth,
                                                  const float4* __restrict__ rows,
                                                  const int* __restrict__ cell_start,
                                                  const unsigned* __restrict__ cell_points,
                                                  float4* __restrict__ out, int n_cells, int c4, int rpw,
                                                  int HW, int DHW) {
  const int cell = (xcd_block() * blockDim.x + threadIdx.x) >> 6;   // grid is a multiple of 8 blocks
  const int lane = threadIdx.x & 63;
  if (cell >= n_cells) return;
  const int st = cell_start[cell];
  const int L = cell_start[cell + 1] - st;
  if (SKIP_HOT && L > HOT_CELL_POINTS) return;   // written by ls_forward_hot
  const int sub = lane / c4;
  const int q = lane - sub * c4;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (sub < rpw && L > 0) {
    const unsigned* pl = cell_points + st;
    // every batch issues UNROLL predicated (index -> depth, row) gathers back to back: a cell of
    // <= UNROLL*rpw points (78 % of the occupied cells at UNROLL=4) is ONE round trip deep instead
    // of a serial tail of dependent loads
    for (int j = sub; j < L; j += UNROLL * rpw) {
      float4 v[UNROLL];
      float s[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        const int jj = j + u * rpw;
        const bool ok = jj < L;
        const unsigned p = ok ? pl[jj] : pl[0];
        v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        s[u] = 0.f;
        if (ok) {
          if (LIFT) {
            const unsigned bn = p / DHW;
            const unsigned hw = p % HW;
            s[u] = depth[p];
            v[u] = rows[(static_cast<size_t>(bn) * HW + hw) * c4 + q];
          } else {
            s[u] = 1.f;
            v[u] = rows[static_cast<size_t>(p) * c4 + q];
          }
        }
      }
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        if (LIFT) fma4(acc, s[u], v[u]);
        else { acc.x += s[u] * v[u].x; acc.y += s[u] * v[u].y; acc.z += s[u] * v[u].z; acc.w += s[u] * v[u].w; }
      }
    }
  }
  float4 tot = acc;
  for (int s2 = 1; s2 < rpw; ++s2) {
    const int src = q + s2 * c4;
    tot.x += __shfl(acc.x, src);
    tot.y += __shfl(acc.y, src);
    tot.z += __shfl(acc.z, src);
    tot.w += __shfl(acc.w, src);
  }
  if (sub == 0) out[static_cast<size_t>(cell) * c4 + q] = tot;  // zeros for an empty cell
}

// C/4 == 16 (C = 64): one 16-lane group per cell, 4 consecutive cells per wave.  The group pulls 16
// point ids per (prefetched) coalesced load, the owning lane fetches the point's depth and computes its
// row offset once, and the rows are gathered 8 at a time.  Compared with ls_forward this quarters the
// number of waves (the kernel is bound by dependent-load round trips x wave generations, not by
// bytes) and takes the id -> depth -> row chain off the per-row critical path.  Each lane adds its
// cell's points in list order (ascending point id) with one fma chain: the sequential sum of the
// reference's bev_pool kernel (bev_pool_cuda.cu:33-40), run-to-run bit identical.
template <bool LIFT>
__global__ __launch_bounds__(256) void ls_forward_c64(const float* __restrict__ depth,
                                                      const float4* __restrict__ rows,
                                                      const int* __restrict__ cell_start,
                                                      const unsigned* __restrict__ cell_points,
                                                      float4* __restrict__ out, int n_cells, int HW,
                                                      int DHW) {
  constexpr int C4 = 16, DU = 8;
  const int wave = (xcd_block() * blockDim.x + threadIdx.x) >> 6;   // grid is a multiple of 8 blocks
  const int lane = threadIdx.x & 63;
  const int sub = lane >> 4, q = lane & 15, g0 = lane & 48;
  const int cell = wave * 4 + sub;
  if (cell >= n_cells) return;                       // whole groups leave together
  const int st = cell_start[cell];
  const int L = cell_start[cell + 1] - st;
  if (L > HOT_CELL_POINTS) return;                   // written by ls_forward_hot
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  const unsigned* pl = cell_points + st;
  unsigned idx = q < L ? pl[q] : 0u;
  for (int j0 = 0; j0 < L; j0 += 16) {
    const int nb = min(16, L - j0);
    float myd = 0.f;
    unsigned myrow = 0u;
    if (q < nb) {
      if (LIFT) {
        myd = depth[idx];
        myrow = (idx / static_cast<unsigned>(DHW)) * HW + idx % static_cast<unsigned>(HW);
      } else {
        myd = 1.f;
        myrow = idx;
      }
    }
    if (j0 + 16 + q < L) idx = pl[j0 + 16 + q];      // next id block is in flight during the gathers
#pragma unroll
    for (int h = 0; h < 16; h += DU) {
      if (h < nb) {                                  // uniform inside the group
        float4 v[DU];
#pragma unroll
        for (int u = 0; u < DU; ++u) {
          const unsigned r = __shfl(myrow, g0 | (h + u));
          v[u] = (h + u) < nb ? rows[static_cast<size_t>(r) * C4 + q] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < DU; ++u) fma4(acc, __shfl(myd, g0 | (h + u)), v[u]);   // padding: 0 * 0
      }
    }
  }
  out[static_cast<size_t>(cell) * C4 + q] = acc;     // zeros for an empty cell
}

// cells whose list is longer than HOT_CELL_POINTS (0.6 % of the cells, 6 % of the points of a
// 6-camera frame; up to ~640 points next to the cameras).  Order of the list is irrelevant.
__global__ __launch_bounds__(256) void ls_hot_cells(const int* __restrict__ cell_start, int n_cells,
                                                    int* __restrict__ hot_cells, int* __restrict__ n_hot) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n_cells) return;
  if (cell_start[c + 1] - cell_start[c] > HOT_CELL_POINTS) hot_cells[atomicAdd(n_hot, 1)] = c;
}

// one 256-thread workgroup per hot cell: each of the 4 waves walks a contiguous quarter of the
// cell's point list (rows in flight: 4 waves x 4 rows x UNROLL), partials meet in LDS and are
// added in a fixed order -> same determinism as the one-wave path, 4x shorter critical path.
template <bool LIFT, int UNROLL>
__global__ __launch_bounds__(256) void ls_forward_hot(const float* __restrict__ depth,
                                                      const float4* __restrict__ rows,
                                                      const int* __restrict__ cell_start,
                                                      const unsigned* __restrict__ cell_points,
                                                      const int* __restrict__ hot_cells,
                                                      const int* __restrict__ n_hot,
                                                      float4* __restrict__ out, int c4, int rpw, int HW,
                                                      int DHW) {
  __shared__ float4 part[4][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int sub = lane / c4;
  const int q = lane - sub * c4;
  const int nh = *n_hot;
  for (int h = blockIdx.x; h < nh; h += gridDim.x) {
    const int cell = hot_cells[h];
    const int st = cell_start[cell];
    const int L = cell_start[cell + 1] - st;
    const int per = (L + 3) >> 2;
    const int j0 = wv * per, j1 = min(L, j0 + per);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (sub < rpw) {
      const unsigned* pl = cell_points + st;
      int j = j0 + sub;
      for (; j + (UNROLL - 1) * rpw < j1; j += UNROLL * rpw) {
        float4 v[UNROLL];
        float s[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
          const unsigned p = pl[j + u * rpw];
          if (LIFT) {
            s[u] = depth[p];
            v[u] = rows[(static_cast<size_t>(p / DHW) * HW + p % HW) * c4 + q];
          } else {
            s[u] = 1.f;
            v[u] = rows[static_cast<size_t>(p) * c4 + q];
          }
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) fma4(acc, s[u], v[u]);
      }
      for (; j < j1; j += rpw) {
        const unsigned p = pl[j];
        if (LIFT) fma4(acc, depth[p], rows[(static_cast<size_t>(p / DHW) * HW + p % HW) * c4 + q]);
        else fma4(acc, 1.f, rows[static_cast<size_t>(p) * c4 + q]);
      }
    }
    __syncthreads();            // previous iteration's readers are done
    part[wv][lane] = acc;
    __syncthreads();
    if (threadIdx.x < c4) {
      float4 tot = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int w2 = 0; w2 < 4; ++w2)
        for (int s2 = 0; s2 < rpw; ++s2) {
          const float4 a = part[w2][s2 * c4 + threadIdx.x];
          tot.x += a.x; tot.y += a.y; tot.z += a.z; tot.w += a.w;
        }
      out[static_cast<size_t>(cell) * c4 + threadIdx.x] = tot;
    }
  }
}

// pixel-stationary backward of the fused lift-splat.  one lane group (c4 lanes) per pixel.
__global__ __launch_bounds__(256) void ls_backward(const float4* __restrict__ grad_out,
                                                   const float* __restrict__ depth,
                                                   const float4* __restrict__ feat,
                                                   const int* __restrict__ point_cell,
                                                   float* __restrict__ grad_depth,
                                                   float4* __restrict__ grad_feat, int n_pix, int c4,
                                                   int rpw, int D, int HW, bool pow2) {
  const int wave = (xcd_block() * blockDim.x + threadIdx.x) >> 6;   // grid is a multiple of 8 blocks
  const int lane = threadIdx.x & 63;
  const int sub = lane / c4;
  const int q = lane - sub * c4;
  const int pix = wave * rpw + sub;           // = bn*HW + hw
  const bool active = sub < rpw && pix < n_pix;
  const int bn = active ? pix / HW : 0;
  const int hw = active ? pix - bn * HW : 0;
  const float4 f = active ? feat[static_cast<size_t>(pix) * c4 + q] : make_float4(0.f, 0.f, 0.f, 0.f);
  float4 gf = make_float4(0.f, 0.f, 0.f, 0.f);
  const size_t pbase = static_cast<size_t>(bn) * D * HW + hw;
  constexpr int DU = 4;   // depth bins in flight per lane group (cell id -> grad row gathers are dependent)
  for (int d0 = 0; d0 < D; d0 += DU) {
    int cell[DU];
    float dp[DU];
    float4 g[DU];
#pragma unroll
    for (int u = 0; u < DU; ++u) {
      const int d = d0 + u;
      const size_t p = pbase + static_cast<size_t>(d) * HW;
      cell[u] = (active && d < D) ? point_cell[p] : -1;
      dp[u] = cell[u] >= 0 ? depth[p] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < DU; ++u)
      g[u] = cell[u] >= 0 ? grad_out[static_cast<size_t>(cell[u]) * c4 + q] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int u = 0; u < DU; ++u) {
      const int d = d0 + u;
      fma4(gf, dp[u], g[u]);
      const float part = fmaf(g[u].x, f.x, fmaf(g[u].y, f.y, fmaf(g[u].z, f.z, g[u].w * f.w)));
      float sum = part;
      if (pow2) {
        for (int o = c4 >> 1; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
      } else {
        sum = 0.f;
        for (int k = 0; k < c4; ++k) sum += __shfl(part, sub * c4 + k);
      }
      if (active && q == 0 && d < D) grad_depth[pbase + static_cast<size_t>(d) * HW] = sum;   // 0 for a dropped point
    }
  }
  if (active) grad_feat[static_cast<size_t>(pix) * c4 + q] = gf;
}

// C/4 == 16 (C = 64) and D <= 64: same pixel-stationary walk, but the lane group first fetches the
// cell id + depth of ALL depth bins in one round trip (lane q owns bins q, q+16, q+32, q+48), then
// streams the grad rows 8 bins per batch: ~1 + D/8 dependent round trips per pixel instead of 2*D/4.
// Accumulation order (bins ascending, xor tree over the 16 lanes) is the one of ls_backward ->
// bit-identical results.
__global__ __launch_bounds__(256) void ls_backward_c64(const float4* __restrict__ grad_out,
                                                       const float* __restrict__ depth,
                                                       const float4* __restrict__ feat,
                                                       const int* __restrict__ point_cell,
                                                       float* __restrict__ grad_depth,
                                                       float4* __restrict__ grad_feat, int n_pix, int D,
                                                       int HW) {
  constexpr int C4 = 16, RPW = 4, DU = 8;
  const int wave = (xcd_block() * blockDim.x + threadIdx.x) >> 6;   // grid is a multiple of 8 blocks
  const int lane = threadIdx.x & 63;
  const int sub = lane >> 4;
  const int q = lane & 15;
  const int pix = wave * RPW + sub;           // = bn*HW + hw
  const bool active = pix < n_pix;
  const int bn = active ? pix / HW : 0;
  const int hw = active ? pix - bn * HW : 0;
  const size_t pbase = static_cast<size_t>(bn) * D * HW + hw;
  int mycell[4];
  float mydp[4], mysum[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int d = q + k * C4;
    const bool ok = active && d < D;
    const size_t p = pbase + static_cast<size_t>(d) * HW;
    mycell[k] = ok ? point_cell[p] : -1;
    mydp[k] = ok ? depth[p] : 0.f;
    mysum[k] = 0.f;
  }
  const float4 f = active ? feat[static_cast<size_t>(pix) * C4 + q] : make_float4(0.f, 0.f, 0.f, 0.f);
  float4 gf = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int d0 = 0; d0 < 64; d0 += DU) {
    if (d0 < D) {                              // wave-uniform
      float4 g[DU];
      float dp[DU];
#pragma unroll
      for (int u = 0; u < DU; ++u) {
        const int d = d0 + u, k = d >> 4, src = (sub << 4) | (d & 15);
        const int cell = __shfl(mycell[k], src);
        dp[u] = cell >= 0 ? __shfl(mydp[k], src) : 0.f;
        g[u] = cell >= 0 ? grad_out[static_cast<size_t>(cell) * C4 + q] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < DU; ++u) {
        const int d = d0 + u, k = d >> 4;
        fma4(gf, dp[u], g[u]);
        float sum = fmaf(g[u].x, f.x, fmaf(g[u].y, f.y, fmaf(g[u].z, f.z, g[u].w * f.w)));
        sum += __shfl_xor(sum, 8);
        sum += __shfl_xor(sum, 4);
        sum += __shfl_xor(sum, 2);
        sum += __shfl_xor(sum, 1);
        if (q == (d & 15)) mysum[k] = sum;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int d = q + k * C4;
    if (active && d < D) grad_depth[pbase + static_cast<size_t>(d) * HW] = mysum[k];   // 0 for a dropped point
  }
  if (active) grad_feat[static_cast<size_t>(pix) * C4 + q] = gf;
}

// backward of splat-from-volume: grad_x[p, :] = grad_out[cell(p), :] or 0
__global__ __launch_bounds__(256) void splat_backward(const float4* __restrict__ grad_out,
                                                      const int* __restrict__ point_cell,
                                                      float4* __restrict__ grad_x, int np, int c4,
                                                      int rpw) {
  const int lane = threadIdx.x & 63;
  const int sub = lane / c4;
  const int q = lane - sub * c4;
  if (sub >= rpw) return;
  const long long wave = (static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
  const long long nwaves = (static_cast<long long>(gridDim.x) * blockDim.x) >> 6;
  for (long long p = wave * rpw + sub; p < np; p += nwaves * rpw) {
    const int cell = point_cell[p];
    grad_x[static_cast<size_t>(p) * c4 + q] =
        cell >= 0 ? grad_out[static_cast<size_t>(cell) * c4 + q] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

// C = 64: a wave takes 64 consecutive points -- one coalesced load of their cell ids, then each 16-lane group copies 16
// gradient rows with 8 gathers in flight (the plain kernel above has ONE dependent id -> row chain per group) and
// streaming stores (grad_x is written once, 0.9 GB at the bench shape, and must not evict the 67 MB of grad rows).
__global__ __launch_bounds__(256) void splat_backward_c64(const float4* __restrict__ grad_out,
                                                          const int* __restrict__ point_cell,
                                                          float4* __restrict__ grad_x, int np) {
  constexpr int DU = 8;
  const int lane = threadIdx.x & 63, sub = lane >> 4, q = lane & 15;
  const long long wave = (static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
  const long long nwaves = (static_cast<long long>(gridDim.x) * blockDim.x) >> 6;
  for (long long p0 = wave * 64; p0 < np; p0 += nwaves * 64) {
    const long long pm = p0 + lane;
    const int mycell = pm < np ? point_cell[pm] : -1;
#pragma unroll
    for (int h = 0; h < 16; h += DU) {
      float4 v[DU];
#pragma unroll
      for (int u = 0; u < DU; ++u) {
        const int cell = __shfl(mycell, (h + u) * 4 + sub);          // point p0 + (h+u)*4 + sub
        v[u] = cell >= 0 ? grad_out[static_cast<size_t>(cell) * 16 + q] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < DU; ++u) {
        const long long p = p0 + (h + u) * 4 + sub;
        if (p < np) st_nt(grad_x + static_cast<size_t>(p) * 16 + q, v[u]);
      }
    }
  }
}

bool vec_ok(int C) { return C > 0 && (C & 3) == 0 && (C >> 2) <= 64; }

}  // namespace

extern "C" size_t dbev_lift_splat_workspace_bytes(int n_points, int n_cells) {
  if (n_points < 0 || n_cells <= 0) return 0;
  return ls_layout(n_points, n_cells).total;
}

static int prepare_impl(const float* geom, const float* cam, const float* frustum, int DHW, int n_points,
                        int batch, const float* dx_host, const float* bx_host, const int32_t* nx_host,
                        const void* coords, bool coords_i64,
                        int32_t* point_cell, int32_t* cell_start, int32_t* cell_points, int32_t* n_kept_out,
                        int32_t* hot_cells, int32_t* n_hot_out, void* workspace, size_t workspace_bytes,
                        dbevStream_t stream) {
  if (n_points < 0 || batch <= 0 || (coords == nullptr && n_points % batch != 0)) return DBEV_EINVAL;
  GridParams G;
  long long per = 1;
  for (int k = 0; k < 3; ++k) {
    if (nx_host[k] <= 0 || !(dx_host[k] > 0.f)) return DBEV_EINVAL;
    G.dx[k] = dx_host[k];
    // (self.bx - self.dx / 2.) as fp32 tensor ops (vt_mine.py:150)
    volatile float half = dx_host[k] / 2.0f;
    volatile float lo = bx_host[k] - half;
    G.lo[k] = lo;
    G.nx[k] = nx_host[k];
    per *= nx_host[k];
  }
  const long long ncell = per * batch;
  if (ncell > 0x7fffffffLL) return DBEV_EINVAL;
  const LsLayout L = ls_layout(n_points, ncell);
  if (workspace == nullptr || workspace_bytes < L.total) return DBEV_EINVAL;
  hipStream_t s = dbev_stream(stream);
  char* ws = static_cast<char*>(workspace);
  int* count = reinterpret_cast<int*>(ws + L.count);
  unsigned* list = reinterpret_cast<unsigned*>(ws + L.list);
  int* scanws = reinterpret_cast<int*>(ws + L.scanws);
  DBEV_HIP_TRY(hipMemsetAsync(count, 0, sizeof(int) * ncell, s));
  DBEV_HIP_TRY(hipMemsetAsync(n_hot_out, 0, sizeof(int), s));
  const int nblk = dbev_ceil_div(n_points > 0 ? n_points : 1, AGG_PTS);
  if (n_points > 0) {
    CamParams cp{cam, frustum, DHW > 0 ? DHW : 1};
    if (coords != nullptr && coords_i64)
      hipLaunchKernelGGL(ls_cell_count_agg<3>, dim3(nblk), dim3(256), 0, s, reinterpret_cast<const float*>(coords), cp,
                         n_points, batch, G, point_cell, count);
    else if (coords != nullptr)
      hipLaunchKernelGGL(ls_cell_count_agg<2>, dim3(nblk), dim3(256), 0, s, reinterpret_cast<const float*>(coords), cp,
                         n_points, batch, G, point_cell, count);
    else if (cam != nullptr)
      hipLaunchKernelGGL(ls_cell_count_agg<1>, dim3(nblk), dim3(256), 0, s, nullptr, cp, n_points,
                         n_points / batch, G, point_cell, count);
    else
      hipLaunchKernelGGL(ls_cell_count_agg<0>, dim3(nblk), dim3(256), 0, s, geom, cp, n_points,
                         n_points / batch, G, point_cell, count);
  }
  int rc = dbev::exclusive_scan_i32(count, cell_start, ncell, false, n_kept_out, scanws, s);
  if (rc) return rc;
  if (n_points > 0) {
    hipLaunchKernelGGL(ls_fill_agg, dim3(nblk), dim3(256), 0, s, point_cell, n_points, cell_start, count, list);
    rc = dbev::segment_sort_u32(cell_start, list, reinterpret_cast<unsigned*>(cell_points),
                                static_cast<int>(ncell), reinterpret_cast<int*>(ws + L.sortws), s);
    if (rc) return rc;
    hipLaunchKernelGGL(ls_hot_cells, dim3(dbev_ceil_div(ncell, 256)), dim3(256), 0, s, cell_start,
                       static_cast<int>(ncell), hot_cells, n_hot_out);
  }
  DBEV_LAUNCH_CHECK();
  return 0;
}

extern "C" int dbev_lift_splat_prepare(const float* geom, int n_points, int batch, const float* dx_host,
                                       const float* bx_host, const int32_t* nx_host, int32_t* point_cell,
                                       int32_t* cell_start, int32_t* cell_points, int32_t* n_kept_out,
                                       int32_t* hot_cells, int32_t* n_hot_out, void* workspace,
                                       size_t workspace_bytes, dbevStream_t stream) {
  if (geom == nullptr && n_points > 0) return DBEV_EINVAL;
  return prepare_impl(geom, nullptr, nullptr, 0, n_points, batch, dx_host, bx_host, nx_host, nullptr, false, point_cell,
                      cell_start, cell_points, n_kept_out, hot_cells, n_hot_out, workspace, workspace_bytes, stream);
}

extern "C" int dbev_lift_splat_prepare_cam(const float* cam_params, const float* frustum, int BN, int D, int H,
                                           int W, int batch, const float* dx_host, const float* bx_host,
                                           const int32_t* nx_host, int32_t* point_cell, int32_t* cell_start,
                                           int32_t* cell_points, int32_t* n_kept_out, int32_t* hot_cells,
                                           int32_t* n_hot_out, void* workspace, size_t workspace_bytes,
                                           dbevStream_t stream) {
  if (cam_params == nullptr || frustum == nullptr || BN <= 0 || D <= 0 || H <= 0 || W <= 0) return DBEV_EINVAL;
  const long long np = static_cast<long long>(BN) * D * H * W;
  if (np > 0x7fffffffLL) return DBEV_EINVAL;
  return prepare_impl(nullptr, cam_params, frustum, D * H * W, static_cast<int>(np), batch, dx_host, bx_host,
                      nx_host, nullptr, false, point_cell, cell_start, cell_points, n_kept_out, hot_cells, n_hot_out, workspace,
                      workspace_bytes, stream);
}

extern "C" int dbev_bev_pool_prepare(const int32_t* coords, int n_points, int B, int D, int H, int W,
                                     int32_t* point_cell, int32_t* cell_start, int32_t* cell_points,
                                     int32_t* n_kept_out, int32_t* hot_cells, int32_t* n_hot_out, void* workspace,
                                     size_t workspace_bytes, dbevStream_t stream) {
  if ((coords == nullptr && n_points > 0) || B <= 0 || D <= 0 || H <= 0 || W <= 0) return DBEV_EINVAL;
  const float one[3] = {1.f, 1.f, 1.f}, zero[3] = {0.f, 0.f, 0.f};
  const int32_t nx[3] = {H, W, D};          // (x, y, z) extents of the coords: 0<=x<H, 0<=y<W, 0<=z<D
  static const int32_t dummy[4] = {0, 0, 0, 0};
  return prepare_impl(nullptr, nullptr, nullptr, 0, n_points, B, one, zero, nx, coords != nullptr ? coords : dummy, false,
                      point_cell, cell_start, cell_points, n_kept_out, hot_cells, n_hot_out, workspace, workspace_bytes,
                      stream);
}

extern "C" int dbev_bev_pool_prepare_i64(const long long* coords, int n_points, int B, int D, int H, int W,
                                         int32_t* point_cell, int32_t* cell_start, int32_t* cell_points,
                                         int32_t* n_kept_out, int32_t* hot_cells, int32_t* n_hot_out, void* workspace,
                                         size_t workspace_bytes, dbevStream_t stream) {
  if ((coords == nullptr && n_points > 0) || B <= 0 || D <= 0 || H <= 0 || W <= 0) return DBEV_EINVAL;
  const float one[3] = {1.f, 1.f, 1.f}, zero[3] = {0.f, 0.f, 0.f};
  const int32_t nx[3] = {H, W, D};
  static const long long dummy[4] = {0, 0, 0, 0};
  return prepare_impl(nullptr, nullptr, nullptr, 0, n_points, B, one, zero, nx,
                      coords != nullptr ? static_cast<const void*>(coords) : static_cast<const void*>(dummy), true,
                      point_cell, cell_start, cell_points, n_kept_out, hot_cells, n_hot_out, workspace, workspace_bytes,
                      stream);
}

static int hot_grid(int n_cells) { return n_cells < 2048 ? (n_cells < 1 ? 1 : n_cells) : 2048; }

extern "C" int dbev_lift_splat_forward(const float* depth, const float* feat_nhwc,
                                       const int32_t* cell_start, const int32_t* cell_points,
                                       const int32_t* hot_cells, const int32_t* n_hot, float* out,
                                       int BN, int D, int H, int W, int C, int n_cells,
                                       dbevStream_t stream) {
  if (BN <= 0 || D <= 0 || H <= 0 || W <= 0 || n_cells <= 0 || !vec_ok(C)) return DBEV_EINVAL;
  const int c4 = C >> 2, rpw = 64 / c4;
  hipStream_t s = dbev_stream(stream);
  if (c4 == 16)
    hipLaunchKernelGGL((ls_forward_c64<true>), dim3(dbev_round_xcd(dbev_ceil_div(n_cells, 16))), dim3(256), 0, s,
                       depth, reinterpret_cast<const float4*>(feat_nhwc), cell_start,
                       reinterpret_cast<const unsigned*>(cell_points), reinterpret_cast<float4*>(out), n_cells,
                       H * W, D * H * W);
  else
  hipLaunchKernelGGL((ls_forward<true, 4, true>), dim3(dbev_round_xcd(dbev_ceil_div(n_cells, 4))), dim3(256), 0, s, depth,
                     reinterpret_cast<const float4*>(feat_nhwc), cell_start,
                     reinterpret_cast<const unsigned*>(cell_points), reinterpret_cast<float4*>(out),
                     n_cells, c4, rpw, H * W, D * H * W);
  hipLaunchKernelGGL((ls_forward_hot<true, 4>), dim3(hot_grid(n_cells)), dim3(256), 0, s, depth,
                     reinterpret_cast<const float4*>(feat_nhwc), cell_start,
                     reinterpret_cast<const unsigned*>(cell_points), hot_cells, n_hot,
                     reinterpret_cast<float4*>(out), c4, rpw, H * W, D * H * W);
  DBEV_LAUNCH_CHECK();
  return 0;
}

extern "C" int dbev_lift_splat_backward(const float* grad_out, const float* depth, const float* feat_nhwc,
                                        const int32_t* point_cell, float* grad_depth,
                                        float* grad_feat_nhwc, int BN, int D, int H, int W, int C,
                                        dbevStream_t stream) {
  if (BN <= 0 || D <= 0 || H <= 0 || W <= 0 || !vec_ok(C)) return DBEV_EINVAL;
  const int c4 = C >> 2, rpw = 64 / c4;
  const int n_pix = BN * H * W;
  const int waves = dbev_ceil_div(n_pix, rpw);
  if (c4 == 16 && D <= 64) {
    hipLaunchKernelGGL(ls_backward_c64, dim3(dbev_round_xcd(dbev_ceil_div(waves, 4))), dim3(256), 0,
                       dbev_stream(stream), reinterpret_cast<const float4*>(grad_out), depth,
                       reinterpret_cast<const float4*>(feat_nhwc), point_cell, grad_depth,
                       reinterpret_cast<float4*>(grad_feat_nhwc), n_pix, D, H * W);
    DBEV_LAUNCH_CHECK();
    return 0;
  }
  hipLaunchKernelGGL(ls_backward, dim3(dbev_round_xcd(dbev_ceil_div(waves, 4))), dim3(256), 0, dbev_stream(stream),
                     reinterpret_cast<const float4*>(grad_out), depth,
                     reinterpret_cast<const float4*>(feat_nhwc), point_cell, grad_depth,
                     reinterpret_cast<float4*>(grad_feat_nhwc), n_pix, c4, rpw, D, H * W,
                     (c4 & (c4 - 1)) == 0);
  DBEV_LAUNCH_CHECK();
  return 0;
}

extern "C" int dbev_splat_forward(const float* x, const int32_t* cell_start, const int32_t* cell_points,
                                  const int32_t* hot_cells, const int32_t* n_hot, float* out, int n_points,
                                  int C, int n_cells, dbevStream_t stream) {
  if (n_points < 0 || n_cells <= 0 || !vec_ok(C)) return DBEV_EINVAL;
  const int c4 = C >> 2, rpw = 64 / c4;
  hipStream_t s = dbev_stream(stream);
  if (c4 == 16)
    hipLaunchKernelGGL((ls_forward_c64<false>), dim3(dbev_round_xcd(dbev_ceil_div(n_cells, 16))), dim3(256), 0, s,
                       nullptr, reinterpret_cast<const float4*>(x), cell_start,
                       reinterpret_cast<const unsigned*>(cell_points), reinterpret_cast<float4*>(out), n_cells,
                       1, 1);
  else
  hipLaunchKernelGGL((ls_forward<false, 4, true>), dim3(dbev_round_xcd(dbev_ceil_div(n_cells, 4))), dim3(256), 0, s, nullptr,
                     reinterpret_cast<const float4*>(x), cell_start,
                     reinterpret_cast<const unsigned*>(cell_points), reinterpret_cast<float4*>(out),
                     n_cells, c4, rpw, 1, 1);
  hipLaunchKernelGGL((ls_forward_hot<false, 8>), dim3(hot_grid(n_cells)), dim3(256), 0, s, nullptr,
                     reinterpret_cast<const float4*>(x), cell_start,
                     reinterpret_cast<const unsigned*>(cell_points), hot_cells, n_hot,
                     reinterpret_cast<float4*>(out), c4, rpw, 1, 1);
  DBEV_LAUNCH_CHECK();
  return 0;
}

extern "C" int dbev_splat_backward(const float* grad_out, const int32_t* point_cell, float* grad_x,
                                   int n_points, int C, dbevStream_t stream) {
  if (n_points < 0 || !vec_ok(C)) return DBEV_EINVAL;
  if (n_points == 0) return 0;
  const int c4 = C >> 2, rpw = 64 / c4;
  if (c4 == 16) {
    long long b64 = (static_cast<long long>(n_points) + 255) / 256;
    if (b64 > DBEV_MAX_GRID * 4) b64 = DBEV_MAX_GRID * 4;
    hipLaunchKernelGGL(splat_backward_c64, dim3(static_cast<unsigned>(b64)), dim3(256), 0, dbev_stream(stream),
                       reinterpret_cast<const float4*>(grad_out), point_cell, reinterpret_cast<float4*>(grad_x), n_points);
    DBEV_LAUNCH_CHECK();
    return 0;
  }
  long long blocks = ((static_cast<long long>(n_points) + rpw - 1) / rpw + 3) / 4;
  if (blocks > DBEV_MAX_GRID * 4) blocks = DBEV_MAX_GRID * 4;
  hipLaunchKernelGGL(splat_backward, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, dbev_stream(stream),
                     reinterpret_cast<const float4*>(grad_out), point_cell,
                     reinterpret_cast<float4*>(grad_x), n_points, c4, rpw);
  DBEV_LAUNCH_CHECK();
  return 0;
}
