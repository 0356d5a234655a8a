/*
 * oracle/voxel.c -- plain-C CPU restatement of the reference's voxel ops.
 * TEST INFRASTRUCTURE ONLY (checker for the HIP path; never shipped, never measured
 * except as bench.py's cpu_baseline "port").
 *
 * Each function cites the reference source it follows (paths relative to the
 * reference root).  dynamic_voxelize / hard_voxelize are pinned bit-exactly against the
 * reference's own voxelization_cpu.cpp compiled into oracle/_ref (tests/golden/voxel_*.npz);
 * dynamic_scatter has NO CPU implementation in the reference
 * (mmdet3d/ops/voxel/src/voxelization.h:118 "do not support cpu yet"), so it restates the
 * CUDA host code and is pinned by construction + brute-force property tests.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* mmdet3d/ops/voxel/src/voxelization_cpu.cpp:157-160 */
static void grid_size_of(const float* voxel_size, const float* coors_range, int ndim, int* grid) {
  for (int i = 0; i < ndim; ++i)
    grid[i] = (int)round((coors_range[ndim + i] - coors_range[i]) / voxel_size[i]);
}

/* mmdet3d/ops/voxel/src/voxelization_cpu.cpp:7-43 (dynamic_voxelize_kernel).
 * coors[i] = (z, y, x) of the cell of point i, or (-1,-1,-1) if any axis is out of range.
 * fp32 subtract, fp32 divide, floor -- exactly the reference expression. */
void oracle_dynamic_voxelize(const float* points, int32_t* coors, int num_points, int num_features,
                             const float* voxel_size, const float* coors_range, int ndim) {
  int grid[3];
  grid_size_of(voxel_size, coors_range, ndim, grid);
  int coor[3];
  for (int i = 0; i < num_points; ++i) {
    int failed = 0;
    for (int j = 0; j < ndim; ++j) {
      volatile float q = (points[(size_t)i * num_features + j] - coors_range[j]) / voxel_size[j];
      int c = (int)floor(q);
      if (c < 0 || c >= grid[j]) { failed = 1; break; }
      coor[ndim - 1 - j] = c;
    }
    for (int k = 0; k < ndim; ++k) coors[(size_t)i * ndim + k] = failed ? -1 : coor[k];
  }
}

/* mmdet3d/ops/voxel/src/voxelization_cpu.cpp:45-101 (hard_voxelize_kernel) + :107-144.
 * Sequential first-come semantics.  voxels[max_voxels,max_points,F], coors[max_voxels,3],
 * num_points_per_voxel[max_voxels] are caller-zeroed (voxelize.py:57-62).  Returns voxel_num. */
int oracle_hard_voxelize(const float* points, float* voxels, int32_t* coors,
                         int32_t* num_points_per_voxel, int num_points, int num_features,
                         const float* voxel_size, const float* coors_range, int max_points,
                         int max_voxels, int ndim) {
  int grid[3];
  grid_size_of(voxel_size, coors_range, ndim, grid);
  int32_t* tmp = (int32_t*)malloc((size_t)num_points * ndim * sizeof(int32_t));
  oracle_dynamic_voxelize(points, tmp, num_points, num_features, voxel_size, coors_range, ndim);
  size_t ncell = (size_t)grid[0] * grid[1] * grid[2];
  int32_t* c2v = (int32_t*)malloc(ncell * sizeof(int32_t));
  for (size_t i = 0; i < ncell; ++i) c2v[i] = -1;
  int voxel_num = 0;
  for (int i = 0; i < num_points; ++i) {
    const int32_t* c = tmp + (size_t)i * ndim;
    if (c[0] == -1) continue;
    size_t cell = ((size_t)c[0] * grid[1] + c[1]) * grid[0] + c[2]; /* [z][y][x] */
    int vid = c2v[cell];
    if (vid == -1) {
      vid = voxel_num;
      if (max_voxels != -1 && voxel_num >= max_voxels) continue;
      voxel_num += 1;
      c2v[cell] = vid;
      for (int k = 0; k < ndim; ++k) coors[(size_t)vid * ndim + k] = c[k];
    }
    int num = num_points_per_voxel[vid];
    if (max_points == -1 || num < max_points) {
      memcpy(voxels + ((size_t)vid * max_points + num) * num_features,
             points + (size_t)i * num_features, sizeof(float) * num_features);
      num_points_per_voxel[vid] += 1;
    }
  }
  free(tmp);
  free(c2v);
  return voxel_num;
}

/* ---- dynamic scatter ----------------------------------------------------------------- */
typedef struct { int32_t c[3]; int32_t idx; } coor_rec;

static int cmp_rec(const void* a, const void* b) {
  const coor_rec* x = (const coor_rec*)a;
  const coor_rec* y = (const coor_rec*)b;
  for (int k = 0; k < 3; ++k) {
    if (x->c[k] < y->c[k]) return -1;
    if (x->c[k] > y->c[k]) return 1;
  }
  return (x->idx > y->idx) - (x->idx < y->idx);
}

/* mmdet3d/ops/voxel/src/scatter_points_cuda.cu:183-239 (dynamic_point_to_voxel_forward_gpu):
 *   coors_clean = any(coor<0) ? (-1,-1,-1) : coor                       (:199)
 *   (out_coors, coors_map, reduce_count) = unique_dim(coors_clean, sorted, inverse, counts) (:201-202)
 *   drop the leading (-1,-1,-1) row, coors_map -= 1 (so invalid points map to -1)  (:204-209)
 *   reduced = max / sum / mean of feats over the map                        (:216-233, :80-103)
 * reduce_type: 0 sum, 1 mean, 2 max (voxelization.h:4).
 * Outputs sized for the worst case (N rows); returns M.  Sum order = point-index order. */
int oracle_dynamic_scatter_forward(const float* feats, const int32_t* coors, int num_input,
                                   int num_feats, int reduce_type, float* reduced,
                                   int32_t* out_coors, int32_t* coors_map, int32_t* reduce_count) {
  if (num_input == 0) return 0;
  coor_rec* rec = (coor_rec*)malloc((size_t)num_input * sizeof(coor_rec));
  for (int i = 0; i < num_input; ++i) {
    const int32_t* c = coors + (size_t)i * 3;
    int bad = c[0] < 0 || c[1] < 0 || c[2] < 0;
    for (int k = 0; k < 3; ++k) rec[i].c[k] = bad ? -1 : c[k];
    rec[i].idx = i;
  }
  qsort(rec, num_input, sizeof(coor_rec), cmp_rec);
  int m = -1; /* current unique row (including the invalid row if present) */
  int has_invalid = rec[0].c[0] < 0;
  for (int j = 0; j < num_input; ++j) {
    if (j == 0 || memcmp(rec[j].c, rec[j - 1].c, sizeof(int32_t) * 3) != 0) ++m;
    int row = has_invalid ? m - 1 : m;
    coors_map[rec[j].idx] = row;
    if (row >= 0) {
      if (j == 0 || memcmp(rec[j].c, rec[j - 1].c, sizeof(int32_t) * 3) != 0) {
        memcpy(out_coors + (size_t)row * 3, rec[j].c, sizeof(int32_t) * 3);
        reduce_count[row] = 0;
      }
      reduce_count[row] += 1;
    }
  }
  int M = has_invalid ? m : m + 1;
  for (size_t i = 0; i < (size_t)M * num_feats; ++i) reduced[i] = reduce_type == 2 ? -INFINITY : 0.f;
  for (int i = 0; i < num_input; ++i) { /* point-index order */
    int to = coors_map[i];
    if (to < 0) continue;
    const float* f = feats + (size_t)i * num_feats;
    float* r = reduced + (size_t)to * num_feats;
    if (reduce_type == 2) {
      for (int k = 0; k < num_feats; ++k) r[k] = fmaxf(r[k], f[k]);
    } else {
      for (int k = 0; k < num_feats; ++k) r[k] += f[k];
    }
  }
  if (reduce_type == 1)
    for (int v = 0; v < M; ++v)
      for (int k = 0; k < num_feats; ++k) reduced[(size_t)v * num_feats + k] /= (float)reduce_count[v];
  free(rec);
  return M;
}

/* mmdet3d/ops/voxel/src/scatter_points_cuda.cu:241-308 (dynamic_point_to_voxel_backward_gpu):
 *   grad_feats = 0
 *   sum : grad_feats[i] = grad_reduced[map[i]]                       (:122-125)
 *   mean: grad_feats[i] = grad_reduced[map[i]] / count[map[i]]        (:126-130)
 *   max : per (voxel, feat) the LOWEST point index i with feats[i]==reduced (atomicMin :154-157)
 *         receives grad_reduced, all others 0                          (:162-179) */
void oracle_dynamic_scatter_backward(float* grad_feats, const float* grad_reduced,
                                     const float* feats, const float* reduced,
                                     const int32_t* coors_map, const int32_t* reduce_count,
                                     int num_input, int num_reduced, int num_feats, int reduce_type) {
  memset(grad_feats, 0, (size_t)num_input * num_feats * sizeof(float));
  if (num_input == 0 || num_reduced == 0) return;
  if (reduce_type == 0 || reduce_type == 1) {
    for (int i = 0; i < num_input; ++i) {
      int to = coors_map[i];
      if (to < 0) continue;
      for (int k = 0; k < num_feats; ++k) {
        float g = grad_reduced[(size_t)to * num_feats + k];
        grad_feats[(size_t)i * num_feats + k] = reduce_type == 0 ? g : g / (float)reduce_count[to];
      }
    }
  } else {
    int32_t* from = (int32_t*)malloc((size_t)num_reduced * num_feats * sizeof(int32_t));
    for (size_t i = 0; i < (size_t)num_reduced * num_feats; ++i) from[i] = num_input;
    for (int i = 0; i < num_input; ++i) {
      int to = coors_map[i];
      if (to < 0) continue;
      for (int k = 0; k < num_feats; ++k)
        if (feats[(size_t)i * num_feats + k] == reduced[(size_t)to * num_feats + k] &&
            i < from[(size_t)to * num_feats + k])
          from[(size_t)to * num_feats + k] = i;
    }
    for (int v = 0; v < num_reduced; ++v)
      for (int k = 0; k < num_feats; ++k) {
        int src = from[(size_t)v * num_feats + k];
        if (src < num_input) /* the reference writes out of bounds otherwise; cannot happen for finite feats */
          grad_feats[(size_t)src * num_feats + k] = grad_reduced[(size_t)v * num_feats + k];
      }
    free(from);
  }
}

/* mmdet3d/models/middle_encoders/pillar_scatter.py:62-102 (forward_batch):
 *   canvas[b, :, y*nx + x] = voxel_features[v, :]  for coors[v] = (b, z, y, x); zeros elsewhere.
 * canvas f32[B, C, ny, nx].  Later rows overwrite earlier ones on duplicates (index_put order). */
void oracle_pillars_scatter(const float* voxel_features, const int32_t* coors, int num_voxels,
                            int C, int B, int ny, int nx, float* canvas) {
  memset(canvas, 0, (size_t)B * C * ny * nx * sizeof(float));
  for (int v = 0; v < num_voxels; ++v) {
    int b = coors[(size_t)v * 4 + 0], y = coors[(size_t)v * 4 + 2], x = coors[(size_t)v * 4 + 3];
    if (b < 0 || b >= B) continue;
    for (int c = 0; c < C; ++c)
      canvas[(((size_t)b * C + c) * ny + y) * nx + x] = voxel_features[(size_t)v * C + c];
  }
}
