// CenterHead training targets on the device.
//
// Replaces CenterHead.get_targets / get_targets_single (mmdet3d/models/dense_heads/centerpoint_head.py:366-413,
// 447-611) + draw_heatmap_gaussian / gaussian_2d / gaussian_radius (mmdet3d/core/utils/gaussian.py:6-88): the
// reference walks the GT boxes of every sample in a Python loop on the label device (hundreds of tiny launches and
// .item() round trips per sample); here two launches build the targets of ALL tasks of the whole batch.
//
//   ct_boxes   : one thread per GT box.  Task / class of the box, its slot k = rank among the sample's boxes of the
//                same task ordered by (class, original index) (the reference concatenates per-class index lists,
//                :470-487), feature-map centre, Gaussian radius (fp32, the reference's 0-dim float32 arithmetic),
//                anno_box / ind / mask rows, and a compact record for the heat-map pass.
//   ct_heatmap : one thread per heat-map element: max over the sample's boxes of that class whose window covers the
//                pixel of float(exp(-(dx^2+dy^2) / (2 sigma^2))), sigma = (2r+1)/6, evaluated in fp64 like numpy
//                (gaussian.py:6-22).  max is order independent -> deterministic, no atomics.
// Arithmetic notes: fp32 divisions / sqrt are IEEE (hipcc default), contraction is off in ct_boxes so that every
// intermediate rounds as in the reference; logf / sinf / cosf and the fp64 exp may differ from numpy's routines in the
// last ulp (tests allow 2 ulp on those columns; the heat map compares equal on the fixtures).
#include <math.h>

#include "common.h"

namespace {

constexpr int CT_MAX_TASKS = 16;
constexpr int CT_MAX_B = 64;
constexpr int CT_MAX_BOXES = 1024;   // per sample, LDS-resident records in ct_heatmap

struct CtParams {
  int B, T, H, W, max_objs, min_radius, norm_bbox;
  int cls_start[CT_MAX_TASKS + 1];   // task t owns global classes [cls_start[t], cls_start[t+1])
  int box_start[CT_MAX_B + 1];       // boxes of sample b: [box_start[b], box_start[b+1])
  float pc0, pc1, vs0, vs1, osf, overlap;
};

struct BoxRec {
  int cls;      // global class id, -1 = draws nothing
  int ix, iy, radius;
};

__device__ __forceinline__ float gaussian_radius_f32(float height, float width, float mo) {
#pragma clang fp contract(off)
  const float b1 = height + width;
  const float c1 = width * height * (1.f - mo) / (1.f + mo);
  const float r1 = (b1 + sqrtf(b1 * b1 - 4.f * c1)) / 2.f;
  const float b2 = 2.f * (height + width);
  const float c2 = (1.f - mo) * width * height;
  const float r2 = (b2 + sqrtf(b2 * b2 - 16.f * c2)) / 2.f;
  const float a3 = 4.f * mo;
  const float b3 = -2.f * mo * (height + width);
  const float c3 = (mo - 1.f) * width * height;
  const float r3 = (b3 + sqrtf(b3 * b3 - 4.f * a3 * c3)) / 2.f;
  return fminf(r1, fminf(r2, r3));
}

__global__ __launch_bounds__(256) void ct_boxes(const float* __restrict__ boxes9, const int* __restrict__ labels,
                                                CtParams P, int n_boxes, BoxRec* __restrict__ recs,
                                                float* __restrict__ anno /* [T,B,max_objs,10] */,
                                                long long* __restrict__ ind /* [T,B,max_objs] */,
                                                unsigned char* __restrict__ mask /* [T,B,max_objs] */) {
#pragma clang fp contract(off)
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_boxes) return;
  BoxRec rec{-1, 0, 0, 0};
  int b = 0;
  while (b + 1 < P.B && i >= P.box_start[b + 1]) ++b;
  const int lab = labels[i];
  int t = -1;
  for (int k = 0; k < P.T; ++k)
    if (lab >= P.cls_start[k] && lab < P.cls_start[k + 1]) t = k;
  if (t >= 0) {
    // slot = number of the sample's boxes of the same task that precede this one in (class, index) order
    int slot = 0;
    for (int j = P.box_start[b]; j < P.box_start[b + 1]; ++j) {
      const int lj = labels[j];
      if (lj >= P.cls_start[t] && lj < P.cls_start[t + 1] && (lj < lab || (lj == lab && j < i))) ++slot;
    }
    if (slot < P.max_objs) {
      const float* bx = boxes9 + static_cast<size_t>(i) * 9;
      const float width = bx[3] / P.vs0 / P.osf;
      const float length = bx[4] / P.vs1 / P.osf;
      if (width > 0.f && length > 0.f) {
        int radius = static_cast<int>(gaussian_radius_f32(length, width, P.overlap));   // int(): toward zero
        radius = radius > P.min_radius ? radius : P.min_radius;
        const float cx = (bx[0] - P.pc0) / P.vs0 / P.osf;
        const float cy = (bx[1] - P.pc1) / P.vs1 / P.osf;
        const float tx = truncf(cx), ty = truncf(cy);
        if (tx >= 0.f && tx < static_cast<float>(P.W) && ty >= 0.f && ty < static_cast<float>(P.H)) {
          const int ix = static_cast<int>(tx), iy = static_cast<int>(ty);
          rec = BoxRec{lab, ix, iy, radius};
          const size_t row = (static_cast<size_t>(t) * P.B + b) * P.max_objs + slot;
          ind[row] = static_cast<long long>(iy) * P.W + ix;
          mask[row] = 1;
          float* a = anno + row * 10;
          a[0] = cx - static_cast<float>(ix);
          a[1] = cy - static_cast<float>(iy);
          a[2] = bx[2];
          a[3] = P.norm_bbox ? logf(bx[3]) : bx[3];
          a[4] = P.norm_bbox ? logf(bx[4]) : bx[4];
          a[5] = P.norm_bbox ? logf(bx[5]) : bx[5];
          a[6] = sinf(bx[6]);
          a[7] = cosf(bx[6]);
          a[8] = bx[7];
          a[9] = bx[8];
        }
      }
    }
  }
  recs[i] = rec;
}

// grid (ceil(H*W/256), total classes, B)
__global__ __launch_bounds__(256) void ct_heatmap(const BoxRec* __restrict__ recs, CtParams P, int n_cls,
                                                  float* __restrict__ heatmap /* [B, n_cls, H, W] */) {
  __shared__ BoxRec srec[CT_MAX_BOXES];
  __shared__ int n_mine;
  const int b = blockIdx.z, cls = blockIdx.y;
  const int j0 = P.box_start[b], nb = P.box_start[b + 1] - j0;
  if (threadIdx.x == 0) n_mine = 0;
  __syncthreads();
  for (int j = threadIdx.x; j < nb; j += blockDim.x) {
    const BoxRec r = recs[j0 + j];
    if (r.cls == cls) srec[atomicAdd(&n_mine, 1)] = r;     // order irrelevant: max below
  }
  __syncthreads();
  const int pix = blockIdx.x * blockDim.x + threadIdx.x;
  if (pix >= P.H * P.W) return;
  const int y = pix / P.W, x = pix - y * P.W;
  float v = 0.f;
  const int n = n_mine;
  for (int j = 0; j < n; ++j) {
    const BoxRec r = srec[j];
    const int dx = x - r.ix, dy = y - r.iy;
    if (dx < -r.radius || dx > r.radius || dy < -r.radius || dy > r.radius) continue;
    const double sigma = static_cast<double>(2 * r.radius + 1) / 6.0;
    const double xd = dx, yd = dy;
    double g = exp(-(xd * xd + yd * yd) / (2.0 * sigma * sigma));
    if (g < 2.220446049250313e-16) g = 0.0;                 // h[h < eps * h.max()] = 0, h.max() == 1
    v = fmaxf(v, static_cast<float>(g));
  }
  heatmap[(static_cast<size_t>(b) * n_cls + cls) * P.H * P.W + pix] = v;
}

}  // namespace

extern "C" int dbev_centerhead_targets(const float* boxes9, const int32_t* labels, const int32_t* box_start_host,
                                       int B, const int32_t* task_num_classes_host, int num_tasks, int H, int W,
                                       int max_objs, int min_radius, float gaussian_overlap, float pc_x, float pc_y,
                                       float voxel_x, float voxel_y, int out_size_factor, int norm_bbox,
                                       float* heatmap, float* anno_box, long long* ind, unsigned char* mask,
                                       void* workspace, size_t workspace_bytes, dbevStream_t stream) {
  if (B <= 0 || B > CT_MAX_B || num_tasks <= 0 || num_tasks > CT_MAX_TASKS || H <= 0 || W <= 0 || max_objs <= 0 ||
      out_size_factor <= 0 || !(voxel_x > 0.f) || !(voxel_y > 0.f) || heatmap == nullptr || anno_box == nullptr ||
      ind == nullptr || mask == nullptr)
    return DBEV_EINVAL;
  CtParams P;
  P.B = B; P.T = num_tasks; P.H = H; P.W = W; P.max_objs = max_objs; P.min_radius = min_radius;
  P.norm_bbox = norm_bbox;
  P.cls_start[0] = 0;
  for (int t = 0; t < num_tasks; ++t) {
    if (task_num_classes_host[t] <= 0) return DBEV_EINVAL;
    P.cls_start[t + 1] = P.cls_start[t] + task_num_classes_host[t];
  }
  for (int b = 0; b <= B; ++b) {
    P.box_start[b] = box_start_host[b];
    if (b > 0 && (P.box_start[b] < P.box_start[b - 1] || P.box_start[b] - P.box_start[b - 1] > CT_MAX_BOXES))
      return DBEV_EINVAL;
  }
  if (P.box_start[0] != 0) return DBEV_EINVAL;
  P.pc0 = pc_x; P.pc1 = pc_y; P.vs0 = voxel_x; P.vs1 = voxel_y;
  P.osf = static_cast<float>(out_size_factor);
  P.overlap = gaussian_overlap;
  const int n_boxes = P.box_start[B];
  const int n_cls = P.cls_start[num_tasks];
  if (n_boxes > 0 && (boxes9 == nullptr || labels == nullptr)) return DBEV_EINVAL;
  const size_t need = sizeof(BoxRec) * static_cast<size_t>(n_boxes > 0 ? n_boxes : 1);
  if (workspace == nullptr || workspace_bytes < need) return DBEV_EINVAL;
  hipStream_t s = dbev_stream(stream);
  const size_t rows = static_cast<size_t>(num_tasks) * B * max_objs;
  DBEV_HIP_TRY(hipMemsetAsync(anno_box, 0, sizeof(float) * rows * 10, s));
  DBEV_HIP_TRY(hipMemsetAsync(ind, 0, sizeof(long long) * rows, s));
  DBEV_HIP_TRY(hipMemsetAsync(mask, 0, rows, s));
  BoxRec* recs = static_cast<BoxRec*>(workspace);
  if (n_boxes > 0)
    hipLaunchKernelGGL(ct_boxes, dim3(dbev_ceil_div(n_boxes, 256)), dim3(256), 0, s, boxes9, labels, P, n_boxes, recs,
                       anno_box, ind, mask);
  hipLaunchKernelGGL(ct_heatmap, dim3(dbev_ceil_div(H * W, 256), n_cls, B), dim3(256), 0, s, recs, P, n_cls, heatmap);
  DBEV_LAUNCH_CHECK();
  return 0;
}
