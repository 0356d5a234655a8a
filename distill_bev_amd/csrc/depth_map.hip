// LiDAR points -> per-camera sparse depth maps (the depth supervision of BEVDepth, img_inputs[-1]).
//
// Replaces the data-loader transform PointToMultiViewDepth (mmdet3d/datasets/pipelines/loading.py:18-61): per camera
//   p   = (point - trans) @ inverse(rots @ inverse(intrins)).T         lidar -> camera pixel ray
//   p   = (p.x / p.z, p.y / p.z, p.z) @ post_rots.T + post_trans       image augmentation
//   u,v = round(p.xy / downsample); kept if inside the map and dbound[0] <= depth < dbound[1]
//   map[v, u] = the NEAREST of the points that land on the pixel   (the reference sorts by pixel + depth / 100 and keeps
//                                                                  the first point of every pixel, :33-41)
// One thread per (point, camera); the per-pixel minimum is an integer atomicMin on the depth's bit pattern (depths are
// positive, so the float order is the unsigned order): exact and order-independent, no sort.  The map is pre-set to
// +inf bits and a second tiny pass turns untouched pixels into 0 (the reference's background value).
#include "common.h"

namespace {

struct CamMats { float cinv[9], post[9], trans[3], ptrans[3]; };

__global__ __launch_bounds__(256) void depth_min_kernel(const float* __restrict__ points, int n, int F,
                                                        const CamMats* __restrict__ cams, int n_cam, int h, int w,
                                                        float inv_ds, float dmin, float dmax,
                                                        unsigned* __restrict__ map) {
  const long long t = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= static_cast<long long>(n) * n_cam) return;
  const int cam = static_cast<int>(t / n), i = static_cast<int>(t - static_cast<long long>(cam) * n);
  const CamMats& m = cams[cam];
  const float x = points[static_cast<size_t>(i) * F] - m.trans[0], y = points[static_cast<size_t>(i) * F + 1] - m.trans[1],
              z = points[static_cast<size_t>(i) * F + 2] - m.trans[2];
  // row-vector times matrix-transpose, accumulated over j = 0, 1, 2 like the reference's matmul
  const float px = fmaf(z, m.cinv[2], fmaf(y, m.cinv[1], x * m.cinv[0]));
  const float py = fmaf(z, m.cinv[5], fmaf(y, m.cinv[4], x * m.cinv[3]));
  const float pz = fmaf(z, m.cinv[8], fmaf(y, m.cinv[7], x * m.cinv[6]));
  const float ux = px / pz, uy = py / pz;
  const float qx = fmaf(pz, m.post[2], fmaf(uy, m.post[1], ux * m.post[0])) + m.ptrans[0];
  const float qy = fmaf(pz, m.post[5], fmaf(uy, m.post[4], ux * m.post[3])) + m.ptrans[1];
  const float d = fmaf(pz, m.post[8], fmaf(uy, m.post[7], ux * m.post[6])) + m.ptrans[2];
  const float cx = rintf(qx * inv_ds), cy = rintf(qy * inv_ds);           // torch.round: half to even
  if (!(cx >= 0.f && cx < static_cast<float>(w) && cy >= 0.f && cy < static_cast<float>(h) && d < dmax && d >= dmin)) return;
  atomicMin(&map[(static_cast<size_t>(cam) * h + static_cast<int>(cy)) * w + static_cast<int>(cx)], __float_as_uint(d));
}

__global__ __launch_bounds__(256) void depth_fill_kernel(unsigned* __restrict__ map, long long n, bool init) {
  const long long t = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= n) return;
  if (init) map[t] = 0x7f800000u;                     // +inf
  else if (map[t] == 0x7f800000u) map[t] = 0u;        // no point on this pixel -> 0.0
}

}  // namespace

extern "C" int dbev_points_to_depth_maps(const float* points, int n_points, int n_feats, const float* cam_mats, int n_cams,
                                         int height, int width, int downsample, float depth_min, float depth_max,
                                         float* depth_maps, dbevStream_t stream) {
  if (n_points < 0 || n_feats < 3 || n_cams <= 0 || height <= 0 || width <= 0 || downsample <= 0 || cam_mats == nullptr ||
      depth_maps == nullptr || (n_points > 0 && points == nullptr) || !(depth_min >= 0.f) || !(depth_max > depth_min))
    return DBEV_EINVAL;
  const int h = height / downsample, w = width / downsample;
  if (h <= 0 || w <= 0) return DBEV_EINVAL;
  hipStream_t s = dbev_stream(stream);
  unsigned* map = reinterpret_cast<unsigned*>(depth_maps);
  const long long cells = static_cast<long long>(n_cams) * h * w;
  hipLaunchKernelGGL(depth_fill_kernel, dim3(dbev_ceil_div(cells, 256)), dim3(256), 0, s, map, cells, true);
  if (n_points > 0)
    hipLaunchKernelGGL(depth_min_kernel, dim3(dbev_ceil_div(static_cast<long long>(n_points) * n_cams, 256)), dim3(256), 0, s, points,
                       n_points, n_feats, reinterpret_cast<const CamMats*>(cam_mats), n_cams, h, w, 1.0f / static_cast<float>(downsample),
                       depth_min, depth_max, map);
  hipLaunchKernelGGL(depth_fill_kernel, dim3(dbev_ceil_div(cells, 256)), dim3(256), 0, s, map, cells, false);
  DBEV_LAUNCH_CHECK();
  return 0;
}
