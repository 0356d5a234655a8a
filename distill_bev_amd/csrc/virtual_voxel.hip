// Dynamic voxel encoders of the voxel teachers (configs[4]) for gfx950 -- replaces the torch op sequences of
// mmdet3d/models/voxel_encoders/dynamic_voxel_encoder.py: `voxelization` (:8-17) and `voxelization_virtual` (:19-68).
//
// The reference (i) boolean-indexes the cloud three times (real / painted / virtual rows), (ii) writes a zero-padded
// [N, 24] copy of it with the three classes in disjoint column groups, (iii) runs `unique(dim=0)` (a sort of the
// coordinate rows) and two `scatter_add_`s, (iv) rescales the "mixed" voxels with two masked divisions.  Here the cloud is
// read twice and nothing per-point is written except one int3:
//   dbev_range_voxel_coords   one thread per point: range test, class test, (z, y, x) = trunc((p - min) / size) as the
//                             reference computes it (fp32 subtract, IEEE divide, truncation), (-1,-1,-1) for dropped
//                             points -- the row convention of the dynamic-scatter grouping (dyn_scatter.hip), which
//                             yields the voxels in `unique`'s lexicographic order and the points of a voxel in ascending id;
//   dbev_virtual_voxel_reduce 32 lanes per voxel (two voxels per wave), lane = one of the 24 padded columns: a point's
//                             17-float row is one coalesced load of the lane group, the class (column 15) and the source
//                             column of every lane come out of it by a lane permute; per-column sums in the REFERENCE's
//                             order (real and painted points touch disjoint columns, virtual points follow the painted
//                             ones in columns 6..22: pass 1 = non-virtual points, pass 2 = virtual points, both in
//                             ascending point id = the sequential scatter_add order on the CPU); mean, indicator and
//                             the two mixed-voxel divisions in registers; 23 floats per voxel written once.
#include "common.h"

namespace {

struct RangeArgs { float lo[3], hi[3], vs[3]; };

// classes of MVP's virtual-point clouds, column F-2 (dynamic_voxel_encoder.py:27-29)
__device__ __forceinline__ int vv_class(float tag) { return tag == 1.f ? 1 : tag == 0.f ? 0 : tag == -1.f ? -1 : 2; }

template <bool VIRTUAL>
__global__ __launch_bounds__(256) void vv_coords(const float* __restrict__ pts, int n, int F, RangeArgs a,
                                                 int* __restrict__ coors) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* p = pts + static_cast<size_t>(i) * F;
  const float x = p[0], y = p[1], z = p[2];
  bool keep = x >= a.lo[0] && x <= a.hi[0] && y >= a.lo[1] && y <= a.hi[1] && z >= a.lo[2] && z <= a.hi[2];
  if (VIRTUAL) keep = keep && vv_class(p[F - 2]) != 2;
  int cz = -1, cy = -1, cx = -1;
  if (keep) {  // .to(torch.int64) truncates; inside the range the quotient is >= 0
    cz = static_cast<int>((z - a.lo[2]) / a.vs[2]);
    cy = static_cast<int>((y - a.lo[1]) / a.vs[1]);
    cx = static_cast<int>((x - a.lo[0]) / a.vs[0]);
  }
  coors[i * 3 + 0] = cz;
  coors[i * 3 + 1] = cy;
  coors[i * 3 + 2] = cx;
}

constexpr int VV_F = 17;    // x y z intensity t + 10 class scores + tag + score  (the reference hard-codes it, :20-21)
constexpr int VV_OUT = 23;  // 24 padded columns minus the indicator

__global__ __launch_bounds__(256) void vv_reduce(const float* __restrict__ pts, const int* __restrict__ vstart,
                                                 const unsigned* __restrict__ vlist, float* __restrict__ out, int m) {
  const int g = (blockIdx.x * blockDim.x + threadIdx.x) >> 5;  // voxel
  const int c = threadIdx.x & 31;                              // padded column
  if (g >= m) return;                                          // whole 32-lane group leaves together
  const int st = vstart[g];
  const int L = vstart[g + 1] - st;
  // source column of padded column c: real rows -> [0,1,2,3,4,16], painted / virtual rows -> 0..14, tag
  const int src_real = c < 5 ? c : 16;
  const int src_pv = c < 21 ? c - 6 : 15;
  float acc = 0.f;
  int n_real = 0;
  for (int pass = 0; pass < 2; ++pass) {
    for (int j = 0; j < L; ++j) {
      const unsigned pid = vlist[st + j];
      const float v = c < VV_F ? pts[static_cast<size_t>(pid) * VV_F + c] : 0.f;
      const int cls = vv_class(__shfl(v, 15, 32));
      const float from_real = __shfl(v, src_real, 32);
      const float from_pv = __shfl(v, src_pv & 31, 32);
      if (pass == 0) {
        if (cls == 1) {
          ++n_real;
          if (c < 6) acc += from_real;
        } else if (cls == 0) {
          if (c >= 6 && c < 22) acc += from_pv;
          else if (c == 22) acc += 1.f;
        }
      } else if (cls == -1) {
        if (c >= 6 && c < 22) acc += from_pv;
      }
    }
  }
  const float cnt = static_cast<float>(L);
  float mean = acc / cnt;                            // scatter_mean: sum / count (scatter.py:37-60)
  const float ind = static_cast<float>(n_real) / cnt;  // padded column 23
  if (ind > 0.f && ind < 1.f) mean = c < 6 ? mean / ind : mean / (1.f - ind);  // :64-66
  if (c < VV_OUT) out[static_cast<size_t>(g) * VV_OUT + c] = mean;
}

}  // namespace

extern "C" int dbev_range_voxel_coords(const float* points, int num_points, int num_feats, const float* pc_range_host,
                                       const float* voxel_size_host, int virtual_classes, int32_t* coors,
                                       dbevStream_t stream) {
  if (num_points < 0 || num_feats < 3 || pc_range_host == nullptr || voxel_size_host == nullptr) return DBEV_EINVAL;
  if (virtual_classes && num_feats != VV_F) return DBEV_EINVAL;
  if (num_points == 0) return 0;
  RangeArgs a;
  for (int k = 0; k < 3; ++k) {
    a.lo[k] = pc_range_host[k];
    a.hi[k] = pc_range_host[3 + k];
    a.vs[k] = voxel_size_host[k];
  }
  const int nb = dbev_ceil_div(num_points, 256);
  if (virtual_classes)
    hipLaunchKernelGGL(vv_coords<true>, dim3(nb), dim3(256), 0, dbev_stream(stream), points, num_points, num_feats, a, coors);
  else
    hipLaunchKernelGGL(vv_coords<false>, dim3(nb), dim3(256), 0, dbev_stream(stream), points, num_points, num_feats, a, coors);
  DBEV_LAUNCH_CHECK();
  return 0;
}

extern "C" int dbev_virtual_voxel_reduce(const float* points, const int32_t* voxel_point_start,
                                         const int32_t* voxel_point_list, float* voxels, int num_voxels,
                                         dbevStream_t stream) {
  if (num_voxels < 0) return DBEV_EINVAL;
  if (num_voxels == 0) return 0;
  hipLaunchKernelGGL(vv_reduce, dim3(dbev_ceil_div(num_voxels, 8)), dim3(256), 0, dbev_stream(stream), points,
                     voxel_point_start, reinterpret_cast<const unsigned*>(voxel_point_list), voxels, num_voxels);
  DBEV_LAUNCH_CHECK();
  return 0;
}
