/*
 * dbev_hip.h -- C ABI of libdbev_hip.so, the MI355X (gfx950) implementation of the
 * DistillBEV training-step hot path.
 *
 * The reference (qcraftai/distill-bev) has no C ABI: its native boundary is a set of
 * pybind11 torch-extension entry points taking at::Tensor.  Every function below
 * replaces one of those entry points (cited per function, paths relative to the
 * reference root) with plain device pointers + sizes + a HIP stream, so that any host
 * (the Python mirror in distill_bev_amd/, a C++ trainer, ...) can bind it without
 * torch in the signature.  INTEGRATION.md shows the reference-side binding.
 *
 * Conventions
 *   - all pointers are DEVICE pointers (HBM) unless the name ends in _host;
 *   - tensors are dense row-major ("contiguous") in the stated shape;
 *   - `stream` is a hipStream_t (NULL = default stream); every launch is asynchronous
 *     on that stream, nothing here synchronises, allocates or frees -> graph-capturable;
 *   - return value: 0 on success, otherwise a hipError_t (launch/config error) or
 *     DBEV_EINVAL for an argument the reference would have rejected with a C++ exception;
 *   - workspaces are caller-allocated; *_workspace_bytes() tells how much.
 */
#ifndef DBEV_HIP_H
#define DBEV_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DBEV_EINVAL 10001
#define DBEV_ABI_VERSION 1

typedef void* dbevStream_t; /* hipStream_t */

/* reduce_t of mmdet3d/ops/voxel/src/voxelization.h:4 */
enum { DBEV_REDUCE_SUM = 0, DBEV_REDUCE_MEAN = 1, DBEV_REDUCE_MAX = 2 };

int dbev_abi_version(void);
/* name of the code object's target ("gfx950") -- lets a host check what it loaded */
const char* dbev_target_arch(void);

/* ------------------------------------------------------------------------------------
 * Per-KERNEL timing (measurement aid, no counterpart in the reference; bench.py `roofline`).
 * While enabled, the multi-kernel entry points bracket every kernel they launch with a
 * hipEvent pair recorded on the launch stream and remember the kernel's algorithmic HBM bytes.
 * dbev_kernel_timing_read synchronises on the recorded events, copies up to `cap` records
 * (kernel id, milliseconds, algorithmic bytes) out in launch order, clears the log and returns
 * the number of records it held.  dbev_kernel_timing_enable(mask): bit k of `mask` switches the
 * log on for kernel id k (DBEV_K_*), -1 = every kernel, 0 (the default) = off: no events, no overhead.
 * ---------------------------------------------------------------------------------- */
enum {
  DBEV_K_BN_STATS = 1, DBEV_K_BN_FINALIZE, DBEV_K_BN_APPLY, DBEV_K_BN_APPLY_RES, DBEV_K_BN_BWD_REDUCE,
  DBEV_K_BN_BWD_REDUCE_Y, DBEV_K_BN_BWD_FINALIZE, DBEV_K_BN_BWD_DX, DBEV_K_BN_BWD_DX_RES, DBEV_K_SPCONV_FWD,
  DBEV_K_MSDA_FWD, DBEV_K_MSDA_BWD_SAMPLE, DBEV_K_MSDA_GV_GATHER, DBEV_K_ADAPT_MSE_FWD, DBEV_K_CONV1X1_FWD, DBEV_K_WINO_FWD, DBEV_K_WINO_WGRAD,
  DBEV_K_GEMM1X1_FWD, DBEV_K_GEMM1X1_WGRAD, DBEV_K_B6_FWD, DBEV_K_B6_WGRAD, DBEV_K_STEM_FWD, DBEV_K_STEM_WGRAD, DBEV_K_COUNT
};
int dbev_kernel_timing_enable(int mask);
int dbev_kernel_timing_read(int* kernel_id, float* ms, long long* algorithmic_bytes, int cap);
const char* dbev_kernel_name(int kernel_id);

/* ------------------------------------------------------------------------------------
 * Fallback ledger (no counterpart in the reference).  The host-side mirrors of the fused ops -- bn_act / BatchNormAct2d,
 * SkinnyConv2d, the fused adaptation + masked-MSE op, the fused pillar path, the batched CenterHead branches -- call dbev_fallback_note(site) whenever a DEVICE tensor they were wired for takes the stock torch path
 * (ineligible layout, channel count or mode).  dbev_fallback_count(site): notes since load / the last reset, site < 0 = all
 * sites.  bench.py prints the count of its timed region (config.fallbacks), tests/test_gpu_full_size.py asserts 0 for the step.
 * ---------------------------------------------------------------------------------- */
enum {
  DBEV_FB_BN_ACT = 0, DBEV_FB_SKINNY_CONV, DBEV_FB_ADAPT_MSE, DBEV_FB_PILLAR_VFE, DBEV_FB_HEAD_BATCH, DBEV_FB_DEPTH_HEAD,
  DBEV_FB_SITES
};
int dbev_fallback_note(int site);
long long dbev_fallback_count(int site);
int dbev_fallback_reset(void);

/* ------------------------------------------------------------------------------------
 * bev_pool  (replaces bev_pool_ext: mmdet3d/ops/bev_pool/src/bev_pool.cpp:22-47,60-87,
 *            kernels src/bev_pool_cuda.cu:20-42,61-84)
 * ---------------------------------------------------------------------------------- */

/* bev_pool_forward(x, geom_feats, interval_lengths, interval_starts, b, d, h, w) -> out
 *   x                f32[n, c]      features sorted so that rows of one voxel are adjacent
 *   geom_feats       i32[n, 4]      (x, y, z, b) voxel coordinate of every row
 *   interval_starts  i32[n_int]     first row of each voxel run
 *   interval_lengths i32[n_int]     rows in each run
 *   out              f32[b, d, h, w, c]  caller-allocated; fully (re)written by the callee:
 *                    zero where no interval lands (the reference returns torch::zeros),
 *                    out[gb, gz, gx, gy, :] = sum of the run's rows otherwise.
 * Summation order inside a run is fixed (run-to-run deterministic). */
int dbev_bev_pool_forward(const float* x, const int32_t* geom_feats,
                          const int32_t* interval_starts, const int32_t* interval_lengths,
                          float* out, int n, int c, int n_intervals,
                          int b, int d, int h, int w, dbevStream_t stream);

/* bev_pool_backward(out_grad, geom_feats, interval_lengths, interval_starts, b,d,h,w) -> x_grad
 *   out_grad f32[b, d, h, w, c] ; x_grad f32[n, c] caller-allocated, every row written:
 *   x_grad[r, :] = out_grad[geom[r].b, geom[r].z, geom[r].x, geom[r].y, :]
 * (identical to the reference's per-interval broadcast because the intervals partition
 *  [0, n) and all rows of a run share geom_feats -- bev_pool.py:39-46). */
int dbev_bev_pool_backward(const float* out_grad, const int32_t* geom_feats,
                           const int32_t* interval_starts, const int32_t* interval_lengths,
                           float* x_grad, int n, int c, int n_intervals,
                           int b, int d, int h, int w, dbevStream_t stream);

/* Gather-free form of the same op for the reference's Python surface `bev_pool(feats, coords, B, D, H, W)`
 * (mmdet3d/ops/bev_pool/bev_pool.py:83-97: rank -> argsort -> feats[indices] -> interval sums): instead of sorting and
 * permuting the [n, C] feature rows (0.9 GB moved twice per call at the training shapes), build the cell -> point-list CSR from
 * the integer voxel coordinates once and let dbev_splat_forward / dbev_splat_backward read / write the rows in place.
 *   coords i32[n, 4] = (x, y, z, b), 0<=x<H, 0<=y<W, 0<=z<D, 0<=b<B (rows outside are dropped, point_cell = -1)
 *   cell = ((b*D + z)*H + x)*W + y, i.e. dbev_splat_forward writes out f32[B, D, H, W, C] exactly like dbev_bev_pool_forward.
 * Outputs / workspace as dbev_lift_splat_prepare (workspace: dbev_lift_splat_workspace_bytes(n, B*D*H*W)). */
int dbev_bev_pool_prepare(const int32_t* coords, int n_points, int B, int D, int H, int W, int32_t* point_cell,
                          int32_t* cell_start, int32_t* cell_points, int32_t* n_kept_out, int32_t* hot_cells,
                          int32_t* n_hot_out, void* workspace, size_t workspace_bytes, dbevStream_t stream);

/* ------------------------------------------------------------------------------------
 * voxel_layer  (replaces mmdet3d/ops/voxel/src/voxelization.cpp:6-11 bindings,
 *               voxelization.h:58-140, kernels voxelization_cuda.cu / scatter_points_cuda.cu)
 * Results are those of the reference's CPU implementation (voxelization_cpu.cpp), bit exact.
 * voxel_size_host[3] / coors_range_host[6] are HOST arrays (the reference passes
 * std::vector<float> by value).
 * ---------------------------------------------------------------------------------- */

/* dynamic_voxelize(points, coors, voxel_size, coors_range, NDim=3)  (voxelization.h:86-95)
 *   points f32[num_points, num_features] (xyz first) ; coors i32[num_points, 3] caller-allocated,
 *   every row written: (z, y, x) cell of the point or (-1,-1,-1) if out of range. */
int dbev_dynamic_voxelize(const float* points, int32_t* coors, int num_points, int num_features,
                          const float* voxel_size_host, const float* coors_range_host, int ndim,
                          dbevStream_t stream);

/* hard_voxelize(points, voxels, coors, num_points_per_voxel, voxel_size, coors_range,
 *               max_points, max_voxels, NDim=3, deterministic=true) -> voxel_num
 *   (voxelization.h:58-84).  Output buffers are caller-allocated at capacity
 *   voxels f32[max_voxels, max_points, F], coors i32[max_voxels, 3], num_points i32[max_voxels]
 *   and fully (re)written (zero beyond the data, as voxelize.py:57-62 provides).
 *   The reference returns voxel_num to the host (a sync); here it is stored to the DEVICE
 *   int *voxel_num_out so the call stays asynchronous.  Always the deterministic result
 *   (first-come order of voxelization_cpu.cpp:70-98). */
size_t dbev_hard_voxelize_workspace_bytes(int num_points, const float* voxel_size_host,
                                          const float* coors_range_host);
int dbev_hard_voxelize(const float* points, float* voxels, int32_t* coors,
                       int32_t* num_points_per_voxel, int32_t* voxel_num_out, int num_points,
                       int num_features, const float* voxel_size_host, const float* coors_range_host,
                       int max_points, int max_voxels, int ndim, void* workspace,
                       size_t workspace_bytes, dbevStream_t stream);

/* dynamic_point_to_voxel_forward(feats, coors, reduce_type) ->
 *        [reduced_feats, out_coors, coors_map, reduce_count]          (voxelization.h:108-121)
 * split in two asynchronous calls because M (number of voxels) is data dependent:
 *   prepare: coors i32[N,3] (z,y,x; any negative entry = invalid point) on a grid of
 *            grid_z x grid_y x grid_x cells (all valid coords must be < grid) ->
 *              out_coors i32[<=N,3]   unique coords in ascending (z,y,x) order
 *                                     (= at::unique_dim sorted with the (-1,-1,-1) row dropped)
 *              coors_map i32[N]       row of out_coors for each point, -1 if invalid
 *              reduce_count i32[N]    points per voxel (entries >= M are 0)
 *              voxel_point_start i32[N+1], voxel_point_list i32[N]
 *                                     CSR: ids of the points of voxel v, ascending
 *              num_voxels_out         DEVICE int: M
 *            all caller-allocated at capacity N.
 *   reduce : reduced f32[M, C] = max / sum / mean over each voxel's points (lanes = channels,
 *            sequential point-id order => deterministic; no float atomics). */
size_t dbev_dynamic_scatter_workspace_bytes(int num_points, int grid_z, int grid_y, int grid_x);
int dbev_dynamic_scatter_prepare(const int32_t* coors, int num_points, int grid_z, int grid_y,
                                 int grid_x, int32_t* out_coors, int32_t* coors_map,
                                 int32_t* reduce_count, int32_t* voxel_point_start,
                                 int32_t* voxel_point_list, int32_t* num_voxels_out,
                                 void* workspace, size_t workspace_bytes, dbevStream_t stream);
int dbev_dynamic_scatter_reduce(const float* feats, const int32_t* voxel_point_start,
                                const int32_t* voxel_point_list, float* reduced, int num_voxels,
                                int num_feats, int reduce_type, dbevStream_t stream);

/* dynamic_point_to_voxel_backward(grad_feats, grad_reduced_feats, feats, reduced_feats,
 *                                 coors_idx, reduce_count, reduce_type)  (voxelization.h:123-140)
 *   grad_feats f32[N, C] caller-allocated, every element written.  max: the lowest point id
 *   attaining the voxel max receives the gradient (scatter_points_cuda.cu:154-157). */
int dbev_dynamic_scatter_backward(float* grad_feats, const float* grad_reduced, const float* feats,
                                  const float* reduced, const int32_t* coors_map,
                                  const int32_t* reduce_count, const int32_t* voxel_point_start,
                                  const int32_t* voxel_point_list, int num_points, int num_voxels,
                                  int num_feats, int reduce_type, dbevStream_t stream);

/* ------------------------------------------------------------------------------------
 * PointPillarsScatter.forward_batch (mmdet3d/models/middle_encoders/pillar_scatter.py:62-102;
 * pure torch in the reference, a python loop over samples)
 *   voxel_features f32[M, C]; coors i32[M, 4] = (b, z, y, x)
 *   canvas f32[B, C, ny, nx] (channels_last=0, the reference layout) or physical
 *   [B, ny, nx, C] (channels_last=1); every element written (0 where no pillar).
 *   cellmap i32[B*ny*nx]: caller-allocated scratch, returned filled with the pillar row of
 *   each BEV cell (-1 = empty); it is the saved state of the backward.
 * ---------------------------------------------------------------------------------- */
int dbev_pillars_scatter(const float* voxel_features, const int32_t* coors, int num_voxels, int C,
                         int B, int ny, int nx, float* canvas, int channels_last, int32_t* cellmap,
                         dbevStream_t stream);
int dbev_pillars_scatter_backward(const float* grad_canvas, const int32_t* coors,
                                  const int32_t* cellmap, int num_voxels, int C, int B, int ny, int nx,
                                  int channels_last, float* grad_feats, dbevStream_t stream);

/* ------------------------------------------------------------------------------------
 * Fused Lift-Splat (LSS view transform).  Replaces, for the training step, the sequence
 *   volume = depth.unsqueeze(1) * feat.unsqueeze(2)            (bevdet_distill_more.py:413-416)
 *   ViewTransformerLiftSplatShoot.voxel_pooling(geom, volume)   (view_transformer_mine.py:141-181)
 * and (same grouping, other call surfaces) voxel_pooling_accelerated (:184-240) and
 * mmdet3d.ops.bev_pool (view_transformer.py:140-169).  The reference has no native entry
 * point for the fused op; these are new ones behind the same Python call sites.
 *
 * Point id p = (((b*N + n)*D + d)*H + h)*W + w, n_points = B*N*D*H*W.
 * BEV cell id = ((b*Y + y)*X + x)*Z + z  (so out[cell, c] is the channels-last image of the
 * reference's output tensor f32[B, C*Z, Y, X] with channel = z*C + c).
 * ---------------------------------------------------------------------------------- */
size_t dbev_lift_splat_workspace_bytes(int n_points, int n_cells);

/* geom f32[n_points, 3] ego-frame (x,y,z) of every frustum point (get_geometry output);
 * dx/bx/nx: HOST arrays of the module's grid parameters (nx = cells in X, Y, Z).
 *   point_cell i32[n_points]   BEV cell of the point, -1 if outside the grid
 *                              index = ((geom - (bx - dx/2)) / dx) truncated TOWARD ZERO, fp32
 *   cell_start i32[n_cells+1], cell_points i32[n_points]   CSR: ascending point ids per cell
 *   n_kept_out DEVICE int: number of points inside the grid
 *   hot_cells  i32[n_cells] (capacity), n_hot_out DEVICE int: the cells holding more than 128
 *              points (dense cells next to the cameras); the forward gives each a whole workgroup
 * n_cells = batch*X*Y*Z. */
int dbev_lift_splat_prepare(const float* geom, int n_points, int batch, const float* dx_host,
                            const float* bx_host, const int32_t* nx_host, int32_t* point_cell,
                            int32_t* cell_start, int32_t* cell_points, int32_t* n_kept_out,
                            int32_t* hot_cells, int32_t* n_hot_out, void* workspace,
                            size_t workspace_bytes, dbevStream_t stream);

/* Same, with get_geometry (view_transformer_mine.py:111-139) fused in: no geometry tensor is read.
 *   cam_params f32[BN, 24] per camera = inverse(post_rots) row-major (9), post_trans (3),
 *              rots @ inverse(intrins) row-major (9), trans (3)   -- tiny host/torch-side matrix work
 *   frustum    f32[D, H, W, 3] (create_frustum, vt_mine.py:98-109)
 * The ego-frame point is evaluated as separate fp32 multiplies and adds in the order
 * ((m0*x + m1*y) + m2*z) (no FMA), i.e. bit-identical to distill_bev_amd.lss.get_geometry. */
int dbev_lift_splat_prepare_cam(const float* cam_params, const float* frustum, int BN, int D, int H, int W,
                                int batch, const float* dx_host, const float* bx_host,
                                const int32_t* nx_host, int32_t* point_cell, int32_t* cell_start,
                                int32_t* cell_points, int32_t* n_kept_out, int32_t* hot_cells,
                                int32_t* n_hot_out, void* workspace, size_t workspace_bytes,
                                dbevStream_t stream);

/* depth f32[BN, D, H, W] (softmaxed depth distribution); feat_nhwc f32[BN, H, W, C]
 * (channels-last image features); out f32[n_cells, C], every cell written:
 *   out[cell, :] = sum over the cell's points p of depth[p] * feat[bn(p), h(p), w(p), :].
 * C must be a multiple of 4 and <= 256. */
int dbev_lift_splat_forward(const float* depth, const float* feat_nhwc, const int32_t* cell_start,
                            const int32_t* cell_points, const int32_t* hot_cells, const int32_t* n_hot,
                            float* out, int BN, int D, int H, int W, int C, int n_cells,
                            dbevStream_t stream);

/* grad_out f32[n_cells, C] -> grad_depth f32[BN, D, H, W], grad_feat_nhwc f32[BN, H, W, C];
 * every element of both written (0 for points outside the grid). */
int dbev_lift_splat_backward(const float* grad_out, const float* depth, const float* feat_nhwc,
                             const int32_t* point_cell, float* grad_depth, float* grad_feat_nhwc, int BN,
                             int D, int H, int W, int C, dbevStream_t stream);

/* voxel_pooling(geom, x) for a caller that holds the volume x f32[n_points, C]:
 * out[cell, :] = sum of x[p, :] over the cell's points; backward = gather (0 if dropped). */
int dbev_splat_forward(const float* x, const int32_t* cell_start, const int32_t* cell_points,
                       const int32_t* hot_cells, const int32_t* n_hot, float* out, int n_points, int C,
                       int n_cells, dbevStream_t stream);
int dbev_splat_backward(const float* grad_out, const int32_t* point_cell, float* grad_x, int n_points,
                        int C, dbevStream_t stream);

/* ------------------------------------------------------------------------------------
 * Foreground-masked feature distillation (FGD).  The reference implements these steps in
 * Python/numpy/torch inside mmdet3d/models/detectors/bevdet_distill.py; there is no native
 * entry point to replace -- these are new ones behind the same Python methods.
 * ---------------------------------------------------------------------------------- */

/* foreground_scale_mask (bevdet_distill.py:755-843; geometry box_np_ops.py:426-446,719-753).
 *   planes    f32[sumM, 6, 4]  (nx, ny, nz, d) of the six inward faces of every box, boxes of all
 *                              samples concatenated (host-computed, float32, box_np_ops order)
 *   box_scale f32[sumM]        sqrt(cell_area / (w*l)) per box
 *   box_offsets i32[B+1]       boxes of sample b are [box_offsets[b], box_offsets[b+1])
 *   xs f32[W], ys f32[H]       cell query coordinates (lower-left cell corners, z = 0.5)
 * -> fg, fg_scale, bg_scale f32[B, H, W] ([b, iy, ix], i.e. the reference's un-transposed
 *    masks after its reshape(W,H).transpose()); fg_count i32[B] scratch (cells inside a box).
 * A cell is foreground iff ((px*nx + py*ny) + pz*nz) + d < 0 for all six planes of some box
 * (fp32, no FMA: bit-identical to the reference's numpy test); fg_scale takes the value of the
 * LOWEST-index box containing the cell; bg_scale = float(1.0 / (H*W - n_fg)). */
int dbev_fg_scale_mask(const float* planes, const float* box_scale, const int32_t* box_offsets,
                       const float* xs, const float* ys, int B, int H, int W, float* fg,
                       float* fg_scale, float* bg_scale, int32_t* fg_count, dbevStream_t stream);

/* x f32[B, C, HW] -> pix_mean f32[B, HW] = mean_c |x| ; ch_mean f32[B, C] = mean_hw |x|
 * (inputs of the spatial / channel attention softmaxes, bevdet_distill.py:1084-1097). */
size_t dbev_abs_mean_maps_workspace_bytes(int B, int C, int HW);
int dbev_abs_mean_maps(const float* x, int B, int C, int HW, float* pix_mean, float* ch_mean,
                       void* workspace, size_t workspace_bytes, dbevStream_t stream);

/* S, T f32[B, C, HW] (adapted student / teacher); Wfg, Wbg, Wfp f32[B, HW] per-pixel weights
 * (Wfp may be NULL), Cc f32[B, C] per-channel factor of the third term (may be NULL = 1).
 *   out3[0] = sum (S-T)^2 Wfg ; out3[1] = sum (S-T)^2 Wbg ; out3[2] = sum (S-T)^2 Wfp Cc[c]
 * (kd_fg / kd_bg / kd_fp of bevdet_distill.py:1253-1262,1282-1287 before * weight / B).
 * backward: dS = 2 (S-T) (g[0] Wfg + g[1] Wbg + g[2] Cc[c] Wfp), g = DEVICE f32[3].
 * HW must be a multiple of 4. */
size_t dbev_fgd_masked_mse_workspace_bytes(int B, int C, int HW);
int dbev_fgd_masked_mse_forward(const float* S, const float* T, const float* Wfg, const float* Wbg,
                                const float* Wfp, const float* Cc, int B, int C, int HW, float* out3,
                                void* workspace, size_t workspace_bytes, dbevStream_t stream);
int dbev_fgd_masked_mse_backward(const float* S, const float* T, const float* Wfg, const float* Wbg,
                                 const float* Wfp, const float* Cc, const float* grad_scale3, int B,
                                 int C, int HW, float* dS, dbevStream_t stream);

/* ------------------------------------------------------------------------------------
 * Fused teacher pillar path (frozen CenterPoint-pillar teacher, eval / no_grad):
 *   DynamicCenterPoint.voxelize (dynamic_centerpoint.py:71-93) ->
 *   DynamicPillarFeatureNet.forward with ONE PFN layer, cluster + voxel centre decoration, max
 *   pooling (pillar_encoder.py:283-338) -> PointPillarsScatter.forward_batch (pillar_scatter.py:62-102)
 * in one asynchronous call; the data-dependent pillar count never reaches the host.
 *   points f32[n_points, F] = the B samples' clouds concatenated; sample_start_host i32[B+1] (HOST)
 *   pfn_weight f32[Cout, F+5] (nn.Linear weight, no bias); bn_* f32[Cout] BatchNorm1d eval statistics
 *   voxel_feats f32[n_points, Cout] (capacity; rows >= M untouched), cellmap i32[B*ny*nx],
 *   num_voxels_out DEVICE int, canvas f32[B, Cout, ny, nx] (or channels-last), all caller-allocated.
 * Pillar rows come out in (b, y, x) order -- the order of the reference's sorted-unique scatter.
 * canvas may be NULL: then only voxel_feats / cellmap / num_voxels are produced and the canvas is written by
 * dbev_pillars_canvas (one launch: canvas[b, :, y, x] = voxel_feats[cellmap[b, y, x], :] or 0). */
size_t dbev_pillar_vfe_workspace_bytes(int n_points, int B, int ny, int nx);
int dbev_pillars_canvas(const float* voxel_feats, const int32_t* cellmap, float* canvas, int C, int B, int ny,
                        int nx, int channels_last, dbevStream_t stream);
int dbev_pillar_vfe_canvas(const float* points, int n_points, int num_features,
                           const int32_t* sample_start_host, int B, const float* voxel_size_host,
                           const float* coors_range_host, const float* pfn_weight, const float* bn_weight,
                           const float* bn_bias, const float* bn_mean, const float* bn_var, float bn_eps,
                           int out_channels, float* voxel_feats, int32_t* cellmap, int32_t* num_voxels_out,
                           float* canvas, int channels_last, void* workspace, size_t workspace_bytes,
                           dbevStream_t stream);

/* Channels-last variants of the two loss kernels above: x, S, T are f32[B, HW, C] (the NHWC image of
 * [B, C, H, W]), so the channels-last activations of the dense stack are consumed / dS is produced without a
 * layout copy.  Same arguments and results otherwise; C % 4 == 0, C <= 1024, any HW.  Two extras serve the spatial
 * term of the FGD loss (torch.mean(feat, [1]) -> spatial adaptation -> MSE, bevdet_distill.py:1272-1278) without further
 * passes over the features: pix_signed_mean f32[B, HW] (may be NULL) = mean_c x from the same read as the |x| means, and
 * grad_pixel_mean f32[B, HW] (may be NULL) = gradient w.r.t. mean_c S, added as grad/C to every channel of dS. */
size_t dbev_abs_mean_maps_nhwc_workspace_bytes(int B, int C, int HW);
int dbev_abs_mean_maps_nhwc(const float* x_nhwc, int B, int C, int HW, float* pix_mean, float* ch_mean,
                            float* pix_signed_mean, void* workspace, size_t workspace_bytes, dbevStream_t stream);
size_t dbev_fgd_masked_mse_nhwc_workspace_bytes(int B, int C, int HW);
int dbev_fgd_masked_mse_forward_nhwc(const float* S, const float* T, const float* Wfg, const float* Wbg,
                                     const float* Wfp, const float* Cc, int B, int C, int HW, float* out3,
                                     void* workspace, size_t workspace_bytes, dbevStream_t stream);
int dbev_fgd_masked_mse_backward_nhwc(const float* S, const float* T, const float* Wfg, const float* Wbg,
                                      const float* Wfp, const float* Cc, const float* grad_scale3,
                                      const float* grad_pixel_mean, int B, int C, int HW, float* dS,
                                      dbevStream_t stream);

/* Bilinear upsampling with align_corners=True (nn.Upsample in the student adaptation layers of the
 * FGD loss, bevdet_distill.py:275-288; ATen upsample_bilinear2d index rule).  x f32[B,C,IH,IW] ->
 * y f32[B,C,OH,OW]; channels_last=1: both tensors are physically [B,H,W,C].  backward: every element
 * of grad_x written (gather formulation, no atomics). */
int dbev_upsample_bilinear_ac_forward(const float* x, float* y, int B, int C, int IH, int IW, int OH, int OW,
                                      int channels_last, dbevStream_t stream);
int dbev_upsample_bilinear_ac_backward(const float* grad_y, float* grad_x, int B, int C, int IH, int IW, int OH,
                                       int OW, int channels_last, dbevStream_t stream);
/* F.grid_sample(input, grid, mode='bilinear', padding_mode='zeros', align_corners=True) on a channels-last input, forward only
 * (BEVDepth4D.shift_feature, detectors/bevdet_distill_more.py:41-94: the warped adjacent-frame map is detached).  grid_xy f32[N, Ho, Wo, 2]
 * normalised (x, y); ATen's arithmetic and accumulation order.  C % 4 == 0 */
int dbev_grid_sample_bilinear_nhwc(const float* x_nhwc, const float* grid_xy, int N, int C, int H, int W, int Ho, int Wo, float* y_nhwc,
                                   dbevStream_t stream);

/* Modulated deformable convolution (DCNv2) sampling stage, channels-last.  Replaces mmcv-full 1.6.0
 * `modulated_deform_conv` (un-vendored dependency; reference call site
 * mmdet3d/models/necks/view_transformer_mine.py:298-306,325-329; algorithm: mmcv
 * modulated_deform_conv_cuda_kernel.cuh im2col / col2im / col2im_coord).  deform_groups = groups = 1.
 *   x_nhwc            f32[N,H,W,C]            input feature (C % 4 == 0; C/4 a power of two or a multiple of 64)
 *   offset_mask_nhwc  f32[N,Ho,Wo,3*kh*kw]    RAW conv_offset output: channels 2k,2k+1 = (dy,dx) of tap k,
 *                                             channels 2K+k = mask logits (sigmoid fused in the kernel)
 *   cols              f32[N*Ho*Wo, kh*kw*C]   sigmoid(logit_k) * bilinear(x, p_k), k-major: the NHWC image of
 *                                             [N, K*C, Ho, Wo]; contract with weight[Co, kh, kw, C] (1x1 conv / GEMM)
 * col2im: every grad_x_nhwc element is written by the callee.  C/4 a power of two in [8, 64]: deterministic gather over a
 * per-input-pixel list of tap corners (no float atomics, fixed summation order; needs the workspace); other channel
 * counts: per-(image, channel slice) LDS accumulation, or global float atomics (as in mmcv) for images that do not fit.
 * grad_offset_mask_nhwc has the layout of offset_mask_nhwc (d/d logit includes the sigmoid derivative).
 * Returns DBEV_EINVAL if Ho/Wo do not match the convolution arithmetic. */
int dbev_dcnv2_im2col(const float* x_nhwc, const float* offset_mask_nhwc, float* cols, int N, int C, int H, int W,
                      int Ho, int Wo, int kh, int kw, int stride, int pad, int dil, dbevStream_t stream);
size_t dbev_dcnv2_col2im_workspace_bytes(int N, int C, int H, int W, int Ho, int Wo, int kh, int kw);
int dbev_dcnv2_col2im(const float* grad_cols, const float* x_nhwc, const float* offset_mask_nhwc,
                      float* grad_x_nhwc, float* grad_offset_mask_nhwc, int N, int C, int H, int W, int Ho, int Wo,
                      int kh, int kw, int stride, int pad, int dil, void* workspace, size_t workspace_bytes,
                      dbevStream_t stream);

/* CenterHead training targets for all tasks of a batch in two launches.  Replaces CenterHead.get_targets /
 * get_targets_single (mmdet3d/models/dense_heads/centerpoint_head.py:366-413,447-611) and draw_heatmap_gaussian /
 * gaussian_radius (mmdet3d/core/utils/gaussian.py:6-88).
 *   boxes9 f32[sumM,9] (x, y, z GRAVITY centre, w, l, h, yaw, vx, vy), labels i32[sumM] (global class ids, -1 ignored),
 *   box_start_host i32[B+1] (HOST; boxes of sample b = [box_start[b], box_start[b+1]), <= 1024 per sample),
 *   task_num_classes_host i32[num_tasks] (HOST): task t owns the next task_num_classes[t] global class ids.
 * -> heatmap f32[B, sum(classes), H, W] (task t = channel slice), anno_box f32[T,B,max_objs,10] = (dx, dy, z, log w|l|h
 *    (dims themselves if !norm_bbox), sin yaw, cos yaw, vx, vy), ind i64[T,B,max_objs] = iy*W+ix, mask u8[T,B,max_objs];
 *    all four fully written by the callee.  A box takes slot k = its rank among the sample's boxes of the same task in
 *    (class, index) order; boxes with k >= max_objs, non-positive size or a centre outside the map draw nothing.
 * workspace >= 16 * max(sumM, 1) bytes. */
int dbev_centerhead_targets(const float* boxes9, const int32_t* labels, const int32_t* box_start_host, int B,
                            const int32_t* task_num_classes_host, int num_tasks, int H, int W, int max_objs,
                            int min_radius, float gaussian_overlap, float pc_x, float pc_y, float voxel_x,
                            float voxel_y, int out_size_factor, int norm_bbox, float* heatmap, float* anno_box,
                            long long* ind, unsigned char* mask, void* workspace, size_t workspace_bytes,
                            dbevStream_t stream);

/* CenterHead training loss of all tasks: clip_sigmoid + GaussianFocalLoss (alpha 2, gamma 4, avg_factor = max(#pos, 1)) on
 * the heat maps and the masked L1 terms of the five regression groups xy | z | whl | yaw | vel (avg_factor = #objects +
 * 1e-4).  Replaces the per-task op chain of CenterHead.loss (mmdet3d/models/dense_heads/centerpoint_head.py:615-686,
 * task_specific variant; mmdet 2.24 gaussian_focal_loss / l1_loss).
 *   heads_host        HOST array [num_tasks*6] of DEVICE pointers: per task reg(2) height(1) dim(3) rot(2) vel(2)
 *                     heatmap(ncls[t]) channels, each f32[B,c,H,W]; nhwc_flags_host[i] != 0: tensor i is NHWC-contiguous
 *   sig_out_host      HOST array [num_tasks] of device pointers: clipped sigmoid of the heat maps (layout of the logits)
 *   heatmap/anno_box/ind/mask   targets in the packed layout of dbev_centerhead_targets
 *   losses            f32[num_tasks*6]: per task (loss_xy, loss_z, loss_whl, loss_yaw, loss_vel, loss_heatmap)
 *   avg_factors       f32[2*num_tasks]: max(#pos,1) per task, then #objects + 1e-4 per task (input of backward)
 * backward: grad_heads_host = HOST array [num_tasks*6] of device pointers with the layouts of the inputs; the 5 regression
 * gradients of every task must be ZERO-FILLED by the caller (only the object pixels are written), the heat-map gradients
 * are fully written; grad_losses f32[num_tasks*6] (device).  Fixed summation orders, no float atomics. */
size_t dbev_centerhead_loss_workspace_bytes(int B, int num_classes_total, int num_tasks, int H, int W);
int dbev_centerhead_loss_forward(const float* const* heads_host, const int32_t* nhwc_flags_host,
                                 float* const* sig_out_host, const int32_t* task_num_classes_host, int num_tasks, int B,
                                 int H, int W, int max_objs, const float* heatmap, const float* anno_box,
                                 const long long* ind, const unsigned char* mask, const float* code_weights_host,
                                 float loss_weight_bbox, float loss_weight_cls, float* losses, float* avg_factors,
                                 void* workspace, size_t workspace_bytes, dbevStream_t stream);
int dbev_centerhead_loss_backward(const float* const* heads_host, const int32_t* nhwc_flags_host,
                                  float* const* grad_heads_host, const int32_t* task_num_classes_host, int num_tasks,
                                  int B, int H, int W, int max_objs, const float* heatmap, const float* anno_box,
                                  const long long* ind, const unsigned char* mask, const float* code_weights_host,
                                  float loss_weight_bbox, float loss_weight_cls, const float* avg_factors,
                                  const float* grad_losses, dbevStream_t stream);

/* 3x3 convolution, stride 1, padding 1, with a skinny output (Cout <= 3), channels-last fp32: the final layer of every
 * CenterHead branch (mmdet3d/models/dense_heads/centerpoint_head.py:17-130, SeparateHead: 64 -> 1..3 channels).  HBM-bound
 * streaming work that MIOpen's implicit-GEMM tiles serve at 0.6 TB/s (N padded from 1-3 to 16).
 *   x_nhwc f32[N,H,W,Cin] (Cin/4 in {8,16,32,64}); weight_ohwi f32[Cout,3,3,Cin] = torch weight.permute(0,2,3,1);
 *   bias f32[Cout] or NULL; y_nhwc f32[N,H,W,Cout].
 * backward: grad_x_nhwc (may be NULL) fully written; grad_weight_ohwi f32[Cout,3,3,Cin] + grad_bias f32[Cout] (both or
 * neither) by fixed-order two-stage reductions (no float atomics; workspace from dbev_skinny_conv3x3_workspace_bytes). */
size_t dbev_skinny_conv3x3_workspace_bytes(int Cin, int Cout);
int dbev_skinny_conv3x3_forward(const float* x_nhwc, const float* weight_ohwi, const float* bias, float* y_nhwc, int N,
                                int Cin, int H, int W, int Cout, dbevStream_t stream);
int dbev_skinny_conv3x3_backward(const float* grad_y_nhwc, const float* x_nhwc, const float* weight_ohwi,
                                 float* grad_x_nhwc, float* grad_weight_ohwi, float* grad_bias, int N, int Cin, int H, int W,
                                 int Cout, void* workspace, size_t workspace_bytes, dbevStream_t stream);
/* All final convolutions of one CenterHead branch group in one launch each way (grid.y = branch): the 36 hidden maps of SeparateHead
 * (centerpoint_head.py:83-101) live in a few wide NHWC tensors (x_pitch / grad_x_pitch = their channel counts, multiples of 4, pointers
 * 16-byte aligned).  Branch b reads channels [b*Cin, (b+1)*Cin) of the wide map in place and writes its slice of the shared gradient; weights_packed f32[n_branch,3,3,3,Cin] = every branch's weight.permute(0,2,3,1) padded with zero rows to 3 output
 * channels, bias_packed f32[n_branch,3]; y_nhwc / grad_y_nhwc: HOST arrays of n_branch device pointers to f32[N,H,W,cout[b]];
 * cout: HOST int32[n_branch] in 1..3; n_branch <= 48.  backward writes every slice of grad_x (pitch grad_x_pitch), the packed
 * weight / bias gradients (rows >= cout[b] are zero) with fixed-order reductions; workspace from
 * dbev_skinny_conv3x3_multi_workspace_bytes. */
size_t dbev_skinny_conv3x3_multi_workspace_bytes(int Cin, int n_branch);
int dbev_skinny_conv3x3_multi_forward(const float* x_nhwc, long long x_pitch, const float* weights_packed,
                                      const float* bias_packed, float* const* y_nhwc, const int32_t* cout, int n_branch, int N,
                                      int Cin, int H, int W, dbevStream_t stream);
int dbev_skinny_conv3x3_multi_backward(const float* const* grad_y_nhwc, const float* x_nhwc, long long x_pitch,
                                       const float* weights_packed, float* grad_x_nhwc, long long grad_x_pitch,
                                       float* grad_weights_packed, float* grad_bias_packed, const int32_t* cout, int n_branch,
                                       int N, int Cin, int H, int W, void* workspace, size_t workspace_bytes, dbevStream_t stream);

/* Training-mode BatchNorm2d fused with the residual add and ReLU that follow it (channels-last fp32):
 *   y = relu( (x - mean_batch) / sqrt(var_batch + eps) * gamma + beta  [+ residual] )
 * = torch.nn.functional.batch_norm(training=True) [+ add] [+ relu] as the reference's dense blocks chain them
 * (mmdet ResNet Bottleneck conv-bn-relu / conv-bn-add-relu; mmdet3d/models/bricks/res_block.py:11-100; mmcv
 * ConvModule).  x, residual, y, grad_*: f32[M, C] with M = N*H*W rows (NHWC); C % 4 == 0 and C/4 a power of two
 * (<= 256) or a multiple of 256.  running_mean/var (may both be NULL) are updated in place with `momentum`,
 * running_var with the unbiased variance; num_batches_tracked (i64 scalar, may be NULL) is incremented.  save_mean, save_invstd f32[C] and save_scale_shift f32[2*C] are
 * outputs of forward / inputs of backward.  backward: y is only read when relu != 0 and grad_residual != NULL
 * (otherwise the gate is recomputed from x); grad_residual (NULL if there was no residual) receives
 * dy * [y > 0].  All reductions have a fixed order (no float atomics).
 * workspace: dbev_bn_act_workspace_bytes(M, C) for forward, + 12*C bytes for backward. */
size_t dbev_bn_act_workspace_bytes(long long M, int C);
int dbev_bn_act_train_forward(const float* x, const float* residual, const float* gamma, const float* beta,
                              float* running_mean, float* running_var, long long* num_batches_tracked,
                              float momentum, float eps, int relu, float* y, float* save_mean, float* save_invstd,
                              float* save_scale_shift, long long M, int C, void* workspace, size_t workspace_bytes,
                              dbevStream_t stream);
/* the same with the statistics pass replaced by partial sums the PRODUCER of x already took (dbev_conv1x1_forward's epilogue):
 * stats_partial f32[partial_rows, 2, C] = per row (sum x, sum x^2) per channel over a disjoint share of the M rows; NULL = as above.
 * y == NULL (then residual must be NULL): statistics only -- save_mean / save_invstd / save_scale_shift and the running statistics
 * are produced, nothing is applied (the consumer normalises inside its own kernel: dbev_depth_head_forward) */
int dbev_bn_act_train_forward_pre(const float* x, const float* residual, const float* gamma, const float* beta,
                                  float* running_mean, float* running_var, long long* num_batches_tracked, float momentum,
                                  float eps, int relu, float* y, float* save_mean, float* save_invstd, float* save_scale_shift,
                                  long long M, int C, const float* stats_partial, int partial_rows, void* workspace,
                                  size_t workspace_bytes, dbevStream_t stream);
/* eval mode (running statistics, no gradient): y = relu(x * scale + shift [+ residual]); workspace >= 8*C bytes */
int dbev_bn_act_infer(const float* x, const float* residual, const float* gamma, const float* beta,
                      const float* running_mean, const float* running_var, float eps, int relu, float* y,
                      long long M, int C, void* workspace, size_t workspace_bytes, dbevStream_t stream);
/* the same in two parts for norms whose parameters and running statistics do not change between calls (a frozen teacher): the
 * coefficients (scale | shift, f32[2C], bit-identical to what dbev_bn_act_infer derives) once, then only the apply pass per call */
int dbev_bn_infer_coef(const float* gamma, const float* beta, const float* running_mean, const float* running_var, float eps, int C,
                       float* scale_shift, dbevStream_t stream);
int dbev_bn_act_apply(const float* x, const float* residual, const float* scale_shift, int relu, float* y, long long M, int C,
                      dbevStream_t stream);
int dbev_bn_act_backward(const float* grad_y, const float* x, const float* y, const float* gamma,
                         const float* save_mean, const float* save_invstd, const float* save_scale_shift,
                         int relu, float* grad_x, float* grad_residual, float* grad_gamma, float* grad_beta,
                         long long M, int C, void* workspace, size_t workspace_bytes, dbevStream_t stream);
/* the same with the incoming gradient given as TWO addends (grad_y2 may be NULL): the output of a residual block feeds the next block's
 * first convolution and its identity branch, autograd would sum their two gradients in a pass of its own (read 2, write 1); the
 * reduction and the dx pass add them on the way in instead */
int dbev_bn_act_backward2(const float* grad_y, const float* grad_y2, const float* x, const float* y, const float* gamma,
                          const float* save_mean, const float* save_invstd, const float* save_scale_shift, int relu,
                          float* grad_x, float* grad_residual, float* grad_gamma, float* grad_beta, long long M, int C,
                          void* workspace, size_t workspace_bytes, dbevStream_t stream);

/* Two training-mode BatchNorm2d layers feeding one add (+ ReLU):  y = [relu]( bn(x) + bn_d(xd) )  -- the main and the
 * `downsample` branch of a stage-first residual block (mmdet ResNet Bottleneck / BasicBlock: `identity = self.downsample(x)`,
 * `out += identity`, `relu`).  Same semantics, layouts and constraints as dbev_bn_act_*; the normalised copy of the branch and the
 * gated gradient dy * [y > 0] are never materialised (forward 4 reads + 1 write of [M, C], backward 8 reads + 2 writes; the two
 * chained single-norm calls take 5 + 2 and 11 + 3).  grad_beta_d == grad_beta (both are sum dz).
 * workspace: dbev_bn_dual_workspace_bytes(M, C) for either direction. */
size_t dbev_bn_dual_workspace_bytes(long long M, int C);
int dbev_bn_dual_train_forward(const float* x, const float* xd, const float* gamma, const float* beta, float* running_mean,
                               float* running_var, long long* num_batches_tracked, float momentum, float eps,
                               const float* gamma_d, const float* beta_d, float* running_mean_d, float* running_var_d,
                               long long* num_batches_tracked_d, float momentum_d, float eps_d, int relu, float* y,
                               float* save_mean, float* save_invstd, float* save_scale_shift, float* save_mean_d,
                               float* save_invstd_d, float* save_scale_shift_d, long long M, int C, void* workspace,
                               size_t workspace_bytes, dbevStream_t stream);
int dbev_bn_dual_train_forward_pre(const float* x, const float* xd, const float* gamma, const float* beta, float* running_mean,
                                   float* running_var, long long* num_batches_tracked, float momentum, float eps,
                                   const float* gamma_d, const float* beta_d, float* running_mean_d, float* running_var_d,
                                   long long* num_batches_tracked_d, float momentum_d, float eps_d, int relu, float* y,
                                   float* save_mean, float* save_invstd, float* save_scale_shift, float* save_mean_d,
                                   float* save_invstd_d, float* save_scale_shift_d, long long M, int C,
                                   const float* stats_partial, int partial_rows, const float* stats_partial_d, int partial_rows_d,
                                   void* workspace, size_t workspace_bytes, dbevStream_t stream);
int dbev_bn_dual_backward(const float* grad_y, const float* x, const float* xd, const float* y, const float* gamma,
                          const float* save_mean, const float* save_invstd, const float* gamma_d, const float* save_mean_d,
                          const float* save_invstd_d, int relu, float* grad_x, float* grad_xd, float* grad_gamma,
                          float* grad_beta, float* grad_gamma_d, float* grad_beta_d, long long M, int C, void* workspace,
                          size_t workspace_bytes, dbevStream_t stream);
int dbev_bn_dual_backward2(const float* grad_y, const float* grad_y2, const float* x, const float* xd, const float* y,
                           const float* gamma, const float* save_mean, const float* save_invstd, const float* gamma_d,
                           const float* save_mean_d, const float* save_invstd_d, int relu, float* grad_x, float* grad_xd,
                           float* grad_gamma, float* grad_beta, float* grad_gamma_d, float* grad_beta_d, long long M, int C,
                           void* workspace, size_t workspace_bytes, dbevStream_t stream);

/* ------------------------------------------------------------------------------------
 * Round 5: the ReLU gate of a residual norm as ONE BYTE per four channels.  The backward of `relu(bn(x) + residual)` reads the saved
 * output only for its sign; the 4x-planes-wide block outputs of the bottlenecks (mmdet3d/models/bricks/res_block.py:102-230,
 * `out += identity; out = relu(out)`) are the largest tensors of the step, and both backward passes read them.
 *   dbev_bn_act_train_forward_mask / dbev_bn_dual_train_forward_mask: as the *_pre entries, plus `relu_mask` u8[M * C / 4]
 *     (NULL: not written): byte i holds the gate bits (bit j = channel 4 i' + j of that pixel positive before the ReLU) of float4 i.
 *   dbev_bn_act_backward3 / dbev_bn_dual_backward3: as the *_backward2 entries with `y` + `y_is_mask`: y_is_mask != 0 -> `y` is that
 *     byte array (16 x less traffic than the saved output); 0 -> the saved output, as before.  Results are bit-identical either way.
 * ---------------------------------------------------------------------------------- */
int dbev_bn_act_train_forward_mask(const float* x, const float* residual, const float* gamma, const float* beta, float* running_mean,
                                   float* running_var, long long* num_batches_tracked, float momentum, float eps, int relu, float* y,
                                   float* save_mean, float* save_invstd, float* save_scale_shift, long long M, int C,
                                   const float* stats_partial, int partial_rows, unsigned char* relu_mask, void* workspace,
                                   size_t workspace_bytes, dbevStream_t stream);
int dbev_bn_act_backward3(const float* grad_y, const float* grad_y2, const float* x, const void* y, int y_is_mask, const float* gamma,
                          const float* save_mean, const float* save_invstd, const float* save_scale_shift, int relu, float* grad_x,
                          float* grad_residual, float* grad_gamma, float* grad_beta, long long M, int C, void* workspace,
                          size_t workspace_bytes, dbevStream_t stream);
int dbev_bn_dual_train_forward_mask(const float* x, const float* xd, const float* gamma, const float* beta, float* running_mean,
                                    float* running_var, long long* num_batches_tracked, float momentum, float eps,
                                    const float* gamma_d, const float* beta_d, float* running_mean_d, float* running_var_d,
                                    long long* num_batches_tracked_d, float momentum_d, float eps_d, int relu, float* y,
                                    float* save_mean, float* save_invstd, float* save_scale_shift, float* save_mean_d,
                                    float* save_invstd_d, float* save_scale_shift_d, long long M, int C, const float* stats_partial,
                                    int partial_rows, const float* stats_partial_d, int partial_rows_d, unsigned char* relu_mask,
                                    void* workspace, size_t workspace_bytes, dbevStream_t stream);
int dbev_bn_dual_backward3(const float* grad_y, const float* grad_y2, const float* x, const float* xd, const void* y, int y_is_mask,
                           const float* gamma, const float* save_mean, const float* save_invstd, const float* gamma_d,
                           const float* save_mean_d, const float* save_invstd_d, int relu, float* grad_x, float* grad_xd,
                           float* grad_gamma, float* grad_beta, float* grad_gamma_d, float* grad_beta_d, long long M, int C,
                           void* workspace, size_t workspace_bytes, dbevStream_t stream);

/* LiDAR sweep -> per-camera sparse depth maps, the depth supervision BEVDepth's img_inputs carry as their last element.
 * Replaces the loader transform PointToMultiViewDepth.__call__ / points2depthmap (mmdet3d/datasets/pipelines/loading.py:18-61).
 *   points f32[n_points, n_feats] (x, y, z first, lidar frame);
 *   cam_mats f32[n_cams, 24] per camera: inverse(rots @ inverse(intrins)) row-major [9], post_rots [9], trans [3], post_trans [3];
 *   depth_maps f32[n_cams, height / downsample, width / downsample] (every element written: 0 where no point lands).
 * A pixel takes the smallest depth in [depth_min, depth_max) of the points rounding onto it (integer atomicMin on the
 * bit pattern: exact, order-independent); the reference reaches the same pixel value through argsort(pixel + depth / 100)
 * in float32, whose ties below ~0.006 m it resolves arbitrarily. */
int dbev_points_to_depth_maps(const float* points, int n_points, int n_feats, const float* cam_mats, int n_cams, int height,
                              int width, int downsample, float depth_min, float depth_max, float* depth_maps,
                              dbevStream_t stream);

/* bev_pool helpers of the Python surface (mmdet3d/ops/bev_pool/bev_pool.py:64-97): the cell lists straight from the
 * int64 coordinates the reference's callers pass (no int32 conversion pass), and the [B, C, S] -> [B, S, C] transpose
 * for an out_grad that arrives in the reference's contiguous [B, C, D, H, W] layout (S = D*H*W). */
int dbev_bev_pool_prepare_i64(const long long* coords, int n_points, int B, int D, int H, int W, int32_t* point_cell,
                              int32_t* cell_start, int32_t* cell_points, int32_t* n_kept_out, int32_t* hot_cells,
                              int32_t* n_hot_out, void* workspace, size_t workspace_bytes, dbevStream_t stream);
int dbev_transpose_bcs_to_bsc(const float* in_bcs, float* out_bsc, int B, int C, int S, dbevStream_t stream);

/* ------------------------------------------------------------------------------------
 * Fused student adaptation (1x1 convolution) + masked-MSE reduction of the FGD loss, 'head' position
 * (replaces the sequence nn.Conv2d(Cs, Ct, 1) -> mean_c|.| -> three masked sums of
 *  mmdet3d/models/detectors/bevdet_distill.py:232-234,1003-1004,1089-1092,1253-1262,1272-1287).
 * Channels-last operands: x [B*HW, Cs], weight [Ct, Cs] (the conv weight [Ct, Cs, 1, 1]), bias [Ct],
 * teacher [B*HW, Ct], channel_weight [B, Ct] or NULL (the fp term's channel attention).
 * forward: diff = (x W^T + bias) - teacher  -> diff_nhwc [B*HW, Ct]   (the adapted tensor itself is never stored)
 *          maps [S][4][B*HW], S = dbev_adapt_mse_map_slices(Cs, Ct) channel slices to be summed by the caller:
 *          [0] sum_c diff^2, [1] sum_c channel_weight diff^2, [2] sum_c |s|, [3] sum_c s.
 * Cs % 32 == 0 and Ct % 32 == 0 (map_slices returns 0 otherwise).  fp32 MFMA (v_mfma_f32_32x32x2_f32).
 * backward_ds: ds = 2 diff (grad_e[p] + grad_efp[p] channel_weight[b, c]) + grad_pool[p] / Ct
 *          (grad_efp / grad_pool / channel_weight may be NULL); the convolution gradients are ordinary GEMMs on ds.
 * ---------------------------------------------------------------------------------- */
int dbev_adapt_mse_map_slices(int Cs, int Ct);
int dbev_adapt_mse_forward(const float* x_nhwc, const float* weight, const float* bias, const float* teacher_nhwc,
                           const float* channel_weight, int B, int HW, int Cs, int Ct, float* diff_nhwc, float* maps,
                           dbevStream_t stream);
int dbev_adapt_mse_backward_ds(const float* diff_nhwc, const float* grad_e, const float* grad_efp,
                               const float* grad_pool, const float* channel_weight, int B, int HW, int Ct,
                               float* ds_nhwc, dbevStream_t stream);

/* ------------------------------------------------------------------------------------
 * Multi-scale deformable attention (replaces mmcv-full 1.6.0 `_ext.ms_deform_attn_forward/backward`, un-vendored;
 * call site mmdet3d/models/transformer_modules/multi_scale_deformable_attn_function.py:10-12,42-49,70-82).
 * value [B, S, NH, D] (S = sum_l H_l*W_l, D % 4 == 0, D/4 a power of two <= 64), spatial shapes (h, w) and level start
 * offsets as HOST int32 arrays (L <= 8), sampling_loc [B, Q, NH, L, P, 2] normalised (x, y), attn_weight
 * [B, Q, NH, L, P]  ->  out [B, Q, NH*D].  Bilinear sampling at loc * (W, H) - 0.5 with zero padding.
 * backward fills grad_value [B, S, NH, D] (every element written; deterministic gather, no float atomics),
 * grad_sampling_loc and grad_attn_weight; workspace of dbev_msda_backward_workspace_bytes (0 = unsupported size).
 * ---------------------------------------------------------------------------------- */
size_t dbev_msda_backward_workspace_bytes(int B, int S, int NH, int Q, int L, int P);
int dbev_msda_forward(const float* value, const int32_t* spatial_shapes_hw_host, const int32_t* level_start_host,
                      const float* sampling_loc, const float* attn_weight, int B, int S, int NH, int D, int Q, int L,
                      int P, float* out, dbevStream_t stream);
int dbev_msda_backward(const float* value, const int32_t* spatial_shapes_hw_host, const int32_t* level_start_host,
                       const float* sampling_loc, const float* attn_weight, const float* grad_out, int B, int S, int NH,
                       int D, int Q, int L, int P, float* grad_value, float* grad_sampling_loc, float* grad_attn_weight,
                       void* workspace, size_t workspace_bytes, dbevStream_t stream);

/* ------------------------------------------------------------------------------------
 * MaxPool2d(kernel 3, stride 2, padding 1) of the ResNet stem, channels-last (replaces ATen max_pool2d_with_indices(_backward) behind
 * mmdet ResNet `self.maxpool`, reached from bevdet_distill_more.py image_encoder -> self.img_backbone).  x_nhwc [N, H, W, C],
 * C % 4 == 0; y [N, Ho, Wo, C], Ho = (H - 1) / 2 + 1; winner u8 [N, Ho, Wo, C] = winning tap 0..8 (first maximum in scan order, a NaN
 * wins: ATen's rule).  backward: gather over the <= 2 x 2 windows that cover an input pixel, no atomics.
 * ---------------------------------------------------------------------------------- */
int dbev_maxpool3x3s2_forward(const float* x_nhwc, int N, int H, int W, int C, float* y_nhwc, unsigned char* winner,
                              dbevStream_t stream);
int dbev_maxpool3x3s2_backward(const float* grad_y_nhwc, const unsigned char* winner, int N, int H, int W, int C, float* grad_x_nhwc,
                               dbevStream_t stream);
/* Round 5: the stem's norm -> ReLU -> max pooling as ONE pass over the convolution's output (mmdet ResNet.forward: `x = self.conv1(x);
 * x = self.norm1(x); x = self.relu(x); x = self.maxpool(x)`): y = maxpool3x3s2(relu(x * scale + shift)), scale_shift f32[2 C] = the
 * coefficient row dbev_bn_act_train_forward_mask(..., y = NULL) leaves in save_scale_shift (training: batch statistics) or
 * dbev_bn_infer_coef (eval).  winner as dbev_maxpool3x3s2_forward (same tie rule on the same values); the backward is
 * dbev_maxpool3x3s2_backward followed by dbev_bn_act_backward3 (gate recomputed from x). */
int dbev_norm_relu_maxpool3x3s2_forward(const float* x_nhwc, const float* scale_shift, int N, int H, int W, int C, float* y_nhwc,
                                        unsigned char* winner, dbevStream_t stream);

/* Round 5: the backward of that fused forward in two passes -- the pooling's gradient gather (dbev_maxpool3x3s2_backward) happens inside
 * the statistics pass and the dx pass of the norm's backward, the 4 x larger gradient of the rectified map is never written:
 * grad_pooled f32[N, Ho, Wo, C] + winner (of the forward) + x (the convolution's output) -> grad_x f32[N, H, W, C], grad_gamma /
 * grad_beta f32[C].  save_* / save_scale_shift: what dbev_bn_act_train_forward_mask left.  C % 4 == 0 and 256 % (C / 4) == 0. */
size_t dbev_stem_pool_norm_backward_workspace_bytes(int N, int H, int W, int C);
int dbev_stem_pool_norm_backward(const float* grad_pooled, const unsigned char* winner, const float* x, const float* gamma,
                                 const float* save_mean, const float* save_invstd, const float* save_scale_shift, int N, int H, int W,
                                 int C, float* grad_x, float* grad_gamma, float* grad_beta, void* workspace, size_t workspace_bytes,
                                 dbevStream_t stream);

/* ------------------------------------------------------------------------------------
 * Round 5: the ResNet stem convolution, nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False) (mmdet ResNet.
 * _make_stem_layer, `self.conv1`; replaces cuDNN behind it), on the fp32 matrix cores (csrc/stem.hip).  x_nhwc f32[N, H, W, 3],
 * weight f32[64][7][7][3] (the channels-last memory of the [64, 3, 7, 7] parameter), z_nhwc f32[N, Ho, Wo, 64] with
 * Ho = (H - 1) / 2 + 1, Wo likewise.  H, W >= 7.
 *   forward: stats_partial f32[dbev_stem7x7s2_stats_rows(N, H, W)][2][64] = per workgroup the channel sums of z and z^2
 *     (bn_finalize's partial-row layout, as the other convolutions' epilogues); NULL: not written.
 *   backward_weight: grad_weight f32[64][7][7][3], every element written; fixed summation order (bit-reproducible).
 *   workspace: dbev_stem7x7s2_workspace_bytes(N, H, W) bytes for either entry (0: unsupported geometry).
 * The data gradient (the image's) is not provided: the image has none in the reference's training step.
 * ---------------------------------------------------------------------------------- */
int dbev_stem7x7s2_stats_rows(int N, int H, int W);
long long dbev_stem7x7s2_workspace_bytes(int N, int H, int W);
int dbev_stem7x7s2_forward(const float* x_nhwc, const float* weight, int N, int H, int W, float* z_nhwc, float* stats_partial,
                           void* workspace, size_t workspace_bytes, dbevStream_t stream);
int dbev_stem7x7s2_backward_weight(const float* x_nhwc, const float* grad_z_nhwc, int N, int H, int W, float* grad_weight,
                                   void* workspace, size_t workspace_bytes, dbevStream_t stream);

/* ------------------------------------------------------------------------------------
 * fp32 GEMM of a 1x1 convolution on the BF16 matrix cores at fp32 accuracy ("bf16x6": every operand split into three bf16 values,
 * six exact partial products per product, fp32 accumulation; csrc/gemm_bf6.hip).  Replaces cuDNN behind nn.Conv2d(k=1) of
 * mmdet3d/models/bricks/res_block.py:102-230 / necks/fpn.py:10-204 like dbev_gemm1x1_forward does.
 *   y[M, N] = x[M, K] * weight[N, K]^T,  x rows of x_row_stride floats, y row-major; K % 64 == 0, N % 64 == 0, M any
 *   positive row count (round 6: the last 128-row block goes through bounds-checked buffer descriptors -- rows past M load zeros, their
 *   stores are dropped; the weight gradient likewise reads pixels past M as zeros).
 * dbev_gemm_bf16x6_pack splits the weight (element (n, k) at weight[n * stride_n + k * stride_k]: the transposed view serves the data
 * gradient) into `packed` (dbev_gemm_bf16x6_packed_bytes(N, K) bytes; 0: unsupported shape) once per weight version.  tile_n: columns
 * of a workgroup tile, the SAME value for the pack and the launches that use it: 0 = 128 when N % 128 == 0 else 64; 64 doubles the
 * workgroups of a layer with few rows.
 * ---------------------------------------------------------------------------------- */
long long dbev_gemm_bf16x6_packed_bytes(int N, int K);
int dbev_gemm_bf16x6_pack(const float* weight, long long stride_n, long long stride_k, int N, int K, int tile_n, void* packed,
                          dbevStream_t stream);
/* the same launch with the BatchNorm statistics of y in its epilogue: stats_partial f32[dbev_gemm_bf16x6_stats_rows(M)][2][N] = per
 * 128-row block the column sums of y and y^2 (bn_finalize's partial-row layout: dbev_bn_act_train_forward_pre(..., pre rows)); NULL: none */
int dbev_gemm_bf16x6_stats_rows(long long M);
/* Round 5: nn.Conv2d(C, Co, 3, stride 2, padding 1, bias=False) -- `conv2` of a stage-first ResNet bottleneck (mmdet ResNet / mmdet3d
 * bricks/res_block.py:102-230; cuDNN behind it) -- as an IMPLICIT bf16x6 GEMM: rows = output pixels, reduction over (ky, kx, c).
 * x_nhwc f32[N, H, W, C], y_nhwc f32[N, H/2, W/2, Co]; `packed` = dbev_gemm_bf16x6_pack of the filter's channels-last memory
 * [Co][3][3][C] taken as the matrix [Co][9 C] (stride_n = 9 C, stride_k = 1, N = Co, K = 9 C) with the same tile_n;
 * stats_partial f32[dbev_gemm_bf16x6_stats_rows(N H/2 W/2)][2][Co] or NULL.  dbev_conv3x3s2_bf16x6_ok: H, W even, C a power of two
 * >= 64, Co % 64 == 0.  (Forward only: both gradients stay with the library this round.) */
int dbev_conv3x3s2_bf16x6_ok(int N, int H, int W, int C, int Co);
int dbev_conv3x3s2_bf16x6_forward_stats(const float* x_nhwc, const void* packed, float* y_nhwc, float* stats_partial, int N, int H, int W,
                                        int C, int Co, int tile_n, dbevStream_t stream);
int dbev_gemm_bf16x6_forward_stats(const float* x, const void* packed, float* y, float* stats_partial, long long M, int K, int N,
                                   int x_row_stride, int tile_n, dbevStream_t stream);
/* both orientations of a [Cout, Cin] filter (element (o, c) at weight[o * stride_o + c * stride_c]) in ONE launch: the forward planes
 * (N = Cout, K = Cin, tile_fwd) and the data gradient's (N = Cin, K = Cout, tile_dgrad) */
int dbev_gemm_bf16x6_pack_pair(const float* weight, long long stride_o, long long stride_c, int Cout, int Cin, int tile_fwd,
                               void* packed_fwd, int tile_dgrad, void* packed_dgrad, dbevStream_t stream);
/* Round 6: y = x weight^T + bias[N] with the bias added in the kernel's epilogue (no pass of its own): torch.nn.Linear on [tokens, C] rows
 * (the BEVFormer encoder's projections and FFNs, projects/mmdet3d_plugin/bevformer/modules/: encoder.py, spatial_cross_attention.py, temporal_self_attention.py)
 * and biased 1x1 convolutions. */
int dbev_gemm_bf16x6_forward_bias(const float* x, const void* packed, const float* bias, float* y, long long M, int K, int N,
                                  int x_row_stride, int tile_n, dbevStream_t stream);
int dbev_gemm_bf16x6_forward(const float* x, const void* packed, float* y, long long M, int K, int N, int x_row_stride, int tile_n,
                             dbevStream_t stream);
/* weight gradient of the same layer, grad_weight[Cout, Cin] = sum_m grad_y[m, Cout] * x[m, Cin] (both operands split on the fly,
 * shares of the pixel range merged in a fixed order: bit-reproducible, no zero-fill launch, no atomics); M % 32 == 0,
 * Cin % 64 == 0, Cout % 64 == 0; workspace: dbev_gemm_bf16x6_backward_weight_workspace_bytes bytes (0: unsupported shape). */
size_t dbev_gemm_bf16x6_backward_weight_workspace_bytes(long long M, int Cin, int Cout, int x_row_stride);
int dbev_gemm_bf16x6_backward_weight(const float* x, const float* grad_y, float* grad_weight, long long M, int Cin, int Cout,
                                     int x_row_stride, void* workspace, size_t workspace_bytes, dbevStream_t stream);

/* ------------------------------------------------------------------------------------
 * Round 5: the per-step re-packing of EVERY trainable layer's weights in one launch per kernel family (the reference has no counterpart:
 * cuDNN consumes `nn.Conv2d.weight` as it is; here the Winograd kernels read G g G^T in their consumption order and the bf16x6 GEMMs the
 * three bf16 planes of a 1x1 filter, both re-derived after every optimizer step -- 72 launches of 6-8 us per step before).
 * `jobs_device`: device array of n_jobs dbevPackJob; a job = the arguments of dbev_wino_filter_pack_pair (kind_a / kind_b = forward /
 * data-gradient kernel codes, 0: skip) resp. dbev_gemm_bf16x6_pack_pair (so / sc = strides of the [Cout, Cin] view, kind_a / kind_b =
 * tile width 64 / 128 of the forward / data-gradient planes, 0: skip).  max_pairs = max Cin * Cout over the jobs; max_units = max
 * Cin * Cout / 8.  The buffers are the ones the single-layer entries filled before (same sizes, same contents).
 * ---------------------------------------------------------------------------------- */
typedef struct dbevPackJob {
  const float* weight;
  long long so, sc, sa, sb;
  int Cout, Cin, kind_a, kind_b;
  void* out_a;
  void* out_b;
} dbevPackJob;
int dbev_wino_filter_pack_multi(const dbevPackJob* jobs_device, int n_jobs, long long max_pairs, dbevStream_t stream);
int dbev_gemm_bf16x6_pack_multi(const dbevPackJob* jobs_device, int n_jobs, long long max_units, dbevStream_t stream);

/* ------------------------------------------------------------------------------------
 * Stride-2 1x1 convolutions as subsample + GEMM (the `downsample` branch of a stage-first bottleneck: nn.Conv2d(k = 1, stride = 2),
 * mmdet ResNet make_res_layer / mmdet3d/models/bricks/res_block.py:102-230; replaces the library's strided implicit GEMM).
 *   dbev_subsample2_nhwc:      y[n, ho, wo, :] = x[n, 2 ho, 2 wo, :]            x f32[N, H, W, C] channels-last -> y f32[N, H/2, W/2, C]
 *   dbev_upsample2_zero_nhwc:  gx[n, 2 ho, 2 wo, :] = g[n, ho, wo, :], every other pixel of gx = 0 (the data gradient of the
 *                              subsample; one pass, every element of gx written once)   g f32[N, H/2, W/2, C] -> gx f32[N, H, W, C]
 * H, W even, C % 4 == 0 (DBEV_EINVAL otherwise); N, H, W are those of the FULL-resolution tensor in both calls.
 * ---------------------------------------------------------------------------------- */
int dbev_subsample2_nhwc(const float* x, float* y, int N, int H, int W, int C, dbevStream_t stream);
int dbev_upsample2_zero_nhwc(const float* g, float* gx, int N, int H, int W, int C, dbevStream_t stream);

/* ------------------------------------------------------------------------------------
 * Per-channel sum of a channels-last tensor, out[c] = sum over the M = N*H*W rows of x_nhwc[M, C]: the bias gradient of a convolution
 * (replaces ATen's grad_output.sum((0, 2, 3)) inside convolution_backward for the nn.Conv2d(bias=True) layers of
 * mmdet3d/models/necks/fpn.py:77-95, necks/view_transformer_mine.py:288-309, backbones/resnet.py:80-96 and the DCN offset
 * convolution).  Any C (float4 lanes when C % 4 == 0); fixed summation order (per-workgroup partial rows in `workspace`, merged by a
 * second launch): bit-reproducible.  workspace: dbev_channel_sum_workspace_bytes(M, C) bytes (0: unsupported size).
 * ---------------------------------------------------------------------------------- */
size_t dbev_channel_sum_workspace_bytes(long long M, int C);
int dbev_channel_sum_nhwc(const float* x_nhwc, long long M, int C, float* out, void* workspace, size_t workspace_bytes,
                          dbevStream_t stream);

/* ------------------------------------------------------------------------------------
 * Depth head of the BEVDepth view transformer, forward (mmdet3d/models/necks/view_transformer_mine.py:300-309 self.dcn's
 * BatchNorm2d, :326 depth_digit = self.depthnet(depth_feat) (nn.Conv2d(c, D, 1), :291), :327 get_depth_dist = softmax(dim=1);
 * same in detectors/bevdet_distill_more.py:398-416): normalise -> 1x1 convolution -> softmax in one pass.
 *  x_nhwc [M, C] the deformable convolution's output, scale_shift f32[2C] (y = x * scale + shift: save_scale_shift of
 *  dbev_bn_act_train_forward_pre(..., y = NULL, ...), or gamma / sqrt(running_var + eps) etc. in eval mode), weight [N, C], bias [N],
 *  C % 4 == 0, C <= 256, N <= 64.  depth_digit_nhwc / depth_prob_nhwc [M, N]; normalised_nhwc [M, C] or NULL (the 1x1's weight
 *  gradient reads it).  fp32 MFMA, fp32 softmax (expf).
 * ---------------------------------------------------------------------------------- */
int dbev_depth_head_forward(const float* x_nhwc, const float* scale_shift, const float* weight, const float* bias, long long M,
                            int C, int N, float* depth_digit_nhwc, float* depth_prob_nhwc, float* normalised_nhwc,
                            dbevStream_t stream);

/* ------------------------------------------------------------------------------------
 * Sparse 3-D convolution (replaces the bundled spconv v1.x extension `sparse_conv_ext`:
 * mmdet3d/ops/spconv/src/all.cc:21-51 get_indice_pairs_3d / indice_conv_fp32 / fused_indice_conv_fp32;
 * kernels include/spconv/indice.cu.h:24-215, reordering.cu.h:22-128, spconv_ops.h:302-348).
 * indices int32 [n, 4] = (batch, z, y, x); dims / ksize / stride / padding / dilation: HOST int32[3] in (z, y, x) order.
 *  dbev_spconv_outputs    output sites of a strided / padded convolution in ascending cell order (the order the
 *                         reference gets from sorting the unique cell ids) -> out_indices, *n_out_device
 *  dbev_spconv_neighbors  output-stationary rulebook nbr [n_out, K]: nbr[o, k] = input row paired with output o by
 *                         kernel offset k (offsets in (kz, ky, kx) row-major order, as the reference's weight
 *                         [kz, ky, kx, Cin, Cout]) or -1; optional inverse table inv [n_in, K] (SparseInverseConv) and
 *                         the reference's own pair lists indice_pairs [K, 2, n_in] (-1 padded) / indice_pair_num [K].
 *                         For a submanifold convolution pass out_indices = indices, n_out = n_in.
 *  dbev_spconv_forward    out[o, :] = bias + sum_k features[nbr[o, k], :] @ weight[k]   (weight [K, Cin, Cout];
 *                         Cin % 16 == 0, Cout % 16 == 0, Cout <= 128, Cin <= 256; fp32 MFMA; bias may be NULL)
 *  dbev_sparse_to_dense   SparseConvTensor.dense(): [B, C, D, H, W] canvas, zero elsewhere (structure.py:52-62)
 *  transposed != 0        SparseConvTranspose2d / 3d (conv.py:300-346, indice.h:88-140 getValidOutPosTranspose): an input at i
 *                         reaches the outputs i * stride - padding + k * dilation; out_dims = get_deconv_output_size (ops.py:33-44)
 *  dbev_spconv_maxpool_*  SparseMaxPool2d / 3d (pool.py:21-88; indice_maxpool_fp32 / indice_maxpool_backward_fp32 of all.cc:40-47,
 *                         pool_ops.h:26-98, maxpool_cuda.cu:28-230): out[o, c] = max(0, max_k features[nbr[o, k], c]) (the reference
 *                         raises a zero-initialised output); backward: every input equal to its output's value receives that output's
 *                         gradient, summed over the input's pairs in offset order (inv = the inverse table).  C % 4 == 0
 * ---------------------------------------------------------------------------------- */
size_t dbev_spconv_build_workspace_bytes(int n_in, int B, const int32_t* in_dims_host, const int32_t* out_dims_host, int K,
                                         int max_out);
int dbev_spconv_outputs(const int32_t* indices, int n_in, int B, const int32_t* in_dims_host, const int32_t* out_dims_host,
                        const int32_t* ksize_host, const int32_t* stride_host, const int32_t* padding_host,
                        const int32_t* dilation_host, int transposed, int32_t* out_indices, int max_out,
                        int32_t* n_out_device, void* workspace, size_t workspace_bytes, dbevStream_t stream);
int dbev_spconv_neighbors(const int32_t* indices, int n_in, const int32_t* out_indices, int n_out, int B,
                          const int32_t* in_dims_host, const int32_t* out_dims_host, const int32_t* ksize_host,
                          const int32_t* stride_host, const int32_t* padding_host, const int32_t* dilation_host,
                          int transposed, int32_t* nbr, int32_t* inv, int32_t* indice_pairs, int32_t* indice_pair_num,
                          void* workspace, size_t workspace_bytes, dbevStream_t stream);
/* the pair lists alone, from an existing table (what dbev_spconv_neighbors writes when indice_pairs != NULL): built on demand
 * for the backward pass and for get_indice_pairs() (ops.py:46-104); an inference forward never reads them */
size_t dbev_spconv_pair_lists_workspace_bytes(int n_out, int K);
int dbev_spconv_pair_lists(const int32_t* nbr, int n_out, int K, int n_in, int32_t* indice_pairs, int32_t* indice_pair_num,
                           void* workspace, size_t workspace_bytes, dbevStream_t stream);
/* the inverse table alone, from an existing neighbour table (what dbev_spconv_neighbors writes when inv != NULL): inv[r, k] = output
 * row paired with input row r by offset k, or -1.  SparseInverseConv3d (conv.py:143-160) and the data gradient
 * (spconv_ops.h:352-420) read it; built on demand */
int dbev_spconv_inverse_table(const int32_t* nbr, int n_out, int K, int n_in, int32_t* inv, dbevStream_t stream);
int dbev_spconv_maxpool_forward(const float* features, const int32_t* nbr, int n_out, int K, int C, float* out_features,
                                dbevStream_t stream);
int dbev_spconv_maxpool_backward(const float* features, const float* out_features, const float* grad_out, const int32_t* inv,
                                 int n_in, int K, int C, float* grad_in, dbevStream_t stream);
int dbev_spconv_forward(const float* features, const float* weight, const float* bias, const int32_t* nbr, int n_out,
                        int K, int Cin, int Cout, float* out_features, dbevStream_t stream);
/* dbev_spconv_forward with the epilogue of a conv -> eval BatchNorm1d -> (+ residual) -> ReLU chain folded in (the reference's
 * fused_indice_conv + SparseSequential.fused(), modules.py:152-196, and the tail of SparseBasicBlock, sparse_block.py:101-121):
 * out = [relu]( conv * scale + shift [+ residual] );  scale / shift [Cout] (NULL = 1 / 0), residual [n_out, Cout] or NULL */
int dbev_spconv_forward_fused(const float* features, const float* weight, const float* scale, const float* shift,
                              const float* residual, int relu, const int32_t* nbr, int n_out, int K, int Cin, int Cout,
                              float* out_features, dbevStream_t stream);
int dbev_sparse_to_dense(const float* features, const int32_t* indices, int n, int C, int B, int D, int H, int W,
                         float* canvas_ncdhw, dbevStream_t stream);
/* indice_conv_backward (spconv_ops.h:352-420; per offset a gather, two GEMMs and a float-atomic scatter-add in the reference):
 *  dbev_spconv_backward_data    grad_features[i, :] = sum_k grad_out[inverse_table[i, k], :] @ weight[k]^T -- the forward
 *                               gather-GEMM on the table of the opposite direction (dbev_spconv_neighbors' `inv` for a regular
 *                               convolution, its `nbr` for an inverse convolution); weight [K, Cin, Cout] as in the forward,
 *                               Cin <= 128, Cout <= 256, multiples of 16; workspace >= 4 * K * Cin * Cout bytes; no atomics.
 *  dbev_spconv_backward_weight  grad_weight[k] = sum over the pairs p < indice_pair_num[k] of
 *                               features[pairs[k, inverse ? 1 : 0, p], :]^T (x) grad_out[pairs[k, inverse ? 0 : 1, p], :]
 *                               from the reference-format lists (dbev_spconv_pair_lists; row stride `pair_stride`, no list longer
 *                               than n_pairs_max); Cin, Cout in {16, 32, 64, 128}; partial sums merged in a fixed order. */
int dbev_spconv_backward_data(const float* grad_out, const float* weight, const int32_t* inverse_table, int n_in, int K, int Cin,
                              int Cout, float* grad_features, void* workspace, size_t workspace_bytes, dbevStream_t stream);
size_t dbev_spconv_backward_weight_workspace_bytes(int K, int Cin, int Cout, int n_pairs_max);
int dbev_spconv_backward_weight(const float* features, const float* grad_out, const int32_t* indice_pairs,
                                const int32_t* indice_pair_num, int pair_stride, int n_pairs_max, int inverse, int K, int Cin,
                                int Cout, float* grad_weight, void* workspace, size_t workspace_bytes, dbevStream_t stream);

/* ------------------------------------------------------------------------------------
 * Dynamic voxel encoders of the voxel teachers (mmdet3d/models/voxel_encoders/dynamic_voxel_encoder.py; torch op sequences
 * in the reference: boolean row selections, a zero-padded [N, 24] copy, `unique(dim=0)`, scatter_mean, masked divisions).
 *  dbev_range_voxel_coords    `voxelization` :9-13 / `voxelization_virtual` :22-26,57: coors i32[N, 3] = (z, y, x) =
 *                             trunc((p - range_min) / voxel_size) for the points inside the CLOSED range, (-1,-1,-1) for the
 *                             others (and, with virtual_classes != 0, for rows whose tag column F-2 is none of 1 / 0 / -1:
 *                             the reference cannot represent them).  Row convention of dbev_dynamic_scatter_prepare, which
 *                             then yields the voxels in `unique`'s order.  pc_range_host f32[6], voxel_size_host f32[3]: HOST.
 *  dbev_virtual_voxel_reduce  `voxelization_virtual` :27-67 on the point lists of dbev_dynamic_scatter_prepare:
 *                             points f32[N, 17] -> voxels f32[M, 23] (real columns 0..5, painted / virtual columns 6..22,
 *                             mixed voxels rescaled by the real fraction); per-column sums in the reference's order. */
int dbev_range_voxel_coords(const float* points, int num_points, int num_feats, const float* pc_range_host,
                            const float* voxel_size_host, int virtual_classes, int32_t* coors, dbevStream_t stream);
int dbev_virtual_voxel_reduce(const float* points, const int32_t* voxel_point_start, const int32_t* voxel_point_list,
                              float* voxels, int num_voxels, dbevStream_t stream);

/* ------------------------------------------------------------------------------------
 * 1x1 convolution of the bottleneck blocks with the batch statistics of the BatchNorm that follows it in the epilogue
 * (mmdet3d/models/bricks/res_block.py:102-230 conv1 / conv3 -> norm; mmdet ResNet `downsample`: nn.Conv2d(k=1) -> BatchNorm2d; the
 * reference runs cuDNN's convolution, then the norm layer re-reads the whole output for mean / variance).
 *   x_nhwc f32[M, x_row_stride >= Cin] (M = N*H*W pixels, channels-last), weight f32[Cout, Cin] (OIHW with a 1x1 kernel, no bias),
 *   y_nhwc f32[M, Cout]; Cin % 32 == 0, Cout % 32 == 0.
 *   stats_partial (may be NULL) f32[rows, 2, Cout], rows = dbev_conv1x1_stats_rows(M, Cin, Cout): per persistent workgroup the sums
 *   of y and y^2 over the pixels it computed -- the partial-row layout dbev_bn_act_from_partials merges (fixed order, no atomics).
 * ---------------------------------------------------------------------------------- */
int dbev_conv1x1_stats_rows(long long M, int Cin, int Cout);
int dbev_conv1x1_forward(const float* x_nhwc, const float* weight, float* y_nhwc, float* stats_partial, long long M, int Cin,
                         int Cout, int x_row_stride, dbevStream_t stream);

/* ------------------------------------------------------------------------------------
 * 3x3 / stride 1 / pad 1 convolution as Winograd F(2x2, 3x3) on the fp32 matrix cores (csrc/wino.hip).  Replaces the cuDNN /
 * MIOpen convolution behind `nn.Conv2d(k=3, s=1, p=1)` of the reference's dense blocks -- the 3x3 layer of the ResNet bottlenecks
 * and BasicBlocks (mmdet3d/models/bricks/res_block.py:11-230, backbones/resnet.py:13-62), FPN_LSS (necks/lss_fpn.py:30-60), the
 * CenterHead shared / branch convolutions (dense_heads/centerpoint_head.py:17-130), SECOND (backbones/second.py:60-78) -- forward
 * and data gradient (the same kernel on grad_y with rotated / transposed filters).
 *   dbev_wino_filter_floats(K, J)   size of the packed transformed filters for K reduction and J output channels (0: unsupported)
 *   dbev_wino_filter_pack           weight element (co, c, a, b) at co*so + c*sc + a*sa + b*sb (element strides: OIHW or
 *                                   channels-last) -> packed = G g G^T in the kernels' consumption orders: two formats one after the
 *                                   other (k groups of 4 for wino_fwd3, of 8 for wino_fwd; the second only when K % 8 == 0).
 *                                   flags bit 0: pack the filters of the data gradient (K = Cout, J = Cin, taps rotated by 180
 *                                   degrees); bits 1-2: 0 both formats, 1 only wino_fwd3's, 2 only wino_fwd's (the caller asked
 *                                   dbev_wino_conv3x3_forward_kernel which kernel its layer gets)
 *   dbev_wino_filter_pack_pair      the forward AND the data-gradient filters of one layer in one launch, each only in the format of
 *                                   the kernel its direction gets: fwd_kernel / dgrad_kernel = dbev_wino_conv3x3_forward_kernel(...) of
 *                                   (N, H, W, Cin, Cout) / (N, H, W, Cout, Cin), 0 = direction not wanted; buffers sized by
 *                                   dbev_wino_filter_floats(Cin, Cout) / (Cout, Cin)
 *   dbev_wino_conv3x3_forward       x_nhwc f32[N, H, W, Cin] -> y_nhwc f32[N, H, W, Cout] (+ bias f32[Cout] or NULL); H, W even,
 *                                   Cin % 4 == 0, Cout % 64 == 0, fewer than 2^31 elements per tensor, H * W * Cin * 4 < 2^31.  stats_partial (may be NULL)
 *                                   f32[rows, 2, Cout], rows = dbev_wino_conv3x3_stats_rows(...): per tile block the sums of y and
 *                                   y^2 per channel (bias included), the partial-row layout of dbev_bn_act_train_forward_pre.
 *   dbev_wino_conv3x3_forward_act   the same with relu != 0: max(y, 0) on the way out -- with the scale of an eval-mode BatchNorm folded into
 *                                   the filters before packing and its shift passed as `bias`, conv -> norm -> ReLU of a frozen stack
 *                                   (the CenterPoint teacher's SECOND / head: second.py:60-78) is this one launch
 *   For the data gradient call it with (x = grad_y, Cin <-> Cout swapped, the data_gradient pack).
 *   dbev_wino_conv3x3_backward_weight   grad_w[co][c][a][b] (element strides as for the pack) = sum over pixels, in the Winograd
 *                                   domain: G^T [ sum_tiles (B^T d B) .* (A dY A^T) ] G; workspace from ..._workspace_bytes, fixed
 *                                   summation order (no atomics).  Cin % 64 == 0, Cout % 64 == 0.
 * Numerics: fp32 throughout (v_mfma_f32_32x32x2_f32); the transforms add / subtract and multiply by 1/2: error vs an fp64
 * convolution 2-4x that of a direct fp32 convolution (tests/test_gpu_wino.py).
 * ---------------------------------------------------------------------------------- */
long long dbev_wino_filter_floats(int K, int J);
int dbev_wino_filter_pack(const float* weight, long long so, long long sc, long long sa, long long sb, int Cout, int Cin,
                          int flags, float* packed, dbevStream_t stream);
int dbev_wino_filter_pack_pair(const float* weight, long long so, long long sc, long long sa, long long sb, int Cout, int Cin,
                              int fwd_kernel, int dgrad_kernel, float* packed_fwd, float* packed_dgrad, dbevStream_t stream);
int dbev_wino_conv3x3_forward_kernel(int N, int H, int W, int Cin, int Cout);    /* 2 | 3: the forward kernel the layer gets; 0: unsupported */
int dbev_wino_conv3x3_stats_rows(int N, int H, int W, int Cin, int Cout);
int dbev_wino_conv3x3_forward(const float* x_nhwc, const float* packed, const float* bias, float* y_nhwc, float* stats_partial,
                              int N, int H, int W, int Cin, int Cout, dbevStream_t stream);
int dbev_wino_conv3x3_forward_act(const float* x_nhwc, const float* packed, const float* bias, float* y_nhwc, float* stats_partial,
                                 int N, int H, int W, int Cin, int Cout, int relu, dbevStream_t stream);
size_t dbev_wino_conv3x3_backward_weight_workspace_bytes(int N, int H, int W, int Cin, int Cout);   /* 0: unsupported geometry */
int dbev_wino_conv3x3_backward_weight(const float* x_nhwc, const float* grad_y_nhwc, float* grad_weight, long long so, long long sc,
                                      long long sa, long long sb, int N, int H, int W, int Cin, int Cout, void* workspace,
                                      size_t workspace_bytes, dbevStream_t stream);

/* ------------------------------------------------------------------------------------
 * 1x1 convolutions as fp32-MFMA GEMMs with no VALU instruction in the main loop (csrc/gemm1x1.hip; round 4): forward, data gradient
 * (the forward on grad_y with the transposed weight) and weight gradient of `nn.Conv2d(k=1)` -> cuDNN / MIOpen in the reference's
 * bottlenecks and necks (mmdet3d/models/bricks/res_block.py:102-230, necks/fpn.py:10-204, necks/lss_fpn.py:10-72).
 *   x_nhwc f32[M, x_row_stride >= Cin] (M = N*H*W pixels), weight f32[Cout, Cin], y_nhwc f32[M, Cout];
 *   M % 128 == 0, Cin % 32 == 0, Cout % 64 == 0 (else DBEV_EINVAL: the caller keeps the library's convolution).
 *   stats_partial (may be NULL) f32[rows, 2, Cout], rows = dbev_gemm1x1_stats_rows(...): per workgroup row the sums of y and y^2
 *   (the partial-row layout of dbev_bn_act_train_forward_pre).
 *   dbev_gemm1x1_backward_weight: grad_weight f32[Cout, Cin] = grad_y^T x, M % 32 == 0, Cin % 64 == 0, Cout % 64 == 0; shares of
 *   the rows summed in a fixed order (no atomics).
 * ---------------------------------------------------------------------------------- */
int dbev_gemm1x1_stats_rows(long long M, int Cin, int Cout, int x_row_stride);
int dbev_gemm1x1_forward(const float* x_nhwc, const float* weight, float* y_nhwc, float* stats_partial, long long M, int Cin,
                         int Cout, int x_row_stride, dbevStream_t stream);
size_t dbev_gemm1x1_backward_weight_workspace_bytes(long long M, int Cin, int Cout, int x_row_stride);
int dbev_gemm1x1_backward_weight(const float* x_nhwc, const float* grad_y_nhwc, float* grad_weight, long long M, int Cin, int Cout,
                                 int x_row_stride, void* workspace, size_t workspace_bytes, dbevStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DBEV_HIP_H */
