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

#ifdef __cplusplus
}
#endif
#endif /* DBEV_HIP_H */
