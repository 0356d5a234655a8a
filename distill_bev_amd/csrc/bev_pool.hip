// bev_pool forward/backward for gfx950 (MI355X).
//
// Replaces mmdet3d/ops/bev_pool/src/bev_pool_cuda.cu:20-42 (bev_pool_kernel) and :61-84
// (bev_pool_grad_kernel) of the reference behind the same argument contract
// (bev_pool.cpp:22-28,60-66).  The reference runs one *thread* per (interval, channel) and
// walks the interval serially; with <=592 rows per BEV cell against a mean of 22 that
// serialises on the dense cells next to the cameras.  Here:
//
//  forward : one 64-lane WAVEFRONT per interval.  A row of c floats is c/4 float4 lanes
//            (16 lanes at c=64), so one wave instruction fetches 64/(c/4) = 4 whole rows
//            = 1 KiB, fully coalesced; UNROLL such loads are kept in flight per lane
//            (32 rows / 8 KiB per wave) to cover HBM latency on the long runs.  The
//            row-group partials are combined with a fixed shuffle tree -> deterministic.
//  backward: one lane-group per ROW (not per interval): x_grad[r] = out_grad[cell(r)].
//            Perfectly balanced, every byte of x_grad written exactly once, out_grad rows
//            served from L2 (4.2 MB per frame).
//
// HBM roofline: algorithmic bytes 4nC + 16n + 8 n_int + 4 BDHWC -- pure
// streaming, no reuse, so no LDS staging: LDS would only add a hop.
#include "common.h"

namespace {

template <int UNROLL>
__global__ __launch_bounds__(256) void bev_pool_fwd_vec4(
    const float4* __restrict__ x, const int4* __restrict__ geom,
    const int* __restrict__ starts, const int* __restrict__ lengths,
    float4* __restrict__ out, int n_intervals, int c4, int rpw, int d, int h, int w) {
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (wave >= n_intervals) return;
  const int start = starts[wave];
  const int len = lengths[wave];
  const int sub = lane / c4;  // which of the rpw rows of a wave-load this lane reads
  const int q = lane - sub * c4;
  const bool active = sub < rpw;

  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (active) {
    const float4* base = x + static_cast<size_t>(start) * c4 + q;
    int r = sub;
    const int stride = rpw * UNROLL;
    for (; r + (UNROLL - 1) * rpw < len; r += stride) {
      float4 v[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) v[u] = base[static_cast<size_t>(r + u * rpw) * c4];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w;
      }
    }
    for (; r < len; r += rpw) {
      const float4 v = base[static_cast<size_t>(r) * c4];
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
  }
  // combine the rpw row-group partials: lane q gathers lanes q + s*c4 in fixed order
  float4 tot = acc;
  for (int s = 1; s < rpw; ++s) {
    const int src = q + s * c4;
    tot.x += __shfl(acc.x, src);
    tot.y += __shfl(acc.y, src);
    tot.z += __shfl(acc.z, src);
    tot.w += __shfl(acc.w, src);
  }
  if (sub == 0) {
    const int4 g = geom[start];  // (x, y, z, b)
    const size_t cell = ((static_cast<size_t>(g.w) * d + g.z) * h + g.x) * w + g.y;
    out[cell * c4 + q] = tot;
  }
}

// any c (not a multiple of 4, or wider than a wave): lane per channel, strided.
__global__ __launch_bounds__(256) void bev_pool_fwd_scalar(
    const float* __restrict__ x, const int4* __restrict__ geom,
    const int* __restrict__ starts, const int* __restrict__ lengths,
    float* __restrict__ out, int n_intervals, int c, int d, int h, int w) {
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (wave >= n_intervals) return;
  const int start = starts[wave];
  const int len = lengths[wave];
  const int4 g = geom[start];
  const size_t cell = ((static_cast<size_t>(g.w) * d + g.z) * h + g.x) * w + g.y;
  for (int ch = lane; ch < c; ch += 64) {
    const float* p = x + static_cast<size_t>(start) * c + ch;
    float acc = 0.f;
    for (int r = 0; r < len; ++r) acc += p[static_cast<size_t>(r) * c];
    out[cell * c + ch] = acc;
  }
}

__global__ __launch_bounds__(256) void bev_pool_bwd_vec4(
    const float4* __restrict__ out_grad, const int4* __restrict__ geom,
    float4* __restrict__ x_grad, int n, int c4, int rpw, int d, int h, int w) {
  const int lane = threadIdx.x & 63;
  const int sub = lane / c4;
  const int q = lane - sub * c4;
  if (sub >= rpw) return;
  const long long wave = (static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
  const long long nwaves = (static_cast<long long>(gridDim.x) * blockDim.x) >> 6;
  for (long long r = wave * rpw + sub; r < n; r += nwaves * rpw) {
    const int4 g = geom[r];
    const size_t cell = ((static_cast<size_t>(g.w) * d + g.z) * h + g.x) * w + g.y;
    st_nt(x_grad + static_cast<size_t>(r) * c4 + q, out_grad[cell * c4 + q]);   // 909 MB written once, streamed
  }
}

__global__ __launch_bounds__(256) void bev_pool_bwd_scalar(
    const float* __restrict__ out_grad, const int4* __restrict__ geom,
    float* __restrict__ x_grad, int n, int c, int d, int h, int w) {
  const long long total = static_cast<long long>(n) * c;
  for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < total;
       i += static_cast<long long>(gridDim.x) * blockDim.x) {
    const long long r = i / c;
    const int ch = static_cast<int>(i - r * c);
    const int4 g = geom[r];
    const size_t cell = ((static_cast<size_t>(g.w) * d + g.z) * h + g.x) * w + g.y;
    x_grad[i] = out_grad[cell * c + ch];
  }
}

}  // namespace

extern "C" int dbev_bev_pool_forward(const float* x, const int32_t* geom_feats,
                                     const int32_t* interval_starts,
                                     const int32_t* interval_lengths, float* out, int n, int c,
                                     int n_intervals, int b, int d, int h, int w,
                                     dbevStream_t stream) {
  if (n < 0 || c <= 0 || n_intervals < 0 || b <= 0 || d <= 0 || h <= 0 || w <= 0) return DBEV_EINVAL;
  hipStream_t s = dbev_stream(stream);
  const size_t out_bytes = static_cast<size_t>(b) * d * h * w * c * sizeof(float);
  DBEV_HIP_TRY(hipMemsetAsync(out, 0, out_bytes, s));
  if (n_intervals == 0 || n == 0) return 0;
  const int blocks = dbev_ceil_div(n_intervals, 4);  // 4 waves (intervals) per 256-thread block
  if ((c & 3) == 0 && (c >> 2) <= 64) {
    const int c4 = c >> 2;
    const int rpw = 64 / c4;
    hipLaunchKernelGGL(bev_pool_fwd_vec4<8>, dim3(blocks), dim3(256), 0, s,
                       reinterpret_cast<const float4*>(x), reinterpret_cast<const int4*>(geom_feats),
                       interval_starts, interval_lengths, reinterpret_cast<float4*>(out),
                       n_intervals, c4, rpw, d, h, w);
  } else {
    hipLaunchKernelGGL(bev_pool_fwd_scalar, dim3(blocks), dim3(256), 0, s, x,
                       reinterpret_cast<const int4*>(geom_feats), interval_starts, interval_lengths,
                       out, n_intervals, c, d, h, w);
  }
  DBEV_LAUNCH_CHECK();
  return 0;
}

extern "C" int dbev_bev_pool_backward(const float* out_grad, const int32_t* geom_feats,
                                      const int32_t* interval_starts,
                                      const int32_t* interval_lengths, float* x_grad, int n, int c,
                                      int n_intervals, int b, int d, int h, int w,
                                      dbevStream_t stream) {
  (void)interval_starts; (void)interval_lengths; (void)n_intervals; (void)b;
  if (n < 0 || c <= 0 || d <= 0 || h <= 0 || w <= 0) return DBEV_EINVAL;
  if (n == 0) return 0;
  hipStream_t s = dbev_stream(stream);
  if ((c & 3) == 0 && (c >> 2) <= 64) {
    const int c4 = c >> 2;
    const int rpw = 64 / c4;
    const long long waves = (static_cast<long long>(n) + rpw - 1) / rpw;
    long long blocks = (waves + 3) / 4;
    if (blocks > DBEV_MAX_GRID * 4) blocks = DBEV_MAX_GRID * 4;
    hipLaunchKernelGGL(bev_pool_bwd_vec4, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, s,
                       reinterpret_cast<const float4*>(out_grad),
                       reinterpret_cast<const int4*>(geom_feats),
                       reinterpret_cast<float4*>(x_grad), n, c4, rpw, d, h, w);
  } else {
    long long blocks = (static_cast<long long>(n) * c + 255) / 256;
    if (blocks > DBEV_MAX_GRID * 4) blocks = DBEV_MAX_GRID * 4;
    hipLaunchKernelGGL(bev_pool_bwd_scalar, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, s,
                       out_grad, reinterpret_cast<const int4*>(geom_feats), x_grad, n, c, d, h, w);
  }
  DBEV_LAUNCH_CHECK();
  return 0;
}


// ---- [B, C, S] <-> [B, S, C] transposes for callers that hand bev_pool a gradient in the reference's contiguous
// [B, C, D, H, W] layout (bev_pool.py:64-81 out_grad) while the kernels work on cell-major rows.  64 x 64 tiles
// through LDS (+1 padding): both the read and the write side move 256-byte rows.
namespace {
__global__ __launch_bounds__(256) void transpose_cs(const float* __restrict__ in, float* __restrict__ out, int C, int S) {
  __shared__ float tile[64][65];
  const int b = blockIdx.z, c0 = blockIdx.y * 64, s0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const float* ib = in + static_cast<size_t>(b) * C * S;
  float* ob = out + static_cast<size_t>(b) * C * S;
  for (int r = ty; r < 64; r += 4) {              // read rows of the [C, S] plane: c = c0 + r, s = s0 + tx
    const int c = c0 + r, sidx = s0 + tx;
    tile[r][tx] = (c < C && sidx < S) ? ib[static_cast<size_t>(c) * S + sidx] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 64; r += 4) {              // write rows of the [S, C] plane: s = s0 + r, c = c0 + tx
    const int sidx = s0 + r, c = c0 + tx;
    if (sidx < S && c < C) ob[static_cast<size_t>(sidx) * C + c] = tile[tx][r];
  }
}
}  // namespace

extern "C" int dbev_transpose_bcs_to_bsc(const float* in_bcs, float* out_bsc, int B, int C, int S, dbevStream_t stream) {
  if (B <= 0 || C <= 0 || S <= 0 || in_bcs == nullptr || out_bsc == nullptr) return DBEV_EINVAL;
  hipLaunchKernelGGL(transpose_cs, dim3(dbev_ceil_div(S, 64), dbev_ceil_div(C, 64), B), dim3(256), 0, dbev_stream(stream),
                     in_bcs, out_bsc, C, S);
  DBEV_LAUNCH_CHECK();
  return 0;
}
