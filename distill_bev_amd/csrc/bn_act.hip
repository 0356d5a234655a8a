// Training-mode BatchNorm2d fused with the residual add and ReLU that follow it, channels-last fp32.
//
// The dense stack of the training step (ResNet-50 image encoder mmdet ResNet `Bottleneck`, the BEV
// encoder mmdet3d/models/bricks/res_block.py:11-100, mmcv ConvModule conv-norm-act triples) spends 20 % of
// the step in HBM-bound BN / ReLU / add passes over activations of up to 1.1 GB (96 x 256 x 64 x 176 fp32).
// Unfused (MIOpen BN + ATen clamp / add / threshold_backward) a conv-BN-ReLU costs 5 full-tensor passes
// forward and 8 backward; here 3 and 5:
//   forward :  bn_stats (read x)                 -> per-workgroup partial (sum, sum of squares) per channel
//              bn_finalize (tiny)                 -> mean, invstd, scale/shift, running-stat update (fp64 merge)
//              bn_apply (read x [,res], write y)  -> y = relu(x * scale + shift [+ res])
//   backward:  bn_bwd_reduce (read dy, x [,y])    -> partial (sum dz, sum dz * xhat),  dz = dy * [y > 0]
//              bn_bwd_finalize (tiny)             -> dgamma, dbeta, per-channel (A, B, C)
//              bn_bwd_dx (read dy, x [,y], write dx [,dres])   dx = A * dz + B * x + C,  dres = dz
// Semantics = torch.nn.functional.batch_norm(training=True) + add + relu: biased variance for the
// normalisation, unbiased for running_var, running = (1 - momentum) * running + momentum * batch.
// Every reduction has a fixed order (no float atomics): bit-reproducible run to run.
//
// Mapping: the tensor is M rows x C channels, C4 = C / 4 float4 columns.  A workgroup of 256 threads covers
// CH = min(C4, 256) consecutive columns x RP = 256 / CH row phases, so a thread keeps ONE column and its
// per-channel accumulators / coefficients in registers while it strides over rows, and a wave always
// touches >= 1 KB of contiguous memory.  gridDim.y walks column chunks when C4 > 256.
#include "common.h"

#include <stdlib.h>

#include <atomic>

namespace {

constexpr int BN_MAX_BLOCKS_X = 2048;   // partial sums per channel merged by the finalize kernels
constexpr int BN_ROWS_UNROLL = 4;
constexpr int BN_FIN_CH = 4, BN_FIN_PH = 64;   // finalize kernels: channels x partial phases per workgroup of 256

struct BnGeom {
  int M, C, C4, CH, RP, GY, NBX;
  int RTPB, RRP, RUNR;     // reduction passes (bn_stats, bn_bwd_reduce): threads per block, row phases, row unroll
  int STRIPE;              // reduction passes: 1 = workgroups interleaved over row stripes, walked back to front (see stripe_rows)
};

// Tunables of the reduction passes, overridable from the environment for A/B runs (tools/kbench_bn2.py):
//   DBEV_BN_RTPB  threads per workgroup (256 / 512 / 1024)      DBEV_BN_RUNR  rows in flight per thread (4 / 8)
//   DBEV_BN_RNBX  cap on workgroups (= partial rows the finalize kernel merges) per column chunk
//   DBEV_BN_STRIPE  1 (default): stripe walk, back to front; 0: one contiguous row range per workgroup (round-1 layout)
struct BnTune { int rtpb, runr, rnbx, stripe; };
const BnTune& bn_tune() {
  static const BnTune t = [] {
    BnTune v{512, 4, 256, 1};   // swept on MI355X (tools/sweep_bn.sh): 512-thread workgroups, 4 rows in flight, <= 256 partial rows
    if (const char* e = getenv("DBEV_BN_RTPB")) v.rtpb = atoi(e);
    if (const char* e = getenv("DBEV_BN_RUNR")) v.runr = atoi(e);
    if (const char* e = getenv("DBEV_BN_RNBX")) v.rnbx = atoi(e);
    if (const char* e = getenv("DBEV_BN_STRIPE")) v.stripe = atoi(e) != 0;
    if (v.rtpb != 256 && v.rtpb != 512 && v.rtpb != 1024) v.rtpb = 512;
    if (v.runr != 4 && v.runr != 8) v.runr = 4;
    if (v.rnbx < 1 || v.rnbx > BN_MAX_BLOCKS_X) v.rnbx = 256;
    return v;
  }();
  return t;
}

bool bn_geom(long long M, int C, BnGeom* g) {
  if (M <= 0 || M > 0x3fffffffLL || C <= 0 || (C & 3)) return false;   // row indices stay clear of int overflow
  const int C4 = C >> 2;
  int CH;
  if (C4 <= 256) {
    if (C4 & (C4 - 1)) return false;
    CH = C4;
  } else {
    if (C4 & 255) return false;
    CH = 256;
  }
  g->M = static_cast<int>(M);
  g->C = C;
  g->C4 = C4;
  g->CH = CH;
  g->RP = 256 / CH;
  g->GY = C4 / CH;
  const BnTune& t = bn_tune();
  g->RTPB = t.rtpb;
  g->RRP = t.rtpb / CH;
  g->RUNR = t.runr;
  g->STRIPE = t.stripe;
  // every reduction workgroup gets at least 4 unrolled trips over its rows; at most `rnbx` partial rows
  const long long per = static_cast<long long>(g->RRP) * g->RUNR * 4;
  const long long tiles = (M + per - 1) / per;
  long long nbx = t.rnbx / g->GY;
  if (nbx < 1) nbx = 1;
  g->NBX = static_cast<int>(tiles < nbx ? tiles : nbx);
  return true;
}

__device__ __forceinline__ float4 f4(float v) { return make_float4(v, v, v, v); }
__device__ __forceinline__ void add4(float4& a, const float4& b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
__device__ __forceinline__ void fma4v(float4& a, const float4& b, const float4& c) {
  a.x = fmaf(b.x, c.x, a.x); a.y = fmaf(b.y, c.y, a.y); a.z = fmaf(b.z, c.z, a.z); a.w = fmaf(b.w, c.w, a.w);
}

// every workgroup streams ONE contiguous range of rows (HBM pages / TLB entries are walked once, in order):
// returns the first row, *r_end = one past the last
// REV: walk the tensor back to front.  The reduction passes (bn_stats, bn_bwd_reduce) run right after a producer /
// before a consumer that streams the same tensor front to back, so the part that is still in (or will still be in) the
// 256 MB memory-side cache is the END for the first pass and the START for the pass after it: measured -6 % / -12 %
// on the forward+backward of 277 MB / 138 MB activations, neutral above 500 MB.
template <bool REV = false>
__device__ __forceinline__ int block_rows(const BnGeom& g, int* r_end, int rp_count, bool rev = false) {
  int per = (g.M + gridDim.x - 1) / gridDim.x;
  per = (per + rp_count - 1) / rp_count * rp_count;
  const long long b = static_cast<long long>((REV || rev) ? gridDim.x - 1 - blockIdx.x : blockIdx.x) * per;
  const long long e = b + per;
  *r_end = static_cast<int>(e < g.M ? e : g.M);
  return static_cast<int>(b < g.M ? b : g.M);
}

// Stripe walk of the reduction passes: a stripe = gridDim.x consecutive pieces of R = RRP * UNR rows, workgroup b takes piece b of every
// stripe, stripes are walked from the LAST to the first.  Against one contiguous range per workgroup: the tensor a reduction pass reads
// was just written front to back by the convolution before it, so its tail sits in the 256 MB memory-side cache and its head in HBM;
// with contiguous ranges the workgroups on the tail finish early and the ones on the head pull from HBM with half the machine idle,
// with stripes every workgroup sees the same mix, the cached tail first (before this pass's own reads evict it).  Inside the step
// (alternating A/B runs on one box): the bn_* family 24.3 -> 23.7 ms, the apply pass that follows finds the head of the tensor
// hot (0.70 -> 0.72 of HBM peak); in isolation unchanged.
__device__ __forceinline__ int stripe_count(const BnGeom& g, int R) {
  const long long per = static_cast<long long>(gridDim.x) * R;
  return static_cast<int>((g.M + per - 1) / per);
}
__device__ __forceinline__ int stripe_row0(int s, int R) { return (s * static_cast<int>(gridDim.x) + static_cast<int>(blockIdx.x)) * R; }

// block-level merge of two float4 accumulators over the row phases, result written by phase 0:
// partial[(blockIdx.x * 2 + which) * C + 4*q .. +4]
template <int TPB>
__device__ __forceinline__ void block_merge_store(float4 a, float4 b, float* __restrict__ partial, int C, int CH, int RP,
                                                  int q, int ql, int rp) {
  __shared__ float4 sa[TPB], sb[TPB];
  sa[threadIdx.x] = a;
  sb[threadIdx.x] = b;
  __syncthreads();
  for (int st = RP >> 1; st > 0; st >>= 1) {
    if (rp < st) {
      add4(sa[threadIdx.x], sa[threadIdx.x + st * CH]);
      add4(sb[threadIdx.x], sb[threadIdx.x + st * CH]);
    }
    __syncthreads();
  }
  if (rp == 0) {
    reinterpret_cast<float4*>(partial + (static_cast<size_t>(blockIdx.x) * 2 + 0) * C)[q] = sa[ql];
    reinterpret_cast<float4*>(partial + (static_cast<size_t>(blockIdx.x) * 2 + 1) * C)[q] = sb[ql];
  }
}


// ---- finalize without a launch (round 3) -------------------------------------------------------------------------------------
// The merge of the per-workgroup partial rows used to be its own 16..64-workgroup kernel between the reduction pass and the
// streaming pass that needs its result (bn_finalize / bn_bwd_finalize*: 235 launches, 1.6 ms per training step).  Now:
//  * producers (bn_stats, bn_bwd_reduce*): the 16 workgroups of a GROUP of consecutive blockIdx.x take a ticket after writing their
//    rows; the last arriver merges the group's rows (fixed order) into the group's first row -- groups finish at different times, so
//    these merges overlap the pass itself;
//  * consumers (bn_apply, bn_bwd_dx*): every thread merges the <= 16 group rows of its own four channels in fp64 in its prologue
//    (<= 32 float4 loads out of L2, the same addresses for every workgroup) and derives its coefficients in registers; workgroup 0 of
//    a column chunk also writes what later kernels read (saved mean / invstd / scale / shift, running statistics, dgamma / dbeta).
// Tickets live in a small device array, self-resetting (the last arriver zeroes its slot); the host hands every call its own slots.
constexpr int BN_GROUP = 16;            // workgroups per group
constexpr int BN_MAX_GROUPS = 16;       // groups per column chunk (NBX <= 256)
constexpr int BN_TICKET_SLOTS = 8192;
__device__ unsigned g_bn_tickets[BN_TICKET_SLOTS];

template <int TPB, int V>
__device__ __forceinline__ void group_merge(float* __restrict__ partial, int C, int CH, int RRP, int q, int ql, int rp, int nbx,
                                            unsigned* __restrict__ tickets) {
  __shared__ int s_last;
  __shared__ float4 sm[V][TPB];
  const int g = blockIdx.x / BN_GROUP;
  const int cnt = min(BN_GROUP, nbx - g * BN_GROUP);
  __syncthreads();                                   // the row's stores are issued ...
  if (threadIdx.x == 0) {
    __threadfence();                                 // ... and visible device-wide before the ticket (ONE release per workgroup: the
                                                     // fence writes the XCD's L2 back; per thread it cost 40 us per launch)
    unsigned* tk = tickets + blockIdx.y * BN_MAX_GROUPS + g;
    const unsigned t = atomicAdd(tk, 1u);
    s_last = (t == static_cast<unsigned>(cnt - 1));
    if (s_last) *tk = 0u;                            // every member has arrived: the slot is free for its next user
  }
  __syncthreads();
  if (!s_last) return;
  __threadfence();                                   // the other members' rows
  float4 acc[V];
#pragma unroll
  for (int v = 0; v < V; ++v) acc[v] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int r = rp; r < cnt; r += RRP) {
    const size_t row = static_cast<size_t>(g) * BN_GROUP + r;
#pragma unroll
    for (int v = 0; v < V; ++v) {
      const float4 x = reinterpret_cast<const float4*>(partial + (row * V + v) * C)[q];
      acc[v].x += x.x; acc[v].y += x.y; acc[v].z += x.z; acc[v].w += x.w;
    }
  }
#pragma unroll
  for (int v = 0; v < V; ++v) sm[v][threadIdx.x] = acc[v];
  __syncthreads();
  for (int st = RRP >> 1; st > 0; st >>= 1) {
    if (rp < st) {
#pragma unroll
      for (int v = 0; v < V; ++v) {
        float4 a = sm[v][threadIdx.x];
        const float4 b = sm[v][threadIdx.x + st * CH];
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        sm[v][threadIdx.x] = a;
      }
    }
    __syncthreads();
  }
  if (rp == 0) {
    const size_t row = static_cast<size_t>(g) * BN_GROUP;
#pragma unroll
    for (int v = 0; v < V; ++v) reinterpret_cast<float4*>(partial + (row * V + v) * C)[q] = sm[v][ql];
  }
}

struct d4 { double x, y, z, w; };
// fp64 sum of the group rows (first row of every group) of value v for this thread's four channels
template <int V>
__device__ __forceinline__ void merged_rows(const float* __restrict__ partial, int C, int q, int nbx, d4 (&out)[V]) {
#pragma unroll
  for (int v = 0; v < V; ++v) out[v] = d4{0.0, 0.0, 0.0, 0.0};
  const int ng = (nbx + BN_GROUP - 1) / BN_GROUP;
  for (int g = 0; g < ng; ++g) {
#pragma unroll
    for (int v = 0; v < V; ++v) {
      const float4 x = reinterpret_cast<const float4*>(partial + (static_cast<size_t>(g) * BN_GROUP * V + v) * C)[q];
      out[v].x += x.x; out[v].y += x.y; out[v].z += x.z; out[v].w += x.w;
    }
  }
}

// forward finalize in a consumer's prologue: scale / shift of this thread's four channels; `write`: also store what later kernels read
struct BnFin {
  const float* partial;           // nullptr: coefficients come from memory (bn_finalize ran, or eval mode)
  int nbx, M;
  const float *gamma, *beta;
  float *running_mean, *running_var;
  float momentum, eps;
  float *save_mean, *save_invstd, *coef;
  long long* nbt;
};

__device__ __forceinline__ void fin_forward(const BnFin& f, int C, int q, bool write, float4* sc_out, float4* sh_out) {
  d4 m[2];
  merged_rows<2>(f.partial, C, q, f.nbx, m);
  const double s[4] = {m[0].x, m[0].y, m[0].z, m[0].w}, sq[4] = {m[1].x, m[1].y, m[1].z, m[1].w};
  float sc[4], sh[4], mf[4], is[4];
  const float4 ga = reinterpret_cast<const float4*>(f.gamma)[q], be = reinterpret_cast<const float4*>(f.beta)[q];
  const float gv[4] = {ga.x, ga.y, ga.z, ga.w}, bv[4] = {be.x, be.y, be.z, be.w};
  double mean[4], var[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    mean[i] = s[i] / f.M;
    var[i] = sq[i] / f.M - mean[i] * mean[i];
    if (var[i] < 0.0) var[i] = 0.0;
    is[i] = static_cast<float>(1.0 / sqrt(var[i] + static_cast<double>(f.eps)));
    mf[i] = static_cast<float>(mean[i]);
    sc[i] = gv[i] * is[i];
    sh[i] = fmaf(-mf[i], sc[i], bv[i]);
  }
  *sc_out = make_float4(sc[0], sc[1], sc[2], sc[3]);
  *sh_out = make_float4(sh[0], sh[1], sh[2], sh[3]);
  if (write) {
    reinterpret_cast<float4*>(f.save_mean)[q] = make_float4(mf[0], mf[1], mf[2], mf[3]);
    reinterpret_cast<float4*>(f.save_invstd)[q] = make_float4(is[0], is[1], is[2], is[3]);
    reinterpret_cast<float4*>(f.coef)[q] = *sc_out;
    reinterpret_cast<float4*>(f.coef + C)[q] = *sh_out;
    if (f.running_mean != nullptr) {
      float4 rm = reinterpret_cast<float4*>(f.running_mean)[q], rv = reinterpret_cast<float4*>(f.running_var)[q];
      float r1[4] = {rm.x, rm.y, rm.z, rm.w}, r2[4] = {rv.x, rv.y, rv.z, rv.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const double unbiased = f.M > 1 ? var[i] * (static_cast<double>(f.M) / (f.M - 1)) : var[i];
        r1[i] = static_cast<float>((1.0 - f.momentum) * r1[i] + f.momentum * mean[i]);
        r2[i] = static_cast<float>((1.0 - f.momentum) * r2[i] + f.momentum * unbiased);
      }
      reinterpret_cast<float4*>(f.running_mean)[q] = make_float4(r1[0], r1[1], r1[2], r1[3]);
      reinterpret_cast<float4*>(f.running_var)[q] = make_float4(r2[0], r2[1], r2[2], r2[3]);
    }
  }
}

template <int TPB, int UNR>
__global__ __launch_bounds__(TPB) void bn_stats(const float4* __restrict__ x, float* __restrict__ partial, BnGeom g,
                                                 unsigned* __restrict__ tickets = nullptr) {
  const int ql = threadIdx.x % g.CH, rp = threadIdx.x / g.CH;
  const int q = blockIdx.y * g.CH + ql;
  float4 s = f4(0.f), ss = f4(0.f);
  const int stride = g.RRP;
  if (g.STRIPE) {
    const int R = stride * UNR;
    for (int st = stripe_count(g, R) - 1; st >= 0; --st) {
      const int r0 = stripe_row0(st, R) + rp;
      float4 v[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int r = r0 + u * stride;
        v[u] = r < g.M ? x[static_cast<size_t>(r) * g.C4 + q] : f4(0.f);
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u) { add4(s, v[u]); fma4v(ss, v[u], v[u]); }
    }
    block_merge_store<TPB>(s, ss, partial, g.C, g.CH, g.RRP, q, ql, rp);
    if (tickets != nullptr) group_merge<TPB, 2>(partial, g.C, g.CH, g.RRP, q, ql, rp, gridDim.x, tickets);
    return;
  }
  int r_end;
  int r = block_rows<true>(g, &r_end, g.RRP) + rp;
  for (; r + (UNR - 1) * stride < r_end; r += UNR * stride) {
    float4 v[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) v[u] = x[static_cast<size_t>(r + u * stride) * g.C4 + q];
#pragma unroll
    for (int u = 0; u < UNR; ++u) { add4(s, v[u]); fma4v(ss, v[u], v[u]); }
  }
  for (; r < r_end; r += stride) {
    const float4 v = x[static_cast<size_t>(r) * g.C4 + q];
    add4(s, v);
    fma4v(ss, v, v);
  }
  block_merge_store<TPB>(s, ss, partial, g.C, g.CH, g.RRP, q, ql, rp);
  if (tickets != nullptr) group_merge<TPB, 2>(partial, g.C, g.CH, g.RRP, q, ql, rp, gridDim.x, tickets);
}

// 4 channels x 64 partial phases per workgroup; fp64 merge in a fixed order.  coef: [0] scale, [1] shift (forward) -- saved for backward.
template <typename PT>
__global__ __launch_bounds__(256) void bn_finalize(const PT* __restrict__ partial, int nbx, int M, int C,
                                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                                   float* __restrict__ running_mean, float* __restrict__ running_var,
                                                   float momentum, float eps, float* __restrict__ save_mean,
                                                   float* __restrict__ save_invstd, float* __restrict__ coef,
                                                   long long* __restrict__ num_batches_tracked) {
  if (num_batches_tracked != nullptr && blockIdx.x == 0 && threadIdx.x == 0) *num_batches_tracked += 1;
  __shared__ double ls[4][BN_FIN_CH], lq[4][BN_FIN_CH];
  const int cl = threadIdx.x % BN_FIN_CH, ph = threadIdx.x / BN_FIN_CH;
  const int c = blockIdx.x * BN_FIN_CH + cl;
  double s = 0.0, sq = 0.0;
  if (c < C) {
    int b = ph;
    for (; b + 7 * BN_FIN_PH < nbx; b += 8 * BN_FIN_PH) {      // 16 independent loads in flight: <= 512 partial rows = ONE trip
      PT a[8], q2[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        a[u] = partial[(static_cast<size_t>(b + u * BN_FIN_PH) * 2 + 0) * C + c];
        q2[u] = partial[(static_cast<size_t>(b + u * BN_FIN_PH) * 2 + 1) * C + c];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) { s += static_cast<double>(a[u]); sq += static_cast<double>(q2[u]); }
    }
    for (; b < nbx; b += BN_FIN_PH) {
      s += static_cast<double>(partial[(static_cast<size_t>(b) * 2 + 0) * C + c]);
      sq += static_cast<double>(partial[(static_cast<size_t>(b) * 2 + 1) * C + c]);
    }
  }
  // 64 phases -> 1: xor-tree over the 16 phases a wave holds (lanes 4 apart), then the 4 waves through LDS -- a fixed
  // order, and 4 shuffle levels + 3 adds instead of a 63-step serial walk through LDS (3 us of a 8 us kernel)
#pragma unroll
  for (int m = BN_FIN_CH; m < 64; m <<= 1) { s += __shfl_xor(s, m); sq += __shfl_xor(sq, m); }
  if ((threadIdx.x & 63) < BN_FIN_CH) { ls[threadIdx.x >> 6][cl] = s; lq[threadIdx.x >> 6][cl] = sq; }
  __syncthreads();
  if (ph == 0 && c < C) {
    s = ((ls[0][cl] + ls[1][cl]) + ls[2][cl]) + ls[3][cl];
    sq = ((lq[0][cl] + lq[1][cl]) + lq[2][cl]) + lq[3][cl];
    const double mean = s / M;
    double var = sq / M - mean * mean;
    if (var < 0.0) var = 0.0;
    const float invstd = static_cast<float>(1.0 / sqrt(var + static_cast<double>(eps)));
    const float mf = static_cast<float>(mean);
    save_mean[c] = mf;
    save_invstd[c] = invstd;
    const float sc = gamma[c] * invstd;
    coef[c] = sc;
    coef[C + c] = fmaf(-mf, sc, beta[c]);
    if (running_mean != nullptr) {
      const double unbiased = M > 1 ? var * (static_cast<double>(M) / (M - 1)) : var;
      running_mean[c] = static_cast<float>((1.0 - momentum) * running_mean[c] + momentum * mean);
      running_var[c] = static_cast<float>((1.0 - momentum) * running_var[c] + momentum * unbiased);
    }
  }
}

// RAFF: the residual is itself a normalised tensor, res * rsc + rsh with the scale / shift of a second BatchNorm
// (`rcoef`): the `downsample` branch of a stage-first residual block, whose normalised copy is then never written
template <bool RES, bool RELU, bool RAFF = false>
__global__ __launch_bounds__(256) void bn_apply(const float4* __restrict__ x, const float4* __restrict__ res,
                                                const float* __restrict__ coef, float4* __restrict__ y, BnGeom g,
                                                const float* __restrict__ rcoef, int rev, BnFin fin, BnFin fin_d,
                                                unsigned char* __restrict__ gate_mask = nullptr) {
  const int ql = threadIdx.x % g.CH, rp = threadIdx.x / g.CH;
  const int q = blockIdx.y * g.CH + ql;
  const bool writer = blockIdx.x == 0 && rp == 0;
  float4 sc, sh;
  if (fin.partial != nullptr) {                                      // finalize in the prologue (see group_merge)
    fin_forward(fin, g.C, q, writer, &sc, &sh);
    if (fin.nbt != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *fin.nbt += 1;
  } else {
    sc = reinterpret_cast<const float4*>(coef)[q];
    sh = reinterpret_cast<const float4*>(coef + g.C)[q];
  }
  float4 rsc = f4(1.f);
  if (RAFF) {
    float4 rsh;
    if (fin_d.partial != nullptr) {
      fin_forward(fin_d, g.C, q, writer, &rsc, &rsh);
      if (fin_d.nbt != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *fin_d.nbt += 1;
    } else {
      rsc = reinterpret_cast<const float4*>(rcoef)[q];
      rsh = reinterpret_cast<const float4*>(rcoef + g.C)[q];
    }
    add4(sh, rsh);                                                   // both shifts in one constant
  }
  const int stride = g.RP;
  int r_end;
  // rev: x was just written front to back by its producer and no statistics pass has touched it since (the 1x1-convolution GEMM
  // took the statistics itself): its TAIL is what the memory-side cache still holds, so the first workgroups start there
  for (int r0 = block_rows(g, &r_end, g.RP, rev != 0) + rp; r0 < r_end; r0 += BN_ROWS_UNROLL * stride) {
    float4 v[BN_ROWS_UNROLL], w[BN_ROWS_UNROLL];
#pragma unroll
    for (int u = 0; u < BN_ROWS_UNROLL; ++u) {
      const int r = r0 + u * stride;
      if (r < r_end) {
        v[u] = x[static_cast<size_t>(r) * g.C4 + q];
        if (RES) w[u] = res[static_cast<size_t>(r) * g.C4 + q];
      }
    }
#pragma unroll
    for (int u = 0; u < BN_ROWS_UNROLL; ++u) {
      const int r = r0 + u * stride;
      if (r < r_end) {
        float4 o;
        o.x = fmaf(v[u].x, sc.x, sh.x); o.y = fmaf(v[u].y, sc.y, sh.y);
        o.z = fmaf(v[u].z, sc.z, sh.z); o.w = fmaf(v[u].w, sc.w, sh.w);
        if (RES && !RAFF) add4(o, w[u]);
        if (RAFF) fma4v(o, w[u], rsc);
        if (RELU && gate_mask != nullptr)             // the ReLU gate of these four channels as one byte: what the backward reads instead of y
          gate_mask[static_cast<size_t>(r) * g.C4 + q] = static_cast<unsigned char>((o.x > 0.f ? 1 : 0) | (o.y > 0.f ? 2 : 0) |
                                                                                    (o.z > 0.f ? 4 : 0) | (o.w > 0.f ? 8 : 0));
        if (RELU) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
        st_nt(y + static_cast<size_t>(r) * g.C4 + q, o);
      }
    }
  }
}

// The saved output of a residual norm is only read for its SIGN (the ReLU gate).  Round 5: the forward writes that gate as one byte
// per float4 (bn_apply's gate_mask) and the backward kernels read the byte instead of the 16 bytes of y -- the two passes of a
// residual norm's backward read 3 + 3 tensors instead of 4 + 4 (the 4-planes-wide block outputs are the largest tensors of the
// step).  ybyte != 0: `y` points at the mask, element i = gate bits of float4 i; the value returned has the gate's signs.
__device__ __forceinline__ float4 load_gate_y(const float4* __restrict__ y, size_t i, int ybyte) {
  if (!ybyte) return y[i];
  const unsigned m = reinterpret_cast<const unsigned char*>(y)[i];
  return make_float4((m & 1u) ? 1.f : 0.f, (m & 2u) ? 1.f : 0.f, (m & 4u) ? 1.f : 0.f, (m & 8u) ? 1.f : 0.f);
}

// dz = dy gated by the ReLU: MASK 0 = no activation, 1 = recompute the pre-activation sign from x
// (no residual: z = x * scale + shift), 2 = read the saved output y (residual variant).
template <int MASK>
__device__ __forceinline__ float4 gate(const float4& dy, const float4& x, const float4& y, const float4& sc,
                                       const float4& sh) {
  if (MASK == 0) return dy;
  float4 o;
  if (MASK == 1) {
    o.x = fmaf(x.x, sc.x, sh.x) > 0.f ? dy.x : 0.f;
    o.y = fmaf(x.y, sc.y, sh.y) > 0.f ? dy.y : 0.f;
    o.z = fmaf(x.z, sc.z, sh.z) > 0.f ? dy.z : 0.f;
    o.w = fmaf(x.w, sc.w, sh.w) > 0.f ? dy.w : 0.f;
  } else {
    o.x = y.x > 0.f ? dy.x : 0.f;
    o.y = y.y > 0.f ? dy.y : 0.f;
    o.z = y.z > 0.f ? dy.z : 0.f;
    o.w = y.w > 0.f ? dy.w : 0.f;
  }
  return o;
}

template <int MASK, int TPB, int UNR>
__global__ __launch_bounds__(TPB) void bn_bwd_reduce(const float4* __restrict__ dy, const float4* __restrict__ dy2, const float4* __restrict__ x,
                                                     const float4* __restrict__ y, const float* __restrict__ coef,
                                                     const float* __restrict__ save_mean,
                                                     const float* __restrict__ save_invstd,
                                                     float* __restrict__ partial, BnGeom g,
                                                     unsigned* __restrict__ tickets = nullptr, int ybyte = 0,
                                                     float4* __restrict__ dz_out = nullptr) {
  // dz_out (round 5, residual norms): the gated gradient dz = gate * (dy + dy2) is WRITTEN here -- it is the gradient of the identity
  // branch anyway (`grad_residual`) -- so that the dx pass reads dz and x instead of dy, dy2, x and the gate: 4 + 3 tensor passes for
  // the pair instead of 3 + 5 when the gradient arrives as two addends
  const int ql = threadIdx.x % g.CH, rp = threadIdx.x / g.CH;
  const int q = blockIdx.y * g.CH + ql;
  const float4 sc = reinterpret_cast<const float4*>(coef)[q];
  const float4 sh = reinterpret_cast<const float4*>(coef + g.C)[q];
  const float4 mu = reinterpret_cast<const float4*>(save_mean)[q];
  const float4 is = reinterpret_cast<const float4*>(save_invstd)[q];
  float4 db = f4(0.f), dg = f4(0.f);
  const int stride = g.RRP;
  int r_end;
  int r0 = block_rows<true>(g, &r_end, g.RRP) + rp, step = UNR * stride, st = 0;
  if (g.STRIPE) {                      // same loop body, rows from the stripe walk (back to front)
    st = stripe_count(g, step) - 1;
    r0 = stripe_row0(st, step) + rp;
    r_end = g.M;
  }
  for (; g.STRIPE ? st >= 0 : r0 < r_end; g.STRIPE ? (--st, r0 = stripe_row0(st, step) + rp) : (r0 += step)) {
    float4 a[UNR], v[UNR], o[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int r = r0 + u * stride;
      a[u] = f4(0.f); v[u] = f4(0.f); o[u] = f4(0.f);
      if (r < r_end) {
        a[u] = dy[static_cast<size_t>(r) * g.C4 + q];
        if (dy2 != nullptr) {                      // the gradient arrives as two addends (a residual junction): summed here, not by a pass of its own
          const float4 b2 = dy2[static_cast<size_t>(r) * g.C4 + q];
          a[u].x += b2.x; a[u].y += b2.y; a[u].z += b2.z; a[u].w += b2.w;
        }
        v[u] = x[static_cast<size_t>(r) * g.C4 + q];
        if (MASK == 2) o[u] = load_gate_y(y, static_cast<size_t>(r) * g.C4 + q, ybyte);
      }
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const float4 dz = gate<MASK>(a[u], v[u], o[u], sc, sh);      // rows past M: dy = 0 -> no contribution
      if (MASK == 2 && dz_out != nullptr) {
        const int r = r0 + u * stride;
        if (r < r_end) st_nt(dz_out + static_cast<size_t>(r) * g.C4 + q, dz);
      }
      float4 xh;
      xh.x = (v[u].x - mu.x) * is.x; xh.y = (v[u].y - mu.y) * is.y;
      xh.z = (v[u].z - mu.z) * is.z; xh.w = (v[u].w - mu.w) * is.w;
      add4(db, dz);
      fma4v(dg, dz, xh);
    }
  }
  block_merge_store<TPB>(db, dg, partial, g.C, g.CH, g.RRP, q, ql, rp);
  if (tickets != nullptr) group_merge<TPB, 2>(partial, g.C, g.CH, g.RRP, q, ql, rp, gridDim.x, tickets);
}

// backward finalize in a consumer's prologue: A, B, Cc of dx = A * dz + B * x + Cc from the merged (sum dz, sum dz * xhat)
struct BnBfin {
  const float* partial;           // nullptr: coefficients come from memory (bn_bwd_finalize ran)
  int nbx, M;
  const float *gamma, *save_mean, *save_invstd;
  float *dgamma, *dbeta;
};

__device__ __forceinline__ void fin_coefs(double s, double sq, double gamma, double mu, double is, int M, float* A, float* B, float* Cc) {
  const double a = gamma * is;
  *A = static_cast<float>(a);
  *B = static_cast<float>(-a * is * sq / M);
  *Cc = static_cast<float>(a * (mu * is * sq - s) / M);
}

__device__ __forceinline__ void fin_backward(const BnBfin& f, int C, int q, bool write, float4* A, float4* B, float4* Cc) {
  d4 m[2];
  merged_rows<2>(f.partial, C, q, f.nbx, m);
  const float4 ga = reinterpret_cast<const float4*>(f.gamma)[q], mu = reinterpret_cast<const float4*>(f.save_mean)[q];
  const float4 is = reinterpret_cast<const float4*>(f.save_invstd)[q];
  fin_coefs(m[0].x, m[1].x, ga.x, mu.x, is.x, f.M, &A->x, &B->x, &Cc->x);
  fin_coefs(m[0].y, m[1].y, ga.y, mu.y, is.y, f.M, &A->y, &B->y, &Cc->y);
  fin_coefs(m[0].z, m[1].z, ga.z, mu.z, is.z, f.M, &A->z, &B->z, &Cc->z);
  fin_coefs(m[0].w, m[1].w, ga.w, mu.w, is.w, f.M, &A->w, &B->w, &Cc->w);
  if (write) {
    reinterpret_cast<float4*>(f.dbeta)[q] = make_float4(static_cast<float>(m[0].x), static_cast<float>(m[0].y), static_cast<float>(m[0].z), static_cast<float>(m[0].w));
    reinterpret_cast<float4*>(f.dgamma)[q] = make_float4(static_cast<float>(m[1].x), static_cast<float>(m[1].y), static_cast<float>(m[1].z), static_cast<float>(m[1].w));
  }
}

// dgamma, dbeta and the per-channel coefficients of dx = A * dz + B * x + Cc  (bcoef: [0] A, [1] B, [2] Cc)
__global__ __launch_bounds__(256) void bn_bwd_finalize(const float* __restrict__ partial, int nbx, int M, int C,
                                                       const float* __restrict__ gamma,
                                                       const float* __restrict__ save_mean,
                                                       const float* __restrict__ save_invstd,
                                                       float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                       float* __restrict__ bcoef) {
  __shared__ double ls[4][BN_FIN_CH], lq[4][BN_FIN_CH];
  const int cl = threadIdx.x % BN_FIN_CH, ph = threadIdx.x / BN_FIN_CH;
  const int c = blockIdx.x * BN_FIN_CH + cl;
  double s = 0.0, sq = 0.0;
  if (c < C) {
    int b = ph;
    for (; b + 7 * BN_FIN_PH < nbx; b += 8 * BN_FIN_PH) {      // 16 independent loads in flight: <= 512 partial rows = ONE trip
      float a[8], q2[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        a[u] = partial[(static_cast<size_t>(b + u * BN_FIN_PH) * 2 + 0) * C + c];
        q2[u] = partial[(static_cast<size_t>(b + u * BN_FIN_PH) * 2 + 1) * C + c];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) { s += static_cast<double>(a[u]); sq += static_cast<double>(q2[u]); }
    }
    for (; b < nbx; b += BN_FIN_PH) {
      s += static_cast<double>(partial[(static_cast<size_t>(b) * 2 + 0) * C + c]);
      sq += static_cast<double>(partial[(static_cast<size_t>(b) * 2 + 1) * C + c]);
    }
  }
  // 64 phases -> 1: xor-tree over the 16 phases a wave holds (lanes 4 apart), then the 4 waves through LDS -- a fixed
  // order, and 4 shuffle levels + 3 adds instead of a 63-step serial walk through LDS (3 us of a 8 us kernel)
#pragma unroll
  for (int m = BN_FIN_CH; m < 64; m <<= 1) { s += __shfl_xor(s, m); sq += __shfl_xor(sq, m); }
  if ((threadIdx.x & 63) < BN_FIN_CH) { ls[threadIdx.x >> 6][cl] = s; lq[threadIdx.x >> 6][cl] = sq; }
  __syncthreads();
  if (ph == 0 && c < C) {
    s = ((ls[0][cl] + ls[1][cl]) + ls[2][cl]) + ls[3][cl];
    sq = ((lq[0][cl] + lq[1][cl]) + lq[2][cl]) + lq[3][cl];
    dbeta[c] = static_cast<float>(s);
    dgamma[c] = static_cast<float>(sq);
    const double is = save_invstd[c], mu = save_mean[c];
    const double A = static_cast<double>(gamma[c]) * is;
    bcoef[c] = static_cast<float>(A);
    bcoef[C + c] = static_cast<float>(-A * is * sq / M);
    bcoef[2 * C + c] = static_cast<float>(A * (mu * is * sq - s) / M);
  }
}

template <int MASK, bool DRES>
__global__ __launch_bounds__(256) void bn_bwd_dx(const float4* __restrict__ dy, const float4* __restrict__ dy2, const float4* __restrict__ x,
                                                 const float4* __restrict__ y, const float* __restrict__ coef,
                                                 const float* __restrict__ bcoef, float4* __restrict__ dx,
                                                 float4* __restrict__ dres, BnGeom g, BnBfin fin, int ybyte = 0) {
  const int ql = threadIdx.x % g.CH, rp = threadIdx.x / g.CH;
  const int q = blockIdx.y * g.CH + ql;
  const float4 sc = reinterpret_cast<const float4*>(coef)[q];
  const float4 sh = reinterpret_cast<const float4*>(coef + g.C)[q];
  float4 A, Bc, Cc;
  if (fin.partial != nullptr) {
    fin_backward(fin, g.C, q, blockIdx.x == 0 && rp == 0, &A, &Bc, &Cc);
  } else {
    A = reinterpret_cast<const float4*>(bcoef)[q];
    Bc = reinterpret_cast<const float4*>(bcoef + g.C)[q];
    Cc = reinterpret_cast<const float4*>(bcoef + 2 * g.C)[q];
  }
  const int stride = g.RP;
  int r_end;
  for (int r0 = block_rows(g, &r_end, g.RP) + rp; r0 < r_end; r0 += BN_ROWS_UNROLL * stride) {
    float4 a[BN_ROWS_UNROLL], v[BN_ROWS_UNROLL], o[BN_ROWS_UNROLL];
#pragma unroll
    for (int u = 0; u < BN_ROWS_UNROLL; ++u) {
      const int r = r0 + u * stride;
      a[u] = f4(0.f); v[u] = f4(0.f); o[u] = f4(0.f);
      if (r < r_end) {
        a[u] = dy[static_cast<size_t>(r) * g.C4 + q];
        if (dy2 != nullptr) {                      // the gradient arrives as two addends (a residual junction): summed here, not by a pass of its own
          const float4 b2 = dy2[static_cast<size_t>(r) * g.C4 + q];
          a[u].x += b2.x; a[u].y += b2.y; a[u].z += b2.z; a[u].w += b2.w;
        }
        v[u] = x[static_cast<size_t>(r) * g.C4 + q];
        if (MASK == 2) o[u] = load_gate_y(y, static_cast<size_t>(r) * g.C4 + q, ybyte);
      }
    }
#pragma unroll
    for (int u = 0; u < BN_ROWS_UNROLL; ++u) {
      const int r = r0 + u * stride;
      if (r < r_end) {
        const float4 dz = gate<MASK>(a[u], v[u], o[u], sc, sh);
        float4 d;
        d.x = fmaf(A.x, dz.x, fmaf(Bc.x, v[u].x, Cc.x));
        d.y = fmaf(A.y, dz.y, fmaf(Bc.y, v[u].y, Cc.y));
        d.z = fmaf(A.z, dz.z, fmaf(Bc.z, v[u].z, Cc.z));
        d.w = fmaf(A.w, dz.w, fmaf(Bc.w, v[u].w, Cc.w));
        st_nt(dx + static_cast<size_t>(r) * g.C4 + q, d);
        if (DRES) st_nt(dres + static_cast<size_t>(r) * g.C4 + q, dz);
      }
    }
  }
}

// ---- two BatchNorms feeding one add (+ ReLU): out = relu(bn(x) + bn_d(xd)) -- the main and the `downsample` branch of a
// stage-first residual block.  dz = dout * [out > 0] is shared, so one reduction pass gives sum dz, sum dz * xhat and
// sum dz * xhat_d (partial rows of THREE values per channel) and one pass writes both input gradients: 4 reads + 2 writes
// and 4 reads, where two chained single-BatchNorm backwards take 8 reads + 3 writes (the gated gradient written and re-read).
template <int TPB>
__device__ __forceinline__ void block_merge_store3(float4 a, float4 b, float4 c, float* __restrict__ partial, int C, int CH,
                                                   int RP, int q, int ql, int rp) {
  __shared__ float4 sa[TPB], sb[TPB], sc[TPB];
  sa[threadIdx.x] = a;
  sb[threadIdx.x] = b;
  sc[threadIdx.x] = c;
  __syncthreads();
  for (int st = RP >> 1; st > 0; st >>= 1) {
    if (rp < st) {
      add4(sa[threadIdx.x], sa[threadIdx.x + st * CH]);
      add4(sb[threadIdx.x], sb[threadIdx.x + st * CH]);
      add4(sc[threadIdx.x], sc[threadIdx.x + st * CH]);
    }
    __syncthreads();
  }
  if (rp == 0) {
    reinterpret_cast<float4*>(partial + (static_cast<size_t>(blockIdx.x) * 3 + 0) * C)[q] = sa[ql];
    reinterpret_cast<float4*>(partial + (static_cast<size_t>(blockIdx.x) * 3 + 1) * C)[q] = sb[ql];
    reinterpret_cast<float4*>(partial + (static_cast<size_t>(blockIdx.x) * 3 + 2) * C)[q] = sc[ql];
  }
}

template <bool RELU, int TPB, int UNR>
__global__ __launch_bounds__(TPB) void bn_bwd_reduce_dual(const float4* __restrict__ dy, const float4* __restrict__ dy2, const float4* __restrict__ x,
                                                          const float4* __restrict__ xd, const float4* __restrict__ y,
                                                          const float* __restrict__ mean, const float* __restrict__ invstd,
                                                          const float* __restrict__ mean_d,
                                                          const float* __restrict__ invstd_d, float* __restrict__ partial,
                                                          BnGeom g, unsigned* __restrict__ tickets = nullptr, int ybyte = 0) {
  const int ql = threadIdx.x % g.CH, rp = threadIdx.x / g.CH;
  const int q = blockIdx.y * g.CH + ql;
  const float4 mu = reinterpret_cast<const float4*>(mean)[q], is = reinterpret_cast<const float4*>(invstd)[q];
  const float4 mud = reinterpret_cast<const float4*>(mean_d)[q], isd = reinterpret_cast<const float4*>(invstd_d)[q];
  float4 db = f4(0.f), dg = f4(0.f), dgd = f4(0.f);
  const int stride = g.RRP;
  int r_end;
  int r0 = block_rows<true>(g, &r_end, g.RRP) + rp, step = UNR * stride, st = 0;
  if (g.STRIPE) {
    st = stripe_count(g, step) - 1;
    r0 = stripe_row0(st, step) + rp;
    r_end = g.M;
  }
  for (; g.STRIPE ? st >= 0 : r0 < r_end; g.STRIPE ? (--st, r0 = stripe_row0(st, step) + rp) : (r0 += step)) {
    float4 a[UNR], v[UNR], vd[UNR], o[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int r = r0 + u * stride;
      a[u] = f4(0.f); v[u] = f4(0.f); vd[u] = f4(0.f); o[u] = f4(0.f);
      if (r < r_end) {
        a[u] = dy[static_cast<size_t>(r) * g.C4 + q];
        if (dy2 != nullptr) {                      // the gradient arrives as two addends (a residual junction): summed here, not by a pass of its own
          const float4 b2 = dy2[static_cast<size_t>(r) * g.C4 + q];
          a[u].x += b2.x; a[u].y += b2.y; a[u].z += b2.z; a[u].w += b2.w;
        }
        v[u] = x[static_cast<size_t>(r) * g.C4 + q];
        vd[u] = xd[static_cast<size_t>(r) * g.C4 + q];
        if (RELU) o[u] = load_gate_y(y, static_cast<size_t>(r) * g.C4 + q, ybyte);
      }
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const float4 dz = gate<RELU ? 2 : 0>(a[u], v[u], o[u], mu, mu);
      float4 xh, xhd;
      xh.x = (v[u].x - mu.x) * is.x; xh.y = (v[u].y - mu.y) * is.y; xh.z = (v[u].z - mu.z) * is.z; xh.w = (v[u].w - mu.w) * is.w;
      xhd.x = (vd[u].x - mud.x) * isd.x; xhd.y = (vd[u].y - mud.y) * isd.y;
      xhd.z = (vd[u].z - mud.z) * isd.z; xhd.w = (vd[u].w - mud.w) * isd.w;
      add4(db, dz);
      fma4v(dg, dz, xh);
      fma4v(dgd, dz, xhd);
    }
  }
  block_merge_store3<TPB>(db, dg, dgd, partial, g.C, g.CH, g.RRP, q, ql, rp);
  if (tickets != nullptr) group_merge<TPB, 3>(partial, g.C, g.CH, g.RRP, q, ql, rp, gridDim.x, tickets);
}

// dgamma / dbeta of both norms (dbeta is the same sum for both) and their dx coefficients: bcoef [0..2] main, [3..5] branch
__global__ __launch_bounds__(256) void bn_bwd_finalize_dual(const float* __restrict__ partial, int nbx, int M, int C,
                                                            const float* __restrict__ gamma, const float* __restrict__ mean,
                                                            const float* __restrict__ invstd,
                                                            const float* __restrict__ gamma_d,
                                                            const float* __restrict__ mean_d,
                                                            const float* __restrict__ invstd_d, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta, float* __restrict__ dgamma_d,
                                                            float* __restrict__ dbeta_d, float* __restrict__ bcoef) {
  __shared__ double l0[4][BN_FIN_CH], l1[4][BN_FIN_CH], l2[4][BN_FIN_CH];
  const int cl = threadIdx.x % BN_FIN_CH, ph = threadIdx.x / BN_FIN_CH;
  const int c = blockIdx.x * BN_FIN_CH + cl;
  double s = 0.0, sq = 0.0, sd = 0.0;
  if (c < C) {
    for (int b = ph; b < nbx; b += BN_FIN_PH) {
      s += static_cast<double>(partial[(static_cast<size_t>(b) * 3 + 0) * C + c]);
      sq += static_cast<double>(partial[(static_cast<size_t>(b) * 3 + 1) * C + c]);
      sd += static_cast<double>(partial[(static_cast<size_t>(b) * 3 + 2) * C + c]);
    }
  }
#pragma unroll
  for (int m = BN_FIN_CH; m < 64; m <<= 1) { s += __shfl_xor(s, m); sq += __shfl_xor(sq, m); sd += __shfl_xor(sd, m); }
  if ((threadIdx.x & 63) < BN_FIN_CH) { l0[threadIdx.x >> 6][cl] = s; l1[threadIdx.x >> 6][cl] = sq; l2[threadIdx.x >> 6][cl] = sd; }
  __syncthreads();
  if (ph == 0 && c < C) {
    s = ((l0[0][cl] + l0[1][cl]) + l0[2][cl]) + l0[3][cl];
    sq = ((l1[0][cl] + l1[1][cl]) + l1[2][cl]) + l1[3][cl];
    sd = ((l2[0][cl] + l2[1][cl]) + l2[2][cl]) + l2[3][cl];
    dbeta[c] = static_cast<float>(s);
    dbeta_d[c] = static_cast<float>(s);
    dgamma[c] = static_cast<float>(sq);
    dgamma_d[c] = static_cast<float>(sd);
    double is = invstd[c], mu = mean[c], A = static_cast<double>(gamma[c]) * is;
    bcoef[c] = static_cast<float>(A);
    bcoef[C + c] = static_cast<float>(-A * is * sq / M);
    bcoef[2 * C + c] = static_cast<float>(A * (mu * is * sq - s) / M);
    is = invstd_d[c]; mu = mean_d[c]; A = static_cast<double>(gamma_d[c]) * is;
    bcoef[3 * C + c] = static_cast<float>(A);
    bcoef[4 * C + c] = static_cast<float>(-A * is * sd / M);
    bcoef[5 * C + c] = static_cast<float>(A * (mu * is * sd - s) / M);
  }
}

template <bool RELU>
__global__ __launch_bounds__(256) void bn_bwd_dx_dual(const float4* __restrict__ dy, const float4* __restrict__ dy2, const float4* __restrict__ x,
                                                      const float4* __restrict__ xd, const float4* __restrict__ y,
                                                      const float* __restrict__ bcoef, float4* __restrict__ dx,
                                                      float4* __restrict__ dxd, BnGeom g, BnBfin fin, BnBfin fin_d, int ybyte = 0) {
  const int ql = threadIdx.x % g.CH, rp = threadIdx.x / g.CH;
  const int q = blockIdx.y * g.CH + ql;
  float4 A, Bc, Cc, Ad, Bd, Cd;
  if (fin.partial != nullptr) {                          // (sum dz, sum dz xhat, sum dz xhat_d) rows of three values
    d4 m[3];
    merged_rows<3>(fin.partial, g.C, q, fin.nbx, m);
    const float4 ga = reinterpret_cast<const float4*>(fin.gamma)[q], mu = reinterpret_cast<const float4*>(fin.save_mean)[q];
    const float4 is = reinterpret_cast<const float4*>(fin.save_invstd)[q];
    const float4 gd = reinterpret_cast<const float4*>(fin_d.gamma)[q], mud = reinterpret_cast<const float4*>(fin_d.save_mean)[q];
    const float4 isd = reinterpret_cast<const float4*>(fin_d.save_invstd)[q];
    fin_coefs(m[0].x, m[1].x, ga.x, mu.x, is.x, fin.M, &A.x, &Bc.x, &Cc.x);  fin_coefs(m[0].x, m[2].x, gd.x, mud.x, isd.x, fin.M, &Ad.x, &Bd.x, &Cd.x);
    fin_coefs(m[0].y, m[1].y, ga.y, mu.y, is.y, fin.M, &A.y, &Bc.y, &Cc.y);  fin_coefs(m[0].y, m[2].y, gd.y, mud.y, isd.y, fin.M, &Ad.y, &Bd.y, &Cd.y);
    fin_coefs(m[0].z, m[1].z, ga.z, mu.z, is.z, fin.M, &A.z, &Bc.z, &Cc.z);  fin_coefs(m[0].z, m[2].z, gd.z, mud.z, isd.z, fin.M, &Ad.z, &Bd.z, &Cd.z);
    fin_coefs(m[0].w, m[1].w, ga.w, mu.w, is.w, fin.M, &A.w, &Bc.w, &Cc.w);  fin_coefs(m[0].w, m[2].w, gd.w, mud.w, isd.w, fin.M, &Ad.w, &Bd.w, &Cd.w);
    if (blockIdx.x == 0 && rp == 0) {
      const float4 s4 = make_float4(static_cast<float>(m[0].x), static_cast<float>(m[0].y), static_cast<float>(m[0].z), static_cast<float>(m[0].w));
      reinterpret_cast<float4*>(fin.dbeta)[q] = s4;
      reinterpret_cast<float4*>(fin_d.dbeta)[q] = s4;
      reinterpret_cast<float4*>(fin.dgamma)[q] = make_float4(static_cast<float>(m[1].x), static_cast<float>(m[1].y), static_cast<float>(m[1].z), static_cast<float>(m[1].w));
      reinterpret_cast<float4*>(fin_d.dgamma)[q] = make_float4(static_cast<float>(m[2].x), static_cast<float>(m[2].y), static_cast<float>(m[2].z), static_cast<float>(m[2].w));
    }
  } else {
    A = reinterpret_cast<const float4*>(bcoef)[q]; Bc = reinterpret_cast<const float4*>(bcoef + g.C)[q];
    Cc = reinterpret_cast<const float4*>(bcoef + 2 * g.C)[q];
    Ad = reinterpret_cast<const float4*>(bcoef + 3 * g.C)[q]; Bd = reinterpret_cast<const float4*>(bcoef + 4 * g.C)[q];
    Cd = reinterpret_cast<const float4*>(bcoef + 5 * g.C)[q];
  }
  const int stride = g.RP;
  int r_end;
  for (int r0 = block_rows(g, &r_end, g.RP) + rp; r0 < r_end; r0 += BN_ROWS_UNROLL * stride) {
    float4 a[BN_ROWS_UNROLL], v[BN_ROWS_UNROLL], vd[BN_ROWS_UNROLL], o[BN_ROWS_UNROLL];
#pragma unroll
    for (int u = 0; u < BN_ROWS_UNROLL; ++u) {
      const int r = r0 + u * stride;
      a[u] = f4(0.f); v[u] = f4(0.f); vd[u] = f4(0.f); o[u] = f4(0.f);
      if (r < r_end) {
        a[u] = dy[static_cast<size_t>(r) * g.C4 + q];
        if (dy2 != nullptr) {                      // the gradient arrives as two addends (a residual junction): summed here, not by a pass of its own
          const float4 b2 = dy2[static_cast<size_t>(r) * g.C4 + q];
          a[u].x += b2.x; a[u].y += b2.y; a[u].z += b2.z; a[u].w += b2.w;
        }
        v[u] = x[static_cast<size_t>(r) * g.C4 + q];
        vd[u] = xd[static_cast<size_t>(r) * g.C4 + q];
        if (RELU) o[u] = load_gate_y(y, static_cast<size_t>(r) * g.C4 + q, ybyte);
      }
    }
#pragma unroll
    for (int u = 0; u < BN_ROWS_UNROLL; ++u) {
      const int r = r0 + u * stride;
      if (r < r_end) {
        const float4 dz = gate<RELU ? 2 : 0>(a[u], v[u], o[u], A, A);
        float4 d, e;
        d.x = fmaf(A.x, dz.x, fmaf(Bc.x, v[u].x, Cc.x)); d.y = fmaf(A.y, dz.y, fmaf(Bc.y, v[u].y, Cc.y));
        d.z = fmaf(A.z, dz.z, fmaf(Bc.z, v[u].z, Cc.z)); d.w = fmaf(A.w, dz.w, fmaf(Bc.w, v[u].w, Cc.w));
        e.x = fmaf(Ad.x, dz.x, fmaf(Bd.x, vd[u].x, Cd.x)); e.y = fmaf(Ad.y, dz.y, fmaf(Bd.y, vd[u].y, Cd.y));
        e.z = fmaf(Ad.z, dz.z, fmaf(Bd.z, vd[u].z, Cd.z)); e.w = fmaf(Ad.w, dz.w, fmaf(Bd.w, vd[u].w, Cd.w));
        st_nt(dx + static_cast<size_t>(r) * g.C4 + q, d);
        st_nt(dxd + static_cast<size_t>(r) * g.C4 + q, e);
      }
    }
  }
}

// eval mode: scale/shift from the running statistics (torch batch_norm(training=False))
__global__ __launch_bounds__(256) void bn_infer_coef(const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     const float* __restrict__ mean, const float* __restrict__ var,
                                                     float eps, int C, float* __restrict__ coef) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float invstd = static_cast<float>(1.0 / sqrt(static_cast<double>(var[c]) + static_cast<double>(eps)));
  const float sc = gamma[c] * invstd;
  coef[c] = sc;
  coef[C + c] = fmaf(-mean[c], sc, beta[c]);
}

#define BN_DISPATCH_TPB_UNR(CALL)                                     \
  do {                                                                \
    if (g.RTPB == 1024) { if (g.RUNR == 8) CALL(1024, 8); else CALL(1024, 4); }   \
    else if (g.RTPB == 512) { if (g.RUNR == 8) CALL(512, 8); else CALL(512, 4); } \
    else { if (g.RUNR == 8) CALL(256, 8); else CALL(256, 4); }        \
  } while (0)

void launch_stats(const BnGeom& g, dim3 grid, hipStream_t s, const float4* x, float* partial, unsigned* tickets = nullptr) {
#define BN_CALL(T, U) hipLaunchKernelGGL((bn_stats<T, U>), grid, dim3(T), 0, s, x, partial, g, tickets)
  BN_DISPATCH_TPB_UNR(BN_CALL);
#undef BN_CALL
}

template <int MASK>
void launch_bwd_reduce(const BnGeom& g, dim3 grid, hipStream_t s, const float4* dy, const float4* dy2, const float4* x,
                       const float4* y, const float* coef, const float* mean, const float* invstd, float* partial,
                       unsigned* tickets = nullptr, int ybyte = 0, float4* dz_out = nullptr) {
#define BN_CALL(T, U) \
  hipLaunchKernelGGL((bn_bwd_reduce<MASK, T, U>), grid, dim3(T), 0, s, dy, dy2, x, y, coef, mean, invstd, partial, g, tickets, ybyte, dz_out)
  BN_DISPATCH_TPB_UNR(BN_CALL);
#undef BN_CALL
}

// ticket slots of one call (see group_merge): a rotating window of the device array; every slot returns to zero after its use
unsigned* bn_tickets(int sets) {
  static unsigned* base = [] {
    void* p = nullptr;
    return hipGetSymbolAddress(&p, HIP_SYMBOL(g_bn_tickets)) == hipSuccess ? static_cast<unsigned*>(p) : nullptr;
  }();
  static std::atomic<unsigned> next{0};
  constexpr unsigned per = 2 * BN_MAX_GROUPS;               // up to two column chunks per set
  if (base == nullptr) return nullptr;
  const unsigned first = next.fetch_add(per * static_cast<unsigned>(sets)) % (BN_TICKET_SLOTS - 4 * per);
  return base + first / per * per;
}
bool bn_ticket_ok(const BnGeom& g) {
  // opt-in (DBEV_BN_TICKET=1, read per call so that a test can switch it): measured SLOWER than the finalize launches it removes --
  // 149.0 vs 146.6 ms per step, bn_stats 4.5 vs 2.6+1.1 ms -- because every producer workgroup pays a release fence (an L2 write-back
  // on gfx950) and every consumer workgroup a dependent 16-row merge before its first load
  const char* e = getenv("DBEV_BN_TICKET");
  return e != nullptr && atoi(e) != 0 && g.GY <= 2 && g.NBX <= BN_GROUP * BN_MAX_GROUPS;
}

// Thousands of partial rows (a convolution epilogue leaves one per 128 pixels: 4224 for the first stage of the image backbone) are
// too many for bn_finalize's C / 4 workgroups -- 28 us for 8.6 MB read as 16-byte pieces.  bn_fold sums `per` consecutive rows of the
// [nrows][2 C] table (fp64, row order) into one fp64 row: coalesced, one workgroup per 256 columns x `per` rows; bn_finalize<double>
// then merges the few folded rows exactly as it merges float rows (fp64 sums either way: the result differs from the one-level
// merge only by fp64 rounding).
__global__ __launch_bounds__(256) void bn_fold(const float* __restrict__ rows, int nrows, int per, int C2, double* __restrict__ fold) {
  const int col = blockIdx.x * 256 + threadIdx.x;
  if (col >= C2) return;
  const int r0 = blockIdx.y * per;
  const int r1 = r0 + per < nrows ? r0 + per : nrows;
  double s = 0.0;
  int r = r0;
  for (; r + 16 <= r1; r += 16) {
    float v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = rows[static_cast<size_t>(r + u) * C2 + col];
#pragma unroll
    for (int u = 0; u < 16; ++u) s += static_cast<double>(v[u]);
  }
  for (; r < r1; ++r) s += static_cast<double>(rows[static_cast<size_t>(r) * C2 + col]);
  fold[static_cast<size_t>(blockIdx.y) * C2 + col] = s;
}
constexpr int BN_FOLD_MIN_ROWS = 1024, BN_FOLD_PER = 32;

// finalize of one norm from `nrows` partial rows; `scratch` (scratch_bytes): room for the folded rows (the call's own partial-row area,
// idle when the rows come from a convolution's epilogue)
void launch_finalize(hipStream_t s, const float* rows, int nrows, int M, int C, const float* gamma, const float* beta, float* running_mean,
                     float* running_var, float momentum, float eps, float* save_mean, float* save_invstd, float* coef,
                     long long* num_batches_tracked, void* scratch, size_t scratch_bytes) {
  static const bool fold_on = getenv("DBEV_BN_FOLD") == nullptr || atoi(getenv("DBEV_BN_FOLD")) != 0;
  const size_t row_bytes = sizeof(double) * 2 * static_cast<size_t>(C);
  const size_t room = scratch != nullptr && scratch != rows ? scratch_bytes / row_bytes : 0;
  if (fold_on && nrows >= BN_FOLD_MIN_ROWS && room >= 2) {
    int per = BN_FOLD_PER;
    if (static_cast<size_t>(dbev_ceil_div(nrows, per)) > room) per = dbev_ceil_div(nrows, static_cast<int>(room));
    const int R = dbev_ceil_div(nrows, per);
    double* fold = static_cast<double*>(scratch);
    DbevKt kt(DBEV_K_BN_FINALIZE, 8LL * nrows * C, s);
    hipLaunchKernelGGL(bn_fold, dim3(dbev_ceil_div(2 * C, 256), R), dim3(256), 0, s, rows, nrows, per, 2 * C, fold);
    hipLaunchKernelGGL(bn_finalize<double>, dim3(dbev_ceil_div(C, BN_FIN_CH)), dim3(256), 0, s, fold, R, M, C, gamma, beta, running_mean,
                       running_var, momentum, eps, save_mean, save_invstd, coef, num_batches_tracked);
    return;
  }
  DbevKt kt(DBEV_K_BN_FINALIZE, 8LL * nrows * C, s);
  hipLaunchKernelGGL(bn_finalize<float>, dim3(dbev_ceil_div(C, BN_FIN_CH)), dim3(256), 0, s, rows, nrows, M, C, gamma, beta, running_mean,
                     running_var, momentum, eps, save_mean, save_invstd, coef, num_batches_tracked);
}

struct BnWs { size_t partial, total; };
BnWs bn_ws(const BnGeom& g) {
  BnWs w;
  w.partial = 0;
  w.total = sizeof(float) * static_cast<size_t>(g.NBX) * 2 * g.C;
  return w;
}

}  // namespace

extern "C" size_t dbev_bn_act_workspace_bytes(long long M, int C) {
  BnGeom g;
  if (!bn_geom(M, C, &g)) return 0;
  return bn_ws(g).total;
}

extern "C" int dbev_bn_act_train_forward(const float* x, const float* residual, const float* gamma, const float* beta,
                                         float* running_mean, float* running_var, long long* num_batches_tracked,
                                         float momentum, float eps, int relu, float* y, float* save_mean,
                                         float* save_invstd, float* save_scale_shift, long long M, int C,
                                         void* workspace, size_t workspace_bytes, dbevStream_t stream) {
  return dbev_bn_act_train_forward_pre(x, residual, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps, relu,
                                       y, save_mean, save_invstd, save_scale_shift, M, C, nullptr, 0, workspace, workspace_bytes,
                                       stream);
}

// ... with the per-channel partial sums (sum x, sum x^2) already taken by the kernel that PRODUCED x (dbev_conv1x1_forward's
// epilogue: `stats_partial` f32[partial_rows, 2, C]): the statistics pass over x does not run.  stats_partial == NULL: as above.
extern "C" int dbev_bn_act_train_forward_pre(const float* x, const float* residual, const float* gamma, const float* beta,
                                             float* running_mean, float* running_var, long long* num_batches_tracked,
                                             float momentum, float eps, int relu, float* y, float* save_mean,
                                             float* save_invstd, float* save_scale_shift, long long M, int C,
                                             const float* stats_partial, int partial_rows, void* workspace,
                                             size_t workspace_bytes, dbevStream_t stream) {
  return dbev_bn_act_train_forward_mask(x, residual, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps, relu, y,
                                        save_mean, save_invstd, save_scale_shift, M, C, stats_partial, partial_rows, nullptr, workspace,
                                        workspace_bytes, stream);
}

// ... and with the ReLU gate of a residual norm written as one byte per four channels (relu_mask u8[M * C / 4], NULL: not written):
// dbev_bn_act_backward3(..., y_is_mask = 1) reads it instead of the saved output
extern "C" int dbev_bn_act_train_forward_mask(const float* x, const float* residual, const float* gamma, const float* beta,
                                              float* running_mean, float* running_var, long long* num_batches_tracked,
                                              float momentum, float eps, int relu, float* y, float* save_mean,
                                              float* save_invstd, float* save_scale_shift, long long M, int C,
                                              const float* stats_partial, int partial_rows, unsigned char* relu_mask,
                                              void* workspace, size_t workspace_bytes, dbevStream_t stream) {
  BnGeom g;
  if (!bn_geom(M, C, &g)) return DBEV_EINVAL;
  if (stats_partial != nullptr && partial_rows <= 0) return DBEV_EINVAL;
  // y == NULL: statistics + finalize only (mean / invstd / scale / shift, running statistics) -- the caller applies them inside its
  // own kernel (dbev_depth_head_forward)
  if (x == nullptr || gamma == nullptr || beta == nullptr || save_mean == nullptr ||
      save_invstd == nullptr || save_scale_shift == nullptr || workspace == nullptr ||
      workspace_bytes < bn_ws(g).total || (running_mean == nullptr) != (running_var == nullptr) ||
      (y == nullptr && residual != nullptr))
    return DBEV_EINVAL;
  hipStream_t s = dbev_stream(stream);
  float* partial = static_cast<float*>(workspace);
  const dim3 grid(g.NBX, g.GY);
  const long long T = 4LL * g.M * C;                       // bytes of one full-tensor pass
  unsigned* tk = (stats_partial == nullptr && y != nullptr && bn_ticket_ok(g)) ? bn_tickets(1) : nullptr;
  int nrows = g.NBX;
  const float* rows = partial;
  if (stats_partial != nullptr) {
    nrows = partial_rows;
    rows = stats_partial;
  } else {
    DbevKt kt(DBEV_K_BN_STATS, T, s);
    launch_stats(g, grid, s, reinterpret_cast<const float4*>(x), partial, tk);
  }
  BnFin fin{};                                               // tickets: the apply pass merges the group rows itself
  if (tk != nullptr) {
    fin = BnFin{partial, g.NBX, g.M, gamma, beta, running_mean, running_var, momentum, eps, save_mean, save_invstd, save_scale_shift,
                num_batches_tracked};
  } else {
    launch_finalize(s, rows, nrows, g.M, C, gamma, beta, running_mean, running_var, momentum, eps, save_mean, save_invstd,
                    save_scale_shift, num_batches_tracked, partial, sizeof(float) * static_cast<size_t>(g.NBX) * 2 * C);
  }
  if (y == nullptr) {
    DBEV_LAUNCH_CHECK();
    return 0;
  }
  const float4* x4 = reinterpret_cast<const float4*>(x);
  const float4* r4 = reinterpret_cast<const float4*>(residual);
  float4* y4 = reinterpret_cast<float4*>(y);
  // the apply pass has no reduction: use enough workgroups for latency hiding on its own
  const long long tiles = (M + static_cast<long long>(g.RP) * BN_ROWS_UNROLL - 1) / (static_cast<long long>(g.RP) * BN_ROWS_UNROLL);
  const long long cap = static_cast<long long>(DBEV_MAX_GRID) * 2 / g.GY;
  const dim3 agrid(static_cast<unsigned>(tiles < cap ? tiles : (cap < 1 ? 1 : cap)), g.GY);
  {
    DbevKt kt(residual != nullptr ? DBEV_K_BN_APPLY_RES : DBEV_K_BN_APPLY, T * (residual != nullptr ? 3 : 2), s);
    const float* none = nullptr;
    const int rev = stats_partial != nullptr ? 1 : 0;
    const BnFin nofin{};
    if (residual != nullptr) {
      if (relu) hipLaunchKernelGGL((bn_apply<true, true>), agrid, dim3(256), 0, s, x4, r4, save_scale_shift, y4, g, none, rev, fin, nofin, relu_mask);
      else hipLaunchKernelGGL((bn_apply<true, false>), agrid, dim3(256), 0, s, x4, r4, save_scale_shift, y4, g, none, rev, fin, nofin);
    } else {
      if (relu) hipLaunchKernelGGL((bn_apply<false, true>), agrid, dim3(256), 0, s, x4, r4, save_scale_shift, y4, g, none, rev, fin, nofin);
      else hipLaunchKernelGGL((bn_apply<false, false>), agrid, dim3(256), 0, s, x4, r4, save_scale_shift, y4, g, none, rev, fin, nofin);
    }
  }
  DBEV_LAUNCH_CHECK();
  return 0;
}

extern "C" int dbev_bn_act_infer(const float* x, const float* residual, const float* gamma, const float* beta,
                                 const float* running_mean, const float* running_var, float eps, int relu, float* y,
                                 long long M, int C, void* workspace, size_t workspace_bytes, dbevStream_t stream) {
  BnGeom g;
  if (!bn_geom(M, C, &g)) return DBEV_EINVAL;
  if (x == nullptr || gamma == nullptr || beta == nullptr || running_mean == nullptr || running_var == nullptr ||
      y == nullptr || workspace == nullptr || workspace_bytes < sizeof(float) * 2 * static_cast<size_t>(C))
    return DBEV_EINVAL;
  hipStream_t s = dbev_stream(stream);
  float* coef = static_cast<float*>(workspace);
  hipLaunchKernelGGL(bn_infer_coef, dim3(dbev_ceil_div(C, 256)), dim3(256), 0, s, gamma, beta, running_mean,
                     running_var, eps, C, coef);
  const float4* x4 = reinterpret_cast<const float4*>(x);
  const float4* r4 = reinterpret_cast<const float4*>(residual);
  float4* y4 = reinterpret_cast<float4*>(y);
  const long long tiles = (M + static_cast<long long>(g.RP) * BN_ROWS_UNROLL - 1) / (static_cast<long long>(g.RP) * BN_ROWS_UNROLL);
  const long long cap = static_cast<long long>(DBEV_MAX_GRID) * 2 / g.GY;
  const dim3 agrid(static_cast<unsigned>(tiles < cap ? tiles : (cap < 1 ? 1 : cap)), g.GY);
  const float* none = nullptr;
  const BnFin nofin{};
  if (residual != nullptr) {
    if (relu) hipLaunchKernelGGL((bn_apply<true, true>), agrid, dim3(256), 0, s, x4, r4, coef, y4, g, none, 0, nofin, nofin);
    else hipLaunchKernelGGL((bn_apply<true, false>), agrid, dim3(256), 0, s, x4, r4, coef, y4, g, none, 0, nofin, nofin);
  } else {
    if (relu) hipLaunchKernelGGL((bn_apply<false, true>), agrid, dim3(256), 0, s, x4, r4, coef, y4, g, none, 0, nofin, nofin);
    else hipLaunchKernelGGL((bn_apply<false, false>), agrid, dim3(256), 0, s, x4, r4, coef, y4, g, none, 0, nofin, nofin);
  }
  DBEV_LAUNCH_CHECK();
  return 0;
}

// eval mode with the coefficients kept by the caller (a frozen teacher's norms: scale / shift never change between steps, the
// coefficient launch of dbev_bn_act_infer is 37 x 5 us of the step): dbev_bn_infer_coef once, dbev_bn_act_apply per call
extern "C" int dbev_bn_infer_coef(const float* gamma, const float* beta, const float* running_mean, const float* running_var, float eps,
                                  int C, float* scale_shift, dbevStream_t stream) {
  if (C <= 0 || gamma == nullptr || beta == nullptr || running_mean == nullptr || running_var == nullptr || scale_shift == nullptr)
    return DBEV_EINVAL;
  hipLaunchKernelGGL(bn_infer_coef, dim3(dbev_ceil_div(C, 256)), dim3(256), 0, dbev_stream(stream), gamma, beta, running_mean,
                     running_var, eps, C, scale_shift);
  DBEV_LAUNCH_CHECK();
  return 0;
}

extern "C" int dbev_bn_act_apply(const float* x, const float* residual, const float* scale_shift, int relu, float* y, long long M,
                                 int C, dbevStream_t stream) {
  BnGeom g;
  if (!bn_geom(M, C, &g)) return DBEV_EINVAL;
  if (x == nullptr || scale_shift == nullptr || y == nullptr) return DBEV_EINVAL;
  hipStream_t s = dbev_stream(stream);
  const float4* x4 = reinterpret_cast<const float4*>(x);
  const float4* r4 = reinterpret_cast<const float4*>(residual);
  float4* y4 = reinterpret_cast<float4*>(y);
  const long long tiles = (M + static_cast<long long>(g.RP) * BN_ROWS_UNROLL - 1) / (static_cast<long long>(g.RP) * BN_ROWS_UNROLL);
  const long long cap = static_cast<long long>(DBEV_MAX_GRID) * 2 / g.GY;
  const dim3 agrid(static_cast<unsigned>(tiles < cap ? tiles : (cap < 1 ? 1 : cap)), g.GY);
  const float* none = nullptr;
  const BnFin nofin{};
  if (residual != nullptr) {
    if (relu) hipLaunchKernelGGL((bn_apply<true, true>), agrid, dim3(256), 0, s, x4, r4, scale_shift, y4, g, none, 0, nofin, nofin);
    else hipLaunchKernelGGL((bn_apply<true, false>), agrid, dim3(256), 0, s, x4, r4, scale_shift, y4, g, none, 0, nofin, nofin);
  } else {
    if (relu) hipLaunchKernelGGL((bn_apply<false, true>), agrid, dim3(256), 0, s, x4, r4, scale_shift, y4, g, none, 0, nofin, nofin);
    else hipLaunchKernelGGL((bn_apply<false, false>), agrid, dim3(256), 0, s, x4, r4, scale_shift, y4, g, none, 0, nofin, nofin);
  }
  DBEV_LAUNCH_CHECK();
  return 0;
}

extern "C" int dbev_bn_act_backward(const float* grad_y, const float* x, const float* y, const float* gamma,
                                    const float* save_mean, const float* save_invstd, const float* save_scale_shift,
                                    int relu, float* grad_x, float* grad_residual, float* grad_gamma, float* grad_beta,
                                    long long M, int C, void* workspace, size_t workspace_bytes, dbevStream_t stream) {
  return dbev_bn_act_backward2(grad_y, nullptr, x, y, gamma, save_mean, save_invstd, save_scale_shift, relu, grad_x, grad_residual,
                               grad_gamma, grad_beta, M, C, workspace, workspace_bytes, stream);
}

extern "C" int dbev_bn_act_backward2(const float* grad_y, const float* grad_y2, const float* x, const float* y, const float* gamma,
                                     const float* save_mean, const float* save_invstd, const float* save_scale_shift,
                                     int relu, float* grad_x, float* grad_residual, float* grad_gamma, float* grad_beta,
                                     long long M, int C, void* workspace, size_t workspace_bytes, dbevStream_t stream) {
  return dbev_bn_act_backward3(grad_y, grad_y2, x, y, 0, gamma, save_mean, save_invstd, save_scale_shift, relu, grad_x, grad_residual,
                               grad_gamma, grad_beta, M, C, workspace, workspace_bytes, stream);
}

// y_is_mask != 0: `y` is the byte mask dbev_bn_act_train_forward_mask wrote (u8[M * C / 4]), not the saved output
extern "C" int dbev_bn_act_backward3(const float* grad_y, const float* grad_y2, const float* x, const void* y, int y_is_mask,
                                     const float* gamma, const float* save_mean, const float* save_invstd,
                                     const float* save_scale_shift, int relu, float* grad_x, float* grad_residual, float* grad_gamma,
                                     float* grad_beta, long long M, int C, void* workspace, size_t workspace_bytes,
                                     dbevStream_t stream) {
  BnGeom g;
  if (!bn_geom(M, C, &g)) return DBEV_EINVAL;
  const size_t need = bn_ws(g).total + sizeof(float) * 3 * static_cast<size_t>(C);
  if (grad_y == nullptr || x == nullptr || gamma == nullptr || save_mean == nullptr || save_invstd == nullptr ||
      save_scale_shift == nullptr || grad_x == nullptr || grad_gamma == nullptr || grad_beta == nullptr ||
      workspace == nullptr || workspace_bytes < need)
    return DBEV_EINVAL;
  // the ReLU gate of the residual variant needs the saved output; without residual it is recomputed from x
  const int mask = !relu ? 0 : (grad_residual != nullptr ? 2 : 1);
  if (mask == 2 && y == nullptr) return DBEV_EINVAL;
  // residual + ReLU: the reduce pass writes dz (= grad_residual), the dx pass reads it back (see bn_bwd_reduce); DBEV_BN_DZ_FIRST=0: the
  // round-4 order (dx pass recomputes dz from dy, dy2 and the gate and writes it)
  static const bool dz_first_on = getenv("DBEV_BN_DZ_FIRST") == nullptr || atoi(getenv("DBEV_BN_DZ_FIRST")) != 0;
  const bool dz_first = mask == 2 && dz_first_on;
  hipStream_t s = dbev_stream(stream);
  float* partial = static_cast<float*>(workspace);
  float* bcoef = reinterpret_cast<float*>(static_cast<char*>(workspace) + bn_ws(g).total);
  const dim3 grid(g.NBX, g.GY);
  const float4* dy4 = reinterpret_cast<const float4*>(grad_y);
  const float4* dy24 = reinterpret_cast<const float4*>(grad_y2);
  const float4* x4 = reinterpret_cast<const float4*>(x);
  const float4* y4 = reinterpret_cast<const float4*>(y);
  const long long T = 4LL * g.M * C;
  unsigned* tk = bn_ticket_ok(g) ? bn_tickets(1) : nullptr;
  {
    DbevKt kt(mask == 2 ? DBEV_K_BN_BWD_REDUCE_Y : DBEV_K_BN_BWD_REDUCE,
              T * (2 + (grad_y2 != nullptr) + dz_first) + (mask == 2 ? (y_is_mask ? T / 16 : T) : 0), s);
    if (mask == 0) launch_bwd_reduce<0>(g, grid, s, dy4, dy24, x4, y4, save_scale_shift, save_mean, save_invstd, partial, tk);
    else if (mask == 1) launch_bwd_reduce<1>(g, grid, s, dy4, dy24, x4, y4, save_scale_shift, save_mean, save_invstd, partial, tk);
    else launch_bwd_reduce<2>(g, grid, s, dy4, dy24, x4, y4, save_scale_shift, save_mean, save_invstd, partial, tk, y_is_mask,
                              dz_first ? reinterpret_cast<float4*>(grad_residual) : nullptr);
  }
  BnBfin fin{};
  if (tk != nullptr) {
    fin = BnBfin{partial, g.NBX, g.M, gamma, save_mean, save_invstd, grad_gamma, grad_beta};
  } else {
    DbevKt kt(DBEV_K_BN_BWD_FINALIZE, 8LL * g.NBX * g.GY * C, s);
    hipLaunchKernelGGL(bn_bwd_finalize, dim3(dbev_ceil_div(C, BN_FIN_CH)), dim3(256), 0, s, partial, g.NBX, g.M, C, gamma,
                       save_mean, save_invstd, grad_gamma, grad_beta, bcoef);
  }
  const long long tiles = (M + static_cast<long long>(g.RP) * BN_ROWS_UNROLL - 1) / (static_cast<long long>(g.RP) * BN_ROWS_UNROLL);
  const long long cap = static_cast<long long>(DBEV_MAX_GRID) * 2 / g.GY;
  const dim3 agrid(static_cast<unsigned>(tiles < cap ? tiles : (cap < 1 ? 1 : cap)), g.GY);
  float4* dx4 = reinterpret_cast<float4*>(grad_x);
  float4* dr4 = reinterpret_cast<float4*>(grad_residual);
  DbevKt kt(grad_residual != nullptr ? DBEV_K_BN_BWD_DX_RES : DBEV_K_BN_BWD_DX,
            dz_first ? T * 3 : T * (3 + (grad_y2 != nullptr) + (grad_residual != nullptr && mask != 0 ? 1 : 0)) + (mask == 2 ? (y_is_mask ? T / 16 : T) : 0), s);
  if (dz_first) {                        // dz is in grad_residual: dx = A dz + B x + C, nothing else to read or write
    hipLaunchKernelGGL((bn_bwd_dx<0, false>), agrid, dim3(256), 0, s, reinterpret_cast<const float4*>(grad_residual),
                       static_cast<const float4*>(nullptr), x4, y4, save_scale_shift, bcoef, dx4, static_cast<float4*>(nullptr), g, fin);
  } else if (grad_residual != nullptr) {
    if (mask == 2) hipLaunchKernelGGL((bn_bwd_dx<2, true>), agrid, dim3(256), 0, s, dy4, dy24, x4, y4, save_scale_shift, bcoef, dx4, dr4, g, fin, y_is_mask);
    else hipLaunchKernelGGL((bn_bwd_dx<0, true>), agrid, dim3(256), 0, s, dy4, dy24, x4, y4, save_scale_shift, bcoef, dx4, dr4, g, fin);
  } else {
    if (mask == 1) hipLaunchKernelGGL((bn_bwd_dx<1, false>), agrid, dim3(256), 0, s, dy4, dy24, x4, y4, save_scale_shift, bcoef, dx4, dr4, g, fin);
    else hipLaunchKernelGGL((bn_bwd_dx<0, false>), agrid, dim3(256), 0, s, dy4, dy24, x4, y4, save_scale_shift, bcoef, dx4, dr4, g, fin);
  }
  DBEV_LAUNCH_CHECK();
  return 0;
}

namespace {

// ---- the stem's pooling gather + norm backward in two passes (round 5) --------------------------------------------------------------
// y = maxpool3x3s2(relu(bn(x))) (mmdet ResNet.forward: conv1 -> norm1 -> relu -> maxpool; forward: dbev_norm_relu_maxpool3x3s2_forward).
// The gradient of the rectified map is a GATHER from the pooled gradient (an input pixel is the winner of at most 2 x 2 windows:
// csrc/maxpool.hip) -- instead of writing it (554 MB) and reading it twice, both passes of the norm's backward gather it themselves:
//   sp_reduce: sum dz, sum dz xhat per channel   reads pooled gradient + winners + x
//   sp_dx:     dx = A dz + B x + Cc              reads the same, writes dx
// with dz = [x scale + shift > 0] * gathered gradient.  A lane = one float4 of channels of a 2 x 2 block of input pixels, as in
// maxpool3x3s2_bwd; SP_IT blocks per lane in a fixed order, one partial row per workgroup (merged by bn_bwd_finalize in fp64).
struct SpDims { int N, H, W, C4, Ho, Wo, Hb, Wb; };
constexpr int SP_IT = 16;

__device__ __forceinline__ void sp_gather(const float4* __restrict__ gy, const uchar4* __restrict__ tap, const SpDims& d, int n, int i, int j,
                                          int c, float4 (&g)[2][2]) {
  float4 v[2][2];
  uchar4 s[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int oh = i + a, ow = j + b;
      if (oh < d.Ho && ow < d.Wo) {
        const size_t o = ((static_cast<size_t>(n) * d.Ho + oh) * d.Wo + ow) * d.C4 + c;
        v[a][b] = gy[o];
        s[a][b] = tap[o];
      } else {
        v[a][b] = f4(0.f);
        s[a][b] = make_uchar4(255, 255, 255, 255);
      }
    }
#pragma unroll
  for (int dh = 0; dh < 2; ++dh)
#pragma unroll
    for (int dw = 0; dw < 2; ++dw) {
      float4 acc = f4(0.f);
#pragma unroll
      for (int a = 0; a <= dh; ++a)
#pragma unroll
        for (int b = 0; b <= dw; ++b) {
          const int k = 3 * (dh + 1 - 2 * a) + (dw + 1 - 2 * b);       // tap of pixel (2 i + dh, 2 j + dw) in window (i + a, j + b)
          if (s[a][b].x == k) acc.x += v[a][b].x;
          if (s[a][b].y == k) acc.y += v[a][b].y;
          if (s[a][b].z == k) acc.z += v[a][b].z;
          if (s[a][b].w == k) acc.w += v[a][b].w;
        }
      g[dh][dw] = acc;
    }
}

template <bool DX>
__global__ __launch_bounds__(256) void sp_pass(const float4* __restrict__ gy, const uchar4* __restrict__ tap, const float4* __restrict__ x,
                                               const float* __restrict__ coef, const float* __restrict__ save_mean,
                                               const float* __restrict__ save_invstd, const float* __restrict__ bcoef,
                                               float* __restrict__ partial, float4* __restrict__ dx, SpDims d, long long total) {
  __shared__ float4 red[2][256];
  const int C = 4 * d.C4;
  const int c = threadIdx.x % d.C4;                                   // 256 % C4 == 0: a lane keeps its channels over its blocks
  const float4 sc = reinterpret_cast<const float4*>(coef)[c], sh = reinterpret_cast<const float4*>(coef + C)[c];
  float4 mu = f4(0.f), is = f4(0.f), A = f4(0.f), Bc = f4(0.f), Cc = f4(0.f);
  if (DX) {
    A = reinterpret_cast<const float4*>(bcoef)[c];
    Bc = reinterpret_cast<const float4*>(bcoef + C)[c];
    Cc = reinterpret_cast<const float4*>(bcoef + 2 * C)[c];
  } else {
    mu = reinterpret_cast<const float4*>(save_mean)[c];
    is = reinterpret_cast<const float4*>(save_invstd)[c];
  }
  float4 s1 = f4(0.f), s2 = f4(0.f);
  for (int it = 0; it < SP_IT; ++it) {
    const long long t = (static_cast<long long>(blockIdx.x) * SP_IT + it) * 256 + threadIdx.x;
    if (t >= total) break;
    long long p = t / d.C4;
    const int j = static_cast<int>(p % d.Wb); p /= d.Wb;
    const int i = static_cast<int>(p % d.Hb);
    const int n = static_cast<int>(p / d.Hb);
    float4 g[2][2];
    sp_gather(gy, tap, d, n, i, j, c, g);
#pragma unroll
    for (int dh = 0; dh < 2; ++dh) {
      const int h = 2 * i + dh;
      if (h >= d.H) continue;
#pragma unroll
      for (int dw = 0; dw < 2; ++dw) {
        const int w = 2 * j + dw;
        if (w >= d.W) continue;
        const size_t o = ((static_cast<size_t>(n) * d.H + h) * d.W + w) * d.C4 + c;
        const float4 v = x[o];
        const float4 dz = gate<1>(g[dh][dw], v, v, sc, sh);
        if (DX) {
          float4 r;
          r.x = fmaf(A.x, dz.x, fmaf(Bc.x, v.x, Cc.x)); r.y = fmaf(A.y, dz.y, fmaf(Bc.y, v.y, Cc.y));
          r.z = fmaf(A.z, dz.z, fmaf(Bc.z, v.z, Cc.z)); r.w = fmaf(A.w, dz.w, fmaf(Bc.w, v.w, Cc.w));
          st_nt(dx + o, r);
        } else {
          float4 xh;
          xh.x = (v.x - mu.x) * is.x; xh.y = (v.y - mu.y) * is.y; xh.z = (v.z - mu.z) * is.z; xh.w = (v.w - mu.w) * is.w;
          add4(s1, dz);
          fma4v(s2, dz, xh);
        }
      }
    }
  }
  if (!DX) {                                                           // the 256 / C4 lanes of a channel quad, in lane order
    red[0][threadIdx.x] = s1;
    red[1][threadIdx.x] = s2;
    __syncthreads();
    if (threadIdx.x < 2 * d.C4) {
      const int which = threadIdx.x / d.C4, q = threadIdx.x % d.C4;
      float4 a = red[which][q];
      for (int l = q + d.C4; l < 256; l += d.C4) add4(a, red[which][l]);
      reinterpret_cast<float4*>(partial + (static_cast<size_t>(blockIdx.x) * 2 + which) * C)[q] = a;
    }
  }
}

}  // namespace

// ---- dual BatchNorm: y = [relu](bn(x) + bn_d(xd)) ------------------------------------------------------------------------
extern "C" size_t dbev_bn_dual_workspace_bytes(long long M, int C) {
  BnGeom g;
  if (!bn_geom(M, C, &g)) return 0;
  // forward: the partial tables of BOTH inputs are live until the apply pass merges them (2 + 2 values per row);
  // backward: three values per row, then the six coefficient rows of the finalize-kernel path
  return sizeof(float) * static_cast<size_t>(g.NBX) * 4 * C + sizeof(float) * 6 * static_cast<size_t>(C);
}

extern "C" int dbev_bn_dual_train_forward(const float* x, const float* xd, const float* gamma, const float* beta,
                                          float* running_mean, float* running_var, long long* num_batches_tracked,
                                          float momentum, float eps, const float* gamma_d, const float* beta_d,
                                          float* running_mean_d, float* running_var_d, long long* num_batches_tracked_d,
                                          float momentum_d, float eps_d, int relu, float* y, float* save_mean,
                                          float* save_invstd, float* save_scale_shift, float* save_mean_d,
                                          float* save_invstd_d, float* save_scale_shift_d, long long M, int C,
                                          void* workspace, size_t workspace_bytes, dbevStream_t stream) {
  return dbev_bn_dual_train_forward_pre(x, xd, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps, gamma_d,
                                        beta_d, running_mean_d, running_var_d, num_batches_tracked_d, momentum_d, eps_d, relu, y,
                                        save_mean, save_invstd, save_scale_shift, save_mean_d, save_invstd_d, save_scale_shift_d, M,
                                        C, nullptr, 0, nullptr, 0, workspace, workspace_bytes, stream);
}

// ... with the partial sums of either input already taken by its producer (see dbev_bn_act_train_forward_pre)
extern "C" int dbev_bn_dual_train_forward_pre(const float* x, const float* xd, const float* gamma, const float* beta,
                                              float* running_mean, float* running_var, long long* num_batches_tracked,
                                              float momentum, float eps, const float* gamma_d, const float* beta_d,
                                              float* running_mean_d, float* running_var_d, long long* num_batches_tracked_d,
                                              float momentum_d, float eps_d, int relu, float* y, float* save_mean,
                                              float* save_invstd, float* save_scale_shift, float* save_mean_d,
                                              float* save_invstd_d, float* save_scale_shift_d, long long M, int C,
                                              const float* stats_partial, int partial_rows, const float* stats_partial_d,
                                              int partial_rows_d, void* workspace, size_t workspace_bytes, dbevStream_t stream) {
  return dbev_bn_dual_train_forward_mask(x, xd, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps, gamma_d, beta_d,
                                         running_mean_d, running_var_d, num_batches_tracked_d, momentum_d, eps_d, relu, y, save_mean,
                                         save_invstd, save_scale_shift, save_mean_d, save_invstd_d, save_scale_shift_d, M, C, stats_partial,
                                         partial_rows, stats_partial_d, partial_rows_d, nullptr, workspace, workspace_bytes, stream);
}

// ... and with the ReLU gate written as one byte per four channels (relu_mask u8[M * C / 4] or NULL; see dbev_bn_act_train_forward_mask)
extern "C" int dbev_bn_dual_train_forward_mask(const float* x, const float* xd, const float* gamma, const float* beta,
                                               float* running_mean, float* running_var, long long* num_batches_tracked,
                                               float momentum, float eps, const float* gamma_d, const float* beta_d,
                                               float* running_mean_d, float* running_var_d, long long* num_batches_tracked_d,
                                               float momentum_d, float eps_d, int relu, float* y, float* save_mean,
                                               float* save_invstd, float* save_scale_shift, float* save_mean_d,
                                               float* save_invstd_d, float* save_scale_shift_d, long long M, int C,
                                               const float* stats_partial, int partial_rows, const float* stats_partial_d,
                                               int partial_rows_d, unsigned char* relu_mask, void* workspace, size_t workspace_bytes,
                                               dbevStream_t stream) {
  BnGeom g;
  if (!bn_geom(M, C, &g)) return DBEV_EINVAL;
  if (x == nullptr || xd == nullptr || gamma == nullptr || beta == nullptr || gamma_d == nullptr || beta_d == nullptr ||
      y == nullptr || save_mean == nullptr || save_invstd == nullptr || save_scale_shift == nullptr ||
      save_mean_d == nullptr || save_invstd_d == nullptr || save_scale_shift_d == nullptr || workspace == nullptr ||
      workspace_bytes < dbev_bn_dual_workspace_bytes(M, C) || (running_mean == nullptr) != (running_var == nullptr) ||
      (running_mean_d == nullptr) != (running_var_d == nullptr))
    return DBEV_EINVAL;
  hipStream_t s = dbev_stream(stream);
  float* partial = static_cast<float*>(workspace);
  const dim3 grid(g.NBX, g.GY);
  const long long T = 4LL * g.M * C;
  const float4* x4 = reinterpret_cast<const float4*>(x);
  const float4* d4 = reinterpret_cast<const float4*>(xd);
  if ((stats_partial != nullptr && partial_rows <= 0) || (stats_partial_d != nullptr && partial_rows_d <= 0)) return DBEV_EINVAL;
  // statistics of the branch, then of the main input, each into its own partial table
  float* partial_d = partial + static_cast<size_t>(g.NBX) * 2 * C;
  const bool tko = bn_ticket_ok(g);
  unsigned* tk = tko ? bn_tickets(2) : nullptr;
  unsigned* tk_d = (tk != nullptr && stats_partial_d == nullptr) ? tk + 2 * BN_MAX_GROUPS : nullptr;
  if (stats_partial != nullptr) tk = nullptr;
  BnFin fin{}, fin_d{};
  if (stats_partial_d == nullptr) { DbevKt kt(DBEV_K_BN_STATS, T, s); launch_stats(g, grid, s, d4, partial_d, tk_d); }
  if (tk_d != nullptr) {
    fin_d = BnFin{partial_d, g.NBX, g.M, gamma_d, beta_d, running_mean_d, running_var_d, momentum_d, eps_d, save_mean_d, save_invstd_d,
                  save_scale_shift_d, num_batches_tracked_d};
  } else {
    const int nr = stats_partial_d != nullptr ? partial_rows_d : g.NBX;
    launch_finalize(s, stats_partial_d != nullptr ? stats_partial_d : partial_d, nr, g.M, C, gamma_d, beta_d, running_mean_d, running_var_d,
                    momentum_d, eps_d, save_mean_d, save_invstd_d, save_scale_shift_d, num_batches_tracked_d, partial_d,
                    sizeof(float) * static_cast<size_t>(g.NBX) * 2 * C);
  }
  if (stats_partial == nullptr) { DbevKt kt(DBEV_K_BN_STATS, T, s); launch_stats(g, grid, s, x4, partial, tk); }
  if (tk != nullptr) {
    fin = BnFin{partial, g.NBX, g.M, gamma, beta, running_mean, running_var, momentum, eps, save_mean, save_invstd, save_scale_shift,
                num_batches_tracked};
  } else {
    const int nr = stats_partial != nullptr ? partial_rows : g.NBX;
    launch_finalize(s, stats_partial != nullptr ? stats_partial : partial, nr, g.M, C, gamma, beta, running_mean, running_var, momentum, eps,
                    save_mean, save_invstd, save_scale_shift, num_batches_tracked, partial, sizeof(float) * static_cast<size_t>(g.NBX) * 2 * C);
  }
  const long long tiles = (M + static_cast<long long>(g.RP) * BN_ROWS_UNROLL - 1) / (static_cast<long long>(g.RP) * BN_ROWS_UNROLL);
  const long long cap = static_cast<long long>(DBEV_MAX_GRID) * 2 / g.GY;
  const dim3 agrid(static_cast<unsigned>(tiles < cap ? tiles : (cap < 1 ? 1 : cap)), g.GY);
  DbevKt kt(DBEV_K_BN_APPLY_RES, T * 3, s);
  float4* y4 = reinterpret_cast<float4*>(y);
  const int rev = (stats_partial != nullptr && stats_partial_d != nullptr) ? 1 : 0;
  if (relu) hipLaunchKernelGGL((bn_apply<true, true, true>), agrid, dim3(256), 0, s, x4, d4, save_scale_shift, y4, g, save_scale_shift_d, rev, fin, fin_d, relu_mask);
  else hipLaunchKernelGGL((bn_apply<true, false, true>), agrid, dim3(256), 0, s, x4, d4, save_scale_shift, y4, g, save_scale_shift_d, rev, fin, fin_d);
  DBEV_LAUNCH_CHECK();
  return 0;
}

extern "C" int dbev_bn_dual_backward(const float* grad_y, const float* x, const float* xd, const float* y,
                                     const float* gamma, const float* save_mean, const float* save_invstd,
                                     const float* gamma_d, const float* save_mean_d, const float* save_invstd_d, int relu,
                                     float* grad_x, float* grad_xd, float* grad_gamma, float* grad_beta,
                                     float* grad_gamma_d, float* grad_beta_d, long long M, int C, void* workspace,
                                     size_t workspace_bytes, dbevStream_t stream) {
  return dbev_bn_dual_backward2(grad_y, nullptr, x, xd, y, gamma, save_mean, save_invstd, gamma_d, save_mean_d, save_invstd_d, relu,
                                grad_x, grad_xd, grad_gamma, grad_beta, grad_gamma_d, grad_beta_d, M, C, workspace, workspace_bytes,
                                stream);
}

extern "C" int dbev_bn_dual_backward2(const float* grad_y, const float* grad_y2, const float* x, const float* xd, const float* y,
                                      const float* gamma, const float* save_mean, const float* save_invstd,
                                      const float* gamma_d, const float* save_mean_d, const float* save_invstd_d, int relu,
                                      float* grad_x, float* grad_xd, float* grad_gamma, float* grad_beta,
                                      float* grad_gamma_d, float* grad_beta_d, long long M, int C, void* workspace,
                                      size_t workspace_bytes, dbevStream_t stream) {
  return dbev_bn_dual_backward3(grad_y, grad_y2, x, xd, y, 0, gamma, save_mean, save_invstd, gamma_d, save_mean_d, save_invstd_d, relu,
                                grad_x, grad_xd, grad_gamma, grad_beta, grad_gamma_d, grad_beta_d, M, C, workspace, workspace_bytes, stream);
}

// y_is_mask != 0: `y` is the byte mask dbev_bn_dual_train_forward_mask wrote
extern "C" int dbev_bn_dual_backward3(const float* grad_y, const float* grad_y2, const float* x, const float* xd, const void* y,
                                      int y_is_mask, const float* gamma, const float* save_mean, const float* save_invstd,
                                      const float* gamma_d, const float* save_mean_d, const float* save_invstd_d, int relu,
                                      float* grad_x, float* grad_xd, float* grad_gamma, float* grad_beta,
                                      float* grad_gamma_d, float* grad_beta_d, long long M, int C, void* workspace,
                                      size_t workspace_bytes, dbevStream_t stream) {
  BnGeom g;
  if (!bn_geom(M, C, &g)) return DBEV_EINVAL;
  if (grad_y == nullptr || x == nullptr || xd == nullptr || (relu && y == nullptr) || gamma == nullptr ||
      save_mean == nullptr || save_invstd == nullptr || gamma_d == nullptr || save_mean_d == nullptr ||
      save_invstd_d == nullptr || grad_x == nullptr || grad_xd == nullptr || grad_gamma == nullptr ||
      grad_beta == nullptr || grad_gamma_d == nullptr || grad_beta_d == nullptr || workspace == nullptr ||
      workspace_bytes < dbev_bn_dual_workspace_bytes(M, C))
    return DBEV_EINVAL;
  hipStream_t s = dbev_stream(stream);
  float* partial = static_cast<float*>(workspace);
  float* bcoef = partial + static_cast<size_t>(g.NBX) * 3 * C;
  unsigned* tk = bn_ticket_ok(g) ? bn_tickets(1) : nullptr;
  const dim3 grid(g.NBX, g.GY);
  const float4* dy4 = reinterpret_cast<const float4*>(grad_y);
  const float4* dy24 = reinterpret_cast<const float4*>(grad_y2);
  const float4* x4 = reinterpret_cast<const float4*>(x);
  const float4* d4 = reinterpret_cast<const float4*>(xd);
  const float4* y4 = reinterpret_cast<const float4*>(y);
  const long long T = 4LL * g.M * C;
  {
    DbevKt kt(DBEV_K_BN_BWD_REDUCE_Y, T * 3 + (relu ? (y_is_mask ? T / 16 : T) : 0), s);
#define BN_CALL(TP, U)                                                                                                        \
  do {                                                                                                                        \
    if (relu) hipLaunchKernelGGL((bn_bwd_reduce_dual<true, TP, U>), grid, dim3(TP), 0, s, dy4, dy24, x4, d4, y4, save_mean,         \
                                 save_invstd, save_mean_d, save_invstd_d, partial, g, tk, y_is_mask);                         \
    else hipLaunchKernelGGL((bn_bwd_reduce_dual<false, TP, U>), grid, dim3(TP), 0, s, dy4, dy24, x4, d4, y4, save_mean, save_invstd, \
                            save_mean_d, save_invstd_d, partial, g, tk);                                                      \
  } while (0)
    BN_DISPATCH_TPB_UNR(BN_CALL);
#undef BN_CALL
  }
  BnBfin fin{}, fin_d{};
  if (tk != nullptr) {
    fin = BnBfin{partial, g.NBX, g.M, gamma, save_mean, save_invstd, grad_gamma, grad_beta};
    fin_d = BnBfin{partial, g.NBX, g.M, gamma_d, save_mean_d, save_invstd_d, grad_gamma_d, grad_beta_d};
  } else {
    DbevKt kt(DBEV_K_BN_BWD_FINALIZE, 12LL * g.NBX * g.GY * C, s);
    hipLaunchKernelGGL(bn_bwd_finalize_dual, dim3(dbev_ceil_div(C, BN_FIN_CH)), dim3(256), 0, s, partial, g.NBX, g.M, C, gamma,
                       save_mean, save_invstd, gamma_d, save_mean_d, save_invstd_d, grad_gamma, grad_beta, grad_gamma_d,
                       grad_beta_d, bcoef);
  }
  const long long tiles = (M + static_cast<long long>(g.RP) * BN_ROWS_UNROLL - 1) / (static_cast<long long>(g.RP) * BN_ROWS_UNROLL);
  const long long cap = static_cast<long long>(DBEV_MAX_GRID) * 2 / g.GY;
  const dim3 agrid(static_cast<unsigned>(tiles < cap ? tiles : (cap < 1 ? 1 : cap)), g.GY);
  DbevKt kt(DBEV_K_BN_BWD_DX_RES, T * 5 + (relu ? (y_is_mask ? T / 16 : T) : 0), s);
  if (relu) hipLaunchKernelGGL((bn_bwd_dx_dual<true>), agrid, dim3(256), 0, s, dy4, dy24, x4, d4, y4, bcoef, reinterpret_cast<float4*>(grad_x),
                               reinterpret_cast<float4*>(grad_xd), g, fin, fin_d, y_is_mask);
  else hipLaunchKernelGGL((bn_bwd_dx_dual<false>), agrid, dim3(256), 0, s, dy4, dy24, x4, d4, y4, bcoef, reinterpret_cast<float4*>(grad_x),
                          reinterpret_cast<float4*>(grad_xd), g, fin, fin_d);
  DBEV_LAUNCH_CHECK();
  return 0;
}

extern "C" size_t dbev_stem_pool_norm_backward_workspace_bytes(int N, int H, int W, int C) {
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3) || 256 % (C / 4) != 0) return 0;
  const long long total = static_cast<long long>(N) * ((H + 1) / 2) * ((W + 1) / 2) * (C / 4);
  const long long rows = (total + 256LL * SP_IT - 1) / (256LL * SP_IT);
  if (rows >= (1LL << 24)) return 0;
  return sizeof(float) * (static_cast<size_t>(rows) * 2 * C + 3 * static_cast<size_t>(C));
}

extern "C" int dbev_stem_pool_norm_backward(const float* grad_pooled, const unsigned char* winner, const float* x, const float* gamma,
                                            const float* save_mean, const float* save_invstd, const float* save_scale_shift, int N, int H,
                                            int W, int C, float* grad_x, float* grad_gamma, float* grad_beta, void* workspace,
                                            size_t workspace_bytes, dbevStream_t stream) {
  const size_t need = dbev_stem_pool_norm_backward_workspace_bytes(N, H, W, C);
  if (need == 0 || grad_pooled == nullptr || winner == nullptr || x == nullptr || gamma == nullptr || save_mean == nullptr ||
      save_invstd == nullptr || save_scale_shift == nullptr || grad_x == nullptr || grad_gamma == nullptr || grad_beta == nullptr ||
      workspace == nullptr || workspace_bytes < need || static_cast<long long>(N) * H * W >= (1LL << 31))
    return DBEV_EINVAL;
  hipStream_t s = dbev_stream(stream);
  SpDims d{N, H, W, C / 4, (H - 1) / 2 + 1, (W - 1) / 2 + 1, (H + 1) / 2, (W + 1) / 2};
  const long long total = static_cast<long long>(N) * d.Hb * d.Wb * d.C4;
  const int rows = static_cast<int>((total + 256LL * SP_IT - 1) / (256LL * SP_IT));
  float* partial = static_cast<float*>(workspace);
  float* bcoef = partial + static_cast<size_t>(rows) * 2 * C;
  const long long M = static_cast<long long>(N) * H * W;
  const long long T = 4LL * M * C;
  const float4* g4 = reinterpret_cast<const float4*>(grad_pooled);
  const uchar4* t4 = reinterpret_cast<const uchar4*>(winner);
  const float4* x4 = reinterpret_cast<const float4*>(x);
  {
    DbevKt kt(DBEV_K_BN_BWD_REDUCE, T + T / 4 + T / 16, s);
    hipLaunchKernelGGL((sp_pass<false>), dim3(rows), dim3(256), 0, s, g4, t4, x4, save_scale_shift, save_mean, save_invstd,
                       static_cast<const float*>(nullptr), partial, static_cast<float4*>(nullptr), d, total);
  }
  {
    DbevKt kt(DBEV_K_BN_BWD_FINALIZE, 8LL * rows * C, s);
    hipLaunchKernelGGL(bn_bwd_finalize, dim3(dbev_ceil_div(C, BN_FIN_CH)), dim3(256), 0, s, partial, rows, static_cast<int>(M), C, gamma,
                       save_mean, save_invstd, grad_gamma, grad_beta, bcoef);
  }
  {
    DbevKt kt(DBEV_K_BN_BWD_DX, 2 * T + T / 4 + T / 16, s);
    hipLaunchKernelGGL((sp_pass<true>), dim3(rows), dim3(256), 0, s, g4, t4, x4, save_scale_shift, save_mean, save_invstd, bcoef,
                       static_cast<float*>(nullptr), reinterpret_cast<float4*>(grad_x), d, total);
  }
  DBEV_LAUNCH_CHECK();
  return 0;
}
