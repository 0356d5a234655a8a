// Device-wide primitives shared by the kernels of libdbev_hip.so: an exclusive scan over
// int32 and a per-segment ascending sort of uint32 values.  Hand-written for gfx950
// (64-lane wave shuffles, LDS block carries); no library dependency.
//
// Together with integer atomics they give every "group points by cell" step of the hot
// path (LSS splat CSR, dynamic scatter, hard voxelization) a DETERMINISTIC order:
//   histogram (int atomicAdd)  ->  exclusive scan  ->  fill (atomic cursor, any order)
//   ->  segment sort by point id  ->  consume in point-id order.
#pragma once
#include "common.h"

namespace dbev {

// number of int32 workspace entries exclusive_scan needs for n items
size_t scan_workspace_ints(long long n);

// out[i] = sum_{j<i} f(in[j]) for i in [0, n], i.e. out has n+1 entries and out[n] is the
// total; f(x) = (x > 0) if as_flags else x.  If total_out != nullptr the total is also
// stored there.  `ws` must hold scan_workspace_ints(n) ints.  in/out may not alias.
int exclusive_scan_i32(const int* in, int* out, long long n, bool as_flags, int* total_out,
                       int* ws, hipStream_t s);

// For every segment g in [0, n_seg): dst[starts[g] .. starts[g+1]) = ascending sort of
// src[starts[g] .. starts[g+1]).  Values inside one segment must be distinct (point ids).
// `ws` (work list of the segments longer than a wave) must hold
// segment_sort_workspace_ints(total number of values) ints.
size_t segment_sort_workspace_ints(long long n_values);
int segment_sort_u32(const int* starts, const unsigned* src, unsigned* dst, int n_seg, int* ws,
                     hipStream_t s);

}  // namespace dbev
