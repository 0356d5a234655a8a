"""ctypes wrapper of oracle/voxel.c (plain-C CPU restatement of the reference's voxel
ops).  TEST INFRASTRUCTURE ONLY.  numpy in / numpy out."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libdbev_oracle.so")
_lib = None

REDUCE = {"sum": 0, "mean": 1, "max": 2}


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "voxel.c")):
            subprocess.check_call(["make", "-C", _HERE], stdout=subprocess.DEVNULL)
        _lib = ctypes.CDLL(_SO)
        _lib.oracle_hard_voxelize.restype = ctypes.c_int
        _lib.oracle_dynamic_scatter_forward.restype = ctypes.c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _f3(v):
    return np.ascontiguousarray(np.asarray(v, dtype=np.float32))


def dynamic_voxelize(points, voxel_size, coors_range, ndim=3):
    """voxelization_cpu.cpp:7-43 -> int32[N, 3] (z, y, x) or (-1,-1,-1)."""
    points = np.ascontiguousarray(points, dtype=np.float32)
    n, f = points.shape
    coors = np.zeros((n, ndim), dtype=np.int32)
    _load().oracle_dynamic_voxelize(_p(points), _p(coors), n, f, _p(_f3(voxel_size)),
                                    _p(_f3(coors_range)), ndim)
    return coors


def hard_voxelize(points, voxel_size, coors_range, max_points, max_voxels, ndim=3):
    """voxelization_cpu.cpp:45-101 -> (voxels[M,max_points,F], coors int32[M,3], num_points int32[M])."""
    points = np.ascontiguousarray(points, dtype=np.float32)
    n, f = points.shape
    voxels = np.zeros((max_voxels, max_points, f), dtype=np.float32)
    coors = np.zeros((max_voxels, ndim), dtype=np.int32)
    num = np.zeros((max_voxels,), dtype=np.int32)
    m = _load().oracle_hard_voxelize(_p(points), _p(voxels), _p(coors), _p(num), n, f,
                                     _p(_f3(voxel_size)), _p(_f3(coors_range)),
                                     int(max_points), int(max_voxels), ndim)
    return voxels[:m], coors[:m], num[:m]


def dynamic_scatter_forward(feats, coors, reduce_type="max"):
    """scatter_points_cuda.cu:183-239 -> (reduced[M,C], out_coors int32[M,3],
    coors_map int32[N], reduce_count int32[M])."""
    feats = np.ascontiguousarray(feats, dtype=np.float32)
    coors = np.ascontiguousarray(coors, dtype=np.int32)
    n, c = feats.shape
    reduced = np.zeros((max(n, 1), c), dtype=np.float32)
    out_coors = np.zeros((max(n, 1), 3), dtype=np.int32)
    cmap = np.zeros((n,), dtype=np.int32)
    cnt = np.zeros((max(n, 1),), dtype=np.int32)
    m = _load().oracle_dynamic_scatter_forward(_p(feats), _p(coors), n, c, REDUCE[reduce_type],
                                               _p(reduced), _p(out_coors), _p(cmap), _p(cnt))
    return reduced[:m].copy(), out_coors[:m].copy(), cmap, cnt[:m].copy()


def dynamic_scatter_backward(grad_reduced, feats, reduced, coors_map, reduce_count, reduce_type="max"):
    """scatter_points_cuda.cu:241-308 -> grad_feats[N, C]."""
    feats = np.ascontiguousarray(feats, dtype=np.float32)
    grad_reduced = np.ascontiguousarray(grad_reduced, dtype=np.float32)
    reduced = np.ascontiguousarray(reduced, dtype=np.float32)
    n, c = feats.shape
    g = np.zeros((n, c), dtype=np.float32)
    _load().oracle_dynamic_scatter_backward(_p(g), _p(grad_reduced), _p(feats), _p(reduced),
                                            _p(np.ascontiguousarray(coors_map, dtype=np.int32)),
                                            _p(np.ascontiguousarray(reduce_count, dtype=np.int32)),
                                            n, reduced.shape[0], c, REDUCE[reduce_type])
    return g


def pillars_scatter(voxel_features, coors, batch_size, ny, nx):
    """pillar_scatter.py:62-102 -> f32[B, C, ny, nx]; coors int32[M,4] = (b, z, y, x)."""
    vf = np.ascontiguousarray(voxel_features, dtype=np.float32)
    coors = np.ascontiguousarray(coors, dtype=np.int32)
    m, c = vf.shape
    canvas = np.zeros((batch_size, c, ny, nx), dtype=np.float32)
    _load().oracle_pillars_scatter(_p(vf), _p(coors), m, c, batch_size, ny, nx, _p(canvas))
    return canvas
