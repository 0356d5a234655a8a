#!/usr/bin/env python3
"""Recipe that compiles the REFERENCE's own CPU voxelization from the sources where they
lie under /root/reference (nothing is copied into the repo) into oracle/_ref/.

Only the reference's few self-contained source files are compiled
(mmdet3d/ops/voxel/src/{voxelization.cpp, voxelization_cpu.cpp, scatter_points_cpu.cpp},
no WITH_CUDA), with torch.utils.cpp_extension -- the same mechanism the reference's
setup.py uses (setup.py:44-56,260-269), not the reference's build system.  The result
(voxel_layer_ref*.so) exports hard_voxelize / dynamic_voxelize; the two dynamic_scatter
entry points exist but raise "do not support cpu yet" (voxelization.h:118,139).

oracle/_ref/ is git-ignored (built artefact) but travels to the GPU box with gpurun.
The rest of the reference's native code (bev_pool_ext: CUDA only; spconv: needs
cuda_runtime_api.h) is unbuildable here -- see DESIGN.md.
"""
import glob
import os
import shutil
import sys

REF = os.environ.get("DBEV_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_ref")


def main():
    src_dir = os.path.join(REF, "mmdet3d", "ops", "voxel", "src")
    srcs = [os.path.join(src_dir, f) for f in
            ("voxelization.cpp", "voxelization_cpu.cpp", "scatter_points_cpu.cpp")]
    if not all(os.path.exists(s) for s in srcs):
        print("reference sources not present; skipping oracle/_ref")
        return 0
    os.makedirs(OUT, exist_ok=True)
    if glob.glob(os.path.join(OUT, "voxel_layer_ref*.so")):
        print("oracle/_ref already built")
        return 0
    from torch.utils.cpp_extension import load
    build_dir = os.path.join(OUT, "build")
    os.makedirs(build_dir, exist_ok=True)
    load(name="voxel_layer_ref", sources=srcs, extra_cflags=["-O2", "-w"],
         build_directory=build_dir, verbose=False)
    for so in glob.glob(os.path.join(build_dir, "voxel_layer_ref*.so")):
        shutil.copy(so, OUT)
    shutil.rmtree(build_dir, ignore_errors=True)
    print("built", glob.glob(os.path.join(OUT, "*.so")))
    return 0


def load_ref():
    """Import the built module (or return None if oracle/_ref is absent)."""
    import importlib.util
    sos = glob.glob(os.path.join(OUT, "voxel_layer_ref*.so"))
    if not sos:
        return None
    import torch  # noqa: F401  (the extension links against libtorch)
    spec = importlib.util.spec_from_file_location("voxel_layer_ref", sos[0])
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


if __name__ == "__main__":
    sys.exit(main())
