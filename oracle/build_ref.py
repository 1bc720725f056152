"""ORACLE — TEST INFRASTRUCTURE ONLY.

Compiles the reference's OWN hard-voxelization CPU sources where they lie under
/root/reference into oracle/_ref/ (git-ignored, but shipped to the GPU box with
the snapshot).  No reference source is copied into the repo and nothing is
patched or stubbed: the three files only need torch's headers, which the image
has.

    mmdet3d/ops/voxel/src/voxelization.cpp        (pybind module)
    mmdet3d/ops/voxel/src/voxelization_cpu.cpp    (hard_voxelize_cpu :105)
    mmdet3d/ops/voxel/src/scatter_points_cpu.cpp  (symbol needed by the module)

The reference's spconv and iou3d_nms CPU sources are NOT buildable under the
no-stand-in rule (every file includes <cuda_runtime_api.h> / <cuda.h>, which this
image lacks) — see DESIGN.md §Oracle.
"""
import importlib.util
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
REF_ROOT = '/root/reference'
OUT_DIR = os.path.join(_HERE, '_ref')
NAME = 'voxel_layer_ref'


def ref_sources():
    src = os.path.join(REF_ROOT, 'mmdet3d/ops/voxel/src')
    return [os.path.join(src, f) for f in
            ('voxelization.cpp', 'voxelization_cpu.cpp', 'scatter_points_cpu.cpp')]


def so_path():
    if not os.path.isdir(OUT_DIR):
        return None
    for f in os.listdir(OUT_DIR):
        if f.startswith(NAME) and f.endswith('.so'):
            return os.path.join(OUT_DIR, f)
    return None


def build(verbose=False):
    """Build oracle/_ref/voxel_layer_ref*.so when /root/reference is present."""
    if so_path() is not None:
        return so_path()
    if not os.path.isdir(REF_ROOT):
        return None
    from torch.utils.cpp_extension import load
    os.makedirs(OUT_DIR, exist_ok=True)
    load(name=NAME, sources=ref_sources(), build_directory=OUT_DIR,
         extra_cflags=['-O2', '-w'], verbose=verbose, is_python_module=True)
    return so_path()


def load_ref():
    """Import the compiled reference module, or None when it is not available."""
    path = so_path()
    if path is None:
        return None
    import torch  # noqa: F401  (libtorch must be loaded first)
    spec = importlib.util.spec_from_file_location(NAME, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def ref_hard_voxelize(points, voxel_size, coors_range, max_points, max_voxels):
    """Reference hard_voxelize on a numpy (N,C) f32 array, through the same call
    sequence as mmdet3d/ops/voxel/voxelize.py:46-58."""
    import numpy as np
    import torch
    mod = load_ref()
    if mod is None:
        raise RuntimeError('oracle/_ref is not built')
    pts = torch.from_numpy(np.ascontiguousarray(points, dtype=np.float32))
    voxels = pts.new_zeros(size=(max_voxels, max_points, pts.size(1)))
    coors = pts.new_zeros(size=(max_voxels, 3), dtype=torch.int)
    num = pts.new_zeros(size=(max_voxels,), dtype=torch.int)
    v = mod.hard_voxelize(pts, voxels, coors, num, [float(x) for x in voxel_size],
                          [float(x) for x in coors_range], int(max_points),
                          int(max_voxels), 3)
    return voxels[:v].numpy(), coors[:v].numpy(), num[:v].numpy()


if __name__ == '__main__':
    p = build(verbose='-v' in sys.argv)
    print('oracle/_ref:', p)
