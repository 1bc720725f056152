"""ORACLE — TEST INFRASTRUCTURE ONLY.

Compiles the reference's OWN hard-voxelization CPU sources where they lie under
/root/reference into oracle/_ref/ (git-ignored, but shipped to the GPU box with
the snapshot).  No reference source is copied into the repo and nothing is
patched or stubbed: the three files only need torch's headers, which the image
has.

    mmdet3d/ops/voxel/src/voxelization.cpp        (pybind module)
    mmdet3d/ops/voxel/src/voxelization_cpu.cpp    (hard_voxelize_cpu :105)
    mmdet3d/ops/voxel/src/scatter_points_cpu.cpp  (symbol needed by the module)

Round 2: the sparse-conv and BEV-IoU CPU sources are built too.  They include
<cuda_runtime_api.h> / <cuda.h>; the image ships the REAL headers inside the triton wheel
(triton/backends/nvidia/include), which is put on the include path — nothing is patched and
no stand-in header is written:

    spconv_ref   oracle/ref_spconv_driver.cc (own driver) + the reference's
                 mmdet3d/ops/spconv/include/spconv/geometry.h (getIndicePairsConv :145,
                 getIndicePairsSubM :248) and mmdet3d/ops/spconv/src/reordering.cc (CPU
                 gather / scatter-add functors :21-50)
    iou3d_ref    oracle/ref_iou3d_driver.cc (own 10-line pybind stub) + the reference's
                 thirdparty/Spconv-OpenPCDet/pcdet/ops/iou3d_nms/src/iou3d_cpu.cpp
                 (boxes_iou_bev_cpu :232)
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


SPCONV_NAME = 'spconv_ref'
IOU3D_NAME = 'iou3d_ref'


def _cuda_include():
    """The real CUDA runtime headers the image ships (triton wheel); None if absent."""
    try:
        import triton
    except Exception:
        return None
    d = os.path.join(os.path.dirname(triton.__file__), 'backends', 'nvidia', 'include')
    return d if os.path.exists(os.path.join(d, 'cuda_runtime_api.h')) else None


def so_path(name=NAME):
    if not os.path.isdir(OUT_DIR):
        return None
    for f in os.listdir(OUT_DIR):
        if f.startswith(name) and f.endswith('.so'):
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


def build_spconv(verbose=False):
    """oracle/_ref/spconv_ref*.so: own driver + the reference's geometry.h / reordering.cc."""
    if so_path(SPCONV_NAME) is not None:
        return so_path(SPCONV_NAME)
    inc = _cuda_include()
    if not os.path.isdir(REF_ROOT) or inc is None:
        return None
    from torch.utils.cpp_extension import load
    sp = os.path.join(REF_ROOT, 'mmdet3d/ops/spconv')
    out = os.path.join(OUT_DIR, SPCONV_NAME + '_build')
    os.makedirs(out, exist_ok=True)
    load(name=SPCONV_NAME, sources=[os.path.join(_HERE, 'ref_spconv_driver.cc'),
                                    os.path.join(sp, 'src/reordering.cc')],
         extra_include_paths=[os.path.join(sp, 'include'), inc], build_directory=out,
         extra_cflags=['-O2', '-w', '-std=c++17'], verbose=verbose, is_python_module=True)
    _hoist(out, SPCONV_NAME)
    return so_path(SPCONV_NAME)


def build_iou3d(verbose=False):
    """oracle/_ref/iou3d_ref*.so: the reference's iou3d_cpu.cpp + an own pybind stub."""
    if so_path(IOU3D_NAME) is not None:
        return so_path(IOU3D_NAME)
    inc = _cuda_include()
    if not os.path.isdir(REF_ROOT) or inc is None:
        return None
    from torch.utils.cpp_extension import load
    src = os.path.join(REF_ROOT, 'thirdparty/Spconv-OpenPCDet/pcdet/ops/iou3d_nms/src')
    out = os.path.join(OUT_DIR, IOU3D_NAME + '_build')
    os.makedirs(out, exist_ok=True)
    load(name=IOU3D_NAME, sources=[os.path.join(_HERE, 'ref_iou3d_driver.cc'),
                                   os.path.join(src, 'iou3d_cpu.cpp')],
         extra_include_paths=[src, inc], build_directory=out,
         extra_cflags=['-O2', '-w', '-std=c++17'], verbose=verbose, is_python_module=True)
    _hoist(out, IOU3D_NAME)
    return so_path(IOU3D_NAME)


def _hoist(build_dir, name):
    """Keep only the .so (next to the voxel one); the ninja scratch dir is not needed."""
    import shutil
    for f in os.listdir(build_dir):
        if f.startswith(name) and f.endswith('.so'):
            shutil.copy2(os.path.join(build_dir, f), os.path.join(OUT_DIR, f))
    shutil.rmtree(build_dir, ignore_errors=True)


def build_all(verbose=False):
    """Every reference-compiled checker that can be built here; returns {name: path|None}."""
    out = {}
    for name, fn in ((NAME, build), (SPCONV_NAME, build_spconv), (IOU3D_NAME, build_iou3d)):
        try:
            out[name] = fn(verbose=verbose)
        except Exception as e:      # optional test infrastructure
            print('[oracle/_ref] %s not built: %s' % (name, e), file=sys.stderr)
            out[name] = None
    return out


def load_ref(name=NAME):
    """Import a compiled reference module, or None when it is not available."""
    path = so_path(name)
    if path is None:
        return None
    import torch  # noqa: F401  (libtorch must be loaded first)
    spec = importlib.util.spec_from_file_location(name, path)
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
    print('oracle/_ref:', build_all(verbose='-v' in sys.argv))
