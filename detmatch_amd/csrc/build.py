"""Builds detmatch_amd/csrc/libdetmatch_hip.so for gfx950 with hipcc (in-tree).

hipcc cross-compiles without a GPU, so this runs in the CPU container and the
resulting .so travels to the GPU box with the repo snapshot.
"""
import concurrent.futures
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, 'libdetmatch_hip.so')
SOURCES = ['voxelize.hip', 'rulebook.hip', 'spconv.hip', 'iou3d_nms.hip',
           'pointnet2_stack.hip', 'points_in_boxes.hip', 'ssl_ops.hip', 'roi_align.hip', 'bn_relu.hip', 'augment.hip', 'kitti_eval.hip', 'anchor_loss.hip', 'misc.hip', 'conv2d.hip', 'anchor_assign.hip', 'roi_targets.hip', 'det2d_targets.hip', 'bev_interp.hip', 'ssl_match.hip', 'box_decode.hip', 'box_project.hip', 'consistency_loss.hip', 'rowgemm.hip', 'spconv16.hip', 'chain.hip', 'chain_tramp.hip', 'chain_ops.hip', 'fc_gemm.hip', 'sort_rows.hip']
HEADERS = ['dm_common.h', 'box_geom.h', '../../include/detmatch_hip.h', 'chain_tramp.inc']
# -ffp-contract=off: fused multiply-adds only where the source says fmaf(), so the
# CPU oracle (same flag) and the device agree bit for bit on index-deciding math.
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off',
         '-fno-fast-math', '-Wno-unused-result', '-DNDEBUG']


def _hipcc():
    for c in ('/opt/rocm/bin/hipcc', 'hipcc'):
        if os.path.exists(c) or c == 'hipcc':
            return c


def _sources():
    return [s for s in SOURCES if os.path.exists(os.path.join(HERE, s))]


def _stamp():
    h = hashlib.sha1()
    for f in _sources() + HEADERS:
        with open(os.path.join(HERE, f), 'rb') as fh:
            h.update(fh.read())
    h.update(' '.join(FLAGS).encode())
    return h.hexdigest()


def _src_stamp(src):
    """Hash of one source, the shared headers and the flags: an object is rebuilt only when this changes."""
    h = hashlib.sha1()
    for f in [src] + HEADERS:
        with open(os.path.join(HERE, f), 'rb') as fh:
            h.update(fh.read())
    h.update(' '.join(FLAGS).encode())
    return h.hexdigest()


def _compile(src):
    obj = os.path.join(HERE, src.replace('.hip', '.o'))
    tag, want = obj + '.stamp', _src_stamp(src)
    if os.path.exists(obj) and os.path.exists(tag) and open(tag).read() == want:
        return obj
    cmd = [_hipcc()] + FLAGS + ['-c', os.path.join(HERE, src), '-o', obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError('hipcc failed for %s:\n%s\n%s' % (src, r.stdout, r.stderr))
    with open(tag, 'w') as fh:
        fh.write(want)
    return obj


def build(force=False, verbose=False):
    stamp_file = os.path.join(HERE, '.build_stamp')
    stamp = _stamp()
    if (not force and os.path.exists(LIB) and os.path.exists(stamp_file)
            and open(stamp_file).read() == stamp):
        return LIB
    srcs = _sources()
    if force:
        for s_ in srcs:
            t_ = os.path.join(HERE, s_.replace('.hip', '.o.stamp'))
            if os.path.exists(t_):
                os.remove(t_)
    if verbose:
        print('[detmatch_amd] hipcc', ' '.join(FLAGS), srcs, file=sys.stderr)
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(6, len(srcs))) as ex:
        objs = list(ex.map(_compile, srcs))
    cmd = [_hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError('link failed:\n%s\n%s' % (r.stdout, r.stderr))
    with open(stamp_file, 'w') as fh:
        fh.write(stamp)
    return LIB


if __name__ == '__main__':
    print(build(force='-f' in sys.argv, verbose=True))
