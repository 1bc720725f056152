"""Which host-side region launches how many kernels?  One profiled DetMatch (or PV-RCNN) step with
record_function ranges around every SSL module, every stage of the 3D / 2D detectors, backward and
the optimizer; prints launches and device time per range (inclusive).

    python tools/launch_census.py [detmatch|pvrcnn|confthr] [kernel-name regex: count only those]
"""
import os as _os
_os.environ.setdefault('GPU_MAX_HW_QUEUES', '4')   # the runtime's default, pinned: with RCCL initialised 5+ hardware queues cost +30 ms per iteration (detmatch_amd/__init__.py)
import collections
import re
import sys

import torch
from torch.profiler import ProfilerActivity, profile, record_function

sys.path.insert(0, '.')


def wrap(obj, attr, label):
    fn = getattr(obj, attr)

    def inner(*a, **k):
        with record_function('DM:' + label):
            return fn(*a, **k)
    setattr(obj, attr, inner)


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else 'detmatch'
    dev = torch.device('cuda', 0)
    from detmatch_amd import synth
    from detmatch_amd.pcdet.workload import DetMatchTrainWorkload, PVRCNNTrainWorkload
    if which == 'pvrcnn':
        wl = PVRCNNTrainWorkload([synth.lidar_frame(i) for i in range(2)], dev)
        dets3d = {'': wl.model}
        dets2d = {}
    else:
        wl = DetMatchTrainWorkload(2, dev, ssl_cfg='confthr_pvrcnn' if which == 'confthr' else None)
        m = wl.model
        for tag, lst in (('lab', m.lab_ssl_modules), ('unlab', m.unlab_ssl_modules)):
            for i, mod in enumerate(lst):
                wrap(mod, 'forward', 'ssl.%s.%02d.%s' % (tag, i, type(mod).__name__))
        wrap(m, '_update_teacher', 'ema')
        dets3d = {'stu.': m.student.detector_3d, 'tea.': m.teacher.detector_3d}
        dets2d = {'stu.': m.student.detector_2d, 'tea.': m.teacher.detector_2d}
        for h in wl.runner._hooks:
            if type(h).__name__ == 'OptimizerHook':
                wrap(h, 'after_train_iter', 'backward+clip+optimizer')
    for tag, d in dets3d.items():
        wrap(d, '_base_batch', tag + '3d.voxelize+batch')
        for name in d.model.module_topology:
            wrap(getattr(d.model, name), 'forward', tag + '3d.' + name)
        wrap(d.model, 'get_training_loss', tag + '3d.losses')
        for name, methods in (('roi_head', ('proposal_layer', 'assign_targets', 'roi_grid_pool', 'get_loss',
                                            'generate_predicted_boxes')),
                              ('dense_head', ('assign_targets', 'generate_predicted_boxes', 'get_loss')),
                              ('point_head', ('assign_targets', 'get_loss')),
                              ('pfe', ('get_sampled_points', 'interpolate_from_bev_features'))):
            mod = getattr(d.model, name, None)
            for meth in methods:
                if mod is not None and hasattr(mod, meth):
                    wrap(mod, meth, tag + '3d.%s.%s' % (name, meth))
        wrap(d.model, 'post_processing', tag + '3d.post_processing')
    for tag, d in dets2d.items():
        wrap(d, 'extract_feat', tag + '2d.backbone+fpn')
        wrap(d, '_trunk', tag + '2d.trunk(backbone+fpn+rpn)')
        wrap(d.rpn_head, 'forward', tag + '2d.rpn.forward')
        wrap(d.rpn_head, 'loss', tag + '2d.rpn.loss')
        wrap(d.rpn_head, 'get_bboxes', tag + '2d.rpn.get_bboxes')
        wrap(d.roi_head, 'forward_train', tag + '2d.roi.forward_train')
        wrap(d.roi_head, 'simple_test_pre_nms', tag + '2d.roi.test')
    for _ in range(4):
        wl.step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        wl.step()
        torch.cuda.synchronize()
    agg = collections.OrderedDict()

    pat = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None

    def walk(e):
        ks = [k for k in e.kernels if pat is None or pat.search(k.name)]
        n, t = len(ks), sum(k.duration for k in ks)
        for c in e.cpu_children:
            cn, ct = walk(c)
            n, t = n + cn, t + ct
        return n, t
    total_n = total_t = 0
    for e in prof.events():
        if e.cpu_parent is None:
            n, t = walk(e)
            total_n, total_t = total_n + n, total_t + t
        if e.name.startswith('DM:'):
            n, t = walk(e)
            a = agg.setdefault(e.name[3:], [0, 0, 0.0, 0.0])
            a[0] += 1
            a[1] += n
            a[2] += t
            a[3] += e.cpu_time_total
    print('%-52s %6s %9s %10s %10s' % ('range (inclusive)', 'calls', 'launches', 'gpu us', 'cpu us'))
    for k, (c, n, t, cpu) in agg.items():
        print('%-52s %6d %9d %10.0f %10.0f' % (k, c, n, t, cpu))
    print('%-52s %6s %9d %10.0f' % ('TOTAL step', '', total_n, total_t))


if __name__ == '__main__':
    main()
