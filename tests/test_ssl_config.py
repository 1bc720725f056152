"""The reference's own DetMatch config files load unchanged and resolve against the registries
(SURVEY §8(b) B1); the in-repo builders (used on the GPU box, where /root/reference is absent)
reproduce them value for value."""
import os

import pytest

from detmatch_amd import configs
from detmatch_amd.mm3d.config import Config

REF = '/root/reference/configs/detmatch/001'
needs_ref = pytest.mark.skipif(not os.path.isdir(REF), reason='reference configs not present')


def _plain(x):
    if isinstance(x, dict):
        return {k: _plain(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_plain(v) for v in x]
    return x


@needs_ref
def test_detmatch_builder_equals_reference_config():
    cfg = Config.fromfile(os.path.join(REF, 'detmatch/split_0.py'))
    ref = _plain(cfg['model'])
    mine = _plain(configs.detmatch_kitti_model(pretrained=ref['pretrained']))
    assert mine == ref
    sched = configs.detmatch_schedule(batch_size=4, num_unlabeled_samples=1, max_iters=5000)
    for k, v in sched.items():
        assert _plain(v) == _plain(cfg[k]), k


@needs_ref
def test_confthr_builder_equals_reference_config():
    cfg = Config.fromfile(os.path.join(REF, 'confthr_pvrcnn/split_0.py'))
    assert _plain(configs.confthr_pvrcnn_ssl_cfg()) == _plain(cfg['model']['ssl_cfg'])


@needs_ref
def test_confthr_frcnn_and_pretrain_builders_equal_reference_configs():
    """the other recipes of configs/detmatch/001 on the same kernels."""
    cfg = Config.fromfile(os.path.join(REF, 'confthr_frcnn/split_0.py'))
    assert _plain(configs.confthr_frcnn_ssl_cfg()) == _plain(cfg['model']['ssl_cfg'])
    ref_model = _plain(cfg['model'])
    mine = _plain(configs.detmatch_kitti_model(ssl_cfg=configs.confthr_frcnn_ssl_cfg(),
                                               pretrained=ref_model['pretrained']))
    assert mine == ref_model
    cfg = Config.fromfile(os.path.join(REF, 'pretrain_pvrcnn/split_0.py'))
    assert _plain(configs.pvrcnn_kitti_model()) == _plain(cfg['model'])
    for k, v in configs.pretrain_pvrcnn_schedule(batch_size=cfg['batch_size']).items():
        assert _plain(v) == _plain(cfg[k]), k
    cfg = Config.fromfile(os.path.join(REF, 'pretrain_frcnn/split_0.py'))
    for k, v in configs.pretrain_frcnn_schedule(batch_size=cfg['batch_size']).items():
        assert _plain(v) == _plain(cfg[k]), k


@needs_ref
def test_reference_config_types_resolve():
    """Every `type` named inside model / optimizer / runner / hooks resolves to a registered class."""
    import detmatch_amd.mm2d  # noqa: F401
    import detmatch_amd.mm3d.openpcdet  # noqa: F401
    import detmatch_amd.mm3d.ssl_modules  # noqa: F401
    from detmatch_amd.mm3d import registry as R
    from detmatch_amd.mm3d import runner  # noqa: F401
    cfg = Config.fromfile(os.path.join(REF, 'detmatch/split_0.py'))
    model = cfg['model']
    assert model['type'] in R.DETECTORS and model['model_cfg']['type'] in R.DETECTORS
    for d in ('detector_2d', 'detector_3d'):
        assert model['model_cfg'][d]['type'] in R.DETECTORS, d
    for chain in model['ssl_cfg'].values():
        for m in chain:
            assert m['type'] in R.SSL_MODULES, m['type']
            if 'assigner_cfg' in m:
                a = m['assigner_cfg']
                assert a['type'] in R.BBOX_ASSIGNERS
                for c in ('cls_cost', 'reg_cost', 'iou_cost'):
                    assert a[c]['type'] in R.MATCH_COST
            for lk in ('loss_cls_cfg', 'loss_iou_cfg', 'loss_l1_cfg'):
                if m.get(lk):
                    assert m[lk]['type'] in R.LOSSES
    assert cfg['runner']['type'] in R.RUNNERS
    assert cfg['optimizer']['constructor'] in R.OPTIMIZER_BUILDERS
    for h in cfg['custom_hooks']:
        assert h['type'] in R.HOOKS


def test_ssl_modules_build_from_builder_config():
    import detmatch_amd.mm3d.ssl_modules  # noqa: F401
    from detmatch_amd.mm3d.registry import build_ssl_module
    for chain in configs.detmatch_ssl_cfg().values():
        for m in chain:
            build_ssl_module(m)


@needs_ref
def test_reference_data_section_resolves_and_compiles():
    """B1 data section: every `type` of the reference config's data / pipeline
    sections is in the DATASETS / PIPELINES registries, and the three pipeline lists compile into the
    device loader's program with exactly the reference's parameters."""
    from detmatch_amd.mm3d import datasets as D
    cfg = Config.fromfile(os.path.join(REF, 'detmatch/split_0.py'))

    def walk(node, found):
        if isinstance(node, dict):
            if 'type' in node and isinstance(node['type'], str):
                found.append(node['type'])
            for v in node.values():
                walk(v, found)
        elif isinstance(node, (list, tuple)):
            for v in node:
                walk(v, found)
        return found

    for name in ('labeled_shared_pipeline', 'labeled_student_pipeline', 'labeled_teacher_pipeline',
                 'unlabeled_shared_pipeline', 'unlabeled_student_pipeline', 'unlabeled_teacher_pipeline',
                 'test_pipeline'):
        for t in walk(cfg[name], []):
            assert t in D.PIPELINES, (name, t)
        D.build_pipeline(cfg[name])                       # argument names are validated too
    for t in ('TS_SSL_Dataset', 'RepeatDataset', cfg['dataset_type']):
        assert t in D.DATASETS, t
    lab = D.compile_pipelines(D.build_pipeline(cfg['labeled_shared_pipeline']),
                              D.build_pipeline(cfg['labeled_student_pipeline']),
                              D.build_pipeline(cfg['labeled_teacher_pipeline']), labeled=True)
    unlab = D.compile_pipelines(D.build_pipeline(cfg['unlabeled_shared_pipeline']),
                                D.build_pipeline(cfg['unlabeled_student_pipeline']),
                                D.build_pipeline(cfg['unlabeled_teacher_pipeline']), labeled=False)
    assert lab['db_sampler']['sample_groups'] == dict(Car=15, Pedestrian=10, Cyclist=10)
    assert unlab['db_sampler'] is None
    for a in (lab, unlab):
        assert a['img_scale'] == ((640, 192), (2560, 768)) and a['flip_ratio'] == 0.5
        assert a['rot_range'] == (-0.78539816, 0.78539816) and a['scale_ratio_range'] == (0.95, 1.05)
        assert a['point_cloud_range'] == list(cfg['point_cloud_range'])
        assert a['img_mean'] == (103.530, 116.280, 123.675) and a['size_divisor'] == 32
        ph = a['student_photometric']
        assert ph['jitter'] == (0.4, 0.4, 0.4, 0.1) and ph['p_jitter'] == 0.8 and ph['p_grey'] == 0.2
        assert ph['blur_sigma'] == (0.1, 2.0) and ph['p_blur'] == 0.5
        assert ph['erasing'] == ((0.7, (0.05, 0.2), (0.3, 3.3)), (0.5, (0.02, 0.2), (0.1, 6)),
                                 (0.3, (0.02, 0.2), (0.05, 8)))
    # a pipeline the device program does not implement is an error, not a silent skip
    bad = [dict(t) for t in cfg['unlabeled_student_pipeline']]
    bad[0], bad[1] = bad[1], bad[0]
    with pytest.raises(ValueError):
        D.compile_pipelines(D.build_pipeline(cfg['unlabeled_shared_pipeline']), D.build_pipeline(bad),
                            D.build_pipeline(cfg['unlabeled_teacher_pipeline']), labeled=False)
    with pytest.raises(TypeError):
        D.build_pipeline([dict(type='PointsRangeFilter', point_cloud_range=[0] * 6, bogus=1)])


def test_rank_aware_device_loader_partitions_the_epoch():
    """ADVICE r1 (ts_ssl_dataset.py:158): ranks share the epoch's permutation and take disjoint slices;
    set_epoch reshuffles consistently."""
    from detmatch_amd.ts_ssl_dataset import TSSSLDeviceLoader

    class Fake(object):
        def __len__(self):
            return 40

    seen = []
    for rank in range(4):
        ld = TSSSLDeviceLoader(Fake(), 2, 'cpu', labeled=False, point_cloud_range=[0, -40, -3, 70.4, 40, 1],
                               seed=3, rank=rank, world_size=4, with_img=False, student_photometric=False)
        idx = np.concatenate(ld._indices())
        assert len(idx) == 10 and len(ld) == 5
        seen.append(idx)
        ld.set_epoch(1)
        assert not np.array_equal(np.concatenate(ld._indices()), idx)
    allidx = np.concatenate(seen)
    assert sorted(allidx.tolist()) == list(range(40))        # a partition of the dataset


import numpy as np  # noqa: E402


@needs_ref
def test_data_builder_equals_reference_config():
    cfg = Config.fromfile(os.path.join(REF, 'detmatch/split_0.py'))
    assert _plain(configs.detmatch_data()) == _plain(cfg['data'])
    pipes = configs.detmatch_pipelines(cfg['data_root'], cfg['db_info_path'])
    for k, v in pipes.items():
        assert _plain(v) == _plain(cfg[k]), k
