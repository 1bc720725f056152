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
