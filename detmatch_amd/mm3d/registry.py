"""Registry / build_from_cfg — just enough of the mmcv registry surface for
configs/detmatch/* to build (mmdet3d/models/builder.py:9-14,101-103)."""
import inspect


class Registry(object):

    def __init__(self, name):
        self._name = name
        self._module_dict = dict()

    @property
    def name(self):
        return self._name

    def get(self, key):
        return self._module_dict.get(key, None)

    def __contains__(self, key):
        return key in self._module_dict

    def _register(self, cls, name=None, force=False):
        name = name or cls.__name__
        if not force and name in self._module_dict:
            raise KeyError('%s is already registered in %s' % (name, self._name))
        self._module_dict[name] = cls

    def register_module(self, cls=None, name=None, force=False):
        """Usable as @R.register_module, @R.register_module() or R.register_module(cls)."""
        if cls is not None and inspect.isclass(cls):
            self._register(cls, name, force)
            return cls

        def _deco(c):
            self._register(c, name, force)
            return c
        return _deco

    def build(self, cfg, default_args=None):
        return build_from_cfg(cfg, self, default_args)


def build_from_cfg(cfg, registry, default_args=None):
    if not isinstance(cfg, dict) or 'type' not in cfg:
        raise TypeError('cfg must be a dict with the key "type", got %r' % (cfg,))
    args = dict(cfg)
    if default_args is not None:
        for k, v in default_args.items():
            args.setdefault(k, v)
    obj_type = args.pop('type')
    if isinstance(obj_type, str):
        obj_cls = registry.get(obj_type)
        if obj_cls is None:
            raise KeyError('%s is not in the %s registry' % (obj_type, registry.name))
    else:
        obj_cls = obj_type
    return obj_cls(**args)


MODELS = Registry('models')
DETECTORS = MODELS
SSL_MODULES = MODELS
LOSSES = Registry('loss')
BBOX_ASSIGNERS = Registry('bbox_assigner')
MATCH_COST = Registry('match_cost')
RUNNERS = Registry('runner')
HOOKS = Registry('hook')
OPTIMIZERS = Registry('optimizer')
OPTIMIZER_BUILDERS = Registry('optimizer builder')


def build_detector(cfg, train_cfg=None, test_cfg=None):
    """mmdet3d/models/builder.py build_detector"""
    return DETECTORS.build(cfg, default_args=dict(train_cfg=train_cfg, test_cfg=test_cfg))


def build_ssl_module(cfg):
    return SSL_MODULES.build(cfg)


def build_assigner(cfg):
    return BBOX_ASSIGNERS.build(cfg)


def build_match_cost(cfg):
    return MATCH_COST.build(cfg)


def build_loss(cfg):
    return LOSSES.build(cfg)
