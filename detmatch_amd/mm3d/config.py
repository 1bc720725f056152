"""Config.fromfile — loads a flattened mmcv-style Python config file (the DetMatch configs
have no `_base_`) into an attribute-access dict, with `--cfg-options a.b=c`
style overrides (tools/train.py:59-68,97-99)."""
import os
import types

from ..pcdet.config import ConfigDict


class Config(ConfigDict):

    @staticmethod
    def fromfile(filename):
        filename = os.path.abspath(os.path.expanduser(filename))
        if not os.path.isfile(filename):
            raise FileNotFoundError(filename)
        ns = {'__file__': filename, '__name__': '_detmatch_cfg_'}
        with open(filename) as f:
            code = compile(f.read(), filename, 'exec')
        exec(code, ns)
        cfg = Config()
        for k, v in ns.items():
            if k.startswith('__') or isinstance(v, (types.ModuleType, types.FunctionType, type)):
                continue
            cfg[k] = v
        cfg['filename'] = filename
        return cfg

    def merge_from_dict(self, options):
        """{'a.b.c': v} -> self.a.b.c = v (lists may be indexed: 'a.0.b')."""
        for full_key, v in options.items():
            d = self
            keys = full_key.split('.')
            for sub in keys[:-1]:
                if isinstance(d, (list, tuple)):
                    d = d[int(sub)]
                else:
                    if sub not in d:
                        d[sub] = ConfigDict()
                    d = d[sub]
            last = keys[-1]
            if isinstance(d, list):
                d[int(last)] = v
            else:
                d[last] = v
