"""The stand-alone detector interface of mmdet's BaseDetector (mmdet 2.14.0, un-vendored: parity
unpinned), as the pre-training recipes use it (configs/detmatch/001/pretrain_pvrcnn,
pretrain_frcnn with `EpochBasedRunner`): forward(return_loss=...) dispatch, train_step / val_step
and _parse_losses.  Mixed into OpenPCDetDetector and FasterRCNN; SSL has its own copies
(mmdet3d/models/detectors/ssl.py:214-253)."""
import torch
import torch.distributed as dist


class LazyLogVars(dict):
    """log_vars of a train_step whose values are produced on first access: {key: 0-d device tensor}, every value
    averaged over the ranks by ONE packed all-reduce.  A dict for every reader (`items`, `[]`, `update`, iteration);
    nothing is launched until one of them looks."""

    def __init__(self, means, loss):
        super().__init__()
        self._pending = (means, loss)

    def _fill(self):
        if self._pending is None:
            return
        (means, loss), self._pending = self._pending, None
        vals = dict(means)
        vals['loss'] = loss
        keys = list(vals.keys())
        with torch.no_grad():
            packed = torch.stack([vals[k].detach().float() for k in keys])
            if dist.is_available() and dist.is_initialized():
                from .parallel import all_reduce
                all_reduce(packed)
                packed = packed / dist.get_world_size()
        extra = dict(super().items())            # entries somebody update()d in before the values existed
        super().clear()
        for i, k in enumerate(keys):
            super().__setitem__(k, packed[i])
        for k, v in extra.items():
            super().__setitem__(k, v)

    def update(self, *a, **k):
        if self._pending is not None and not a and not k:
            return
        if self._pending is not None:
            # merged in without forcing the values (SSL.train_step adds its own log_vars): remembered, re-applied by _fill
            for key, v in dict(*a, **k).items():
                super().__setitem__(key, v)
            return
        super().update(*a, **k)

    def __getitem__(self, k):
        self._fill()
        return super().__getitem__(k)

    def __contains__(self, k):
        self._fill()
        return super().__contains__(k)

    def __iter__(self):
        self._fill()
        return super().__iter__()

    def __len__(self):
        self._fill()
        return super().__len__()

    def keys(self):
        self._fill()
        return super().keys()

    def values(self):
        self._fill()
        return super().values()

    def items(self):
        self._fill()
        return super().items()

    def get(self, k, d=None):
        self._fill()
        return super().get(k, d)


class DetectorStepMixin(object):

    def _parse_losses(self, losses):
        """loss = sum of the entries whose key contains 'loss'; every logged value is averaged over
        ranks (one packed all-reduce instead of one per key) and stays on the device.
        The LOSS comes first and alone: the logged values (a mean, a cast and a stack entry per key, ~70 tiny launches
        for the DetMatch recipe) are packed lazily — `LazyLogVars` computes them when somebody reads them, which the
        runner does after it has issued the backward pass (the device's main lane was idle for that long)."""
        means = {}
        for name, value in losses.items():
            if isinstance(value, torch.Tensor):
                means[name] = value if value.dim() == 0 else value.mean()
            elif isinstance(value, list):
                means[name] = sum(_l.mean() for _l in value)
            else:
                raise TypeError('%s is not a tensor or list of tensors' % name)
        terms = [v for k, v in means.items() if 'loss' in k]
        loss = terms[0] if len(terms) == 1 else torch.stack(terms).sum()
        return loss, LazyLogVars(means, loss)

    def train_step(self, data, optimizer=None):
        losses = self(**data)
        loss, log_vars = self._parse_losses(losses)
        return dict(loss=loss, log_vars=log_vars, num_samples=len(data['img_metas']))

    val_step = train_step

    def forward(self, return_loss=True, **kwargs):
        if return_loss:
            return self.forward_train(**kwargs)
        return self.forward_test(**kwargs)

    def forward_test(self, img_metas, **kwargs):
        """Single-augmentation test batches: `[batch]` lists (as MultiScaleFlipAug collates them) are
        unwrapped; test-time augmentation is outside the DetMatch path."""
        if img_metas and not isinstance(img_metas[0], dict):
            assert len(img_metas) == 1, 'test-time augmentation is outside the DetMatch path'
            kwargs = {k: (v[0] if isinstance(v, (list, tuple)) and len(v) == 1 else v)
                      for k, v in kwargs.items()}
            img_metas = img_metas[0]
        return self.simple_test(img_metas=img_metas, **kwargs)
