"""The stand-alone detector interface of mmdet's BaseDetector (mmdet 2.14.0, un-vendored: parity
unpinned), as the pre-training recipes use it (configs/detmatch/001/pretrain_pvrcnn,
pretrain_frcnn with `EpochBasedRunner`): forward(return_loss=...) dispatch, train_step / val_step
and _parse_losses.  Mixed into OpenPCDetDetector and FasterRCNN; SSL has its own copies
(mmdet3d/models/detectors/ssl.py:214-253)."""
import torch
import torch.distributed as dist


class DetectorStepMixin(object):

    def _parse_losses(self, losses):
        """loss = sum of the entries whose key contains 'loss'; every logged value is averaged over
        ranks (one packed all-reduce instead of one per key) and stays on the device."""
        log_vars = {}
        for name, value in losses.items():
            if isinstance(value, torch.Tensor):
                log_vars[name] = value.mean()
            elif isinstance(value, list):
                log_vars[name] = sum(_l.mean() for _l in value)
            else:
                raise TypeError('%s is not a tensor or list of tensors' % name)
        loss = sum(v for k, v in log_vars.items() if 'loss' in k)
        log_vars['loss'] = loss
        keys = list(log_vars.keys())
        packed = torch.stack([log_vars[k].detach().float() for k in keys])
        if dist.is_available() and dist.is_initialized():
            from .parallel import all_reduce
            all_reduce(packed)
            packed = packed / dist.get_world_size()
        return loss, {k: packed[i] for i, k in enumerate(keys)}

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
