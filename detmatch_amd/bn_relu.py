"""Fused training-mode BatchNorm (+ReLU) over (N, C) row-major activations — host side of
csrc/bn_relu.hip.  `bn_relu_rows(x, bn, relu)` computes exactly what `relu(bn(x))` does for an
nn.BatchNorm1d / nn.BatchNorm2d module `bn` whose channels are the last dim of `x`, including the
running-statistics and num_batches_tracked updates; evaluation mode without gradients (the EMA teacher)
is one launch of its own; shapes the kernels do not take, and evaluation mode under autograd, fall
through to torch.nn.functional.batch_norm."""

import torch
import torch.nn.functional as F
from torch.autograd import Function

from . import _lib


class _BNReLURows(Function):

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, eps, momentum, relu, pre=None):
        x = x.contiguous()
        n, c = x.shape
        L = _lib.lib()
        y = torch.empty_like(x)
        mean = torch.empty((c,), dtype=torch.float32, device=x.device)
        invstd = torch.empty_like(mean)
        if pre is not None:      # column statistics already reduced by the producer (dm_rowgemm_stats)
            _lib.check(L.dm_bn_rows_forward_pre(
                _lib.ptr(x), n, c, _lib.ptr(gamma), _lib.ptr(beta), float(eps), float(momentum),
                _lib.ptr(running_mean), _lib.ptr(running_var), int(relu), _lib.ptr(y), _lib.ptr(mean),
                _lib.ptr(invstd), _lib.ptr(pre[0]), _lib.ptr(pre[1]), int(pre[2]), _lib.stream()),
                'dm_bn_rows_forward_pre')
        else:
            ws = _lib.workspace(L.dm_bn_rows_workspace_bytes(n, c), x.device, 'bn_rows')
            _lib.check(L.dm_bn_rows_forward(
                _lib.ptr(x), n, c, _lib.ptr(gamma), _lib.ptr(beta), float(eps), float(momentum),
                _lib.ptr(running_mean), _lib.ptr(running_var), int(relu), _lib.ptr(y), _lib.ptr(mean),
                _lib.ptr(invstd), _lib.ptr(ws), ws.numel(), _lib.stream()), 'dm_bn_rows_forward')
        ctx.save_for_backward(x, gamma, beta, mean, invstd)
        ctx.relu = relu
        return y

    @staticmethod
    def backward(ctx, gy):
        x, gamma, beta, mean, invstd = ctx.saved_tensors
        gy = gy.contiguous()
        n, c = x.shape
        L = _lib.lib()
        gx = torch.empty_like(x)
        ggamma = torch.empty((c,), dtype=torch.float32, device=x.device)
        gbeta = torch.empty_like(ggamma)
        ws = _lib.workspace(L.dm_bn_rows_workspace_bytes(n, c), x.device, 'bn_rows')
        _lib.check(L.dm_bn_rows_backward(
            _lib.ptr(gy), _lib.ptr(x), n, c, _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(mean),
            _lib.ptr(invstd), int(ctx.relu), _lib.ptr(gx), _lib.ptr(ggamma), _lib.ptr(gbeta), _lib.ptr(ws),
            ws.numel(), _lib.stream()), 'dm_bn_rows_backward')
        return (gx, ggamma if gamma is not None else None, gbeta if beta is not None else None,
                None, None, None, None, None, None)


class _BNReLUMaxRows(Function):
    """max over the `ns` rows of every group of relu(bn(x)), x (M * ns, C) -> (M, C): the normalised tensor
    is never written, the backward goes from the pooled gradient straight to the dense input gradient
    (csrc/bn_relu.hip: bn_apply_max / bn_max_bwd_*)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, eps, momentum, ns, pre=None):
        x = x.contiguous()
        n, c = x.shape
        m = n // ns
        L = _lib.lib()
        pooled = torch.empty((m, c), dtype=torch.float32, device=x.device)
        arg = torch.empty((m, c), dtype=torch.uint8, device=x.device)
        mean = torch.empty((c,), dtype=torch.float32, device=x.device)
        invstd = torch.empty_like(mean)
        if pre is not None:
            _lib.check(L.dm_bn_rows_max_forward_pre(
                _lib.ptr(x), m, int(ns), c, _lib.ptr(gamma), _lib.ptr(beta), float(eps), float(momentum),
                _lib.ptr(running_mean), _lib.ptr(running_var), _lib.ptr(pooled), _lib.ptr(arg), _lib.ptr(mean),
                _lib.ptr(invstd), _lib.ptr(pre[0]), _lib.ptr(pre[1]), int(pre[2]), _lib.stream()),
                'dm_bn_rows_max_forward_pre')
        else:
            ws = _lib.workspace(L.dm_bn_rows_workspace_bytes(n, c), x.device, 'bn_rows')
            _lib.check(L.dm_bn_rows_max_forward(
                _lib.ptr(x), m, int(ns), c, _lib.ptr(gamma), _lib.ptr(beta), float(eps), float(momentum),
                _lib.ptr(running_mean), _lib.ptr(running_var), _lib.ptr(pooled), _lib.ptr(arg), _lib.ptr(mean),
                _lib.ptr(invstd), _lib.ptr(ws), ws.numel(), _lib.stream()), 'dm_bn_rows_max_forward')
        ctx.save_for_backward(x, gamma, beta, mean, invstd, arg)
        ctx.ns = int(ns)
        return pooled

    @staticmethod
    def backward(ctx, gp):
        x, gamma, beta, mean, invstd, arg = ctx.saved_tensors
        # the pooled gradient usually is a column block of the concatenated features' gradient: read in place
        if not (gp.dim() == 2 and gp.stride(1) == 1 and gp.stride(0) % 4 == 0 and gp.stride(0) >= gp.shape[1]
                and gp.data_ptr() % 16 == 0):
            gp = gp.contiguous()
        n, c = x.shape
        m = n // ctx.ns
        L = _lib.lib()
        gx = torch.empty_like(x)
        ggamma = torch.empty((c,), dtype=torch.float32, device=x.device)
        gbeta = torch.empty_like(ggamma)
        ws = _lib.workspace(L.dm_bn_rows_workspace_bytes(n, c), x.device, 'bn_rows')
        _lib.check(L.dm_bn_rows_max_backward_ld(
            gp.data_ptr(), int(gp.stride(0)), _lib.ptr(arg), _lib.ptr(x), m, ctx.ns, c, _lib.ptr(gamma), _lib.ptr(beta),
            _lib.ptr(mean), _lib.ptr(invstd), _lib.ptr(gx), _lib.ptr(ggamma), _lib.ptr(gbeta), _lib.ptr(ws),
            ws.numel(), _lib.stream()), 'dm_bn_rows_max_backward_ld')
        return (gx, ggamma if gamma is not None else None, gbeta if beta is not None else None,
                None, None, None, None, None, None)


# num_batches_tracked of the training-mode layers: inside SSL.forward_train the +1 of every layer call is
# collected and applied by ONE multi-tensor add before the EMA reads the counters (118 one-element launches per
# iteration otherwise); anywhere else the counter moves at once, like nn.BatchNorm's.
_DEFER = [False]
_PENDING = []


def _bump(bn):
    if _DEFER[0]:
        _PENDING.append(bn.num_batches_tracked)
    else:
        bn.num_batches_tracked.add_(1)


def flush_counters():
    if not _PENDING:
        return
    times = {}
    for t in _PENDING:
        e = times.setdefault(id(t), [t, 0])
        e[1] += 1
    del _PENDING[:]
    by_count = {}
    for t, n in times.values():
        by_count.setdefault((n, t.device, t.dtype), []).append(t)
    with torch.no_grad():
        for (n, _, _), ts in by_count.items():
            torch._foreach_add_(ts, n)


class deferred_counters(object):
    """with deferred_counters(): ... — see _DEFER; flushes on exit (also when the body raises)."""

    def __enter__(self):
        self.outer = _DEFER[0]
        _DEFER[0] = True
        return self

    def __exit__(self, *exc):
        _DEFER[0] = self.outer
        if not self.outer:
            flush_counters()
        return False


def _pre_stats(x, c):
    """Column statistics attached to x by the GEMM that produced it (TallSkinnyLinear, `dm_bn_pre`)."""
    pre = getattr(x, 'dm_bn_pre', None)
    return pre if (pre is not None and pre[0].shape[1] == c and x.is_contiguous()) else None


def bn_relu_rows_max(x, bn, ns):
    """relu(bn(x)).view(M, ns, C).max(dim=1)[0] for x (M * ns, C): one fused pass in training mode (the
    normalised tensor is not materialised), the two-step form otherwise."""
    c = x.shape[-1]
    training = bn.training or not bn.track_running_stats
    if training and bn.momentum is not None and _kernel_takes(x, c) and 1 <= ns <= 255 and \
            x.shape[0] % ns == 0 and (bn.weight is None) == (bn.bias is None):
        if bn.track_running_stats and bn.num_batches_tracked is not None:
            _bump(bn)
        rm = bn.running_mean if bn.track_running_stats else None
        rv = bn.running_var if bn.track_running_stats else None
        return _BNReLUMaxRows.apply(x, bn.weight, bn.bias, rm, rv, bn.eps, bn.momentum, ns, _pre_stats(x, c))
    if not training and _kernel_takes(x, c) and 1 <= ns <= 255 and x.shape[0] % ns == 0 and \
            (bn.weight is None) == (bn.bias is None) and \
            not (torch.is_grad_enabled() and (x.requires_grad or (bn.weight is not None and bn.weight.requires_grad))):
        x = x.contiguous()
        pooled = torch.empty((x.shape[0] // ns, c), dtype=torch.float32, device=x.device)
        _lib.check(_lib.lib().dm_bn_rows_eval_max(
            _lib.ptr(x), x.shape[0] // ns, int(ns), c, _lib.ptr(bn.weight), _lib.ptr(bn.bias),
            _lib.ptr(bn.running_mean), _lib.ptr(bn.running_var), float(bn.eps), _lib.ptr(pooled), _lib.stream()),
            'dm_bn_rows_eval_max')
        return pooled
    y = bn_relu_rows(x, bn, relu=True)
    return y.view(x.shape[0] // ns, ns, c).max(dim=1)[0]


def _kernel_takes(x, c):
    return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.shape[0] > 1 and c % 4 == 0
            and 4 <= c <= 1024 and 256 % (c // 4) == 0)


def bn_relu_rows(x, bn, relu=True):
    """relu(bn(x)) for x (N, C), bn an nn.BatchNorm{1,2}d over C."""
    c = x.shape[-1]
    training = bn.training or not bn.track_running_stats
    if training and bn.momentum is not None and _kernel_takes(x, c) and \
            (bn.weight is None) == (bn.bias is None):
        if bn.track_running_stats and bn.num_batches_tracked is not None:
            _bump(bn)
        rm = bn.running_mean if bn.track_running_stats else None
        rv = bn.running_var if bn.track_running_stats else None
        return _BNReLURows.apply(x, bn.weight, bn.bias, rm, rv, bn.eps, bn.momentum, relu, _pre_stats(x, c))
    if not training and _kernel_takes(x, c) and (bn.weight is None) == (bn.bias is None) and \
            not (torch.is_grad_enabled() and (x.requires_grad or (bn.weight is not None and bn.weight.requires_grad))):
        # inference (the EMA teacher): normalisation with the running statistics + ReLU in one launch
        x = x.contiguous()
        y = torch.empty_like(x)
        _lib.check(_lib.lib().dm_bn_rows_eval(
            _lib.ptr(x), x.shape[0], c, _lib.ptr(bn.weight), _lib.ptr(bn.bias), _lib.ptr(bn.running_mean),
            _lib.ptr(bn.running_var), float(bn.eps), int(relu), _lib.ptr(y), _lib.stream()), 'dm_bn_rows_eval')
        return y
    if bn.training and bn.track_running_stats and bn.num_batches_tracked is not None:
        bn.num_batches_tracked.add_(1)
    y = F.batch_norm(x, bn.running_mean, bn.running_var, bn.weight, bn.bias, training,
                     bn.momentum if bn.momentum is not None else 0.0, bn.eps)
    return F.relu(y, inplace=True) if relu else y


def _linear_rows(x, m):
    # (the RoI head's 27648 -> 256 layer was also tried as a split-K 1x1 convolution on conv2d.hip:
    # 290 us forward + backward against 212 us of the BLAS kernels torch picks — kept on BLAS)
    # (vendor GEMM: only ever inside _lib.blas_turn — one Stream-K kernel at a time)
    w = m.weight
    return _lib.blas_linear(x, w.squeeze(-1) if w.dim() == 3 else w, m.bias)


def fc_rows(seq, x):
    """An nn.Sequential of Linear / Conv1d(kernel 1) + BatchNorm1d + ReLU (+ Dropout) layers — the FC
    stacks of the PV-RCNN heads (pvrcnn_head.py:25-52 builds them from Conv1d on (N, C, 1) tensors,
    point_head_template.py:34-47 and voxel_set_abstraction.py:107-111 from Linear) — evaluated on the
    (N, C) row view with the same parameters and state-dict keys: a plain GEMM per layer and the
    fused BatchNorm+ReLU row kernel instead of one convolution / batch-norm library call per layer."""
    import torch.nn as nn
    mods = list(seq)
    i = 0
    while i < len(mods):
        m = mods[i]
        if isinstance(m, (nn.Conv1d, nn.Linear)):
            if isinstance(m, nn.Conv1d):
                assert m.kernel_size == (1,) and m.stride == (1,) and m.padding == (0,) and m.groups == 1
            x = _linear_rows(x, m)
        elif isinstance(m, nn.BatchNorm1d):
            relu = i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)
            x = bn_relu_rows(x, m, relu=relu)
            i += int(relu)
        elif isinstance(m, nn.Dropout):
            x = F.dropout(x, m.p, m.training, False)
        elif isinstance(m, nn.ReLU):
            x = F.relu(x)
        else:
            raise TypeError('unexpected layer %s in an FC stack' % type(m).__name__)
        i += 1
    return x
