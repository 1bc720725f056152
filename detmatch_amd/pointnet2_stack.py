"""Stacked-batch PointNet++ ops and modules — host-side mirror of
pcdet/ops/pointnet2/pointnet2_stack/{pointnet2_utils,pointnet2_modules}.py.

Same names / argument meaning (`ball_query`, `grouping_operation`,
`furthest_point_sample`, `QueryAndGroup`, `StackSAModuleMSG`); compute in
libdetmatch_hip.so (pointnet2_stack.hip).
"""
import os
import weakref
from typing import List

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.autograd import Function

from . import _lib
from .bn_relu import bn_relu_rows, bn_relu_rows_max
from .fused import on as fused_on


def _i32(t):
    return t if t.dtype == torch.int32 else t.int()


class BallQuery(Function):
    """pointnet2_utils.py:8-44"""

    @staticmethod
    def forward(ctx, radius, nsample, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt,
                max_m_per_sample=0):
        xyz = xyz.contiguous()
        new_xyz = new_xyz.contiguous()
        xyz_batch_cnt = _i32(xyz_batch_cnt).contiguous()
        new_xyz_batch_cnt = _i32(new_xyz_batch_cnt).contiguous()
        _lib.require_device(xyz, new_xyz, xyz_batch_cnt, new_xyz_batch_cnt)
        B = xyz_batch_cnt.shape[0]
        M = new_xyz.shape[0]
        # the kernel writes all nsample slots and the empty flag (one byte, 0 / 1) of every query
        idx = torch.empty((M, nsample), dtype=torch.int32, device=xyz.device)
        empty = torch.empty((M,), dtype=torch.bool, device=xyz.device)
        rc = _lib.lib().dm_ball_query_stack(B, M, float(radius), int(nsample), _lib.ptr(new_xyz),
                                            _lib.ptr(new_xyz_batch_cnt), _lib.ptr(xyz),
                                            _lib.ptr(xyz_batch_cnt), int(max_m_per_sample),
                                            _lib.ptr(idx), _lib.ptr(empty), _lib.stream())
        _lib.check(rc, 'dm_ball_query_stack')
        ctx.mark_non_differentiable(idx, empty)
        return idx, empty

    @staticmethod
    def backward(ctx, a=None, b=None):
        return None, None, None, None, None, None, None


ball_query = BallQuery.apply


@torch.no_grad()
def ball_query_pair(radius_a, nsample_a, radius_b, nsample_b, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt):
    """Two ball queries around the same centres in one scan of the points (dm_ball_query_stack2):
    -> ((idx_a, empty_a), (idx_b, empty_b)), each as `ball_query` returns it."""
    xyz, new_xyz = xyz.contiguous(), new_xyz.contiguous()
    xyz_batch_cnt, new_xyz_batch_cnt = _i32(xyz_batch_cnt).contiguous(), _i32(new_xyz_batch_cnt).contiguous()
    _lib.require_device(xyz, new_xyz, xyz_batch_cnt, new_xyz_batch_cnt)
    m, dev = new_xyz.shape[0], xyz.device
    ia = torch.empty((m, nsample_a), dtype=torch.int32, device=dev)
    ib = torch.empty((m, nsample_b), dtype=torch.int32, device=dev)
    ea = torch.empty((m,), dtype=torch.bool, device=dev)
    eb = torch.empty((m,), dtype=torch.bool, device=dev)
    _lib.check(_lib.lib().dm_ball_query_stack2(
        xyz_batch_cnt.shape[0], m, float(radius_a), int(nsample_a), float(radius_b), int(nsample_b),
        _lib.ptr(new_xyz), _lib.ptr(new_xyz_batch_cnt), _lib.ptr(xyz), _lib.ptr(xyz_batch_cnt), _lib.ptr(ia),
        _lib.ptr(ib), _lib.ptr(ea), _lib.ptr(eb), _lib.stream()), 'dm_ball_query_stack2')
    return (ia, ea), (ib, eb)


class GroupingOperation(Function):
    """pointnet2_utils.py:48-113.  `empty_mask` (extension): rows of empty balls are
    written as zeros by the kernel instead of a separate masked assignment."""

    @staticmethod
    def forward(ctx, features, features_batch_cnt, idx, idx_batch_cnt, empty_mask=None):
        features = features.contiguous()
        features_batch_cnt = _i32(features_batch_cnt).contiguous()
        idx_batch_cnt = _i32(idx_batch_cnt).contiguous()
        idx = idx.contiguous()
        _lib.require_device(features, features_batch_cnt, idx, idx_batch_cnt)
        M, nsample = idx.size()
        N, C = features.size()
        B = idx_batch_cnt.shape[0]
        output = torch.empty((M, C, nsample), dtype=torch.float32, device=features.device)
        em = None
        if empty_mask is not None:
            em = empty_mask.to(torch.uint8).contiguous()
        rc = _lib.lib().dm_group_points_stack(B, M, C, nsample, _lib.ptr(features),
                                              _lib.ptr(features_batch_cnt), _lib.ptr(idx),
                                              _lib.ptr(idx_batch_cnt), _lib.ptr(em),
                                              _lib.ptr(output), _lib.stream())
        _lib.check(rc, 'dm_group_points_stack')
        ctx.for_backwards = (B, N, idx, features_batch_cnt, idx_batch_cnt, em)
        return output

    @staticmethod
    def backward(ctx, grad_out):
        B, N, idx, features_batch_cnt, idx_batch_cnt, em = ctx.for_backwards
        M, C, nsample = grad_out.size()
        grad_out = grad_out.contiguous()
        if em is not None:  # zeroed rows carry no gradient
            grad_out = grad_out * (em == 0).view(-1, 1, 1).to(grad_out.dtype)
        grad_features = torch.empty((N, C), dtype=torch.float32, device=grad_out.device)
        rc = _lib.lib().dm_group_points_grad_stack(B, M, C, N, nsample, _lib.ptr(grad_out),
                                                   _lib.ptr(idx), _lib.ptr(idx_batch_cnt),
                                                   _lib.ptr(features_batch_cnt),
                                                   _lib.ptr(grad_features), _lib.stream())
        _lib.check(rc, 'dm_group_points_grad_stack')
        return grad_features, None, None, None, None


grouping_operation = GroupingOperation.apply


class QueryAndGroup(nn.Module):
    """pointnet2_utils.py:116-156"""

    def __init__(self, radius: float, nsample: int, use_xyz: bool = True):
        super().__init__()
        self.radius, self.nsample, self.use_xyz = radius, nsample, use_xyz

    def forward(self, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, features=None):
        idx, empty_ball_mask = ball_query(self.radius, self.nsample, xyz, xyz_batch_cnt, new_xyz,
                                          new_xyz_batch_cnt)
        grouped_xyz = grouping_operation(xyz, xyz_batch_cnt, idx, new_xyz_batch_cnt)
        grouped_xyz = grouped_xyz - new_xyz.unsqueeze(-1)
        grouped_xyz = grouped_xyz * (~empty_ball_mask).view(-1, 1, 1).to(grouped_xyz.dtype)
        if features is not None:
            grouped_features = grouping_operation(features, xyz_batch_cnt, idx, new_xyz_batch_cnt,
                                                  empty_ball_mask)
            if self.use_xyz:
                new_features = torch.cat([grouped_xyz, grouped_features], dim=1)
            else:
                new_features = grouped_features
        else:
            assert self.use_xyz, 'Cannot have not features and not use xyz as a feature!'
            new_features = grouped_xyz
        return new_features, idx


class QueryGroupRows(Function):
    """Fused QueryAndGroup gather in row layout: -> (M, nsample, [4+]C) with
    rows[m,s,0:4] = [xyz[src] - new_xyz[m], 0], rows[m,s,4:] = features[src], zero rows for empty balls
    (the content of pointnet2_utils.py:139-153, one launch, no cat).  Gradient: features only."""

    @staticmethod
    def forward(ctx, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, features, idx, empty_mask,
                use_xyz):
        _lib.require_device(xyz)
        xyz, new_xyz = xyz.contiguous(), new_xyz.contiguous()
        xyz_batch_cnt = _i32(xyz_batch_cnt).contiguous()
        new_xyz_batch_cnt = _i32(new_xyz_batch_cnt).contiguous()
        c = 0 if features is None else features.shape[1]
        if features is not None:
            features = features.contiguous()
        m, ns = idx.shape
        width = (4 if use_xyz else 0) + c      # xyz rides in a 16-byte slot [dx, dy, dz, 0]
        out = torch.empty((m, ns, width), dtype=torch.float32, device=xyz.device)
        em = empty_mask.contiguous().view(torch.uint8) if empty_mask is not None else None
        _lib.check(_lib.lib().dm_query_group_rows(
            xyz_batch_cnt.numel(), m, c, ns, int(use_xyz), _lib.ptr(xyz), _lib.ptr(new_xyz),
            _lib.ptr(features) if features is not None else None, _lib.ptr(xyz_batch_cnt),
            _lib.ptr(new_xyz_batch_cnt), _lib.ptr(idx), _lib.ptr(em) if em is not None else None,
            _lib.ptr(out), _lib.stream()), 'dm_query_group_rows')
        ctx.save_for_backward(idx, xyz_batch_cnt, new_xyz_batch_cnt, em)
        ctx.meta = (features.shape[0] if features is not None else 0, c, width, 4 if use_xyz else 0)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        idx, xyz_cnt, new_cnt, em = ctx.saved_tensors
        n, c, width, off = ctx.meta
        gfeat = None
        if c > 0 and ctx.needs_input_grad[4]:
            grad_out = grad_out.contiguous()
            m, ns = idx.shape
            gfeat = torch.empty((n, c), dtype=torch.float32, device=grad_out.device)
            _lib.check(_lib.lib().dm_group_rows_grad(
                xyz_cnt.numel(), m, c, n, ns, width, off, _lib.ptr(grad_out), _lib.ptr(idx),
                _lib.ptr(new_cnt), _lib.ptr(xyz_cnt), _lib.ptr(em) if em is not None else None,
                _lib.ptr(gfeat), _lib.stream()), 'dm_group_rows_grad')
        return None, None, None, None, gfeat, None, None, None


class TallSkinnyLinear(Function):
    """y = x @ w^T for x (R, Cin) with R in the hundreds of thousands and Cin, Cout <= a few hundred
    (the shared MLP over all grouped references).  Forward / input gradient are ordinary GEMMs; the
    weight gradient dW = dy^T x contracts over R, a shape (64 x 131 x 884736) a single GEMM launch
    fills the chip poorly with: it is computed split-K as one batched GEMM over row chunks + a sum."""

    # rows from which csrc/rowgemm.hip takes the forward / input-gradient GEMM (A/B of the DetMatch step:
    # 16 k rows 107.0 ms, 256 k rows 107.8 ms, BLAS only 109.5 ms; below ~16 k rows a launch is latency sized)
    ROWGEMM_MIN_ROWS = 16384       # (class attribute: the equality tests and tools/bench_tall_skinny.py lower it)

    @staticmethod
    def _rowgemm(x, w, col0=0):
        """x (R, K) . w (N, K)^T on csrc/rowgemm.hip (weights resident in LDS), or None if not taken.
        col0 > 0: the result has col0 leading zero columns (rows of col0 + N floats)."""
        if not (x.is_cuda and x.dtype == torch.float32 and w.dtype == torch.float32
                and x.shape[0] >= TallSkinnyLinear.ROWGEMM_MIN_ROWS):
            return None
        L = _lib.lib()
        r, k = x.shape
        n = w.shape[0]
        if not L.dm_rowgemm_supported(k, n) or col0 % 4:
            return None
        x, w = x.contiguous(), w.contiguous()
        y = torch.empty((r, col0 + n), dtype=torch.float32, device=x.device)
        _lib.check(L.dm_rowgemm_strided(_lib.ptr(x), _lib.ptr(w), _lib.ptr(y), r, k, n, col0 + n, col0,
                                        _lib.stream()), 'dm_rowgemm')
        return y

    @staticmethod
    def _rowgemm_dgrad(gy, w, dead):
        """gx = gy @ w with the first `dead` columns written as zeros, read from the stored (out, in) weight
        (csrc/rowgemm.hip: dm_rowgemm_wt) — no `w.t().contiguous()` copy per call; None if not taken."""
        if not (gy.is_cuda and gy.dtype == torch.float32 and w.dtype == torch.float32 and w.dim() == 2
                and w.stride(1) == 1 and gy.shape[0] >= TallSkinnyLinear.ROWGEMM_MIN_ROWS):
            return None
        L = _lib.lib()
        r, k = gy.shape
        n = w.shape[1] - dead
        if not L.dm_rowgemm_supported(k, n) or dead % 4:
            return None
        gy = gy.contiguous()
        gx = torch.empty((r, dead + n), dtype=torch.float32, device=gy.device)
        _lib.check(L.dm_rowgemm_wt(_lib.ptr(gy), w.data_ptr() + 4 * dead, int(w.stride(0)), _lib.ptr(gx), r, k, n,
                                   dead + n, dead, _lib.stream()), 'dm_rowgemm_wt')
        return gx

    # csrc/conv2d.hip's streaming weight-gradient kernel (dm_tall_wgrad): correct and reproducible, but on the shapes of
    # the step it only ties the batched BLAS call (profiles/r04_tall_skinny_wgrad_blas_vs_own.txt: 250 vs 240 us on
    # 884 736 x 132 x 64, 160 vs 98 us on 884 736 x 64 x 64, 24-49 vs 26-33 us on the small ones).  Round 6: the default
    # all the same — the chained path uses it anyway, and no vendor GEMM is left in the process (a Stream-K kernel of the
    # vendor library next to a second one dead-locks the device); False: the split batched BLAS call, in a turn
    OWN_WGRAD = True

    @staticmethod
    def _wgrad(gy, x):
        """dW = gy^T x on csrc/conv2d.hip's streaming kernel (dm_tall_wgrad), or None if not taken."""
        if not (gy.is_cuda and gy.dtype == torch.float32 and x.dtype == torch.float32 and x.shape[0] >= 4096):
            return None
        L = _lib.lib()
        r, k = x.shape
        n = gy.shape[1]
        if not L.dm_tall_wgrad_supported(n, k):
            return None
        gy, x = gy.contiguous(), x.contiguous()
        gw = torch.empty((n, k), dtype=torch.float32, device=x.device)
        ws = _lib.workspace(int(L.dm_tall_wgrad_workspace_bytes(r, n, k)), x.device, 'tall_wgrad')
        _lib.check(L.dm_tall_wgrad(_lib.ptr(gy), _lib.ptr(x), _lib.ptr(gw), r, n, k, 0, _lib.ptr(ws), ws.numel(),
                                   _lib.stream()), 'dm_tall_wgrad')
        return gw

    @staticmethod
    def _rowgemm_stats(x, w):
        """_rowgemm that also reduces the column statistics of its output for the BatchNorm that follows
        (per-workgroup (mean, M2) partials from the output tile it already holds in LDS): -> y with the
        attribute `dm_bn_pre = (partial, counts, parts)`, or None if not taken."""
        if not (x.is_cuda and x.dtype == torch.float32 and w.dtype == torch.float32
                and x.shape[0] >= TallSkinnyLinear.ROWGEMM_MIN_ROWS):
            return None
        L = _lib.lib()
        r, k = x.shape
        n = w.shape[0]
        if not L.dm_rowgemm_supported(k, n) or n > 256:
            return None
        x, w = x.contiguous(), w.contiguous()
        parts = L.dm_rowgemm_parts(r, k, n)
        y = torch.empty((r, n), dtype=torch.float32, device=x.device)
        partial = torch.empty((2, n, parts), dtype=torch.float32, device=x.device)
        counts = torch.empty((parts,), dtype=torch.float32, device=x.device)
        _lib.check(L.dm_rowgemm_stats(_lib.ptr(x), _lib.ptr(w), _lib.ptr(y), r, k, n, _lib.ptr(partial),
                                      _lib.ptr(counts), _lib.stream()), 'dm_rowgemm_stats')
        y.dm_bn_pre = (partial, counts, parts)
        return y

    @staticmethod
    def forward(ctx, x, w, dead_cols=0, bn_stats=False):
        """dead_cols: leading input columns nobody differentiates (the xyz + padding floats of a grouped
        row: QueryGroupRows.backward reads the feature columns only) — their input gradient is written
        as zeros instead of being computed.  bn_stats: a training-mode BatchNorm follows; its column
        statistics ride along on the output (`dm_bn_pre`, read by bn_relu_rows / bn_relu_rows_max)."""
        ctx.save_for_backward(x, w)
        ctx.dead_cols = int(dead_cols)
        y = TallSkinnyLinear._rowgemm_stats(x, w) if bn_stats else None
        if y is None:
            y = TallSkinnyLinear._rowgemm(x, w)
        if y is None:
            with _lib.blas_turn():
                y = x @ w.t()
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gx = None
        if ctx.needs_input_grad[0]:
            d = ctx.dead_cols
            gx = TallSkinnyLinear._rowgemm_dgrad(gy, w, d)
            if gx is None:
                with _lib.blas_turn():
                    gx = gy @ w
        gw = None
        if ctx.needs_input_grad[1] and TallSkinnyLinear.OWN_WGRAD:
            gw = TallSkinnyLinear._wgrad(gy, x)
        if ctx.needs_input_grad[1] and gw is None:
            rows = x.shape[0]
            split = next((s for s in (256, 128, 64, 32, 16, 8) if rows % s == 0 and rows // s >= 2048), 1)
            with _lib.blas_turn():
                if split > 1:
                    gw = torch.bmm(gy.view(split, rows // split, -1).transpose(1, 2),
                                   x.view(split, rows // split, -1)).sum(dim=0)
                else:
                    gw = gy.t() @ x
        return gx, gw, None, None


def query_group_rows(radius, nsample, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, features=None,
                     use_xyz=True, found=None):
    """QueryAndGroup (pointnet2_utils.py:116-156) in row layout -> (M, nsample, [4+]C), idx.
    `found`: (idx, empty) of this query computed elsewhere (ball_query_pair)."""
    idx, empty = found if found is not None else \
        ball_query(radius, nsample, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt)
    assert use_xyz or features is not None, 'Cannot have not features and not use xyz as a feature!'
    rows = QueryGroupRows.apply(xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, features, idx, empty,
                                use_xyz)
    return rows, idx


class FurthestPointSampling(Function):
    """pointnet2_utils.py:158-180"""

    @staticmethod
    def forward(ctx, xyz, npoint):
        xyz = xyz.contiguous()
        _lib.require_device(xyz)
        B, N, _ = xyz.size()
        output = torch.empty((B, npoint), dtype=torch.int32, device=xyz.device)
        temp = torch.full((B, N), 1e10, dtype=torch.float32, device=xyz.device)
        rc = _lib.lib().dm_furthest_point_sampling(B, N, int(npoint), _lib.ptr(xyz), _lib.ptr(temp),
                                                   _lib.ptr(output), _lib.stream())
        _lib.check(rc, 'dm_furthest_point_sampling')
        ctx.mark_non_differentiable(output)
        return output

    @staticmethod
    def backward(ctx, a=None):
        return None, None


furthest_point_sample = FurthestPointSampling.apply


def furthest_point_sample_stack(xyz, batch_cnt_host, npoint):
    """Ragged FPS: xyz (sum N, 3) stacked, batch_cnt_host the per-sample point counts (host
    ints) -> (B, npoint) int32 indices local to each sample.  All samples run concurrently
    (the reference loops samples in Python, voxel_set_abstraction.py:135-151)."""
    import ctypes
    xyz = xyz.contiguous()
    _lib.require_device(xyz)
    B = len(batch_cnt_host)
    offs = [0]
    for c in batch_cnt_host:
        offs.append(offs[-1] + int(c))
    assert offs[-1] == xyz.shape[0]
    output = torch.empty((B, npoint), dtype=torch.int32, device=xyz.device)
    temp = torch.full((xyz.shape[0],), 1e10, dtype=torch.float32, device=xyz.device)
    rc = _lib.lib().dm_furthest_point_sampling_stack(
        B, (ctypes.c_int32 * (B + 1))(*offs), int(npoint), _lib.ptr(xyz), _lib.ptr(temp),
        _lib.ptr(output), _lib.stream())
    _lib.check(rc, 'dm_furthest_point_sampling_stack')
    return output


class _PadXyzColumn(torch.autograd.Function):
    """(Cout, 3 + C, 1, 1) first-layer weight of a shared MLP -> (Cout, 4 + C) with a zero column for the
    padding float that follows xyz in a grouped row.  The padded copy is cached per parameter and
    rewritten (two strided copies) only after the weight changed — in place (`_version`) or through the
    raw pointers of the fused optimizer / EMA kernels (dense_conv.weights_changed) — instead of a
    slice / zeros / cat chain per grouper and pass; the backward is one cat of the two column blocks."""

    _cache = {}

    @staticmethod
    def forward(ctx, weight):
        from . import dense_conv
        w = weight.detach().view(weight.shape[0], weight.shape[1])
        key = (id(weight), w.data_ptr())
        hit = _PadXyzColumn._cache.get(key)
        if hit is not None and hit[0]() is not weight:
            hit = None
        gen = dense_conv._GENERATION[0]
        if hit is None:
            if len(_PadXyzColumn._cache) > 512:
                _PadXyzColumn._cache.clear()
            buf = torch.zeros((w.shape[0], w.shape[1] + 1), dtype=w.dtype, device=w.device)
            hit = [weakref.ref(weight), -1, -1, buf]
            _PadXyzColumn._cache[key] = hit
        if hit[1] != weight._version or dense_conv._stale(hit[2], w.data_ptr()):
            hit[3][:, :3].copy_(w[:, :3])
            hit[3][:, 4:].copy_(w[:, 3:])
        hit[1], hit[2] = weight._version, gen
        ctx.shape = tuple(weight.shape)
        return hit[3].view(hit[3].shape)       # a fresh tensor object per call over the cached storage

    @staticmethod
    def backward(ctx, g):
        return torch.cat([g[:, :3], g[:, 4:]], dim=1).view(ctx.shape)


def _shared_mlp_1x1(mlp, x):
    """nn.Sequential of [Conv2d 1x1, BatchNorm2d, ReLU]* on a (1, C, M, nsample) tensor without a convolution
    call (pointnet2_modules.py:31-40 builds exactly this pattern)."""
    for mod in mlp:
        if isinstance(mod, nn.Conv2d):
            if mod.kernel_size != (1, 1) or mod.stride != (1, 1) or mod.groups != 1:
                raise ValueError('shared MLPs are 1x1 convolutions')
            b, c, m, ns = x.shape
            y = _lib.blas_linear(x.reshape(b, c, m * ns).transpose(1, 2), mod.weight.view(mod.out_channels, c)).transpose(1, 2)
            if mod.bias is not None:
                y = y + mod.bias.view(1, -1, 1)
            x = y.view(b, mod.out_channels, m, ns)
        elif isinstance(mod, nn.BatchNorm2d):
            with torch.backends.cudnn.flags(enabled=False):
                x = mod(x)
        else:
            x = mod(x)
    return x


class StackSAModuleMSG(nn.Module):
    """pointnet2_modules.py:10-92: multi-scale grouping + shared MLP + max over samples."""

    def __init__(self, *, radii: List[float], nsamples: List[int], mlps: List[List[int]],
                 use_xyz: bool = True, pool_method='max_pool'):
        super().__init__()
        assert len(radii) == len(nsamples) == len(mlps)
        self.groupers = nn.ModuleList()
        self.mlps = nn.ModuleList()
        for i in range(len(radii)):
            self.groupers.append(QueryAndGroup(radii[i], nsamples[i], use_xyz=use_xyz))
            mlp_spec = mlps[i]
            if use_xyz:
                mlp_spec[0] += 3
            shared_mlps = []
            for k in range(len(mlp_spec) - 1):
                shared_mlps.extend([nn.Conv2d(mlp_spec[k], mlp_spec[k + 1], kernel_size=1,
                                              bias=False),
                                    nn.BatchNorm2d(mlp_spec[k + 1]), nn.ReLU()])
            self.mlps.append(nn.Sequential(*shared_mlps))
        self.pool_method = pool_method
        self.row_layout = True      # False: the reference's (1, C, M, nsample) Conv2d formulation
        self.init_weights()

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            if isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1.0)
                nn.init.constant_(m.bias, 0)

    def forward(self, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, features=None,
                empty_voxel_set_zeros=True):
        new_features_list = []
        if self.row_layout and self.pool_method == 'max_pool' and features is not None and xyz.is_cuda:
            # the whole call as one chained launch table behind one autograd node (sa_chain.py), when it applies
            from . import chain as _chain
            train = self.mlps[0][1].training
            if _chain.on('sa', train) and fused_on():
                c_real = features.shape[1]
                c_pad = (c_real + 3) // 4 * 4
                fgrad = bool(features.requires_grad and torch.is_grad_enabled())
                if c_pad != c_real and fgrad:
                    c_pad = -1                         # padded features are plain values only (raw points)
                key = (xyz_batch_cnt.numel(), new_xyz.shape[0], c_real, train, fgrad or c_pad == c_real)
                cache = self.__dict__.setdefault('_chains', {})
                ch = cache.get(key)
                if ch is None or (ch is not False and not ch.valid()):
                    from .sa_chain import SAChain
                    if len(cache) > 8:
                        cache.clear()
                    ch = cache[key] = SAChain(self, key[0], key[1], c_pad, xyz.device, train, c_real=c_real, feat_grad=key[4]) \
                        if c_pad > 0 and SAChain.applicable(self, key[1], c_pad, train, c_real) else False
                if ch is not False and (train or not (torch.is_grad_enabled() and (
                        features.requires_grad or any(p.requires_grad for p in self.parameters())))):
                    if c_pad != c_real:                # zero columns behind the real ones (their weights are zero too)
                        features = F.pad(features, (0, c_pad - c_real))
                    return new_xyz, ch(xyz, _i32(xyz_batch_cnt).contiguous(), new_xyz, _i32(new_xyz_batch_cnt).contiguous(),
                                       features)
        if self.row_layout and self.pool_method == 'max_pool':
            # Row layout: a grouped reference is one contiguous row, the shared 1x1-conv MLP a GEMM
            # over (M*nsample, C) rows, BatchNorm2d a column reduction over the same M*nsample
            # elements per channel — the math of :72-83 without the (1, C, M, nsample) copies.
            found = [None] * len(self.groupers)
            if len(self.groupers) == 2 and fused_on() and xyz.is_cuda:
                # both radii of the source in one scan of its points
                ga, gb = self.groupers
                found = ball_query_pair(ga.radius, ga.nsample, gb.radius, gb.nsample, xyz, xyz_batch_cnt,
                                        new_xyz, new_xyz_batch_cnt)
            for k, g in enumerate(self.groupers):
                rows, _ = query_group_rows(g.radius, g.nsample, xyz, xyz_batch_cnt, new_xyz,
                                           new_xyz_batch_cnt, features, g.use_xyz, found=found[k])
                m, ns, width = rows.shape
                x = rows.view(m * ns, width)
                mods = list(self.mlps[k])
                n_layers = len(mods) // 3
                for li, (conv, bn) in enumerate(zip(mods[0::3], mods[1::3])):
                    if li == 0 and g.use_xyz:      # zero column for the padding float after xyz
                        w = _PadXyzColumn.apply(conv.weight)
                    else:
                        w = conv.weight.view(conv.out_channels, conv.in_channels)
                    x = TallSkinnyLinear.apply(x, w, 4 if (li == 0 and g.use_xyz) else 0,
                                               bool(bn.training and conv.bias is None and fused_on()))
                    if conv.bias is not None:
                        x = x + conv.bias
                    if li == n_layers - 1 and fused_on():
                        x = bn_relu_rows_max(x, bn, ns)     # ... and the max over nsample in the same pass
                    else:
                        x = bn_relu_rows(x, bn, relu=True)  # fused BatchNorm + ReLU over rows
                new_features_list.append(x if x.shape[0] == m else x.view(m, ns, -1).max(dim=1)[0])      # (M, C)
            return new_xyz, torch.cat(new_features_list, dim=1)
        for k in range(len(self.groupers)):
            new_features, _ = self.groupers[k](xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt,
                                               features)  # (M, C, nsample)
            new_features = new_features.permute(1, 0, 2).unsqueeze(dim=0)  # (1, C, M, nsample)
            # Never MIOpen (it may compile a kernel at run time, which forks a compiler from a process that has
            # initialised the GPU): the 1x1 convolutions are matrix products (their backward is a matrix
            # product too — a `cudnn.flags` context around the forward would not cover convolution_backward,
            # which picks its backend when it runs), BatchNorm records its backend at forward time.
            new_features = _shared_mlp_1x1(self.mlps[k], new_features)
            if self.pool_method == 'max_pool':
                new_features = F.max_pool2d(new_features,
                                            kernel_size=[1, new_features.size(3)]).squeeze(dim=-1)
            elif self.pool_method == 'avg_pool':
                new_features = F.avg_pool2d(new_features,
                                            kernel_size=[1, new_features.size(3)]).squeeze(dim=-1)
            else:
                raise NotImplementedError
            new_features_list.append(new_features.squeeze(dim=0).permute(1, 0))  # (M, C)
        return new_xyz, torch.cat(new_features_list, dim=1)
