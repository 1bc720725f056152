"""One StackSAModuleMSG call as a chained forward / backward pair (chain.py): both ball queries in one scan, the
grouped rows, the shared-MLP GEMMs with their BatchNorm statistics, BatchNorm + ReLU (+ max over nsample) and the
concatenation of the groupers' pooled features — pcdet/ops/pointnet2/pointnet2_stack/pointnet2_modules.py:58-104 — behind
ONE autograd node instead of ~12 (`pointnet2_stack.StackSAModuleMSG.forward` issues the same kernels one Function at
a time).  Used by the five sources of VoxelSetAbstraction and by the RoI head's grid pooling.

Shapes are static except the number of source rows N (voxels / points of the batch): N only appears as an argument
of the grouping kernels and in the size of the feature gradient, both handed in through the slot table per call.
The backward takes the weight gradients of the tall-skinny GEMMs from the library's own streaming kernel
(`dm_tall_wgrad`) — the op-by-op path's default is a batched BLAS call, so weight gradients agree with it to fp32
rounding (bit for bit with `TallSkinnyLinear.OWN_WGRAD = True`); everything else is bit-identical.
"""
import torch
import torch.nn as nn

from . import _lib, bn_relu, dense_conv
from .chain import Program, S
from .dense_chain import _Weights, _watch_loads


def _bn_ok(bn, c):
    return (bn.track_running_stats and bn.affine and bn.momentum is not None and c % 4 == 0 and 4 <= c <= 1024
            and 256 % (c // 4) == 0)


class SAChain(object):

    @staticmethod
    def applicable(module, m, c_feat, train, c_real=None):
        """The configurations the chain is built for (anything else keeps the op-by-op path).  c_real < c_feat: the
        features were zero-padded to c_feat columns (raw points: one intensity column -> four)."""
        L = _lib.lib()
        c_real = c_feat if c_real is None else c_real
        if len(module.groupers) != 2 or module.pool_method != 'max_pool' or not module.row_layout or c_feat % 4:
            return False
        for g, mlp in zip(module.groupers, module.mlps):
            if not g.use_xyz or not 1 <= g.nsample <= 255 or m * g.nsample < 16384:
                return False
            mods = list(mlp)
            k = 4 + c_feat
            if not mods or not isinstance(mods[0], nn.Conv2d) or mods[0].in_channels != 3 + c_real:
                return False
            for conv, bn, act in zip(mods[0::3], mods[1::3], mods[2::3]):
                if not isinstance(conv, nn.Conv2d) or not isinstance(bn, nn.BatchNorm2d) or not isinstance(act, nn.ReLU):
                    return False
                n = conv.out_channels
                if conv.bias is not None or conv.kernel_size != (1, 1) or not L.dm_rowgemm_supported(k, n) or n > 64 \
                        or not _bn_ok(bn, n) or bn.training != train or not L.dm_tall_wgrad_supported(n, k):
                    return False
                k = n
        return True

    def __init__(self, module, batch, m, c_feat, device, train, name='sa', c_real=None, feat_grad=True):
        L = _lib.lib()
        self.module, self.train, self.device = module, train, device
        self.batch, self.m, self.c = batch, m, c_feat
        self.c_real = c_feat if c_real is None else c_real
        self.feat_grad = feat_grad      # False: nobody differentiates the features (raw points): no scatter of row gradients
        self.weights = W = _Weights(device)
        fwd = self.fwd = Program(name + '.fwd')
        fa = fwd.layout('arena')
        fws = fwd.slot('ws')
        xyz, xyz_cnt, new_xyz, new_cnt, feat, n_src = (fwd.slot(k) for k in
                                                        ('xyz', 'xyz_cnt', 'new_xyz', 'new_cnt', 'feat', 'n'))
        ST = Program.STREAM
        ga, gb = module.groupers
        idx = [fa.take(m * g.nsample * 4) for g in module.groupers]
        emp = [fa.take(m) for _ in module.groupers]
        fwd.call('dm_ball_query_stack2', batch, m, float(ga.radius), ga.nsample, float(gb.radius), gb.nsample, new_xyz,
                 new_cnt, xyz, xyz_cnt, idx[0], idx[1], emp[0], emp[1], ST)
        self.c_out = sum(list(mlp)[-3].out_channels for mlp in module.mlps)
        self.out_ref = fa.floats(m, self.c_out)
        self.bn_modules = []
        self.tape = []
        width = 4 + c_feat
        col = 0
        for gi, (g, mlp) in enumerate(zip(module.groupers, module.mlps)):
            ns = g.nsample
            rows = m * ns
            x = fa.floats(rows, width)
            fwd.call('dm_query_group_rows', batch, m, c_feat, ns, 1, xyz, new_xyz, feat, xyz_cnt, new_cnt, idx[gi],
                     emp[gi], x, ST)
            mods = list(mlp)
            n_layers = len(mods) // 3
            k = width
            layers = []
            for li, (conv, bn) in enumerate(zip(mods[0::3], mods[1::3])):
                n = conv.out_channels
                W.watch.append(conv.weight)
                if li == 0:       # zero column for the padding float after xyz (pointnet2_stack._PadXyzColumn)
                    w = torch.zeros((n, k), dtype=torch.float32, device=device)      # (+ zero columns of padded features)
                    ks = 3 + self.c_real
                    src = conv.weight.detach().view(n, ks)
                    W.calls.append(('dm_copy2d_f32', src, ks, w, k, n, 3, ST))
                    W.calls.append(('dm_copy2d_f32', src.data_ptr() + 12, ks, w.data_ptr() + 16, k, n, self.c_real, ST))
                    W.keep += [src, w]
                else:
                    w = conv.weight.detach().view(n, k)
                    W.keep.append(w)
                y = fa.floats(rows, n)
                last = li == n_layers - 1
                rec = dict(conv=conv, bn=bn, x=x, y=y, w=w, k=k, n=n, rows=rows, ns=ns, last=last, first=li == 0, col=col,
                           idx=idx[gi], emp=emp[gi])
                if train:
                    parts = int(L.dm_rowgemm_parts(rows, k, n))
                    partial, counts = fa.floats(2, n, parts), fa.floats(parts)
                    mean, invstd = fa.floats(n), fa.floats(n)
                    fwd.call('dm_rowgemm_stats', x, w, y, rows, k, n, partial, counts, ST)
                    g_, b_ = bn.weight.detach(), bn.bias.detach()
                    if last:
                        pooled, arg = fa.floats(m, n), fa.take(m * n)
                        fwd.call('dm_bn_rows_max_forward_pre', y, m, ns, n, g_, b_, float(bn.eps), float(bn.momentum),
                                 bn.running_mean, bn.running_var, pooled, arg, mean, invstd, partial, counts, parts, ST)
                        rec.update(arg=arg)
                    else:
                        out = fa.floats(rows, n)
                        fwd.call('dm_bn_rows_forward_pre', y, rows, n, g_, b_, float(bn.eps), float(bn.momentum),
                                 bn.running_mean, bn.running_var, 1, out, mean, invstd, partial, counts, parts, ST)
                    rec.update(mean=mean, invstd=invstd)
                    self.bn_modules.append(bn)
                else:
                    fwd.call('dm_rowgemm', x, w, y, rows, k, n, ST)
                    g_, b_ = bn.weight.detach(), bn.bias.detach()
                    if last:
                        pooled = fa.floats(m, n)
                        fwd.call('dm_bn_rows_eval_max', y, m, ns, n, g_, b_, bn.running_mean, bn.running_var,
                                 float(bn.eps), pooled, ST)
                    else:
                        out = fa.floats(rows, n)
                        fwd.call('dm_bn_rows_eval', y, rows, n, g_, b_, bn.running_mean, bn.running_var, float(bn.eps),
                                 1, out, ST)
                if last:
                    fwd.call('dm_copy2d_f32', pooled, n, self.out_ref + col * 4, self.c_out, m, n, ST)
                    col += n
                else:
                    x = out
                layers.append(rec)
                k = n
            self.tape.append(layers)
        self.fwd_bytes = fa.size
        fwd.finalize()
        self.bwd = None
        self.bwd_d = self.bwd_w = None
        if train:
            self._build_backward(width)
        W.finalize()
        _watch_loads([module])
        self.first_ptr = W.watch[0].data_ptr()

    def _build_backward(self, width):
        L = _lib.lib()
        m, c_feat, batch = self.m, self.c, self.batch
        bwd = self.bwd = Program(self.fwd.name[:-4] + '.bwd')
        fa_slot = bwd.slot('fwd_arena').slot
        ba = bwd.layout('bwd_arena')
        bws = bwd.slot('ws')
        xyz_cnt, new_cnt, n_src, nc_src, gout, gfeat, gfeat2 = (bwd.slot(k) for k in
                                                                 ('xyz_cnt', 'new_cnt', 'n', 'n*c', 'gout', 'gfeat', 'gfeat2'))
        ST = Program.STREAM
        F = lambda ref: S(fa_slot, ref.off)
        self.params, self.param_refs = [], []

        def ws(nbytes):
            bwd.need_workspace(nbytes)
            return bws, int(nbytes)

        for gi, layers in enumerate(self.tape):
            g = None
            for rec in reversed(layers):
                bn, conv = rec['bn'], rec['conv']
                rows, n, k, ns = rec['rows'], rec['n'], rec['k'], rec['ns']
                gy = ba.floats(rows, n)
                gg, gb = ba.floats(n), ba.floats(n)
                w_, wb = ws(L.dm_bn_rows_workspace_bytes(rows, n))
                g_, b_ = bn.weight.detach(), bn.bias.detach()
                if rec['last']:
                    bwd.call('dm_bn_rows_max_backward_ld', gout + rec['col'] * 4, self.c_out, F(rec['arg']), F(rec['y']), m,
                             ns, n, g_, b_, F(rec['mean']), F(rec['invstd']), gy, gg, gb, w_, wb, ST)
                else:
                    bwd.call('dm_bn_rows_backward', g, F(rec['y']), rows, n, g_, b_, F(rec['mean']), F(rec['invstd']), 1, gy,
                             gg, gb, w_, wb, ST)
                self.params += [bn.weight, bn.bias]
                self.param_refs += [(gg, n), (gb, n)]
                # weight gradient of the GEMM (padded for the first layer, unpadded into the parameter's gradient)
                gw = ba.floats(n, k)
                w_, wb = ws(L.dm_tall_wgrad_workspace_bytes(rows, n, k))
                bwd.call('dm_tall_wgrad', gy, F(rec['x']), gw, rows, n, k, 0, w_, wb, ST)
                if rec['first']:
                    ks = 3 + self.c_real
                    gwu = ba.floats(n, ks)
                    bwd.call('dm_copy2d_f32', gw, k, gwu, ks, n, 3, ST)
                    bwd.call('dm_copy2d_f32', gw + 16, k, gwu + 12, ks, n, self.c_real, ST)
                    gw = gwu
                    self.param_refs.append((gw, n * ks))
                else:
                    self.param_refs.append((gw, n * k))
                self.params.append(conv.weight)
                # input gradient (of the first layer only when somebody differentiates the features)
                if rec['first'] and not self.feat_grad:
                    continue
                dead = 4 if rec['first'] else 0
                g = ba.floats(rows, k)
                bwd.call('dm_rowgemm_wt', gy, rec['w'].data_ptr() + 4 * dead, k, g, rows, n, k - dead, k, dead, ST)
            if self.feat_grad:
                rec0 = layers[0]
                bwd.call('dm_group_rows_grad', batch, m, c_feat, n_src, rec0['ns'], width, 4, g, F(rec0['idx']), new_cnt,
                         xyz_cnt, F(rec0['emp']), gfeat if gi == 0 else gfeat2, ST)
        if self.feat_grad:
            bwd.call('dm_add_mask_f32', gfeat, gfeat2, None, gfeat, nc_src, ST)
        self.bwd_bytes = ba.size
        # weight gradients (+ the two copies that unpad the first layer's) as a second half: chain.SIDE_WGRAD
        from .chain import split_program
        self.bwd_d, self.bwd_w = split_program(bwd, ('dm_tall_wgrad', 'dm_copy2d_f32'))
        self.bws_index = bws.slot - 1
        bwd.finalize()

    def valid(self):
        return self.weights.watch[0].data_ptr() == self.first_ptr

    # ---- execution --------------------------------------------------------------------------------------------------
    def forward_raw(self, xyz, xyz_cnt, new_xyz, new_cnt, feat):
        self.weights.refresh()
        arena = torch.empty(self.fwd_bytes, dtype=torch.uint8, device=self.device)
        ws = _lib.workspace(self.fwd.ws_bytes, self.device, 'chain') if self.fwd.ws_bytes else None
        self.fwd.run([arena.data_ptr(), 0 if ws is None else ws.data_ptr(), xyz.data_ptr(), xyz_cnt.data_ptr(),
                      new_xyz.data_ptr(), new_cnt.data_ptr(), feat.data_ptr(), feat.shape[0]])
        for bn in self.bn_modules:
            if bn.num_batches_tracked is not None:
                bn_relu._bump(bn)
        off = self.out_ref.off // 4
        out = arena.view(torch.float32)[off:off + self.m * self.c_out].view(self.m, self.c_out)
        return out, arena

    def backward_raw(self, arena, xyz_cnt, new_cnt, n_src, gout):
        garena = torch.empty(self.bwd_bytes, dtype=torch.uint8, device=self.device)
        ws = _lib.workspace(self.bwd.ws_bytes, self.device, 'chain') if self.bwd.ws_bytes else None
        gfeat = torch.empty((2, n_src if self.feat_grad else 1, self.c), dtype=torch.float32, device=self.device)
        gout = gout.contiguous()
        vals = [arena.data_ptr(), garena.data_ptr(), 0 if ws is None else ws.data_ptr(), xyz_cnt.data_ptr(),
                new_cnt.data_ptr(), n_src, n_src * self.c, gout.data_ptr(), gfeat.data_ptr(),
                gfeat.data_ptr() + 4 * n_src * self.c]
        from . import chain as _chain
        if _chain.SIDE_WGRAD[0] and self.bwd_w is not None and _lib.raw_stream() != _lib.aux_stream(self.device).cuda_stream:
            _chain.run_split(self.bwd_d, self.bwd_w, vals, self.bws_index, self.bwd.ws_bytes, self.device,
                             [arena, garena, gout, xyz_cnt, new_cnt])
        else:
            self.bwd.run(vals)
        f = garena.view(torch.float32)
        grads = [f[ref.off // 4:ref.off // 4 + numel].view(p.shape) for p, (ref, numel) in zip(self.params, self.param_refs)]
        return (gfeat[0] if self.feat_grad else None), grads

    def __call__(self, xyz, xyz_cnt, new_xyz, new_cnt, feat):
        for t in (xyz, xyz_cnt, new_xyz, new_cnt, feat):
            if not t.is_cuda:
                raise _lib.DetMatchHipError('chains run on the MI355X only (got a %s tensor); there is no CPU path' % t.device)
        assert feat.shape[1] == self.c and new_xyz.shape[0] == self.m and xyz_cnt.numel() == self.batch
        if self.bwd is None or not torch.is_grad_enabled():
            return self.forward_raw(xyz.contiguous(), xyz_cnt, new_xyz.contiguous(), new_cnt, feat.detach().contiguous())[0]
        return _SAFn.apply(self, xyz, xyz_cnt, new_xyz, new_cnt, feat, *self.params)


class _SAFn(torch.autograd.Function):

    @staticmethod
    def forward(ctx, chain, xyz, xyz_cnt, new_xyz, new_cnt, feat, *params):
        feat = feat.detach().contiguous()
        out, arena = chain.forward_raw(xyz.contiguous(), xyz_cnt, new_xyz.contiguous(), new_cnt, feat)
        ctx.chain, ctx.arena, ctx.cnts, ctx.n_src = chain, arena, (xyz_cnt, new_cnt), feat.shape[0]
        return out

    @staticmethod
    def backward(ctx, gout):
        gfeat, grads = ctx.chain.backward_raw(ctx.arena, ctx.cnts[0], ctx.cnts[1], ctx.n_src, gout)
        ctx.arena = None
        return (None, None, None, None, None, gfeat if ctx.needs_input_grad[5] else None) + tuple(grads)
