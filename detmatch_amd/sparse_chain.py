"""VoxelBackBone8x as a chained forward / backward pair (chain.py): the 12 gather-GEMMs with their BatchNorm1d + ReLU
rows (pcdet/models/backbones_3d/spconv_backbone.py:80-175, spconv/conv.py:146-172) behind ONE autograd node, the
weight gradients of all layers in one `dm_spconv_wgrad_batch` call inside the backward table — VERDICT r4 item 1 (iv).

The rulebooks are given (prefetched with the geometry of the pass, or built here): everything data dependent — the
row counts of the five resolution levels, the gather tables, their packed / ordered variants — travels in the slot
table, the op tables are built once.  Same entry points, same arguments, same order as the op-by-op path
(`spconv.conv.SparseConvolution.forward` + `bn_relu.bn_relu_rows`): bit-identical results.
"""
import ctypes

import torch
import torch.nn as nn

from . import _lib, bn_relu, precision
from .chain import Program, S
from .spconv import ops as sp_ops
from .spconv.conv import SparseConvolution
from .spconv.structure import SparseConvTensor


def _layers(backbone):
    """[(conv, bn)] in forward order + the indices after which x_conv1..4 / the output are taken."""
    seqs = [backbone.conv_input, backbone.conv1, backbone.conv2, backbone.conv3, backbone.conv4, backbone.conv_out]
    layers, taps = [], []
    for seq in seqs:
        mods = [m for m in seq.modules() if isinstance(m, (SparseConvolution, nn.BatchNorm1d, nn.ReLU))]
        if len(mods) % 3:
            return None, None
        for conv, bn, act in zip(mods[0::3], mods[1::3], mods[2::3]):
            if not (isinstance(conv, SparseConvolution) and isinstance(bn, nn.BatchNorm1d) and isinstance(act, nn.ReLU)):
                return None, None
            layers.append((conv, bn))
        taps.append(len(layers) - 1)
    return layers, taps[1:]      # taps: x_conv1, x_conv2, x_conv3, x_conv4, out


class SparseBackboneChain(object):

    @staticmethod
    def applicable(backbone, train):
        layers, taps = _layers(backbone)
        if layers is None or backbone.frozen_stages >= 0:
            return False
        for conv, bn in layers:
            c = conv.out_channels
            if conv.bias is not None or conv.conv1x1 or conv.indice_key is None or bn.training != train or \
                    not (bn.track_running_stats and bn.affine and bn.momentum is not None) or \
                    c % 4 or not 4 <= c <= 1024 or 256 % (c // 4) or conv.weight.dtype != torch.float32:
                return False
        return True

    def __init__(self, backbone, device, train):
        L = _lib.lib()
        self.backbone, self.device, self.train = backbone, device, train
        self.layers, self.taps = _layers(backbone)
        self.keys = []
        for conv, _ in self.layers:
            if conv.indice_key not in self.keys:
                self.keys.append(conv.indice_key)
        self.storage = None
        if precision.sparse_bf16():
            self.storage = 0
        elif precision.fp32_flavour() == 'fp32_split':
            self.storage = 3
        self.signature = (precision.sparse_bf16(), precision.fp32_flavour(), sp_ops.PACK_ROWS)
        shape, self.out_shapes = list(backbone.sparse_shape), []
        for conv, _ in self.layers:
            if not conv.subm:
                shape = sp_ops.get_conv_output_size(shape, conv.kernel_size, conv.stride, conv.padding, conv.dilation)
            self.out_shapes.append(list(shape))
        self._bn_ws = {}
        ST = Program.STREAM
        fwd = self.fwd = Program('voxel_backbone.fwd')
        self.f_arena, self.f_ws, self.f_wsb = fwd.slot('arena'), fwd.slot('ws'), fwd.slot('ws_bytes')
        f_in = fwd.slot('voxel_features')
        # per rulebook: rows in / out and the forward gather table (+ tile order, row permutation or 0)
        self.f_key = {k: dict(n_in=fwd.slot(k + '.n_in'), n_out=fwd.slot(k + '.n_out'), tab=fwd.slot(k + '.tab'),
                              order=fwd.slot(k + '.order'), perm=fwd.slot(k + '.perm'),
                              tab_raw=fwd.slot(k + '.tab_raw')) for k in self.keys}
        # per layer buffers: conv rows, BatchNorm rows, mean, invstd — byte offsets into the arena, per call
        self.f_buf = [dict(y=fwd.slot('y%d' % i), z=fwd.slot('z%d' % i), mean=fwd.slot('m%d' % i), inv=fwd.slot('i%d' % i))
                      for i in range(len(self.layers))]
        x = f_in
        self.ws_static = 0
        for i, (conv, bn) in enumerate(self.layers):
            k = self.f_key[conv.indice_key]
            cin, cout, kvol = conv.in_channels, conv.out_channels, int(conv.weight.numel() // (conv.in_channels * conv.out_channels))
            b = self.f_buf[i]
            w = conv.weight.detach()
            fwd.keep.append(w)
            use16 = self.storage is not None and min(cin, cout) >= 16
            packed = cin >= 32      # (and rows >= TILE_ORDER_MIN_ROWS: decided per call — the slots then hold 0)
            tab = k['tab'] if packed else k['tab_raw']
            order, perm = (k['order'], k['perm']) if packed else (None, None)
            if use16:
                wsb = int(L.dm_spconv16_workspace_bytes(kvol, cin, cout))
                fwd.call('dm_spconv_gather_gemm16', x, k['n_in'], w, self.storage, tab, k['n_out'], kvol, cin, cout, 0, 0,
                         b['y'], order, perm, self.f_ws, wsb, ST)
            else:
                wsb = int(L.dm_spconv_workspace_bytes(kvol, cin, cout))
                fwd.call('dm_spconv_gather_gemm', x, k['n_in'], w, tab, k['n_out'], kvol, cin, cout, 0, 0, b['y'], order, perm,
                         self.f_ws, wsb, ST)
            self.ws_static = max(self.ws_static, wsb)
            g_, b_ = bn.weight.detach(), bn.bias.detach()
            if train:
                fwd.call('dm_bn_rows_forward', b['y'], k['n_out'], cout, g_, b_, float(bn.eps), float(bn.momentum),
                         bn.running_mean, bn.running_var, 1, b['z'], b['mean'], b['inv'], self.f_ws, self.f_wsb, ST)
            else:
                fwd.call('dm_bn_rows_eval', b['y'], k['n_out'], cout, g_, b_, bn.running_mean, bn.running_var, float(bn.eps), 1,
                         b['z'], ST)
            x = b['z']
        fwd.finalize()
        self.bwd = None
        if train:
            self._build_backward()
        self.first_ptr = self.layers[0][0].weight.data_ptr()

    def _build_backward(self):
        L = _lib.lib()
        ST = Program.STREAM
        bwd = self.bwd = Program('voxel_backbone.bwd')
        self.b_ws, self.b_wsb = bwd.slot('ws'), bwd.slot('ws_bytes')
        b_in = bwd.slot('voxel_features')
        self.b_key = {k: dict(n_in=bwd.slot(k + '.n_in'), n_out=bwd.slot(k + '.n_out'), tab=bwd.slot(k + '.tab'),
                              order=bwd.slot(k + '.order'), perm=bwd.slot(k + '.perm'), tab_raw=bwd.slot(k + '.tab_raw'))
                      for k in self.keys}
        n = len(self.layers)
        # forward buffers (saved), gradient buffers: gz (grad of BN rows), gy (grad of conv rows), param grads
        self.b_fwd = [dict(y=bwd.slot('y%d' % i), z=bwd.slot('z%d' % i), mean=bwd.slot('m%d' % i), inv=bwd.slot('i%d' % i))
                      for i in range(n)]
        self.b_grad = [dict(gz=bwd.slot('gz%d' % i), gy=bwd.slot('gy%d' % i), gg=bwd.slot('gg%d' % i), gb=bwd.slot('gb%d' % i),
                            gw=bwd.slot('gw%d' % i), nel=bwd.slot('nel%d' % i)) for i in range(n)]
        self.b_gout = [bwd.slot('gout%d' % j) for j in range(len(self.taps))]
        self.jobs = (_lib.SpconvWgradJob * n)()
        self.b_jobws, self.b_jobwsb = bwd.slot('jobs_ws'), bwd.slot('jobs_ws_bytes')
        for i in reversed(range(n)):
            conv, bn = self.layers[i]
            k = self.b_key[conv.indice_key]
            cin, cout = conv.in_channels, conv.out_channels
            kvol = int(conv.weight.numel() // (cin * cout))
            f, g = self.b_fwd[i], self.b_grad[i]
            # gradient arriving at the layer's output rows: from the next layer (gz written by its dgrad) and / or from
            # the consumers of the multi-scale features (an extra output gradient, added in)
            if i in self.taps:
                j = self.taps.index(i)
                if i == n - 1:
                    gz = self.b_gout[j]
                else:
                    bwd.call('dm_add_mask_f32', g['gz'], self.b_gout[j], None, g['gz'], g['nel'], ST)
                    gz = g['gz']
            else:
                gz = g['gz']
            bwd.call('dm_bn_rows_backward', gz, f['y'], k['n_out'], cout, bn.weight.detach(), bn.bias.detach(), f['mean'],
                     f['inv'], 1, g['gy'], g['gg'], g['gb'], self.b_ws, self.b_wsb, ST)
            if i > 0:        # the voxel features themselves need no gradient
                packed = cout >= 32
                tab = k['tab'] if packed else k['tab_raw']
                order, perm = (k['order'], k['perm']) if packed else (None, None)
                w = conv.weight.detach()
                use16 = self.storage is not None and min(cin, cout) >= 16
                flip = 1 if conv.subm else 0
                dst = self.b_grad[i - 1]['gz']
                if use16:
                    wsb = int(L.dm_spconv16_workspace_bytes(kvol, cin, cout))
                    bwd.call('dm_spconv_gather_gemm16', g['gy'], k['n_out'], w, self.storage, tab, k['n_in'], kvol, cin, cout, 1,
                             flip, dst, order, perm, self.b_ws, wsb, ST)
                else:
                    wsb = int(L.dm_spconv_workspace_bytes(kvol, cin, cout))
                    bwd.call('dm_spconv_gather_gemm', g['gy'], k['n_out'], w, tab, k['n_in'], kvol, cin, cout, 1, flip, dst,
                             order, perm, self.b_ws, wsb, ST)
                self.ws_static = max(self.ws_static, wsb)
        bwd.call('dm_spconv_wgrad_batch', self.jobs, n, 0, self.b_jobws, self.b_jobwsb, ST)
        bwd.finalize()

    def valid(self):
        return self.layers[0][0].weight.data_ptr() == self.first_ptr and \
            self.signature == (precision.sparse_bf16(), precision.fp32_flavour(), sp_ops.PACK_ROWS)

    # ---- execution --------------------------------------------------------------------------------------------------
    @staticmethod
    def _table(nbr, n_rows, ci):
        """(table, order, perm, raw) pointers of one gather launch: `spconv.ops._gather_gemm`'s choice."""
        if ci >= 32 and n_rows >= sp_ops.TILE_ORDER_MIN_ROWS:
            if sp_ops.PACK_ROWS:
                t, perm, order = sp_ops.packed_rows(nbr)
                return t.data_ptr(), order.data_ptr(), perm.data_ptr(), nbr.data_ptr()
            order = sp_ops.tile_order(nbr)
            return nbr.data_ptr(), order.data_ptr(), 0, nbr.data_ptr()
        return nbr.data_ptr(), 0, 0, nbr.data_ptr()

    def _bn_workspace(self, vals_key):
        L = _lib.lib()
        need = 256
        for conv, _ in self.layers:
            key = (vals_key[conv.indice_key][1], conv.out_channels)
            b = self._bn_ws.get(key)
            if b is None:
                if len(self._bn_ws) > 4096:
                    self._bn_ws.clear()
                b = self._bn_ws[key] = int(L.dm_bn_rows_workspace_bytes(key[0], key[1]))
            need = max(need, b)
        return need

    def _books(self, x0, batch_size, indice_dict):
        """{key: (outids, in_indices, pairs, num, shape)} for every rulebook of the chain (built when absent)."""
        if not all(k in indice_dict for k in self.keys):
            built = self.backbone.build_rulebooks(x0, batch_size)
            for k, v in built.items():
                indice_dict.setdefault(k, v)
        return indice_dict

    def forward_raw(self, voxel_features, voxel_coords, batch_size, indice_dict):
        L = _lib.lib()
        books = self._books(voxel_coords, batch_size, indice_dict)
        n_layers = len(self.layers)
        vals_key = {}
        max_rows = 1
        ci_of = {}
        for conv, _ in self.layers:
            ci_of[conv.indice_key] = max(ci_of.get(conv.indice_key, 0), conv.in_channels)
        for k in self.keys:
            outids, inids, pairs, num, _ = books[k]
            nbr_out = pairs.dm_tables[0]
            n_in, n_out = int(inids.shape[0]), int(outids.shape[0])
            tab, order, perm, raw = self._table(nbr_out, n_out, ci_of[k])
            vals_key[k] = (n_in, n_out, tab, order, perm, raw)
            max_rows = max(max_rows, n_out)
        # arena layout of this call
        off, offs = 0, []
        for conv, _ in self.layers:
            n_out = vals_key[conv.indice_key][1]
            c = conv.out_channels
            rows_b = (n_out * c * 4 + 255) // 256 * 256
            offs.append((off, off + rows_b, off + 2 * rows_b, off + 2 * rows_b + 256 * ((c * 4 + 255) // 256)))
            off += 2 * rows_b + 2 * 256 * ((c * 4 + 255) // 256)
        arena = torch.empty(max(off, 256), dtype=torch.uint8, device=self.device)
        base = arena.data_ptr()
        wsb = max(self.ws_static, self._bn_workspace(vals_key))
        ws = _lib.workspace(wsb, self.device, 'chain')
        vals = [base, ws.data_ptr(), ws.numel(), voxel_features.data_ptr()]
        for k in self.keys:
            vals += list(vals_key[k])
        for o in offs:
            vals += [base + o[0], base + o[1], base + o[2], base + o[3]]
        self.fwd.run(vals)
        if self.train:
            for _, bn in self.layers:
                if bn.num_batches_tracked is not None:
                    bn_relu._bump(bn)
        f = arena.view(torch.float32)
        outs = []
        for i in self.taps:
            conv = self.layers[i][0]
            n_out = vals_key[conv.indice_key][1]
            o = offs[i][1] // 4
            outs.append(f[o:o + n_out * conv.out_channels].view(n_out, conv.out_channels))
        return outs, (arena, offs, books, vals_key)

    def backward_raw(self, saved, voxel_features, gouts):
        L = _lib.lib()
        arena, offs, books, vals_key = saved
        n = len(self.layers)
        base = arena.data_ptr()
        # tables of the input-gradient launches
        keyvals = {}
        co_of = {}
        for conv, _ in self.layers[1:]:
            co_of[conv.indice_key] = max(co_of.get(conv.indice_key, 0), conv.out_channels)
        max_rows = 1
        for k in self.keys:
            outids, inids, pairs, num, _ = books[k]
            nbr_out, nbr_in, subm = pairs.dm_tables
            n_in, n_out = vals_key[k][0], vals_key[k][1]
            nbr = nbr_out if subm else nbr_in
            if k in co_of:
                tab, order, perm, raw = self._table(nbr, n_in, co_of[k])
            else:
                tab, order, perm, raw = nbr.data_ptr(), 0, 0, nbr.data_ptr()
            keyvals[k] = (n_in, n_out, tab, order, perm, raw)
            max_rows = max(max_rows, n_out, n_in)
        # gradient arena
        goff, go = 0, []
        for conv, _ in self.layers:
            n_out = vals_key[conv.indice_key][1]
            c = conv.out_channels
            rows_b = (n_out * c * 4 + 255) // 256 * 256
            cb = 256 * ((c * 4 + 255) // 256)
            wb = (conv.weight.numel() * 4 + 255) // 256 * 256
            go.append((goff, goff + rows_b, goff + 2 * rows_b, goff + 2 * rows_b + cb, goff + 2 * rows_b + 2 * cb, n_out * c))
            goff += 2 * rows_b + 2 * cb + wb
        garena = torch.empty(goff, dtype=torch.uint8, device=self.device)
        gbase = garena.data_ptr()
        wsb = max(self.ws_static, self._bn_workspace(vals_key))
        ws = _lib.workspace(wsb, self.device, 'chain')
        vals = [ws.data_ptr(), ws.numel(), voxel_features.data_ptr()]
        for k in self.keys:
            vals += list(keyvals[k])
        for o in offs:
            vals += [base + o[0], base + o[1], base + o[2], base + o[3]]
        for o in go:
            vals += [gbase + o[0], gbase + o[1], gbase + o[2], gbase + o[3], gbase + o[4], o[5]]
        keep = []
        for j, i in enumerate(self.taps):
            conv = self.layers[i][0]
            g = gouts[j]
            n_out = vals_key[conv.indice_key][1]
            if g is None:
                g = torch.zeros((n_out, conv.out_channels), dtype=torch.float32, device=self.device)
            g = g.contiguous()
            keep.append(g)
            vals.append(g.data_ptr())
        # the weight-gradient jobs of this pass
        for i, (conv, _) in enumerate(self.layers):
            a = self.jobs[i]
            outids, inids, pairs, num, _ = books[conv.indice_key]
            a.feat = voxel_features.data_ptr() if i == 0 else base + offs[i - 1][1]
            a.out_grad = gbase + go[i][1]
            a.indice_pairs, a.indice_num = pairs.data_ptr(), num.data_ptr()
            a.filt_grad = gbase + go[i][4]
            a.pair_stride, a.kvol = int(pairs.shape[2]), int(pairs.shape[0])
            a.cin, a.cout = conv.in_channels, conv.out_channels
        jwsb = int(L.dm_spconv_wgrad_batch_workspace_bytes(self.jobs, n))
        jws = _lib.workspace(jwsb, self.device, 'chain_wgrad')
        vals += [jws.data_ptr(), jws.numel()]
        self.bwd.run(vals)
        f = garena.view(torch.float32)
        grads = []
        for (conv, bn), o in zip(self.layers, go):
            c = conv.out_channels
            grads.append(f[o[4] // 4:o[4] // 4 + conv.weight.numel()].view(conv.weight.shape))
            grads.append(f[o[2] // 4:o[2] // 4 + c])
            grads.append(f[o[3] // 4:o[3] // 4 + c])
        return grads

    def params(self):
        out = []
        for conv, bn in self.layers:
            out += [conv.weight, bn.weight, bn.bias]
        return out

    def __call__(self, voxel_features, voxel_coords, batch_size, sparse_shape, indice_dict):
        if not voxel_features.is_cuda:
            raise _lib.DetMatchHipError('chains run on the MI355X only (got a %s tensor); there is no CPU path'
                                        % voxel_features.device)
        vf = voxel_features.detach().contiguous()
        if self.bwd is None or not torch.is_grad_enabled():
            outs, saved = self.forward_raw(vf, voxel_coords, batch_size, indice_dict)
        else:
            outs = _SparseFn.apply(self, vf, voxel_coords, batch_size, indice_dict, *self.params())
            saved = None
        books = indice_dict
        tensors, shape = [], list(sparse_shape)
        for j, i in enumerate(self.taps):
            conv = self.layers[i][0]
            outids, _, _, _, _ = books[conv.indice_key]
            # spatial shape of the level: the rulebook cache holds the INPUT shape of each key
            t = SparseConvTensor(outs[j], outids, self.out_shapes[i], batch_size)
            t.indice_dict = indice_dict
            tensors.append(t)
        return tensors


class _SparseFn(torch.autograd.Function):

    @staticmethod
    def forward(ctx, chain, vf, coords, batch_size, indice_dict, *params):
        outs, saved = chain.forward_raw(vf, coords, batch_size, indice_dict)
        ctx.chain, ctx.saved, ctx.vf = chain, saved, vf
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gouts):
        grads = ctx.chain.backward_raw(ctx.saved, ctx.vf, gouts)
        ctx.saved = None
        return (None, None, None, None, None) + tuple(grads)
