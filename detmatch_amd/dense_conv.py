"""Dense 2-D convolutions on the hand-written fp32-MFMA implicit-GEMM kernels (csrc/conv2d.hip).

Host-side mirror of `torch.nn.functional.conv2d` / `conv_transpose2d` for the shapes of the DetMatch
path (pcdet/models/backbones_2d/base_bev_backbone.py:38-69, anchor_head_single.py:20-37, mmdet
ResNet-50/FPN/RPN at configs/detmatch/001/detmatch/split_0.py:39-99): same argument meaning, same
(N, C, H, W) logical shapes; tensors are kept in `torch.channels_last` memory format (= NHWC rows),
which is what the kernels read and write.  Forward, input gradient and weight gradient all run in
libdetmatch_hip.so; there is no MIOpen / cuDNN call and no CPU path: a CPU tensor raises
`DetMatchHipError`.  (CPU-only tests of the HOST logic around the convolutions install their own
stand-in for host tensors through `HOST_TENSOR_HOOK`; the stand-in lives in tests/_host_conv.py, the
product, bench.py and smoke() never set the hook.)
"""
import bisect
import ctypes
import os
import weakref

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib

# None in the product.  tests/conftest.py may set it to an object with conv2d(...) / conv_transpose2d(...)
# that is handed the arguments of a call on HOST tensors (tests/_host_conv.py).
HOST_TENSOR_HOOK = None

# Bumped by whoever rewrites weights through raw pointers (fused optimizer / EMA kernels): the
# packed-weight cache is keyed on it (ordinary in-place torch updates bump Tensor._version).
_GENERATION = [0]
_PACK_CACHE = {}


_MATH = {'fp32_mfma': 0, 'bf16': 1, 'fp32_split': 2, 'fp32_split_nopatch': 18, 'fp32_split_tapwise_wgrad': 34}


_MATH_CODE = [None]      # the library's arithmetic code, mirrored here (a ctypes query per convolution otherwise)


def set_math(mode):
    """Arithmetic of the convolution GEMMs (forward, input gradient, weight gradient of the >= 128-channel layers):
      'fp32_split'  fp32-class: every operand is split into three bf16 numbers (together its 24 significand
                    bits) and the six significant cross products run on the bf16 matrix instruction with fp32
                    accumulation — MORE accurate against float64 than 'fp32_mfma' (2.0e-7 vs 6.0e-7 rms,
                    tools/probe_bf16_split.py) at 2.7x less matrix-pipe time;
      'fp32_mfma'   the matrix pipe's own fp32 instruction (v_mfma_f32_32x32x2_f32);
      'bf16'        mixed precision: operands ROUNDED to bf16, fp32 accumulate / storage — the counterpart of
                    the reference's fp16 autocast configs (BASELINE configs[4]); see detmatch_amd/precision.py.
    'fp32' selects the default fp32 flavour (FP32_DEFAULT)."""
    if mode == 'fp32':
        mode = FP32_DEFAULT
    if mode not in _MATH:
        raise ValueError("math mode must be one of 'fp32', %s" % sorted(_MATH))
    _lib.check(_lib.lib().dm_dconv_set_math(_MATH[mode]), 'dm_dconv_set_math')
    _MATH_CODE[0] = _lib.lib().dm_dconv_get_math()


def get_math_code():
    """The library's arithmetic code; chains are built per code."""
    if _MATH_CODE[0] is None:
        _MATH_CODE[0] = _lib.lib().dm_dconv_get_math()
    return _MATH_CODE[0]


def get_math():
    code = _lib.lib().dm_dconv_get_math()
    return [k for k, v in _MATH.items() if v == code][0]      # ('fp32_split_nopatch' reads back as 'fp32_split')


# which kernels serve exact-class fp32 (environment DM_FP32_CONV=fp32_mfma|fp32_split for A/B runs): the split
# kernels — more accurate against float64 AND faster (107 -> 96 ms per DetMatch iteration)
FP32_DEFAULT = os.environ.get('DM_FP32_CONV', 'fp32_split')
if FP32_DEFAULT not in _MATH:
    raise ValueError('DM_FP32_CONV must be one of %s' % sorted(_MATH))
# The library starts in mode 0 (fp32_mfma); the default flavour is applied by a post-load hook, so importing
# this module never loads (or needs) the .so and the mode is set whenever the library is first loaded.
def _apply_default(l):
    _lib.check(l.dm_dconv_set_math(_MATH[FP32_DEFAULT]), 'dm_dconv_set_math')
    _MATH_CODE[0] = l.dm_dconv_get_math()


_lib.on_load(_apply_default)


_EVENTS = []        # (generation after the event, ptr_lo, ptr_hi) of raw-pointer rewrites; None = everything


def weights_changed(ptr_lo=None, ptr_hi=None):
    """Somebody rewrote weights through raw pointers (fused optimizer / EMA kernels).  With a byte range
    [ptr_lo, ptr_hi) only packed copies whose source lies inside become stale — the EMA of the teacher
    then does not force a re-pack of the student's weights in the middle of the iteration."""
    _GENERATION[0] += 1
    _EVENTS.append((_GENERATION[0], ptr_lo, ptr_hi))
    if len(_EVENTS) > 64:
        del _EVENTS[:]
        _EVENTS.append((_GENERATION[0], None, None))
        _FLOOR[0] = _GENERATION[0]
    if len(_PACK_CACHE) > 4096:
        _PACK_CACHE.clear()
        _TABLES.clear()


_FLOOR = [0]        # entries packed before this generation are stale whatever the events say
LOAD_EPOCH = [0]    # bumped by load_state_dict() of a module a chain was built over (dense_chain._watch_loads)


def _stale(entry_gen, src_ptr):
    if entry_gen < _FLOOR[0]:
        return True
    for gen, lo, hi in reversed(_EVENTS):
        if gen <= entry_gen:
            break
        if lo is None or lo <= src_ptr < hi:
            return True
    return False


def _stale_any(entry_gen, sorted_ptrs):
    """_stale over SEVERAL source tensors (ascending addresses): True when any raw-pointer rewrite after
    `entry_gen` covers at least one of them.  A chain's derived weights come from many parameters and buffers —
    frozen ones among them, which the optimizer's byte ranges never cover — so asking about one tensor only is not
    enough (the 2D trunk chain watched a frozen stem weight first and kept stale packed weights)."""
    if entry_gen < _FLOOR[0]:
        return True
    for gen, lo, hi in reversed(_EVENTS):
        if gen <= entry_gen:
            break
        if lo is None:
            return True
        i = bisect.bisect_left(sorted_ptrs, lo)
        if i < len(sorted_ptrs) and sorted_ptrs[i] < hi:
            return True
    return False


def _shorts(v):
    return (ctypes.c_short * len(v))(*[int(x) for x in v])


def _pad4(c):
    return (c + 3) // 4 * 4


def _cl(x):
    """(N,C,H,W) tensor whose memory is dense NHWC."""
    if x.dim() != 4:
        raise ValueError('expected a (N, C, H, W) tensor')
    st = x.stride()
    n, c, h, w = x.shape
    if st == (h * w * c, 1, w * c, c):
        return x
    return x.contiguous(memory_format=torch.channels_last) if (c > 1 and h * w > 1) else \
        x.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)


def _empty_cl(n, c, h, w, like):
    return torch.empty((n, h, w, c), dtype=torch.float32, device=like.device).permute(0, 3, 1, 2)


def _pad_channels(x, c4):
    """Zero-pad the channel dimension of an NHWC tensor to c4 (the RGB stem: 3 -> 4)."""
    n, c, h, w = x.shape
    if c == c4:
        return x
    out = torch.zeros((n, h, w, c4), dtype=x.dtype, device=x.device).permute(0, 3, 1, 2)
    out[:, :c] = x
    return out


class _Packed(object):
    """One cached packed weight: destination buffer, descriptor and what it was packed from."""
    __slots__ = ('ver', 'dst', 'src', 'wref', 'scale_n', 'scale_k', 'desc', 'batchable', 'stream')


# workgroups per table row of dm_dconv_pack_batch (every row gets as many as the LARGEST one needs at 2 048 elements each, up to this)
PACK_BLOCKS_CAP = int(os.environ.get('DM_PACK_BLOCKS_CAP', '1024'))    # 256: 207 us per launch, 1024: 176, 4096: 168 (profiles/r06_ab_step_variants.txt)

# 80-byte rows of dm_dconv_pack_batch (csrc/conv2d.hip: DConvPackDesc)
_DESC = None
_TABLES = {}      # HIP stream -> {'entries': [...], 'table': device tensor or None}


def _desc_dtype():
    global _DESC
    if _DESC is None:
        import numpy as np
        _DESC = np.dtype([('src', '<u8'), ('dst', '<u8'), ('scale_n', '<u8'), ('scale_k', '<u8'),
                          ('sn', '<i8'), ('sk', '<i8'), ('st', '<i8'), ('S', '<i4'), ('N', '<i4'),
                          ('K', '<i4'), ('Nsrc', '<i4'), ('Ksrc', '<i4'), ('pad', '<i4')])
        assert _DESC.itemsize == 80
    return _DESC


def _scale_id(t):
    return None if t is None else (t.data_ptr(), t._version)


def _constant(t):
    """Scale vectors that never change while they live (FrozenBN.scale_shift marks its cached map)."""
    return t is None or getattr(t, 'dm_constant', False)


def _refresh_stream(stream, device):
    """Re-pack the stale batchable cached weights of this stream in ONE launch (after the fused optimizer
    / EMA kernels rewrote them: `weights_changed`); the others just move to the current generation."""
    import numpy as np
    reg = _TABLES[stream]
    live = []
    for e in reg['entries']:
        w = e.wref()
        if w is not None and w.data_ptr() == e.desc[0] and _PACK_CACHE.get(e.desc[-1]) is e:
            live.append(e)
    reg['entries'] = live
    gen = _GENERATION[0]
    dirty = [e for e in live if e.ver[1] != e.wref()._version or _stale(e.ver[0], e.desc[0])]
    if dirty:
        sig = tuple(id(e) for e in dirty)
        tables = reg.setdefault('tables', {})
        hit = tables.get(sig)
        if hit is None:
            if len(tables) > 8:
                tables.clear()
            rows = np.zeros(len(dirty), dtype=_desc_dtype())
            for i, e in enumerate(dirty):
                rows[i] = e.desc[:-1]
            blocks = max(1, min(PACK_BLOCKS_CAP, max(e.dst.numel() for e in dirty) // 2048))
            hit = tables[sig] = (torch.from_numpy(rows.view(np.uint8)).to(device), blocks, list(dirty))
        _lib.check(_lib.lib().dm_dconv_pack_batch(hit[0].data_ptr(), len(dirty), hit[1], _lib.raw_stream()),
                   'dm_dconv_pack_batch')
    for e in live:
        e.ver = (gen, e.wref()._version, e.ver[2], e.ver[3])


PLANES = True     # pre-split weight planes for the patch kernel (module switch of the A/B tools)


def _wants_planes(T, N, K, stride_one):
    """The layers dconv_patch_* takes in the split arithmetic (csrc/conv2d.hip: dm_dconv_gemm_planes)."""
    if _MATH_CODE[0] is None:
        _MATH_CODE[0] = _lib.lib().dm_dconv_get_math()
    return PLANES and stride_one and 4 <= T <= 9 and K % 32 == 0 and N % 4 == 0 and N >= 64 and _MATH_CODE[0] == 2


def _pack(weight, tag, S, N, K, n_src, k_src, sn, sk, st, scale_n=None, scale_k=None, planes=False):
    """[S][N][K] packed copy of `weight` (cached until the weight or a scale changes).  planes=True: the flat
    buffer also holds the pre-split bf16 planes behind the fp32 block (`dst.dm_planes` = their byte offset)."""
    # temporaries may recycle an address; `dm_cacheable`: a long-lived view of a Parameter (_lib.own_linear keeps one per FC weight)
    cacheable = isinstance(weight, nn.Parameter) or getattr(weight, 'dm_cacheable', False)
    stream = _lib.raw_stream()
    if planes:
        tag = tag + 'P'
    key = (id(weight), weight.data_ptr(), tag, N, K, stream)
    ver = (_GENERATION[0], weight._version, _scale_id(scale_n), _scale_id(scale_k))
    hit = _PACK_CACHE.get(key) if cacheable else None
    if hit is not None and hit.wref() is not weight:
        hit = None
    if hit is not None:
        if hit.ver == ver:
            return hit.dst
        if hit.ver[1:] == ver[1:] and hit.desc[0] == weight.data_ptr() and not _stale(hit.ver[0], hit.desc[0]):
            hit.ver = ver              # rewritten weights were somebody else's
            return hit.dst
        if hit.batchable and hit.ver[2:] == ver[2:]:
            _refresh_stream(stream, weight.device)
            if hit.ver == ver:
                return hit.dst
    w = weight.detach()
    if not w.is_contiguous():
        w = w.contiguous()
    if hit is not None:
        dst = hit.dst
    elif planes:
        nb = int(_lib.lib().dm_dconv_planes_bytes(S, N, K))
        dst = torch.empty(S * N * K + (nb + 3) // 4, dtype=torch.float32, device=weight.device)
        dst.dm_planes = S * N * K * 4
    else:
        dst = torch.empty((S, N, K), dtype=torch.float32, device=weight.device)
    _lib.check(_lib.lib().dm_dconv_pack(_lib.ptr(w), _lib.ptr(dst), _lib.ptr(scale_n),
                                        _lib.ptr(scale_k), S, N, K, n_src, k_src, sn, sk, st,
                                        _lib.stream()), 'dm_dconv_pack')
    if planes:
        _lib.check(_lib.lib().dm_dconv_pack_planes(_lib.ptr(w), dst.data_ptr() + dst.dm_planes, _lib.ptr(scale_n),
                                                   _lib.ptr(scale_k), S, N, K, n_src, k_src, sn, sk, st,
                                                   _lib.stream()), 'dm_dconv_pack_planes')
    if cacheable:
        e = _Packed()
        e.ver, e.dst, e.src, e.wref = ver, dst, w, weakref.ref(weight)
        e.scale_n, e.scale_k, e.stream = scale_n, scale_k, stream
        e.batchable = w.data_ptr() == weight.data_ptr() and _constant(scale_n) and _constant(scale_k)
        e.desc = (w.data_ptr(), dst.data_ptr(), 0 if scale_n is None else scale_n.data_ptr(),
                  0 if scale_k is None else scale_k.data_ptr(), sn, sk, st, S, N, K, n_src, k_src,
                  1 if planes else 0, key)
        _PACK_CACHE[key] = e
        if e.batchable:
            reg = _TABLES.setdefault(stream, {'entries': []})
            reg['entries'].append(e)
            reg.pop('tables', None)
    return dst


_PLAN_BY_ID = {}
_FWD_GEOM = {}       # launch geometry of a forward convolution by its shape signature
_PLAN_CACHE = {}     # (kind, geometry, taps) -> (ctypes geometry, ctypes taps, workspace bytes)


def _plan(kind, geom, taps):
    """The marshalled argument arrays of one launch geometry (a network has a few dozen; building
    ctypes arrays per call costs more host time than the launch)."""
    fast = (kind, id(geom), id(taps))       # cached geometry lists (_FWD_GEOM) are looked up by identity
    hit = _PLAN_BY_ID.get(fast)
    if hit is not None and hit[3] is geom and hit[4] is taps:
        return hit[:3]
    key = (kind, tuple(geom), tuple(taps))
    hit = _PLAN_CACHE.get(key)
    if hit is not None and any(geom is v[2] for v in _FWD_GEOM.values()):
        # identity shortcut only for the cached forward geometries (lists that live as long as _FWD_GEOM): the
        # backward / transposed paths build fresh lists per call, which would only churn this dict
        if len(_PLAN_BY_ID) > 8192:
            _PLAN_BY_ID.clear()
        _PLAN_BY_ID[fast] = hit + (geom, taps)
    if hit is None:
        L = _lib.lib()
        g = _lib.ints(geom)
        if kind == 'gemm':
            t = [a for a, _, _ in taps] + [b for _, b, _ in taps] + [c for _, _, c in taps]
            nbytes = L.dm_dconv_gemm_workspace_bytes(g)
        else:
            t = [a for a, _ in taps] + [b for _, b in taps]
            nbytes = L.dm_dconv_wgrad_workspace_bytes(g)
        if len(_PLAN_CACHE) > 4096:
            _PLAN_CACHE.clear()
        hit = _PLAN_CACHE[key] = (g, _shorts(t), int(nbytes))
    return hit


def _gemm(x, wp, bias, y, geom, taps, residual=None):
    """geom: the 17 ints of dm_dconv_gemm; taps: [(dy, dx, slice)]; residual: tensor in y's layout that
    the epilogue adds before the ReLU."""
    g, t, nbytes = _plan('gemm', geom, taps)
    ws = _lib.workspace(nbytes, x.device, 'dconv_gemm') if nbytes else None
    off = getattr(wp, 'dm_planes', None)
    _lib.check(_lib.lib().dm_dconv_gemm_planes(
        x.data_ptr(), wp.data_ptr(), None if off is None else wp.data_ptr() + off,
        None if bias is None else bias.data_ptr(),
        None if residual is None else residual.data_ptr(), y.data_ptr(), g, t,
        None if ws is None else ws.data_ptr(), ws.numel() if ws is not None else 0, _lib.raw_stream()),
        'dm_dconv_gemm')


def _wgrad(U, V, out, scale_u, geom, taps, cv_out, su, sv, st):
    g, t, nbytes = _plan('wgrad', geom, taps)
    ws = _lib.workspace(nbytes, U.device, 'dconv_wgrad')
    _lib.check(_lib.lib().dm_dconv_wgrad(U.data_ptr(), V.data_ptr(), out.data_ptr(),
                                         None if scale_u is None else scale_u.data_ptr(), g, t, cv_out, su, sv,
                                         st, 0, ws.data_ptr(), ws.numel(), _lib.raw_stream()), 'dm_dconv_wgrad')


def _pair(v):
    return (int(v[0]), int(v[1])) if isinstance(v, (tuple, list)) else (int(v), int(v))


def _require(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.DetMatchHipError(
                'detmatch_amd.dense_conv runs on the MI355X only (got a %s tensor); there is no '
                'CPU path' % t.device)


class _Conv2dFn(torch.autograd.Function):
    """y = conv2d(x, weight * w_scale[:, None, None, None], bias, stride, padding) [+ ReLU]."""

    @staticmethod
    def forward(ctx, x, weight, bias, w_scale, stride, padding, relu, residual=None):
        cout, cin, kh, kw = weight.shape
        (sh, sw), (ph, pw) = stride, padding
        x = _pad_channels(_cl(x.detach()), _pad4(cin))
        n, c4, h, w = x.shape
        # geometry of the launch: a network has a few dozen distinct ones, rebuilt lists cost more than the launch
        key = (n, c4, h, w, cout, kh, kw, sh, sw, ph, pw, relu)
        hit = _FWD_GEOM.get(key)
        if hit is None:
            ho, wo = (h + 2 * ph - kh) // sh + 1, (w + 2 * pw - kw) // sw + 1
            taps = [(a - ph, b - pw, a * kw + b) for a in range(kh) for b in range(kw)]
            if len(_FWD_GEOM) > 4096:
                _FWD_GEOM.clear()
            hit = _FWD_GEOM[key] = (ho, wo, [n, h, w, c4, ho, wo, cout, ho, wo, 0, 0, 1, 1, sh, sw, kh * kw, int(relu)], taps)
        ho, wo, geom, taps = hit
        T = kh * kw
        wp = _pack(weight, 'f', T, cout, c4, cout, cin, cin * T, T, 1, scale_n=w_scale,
                   planes=residual is None and _wants_planes(T, cout, c4, sh == 1 and sw == 1 and ho == h and wo == w))
        y = _empty_cl(n, cout, ho, wo, x)
        if residual is not None:
            residual = _cl(residual.detach())
            if tuple(residual.shape) != tuple(y.shape):
                raise _lib.DetMatchHipError('dense_conv: residual %s does not match the output %s'
                                            % (tuple(residual.shape), tuple(y.shape)))
        _gemm(x, wp, None if bias is None else bias.detach(), y, geom, taps, residual)
        ctx.geom = (stride, padding, relu, (n, cin, h, w))
        ctx.has_bias = bias is not None
        ctx.save_for_backward(x, weight, w_scale, y if relu else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, w_scale, y = ctx.saved_tensors
        (sh, sw), (ph, pw), relu, (n, cin, h, w) = ctx.geom
        cout, _, kh, kw = weight.shape
        c4 = x.shape[1]
        T = kh * kw
        if relu:
            dy = torch.ops.aten.threshold_backward(dy, y, 0)     # dy where y > 0 else 0, one launch
        if cout % 4:       # the kernels read rows of the gradient 16 bytes at a time
            raise _lib.DetMatchHipError('dense_conv: Cout must be a multiple of 4 (pad the layer)')
        dy = _cl(dy)
        ho, wo = dy.shape[2], dy.shape[3]
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            wt = _pack(weight, 'b', T, c4, cout, cin, cout, T, cin * T, 1, scale_k=w_scale,
                       planes=_wants_planes(T, c4, cout, sh == 1 and sw == 1 and ho == h and wo == w))
            dxp = _empty_cl(n, c4, h, w, dy)
            if sh == 1 and sw == 1:
                taps = [(ph - a, pw - b, a * kw + b) for a in range(kh) for b in range(kw)]
                _gemm(dy, wt, None, dxp, [n, ho, wo, cout, h, w, c4, h, w, 0, 0, 1, 1, 1, 1, T, 0],
                      taps)
            else:
                classes = []
                for ry in range(sh):
                    for rx in range(sw):
                        taps = [((ry + ph - a) // sh, (rx + pw - b) // sw, a * kw + b)
                                for a in range(kh) for b in range(kw)
                                if (ry + ph - a) % sh == 0 and (rx + pw - b) % sw == 0]
                        lh, lw = (h - ry + sh - 1) // sh, (w - rx + sw - 1) // sw
                        classes.append((ry, rx, lh, lw, taps))
                if any(not c[4] for c in classes):
                    dxp.zero_()
                for ry, rx, lh, lw, taps in classes:
                    if taps and lh > 0 and lw > 0:
                        _gemm(dy, wt, None, dxp, [n, ho, wo, cout, h, w, c4, lh, lw, ry, rx, sh, sw,
                                                  1, 1, len(taps), 0], taps)
            dx = dxp if c4 == cin else dxp[:, :cin]
        if ctx.needs_input_grad[1]:
            dw = torch.empty_like(weight, memory_format=torch.contiguous_format)
            taps = [(a - ph, b - pw) for a in range(kh) for b in range(kw)]
            _wgrad(dy, x, dw, w_scale, [n, ho, wo, cout, c4, h, w, sh, sw, T], taps, cin,
                   cin * T, T, 1)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = dy.sum(dim=(0, 2, 3))
        # the shortcut branch receives the (ReLU-masked) output gradient as it is
        dres = dy if (len(ctx.needs_input_grad) > 7 and ctx.needs_input_grad[7]) else None
        return dx, dw, db, None, None, None, None, dres


class _ConvTranspose2dFn(torch.autograd.Function):
    """ConvTranspose2d with kernel == stride, no padding, no bias (the BEV backbone's deblocks,
    base_bev_backbone.py:52-58): every output pixel has exactly one tap."""

    @staticmethod
    def forward(ctx, x, weight, k):
        cin, cout = weight.shape[0], weight.shape[1]
        x = _cl(x.detach())
        n, _, h, w = x.shape
        T = k * k
        wp = _pack(weight, 'tf', T, cout, cin, cout, cin, T, cout * T, 1)
        y = _empty_cl(n, cout, h * k, w * k, x)
        for a in range(k):
            for b in range(k):
                _gemm(x, wp, None, y, [n, h, w, cin, h * k, w * k, cout, h, w, a, b, k, k, 1, 1, 1, 0],
                      [(0, 0, a * k + b)])
        ctx.k = k
        ctx.save_for_backward(x, weight)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        k = ctx.k
        cin, cout = weight.shape[0], weight.shape[1]
        n, _, h, w = x.shape
        T = k * k
        dy = _cl(dy)
        dx = dw = None
        taps = [(a, b) for a in range(k) for b in range(k)]
        if ctx.needs_input_grad[0]:
            wt = _pack(weight, 'tb', T, cin, cout, cin, cout, cout * T, T, 1)
            dx = _empty_cl(n, cin, h, w, dy)
            _gemm(dy, wt, None, dx, [n, h * k, w * k, cout, h, w, cin, h, w, 0, 0, 1, 1, k, k, T, 0],
                  [(a, b, a * k + b) for a, b in taps])
        if ctx.needs_input_grad[1]:
            dw = torch.empty_like(weight, memory_format=torch.contiguous_format)
            _wgrad(x, dy, dw, None, [n, h, w, cin, cout, h * k, w * k, k, k, T], taps, cout,
                   cout * T, T, 1)
        return dx, dw, None


def conv2d(x, weight, bias=None, stride=1, padding=0, relu=False, w_scale=None, residual=None):
    """F.conv2d (dilation 1, groups 1) [+ residual] [+ fused ReLU]; `w_scale` (Cout) multiplies the
    weight rows (frozen-BatchNorm fold); `residual` (the output's shape) is added in the GEMM's epilogue,
    before the ReLU.  Returns a channels_last tensor."""
    stride, padding = _pair(stride), _pair(padding)
    if not x.is_cuda:
        if HOST_TENSOR_HOOK is not None:
            return HOST_TENSOR_HOOK.conv2d(x, weight, bias, stride, padding, relu, w_scale, residual)
        _require(x)
    _require(weight, bias, w_scale, residual)
    return _Conv2dFn.apply(x, weight, bias, w_scale, stride, padding, bool(relu), residual)


def conv_transpose2d(x, weight, stride):
    """F.conv_transpose2d for kernel_size == stride, no padding / bias."""
    k = int(stride[0] if isinstance(stride, (tuple, list)) else stride)
    if tuple(weight.shape[2:]) != (k, k):
        raise NotImplementedError('conv_transpose2d: only kernel_size == stride is on the path')
    if not x.is_cuda:
        if HOST_TENSOR_HOOK is not None:
            return HOST_TENSOR_HOOK.conv_transpose2d(x, weight, k)
        _require(x)
    _require(weight)
    return _ConvTranspose2dFn.apply(x, weight, k)


class Conv2d(nn.Conv2d):
    """nn.Conv2d (same constructor, parameters and state-dict keys) computed by the HIP kernels."""

    def forward(self, x):
        if self.dilation != (1, 1) or self.groups != 1 or self.padding_mode != 'zeros' or \
                isinstance(self.padding, str):
            raise NotImplementedError('dense_conv.Conv2d: dilation / groups / padding modes are '
                                      'not on the DetMatch path')
        return conv2d(x, self.weight, self.bias, self.stride, self.padding)


class ConvTranspose2d(nn.ConvTranspose2d):
    def forward(self, x):
        if self.bias is not None or self.padding != (0, 0) or self.output_padding != (0, 0) or \
                self.kernel_size != self.stride:
            raise NotImplementedError('dense_conv.ConvTranspose2d: only kernel == stride, no bias')
        return conv_transpose2d(x, self.weight, self.stride)
