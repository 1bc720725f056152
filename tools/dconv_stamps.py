"""Where a K-tile of dconv_gemm_kernel<128,128> spends its cycles: s_memtime stamps around the
phases of the steady-state loop (debug build tools/_alt/libdconv_stamps.so, -DDCONV_STAMPS).
    python tools/dconv_stamps.py
"""
import ctypes
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), '_alt', 'libdconv_stamps.so'))
vp = ctypes.c_void_p
lib.dm_dconv_gemm.argtypes = [vp, vp, vp, vp, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_short), vp, ctypes.c_size_t, vp]
dev = torch.device('cuda:0')
B, C, H, W, N = 2, 128, 200, 176, 128
x = torch.randn(B, H, W, C, device=dev)
wp = torch.randn(9, N, C, device=dev) * 0.05
y = torch.empty(B, H, W, N, device=dev)
nw = 8 * 4 * 1024
dbg = torch.zeros(nw * 8, dtype=torch.int64, device=dev)
geom = (ctypes.c_int * 17)(B, H, W, C, H, W, N, H, W, 0, 0, 1, 1, 1, 1, 9, 2)
taps = [a - 1 for a in range(3) for b in range(3)] + [b - 1 for a in range(3) for b in range(3)] + list(range(9))
taps = (ctypes.c_short * 27)(*taps)
for _ in range(3):
    rc = lib.dm_dconv_gemm(vp(x.data_ptr()), vp(wp.data_ptr()), vp(dbg.data_ptr()), vp(y.data_ptr()), geom, taps, None, 0, None)
torch.cuda.synchronize()
assert rc == 0
d = dbg.view(-1, 8).cpu().double()
d = d[d.sum(1) > 0]
print('entry -> loop start: %.1f us (max %.1f)   loop end -> last store issued+drained: %.1f us (max %.1f)' % (d[:, 6].mean() / 100, d[:, 6].max() / 100, d[:, 7].mean() / 100, d[:, 7].max() / 100))
import numpy as np
tot_w = (d[:, 6] + d[:, 0] + d[:, 7]).numpy() / 100
lp = d[:, 0].numpy() / 100
print('per-wave entry->end (without the last tile): mean %.1f  p50 %.1f  p90 %.1f  p99 %.1f  max %.1f us' % (tot_w.mean(), np.percentile(tot_w, 50), np.percentile(tot_w, 90), np.percentile(tot_w, 99), tot_w.max()))
print('loop only: mean %.1f  p50 %.1f  p90 %.1f  p99 %.1f  max %.1f  min %.1f us' % (lp.mean(), np.percentile(lp, 50), np.percentile(lp, 90), np.percentile(lp, 99), lp.max(), lp.min()))
d = d[:, :6]
print('waves with stamps:', len(d), ' K-tiles per wave: 36 (35 in the loop)')
names = ['prologue (first tile staged)', 'frag reads kb0,kb1 + issue of next loads', '48 MFMAs (+frag kb2,kb3)',
         'mask + LDS stores (waits for the loads)', '16 MFMAs', 'barrier']
real = d[:, 0].mean()
d[:, 0] = 0
tot = d.sum(1).mean()
print('whole loop: %.1f us (s_memrealtime), %.0f s_memtime ticks -> %.2f ticks/ns' % (real / 100.0, tot, tot / (real * 10.0)))
for i, n in enumerate(names):
    per = d[:, i].mean() / (35 if i else 1)
    print('%-45s %9.0f cycles%s   (%.1f%% of the wave)' % (n, per, '/tile' if i else '     ', 100 * d[:, i].mean() / tot))
print('per K-tile total: %.0f cycles; MFMA floor with the pipe shared by 2 waves: 8192, alone: 4096' % (d[:, 1:].sum(1).mean() / 35))
blk = (dbg.view(-1, 8)[:, 0].cpu().double().numpy() / 100).reshape(-1, 4)[:512]
print('blocks with loop < 100 us:', int((blk.max(1) < 100).sum()), 'of', len(blk))
print('loop us by block id (every 8th block, i.e. one XCD):', np.round(blk[::8, 0][:64], 0).tolist())
h, e = np.histogram(blk.max(1), bins=12)
print('histogram of per-block loop time:', list(zip(np.round(e[:-1]).tolist(), h.tolist())))
