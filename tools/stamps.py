"""Diagnostic: per-workgroup in-kernel stamps of spconv_gg for one layer."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from detmatch_amd import synth, voxel, _lib
from detmatch_amd.pcdet.workload import BACKBONE_LAYERS
from detmatch_amd.spconv import ops
key_want = sys.argv[1] if len(sys.argv) > 1 else 'subm3'
dev = torch.device('cuda:0')
_lib.lib().dm_spconv_set_variant(int(sys.argv[2]) if len(sys.argv) > 2 else -1)
pts = [torch.from_numpy(synth.lidar_frame(s)['points']).to(dev) for s in range(2)]
_, coors, _, mean, _ = voxel.voxelize_batch(pts, synth.KITTI_VOXEL, synth.KITTI_RANGE, 5, 16000)
idx, shape = coors, [41, 1600, 1408]
done = set()
for key, subm, cin, cout, ks, st, pd in BACKBONE_LAYERS:
    if key in done: continue
    done.add(key)
    rb = ops.build_rulebook(idx, 2, shape, ks, st, pd, 1, subm)
    if key == key_want:
        x = torch.randn(rb.n_in, cin, device=dev); w = torch.randn(*ks, cin, cout, device=dev) * 0.05
        rb.indice_pairs.dm_tables = (rb.nbr_out, rb.nbr_in, rb.subm)
        for _ in range(5): ops.indice_conv(x, w, rb.indice_pairs, rb.indice_num, rb.n_out, False, subm)
        buf = torch.zeros(4096 * 6, dtype=torch.int64, device=dev)
        _lib.lib().dm_spconv_debug_stamps(_lib.ptr(buf))
        ops.indice_conv(x, w, rb.indice_pairs, rb.indice_num, rb.n_out, False, subm)
        torch.cuda.synchronize()
        _lib.lib().dm_spconv_debug_stamps(None)
        b = buf.cpu().numpy().reshape(-1, 6)
        b = b[b[:, 0] != 0]
        t0 = b[:, 0].min()
        life = b[:, 2] - b[:, 0]; pro = b[:, 1] - b[:, 0]
        rt = (b[:, 4] - b[:, 3]) * 10.0  # ns (100 MHz)
        print('WGs', len(b), 'kernel span cycles', b[:, 2].max() - t0, 'span ns', (b[:, 4].max() - b[:, 3].min()) * 10)
        print('life cycles mean %.0f min %d max %d | prologue mean %.0f | n_active mean %.1f max %d' % (life.mean(), life.min(), life.max(), pro.mean(), b[:, 5].mean(), b[:, 5].max()))
        print('clock GHz ~ %.2f' % (life.sum() / rt.sum()))
        print('start offsets (cycles) pct: ', np.percentile(b[:, 0] - t0, [0, 50, 90, 100]))
        print('cycles per active offset: %.0f' % ((life - pro).sum() / b[:, 5].sum()))
        break
    idx, shape = rb.outids, rb.out_shape
