"""Run one backbone layer's forward gather-GEMM N times (for rocprofv3 --pmc runs)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from detmatch_amd import synth, voxel  # noqa: E402
from detmatch_amd.pcdet.workload import BACKBONE_LAYERS  # noqa: E402
from detmatch_amd.spconv import ops  # noqa: E402

key_want = sys.argv[1] if len(sys.argv) > 1 else 'subm3'
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device('cuda:0')
pts = [torch.from_numpy(synth.lidar_frame(s)['points']).to(dev) for s in range(2)]
_, coors, _, mean, _ = voxel.voxelize_batch(pts, synth.KITTI_VOXEL, synth.KITTI_RANGE, 5, 16000)
idx, shape = coors, [41, 1600, 1408]
done = set()
for key, subm, cin, cout, ks, st, pd in BACKBONE_LAYERS:
    if key in done:
        continue
    done.add(key)
    rb = ops.build_rulebook(idx, 2, shape, ks, st, pd, 1, subm)
    if key == key_want:
        x = torch.randn(rb.n_in, cin, device=dev)
        w = torch.randn(*ks, cin, cout, device=dev) * 0.05
        rb.indice_pairs.dm_tables = (rb.nbr_out, rb.nbr_in, rb.subm)
        for _ in range(reps):
            ops.indice_conv(x, w, rb.indice_pairs, rb.indice_num, rb.n_out, False, subm)
        torch.cuda.synchronize()
        break
    idx, shape = rb.outids, rb.out_shape
