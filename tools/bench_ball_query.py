"""Ball-query launch time at the shapes of the PV-RCNN set abstraction (4096 key points against the
raw points / the four sparse levels of a 2-sample batch) and of RoI-grid pooling.
    python tools/bench_ball_query.py
"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from detmatch_amd import pointnet2_stack as pn  # noqa: E402


def main():
    dev = torch.device('cuda:0')
    rng = np.random.default_rng(0)
    cases = [('raw_points', 2048, 17000, 0.4, 16), ('raw_points', 2048, 17000, 0.8, 16),
             ('conv2', 2048, 30000, 0.8, 16), ('conv2', 2048, 30000, 1.2, 32),
             ('conv4', 2048, 5000, 2.4, 16), ('conv4', 2048, 5000, 4.8, 32),
             ('roi_grid', 27648, 2048, 0.8, 16), ('roi_grid', 27648, 2048, 1.6, 16)]
    out = []
    for name, m, n, radius, ns in cases:
        lo, hi = np.array([0, -40, -3]), np.array([70.4, 40, 1])
        xyz = torch.from_numpy(rng.uniform(lo, hi, (2 * n, 3)).astype(np.float32)).to(dev)
        new = torch.from_numpy(rng.uniform(lo, hi, (2 * m, 3)).astype(np.float32)).to(dev)
        cnt = torch.tensor([n, n], dtype=torch.int32, device=dev)
        ncnt = torch.tensor([m, m], dtype=torch.int32, device=dev)
        for _ in range(3):
            pn.ball_query(radius, ns, xyz, cnt, new, ncnt)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            pn.ball_query(radius, ns, xyz, cnt, new, ncnt)
        e1.record()
        torch.cuda.synchronize()
        out.append(dict(case=name, queries=2 * m, points_per_sample=n, radius=radius, nsample=ns,
                        us_per_call=round(e0.elapsed_time(e1) / 20 * 1e3, 1)))
        print(json.dumps(out[-1]))


if __name__ == '__main__':
    main()
