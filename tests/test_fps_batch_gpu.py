"""Key-point FPS of several passes as ONE launch (pcdet/pfe.py:FpsBatch): the same key points per pass as the per-pass
launches (each workgroup samples its own cloud; voxel_set_abstraction.py:119-158)."""
import pytest
import torch

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(300)]


def test_batched_fps_equals_per_pass(dev):
    from detmatch_amd import synth
    from detmatch_amd.pcdet import pfe as P

    class _Cfg(object):
        POINT_SOURCE, SAMPLE_METHOD, NUM_KEYPOINTS = 'raw_points', 'FPS', 2048

    class _Mod(object):
        model_cfg = _Cfg()
        get_sampled_points = P.VoxelSetAbstraction.get_sampled_points
        sample_keypoints_async = P.VoxelSetAbstraction.sample_keypoints_async

    def batch(seeds, few=False):
        pts, cnt = [], []
        for k, s in enumerate(seeds):
            p = torch.from_numpy(synth.lidar_frame(s)['points']).to(dev)
            if few and k == 0:
                p = p[:700]                        # fewer points than key points: the repeat-padding branch
            pts.append(torch.nn.functional.pad(p, (1, 0), value=float(k)))
            cnt.append(int(p.shape[0]))
        return dict(batch_size=len(seeds), points=torch.cat(pts), points_batch_cnt_host=cnt)

    mods = [_Mod(), _Mod(), _Mod()]
    sets = [batch((0, 1)), batch((2, 3), few=True), batch((4, 5))]
    want = [m.get_sampled_points(m, b) if False else P.VoxelSetAbstraction.get_sampled_points(m, b) for m, b in zip(mods, sets)]
    with P.FpsBatch():
        for m, b in zip(mods, sets):
            P.VoxelSetAbstraction.sample_keypoints_async(m, b)
            assert 'keypoints_async' not in b          # collected, not launched
    evs = set()
    for b, w in zip(sets, want):
        kp, ev = b['keypoints_async']
        ev.synchronize()
        evs.add(id(ev))
        assert kp.shape == w.shape == (2, 2048, 3) and torch.equal(kp, w)
    assert len(evs) == 1                               # one launch, one event
