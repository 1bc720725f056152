"""Key-point features from the BEV map (csrc/bev_interp.hip) against the tensor formulation of
voxel_set_abstraction.py:9-40,113-117 (itself pinned to the reference's outputs in
tests/test_pcdet_torch_golden.py): forward bit-exact, gradient of the map incl. cells hit by several key
points, key points outside the map (clamped cells), run-to-run reproducibility."""
import numpy as np
import pytest
import torch

from detmatch_amd import configs
from detmatch_amd.pcdet.config import ConfigDict

pytestmark = pytest.mark.gpu


def _pfe(dev):
    from detmatch_amd.pcdet.pfe import VoxelSetAbstraction
    cfg = ConfigDict(configs.pvrcnn_kitti_model()['pcdet_model'])
    return VoxelSetAbstraction(cfg.PFE, voxel_size=[0.05, 0.05, 0.1],
                               point_cloud_range=[0, -40, -3, 70.4, 40, 1], num_bev_features=256,
                               num_rawpoint_features=4).to(dev)


@pytest.mark.parametrize('c,h,w,k', [(256, 200, 176, 2048), (64, 50, 44, 300)])
def test_bev_interpolation_matches_tensor_formulation(dev, c, h, w, k):
    pfe = _pfe(dev)
    g = torch.Generator().manual_seed(3)
    stride = 8 if h == 200 else 32
    kp = torch.rand(2, k, 3, generator=g) * torch.tensor([70.4, 80.0, 4.0]) + torch.tensor([0.0, -40.0, -3.0])
    kp[:, :20] = kp[:, 20:40]                      # several key points in the same cell
    kp[0, 40:60, :2] += 100.0                      # outside the map: clamped cells, reference weights
    kp[1, 60:70, :2] -= 100.0
    kp[1, 70] = torch.tensor([70.4, 40.0, 0.0])    # exactly on the far border
    kp = kp.to(dev)
    outs = []
    for fused in (True, False):
        bev = torch.randn(2, c, h, w, generator=torch.Generator().manual_seed(5)).to(dev) \
            .contiguous(memory_format=torch.channels_last).requires_grad_(True)
        y = pfe.interpolate_from_bev_features(kp, bev, 2, stride, fused=fused)
        gy = torch.randn(y.shape, generator=torch.Generator().manual_seed(6)).to(dev)
        y.backward(gy)
        outs.append((y.detach(), bev.grad.clone()))
    assert outs[0][0].shape == (2, k, c)
    assert torch.equal(outs[0][0], outs[1][0])
    np.testing.assert_allclose(outs[0][1].cpu().numpy(), outs[1][1].cpu().numpy(), rtol=1e-5, atol=1e-5)
    assert outs[0][1].is_contiguous(memory_format=torch.channels_last)
    # the scatter has no atomics: bitwise reproducible
    bev = torch.randn(2, c, h, w, generator=torch.Generator().manual_seed(5)).to(dev) \
        .contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y = pfe.interpolate_from_bev_features(kp, bev, 2, stride)
    y.backward(torch.randn(y.shape, generator=torch.Generator().manual_seed(6)).to(dev))
    assert torch.equal(bev.grad, outs[0][1])
