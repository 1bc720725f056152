"""Pure-tensor utilities of the PV-RCNN path — restated from
pcdet/utils/{common_utils,box_utils,box_coder_utils,loss_utils}.py (line refs per function).
Device agnostic: they run on whatever device the tensors live on."""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..devconst import const


# ---------------------------------------------------------------- common_utils
def limit_period(val, offset=0.5, period=np.pi):
    """common_utils.py:20-23"""
    return val - torch.floor(val / period + offset) * period


def rotate_points_along_z(points, angle):
    """Same result as common_utils.py:34-56 for points (B, N, 3+C), angle (B): x' = x cos - y sin, y' = x sin + y cos,
    everything else untouched — as a turn of the two coordinate slices (the reference builds B 3x3 matrices and calls a
    batched matrix product)."""
    c, s = torch.cos(angle).float().view(-1, 1), torch.sin(angle).float().view(-1, 1)
    x, y = points[:, :, 0], points[:, :, 1]
    return torch.cat((torch.stack((x * c - y * s, x * s + y * c), dim=-1), points[:, :, 2:]), dim=-1)


def get_voxel_centers(voxel_coords, downsample_times, voxel_size, point_cloud_range):
    """common_utils.py:65-82: (idx[x,y,z] + 0.5) * voxel * stride + range_min."""
    assert voxel_coords.shape[1] == 3
    voxel_centers = voxel_coords.flip(-1).float()      # (z, y, x) -> (x, y, z)
    voxel_size = const([float(v) * downsample_times for v in voxel_size], voxel_centers.device)
    pc_range = const([float(v) for v in point_cloud_range[0:3]], voxel_centers.device)
    return (voxel_centers + 0.5) * voxel_size + pc_range


# ---------------------------------------------------------------- box_utils
_CORNER_TEMPLATE = ([1, 1, -1], [1, -1, -1], [-1, -1, -1], [-1, 1, -1],
                    [1, 1, 1], [1, -1, 1], [-1, -1, 1], [-1, 1, 1])


def boxes_to_corners_3d(boxes3d):
    """box_utils.py:28-56 (corner order of the docstring there)."""
    template = const(_CORNER_TEMPLATE, boxes3d.device, boxes3d.dtype) / 2
    corners3d = boxes3d[:, None, 3:6].repeat(1, 8, 1) * template[None, :, :]
    corners3d = rotate_points_along_z(corners3d.view(-1, 8, 3), boxes3d[:, 6]).view(-1, 8, 3)
    return corners3d + boxes3d[:, None, 0:3]


def enlarge_box3d(boxes3d, extra_width=(0, 0, 0)):
    """box_utils.py:145-158"""
    large = boxes3d.clone()
    large[:, 3:6] += const([float(v) for v in extra_width], boxes3d.device, boxes3d.dtype)[None, :]
    return large


def boxes_iou_normal(boxes_a, boxes_b):
    """Axis-aligned IoU of (N, 4) x (M, 4) [x1, y1, x2, y2] boxes, == box_utils.py:248-269: the overlap rectangle from
    the broadcast corner max / min (both axes at once), union floored at 1e-6."""
    a, b = boxes_a[:, None, :], boxes_b[None, :, :]
    wh = (torch.min(a[..., 2:4], b[..., 2:4]) - torch.max(a[..., 0:2], b[..., 0:2])).clamp_min(0)
    inter = wh[..., 0] * wh[..., 1]
    area = lambda t: (t[..., 2] - t[..., 0]) * (t[..., 3] - t[..., 1])
    return inter / (area(a) + area(b) - inter).clamp_min(1e-6)


def boxes3d_lidar_to_aligned_bev_boxes(boxes3d):
    """box_utils.py:272-283"""
    rot_angle = limit_period(boxes3d[:, 6], offset=0.5, period=np.pi).abs()
    dims = boxes3d[:, 3:5]
    choose_dims = torch.where(rot_angle[:, None] < np.pi / 4, dims, dims.flip(-1))
    return torch.cat((boxes3d[:, 0:2] - choose_dims / 2, boxes3d[:, 0:2] + choose_dims / 2), dim=1)


def boxes3d_nearest_bev_iou(boxes_a, boxes_b):
    """box_utils.py:286-298"""
    return boxes_iou_normal(boxes3d_lidar_to_aligned_bev_boxes(boxes_a),
                            boxes3d_lidar_to_aligned_bev_boxes(boxes_b))


# ---------------------------------------------------------------- box_coder_utils
class ResidualCoder(object):
    """box_coder_utils.py:5-77 (encode_angle_by_sincos=False branch, the configs' default)."""

    def __init__(self, code_size=7, encode_angle_by_sincos=False, **kwargs):
        assert not encode_angle_by_sincos
        self.code_size = code_size
        self.encode_angle_by_sincos = False

    @staticmethod
    def _diag(anchors):
        """length of the anchor's BEV diagonal, (.., 1)"""
        return torch.sqrt(anchors[..., 3:4] ** 2 + anchors[..., 4:5] ** 2)

    def encode_torch(self, boxes, anchors):
        """SECOND residuals (box_coder_utils.py:14-47), block-wise: centre offsets over (diag, diag, dz_a), log size
        ratios, plain differences for the angle and any extra columns.  Sizes are floored at 1e-5 (the reference
        clamps its inputs in place, :22-23; here the inputs are left alone)."""
        size_a, size_g = anchors[..., 3:6].clamp_min(1e-5), boxes[..., 3:6].clamp_min(1e-5)
        diag = torch.sqrt(size_a[..., 0:1] ** 2 + size_a[..., 1:2] ** 2)
        scale = torch.cat([diag, diag, size_a[..., 2:3]], dim=-1)
        return torch.cat([(boxes[..., 0:3] - anchors[..., 0:3]) / scale, torch.log(size_g / size_a),
                          boxes[..., 6:] - anchors[..., 6:]], dim=-1)

    def decode_torch(self, box_encodings, anchors):
        """inverse of encode_torch (box_coder_utils.py:49-77): no flooring here, as in the reference"""
        size_a = anchors[..., 3:6]
        diag = self._diag(anchors)
        scale = torch.cat([diag, diag, size_a[..., 2:3]], dim=-1)
        return torch.cat([box_encodings[..., 0:3] * scale + anchors[..., 0:3], torch.exp(box_encodings[..., 3:6]) * size_a,
                          box_encodings[..., 6:] + anchors[..., 6:]], dim=-1)


# ---------------------------------------------------------------- loss_utils
class SigmoidFocalClassificationLoss(nn.Module):
    """loss_utils.py:9-73"""

    def __init__(self, gamma=2.0, alpha=0.25):
        super().__init__()
        self.alpha = alpha
        self.gamma = gamma

    @staticmethod
    def sigmoid_cross_entropy_with_logits(input, target):
        return torch.clamp(input, min=0) - input * target + torch.log1p(torch.exp(-torch.abs(input)))

    def forward(self, input, target, weights):
        pred_sigmoid = torch.sigmoid(input)
        alpha_weight = target * self.alpha + (1 - target) * (1 - self.alpha)
        pt = target * (1.0 - pred_sigmoid) + (1.0 - target) * pred_sigmoid
        focal_weight = alpha_weight * torch.pow(pt, self.gamma)
        loss = focal_weight * self.sigmoid_cross_entropy_with_logits(input, target)
        if len(weights.shape) == 2 or (len(weights.shape) == 1 and len(target.shape) == 2):
            weights = weights.unsqueeze(-1)
        assert len(weights.shape) == len(loss.shape)
        return loss * weights


class WeightedSmoothL1Loss(nn.Module):
    """loss_utils.py:75-137 (beta = 1/9)."""

    def __init__(self, beta=1.0 / 9.0, code_weights=None):
        super().__init__()
        self.beta = beta
        self.code_weights = None
        if code_weights is not None:
            self.register_buffer('code_weights_buf',
                                 torch.tensor(code_weights, dtype=torch.float32), persistent=False)
            self.code_weights = True

    @staticmethod
    def smooth_l1_loss(diff, beta):
        if beta < 1e-5:
            return torch.abs(diff)
        n = torch.abs(diff)
        return torch.where(n < beta, 0.5 * n ** 2 / beta, n - 0.5 * beta)

    def forward(self, input, target, weights=None):
        target = torch.where(torch.isnan(target), input, target)
        diff = input - target
        if self.code_weights is not None:
            diff = diff * self.code_weights_buf.to(diff.device).view(1, 1, -1)
        loss = self.smooth_l1_loss(diff, self.beta)
        if weights is not None:
            assert weights.shape[0] == loss.shape[0] and weights.shape[1] == loss.shape[1]
            loss = loss * weights.unsqueeze(-1)
        return loss


class WeightedCrossEntropyLoss(nn.Module):
    """loss_utils.py:181-206"""

    def forward(self, input, target, weights):
        input = input.permute(0, 2, 1)
        target = target.argmax(dim=-1)
        return F.cross_entropy(input, target, reduction='none') * weights


def get_corner_loss_lidar(pred_bbox3d, gt_bbox3d):
    """loss_utils.py:209-233"""
    assert pred_bbox3d.shape[0] == gt_bbox3d.shape[0]
    pred_box_corners = boxes_to_corners_3d(pred_bbox3d)
    gt_box_corners = boxes_to_corners_3d(gt_bbox3d)
    gt_bbox3d_flip = gt_bbox3d.clone()
    gt_bbox3d_flip[:, 6] += np.pi
    gt_box_corners_flip = boxes_to_corners_3d(gt_bbox3d_flip)
    corner_dist = torch.min(torch.norm(pred_box_corners - gt_box_corners, dim=2),
                            torch.norm(pred_box_corners - gt_box_corners_flip, dim=2))
    return WeightedSmoothL1Loss.smooth_l1_loss(corner_dist, beta=1.0).mean(dim=1)
