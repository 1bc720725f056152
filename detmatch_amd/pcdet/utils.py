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
    """common_utils.py:34-56: points (B, N, 3+C), angle (B) — x' = x cos - y sin."""
    cosa = torch.cos(angle)
    sina = torch.sin(angle)
    zeros = angle.new_zeros(points.shape[0])
    ones = angle.new_ones(points.shape[0])
    rot_matrix = torch.stack((cosa, sina, zeros, -sina, cosa, zeros, zeros, zeros, ones),
                             dim=1).view(-1, 3, 3).float()
    points_rot = torch.matmul(points[:, :, 0:3], rot_matrix)
    return torch.cat((points_rot, points[:, :, 3:]), dim=-1)


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
    """box_utils.py:248-269: axis-aligned IoU of (N,4) x (M,4) [x1,y1,x2,y2]."""
    x_min = torch.max(boxes_a[:, 0, None], boxes_b[None, :, 0])
    x_max = torch.min(boxes_a[:, 2, None], boxes_b[None, :, 2])
    y_min = torch.max(boxes_a[:, 1, None], boxes_b[None, :, 1])
    y_max = torch.min(boxes_a[:, 3, None], boxes_b[None, :, 3])
    x_len = torch.clamp_min(x_max - x_min, min=0)
    y_len = torch.clamp_min(y_max - y_min, min=0)
    area_a = (boxes_a[:, 2] - boxes_a[:, 0]) * (boxes_a[:, 3] - boxes_a[:, 1])
    area_b = (boxes_b[:, 2] - boxes_b[:, 0]) * (boxes_b[:, 3] - boxes_b[:, 1])
    inter = x_len * y_len
    return inter / torch.clamp_min(area_a[:, None] + area_b[None, :] - inter, min=1e-6)


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

    def encode_torch(self, boxes, anchors):
        # the reference clamps in place (:22-23); kept out-of-place here (same values)
        anchors = torch.cat([anchors[:, :3], torch.clamp_min(anchors[:, 3:6], 1e-5), anchors[:, 6:]], -1)
        boxes = torch.cat([boxes[:, :3], torch.clamp_min(boxes[:, 3:6], 1e-5), boxes[:, 6:]], -1)
        xa, ya, za, dxa, dya, dza, ra, *cas = torch.split(anchors, 1, dim=-1)
        xg, yg, zg, dxg, dyg, dzg, rg, *cgs = torch.split(boxes, 1, dim=-1)
        diagonal = torch.sqrt(dxa ** 2 + dya ** 2)
        xt = (xg - xa) / diagonal
        yt = (yg - ya) / diagonal
        zt = (zg - za) / dza
        dxt = torch.log(dxg / dxa)
        dyt = torch.log(dyg / dya)
        dzt = torch.log(dzg / dza)
        cts = [g - a for g, a in zip(cgs, cas)]
        return torch.cat([xt, yt, zt, dxt, dyt, dzt, rg - ra, *cts], dim=-1)

    def decode_torch(self, box_encodings, anchors):
        xa, ya, za, dxa, dya, dza, ra, *cas = torch.split(anchors, 1, dim=-1)
        xt, yt, zt, dxt, dyt, dzt, rt, *cts = torch.split(box_encodings, 1, dim=-1)
        diagonal = torch.sqrt(dxa ** 2 + dya ** 2)
        xg = xt * diagonal + xa
        yg = yt * diagonal + ya
        zg = zt * dza + za
        dxg = torch.exp(dxt) * dxa
        dyg = torch.exp(dyt) * dya
        dzg = torch.exp(dzt) * dza
        cgs = [t + a for t, a in zip(cts, cas)]
        return torch.cat([xg, yg, zg, dxg, dyg, dzg, rt + ra, *cgs], dim=-1)


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
