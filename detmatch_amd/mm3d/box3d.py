"""LiDARInstance3DBoxes — the subset of mmdet3d/core/bbox/structures/{base_box3d,lidar_box3d,
utils}.py that the DetMatch path touches.  Box rows are (x, y, z_bottom, x_size, y_size,
z_size, yaw); the origin argument re-anchors on construction (base_box3d.py:36-66)."""
import numpy as np
import torch

from ..devconst import const


def limit_period(val, offset=0.5, period=np.pi):
    """structures/utils.py:5-21"""
    return val - torch.floor(val / period + offset) * period


def _planar_turn(u, v, angles):
    """(u, v) -> (u cos + v sin, -u sin + v cos) with one angle per leading row; u, v (N, M), angles (N)."""
    c, s = torch.cos(angles).unsqueeze(-1), torch.sin(angles).unsqueeze(-1)
    return u * c + v * s, v * c - u * s


def rotation_3d_in_axis(points, angles, axis=0):
    """Same result as structures/utils.py:24-61 for points (N, M, 3), angles (N) — written as a planar turn of two
    coordinate slices (no 3x3 matrices, no batched matrix product: a vendor GEMM per call for 8 corners per box is
    what the reference pays).  The reference's conventions: axis 2 turns (x, y), axis 1 turns (x, z), and axis 0
    ALSO moves the axes — (x, y, z) -> (z, x', y') — exactly as its matrix for that case does."""
    x, y, z = points.unbind(-1)
    if axis == 2 or axis == -1:
        a, b = _planar_turn(x, y, angles)
        out = (a, b, z)
    elif axis == 1:
        a, b = _planar_turn(x, z, angles)
        out = (a, y, b)
    elif axis == 0:
        a, b = _planar_turn(x, y, angles)
        out = (z, a, b)
    else:
        raise ValueError('axis should in range [0, 1, 2], got %s' % axis)
    return torch.stack(out, dim=-1)


def _rows_times_3x3(xyz, m):
    """xyz (N, 3) @ m (3, 3) as three scaled row sums (no vendor GEMM for a 3-wide contraction)."""
    return xyz[:, 0:1] * m[0] + xyz[:, 1:2] * m[1] + xyz[:, 2:3] * m[2]


class LiDARInstance3DBoxes(object):

    def __init__(self, tensor, box_dim=7, with_yaw=True, origin=(0.5, 0.5, 0)):
        device = tensor.device if isinstance(tensor, torch.Tensor) else torch.device('cpu')
        tensor = torch.as_tensor(tensor, dtype=torch.float32, device=device)
        if tensor.numel() == 0:
            tensor = tensor.reshape((0, box_dim)).to(dtype=torch.float32, device=device)
        assert tensor.dim() == 2 and tensor.size(-1) == box_dim, tensor.size()
        self.box_dim = box_dim
        self.with_yaw = with_yaw
        self.tensor = tensor.clone()
        if tuple(origin) != (0.5, 0.5, 0):
            shift = const([0.5 - origin[0], 0.5 - origin[1], 0.0 - origin[2]], self.tensor.device)
            self.tensor[:, :3] += self.tensor[:, 3:6] * shift

    # ---- views (base_box3d.py:68-147) ----------------------------------------
    @property
    def volume(self):
        return self.tensor[:, 3] * self.tensor[:, 4] * self.tensor[:, 5]

    @property
    def dims(self):
        return self.tensor[:, 3:6]

    @property
    def yaw(self):
        return self.tensor[:, 6]

    @property
    def height(self):
        return self.tensor[:, 5]

    @property
    def bottom_center(self):
        return self.tensor[:, :3]

    center = bottom_center

    @property
    def gravity_center(self):
        """lidar_box3d.py:39-46"""
        bc = self.bottom_center
        gc = torch.zeros_like(bc)
        gc[:, :2] = bc[:, :2]
        gc[:, 2] = bc[:, 2] + self.tensor[:, 5] * 0.5
        return gc

    @property
    def corners(self):
        """lidar_box3d.py:48-87: (N, 8, 3), order (x0y0z0, x0y0z1, x0y1z1, x0y1z0, x1y0z0, ...)."""
        assert len(self.tensor) != 0
        dims = self.dims
        cn = np.stack(np.unravel_index(np.arange(8), [2] * 3), axis=1).astype(np.float32)
        cn = cn[[0, 1, 3, 2, 4, 5, 7, 6]] - np.array([0.5, 0.5, 0], np.float32)
        corners_norm = const(cn, dims.device, dims.dtype)
        corners = dims.view([-1, 1, 3]) * corners_norm.reshape([1, 8, 3])
        corners = rotation_3d_in_axis(corners, self.tensor[:, 6], axis=2)
        return corners + self.tensor[:, :3].view(-1, 1, 3)

    @property
    def bev(self):
        t = self.tensor
        return torch.cat([t[:, 0:2], t[:, 3:5], t[:, 6:7]], dim=1)

    # ---- in-place transforms -----------------------------------------------------
    def rotate(self, angle, points=None):
        """lidar_box3d.py:116-167 (angle scalar or a 3x3 matrix M applied as xyz @ M)."""
        if not isinstance(angle, torch.Tensor):
            angle = self.tensor.new_tensor(angle)
        assert angle.shape == torch.Size([3, 3]) or angle.numel() == 1
        if angle.numel() == 1:
            rot_sin = torch.sin(angle)
            rot_cos = torch.cos(angle)
            rot_mat_T = self.tensor.new_tensor([[rot_cos, -rot_sin, 0], [rot_sin, rot_cos, 0],
                                                [0, 0, 1]])
        else:
            rot_mat_T = angle.to(self.tensor)
            angle = torch.atan2(rot_mat_T[1, 0], rot_mat_T[0, 0])
        self.tensor = torch.cat([_rows_times_3x3(self.tensor[:, :3], rot_mat_T), self.tensor[:, 3:6],
                                 (self.tensor[:, 6] + angle).unsqueeze(-1), self.tensor[:, 7:]], -1)
        if points is not None:
            points[:, :3] = _rows_times_3x3(points[:, :3], rot_mat_T)
            return points, rot_mat_T

    def flip(self, bev_direction='horizontal', points=None):
        """lidar_box3d.py:169-203"""
        assert bev_direction in ('horizontal', 'vertical')
        t = self.tensor
        if bev_direction == 'horizontal':
            yaw = -t[:, 6] + np.pi if self.with_yaw else t[:, 6]
            self.tensor = torch.cat([t[:, 0:1], -t[:, 1:2], t[:, 2:6], yaw.unsqueeze(-1), t[:, 7:]], -1)
        else:
            yaw = -t[:, 6] if self.with_yaw else t[:, 6]
            self.tensor = torch.cat([-t[:, 0:1], t[:, 1:6], yaw.unsqueeze(-1), t[:, 7:]], -1)
        if points is not None:
            if bev_direction == 'horizontal':
                points[:, 1] = -points[:, 1]
            else:
                points[:, 0] = -points[:, 0]
            return points

    def translate(self, trans_vector):
        if not isinstance(trans_vector, torch.Tensor):
            trans_vector = self.tensor.new_tensor(trans_vector)
        self.tensor = torch.cat([self.tensor[:, :3] + trans_vector.to(self.tensor),
                                 self.tensor[:, 3:]], -1)

    def scale(self, scale_factor):
        self.tensor = torch.cat([self.tensor[:, :6] * scale_factor, self.tensor[:, 6:7],
                                 self.tensor[:, 7:] * scale_factor], -1)

    def limit_yaw(self, offset=0.5, period=np.pi):
        self.tensor = torch.cat([self.tensor[:, :6],
                                 limit_period(self.tensor[:, 6], offset, period).unsqueeze(-1),
                                 self.tensor[:, 7:]], -1)

    # ---- container protocol ----------------------------------------------------
    def __getitem__(self, item):
        if isinstance(item, int):
            return LiDARInstance3DBoxes(self.tensor[item].view(1, -1), box_dim=self.box_dim,
                                        with_yaw=self.with_yaw)
        if isinstance(item, torch.Tensor) and item.dim() == 1 and item.dtype == torch.int64:
            b = self.tensor.index_select(0, item)      # == tensor[item]; cheaper backward (bbox_utils.take)
        else:
            b = self.tensor[item]
        assert b.dim() == 2
        return LiDARInstance3DBoxes(b, box_dim=self.box_dim, with_yaw=self.with_yaw)

    def __len__(self):
        return self.tensor.shape[0]

    @property
    def shape(self):
        return self.tensor.shape

    @property
    def device(self):
        return self.tensor.device

    def __repr__(self):
        return self.__class__.__name__ + '(\n    ' + str(self.tensor) + ')'

    @classmethod
    def cat(cls, boxes_list):
        assert isinstance(boxes_list, (list, tuple))
        if len(boxes_list) == 0:
            return cls(torch.empty(0))
        return cls(torch.cat([b.tensor for b in boxes_list], dim=0),
                   box_dim=boxes_list[0].tensor.shape[1], with_yaw=boxes_list[0].with_yaw)

    def _wrap(self, t):
        out = LiDARInstance3DBoxes.__new__(LiDARInstance3DBoxes)
        out.box_dim, out.with_yaw, out.tensor = self.box_dim, self.with_yaw, t
        return out

    def to(self, device):
        return self._wrap(self.tensor.to(device))

    def clone(self):
        return self._wrap(self.tensor.clone())

    def detach(self):
        """base_box3d.py:350 (added by DetMatch)"""
        return self._wrap(self.tensor.detach())

    def new_box(self, data):
        new_tensor = self.tensor.new_tensor(data) if not isinstance(data, torch.Tensor) \
            else data.to(self.device)
        return LiDARInstance3DBoxes(new_tensor, box_dim=self.box_dim, with_yaw=self.with_yaw)
