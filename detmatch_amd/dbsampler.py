"""GT-paste augmentation — host mirror of mmdet3d/datasets/pipelines/dbsampler.py
(`BatchSampler` :15-84, `DataBaseSampler` :87-387) and `ObjectSample`
(mmdet3d/datasets/pipelines/transforms_3d.py:248-352), with the two geometric tests on the device:

  * collision of a candidate with the scene's boxes and with earlier candidates: the reference builds
    BEV corner boxes (box_np_ops.center_to_corner_box2d: corners turned CLOCKWISE by yaw) and runs the
    numba `box_collision_test`; here the rotated-overlap kernel of iou3d_nms.hip (counter-clockwise
    headings: the boxes are handed over with heading = -yaw) — a pair collides when its BEV overlap area
    is positive;
  * removal of the scene points that fall inside a pasted box: numba `points_in_rbbox` in the reference,
    the points-in-boxes kernel here.

Database format (unchanged): `kitti_dbinfos_*.pkl` = {class name: [dict(name, path, box3d_lidar (7,)
bottom-centre x,y,z,w,l,h,yaw, num_points_in_gt, difficulty, ...)]}, one float32 x 4 `.bin` of
box-relative points per object.  Random draws come from a `numpy.random.RandomState` in the reference's
order (one shuffle per class at construction, reshuffle on wrap-around)."""
import copy
import os
import pickle

import numpy as np
import torch

from . import _lib


class BatchSampler(object):
    """dbsampler.py:15-84: cyclic sampler over a shuffled index list."""

    def __init__(self, sampled_list, name=None, epoch=None, shuffle=True, drop_reminder=False, rng=None):
        self._rng = rng if rng is not None else np.random
        self._sampled_list = sampled_list
        self._indices = np.arange(len(sampled_list))
        if shuffle:
            self._rng.shuffle(self._indices)
        self._idx = 0
        self._example_num = len(sampled_list)
        self._name, self._shuffle = name, shuffle

    def _sample(self, num):
        if self._idx + num >= self._example_num:
            ret = self._indices[self._idx:].copy()
            assert self._name is not None
            if self._shuffle:
                self._rng.shuffle(self._indices)
            self._idx = 0
        else:
            ret = self._indices[self._idx:self._idx + num]
            self._idx += num
        return ret

    def sample(self, num):
        return [self._sampled_list[i] for i in self._sample(num)]


def _boxes7_bev(b):
    """mm3d LiDAR boxes (n, 7) [x, y, z_bottom, w, l, h, yaw] as rows the BEV overlap kernel takes:
    x-extent w, y-extent l, turned clockwise by yaw = heading -yaw."""
    z = torch.zeros_like(b[:, :1])
    return torch.cat([b[:, 0:2], z, b[:, 3:5], z + 1, -b[:, 6:7]], dim=1).contiguous()


def collision_matrix(boxes, device):
    """(n, 7) host array -> (n, n) bool numpy: BEV rectangles i and j intersect (diagonal False)."""
    from . import iou3d_nms
    t = _boxes7_bev(torch.as_tensor(np.asarray(boxes, dtype=np.float32), device=device))
    ov = iou3d_nms.boxes_overlap_bev_exact(t, t) > 0     # exact: boxes 1 cm apart do not collide
    ov.fill_diagonal_(False)
    return ov.cpu().numpy()


class DataBaseSampler(object):

    def __init__(self, info_path, data_root, rate, prepare, sample_groups, classes=None, use_road_plane=False,
                 limit_whole_scene=True, points_loader=None, rng=None, device=None):
        self.data_root, self.info_path, self.rate, self.prepare = data_root, info_path, rate, prepare
        self.classes = classes
        self.cat2label = {name: i for i, name in enumerate(classes)}
        self.label2cat = {i: name for i, name in enumerate(classes)}
        self.use_road_plane, self.limit_whole_scene = use_road_plane, limit_whole_scene
        self.load_dim = (points_loader or {}).get('load_dim', 4)
        self.rng = rng if rng is not None else np.random
        self.device = torch.device(device) if device is not None else torch.device('cuda', 0)
        with open(info_path, 'rb') as f:
            db_infos = pickle.load(f)
        for prep_func, val in prepare.items():
            db_infos = getattr(self, prep_func)(db_infos, val)
        self.db_infos = db_infos
        self.sample_classes = list(sample_groups.keys())
        self.sample_max_nums = [int(v) for v in sample_groups.values()]
        self.sampler_dict = {k: BatchSampler(v, k, shuffle=True, rng=self.rng) for k, v in db_infos.items()}

    @staticmethod
    def filter_by_difficulty(db_infos, removed_difficulty):
        return {k: [i for i in v if i['difficulty'] not in removed_difficulty] for k, v in db_infos.items()}

    @staticmethod
    def filter_by_min_points(db_infos, min_gt_points_dict):
        for name, min_num in min_gt_points_dict.items():
            if int(min_num) > 0:
                db_infos[name] = [i for i in db_infos[name] if i['num_points_in_gt'] >= int(min_num)]
        return db_infos

    @staticmethod
    def put_boxes_on_road_planes(sampled_bboxes, sampled_points, input_dict):
        """dbsampler.py:192-243: move every pasted object (box + its points) onto the frame's road plane
        a x + b y + c z + d = 0 (camera frame)."""
        calib = input_dict['calib']
        rect, trv2c = calib['R0_rect'].astype(np.float32), calib['Tr_velo_to_cam'].astype(np.float32)
        a, b, c, d = input_dict['road_plane']
        centre = sampled_bboxes[:, :3].copy()
        centre[:, 2] = sampled_bboxes[:, 2] + sampled_bboxes[:, 5] * 0.5
        hom = np.concatenate([centre, np.ones((len(centre), 1), dtype=np.float32)], axis=1)
        cam = hom @ trv2c.T @ rect.T
        cam = cam[:, :3] / cam[:, [3]]
        cam[:, 1] = (-d - a * cam[:, 0] - c * cam[:, 2]) / b
        back = np.concatenate([cam, np.ones((len(cam), 1), dtype=np.float32)], axis=1) @ np.linalg.inv((rect @ trv2c).T)
        mv_height = centre[:, 2] - sampled_bboxes[:, 5] / 2 - back[:, 2]
        sampled_bboxes[:, 2] -= mv_height
        for i, p in enumerate(sampled_points):
            p[:, 2] -= float(mv_height[i])
        return sampled_bboxes, sampled_points

    def sample_class_v2(self, name, num, gt_bboxes):
        """dbsampler.py:345-387: draw `num` objects, keep those that hit neither the scene nor an earlier
        kept candidate (a rejected candidate no longer blocks later ones)."""
        sampled = copy.deepcopy(self.sampler_dict[name].sample(num))
        num_gt = gt_bboxes.shape[0]
        if not sampled:
            return []
        sp_boxes = np.stack([i['box3d_lidar'] for i in sampled], axis=0)
        coll = collision_matrix(np.concatenate([gt_bboxes, sp_boxes], axis=0), self.device)
        valid = []
        for i in range(num_gt, num_gt + len(sampled)):
            if coll[i].any():
                coll[i] = False
                coll[:, i] = False
            else:
                valid.append(sampled[i - num_gt])
        return valid

    def _load_points(self, info):
        path = os.path.join(self.data_root, info['path']) if self.data_root else info['path']
        pts = np.fromfile(path, dtype=np.float32).reshape(-1, self.load_dim)[:, :4].copy()
        pts[:, :3] += info['box3d_lidar'][:3]                    # stored relative to the box
        return pts

    def sample_all(self, gt_bboxes, gt_labels, img=None, input_dict=None):
        """dbsampler.py:245-343 -> None or dict(gt_labels_3d, gt_bboxes_3d (k,7), points (m,4) numpy,
        group_ids)."""
        gt_bboxes = np.asarray(gt_bboxes, dtype=np.float32).reshape(-1, 7)
        nums = []
        for name, max_num in zip(self.sample_classes, self.sample_max_nums):
            label = self.cat2label[name]
            n = int(max_num - np.sum([l == label for l in gt_labels])) if self.limit_whole_scene else int(max_num)
            nums.append(np.round(self.rate * n).astype(np.int64))
        sampled, sampled_boxes, avoid = [], [], gt_bboxes
        for name, n in zip(self.sample_classes, nums):
            if n > 0:
                got = self.sample_class_v2(name, n, avoid)
                sampled += got
                if got:
                    box = np.stack([s['box3d_lidar'] for s in got], axis=0)
                    sampled_boxes.append(box)
                    avoid = np.concatenate([avoid, box], axis=0)
        if not sampled:
            return None
        sampled_boxes = np.concatenate(sampled_boxes, axis=0)
        pts = [self._load_points(info) for info in sampled]
        if self.use_road_plane:
            sampled_boxes, pts = self.put_boxes_on_road_planes(sampled_boxes, pts, input_dict)
        return dict(gt_labels_3d=np.array([self.cat2label[s['name']] for s in sampled], dtype=np.int64),
                    gt_bboxes_3d=sampled_boxes, points=np.concatenate(pts, axis=0),
                    group_ids=np.arange(gt_bboxes.shape[0], gt_bboxes.shape[0] + len(sampled)))


class ObjectSample(object):
    """transforms_3d.py:248-352 (sample_2d=False, the DetMatch setting): paste database objects into a
    frame dict(points (N,4) device tensor, gt_bboxes_3d, gt_labels_3d[, calib, road_plane])."""

    def __init__(self, db_sampler, sample_2d=False):
        assert not sample_2d, 'the DetMatch recipes paste in 3D only'
        if isinstance(db_sampler, dict):
            cfg = dict(db_sampler)
            cfg.pop('type', None)
            db_sampler = DataBaseSampler(**cfg)
        self.db_sampler = db_sampler

    @staticmethod
    def remove_points_in_boxes(points, boxes):
        """Drop the scene points inside any of the (k,7) bottom-centre boxes (device kernel)."""
        from . import roiaware_pool3d
        _lib.require_device(points)
        b = torch.as_tensor(np.asarray(boxes, dtype=np.float32), device=points.device)
        # mm3d (x, y, z_bottom, w, l, h, yaw) -> the kernel's (x, y, z_centre, dx, dy, dz, heading):
        # mmdet3d/models/detectors/openpcdet.py:100-122
        pc = torch.stack([b[:, 0], b[:, 1], b[:, 2] + b[:, 5] / 2, b[:, 4], b[:, 3], b[:, 5],
                          -(b[:, 6] + np.pi / 2)], dim=1)
        idx = roiaware_pool3d.points_in_boxes_gpu(points[None, :, :3].contiguous(), pc[None].contiguous())[0]
        return points[idx < 0]

    def __call__(self, frame):
        from .mm3d.box3d import LiDARInstance3DBoxes
        boxes, labels = frame['gt_bboxes_3d'], frame['gt_labels_3d']
        lab_np = labels.cpu().numpy() if torch.is_tensor(labels) else np.asarray(labels)
        got = self.db_sampler.sample_all(boxes.tensor.cpu().numpy(), lab_np, img=None, input_dict=frame)
        if got is not None:
            pts = self.remove_points_in_boxes(frame['points'], got['gt_bboxes_3d'])
            new = torch.as_tensor(got['points'], device=pts.device)
            frame['points'] = torch.cat([new, pts], dim=0)
            all_boxes = np.concatenate([boxes.tensor.cpu().numpy(), got['gt_bboxes_3d']])
            frame['gt_bboxes_3d'] = LiDARInstance3DBoxes(torch.from_numpy(all_boxes), box_dim=7)
            lab = np.concatenate([lab_np, got['gt_labels_3d']], axis=0).astype(np.int64)
            frame['gt_labels_3d'] = torch.from_numpy(lab)
        return frame
