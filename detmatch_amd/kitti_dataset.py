"""KITTI on-disk reader and result formatting — host mirror of
mmdet3d/datasets/kitti_dataset.py: `get_data_info` :109-151, `get_ann_info` :153-216,
`format_results` :265-318, `evaluate` :320-440 (teacher / student result fan-out included),
`bbox2result_kitti` :441-552, `bbox2result_kitti2d` :554-658, `convert_valid_bboxes` :660-740, with the
box-mode conversions it needs (core/bbox/structures/box_3d_mode.py:63-166, cam_box3d.py:102-142,
utils.py:100-149).  File formats: `kitti_infos_*.pkl` (list of dicts: image / point_cloud / calib /
annos), velodyne `.bin` float32 x 4, `image_2/*.png`.

The AP itself is detmatch_amd/kitti_eval.py.  Data pipelines are not run here: `load_points` /
`load_image` hand the raw frame to the device pipeline (detmatch_amd/pipeline3d.py)."""
import os
import pickle
import tempfile

import numpy as np
import torch

from .mm3d.box3d import LiDARInstance3DBoxes
from .mm3d.ssl import add_prefix


# ---------------------------------------------------------------- box-mode conversions (yaw is carried over unchanged)
def lidar_to_cam(arr, rt_mat):
    """Box3DMode.convert(LIDAR -> CAM): centres through rt_mat (3x3 or 4x4), sizes (x,y,z) -> (y,z,x)."""
    arr = torch.as_tensor(arr).clone()
    rt = torch.as_tensor(np.asarray(rt_mat), dtype=arr.dtype)
    xyz = _apply(arr[:, :3], rt)
    return torch.cat([xyz, arr[:, 4:5], arr[:, 5:6], arr[:, 3:4], arr[:, 6:]], dim=-1)


def cam_to_lidar(arr, rt_mat):
    """Box3DMode.convert(CAM -> LIDAR): sizes (x,y,z) -> (z,x,y)."""
    arr = torch.as_tensor(arr).clone()
    rt = torch.as_tensor(np.asarray(rt_mat), dtype=arr.dtype)
    xyz = _apply(arr[:, :3], rt)
    return torch.cat([xyz, arr[:, 5:6], arr[:, 3:4], arr[:, 4:5], arr[:, 6:]], dim=-1)


def _apply(xyz, rt):
    if rt.size(1) == 4:
        return (torch.cat([xyz, xyz.new_ones(xyz.size(0), 1)], dim=-1) @ rt.t())[:, :3]
    return xyz @ rt.t()


def cam_box_corners(cam):
    """CameraInstance3DBoxes.corners: (N,7) [x,y,z, l(x),h(y),w(z), ry], origin (0.5, 1, 0.5) -> (N,8,3)."""
    assert len(cam) != 0
    dims = cam[:, 3:6]
    norm = torch.from_numpy(np.stack(np.unravel_index(np.arange(8), [2] * 3), axis=1)).to(dims)
    norm = norm[[0, 1, 3, 2, 4, 5, 7, 6]] - dims.new_tensor([0.5, 1, 0.5])
    corners = dims.view(-1, 1, 3) * norm.reshape(1, 8, 3)
    s, c = torch.sin(cam[:, 6]), torch.cos(cam[:, 6])
    one, zero = torch.ones_like(c), torch.zeros_like(c)
    rot_t = torch.stack([torch.stack([c, zero, -s]), torch.stack([zero, one, zero]), torch.stack([s, zero, c])])
    corners = torch.einsum('aij,jka->aik', (corners, rot_t))            # rotation_3d_in_axis(axis=1)
    return corners + cam[:, :3].view(-1, 1, 3)


def points_cam2img(points_3d, proj_mat):
    """utils.py:100-149: (..., 3) camera points through a 3x4 / 4x4 projection -> (..., 2) pixels."""
    proj_mat = torch.as_tensor(proj_mat, dtype=points_3d.dtype)
    if proj_mat.shape[0] == 3:
        full = torch.eye(4, dtype=proj_mat.dtype)
        full[:3, :proj_mat.shape[1]] = proj_mat
        proj_mat = full
    p4 = torch.cat([points_3d, points_3d.new_ones(*points_3d.shape[:-1], 1)], dim=-1)
    p2 = torch.matmul(p4, proj_mat.t())
    return p2[..., :2] / p2[..., 2:3]


_EMPTY_ANNO = dict(name=np.array([]), truncated=np.array([]), occluded=np.array([]), alpha=np.array([]),
                   bbox=np.zeros([0, 4]), dimensions=np.zeros([0, 3]), location=np.zeros([0, 3]),
                   rotation_y=np.array([]), score=np.array([]))


class KittiDataset(object):
    CLASSES = ('car', 'pedestrian', 'cyclist')

    def __init__(self, data_root, ann_file, split, pts_prefix='velodyne', pipeline=None, classes=None,
                 modality=None, box_type_3d='LiDAR', filter_empty_gt=True, test_mode=False,
                 pcd_limit_range=(0, -40, -3, 70.4, 40, 0.0), completely_remove_other_classes=False,
                 load_interval=1):
        assert box_type_3d == 'LiDAR'
        self.data_root, self.ann_file, self.split = data_root, ann_file, split
        self.root_split = os.path.join(data_root, split)
        self.pts_prefix, self.test_mode = pts_prefix, test_mode
        self.modality = modality if modality is not None else dict(use_lidar=True, use_camera=False)
        self.pcd_limit_range = list(pcd_limit_range)
        self.completely_remove_other_classes = completely_remove_other_classes
        if classes is not None:
            self.CLASSES = tuple(classes)
        with open(ann_file, 'rb') as f:
            self.data_infos = pickle.load(f)[::load_interval]
        self.pipeline = pipeline

    def __len__(self):
        return len(self.data_infos)

    # ---- reading ------------------------------------------------------------------------------
    def _get_pts_filename(self, idx):
        return os.path.join(self.root_split, self.pts_prefix, '%06d.bin' % idx)

    def get_data_info(self, index):
        info = self.data_infos[index]
        sample_idx = info['image']['image_idx']
        rect = info['calib']['R0_rect'].astype(np.float32)
        trv2c = info['calib']['Tr_velo_to_cam'].astype(np.float32)
        p2 = info['calib']['P2'].astype(np.float32)
        out = dict(sample_idx=sample_idx, pts_filename=self._get_pts_filename(sample_idx), img_prefix=None,
                   img_info=dict(filename=os.path.join(self.data_root, info['image']['image_path'])),
                   lidar2img=p2 @ rect @ trv2c)
        if not self.test_mode:
            out['ann_info'] = self.get_ann_info(index)
        return out

    def get_ann_info(self, index):
        info = self.data_infos[index]
        rect = info['calib']['R0_rect'].astype(np.float32)
        trv2c = info['calib']['Tr_velo_to_cam'].astype(np.float32)
        annos = info['annos']
        keep = np.array([i for i, x in enumerate(annos['name']) if x != 'DontCare'], dtype=np.int64)
        annos = {k: v[keep] for k, v in annos.items()}
        cam = np.concatenate([annos['location'], annos['dimensions'], annos['rotation_y'][..., None]],
                             axis=1).astype(np.float32)
        lidar = cam_to_lidar(torch.from_numpy(cam), np.linalg.inv(rect @ trv2c))
        names, bboxes = annos['name'], annos['bbox'].astype('float32')
        if self.completely_remove_other_classes:
            sel = np.array([i for i, x in enumerate(names) if x in self.CLASSES], dtype=np.int64)
            names, bboxes, lidar = names[sel], bboxes[sel], lidar[sel]
        labels = np.array([self.CLASSES.index(c) if c in self.CLASSES else -1 for c in names]).astype(np.int64)
        return dict(gt_bboxes_3d=LiDARInstance3DBoxes(lidar, box_dim=7), gt_labels_3d=labels.copy(),
                    bboxes=bboxes, labels=labels, gt_names=names)

    def load_points(self, index, load_dim=4, use_dim=4):
        """LoadPointsFromFile (coord_type='LIDAR'): float32 rows of the velodyne .bin."""
        pts = np.fromfile(self._get_pts_filename(self.data_infos[index]['image']['image_idx']), dtype=np.float32)
        return pts.reshape(-1, load_dim)[:, :use_dim]

    def load_image(self, index):
        """LoadImageFromFile: (H, W, 3) uint8, BGR channel order (mmcv / cv2 convention)."""
        from PIL import Image
        info = self.data_infos[index]
        rgb = np.asarray(Image.open(os.path.join(self.data_root, info['image']['image_path'])).convert('RGB'))
        return np.ascontiguousarray(rgb[:, :, ::-1])

    # ---- results ------------------------------------------------------------------------------
    def convert_valid_bboxes(self, box_dict, info):
        boxes, scores, labels = box_dict['boxes_3d'], box_dict['scores_3d'], box_dict['labels_3d']
        sample_idx = info['image']['image_idx']
        t = boxes.tensor.detach().cpu().clone()
        t[:, -1] = t[:, -1] - np.pi                       # "the hack of yaw" (:694)
        lidar = LiDARInstance3DBoxes(t, box_dim=t.shape[-1]) if len(t) else boxes
        if len(t):
            lidar.limit_yaw(offset=0.5, period=np.pi * 2)
        if len(t) == 0:
            return dict(bbox=np.zeros([0, 4]), box3d_camera=np.zeros([0, 7]), box3d_lidar=np.zeros([0, 7]),
                        scores=np.zeros([0]), label_preds=np.zeros([0, 4]), sample_idx=sample_idx)
        rect = info['calib']['R0_rect'].astype(np.float32)
        trv2c = info['calib']['Tr_velo_to_cam'].astype(np.float32)
        p2 = torch.from_numpy(info['calib']['P2'].astype(np.float32))
        cam = lidar_to_cam(lidar.tensor, rect @ trv2c)
        px = points_cam2img(cam_box_corners(cam), p2)
        box_2d = torch.cat([px.min(dim=1)[0], px.max(dim=1)[0]], dim=1)
        shape = lidar.tensor.new_tensor(np.asarray(info['image']['image_shape'], dtype=np.float32))
        valid_cam = (box_2d[:, 0] < shape[1]) & (box_2d[:, 1] < shape[0]) & (box_2d[:, 2] > 0) & (box_2d[:, 3] > 0)
        rng = lidar.tensor.new_tensor(self.pcd_limit_range)
        center = lidar.tensor[:, :3]                      # bottom centre (BaseInstance3DBoxes.center)
        valid = valid_cam & ((center > rng[:3]) & (center < rng[3:])).all(-1)
        if valid.sum() > 0:
            scores, labels = scores.detach().cpu(), labels.detach().cpu()
            return dict(bbox=box_2d[valid].numpy(), box3d_camera=cam[valid].numpy(),
                        box3d_lidar=lidar.tensor[valid].numpy(), scores=scores[valid].numpy(),
                        label_preds=labels[valid].numpy(), sample_idx=sample_idx)
        return dict(bbox=np.zeros([0, 4]), box3d_camera=np.zeros([0, 7]), box3d_lidar=np.zeros([0, 7]),
                    scores=np.zeros([0]), label_preds=np.zeros([0, 4]), sample_idx=sample_idx)

    def bbox2result_kitti(self, net_outputs, class_names, pklfile_prefix=None, submission_prefix=None):
        assert len(net_outputs) == len(self.data_infos), 'invalid list length of network outputs'
        if submission_prefix is not None:
            os.makedirs(submission_prefix, exist_ok=True)
        det_annos = []
        for idx, pred in enumerate(net_outputs):
            info = self.data_infos[idx]
            sample_idx = info['image']['image_idx']
            image_shape = info['image']['image_shape'][:2]
            bd = self.convert_valid_bboxes(pred, info)
            if len(bd['bbox']) > 0:
                bbox = bd['bbox'].copy()
                bbox[:, 2:] = np.minimum(bbox[:, 2:], image_shape[::-1])
                bbox[:, :2] = np.maximum(bbox[:, :2], [0, 0])
                cam, lid = bd['box3d_camera'], bd['box3d_lidar']
                anno = dict(name=np.array([class_names[int(l)] for l in bd['label_preds']]),
                            truncated=np.zeros(len(bbox)), occluded=np.zeros(len(bbox), dtype=np.int64),
                            alpha=-np.arctan2(-lid[:, 1], lid[:, 0]) + cam[:, 6], bbox=bbox,
                            dimensions=cam[:, 3:6], location=cam[:, :3], rotation_y=cam[:, 6],
                            score=bd['scores'])
            else:
                anno = {k: v.copy() for k, v in _EMPTY_ANNO.items()}
            if submission_prefix is not None:
                with open('%s/%06d.txt' % (submission_prefix, sample_idx), 'w') as f:
                    b, loc, dims = anno['bbox'], anno['location'], anno['dimensions']
                    for i in range(len(b)):
                        print('{} -1 -1 {:.4f} {:.4f} {:.4f} {:.4f} {:.4f} {:.4f} {:.4f} {:.4f} {:.4f} {:.4f} {:.4f} '
                              '{:.4f} {:.4f}'.format(anno['name'][i], anno['alpha'][i], b[i][0], b[i][1], b[i][2],
                                                     b[i][3], dims[i][1], dims[i][2], dims[i][0], loc[i][0],
                                                     loc[i][1], loc[i][2], anno['rotation_y'][i],
                                                     anno['score'][i]), file=f)
            anno['sample_idx'] = np.array([sample_idx] * len(anno['score']), dtype=np.int64)
            det_annos.append(anno)
        if pklfile_prefix is not None:
            out = pklfile_prefix if pklfile_prefix.endswith(('.pkl', '.pickle')) else pklfile_prefix + '.pkl'
            with open(out, 'wb') as f:
                pickle.dump(det_annos, f)
        return det_annos

    def bbox2result_kitti2d(self, net_outputs, class_names, pklfile_prefix=None, submission_prefix=None):
        """Per image a list (per class) of (k, 5) [x1,y1,x2,y2,score] arrays -> KITTI annos with the 3D
        fields set to the devkit's "unknown" values."""
        assert len(net_outputs) == len(self.data_infos), 'invalid list length of network outputs'
        det_annos = []
        for i, per_class in enumerate(net_outputs):
            rows = [(class_names[int(l)], b) for l, arr in enumerate(per_class) for b in np.asarray(arr).reshape(-1, 5)]
            n = len(rows)
            if n == 0:
                anno = {k: v.copy() for k, v in _EMPTY_ANNO.items()}
            else:
                anno = dict(name=np.array([r[0] for r in rows]), truncated=np.zeros(n), occluded=np.zeros(n, np.int64),
                            alpha=np.zeros(n), bbox=np.stack([r[1][:4] for r in rows]),
                            dimensions=np.zeros((n, 3), np.float32), location=np.full((n, 3), -1000.0, np.float32),
                            rotation_y=np.zeros(n), score=np.array([r[1][4] for r in rows]))
            anno['sample_idx'] = np.array([self.data_infos[i]['image']['image_idx']] * n, dtype=np.int64)
            det_annos.append(anno)
        return det_annos

    def format_results(self, outputs, pklfile_prefix=None, submission_prefix=None):
        tmp_dir = None
        if pklfile_prefix is None:
            tmp_dir = tempfile.TemporaryDirectory()
            pklfile_prefix = os.path.join(tmp_dir.name, 'results')
        if not isinstance(outputs[0], dict):
            files = self.bbox2result_kitti2d(outputs, self.CLASSES, pklfile_prefix, submission_prefix)
        elif 'pts_bbox' in outputs[0] or 'img_bbox' in outputs[0]:
            files = dict()
            for name in outputs[0]:
                res = [out[name] for out in outputs]
                sub = submission_prefix + name if submission_prefix is not None else None
                fn = self.bbox2result_kitti2d if 'img' in name else self.bbox2result_kitti
                files[name] = fn(res, self.CLASSES, pklfile_prefix + name, sub)
        else:
            files = self.bbox2result_kitti(outputs, self.CLASSES, pklfile_prefix, submission_prefix)
        return files, tmp_dir

    def evaluate(self, results, *args, **kwargs):
        """kitti_dataset.py:320-375: SSL.simple_test results carry 'teacher' and 'student' entries (each
        optionally split into 'results_2d' / 'results_3d'); evaluate all and prefix the keys."""
        if isinstance(results[0], dict) and 'teacher' in results[0] and 'student' in results[0]:
            out = dict()
            for who, tag in (('teacher', 'tea'), ('student', 'stu')):
                res = [r[who] for r in results]
                if 'results_2d' in res[0] and 'results_3d' in res[0]:
                    both = dict()
                    both.update(add_prefix(self.evaluate([r['results_2d'] for r in res], *args, **kwargs), '2d'))
                    both.update(add_prefix(self.evaluate([r['results_3d'] for r in res], *args, **kwargs), '3d'))
                    out.update(add_prefix(both, tag))
                else:
                    out.update(add_prefix(self.evaluate(res, *args, **kwargs), tag))
            return out
        return self._evaluate(results, *args, **kwargs)

    def _evaluate(self, results, metric=None, logger=None, pklfile_prefix=None, submission_prefix=None,
                  show=False, out_dir=None, pipeline=None):
        from .kitti_eval import kitti_eval
        files, tmp_dir = self.format_results(results, pklfile_prefix)
        gt_annos = [info['annos'] for info in self.data_infos]
        if isinstance(files, dict):
            ap_dict = dict()
            for name, f in files.items():
                text, ap = kitti_eval(gt_annos, f, self.CLASSES, eval_types=['bbox'] if 'img' in name else
                                      ['bbox', 'bev', '3d'])
                for k, v in ap.items():
                    ap_dict['%s/%s' % (name, k)] = float('{:.4f}'.format(v))
                _log('Results of %s:\n%s' % (name, text), logger)
        else:
            text, ap_dict = kitti_eval(gt_annos, files, self.CLASSES,
                                       eval_types=['bbox'] if metric == 'img_bbox' else ['bbox', 'bev', '3d'])
            _log('\n' + text, logger)
        if tmp_dir is not None:
            tmp_dir.cleanup()
        return ap_dict


def _log(msg, logger):
    if logger is not None and hasattr(logger, 'info'):
        logger.info(msg)
    else:
        print(msg)
