"""Seeded synthetic KITTI-/Waymo-shaped frames (SURVEY.md §8(d) "Synthetic inputs").

A tiny ray-cast LiDAR simulator: sensor at the origin, ground plane, two facade
planes and a handful of oriented cuboids (Car / Pedestrian / Cyclist sizes).
Uniform-random points are NOT acceptable for this path (they give ~1 rulebook
pair per voxel); ray-cast surfaces reproduce the sub-manifold neighbour counts
of real scans (≈4 / 8.6 / 10.5 / 10.5 pairs per voxel at strides 1/2/4/8).

Everything here is numpy on the host; it is data generation, not the hot path.
"""
import numpy as np

KITTI_RANGE = (0.0, -40.0, -3.0, 70.4, 40.0, 1.0)
KITTI_VOXEL = (0.05, 0.05, 0.1)
WAYMO_RANGE = (-75.2, -75.2, -2.0, 75.2, 75.2, 4.0)
WAYMO_VOXEL = (0.1, 0.1, 0.15)

# (l, w, h) along the box's own x / y / z
CLASS_SIZES = np.array([[3.9, 1.6, 1.56],    # Car
                        [0.8, 0.6, 1.73],    # Pedestrian
                        [1.76, 0.6, 1.73]],  # Cyclist
                       dtype=np.float64)

# fixed KITTI-like projection P2 @ R0_rect @ Tr_velo_to_cam (4x4)
_P2 = np.array([[721.5377, 0.0, 609.5593, 44.85728],
                [0.0, 721.5377, 172.854, 0.2163791],
                [0.0, 0.0, 1.0, 0.002745884],
                [0.0, 0.0, 0.0, 1.0]])
_R0 = np.array([[0.9999239, 0.00983776, -0.00744505, 0.0],
                [-0.0098698, 0.9999421, -0.00427846, 0.0],
                [0.00740253, 0.00435161, 0.9999631, 0.0],
                [0.0, 0.0, 0.0, 1.0]])
_TR = np.array([[0.00753374, -0.9999714, -0.00061660, -0.00406977],
                [0.01480249, 0.00072807, -0.9998902, -0.07631618],
                [0.9998621, 0.00752379, 0.01480755, -0.2717806],
                [0.0, 0.0, 0.0, 1.0]])
KITTI_LIDAR2IMG = (_P2 @ _R0 @ _TR).astype(np.float32)


def _ray_obb(dirs, center, size, yaw):
    """Slab-method ray / oriented-box intersection; rays start at the origin.

    Returns the entry distance t (inf where the ray misses)."""
    c, s = np.cos(yaw), np.sin(yaw)
    # world -> box frame (rotation by -yaw about z)
    ox = -(center[0] * c + center[1] * s)
    oy = -(-center[0] * s + center[1] * c)
    oz = -center[2]
    dx = dirs[:, 0] * c + dirs[:, 1] * s
    dy = -dirs[:, 0] * s + dirs[:, 1] * c
    dz = dirs[:, 2]
    tmin = np.full(dirs.shape[0], -np.inf)
    tmax = np.full(dirs.shape[0], np.inf)
    for o, d, h in ((ox, dx, size[0] / 2), (oy, dy, size[1] / 2),
                    (oz, dz, size[2] / 2)):
        with np.errstate(divide='ignore', invalid='ignore'):
            inv = 1.0 / d
            t0 = (-h - o) * inv
            t1 = (h - o) * inv
        lo = np.minimum(t0, t1)
        hi = np.maximum(t0, t1)
        par = np.abs(d) < 1e-12
        lo = np.where(par, np.where(np.abs(o) <= h, -np.inf, np.inf), lo)
        hi = np.where(par, np.where(np.abs(o) <= h, np.inf, -np.inf), hi)
        tmin = np.maximum(tmin, lo)
        tmax = np.minimum(tmax, hi)
    hit = (tmax >= tmin) & (tmax > 0) & (tmin > 0)
    return np.where(hit, tmin, np.inf)


def lidar_frame(seed=0, full360=False, az_step_deg=None, n_boxes=None,
                pc_range=None):
    """One synthetic frame.

    Returns dict(points (N,4) f32 [x,y,z,intensity], gt_boxes (G,7) f32 in the
    mmdet3d LiDAR convention [x,y,z_bottom... no: gravity-centre x,y,z,l,w,h,yaw],
    gt_labels (G,) i64)."""
    rng = np.random.default_rng(seed)
    if pc_range is None:
        pc_range = WAYMO_RANGE if full360 else KITTI_RANGE
    if az_step_deg is None:
        az_step_deg = 0.113 if full360 else 0.28
    if n_boxes is None:
        n_boxes = 40 if full360 else 12
    ground_z = -1.73
    elev = np.deg2rad(np.linspace(2.0, -24.8, 64))
    if full360:
        az = np.deg2rad(np.arange(-180.0, 180.0, az_step_deg))
    else:
        az = np.deg2rad(np.arange(-45.0, 45.0, az_step_deg))
    ee, aa = np.meshgrid(elev, az, indexing='ij')
    dirs = np.stack([np.cos(ee) * np.cos(aa), np.cos(ee) * np.sin(aa),
                     np.sin(ee)], axis=-1).reshape(-1, 3)
    n_rays = dirs.shape[0]
    t = np.full(n_rays, np.inf)
    # ground
    with np.errstate(divide='ignore'):
        tg = ground_z / dirs[:, 2]
    tg = np.where((dirs[:, 2] < 0) & (tg > 0), tg, np.inf)
    t = np.minimum(t, tg)
    # two facades y = +a, y = -b, kept for ground_z < z < 3
    for sign in (1.0, -1.0):
        yw = sign * rng.uniform(9.0, 16.0)
        with np.errstate(divide='ignore', invalid='ignore'):
            tf = yw / dirs[:, 1]
        z = tf * dirs[:, 2]
        ok = (tf > 0) & np.isfinite(tf) & (z > ground_z) & (z < 3.0)
        t = np.minimum(t, np.where(ok, tf, np.inf))
    # cuboids
    labels = rng.integers(0, 3, size=n_boxes)
    rmax = 55.0
    rr = rng.uniform(6.0, rmax, size=n_boxes)
    if full360:
        bearing = rng.uniform(-np.pi, np.pi, size=n_boxes)
    else:
        bearing = rng.uniform(-0.6, 0.6, size=n_boxes)
    yaw = rng.uniform(-np.pi, np.pi, size=n_boxes)
    boxes = np.zeros((n_boxes, 7), dtype=np.float64)
    for b in range(n_boxes):
        size = CLASS_SIZES[labels[b]]
        ctr = np.array([rr[b] * np.cos(bearing[b]), rr[b] * np.sin(bearing[b]),
                        ground_z + size[2] / 2])
        boxes[b] = [ctr[0], ctr[1], ctr[2], size[0], size[1], size[2], yaw[b]]
        t = np.minimum(t, _ray_obb(dirs, ctr, size, yaw[b]))
    hit = np.isfinite(t) & (t < 75.0 * (1.5 if full360 else 1.0))
    pts = dirs[hit] * t[hit, None]
    pts = pts + rng.normal(0.0, 0.01, size=pts.shape)
    keep = ((pts[:, 0] >= pc_range[0]) & (pts[:, 0] < pc_range[3]) &
            (pts[:, 1] >= pc_range[1]) & (pts[:, 1] < pc_range[4]) &
            (pts[:, 2] >= pc_range[2]) & (pts[:, 2] < pc_range[5]))
    pts = pts[keep]
    inten = rng.uniform(0.0, 1.0, size=(pts.shape[0], 1))
    pts = np.concatenate([pts, inten], axis=1).astype(np.float32)
    perm = rng.permutation(pts.shape[0])
    pts = np.ascontiguousarray(pts[perm])
    return dict(points=pts, gt_boxes=boxes.astype(np.float32),
                gt_labels=labels.astype(np.int64))


def kitti_batch(batch_size, seed=0):
    """`batch_size` KITTI-shaped frames with seeds seed, seed+1, …"""
    return [lidar_frame(seed + i) for i in range(batch_size)]


def kitti_image(seed=0, shape=(384, 1280)):
    """Seeded uint8 noise image already resized/padded to a /32 shape."""
    rng = np.random.default_rng(10_000 + seed)
    return rng.integers(0, 256, size=(shape[0], shape[1], 3), dtype=np.uint8)


# class order of configs/detmatch: ['Pedestrian', 'Cyclist', 'Car']; simulator: Car, Ped, Cyc
_SIM_TO_CFG_LABEL = np.array([2, 0, 1], dtype=np.int64)


def frame_to_mm3d_gt(frame):
    """Simulator GT (pcdet convention: gravity centre, dx along heading) -> the mmdet3d LiDAR
    convention the detectors take (bottom centre, (x, y, z, w, l, h, yaw), the inverse of
    mmdet3d/models/detectors/openpcdet.py:100-122) and config-order labels."""
    b = frame['gt_boxes'].astype(np.float32)
    out = np.stack([b[:, 0], b[:, 1], b[:, 2] - b[:, 5] / 2, b[:, 4], b[:, 3], b[:, 5],
                    -b[:, 6] - np.pi / 2], axis=1).astype(np.float32)
    return out, _SIM_TO_CFG_LABEL[frame['gt_labels']]


# ---------------------------------------------------------------------------------------------
# DetMatch iteration inputs (SURVEY §8(b) "img_metas semantics", §8(d) "Batches")
# ---------------------------------------------------------------------------------------------
KITTI_ORI_SHAPE = (375, 1242, 3)
KITTI_IMG_SHAPE = (384, 1280, 3)       # after Resize (keep_ratio) + Pad(32) in the synthetic setup
IMG_MEAN_BGR = np.array([103.530, 116.280, 123.675], dtype=np.float32)   # caffe-style, std 1


WAYMO_ORI_SHAPE = (1280, 1920, 3)
# Waymo-shaped camera: the KITTI-like projection with its intrinsics scaled to a 1280 x 1920 sensor
_SCALE_W = np.diag([1920.0 / 1242.0, 1280.0 / 375.0, 1.0, 1.0]).astype(np.float32)
PROFILES = {
    'kitti': dict(pc_range=KITTI_RANGE, full360=False, ori_shape=KITTI_ORI_SHAPE, img_shape=KITTI_IMG_SHAPE,
                  lidar2img=KITTI_LIDAR2IMG),
    # BASELINE.json configs[4]: Waymo-SHAPED synthetic (no reference config exists): full
    # 360 deg sweep of ~200 k points, range +-75.2 m x [-2, 4), 1280 x 1920 image
    'waymo': dict(pc_range=WAYMO_RANGE, full360=True, ori_shape=WAYMO_ORI_SHAPE, img_shape=WAYMO_ORI_SHAPE,
                  lidar2img=(_SCALE_W @ KITTI_LIDAR2IMG).astype(np.float32)),
}


def _in_range(pts, pc_range):
    return ((pts[:, 0] > pc_range[0]) & (pts[:, 1] > pc_range[1]) & (pts[:, 2] > pc_range[2]) &
            (pts[:, 0] < pc_range[3]) & (pts[:, 1] < pc_range[4]) & (pts[:, 2] < pc_range[5]))


def ssl_sample(seed, labeled, device='cpu', with_img=True, profile='kitti'):
    """One TS_SSL_Dataset item (teacher_student_ssl_dataset.py:26-33): the shared pipeline (Resize,
    RandomFlip3D with sync_2d) runs once, then the student pipeline (GlobalRotScaleTrans,
    range filters, shuffle) and the teacher pipeline (range filter, shuffle) on copies.

    Returns (stu, tea) dicts of torch tensors / LiDARInstance3DBoxes with the img_metas keys the
    SSL modules consume.  Image augmentations that do not move boxes (colour jitter, erasing) are
    represented by independent noise in the student image."""
    import torch
    from .mm3d.bbox_utils import bbox_2d_transform, bbox_3d_to_bbox_2d
    from .mm3d.box3d import LiDARInstance3DBoxes
    prof = PROFILES[profile]
    KITTI_RANGE, KITTI_ORI_SHAPE, KITTI_IMG_SHAPE, KITTI_LIDAR2IMG = (
        prof['pc_range'], prof['ori_shape'], prof['img_shape'], prof['lidar2img'])   # shadow the module constants
    rng = np.random.default_rng(77_000 + seed)
    frame = lidar_frame(seed, full360=prof['full360'])
    pts = frame['points'].copy()
    boxes_np, labels_np = frame_to_mm3d_gt(frame)
    boxes = LiDARInstance3DBoxes(torch.from_numpy(boxes_np))
    labels = torch.from_numpy(labels_np)
    flip = bool(rng.random() < 0.5)
    sx, sy = KITTI_IMG_SHAPE[1] / KITTI_ORI_SHAPE[1], KITTI_IMG_SHAPE[0] / KITTI_ORI_SHAPE[0]
    meta = dict(sample_idx=int(seed), lidar2img=KITTI_LIDAR2IMG.copy(), ori_shape=KITTI_ORI_SHAPE,
                img_shape=KITTI_IMG_SHAPE, pad_shape=KITTI_IMG_SHAPE,
                scale_factor=np.array([sx, sy, sx, sy], dtype=np.float32), flip=flip,
                pcd_horizontal_flip=flip, pcd_vertical_flip=False, box_type_3d=LiDARInstance3DBoxes,
                transformation_3d_flow=['HF'] if flip else [], pcd_trans=np.zeros(3, np.float32),
                pcd_scale_factor=1.0, pcd_rotation=np.eye(3, dtype=np.float32))
    # ---- 2D GT: projected 3D GT on the original image, then Resize + flip -------------------
    gt2d = gt2d_labels = None
    if labeled:
        b2d, valid = bbox_3d_to_bbox_2d(boxes, KITTI_LIDAR2IMG, KITTI_ORI_SHAPE)
        wh_ok = ((b2d[:, 2] - b2d[:, 0]) > 2) & ((b2d[:, 3] - b2d[:, 1]) > 2)
        keep = valid & wh_ok
        gt2d = bbox_2d_transform(meta, b2d[keep], True)
        gt2d_labels = labels[keep]
    # ---- shared flip -------------------------------------------------------------------------
    if flip:
        pts[:, 1] = -pts[:, 1]
        boxes.flip('horizontal')
    tea_pts = pts[_in_range(pts, KITTI_RANGE)]
    tea_pts = tea_pts[rng.permutation(len(tea_pts))]
    # ---- student: GlobalRotScaleTrans(rot ±pi/4, scale .95-1.05, no translation) ------------
    ang = float(rng.uniform(-0.78539816, 0.78539816))
    c, s = np.cos(ang), np.sin(ang)
    if labeled and len(boxes_np):      # transforms_3d.py:611-616: the box matrix (not transposed)
        rot = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]], dtype=np.float32)
    else:                              # :604-607: points.rotate(angle) returns the transposed one
        rot = np.array([[c, s, 0], [-s, c, 0], [0, 0, 1]], dtype=np.float32)
    scale = float(rng.uniform(0.95, 1.05))
    stu_pts = pts.copy()
    stu_pts[:, :3] = (stu_pts[:, :3] @ rot) * scale
    stu_meta = dict(meta)
    stu_meta.update(pcd_rotation=rot, pcd_scale_factor=scale,
                    transformation_3d_flow=list(meta['transformation_3d_flow']) + ['R', 'S', 'T'])
    stu_boxes = boxes.clone()
    stu_boxes.rotate(torch.from_numpy(rot))
    stu_boxes.scale(scale)
    stu_pts = stu_pts[_in_range(stu_pts, KITTI_RANGE)]
    stu_pts = stu_pts[rng.permutation(len(stu_pts))]
    stu = dict(points=torch.from_numpy(stu_pts).to(device), img_metas=stu_meta)
    tea = dict(points=torch.from_numpy(tea_pts).to(device), img_metas=dict(meta))
    if labeled:
        bev = stu_boxes.bev
        inr = ((bev[:, 0] > KITTI_RANGE[0]) & (bev[:, 1] > KITTI_RANGE[1]) &
               (bev[:, 0] < KITTI_RANGE[3]) & (bev[:, 1] < KITTI_RANGE[4]))   # ObjectRangeFilter
        sb = stu_boxes[inr]
        sb.limit_yaw(offset=0.5, period=2 * np.pi)
        stu['gt_bboxes_3d'] = sb.to(device)
        stu['gt_labels_3d'] = labels[inr].to(device)
        stu['gt_bboxes'] = gt2d.to(device)
        stu['gt_labels'] = gt2d_labels.to(device)
    if with_img:
        base = kitti_image(seed, KITTI_IMG_SHAPE[:2]).astype(np.float32)
        if flip:
            base = base[:, ::-1]
        tea_img = (base - IMG_MEAN_BGR).transpose(2, 0, 1)
        noise = rng.normal(0, 8, size=base.shape).astype(np.float32)
        stu_img = (np.clip(base + noise, 0, 255) - IMG_MEAN_BGR).transpose(2, 0, 1)
        tea['img'] = torch.from_numpy(np.ascontiguousarray(tea_img)).to(device)
        stu['img'] = torch.from_numpy(np.ascontiguousarray(stu_img)).to(device)
    return stu, tea


def _collate(samples):
    import torch
    out = dict()
    for k in samples[0].keys():
        vals = [s[k] for s in samples]
        out[k] = torch.stack(vals, dim=0) if k == 'img' else vals
    return out


def _device_points(samples, seeds, device, generator=None):
    """Replace the host-augmented point clouds of `samples` [(stu, tea), ...] by the output of the
    device pipeline (pipeline3d.augment_points): the raw frames go to HBM once and every student /
    teacher view replays its recorded img_metas there (flip, rotation, scale, range filter, shuffle)."""
    import torch
    from .pipeline3d import View3D, augment_points
    raw = [torch.from_numpy(lidar_frame(s)['points']).to(device) for s in seeds]
    views = []
    for i, (stu, tea) in enumerate(samples):
        views += [View3D.from_meta(stu['img_metas'], i, KITTI_RANGE, shuffle=True),
                  View3D.from_meta(tea['img_metas'], i, KITTI_RANGE, shuffle=True)]
    outs = augment_points(raw, views, generator)
    for i, (stu, tea) in enumerate(samples):
        stu['points'], tea['points'] = outs[2 * i], outs[2 * i + 1]


def ssl_batch(batch_size, seed=0, device='cpu', with_img=True, device_pipeline=False, profile='kitti'):
    """The data_batch IterBasedSSLRunner.train assembles (iter_based_ssl_runner.py:21-27): `batch_size`
    labeled + `batch_size` unlabeled samples, keys lab_stu / lab_tea / unlab_stu / unlab_tea.
    device_pipeline=True: the point clouds are augmented on the GPU (one dm_points_augment call for all
    4 * batch_size views) instead of in numpy."""
    lab = [ssl_sample(seed + i, True, device, with_img, profile) for i in range(batch_size)]
    unlab = [ssl_sample(seed + 1000 + i, False, device, with_img, profile) for i in range(batch_size)]
    if device_pipeline:
        assert profile == 'kitti'
        _device_points(lab + unlab, [seed + i for i in range(batch_size)] +
                       [seed + 1000 + i for i in range(batch_size)], device)
    return dict(lab_stu=_collate([s for s, _ in lab]), lab_tea=_collate([t for _, t in lab]),
                unlab_stu=_collate([s for s, _ in unlab]), unlab_tea=_collate([t for _, t in unlab]),
                img_metas=[s['img_metas'] for s, _ in lab])
