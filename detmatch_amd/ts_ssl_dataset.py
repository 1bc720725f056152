"""TS_SSL_Dataset batches from KITTI files, assembled on the device.

What `mmdet3d/datasets/teacher_student_ssl_dataset.py:26-33` + the pipelines of
configs/detmatch/001/detmatch/split_0.py:552-745 produce per sample in the loader workers, here per
BATCH on the GPU: the raw frame (points `.bin`, image `.png`, annotations) is uploaded once; the 3D side
runs through detmatch_amd/pipeline3d.py (one dm_points_augment call for all student + teacher views),
the 2D side (Resize with keep_ratio over a scale range, horizontal flip synchronised with the 3D flip,
Normalize, Pad to a multiple of 32) with torch ops on the device.

`TSSSLDeviceLoader` yields what `IterBasedSSLRunner.train` takes from a mmcv DataLoader:
dict(stu=..., tea=..., img_metas=...) with `points` / `gt_bboxes_3d` / `gt_labels_3d` / `gt_bboxes` /
`gt_labels` lists, a stacked `img` tensor and the img_metas keys the SSL modules replay.

Not restated (documented gaps): `ObjectSample` GT-paste (needs the db-info crops and numba collision
tests), the student-only photometric augmentations (torchvision ColorJitter / Grayscale / GaussianBlur /
RandomErasing — they do not move boxes), mmcv's cv2 interpolation (torch bilinear here).  mmdet / mmcv
transform semantics are un-vendored: parity unpinned for the 2D side."""
import numpy as np
import torch
import torch.nn.functional as F

from . import pipeline3d as P3
from .mm3d.box3d import LiDARInstance3DBoxes


class ImageResizeFlipNormPad(object):
    """mmdet Resize(img_scale=[(640,192),(2560,768)], multiscale_mode='range', keep_ratio=True) ->
    RandomFlip (decision shared with the 3D flip) -> Normalize(mean, std, to_rgb=False) ->
    Pad(size_divisor=32), for one image tensor (H, W, 3) uint8/float BGR on the device."""

    def __init__(self, img_scale=((640, 192), (2560, 768)), mean=(103.530, 116.280, 123.675),
                 std=(1.0, 1.0, 1.0), size_divisor=32):
        self.img_scale = [tuple(s) for s in img_scale]
        self.mean, self.std, self.size_divisor = mean, std, size_divisor

    def draw_scale(self, rng):
        """mmdet Resize.random_sample: long and short edge drawn independently, inclusive."""
        longs = [max(s) for s in self.img_scale]
        shorts = [min(s) for s in self.img_scale]
        return (int(rng.randint(min(longs), max(longs) + 1)), int(rng.randint(min(shorts), max(shorts) + 1)))

    def __call__(self, img, scale, flip):
        h, w = int(img.shape[0]), int(img.shape[1])
        k = min(max(scale) / max(h, w), min(scale) / min(h, w))            # mmcv.rescale_size
        nw, nh = int(w * float(k) + 0.5), int(h * float(k) + 0.5)
        x = img.permute(2, 0, 1).float().unsqueeze(0)
        x = F.interpolate(x, size=(nh, nw), mode='bilinear', align_corners=False)
        if flip:
            x = x.flip(-1)
        x = (x - x.new_tensor(self.mean).view(1, 3, 1, 1)) / x.new_tensor(self.std).view(1, 3, 1, 1)
        d = self.size_divisor
        ph, pw = (nh + d - 1) // d * d, (nw + d - 1) // d * d
        x = F.pad(x, (0, pw - nw, 0, ph - nh))
        meta = dict(ori_shape=(h, w, 3), img_shape=(nh, nw, 3), pad_shape=(ph, pw, 3),
                    scale_factor=np.array([nw / w, nh / h, nw / w, nh / h], dtype=np.float32), flip=bool(flip),
                    flip_direction='horizontal' if flip else None,
                    img_norm_cfg=dict(mean=np.array(self.mean, np.float32), std=np.array(self.std, np.float32),
                                      to_rgb=False))
        return x[0], meta

    @staticmethod
    def boxes(bboxes, meta):
        """Resize._resize_bboxes + RandomFlip.bbox_flip on (n,4) xyxy in original-image pixels."""
        b = bboxes * bboxes.new_tensor(meta['scale_factor'])
        h, w = meta['img_shape'][:2]
        b = torch.stack([b[:, 0].clamp(0, w), b[:, 1].clamp(0, h), b[:, 2].clamp(0, w), b[:, 3].clamp(0, h)], 1)
        if meta['flip']:
            b = torch.stack([w - b[:, 2], b[:, 1], w - b[:, 0], b[:, 3]], 1)
        return b


class TSSSLDeviceLoader(object):
    """One of the two loaders `IterBasedSSLRunner.run([labeled, unlabeled])` takes."""

    def __init__(self, dataset, samples_per_gpu, device, labeled, point_cloud_range, seed=0,
                 rot_range=(-0.78539816, 0.78539816), scale_ratio_range=(0.95, 1.05), flip_ratio=0.5,
                 img_scale=((640, 192), (2560, 768)), shuffle=True, with_img=True):
        self.dataset, self.bs, self.device, self.labeled = dataset, samples_per_gpu, torch.device(device), labeled
        self.rng = np.random.RandomState(seed)
        self.shuffle, self.with_img = shuffle, with_img
        self.flip_ratio = flip_ratio
        self.image_tf = ImageResizeFlipNormPad(img_scale)
        self.pipe = P3.TSSSLPipeline3D(
            shared=[P3.RandomFlip3D(sync_2d=True, flip_ratio_bev_horizontal=flip_ratio)],
            student=[P3.GlobalRotScaleTrans(rot_range=list(rot_range), scale_ratio_range=list(scale_ratio_range)),
                     P3.PointsRangeFilter(point_cloud_range), P3.PointShuffle()],
            teacher=[P3.PointsRangeFilter(point_cloud_range), P3.PointShuffle()],
            object_range=point_cloud_range if labeled else None)
        self.sampler = None

    def __len__(self):
        return max(len(self.dataset) // self.bs, 1)

    def _indices(self):
        n = len(self.dataset)
        order = self.rng.permutation(n) if self.shuffle else np.arange(n)
        if n < self.bs:                                   # tiny sets (tests): repeat
            order = np.resize(order, self.bs)
        return [order[i:i + self.bs] for i in range(0, len(order) - self.bs + 1, self.bs)]

    def __iter__(self):
        for idx in self._indices():
            yield self.batch([int(i) for i in idx])

    def batch(self, indices):
        dev, ds = self.device, self.dataset
        frames, metas2d, imgs = [], [], []
        for i in indices:
            info = ds.get_data_info(i)
            flip = bool(self.rng.rand() < self.flip_ratio)                 # mmdet RandomFlip's draw
            meta = dict(sample_idx=info['sample_idx'], lidar2img=info['lidar2img'], flip=flip,
                        box_type_3d=LiDARInstance3DBoxes)
            if self.with_img:
                raw = torch.from_numpy(ds.load_image(i)).to(dev)
                img, m2 = self.image_tf(raw, self.image_tf.draw_scale(self.rng), flip)
                meta.update(m2)
                imgs.append(img)
            f = dict(points=torch.from_numpy(ds.load_points(i)).to(dev), meta=meta)
            if self.labeled:
                ann = info['ann_info']
                f['gt_bboxes_3d'] = ann['gt_bboxes_3d']
                f['gt_labels_3d'] = torch.from_numpy(ann['gt_labels_3d'])
                f['bboxes'], f['labels'] = torch.from_numpy(ann['bboxes']), torch.from_numpy(ann['labels'])
            frames.append(f)
        stu, tea = self.pipe(frames, self.rng)
        out_s = dict(points=[s['points'] for s in stu], img_metas=[s['img_metas'] for s in stu])
        out_t = dict(points=[t['points'] for t in tea], img_metas=[t['img_metas'] for t in tea])
        if self.labeled:
            out_s['gt_bboxes_3d'] = [s['gt_bboxes_3d'].to(dev) for s in stu]
            out_s['gt_labels_3d'] = [s['gt_labels_3d'].to(dev) for s in stu]
            if self.with_img:
                out_s['gt_bboxes'] = [self.image_tf.boxes(f['bboxes'], f['meta']).to(dev) for f in frames]
                out_s['gt_labels'] = [f['labels'].to(dev) for f in frames]
        if self.with_img:
            ph = max(int(i.shape[1]) for i in imgs)
            pw = max(int(i.shape[2]) for i in imgs)
            batch = torch.stack([F.pad(i, (0, pw - i.shape[2], 0, ph - i.shape[1])) for i in imgs])
            out_s['img'], out_t['img'] = batch, batch            # no photometric student augmentation (see above)
            for s, t in zip(out_s['img_metas'], out_t['img_metas']):
                s['pad_shape'] = t['pad_shape'] = (ph, pw, 3)
        return dict(stu=out_s, tea=out_t, img_metas=out_s['img_metas'])
