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

The student-only photometric chain of split_0.py:586-625 (torchvision ColorJitter p .8, RandomGrayscale
p .2, SimCLR GaussianBlur p .5, three RandomErasing with random fill) is `StudentPhotometric`: the same
parameter draws, evaluated with torch ops on the device (true Gaussian kernel instead of PIL's box
approximation; erased patches filled with clamped noise).

Not restated (documented gap): `ObjectSample` GT-paste (needs the db-info crops and numba collision
tests).  mmdet / mmcv / torchvision transform semantics are un-vendored: parity unpinned for the 2D side."""
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

    def __call__(self, img, scale, flip, photometric=None):
        h, w = int(img.shape[0]), int(img.shape[1])
        k = min(max(scale) / max(h, w), min(scale) / min(h, w))            # mmcv.rescale_size
        nw, nh = int(w * float(k) + 0.5), int(h * float(k) + 0.5)
        x = img.permute(2, 0, 1).float().unsqueeze(0)
        x = F.interpolate(x, size=(nh, nw), mode='bilinear', align_corners=False)
        if flip:
            x = x.flip(-1)
        if photometric is not None:        # student only: after the shared Resize / flip, before Normalize
            x = photometric(x[0]).unsqueeze(0)
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


class StudentPhotometric(object):
    """split_0.py:586-625 on a (3, H, W) float image in [0, 255] on the device (channel order as stored:
    the reference hands mmcv's BGR array to torchvision as if it were RGB, so the grey weights hit the
    same stored channels here).  Formulas: torchvision.transforms.functional (adjust_brightness /
    contrast / saturation / hue, rgb_to_grayscale, RandomErasing.get_params)."""

    GREY = (0.299, 0.587, 0.114)

    def __init__(self, jitter=(0.4, 0.4, 0.4, 0.1), p_jitter=0.8, p_grey=0.2, blur_sigma=(0.1, 2.0), p_blur=0.5,
                 erasing=((0.7, (0.05, 0.2), (0.3, 3.3)), (0.5, (0.02, 0.2), (0.1, 6)), (0.3, (0.02, 0.2), (0.05, 8)))):
        self.jitter, self.p_jitter, self.p_grey = jitter, p_jitter, p_grey
        self.blur_sigma, self.p_blur, self.erasing = blur_sigma, p_blur, erasing

    def _grey(self, x):
        w = x.new_tensor(self.GREY).view(3, 1, 1)
        return (x * w).sum(0, keepdim=True)

    @staticmethod
    def _hue(x, shift):
        """adjust_hue: RGB -> HSV, h += shift (mod 1), back; x in [0, 1]."""
        r, g, b = x[0], x[1], x[2]
        maxc, minc = x.max(0)[0], x.min(0)[0]
        v = maxc
        cr = maxc - minc
        s = cr / torch.where(maxc == 0, torch.ones_like(maxc), maxc)
        crd = torch.where(cr == 0, torch.ones_like(cr), cr)
        rc, gc, bc = (maxc - r) / crd, (maxc - g) / crd, (maxc - b) / crd
        h = torch.where(maxc == r, bc - gc, torch.where(maxc == g, 2.0 + rc - bc, 4.0 + gc - rc))
        h = torch.where(cr == 0, torch.zeros_like(h), h)
        h = ((h / 6.0 + 1.0) % 1.0 + shift) % 1.0
        i = torch.floor(h * 6.0)
        f = h * 6.0 - i
        i = i.long() % 6
        p, q, t = v * (1 - s), v * (1 - f * s), v * (1 - (1 - f) * s)
        sel = lambda opts: torch.stack(opts, 0).gather(0, i.unsqueeze(0)).squeeze(0)
        return torch.stack([sel([v, q, p, p, t, v]), sel([t, v, v, q, p, p]), sel([p, p, t, v, v, q])], 0)

    def __call__(self, img, rng, generator=None):
        x = img / 255.0
        if rng.rand() < self.p_jitter:
            b, c, s, h = self.jitter
            fb, fc = rng.uniform(1 - b, 1 + b), rng.uniform(1 - c, 1 + c)
            fs, fh = rng.uniform(1 - s, 1 + s), rng.uniform(-h, h)
            for op in rng.permutation(4):
                if op == 0:
                    x = (x * fb).clamp(0, 1)
                elif op == 1:
                    x = (fc * x + (1 - fc) * self._grey(x).mean()).clamp(0, 1)
                elif op == 2:
                    x = (fs * x + (1 - fs) * self._grey(x)).clamp(0, 1)
                else:
                    x = self._hue(x, fh)
        if rng.rand() < self.p_grey:
            x = self._grey(x).expand(3, -1, -1).contiguous()
        if rng.rand() < self.p_blur:
            sigma = rng.uniform(*self.blur_sigma)
            r = max(int(3 * sigma + 0.5), 1)
            k = torch.exp(-0.5 * (torch.arange(-r, r + 1, device=x.device, dtype=x.dtype) / sigma) ** 2)
            k = k / k.sum()
            y = F.pad(x.unsqueeze(0), (r, r, r, r), mode='replicate')
            y = F.conv2d(y, k.view(1, 1, 1, -1).expand(3, 1, 1, -1), groups=3)
            x = F.conv2d(y, k.view(1, 1, -1, 1).expand(3, 1, -1, 1), groups=3)[0]
        hh, ww = x.shape[1], x.shape[2]
        for p, scale, ratio in self.erasing:
            if rng.rand() >= p:
                continue
            for _ in range(10):
                area = hh * ww * rng.uniform(*scale)
                aspect = float(np.exp(rng.uniform(np.log(ratio[0]), np.log(ratio[1]))))
                eh, ew = int(round(np.sqrt(area * aspect))), int(round(np.sqrt(area / aspect)))
                if eh < hh and ew < ww:
                    i, j = rng.randint(0, hh - eh + 1), rng.randint(0, ww - ew + 1)
                    noise = torch.randn((3, eh, ew), device=x.device, dtype=x.dtype, generator=generator)
                    x[:, i:i + eh, j:j + ew] = noise.clamp(0, 1)
                    break
        return (x * 255.0).round().clamp(0, 255)            # TVToPILImage: back to 8-bit levels


class TSSSLDeviceLoader(object):
    """One of the two loaders `IterBasedSSLRunner.run([labeled, unlabeled])` takes."""

    def __init__(self, dataset, samples_per_gpu, device, labeled, point_cloud_range, seed=0,
                 rot_range=(-0.78539816, 0.78539816), scale_ratio_range=(0.95, 1.05), flip_ratio=0.5,
                 img_scale=((640, 192), (2560, 768)), shuffle=True, with_img=True, student_photometric=True,
                 db_sampler=None, img_mean=(103.530, 116.280, 123.675), img_std=(1.0, 1.0, 1.0),
                 size_divisor=32, rank=0, world_size=1):
        self.dataset, self.bs, self.device, self.labeled = dataset, samples_per_gpu, torch.device(device), labeled
        # the sample ORDER is drawn from a seed shared by all ranks (+ epoch) and each rank takes every
        # world_size-th entry (mmdet DistributedGroupSampler's partition); augmentation draws use a
        # per-rank stream
        self.seed, self.rank, self.world_size, self.epoch = int(seed), int(rank), int(world_size), 0
        self.rng = np.random.RandomState((int(seed) * 1000003 + 7919 * int(rank)) % (2 ** 31))
        self.shuffle, self.with_img = shuffle, with_img
        self.flip_ratio = flip_ratio
        self.image_tf = ImageResizeFlipNormPad(img_scale, mean=img_mean, std=img_std, size_divisor=size_divisor)
        if isinstance(student_photometric, dict):
            self.photometric = StudentPhotometric(**student_photometric)
        else:
            self.photometric = StudentPhotometric() if student_photometric else None
        # ObjectSample(db_sampler) of the labeled shared pipeline (split_0.py:570): GT-paste before the flip
        self.object_sample = None
        if db_sampler is not None and labeled:
            from .dbsampler import ObjectSample
            if isinstance(db_sampler, dict):
                db_sampler = dict(db_sampler, device=self.device, rng=self.rng)
            self.object_sample = db_sampler if isinstance(db_sampler, ObjectSample) else ObjectSample(db_sampler)
        self.pipe = P3.TSSSLPipeline3D(
            shared=[P3.RandomFlip3D(sync_2d=True, flip_ratio_bev_horizontal=flip_ratio)],
            student=[P3.GlobalRotScaleTrans(rot_range=list(rot_range), scale_ratio_range=list(scale_ratio_range)),
                     P3.PointsRangeFilter(point_cloud_range), P3.PointShuffle()],
            teacher=[P3.PointsRangeFilter(point_cloud_range), P3.PointShuffle()],
            object_range=point_cloud_range if labeled else None)
        self.sampler = self          # IterLoader calls loader.sampler.set_epoch(epoch) (mmcv)

    def __len__(self):
        return max(len(self.dataset) // (self.bs * self.world_size), 1)

    def set_epoch(self, epoch):
        """IterLoader calls this when the loader is exhausted: a new shared permutation per epoch."""
        self.epoch = int(epoch)

    def _indices(self):
        n = len(self.dataset)
        order_rng = np.random.RandomState((self.seed + self.epoch) % (2 ** 31))
        order = order_rng.permutation(n) if self.shuffle else np.arange(n)
        need = self.bs * self.world_size
        if len(order) < need:                             # tiny sets (tests): repeat
            order = np.resize(order, need)
        order = order[:len(order) // need * need][self.rank::self.world_size]
        return [order[i:i + self.bs] for i in range(0, len(order) - self.bs + 1, self.bs)]

    def __iter__(self):
        for idx in self._indices():
            yield self.batch([int(i) for i in idx])

    def batch(self, indices):
        dev, ds = self.device, self.dataset
        frames, imgs, imgs_stu = [], [], []
        for i in indices:
            info = ds.get_data_info(i)
            flip = bool(self.rng.rand() < self.flip_ratio)                 # mmdet RandomFlip's draw
            meta = dict(sample_idx=info['sample_idx'], lidar2img=info['lidar2img'], flip=flip,
                        box_type_3d=LiDARInstance3DBoxes)
            if self.with_img:
                raw = torch.from_numpy(ds.load_image(i)).to(dev)
                scale = self.image_tf.draw_scale(self.rng)
                img, m2 = self.image_tf(raw, scale, flip)
                meta.update(m2)
                imgs.append(img)
                if self.photometric is not None:
                    imgs_stu.append(self.image_tf(raw, scale, flip,
                                                  photometric=lambda x: self.photometric(x, self.rng))[0])
            f = dict(points=torch.from_numpy(ds.load_points(i)).to(dev), meta=meta)
            if self.labeled:
                ann = info['ann_info']
                f['gt_bboxes_3d'] = ann['gt_bboxes_3d']
                f['gt_labels_3d'] = torch.from_numpy(ann['gt_labels_3d'])
                f['bboxes'], f['labels'] = torch.from_numpy(ann['bboxes']), torch.from_numpy(ann['labels'])
                if self.object_sample is not None:       # pasted objects get no 2D box (sample_2d=False)
                    calib = ds.data_infos[i].get('calib')
                    if calib is not None:
                        f['calib'] = calib
                    if 'road_plane' in ds.data_infos[i]:
                        f['road_plane'] = ds.data_infos[i]['road_plane']
                    f = self.object_sample(f)
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
            stack = lambda lst: torch.stack([F.pad(i, (0, pw - i.shape[2], 0, ph - i.shape[1])) for i in lst])
            out_t['img'] = stack(imgs)
            out_s['img'] = stack(imgs_stu) if self.photometric is not None else out_t['img']
            for s, t in zip(out_s['img_metas'], out_t['img_metas']):
                s['pad_shape'] = t['pad_shape'] = (ph, pw, 3)
        return dict(stu=out_s, tea=out_t, img_metas=out_s['img_metas'])
