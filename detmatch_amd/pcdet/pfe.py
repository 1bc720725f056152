"""VoxelSetAbstraction — pcdet/models/backbones_3d/pfe/voxel_set_abstraction.py:9-238.

Keypoint FPS, bilinear BEV sampling, multi-scale set abstraction over raw points and the
four sparse-conv stages, 640 -> 128 fusion.  Feature column order is
[bev(256), raw_points(32), x_conv1(32), x_conv2(64), x_conv3(128), x_conv4(128)]
(:181-230), which fixes the layout of vsa_point_feature_fusion.0.weight for checkpoints.

Host syncs of the reference removed: per-sample point counts come from the caller
(`batch_dict['points_batch_cnt_host']`), voxel counts per sample are device bincounts.
"""
import copy

import torch
import torch.nn as nn

from .. import pointnet2_stack as pn2
from ..bn_relu import fc_rows
from ..fused import on as fused_on
from ..devconst import upload
from .utils import get_voxel_centers


def batch_row_counts(batch_idx, batch_size):
    """Rows per sample of a batch-index column (torch.bincount reads its output size back: sync)."""
    ar = torch.arange(batch_size, device=batch_idx.device, dtype=batch_idx.dtype)
    return (batch_idx.view(-1, 1) == ar.view(1, -1)).sum(dim=0).int()


def voxel_centers_and_counts(coords, batch_size, downsample_times, voxel_size, point_cloud_range):
    """(xyz (N, 3), rows per sample (B) int32) of a sparse level's (N, 4) [b, z, y, x] coordinates: one launch
    (dm_voxel_centers) instead of get_voxel_centers' five and batch_row_counts' four."""
    if coords.is_cuda and coords.dtype == torch.int32 and coords.is_contiguous() and coords.shape[1] == 4:
        from .. import _lib
        n = coords.shape[0]
        xyz = torch.empty((n, 3), dtype=torch.float32, device=coords.device)
        counts = torch.empty((batch_size,), dtype=torch.int32, device=coords.device)
        v = [float(s) * downsample_times for s in voxel_size]
        r = [float(c) for c in point_cloud_range[0:3]]
        _lib.check(_lib.lib().dm_voxel_centers(coords.data_ptr(), n, int(batch_size), v[0], v[1], v[2], r[0], r[1], r[2],
                                               xyz.data_ptr(), counts.data_ptr(), _lib.raw_stream()), 'dm_voxel_centers')
        return xyz, counts
    xyz = get_voxel_centers(coords[:, 1:4], downsample_times=downsample_times, voxel_size=voxel_size,
                            point_cloud_range=point_cloud_range)
    return xyz.contiguous(), batch_row_counts(coords[:, 0], batch_size)


def bilinear_interpolate_torch(im, x, y):
    """voxel_set_abstraction.py:9-40: im (H, W, C), x / y (N) in pixel units, clamped."""
    x0 = torch.floor(x).long()
    x1 = x0 + 1
    y0 = torch.floor(y).long()
    y1 = y0 + 1
    x0 = torch.clamp(x0, 0, im.shape[1] - 1)
    x1 = torch.clamp(x1, 0, im.shape[1] - 1)
    y0 = torch.clamp(y0, 0, im.shape[0] - 1)
    y1 = torch.clamp(y1, 0, im.shape[0] - 1)
    Ia, Ib, Ic, Id = im[y0, x0], im[y1, x0], im[y0, x1], im[y1, x1]
    wa = (x1.type_as(x) - x) * (y1.type_as(y) - y)
    wb = (x1.type_as(x) - x) * (y - y0.type_as(y))
    wc = (x - x0.type_as(x)) * (y1.type_as(y) - y)
    wd = (x - x0.type_as(x)) * (y - y0.type_as(y))
    return Ia * wa[:, None] + Ib * wb[:, None] + Ic * wc[:, None] + Id * wd[:, None]


class _BevInterpolate(torch.autograd.Function):
    """dm_bev_interpolate_forward / _backward (csrc/bev_interp.hip): (B, C, H, W) channels_last map,
    key points (B, K, 3) -> (B, K, C); gradient w.r.t. the map only (key points are sampled points)."""

    @staticmethod
    def forward(ctx, bev, keypoints, geom):
        from .. import _lib, dense_conv
        im = dense_conv._cl(bev.detach())
        kp = keypoints.detach().float().contiguous()
        b, c, h, w = im.shape
        k = int(kp.shape[1])
        out = torch.empty((b, k, c), dtype=torch.float32, device=im.device)
        cells = torch.empty((b, k, 4), dtype=torch.int32, device=im.device)
        weights = torch.empty((b, k, 4), dtype=torch.float32, device=im.device)
        _lib.check(_lib.lib().dm_bev_interpolate_forward(
            im.data_ptr(), b, h, w, c, kp.data_ptr(), int(kp.shape[2]), k, _lib.floats(geom), out.data_ptr(),
            cells.data_ptr(), weights.data_ptr(), _lib.raw_stream()), 'dm_bev_interpolate_forward')
        ctx.save_for_backward(cells, weights)
        ctx.shape = (b, c, h, w)
        return out

    @staticmethod
    def backward(ctx, grad):
        from .. import _lib
        cells, weights = ctx.saved_tensors
        b, c, h, w = ctx.shape
        grad = grad.contiguous()
        gim = torch.empty((b, h, w, c), dtype=torch.float32, device=grad.device)
        _lib.check(_lib.lib().dm_bev_interpolate_backward(
            grad.data_ptr(), cells.data_ptr(), weights.data_ptr(), b, h, w, c, int(cells.shape[1]),
            gim.data_ptr(), _lib.raw_stream()), 'dm_bev_interpolate_backward')
        return gim.permute(0, 3, 1, 2), None, None


class FpsBatch(object):
    """with FpsBatch(): <prepare the geometry of several passes>  — their key-point FPS becomes one launch at exit."""

    def __enter__(self):
        self.outer = VoxelSetAbstraction._collect
        VoxelSetAbstraction._collect = []
        return self

    def __exit__(self, *exc):
        items, VoxelSetAbstraction._collect = VoxelSetAbstraction._collect, self.outer
        if exc[0] is None and items:
            VoxelSetAbstraction.flush_fps(items)
        return False


class VoxelSetAbstraction(nn.Module):

    def __init__(self, model_cfg, voxel_size, point_cloud_range, num_bev_features=None,
                 num_rawpoint_features=None, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        self.voxel_size = voxel_size
        self.point_cloud_range = point_cloud_range
        SA_cfg = copy.deepcopy(self.model_cfg.SA_LAYER)  # the reference mutates MLPS in place
        self.SA_layers = nn.ModuleList()
        self.SA_layer_names = []
        self.downsample_times_map = {}
        c_in = 0
        for src_name in self.model_cfg.FEATURES_SOURCE:
            if src_name in ['bev', 'raw_points']:
                continue
            self.downsample_times_map[src_name] = SA_cfg[src_name].DOWNSAMPLE_FACTOR
            mlps = [list(m) for m in SA_cfg[src_name].MLPS]
            for k in range(len(mlps)):
                mlps[k] = [mlps[k][0]] + mlps[k]
            self.SA_layers.append(pn2.StackSAModuleMSG(
                radii=SA_cfg[src_name].POOL_RADIUS, nsamples=SA_cfg[src_name].NSAMPLE, mlps=mlps,
                use_xyz=True, pool_method='max_pool'))
            self.SA_layer_names.append(src_name)
            c_in += sum([x[-1] for x in mlps])
        if 'bev' in self.model_cfg.FEATURES_SOURCE:
            c_in += num_bev_features
        if 'raw_points' in self.model_cfg.FEATURES_SOURCE:
            mlps = [list(m) for m in SA_cfg['raw_points'].MLPS]
            for k in range(len(mlps)):
                mlps[k] = [num_rawpoint_features - 3] + mlps[k]
            self.SA_rawpoints = pn2.StackSAModuleMSG(
                radii=SA_cfg['raw_points'].POOL_RADIUS, nsamples=SA_cfg['raw_points'].NSAMPLE,
                mlps=mlps, use_xyz=True, pool_method='max_pool')
            c_in += sum([x[-1] for x in mlps])
        self.vsa_point_feature_fusion = nn.Sequential(
            nn.Linear(c_in, self.model_cfg.NUM_OUTPUT_FEATURES, bias=False),
            nn.BatchNorm1d(self.model_cfg.NUM_OUTPUT_FEATURES), nn.ReLU())
        self.num_point_features = self.model_cfg.NUM_OUTPUT_FEATURES
        self.num_point_features_before_fusion = c_in

    def interpolate_from_bev_features(self, keypoints, bev_features, batch_size, bev_stride, fused=None):
        if fused_on(fused) and keypoints.is_cuda and bev_features.shape[1] % 4 == 0 and bev_features.shape[1] <= 1024:
            geom = (float(self.point_cloud_range[0]), float(self.point_cloud_range[1]),
                    float(self.voxel_size[0]), float(self.voxel_size[1]), float(bev_stride))
            return _BevInterpolate.apply(bev_features, keypoints, geom)
        x_idxs = (keypoints[:, :, 0] - self.point_cloud_range[0]) / self.voxel_size[0] / bev_stride
        y_idxs = (keypoints[:, :, 1] - self.point_cloud_range[1]) / self.voxel_size[1] / bev_stride
        out = []
        for k in range(batch_size):
            cur_bev = bev_features[k].permute(1, 2, 0)  # (H, W, C)
            out.append(bilinear_interpolate_torch(cur_bev, x_idxs[k], y_idxs[k]).unsqueeze(0))
        return torch.cat(out, dim=0)

    def get_sampled_points(self, batch_dict):
        """:119-158 — FPS of each sample's raw points (repeat-padded when N < NUM_KEYPOINTS)."""
        assert self.model_cfg.POINT_SOURCE == 'raw_points' and self.model_cfg.SAMPLE_METHOD == 'FPS'
        batch_size = batch_dict['batch_size']
        points = batch_dict['points']
        cnt = batch_dict['points_batch_cnt_host']
        num_kp = self.model_cfg.NUM_KEYPOINTS
        xyz = points[:, 1:4].contiguous()
        idx = pn2.furthest_point_sample_stack(xyz, cnt, num_kp).long()      # (B, num_kp)
        keypoints_list = []
        start = 0
        for bs_idx in range(batch_size):
            n = int(cnt[bs_idx])
            cur = idx[bs_idx]
            if n < num_kp:  # :143-146 repeat-pad the n distinct picks
                times = int(num_kp / n) + 1
                cur = cur[:n].repeat(times)[:num_kp]
            keypoints_list.append(xyz[start:start + n][cur].unsqueeze(dim=0))
            start += n
        return torch.cat(keypoints_list, dim=0)

    _side_streams = {}
    fps_stream = None        # SSL.forward_train (teacher ahead): the stream the key-point FPS is launched on instead of the side stream
    _collect = None          # FpsBatch: [(module, batch dict)] of the passes whose FPS is to be ONE launch

    def sample_keypoints_async(self, batch_dict):
        """FPS depends only on the raw points, runs 2047 dependent rounds and occupies one CU per
        sample: launch it on a side HIP stream as soon as the batch is assembled so that it
        overlaps voxelisation + the sparse and BEV backbones; forward() joins on the event."""
        pts = batch_dict['points']
        if not pts.is_cuda:
            return
        if VoxelSetAbstraction._collect is not None:
            VoxelSetAbstraction._collect.append((self, batch_dict))
            return
        # a small POOL of side streams, used round-robin: the three passes of a DetMatch iteration
        # (labeled student, teacher, unlabeled student) issue their FPS back to back at the step
        # boundary; on ONE side stream the launches would run one after the other (3 x 7 ms with two
        # CUs busy each), on three they run side by side
        key = pts.device.index
        pool = VoxelSetAbstraction._side_streams.get(key)
        if pool is None:
            # (round 5: ONE side stream, shared with the geometry look-ahead — `_lib.aux_stream` says why; the FPS
            # launches of an iteration are issued one iteration ahead, so running them one after the other costs
            # nothing; a pool of three streams was one of the conditions of the round-5 dead-lock.)
            from .. import _lib
            pool = VoxelSetAbstraction._side_streams[key] = dict(streams=[_lib.aux_stream(pts.device)], next=0)
        side = pool['streams'][pool['next'] % len(pool['streams'])]
        pool['next'] += 1
        if VoxelSetAbstraction.fps_stream is not None:
            side = VoxelSetAbstraction.fps_stream
        main = torch.cuda.current_stream(pts.device)
        side.wait_stream(main)
        with torch.cuda.stream(side), torch.no_grad():
            kp = self.get_sampled_points(batch_dict)
            ev = torch.cuda.Event()
            ev.record(side)
        batch_dict['keypoints_async'] = (kp, ev)

    @staticmethod
    def flush_fps(items):
        """The key points of SEVERAL passes (labeled student, teacher, unlabeled student of one DetMatch iteration) from
        ONE launch: FPS is a chain of NUM_KEYPOINTS dependent rounds on one workgroup per sample (3.4 ms for 20 000 ->
        2048 whatever else the device does), so three launches behind each other on the side stream deliver the last
        pass's key points after 10 ms, one launch over all six samples after 3.4 — same indices per sample (each
        workgroup sees its own sample only)."""
        groups = {}
        for mod, bd in items:
            groups.setdefault((bd['points'].device, int(mod.model_cfg.NUM_KEYPOINTS)), []).append((mod, bd))
        for (dev, num_kp), grp in groups.items():
            if len(grp) == 1:
                grp[0][0].sample_keypoints_async(grp[0][1])
                continue
            from .. import _lib
            side = VoxelSetAbstraction.fps_stream if VoxelSetAbstraction.fps_stream is not None else _lib.aux_stream(dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side), torch.no_grad():
                xyz = torch.cat([bd['points'][:, 1:4] for _, bd in grp], dim=0).contiguous()
                cnt = [int(c) for _, bd in grp for c in bd['points_batch_cnt_host']]
                idx = pn2.furthest_point_sample_stack(xyz, cnt, num_kp).long()          # (sum of batch sizes, num_kp)
                row, start, out = 0, 0, []
                for mod, bd in grp:
                    kps = []
                    for n in bd['points_batch_cnt_host']:
                        n = int(n)
                        cur = idx[row]
                        if n < num_kp:       # get_sampled_points: repeat-pad the n distinct picks
                            cur = cur[:n].repeat(int(num_kp / n) + 1)[:num_kp]
                        kps.append(xyz[start:start + n][cur].unsqueeze(dim=0))
                        row += 1
                        start += n
                    out.append(torch.cat(kps, dim=0))
                ev = torch.cuda.Event()
                ev.record(side)
            for (mod, bd), kp in zip(grp, out):
                bd['keypoints_async'] = (kp, ev)

    # ---- feature sources ---------------------------------------------------------------------
    def _branch_fns(self, batch_dict, keypoints, new_xyz, new_xyz_batch_cnt):
        """One callable per feature source, in the column order of the fused feature (bev, raw points,
        x_conv1..4): each reads only the key points and its own source and returns (B, K, C_src)."""
        batch_size, num_keypoints, _ = keypoints.shape
        dev = keypoints.device
        fns = []
        if 'bev' in self.model_cfg.FEATURES_SOURCE:
            spatial = batch_dict['spatial_features']
            stride = batch_dict['spatial_features_stride']
            fns.append(lambda: self.interpolate_from_bev_features(keypoints, spatial, batch_size,
                                                                   bev_stride=stride))
        if 'raw_points' in self.model_cfg.FEATURES_SOURCE:
            raw_points = batch_dict['points']
            counts = [int(c) for c in batch_dict['points_batch_cnt_host']]

            def raw():
                xyz_batch_cnt = upload(counts, dev, torch.int32)
                point_features = raw_points[:, 4:].contiguous() if raw_points.shape[1] > 4 else None
                _, pooled = self.SA_rawpoints(xyz=raw_points[:, 1:4].contiguous(),
                                              xyz_batch_cnt=xyz_batch_cnt, new_xyz=new_xyz,
                                              new_xyz_batch_cnt=new_xyz_batch_cnt,
                                              features=point_features)
                return pooled.view(batch_size, num_keypoints, -1)
            fns.append(raw)
        for k, src_name in enumerate(self.SA_layer_names):
            sp = batch_dict['multi_scale_3d_features'][src_name]
            feats = sp.features

            def sa(k=k, src_name=src_name, sp=sp, feats=feats):
                xyz, xyz_batch_cnt = voxel_centers_and_counts(
                    sp.indices, batch_size, self.downsample_times_map[src_name], self.voxel_size, self.point_cloud_range)
                _, pooled = self.SA_layers[k](xyz=xyz, xyz_batch_cnt=xyz_batch_cnt,
                                              new_xyz=new_xyz, new_xyz_batch_cnt=new_xyz_batch_cnt,
                                              features=feats.contiguous())
                return pooled.view(batch_size, num_keypoints, -1)
            fns.append(sa)
        return fns

    def forward(self, batch_dict):
        # (Measured and dropped: the six sources on three side streams underneath the BEV backbone, with
        # backward gates in front of the sparse backbone: 117.8-118.6 ms against 111.5-117.0 ms on one
        # stream, 3 alternations of 40 iterations — the event hand-offs and the convolutions sharing the
        # device cost more than the dispatch gaps they hide.)
        if 'keypoints_async' in batch_dict:
            keypoints, ev = batch_dict.pop('keypoints_async')
            main = torch.cuda.current_stream(keypoints.device)
            main.wait_event(ev)
            keypoints.record_stream(main)
        else:
            keypoints = self.get_sampled_points(batch_dict)
        new_xyz = keypoints.view(-1, 3).contiguous()
        new_xyz_batch_cnt = torch.full((keypoints.shape[0],), keypoints.shape[1], dtype=torch.int32,
                                       device=keypoints.device)
        point_features_list = [fn() for fn in self._branch_fns(batch_dict, keypoints, new_xyz,
                                                               new_xyz_batch_cnt)]
        batch_size, num_keypoints, _ = keypoints.shape
        dev = keypoints.device
        point_features = torch.cat(point_features_list, dim=2)
        batch_idx = torch.arange(batch_size, device=dev).view(-1, 1).repeat(1, num_keypoints).view(-1)
        point_coords = torch.cat((batch_idx.view(-1, 1).float(), keypoints.view(-1, 3)), dim=1)
        batch_dict['point_features_before_fusion'] = point_features.view(-1, point_features.shape[-1])
        batch_dict['point_features'] = fc_rows(self.vsa_point_feature_fusion,
                                               point_features.view(-1, point_features.shape[-1]))
        batch_dict['point_coords'] = point_coords
        batch_dict['point_batch_cnt'] = new_xyz_batch_cnt      # rows of point_coords per sample (the RoI head's ball queries)
        return batch_dict
