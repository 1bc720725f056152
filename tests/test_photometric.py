"""Student-only photometric augmentation (detmatch_amd/ts_ssl_dataset.py:StudentPhotometric; the
torchvision chain of configs/detmatch/001/detmatch/split_0.py:586-625 — un-vendored, parity unpinned):
property tests of the formulas."""
import colorsys

import numpy as np
import torch

from detmatch_amd.ts_ssl_dataset import ImageResizeFlipNormPad, StudentPhotometric


def _img(seed=0, h=40, w=64):
    return torch.from_numpy(np.random.RandomState(seed).randint(0, 256, (3, h, w)).astype(np.float32))


def test_disabled_chain_is_the_identity_and_levels_stay_8_bit():
    aug = StudentPhotometric(p_jitter=0.0, p_grey=0.0, p_blur=0.0, erasing=())
    x = _img()
    assert torch.equal(aug(x, np.random.RandomState(0)), x)
    out = StudentPhotometric()(x, np.random.RandomState(1))
    assert out.shape == x.shape and float(out.min()) >= 0 and float(out.max()) <= 255
    assert torch.equal(out, out.round())


def test_hue_matches_colorsys_and_zero_shift_is_identity():
    x = _img(3) / 255.0
    assert torch.allclose(StudentPhotometric._hue(x, 0.0), x, atol=1e-6)
    got = StudentPhotometric._hue(x, 0.07).permute(1, 2, 0).reshape(-1, 3).numpy()
    want = []
    for r, g, b in x.permute(1, 2, 0).reshape(-1, 3).numpy().tolist():
        h, s, v = colorsys.rgb_to_hsv(r, g, b)
        want.append(colorsys.hsv_to_rgb((h + 0.07) % 1.0, s, v))
    np.testing.assert_allclose(got, np.array(want), atol=2e-6)


def test_grey_blur_and_erasing():
    x = _img(5)
    g = StudentPhotometric(p_jitter=0.0, p_grey=1.0, p_blur=0.0, erasing=())(x, np.random.RandomState(0))
    assert torch.equal(g[0], g[1]) and torch.equal(g[1], g[2])
    want = (0.299 * x[0] + 0.587 * x[1] + 0.114 * x[2]).round()
    assert float((g[0] - want).abs().max()) <= 1.0
    b = StudentPhotometric(p_jitter=0.0, p_grey=0.0, p_blur=1.0, erasing=())(x, np.random.RandomState(0))
    assert float(b.var()) < float(x.var()) and abs(float(b.mean()) - float(x.mean())) < 2.0   # smooths, keeps the mean
    e = StudentPhotometric(p_jitter=0.0, p_grey=0.0, p_blur=0.0, erasing=((1.0, (0.05, 0.2), (0.3, 3.3)),))(
        x.clone(), np.random.RandomState(2))
    changed = (e != x).any(0)
    ys, xs = np.nonzero(changed.numpy())
    hh, ww = ys.max() - ys.min() + 1, xs.max() - xs.min() + 1
    assert 0.04 * 40 * 64 <= hh * ww <= 0.22 * 40 * 64              # one rectangle of 5-20 % of the image
    assert int(changed.sum()) >= 0.9 * hh * ww


def test_student_and_teacher_images_share_geometry():
    raw = _img(7, 37, 123).permute(1, 2, 0).contiguous()
    tf = ImageResizeFlipNormPad(((128, 40), (128, 40)))
    t, mt = tf(raw, (128, 40), True)
    s, ms = tf(raw, (128, 40), True, photometric=lambda x: StudentPhotometric()(x, np.random.RandomState(0)))
    assert t.shape == s.shape and mt['img_shape'] == ms['img_shape'] and mt['pad_shape'] == ms['pad_shape']
    assert t.shape[1] % 32 == 0 and t.shape[2] % 32 == 0 and not torch.equal(t, s)
