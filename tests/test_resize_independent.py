"""CPU: the restated third-party resizes (cv2.resize, reference pose_detector.py:493 INTER_LINEAR uint8; :443, :461-467 INTER_CUBIC uint8 /
float32) against TWO independent libraries that implement the same sampling geometry -- torch's `interpolate` (half-pixel centres,
align_corners=False, no antialiasing; bicubic with A = -0.75 like OpenCV) and Pillow's affine transform with bilinear sampling (point
sampled, replicated border).  No OpenCV exists in this image, so the restatements cannot be held against cv2 itself; what these tests pin
is everything but OpenCV's fixed-point rounding: the source coordinate of every output pixel, the border rule, the kernel -- to +-1 grey
level (uint8) / 1e-5 (float32), for up- AND down-scaling at the ratios the reference uses (584 -> 368 for data/person.png; the 0.5x .. 2x
scales of detect_precise)."""
import numpy as np
import pytest
import torch

from oracle import precise_ref as PR
from oracle import resize_ref as RR

SIZES = [((584, 584), (368, 368)),            # data/person.png -> network input (pose_detector.py:493)
         ((482, 642), (368, 496)),            # a 4:3 frame
         ((120, 90), (368, 280)),             # up-scaling
         ((97, 131), (64, 200)),              # down in one axis, up in the other
         ((50, 70), (33, 41))]


def _torch_resize(a, dh, dw, mode):
    t = torch.from_numpy(a.astype(np.float64).transpose(2, 0, 1))[None]
    return torch.nn.functional.interpolate(t, size=(dh, dw), mode=mode, align_corners=False)[0].numpy().transpose(1, 2, 0)


@pytest.mark.parametrize('src,dst', SIZES)
def test_linear_u8_within_one_level_of_torch_and_pillow(src, dst):
    from PIL import Image
    rng = np.random.default_rng(src[0] + dst[1])
    # smooth + noise: noise alone would hide a half-pixel shift behind its own variance
    yy, xx = np.mgrid[0:src[0], 0:src[1]]
    base = 127 + 90 * np.sin(xx / 9.0)[..., None] * np.cos(yy / 7.0)[..., None] + rng.normal(0, 12, src + (3,))
    img = np.clip(base, 0, 255).astype(np.uint8)
    (dh, dw) = dst
    ours = RR.resize_linear_u8(img, dw, dh).astype(np.float64)
    t = _torch_resize(img, dh, dw, 'bilinear')
    assert np.abs(ours - t).max() <= 1.0, np.abs(ours - t).max()            # 11-bit fixed point vs float: rounding only
    assert np.abs(ours - t).mean() < 0.3
    # Pillow: output pixel (x, y) samples the input at (a (x + 0.5), e (y + 0.5)) with centres at +0.5 -- OpenCV's geometry -- bilinear, edge
    # pixels replicated, float arithmetic
    sx, sy = src[1] / dw, src[0] / dh
    p = np.stack([np.asarray(Image.fromarray(img[..., c]).transform((dw, dh), Image.AFFINE, (sx, 0, 0, 0, sy, 0), resample=Image.BILINEAR))
                  for c in range(3)], axis=-1).astype(np.float64)
    inner = (slice(1, -1), slice(1, -1))                                    # (Pillow fills samples whose centre falls outside the image: skip the rim)
    assert np.abs(ours[inner] - p[inner]).max() <= 1.0, np.abs(ours[inner] - p[inner]).max()
    # and a deliberately wrong geometry (corner-aligned) is NOT within a level: the bar means something
    wrong = torch.nn.functional.interpolate(torch.from_numpy(img.astype(np.float64).transpose(2, 0, 1))[None], size=(dh, dw), mode='bilinear',
                                            align_corners=True)[0].numpy().transpose(1, 2, 0)
    assert np.abs(ours - wrong).max() > 3.0


def _keys_half(a, dw, dh):
    """separable cubic convolution with A = -0.5, half-pixel centres, replicated border (NumPy, float64): the 'other' bicubic"""
    def w(t):
        t = np.abs(t)
        return np.where(t <= 1, (1.5 * t - 2.5) * t * t + 1, np.where(t < 2, ((-0.5 * t + 2.5) * t - 4) * t + 2, 0.0))

    def axis(x, n_dst, ax):
        n = x.shape[ax]
        c = (np.arange(n_dst) + 0.5) * n / n_dst - 0.5
        i0 = np.floor(c).astype(int)
        out = 0
        for k in range(-1, 3):
            idx = np.clip(i0 + k, 0, n - 1)
            wk = w(c - (i0 + k))
            shape = [1] * x.ndim; shape[ax] = n_dst
            out = out + np.take(x, idx, axis=ax) * wk.reshape(shape)
        return out
    return axis(axis(a.astype(np.float64), dw, 1), dh, 0)


@pytest.mark.parametrize('src,dst', SIZES)
def test_cubic_within_rounding_of_torch_bicubic(src, dst):
    rng = np.random.default_rng(src[1] + dst[0])
    (dh, dw) = dst
    # OpenCV computes the source coordinate in float32 -- restated as such: at x ~ 600 it carries 3e-5 of rounding, which the local slope
    # of the map turns into value error (torch works in float64).  A smooth field (slope <= 0.3 per pixel) shows the geometry and the
    # kernel to 3e-5; white noise (slope of several units per pixel) is the worst case and stays within 5e-4.
    yy, xx = np.mgrid[0:src[0], 0:src[1]]
    smooth = (np.sin(xx / 11.0)[..., None] * np.cos(yy / 13.0)[..., None] * np.ones(3)).astype(np.float32)
    d = np.abs(PR.resize_cubic_f32_ref(smooth, dw, dh) - _torch_resize(smooth, dh, dw, 'bicubic')).max()
    assert d < 3e-5, d
    a = rng.standard_normal(src + (3,)).astype(np.float32)
    d = np.abs(PR.resize_cubic_f32_ref(a, dw, dh) - _torch_resize(a, dh, dw, 'bicubic')).max()
    assert d < 5e-4, d
    # (a different cubic -- Keys' A = -0.5, Pillow's and many others' -- is NOT within that: the bar separates the kernels)
    assert np.abs(PR.resize_cubic_f32_ref(a, dw, dh) - _keys_half(a, dw, dh)).max() > 1e-2
    img = rng.integers(0, 256, src + (3,), dtype=np.uint8)
    ours8 = PR.resize_cubic_u8_ref(img, dw, dh).astype(np.float64)
    t8 = np.clip(np.rint(_torch_resize(img, dh, dw, 'bicubic')), 0, 255)
    assert np.abs(ours8 - t8).max() <= 1.0, np.abs(ours8 - t8).max()         # fixed point vs float: rounding only
    assert (ours8 != t8).mean() < 0.2


def test_detect_precise_scales_against_torch():
    """The four scales of detect_precise on a 482 x 642 frame (pose_detector.py:441-443: multiplier = scale * 368 / min(h, w), ceil):
    the uint8 cubic resize of the frame at every one of them within a grey level of torch's bicubic."""
    import math
    rng = np.random.default_rng(5)
    h, w = 482, 642
    img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    for scale in (0.5, 1.0, 1.5, 2.0):
        m = scale * 368 / min(h, w)
        dw, dh = math.ceil(w * m), math.ceil(h * m)
        ours = PR.resize_cubic_u8_ref(img, dw, dh).astype(np.float64)
        t = np.clip(np.rint(_torch_resize(img, dh, dw, 'bicubic')), 0, 255)
        assert np.abs(ours - t).max() <= 1.0, (scale, np.abs(ours - t).max())
