"""TEST INFRASTRUCTURE -- CPU restatement of `PoseDetector.detect_precise` (reference pose_detector.py:433-482).

PARITY UNPINNED for the cv2 steps: OpenCV (unpinned third-party dependency of the reference, not installable
here) provides `cv2.resize(..., interpolation=cv2.INTER_CUBIC)`; it is restated below from OpenCV's published
algorithm (bicubic with A = -0.75, half-pixel source coordinates computed as float32, replicate border; uint8 path
in 11-bit fixed point).  Written tap-by-tap with explicit loops over the 4 taps and an einsum-free gather so that it
is an independent implementation from the vectorised product code in pose_detector.py.

Only tests/ may import this module.
"""
import math

import numpy as np

from . import postprocess_ref as P

INFERENCE_SCALES = [0.5, 1, 1.5, 2]      # entity.py:72
INFERENCE_IMG_SIZE = 368                 # entity.py:71
DOWNSCALE = 8                            # entity.py:59


def _coeffs(fx):
    A = np.float32(-0.75)
    x = np.float32(fx)
    c = np.empty(4, np.float32)
    c[0] = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A
    c[1] = ((A + 2) * x - (A + 3)) * x * x + 1
    c[2] = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1
    c[3] = np.float32(1) - c[0] - c[1] - c[2]
    return c


def _axis_table(dst, src):
    scale = 1.0 / (dst / src)
    idx = np.empty((dst, 4), np.int64)
    co = np.empty((dst, 4), np.float32)
    for d in range(dst):
        f = np.float32((d + 0.5) * scale - 0.5)
        s = int(math.floor(f))
        co[d] = _coeffs(f - np.float32(s))
        for k in range(4):
            idx[d, k] = min(max(s - 1 + k, 0), src - 1)
    return idx, co


def resize_cubic_f32_ref(img, dst_w, dst_h):
    img = np.asarray(img, dtype=np.float32)
    sh, sw = img.shape[:2]
    if (sw, sh) == (dst_w, dst_h):
        return img.copy()
    ix, cx = _axis_table(dst_w, sw)
    iy, cy = _axis_table(dst_h, sh)
    rows = np.empty((sh, dst_w) + img.shape[2:], np.float32)
    for d in range(dst_w):
        acc = img[:, ix[d, 0]] * cx[d, 0]
        for k in (1, 2, 3):
            acc = acc + img[:, ix[d, k]] * cx[d, k]
        rows[:, d] = acc
    out = np.empty((dst_h, dst_w) + img.shape[2:], np.float32)
    for d in range(dst_h):
        acc = rows[iy[d, 0]] * cy[d, 0]
        for k in (1, 2, 3):
            acc = acc + rows[iy[d, k]] * cy[d, k]
        out[d] = acc
    return out


def resize_cubic_u8_ref(img, dst_w, dst_h):
    img = np.asarray(img, dtype=np.uint8)
    sh, sw = img.shape[:2]
    if (sw, sh) == (dst_w, dst_h):
        return img.copy()
    ix, cx = _axis_table(dst_w, sw)
    iy, cy = _axis_table(dst_h, sh)
    ax = np.clip(np.rint(cx * np.float32(2048)), -32768, 32767).astype(np.int64)
    ay = np.clip(np.rint(cy * np.float32(2048)), -32768, 32767).astype(np.int64)
    src = img.astype(np.int64)
    rows = np.zeros((sh, dst_w) + img.shape[2:], np.int64)
    for d in range(dst_w):
        for k in range(4):
            rows[:, d] += src[:, ix[d, k]] * ax[d, k]
    out = np.zeros((dst_h, dst_w) + img.shape[2:], np.int64)
    for d in range(dst_h):
        for k in range(4):
            out[d] += rows[iy[d, k]] * ay[d, k]
    return np.clip((out + (1 << 21)) >> 22, 0, 255).astype(np.uint8)


def pad_image(img, stride, pad_value):
    """pose_detector.py:46-55"""
    h, w, _ = img.shape
    pad = [(stride - (h % stride)) % stride, (stride - (w % stride)) % stride]
    out = np.zeros((h + pad[0], w + pad[1], 3), np.int64) + np.asarray(pad_value)
    out[:h, :w, :] = img
    return out.astype(np.uint8), pad


def averaged_maps(model, orig_img):
    """pose_detector.py:436-470; `model(x_nchw_f32) -> (paf (1,38,h,w), heat (1,19,h,w))`."""
    oh, ow, _ = orig_img.shape
    pafs_sum = 0
    heat_sum = 0
    sizes = []
    for scale in INFERENCE_SCALES:
        mult = scale * INFERENCE_IMG_SIZE / min(oh, ow)
        img = resize_cubic_u8_ref(orig_img, math.ceil(ow * mult), math.ceil(oh * mult))
        padded, pad = pad_image(img, DOWNSCALE, (104, 117, 123))
        ph, pw = padded.shape[:2]
        sizes.append((ph, pw))
        paf, heat = model(P.preprocess(padded))
        tp = resize_cubic_f32_ref(paf[0].transpose(1, 2, 0), pw, ph)[:ph - pad[0], :pw - pad[1]]
        pafs_sum = pafs_sum + resize_cubic_f32_ref(tp, ow, oh)
        th = heat[0].transpose(1, 2, 0)
        th = resize_cubic_f32_ref(th, th.shape[1] * DOWNSCALE, th.shape[0] * DOWNSCALE)[:ph - pad[0], :pw - pad[1]]
        heat_sum = heat_sum + resize_cubic_f32_ref(th, ow, oh)
    n = len(INFERENCE_SCALES)
    return (pafs_sum / n).transpose(2, 0, 1), (heat_sum / n).transpose(2, 0, 1), sizes


def detect_precise_from_maps(pafs, heatmaps):
    """pose_detector.py:475-482 on the averaged full-resolution maps: img_len = orig_img_w, no rescale."""
    w = heatmaps.shape[2]
    return P.postprocess(np.ascontiguousarray(heatmaps, dtype=np.float32), np.ascontiguousarray(pafs, dtype=np.float32), w)
