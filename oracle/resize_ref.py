"""TEST INFRASTRUCTURE -- NumPy restatement of `cv2.resize(img, (w, h))` (INTER_LINEAR, uint8) used at reference
pose_detector.py:493.  PARITY UNPINNED: OpenCV is an unpinned third-party dependency of the reference that cannot be
installed here, so this follows OpenCV's published fixed-point algorithm; the product's HIP kernel
(csrc/prep.hip::resize_linear_u8_kernel, tables from pmx_api.hip::make_resize_table) is tested bit-exactly against it.
Only tests/ may import this module."""
import numpy as np


def resize_linear_u8(img, dst_w, dst_h):
    """Restatement of `cv2.resize(img, (dst_w, dst_h))` (INTER_LINEAR, uint8) used at pose_detector.py:493.

    OpenCV is a third-party dependency that is not vendored by the reference and not installable here, so this
    follows OpenCV's published fixed-point algorithm (imgproc/resize.cpp: half-pixel source coordinates computed
    in float32, 11-bit coefficients `saturate_cast<short>(c * 2048)`, horizontal pass in int32, vertical pass
    `(((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2`).  PARITY UNPINNED (no cv2 to compare
    against); it is the identity when the size does not change, which is the case for every 368 x 368 input.
    """
    img = np.ascontiguousarray(img, dtype=np.uint8)
    src_h, src_w, cn = img.shape
    if (src_w, src_h) == (dst_w, dst_h):
        return img.copy()

    def coeffs(dst, src):
        scale = 1.0 / (float(dst) / float(src))
        d = np.arange(dst, dtype=np.float64)
        f = ((d + 0.5) * scale - 0.5).astype(np.float32)
        s = np.floor(f).astype(np.int64)
        f = f - s.astype(np.float32)
        lo = s < 0
        f[lo] = 0
        s[lo] = 0
        hi = s >= src - 1
        f[hi] = 0
        s[hi] = src - 1
        c0 = np.rint((np.float32(1.0) - f) * np.float32(2048)).astype(np.int64)
        c1 = np.rint(f * np.float32(2048)).astype(np.int64)
        return s, np.minimum(s + 1, src - 1), c0, c1

    sx, sx1, ax0, ax1 = coeffs(dst_w, src_w)
    sy, sy1, by0, by1 = coeffs(dst_h, src_h)
    src = img.astype(np.int64)
    rows = src[:, sx, :] * ax0[None, :, None] + src[:, sx1, :] * ax1[None, :, None]       # (src_h, dst_w, cn)
    s0 = rows[sy]
    s1 = rows[sy1]
    out = (((by0[:, None, None] * (s0 >> 4)) >> 16) + ((by1[:, None, None] * (s1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)
