"""`FaceDetector` / `HandDetector` -- mirrors of the reference classes (face_detector.py:12-77, hand_detector.py:12-87) on
the same MI355X conv kernels as the pose network (single-branch CPM, models/FaceNet.py / models/HandNet.py).

    FaceDetector(arch='facenet', weights_file=None, model=None, device=-1)(face_img, fast_mode=False) -> 70 key points
    HandDetector(arch='handnet', weights_file=None, model=None, device=-1)(hand_img, fast_mode=False, hand_type="right") -> 21

A key point is `[x, y, confidence]` (ints, np.float32) or `None` when the smoothed maximum does not exceed the threshold,
in the pixel frame of the crop that was passed in -- exactly the reference's return value.  The whole path runs on the
GPU: cv2.resize to 368 x 368 (restated INTER_LINEAR kernel), x / 256 - 0.5, network, corner-aligned resize of the last
stage to the crop size, SciPy-equivalent Gaussian, arg-max (CPU-branch semantics, including the reference's quirk for
tied maxima).  `model=` may be a weights dict or a callable returning the list of stage outputs (test seam; the reference
ignores its `model` argument).
"""
import numpy as np

from . import native
from . import weights as weights_mod
from .entity import params


class _KeypointDetector(object):
    ARCH = None
    SIZE_KEY = None
    THRESH_KEY = None

    def __init__(self, arch=None, weights_file=None, model=None, device=-1, weights=None):
        self.arch = arch or self.ARCH
        if self.arch != self.ARCH:
            raise ValueError('%s needs arch=%r' % (type(self).__name__, self.ARCH))
        self.device = device
        self.model = model if callable(model) else None
        w = model if isinstance(model, dict) else weights
        if w is None and weights_file:
            w = weights_mod.load_npz(weights_file, self.ARCH)          # serializers.load_npz (face_detector.py:16)
        size = params[self.SIZE_KEY]
        self.engine = native.Engine(device if device >= 0 else 0, max_batch=1, max_h=size, max_w=size,
                                    gaussian_sigma=params['gaussian_sigma'], arch=self.ARCH)
        if w is not None:
            self.engine.set_weights(w)

    def _detect(self, img, flip_maps=False):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        h, w, _ = img.shape
        size = params[self.SIZE_KEY]
        if self.model is None:
            if self.engine.weights_missing():
                raise RuntimeError('%s has no weights: pass weights_file=, weights= or model=' % type(self).__name__)
            self.engine.forward_u8_resized(img[None], size, size)          # cv2.resize + /256 - 0.5 + network (:31-36)
        else:
            resized = self.engine.resize_u8(img[None], size, size)[0]
            x = np.array(resized[np.newaxis], dtype=np.float32).transpose(0, 3, 1, 2) / 256 - 0.5     # :32
            hs = self.model(x)
            self.engine.set_heat(np.asarray(getattr(hs[-1], 'data', hs[-1]), dtype=np.float32))
        self.engine.set_option('kp_flip_x', int(flip_maps))               # cv2.flip(heatmaps, 1) for left hands (hand_detector.py:46-47)
        kp = self.engine.keypoints(h, w, params[self.THRESH_KEY])[0]      # F.resize_images + peaks (:37-38)
        out = []
        for x, y, conf, valid in kp:
            out.append([int(x), int(y), np.float32(conf)] if valid else None)
        return out


class FaceDetector(_KeypointDetector):
    ARCH, SIZE_KEY, THRESH_KEY = 'facenet', 'face_inference_img_size', 'face_heatmap_peak_thresh'

    def __call__(self, face_img, fast_mode=False):
        """reference face_detector.py:28-40 (`fast_mode` is unused there as well)"""
        return self._detect(face_img)


class HandDetector(_KeypointDetector):
    ARCH, SIZE_KEY, THRESH_KEY = 'handnet', 'hand_inference_img_size', 'hand_heatmap_peak_thresh'

    def __call__(self, hand_img, fast_mode=False, hand_type="right"):
        """reference hand_detector.py:28-50: a left hand is mirrored (cv2.flip(img, 1)) before the network and its resized
        heat maps are mirrored back before the Gaussian and the arg-max -- here on the device (the column tables of the
        resize are reversed: the same samples, so also the same row-major order among exactly equal maxima)."""
        hand_img = np.asarray(hand_img)
        if hand_type == "left":
            return self._detect(hand_img[:, ::-1], flip_maps=True)
        return self._detect(hand_img)


# ---- visualisation / crop helpers of the reference modules (host-only; cv2.circle / cv2.line rasters are stand-ins) -----------
def _shifted(kp, left_top):
    return (kp[0] + left_top[0], kp[1] + left_top[1])


def draw_face_keypoints(orig_img, face_keypoints, left_top):
    """reference face_detector.py:79-97: the 70 key points as radius-2 discs and the `face_line_indices` polylines
    (thickness 1), colour (255, 255, 0), on a copy of the image; `left_top` = origin of the face crop."""
    from .pose_detector import _draw_disc, _draw_line
    img = np.array(orig_img, copy=True)
    for kp in face_keypoints:
        if kp:
            _draw_disc(img, _shifted(kp, left_top), 2, (255, 255, 0))
    for a, b in params['face_line_indices']:
        if face_keypoints[a] and face_keypoints[b]:
            _draw_line(img, _shifted(face_keypoints[a], left_top), _shifted(face_keypoints[b], left_top), (255, 255, 0), 1)
    return img


FINGER_COLORS = [(0, 0, 255), (0, 255, 255), (0, 255, 0), (255, 0, 0), (255, 0, 255)]


def draw_hand_keypoints(orig_img, hand_keypoints, left_top):
    """reference hand_detector.py:89-115: per finger, radius-3 discs at both ends of every bone whose key point exists and
    a thickness-1 line where both exist."""
    from .pose_detector import _draw_disc, _draw_line
    img = np.array(orig_img, copy=True)
    for color, bones in zip(FINGER_COLORS, params['fingers_indices']):
        for a, b in bones:
            ka, kb = hand_keypoints[a], hand_keypoints[b]
            for k in (ka, kb):
                if k:
                    _draw_disc(img, _shifted(k, left_top), 3, color)
            if ka and kb:
                _draw_line(img, _shifted(ka, left_top), _shifted(kb, left_top), color, 1)
    return img


def crop_face(img, rect):
    """reference face_detector.py:99-114 (camera_face_demo.py): `rect` = (x, y, w, h) of a face box; the box is scaled by
    `face_crop_scale` about its centre, clipped to the image and zero-padded to a square.  -> (crop, (left, top))."""
    h, w = img.shape[:2]
    cx, cy = rect[0] + rect[2] / 2, rect[1] + rect[3] / 2
    half_w, half_h = rect[2] * params['face_crop_scale'] / 2, rect[3] * params['face_crop_scale'] / 2
    left, top = max(0, int(cx - half_w)), max(0, int(cy - half_h))
    right, bottom = min(w - 1, int(cx + half_w)), min(h - 1, int(cy + half_h))
    face = img[top:bottom, left:right]
    edge = max(face.shape[:2])
    out = np.zeros((edge, edge, face.shape[2]), dtype=np.uint8)
    out[:face.shape[0], :face.shape[1]] = face
    return out, (left, top)
