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

    def _detect(self, img):
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
        """reference hand_detector.py:28-50: a left hand is mirrored (cv2.flip(img, 1)) before the network and its heat
        maps are mirrored back before the peaks are taken; the Gaussian and the arg-max commute with the mirror (reflect
        border, commutative pair sums), so the key points of the mirrored maps are mirrored back instead: x -> W - 1 - x
        (only the row-major tie order of exactly equal maxima could differ)."""
        hand_img = np.asarray(hand_img)
        if hand_type == "left":
            kps = self._detect(hand_img[:, ::-1])
            w = hand_img.shape[1]
            return [None if k is None else [w - 1 - k[0], k[1], k[2]] for k in kps]
        return self._detect(hand_img)
