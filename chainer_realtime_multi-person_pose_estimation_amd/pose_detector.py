"""`PoseDetector` -- drop-in mirror of the reference class (reference `pose_detector.py:15-517`) whose
hot path runs on one MI355X through libpose_mi355x.so (C ABI: include/pose_mi355x.h).

Same constructor and call signature as the reference:

    PoseDetector(arch=None, weights_file=None, model=None, device=-1, precise=False)   # pose_detector.py:16
    poses, scores = detector(orig_img)          # orig_img: H x W x 3 uint8 BGR             pose_detector.py:484

`poses` is float64 (n, 18, 3) rows [x, y, 2] / [0, 0, 0] in original-image pixels, `scores` float64 (n,);
the "nothing found" returns have the reference's shapes (:509-510, :264).  The numerics are those of the
reference's CPU branch (the golden one): SciPy-equivalent Gaussian, 4-neighbour strict NMS, NumPy-order
float64 PAF scoring, greedy matching and grouping -- executed by HIP kernels.

What `model=` means here (the reference's plug-in seam, :19-20):
  * None                -> the built-in CocoPoseNet kernels with `weights_file` (Chainer NPZ) or `weights=`;
  * dict name->(W, b)   -> the built-in kernels with these weights;
  * a callable          -> called as `model(x)` with the float32 NCHW input exactly like the reference calls its
                           Chain (:499); it must return `(pafs, heatmaps)` lists whose last elements are arrays
                           of shape (1, 38, h, w) / (1, 19, h, w).  Their values are installed on the device and
                           the post-process runs on the GPU (this is how the tests inject synthetic skeletons).

`device`: GPU index; -1 (the reference's "CPU") selects GPU 0 -- there is one numerics path (the CPU-branch
semantics) and it always runs on the MI355X; there is no CPU fallback, construction raises without a GPU.
"""
import math

import numpy as np

from . import native
from . import weights as weights_mod
from .entity import JointType, params


class PoseDetector(object):
    precise_largest_first = True      # detect_precise enqueues its largest scale first (same results: the parts are summed in the reference's order)

    def __init__(self, arch=None, weights_file=None, model=None, device=-1, precise=False, weights=None,
                 max_batch=1, max_size=None, gpu_branch_peaks=False, precision='f32'):
        self.arch = arch
        self.precise = precise
        self.device = device
        self._gpu = device if device >= 0 else 0
        self.model = None
        w = None
        if callable(model):
            self.model = model
        elif isinstance(model, dict):
            w = model
        elif model is not None:
            raise TypeError('model must be None, a weights dict or a callable returning (pafs, heatmaps)')
        else:
            if arch not in (None, 'posenet'):
                raise ValueError("only arch='posenet' is on the accelerated path (got %r)" % (arch,))
            if weights is not None:
                w = weights
            elif weights_file:
                w = weights_mod.load_npz(weights_file)     # serializers.load_npz (:26)
        size = params['inference_img_size']
        mh, mw = (size, size) if max_size is None else max_size
        self._weights = w
        self._gpu_branch_peaks = bool(gpu_branch_peaks)
        if precision not in ('f32', 'bf16x3'):
            raise ValueError("precision must be 'f32' (default: the fp32 arithmetic the parity tests specify) or 'bf16x3' (opt-in, frozen: "
                             "3x3 / 7x7 layers on the bf16 matrix cores with three-term splits, fp32-grade accuracy, SLOWER than the fp32 "
                             "Winograd path since round 5; needs a library built with PMX_BUILD_BF16X3=1)")
        self._precision = precision
        self.engine = None
        self._make_engine(max_batch, mh, mw)

    def _make_engine(self, max_batch, mh, mw):
        # growth keeps the engine's state: weights (also those installed through detector.engine.set_weights / set_layer), options,
        # stream, capacities -- taken from the old context, which is destroyed BEFORE the larger one is created (the two never hold
        # their buffers and weight packs at the same time)
        st = None
        if self.engine is not None:
            st = self.engine.state()
            self.engine.close()
            self.engine = None
        try:
            self.engine = self._new_engine(max_batch, mh, mw, st)
        except Exception:
            # a failed growth (device out of memory for an oversized image / batch, rejected capacities) must not leave the
            # detector without a context: rebuild the previous one from the saved state, keep its capacity, re-raise
            self.engine = None
            if st is not None:
                self.engine = self._new_engine(*self._cap, st)
            raise
        self._cap = (max_batch, mh, mw)            # only once the new context is complete

    def _new_engine(self, max_batch, mh, mw, st):
        eng = native.Engine(self._gpu, max_batch=max_batch, max_h=mh, max_w=mw, gaussian_sigma=params['gaussian_sigma'])
        try:
            if st is not None:
                eng.load_state(st)
            elif self._weights is not None:
                eng.set_weights(self._weights)
            if self._precision == 'bf16x3':
                eng.set_option('precision', 1)
            if self._gpu_branch_peaks:
                # the reference's own GPU branch of compute_peaks_from_heatmaps (:111-133): 17x17 un-normalised kernel, zero
                # padding, '>=' NMS -- NOT the golden CPU semantics; off by default
                eng.set_option('peaks_gpu_branch', 1)
        except Exception:
            eng.close()
            raise
        return eng

    # ---- host helpers with the reference's names and semantics -------------------------------------
    def compute_optimal_size(self, orig_img, img_size, stride=8):
        """Network input size (w, h) for an image: the SHORT side becomes `img_size`, the long side follows the aspect ratio
        (round-half-even, as np.round does) and is then rounded UP to a multiple of `stride`; a square image takes the second rule
        for its height.  Same results as reference pose_detector.py:57-73 (tests/test_oracle_vs_reference.py)."""
        h, w = orig_img.shape[:2]
        ratio = h / w                                        # the reference divides / multiplies by this float: keep its rounding

        def long_side(exact):
            return -(-int(np.round(exact)) // stride) * stride
        if h < w:
            return (long_side(img_size / ratio), int(img_size))
        return (int(img_size), long_side(img_size * ratio))

    def preprocess(self, img):
        """reference pose_detector.py:426-431 (host version, for `model=` callables; the built-in network
        fuses it into the first kernel)."""
        x_data = img.astype('f')
        x_data /= 255
        x_data -= 0.5
        x_data = x_data.transpose(2, 0, 1)[None]
        return x_data

    def _grow(self, batch, h, w):
        """Re-create the device context when an input exceeds its capacity (buffers are sized at creation)."""
        mb, mh, mw = self._cap
        if batch <= mb and h * w <= mh * mw:
            return
        self._make_engine(max(mb, batch), max(mh, h), max(mw, w))

    # ---- the hot path -----------------------------------------------------------------------------
    def __call__(self, orig_img):
        """reference pose_detector.py:484-517"""
        orig_img = np.asarray(orig_img)
        if self.precise:
            return self.detect_precise(orig_img)
        return self.detect_batch([orig_img])[0]

    def pad_image(self, img, stride, pad_value):
        """reference pose_detector.py:46-55: pad bottom / right to a multiple of `stride`."""
        h, w, _ = img.shape
        pad = [0] * 2
        pad[0] = (stride - (h % stride)) % stride  # down
        pad[1] = (stride - (w % stride)) % stride  # right
        img_padded = np.empty((h + pad[0], w + pad[1], 3), np.uint8)
        img_padded[...] = np.asarray(pad_value, dtype=np.uint8)
        img_padded[:h, :w, :] = img
        return img_padded, pad

    def detect_precise(self, orig_img):
        """reference pose_detector.py:433-482: average the network outputs over `inference_scales`, resized
        (cv2 INTER_CUBIC, restated in resize_cubic_*) to the ORIGINAL resolution, then the same post-process at that
        resolution with img_len = orig_img_w (:478) and no coordinate rescale.  The four forward passes and the
        full-resolution post-process run on the GPU.  With the native network the cubic resizes and the accumulation run on
        the device too (pmx_precise_*); with a plugged-in `model=` callable they are host code as in the reference."""
        orig_img = np.ascontiguousarray(orig_img, dtype=np.uint8)
        orig_img_h, orig_img_w, _ = orig_img.shape
        if self.model is None:
            return self._detect_precise_device(orig_img)
        pafs_sum = 0
        heatmaps_sum = 0
        for scale in params['inference_scales']:
            multiplier = scale * params['inference_img_size'] / min(orig_img.shape[:2])            # :442
            img = resize_cubic_u8(orig_img, math.ceil(orig_img_w * multiplier), math.ceil(orig_img_h * multiplier))
            padded_img, pad = self.pad_image(img, params['downscale'], (104, 117, 123))              # :445
            p_h, p_w = padded_img.shape[:2]
            h1s, h2s = self.model(self.preprocess(padded_img))
            paf = np.asarray(_data(h1s[-1]), dtype=np.float32)[0]
            heat = np.asarray(_data(h2s[-1]), dtype=np.float32)[0]
            tmp_paf = np.ascontiguousarray(paf.transpose(1, 2, 0))                                   # :453
            tmp_heatmap = np.ascontiguousarray(heat.transpose(1, 2, 0))                              # :454
            tmp_paf = resize_cubic_f32(tmp_paf, p_w, p_h)                                            # :461
            tmp_paf = tmp_paf[:p_h - pad[0], :p_w - pad[1], :]                                       # :462
            pafs_sum = pafs_sum + resize_cubic_f32(tmp_paf, orig_img_w, orig_img_h)                  # :463
            ds = params['downscale']
            tmp_heatmap = resize_cubic_f32(tmp_heatmap, tmp_heatmap.shape[1] * ds, tmp_heatmap.shape[0] * ds)   # :465
            tmp_heatmap = tmp_heatmap[:p_h - pad[0], :p_w - pad[1], :]                               # :466
            heatmaps_sum = heatmaps_sum + resize_cubic_f32(tmp_heatmap, orig_img_w, orig_img_h)      # :467
        n = len(params['inference_scales'])
        self.pafs = (pafs_sum / n).transpose(2, 0, 1)                                                # :469
        self.heatmaps = (heatmaps_sum / n).transpose(2, 0, 1)                                        # :470
        # post-process at the original resolution on the device (resize to the same size is the identity)
        self._grow(1, 8, 8)
        self.engine.set_maps(np.ascontiguousarray(self.pafs)[None], np.ascontiguousarray(self.heatmaps)[None])
        self.engine.postprocess(orig_img_h, orig_img_w, img_len=orig_img_w, scale_xy=None)           # :475-481
        self.all_peaks = self.engine.peaks(0)
        return unpack_results(self.engine.results())[0]

    def _detect_precise_device(self, orig_img, fetch_maps=True):
        """detect_precise with everything between the uint8 image and the result record on the device."""
        try:
            return self.detect_precise_batch([orig_img], fetch_maps=fetch_maps)[0]
        finally:                                    # (also when the reference's IndexError condition is raised: the maps and peaks exist)
            if fetch_maps and getattr(self, 'pafs', None) is not None and np.ndim(self.pafs) == 4:
                self.pafs, self.heatmaps = self.pafs[0], self.heatmaps[0]

    def detect_precise_batch(self, imgs, fetch_maps=False, return_exceptions=False):
        """`detect_precise` (reference pose_detector.py:433-482) for a list of uint8 BGR images of ONE common size -> list of (poses, scores).
        The reference handles one image per call; here every inference scale runs the n images as ONE batch through the network (a single
        0.5x input is 23 x 31 feature maps -- too little for 256 CUs), the cubic resizes and the accumulation stay on the device per image,
        and the full-resolution post-process runs on the n averaged map sets at once.  Per image the result equals the single-image call up
        to the network's kernel-choice-by-launch-size rounding.  Native network only (`model=` callables: loop).
        Where the reference would raise for ONE image (IndexError, :197) the default raises as it does; `return_exceptions=True` puts the
        exception object into that image's slot and returns everybody else's result (a per-image loop over the reference loses nothing)."""
        if len(imgs) == 0:
            raise ValueError('detect_precise_batch needs at least one image')
        if self.model is not None:
            return [self.detect_precise(im) for im in imgs]
        if self._weights is None:
            raise RuntimeError('PoseDetector has no weights: pass weights_file=, weights= or model=')
        imgs = [np.ascontiguousarray(im, dtype=np.uint8) for im in imgs]
        shape = imgs[0].shape
        for im in imgs:
            if im.shape != shape or im.ndim != 3 or im.shape[2] != 3:
                raise ValueError('detect_precise_batch needs uint8 H x W x 3 images of one common size')
        orig_img_h, orig_img_w, _ = shape
        n = len(imgs)
        sizes = []
        for scale in params['inference_scales']:
            multiplier = scale * params['inference_img_size'] / min(shape[:2])                     # :442
            sizes.append((math.ceil(orig_img_h * multiplier), math.ceil(orig_img_w * multiplier)))
        ds = params['downscale']
        big = max(sizes)
        self._grow(n, -(-big[0] // ds) * ds, -(-big[1] // ds) * ds)
        batch = np.stack(imgs)
        self.engine.precise_begin(orig_img_h, orig_img_w, n)
        # the largest scale is enqueued FIRST (its chain of launches is the critical path of the sequence) but keeps its position in the
        # reference's loop: the parts are summed in slot order (:463,467)
        order = sorted(range(len(sizes)), key=lambda i: -sizes[i][0] * sizes[i][1]) if self.precise_largest_first else range(len(sizes))
        for slot in order:
            self.engine.precise_add_scale(batch, sizes[slot][0], sizes[slot][1], slot=slot)          # :443-467
        self.engine.precise_finish()                                                                 # :469-470
        if fetch_maps:
            self.pafs, self.heatmaps = self.engine.get_maps()                                       # (n, 38 | 19, H, W)
        self.engine.postprocess(orig_img_h, orig_img_w, img_len=orig_img_w, scale_xy=None)           # :475-481
        rec = self.engine.results()
        self.all_peaks = self.engine.peaks(0) if n == 1 else None                                    # :475 (kept as the reference keeps it; a batch has no single set)
        return unpack_results(rec, return_exceptions=return_exceptions)

    # ---- demo-chain helpers (reference pose_detector.py:267-424): host geometry that feeds the face / hand detectors -------
    _UNIT_BASE_LIMBS = (14, 3, 0, 13, 9)            # nose-neck, neck-left hip, neck-right hip, shoulder-ear (left, right)
    _UNIT_BASE_RATIO = (0.85, 2.2, 2.2, 0.85, 0.85)
    _UNIT_ALL_RATIO = (2.2, 1.7, 1.7, 2.2, 1.7, 1.7, 0.6, 0.93, 0.65, 0.85, 0.6, 0.93, 0.65, 0.85, 1, 0.2, 0.2, 0.25, 0.25)

    def compute_limbs_length(self, joints):
        """reference :267-277 -- per-limb length (0 where an endpoint is None) and the endpoint pairs."""
        pairs, lengths = [], np.zeros(len(params['limbs_point']))
        for i, (ja, jb) in enumerate(params['limbs_point']):
            a, b = joints[ja], joints[jb]
            if a is None or b is None:
                pairs.append(None)
                continue
            pairs.append([a, b])
            lengths[i] = np.linalg.norm(b[:-1] - a[:-1])
        return lengths, pairs

    def compute_unit_length(self, limbs_len):
        """reference :279-291 -- body 'unit length' from limb-length ratios; torso/head limbs are preferred."""
        base = limbs_len[list(self._UNIT_BASE_LIMBS)]
        known = base > 0
        if known.any():
            ratio = np.array(self._UNIT_BASE_RATIO)
            return np.sum(base[known] / ratio[known]) / np.count_nonzero(known)
        ratio = np.array(self._UNIT_ALL_RATIO)
        known = limbs_len > 0
        return np.sum(limbs_len[known] / ratio[known]) / np.count_nonzero(known)

    def get_unit_length(self, person_pose):
        """reference :293-297 (called by demo.py:32)"""
        return self.compute_unit_length(self.compute_limbs_length(person_pose)[0])

    # row j: how far (in unit lengths) the person's box extends above / below joint j when j is the top / bottom anchor,
    # and the preference of joint j as top / bottom anchor (smaller = preferred)   (reference :312-313, :342-343)
    _PERSON_TOP = ((0.9, 4), (1.9, 5), (1.9, 6), (2.9, 12), (3.7, 16), (1.9, 7), (2.9, 13), (3.7, 17), (4.0, 8), (5.5, 10), (7.0, 14),
                   (4.0, 9), (5.5, 11), (7.0, 15), (0.7, 2), (0.8, 3), (0.7, 0), (0.8, 1))
    _PERSON_BOTTOM = ((6.9, 9), (5.9, 6), (5.9, 7), (4.9, 14), (4.1, 16), (5.9, 8), (4.9, 15), (4.1, 17), (3.8, 4), (2.3, 2), (0.8, 0),
                      (3.8, 5), (2.3, 3), (0.8, 1), (7.1, 10), (7.0, 11), (7.1, 12), (7.0, 13))

    def crop_person(self, img, person_pose, unit_length):
        """reference :311-352: box around all visible joints -- 0.3 units to the sides, above / below by the padding of the
        preferred anchor joints.  One pass like the reference's, including its `elif` coupling: a joint that becomes the
        top anchor (or the new minimum) is not considered as bottom anchor (or maximum) in the same step."""
        top_i = bot_i = None                    # None = nothing chosen yet (the reference's sentinel of priority sys.maxsize)
        top_pos, left_pos = float('inf'), float('inf')
        bottom_pos, right_pos = 0, 0
        for i, joint in enumerate(person_pose):
            if not joint[2] > 0:
                continue
            if top_i is None or self._PERSON_TOP[i][1] < self._PERSON_TOP[top_i][1]:
                top_i = i
            elif bot_i is None or self._PERSON_BOTTOM[i][1] < self._PERSON_BOTTOM[bot_i][1]:
                bot_i = i
            if joint[1] < top_pos:
                top_pos = joint[1]
            elif joint[1] > bottom_pos:
                bottom_pos = joint[1]
            if joint[0] < left_pos:
                left_pos = joint[0]
            elif joint[0] > right_pos:
                right_pos = joint[0]
        if top_i is None or bot_i is None:      # the reference indexes its 18-entry padding tables with the sentinel 18
            raise IndexError('list index out of range')
        bbox = (int(left_pos - 0.3 * unit_length), int(top_pos - self._PERSON_TOP[top_i][0] * unit_length),
                int(right_pos + 0.3 * unit_length), int(bottom_pos + self._PERSON_BOTTOM[bot_i][0] * unit_length))
        return self.crop_image(img, bbox), bbox

    def crop_image(self, img, bbox):
        """reference :401-424 -- crop with zero padding where the box leaves the image."""
        left, top, right, bottom = bbox
        h, w, ch = img.shape
        out = np.zeros((bottom - top, right - left, ch), dtype=np.uint8)
        x0, y0, x1, y1 = max(0, left), max(0, top), min(w, right), min(h, bottom)
        if x1 > x0 and y1 > y0:
            out[y0 - top:y1 - top, x0 - left:x1 - left] = img[y0:y1, x0:x1]
        return out

    def crop_around_keypoint(self, img, keypoint, crop_size):
        """reference :299-309"""
        x, y = keypoint
        bbox = (int(x - crop_size), int(y - crop_size), int(x + crop_size), int(y + crop_size))
        return self.crop_image(img, bbox), bbox

    def crop_face(self, img, person_pose, unit_length):
        """reference :354-369 (demo.py:36): box around the nose, 1.2 units up, 0.8 down, 1 unit to each side."""
        nose = person_pose[JointType.Nose]
        if not nose[2] > 0:
            return None, None
        bbox = (int(nose[0] - unit_length), int(nose[1] - unit_length * 1.2),
                int(nose[0] + unit_length), int(nose[1] + unit_length * 0.8))
        return self.crop_image(img, bbox), bbox

    def crop_hands(self, img, person_pose, unit_length):
        """reference :371-399 (demo.py:44): a 0.95-unit box centred 30 % past the wrist along the forearm."""
        hands = {"left": None, "right": None}
        for side, wrist_j, elbow_j in (("left", JointType.LeftHand, JointType.LeftElbow),
                                       ("right", JointType.RightHand, JointType.RightElbow)):
            if person_pose[wrist_j][2] > 0:
                center = person_pose[wrist_j][:-1]          # a view, updated in place exactly as the reference does
                if person_pose[elbow_j][2] > 0:
                    center += (0.3 * (person_pose[wrist_j][:-1] - person_pose[elbow_j][:-1])).astype(center.dtype)
                hand_img, bbox = self.crop_around_keypoint(img, center, unit_length * 0.95)
                hands[side] = {"img": hand_img, "bbox": bbox}
        return hands

    def detect_batch(self, imgs):
        """Batched `__call__`: list of H x W x 3 uint8 BGR images -> list of (poses, scores), one per image in the order given (the
        reference handles one image per call).  Images of ONE common size run as a uniform batch; images of DIFFERENT sizes (the
        reference picks the network size per image, :490-493) run as a mixed batch -- one launch per layer over all size classes
        (include/pose_mi355x.h::pmx_detect_images) -- instead of one call per image.  Per image the result is what `__call__` returns
        for it, up to the kernel-choice-by-launch-size rounding of the network."""
        imgs = [np.asarray(im) for im in imgs]
        if len(imgs) == 0:
            raise ValueError('detect_batch needs at least one image')
        for im in imgs:
            if im.dtype != np.uint8 or im.ndim != 3 or im.shape[2] != 3:
                raise ValueError('detect_batch needs uint8 H x W x 3 images')
        shape = imgs[0].shape
        if any(im.shape != shape for im in imgs):
            return self._detect_mixed(imgs)
        orig_h, orig_w, _ = shape
        input_w, input_h = self.compute_optimal_size(imgs[0], params['inference_img_size'])   # :490
        map_w, map_h = self.compute_optimal_size(imgs[0], params['heatmap_size'])            # :491
        B = len(imgs)
        self._grow(B, input_h, input_w)
        batch = np.stack(imgs)
        scale = np.tile(np.array([orig_w / map_w, orig_h / map_h], dtype=np.float64), (B, 1))   # :513-514
        if self.model is None:
            if self.engine.weights_missing():
                raise RuntimeError('PoseDetector has no weights: pass weights_file=, weights= or model=')
            # cv2.resize (:493, identity when the size is unchanged) + preprocess + network + post-process, all on the GPU
            self.engine.forward_u8_resized(batch, input_h, input_w)                              # :493-499
            self.engine.postprocess(map_h, map_w, img_len=map_w, scale_xy=scale)                 # :501-516
        else:
            pafs, heats = [], []
            if (input_h, input_w) != (orig_h, orig_w):
                batch = self.engine.resize_u8(batch, input_h, input_w)                           # :493 on the device
            for im in batch:
                h1s, h2s = self.model(self.preprocess(im))                                      # :499
                pafs.append(np.asarray(_data(h1s[-1]), dtype=np.float32)[0])
                heats.append(np.asarray(_data(h2s[-1]), dtype=np.float32)[0])
            self.engine.set_maps(np.stack(pafs), np.stack(heats))
            self.engine.postprocess(map_h, map_w, img_len=map_w, scale_xy=scale)                # :501-516
        return unpack_results(self.engine.results())

    def _detect_mixed(self, imgs):
        """detect_batch for images of different sizes.  Images are sorted by network size (few segments), results return in the caller's
        order.  `model=` callables: one call per image, as the reference does."""
        if self.model is not None:
            return [self.detect_batch([im])[0] for im in imgs]
        if self._weights is None or self.engine.weights_missing():
            raise RuntimeError('PoseDetector has no weights: pass weights_file=, weights= or model=')
        net, mp = [], []
        for im in imgs:
            w_, h_ = self.compute_optimal_size(im, params['inference_img_size'])                 # :490
            mw_, mh_ = self.compute_optimal_size(im, params['heatmap_size'])                     # :491
            net.append((h_, w_))
            mp.append((mh_, mw_))
        order = sorted(range(len(imgs)), key=lambda i: (net[i], mp[i], i))
        px = sum(h * w for h, w in net)
        mb, mh, mw = self._cap
        if len(imgs) > mb or px > mb * mh * mw:
            # capacity is a pixel budget (max_batch x max_h x max_w) and an image count: grow both as needed
            nb = max(mb, len(imgs))
            side = max(mh * mw, -(-px // nb))
            self._make_engine(nb, mh, -(-side // (mh * 8)) * 8)          # (max_w: a multiple of 8)
        self.engine.detect_images([imgs[i] for i in order], [net[i] for i in order], [mp[i] for i in order])
        res = unpack_results(self.engine.results(), return_exceptions=True)
        out = [None] * len(imgs)
        for k, i in enumerate(order):
            out[i] = res[k]
        for r in out:                                  # (the reference would have raised on that image's call)
            if isinstance(r, Exception):
                raise r
        return out

    def detect_maps(self, paf, heat, map_h, map_w, img_len=None, scale_xy=None):
        """Post-process only (pose_detector.py:501-517) on network outputs paf (B,38,h,w), heat (B,19,h,w)."""
        self.engine.set_maps(paf, heat)
        self.engine.postprocess(map_h, map_w, img_len=map_w if img_len is None else img_len, scale_xy=scale_xy)
        return unpack_results(self.engine.results())


def _data(v):
    return getattr(v, 'data', v)      # chainer.Variable-like or plain array


def unpack_results(records, return_exceptions=False):
    """Device result records -> [(poses, scores)] with the reference's return shapes and error behaviour.
    `return_exceptions`: an image on which the reference would raise (IndexError, :197) gets the exception OBJECT in its list slot
    instead of aborting the whole batch -- the reference, called once per image, only ever fails that image."""
    out = []
    for r in records:
        st = int(r['status'])
        if st & native.IMG_TRIPLE_MATCH:
            # the reference fails with IndexError at pose_detector.py:197 when a third subset matches
            err = IndexError('list assignment index out of range')
            if not return_exceptions:
                raise err
            out.append(err)
            continue
        # (capacity bits never reach this point: the library grows its buffers and re-runs the post-process, the reference
        #  has no limits on peaks / candidates / subsets / people)
        n = int(r['n_people'])
        if int(r['n_peaks']) == 0:
            out.append((np.empty((0, len(JointType), 3)), np.empty(0)))      # :509-510
        elif n == 0:
            out.append((np.array([]), np.empty(0)))                          # :264 np.array([]) / :516
        else:
            out.append((r['poses'][:n].copy(), r['scores'][:n].copy()))
    return out


# ---- cv2.resize(..., interpolation=cv2.INTER_CUBIC) restated (used by detect_precise only) ---------------------------
def _cubic_taps(dst, src):
    """Source indices (4 taps, replicate border) and float32 coefficients of OpenCV's bicubic (A = -0.75):
    fx = float((dx + 0.5) * scale - 0.5), scale = 1 / (dst / src) in double; sx = floor(fx); taps sx-1 .. sx+2."""
    scale = 1.0 / (float(dst) / float(src))
    d = np.arange(dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    x = (f - s.astype(np.float32)).astype(np.float32)
    A = np.float32(-0.75)
    one = np.float32(1.0)
    c0 = ((A * (x + one) - np.float32(5) * A) * (x + one) + np.float32(8) * A) * (x + one) - np.float32(4) * A
    c1 = ((A + np.float32(2)) * x - (A + np.float32(3))) * x * x + one
    c2 = ((A + np.float32(2)) * (one - x) - (A + np.float32(3))) * (one - x) * (one - x) + one
    c3 = one - c0 - c1 - c2
    idx = np.stack([np.clip(s + k, 0, src - 1) for k in (-1, 0, 1, 2)])
    return idx, np.stack([c0, c1, c2, c3]).astype(np.float32)


def resize_cubic_f32(img, dst_w, dst_h):
    """`cv2.resize(float32 H x W x C, (dst_w, dst_h), interpolation=cv2.INTER_CUBIC)` restated (pose_detector.py:461-467):
    horizontal 4-tap pass then vertical 4-tap pass, float32 products summed left to right.  PARITY UNPINNED: OpenCV is
    not installable here (its SIMD paths may fuse multiply-adds)."""
    img = np.ascontiguousarray(img, dtype=np.float32)
    src_h, src_w = img.shape[:2]
    if (src_w, src_h) == (dst_w, dst_h):
        return img.copy()
    ix, cx = _cubic_taps(dst_w, src_w)
    iy, cy = _cubic_taps(dst_h, src_h)
    rows = img[:, ix[0]] * cx[0][None, :, None]
    for k in (1, 2, 3):
        rows = rows + img[:, ix[k]] * cx[k][None, :, None]
    out = rows[iy[0]] * cy[0][:, None, None]
    for k in (1, 2, 3):
        out = out + rows[iy[k]] * cy[k][:, None, None]
    return out


def resize_cubic_u8(img, dst_w, dst_h):
    """`cv2.resize(uint8 image, (dst_w, dst_h), interpolation=cv2.INTER_CUBIC)` restated (pose_detector.py:443): OpenCV's
    fixed-point path -- coefficients `saturate_cast<short>(c * 2048)`, int32 horizontal pass, vertical pass
    `(sum + (1 << 21)) >> 22`, saturated to uint8.  PARITY UNPINNED (no cv2 to compare against)."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    src_h, src_w = img.shape[:2]
    if (src_w, src_h) == (dst_w, dst_h):
        return img.copy()
    ix, cx = _cubic_taps(dst_w, src_w)
    iy, cy = _cubic_taps(dst_h, src_h)
    ax = np.clip(np.rint(cx * np.float32(2048)), -32768, 32767).astype(np.int64)
    ay = np.clip(np.rint(cy * np.float32(2048)), -32768, 32767).astype(np.int64)
    src = img.astype(np.int64)
    rows = sum(src[:, ix[k]] * ax[k][None, :, None] for k in range(4))
    out = sum(rows[iy[k]] * ay[k][:, None, None] for k in range(4))
    out = (out + (1 << 21)) >> 22
    return np.clip(out, 0, 255).astype(np.uint8)


# ---- visualisation + CLI (reference pose_detector.py:520-579); host-only, no compute on the hot path ---------------------
LIMB_COLORS = [
    [0, 255, 0], [0, 255, 85], [0, 255, 170], [0, 255, 255], [0, 170, 255],
    [0, 85, 255], [255, 0, 0], [255, 85, 0], [255, 170, 0], [255, 255, 0],
    [255, 0, 85], [170, 255, 0], [85, 255, 0], [170, 0, 255], [0, 0, 255],
    [0, 0, 255], [255, 0, 255], [170, 0, 255], [255, 0, 170]]
JOINT_COLORS = [
    [255, 0, 0], [255, 85, 0], [255, 170, 0], [255, 255, 0], [170, 255, 0],
    [85, 255, 0], [0, 255, 0], [0, 255, 85], [0, 255, 170], [0, 255, 255],
    [0, 170, 255], [0, 85, 255], [0, 0, 255], [85, 0, 255], [170, 0, 255],
    [255, 0, 255], [255, 0, 170], [255, 0, 85]]


def _draw_line(canvas, p1, p2, color, thickness=2):
    """All pixels within thickness/2 of the segment p1-p2 (stands in for cv2.line; OpenCV's exact raster is unpinned)."""
    h, w = canvas.shape[:2]
    x1, y1, x2, y2 = float(p1[0]), float(p1[1]), float(p2[0]), float(p2[1])
    r = thickness / 2.0
    xa, xb = int(max(0, math.floor(min(x1, x2) - r))), int(min(w - 1, math.ceil(max(x1, x2) + r)))
    ya, yb = int(max(0, math.floor(min(y1, y2) - r))), int(min(h - 1, math.ceil(max(y1, y2) + r)))
    if xa > xb or ya > yb:
        return
    ys, xs = np.mgrid[ya:yb + 1, xa:xb + 1]
    dx, dy = x2 - x1, y2 - y1
    L2 = dx * dx + dy * dy
    t = np.clip(((xs - x1) * dx + (ys - y1) * dy) / L2, 0, 1) if L2 > 0 else np.zeros_like(xs, dtype=float)
    d2 = (xs - (x1 + t * dx)) ** 2 + (ys - (y1 + t * dy)) ** 2
    canvas[ya:yb + 1, xa:xb + 1][d2 <= r * r + 0.25] = color


def _draw_disc(canvas, c, radius, color):
    h, w = canvas.shape[:2]
    cx, cy = int(c[0]), int(c[1])
    xa, xb, ya, yb = max(0, cx - radius), min(w - 1, cx + radius), max(0, cy - radius), min(h - 1, cy + radius)
    if xa > xb or ya > yb:
        return
    ys, xs = np.mgrid[ya:yb + 1, xa:xb + 1]
    canvas[ya:yb + 1, xa:xb + 1][(xs - cx) ** 2 + (ys - cy) ** 2 <= radius * radius] = color


def draw_person_pose(orig_img, poses):
    """reference pose_detector.py:520-553: limbs (thickness 2, shoulder-ear limbs 9 and 13 skipped) then joints (discs
    of radius 3) on a copy of the BGR image."""
    if len(poses) == 0:
        return orig_img
    canvas = np.array(orig_img, copy=True)
    rounded = np.asarray(poses).round().astype('i')
    for pose in rounded:
        for i, (limb, color) in enumerate(zip(params['limbs_point'], LIMB_COLORS)):
            if i != 9 and i != 13:
                a, b = pose[int(limb[0])], pose[int(limb[1])]
                if a[2] != 0 and b[2] != 0:
                    _draw_line(canvas, a[:2], b[:2], color, 2)
    for pose in rounded:
        for (x, y, v), color in zip(pose, JOINT_COLORS):
            if v != 0:
                _draw_disc(canvas, (x, y), 3, color)
    return canvas


def imread_bgr(path):
    """cv2.imread equivalent for the CLI: 8-bit BGR, alpha dropped (pose_detector.py:571)."""
    from PIL import Image
    return np.ascontiguousarray(np.asarray(Image.open(path).convert('RGB'))[:, :, ::-1])


def imwrite_bgr(path, img):
    from PIL import Image
    Image.fromarray(np.ascontiguousarray(img[:, :, ::-1])).save(path)


def main(argv=None):
    """`python -m <package>.pose_detector posenet weights.npz --img X [--gpu N] [--precise]` (reference :555-579)."""
    import argparse
    parser = argparse.ArgumentParser(description='Pose detector')
    parser.add_argument('arch', choices=['posenet'], default='posenet', help='Model architecture')
    parser.add_argument('weights', help='weights file path (Chainer NPZ)')
    parser.add_argument('--img', '-i', default=None, help='image file path')
    parser.add_argument('--gpu', '-g', type=int, default=-1, help='GPU ID (negative value selects GPU 0: there is no CPU path)')
    parser.add_argument('--precise', action='store_true', help='do precise inference')
    parser.add_argument('--out', '-o', default='result.png', help='output image path')
    args = parser.parse_args(argv)
    pose_detector = PoseDetector(args.arch, args.weights, device=args.gpu, precise=args.precise)
    img = imread_bgr(args.img)
    poses, _ = pose_detector(img)
    img = draw_person_pose(img, poses)
    print('Saving result into %s...' % args.out)
    imwrite_bgr(args.out, img)
    return 0


if __name__ == '__main__':
    raise SystemExit(main())
