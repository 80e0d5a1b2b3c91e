"""TEST INFRASTRUCTURE -- runs the *verbatim* reference code from /root/reference (read-only, imported where it lies).

Used only by `oracle/make_golden.py` and the `tests/test_oracle_vs_reference.py` / `tests/test_reference_network.py`
CPU tests, and only in the authoring container (where /root/reference exists).  It never runs on the GPU box and is
never imported by the product package.

The reference cannot be imported as-is here: chainer, cupy, cv2 and pycocotools are not installed.  Its own code is
pure Python on top of a handful of third-party calls, so those calls -- and only those -- are backed by stand-ins in
`sys.modules`; everything the reference itself wrote (layer tables, dataflow, pre/post-process, detectors) then executes
unchanged:

  third-party call the reference makes                      stand-in used here
  --------------------------------------------------------  -------------------------------------------------------------
  chainer.Chain(**links)                                    attribute holder that remembers the link names in order
  L.Convolution2D(in, out, ksize, stride, pad)              object with W.data (OIHW) / b.data; __call__ = torch-CPU fp32
                                                            conv2d (cross-correlation, zero pad) -- Chainer's own CPU conv
                                                            is im2col + BLAS: same sums, summation order undefined in both
  F.relu, F.max_pooling_2d(ksize, stride) [cover_all=True]  torch relu / max_pool2d(ceil_mode=True), F.concat -> torch.cat
  F.resize_images(x, (h, w))                                oracle/postprocess_ref.py::resize_images_ref (restated; Chainer
                                                            "2.0+" is unpinned -- named third-party step)
  serializers.load_npz(file, chain)                         NumPy reader of `<link>/W`, `<link>/b`
  cv2.resize / cv2.flip / cv2.imread                        oracle/resize_ref.py, oracle/precise_ref.py (restated OpenCV
                                                            INTER_LINEAR u8 / INTER_CUBIC u8+f32; OpenCV is unpinned), PIL
  cuda.get_array_module                                     -> numpy (forces the reference's CPU branch, :80-82)

plus ONE NumPy>=1.23 compatibility shim: `pose_detector.py:147` indexes with a *list* of arrays
(`paf[0][np.hsplit(integ_points, 2)]`), which NumPy < 1.23 treated as a tuple index (`paf[0][ys, xs]`); `np.hsplit` is
wrapped to return a tuple -- the intended (and historically actual) meaning.

Nothing of the reference is copied.
"""
import os
import sys
import types
import warnings

import numpy as np

REFERENCE_DIR = os.environ.get('PMX_REFERENCE_DIR', '/root/reference')


def reference_available():
    return os.path.isfile(os.path.join(REFERENCE_DIR, 'pose_detector.py'))


_cached = None
_mods = None


# ---- stand-ins for the third-party calls ---------------------------------------------------------------------------------
class RefVar(object):
    """Stands in for chainer.Variable: `.data` is a NumPy array, indexing yields another RefVar (`h1s[-1][0].data`,
    pose_detector.py:453)."""

    def __init__(self, data):
        self.data = np.ascontiguousarray(data)

    def __getitem__(self, idx):
        return RefVar(self.data[idx])

    @property
    def shape(self):
        return self.data.shape


def _arr(v):
    return v.data if isinstance(v, RefVar) else np.asarray(v)


class _Param(object):
    def __init__(self, data):
        self.data = data


class TorchConvolution2D(object):
    """L.Convolution2D(in_channels, out_channels, ksize, stride, pad): y = cross-correlation(x, W) + b."""

    def __init__(self, in_channels, out_channels, ksize, stride=1, pad=0):
        self.in_channels, self.out_channels, self.ksize, self.stride, self.pad = in_channels, out_channels, ksize, stride, pad
        self.W = _Param(np.zeros((out_channels, in_channels, ksize, ksize), np.float32))
        self.b = _Param(np.zeros((out_channels,), np.float32))

    def __call__(self, x):
        import torch
        import torch.nn.functional as TF
        with torch.no_grad():
            y = TF.conv2d(torch.from_numpy(np.ascontiguousarray(_arr(x), dtype=np.float32)), torch.from_numpy(self.W.data),
                          torch.from_numpy(self.b.data), stride=self.stride, padding=self.pad)
        return RefVar(y.numpy())


def _relu(x):
    import torch
    return RefVar(torch.relu(torch.from_numpy(_arr(x))).numpy())


def _max_pooling_2d(x, ksize, stride=None, pad=0, cover_all=True):
    import torch
    import torch.nn.functional as TF
    return RefVar(TF.max_pool2d(torch.from_numpy(_arr(x)), ksize, stride or ksize, pad, ceil_mode=bool(cover_all)).numpy())


def _concat(xs, axis=1):
    return RefVar(np.concatenate([_arr(x) for x in xs], axis=axis))


def _resize_images(x, output_shape):
    from . import postprocess_ref as P
    a = _arr(x)
    return RefVar(np.stack([P.resize_images_ref(m, int(output_shape[0]), int(output_shape[1])) for m in a]))


class Chain(object):
    """Stands in for chainer.Chain (models/CocoPoseNet.py:20,24): links become attributes, declaration order kept."""

    def __init__(self, **links):
        self._link_names = list(links)
        self.__dict__.update(links)

    def namedlinks(self):
        return [(n, getattr(self, n)) for n in self._link_names]

    def to_gpu(self):
        raise RuntimeError('the reference runs on its CPU branch here')


def _load_npz(path, chain):
    """chainer.serializers.load_npz for a Chain of Convolution2D links: arrays `<link>/W`, `<link>/b`."""
    with np.load(path) as z:
        for name, link in chain.namedlinks():
            W, b = z[name + '/W'], z[name + '/b']
            if W.shape != link.W.data.shape or b.shape != link.b.data.shape:
                raise ValueError('%s: shape mismatch %s vs %s' % (name, W.shape, link.W.data.shape))
            link.W.data = np.ascontiguousarray(W, dtype=np.float32)
            link.b.data = np.ascontiguousarray(b, dtype=np.float32)


INTER_LINEAR, INTER_CUBIC = 1, 2


def _cv2_resize(src, dsize, fx=0, fy=0, interpolation=INTER_LINEAR):
    from . import precise_ref, resize_ref
    src = np.asarray(src)
    if not dsize or dsize[0] == 0:
        dsize = (int(np.rint(src.shape[1] * fx)), int(np.rint(src.shape[0] * fy)))      # cvRound(src * f)
    w, h = int(dsize[0]), int(dsize[1])
    if interpolation == INTER_CUBIC:
        if src.dtype == np.uint8:
            return precise_ref.resize_cubic_u8_ref(src, w, h)
        return precise_ref.resize_cubic_f32_ref(src.astype(np.float32), w, h)
    if src.dtype != np.uint8:
        raise NotImplementedError('cv2.resize stand-in: INTER_LINEAR is restated for uint8 only')
    return resize_ref.resize_linear_u8(src, w, h)


def _cv2_flip(src, code):
    if code != 1:
        raise NotImplementedError
    return np.ascontiguousarray(np.asarray(src)[:, ::-1])


def _cv2_imread(path):
    from PIL import Image
    return np.ascontiguousarray(np.asarray(Image.open(path).convert('RGB'))[:, :, ::-1])     # 8-bit BGR, alpha dropped


class _NullCtx(object):
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _install_stand_ins():
    _mod('cv2', INTER_CUBIC=INTER_CUBIC, INTER_LINEAR=INTER_LINEAR, resize=_cv2_resize, flip=_cv2_flip, imread=_cv2_imread)
    cuda = _mod('chainer.cuda', get_array_module=lambda *a: np)  # forces CPU branch, :80-82
    F = _mod('chainer.functions', relu=_relu, max_pooling_2d=_max_pooling_2d, concat=_concat, resize_images=_resize_images)
    L = _mod('chainer.links', Convolution2D=TorchConvolution2D)
    L.caffe = _mod('chainer.links.caffe')
    ser = _mod('chainer.serializers', load_npz=_load_npz)
    ds = _mod('chainer.dataset', DatasetMixin=object)
    _mod('chainer', Chain=Chain, cuda=cuda, functions=F, links=L, serializers=ser, dataset=ds, Variable=RefVar,
         using_config=lambda *a, **k: _NullCtx())
    _mod('pycocotools')
    _mod('pycocotools.coco', COCO=object)


def import_reference_modules():
    """{'pose_detector', 'coco_data_loader', 'face_detector', 'hand_detector', 'entity', 'CocoPoseNet', 'FaceNet', 'HandNet'}
    -> the reference's own modules / classes, imported verbatim on top of the stand-ins."""
    global _mods
    if _mods is not None:
        return _mods
    if not reference_available():
        raise RuntimeError('reference not present at %s' % REFERENCE_DIR)
    _install_stand_ins()
    sys.path.insert(0, REFERENCE_DIR)
    try:
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            import pose_detector as PD          # noqa: E402
            import coco_data_loader as CDL      # noqa: E402
            import face_detector as FD          # noqa: E402
            import hand_detector as HD          # noqa: E402
            import entity as EN                 # noqa: E402
            from models.CocoPoseNet import CocoPoseNet
            from models.FaceNet import FaceNet
            from models.HandNet import HandNet
    finally:
        sys.path.remove(REFERENCE_DIR)
    _mods = dict(pose_detector=PD, coco_data_loader=CDL, face_detector=FD, hand_detector=HD, entity=EN,
                 CocoPoseNet=CocoPoseNet, FaceNet=FaceNet, HandNet=HandNet)
    return _mods


def import_reference():
    """Returns (pose_detector_module, coco_data_loader_module, detector_instance, label_gen)."""
    global _cached
    if _cached is not None:
        return _cached
    m = import_reference_modules()
    PD, CDL = m['pose_detector'], m['coco_data_loader']
    det = PD.PoseDetector(model=object(), device=-1)          # pose_detector.py:19-20
    gen = CDL.CocoDataLoader.__new__(CDL.CocoDataLoader)      # label generator without COCO
    _cached = (PD, CDL, det, gen)
    return _cached


class hsplit_shim(object):
    """Context manager: NumPy>=1.23 shim for pose_detector.py:147 (see module docstring)."""

    def __enter__(self):
        self._orig = np.hsplit
        orig = self._orig
        np.hsplit = lambda a, n: tuple(orig(a, n))
        return self

    def __exit__(self, *exc):
        np.hsplit = self._orig
        return False


# ---- the reference's own networks ------------------------------------------------------------------------------------------
_ARCH_CLASS = {'posenet': 'CocoPoseNet', 'facenet': 'FaceNet', 'handnet': 'HandNet'}


def ref_model(arch, weights):
    """Instantiate the reference's own Chain (`params['archs'][arch]()`, entity.py:50-54) and fill it with
    weights {name: (W OIHW, b)}.  Every link of the Chain must be present in `weights` and vice versa."""
    m = import_reference_modules()
    cls = m['entity'].params['archs'][arch]
    assert cls is m[_ARCH_CLASS[arch]]
    model = cls()
    names = [n for n, _ in model.namedlinks()]
    if set(names) != set(weights):
        raise ValueError('weight names differ from the reference Chain: %s' % sorted(set(names) ^ set(weights)))
    for n, link in model.namedlinks():
        W, b = weights[n]
        assert W.shape == link.W.data.shape and b.shape == link.b.data.shape, n
        link.W.data = np.ascontiguousarray(W, dtype=np.float32)
        link.b.data = np.ascontiguousarray(b, dtype=np.float32)
    return model


def ref_layer_table(arch):
    """[(name, cin, cout, ksize, stride, pad)] read off the instantiated reference Chain, declaration order."""
    m = import_reference_modules()
    model = m[_ARCH_CLASS[arch]]()
    return [(n, l.in_channels, l.out_channels, l.ksize, l.stride, l.pad) for n, l in model.namedlinks()]


def ref_network_forward(arch, weights, x, all_stages=False):
    """The reference's `model(x)` (models/CocoPoseNet.py:132-262, FaceNet.py:78-160, HandNet.py) run verbatim.
    posenet -> (paf, heat) of the last stage (or the lists of all six); facenet / handnet -> heat of the last stage."""
    model = ref_model(arch, weights)
    out = model(np.ascontiguousarray(x, dtype=np.float32))
    if arch == 'posenet':
        pafs, heats = out
        if all_stages:
            return [p.data for p in pafs], [h.data for h in heats]
        return pafs[-1].data, heats[-1].data
    return [h.data for h in out] if all_stages else out[-1].data


def ref_pose_detector(weights=None, weights_file=None, precise=False):
    """The reference's PoseDetector on its CPU branch: `PoseDetector(model=<its own CocoPoseNet>)` or, with `weights_file`,
    `PoseDetector('posenet', weights_file)` (its own load_npz path, :23-26)."""
    m = import_reference_modules()
    PD = m['pose_detector']
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):      # "Loading the model..."
        if weights_file is not None:
            return PD.PoseDetector('posenet', weights_file, device=-1, precise=precise)
        return PD.PoseDetector(model=ref_model('posenet', weights), device=-1, precise=precise)


def ref_call(detector, img):
    """`detector(img)` with the NumPy shim active; returns (poses, scores) exactly as the reference returns them."""
    with hsplit_shim(), warnings.catch_warnings():
        warnings.simplefilter('ignore')
        return detector(img)


def ref_postprocess(heatmaps, pafs, img_len, orig_w=None, orig_h=None):
    """Run the VERBATIM reference post-process (pose_detector.py:508-516) on full-size maps.

    heatmaps: (19, H, W) float32, pafs: (38, H, W) float32 (already resized to map size).
    Returns dict with all_peaks (before rescale), connections (list of 19), subsets, poses, scores,
    or raises whatever the reference raises (IndexError at :197 for >2 matching subsets).
    """
    PD, _, det, _ = import_reference()
    map_h, map_w = heatmaps.shape[1:]
    if orig_w is None:
        orig_w = map_w
    if orig_h is None:
        orig_h = map_h
    out = {}
    with hsplit_shim(), warnings.catch_warnings():
        warnings.simplefilter('ignore')
        all_peaks = det.compute_peaks_from_heatmaps(heatmaps)
        if len(all_peaks) == 0:
            out['all_peaks'] = np.zeros((0, 5))
            out['connections'] = [np.zeros((0, 3)) for _ in range(19)]
            out['subsets'] = np.zeros((0, 20))
            out['poses'] = np.empty((0, 18, 3))
            out['scores'] = np.empty(0)
            return out
        out['all_peaks'] = all_peaks.copy()
        conns = det.compute_connections(pafs, all_peaks, img_len, PD.params)
        out['connections'] = [np.asarray(c, dtype=np.float64).reshape(-1, 3) for c in conns]
        subsets = det.grouping_key_points(conns, all_peaks, PD.params)
        out['subsets'] = subsets.copy()
        all_peaks[:, 1] *= orig_w / map_w      # pose_detector.py:513
        all_peaks[:, 2] *= orig_h / map_h      # pose_detector.py:514
        out['poses'] = det.subsets_to_pose_array(subsets, all_peaks)
        out['scores'] = subsets[:, -2]
    return out
