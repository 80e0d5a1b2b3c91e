"""TEST INFRASTRUCTURE -- imports the *verbatim* reference post-process from /root/reference.

Only `oracle/make_golden.py` and `tests/test_oracle_vs_reference.py` use this, and only in the
authoring container (where /root/reference exists).  It never runs on the GPU box and is never
imported by the product package.

The reference (`pose_detector.py`) cannot be imported as-is here: chainer, cupy, cv2 and
pycocotools are not installed.  The post-process half (`pose_detector.py:75-265`) is pure
NumPy + SciPy, so we stub the missing third-party modules in `sys.modules` and apply ONE
NumPy>=1.23 compatibility shim:

  * `pose_detector.py:147` indexes with a *list* of arrays (`paf[0][np.hsplit(integ_points, 2)]`),
    which NumPy < 1.23 treated as a tuple index (`paf[0][ys, xs]`).  We wrap `np.hsplit` so that
    it returns a tuple -- the intended (and historically actual) meaning.

Nothing of the reference is copied; it is imported read-only from where it lies.
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


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def import_reference():
    """Returns (pose_detector_module, coco_data_loader_module, detector_instance, label_gen)."""
    global _cached
    if _cached is not None:
        return _cached
    if not reference_available():
        raise RuntimeError('reference not present at %s' % REFERENCE_DIR)

    class Chain(object):  # stands in for chainer.Chain (models/CocoPoseNet.py:20,24)
        def __init__(self, **links):
            self.__dict__.update(links)

    saved = {k: sys.modules.get(k) for k in (
        'cv2', 'chainer', 'chainer.cuda', 'chainer.functions', 'chainer.links',
        'chainer.links.caffe', 'chainer.serializers', 'chainer.dataset',
        'pycocotools', 'pycocotools.coco', 'entity', 'models', 'pose_detector',
        'coco_data_loader')}

    _mod('cv2', INTER_CUBIC=2)
    cuda = _mod('chainer.cuda', get_array_module=lambda *a: np)  # forces CPU branch, :80-82
    F = _mod('chainer.functions')
    L = _mod('chainer.links', Convolution2D=lambda **kw: kw)
    L.caffe = _mod('chainer.links.caffe')
    ser = _mod('chainer.serializers')
    ds = _mod('chainer.dataset', DatasetMixin=object)
    _mod('chainer', Chain=Chain, cuda=cuda, functions=F, links=L, serializers=ser, dataset=ds)
    _mod('pycocotools')
    _mod('pycocotools.coco', COCO=object)

    sys.path.insert(0, REFERENCE_DIR)
    try:
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            import pose_detector as PD          # noqa: E402
            import coco_data_loader as CDL      # noqa: E402
    finally:
        sys.path.remove(REFERENCE_DIR)

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
