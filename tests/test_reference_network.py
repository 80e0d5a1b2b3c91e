"""CPU: the network half of the oracle pinned to the reference's OWN model code.

Authoring container (where /root/reference exists): `oracle/_refimport.py` executes models/CocoPoseNet.py, models/FaceNet.py,
models/HandNet.py, pose_detector.py, face_detector.py and hand_detector.py verbatim on torch-backed stand-ins for the few
Chainer / OpenCV calls they make; the restatements in oracle/ (network_ref, face_hand_ref, postprocess_ref, precise_ref) and the
product's layer tables must agree with them exactly.  Everywhere (also on the GPU box's CPU): the restatements reproduce the
committed goldens those reference runs wrote (tests/golden/net_*.npz, e2e_*.npz, kp_*.npz; generator oracle/make_golden_net.py).
"""
import os
import tempfile

import numpy as np
import pytest

from conftest import GOLDEN, pkg
from oracle import _refimport as R
from oracle import face_hand_ref as FH
from oracle import network_ref as N
from oracle import postprocess_ref as P
from oracle import precise_ref
from oracle import resize_ref

needs_reference = pytest.mark.skipif(not R.reference_available(), reason='/root/reference not present')
HEAD = ('Mconv7_stage6_L1', 'Mconv7_stage6_L2')


def _x(arch, seed, h, w):
    img = np.random.default_rng(seed + 1000).integers(0, 256, (1, h, w, 3), dtype=np.uint8)
    div = 255.0 if arch == 'posenet' else 256.0
    return img, (img.astype(np.float32) / np.float32(div) - np.float32(0.5)).transpose(0, 3, 1, 2)


def load_e2e(name):
    z = np.load(os.path.join(GOLDEN, name + '.npz'))
    d = {k: z[k] for k in z.files}
    d['poses'] = d['poses'].reshape(tuple(d['poses_shape']))
    w = pkg('weights').synthetic_weights(int(d['seed']))
    w[HEAD[0]] = (d['head_W1'], d['head_b1'])
    w[HEAD[1]] = (d['head_W2'], d['head_b2'])
    d['weights'] = w
    return d


# ---- against the reference's own code (authoring container) ---------------------------------------------------------------
@needs_reference
@pytest.mark.parametrize('arch', ['posenet', 'facenet', 'handnet'])
def test_layer_tables_equal_the_reference_chain(arch):
    """Names, order, shapes, stride and padding read off the instantiated reference Chain == the product's layer table
    (weights.py, which pmx_set_layer checks every upload against) == the oracle's."""
    ref = R.ref_layer_table(arch)
    assert [(n, ci, co, k) for n, ci, co, k, s, p in ref] == [tuple(t) for t in pkg('weights').layer_table(arch)]
    assert all(s == 1 and p == k // 2 for _, _, _, k, s, p in ref)
    if arch == 'posenet':
        assert [(n, ci, co, k) for n, ci, co, k, s, p in ref] == N.layer_table()
    else:
        assert [(n, ci, co, k) for n, ci, co, k, s, p in ref] == FH.layer_table({'facenet': 71, 'handnet': 22}[arch])


@needs_reference
def test_network_restatement_equals_reference_posenet_bit_for_bit():
    """oracle/network_ref.forward == the reference's CocoPoseNet.__call__ (models/CocoPoseNet.py:132-262) executed verbatim, all
    six stages, same conv primitive -> identical bits.  A swapped concat order, a misplaced pool or a missing ReLU in the
    restatement cannot pass."""
    w = pkg('weights').synthetic_weights(0)
    _, x = _x('posenet', 7, 64, 96)
    rp, rh = R.ref_network_forward('posenet', w, x, all_stages=True)
    mine = N.forward(w, x, all_stages=True)
    assert len(rp) == len(rh) == len(mine) == 6
    for s in range(6):
        assert np.array_equal(rp[s], mine[s][0]) and np.array_equal(rh[s], mine[s][1]), s


@needs_reference
@pytest.mark.parametrize('arch,n', [('facenet', 71), ('handnet', 22)])
def test_network_restatement_equals_reference_cpm_bit_for_bit(arch, n):
    w = pkg('weights').synthetic_weights(1, arch)
    _, x = _x(arch, 8, 56, 72)
    ref = R.ref_network_forward(arch, w, x, all_stages=True)
    mine = FH.cpm_forward(w, x)
    assert len(ref) == len(mine) == 6 and ref[-1].shape == (1, n, 7, 9)
    for a, b in zip(ref, mine):
        assert np.array_equal(a, b)


@needs_reference
def test_reference_call_equals_oracle_pipeline_and_npz_round_trip():
    """The whole reference `PoseDetector('posenet', weights_file)(img)` (its own load_npz path, :23-26, and __call__, :484-517)
    == resize_ref + preprocess + network_ref + postprocess_ref; the NPZ written by the product's writer loads into the
    reference Chain."""
    W = pkg('weights')
    w = W.synthetic_weights(3)
    img = np.random.default_rng(5).integers(0, 256, (70, 90, 3), dtype=np.uint8)
    in_w, in_h = P.compute_optimal_size(70, 90, 368)
    map_w, map_h = P.compute_optimal_size(70, 90, 320)
    small = resize_ref.resize_linear_u8(img, in_w, in_h)
    paf, heat = N.forward(w, P.preprocess(small))
    w = W.calibrate_head(w, paf[0], heat[0])
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, 'coco_posenet.npz')
        W.save_npz(path, w)
        det = R.ref_pose_detector(weights_file=path)
    poses, scores = R.ref_call(det, img)
    paf, heat = N.forward(w, P.preprocess(small))
    mine = P.postprocess_from_net_output(paf[0], heat[0], map_h, map_w, orig_w=90, orig_h=70)
    assert len(mine['all_peaks']) > 20 and len(mine['poses']) > 0
    assert np.array_equal(np.asarray(poses), np.asarray(mine['poses']))
    assert np.allclose(scores, mine['scores'], rtol=0, atol=1e-12)


@needs_reference
def test_reference_detect_precise_equals_oracle_pipeline():
    """pose_detector.py:433-482 run verbatim (four scales) == oracle/precise_ref.py on the same network."""
    W = pkg('weights')
    w = W.synthetic_weights(2)
    img = np.random.default_rng(6).integers(0, 256, (40, 56, 3), dtype=np.uint8)
    paf, heat = N.forward(w, P.preprocess(resize_ref.resize_linear_u8(img, 368 * 56 // 40 // 8 * 8, 368)))
    w = W.calibrate_head(w, paf[0], heat[0], heat_s=0.2, heat_t=-0.2, paf_s=1.2)
    det = R.ref_pose_detector(w, precise=True)
    poses, scores = R.ref_call(det, img)
    pafs, heats, _ = precise_ref.averaged_maps(lambda x: N.forward(w, x), img)
    assert np.array_equal(det.pafs, pafs) and np.array_equal(det.heatmaps, heats)
    mine = precise_ref.detect_precise_from_maps(pafs, heats)
    assert len(mine['all_peaks']) > 0
    assert np.array_equal(np.asarray(poses).reshape(-1, 18, 3), np.asarray(mine['poses']).reshape(-1, 18, 3))


@needs_reference
@pytest.mark.parametrize('arch,hand_type', [('facenet', None), ('handnet', 'right'), ('handnet', 'left')])
def test_reference_keypoint_detectors_equal_oracle(arch, hand_type):
    """face_detector.py:28-40 / hand_detector.py:28-50 run verbatim == oracle/face_hand_ref.detect."""
    m = R.import_reference_modules()
    W = pkg('weights')
    w = W.synthetic_weights(4, arch)
    img = np.random.default_rng(9).integers(0, 256, (60, 52, 3), dtype=np.uint8)
    import contextlib
    import io
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, 'w.npz')
        W.save_npz(path, w)
        with contextlib.redirect_stdout(io.StringIO()):
            det = (m['face_detector'].FaceDetector if arch == 'facenet' else m['hand_detector'].HandDetector)(arch, path, device=-1)
    # the full-size (368 x 368) network is slow on the CPU; both sides see the same shrunken inference size
    key = 'face_inference_img_size' if arch == 'facenet' else 'hand_inference_img_size'
    thresh = m['entity'].params['face_heatmap_peak_thresh' if arch == 'facenet' else 'hand_heatmap_peak_thresh']
    old = m['entity'].params[key]
    m['entity'].params[key] = 64
    try:
        ref = det(img) if hand_type is None else det(img, hand_type=hand_type)
    finally:
        m['entity'].params[key] = old
    mine, _ = FH.detect(lambda x: FH.cpm_forward(w, x)[-1], img, thresh, size=64, hand_type=hand_type or 'right')
    assert len(ref) == len(mine)
    for a, b in zip(ref, mine):
        assert (a is None) == (b is None)
        if a is not None:
            assert a[0] == b[0] and a[1] == b[1] and a[2] == b[2]


# ---- against the committed goldens (everywhere) -------------------------------------------------------------------------------
@pytest.mark.parametrize('name', ['net_posenet_64x96', 'net_posenet_184x248', 'net_facenet_64x64', 'net_handnet_72x56'])
def test_network_restatement_reproduces_reference_goldens(name):
    """The goldens were written by the reference's own Chain; oneDNN may pick another summation order on another CPU, hence
    1e-5 (relative to the map scale) instead of bit equality."""
    z = np.load(os.path.join(GOLDEN, name + '.npz'))
    arch = name.split('_')[1]
    h, w = [int(v) for v in z['hw']]
    weights = pkg('weights').synthetic_weights(int(z['seed']), arch)
    _, x = _x(arch, int(z['seed']), h, w)
    if arch == 'posenet':
        paf, heat = N.forward(weights, x)
        assert np.abs(paf - z['paf']).max() <= 1e-5 * max(1.0, np.abs(z['paf']).max())
    else:
        heat = FH.cpm_forward(weights, x)[-1]
    assert np.abs(heat - z['heat']).max() <= 1e-5 * max(1.0, np.abs(z['heat']).max())


@pytest.mark.parametrize('name', ['e2e_person', 'e2e_people', 'e2e_dinner'])
def test_oracle_pipeline_reproduces_reference_e2e_goldens(name):
    """BASELINE config 1 on the CPU: the oracle pipeline == what the reference's PoseDetector returned for its own
    data/<image>.png (peak indices and poses identical, scores to 1e-4)."""
    g = load_e2e(name)
    img = g['img']
    oh, ow = img.shape[:2]
    in_w, in_h = P.compute_optimal_size(oh, ow, 368)
    map_w, map_h = P.compute_optimal_size(oh, ow, 320)
    small = resize_ref.resize_linear_u8(img, in_w, in_h)
    assert np.array_equal(small, g['resized'])
    paf, heat = N.forward(g['weights'], P.preprocess(small))
    out = P.postprocess_from_net_output(paf[0], heat[0], map_h, map_w, orig_w=ow, orig_h=oh)
    assert np.array_equal(out['all_peaks'][:, [0, 1, 2, 4]], g['all_peaks'][:, [0, 1, 2, 4]])
    assert np.abs(out['all_peaks'][:, 3] - g['all_peaks'][:, 3]).max() <= 1e-4
    assert np.array_equal(np.asarray(out['poses']), g['poses'])
    assert np.abs(np.asarray(out['scores']) - g['scores']).max() <= 1e-4
