"""GPU: the HIP path against goldens written by the reference's OWN code (oracle/make_golden_net.py: models/CocoPoseNet.py,
FaceNet.py, HandNet.py, pose_detector.py, face_detector.py, hand_detector.py executed verbatim on torch-backed stand-ins).

Bars (BASELINE.json): network maps within 1e-4 of the reference's (fp32 summation order differs: MFMA chain here, oneDNN
there, Chainer's im2col + BLAS in the original -- none defined); integer peak indices / poses identical; scores within 1e-4.
BASELINE config 1 = `e2e_person` (data/person.png, 584x584 RGBA -> BGR, 584 -> 368 resize on the device, rescale 584/320).
"""
import os

import numpy as np
import pytest

from conftest import GOLDEN, pkg
from test_reference_network import _x, load_e2e

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float(np.abs(a - b).max() / max(1.0, np.abs(b).max()))


@pytest.mark.parametrize('name', ['net_posenet_64x96', 'net_posenet_184x248'])
def test_posenet_matches_reference_chain_goldens(native, name):
    z = np.load(os.path.join(GOLDEN, name + '.npz'))
    h, w = [int(v) for v in z['hw']]
    img, _ = _x('posenet', int(z['seed']), h, w)
    eng = native.Engine(0, max_batch=1, max_h=h, max_w=w)
    eng.set_weights(pkg('weights').synthetic_weights(int(z['seed'])))
    eng.forward_u8(img)
    paf, heat = eng.get_maps()
    eng.close()
    assert paf.shape == z['paf'].shape and heat.shape == z['heat'].shape
    assert _rel(paf, z['paf']) < 1e-4 and _rel(heat, z['heat']) < 1e-4, (_rel(paf, z['paf']), _rel(heat, z['heat']))


@pytest.mark.parametrize('name', ['net_facenet_64x64', 'net_handnet_72x56'])
def test_cpm_nets_match_reference_chain_goldens(native, name):
    z = np.load(os.path.join(GOLDEN, name + '.npz'))
    arch = name.split('_')[1]
    h, w = [int(v) for v in z['hw']]
    img, _ = _x(arch, int(z['seed']), h, w)
    eng = native.Engine(0, max_batch=1, max_h=h, max_w=w, arch=arch)
    eng.set_weights(pkg('weights').synthetic_weights(int(z['seed']), arch))
    eng.forward_u8(img)
    heat = eng.get_maps()
    eng.close()
    assert heat.shape == z['heat'].shape
    assert _rel(heat, z['heat']) < 1e-4, _rel(heat, z['heat'])


@pytest.mark.parametrize('name', ['e2e_person', 'e2e_people', 'e2e_dinner'])
def test_config1_reference_images_end_to_end(native, name):
    """`PoseDetector(weights=...)(img)` on the reference's own images == what the reference's PoseDetector returned
    (pose_detector.py:484-517, :571-574): the device resize reproduces the resized network input, all_peaks indices and the
    pose arrays are identical, peak / person scores agree to 1e-4."""
    PD = pkg('pose_detector')
    g = load_e2e(name)
    img = g['img']
    det = PD.PoseDetector(weights=g['weights'], device=0)
    poses, scores = det(img)
    in_h, in_w = g['resized'].shape[:2]
    assert np.array_equal(det.engine.get_resized(in_h, in_w)[0], g['resized'])
    peaks = det.engine.peaks(0)
    det.engine.close()
    assert peaks.shape == g['all_peaks'].shape
    assert np.array_equal(peaks[:, [0, 1, 2, 4]], g['all_peaks'][:, [0, 1, 2, 4]])
    d_peak = float(np.abs(peaks[:, 3] - g['all_peaks'][:, 3]).max())
    assert d_peak <= 1e-4
    assert np.asarray(poses).shape == g['poses'].shape and np.array_equal(np.asarray(poses), g['poses'])
    d_person = float(np.abs(np.asarray(scores) - g['scores']).max())
    assert d_person <= 1e-4
    # (the generator measured the same kind of deltas with the order-defined network oracle: g['order_noise'])
    print('%s: %d peaks, %d people; max |d peak score| %.2g, max |d person score| %.2g' % (name, len(peaks), len(g['poses']), d_peak, d_person))


def test_precise_reference_golden(native):
    """`PoseDetector(..., precise=True)` (pose_detector.py:433-482, four scales) on a crop of data/people.png vs the
    reference run: identical peak indices and poses, scores to 1e-4 (cubic resizes restated on both sides)."""
    PD = pkg('pose_detector')
    g = load_e2e('e2e_precise_people_crop')
    det = PD.PoseDetector(weights=g['weights'], device=0, precise=True)
    poses, scores = det(g['img'])
    peaks = det.all_peaks
    det.engine.close()
    assert peaks.shape == g['all_peaks'].shape and np.array_equal(peaks[:, [0, 1, 2, 4]], g['all_peaks'][:, [0, 1, 2, 4]])
    assert np.abs(peaks[:, 3] - g['all_peaks'][:, 3]).max() <= 1e-4
    assert np.array_equal(np.asarray(poses).reshape(-1, 18, 3), g['poses'].reshape(-1, 18, 3))
    assert np.abs(np.asarray(scores) - g['scores']).max() <= 1e-4


@pytest.mark.parametrize('name,arch,cls', [('kp_face', 'facenet', 'FaceDetector'), ('kp_hand', 'handnet', 'HandDetector'),
                                           ('kp_hand_left', 'handnet', 'HandDetector')])
def test_keypoint_detectors_match_reference_goldens(native, name, arch, cls):
    """FaceDetector / HandDetector on the reference's data/face.png / data/hand.png vs the reference classes' own output.
    A channel whose two largest smoothed values are closer than 1e-4 (relative) may legitimately move by one pixel under
    fp32 summation-order noise; everything else must be identical."""
    D = pkg('face_hand_detector')
    z = np.load(os.path.join(GOLDEN, name + '.npz'))
    det = getattr(D, cls)(arch, weights=pkg('weights').synthetic_weights(int(z['seed']), arch), device=0)
    ht = str(z['hand_type'])
    got = det(z['img'], hand_type=ht) if ht else det(z['img'])
    det.engine.close()
    ref = z['keypoints']
    assert len(got) == len(ref)
    moved = 0
    for k, r, gap in zip(got, ref, z['argmax_gap']):
        assert (k is None) == (r[3] == 0)
        if k is None:
            continue
        assert abs(float(k[2]) - r[2]) <= 1e-4 * max(1.0, abs(r[2]))
        if k[0] != r[0] or k[1] != r[1]:
            assert gap < 1e-4 and abs(k[0] - r[0]) <= 1 and abs(k[1] - r[1]) <= 1, (k, r, gap)
            moved += 1
    assert moved <= 2


def test_demo_chain_matches_reference_chain(native):
    """reference demo.py:27-55 as a whole on data/dinner.png: PoseDetector -> get_unit_length -> crop_face / crop_hands ->
    FaceDetector / HandDetector (-> draw_*), every product step against what the reference's own chain produced."""
    PD, FH, W = pkg('pose_detector'), pkg('face_hand_detector'), pkg('weights')
    g = load_e2e('e2e_dinner')
    z = np.load(os.path.join(GOLDEN, 'demo_chain_dinner.npz'))
    img = g['img']
    det = PD.PoseDetector(weights=g['weights'], device=0)
    poses, _ = det(img)
    assert np.array_equal(np.asarray(poses), z['poses'])
    fdet = FH.FaceDetector('facenet', weights=W.synthetic_weights(int(z['face_seed']), 'facenet'), device=0)
    hdet = FH.HandDetector('handnet', weights=W.synthetic_weights(int(z['hand_seed']), 'handnet'), device=0)

    def check(kps, ref, gaps):
        assert len(kps) == len(ref)
        for k, r, gap in zip(kps, ref, gaps):
            assert (k is None) == (r[3] == 0)
            if k is None:
                continue
            assert abs(float(k[2]) - r[2]) <= 1e-4 * max(1.0, abs(r[2]))
            if k[0] != r[0] or k[1] != r[1]:        # an arg-max within 1e-4 of a tie may move by a pixel
                assert gap < 1e-4 and abs(k[0] - r[0]) <= 1 and abs(k[1] - r[1]) <= 1, (k, r, gap)
    canvas = PD.draw_person_pose(img, poses)
    crops = 0
    for i in z['persons']:
        pose = np.asarray(poses[int(i)]).copy()
        unit = det.get_unit_length(pose)
        assert unit == float(z['unit_%d' % i])
        face, bbox = det.crop_face(img, pose, unit)
        if 'face_kp_%d' % i in z.files:
            assert tuple(bbox) == tuple(int(v) for v in z['face_bbox_%d' % i])
            kps = fdet(face)
            check(kps, z['face_kp_%d' % i], z['face_gap_%d' % i])
            canvas = FH.draw_face_keypoints(canvas, kps, (bbox[0], bbox[1]))
            crops += 1
        else:
            assert face is None
        hands = det.crop_hands(img, pose, unit)
        for side in ('left', 'right'):
            if '%s_kp_%d' % (side, i) in z.files:
                assert tuple(hands[side]['bbox']) == tuple(int(v) for v in z['%s_bbox_%d' % (side, i)])
                kps = hdet(hands[side]['img'], hand_type=side)
                check(kps, z['%s_kp_%d' % (side, i)], z['%s_gap_%d' % (side, i)])
                canvas = FH.draw_hand_keypoints(canvas, kps, hands[side]['bbox'][:2])
                crops += 1
            else:
                assert hands[side] is None
    assert crops >= 3 and canvas.shape == img.shape and not np.array_equal(canvas, img)
    for d in (det, fdet, hdet):
        d.engine.close()
