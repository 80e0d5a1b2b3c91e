"""TEST INFRASTRUCTURE -- generates tests/golden/net_*.npz and tests/golden/e2e_*.npz by running the reference's OWN
network and detector code (models/CocoPoseNet.py:132-262, models/FaceNet.py:78-160, models/HandNet.py,
pose_detector.py:484-517, face_detector.py:28-40, hand_detector.py:28-50) through oracle/_refimport.py.

Run in the authoring container only (needs /root/reference):

    python -m oracle.make_golden_net

What is pinned ("outputs of the reference itself run here"; the reference holds no golden vectors):

  net_<arch>_<h>x<w>.npz   the reference Chain's `__call__` on a seeded input with seeded weights: last-stage outputs
                           (posenet: PAF + heat; facenet / handnet: heat).  Inputs and weights are re-created from their
                           seeds by the tests (`weights.synthetic_weights(seed, arch)`, `default_rng(seed).integers`).
  e2e_<image>.npz          BASELINE config 1: the reference `PoseDetector(model=<its CocoPoseNet>)(img)` on the reference's
                           own data/person.png (584x584 RGBA -> BGR), data/people.png (480x480) and data/dinner.png
                           (482 wide x 642 tall): the uint8 BGR image, the resized network input, the two calibrated head
                           layers, all_peaks (before the rescale), poses and scores.
                           No trained weights exist offline, so the weights are seeded He weights whose last two (affine) layers
                           are calibrated on the image (`weights.calibrate_head`): plumbing + golden output, as config 1 says.
                           GPU-safety of each golden is checked here before it is written: the same image through the
                           order-defined network oracle (oracle/conv_fma_ref.py, bit-identical to the HIP kernels) and the
                           verbatim reference post-process must give the same peaks and poses, i.e. no peak / match sits on
                           a tie that fp32 summation-order noise (~1e-6) could flip.
  e2e_precise_<image>.npz  `PoseDetector(..., precise=True)` (pose_detector.py:433-482) on a down-scaled crop, same contents.
  kp_face.npz / kp_hand.npz  the reference FaceDetector / HandDetector on data/face.png / data/hand.png.
  demo_chain_dinner.npz    reference demo.py:27-55 on data/dinner.png (image stored in e2e_dinner.npz): poses -> unit length ->
                           face / hand crops -> key points, for the three people with the most crops.

Third-party steps inside these runs that are restated, not reference-run (named): cv2.resize (OpenCV unpinned),
F.resize_images (Chainer unpinned), the convolution primitive (torch-CPU conv2d standing in for Chainer's im2col + BLAS).
"""
import importlib
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle import _refimport as R            # noqa: E402
from oracle import postprocess_ref as P       # noqa: E402
from oracle import resize_ref                 # noqa: E402
from oracle import conv_fma_ref               # noqa: E402

GOLDEN = os.path.join(ROOT, 'tests', 'golden')
W = importlib.import_module('chainer_realtime_multi-person_pose_estimation_amd.weights')
HEAD = ('Mconv7_stage6_L1', 'Mconv7_stage6_L2')


def net_case(arch, seed, h, w):
    weights = W.synthetic_weights(seed, arch)
    img = np.random.default_rng(seed + 1000).integers(0, 256, (1, h, w, 3), dtype=np.uint8)
    div = 255.0 if arch == 'posenet' else 256.0
    x = (img.astype(np.float32) / np.float32(div) - np.float32(0.5)).transpose(0, 3, 1, 2)
    out = R.ref_network_forward(arch, weights, x)
    name = 'net_%s_%dx%d' % (arch, h, w)
    if arch == 'posenet':
        np.savez_compressed(os.path.join(GOLDEN, name + '.npz'), seed=seed, hw=np.array([h, w]), paf=out[0], heat=out[1])
        print('%-24s paf %s heat %s  |paf|max %.3g |heat|max %.3g' % (name, out[0].shape, out[1].shape,
                                                                     np.abs(out[0]).max(), np.abs(out[1]).max()))
    else:
        np.savez_compressed(os.path.join(GOLDEN, name + '.npz'), seed=seed, hw=np.array([h, w]), heat=out)
        print('%-24s heat %s |heat|max %.3g' % (name, out.shape, np.abs(out).max()))


class _PeakTap(object):
    """Records what the reference's compute_peaks_from_heatmaps returns (all_peaks before the in-place rescale)."""

    def __init__(self, det):
        self.det, self.orig, self.peaks = det, det.compute_peaks_from_heatmaps, None
        det.compute_peaks_from_heatmaps = self

    def __call__(self, heatmaps):
        out = self.orig(heatmaps)
        self.peaks = np.array(out, dtype=np.float64, copy=True).reshape(-1, 5)
        return out


def calibrated_weights(img, seed, **cal):
    """Seeded weights whose head is calibrated on the reference's own outputs for this image."""
    weights = W.synthetic_weights(seed)
    m = R.import_reference_modules()
    det = R.ref_pose_detector(weights)
    in_w, in_h = det.compute_optimal_size(img, m['entity'].params['inference_img_size'])
    small = m['pose_detector'].cv2.resize(img, (in_w, in_h))
    paf, heat = R.ref_network_forward('posenet', weights, det.preprocess(small))
    return W.calibrate_head(weights, paf[0], heat[0], **cal), small


def e2e_case(name, img, seed=0):
    weights, small = calibrated_weights(img, seed)
    det = R.ref_pose_detector(weights)
    tap = _PeakTap(det)
    poses, scores = R.ref_call(det, img)
    poses = np.asarray(poses, dtype=np.float64)
    all_peaks = tap.peaks if tap.peaks is not None else np.zeros((0, 5))
    # GPU-safety: order-defined network oracle (bit-identical to the HIP kernels) + verbatim reference post-process
    map_w, map_h = det.compute_optimal_size(img, 320)
    fpaf, fheat = conv_fma_ref.forward_fma(weights, P.preprocess(small))
    ref2 = R.ref_postprocess(P.resize_images_ref(fheat[0], map_h, map_w), P.resize_images_ref(fpaf[0], map_h, map_w), map_w,
                             orig_w=img.shape[1], orig_h=img.shape[0])
    same = (ref2['all_peaks'].shape == all_peaks.shape and np.array_equal(ref2['all_peaks'][:, [0, 1, 2, 4]], all_peaks[:, [0, 1, 2, 4]])
            and np.asarray(ref2['poses']).shape == poses.shape and np.array_equal(np.asarray(ref2['poses']), poses))
    if not same:
        raise SystemExit('%s: peaks / poses sit on a summation-order tie (seed %d); pick another seed' % (name, seed))
    dpk = float(np.abs(ref2['all_peaks'][:, 3] - all_peaks[:, 3]).max()) if len(all_peaks) else 0.0
    dsc = float(np.abs(np.asarray(ref2['scores']) - np.asarray(scores)).max()) if len(poses) else 0.0
    np.savez_compressed(os.path.join(GOLDEN, name + '.npz'), img=img, resized=small, seed=seed,
                        head_W1=weights[HEAD[0]][0], head_b1=weights[HEAD[0]][1], head_W2=weights[HEAD[1]][0], head_b2=weights[HEAD[1]][1],
                        all_peaks=all_peaks, poses=poses, poses_shape=np.array(poses.shape), scores=np.asarray(scores, dtype=np.float64),
                        order_noise=np.array([dpk, dsc]))
    print('%-22s img %s -> net %s  peaks %4d people %2d  order-defined-oracle deltas: peak score %.2g person score %.2g'
          % (name, img.shape[:2], small.shape[:2], len(all_peaks), len(poses), dpk, dsc))


def e2e_precise_case(name, img, seed=0):
    # the four-scale average damps the random maps: a stronger head keeps ~150 peaks / ~10 people in the averaged maps
    weights, _ = calibrated_weights(img, seed, heat_s=0.2, heat_t=-0.2, paf_s=1.2)
    det = R.ref_pose_detector(weights, precise=True)
    tap = _PeakTap(det)
    poses, scores = R.ref_call(det, img)
    poses = np.asarray(poses, dtype=np.float64)
    np.savez_compressed(os.path.join(GOLDEN, name + '.npz'), img=img, seed=seed,
                        head_W1=weights[HEAD[0]][0], head_b1=weights[HEAD[0]][1], head_W2=weights[HEAD[1]][0], head_b2=weights[HEAD[1]][1],
                        all_peaks=tap.peaks, poses=poses, poses_shape=np.array(poses.shape), scores=np.asarray(scores, dtype=np.float64))
    print('%-22s img %s precise: peaks %4d people %2d' % (name, img.shape[:2], len(tap.peaks), len(poses)))


def keypoint_case(name, arch, img, seed, hand_type=None):
    m = R.import_reference_modules()
    weights = W.synthetic_weights(seed, arch)
    import contextlib
    import io
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, arch + '.npz')
        W.save_npz(path, weights)                 # the reference's own load_npz path (face_detector.py:15-16)
        with contextlib.redirect_stdout(io.StringIO()):
            det = (m['face_detector'].FaceDetector if arch == 'facenet' else m['hand_detector'].HandDetector)(arch, path, device=-1)
    kps = det(img) if hand_type is None else det(img, hand_type=hand_type)
    rows = np.array([[k[0], k[1], k[2], 1.0] if k is not None else [0, 0, 0, 0] for k in kps], dtype=np.float64)
    # relative gap between the two largest smoothed values of every channel (how far the arg-max is from a tie), from the
    # restatement (equal to the reference run: tests/test_reference_network.py)
    from oracle import face_hand_ref as FH
    thresh = m['entity'].params['face_heatmap_peak_thresh' if arch == 'facenet' else 'hand_heatmap_peak_thresh']
    kps2, up = FH.detect(lambda x: FH.cpm_forward(weights, x)[-1], img, thresh, hand_type=hand_type or 'right')
    assert all((a is None) == (b is None) and (a is None or (a[0] == b[0] and a[1] == b[1])) for a, b in zip(kps, kps2))
    gaps = []
    for i in range(up.shape[0] - 1):
        f = np.sort(P.gaussian_filter_ref(up[i]).ravel())
        gaps.append((f[-1] - f[-2]) / abs(f[-1]))
    np.savez_compressed(os.path.join(GOLDEN, name + '.npz'), img=img, seed=seed, keypoints=rows,
                        hand_type=np.array(hand_type or ''), argmax_gap=np.array(gaps, dtype=np.float64))
    print('%-22s %s: %d key points, %d valid, max conf %.3g' % (name, arch, len(rows), int(rows[:, 3].sum()), rows[:, 2].max()))


def demo_chain_case(name, img, n_persons=3, seed=0, face_seed=4, hand_seed=5):
    """reference demo.py:27-55 as a whole: PoseDetector -> get_unit_length -> crop_face / crop_hands -> FaceDetector / HandDetector,
    every step the reference's own code; the first `n_persons` people of the image."""
    import contextlib
    import io
    from oracle import face_hand_ref as FH
    m = R.import_reference_modules()
    weights, _ = calibrated_weights(img, seed)
    det = R.ref_pose_detector(weights)
    poses, _ = R.ref_call(det, img)
    fw, hw = W.synthetic_weights(face_seed, 'facenet'), W.synthetic_weights(hand_seed, 'handnet')
    with tempfile.TemporaryDirectory() as td:
        W.save_npz(os.path.join(td, 'f.npz'), fw)
        W.save_npz(os.path.join(td, 'h.npz'), hw)
        with contextlib.redirect_stdout(io.StringIO()):
            fdet = m['face_detector'].FaceDetector('facenet', os.path.join(td, 'f.npz'), device=-1)
            hdet = m['hand_detector'].HandDetector('handnet', os.path.join(td, 'h.npz'), device=-1)

    def rows_and_gaps(kps, crop, weights_, thresh, hand_type='right'):
        rows = np.array([[k[0], k[1], k[2], 1.0] if k is not None else [0, 0, 0, 0] for k in kps], dtype=np.float64)
        kps2, up = FH.detect(lambda x: FH.cpm_forward(weights_, x)[-1], crop, thresh, hand_type=hand_type)
        gaps = []
        for i in range(up.shape[0] - 1):
            f = np.sort(P.gaussian_filter_ref(up[i]).ravel())
            gaps.append((f[-1] - f[-2]) / max(abs(f[-1]), 1e-30))
        return rows, np.array(gaps)
    out = {'seed': seed,          # (the image itself is in e2e_<image>.npz)
           'face_seed': face_seed, 'hand_seed': hand_seed, 'n_persons': n_persons,
           'head_W1': weights[HEAD[0]][0], 'head_b1': weights[HEAD[0]][1], 'head_W2': weights[HEAD[1]][0], 'head_b2': weights[HEAD[1]][1],
           'poses': np.asarray(poses, dtype=np.float64)}
    # the people with the most face / hand crops (random-weight skeletons often lack a nose or wrists)
    def n_crops_of(pose):
        return int(pose[0][2] > 0) + int(pose[4][2] > 0) + int(pose[7][2] > 0)
    chosen = sorted(range(len(poses)), key=lambda i: -n_crops_of(poses[i]))[:n_persons]
    out['persons'] = np.array(chosen)
    n_crops = 0
    for i in chosen:
        pose = poses[i].copy()
        unit = det.get_unit_length(pose)
        out['unit_%d' % i] = np.float64(unit)
        face, bbox = det.crop_face(img, pose, unit)
        out['face_bbox_%d' % i] = np.array(bbox if bbox is not None else (0, 0, 0, 0))
        if face is not None:
            k, g = rows_and_gaps(fdet(face), face, fw, m['entity'].params['face_heatmap_peak_thresh'])
            out['face_kp_%d' % i], out['face_gap_%d' % i] = k, g
            n_crops += 1
        hands = det.crop_hands(img, pose, unit)
        for side in ('left', 'right'):
            if hands[side] is not None:
                out['%s_bbox_%d' % (side, i)] = np.array(hands[side]['bbox'])
                k, g = rows_and_gaps(hdet(hands[side]['img'], hand_type=side), hands[side]['img'], hw,
                                     m['entity'].params['hand_heatmap_peak_thresh'], hand_type=side)
                out['%s_kp_%d' % (side, i)], out['%s_gap_%d' % (side, i)] = k, g
                n_crops += 1
    np.savez_compressed(os.path.join(GOLDEN, name + '.npz'), **out)
    print('%-22s persons %s, %d face / hand crops through the reference chain' % (name, chosen, n_crops))


def main():
    os.makedirs(GOLDEN, exist_ok=True)
    m = R.import_reference_modules()
    cv2 = m['pose_detector'].cv2
    data = os.path.join(R.REFERENCE_DIR, 'data')
    # ---- networks --------------------------------------------------------------------------------------------------
    net_case('posenet', 0, 64, 96)
    net_case('posenet', 1, 184, 248)
    net_case('facenet', 2, 64, 64)
    net_case('handnet', 3, 72, 56)
    # ---- BASELINE config 1: pose_detector.py:571-574 on the reference's own images -------------------------------
    e2e_case('e2e_person', cv2.imread(os.path.join(data, 'person.png')))
    e2e_case('e2e_people', cv2.imread(os.path.join(data, 'people.png')))
    e2e_case('e2e_dinner', cv2.imread(os.path.join(data, 'dinner.png')))
    # precise mode on a small crop of people.png (four scales of a 120 x 160 image: 184 .. 736 px network inputs)
    crop = np.ascontiguousarray(cv2.imread(os.path.join(data, 'people.png'))[100:220, 150:310])
    e2e_precise_case('e2e_precise_people_crop', crop)
    # ---- face / hand detectors on the reference's own crops ---------------------------------------------------------
    keypoint_case('kp_face', 'facenet', cv2.imread(os.path.join(data, 'face.png')), 4)
    keypoint_case('kp_hand', 'handnet', cv2.imread(os.path.join(data, 'hand.png')), 5)
    keypoint_case('kp_hand_left', 'handnet', cv2.imread(os.path.join(data, 'hand.png')), 5, hand_type='left')
    # ---- the demo.py chain as a whole -----------------------------------------------------------------------------------
    demo_chain_case('demo_chain_dinner', cv2.imread(os.path.join(data, 'dinner.png')))
    print('golden fixtures written to', GOLDEN)


if __name__ == '__main__':
    main()
