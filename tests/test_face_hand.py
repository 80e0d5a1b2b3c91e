"""Face / hand key-point detectors (reference face_detector.py, hand_detector.py; models/FaceNet.py, models/HandNet.py)."""
import numpy as np
import pytest

from conftest import pkg
from oracle import face_hand_ref as FH


def test_layer_tables_match_reference_shapes():
    W = pkg('weights')
    for arch, n in (('facenet', 71), ('handnet', 22)):
        t = W.layer_table(arch)
        assert t == FH.layer_table(n)
        assert len(t) == 17 + 5 * 7
        assert ('Mconv1_stage2', n + 128, 128, 7) in t and ('conv6_2_CPM', 512, n, 1) in t


def _blob_maps(rng, n_maps, h, w, present):
    maps = np.zeros((1, n_maps, h, w), 'f')
    truth = []
    for c in range(n_maps - 1):
        if present[c]:
            y, x = int(rng.integers(3, h - 3)), int(rng.integers(3, w - 3))
            gy, gx = np.mgrid[0:h, 0:w]
            maps[0, c] = (0.5 + 0.5 * rng.random()) * np.exp(-((gy - y) ** 2 + (gx - x) ** 2) / 3.0)
            truth.append((x, y))
        else:
            truth.append(None)
    maps[0, -1] = 1 - maps[0, :-1].max(0)
    return maps, truth


@pytest.mark.gpu
@pytest.mark.parametrize('arch,cls,n_maps', [('facenet', 'FaceDetector', 71), ('handnet', 'HandDetector', 22)])
def test_keypoints_from_injected_maps_match_oracle(native, arch, cls, n_maps):
    D = pkg('face_hand_detector')
    rng = np.random.default_rng(n_maps)
    present = rng.random(n_maps - 1) > 0.2
    maps, _ = _blob_maps(rng, n_maps, 46, 46, present)
    img = rng.integers(0, 256, (150, 131, 3), dtype=np.uint8)       # the crop size sets the key-point frame
    det = getattr(D, cls)(arch, model=lambda x: [maps], device=0)
    types = ('right', 'left') if arch == 'handnet' else ('right',)
    for hand_type in types:
        got = det(img, hand_type=hand_type) if arch == 'handnet' else det(img)
        ref, up = FH.detect(lambda x: maps, img, 0.1, hand_type=hand_type)
        assert len(got) == n_maps - 1 == len(ref)
        for g, r, p in zip(got, ref, present):
            assert (g is None) == (r is None)
            if r is not None:
                assert g[0] == r[0] and g[1] == r[1] and g[2] == r[2] and isinstance(g[2], np.float32)
            assert p or g is None
    det.engine.close()


@pytest.mark.gpu
def test_keypoint_tie_quirk_and_threshold(native):
    """Two exactly equal maxima: the reference returns [y1, y0] (flattened np.where); below-threshold maps give None."""
    D = pkg('face_hand_detector')
    maps = np.zeros((1, 22, 40, 40), 'f')
    maps[0, 0, 10, 8] = 4.0
    maps[0, 0, 10, 30] = 4.0           # mirror-symmetric pair -> bit-identical smoothed maxima
    maps[0, 1, 20, 20] = 0.5           # smoothed peak = 0.5 * 0.0255 < 0.1 -> None
    det = D.HandDetector('handnet', model=lambda x: [maps], device=0)
    img = np.zeros((40, 40, 3), np.uint8)
    got = det(img)
    ref, _ = FH.detect(lambda x: maps, img, 0.1)
    assert ref[0] is not None and ref[1] is None
    assert got[0][:2] == ref[0][:2] and got[0][2] == ref[0][2]
    assert got[1] is None
    # left hand: the reference mirrors the resized maps before the peaks (hand_detector.py:46-47), so the row-major order among
    # the two exactly equal maxima is taken on the MIRRORED map -- reproduced on the device by reversing the resize columns
    maps2 = maps.copy()
    maps2[0, 2, 12, 5] = 8.0
    maps2[0, 2, 30, 5] = 8.0           # two equal maxima in one column: rows decide, the mirror changes nothing
    maps2[0, 3, 7, 9] = 8.0
    maps2[0, 3, 7, 30] = 8.0           # two equal maxima in one row: the mirror swaps their order
    det2 = D.HandDetector('handnet', model=lambda x: [maps2], device=0)
    img2 = np.zeros((40, 40, 3), np.uint8)
    got_l = det2(img2, hand_type='left')
    ref_l, _ = FH.detect(lambda x: maps2, img2, 0.1, hand_type='left')
    for c in (0, 2, 3):
        assert ref_l[c] is not None and got_l[c][:2] == ref_l[c][:2] and got_l[c][2] == ref_l[c][2], (c, got_l[c], ref_l[c])
    det.engine.close()
    det2.engine.close()


@pytest.mark.gpu
@pytest.mark.parametrize('arch,cls,n_maps', [('facenet', 'FaceDetector', 71), ('handnet', 'HandDetector', 22)])
def test_network_matches_torch_oracle(native, arch, cls, n_maps):
    W = pkg('weights')
    D = pkg('face_hand_detector')
    weights = W.synthetic_weights(3, arch)
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (368, 368, 3), dtype=np.uint8)
    det = getattr(D, cls)(arch, weights=weights, device=0)
    det(img)
    heat = det.engine.get_maps()
    x = np.array(img[np.newaxis], dtype=np.float32).transpose(0, 3, 1, 2) / 256 - 0.5
    ref = FH.cpm_forward(weights, x)[-1]
    assert heat.shape == ref.shape == (1, n_maps, 46, 46)
    assert np.abs(heat - ref).max() < 1e-4 * max(1.0, np.abs(ref).max())
    # key points: exact on the device's own maps
    from oracle import postprocess_ref as P
    kp_ref = FH.compute_keypoints(P.resize_images_ref(heat[0], 368, 368), 0.1)
    kp = det(img)
    for g, r in zip(kp, kp_ref):
        assert (g is None) == (r is None)
        if r is not None:
            assert g[0] == r[0] and g[1] == r[1] and g[2] == r[2]
    det.engine.close()


@pytest.mark.gpu
@pytest.mark.parametrize('arch', ['facenet', 'handnet'])
def test_cpm_nets_batch64_kernel_generations_identical(native, arch):
    """At batch 64 x 368 x 368 the single-branch CPM nets take the one-block-per-CU v6 kernels (17- and 9-tile blocks, the
    pooled variant, 13 / 10 input chunks); the maps must equal the v5 path's and a single image's bit for bit."""
    W = pkg('weights')
    eng = native.Engine(0, max_batch=64, max_h=368, max_w=368, arch=arch)
    eng.set_weights(W.synthetic_weights(0, arch))
    imgs = np.random.default_rng(3).integers(0, 256, (64, 368, 368, 3), dtype=np.uint8)
    outs = {}
    eng.set_option('conv_algo', 0)       # the direct kernel generations are compared (batch 64 would take the Winograd kernel)
    for gen in (5, 6):
        eng.set_option('kernel_gen', gen)
        eng.profile_reset()
        eng.profile_enable(True)
        eng.forward_u8(imgs)
        outs[gen] = eng.get_maps()
        names = {e['kernel'] for e in eng.profile()}
        eng.profile_enable(False)
        assert any('_v6' in k for k in names) == (gen == 6), names
    assert np.array_equal(outs[5], outs[6])
    eng.set_option('ksplit', 1)          # unsplit single-image kernels: same K order as the batch kernels
    eng.forward_u8(imgs[7:8])
    assert np.array_equal(eng.get_maps()[0], outs[6][7])
    eng.close()
