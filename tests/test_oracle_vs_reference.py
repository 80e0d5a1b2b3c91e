"""CPU, authoring container only: the oracle against the VERBATIM reference functions imported from
/root/reference (skipped where the reference is absent, e.g. on the GPU box)."""
import numpy as np
import pytest

from oracle import _refimport as R
from oracle import postprocess_ref as P
from oracle import fixtures as Fx

pytestmark = pytest.mark.skipif(not R.reference_available(), reason='/root/reference not present')


@pytest.mark.parametrize('seed,n,noise', [(11, 2, 0.0), (12, 7, 0.02), (13, 10, 0.03)])
def test_postprocess_matches_verbatim_reference(seed, n, noise):
    heat, paf, _ = Fx.synthetic_maps(seed, n, 46, 46, 1.0, 0.9, noise=noise, height_range=(0.3, 0.7), drop_prob=0.15)
    up_h = P.resize_images_ref(heat, 320, 320)
    up_p = P.resize_images_ref(paf, 320, 320)
    ref = R.ref_postprocess(up_h, up_p, 320, orig_w=640, orig_h=480)
    mine = P.postprocess(up_h, up_p, 320, orig_w=640, orig_h=480)
    assert np.array_equal(ref['all_peaks'], mine['all_peaks'])
    for a, b in zip(ref['connections'], mine['connections']):
        assert a.shape == np.asarray(b).reshape(-1, 3).shape
        if len(a):
            assert np.array_equal(a[:, :2], b[:, :2]) and np.allclose(a[:, 2], b[:, 2], rtol=0, atol=1e-12)
    assert np.array_equal(np.asarray(ref['poses']), np.asarray(mine['poses']))
    assert np.allclose(ref['scores'], mine['scores'], rtol=0, atol=1e-12)


def test_label_renderer_matches_reference_generator():
    _, _, _, gen = R.import_reference()
    heat, paf, poses = Fx.synthetic_maps(5, 4, 40, 52, 1.2, 1.0)
    img = np.zeros((40, 52, 3), 'uint8')
    assert np.array_equal(heat, gen.generate_heatmaps(img, poses, 1.2))
    assert np.array_equal(paf, gen.generate_pafs(img, poses, 1.0))


def test_host_helpers_match_reference():
    _, _, det, _ = R.import_reference()
    rng = np.random.default_rng(0)
    for _ in range(50):
        h, w = int(rng.integers(20, 2000)), int(rng.integers(20, 2000))
        img = np.zeros((h, w, 3), 'uint8')
        for t in (368, 320):
            rw, rh = det.compute_optimal_size(img, t)
            assert (int(rw), int(rh)) == P.compute_optimal_size(h, w, t)
    img = rng.integers(0, 256, (9, 13, 3), dtype=np.uint8)
    assert np.array_equal(det.preprocess(img), P.preprocess(img))
