"""GPU (T2): the HIP post-process through the C ABI against (a) the committed golden fixtures produced by the
VERBATIM reference and (b) the NumPy oracle on seeded inputs.  Contract: integer / index outputs bit-exact
(peak coordinates, ids, matches, subsets, poses), smoothed scores bit-exact (float32), PAF scores within 1e-9
(BASELINE.json allows 1e-4)."""
import numpy as np
import pytest

from conftest import golden_cases, load_golden, conns_by_limb, pkg
from oracle import postprocess_ref as P
from oracle import fixtures as Fx

pytestmark = pytest.mark.gpu
SCORE_TOL = 1e-9


def _compare(engine, image, ref_peaks, ref_conns, ref_subsets, ref_poses, ref_scores, rec):
    peaks = engine.peaks(image)
    assert peaks.shape == ref_peaks.reshape(-1, 5).shape, (peaks.shape, ref_peaks.shape)
    assert np.array_equal(peaks, ref_peaks.reshape(-1, 5)), 'all_peaks (type, x, y, score, id) must be bit-exact'
    conns = conns_by_limb(engine.connections(image))
    for l in range(19):
        r = np.asarray(ref_conns[l], dtype=np.float64).reshape(-1, 3)
        assert conns[l].shape == r.shape, 'limb %d: %s vs %s' % (l, conns[l].shape, r.shape)
        assert np.array_equal(conns[l][:, :2], r[:, :2]), 'limb %d ids' % l
        assert np.allclose(conns[l][:, 2], r[:, 2], rtol=0, atol=SCORE_TOL), 'limb %d scores' % l
    subsets = engine.subsets(image)
    assert subsets.shape == ref_subsets.shape
    assert np.array_equal(subsets[:, :18], ref_subsets[:, :18])
    assert np.allclose(subsets[:, 18:], ref_subsets[:, 18:], rtol=0, atol=SCORE_TOL)
    n = int(rec['n_people'])
    assert rec['status'] == 0
    assert n == len(ref_subsets)
    assert rec['n_peaks'] == len(ref_peaks.reshape(-1, 5))
    if n:
        assert np.array_equal(rec['poses'][:n], np.asarray(ref_poses, dtype=np.float64))
        assert np.allclose(rec['scores'][:n], ref_scores, rtol=0, atol=SCORE_TOL)


@pytest.mark.parametrize('name', golden_cases())
def test_reference_golden(engine, name):
    g = load_golden(name)
    map_h, map_w = [int(v) for v in g['map_hw']]
    orig_h, orig_w = [int(v) for v in g['orig_hw']]
    engine.set_maps(g['paf_lo'][None], g['heat_lo'][None])
    engine.postprocess(map_h, map_w, img_len=map_w, scale_xy=[[orig_w / map_w, orig_h / map_h]])
    rec = engine.results()[0]
    _compare(engine, 0, g['all_peaks'], conns_by_limb(g['connections']), g['subsets'], g['poses'], g['scores'], rec)


def test_golden_through_pose_detector_api(native):
    """Same fixtures through the drop-in class, using the reference's `model=` seam."""
    PD = pkg('pose_detector')
    g = load_golden('pp_people3')

    def model(x):
        assert x.shape == (1, 3, 368, 368) and x.dtype == np.float32
        return [g['paf_lo'][None]], [g['heat_lo'][None]]
    det = PD.PoseDetector(model=model, device=0)
    poses, scores = det(np.zeros((368, 368, 3), np.uint8))
    # golden was produced with orig == map (320); the detector rescales by 368/320 (pose_detector.py:513-514)
    ref = g['poses'].copy()
    ref[:, :, 0] *= 368 / 320
    ref[:, :, 1] *= 368 / 320
    assert np.array_equal(poses, ref)
    assert np.allclose(scores, g['scores'], rtol=0, atol=SCORE_TOL)
    det.engine.close()


def test_empty_returns_have_reference_shapes(native):
    PD = pkg('pose_detector')
    z = lambda x: ([np.zeros((1, 38, 46, 46), 'f')], [np.zeros((1, 19, 46, 46), 'f')])
    det = PD.PoseDetector(model=z, device=0)
    poses, scores = det(np.zeros((368, 368, 3), np.uint8))
    assert poses.shape == (0, 18, 3) and scores.shape == (0,)          # pose_detector.py:509-510
    g = load_golden('pp_nolimbs')
    det2 = PD.PoseDetector(model=lambda x: ([g['paf_lo'][None]], [g['heat_lo'][None]]), device=0)
    poses, scores = det2(np.zeros((368, 368, 3), np.uint8))
    assert poses.shape == (0,) and scores.shape == (0,)                # :264 np.array([]) and :516
    det.engine.close()
    det2.engine.close()


@pytest.mark.parametrize('seed,n,hw,mapsz', [(1, 4, (46, 46), (320, 320)), (2, 9, (46, 46), (320, 320)),
                                              (3, 5, (30, 52), (208, 368)), (4, 3, (23, 23), (160, 160)),
                                              (5, 8, (46, 46), (333, 301))])
def test_vs_oracle_seeded(engine, seed, n, hw, mapsz):
    heat, paf, _ = Fx.synthetic_maps(seed, n, hw[0], hw[1], 1.0, 0.9, noise=0.02, height_range=(0.3, 0.7), drop_prob=0.15)
    ref = P.postprocess_from_net_output(paf, heat, mapsz[0], mapsz[1], orig_w=2 * mapsz[1], orig_h=3 * mapsz[0])
    engine.set_option('keep_smoothed', 1)
    engine.set_maps(paf[None], heat[None])
    engine.postprocess(mapsz[0], mapsz[1], img_len=mapsz[1], scale_xy=[[2.0, 3.0]])
    for j in (0, 7, 17):
        assert np.array_equal(engine.smoothed(0, j), ref['smoothed'][j]), 'smoothed heat map %d not bit-exact' % j
    engine.set_option('keep_smoothed', 0)
    rec = engine.results()[0]
    _compare(engine, 0, ref['all_peaks'], ref['connections'], ref['subsets'], ref['poses'], ref['scores'], rec)


def test_batch_images_are_independent(engine):
    cases = [Fx.synthetic_maps(s, 3 + s, 46, 46, 1.0, 0.9, noise=0.01) for s in range(4)]
    heat = np.stack([c[0] for c in cases])
    paf = np.stack([c[1] for c in cases])
    engine.set_maps(paf, heat)
    engine.postprocess(320, 320, img_len=320)
    recs = engine.results()
    for b in range(4):
        ref = P.postprocess_from_net_output(paf[b], heat[b], 320, 320)
        _compare(engine, b, ref['all_peaks'], ref['connections'], ref['subsets'], ref['poses'], ref['scores'], recs[b])


def test_full_resolution_maps_no_upsampling(engine):
    """in == out size: F.resize_images is the identity; this is the detect_precise-style entry."""
    rng = np.random.default_rng(9)
    heat, paf, _ = Fx.synthetic_maps(9, 5, 96, 128, 6.0, 5.0, noise=0.02)
    ref = P.postprocess(heat, paf, 128)
    engine.set_maps(paf[None], heat[None])
    engine.postprocess(96, 128, img_len=128)
    rec = engine.results()[0]
    _compare(engine, 0, ref['all_peaks'], ref['connections'], ref['subsets'], ref['poses'], ref['scores'], rec)


def test_peak_capacity_overflow_is_reported_not_truncated(engine, native):
    rng = np.random.default_rng(3)
    heat = np.zeros((1, 19, 46, 46), 'f')
    heat[0, 0] = rng.random((46, 46)).astype('f') * 4      # hundreds of maxima on joint 0 at 4x upsampling
    engine.set_maps(np.zeros((1, 38, 46, 46), 'f'), heat)
    engine.postprocess(368, 368, img_len=368)
    rec = engine.results()[0]
    ref_peaks, _ = P.compute_peaks_from_heatmaps(P.resize_images_ref(heat[0], 368, 368))
    if (ref_peaks[:, 0] == 0).sum() > native.MAX_PEAKS_PER_JOINT:
        assert rec['status'] & native.IMG_PEAK_OVERFLOW
        PD = pkg('pose_detector')
        with pytest.raises(RuntimeError):
            PD.unpack_results(np.array([rec]))
    else:
        assert rec['status'] == 0


def test_fast_and_generic_peak_kernels_agree(engine):
    """The radius-10 fast path (LDS tables, sliding windows) and the generic-radius kernel: identical smoothed maps
    and peaks (both are also compared with the oracle elsewhere)."""
    heat, paf, _ = Fx.synthetic_maps(21, 7, 46, 46, 1.0, 0.9, noise=0.03)
    out = {}
    for generic in (0, 1):
        engine.set_option('pp_generic', generic)
        engine.set_option('keep_smoothed', 1)
        engine.set_maps(paf[None], heat[None])
        engine.postprocess(333, 301, img_len=301)
        out[generic] = (engine.peaks(0), [engine.smoothed(0, j) for j in (0, 5, 17)], engine.results()[0].copy())
    engine.set_option('pp_generic', 0)
    engine.set_option('keep_smoothed', 0)
    assert np.array_equal(out[0][0], out[1][0])
    for a, b in zip(out[0][1], out[1][1]):
        assert np.array_equal(a, b)
    assert out[0][2]['n_people'] == out[1][2]['n_people'] and np.array_equal(out[0][2]['poses'], out[1][2]['poses'])
    ref = P.postprocess_from_net_output(paf, heat, 333, 301)
    assert np.array_equal(out[0][0], ref['all_peaks'])


def test_reference_gpu_branch_peaks_variant(engine):
    """Documented NON-golden variant: the reference's own GPU branch (pose_detector.py:111-133) -- un-normalised 17x17
    kernel, zero padding, '>=' NMS.  Scores to float32 rounding (cuDNN order is undefined), coordinates exact on a
    fixture without near-ties."""
    heat, paf, _ = Fx.synthetic_maps(31, 5, 46, 46, 1.0, 0.9)
    up = P.resize_images_ref(heat, 320, 320)
    ref_peaks, ref_sm = P.compute_peaks_gpu_branch(up)
    engine.set_option('peaks_gpu_branch', 1)
    engine.set_option('keep_smoothed', 1)
    engine.set_maps(paf[None], heat[None])
    engine.postprocess(320, 320, img_len=320)
    got = engine.peaks(0)
    sm0 = engine.smoothed(0, 0)
    engine.set_option('keep_smoothed', 0)
    engine.set_option('peaks_gpu_branch', 0)
    assert np.abs(sm0 - ref_sm[0]).max() < 2e-6 * max(1.0, np.abs(ref_sm[0]).max())
    assert got.shape == ref_peaks.shape
    assert np.array_equal(got[:, [0, 1, 2, 4]], ref_peaks[:, [0, 1, 2, 4]])
    assert np.allclose(got[:, 3], ref_peaks[:, 3], rtol=0, atol=2e-6)
    # and it is really a different result from the golden CPU branch (scores are not normalised the same way)
    cpu, _ = P.compute_peaks_from_heatmaps(up)
    assert cpu.shape != got.shape or not np.array_equal(cpu[:, 3], got[:, 3])
