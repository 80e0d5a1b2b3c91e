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


@pytest.mark.parametrize('slices', [0, 8, 3])
@pytest.mark.parametrize('name', golden_cases())
def test_reference_golden(engine, name, slices):
    """slices: blocks per (limb, image) of the candidate-pair scan (option pp_limbs_slices; 0 = the one-block form of the batch path,
    8 = what external full-resolution maps get by default): the list of accepted candidates comes in another order, the result is the same."""
    g = load_golden(name)
    map_h, map_w = [int(v) for v in g['map_hw']]
    orig_h, orig_w = [int(v) for v in g['orig_hw']]
    engine.set_option('pp_limbs_slices', slices)
    try:
        engine.set_maps(g['paf_lo'][None], g['heat_lo'][None])
        engine.postprocess(map_h, map_w, img_len=map_w, scale_xy=[[orig_w / map_w, orig_h / map_h]])
        rec = engine.results()[0]
        _compare(engine, 0, g['all_peaks'], conns_by_limb(g['connections']), g['subsets'], g['poses'], g['scores'], rec)
    finally:
        engine.set_option('pp_limbs_slices', -1)


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


# ---- no result caps: the reference grows its lists without limit (pose_detector.py:104-110,157,243) -----------------------
def _fresh(native, **caps):
    e = native.Engine(0, max_batch=2, max_h=368, max_w=368)
    if caps:
        e.set_capacities(**caps)
    return e


def _check_vs_oracle(e, paf, heat, map_h, map_w, img_len=None):
    ref = P.postprocess_from_net_output(paf, heat, map_h, map_w)
    e.set_maps(paf[None], heat[None])
    e.postprocess(map_h, map_w, img_len=map_w if img_len is None else img_len)
    rec = e.results()[0]
    _compare(e, 0, ref['all_peaks'], ref['connections'], ref['subsets'], ref['poses'], ref['scores'], rec)
    return ref, rec


def test_more_than_128_peaks_per_joint_equals_oracle(native):
    """Hundreds of maxima of one joint type (noise at 8x up-sampling): the peak capacity grows, nothing is truncated."""
    rng = np.random.default_rng(3)
    heat = np.zeros((19, 46, 46), 'f')
    heat[0] = rng.random((46, 46)).astype('f') * 4
    heat[14] = rng.random((46, 46)).astype('f') * 3          # nose (0) - right eye (14): limb 15 gets n0 x n14 pairs
    paf = np.zeros((38, 46, 46), 'f')
    e = _fresh(native)
    ref, rec = _check_vs_oracle(e, paf, heat, 368, 368)
    n0 = int((ref['all_peaks'][:, 0] == 0).sum())
    assert n0 > native.INIT_PEAKS_PER_JOINT, n0
    assert e.capacities()['peaks_per_joint'] >= n0
    e.close()


def test_crowd_of_80_people_equals_oracle(native):
    """>= 80 people in one full-resolution map (in == out size, the detect_precise-style entry): more live subsets than the
    LDS table holds and more persons than the initial record -- capacities grow, results equal the oracle's."""
    heat, paf, poses = Fx.synthetic_maps(77, 96, 320, 448, 3.0, 2.5, height_range=(0.10, 0.16), drop_prob=0.05)
    ref = P.postprocess(heat, paf, 448)
    assert len(ref['subsets']) >= 80, len(ref['subsets'])
    e = _fresh(native)
    e.set_maps(paf[None], heat[None])
    e.postprocess(320, 448, img_len=448)
    rec = e.results()[0]
    _compare(e, 0, ref['all_peaks'], ref['connections'], ref['subsets'], ref['poses'], ref['scores'], rec)
    caps = e.capacities()
    assert caps['people'] >= len(ref['subsets']) > native.INIT_PEOPLE
    # through the drop-in class: no RuntimeError, all people returned
    PD = pkg('pose_detector')
    out = PD.unpack_results(np.array([rec]))[0]
    assert out[0].shape == (len(ref['subsets']), 18, 3)
    e.close()


def test_candidate_store_overflow_equals_oracle(native):
    """A constant PAF field accepts about half of all nA x nB pairs of a limb: far more accepted candidates than the LDS
    store holds (4096) -> device-memory candidate store, same greedy matching."""
    rng = np.random.default_rng(5)
    heat = np.zeros((19, 46, 46), 'f')
    heat[1] = rng.random((46, 46)).astype('f') * 4           # neck
    heat[8] = rng.random((46, 46)).astype('f') * 4           # right waist: limb 0 = neck -> right waist
    paf = np.zeros((38, 46, 46), 'f')
    paf[0] = 1.0                                             # x component of limb 0
    for slices in (-1, 0):                                   # (sliced scan -- the default for external maps -- and the one-block form)
        e = _fresh(native)
        e.set_option('pp_limbs_slices', slices)
        ref, rec = _check_vs_oracle(e, paf, heat, 184, 184)
        assert e.capacities()['candidates'] > 4096
        e.close()


@pytest.mark.parametrize('name', ['pp_crowd12_noise', 'pp_merge', 'pp_twosub', 'pp_netlike'])
def test_growth_from_tiny_capacities_reproduces_goldens(native, name):
    """Contexts shrunk to 4 peaks / joint, 2 subsets, 1 person: every capacity has to grow (several rounds) before the
    reference-generated golden is reproduced exactly; a second image in the same batch is unaffected."""
    g = load_golden(name)
    map_h, map_w = [int(v) for v in g['map_hw']]
    orig_h, orig_w = [int(v) for v in g['orig_hw']]
    e = _fresh(native, peaks_per_joint=4, subsets=2, people=1)
    g2 = load_golden('pp_people3')
    same = g2['paf_lo'].shape == g['paf_lo'].shape
    paf = np.stack([g['paf_lo'], g2['paf_lo']]) if same else g['paf_lo'][None]
    heat = np.stack([g['heat_lo'], g2['heat_lo']]) if same else g['heat_lo'][None]
    e.set_maps(paf, heat)
    sc = [[orig_w / map_w, orig_h / map_h]] * len(paf)
    e.postprocess(map_h, map_w, img_len=map_w, scale_xy=sc)
    recs = e.results()
    _compare(e, 0, g['all_peaks'], conns_by_limb(g['connections']), g['subsets'], g['poses'], g['scores'], recs[0])
    if same and (map_h, map_w) == tuple(int(v) for v in g2['map_hw']):
        assert int(recs[1]['n_people']) == len(g2['subsets'])
        assert np.array_equal(e.peaks(1), g2['all_peaks'].reshape(-1, 5))
    caps = e.capacities()
    per_joint = int(np.bincount(g['all_peaks'][:, 0].astype(int)).max())
    assert caps['peaks_per_joint'] >= per_joint and caps['subsets'] > 2 and caps['people'] >= max(1, len(g['subsets']))
    e.close()


@pytest.mark.parametrize('name', ['pp_crowd12_noise', 'pp_merge', 'pp_twosub'])
def test_subset_table_in_device_memory_reproduces_goldens(native, name):
    """The live subset rows of the grouping sit in LDS up to 896 rows (dynamic LDS; crowds included) and in device memory beyond: a context
    pre-sized for 1024 subsets takes the device-memory form -- same goldens."""
    g = load_golden(name)
    map_h, map_w = [int(v) for v in g['map_hw']]
    orig_h, orig_w = [int(v) for v in g['orig_hw']]
    e = _fresh(native, subsets=1024, people=1024)
    assert e.capacities()['subsets'] >= 1024
    e.set_maps(g['paf_lo'][None], g['heat_lo'][None])
    e.postprocess(map_h, map_w, img_len=map_w, scale_xy=[[orig_w / map_w, orig_h / map_h]])
    _compare(e, 0, g['all_peaks'], conns_by_limb(g['connections']), g['subsets'], g['poses'], g['scores'], e.results()[0])
    e.close()


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
