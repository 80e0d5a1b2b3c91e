"""CPU tests of the margin census (oracle/census.py): the margin restates the reference's five strict comparisons
(reference pose_detector.py:96-102), and two maps that differ by d can only disagree on pixels whose margins are within 2 d."""
import numpy as np

from oracle import census, postprocess_ref as P


def _maps(seed, n=3, h=40, w=52):
    rng = np.random.default_rng(seed)
    heat = (rng.standard_normal((n + 1, h, w)) * 1.0).astype(np.float32)      # (smoothing with sigma 2.5 divides the std by ~9)
    return heat


def test_margin_positive_exactly_where_the_reference_test_finds_a_peak():
    heat = _maps(0)
    peaks, smoothed = P.compute_peaks_from_heatmaps(heat)
    assert len(peaks) > 5
    got = set()
    for j in range(smoothed.shape[0]):
        ys, xs = np.nonzero(census.margin_map(smoothed[j]) > 0)
        got |= {(j, int(x), int(y)) for y, x in zip(ys, xs)}
    assert got == census.peak_set(peaks)


def test_plateau_and_border_pixels_are_not_peaks():
    s = np.zeros((6, 6), np.float32)
    s[2, 2] = s[2, 3] = 0.5            # two equal neighbours: strict '>' rejects both
    s[0, 5] = 0.3                      # corner: compared against zeros outside the map -> a peak
    m = census.margin_map(s)
    assert m[2, 2] == 0 and m[2, 3] == 0 and m[0, 5] > 0


def test_disagreements_of_perturbed_maps_are_near_ties_and_are_all_reported():
    heat = _maps(1, n=2, h=64, w=64)
    peaks_a, sm_a = P.compute_peaks_from_heatmaps(heat)
    rng = np.random.default_rng(5)
    # a "second network": the same smoothed maps + noise of 2e-3 (large on purpose, so that some decisions flip)
    sm_b = (sm_a.astype(np.float64) + rng.uniform(-2e-3, 2e-3, sm_a.shape)).astype(np.float32)
    rows = []
    for j in range(sm_b.shape[0]):
        ys, xs = np.nonzero(census.margin_map(sm_b[j]) > 0)
        rows += [(j, int(x), int(y), float(sm_b[j, y, x]), 0) for y, x in zip(ys, xs)]
    rows = [r[:4] + (i,) for i, r in enumerate(rows)]
    peaks_b = np.array(rows, dtype=np.float64).reshape(-1, 5)
    f = census.compare_frame(peaks_b, peaks_a, sm_a, lambda j: sm_b[j], np.zeros((0, 18, 3)), np.zeros(0), np.zeros((0, 18, 3)), np.zeros(0))
    diff = census.peak_set(peaks_a) ^ census.peak_set(peaks_b)
    assert len(diff) >= 1, 'fixture: the noise should flip at least one decision'
    assert {(m['joint'], m['x'], m['y']) for m in f['mismatches']} == diff
    for m in f['mismatches']:
        assert m['margin_sum'] <= 2 * m['local_abs_diff_smoothed'] * (1 + 1e-9)
        assert (m['margin_gpu'] > 0) != (m['margin_cpu'] > 0)
        assert m['decided_by'] in census.TESTS
    s = census.summarize([f], 'test')
    assert s['all_mismatches_are_near_ties'] and s['mismatching_peaks'] == len(diff) and s['frames_identical'] == 0
    # the same maps on both sides: nothing to report
    f0 = census.compare_frame(peaks_a, peaks_a, sm_a, lambda j: sm_a[j], np.zeros((0, 18, 3)), np.zeros(0), np.zeros((0, 18, 3)), np.zeros(0))
    assert f0['identical_peaks'] and f0['identical_poses'] and not f0['mismatches'] and f0['min_margin_of_accepted_peaks'] > 0


def test_matched_people_scores():
    pa = np.zeros((2, 18, 3)); pa[0, 0] = (3, 4, 2); pa[1, 5] = (7, 8, 2)
    pb = pa[::-1].copy()
    heat = _maps(2, n=1)
    peaks, sm = P.compute_peaks_from_heatmaps(heat)
    f = census.compare_frame(peaks, peaks, sm, lambda j: sm[j], pa, np.array([1.0, 2.0]), pb, np.array([2.0 + 3e-6, 1.0]))
    assert f['matched_people'] == 2 and not f['identical_poses']
    assert abs(f['max_abs_score_diff_matched_people'] - 3e-6) < 1e-12
