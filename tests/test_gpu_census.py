"""GPU parity test in its honest form (SURVEY.md section 4 T3 / section 7 "hard parts"): the default batch path against
the CPU oracle on fresh frames, where "integer peak indices bit-exact" is asserted on every decision whose margin is healthy and every
disagreement must be a near-tie -- a pixel whose decision margins (reference pose_detector.py:96-102, oracle/census.py) lie inside the
local difference of the two networks' smoothed maps.  The 512-frame run of the same census: tools/parity_census.py ->
profiles/r04_parity_census.json."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))

pytestmark = pytest.mark.gpu


def test_census_64_frames_default_path(native):
    """(+ the opt-in bf16x3 mode where the library was built with it: PMX_BUILD_BF16X3=1)"""
    import parity_census
    out = parity_census.run_census(frames=64, batch=32, seed0=9100, bf16x3=native.has_bf16x3())
    parity_census.check(out)                     # every disagreement a near-tie; scores of matched peaks / people within 1e-4
    for name, s in out['paths'].items():
        print('\n[census %s] %d / %d frames identical; %d of %d peaks disagree (max margin %.3g); smallest margin of an accepted peak %.3g, '
              'of any agreed decision %.3g; max |d score| peaks %.3g people %.3g'
              % (name, s['frames_identical'], s['frames'], s['mismatching_peaks'], s['peaks_compared'], s['max_margin_of_a_mismatch'],
                 s['min_margin_of_accepted_peaks'], s['min_abs_margin_of_agreed_decisions'], s['max_abs_peak_score_diff'],
                 s['max_abs_score_diff_matched_people']))
        assert s['frames'] == 64 and s['peaks_compared'] > 64 * 50
        assert s['frames_identical'] >= 56, json.dumps(s)           # (measured flip rate: a few per cent of the frames)
        assert s['matched_people'] >= 0.9 * s['people_cpu']
    f32 = out['paths']['f32_default_batch_path']
    # healthy fixture => identical: a frame without any near-tie pixel (all margins above the two networks' difference) must match exactly
    assert f32['mismatching_peaks'] == 0 or f32['max_margin_of_a_mismatch'] < 1e-5
    # CEILINGS of the default path, so that a kernel change cannot widen the disagreement silently.  Measured on these
    # 64 frames with the round-6 kernels: 62 identical, 2 disagreeing peaks of 9 796, largest margin of a flip 8.2e-8; on 512 frames
    # (profiles/r06_parity_census.json): 501 identical, 14 peaks, 3.8e-7 -- and the same census with conv1 on the direct kernel (503, 12,
    # 1.3e-7) or every layer on the direct kernels (496, 24, 3.8e-7) shows the spread between summation orders, i.e. the 3.8e-7 is a
    # property of those frames' near-ties, not of one kernel.  A flip needs |margin| <= twice the maps' difference (~1e-6 here).
    assert f32['max_margin_of_a_mismatch'] <= 1e-6, json.dumps(f32)
    assert f32['frames_identical'] >= 60, json.dumps(f32)                       # <= 6 % of the frames (measured 3 %)
    assert f32['mismatching_peaks'] <= 6, json.dumps(f32)
    assert f32['max_abs_peak_score_diff'] <= 1e-5 and f32['max_abs_score_diff_matched_people'] <= 2e-5, json.dumps(f32)


def test_config5_precise_482x642_native_network_vs_precise_ref(native):
    """BASELINE config 5 on the NATIVE network (not only through the `model=` seam): PoseDetector(precise=True) on a 482 x 642 frame
    (four scales up to 736 x 984, cubic resizes and accumulation on the device, post-process at the original resolution: ~800 peaks,
    ~30 people) against oracle/precise_ref driving the torch-CPU network restatement -- averaged maps within 2e-5 of the map scale, every
    peak both sides found within 1e-4, every disagreement (if any) a near-tie, every person both sides found within 1e-4."""
    sys.path.insert(0, ROOT)
    import bench
    import importlib
    out = bench.precise_mode(importlib.import_module(bench.PKG + '.weights'), 0, with_oracle=True)
    m = out['keypoint_match_vs_precise_ref']
    print('\n[precise 482x642] %.1f ms per image; %d peaks, %d people; vs precise_ref: %s' % (out['ms_per_image'], out['peaks'], out['people'], json.dumps(m)))
    assert 'error' not in m and 'oracle_raised' not in m, m
    assert out['peaks'] >= 300 and out['people'] >= 10
    assert m['max_abs_diff_averaged_maps_over_scale'] <= 2e-5
    assert m['all_mismatches_are_near_ties'] and m['max_margin_of_a_mismatch'] <= 1e-5
    assert m['max_abs_peak_score_diff'] <= 1e-4 and m['max_abs_score_diff_matched_people'] <= 1e-4
    assert m['matched_people'] >= 0.9 * m['people_cpu']
