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


def test_census_64_frames_default_path_and_bf16x3(native):
    import parity_census
    out = parity_census.run_census(frames=64, batch=32, seed0=9100, bf16x3=True)
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
