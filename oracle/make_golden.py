"""TEST INFRASTRUCTURE -- generates tests/golden/* by running the VERBATIM reference post-process.

Run in the authoring container only (needs /root/reference):

    python -m oracle.make_golden

What is pinned (SURVEY.md section 8c -- the reference itself holds no golden vectors, so these are
"outputs of the reference itself run here"):

  * pp_*.npz     -- post-process cases.  Inputs are low-resolution network-output-shaped maps
                    (heat_lo (19,h,w), paf_lo (38,h,w)); they are resized to the map size with the
                    restated F.resize_images (oracle/postprocess_ref.py; Chainer is not installable, so
                    that one step is restated, not reference-run) and then pushed through the reference's
                    own compute_peaks_from_heatmaps / compute_connections / grouping_key_points /
                    subsets_to_pose_array (pose_detector.py:75-265), imported verbatim.
  * host_fns.json -- compute_optimal_size (pose_detector.py:57-73) and entity constants
                    (entity.py:9-45, 71-105) read from the reference modules.
  * preprocess.npz -- PoseDetector.preprocess (pose_detector.py:426-431) on a seeded uint8 image.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle import _refimport as R            # noqa: E402
from oracle import postprocess_ref as P       # noqa: E402
from oracle import fixtures as Fx             # noqa: E402

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pack_connections(conns):
    rows = []
    for li, c in enumerate(conns):
        for r in np.asarray(c).reshape(-1, 3):
            rows.append([li, r[0], r[1], r[2]])
    return np.array(rows, dtype=np.float64).reshape(-1, 4)


def run_case(name, heat_lo, paf_lo, map_h, map_w, orig_h=None, orig_w=None, note=''):
    heat_lo = np.ascontiguousarray(heat_lo, dtype=np.float32)
    paf_lo = np.ascontiguousarray(paf_lo, dtype=np.float32)
    orig_h = map_h if orig_h is None else orig_h
    orig_w = map_w if orig_w is None else orig_w
    heat = P.resize_images_ref(heat_lo, map_h, map_w)
    paf = P.resize_images_ref(paf_lo, map_h, map_w)
    ref = R.ref_postprocess(heat, paf, map_w, orig_w=orig_w, orig_h=orig_h)
    poses = np.asarray(ref['poses'], dtype=np.float64)
    np.savez_compressed(
        os.path.join(GOLDEN, name + '.npz'),
        heat_lo=heat_lo.astype(np.float16) if _fits_f16(heat_lo) else heat_lo,
        paf_lo=paf_lo.astype(np.float16) if _fits_f16(paf_lo) else paf_lo,
        map_hw=np.array([map_h, map_w]), orig_hw=np.array([orig_h, orig_w]),
        all_peaks=ref['all_peaks'], connections=pack_connections(ref['connections']),
        subsets=ref['subsets'], poses=poses, poses_shape=np.array(poses.shape),
        scores=np.asarray(ref['scores'], dtype=np.float64), note=np.array(note))
    # keep the restatement honest while we are here
    mine = P.postprocess(heat, paf, map_w, orig_w=orig_w, orig_h=orig_h)
    assert np.array_equal(mine['all_peaks'], ref['all_peaks']), name
    assert np.array_equal(np.asarray(mine['poses']), poses), name
    print('%-22s map %dx%d peaks %4d conns %3d people %2d  %s' % (
        name, map_h, map_w, len(ref['all_peaks']), sum(len(c) for c in ref['connections']),
        len(ref['subsets']), note))


def _fits_f16(a):
    return np.array_equal(a.astype(np.float16).astype(np.float32), a)


def quantize_f16(a):
    """Store-friendly inputs: values exactly representable in float16 (inputs only; all math stays f32)."""
    return a.astype(np.float16).astype(np.float32)


def main():
    os.makedirs(GOLDEN, exist_ok=True)
    PD, CDL, det, gen = R.import_reference()

    # ---- post-process cases ---------------------------------------------------------------------
    h, p, _ = Fx.synthetic_maps(1, 3, 46, 46, 1.0, 0.9)
    run_case('pp_people3', quantize_f16(h), quantize_f16(p), 320, 320, note='3 clean skeletons')

    h, p, _ = Fx.synthetic_maps(2, 12, 46, 46, 0.9, 0.8, noise=0.02, height_range=(0.25, 0.6), drop_prob=0.2)
    run_case('pp_crowd12_noise', quantize_f16(h), quantize_f16(p), 320, 320, note='12 people, noise, dropped joints')

    h, p, _ = Fx.synthetic_maps(3, 6, 46, 69, 1.0, 0.9, noise=0.01, drop_prob=0.15)
    run_case('pp_rect_46x69', quantize_f16(h), quantize_f16(p), 320, 480, orig_h=427, orig_w=640,
             note='non-square map, orig-size rescale')

    h, p, _ = Fx.synthetic_maps(4, 5, 69, 46, 1.1, 1.0, drop_prob=0.3)
    run_case('pp_rect_69x46', quantize_f16(h), quantize_f16(p), 480, 320, orig_h=960, orig_w=640,
             note='portrait map')

    run_case('pp_empty', np.zeros((19, 46, 46), 'f'), np.zeros((38, 46, 46), 'f'), 320, 320, note='no peaks')

    # peaks but no valid limbs (heat only)
    h, p, _ = Fx.synthetic_maps(5, 4, 46, 46, 1.0, 0.9)
    run_case('pp_nolimbs', quantize_f16(h), np.zeros_like(p), 320, 320, note='peaks, zero PAF')

    # network-like smooth random fields (what seeded synthetic weights produce): many peaks/limbs
    rng = np.random.default_rng(7)
    from scipy.ndimage import gaussian_filter as gf
    z = np.stack([gf(rng.standard_normal((46, 46)), 1.2) for _ in range(19)])
    z = (z - z.mean((1, 2), keepdims=True)) / z.std((1, 2), keepdims=True)
    zp = np.stack([gf(rng.standard_normal((46, 46)), 2.0) for _ in range(38)])
    zp = (zp - zp.mean((1, 2), keepdims=True)) / zp.std((1, 2), keepdims=True)
    run_case('pp_netlike', quantize_f16((z * 0.1 - 0.12).astype('f')), quantize_f16((zp * 0.5).astype('f')),
             320, 320, note='smooth random fields')

    # full-resolution adversarial map (no upsampling: in == out): plateaus, border peaks, exact ties
    H = W = 64
    heat = np.zeros((19, H, W), 'f')
    heat[0, 0, 0] = 1.0            # corner peak
    heat[0, 0, 31] = 1.0           # top border
    heat[0, 63, 63] = 1.0          # far corner
    heat[0, 30, 63] = 0.9          # right border
    heat[1, 20:22, 20:22] = 1.0    # 2x2 plateau -> strict '>' rejects all four after symmetric smoothing
    heat[1, 40, 10] = 1.0
    heat[1, 40, 12] = 1.0          # two equal peaks 2 px apart (merge to plateau/tie)
    heat[2, 10, 10] = 0.3          # below-threshold after smoothing? (0.3 * 0.0255 < 0.05)
    heat[2, 32, 32] = 2.0
    heat[3, 5, :] = 0.5            # ridge: equal along x
    heat[4, :, 7] = 0.5            # ridge along y
    heat[5] = 0.06                 # constant map just above threshold: no strict maxima inside
    heat[6, 31, 31] = 2.0
    heat[6, 31, 33] = 2.0 + 2 ** -20
    paf = np.zeros((38, H, W), 'f')
    paf[28] = 0.0
    paf[29] = -1.0                 # limb 14 Neck->Nose pointing up
    run_case('pp_adversarial64', heat, paf, H, W, note='full-res: plateaus, borders, ridges, ties')

    # cases that drive the two-subset branches of grouping (pose_detector.py:208-235)
    # (a) MERGE (:214-218, incl. the `[-2:] += score` quirk): a person whose Neck is not detected splits into a
    #     shoulder/arm subset (which picks up the ear at limb 9/13) and a nose/eye subset; limb 17/18 joins them.
    rng = np.random.default_rng(21)
    poses = Fx.random_poses(rng, 3, 46, 46, height_range=(0.55, 0.75), drop_prob=0.0, jitter=0.0)
    poses[0, 1, 2] = 0       # person 0: no Neck
    poses[2, 1, 2] = 0       # person 2: no Neck, no left eye
    poses[2, 15, 2] = 0
    h = Fx.render_heatmaps((46, 46), poses, 1.0)
    p = Fx.render_pafs((46, 46), poses, 0.9)
    h, p = quantize_f16(h), quantize_f16(p)
    stats = {}
    out = P.postprocess_from_net_output(p, h, 320, 320)
    P.grouping_key_points(out['connections'], out['all_peaks'], stats=stats)
    assert stats.get('merge', 0) > 0, stats
    run_case('pp_merge', h, p, 320, 320, note='two-subset MERGE branch x%d, no-merge x%d'
             % (stats.get('merge', 0), stats.get('two_nomerge', 0)))
    # (b) NO-merge (:219-235): found by seed search over noisy crowds
    for seed in range(100, 400):
        h, p, _ = Fx.synthetic_maps(seed, 6, 46, 46, 0.9, 0.8, noise=0.03, height_range=(0.3, 0.6), drop_prob=0.35)
        h, p = quantize_f16(h), quantize_f16(p)
        stats = {}
        out = P.postprocess_from_net_output(p, h, 320, 320)
        P.grouping_key_points(out['connections'], out['all_peaks'], stats=stats)
        if stats.get('two_nomerge', 0) > 0:
            run_case('pp_twosub', h, p, 320, 320, note='two-subset NO-merge branch x%d (seed %d)'
                     % (stats['two_nomerge'], seed))
            break

    # ---- host helper functions --------------------------------------------------------------------
    sizes = [(584, 584), (480, 480), (482, 642), (642, 482), (368, 368), (720, 1280), (1080, 1920),
             (375, 500), (333, 500), (500, 333), (427, 640), (100, 37), (37, 100), (369, 371)]
    table = []
    for (hh, ww) in sizes:
        img = np.zeros((hh, ww, 3), 'uint8')
        for target in (368, 320):
            w_, h_ = det.compute_optimal_size(img, target)
            table.append([hh, ww, target, int(w_), int(h_)])
    ent = {
        'JointType': {j.name: int(j) for j in PD.JointType},
        'limbs_point': [[int(a), int(b)] for a, b in PD.params['limbs_point']],
        'params': {k: PD.params[k] for k in (
            'inference_img_size', 'inference_scales', 'heatmap_size', 'gaussian_sigma', 'ksize',
            'n_integ_points', 'n_integ_points_thresh', 'heatmap_peak_thresh', 'inner_product_thresh',
            'limb_length_ratio', 'length_penalty_value', 'n_subset_limbs_thresh', 'subset_score_thresh',
            'downscale')},
    }
    with open(os.path.join(GOLDEN, 'host_fns.json'), 'w') as f:
        json.dump({'compute_optimal_size': table, 'entity': ent}, f, indent=1)

    rng = np.random.default_rng(11)
    img = rng.integers(0, 256, (16, 24, 3), dtype=np.uint8)
    np.savez_compressed(os.path.join(GOLDEN, 'preprocess.npz'), img=img, x=det.preprocess(img))
    print('golden fixtures written to', GOLDEN)


if __name__ == '__main__':
    main()
