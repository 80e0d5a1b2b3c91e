#!/usr/bin/env python
"""bf16x3 conv path ("precision" = 1) vs the fp32-MFMA path: accuracy against a float64 reference, speed, and the effect on the
whole detector (maps, peak indices, poses, scores) over many frames."""
import argparse, importlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = 'chainer_realtime_multi-person_pose_estimation_amd'
ap = argparse.ArgumentParser()
ap.add_argument('--frames', type=int, default=256)
ap.add_argument('--json', default=None)
a = ap.parse_args()
native = importlib.import_module(PKG + '.native')
W = importlib.import_module(PKG + '.weights')
import torch
out = {'conv': []}
eng = native.Engine(0, max_batch=32, max_h=368, max_w=368)
rng = np.random.default_rng(0)
for name, B, cin, H, Wd, cout, k, pool in [('7x7 128->256 B32', 32, 128, 46, 46, 256, 7, False), ('7x7 192->256 B32', 32, 192, 46, 46, 256, 7, False),
                                            ('3x3 256->256 92 B32', 32, 256, 92, 92, 256, 3, False), ('3x3 128->128 184 pool B32', 32, 128, 184, 184, 128, 3, True),
                                            ('3x3 512->512 46 B32', 32, 512, 46, 46, 512, 3, False)]:
    x = np.maximum(rng.standard_normal((B, cin, H, Wd)), 0).astype('f')
    w = (rng.standard_normal((cout, cin, k, k)) / np.sqrt(cin * k * k)).astype('f')
    b = rng.standard_normal(cout).astype('f')
    # float64 reference on 2 images
    with torch.no_grad():
        ref = torch.nn.functional.conv2d(torch.from_numpy(x[:2]).double(), torch.from_numpy(w).double(), torch.from_numpy(b).double(), padding=k // 2)
        ref = torch.relu(ref)
        if pool:
            ref = torch.nn.functional.max_pool2d(ref, 2, 2)
        ref = ref.numpy()
    res = {}
    for prec in (0, 1):
        eng.set_option('precision', prec)
        y, ms = eng.conv2d(x, w, b, relu=True, pool=pool, iters=20)
        res[prec] = (y, ms)
    eng.set_option('precision', 0)
    scale = np.abs(ref).max()
    e0 = np.abs(res[0][0][:2] - ref).max() / scale
    e1 = np.abs(res[1][0][:2] - ref).max() / scale
    d01 = np.abs(res[0][0] - res[1][0]).max() / scale
    flop = 2.0 * B * H * Wd * cout * cin * k * k
    row = dict(shape=name, fp32_ms=res[0][1], bf16x3_ms=res[1][1], speedup=res[0][1] / res[1][1], fp32_tflops=flop / res[0][1] / 1e9,
               bf16x3_equiv_tflops=flop / res[1][1] / 1e9, err_fp32_vs_f64=float(e0), err_bf16x3_vs_f64=float(e1), diff_paths=float(d01))
    out['conv'].append(row)
    print('%-28s fp32 %.3f ms (%.1f TF)  bf16x3 %.3f ms (%.1f TF-equiv)  x%.2f | max err / scale vs float64: fp32 %.2e  bf16x3 %.2e | fp32 vs bf16x3 %.2e'
          % (name, res[0][1], row['fp32_tflops'], res[1][1], row['bf16x3_equiv_tflops'], row['speedup'], e0, e1, d01), flush=True)

# ---- whole detector over `frames` frames
w = W.synthetic_weights(0); eng.set_weights(w)
cal = np.random.default_rng(1234).integers(0, 256, (1, 368, 368, 3), dtype=np.uint8)
eng.forward_u8(cal); paf, heat = eng.get_maps()
w = W.calibrate_head(w, paf[0], heat[0]); eng.set_weights({k: w[k] for k in ('Mconv7_stage6_L1', 'Mconv7_stage6_L2')})
def new_stats():
    return dict(frames=0, frames_same_peaks=0, frames_same_poses=0, max_peak_score_diff=0.0, max_person_score_diff=0.0, max_map_diff_rel=0.0)


stats, control = new_stats(), new_stats()
times = {0: [], 1: [], 2: []}
MODES = {0: dict(precision=0, kernel_gen=6, ksplit=0), 1: dict(precision=1, kernel_gen=6, ksplit=0),
         2: dict(precision=0, kernel_gen=5, ksplit=2)}      # 2 = CONTROL: the same fp32 arithmetic in another summation order (v5 strips, 2 K slices)
for it in range((a.frames + 31) // 32):
    imgs = np.random.default_rng(100 + it).integers(0, 256, (32, 368, 368, 3), dtype=np.uint8)
    r = {}
    for prec in (0, 1, 2):
        for k_, v_ in MODES[prec].items():
            eng.set_option(k_, v_)
        eng.detect_batch(imgs, 320, 320); eng.results()
        t0 = time.perf_counter()
        eng.detect_batch(imgs, 320, 320); rec = eng.results()
        times[prec].append(time.perf_counter() - t0)
        r[prec] = (rec.copy(), [eng.peaks(i) for i in range(32)], eng.get_maps())
    for other, st_ in ((2, control),):
        for i in range(32):
            p0, p1 = r[0][1][i], r[other][1][i]
            same = p0.shape == p1.shape and np.array_equal(p0[:, [0, 1, 2, 4]], p1[:, [0, 1, 2, 4]])
            st_['frames'] += 1
            st_['frames_same_peaks'] += int(same)
            if same and len(p0):
                st_['max_peak_score_diff'] = max(st_['max_peak_score_diff'], float(np.abs(p0[:, 3] - p1[:, 3]).max()))
            a0, a1 = r[0][0][i], r[other][0][i]
            if a0['n_people'] == a1['n_people'] and np.array_equal(a0['poses'], a1['poses']):
                st_['frames_same_poses'] += 1
                st_['max_person_score_diff'] = max(st_['max_person_score_diff'], float(np.abs(a0['scores'] - a1['scores']).max()))
        for m0, m1 in zip(r[0][2], r[other][2]):
            st_['max_map_diff_rel'] = max(st_['max_map_diff_rel'], float(np.abs(m0 - m1).max() / np.abs(m0).max()))
    for i in range(32):
        p0, p1 = r[0][1][i], r[1][1][i]
        same = p0.shape == p1.shape and np.array_equal(p0[:, [0, 1, 2, 4]], p1[:, [0, 1, 2, 4]])
        stats['frames'] += 1
        stats['frames_same_peaks'] += int(same)
        if same and len(p0):
            stats['max_peak_score_diff'] = max(stats['max_peak_score_diff'], float(np.abs(p0[:, 3] - p1[:, 3]).max()))
        a0, a1 = r[0][0][i], r[1][0][i]
        if a0['n_people'] == a1['n_people'] and np.array_equal(a0['poses'], a1['poses']):
            stats['frames_same_poses'] += 1
            stats['max_person_score_diff'] = max(stats['max_person_score_diff'], float(np.abs(a0['scores'] - a1['scores']).max()))
    for m0, m1 in zip(r[0][2], r[1][2]):
        stats['max_map_diff_rel'] = max(stats['max_map_diff_rel'], float(np.abs(m0 - m1).max() / np.abs(m0).max()))
for k_, v_ in MODES[0].items():
    eng.set_option(k_, v_)
out['control_fp32_other_summation_order'] = control
print('control (fp32, other summation order):', json.dumps(control), flush=True)
stats['fp32_frames_per_s'] = 32 / float(np.median(times[0]))
stats['bf16x3_frames_per_s'] = 32 / float(np.median(times[1]))
out['detector'] = stats
print(json.dumps(stats), flush=True)
if a.json:
    json.dump(out, open(a.json, 'w'), indent=1)
