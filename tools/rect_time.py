#!/usr/bin/env python
"""Throughput of the hot path on non-square network inputs (e.g. 368 x 496 = a 4:3 COCO frame): frames/s and TFLOP/s."""
import argparse, importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument('--h', type=int, default=368); ap.add_argument('--w', type=int, default=496)
ap.add_argument('--batch', type=int, default=32); ap.add_argument('--steps', type=int, default=5)
ap.add_argument('--gen', type=int, default=0); ap.add_argument('--profile', action='store_true')
a = ap.parse_args()
native = importlib.import_module('chainer_realtime_multi-person_pose_estimation_amd.native')
W = importlib.import_module('chainer_realtime_multi-person_pose_estimation_amd.weights')
eng = native.Engine(0, max_batch=a.batch, max_h=a.h, max_w=a.w)
w = W.synthetic_weights(0); eng.set_weights(w)
cal = np.random.default_rng(1234).integers(0, 256, (1, a.h, a.w, 3), dtype=np.uint8)
eng.forward_u8(cal); paf, heat = eng.get_maps()
w = W.calibrate_head(w, paf[0], heat[0]); eng.set_weights({k: w[k] for k in ('Mconv7_stage6_L1', 'Mconv7_stage6_L2')})
if a.gen:
    eng.set_option('kernel_gen', a.gen)
imgs = np.random.default_rng(1).integers(0, 256, (a.batch, a.h, a.w, 3), dtype=np.uint8)
mh, mw = a.h * 320 // 368 // 8 * 8, a.w * 320 // 368 // 8 * 8
eng.detect_batch(imgs, mh, mw); eng.results()
t0 = time.perf_counter()
for _ in range(a.steps):
    eng.detect_batch(imgs, mh, mw); rec = eng.results()
dt = (time.perf_counter() - t0) / a.steps
flop = 271868013568 * (a.h * a.w) / (368.0 * 368.0)
print('%dx%d B=%d gen=%d: %.2f ms/step  %.1f frames/s  %.1f TFLOP/s whole net (host upload included)  people/frame %.1f'
      % (a.h, a.w, a.batch, a.gen, dt * 1e3, a.batch / dt, flop * a.batch / dt / 1e12, rec['n_people'].mean()))
if a.profile:
    eng.profile_enable(True)
    eng.detect_batch(imgs, mh, mw); eng.results()
    by = {}
    for e in eng.profile():
        k = by.setdefault(e['kernel'], [0.0, 0.0, 0])
        k[0] += e['total_ms']; k[1] += e['flop_per_launch'] * e['launches']; k[2] += e['launches']
    for name, (ms, fl, n) in sorted(by.items(), key=lambda kv: -kv[1][0]):
        print('  %-28s %3d launches %8.3f ms  %6.1f TFLOP/s' % (name, n, ms, fl / ms / 1e9 if ms else 0))
