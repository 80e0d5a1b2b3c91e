#!/usr/bin/env python
"""Soak: the same batch through the whole hot path N times; maps and records must be bit-identical every time (races,
uninitialised reads and missing barriers show up as run-to-run differences under sustained load)."""
import argparse, importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument('--steps', type=int, default=300); ap.add_argument('--batch', type=int, default=32)
ap.add_argument('--precision', type=int, default=0, help='1 = the opt-in bf16x3 kernels')
a = ap.parse_args()
native = importlib.import_module('chainer_realtime_multi-person_pose_estimation_amd.native')
W = importlib.import_module('chainer_realtime_multi-person_pose_estimation_amd.weights')
eng = native.Engine(0, max_batch=a.batch, max_h=368, max_w=368)
w = W.synthetic_weights(0); eng.set_weights(w)
if a.precision:
    eng.set_option('precision', a.precision)
cal = np.random.default_rng(1234).integers(0, 256, (1, 368, 368, 3), dtype=np.uint8)
eng.forward_u8(cal); paf, heat = eng.get_maps()
w = W.calibrate_head(w, paf[0], heat[0]); eng.set_weights({k: w[k] for k in ('Mconv7_stage6_L1', 'Mconv7_stage6_L2')})
imgs = np.random.default_rng(1).integers(0, 256, (a.batch, 368, 368, 3), dtype=np.uint8)
eng.detect_batch(imgs, 320, 320)
ref_rec = eng.results().tobytes()
ref_paf, ref_heat = eng.get_maps()
bad = 0
t0 = time.perf_counter()
for i in range(a.steps):
    eng.detect_batch(imgs, 320, 320)
    rec = eng.results().tobytes()
    if rec != ref_rec:
        bad += 1
    if i % 50 == 49:
        p, h = eng.get_maps()
        if not (np.array_equal(p, ref_paf) and np.array_equal(h, ref_heat)):
            bad += 1
dt = time.perf_counter() - t0
print('soak (precision %d): %d steps of batch %d in %.1f s (%.1f frames/s incl. host upload), mismatching steps: %d' % (a.precision, a.steps, a.batch, dt, a.steps * a.batch / dt, bad))
sys.exit(1 if bad else 0)
