#!/usr/bin/env python
"""Latency of the whole hot path for small batches (the reference's own usage: one image per call)."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
native = importlib.import_module('chainer_realtime_multi-person_pose_estimation_amd.native')
W = importlib.import_module('chainer_realtime_multi-person_pose_estimation_amd.weights')
eng = native.Engine(0, max_batch=32, max_h=368, max_w=368)
w = W.synthetic_weights(0); eng.set_weights(w)
cal = np.random.default_rng(1234).integers(0, 256, (1, 368, 368, 3), dtype=np.uint8)
eng.forward_u8(cal); paf, heat = eng.get_maps()
w = W.calibrate_head(w, paf[0], heat[0]); eng.set_weights({k: w[k] for k in ('Mconv7_stage6_L1', 'Mconv7_stage6_L2')})
for B in (1, 2, 4, 8, 16, 32):
    imgs = np.random.default_rng(B).integers(0, 256, (B, 368, 368, 3), dtype=np.uint8)
    for _ in range(2):
        eng.detect_batch(imgs, 320, 320); eng.results()
    n = 10
    t0 = time.perf_counter()
    for _ in range(n):
        eng.detect_batch(imgs, 320, 320); eng.results()
    dt = (time.perf_counter() - t0) / n
    print('B=%2d  %.2f ms/call  %.2f ms/frame  %.1f frames/s (host upload included)' % (B, dt * 1e3, dt * 1e3 / B, B / dt), flush=True)
