#!/usr/bin/env python
"""Run-to-run determinism of the paths with concurrency or heterogeneous launches: detect_precise (four scales in flight on prioritised
lanes) N times on one frame, a mixed-size batch N times, single images N times -- records and maps must be bit-identical every time."""
import argparse, importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
ap = argparse.ArgumentParser()
ap.add_argument('--precise', type=int, default=60); ap.add_argument('--mixed', type=int, default=60); ap.add_argument('--single', type=int, default=300)
a = ap.parse_args()
native = importlib.import_module(bench.PKG + '.native'); W = importlib.import_module(bench.PKG + '.weights'); PD = importlib.import_module(bench.PKG + '.pose_detector')
w = W.synthetic_weights(0)
eng = native.Engine(0, max_batch=1, max_h=368, max_w=368); eng.set_weights(w)
eng.forward_u8(np.random.default_rng(1234).integers(0, 256, (1, 368, 368, 3), dtype=np.uint8)); paf, heat = eng.get_maps()
w = W.calibrate_head(w, paf[0], heat[0]); eng.set_weights({k: w[k] for k in ('Mconv7_stage6_L1', 'Mconv7_stage6_L2')})
bad = 0
# single images
img = np.random.default_rng(3).integers(0, 256, (1, 368, 368, 3), dtype=np.uint8)
eng.detect_batch(img, 320, 320); ref = eng.results().tobytes(); refm = eng.get_maps()
t0 = time.perf_counter()
for i in range(a.single):
    eng.detect_batch(img, 320, 320)
    bad += eng.results().tobytes() != ref
    if i % 100 == 99:
        m = eng.get_maps(); bad += not (np.array_equal(m[0], refm[0]) and np.array_equal(m[1], refm[1]))
print('single image: %d calls in %.1f s, mismatching: %d' % (a.single, time.perf_counter() - t0, bad)); sys.stdout.flush()
eng.close()
# mixed batches
rng = np.random.default_rng(7)
classes = [(480, 640), (640, 480), (427, 640), (375, 500), (640, 640), (333, 500)]
imgs = [rng.integers(0, 256, classes[int(rng.integers(0, len(classes)))] + (3,), dtype=np.uint8) for _ in range(16)]
det = PD.PoseDetector(weights=w, device=0, max_batch=16, max_size=(368, 496))
r0 = det.detect_batch(imgs)
m0 = [det.engine.image_maps(i) for i in range(len(imgs))]
bad_m = 0
t0 = time.perf_counter()
for i in range(a.mixed):
    r = det.detect_batch(imgs)
    bad_m += not all(np.array_equal(np.asarray(x[0]), np.asarray(y[0])) and np.array_equal(np.asarray(x[1]), np.asarray(y[1])) for x, y in zip(r, r0))
    if i % 20 == 19:
        bad_m += not all(np.array_equal(det.engine.image_maps(k)[0], m0[k][0]) and np.array_equal(det.engine.image_maps(k)[1], m0[k][1]) for k in range(len(imgs)))
print('mixed batch of 16 (6 sizes): %d calls in %.1f s, mismatching: %d' % (a.mixed, time.perf_counter() - t0, bad_m)); sys.stdout.flush()
det.engine.close()
# detect_precise
frame = np.random.default_rng(55).integers(0, 256, (482, 642, 3), dtype=np.uint8)
det = PD.PoseDetector(weights=w, device=0, precise=True, max_size=(736, 984))
def run():
    try:
        det._detect_precise_device(frame, fetch_maps=True)
    except IndexError:
        pass
    return det.pafs.copy(), det.heatmaps.copy(), det.engine.results().tobytes()
p0 = run()
bad_p = 0
t0 = time.perf_counter()
for i in range(a.precise):
    p = run()
    bad_p += not (np.array_equal(p[0], p0[0]) and np.array_equal(p[1], p0[1]) and p[2] == p0[2])
print('detect_precise (482 x 642, four scales in flight, priorities): %d calls in %.1f s, mismatching: %d' % (a.precise, time.perf_counter() - t0, bad_p))
det.engine.close()
sys.exit(1 if bad + bad_m + bad_p else 0)
