#!/usr/bin/env python
"""A stream of frames of DIFFERENT sizes (COCO-val style: 640 x 480, 480 x 640, 640 x 427, 500 x 375, 640 x 640 ...) through
PoseDetector.detect_batch: the mixed batch (one launch per layer over all size classes, pmx_detect_images) against the reference's way
(one image per call) and against a uniform batch of the same pixel count.  -> profiles/rNN_mixed_batch.json"""
import argparse, importlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = 'chainer_realtime_multi-person_pose_estimation_amd'
ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=32); ap.add_argument('--steps', type=int, default=5); ap.add_argument('--json', default=None)
a = ap.parse_args()
native = importlib.import_module(PKG + '.native')
W = importlib.import_module(PKG + '.weights')
PD = importlib.import_module(PKG + '.pose_detector')
weights = W.synthetic_weights(0)
eng = native.Engine(0, max_batch=1, max_h=368, max_w=368); eng.set_weights(weights)
eng.forward_u8(np.random.default_rng(1234).integers(0, 256, (1, 368, 368, 3), dtype=np.uint8)); paf, heat = eng.get_maps(); eng.close()
weights = W.calibrate_head(weights, paf[0], heat[0])
rng = np.random.default_rng(7)
classes = [(480, 640), (640, 480), (427, 640), (375, 500), (640, 640), (426, 640), (480, 640), (333, 500), (500, 375), (640, 427)]
sizes = [classes[int(rng.integers(0, len(classes)))] for _ in range(a.batch)]
imgs = [rng.integers(0, 256, s + (3,), dtype=np.uint8) for s in sizes]
det = PD.PoseDetector(weights=weights, device=0, max_batch=a.batch, max_size=(368, 496))
net = [det.compute_optimal_size(im, 368)[::-1] for im in imgs]
npx = sum(h * w for h, w in net)


def timed(fn, steps):
    fn(); det.engine.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        r = fn()
    det.engine.synchronize()
    return (time.perf_counter() - t0) / steps, r


t_mixed, res = timed(lambda: det.detect_batch(imgs), a.steps)
t_loop, res1 = timed(lambda: [det(im) for im in imgs], max(1, a.steps // 2))
same = sum(1 for x, y in zip(res, res1) if np.asarray(x[0]).shape == np.asarray(y[0]).shape and np.array_equal(np.asarray(x[0]), np.asarray(y[0])))
uni = [rng.integers(0, 256, (368, 496, 3), dtype=np.uint8) for _ in range(a.batch)]
t_uni, _ = timed(lambda: det.detect_batch(uni), a.steps)
out = {'what': __doc__.split('->')[0].strip(), 'batch': a.batch, 'distinct_network_sizes': len(set(net)), 'network_pixels_vs_368x368_frames': npx / (368.0 * 368.0),
       'mixed_batch_ms': t_mixed * 1e3, 'one_image_per_call_ms': t_loop * 1e3, 'speedup_vs_one_image_per_call': t_loop / t_mixed,
       'uniform_368x496_batch_ms': t_uni * 1e3,
       'ms_per_megapixel': {'mixed': t_mixed * 1e3 / (npx / 1e6), 'one_image_per_call': t_loop * 1e3 / (npx / 1e6), 'uniform_368x496': t_uni * 1e3 / (a.batch * 368 * 496 / 1e6)},
       'mixed_rate_per_pixel_vs_uniform_batch': (t_uni / (a.batch * 368 * 496)) / (t_mixed / npx),
       'frames_with_poses_identical_to_the_default_single_image_call': same, 'people_found': int(sum(len(r[1]) for r in res)),
       'note': 'host images (pageable) in every call: uploads, device cv2.resize, network, post-process, records; the per-image loop uses the default single-image kernels (unit mode / split-K)'}
print(json.dumps(out, indent=1))
if a.json:
    json.dump(out, open(a.json, 'w'), indent=1)
