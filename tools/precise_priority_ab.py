"""A/B of detect_precise (bench.py's config-5 frame, 482 x 642; one image per call and eight per call) with and without stream
priorities on the lanes (option precise_lane_priority: the lane of the largest scale first) -- and, for each, the default kernel selection
against the plain Winograd kernels everywhere (conv_algo 2).  GPU box: python tools/precise_priority_ab.py [--json out.json]"""
import importlib, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
PD = importlib.import_module(bench.PKG + '.pose_detector'); W_ = importlib.import_module(bench.PKG + '.weights')
H, W = 482, 642
img = np.random.default_rng(55).integers(0, 256, (H, W, 3), dtype=np.uint8)
batch = [img] + [np.random.default_rng(56 + i).integers(0, 256, (H, W, 3), dtype=np.uint8) for i in range(7)]
wts = None
out = {}
for rep in range(2):
    for prio in (0, 1):
        det = PD.PoseDetector(weights=wts or W_.synthetic_weights(0), device=0, precise=True, max_batch=8, max_size=(736, 984))
        det.engine.set_option('precise_lane_priority', prio)          # before the first detect_precise: read when a lane's stream is created
        if wts is None:
            cal = PD.resize_cubic_u8(img, int(np.ceil(W * 368 / min(H, W))), int(np.ceil(H * 368 / min(H, W))))
            cal, _ = det.pad_image(cal, 8, (104, 117, 123))
            det.engine.forward_u8(cal[None])
            paf0, heat0 = det.engine.get_maps()
            wts = W_.calibrate_head(det._weights, paf0[0], heat0[0], heat_s=0.2, heat_t=-0.2, paf_s=1.2)
            det._weights = wts
            det.engine.set_weights({k: wts[k] for k in ('Mconv7_stage6_L1', 'Mconv7_stage6_L2')})
        for algo in (1, 2):
            det.engine.set_option('conv_algo', algo)
            def run1():
                try: det._detect_precise_device(img, fetch_maps=False)
                except IndexError: pass
            def run8():
                try: det.detect_precise_batch(batch)
                except IndexError: pass
            for f, n, key in ((run1, 8, 'one'), (run8, 3, 'eight')):
                f(); f()
                t0 = time.perf_counter()
                for _ in range(n): f()
                ms = (time.perf_counter() - t0) / n * 1e3 / (8 if key == 'eight' else 1)
                out.setdefault('priority_%d_algo_%d' % (prio, algo), {}).setdefault(key + '_ms_per_image', []).append(round(ms, 3))
            print('priority', prio, 'conv_algo', algo, out['priority_%d_algo_%d' % (prio, algo)]); sys.stdout.flush()
        det.engine.close()
if '--json' in sys.argv:
    json.dump(out, open(sys.argv[sys.argv.index('--json') + 1], 'w'), indent=1)
