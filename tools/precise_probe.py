"""detect_precise (482 x 642, one image per call) in one process: stream priorities x kernel selection in the lanes x enqueue order, every
configuration twice, interleaved; median of 10 calls after 2 warm-up calls.  Optional first argument `torch`: initialise torch on the
device first (as bench.py does).  -> profiles/rNN_precise_probe.json"""
import importlib, json, os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
if 'torch' in sys.argv:
    import torch
    torch.zeros(8, device='cuda:0'); torch.cuda.synchronize()
import bench
PD = importlib.import_module(bench.PKG + '.pose_detector'); W_ = importlib.import_module(bench.PKG + '.weights')
H, W = 482, 642
img = np.random.default_rng(55).integers(0, 256, (H, W, 3), dtype=np.uint8)
wts = [None]


def probe(prio, plain, largest_first, lanes=4):
    det = PD.PoseDetector(weights=wts[0] or W_.synthetic_weights(0), device=0, precise=True, max_size=(736, 984))
    det.precise_largest_first = bool(largest_first)
    det.engine.set_option('precise_lane_priority', prio)
    det.engine.set_option('precise_plain', plain)
    det.engine.set_option('precise_lanes', lanes)
    if wts[0] is None:
        cal = PD.resize_cubic_u8(img, int(np.ceil(W * 368 / min(H, W))), int(np.ceil(H * 368 / min(H, W))))
        cal, _ = det.pad_image(cal, 8, (104, 117, 123))
        det.engine.forward_u8(cal[None])
        paf0, heat0 = det.engine.get_maps()
        wts[0] = W_.calibrate_head(det._weights, paf0[0], heat0[0], heat_s=0.2, heat_t=-0.2, paf_s=1.2)
        det._weights = wts[0]
        det.engine.set_weights({k: wts[0][k] for k in ('Mconv7_stage6_L1', 'Mconv7_stage6_L2')})
    ts = []
    for i in range(12):
        t0 = time.perf_counter()
        try:
            det._detect_precise_device(img, fetch_maps=False)
        except IndexError:
            pass
        ts.append((time.perf_counter() - t0) * 1e3)
    det.engine.close()
    return round(statistics.median(ts[2:]), 2), round(min(ts[2:]), 2)


out = {}
if 'lanes' in sys.argv:              # second experiment: lanes in use x kernel selection, priorities on, largest scale first
    for rep in range(2):
        for lanes in (4, 3, 2, 1):
            for plain in (0, 1):
                key = 'lanes_%d_plain_%d' % (lanes, plain)
                out.setdefault(key, []).append(probe(1, plain, 1, lanes))
                print(key, out[key]); sys.stdout.flush()
for rep in range(0 if 'lanes' in sys.argv else 2):
    for prio in (0, 1):
        for plain in (0, 1):
            for lf in (0, 1):
                key = 'priority_%d_plain_%d_largest_first_%d' % (prio, plain, lf)
                out.setdefault(key, []).append(probe(prio, plain, lf))
                print(key, out[key]); sys.stdout.flush()
js = [a for a in sys.argv[1:] if a.endswith('.json')]
if js:
    json.dump({'what': __doc__.split('->')[0].strip(), 'torch_initialised_first': 'torch' in sys.argv, 'median_min_ms': out}, open(js[0], 'w'), indent=1)
