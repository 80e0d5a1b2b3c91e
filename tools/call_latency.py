#!/usr/bin/env python
"""What a caller of the mirror sees per image: PoseDetector.__call__ on a host uint8 frame (upload, device cv2.resize, network,
post-process, records, unpack) for a few frame sizes, next to Engine.detect_batch on the same frame already in HBM; and the host's share:
how long the enqueue of one call takes before it blocks on the result."""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
PD = importlib.import_module(bench.PKG + '.pose_detector'); W_ = importlib.import_module(bench.PKG + '.weights')
native = importlib.import_module(bench.PKG + '.native')
wts = W_.synthetic_weights(0)
eng = native.Engine(0, max_batch=1, max_h=368, max_w=368)
eng.set_weights(wts)
eng.forward_u8(np.random.default_rng(1234).integers(0, 256, (1, 368, 368, 3), dtype=np.uint8))
paf, heat = eng.get_maps()
eng.close()
wts = W_.calibrate_head(wts, paf[0], heat[0])
out = {'what': __doc__.strip(), 'rows': []}
det = PD.PoseDetector(weights=wts, device=0, max_batch=1, max_size=(368, 496))
for (h, w) in ((368, 368), (480, 640), (640, 480), (720, 1280)):
    img = np.random.default_rng(h + w).integers(0, 256, (h, w, 3), dtype=np.uint8)
    for _ in range(3): det(img)
    n = 30
    t0 = time.perf_counter()
    for _ in range(n): det(img)
    call_ms = (time.perf_counter() - t0) / n * 1e3
    row = {'frame': '%dx%d' % (h, w), 'network_input': 'x'.join(str(v) for v in det.compute_optimal_size(img, 368)[::-1]), 'PoseDetector_call_ms': call_ms}
    out['rows'].append(row); print(row, flush=True)
det.close() if hasattr(det, 'close') else None
eng = native.Engine(0, max_batch=1, max_h=368, max_w=368)
eng.set_weights(wts)
img = np.random.default_rng(2).integers(0, 256, (1, 368, 368, 3), dtype=np.uint8)
for _ in range(3): eng.detect_batch(img, 320, 320); eng.results()
enq, tot = [], []
for _ in range(30):
    t0 = time.perf_counter(); eng.detect_batch(img, 320, 320); t1 = time.perf_counter(); eng.results(); t2 = time.perf_counter()
    enq.append((t1 - t0) * 1e3); tot.append((t2 - t0) * 1e3)
out['engine_host_frame_368'] = {'enqueue_ms_median': float(np.median(enq)), 'call_ms_median': float(np.median(tot))}
print(out['engine_host_frame_368'])
if len(sys.argv) > 1: json.dump(out, open(sys.argv[1], 'w'), indent=1)
