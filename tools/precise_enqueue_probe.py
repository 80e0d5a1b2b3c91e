"""detect_precise (482 x 642): how long does the HOST take to enqueue the four scales + finish + post-process (before it blocks on the
results), against the whole call?  Is the path enqueue-bound on this box?"""
import importlib, math, os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
PD = importlib.import_module(bench.PKG + '.pose_detector'); W_ = importlib.import_module(bench.PKG + '.weights'); E = importlib.import_module(bench.PKG + '.entity')
H, W = 482, 642
img = np.random.default_rng(55).integers(0, 256, (H, W, 3), dtype=np.uint8)
base = W_.synthetic_weights(0)
eng0 = importlib.import_module(bench.PKG + '.native').Engine(0, max_batch=1, max_h=368, max_w=496)
eng0.set_weights(base)
cal = PD.resize_cubic_u8(img, int(np.ceil(W * 368 / min(H, W))), int(np.ceil(H * 368 / min(H, W))))
cal = np.pad(cal, ((0, 368 - cal.shape[0]), (0, 496 - cal.shape[1]), (0, 0)), constant_values=110)
eng0.forward_u8(cal[None]); paf0, heat0 = eng0.get_maps(); eng0.close()
wts = W_.calibrate_head(base, paf0[0], heat0[0], heat_s=0.2, heat_t=-0.2, paf_s=1.2)
det = PD.PoseDetector(weights=wts, device=0, precise=True, max_size=(736, 984))
eng = det.engine
sizes = []
for scale in E.params['inference_scales']:
    m = scale * E.params['inference_img_size'] / min(H, W)
    sizes.append((math.ceil(H * m), math.ceil(W * m)))
rows = []
for it in range(14):
    eng.synchronize()
    t0 = time.perf_counter()
    eng.precise_begin(H, W, 1)
    marks = []
    for slot in (3, 2, 1, 0):
        eng.precise_add_scale(img, sizes[slot][0], sizes[slot][1], slot=slot)
        marks.append(time.perf_counter() - t0)
    eng.precise_finish()
    eng.postprocess(H, W, img_len=W)
    t_enq = time.perf_counter() - t0
    try:
        eng.results()
    except Exception:
        pass
    t_all = time.perf_counter() - t0
    rows.append((t_enq * 1e3, t_all * 1e3, [m * 1e3 for m in marks]))
for r in rows[2:]:
    print('enqueue %.2f ms  whole call %.2f ms   add_scale done at %s' % (r[0], r[1], ' '.join('%.2f' % m for m in r[2])))
print('median enqueue %.2f  median call %.2f' % (statistics.median(r[0] for r in rows[2:]), statistics.median(r[1] for r in rows[2:])))
det.engine.close()
