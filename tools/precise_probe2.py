"""Does a detector that ran a calibration forward (one 368 x 496 image, default kernel selection) + set_weights before its first
detect_precise run detect_precise slower than one constructed with the final weights?  Alternating, same options, one process."""
import importlib, os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
PD = importlib.import_module(bench.PKG + '.pose_detector'); W_ = importlib.import_module(bench.PKG + '.weights')
H, W = 482, 642
img = np.random.default_rng(55).integers(0, 256, (H, W, 3), dtype=np.uint8)
base = W_.synthetic_weights(0)
final = [None]


def probe(mode):
    det = PD.PoseDetector(weights=final[0] if mode == 'final' else base, device=0, precise=True, max_size=(736, 984))
    if mode != 'final':
        cal = PD.resize_cubic_u8(img, int(np.ceil(W * 368 / min(H, W))), int(np.ceil(H * 368 / min(H, W))))
        cal, _ = det.pad_image(cal, 8, (104, 117, 123))
        if mode == 'calibrate_plain':
            det.engine.set_option('conv_algo', 2)
        det.engine.forward_u8(cal[None])
        det.engine.set_option('conv_algo', 1)
        paf0, heat0 = det.engine.get_maps()
        w = W_.calibrate_head(base, paf0[0], heat0[0], heat_s=0.2, heat_t=-0.2, paf_s=1.2)
        final[0] = w
        det._weights = w
        if mode != 'forward_only':
            det.engine.set_weights({k: w[k] for k in ('Mconv7_stage6_L1', 'Mconv7_stage6_L2')})
    ts = []
    for i in range(12):
        t0 = time.perf_counter()
        try:
            det._detect_precise_device(img, fetch_maps=False)
        except IndexError:
            pass
        ts.append((time.perf_counter() - t0) * 1e3)
    det.engine.close()
    print(mode, round(statistics.median(ts[2:]), 2), [round(t, 1) for t in ts]); sys.stdout.flush()


for rep in range(2):
    for mode in ('calibrate', 'final', 'forward_only', 'calibrate_plain', 'final'):
        probe(mode)
