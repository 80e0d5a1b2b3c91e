#!/usr/bin/env python
"""ms per image of detect_batch + results for every batch size 1 .. 32 (default options), square and landscape network inputs: where does
the launch-form selection leave steps in the curve?  Device-resident inputs.  usage: batch_sweep.py out.json [KEY=V ...]"""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
import torch
native = importlib.import_module(bench.PKG + '.native')
weights_mod = importlib.import_module(bench.PKG + '.weights')
opts = [kv.split('=') for kv in sys.argv[2:]]
out = {'what': __doc__.split('usage')[0].strip(), 'options': dict(opts), 'sizes': {}}
for (h, w) in ((368, 368), (368, 496)):
    eng = native.Engine(0, max_batch=32, max_h=h, max_w=w)
    wts = weights_mod.synthetic_weights(0)
    eng.set_weights(wts)
    eng.forward_u8(np.random.default_rng(1234).integers(0, 256, (1, h, w, 3), dtype=np.uint8))
    paf, heat = eng.get_maps()
    wts = weights_mod.calibrate_head(wts, paf[0], heat[0])
    eng.set_weights({k: wts[k] for k in ('Mconv7_stage6_L1', 'Mconv7_stage6_L2')})
    for k, v in opts: eng.set_option(k, int(v))
    imgs = torch.from_numpy(np.random.default_rng(2).integers(0, 256, (32, h, w, 3), dtype=np.uint8)).to('cuda:0')
    mh, mw = h * 320 // 368 // 8 * 8, w * 320 // 368 // 8 * 8
    rows = {}
    for B in range(1, 33):
        def step():
            eng.detect_batch(device_ptr=imgs.data_ptr(), shape=(B, h, w), map_h=mh, map_w=mw); eng.results()
        for _ in range(2): step()
        n = 12 if B <= 4 else 5
        best = 1e9
        for rep in range(2):
            t0 = time.perf_counter()
            for _ in range(n): step()
            best = min(best, (time.perf_counter() - t0) / n * 1e3)
        rows[B] = {'ms': best, 'ms_per_image': best / B}
        print('%dx%d B=%2d %8.3f ms %7.3f ms/image' % (h, w, B, best, best / B), flush=True)
    out['sizes']['%dx%d' % (h, w)] = rows
    eng.close()
json.dump(out, open(sys.argv[1], 'w'), indent=1)
