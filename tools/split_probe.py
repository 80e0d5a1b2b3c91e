#!/usr/bin/env python
"""Would a batch whose launches end in a half-empty round of the 256 CUs be cheaper as TWO calls -- the images that fill whole rounds, then the
remainder (which the selection runs in its latency-oriented forms)?  Times detect_batch + results at batch B and at (n0, B - n0) for the
landscape (46 x 62 maps: 48 blocks per image and 7x7 layer) and square (32 blocks per image) cases.  The sum of two calls over-states the
cost of a split inside one forward (second post-process, second record copy, a synchronisation).  -> gpurun_out/<tag>/split_probe.json"""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
import torch
native = importlib.import_module(bench.PKG + '.native')
weights_mod = importlib.import_module(bench.PKG + '.weights')
dev = torch.device('cuda:0')
cases = [((368, 496), 48, (8, 24, 12, 20)), ((368, 368), 32, (12, 20, 28, 4))]
out = {'what': __doc__.split('->')[0].strip(), 'cases': []}
for (h, w), bpi, batches in cases:
    eng = native.Engine(0, max_batch=32, max_h=h, max_w=w)
    wts = weights_mod.synthetic_weights(0)
    eng.set_weights(wts)
    cal = np.random.default_rng(1234).integers(0, 256, (1, h, w, 3), dtype=np.uint8)
    eng.forward_u8(cal)
    paf, heat = eng.get_maps()
    wts = weights_mod.calibrate_head(wts, paf[0], heat[0])
    eng.set_weights({k: wts[k] for k in ('Mconv7_stage6_L1', 'Mconv7_stage6_L2')})
    imgs = torch.from_numpy(np.random.default_rng(2).integers(0, 256, (32, h, w, 3), dtype=np.uint8)).to(dev)
    mh, mw = h * 320 // 368 // 8 * 8, w * 320 // 368 // 8 * 8
    def t(parts, steps=8):
        def once():
            o = 0
            for n in parts:
                eng.detect_batch(device_ptr=imgs.data_ptr() + o * h * w * 3, shape=(n, h, w), map_h=mh, map_w=mw); eng.results()
                o += n
        for _ in range(2): once()
        t0 = time.perf_counter()
        for _ in range(steps): once()
        return (time.perf_counter() - t0) / steps * 1e3
    for B in batches:
        row = {'size': '%dx%d' % (h, w), 'batch': B, 'whole_ms': t([B])}
        full = (B * bpi) // 256 * 256 // bpi           # images whose 7x7 blocks fill whole rounds
        cands = sorted({n for n in (full, full + 1, B - 1, B - 2, B - 3, B - 4) if 0 < n < B})
        row['splits'] = {'%d+%d' % (n, B - n): t([n, B - n]) for n in cands}
        best = min(row['splits'].items(), key=lambda kv: kv[1])
        row['best'] = best[0]; row['best_over_whole'] = best[1] / row['whole_ms']
        out['cases'].append(row)
        print(json.dumps(row), flush=True)
    eng.close()
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], 'w'), indent=1)
