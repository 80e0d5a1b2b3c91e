#!/usr/bin/env python
"""The batch cut in two by images at the last whole round of the CUs (conv_select.hip::wino_split_images, option wino_split: 0 off, 1 =
default threshold, >= 50 = threshold in percent) against the whole batch in one launch per layer: ms per detect_batch + results step for
landscape / square inputs at the batch sizes whose 7x7 launches end in a part-filled round.  Same process, interleaved, device-resident
inputs.   usage: split_ab.py out.json [settings ...]   default settings: 0 1 97"""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
import torch
native = importlib.import_module(bench.PKG + '.native')
weights_mod = importlib.import_module(bench.PKG + '.weights')
out_path = sys.argv[1]
settings = [int(v) for v in sys.argv[2:]] or [0, 1, 97]
cases = [((368, 496), (8, 12, 20, 24, 28)), ((368, 368), (4, 12, 20, 28)), ((496, 368), (8, 24))]
out = {'what': __doc__.split('usage')[0].strip(), 'settings': settings, 'rows': []}
for (h, w), batches in cases:
    eng = native.Engine(0, max_batch=32, max_h=h, max_w=w)
    wts = weights_mod.synthetic_weights(0)
    eng.set_weights(wts)
    eng.forward_u8(np.random.default_rng(1234).integers(0, 256, (1, h, w, 3), dtype=np.uint8))
    paf, heat = eng.get_maps()
    wts = weights_mod.calibrate_head(wts, paf[0], heat[0])
    eng.set_weights({k: wts[k] for k in ('Mconv7_stage6_L1', 'Mconv7_stage6_L2')})
    imgs = torch.from_numpy(np.random.default_rng(2).integers(0, 256, (32, h, w, 3), dtype=np.uint8)).to('cuda:0')
    mh, mw = h * 320 // 368 // 8 * 8, w * 320 // 368 // 8 * 8
    for B in batches:
        res = {s: [] for s in settings}
        recs = {}
        for rep in range(2):
            for s in settings:
                eng.set_option('wino_split', s)
                for _ in range(2):
                    eng.detect_batch(device_ptr=imgs.data_ptr(), shape=(B, h, w), map_h=mh, map_w=mw); r = eng.results()
                t0 = time.perf_counter()
                for _ in range(6):
                    eng.detect_batch(device_ptr=imgs.data_ptr(), shape=(B, h, w), map_h=mh, map_w=mw); r = eng.results()
                res[s].append((time.perf_counter() - t0) / 6 * 1e3)
                recs[s] = r.copy()
        eng.set_option('wino_split', 1)
        eng.profile_reset(); eng.profile_enable(1)
        eng.detect_batch(device_ptr=imgs.data_ptr(), shape=(B, h, w), map_h=mh, map_w=mw); eng.results()
        prof = eng.profile(); eng.profile_enable(False); eng.profile_reset()
        split_layers = sorted({p['layer'] for p in prof if '@' in p['kernel']})
        base = recs[settings[0]]
        same = {str(s): bool(np.array_equal(recs[s]['n_people'], base['n_people']) and np.array_equal(recs[s]['poses'], base['poses'])) for s in settings}
        dmax = {str(s): float(np.abs(recs[s]['scores'] - base['scores']).max()) for s in settings}
        row = {'size': '%dx%d' % (h, w), 'batch': B, 'ms': {str(s): min(v) for s, v in res.items()}, 'layers_split_at_default': len(split_layers),
               'example': next((p['kernel'] for p in prof if '@' in p['kernel']), None), 'poses_equal_to_first_setting': same, 'max_abs_score_diff_vs_first_setting': dmax}
        out['rows'].append(row)
        print(json.dumps(row), flush=True)
    eng.close()
json.dump(out, open(out_path, 'w'), indent=1)
