#!/usr/bin/env python
"""One 368 x 368 image per call (BASELINE config 2, the reference's own usage) under alternative engine options, same process, interleaved:
ms per detect_batch + results call and the per-layer profile of each setting.
usage: single_image_ab.py out.json KEY=V[,KEY=V...] KEY=V ...      (each argument one setting; `base` = defaults)"""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
import torch
native = importlib.import_module(bench.PKG + '.native')
weights_mod = importlib.import_module(bench.PKG + '.weights')
out_path, settings = sys.argv[1], sys.argv[2:] or ['base']
S = int(os.environ.get('AB_SIZE', '368'))
H, W = S, int(os.environ.get('AB_W', str(S)))
eng = native.Engine(0, max_batch=1, max_h=H, max_w=W)
wts = weights_mod.synthetic_weights(0)
eng.set_weights(wts)
cal = np.random.default_rng(1234).integers(0, 256, (1, H, W, 3), dtype=np.uint8)
eng.forward_u8(cal)
paf, heat = eng.get_maps()
wts = weights_mod.calibrate_head(wts, paf[0], heat[0])
eng.set_weights({k: wts[k] for k in ('Mconv7_stage6_L1', 'Mconv7_stage6_L2')})
img = torch.from_numpy(np.random.default_rng(2).integers(0, 256, (1, H, W, 3), dtype=np.uint8)).to('cuda:0')
mh, mw = H * 320 // 368 // 8 * 8, W * 320 // 368 // 8 * 8
def parse(s):
    return [] if s == 'base' else [(kv.split('=')[0], int(kv.split('=')[1])) for kv in s.split(',')]
def apply(s, on):
    for k, v in parse(s):
        eng.set_option(k, v if on else DEFAULTS.get(k, 0))
DEFAULTS = {'wino_unit_g': 0, 'conv_algo': 1, 'wino_unit_eff': 80}
def call():
    eng.detect_batch(device_ptr=img.data_ptr(), shape=(1, H, W), map_h=mh, map_w=mw); return eng.results()
res = {s: [] for s in settings}
for rep in range(3):
    for s in settings:
        apply(s, True)
        for _ in range(3): call()
        t0 = time.perf_counter()
        for _ in range(30): call()
        res[s].append((time.perf_counter() - t0) / 30 * 1e3)
        apply(s, False)
out = {'what': __doc__.split('usage')[0].strip(), 'size': [H, W], 'ms_per_call': {s: {'runs': v, 'min': min(v)} for s, v in res.items()}, 'layers': {}}
for s in settings:
    apply(s, True)
    call()
    eng.profile_reset(); eng.profile_enable(1)
    for _ in range(5): call()
    prof = eng.profile(); eng.profile_enable(False); eng.profile_reset()
    apply(s, False)
    out['layers'][s] = [{'layer': p['layer'], 'kernel': p['kernel'], 'avg_us': p['total_ms'] / p['launches'] * 1e3} for p in prof]
    out['ms_per_call'][s]['kernel_ms'] = sum(p['total_ms'] for p in prof) / 5
json.dump(out, open(out_path, 'w'), indent=1)
for s in settings:
    print(s, out['ms_per_call'][s])
a, b = settings[0], settings[-1]
la, lb = {(l['layer']): l for l in out['layers'][a]}, {(l['layer']): l for l in out['layers'][b]}
for k in la:
    if k in lb and (la[k]['kernel'] != lb[k]['kernel'] or abs(la[k]['avg_us'] - lb[k]['avg_us']) > 3):
        print('%-22s %-34s %7.1f us | %-34s %7.1f us' % (k, la[k]['kernel'], la[k]['avg_us'], lb[k]['kernel'], lb[k]['avg_us']))
