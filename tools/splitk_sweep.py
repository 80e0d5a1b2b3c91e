#!/usr/bin/env python
"""Split-K tuning: whole-call latency and the per-layer kernel times of small batches for forced numbers of K slices.

    python tools/splitk_sweep.py [--batches 1 2 4] [--ksplit 1 0 2 4 8]      (0 = the library's automatic choice)
"""
import argparse, importlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = 'chainer_realtime_multi-person_pose_estimation_amd'
ap = argparse.ArgumentParser()
ap.add_argument('--batches', type=int, nargs='+', default=[1, 2, 4])
ap.add_argument('--ksplit', type=int, nargs='+', default=[1, 0, 2, 4, 8])
ap.add_argument('--json', default=None)
ap.add_argument('--precision', type=int, default=0)
a = ap.parse_args()
native = importlib.import_module(PKG + '.native')
W = importlib.import_module(PKG + '.weights')
eng = native.Engine(0, max_batch=max(a.batches), max_h=368, max_w=368)
w = W.synthetic_weights(0); eng.set_weights(w)
if a.precision:
    eng.set_option('precision', a.precision)
cal = np.random.default_rng(1234).integers(0, 256, (1, 368, 368, 3), dtype=np.uint8)
eng.forward_u8(cal); paf, heat = eng.get_maps()
w = W.calibrate_head(w, paf[0], heat[0]); eng.set_weights({k: w[k] for k in ('Mconv7_stage6_L1', 'Mconv7_stage6_L2')})
out = []
for B in a.batches:
    imgs = np.random.default_rng(B).integers(0, 256, (B, 368, 368, 3), dtype=np.uint8)
    for S in a.ksplit:
        eng.set_option('ksplit', S)
        for _ in range(3):
            eng.detect_batch(imgs, 320, 320); eng.results()
        n = 20
        t0 = time.perf_counter()
        for _ in range(n):
            eng.detect_batch(imgs, 320, 320); eng.results()
        ms = (time.perf_counter() - t0) / n * 1e3
        eng.profile_reset(); eng.profile_enable(True)
        for _ in range(5):
            eng.detect_batch(imgs, 320, 320); eng.results()
        prof = eng.profile(); eng.profile_enable(False)
        def grp(pred):
            return sum(p['avg_ms'] for p in prof if pred(p['layer']))
        row = dict(B=B, ksplit=S, ms_per_call=ms, conv7_128=grp(lambda l: l.startswith(('Mconv2', 'Mconv3', 'Mconv4', 'Mconv5'))),
                   conv7_185=grp(lambda l: l.startswith('Mconv1')), stem=grp(lambda l: l.startswith(('conv1', 'conv2', 'conv3', 'conv4'))),
                   stage1=grp(lambda l: l.startswith('conv5')), heads=grp(lambda l: l.startswith(('Mconv6', 'Mconv7'))),
                   pp=grp(lambda l: l.startswith('pp_')),
                   kernels={p['layer']: p['kernel'] for p in prof if p['layer'] in ('Mconv2_stage2', 'Mconv1_stage2', 'conv4_2', 'conv3_2', 'conv2_2', 'conv1_2', 'conv4_1', 'conv5_1_CPM')},
                   layer_ms={p['layer']: round(p['avg_ms'], 4) for p in prof if p['layer'] in ('Mconv2_stage2', 'Mconv1_stage2', 'conv4_1', 'conv4_2', 'conv4_3_CPM', 'conv4_4_CPM', 'conv3_2', 'conv3_4', 'conv2_2', 'conv1_2', 'conv5_1_CPM')})
        out.append(row)
        print('B=%d ksplit=%d: %.3f ms/call | 7x7x128 %.3f  7x7x185 %.3f  stem %.3f  stage1 %.3f  heads %.3f  pp %.3f | %s' % (
            B, S, ms, row['conv7_128'], row['conv7_185'], row['stem'], row['stage1'], row['heads'], row['pp'], row['layer_ms']), flush=True)
        if S == 0:
            print('   auto plan:', row['kernels'], flush=True)
if a.json:
    json.dump(out, open(a.json, 'w'), indent=1)
