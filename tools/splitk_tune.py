#!/usr/bin/env python
"""In-network split-K tuning: for each batch size and each class of layers (by number of 16-channel chunks) try explicit slice
plans and report the measured per-layer time (HIP events) -- the ground truth behind the plan table in conv_mfma.hip.
    python tools/splitk_tune.py --batches 1 2 3 4"""
import argparse, importlib, itertools, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = 'chainer_realtime_multi-person_pose_estimation_amd'
ap = argparse.ArgumentParser()
ap.add_argument('--batches', type=int, nargs='+', default=[1, 2, 3, 4])
ap.add_argument('--precision', type=int, default=0)
a = ap.parse_args()
native = importlib.import_module(PKG + '.native')
W = importlib.import_module(PKG + '.weights')
eng = native.Engine(0, max_batch=max(a.batches), max_h=368, max_w=368)
eng.set_weights(W.synthetic_weights(0))
if a.precision:
    eng.set_option('precision', a.precision)


def partitions(n, parts, maxpart):
    if parts == 0:
        if n == 0:
            yield []
        return
    for v in range(min(n - (parts - 1), maxpart), 0, -1):
        if v * parts < n:
            break
        for rest in partitions(n - v, parts - 1, v):
            yield [v] + rest


CLASSES = {8: ('Mconv2_stage2', 'Mconv3_stage2', 'Mconv4_stage3', 'Mconv5_stage4'), 12: ('Mconv1_stage2', 'Mconv1_stage5')}
EVEN = {'conv4_2': 32, 'conv4_3_CPM': 32, 'conv4_1': 16, 'conv4_4_CPM': 16, 'conv3_2': 16, 'conv3_4': 16, 'conv3_1': 8, 'conv2_2': 8,
        'conv5_1_CPM': 8, 'conv5_2_CPM': 8}


def measure(B, layers):
    imgs = np.random.default_rng(B).integers(0, 256, (B, 368, 368, 3), dtype=np.uint8)
    eng.forward_u8(imgs); eng.synchronize()
    eng.profile_reset(); eng.profile_enable(True)
    for _ in range(8):
        eng.forward_u8(imgs)
    prof = eng.profile(); eng.profile_enable(False)
    return {p['layer']: (p['avg_ms'] * 1e3, p['kernel']) for p in prof if p['layer'] in layers}


for B in a.batches:
    for nch, layers in CLASSES.items():
        cands = [[nch]]
        for S in range(2, 5):
            cands += [p for p in partitions(nch, S, 9)]
        cands += [[nch // S + (1 if s < nch % S else 0) for s in range(S)] for S in (5, 6, 8)]
        rows = []
        for plan in cands:
            if len(plan) == 1:
                eng.set_option('ksplit', 1)
            else:
                eng.set_option('ksplit_plan', int(''.join(str(v) for v in plan)))
            m = measure(B, layers)
            rows.append((np.mean([v[0] for v in m.values()]), plan))
        rows.sort(key=lambda r: r[0])
        print('B=%d nch=%d (7x7): ' % (B, nch) + '  '.join('%s %.1f' % ('-'.join(map(str, p)), t) for t, p in rows[:8]) +
              '  ...  unsplit %.1f' % [t for t, p in rows if len(p) == 1][0], flush=True)
    out = {}
    for S in (1, 2, 3, 4, 5, 6, 8):
        eng.set_option('ksplit', S)
        m = measure(B, EVEN.keys())
        for k, (t, kern) in m.items():
            out.setdefault(k, []).append((t, S))
    for k, v in out.items():
        print('B=%d %-12s nch=%2d: ' % (B, k, EVEN[k]) + '  '.join('k%d %.1f' % (S, t) for t, S in v), flush=True)
eng.set_option('ksplit', 0)
