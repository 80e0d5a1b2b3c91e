#!/usr/bin/env python
"""Whole-network forward time per batch size with the direct kernels (conv_algo 0), the launch-size rule (1) and Winograd on every
eligible layer (2): where does the Winograd kernel start to pay?"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = 'chainer_realtime_multi-person_pose_estimation_amd'
native = importlib.import_module(PKG + '.native')
W = importlib.import_module(PKG + '.weights')
eng = native.Engine(0, max_batch=32, max_h=368, max_w=368)
eng.set_weights(W.synthetic_weights(0))
imgs = np.random.default_rng(1).integers(0, 256, (32, 368, 368, 3), dtype=np.uint8)
def run(B):
    x = imgs[:B]
    for _ in range(2):
        eng.forward_u8(x)
    eng.get_maps()
    n = max(3, 24 // B)
    t0 = time.perf_counter()
    for _ in range(n):
        eng.forward_u8(x)
    eng.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


FILLS = (45, 50, 56, 62, 70, 80)
EFFS = (1, 60, 80, 100)         # wino_unit_eff: 1 = unit mode practically never
print('ms per forward; rule = conv_algo 1 with wino_min_fill = ' + ' / '.join(str(f) for f in FILLS) + '; units = conv_algo 1 with wino_unit_eff = ' + ' / '.join(str(f) for f in EFFS))
for B in (1, 2, 3, 4, 5, 6, 8, 10, 12, 16, 24, 32):
    eng.set_option('conv_algo', 0); d = run(B)
    eng.set_option('conv_algo', 2); w = run(B)
    eng.set_option('conv_algo', 1)
    r = []
    for f in FILLS:
        eng.set_option('wino_min_fill', f); r.append(run(B))
    eng.set_option('wino_min_fill', 50)
    u = []
    for e in EFFS:
        eng.set_option('wino_unit_eff', e); u.append(run(B))
    eng.set_option('wino_unit_eff', 80)
    print('B=%2d  direct %6.2f  winograd-all %6.2f  rule %s  units %s' % (B, d, w, ' / '.join('%6.2f' % v for v in r), ' / '.join('%6.2f' % v for v in u)), flush=True)
