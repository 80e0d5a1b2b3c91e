#!/usr/bin/env python
"""Occupancy experiment: same kernels, co-resident blocks per CU capped through the dynamic LDS request."""
import importlib, os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
native = importlib.import_module('chainer_realtime_multi-person_pose_estimation_amd.native')
eng = native.Engine(0, max_batch=1, max_h=64, max_w=64)
rng = np.random.default_rng(0)
def run(tag, B, cin, cout, k, hw, variant, min_lds, iters=10, relu_in=True):
    x = rng.standard_normal((B, cin, hw, hw)).astype('f')
    if relu_in:
        x = np.maximum(x, 0)
    w = (rng.standard_normal((cout, cin, k, k)) / np.sqrt(cin * k * k)).astype('f')
    b = rng.standard_normal(cout).astype('f')
    eng.set_option('force_variant_k%d' % k, variant)
    eng.set_option('conv_min_lds', min_lds)
    y, ms = eng.conv2d(x, w, b, relu=True, iters=iters)
    eng.set_option('conv_min_lds', 0)
    tf = 2.0 * B * hw * hw * cin * cout * k * k / ms / 1e9
    print('%-22s B=%3d v=%2d min_lds=%6d  %8.3f ms  %6.1f TF/s (%.1f%%)' % (tag, B, variant, min_lds, ms, tf, tf / 1.573), flush=True)
for v, name in ((18, 'v3 strip 7x7'), (10, 'v2 strip 7x7')):
    for lds in (0, 82 * 1024):     # v3: 2 blocks per CU by default; 82 KB -> 1 block per CU
        for B in (64, 128):
            run(name, B, 128, 128, 7, 46, v, lds)
run('v3 strip 7x7 dense in', 64, 128, 128, 7, 46, 18, 0, relu_in=False)
run('v3 8x16 7x7', 64, 128, 128, 7, 46, 20, 0)
for v, name in ((21, 'v3 8x16 3x3'), (13, 'v2 8x16 3x3')):
    for lds in (0, 82 * 1024):
        run(name, 8, 128, 128, 3, 184, v, lds)
run('v3 8x16 3x3 n64', 4, 64, 64, 3, 368, 22, 0)
run('v1 8x16 3x3 n64', 4, 64, 64, 3, 368, 2, 0)
run('v3 strip 3x3 c512', 32, 512, 512, 3, 46, 19, 0)
run('v2 strip 3x3 c512', 32, 512, 512, 3, 46, 11, 0)
