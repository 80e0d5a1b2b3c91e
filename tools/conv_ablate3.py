#!/usr/bin/env python
"""Compile-time ablations of the v3 7x7 strip kernel (variants 100+ABL) at 1 and 2 blocks per CU."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
native = importlib.import_module('chainer_realtime_multi-person_pose_estimation_amd.native')
eng = native.Engine(0, max_batch=1, max_h=64, max_w=64)
rng = np.random.default_rng(0)
B = 128
x = np.maximum(rng.standard_normal((B, 128, 46, 46)), 0).astype('f')
w = (rng.standard_normal((128, 128, 7, 7)) / np.sqrt(128 * 49)).astype('f')
b = np.zeros(128, 'f')
names = {120: 'v4 twin', 121: 'v4 no B loads', 122: 'v4 no A reads', 123: 'v4 no A, no B', 124: 'v4 halo once', 127: 'v4 no A/B/staging', 25: 'v4', 10: 'v2', 100: 'full v3', 101: 'no B loads', 102: 'no A reads', 103: 'no A, no B', 104: 'halo staged once', 107: 'no A/B/staging',
         108: 'no B reg copies', 115: 'pure MFMA loop', 18: 'v3 (product)'}
for lds in (0, 84 * 1024):
    for v in (25, 120, 121, 122, 123, 124, 127, 115):
        eng.set_option('force_variant_k7', v)
        eng.set_option('conv_min_lds', lds)
        y, ms = eng.conv2d(x, w, b, relu=True, iters=8)
        tf = 2.0 * B * 46 * 46 * 128 * 128 * 49 / ms / 1e9
        print('%s blocks/CU  %-20s %8.3f ms %6.1f TF/s (%.1f%%)' % ('1' if lds else '2', names[v], ms, tf, tf / 1.573), flush=True)
