#!/usr/bin/env python
"""One conv configuration through pmx_conv2d (for rocprofv3 PMC runs)."""
import argparse, importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument('--variant', type=int, default=17, help='index into conv_mfma.hip g_variants (17 = 7x7 v6)'); ap.add_argument('--B', type=int, default=64)
ap.add_argument('--k', type=int, default=7); ap.add_argument('--hw', type=int, default=46)
ap.add_argument('--cin', type=int, default=128); ap.add_argument('--cout', type=int, default=128)
ap.add_argument('--min-lds', type=int, default=0); ap.add_argument('--iters', type=int, default=3)
a = ap.parse_args()
native = importlib.import_module('chainer_realtime_multi-person_pose_estimation_amd.native')
eng = native.Engine(0, max_batch=1, max_h=64, max_w=64)
rng = np.random.default_rng(0)
x = np.maximum(rng.standard_normal((a.B, a.cin, a.hw, a.hw)), 0).astype('f')
w = (rng.standard_normal((a.cout, a.cin, a.k, a.k)) / np.sqrt(a.cin * a.k * a.k)).astype('f')
eng.set_option('force_variant_k%d' % a.k, a.variant)
eng.set_option('conv_min_lds', a.min_lds)
y, ms = eng.conv2d(x, w, np.zeros(a.cout, 'f'), relu=True, iters=a.iters)
print('variant %d B %d min_lds %d: %.3f ms %.1f TF/s' % (a.variant, a.B, a.min_lds, ms, 2.0 * a.B * a.hw * a.hw * a.cin * a.cout * a.k * a.k / ms / 1e9))
