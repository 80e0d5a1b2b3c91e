#!/usr/bin/env python
"""Torch-free driver for rocprofv3 runs: B images through the whole hot path, `steps` times (host-uploaded input).

    rocprofv3 --kernel-trace --stats -d out -o name -- python tools/profile_driver.py --batch 32 --steps 3
    rocprofv3 --pmc FETCH_SIZE -d out -o name -- python tools/profile_driver.py --batch 32 --steps 1
"""
import argparse
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = 'chainer_realtime_multi-person_pose_estimation_amd'

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=32)
ap.add_argument('--steps', type=int, default=3)
ap.add_argument('--size', type=int, default=368)
ap.add_argument('--k7', type=int, default=-1, help='force a 7x7 kernel variant')
ap.add_argument('--k3', type=int, default=-1)
ap.add_argument('--gen', type=int, default=0, help='kernel generation (0 = library default)')
ap.add_argument('--profile-json', default=None)
ap.add_argument('--precision', type=int, default=0, help='1 = bf16x3 kernels where a v6 kernel would run')
ap.add_argument('--algo', type=int, default=0, help='conv_algo option: 1 = Winograd F(2x2,3x3) for the 3x3 / 7x7 layers of large launches')
ap.add_argument('--lib', default=None, help='load this build of the library instead of the product one (tools/kernel_variants.py)')
ap.add_argument('--opt', action='append', default=[], help='engine option key=value (repeatable), e.g. --opt wino_geom=0')
a = ap.parse_args()
native = importlib.import_module(PKG + '.native')
if a.lib:
    native.LIB_PATH = os.path.abspath(a.lib)
weights_mod = importlib.import_module(PKG + '.weights')
B, S = a.batch, a.size
eng = native.Engine(0, max_batch=B, max_h=S, max_w=S)
w = weights_mod.synthetic_weights(0)
eng.set_weights(w)
cal = np.random.default_rng(1234).integers(0, 256, (1, S, S, 3), dtype=np.uint8)
eng.forward_u8(cal)
paf, heat = eng.get_maps()
w = weights_mod.calibrate_head(w, paf[0], heat[0])
eng.set_weights({k: w[k] for k in ('Mconv7_stage6_L1', 'Mconv7_stage6_L2')})
eng.set_option('force_variant_k7', a.k7)
eng.set_option('force_variant_k3', a.k3)
if a.gen:
    eng.set_option('kernel_gen', a.gen)
if a.precision:
    eng.set_option('precision', a.precision)
if a.algo:
    eng.set_option('conv_algo', a.algo)
for kv in a.opt:
    k, v = kv.split('=')
    eng.set_option(k, int(v))
imgs = np.random.default_rng(1).integers(0, 256, (B, S, S, 3), dtype=np.uint8)
if a.profile_json:
    eng.profile_enable(True)
import time
eng.detect_batch(imgs, 320 * S // 368 // 8 * 8 if S != 368 else 320, 320 * S // 368 // 8 * 8 if S != 368 else 320); eng.results()
_t0 = time.perf_counter()
for _ in range(a.steps):
    eng.detect_batch(imgs, 320 * S // 368 // 8 * 8 if S != 368 else 320, 320 * S // 368 // 8 * 8 if S != 368 else 320)
    rec = eng.results()
if a.profile_json:
    import json
    json.dump({'batch': B, 'steps': a.steps, 'entries': eng.profile()}, open(a.profile_json, 'w'), indent=1)
print('B=%d k7=%d k3=%d gen=%d %s: %.3f ms/step  %.3f ms/frame' % (B, a.k7, a.k3, a.gen, ' '.join(a.opt), (time.perf_counter() - _t0) / a.steps * 1e3, (time.perf_counter() - _t0) / a.steps * 1e3 / B))
print('people/frame %.2f peaks/frame %.1f status %d' % (rec['n_people'].mean(), rec['n_peaks'].mean(), int(np.bitwise_or.reduce(rec['status']))))
eng.close()
