#!/usr/bin/env python
"""Ablation / sweep of the dominant conv kernel through pmx_conv2d (timing only; dbg flags give wrong results)."""
import importlib, os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
native = importlib.import_module('chainer_realtime_multi-person_pose_estimation_amd.native')
eng = native.Engine(0, max_batch=1, max_h=64, max_w=64)
rng = np.random.default_rng(0)
out = []
def run(tag, B, cin, cout, k, hw, variant, dbg=0, iters=10):
    x = rng.standard_normal((B, cin, hw, hw)).astype('f')
    w = (rng.standard_normal((cout, cin, k, k)) / np.sqrt(cin * k * k)).astype('f')
    b = rng.standard_normal(cout).astype('f')
    eng.set_option('force_variant_k%d' % k, variant)
    eng.set_option('conv_dbg', dbg)
    y, ms = eng.conv2d(x, w, b, relu=True, iters=iters)
    eng.set_option('conv_dbg', 0)
    tf = 2.0 * B * hw * hw * cin * cout * k * k / ms / 1e9
    r = dict(tag=tag, B=B, cin=cin, cout=cout, k=k, hw=hw, variant=variant, dbg=dbg, ms=ms, tflops=tf)
    out.append(r)
    print('%-28s B=%3d cout=%3d v=%2d dbg=%2d  %8.3f ms  %6.1f TF/s' % (tag, B, cout, variant, dbg, ms, tf), flush=True)
# 1. occupancy / tail sweep: 23 strips per image, single group, BN=128 -> 23*B blocks; 1024 resident slots (v2)
for B in (22, 44, 45, 64, 88, 89, 128):
    run('sweep_v2_strip', B, 128, 128, 7, 46, 10)
for B in (33, 64, 66, 67):
    run('sweep_v1_strip', B, 128, 128, 7, 46, 8)
# 2. ablation at B=64 (1472 blocks, like the two-group launch of the network at B=32)
for dbg in (0, 1, 2, 4, 8, 3, 15):
    run('ablate_v2_strip', 64, 128, 128, 7, 46, 10, dbg)
# 3. cout=256 single launch (2 N blocks) vs 128
run('v2_strip_cout256', 32, 128, 256, 7, 46, 10)
run('v2_t8x16', 64, 128, 128, 7, 46, 12)
json.dump(out, open(os.path.join(ROOT, 'gpurun_out', 'conv_ablate.json'), 'w'), indent=1)
