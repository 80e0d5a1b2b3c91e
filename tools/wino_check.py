#!/usr/bin/env python
"""Winograd F(2x2, 3x3) conv path (option "conv_algo" = 1) vs the direct fp32-MFMA kernels: bit-exactness against its plain-C twin
(oracle/conv_fma_ref.c::conv_wino_ref), accuracy against float64, per-layer speed, and whole-network time at batch 32."""
import argparse, importlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = 'chainer_realtime_multi-person_pose_estimation_amd'
ap = argparse.ArgumentParser()
ap.add_argument('--json', default=None)
ap.add_argument('--batch', type=int, default=32)
a = ap.parse_args()
native = importlib.import_module(PKG + '.native')
W = importlib.import_module(PKG + '.weights')
from oracle import conv_fma_ref as R
import torch
out = {'exact': [], 'conv': []}
eng = native.Engine(0, max_batch=a.batch, max_h=368, max_w=368)
rng = np.random.default_rng(0)
# ---- bit-exactness vs the C twin on small / ragged shapes (blocks >= 2 x CUs is needed for the Winograd kernel to be chosen)
for (B, cin, H, Wd, cout, ks, relu, pool) in [(2, 32, 30, 34, 128, 3, 1, 0), (3, 64, 22, 18, 128, 3, 0, 1), (1, 96, 17, 33, 256, 3, 1, 0), (2, 32, 9, 15, 130, 3, 1, 0),
                                                (1, 64, 46, 46, 128, 3, 1, 0), (2, 32, 19, 21, 128, 7, 1, 0), (1, 64, 46, 46, 128, 7, 1, 0), (1, 96, 9, 40, 100, 7, 0, 0)]:
    x = rng.standard_normal((B, cin, H, Wd)).astype('f'); w = (rng.standard_normal((cout, cin, ks, ks)) / np.sqrt(cin * ks * ks)).astype('f')
    b = rng.standard_normal(cout).astype('f')
    eng.set_option('conv_algo', 2)
    y = eng.conv2d(x, w, b, relu=bool(relu), pool=bool(pool))
    eng.set_option('conv_algo', 0)
    yd = eng.conv2d(x, w, b, relu=bool(relu), pool=bool(pool))
    o = R.conv_wino(x, w, b, relu, pool)
    row = dict(shape=[B, cin, H, Wd, cout, ks, relu, pool], identical=bool(np.array_equal(y, o)), max_diff_vs_twin=float(np.abs(y - o).max()),
               max_diff_vs_direct=float(np.abs(y - yd).max()), ran_wino=not np.array_equal(y, yd))
    out['exact'].append(row); print(row, flush=True)
# ---- speed / accuracy on the network's 3x3 shapes
for name, B, cin, H, Wd, cout, pool in [('Mconv 7x7 128->128 46', 2 * a.batch, 128, 46, 46, 128, False), ('Mconv1 7x7 192->128 46', 2 * a.batch, 192, 46, 46, 128, False),
                                         ('conv2_1 64->128 184', a.batch, 64, 184, 184, 128, False), ('conv2_2 128->128 184 pool', a.batch, 128, 184, 184, 128, True),
                                         ('conv3_1 128->256 92', a.batch, 128, 92, 92, 256, False), ('conv3_2 256->256 92', a.batch, 256, 92, 92, 256, False),
                                         ('conv3_4 256->256 92 pool', a.batch, 256, 92, 92, 256, True), ('conv4_1 256->512 46', a.batch, 256, 46, 46, 512, False),
                                         ('conv4_2 512->512 46', a.batch, 512, 46, 46, 512, False), ('conv4_4 256->128 46', a.batch, 256, 46, 46, 128, False)]:
    x = np.maximum(rng.standard_normal((B, cin, H, Wd)), 0).astype('f')
    ks = 7 if '7x7' in name else 3
    w = (rng.standard_normal((cout, cin, ks, ks)) / np.sqrt(cin * ks * ks)).astype('f')
    b = rng.standard_normal(cout).astype('f')
    with torch.no_grad():
        ref = torch.relu(torch.nn.functional.conv2d(torch.from_numpy(x[:1]).double(), torch.from_numpy(w).double(), torch.from_numpy(b).double(), padding=ks // 2))
        if pool:
            ref = torch.nn.functional.max_pool2d(ref, 2, 2)
        ref = ref.numpy()
    res = {}
    for algo in (0, 1):
        eng.set_option('conv_algo', algo)
        res[algo] = eng.conv2d(x, w, b, relu=True, pool=pool, iters=10)
    eng.set_option('conv_algo', 0)
    scale = np.abs(ref).max()
    flop = 2.0 * B * H * Wd * cout * cin * ks * ks
    row = dict(shape=name, direct_ms=res[0][1], wino_ms=res[1][1], speedup=res[0][1] / res[1][1], direct_tflops=flop / res[0][1] / 1e9,
               wino_equiv_tflops=flop / res[1][1] / 1e9, err_direct_vs_f64=float(np.abs(res[0][0][:1] - ref).max() / scale),
               err_wino_vs_f64=float(np.abs(res[1][0][:1] - ref).max() / scale))
    out['conv'].append(row)
    print('%-28s direct %.3f ms (%.1f TF)  winograd %.3f ms (%.1f TF-equiv)  x%.2f | err/scale vs f64: direct %.2e  winograd %.2e'
          % (name, row['direct_ms'], row['direct_tflops'], row['wino_ms'], row['wino_equiv_tflops'], row['speedup'], row['err_direct_vs_f64'], row['err_wino_vs_f64']), flush=True)
# ---- whole network
w = W.synthetic_weights(0); eng.set_weights(w)
imgs = np.random.default_rng(1).integers(0, 256, (a.batch, 368, 368, 3), dtype=np.uint8)
net = {}
for algo in (0, 1):
    eng.set_option('conv_algo', algo)
    for _ in range(2):
        eng.forward_u8(imgs)
    eng.get_maps()
    t0 = time.perf_counter()
    for _ in range(5):
        eng.forward_u8(imgs)
    maps = eng.get_maps()
    net[algo] = ((time.perf_counter() - t0) / 5 * 1e3, maps)
eng.set_option('conv_algo', 0)
d = max(np.abs(net[0][1][0] - net[1][1][0]).max(), np.abs(net[0][1][1] - net[1][1][1]).max())
sc = max(np.abs(net[0][1][0]).max(), np.abs(net[0][1][1]).max())
out['network'] = dict(batch=a.batch, direct_ms=net[0][0], wino_ms=net[1][0], max_map_diff=float(d), map_scale=float(sc))
print('network forward batch %d: direct %.2f ms, winograd %.2f ms (%.1f -> %.1f frames/s); max map diff %.2e (scale %.2e)'
      % (a.batch, net[0][0], net[1][0], a.batch / net[0][0] * 1e3, a.batch / net[1][0] * 1e3, d, sc))
if a.json:
    json.dump(out, open(a.json, 'w'), indent=1)
