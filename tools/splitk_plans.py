#!/usr/bin/env python
"""Ground truth for the split-K schedule model: one conv launch shape, explicit slice plans, measured time (pmx_conv2d).
    python tools/splitk_plans.py            (shapes of a single 368x368 image: 7x7 128->2x128, 7x7 192->2x128, 3x3 512->512 ...)"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
native = importlib.import_module('chainer_realtime_multi-person_pose_estimation_amd.native')
eng = native.Engine(0, max_batch=4, max_h=368, max_w=368)
rng = np.random.default_rng(0)
CASES = [  # (name, B, cin, H, W, cout, k, plans)
    ('7x7 128ch B1', 1, 128, 46, 46, 256, 7, [1, 2, 3, 4, 8, -3221, -2222, -332, -233, -2321, -3311, -4211, -2211_11 if False else -221111, -11111111, -44, -431, -422, -3212]),
    ('7x7 192ch B1', 1, 192, 46, 46, 256, 7, [1, 3, 4, 6, -4332, -3333, -444, -5322, -4422, -43221, -33222, -222222, -633, -6222]),
    ('3x3 512->512 B1', 1, 512, 46, 46, 512, 3, [1, 2, 4, 5, 8]),
    ('3x3 256->512 B1', 1, 256, 46, 46, 512, 3, [1, 2, 3, 4]),
    ('3x3 512->256 B1', 1, 512, 46, 46, 256, 3, [1, 2, 3, 4, 6, 8]),
    ('3x3 256->256 92 B1', 1, 256, 92, 92, 256, 3, [1, 2, 3, 4]),
    ('3x3 128->128 184 B1', 1, 128, 184, 184, 128, 3, [1, 2]),
    ('7x7 128ch B2', 2, 128, 46, 46, 256, 7, [1, 2, 3, 4, -3221, -332, -521, -44, -53, -62]),
    ('7x7 128ch B4', 4, 128, 46, 46, 256, 7, [1, 2, 3, 4, -4211, -53, -62, -71]),
]
for name, B, cin, H, W, cout, k, plans in CASES:
    x = rng.standard_normal((B, cin, H, W)).astype('f')
    w = (rng.standard_normal((cout, cin, k, k)) / np.sqrt(cin * k * k)).astype('f')
    b = rng.standard_normal(cout).astype('f')
    res = []
    for p in plans:
        if p < 0:
            eng.set_option('ksplit_plan', -p)
        else:
            eng.set_option('ksplit', p)
        _, ms = eng.conv2d(x, w, b, relu=True, iters=30)
        res.append((ms * 1e3, p))
    eng.set_option('ksplit', 0)
    _, ms = eng.conv2d(x, w, b, relu=True, iters=30)
    flop = 2.0 * B * H * W * cout * cin * k * k
    print('%-22s auto %.1f us | ' % (name, ms * 1e3) + '  '.join('%s:%.1f' % (('k%d' % p) if p > 0 else ('p%d' % -p), t) for t, p in res) +
          '   best %.1f TF/s' % (flop / min(t for t, _ in res) / 1e6), flush=True)
