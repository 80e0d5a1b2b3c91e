import importlib, numpy as np, sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT','/root/repo'))
native = importlib.import_module('chainer_realtime_multi-person_pose_estimation_amd.native')
eng = native.Engine(0, max_batch=1, max_h=64, max_w=64)
rng = np.random.default_rng(0)
B=8
x = np.maximum(rng.standard_normal((B, 64, 368, 368)), 0).astype('f')
for cout in (64, 128):
    w = (rng.standard_normal((cout, 64, 3, 3)) / 24).astype('f'); b = np.zeros(cout, 'f')
    for algo in (0, 2):
        eng.set_option('conv_algo', algo)
        y, ms = eng.conv2d(x, w, b, relu=True, pool=True, iters=5)
        fl = 2.0 * B * 368 * 368 * 64 * cout * 9
        print('cout %d algo %d: %.3f ms  (x4 for batch 32: %.2f ms)  %.1f TF/s algorithmic' % (cout, algo, ms, ms * 4, fl / ms / 1e9), flush=True)
