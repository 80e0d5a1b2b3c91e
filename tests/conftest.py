import glob
import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
PKG = 'chainer_realtime_multi-person_pose_estimation_amd'
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pkg(sub=None):
    return importlib.import_module(PKG + ('.' + sub if sub else ''))


def golden_cases():
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, 'pp_*.npz')))


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + '.npz'))
    d = {k: z[k] for k in z.files}
    d['heat_lo'] = d['heat_lo'].astype(np.float32)
    d['paf_lo'] = d['paf_lo'].astype(np.float32)
    d['poses'] = d['poses'].reshape(tuple(d['poses_shape']))
    return d


def conns_by_limb(flat):
    """(n,4) rows (limb, a, b, score) -> list of 19 (k,3) arrays"""
    flat = np.asarray(flat, dtype=np.float64).reshape(-1, 4)
    return [flat[flat[:, 0] == l][:, 1:] for l in range(19)]


@pytest.fixture(scope='session')
def native():
    n = pkg('native')
    if n.needs_build():
        n.build()
    return n


@pytest.fixture(scope='session')
def engine(native):
    """A shared engine big enough for every GPU test (batch 4, 368 x 368)."""
    e = native.Engine(0, max_batch=4, max_h=368, max_w=368)
    yield e
    e.close()


def forward_plan(engine, run, with_wino=False, image=None, with_profile=False):
    """Run `run()` (one forward through `engine`) with the per-launch profiler on and return the split-K plan the kernels used
    ({layer label: K slices}, oracle/conv_fma_ref.py::splitk_plan) -- what the order-defined oracle needs to reproduce a
    small-launch forward bit for bit.  with_wino: return (plan, labels of the layers that ran as Winograd, out).  with_profile: return
    (profile, out) instead -- a forward that cut its batch in two by images has one plan PER IMAGE (splitk_plan(profile, image=i));
    `image`: the plan of that image."""
    from oracle import conv_fma_ref
    engine.profile_reset()
    engine.profile_enable(True)
    try:
        out = run()
        prof = engine.profile()
        if with_profile:
            return prof, out
        plan = conv_fma_ref.splitk_plan(prof, image=image)
        wino = conv_fma_ref.wino_layers(conv_fma_ref._for_image(prof, image))
    finally:
        engine.profile_enable(False)
        engine.profile_reset()
    return (plan, wino, out) if with_wino else (plan, out)
