"""GPU: size-independent properties at BASELINE.json's full sizes (368 x 368 frames, batch 32), where the CPU oracle
would take minutes -- determinism, batch independence, permutation equivariance, sharding == single rank -- plus
hypothesis-driven post-process cases against the NumPy oracle (T4)."""
import os

import numpy as np
import pytest
from hypothesis import given, settings, strategies as st, HealthCheck

from conftest import pkg
from oracle import postprocess_ref as P
from oracle import fixtures as Fx

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def big(native):
    W = pkg('weights')
    eng = native.Engine(0, max_batch=32, max_h=368, max_w=368)
    w = W.synthetic_weights(0)
    eng.set_weights(w)
    cal = np.random.default_rng(1234).integers(0, 256, (1, 368, 368, 3), dtype=np.uint8)
    eng.forward_u8(cal)
    paf, heat = eng.get_maps()
    w = W.calibrate_head(w, paf[0], heat[0])
    eng.set_weights({k: w[k] for k in ('Mconv7_stage6_L1', 'Mconv7_stage6_L2')})
    imgs = np.random.default_rng(7).integers(0, 256, (32, 368, 368, 3), dtype=np.uint8)
    yield eng, imgs
    eng.close()


def _run(eng, imgs):
    eng.detect_batch(imgs, 320, 320)
    return eng.results().copy()


def _same(a, b):
    return all(np.array_equal(a[f], b[f]) for f in ('n_people', 'n_peaks', 'status', 'scores', 'poses'))


def test_full_batch32_is_deterministic_and_clean(big):
    eng, imgs = big
    r1 = _run(eng, imgs)
    r2 = _run(eng, imgs)
    assert _same(r1, r2), 'two runs on the same batch must be bit-identical'
    assert np.all(r1['status'] == 0)
    assert r1['n_peaks'].min() > 20 and r1['n_people'].sum() > 32, 'synthetic workload should exercise the post-process'


@pytest.mark.parametrize('algo', [0, 2])
def test_batch_independence_and_permutation(big, algo):
    eng, imgs = big
    # bitwise batch-size independence is a property of ONE summation order: unsplit kernels, and one algorithm at every launch size
    # (0: the direct kernels everywhere; 2: the Winograd kernel on every eligible layer, direct elsewhere)
    eng.set_option('ksplit', 1)
    eng.set_option('conv_algo', algo)
    full = _run(eng, imgs)
    perm = np.random.default_rng(0).permutation(32)
    permuted = _run(eng, imgs[perm])
    assert _same(full[perm], permuted), 'results must follow the images under a batch permutation'
    # single images and a sub-batch give the same records as inside the full batch (different kernel tile variants
    # are used for small batches: the K order of the FMA chains is identical, so results are bit-identical)
    for i in (0, 13, 31):
        one = _run(eng, imgs[i:i + 1])
        assert _same(one, full[i:i + 1]), 'image %d alone differs from the same image inside the batch' % i
    sub = _run(eng, imgs[8:16])
    assert _same(sub, full[8:16])
    # default configuration: single images are split over K and batches take the Winograd kernel (defined, different summation
    # trees): same people, scores to 1e-5
    eng.set_option('ksplit', 0)
    eng.set_option('conv_algo', 1)
    full = _run(eng, imgs)
    for i in (0, 13, 31):
        one = _run(eng, imgs[i:i + 1])[0]
        ref = full[i]
        assert one['n_peaks'] == ref['n_peaks'] and one['n_people'] == ref['n_people'] and one['status'] == 0
        assert np.array_equal(one['poses'], ref['poses'])
        assert np.abs(one['scores'] - ref['scores']).max() <= 1e-5


def test_sharded_equals_single_rank(big):
    """What bench.py does across ranks: contiguous shards processed independently == the whole batch."""
    eng, imgs = big
    d = pkg('dist')
    eng.set_option('ksplit', 1)
    eng.set_option('conv_algo', 0)
    full = _run(eng, imgs)
    for world in (2, 4, 8):
        parts = []
        for r in range(world):
            lo, hi = d.shard_range(32, r, world)
            parts.append(_run(eng, imgs[lo:hi]))
        assert _same(np.concatenate(parts), full)
    eng.set_option('ksplit', 0)
    eng.set_option('conv_algo', 1)
    # default configuration (kernel choice by launch size: shards of 4 images run other kernels than the batch of 32): same people,
    # identical poses, scores to 1e-5
    full = _run(eng, imgs)
    parts = np.concatenate([_run(eng, imgs[lo:lo + 4]) for lo in range(0, 32, 4)])
    assert all(np.array_equal(parts[f], full[f]) for f in ('n_people', 'n_peaks', 'status', 'poses'))
    assert np.abs(parts['scores'] - full['scores']).max() <= 1e-5


def test_maps_are_a_pure_function_of_the_image(big):
    eng, imgs = big
    eng.forward_u8(imgs[:4])
    p1, h1 = eng.get_maps()
    eng.forward_u8(imgs[:4][::-1].copy())
    p2, h2 = eng.get_maps()
    assert np.array_equal(p1, p2[::-1]) and np.array_equal(h1, h2[::-1])


# deterministic example set by default (the round-end run must be reproducible); PMX_FUZZ=<n> draws n fresh random examples
_FUZZ = int(os.environ.get('PMX_FUZZ', '0'))


@settings(max_examples=_FUZZ or 25, derandomize=not _FUZZ, deadline=None, database=None,
          suppress_health_check=[HealthCheck.function_scoped_fixture])
@given(seed=st.integers(0, 10 ** 6), n=st.integers(0, 10), fh=st.integers(12, 48), fw=st.integers(12, 48),
       up=st.sampled_from([1, 3, 5, 7]), noise=st.sampled_from([0.0, 0.01, 0.04]))
def test_hypothesis_postprocess_matches_oracle(engine, seed, n, fh, fw, up, noise):
    heat, paf, _ = Fx.synthetic_maps(seed, n, fh, fw, 1.0, 0.9, noise=noise, height_range=(0.3, 0.8), drop_prob=0.2)
    mh, mw = fh * up, fw * up
    try:
        ref = P.postprocess_from_net_output(paf, heat, mh, mw)
    except IndexError:
        ref = None
    engine.set_maps(paf[None], heat[None])
    engine.postprocess(mh, mw, img_len=mw)
    rec = engine.results()[0]
    if ref is None:
        assert rec['status'] & 8
        return
    assert rec['status'] == 0       # capacities grow on demand: nothing is ever truncated
    assert np.array_equal(engine.peaks(0), ref['all_peaks'])
    n_ref = len(ref['subsets'])
    assert rec['n_people'] == n_ref
    if n_ref:
        assert np.array_equal(rec['poses'][:n_ref], np.asarray(ref['poses'], dtype=np.float64))
        assert np.allclose(rec['scores'][:n_ref], ref['scores'], rtol=0, atol=1e-9)


@settings(max_examples=_FUZZ or 12, derandomize=not _FUZZ, deadline=None, database=None,
          suppress_health_check=[HealthCheck.function_scoped_fixture])
@given(seed=st.integers(0, 10 ** 6), n=st.integers(1, 14), fh=st.integers(16, 46), fw=st.integers(16, 46), up=st.sampled_from([3, 5, 7]),
       noise=st.sampled_from([0.0, 0.02, 0.05]), cap_pk=st.integers(1, 12), cap_sub=st.integers(1, 6), cap_ppl=st.integers(1, 4),
       cap_cand=st.sampled_from([0, 0, 16, 64]))
def test_hypothesis_capacity_growth_matches_oracle(native, seed, n, fh, fw, up, noise, cap_pk, cap_sub, cap_ppl, cap_cand):
    """Random crowds through contexts with random TINY capacities (peaks / joint, subsets, people per record, device-memory
    candidate store): whatever has to grow, however many rounds it takes, the result equals the oracle's."""
    heat, paf, _ = Fx.synthetic_maps(seed, n, fh, fw, 1.0, 0.9, noise=noise, height_range=(0.3, 0.8), drop_prob=0.2)
    mh, mw = fh * up, fw * up
    try:
        ref = P.postprocess_from_net_output(paf, heat, mh, mw)
    except IndexError:
        ref = None
    e = native.Engine(0, max_batch=1, max_h=64, max_w=64)
    e.set_capacities(peaks_per_joint=cap_pk, subsets=cap_sub, people=min(cap_ppl, cap_sub), candidates=cap_cand)      # people <= subsets (ABI)
    e.set_maps(paf[None], heat[None])
    e.postprocess(mh, mw, img_len=mw)
    rec = e.results()[0]
    try:
        if ref is None:
            assert rec['status'] & 8
            return
        assert rec['status'] == 0
        assert np.array_equal(e.peaks(0), ref['all_peaks'])
        subs = e.subsets(0)
        assert subs.shape == ref['subsets'].shape and np.array_equal(subs[:, :18], ref['subsets'][:, :18])
        n_ref = len(ref['subsets'])
        assert rec['n_people'] == n_ref
        if n_ref:
            assert np.array_equal(rec['poses'][:n_ref], np.asarray(ref['poses'], dtype=np.float64))
            assert np.allclose(rec['scores'][:n_ref], ref['scores'], rtol=0, atol=1e-9)
    finally:
        e.close()
