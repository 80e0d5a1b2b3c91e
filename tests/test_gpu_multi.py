"""GPU: batches of images of DIFFERENT sizes (include/pose_mi355x.h: pmx_detect_images / pmx_forward_u8_images / pmx_postprocess_images;
csrc/pmx_multi.hip).  The reference takes any image in any call and picks the network size per image (pose_detector.py:490-493, :57-73);
here a mixed batch is ONE launch per layer over all size classes.  Bars:
  * the maps of every image are bit-identical to the order-defined C twin (oracle/conv_fma_ref: plain Winograd arithmetic on every 3x3 /
    7x7 layer, conv1_1 direct + conv1_2 Winograd) and to a single-image call that runs the same plain kernels;
  * `PoseDetector.detect_batch` on mixed sizes (device cv2.resize per image, per-class post-process, rescale) returns, image by image and
    in the caller's order, exactly what a per-image call with those kernels returns -- also when the post-process capacities have to grow;
  * the uniform entry points refuse the maps of a mixed batch instead of misreading them."""
import numpy as np
import pytest

from conftest import forward_plan, pkg
from oracle import conv_fma_ref as R
from oracle import postprocess_ref as P

pytestmark = pytest.mark.gpu


def _plain(engine):
    """single-image calls on the kernels a mixed batch uses: the plain Winograd kernel on every eligible layer, conv1 as conv1_wino_kernel"""
    engine.set_option('conv_algo', 2)
    engine.set_option('conv1_wino', 2)


def _default(engine):
    engine.set_option('conv_algo', 1)
    engine.set_option('conv1_wino', 1)


def test_mixed_forward_maps_bit_exact_vs_twin_and_single_image_calls(native):
    weights = pkg('weights').synthetic_weights(0)
    eng = native.Engine(0, max_batch=8, max_h=128, max_w=128)
    eng.set_weights(weights)
    rng = np.random.default_rng(5)
    # three size classes, one of them twice (non-consecutive: its own segment), ragged against the 8 x 16 rectangles and 16 x 16 squares
    shapes = [(64, 96), (64, 96), (104, 72), (40, 136), (40, 136), (40, 136), (64, 96)]
    imgs = [rng.integers(0, 256, s + (3,), dtype=np.uint8) for s in shapes]
    plan, wino, _ = forward_plan(eng, lambda: eng.forward_u8_images(imgs), with_wino=True)
    assert {'conv1_2', 'conv2_1', 'conv2_2', 'conv3_4', 'conv4_4_CPM', 'conv5_3_CPM', 'Mconv1_stage2', 'Mconv5_stage6'} <= wino, wino
    assert not plan and not plan.wino_units and not plan.wino_tails        # no split-K, no unit mode: plain launches only
    got = [eng.image_maps(i) for i in range(len(imgs))]
    with pytest.raises(native.PmxError):                                    # the uniform accessors refuse a mixed layout
        eng.get_maps()
    with pytest.raises(native.PmxError):
        eng._fhw = (8, 12)
        eng.postprocess(64, 96, img_len=96)
    # (a) the C twin, image by image
    for i in (0, 2, 3, 6):
        rpaf, rheat = R.forward_fma(weights, P.preprocess(imgs[i]), splitk=None, wino=wino)
        assert np.array_equal(got[i][0], rpaf[0]) and np.array_equal(got[i][1], rheat[0]), (i, np.abs(got[i][0] - rpaf[0]).max())
    # (b) single-image calls on the same kernels
    _plain(eng)
    try:
        for i, im in enumerate(imgs):
            eng.forward_u8(im[None])
            paf, heat = eng.get_maps()
            assert np.array_equal(got[i][0], paf[0]) and np.array_equal(got[i][1], heat[0]), i
    finally:
        _default(eng)
    # (c) a uniform batch through the mixed entry == the same batch through the uniform entry on the plain kernels
    same = [imgs[3], imgs[4], imgs[5]]
    eng.forward_u8_images(same)
    a = [eng.image_maps(i) for i in range(3)]
    _plain(eng)
    try:
        eng.forward_u8(np.stack(same))
        paf, heat = eng.get_maps()
    finally:
        _default(eng)
    for i in range(3):
        assert np.array_equal(a[i][0], paf[i]) and np.array_equal(a[i][1], heat[i])
        assert np.array_equal(eng.image_maps(i)[0], paf[i])                 # (the per-image accessor on a uniform batch)
    eng.close()


def _calibrated(native):
    W = pkg('weights')
    weights = W.synthetic_weights(0)
    eng = native.Engine(0, max_batch=1, max_h=368, max_w=368)
    eng.set_weights(weights)
    eng.forward_u8(np.random.default_rng(1234).integers(0, 256, (1, 368, 368, 3), dtype=np.uint8))
    paf, heat = eng.get_maps()
    eng.close()
    return W.calibrate_head(weights, paf[0], heat[0])


def test_detect_batch_on_mixed_sizes_equals_per_image_calls(native):
    PD = pkg('pose_detector')
    weights = _calibrated(native)
    rng = np.random.default_rng(11)
    # originals of five sizes in no particular order: some already at their network size, most need the device cv2.resize
    sizes = [(368, 368), (240, 320), (368, 496), (240, 320), (480, 360), (368, 368), (300, 420), (240, 320), (368, 496)]
    imgs = [rng.integers(0, 256, s + (3,), dtype=np.uint8) for s in sizes]
    det = PD.PoseDetector(weights=weights, device=0, max_batch=4)           # (grows to 9 images / the pixel budget of the batch)
    res = det.detect_batch(imgs)
    assert len(res) == len(imgs) and det._cap[0] >= len(imgs)
    one = PD.PoseDetector(weights=weights, device=0, max_size=(368, 496))
    _plain(one.engine)
    people = 0
    for i, im in enumerate(imgs):
        poses, scores = one(im)
        assert np.asarray(res[i][0]).shape == np.asarray(poses).shape, i
        assert np.array_equal(np.asarray(res[i][0]), np.asarray(poses)) and np.array_equal(np.asarray(res[i][1]), np.asarray(scores)), i
        people += len(scores)
    assert people >= 10, 'the fixture should find people'
    # the default single-image path (unit-mode / split-K kernels) differs from it by fp32 rounding only
    _default(one.engine)
    n_same = 0
    for i, im in enumerate(imgs):
        poses, scores = one(im)
        if np.asarray(poses).shape == np.asarray(res[i][0]).shape and np.array_equal(np.asarray(poses), np.asarray(res[i][0])):
            n_same += 1
            assert np.allclose(scores, res[i][1], rtol=0, atol=1e-5)
    assert n_same >= len(imgs) - 1                                          # (a near-tie peak may flip on one frame)
    one.engine.close()
    # a second, differently composed batch through the same context (tables, segment buffers and caches are reused)
    res2 = det.detect_batch(imgs[::-1][:5])
    for k, i in enumerate(range(len(imgs) - 1, len(imgs) - 6, -1)):
        assert np.array_equal(np.asarray(res2[k][0]), np.asarray(res[i][0])) and np.array_equal(np.asarray(res2[k][1]), np.asarray(res[i][1]))
    # and a uniform batch afterwards still takes the uniform path
    uni = det.detect_batch([imgs[1], imgs[3], imgs[7]])
    assert len(uni) == 3 and det.engine._fhw is not None
    det.engine.close()


def test_mixed_batch_grows_capacities_like_a_uniform_one(native):
    """The post-process capacities (peaks per joint type, subsets, people) grow on demand: after an overflow the post-process of EVERY
    segment runs again on its slice of the new buffers -- same records as with capacities that were large enough from the start."""
    weights = _calibrated(native)
    rng = np.random.default_rng(3)
    imgs = [rng.integers(0, 256, s + (3,), dtype=np.uint8) for s in [(368, 368), (184, 248), (368, 368), (368, 496)]]
    net = [im.shape[:2] for im in imgs]
    mp = [(h * 320 // 368 // 8 * 8, w * 320 // 368 // 8 * 8) for h, w in net]
    big = native.Engine(0, max_batch=4, max_h=368, max_w=496)
    big.set_weights(weights)
    big.detect_images(imgs, net, mp)
    ref = big.results()
    assert int(ref['n_peaks'].sum()) > 100 and int(ref['n_people'].sum()) >= 4
    small = native.Engine(0, max_batch=4, max_h=368, max_w=496)
    small.set_weights(weights)
    small.set_capacities(peaks_per_joint=2, subsets=2, people=1)
    small.detect_images(imgs, net, mp)
    got = small.results()
    caps = small.capacities()
    assert caps['peaks_per_joint'] > 2 and caps['people'] > 1
    n = ref['n_people']
    assert np.array_equal(got['n_people'], n) and np.array_equal(got['n_peaks'], ref['n_peaks']) and np.array_equal(got['status'], ref['status'])
    for i in range(4):
        assert np.array_equal(got['poses'][i, :n[i]], ref['poses'][i, :n[i]]) and np.array_equal(got['scores'][i, :n[i]], ref['scores'][i, :n[i]])
        assert np.array_equal(small.peaks(i), big.peaks(i))
    small.close()
    big.close()


def test_mixed_batch_argument_errors(native):
    eng = native.Engine(0, max_batch=2, max_h=64, max_w=64)
    eng.set_weights(pkg('weights').synthetic_weights(0))
    rng = np.random.default_rng(0)
    im = lambda h, w: rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    with pytest.raises(native.PmxError) as e:
        eng.forward_u8_images([im(64, 64), im(32, 32), im(32, 32)])         # 3 images, capacity 2
    assert e.value.code == 5                                        # PMX_ERR_CAPACITY
    with pytest.raises(native.PmxError) as e:
        eng.forward_u8_images([im(64, 64), im(64, 72)])                     # pixel budget 2 x 64 x 64
    assert e.value.code == 5                                        # PMX_ERR_CAPACITY
    with pytest.raises(native.PmxError) as e:
        eng.forward_u8_images([im(60, 64)])                                 # not a multiple of 8
    assert e.value.code == 1
    eng.forward_u8_images([im(64, 64), im(32, 96)])                         # (32 x 96 fits the pixel budget although W > max_w)
    with pytest.raises(native.PmxError):
        eng.postprocess_images([(56, 56)])                                  # batch mismatch
    eng.postprocess_images([(56, 56), (24, 80)])
    assert len(eng.results()) == 2
    eng.forward_u8(im(64, 64)[None])
    with pytest.raises(native.PmxError):
        eng.postprocess_images([(56, 56)])                                  # the current maps are a uniform batch
    eng.close()


# ---- randomised: any list of sizes (deterministic example set by default; PMX_FUZZ=<n> draws n fresh random examples) ----------------------
import os
from hypothesis import given, settings, strategies as st, HealthCheck

_FUZZ = int(os.environ.get('PMX_FUZZ', '0'))


@pytest.fixture(scope='module')
def fuzz_engines(native):
    weights = _calibrated(native)
    mixed = native.Engine(0, max_batch=8, max_h=160, max_w=160)
    mixed.set_weights(weights)
    single = native.Engine(0, max_batch=1, max_h=160, max_w=160)
    single.set_weights(weights)
    _plain(single)
    yield mixed, single
    mixed.close()
    single.close()


@settings(max_examples=_FUZZ or 12, derandomize=not _FUZZ, deadline=None, database=None,
          suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])
@given(seed=st.integers(0, 10 ** 6), sizes=st.lists(st.tuples(st.integers(1, 20), st.integers(1, 20)), min_size=1, max_size=8))
def test_mixed_batch_fuzz_any_sizes(fuzz_engines, seed, sizes):
    """Random lists of 1 .. 8 images of 8 x 8 ... 160 x 160 pixels (maps down to 1 x 1, every kind of ragged edge against the 16 x 16
    squares and 8 x 16 rectangles, repeated and alternating sizes): maps, peaks and records of every image == the single-image call on
    the plain kernels, bit for bit."""
    mixed, single = fuzz_engines
    rng = np.random.default_rng(seed)
    imgs = [rng.integers(0, 256, (8 * h, 8 * w, 3), dtype=np.uint8) for h, w in sizes]
    net = [im.shape[:2] for im in imgs]
    mp = [(max(8, h * 320 // 368 // 8 * 8), max(8, w * 320 // 368 // 8 * 8)) for h, w in net]
    mixed.detect_images(imgs, net, mp)
    rec = mixed.results()
    maps = [mixed.image_maps(i) for i in range(len(imgs))]
    peaks = [mixed.peaks(i) for i in range(len(imgs))]
    for i, im in enumerate(imgs):
        single.detect_batch(im[None], mp[i][0], mp[i][1], img_len=mp[i][1], scale_xy=[[im.shape[1] / mp[i][1], im.shape[0] / mp[i][0]]])
        paf, heat = single.get_maps()
        assert np.array_equal(maps[i][0], paf[0]) and np.array_equal(maps[i][1], heat[0]), (i, sizes)
        assert np.array_equal(peaks[i], single.peaks(0)), (i, sizes)
        r1 = single.results()[0]
        n = int(r1['n_people'])
        assert int(rec[i]['n_people']) == n and int(rec[i]['n_peaks']) == int(r1['n_peaks']) and int(rec[i]['status']) == int(r1['status'])
        assert np.array_equal(rec[i]['poses'][:n], r1['poses'][:n]) and np.array_equal(rec[i]['scores'][:n], r1['scores'][:n])
