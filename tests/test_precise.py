"""detect_precise (reference pose_detector.py:433-482): CPU tests of the restated cv2 cubic resizes, GPU test of the
multi-scale path through the drop-in class."""
import numpy as np
import pytest

from conftest import pkg
from oracle import precise_ref as R
from oracle import postprocess_ref as P
from oracle import fixtures as Fx


def test_cubic_f32_product_equals_independent_oracle():
    PD = pkg('pose_detector')
    rng = np.random.default_rng(0)
    for (h, w, c, dh, dw) in [(6, 9, 5, 48, 72), (23, 31, 3, 17, 40), (10, 10, 2, 10, 25), (46, 46, 4, 368, 368)]:
        a = rng.standard_normal((h, w, c)).astype('f')
        assert np.array_equal(PD.resize_cubic_f32(a, dw, dh), R.resize_cubic_f32_ref(a, dw, dh))


def test_cubic_u8_product_equals_independent_oracle_and_is_sane():
    PD = pkg('pose_detector')
    rng = np.random.default_rng(1)
    for (h, w, dh, dw) in [(20, 30, 31, 45), (48, 64, 24, 32), (37, 29, 100, 64)]:
        a = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        assert np.array_equal(PD.resize_cubic_u8(a, dw, dh), R.resize_cubic_u8_ref(a, dw, dh))
    flat = np.full((9, 7, 3), 200, np.uint8)
    assert np.all(PD.resize_cubic_u8(flat, 20, 33) == 200)
    # against torch's bicubic (same A = -0.75, half-pixel centres): float path to rounding, uint8 path to +-1
    import torch
    a = rng.standard_normal((10, 12, 3)).astype('f')
    t = torch.nn.functional.interpolate(torch.from_numpy(a.transpose(2, 0, 1))[None], size=(80, 96), mode='bicubic',
                                        align_corners=False)[0].numpy().transpose(1, 2, 0)
    assert np.abs(PD.resize_cubic_f32(a, 96, 80) - t).max() < 1e-5


def test_pad_image_matches_reference_semantics():
    PD = pkg('pose_detector')
    det = PD.PoseDetector.__new__(PD.PoseDetector)
    img = np.arange(5 * 11 * 3, dtype=np.uint8).reshape(5, 11, 3)
    p, pad = det.pad_image(img, 8, (104, 117, 123))
    q, pad2 = R.pad_image(img, 8, (104, 117, 123))
    assert pad == pad2 == [3, 5] and np.array_equal(p, q)
    assert p.shape == (8, 16, 3) and tuple(p[7, 15]) == (104, 117, 123)


@pytest.mark.gpu
def test_detect_precise_with_model_seam(native):
    """Synthetic skeleton maps injected through the reference's `model=` seam at every scale; the averaged maps and
    the final poses must equal the oracle's (same restated cubic; post-process exact on identical maps)."""
    PD = pkg('pose_detector')
    H, W = 96, 128                     # original image
    rng = np.random.default_rng(4)
    poses = Fx.random_poses(rng, 3, H, W, height_range=(0.5, 0.8), drop_prob=0.1)

    def model(x):
        # network-output-shaped maps for whatever padded input size the scale loop produced
        h8, w8 = x.shape[2] // 8, x.shape[3] // 8
        sp = poses.copy()
        sp[:, :, 0] *= w8 / W
        sp[:, :, 1] *= h8 / H
        heat = Fx.render_heatmaps((h8, w8), sp, max(0.6, 0.02 * h8))
        paf = Fx.render_pafs((h8, w8), sp, max(0.6, 0.015 * h8))
        return [paf[None]], [heat[None]]
    img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    det = PD.PoseDetector(model=model, device=0, precise=True)
    got_poses, got_scores = det(img)
    ref_paf, ref_heat, sizes = R.averaged_maps(lambda x: tuple(m[-1] for m in model(x)), img)
    assert sizes[0][0] % 8 == 0 and len(sizes) == 4
    assert np.array_equal(det.pafs, ref_paf) and np.array_equal(det.heatmaps, ref_heat)
    ref = R.detect_precise_from_maps(ref_paf, ref_heat)
    assert np.array_equal(det.all_peaks, ref['all_peaks'])
    assert len(ref['subsets']) >= 2, 'fixture should find people'
    assert np.array_equal(np.asarray(got_poses), np.asarray(ref['poses']))
    assert np.allclose(got_scores, ref['scores'], rtol=0, atol=1e-9)
    det.engine.close()


@pytest.mark.gpu
def test_detect_precise_native_network(native):
    """The four forward passes run on the GPU kernels (sizes 0.5x..2x); averaged maps vs the oracle network within
    tolerance, post-process exact on the device's own averaged maps."""
    PD = pkg('pose_detector')
    W_ = pkg('weights')
    from oracle import network_ref as N
    weights = W_.synthetic_weights(0)
    rng = np.random.default_rng(6)
    img = rng.integers(0, 256, (120, 152, 3), dtype=np.uint8)
    det = PD.PoseDetector(weights=weights, device=0, precise=True, max_size=(368, 472))
    # calibrate the synthetic head on the scale-1 input so that a sane number of peaks survives at full resolution
    cal = PD.resize_cubic_u8(img, 467, 368)
    cal, _ = det.pad_image(cal, 8, (104, 117, 123))
    det.engine.forward_u8(cal[None])
    paf0, heat0 = det.engine.get_maps()
    weights = W_.calibrate_head(weights, paf0[0], heat0[0], heat_s=0.1, heat_t=-0.2)
    det._weights = weights
    det.engine.set_weights({k: weights[k] for k in ('Mconv7_stage6_L1', 'Mconv7_stage6_L2')})
    poses, scores = det(img)
    ref_paf, ref_heat, sizes = R.averaged_maps(lambda x: N.forward(weights, x), img)
    assert sizes[-1][0] >= 2 * 368
    scale = max(1.0, np.abs(ref_heat).max(), np.abs(ref_paf).max())
    assert np.abs(det.pafs - ref_paf).max() < 2e-4 * scale
    assert np.abs(det.heatmaps - ref_heat).max() < 2e-4 * scale
    try:
        ref = R.detect_precise_from_maps(det.pafs, det.heatmaps)
    except IndexError:
        pytest.skip('reference raises IndexError on this random field')
    assert np.array_equal(det.all_peaks, ref['all_peaks'])
    assert np.array_equal(np.asarray(poses), np.asarray(ref['poses']))
    det.engine.close()


@pytest.mark.gpu
@pytest.mark.parametrize('rows', [1, 0])
@pytest.mark.parametrize('shape', [(120, 152), (97, 141), (368, 368)])
def test_device_cubic_path_equals_host_restatement(native, shape, rows):
    """pmx_precise_* (cubic uint8 / float32 resizes, crop, accumulation, average on the device) is bit-identical to the
    host restatement of the same steps fed by the same network kernels (the `model=` seam wrapping the engine) -- with the
    float32 resizes in their separable LDS form (rows = 1, the default) and one thread per element (rows = 0)."""
    PD = pkg('pose_detector')
    W_ = pkg('weights')
    weights = W_.synthetic_weights(0)
    rng = np.random.default_rng(shape[0])
    img = rng.integers(0, 256, shape + (3,), dtype=np.uint8)
    dev = PD.PoseDetector(weights=weights, device=0, precise=True, max_size=(368, 472))
    dev.engine.set_option('cubic_rows', rows)
    dev.engine.set_option('precise_plain', 0)      # the seam below runs single forwards with the default kernel selection: same kernels on both sides
    res_dev = None
    try:
        res_dev = dev(img)
    except (IndexError, RuntimeError):        # random-weight maps may overflow the peak capacity; maps are still compared
        pass
    eng = dev.engine

    def model(x):
        eng.forward_f32(x)
        paf, heat = eng.get_maps()
        return [paf], [heat]
    host = PD.PoseDetector(model=model, device=0, precise=True)
    res_host = None
    try:
        res_host = host(img)
    except (IndexError, RuntimeError):        # random-weight maps may overflow the peak capacity; maps are still compared
        pass
    assert np.array_equal(dev.pafs, host.pafs)
    assert np.array_equal(dev.heatmaps, host.heatmaps)
    assert (res_dev is None) == (res_host is None)
    if res_dev is not None:
        assert np.array_equal(np.asarray(res_dev[0]), np.asarray(res_host[0]))
        assert np.array_equal(np.asarray(res_dev[1]), np.asarray(res_host[1]))
    host.engine.close()
    dev.engine.set_option('cubic_rows', 1)
    dev.engine.close()


@pytest.mark.gpu
def test_detect_precise_crowd_config5(native):
    """BASELINE config 5: multi-scale (0.5/1.0/1.5/2.0) crowd image -- 22 synthetic people at 482 x 642 (dinner.png's
    size), full PAF grouping stress at the ORIGINAL resolution.  Maps enter through the `model=` seam at every scale;
    poses must equal the oracle's exactly (identical restated cubic resizes -> identical averaged maps)."""
    PD = pkg('pose_detector')
    H, W = 482, 642
    rng = np.random.default_rng(12)
    poses = Fx.random_poses(rng, 22, H, W, height_range=(0.25, 0.5), drop_prob=0.1)

    def model(x):
        h8, w8 = x.shape[2] // 8, x.shape[3] // 8
        sp = poses.copy()
        sp[:, :, 0] *= w8 / W
        sp[:, :, 1] *= h8 / H
        heat = Fx.render_heatmaps((h8, w8), sp, max(0.7, 0.012 * h8))
        paf = Fx.render_pafs((h8, w8), sp, max(0.7, 0.01 * h8))
        return [paf[None]], [heat[None]]
    img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    det = PD.PoseDetector(model=model, device=0, precise=True)
    got_poses, got_scores = det(img)
    ref_paf, ref_heat, sizes = R.averaged_maps(lambda x: tuple(m[-1] for m in model(x)), img)
    assert np.array_equal(det.pafs, ref_paf) and np.array_equal(det.heatmaps, ref_heat)
    ref = R.detect_precise_from_maps(ref_paf, ref_heat)
    assert len(ref['subsets']) >= 15, 'crowd fixture should recover most of the 22 people (got %d)' % len(ref['subsets'])
    assert np.array_equal(det.all_peaks, ref['all_peaks'])
    assert np.array_equal(np.asarray(got_poses), np.asarray(ref['poses']))
    assert np.allclose(got_scores, ref['scores'], rtol=0, atol=1e-9)
    # recovered people sit on the ground-truth skeletons: keypoints within tolerance of the rendered joints (the 1/8
    # resolution rendering + cubic up-sampling shifts peaks by ~5 px at this size; fragments of mixed people are rare)
    gp = np.asarray(got_poses)
    ds = []
    for person in gp:
        vis = person[:, 2] > 0
        ds.append(min(np.mean(np.hypot(person[vis, 0] - q[vis, 0], person[vis, 1] - q[vis, 1])) for q in poses))
    assert np.median(ds) < 8.0, np.median(ds)
    det.engine.close()


@pytest.mark.gpu
def test_detect_precise_batch_equals_one_image_per_call(native):
    """PoseDetector.detect_precise_batch (every inference scale of n same-size images as ONE network batch, pmx_precise_*_batch) against
    detect_precise image by image: the averaged full-resolution maps agree to the network's kernel-choice-by-launch-size rounding
    (<= 2e-5 of the map scale: the resizes and the accumulation are the same device code per image), the peak sets differ at most by
    near-ties, and a batch of ONE is the single call bit for bit."""
    PD = pkg('pose_detector')
    W_ = pkg('weights')
    weights = W_.synthetic_weights(0)
    rng = np.random.default_rng(21)
    imgs = [rng.integers(0, 256, (120, 152, 3), dtype=np.uint8) for _ in range(3)]
    det = PD.PoseDetector(weights=weights, device=0, precise=True, max_size=(368, 472))
    cal = PD.resize_cubic_u8(imgs[0], 467, 368)
    cal, _ = det.pad_image(cal, 8, (104, 117, 123))
    det.engine.forward_u8(cal[None])
    paf0, heat0 = det.engine.get_maps()
    weights = W_.calibrate_head(weights, paf0[0], heat0[0], heat_s=0.1, heat_t=-0.2)
    det._weights = weights
    det.engine.set_weights({k: weights[k] for k in ('Mconv7_stage6_L1', 'Mconv7_stage6_L2')})
    single = []
    for im in imgs:
        try:
            r = det(im)
        except IndexError:
            r = None
        single.append((r, det.pafs.copy(), det.heatmaps.copy(), det.all_peaks.copy()))
    # a batch of one IS the single call
    try:
        r1 = det.detect_precise_batch([imgs[1]], fetch_maps=True)[0]
    except IndexError:
        r1 = None
    assert np.array_equal(det.pafs[0], single[1][1]) and np.array_equal(det.heatmaps[0], single[1][2])
    assert (r1 is None) == (single[1][0] is None)
    if r1 is not None:
        assert np.array_equal(np.asarray(r1[0]), np.asarray(single[1][0][0])) and np.array_equal(np.asarray(r1[1]), np.asarray(single[1][0][1]))
    try:
        rb = det.detect_precise_batch(imgs, fetch_maps=True)
    except IndexError:
        rb = None
    assert det.pafs.shape == (3, 38, 120, 152) and det.heatmaps.shape == (3, 19, 120, 152)
    for i in range(3):
        scale = max(1.0, float(np.abs(single[i][1]).max()), float(np.abs(single[i][2]).max()))
        assert float(np.abs(det.pafs[i] - single[i][1]).max()) <= 2e-5 * scale
        assert float(np.abs(det.heatmaps[i] - single[i][2]).max()) <= 2e-5 * scale
        pb, ps = det.engine.peaks(i), single[i][3]
        sb = {(int(r[0]), int(r[1]), int(r[2])) for r in pb}
        ss = {(int(r[0]), int(r[1]), int(r[2])) for r in ps}
        assert len(sb ^ ss) <= max(2, len(ss) // 100), (i, len(sb), len(ss), len(sb ^ ss))
        if rb is not None and single[i][0] is not None and sb == ss:
            assert len(rb[i][0]) == len(single[i][0][0])
    det.engine.close()


@pytest.mark.gpu
def test_precise_scales_in_flight_same_bits(native):
    """detect_precise runs its four inference scales concurrently, one per lane (own stream and working set), and adds their parts in
    scale order at the end (option precise_lanes, default 4): the averaged maps, peaks and poses are those of the scales run one after
    the other on one stream (precise_lanes = 1) -- bit for bit, twice in a row (the second call reuses parts and lanes)."""
    PD = pkg('pose_detector')
    W_ = pkg('weights')
    weights = W_.synthetic_weights(0)
    rng = np.random.default_rng(77)
    imgs = [rng.integers(0, 256, (120, 152, 3), dtype=np.uint8), rng.integers(0, 256, (120, 152, 3), dtype=np.uint8)]
    det = PD.PoseDetector(weights=weights, device=0, precise=True, max_size=(368, 472))
    got = {}
    for lanes in (4, 1, 3):
        det.engine.set_option('precise_lanes', lanes)
        det.engine.set_option('precise_plain', 1)            # (-1, the default, means "plain kernels with all four lanes in use": the other runs would pick other kernels)
        for rep, img in enumerate(imgs + imgs[:1]):
            try:
                det._detect_precise_device(img, fetch_maps=True)
            except (IndexError, RuntimeError):
                pass
            got[(lanes, rep)] = (det.pafs.copy(), det.heatmaps.copy())
    det.engine.set_option('precise_lanes', 4)
    det.engine.set_option('precise_plain', -1)
    det.engine.close()
    for rep in range(3):
        for lanes in (1, 3):
            assert np.array_equal(got[(4, rep)][0], got[(lanes, rep)][0]) and np.array_equal(got[(4, rep)][1], got[(lanes, rep)][1])
    assert np.array_equal(got[(4, 0)][0], got[(4, 2)][0]) and not np.array_equal(got[(4, 0)][0], got[(4, 1)][0])


@pytest.mark.gpu
def test_precise_table_cache_trims_only_between_sequences(native):
    """A context fed ever new image sizes (detect_precise over a data set) starts its cubic-table cache over -- but only inside
    pmx_precise_begin*, behind a device synchronisation, never while a sequence holds table pointers (the round-5 code evicted inside
    cubic_table(), i.e. between the six lookups of ONE scale, with lanes still in flight: freed / recycled tables under running
    kernels).  14 distinct sizes through one context whose cap is lowered so that the cache is trimmed several times; every result is
    bit-identical to that of a fresh context that never trims."""
    PD = pkg('pose_detector')
    W_ = pkg('weights')
    weights = W_.synthetic_weights(0)
    rng = np.random.default_rng(91)
    sizes = [(48 + 4 * i, 64 + 4 * i + 4 * ((3 * i) % 7)) for i in range(14)]          # the 2x scale is at most 736 x 1264
    assert len(set(sizes)) == 14
    imgs = [rng.integers(0, 256, s + (3,), dtype=np.uint8) for s in sizes]
    det = PD.PoseDetector(weights=weights, device=0, precise=True, max_size=(744, 1280))      # (no re-created context on the way: it would forget the option)
    det.engine.set_option('precise_table_cap', 30)          # one sequence adds up to 24 tables: trimmed at (almost) every second begin
    got = []
    for img in imgs:
        try:
            det._detect_precise_device(img, fetch_maps=True)
        except (IndexError, RuntimeError):                  # random-weight maps may overflow a capacity; the maps are what is compared
            pass
        got.append((det.pafs.copy(), det.heatmaps.copy()))
    cached, trims = det.engine.precise_table_stats()
    assert trims >= 4 and cached <= 30 + 24, (cached, trims)
    det.engine.close()
    for i in (0, 5, 9, 13):
        fresh = PD.PoseDetector(weights=weights, device=0, precise=True, max_size=(744, 1280))
        try:
            fresh._detect_precise_device(imgs[i], fetch_maps=True)
        except (IndexError, RuntimeError):
            pass
        assert fresh.engine.precise_table_stats()[1] == 0
        assert np.array_equal(fresh.pafs, got[i][0]) and np.array_equal(fresh.heatmaps, got[i][1]), i
        fresh.engine.close()


@pytest.mark.gpu
def test_precise_enqueue_order_does_not_change_bits(native):
    """pmx_precise_add_scale_at: the scale's position in the reference's loop (slot) is given explicitly, the parts are summed in slot order
    (:463,467) -- so the averaged maps do not depend on the order the scales are ENQUEUED in (PoseDetector enqueues the largest first: its
    chain is the critical path).  In order without slots == in order with slots == reversed == what PoseDetector does; a slot used twice
    and a gap at finish are refused."""
    import math
    PD = pkg('pose_detector')
    E = pkg('entity')
    weights = pkg('weights').synthetic_weights(0)
    rng = np.random.default_rng(8)
    img = rng.integers(0, 256, (120, 152, 3), dtype=np.uint8)
    det = PD.PoseDetector(weights=weights, device=0, precise=True, max_size=(368, 472))
    try:
        det._detect_precise_device(img, fetch_maps=True)
    except (IndexError, RuntimeError):
        pass
    ref = (det.pafs.copy(), det.heatmaps.copy())
    eng = det.engine
    sizes = []
    for scale in E.params['inference_scales']:
        m = scale * E.params['inference_img_size'] / min(img.shape[:2])
        sizes.append((math.ceil(img.shape[0] * m), math.ceil(img.shape[1] * m)))
    for order, with_slot in (([0, 1, 2, 3], False), ([0, 1, 2, 3], True), ([3, 2, 1, 0], True), ([2, 0, 3, 1], True)):
        eng.precise_begin(img.shape[0], img.shape[1], 1)
        for k in order:
            eng.precise_add_scale(img, sizes[k][0], sizes[k][1], slot=k if with_slot else None)
        eng.precise_finish()
        paf, heat = eng.get_maps()
        assert np.array_equal(paf[0], ref[0]) and np.array_equal(heat[0], ref[1]), (order, with_slot)
    eng.precise_begin(img.shape[0], img.shape[1], 1)
    eng.precise_add_scale(img, sizes[1][0], sizes[1][1], slot=1)
    with pytest.raises(native.PmxError):
        eng.precise_add_scale(img, sizes[1][0], sizes[1][1], slot=1)          # the slot is taken
    with pytest.raises(native.PmxError):
        eng.precise_finish()                                                   # slot 0 is missing
    eng.close()
