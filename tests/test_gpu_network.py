"""GPU (T1/T3): the 92-layer network through the C ABI against the torch-CPU fp32 restatement, and the end-to-end
PoseDetector on top of it."""
import numpy as np
import pytest

from conftest import forward_plan, pkg
from oracle import network_ref as N
from oracle import postprocess_ref as P

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def weights():
    return pkg('weights').synthetic_weights(0)


def _rel_err(a, b):
    return np.abs(a - b).max() / max(1.0, np.abs(b).max())


def test_network_forward_matches_torch_oracle(engine, weights):
    engine.set_weights(weights)
    rng = np.random.default_rng(0)
    imgs = rng.integers(0, 256, (2, 184, 248, 3), dtype=np.uint8)      # non-square, multiples of 8
    engine.forward_u8(imgs)
    paf, heat = engine.get_maps()
    x = np.concatenate([P.preprocess(im) for im in imgs])
    rpaf, rheat = N.forward(weights, x)
    assert paf.shape == rpaf.shape == (2, 38, 23, 31) and heat.shape == rheat.shape == (2, 19, 23, 31)
    # tolerance: fp32 accumulation-order differences through 6 stages (BASELINE tolerance for scores is 1e-4)
    assert _rel_err(paf, rpaf) < 1e-4, _rel_err(paf, rpaf)
    assert _rel_err(heat, rheat) < 1e-4, _rel_err(heat, rheat)


@pytest.mark.parametrize('shape', [(1, 64, 64), (2, 96, 128), (1, 40, 184)])
def test_network_bit_exact_vs_order_defined_oracle(engine, weights, shape):
    """The whole 92-layer forward (fused preprocess, pooling, the concat layout of the stage inputs, both branches) equals
    oracle/conv_fma_ref.py::forward_fma BIT FOR BIT: the kernels' summation order is defined (sequential fused multiply-add
    chain over chunk -> tap -> half -> k) and the plain-C oracle walks K the same way."""
    from oracle import conv_fma_ref as R
    engine.set_weights(weights)
    rng = np.random.default_rng(sum(shape))
    imgs = rng.integers(0, 256, shape + (3,), dtype=np.uint8)
    plan, _ = forward_plan(engine, lambda: engine.forward_u8(imgs))      # small launches are split over K: same plan on both sides
    paf, heat = engine.get_maps()
    x = np.concatenate([P.preprocess(im) for im in imgs])
    rpaf, rheat = R.forward_fma(weights, x, splitk=plan)
    assert np.array_equal(paf, rpaf), np.abs(paf - rpaf).max()
    assert np.array_equal(heat, rheat), np.abs(heat - rheat).max()


def test_end_to_end_identical_to_order_defined_oracle(native, weights, monkeypatch):
    """PoseDetector.__call__ on the GPU == (order-defined network oracle -> NumPy restatement of the reference post-process):
    identical pose arrays and scores, no tolerance anywhere.  The image is not at the network size, so the device resize is
    in the path; the network / map sizes are shrunk through entity.params to keep the scalar C oracle fast."""
    from oracle import conv_fma_ref as R
    from oracle import resize_ref
    PD, W, ent = pkg('pose_detector'), pkg('weights'), pkg('entity')
    monkeypatch.setitem(ent.params, 'inference_img_size', 64)
    monkeypatch.setitem(ent.params, 'heatmap_size', 56)
    det = PD.PoseDetector(weights=weights, device=0, max_size=(64, 96))
    rng = np.random.default_rng(11)
    img = rng.integers(0, 256, (50, 70, 3), dtype=np.uint8)
    in_w, in_h = det.compute_optimal_size(img, 64)
    map_w, map_h = det.compute_optimal_size(img, 56)
    small = resize_ref.resize_linear_u8(img, in_w, in_h)
    plan, _ = forward_plan(det.engine, lambda: det.engine.forward_u8(small[None]))      # the launch plan depends on shapes only
    paf, heat = R.forward_fma(weights, P.preprocess(small), splitk=plan)
    w2 = W.calibrate_head(weights, paf[0], heat[0])              # a head that produces people on this image
    det.engine.set_weights({k: w2[k] for k in ('Mconv7_stage6_L1', 'Mconv7_stage6_L2')})
    paf, heat = R.forward_fma(w2, P.preprocess(small), splitk=plan)
    ref = P.postprocess_from_net_output(paf[0], heat[0], map_h, map_w, orig_w=img.shape[1], orig_h=img.shape[0])
    poses, scores = det(img)
    assert len(ref['all_peaks']) > 0
    assert np.array_equal(np.asarray(poses, dtype=np.float64).reshape(-1, 18, 3), np.asarray(ref['poses']).reshape(-1, 18, 3))
    assert np.allclose(scores, ref['scores'], rtol=0, atol=1e-9)
    det.engine.close()


def test_forward_f32_seam_equals_u8_path(engine, weights):
    engine.set_weights(weights)
    rng = np.random.default_rng(1)
    imgs = rng.integers(0, 256, (1, 64, 96, 3), dtype=np.uint8)
    imgs.reshape(-1)[:256] = np.arange(256)          # every byte value goes through the fused preprocess
    engine.forward_u8(imgs)
    paf_a, heat_a = engine.get_maps()
    engine.forward_f32(P.preprocess(imgs[0]))
    paf_b, heat_b = engine.get_maps()
    assert np.array_equal(paf_a, paf_b) and np.array_equal(heat_a, heat_b)   # same kernels, same fp32 input bits


def test_full_size_368_single_image(engine, weights):
    engine.set_weights(weights)
    rng = np.random.default_rng(2)
    img = rng.integers(0, 256, (1, 368, 368, 3), dtype=np.uint8)
    engine.forward_u8(img)
    paf, heat = engine.get_maps()
    rpaf, rheat = N.forward(weights, P.preprocess(img[0]))
    assert _rel_err(paf, rpaf) < 1e-4 and _rel_err(heat, rheat) < 1e-4


def test_end_to_end_pose_detector(native, weights):
    """T3: PoseDetector.__call__ == oracle network o oracle post-process on the device's own maps."""
    W = pkg('weights')
    PD = pkg('pose_detector')
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (368, 368, 3), dtype=np.uint8)
    det = PD.PoseDetector(weights=weights, device=0, max_batch=2)
    # calibrate the synthetic head so that a realistic number of peaks survives (weights stay synthetic)
    det.engine.forward_u8(img[None])
    paf, heat = det.engine.get_maps()
    w2 = W.calibrate_head(weights, paf[0], heat[0])
    det.engine.set_weights({k: w2[k] for k in ('Mconv7_stage6_L1', 'Mconv7_stage6_L2')})
    poses, scores = det(img)
    paf, heat = det.engine.get_maps()
    # (a) the device's maps agree with the oracle network within tolerance
    rpaf, rheat = N.forward(w2, P.preprocess(img))
    assert _rel_err(paf, rpaf) < 1e-4 and _rel_err(heat, rheat) < 1e-4
    # (b) given identical maps, everything downstream is exact
    ref = P.postprocess_from_net_output(paf[0], heat[0], 320, 320, orig_w=368, orig_h=368)
    assert len(ref['all_peaks']) > 20, 'calibration should leave a workload'
    assert np.array_equal(det.engine.peaks(0), ref['all_peaks'])
    assert np.array_equal(np.asarray(poses), np.asarray(ref['poses']))
    assert np.allclose(scores, ref['scores'], rtol=0, atol=1e-9)
    # (c) batched entry == per-image calls: the same people; a batch of two and a single image use different split-K plans
    #     (defined, different summation trees), so the scores agree to summation-order noise; with split-K off, bit for bit
    img2 = rng.integers(0, 256, (368, 368, 3), dtype=np.uint8)
    (p1, s1), (p2, s2) = det.detect_batch([img, img2])
    assert np.array_equal(p1, poses) and np.allclose(s1, scores, rtol=0, atol=1e-5)
    q2, t2 = det(img2)
    assert np.array_equal(p2, q2) and np.allclose(s2, t2, rtol=0, atol=1e-5)
    det.engine.set_option('ksplit', 1)
    det.engine.set_option('conv_algo', 0)
    (p1, s1), (p2, s2) = det.detect_batch([img, img2])
    q1, t1 = det(img)
    q2, t2 = det(img2)
    assert np.array_equal(p1, q1) and np.array_equal(s1, t1) and np.array_equal(p2, q2) and np.array_equal(s2, t2)
    det.engine.close()


@pytest.mark.parametrize('src,dst', [((584, 584), (368, 368)), ((482, 642), (368, 496)), ((100, 37), (1000, 368)), ((37, 100), (368, 1000)),
                                     ((720, 1280), (368, 656))])
def test_gpu_resize_u8_bit_exact_vs_restatement(native, src, dst):
    """cv2.resize (INTER_LINEAR uint8, pose_detector.py:493) as a HIP kernel == the NumPy restatement, bit for bit."""
    from oracle import resize_ref as RR
    rng = np.random.default_rng(src[0])
    imgs = rng.integers(0, 256, (2, src[0], src[1], 3), dtype=np.uint8)
    eng = native.Engine(0, max_batch=2, max_h=dst[0], max_w=dst[1])
    out = eng.resize_u8(imgs, dst[0], dst[1])
    for b in range(2):
        assert np.array_equal(out[b], RR.resize_linear_u8(imgs[b], dst[1], dst[0]))
    eng.close()


def test_pose_detector_non_square_input_resized_on_device(native, weights):
    """Config 1 shape class: an image that is not at the network size goes through the device resize; the maps equal
    the oracle network on the restated-resize image."""
    from oracle import resize_ref as RR
    PD = pkg('pose_detector')
    rng = np.random.default_rng(8)
    img = rng.integers(0, 256, (292, 390, 3), dtype=np.uint8)
    det = PD.PoseDetector(weights=weights, device=0, max_size=(368, 496))
    iw, ih = det.compute_optimal_size(img, 368)
    assert (iw, ih) == (496, 368)
    try:
        det(img)
    except (RuntimeError, IndexError):
        pass          # uncalibrated random head: capacity overflow is fine here, only the maps are checked
    paf, heat = det.engine.get_maps()
    resized = RR.resize_linear_u8(img, iw, ih)
    assert np.array_equal(det.engine.get_resized(ih, iw)[0], resized)
    rpaf, rheat = N.forward(weights, P.preprocess(resized))
    assert _rel_err(paf, rpaf) < 1e-4 and _rel_err(heat, rheat) < 1e-4
    det.engine.close()


def test_cli_end_to_end_with_npz_weights(native, weights, tmp_path):
    """`pose_detector.py posenet weights.npz --img X --gpu 0` (reference :555-579) through the Chainer-NPZ reader."""
    W = pkg('weights')
    PD = pkg('pose_detector')
    rng = np.random.default_rng(10)
    img = rng.integers(0, 256, (368, 368, 3), dtype=np.uint8)
    eng = native.Engine(0, max_batch=1, max_h=368, max_w=368)
    eng.set_weights(weights)
    eng.forward_u8(img[None])
    paf, heat = eng.get_maps()
    eng.close()
    w2 = W.calibrate_head(weights, paf[0], heat[0])
    npz = str(tmp_path / 'coco_posenet.npz')
    W.save_npz(npz, w2)
    png_in, png_out = str(tmp_path / 'in.png'), str(tmp_path / 'result.png')
    PD.imwrite_bgr(png_in, img)
    assert PD.main(['posenet', npz, '--img', png_in, '--gpu', '0', '--out', png_out]) == 0
    drawn = PD.imread_bgr(png_out)
    det = PD.PoseDetector('posenet', npz, device=0)
    poses, scores = det(img)
    assert len(poses) > 0
    assert np.array_equal(drawn, PD.draw_person_pose(img, poses))
    det.engine.close()
