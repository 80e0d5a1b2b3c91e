"""GPU: the Winograd F(2x2, 3x3) fp32 kernel (csrc/conv_wino.hip::conv_wino_kernel, option "conv_algo": 1 = the 3x3 / 7x7
layers of launches that fill the chip, 2 = every eligible layer) -- the replacement for L.Convolution2D on those layers
(models/CocoPoseNet.py:28-129).  Bars: bit-identical to its plain-C twin (oracle/conv_fma_ref.c::conv_wino_ref: transforms,
plane-wise FMA chains and output transforms in the kernel's order); within fp32 rounding of the float64 convolution; the whole
network and the reference's end-to-end goldens (identical peak indices / poses, scores to 1e-4) also hold with it."""
import numpy as np
import pytest

from conftest import forward_plan, pkg
from oracle import conv_fma_ref as R
from oracle import network_ref as N
from oracle import postprocess_ref as P
from test_reference_network import load_e2e

pytestmark = pytest.mark.gpu
TOL = 2e-5


def _run(engine, x, w, b, relu, pool, algo):
    engine.set_option('conv_algo', algo)
    try:
        return engine.conv2d(x, w, b, relu=relu, pool=pool)
    finally:
        engine.set_option('conv_algo', 0)


def _data(seed, B, cin, H, W, cout, k):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((B, cin, H, W)).astype('f')
    w = (rng.standard_normal((cout, cin, k, k)) / np.sqrt(cin * k * k)).astype('f')
    return x, w, rng.standard_normal(cout).astype('f')


@pytest.mark.parametrize('B,cin,H,W,cout,k,relu,pool', [
    (2, 32, 30, 34, 128, 3, True, False),       # several 8x16 tiles, ragged right / bottom edges
    (3, 64, 22, 18, 128, 3, False, True),       # fused 2x2 max-pool
    (1, 96, 17, 33, 256, 3, True, False),       # odd H and W (half-used Winograd tiles), two 128-channel blocks
    (2, 32, 9, 15, 130, 3, True, False),        # cout not a multiple of 128 (padded to 256)
    (1, 64, 46, 46, 128, 3, True, False),       # the feature-map size of the network
    (2, 32, 19, 21, 128, 7, True, False),       # 7x7: four 3x3 sub-kernels + row 6 / column 6 (1-D) + tap (6, 6)
    (1, 64, 46, 46, 128, 7, True, False),
    (1, 96, 9, 40, 100, 7, False, False),
    (1, 32, 3, 5, 128, 7, True, False),         # image smaller than the kernel: every window crosses the border
    (1, 32, 2, 2, 128, 3, True, True)])
def test_winograd_conv_bit_exact_vs_c_twin(engine, B, cin, H, W, cout, k, relu, pool):
    x, w, b = _data(B * 1000 + cin + H + k, B, cin, H, W, cout, k)
    y = _run(engine, x, w, b, relu, pool, 2)
    yd = _run(engine, x, w, b, relu, pool, 0)
    ref = R.conv_wino(x, w, b, relu, pool)
    assert y.shape == ref.shape and np.isfinite(y).all()
    assert np.array_equal(y, ref), (np.abs(y - ref).max(), int((y != ref).sum()))
    assert not np.array_equal(y, yd), 'the Winograd kernel did not run'
    t = N.conv2d_ref(x, w, b, relu=relu, pool=pool)
    assert np.abs(y - t).max() <= TOL * max(1.0, np.abs(t).max())


def test_winograd_is_not_used_where_it_does_not_apply(engine):
    """cin not a multiple of 32, 1x1 layers, the bf16x3 mode and forced variants keep their kernels (bit-identical to conv_algo 0)."""
    for (cin, cout, k) in [(48, 128, 3), (64, 128, 1), (3, 64, 3)]:
        x, w, b = _data(cin + k, 1, cin, 12, 20, cout, k)
        assert np.array_equal(_run(engine, x, w, b, True, False, 2), _run(engine, x, w, b, True, False, 0)), (cin, cout, k)


def test_winograd_launch_selection_by_round_fill(engine):
    """conv_algo = 1 takes a layer when its blocks fill at least half of the CU rounds they need (one equal block per CU at a time:
    a round costs the same however full it is); the result is the respective twin's either way."""
    ncu = 256
    engine.set_option('ksplit', 1)                                          # the direct launches unsplit: one twin call each
    for B, wino in [(1, False), (5, False), (7, False), (8, True), (15, True), (32, True)]:
        blocks = 17 * B                                                     # 46x46 map, 128 output channels: 529 tiles in runs of 32
        rounds = -(-blocks // ncu)
        assert (blocks * 100 >= 50 * rounds * ncu) == wino, (B, blocks)     # the rule, restated
        x, w, b = _data(B, B, 32, 46, 46, 128, 3)
        y = _run(engine, x, w, b, True, False, 1)
        ref = R.conv_wino(x, w, b, True, False) if wino else R.conv_fma(x, w, b, True, False)
        assert np.array_equal(y, ref), (B, blocks, wino)
    engine.set_option('ksplit', 0)


@pytest.mark.parametrize('B,cin,H,W,cout,k,relu,pool', [
    (2, 64, 46, 46, 128, 7, True, False),          # 46 x 46: 16 full runs of 32 tiles + 17 tiles
    (1, 192, 46, 46, 128, 7, True, False),         # Mconv1-shaped (6 chunks)
    (3, 32, 20, 46, 130, 7, False, False),         # 230 tiles = 7 runs + 6 tiles; cout padded to 256
    (1, 96, 45, 46, 256, 3, True, False),          # odd H: the last tile row is half used
    (2, 64, 46, 46, 128, 3, True, False),
    (1, 32, 2, 46, 128, 7, True, False),           # one part-filled run, every window crosses the border
    (1, 64, 3, 46, 128, 3, False, False),
    (1, 64, 92, 46, 128, 7, True, False),          # taller than wide: 34 runs
    (1, 64, 30, 92, 128, 3, True, False),          # two slabs of 46 columns: halo columns come from the neighbour slab
    (2, 32, 24, 92, 128, 3, True, True),           # ... with the fused 2x2 max-pool (conv3_4-shaped)
    (1, 64, 14, 184, 128, 3, False, True),         # four slabs (conv2_2-shaped)
    (1, 32, 9, 138, 128, 7, True, False)])         # three slabs, 7x7, odd H
def test_winograd_run_geometry_bit_exact_vs_c_twin(engine, B, cin, H, W, cout, k, relu, pool):
    """Maps whose width is a multiple of 46 take the run geometry of the Winograd kernel (vertical slabs of 46 columns, blocks of 32
    consecutive tiles of a slab in row-major order instead of 8 x 16 pixel rectangles; conv_wino.hip GEOM 1): same arithmetic per
    tile, so the same bits as the rectangles and as the twin."""
    x, w, b = _data(7 * B + cin + H + k + W, B, cin, H, W, cout, k)
    y = _run(engine, x, w, b, relu, pool, 2)
    engine.set_option('wino_geom', 0)
    try:
        y_rect = _run(engine, x, w, b, relu, pool, 2)
    finally:
        engine.set_option('wino_geom', -1)
    ref = R.conv_wino(x, w, b, relu, pool)
    assert y.shape == ref.shape and np.isfinite(y).all()
    assert np.array_equal(y, ref), (np.abs(y - ref).max(), int((y != ref).sum()))
    assert np.array_equal(y_rect, ref)
    assert not np.array_equal(y, _run(engine, x, w, b, relu, pool, 0)), 'the Winograd kernel did not run'


@pytest.mark.parametrize('B,cin,H,W,cout,k,relu,pool', [
    (2, 128, 46, 46, 128, 7, True, False),         # the 7x7 layers of stages 2-6: units of 1 chunk -> 4 + row 6 + column 6 + tap (6, 6) = 7 slabs
    (1, 192, 46, 46, 128, 7, True, False),         # Mconv1: units of 2 chunks -> 6 slabs
    (1, 64, 46, 46, 256, 7, False, False),
    (2, 256, 46, 46, 128, 3, True, False),         # conv4_4: 8 chunk units
    (1, 512, 46, 46, 512, 3, True, False),         # conv4_2: 16 chunks in units of 2
    (1, 96, 45, 46, 132, 7, True, False),          # odd H, cout padded to 256
    (1, 64, 10, 46, 128, 3, True, False),          # 115 tiles = 3 runs + 19 tiles
    (2, 64, 12, 92, 128, 3, True, False),          # two slabs, 138 tiles each = 4 runs + 10 tiles
    (1, 128, 20, 184, 128, 3, True, True)])        # four slabs, pooled: the combine kernel pools the tail tiles
def test_winograd_run_tail_in_unit_mode_bit_exact_vs_c_twin(engine, B, cin, H, W, cout, k, relu, pool):
    """Run geometry with the part-filled last block of every image in unit mode (option wino_tail = 1; by the cost model at batch
    32): the full runs as one plain launch, the tail as K units writing compact slabs + conv_wino_tail_reduce_kernel == the twin with
    `unit_from` = first tile of that block, bit for bit; the tiles in front of it keep the plain chain."""
    x, w, b = _data(11 * B + cin + H + k + W, B, cin, H, W, cout, k)
    nch = (cin + 31) // 32
    g = -(-nch // (8 - (3 if k == 7 else 0)))
    engine.set_option('wino_tail', 1)
    try:
        y = _run(engine, x, w, b, relu, pool, 2)
    finally:
        engine.set_option('wino_tail', -1)
    uf = R.wino_run_unit_from(H, W)
    ntiles = 23 * ((H + 1) // 2)
    assert 0 < uf < ntiles
    ref = R.conv_wino(x, w, b, relu, pool, unit_g=g, unit_from=uf)
    plain = R.conv_wino(x, w, b, relu, pool)
    assert np.array_equal(y, ref), (np.abs(y - ref).max(), int((y != ref).sum()))
    assert not np.array_equal(y, plain), 'the tail did not run in unit mode'
    if not pool:
        m = np.zeros((H + 1, W), bool)                    # pixels of the tiles in front of the tails: the plain chain
        for sl in range(W // 46):
            for t in range(uf):
                m[2 * (t // 23):2 * (t // 23) + 2, 46 * sl + 2 * (t % 23):46 * sl + 2 * (t % 23) + 2] = True
        assert np.array_equal(y[..., m[:H]], plain[..., m[:H]])
    t = N.conv2d_ref(x, w, b, relu=relu, pool=pool)
    assert np.abs(y - t).max() <= TOL * max(1.0, np.abs(t).max())


@pytest.mark.parametrize('B,cin,H,cout,k,relu', [
    (2, 128, 46, 128, 7, True),        # 34 tail tiles: a block of 17 + 15 tiles, one of 2
    (5, 192, 46, 128, 7, True),        # 85 tiles: blocks 17 + 15 | 2 + 17 + 13 (three images in one block) | 4 + 17 (part-filled)
    (3, 64, 38, 256, 7, False),        # 19 tile rows: 21 tail tiles per image; two 128-channel blocks
    (6, 64, 45, 128, 7, True),         # odd H (the last tile row is half outside the map)
    (4, 256, 46, 128, 3, True),        # conv4_4
    (7, 64, 24, 132, 3, True)])        # 20 tail tiles per image, cout padded to 256
def test_winograd_merged_tails_bit_exact_vs_c_twin(engine, B, cin, H, cout, k, relu):
    """Batches: the part-filled last blocks of all images run as ONE stream of tiles, 32 per block (conv_wino_kernel<KS, 0, 1, 3>:
    up to three images' segments side by side in a block) instead of one part-filled block per image.  Same units, same chains: the
    output equals the per-image form (option wino_tail_merge = 0) and the twin bit for bit."""
    W = 46
    x, w, b = _data(13 * B + cin + H + k, B, cin, H, W, cout, k)
    nch = (cin + 31) // 32
    g = -(-nch // (8 - (3 if k == 7 else 0)))
    engine.set_option('wino_tail', 1)
    try:
        y = _run(engine, x, w, b, relu, False, 2)
        engine.set_option('wino_tail_merge', 0)
        y_img = _run(engine, x, w, b, relu, False, 2)
    finally:
        engine.set_option('wino_tail_merge', 1)
        engine.set_option('wino_tail', -1)
    uf = R.wino_run_unit_from(H, W)
    ntiles = 23 * ((H + 1) // 2)
    assert 16 <= ntiles - uf <= 23 and uf % 23 + (ntiles - uf) == 23, 'not a mergeable tail: the test would compare the per-image form with itself'
    ref = R.conv_wino(x, w, b, relu, False, unit_g=g, unit_from=uf)
    assert np.array_equal(y_img, ref), (np.abs(y_img - ref).max(), int((y_img != ref).sum()))
    assert np.array_equal(y, ref), (np.abs(y - ref).max(), int((y != ref).sum()), np.argwhere(y != ref)[:8].tolist())


def test_two_images_368_merged_tails_bit_exact(native):
    """Two 368x368 images through the whole network, plain Winograd kernel forced on every eligible layer, tails in unit mode: the
    46x46 layers take the merged-tail launch (profile labels ".../t<g>m") and forward_fma with that plan reproduces both images' maps
    bit for bit."""
    weights = pkg('weights').synthetic_weights(0)
    eng = native.Engine(0, max_batch=2, max_h=368, max_w=368)
    eng.set_weights(weights)
    imgs = np.random.default_rng(23).integers(0, 256, (2, 368, 368, 3), dtype=np.uint8)
    eng.set_option('conv_algo', 2)
    eng.set_option('wino_tail', 1)
    eng.profile_enable(True); eng.forward_u8(imgs); prof = eng.profile(); eng.profile_enable(False)
    paf, heat = eng.get_maps()
    eng.close()
    import re
    merged = {e['layer'] for e in prof if re.search(r'/t\d+m', e['kernel'])}
    assert 'Mconv3_stage4' in merged and 'conv4_2' in merged and 'conv3_2' not in merged and len(merged) >= 30, sorted(merged)
    plan = R.splitk_plan(prof)
    x = np.stack([P.preprocess(im)[0] for im in imgs])
    rpaf, rheat = R.forward_fma(weights, x, splitk=plan)
    assert np.array_equal(paf, rpaf) and np.array_equal(heat, rheat), (np.abs(paf - rpaf).max(), np.abs(heat - rheat).max())


def test_single_image_368_runs_and_tails_bit_exact(native):
    """One 368x368 image with the plain Winograd kernel forced on every eligible layer and the tails in unit mode: the 46x46 layers
    (conv4_x, conv5_x, all 7x7 layers) and the 92- / 184-wide ones of the stem (two / four slabs) take the run geometry (labels
    "...r/t<g>") and forward_fma with that plan reproduces the maps
    bit for bit -- the launch forms batch 32 uses by default, on one frame."""
    weights = pkg('weights').synthetic_weights(0)
    eng = native.Engine(0, max_batch=1, max_h=368, max_w=368)
    eng.set_weights(weights)
    img = np.random.default_rng(22).integers(0, 256, (1, 368, 368, 3), dtype=np.uint8)
    eng.set_option('conv_algo', 2)
    eng.set_option('wino_tail', 1)
    plan, _ = forward_plan(eng, lambda: eng.forward_u8(img))
    paf, heat = eng.get_maps()
    eng.close()
    assert plan.wino_tails.get('Mconv2_stage3') == 1 and plan.wino_tails.get('Mconv1_stage2') == 2 and plan.wino_tails.get('conv4_2') == 2 \
        and plan.wino_tails.get('conv5_1_CPM') == 1 and plan.wino_tails.get('conv3_2') == 1 and plan.wino_tails.get('conv2_2') == 1 \
        and len(plan.wino_tails) == 38, plan.wino_tails
    rpaf, rheat = R.forward_fma(weights, P.preprocess(img[0]), splitk=plan)
    assert np.array_equal(paf, rpaf) and np.array_equal(heat, rheat), (np.abs(paf - rpaf).max(), np.abs(heat - rheat).max())


@pytest.mark.parametrize('shape', [(1, 64, 64), (2, 96, 128)])
def test_network_with_winograd_bit_exact_vs_order_defined_oracle(engine, shape):
    """The whole forward with every eligible layer on the Winograd kernel (32 of the 92 layers carry 98 % of the work:
    conv2_1 .. conv5_3 and the 25 7x7 layers) == forward_fma with the same layers restated by the C twin, bit for bit."""
    weights = pkg('weights').synthetic_weights(0)
    engine.set_weights(weights)
    rng = np.random.default_rng(sum(shape) + 1)
    imgs = rng.integers(0, 256, shape + (3,), dtype=np.uint8)
    engine.set_option('conv_algo', 2)
    try:
        plan, wino, _ = forward_plan(engine, lambda: engine.forward_u8(imgs), with_wino=True)
        paf, heat = engine.get_maps()
    finally:
        engine.set_option('conv_algo', 0)
    assert {'conv2_1', 'conv3_2', 'conv4_2', 'conv5_1_CPM', 'Mconv1_stage2', 'Mconv5_stage6'} <= wino and 'conv1_2' not in wino, wino
    x = np.concatenate([P.preprocess(im) for im in imgs])
    rpaf, rheat = R.forward_fma(weights, x, splitk=plan, wino=wino)
    assert np.array_equal(paf, rpaf), np.abs(paf - rpaf).max()
    assert np.array_equal(heat, rheat), np.abs(heat - rheat).max()
    # and it is the same network as the torch-CPU restatement / the direct kernels, to fp32 rounding
    tpaf, theat = N.forward(weights, x)
    assert np.abs(paf - tpaf).max() <= 1e-4 * max(1.0, np.abs(tpaf).max())
    assert np.abs(heat - theat).max() <= 1e-4 * max(1.0, np.abs(theat).max())


@pytest.mark.parametrize('name', ['e2e_person', 'e2e_people', 'e2e_dinner'])
def test_config1_reference_images_with_winograd(native, name):
    """BASELINE config 1 with every eligible layer on the Winograd kernel: what the reference's own PoseDetector returned on its
    own images (pose_detector.py:484-517) -- identical peak indices and poses, scores to 1e-4."""
    PD = pkg('pose_detector')
    g = load_e2e(name)
    det = PD.PoseDetector(weights=g['weights'], device=0)
    det.engine.set_option('conv_algo', 2)
    poses, scores = det(g['img'])
    peaks = det.engine.peaks(0)
    det.engine.close()
    assert peaks.shape == g['all_peaks'].shape
    assert np.array_equal(peaks[:, [0, 1, 2, 4]], g['all_peaks'][:, [0, 1, 2, 4]])
    assert float(np.abs(peaks[:, 3] - g['all_peaks'][:, 3]).max()) <= 1e-4
    assert np.asarray(poses).shape == g['poses'].shape and np.array_equal(np.asarray(poses), g['poses'])
    assert float(np.abs(np.asarray(scores) - g['scores']).max()) <= 1e-4


def test_batch_32_default_path_vs_cpu_oracle_and_single_images(native):
    """Batch 32 at 368x368 with default options (run-geometry Winograd kernel + unit-mode tails on the 46x46 layers, plain Winograd
    on conv2_x / conv3_x) -- ALL 32 frames against (a) the CPU oracle (torch fp32 restatement of models/CocoPoseNet.py + NumPy
    restatement of pose_detector.py:75-265): identical peak indices and poses, scores to 1e-4 (the north_star tolerance); (b) the
    same images one at a time (unit-mode Winograd / direct kernels + split-K): identical peaks and poses, scores to 1e-5 -- the
    launch forms differ by fp32 rounding only."""
    W = pkg('weights')
    eng = native.Engine(0, max_batch=32, max_h=368, max_w=368)
    w = W.synthetic_weights(0); eng.set_weights(w)
    cal = np.random.default_rng(1234).integers(0, 256, (1, 368, 368, 3), dtype=np.uint8)
    eng.forward_u8(cal); paf, heat = eng.get_maps()
    w = W.calibrate_head(w, paf[0], heat[0]); eng.set_weights({k: w[k] for k in ('Mconv7_stage6_L1', 'Mconv7_stage6_L2')})
    imgs = np.random.default_rng(3).integers(0, 256, (32, 368, 368, 3), dtype=np.uint8)
    eng.profile_enable(True); eng.detect_batch(imgs, 320, 320); prof = eng.profile(); eng.profile_enable(False)
    plan = R.splitk_plan(prof)
    assert len(plan.wino) >= 30 and len(plan.wino_tails) >= 25, (sorted(plan.wino), plan.wino_tails)
    rec = eng.results().copy()
    peaks32 = [eng.peaks(i).copy() for i in range(32)]
    assert int(rec['n_people'].sum()) > 32 and int(np.bitwise_or.reduce(rec['status'])) == 0
    worst_peak = worst_person = 0.0
    smallest_margin = np.inf
    from oracle import census
    for i in range(32):                                   # (a) the CPU oracle
        x = P.preprocess(imgs[i])
        opaf, oheat = N.forward(w, x)
        o = P.postprocess_from_net_output(opaf[0], oheat[0], 320, 320)
        op = np.asarray(o['all_peaks'], dtype=np.float64).reshape(-1, 5)
        # margins of this fixture's decisions: the smallest one says how far the frame is from a legitimate flip
        margins = np.abs(np.stack([census.margin_map(sm) for sm in o['smoothed']]))
        smallest_margin = min(smallest_margin, float(margins.min()))
        assert peaks32[i].shape == op.shape and np.array_equal(peaks32[i][:, [0, 1, 2, 4]], op[:, [0, 1, 2, 4]]), \
            (i, 'smallest decision margin of the frame %.3g (a margin below ~1e-5 is a near-tie: see tests/test_gpu_census.py)' % margins.min())
        worst_peak = max(worst_peak, float(np.abs(peaks32[i][:, 3] - op[:, 3]).max()))
        n = int(rec['n_people'][i])
        oposes = np.asarray(o['poses'], dtype=np.float64).reshape(-1, 18, 3)
        assert oposes.shape[0] == n and np.array_equal(rec['poses'][i][:n], oposes), i
        if n:
            worst_person = max(worst_person, float(np.abs(rec['scores'][i][:n] - np.asarray(o['scores']).reshape(-1)).max()))
    print('\n[batch-32 fixture, seed 3] smallest |margin| of any peak decision over the 32 frames: %.3g; max |d score| peaks %.3g, people %.3g'
          % (smallest_margin, worst_peak, worst_person))
    assert worst_peak <= 1e-4 and worst_person <= 1e-4, (worst_peak, worst_person)
    for i in range(32):                                   # (b) one image per call
        eng.detect_batch(imgs[i:i + 1], 320, 320)
        r1 = eng.results()
        p1 = eng.peaks(0)
        assert r1['n_people'][0] == rec['n_people'][i] and r1['n_peaks'][0] == rec['n_peaks'][i]
        assert np.array_equal(p1[:, [0, 1, 2, 4]], peaks32[i][:, [0, 1, 2, 4]])
        assert np.abs(p1[:, 3] - peaks32[i][:, 3]).max() <= 1e-5
        n = int(r1['n_people'][0])
        assert np.array_equal(r1['poses'][0][:n], rec['poses'][i][:n])
        assert n == 0 or np.abs(r1['scores'][0][:n] - rec['scores'][i][:n]).max() <= 1e-5
    eng.close()


@pytest.mark.parametrize('name', ['net_posenet_64x96', 'net_posenet_184x248', 'net_facenet_64x64', 'net_handnet_72x56'])
def test_reference_chain_goldens_with_winograd(native, name):
    """The goldens written by the reference's own CocoPoseNet / FaceNet / HandNet.__call__ (models/*.py) with every eligible layer
    on the Winograd kernel: maps within 1e-4 (the bar of test_gpu_reference_goldens.py, which runs the default kernel choice)."""
    import os
    from conftest import GOLDEN
    from test_reference_network import _x
    z = np.load(os.path.join(GOLDEN, name + '.npz'))
    arch = name.split('_')[1]
    h, w = [int(v) for v in z['hw']]
    img, _ = _x(arch, int(z['seed']), h, w)
    eng = native.Engine(0, max_batch=1, max_h=h, max_w=w, arch=arch)
    eng.set_weights(pkg('weights').synthetic_weights(int(z['seed']), arch) if arch != 'posenet' else pkg('weights').synthetic_weights(int(z['seed'])))
    eng.set_option('conv_algo', 2)
    eng.profile_enable(True); eng.forward_u8(img); prof = eng.profile(); eng.profile_enable(False)
    assert len(R.wino_layers(prof)) >= 10, sorted(R.wino_layers(prof))
    maps = eng.get_maps()
    eng.close()

    def rel(a, b):
        return float(np.abs(a - b).max() / max(1.0, np.abs(b).max()))
    if arch == 'posenet':
        assert rel(maps[0], z['paf']) < 1e-4 and rel(maps[1], z['heat']) < 1e-4
    else:
        assert rel(maps, z['heat']) < 1e-4


@pytest.mark.parametrize('B,cin,H,W,cout,k,relu,pool', [(1, 128, 46, 46, 128, 7, True, False), (1, 192, 46, 46, 128, 7, True, False),
                                                         (2, 64, 20, 30, 256, 7, False, False), (1, 96, 9, 11, 100, 7, True, False),
                                                         (1, 512, 46, 46, 512, 3, True, False), (1, 256, 46, 46, 128, 3, True, False),
                                                         (2, 96, 14, 18, 132, 3, False, True)])
def test_winograd_unit_mode_bit_exact_vs_c_twin(engine, B, cin, H, W, cout, k, relu, pool):
    """Unit mode of the 7x7 Winograd kernel (single images; conv_algo 3 forces it): pass 1 in units of g chunks, pass 2a, pass 2b as
    separate blocks writing slabs that the combine kernel adds in unit order == the twin's unit_g form, bit for bit."""
    x, w, b = _data(cin + H, B, cin, H, W, cout, k)
    nch = (cin + 31) // 32
    nu1_max = 8 - (3 if k == 7 else 0)                  # wino_unit_g -1: as many units as 8 slabs allow (conv_select.hip)
    g = -(-nch // nu1_max)
    engine.set_option('wino_unit_g', -1)
    try:
        y = _run(engine, x, w, b, relu, pool, 3)
    finally:
        engine.set_option('wino_unit_g', 0)
    ref = R.conv_wino(x, w, b, relu, pool, unit_g=g)
    assert np.array_equal(y, ref), (np.abs(y - ref).max(), int((y != ref).sum()))
    assert not np.array_equal(y, R.conv_wino(x, w, b, relu, pool)), 'unit mode did not run'
    t = N.conv2d_ref(x, w, b, relu=relu, pool=pool)
    assert np.abs(y - t).max() <= TOL * max(1.0, np.abs(t).max())


@pytest.mark.parametrize('B,cin,H,W,cout,k,g', [(1, 512, 46, 46, 512, 3, 6), (1, 512, 46, 46, 512, 3, 3), (1, 256, 46, 46, 512, 3, 3),
                                                 (1, 512, 46, 46, 256, 3, 5), (1, 128, 46, 46, 128, 7, 2), (2, 192, 20, 30, 128, 7, 3),
                                                 (1, 256, 46, 46, 128, 3, 4), (1, 128, 46, 46, 128, 7, 4)])      # (7x7 with ONE pass-1 unit + row 6, column 6, tap (6, 6))
def test_winograd_unit_plans_bit_exact_vs_c_twin(engine, B, cin, H, W, cout, k, g):
    """Any unit plan -- g chunks per pass-1 unit, the last unit shorter (512 channels as 6 / 6 / 4 chunks, 5 / 5 / 5 / 1) -- equals the
    twin's unit_g form bit for bit: since round 6 the selection takes the plan a dispatch simulation finishes first (one 368 x 368 image:
    conv4_2 as 3 units = 216 blocks in one round of the CUs instead of 8 units = 576 blocks), here forced through `wino_unit_g`."""
    x, w, b = _data(cin + H + g, B, cin, H, W, cout, k)
    engine.set_option('wino_unit_g', g)
    try:
        y = _run(engine, x, w, b, True, False, 3)
    finally:
        engine.set_option('wino_unit_g', 0)
    ref = R.conv_wino(x, w, b, True, False, unit_g=g)
    assert np.array_equal(y, ref), (np.abs(y - ref).max(), int((y != ref).sum()))
    assert not np.array_equal(y, R.conv_wino(x, w, b, True, False, unit_g=1)), 'the forced plan did not run'


def test_single_image_368_default_plan_bit_exact(native):
    """One 368x368 image with default options: the 46x46 7x7 layers run the Winograd kernel in unit mode, conv2_x / conv3_x the plain
    Winograd kernel, the rest the direct kernels (split-K where planned) -- and forward_fma with the plan read from the profile
    reproduces the maps bit for bit."""
    weights = pkg('weights').synthetic_weights(0)
    eng = native.Engine(0, max_batch=1, max_h=368, max_w=368)
    eng.set_weights(weights)
    img = np.random.default_rng(21).integers(0, 256, (1, 368, 368, 3), dtype=np.uint8)
    plan, _ = forward_plan(eng, lambda: eng.forward_u8(img))
    paf, heat = eng.get_maps()
    eng.close()
    assert len(plan.wino_units) >= 24 and 'conv4_2' in plan.wino_units and 'conv2_2' in plan.wino and 'conv2_2' not in plan.wino_units, \
        (plan.wino_units, sorted(plan.wino))
    rpaf, rheat = R.forward_fma(weights, P.preprocess(img[0]), splitk=plan)
    assert np.array_equal(paf, rpaf) and np.array_equal(heat, rheat), (np.abs(paf - rpaf).max(), np.abs(heat - rheat).max())


@pytest.mark.parametrize('B,h,w,must_cut,min_cut', [(32, 184, 248, 'Mconv2_stage3', 25), (24, 64, 144, 'conv3_4', 1)])      # (conv3_4: a POOLED layer cut)
def test_batch_cut_in_two_by_images_bit_exact_per_image(native, B, h, w, must_cut, min_cut):
    """A batch whose plain 7x7 launches would end in a part-filled round of the CUs (32 frames of 184 x 248: 12 blocks per image = 384 =
    1.5 rounds) is cut in two by images (conv_select.hip::wino_split_images): whole rounds through the plain kernel, the rest through
    the selection of their own count (unit mode).  The profile labels say which image took which form ("@<first>+<count>"), and
    forward_fma with the plan OF AN IMAGE reproduces that image's maps bit for bit -- on both sides of the cut; with the option off the
    batch runs as one launch per layer and every image has the same plan."""
    weights = pkg('weights').synthetic_weights(0)
    eng = native.Engine(0, max_batch=B, max_h=h, max_w=w)
    eng.set_weights(weights)
    imgs = np.random.default_rng(77).integers(0, 256, (B, h, w, 3), dtype=np.uint8)
    prof, _ = forward_plan(eng, lambda: eng.forward_u8(imgs), with_profile=True)
    paf, heat = eng.get_maps()
    cut = sorted({p['layer'] for p in prof if '@' in p['kernel']})
    if not cut:
        eng.close()
        pytest.skip('no layer was cut: the cut is a function of the CU count, the shapes of this test are chosen for 256 CUs')
    assert len(cut) >= min_cut and must_cut in cut, cut
    import re
    n0 = {int(re.search(r'@0\+(\d+)$', p['kernel']).group(1)) for p in prof if p['layer'] == must_cut and '@0+' in p['kernel']}
    assert len(n0) == 1
    n0 = n0.pop()
    assert 0 < n0 < B
    for i in (0, n0 - 1, n0, B - 1):
        plan = R.splitk_plan(prof, image=i)
        assert (must_cut in plan.wino_units) == (i >= n0), (i, n0, plan.wino_units)
        rpaf, rheat = R.forward_fma(weights, P.preprocess(imgs[i]), splitk=plan)
        assert np.array_equal(paf[i], rpaf[0]) and np.array_equal(heat[i], rheat[0]), (i, np.abs(paf[i] - rpaf[0]).max(), np.abs(heat[i] - rheat[0]).max())
    eng.set_option('wino_split', 0)
    prof0, _ = forward_plan(eng, lambda: eng.forward_u8(imgs), with_profile=True)
    paf0, heat0 = eng.get_maps()
    eng.close()
    assert not any('@' in p['kernel'] for p in prof0)
    nf = min(int(re.search(r'@(\d+)\+\d+$', p['kernel']).group(1)) for p in prof if '@' in p['kernel'] and '@0+' not in p['kernel'])      # images in front of EVERY cut
    assert np.array_equal(paf0[:nf], paf[:nf]) and np.array_equal(heat0[:nf], heat[:nf])          # plain blocks do not depend on the batch they are part of
    assert not np.array_equal(paf0[n0:], paf[n0:])
    scale = max(1.0, float(np.abs(paf0).max()), float(np.abs(heat0).max()))
    assert np.abs(paf0 - paf).max() <= 2e-5 * scale and np.abs(heat0 - heat).max() <= 2e-5 * scale


def test_cut_batches_equal_uncut_batches_to_rounding(native):
    """Random uniform batches (2-32 frames, 64-400 pixels a side) with the cut on and off: the maps agree to fp32 rounding (2e-5 of the map
    scale -- the bar between any two launch forms), the images in front of a cut bit for bit, a second run of the same batch reproduces
    the first bit for bit; and the case list does contain cuts."""
    weights = pkg('weights').synthetic_weights(0)
    rng = np.random.default_rng(2026)
    eng = native.Engine(0, max_batch=32, max_h=496, max_w=496)
    eng.set_weights(weights)
    cuts = 0
    cases = [(8, 368, 496), (12, 368, 496), (24, 496, 368)] + [(int(rng.integers(2, 33)), int(rng.integers(8, 51)) * 8, int(rng.integers(8, 51)) * 8) for _ in range(9)]
    for B, h, w in cases:
        imgs = rng.integers(0, 256, (B, h, w, 3), dtype=np.uint8)
        eng.set_option('wino_split', 1)
        prof, _ = forward_plan(eng, lambda: eng.forward_u8(imgs), with_profile=True)
        paf1, heat1 = eng.get_maps()
        eng.forward_u8(imgs)
        paf1b, heat1b = eng.get_maps()
        assert np.array_equal(paf1, paf1b) and np.array_equal(heat1, heat1b), (B, h, w)
        eng.set_option('wino_split', 0)
        eng.forward_u8(imgs)
        paf0, heat0 = eng.get_maps()
        import re
        firsts = [int(re.search(r'@(\d+)\+\d+$', p['kernel']).group(1)) for p in prof if '@' in p['kernel']]
        n_front = min([f for f in firsts if f > 0], default=B)       # images in front of every cut of this forward
        cuts += bool(firsts)
        assert np.array_equal(paf0[:n_front], paf1[:n_front]) and np.array_equal(heat0[:n_front], heat1[:n_front]), (B, h, w, n_front)
        scale = max(1.0, float(np.abs(paf0).max()), float(np.abs(heat0).max()))
        assert np.abs(paf0 - paf1).max() <= 2e-5 * scale and np.abs(heat0 - heat1).max() <= 2e-5 * scale, (B, h, w)
    eng.set_option('wino_split', 1)
    eng.close()
    assert cuts >= 3, cuts


# ---- randomised shapes for the run geometry (deterministic example set by default; PMX_FUZZ=<n> draws n fresh random examples) ----------
import os as _os
from hypothesis import given, settings, strategies as st, HealthCheck

_FUZZ = int(_os.environ.get('PMX_FUZZ', '0'))


@settings(max_examples=_FUZZ or 24, derandomize=not _FUZZ, deadline=None, database=None,
          suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])
@given(seed=st.integers(0, 10 ** 6), k=st.sampled_from([3, 7]), B=st.integers(1, 3), H=st.integers(1, 60), slabs=st.integers(1, 3),
       nch=st.integers(1, 6), cout=st.sampled_from([100, 128, 132, 256]), relu=st.booleans(), pool=st.booleans(), tail=st.booleans())
def test_random_run_geometry_shapes_bit_exact(engine, seed, k, B, H, slabs, nch, cout, relu, pool, tail):
    """Random maps of 1-3 slabs of 46 columns, any height (part-filled runs, single rows, odd heights), 1-6 chunks, padded output
    channels, with / without the fused pool, tails in unit mode or in the plain launch: the run-geometry kernels (GEOM 1 / 2, their
    unit-mode twins and conv_wino_tail_reduce_kernel) equal the plain-C twin bit for bit."""
    W = 46 * slabs
    pool = pool and k == 3 and H % 2 == 0
    cin = 32 * nch
    x, w, b = _data(seed, B, cin, H, W, cout, k)
    engine.set_option('wino_tail', 1 if tail else 0)
    try:
        y = _run(engine, x, w, b, relu, pool, 2)
    finally:
        engine.set_option('wino_tail', -1)
    g = -(-nch // (8 - (3 if k == 7 else 0)))
    ntiles = 23 * ((H + 1) // 2)
    uf = R.wino_run_unit_from(H, W)
    tailed = tail and nch >= 2 and cout % 4 == 0 and 0 < uf < ntiles and -(-nch // g) >= 2        # the plan wino_select makes
    ref = R.conv_wino(x, w, b, relu, pool, unit_g=g, unit_from=uf) if tailed else R.conv_wino(x, w, b, relu, pool)
    assert y.shape == ref.shape
    assert np.array_equal(y, ref), (k, B, cin, H, W, cout, relu, pool, tailed, float(np.abs(y - ref).max()), int((y != ref).sum()))


@settings(max_examples=_FUZZ or 16, derandomize=not _FUZZ, deadline=None, database=None,
          suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])
@given(seed=st.integers(0, 10 ** 6), k=st.sampled_from([3, 7]), B=st.integers(1, 3), H=st.integers(1, 50), W=st.integers(1, 70),
       nch=st.integers(2, 16), cout=st.sampled_from([100, 128, 132, 256]), gsel=st.integers(0, 15), pool=st.booleans())
def test_random_unit_plans_bit_exact(engine, seed, k, B, H, W, nch, cout, gsel, pool):
    """Whole launches in unit mode (rectangles) under a random unit plan -- g chunks per pass-1 unit among the plans with 2 .. 8 slabs,
    any map size, 2-16 chunks, padded output channels, pooled 3x3 layers (the combine pools) -- equal the twin's unit_g form bit for bit."""
    extra = 3 if k == 7 else 0
    plans = [g for g in range(1, nch + 1) if 2 <= -(-nch // g) + extra <= 8]
    if k == 7:
        nch = min(nch, 8)                      # (a 7x7 layer of 16 chunks at 50 x 70 is seconds of twin time; the network's widest is 6)
        plans = [g for g in range(1, nch + 1) if 2 <= -(-nch // g) + extra <= 8]
    g = plans[gsel % len(plans)]
    pool = pool and k == 3 and H % 2 == 0 and W % 2 == 0
    x, w, b = _data(seed, B, 32 * nch, H, W, cout, k)
    engine.set_option('wino_unit_g', g)
    try:
        y = _run(engine, x, w, b, True, pool, 3)
    finally:
        engine.set_option('wino_unit_g', 0)
    ref = R.conv_wino(x, w, b, True, pool, unit_g=g)
    assert y.shape == ref.shape
    assert np.array_equal(y, ref), (k, B, nch, H, W, cout, g, pool, float(np.abs(y - ref).max()), int((y != ref).sum()))
