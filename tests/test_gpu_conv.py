"""GPU (T0): every convolution kernel variant through the C ABI (pmx_conv2d) against the torch-CPU fp32 reference
of the same op (oracle/network_ref.py::conv2d_ref = L.Convolution2D semantics [+relu] [+2x2 max-pool])."""
import numpy as np
import pytest

from oracle import network_ref as N

pytestmark = pytest.mark.gpu

# fp32 MFMA = exact fp32 FMA chain; the torch reference sums in a different order -> tolerance scaled by sqrt(K)
TOL = 2e-5


def _case(engine, B, cin, H, W, cout, k, relu, pool, seed, variant=None):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((B, cin, H, W)).astype('f')
    w = (rng.standard_normal((cout, cin, k, k)) / np.sqrt(cin * k * k)).astype('f')
    b = rng.standard_normal(cout).astype('f')
    for kk in (1, 3, 7):
        engine.set_option('force_variant_k%d' % kk, -1)
    if variant is not None:
        engine.set_option('force_variant_k%d' % k, variant)
    y = engine.conv2d(x, w, b, relu=relu, pool=pool)
    engine.set_option('force_variant_k%d' % k, -1)
    ref = N.conv2d_ref(x, w, b, relu=relu, pool=pool)
    assert y.shape == ref.shape
    assert np.isfinite(y).all(), 'unwritten (poisoned) or non-finite outputs'
    err = np.abs(y - ref).max()
    assert err <= TOL * max(1.0, np.abs(ref).max()), (err, np.abs(ref).max())


def _exact_case(engine, B, cin, H, W, cout, k, relu, pool, seed, variant=None, ksplit=1):
    """GPU conv == oracle/conv_fma_ref.c bit for bit (same K order, sequential fused multiply-add chain; `ksplit` K slices
    combined left to right on both sides)."""
    from oracle import conv_fma_ref as R
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((B, cin, H, W)).astype('f')
    w = (rng.standard_normal((cout, cin, k, k)) / np.sqrt(cin * k * k)).astype('f')
    b = rng.standard_normal(cout).astype('f')
    for kk in (1, 3, 7):
        engine.set_option('force_variant_k%d' % kk, -1)
    if variant is not None:
        engine.set_option('force_variant_k%d' % k, variant)
    engine.set_option('ksplit', ksplit)
    y = engine.conv2d(x, w, b, relu=relu, pool=pool)
    engine.set_option('ksplit', 0)
    engine.set_option('force_variant_k%d' % k, -1)
    ref = R.conv_fma(x, w, b, relu=relu, pool=pool, splitk=ksplit)
    assert y.shape == ref.shape
    assert np.array_equal(y, ref), (np.abs(y - ref).max(), int((y != ref).sum()))
    if ksplit > 1:      # and the split result is the same convolution (vs torch, tolerance)
        t = N.conv2d_ref(x, w, b, relu=relu, pool=pool)
        assert np.abs(y - t).max() <= TOL * max(1.0, np.abs(t).max())


@pytest.mark.parametrize('variant,k,cin,h,w,cout,pool', [
    (None, 7, 40, 12, 15, 128, False), (None, 3, 70, 14, 10, 64, True), (None, 1, 100, 9, 13, 38, False),
    (3, 1, 128, 8, 16, 128, False), (2, 3, 16, 10, 12, 64, False),            # v1
    (10, 7, 48, 6, 46, 128, False), (14, 3, 35, 16, 16, 64, True), (16, 3, 20, 9, 9, 19, False),    # v5
    (17, 7, 33, 13, 46, 128, False), (18, 3, 140, 25, 92, 256, False), (19, 3, 128, 12, 46, 128, True),
    (21, 7, 17, 7, 46, 200, False),                                                                    # v6
    (20, 3, 3, 20, 24, 64, False)])                                                                    # conv1_1
def test_conv_bit_exact_vs_order_defined_c_oracle(engine, variant, k, cin, h, w, cout, pool):
    _exact_case(engine, 2, cin, h, w, cout, k, True, pool, seed=700 + k + cin, variant=variant)


@pytest.mark.parametrize('variant,k,cin,h,w,cout,pool,S', [
    (15, 7, 128, 46, 46, 128, False, 4), (15, 7, 128, 20, 23, 128, False, 8), (15, 7, 185, 16, 16, 64, False, 5),   # 12 chunks in 5 slices
    (16, 3, 512, 23, 23, 256, False, 4), (16, 3, 256, 24, 40, 128, True, 2), (13, 3, 100, 24, 32, 128, True, 3),
    (12, 7, 48, 16, 32, 128, False, 3), (14, 3, 64, 16, 32, 64, False, 4), (10, 7, 64, 6, 46, 128, False, 2)])
def test_split_k_bit_exact_vs_order_defined_c_oracle(engine, variant, k, cin, h, w, cout, pool, S):
    """Split-K of the v5 kernels (single-image launches): S slices over the 16-channel chunks, slabs combined left to right,
    then bias / ReLU / pool -- bit-identical to the plain-C oracle with the same S, for even and uneven chunk splits.
    (The automatic choice is covered at network level: the launch plan is read from the profile labels.)"""
    _exact_case(engine, 1, cin, h, w, cout, k, True, pool, seed=900 + k + cin + S, variant=variant, ksplit=S)


# variant indices: see conv_mfma.hip g_variants
@pytest.mark.parametrize('variant,k,cout', [(0, 7, 128), (5, 7, 128), (1, 3, 128), (2, 3, 64), (6, 3, 64),
                                            (3, 1, 128), (4, 1, 64), (7, 1, 64)])
def test_variants_basic(engine, variant, k, cout):
    _case(engine, 2, 32, 24, 40, cout, k, True, False, seed=variant, variant=variant)


@pytest.mark.parametrize('variant,k,cout', [(10, 7, 128), (11, 3, 128), (12, 7, 128), (13, 3, 256), (14, 3, 64), (15, 7, 128),
                                            (16, 3, 38), (24, 7, 128), (25, 3, 256), (26, 3, 64)])
@pytest.mark.parametrize('hw,cin', [((46, 46), 48), ((9, 21), 16), ((20, 50), 185)])
def test_v5_variants(engine, variant, k, cout, hw, cin):
    # v5: v4 with fully unrolled taps (immediate LDS offsets) and buffer-resource weight loads (no address VALU in the loop)
    _case(engine, 3, cin, hw[0], hw[1], cout, k, True, False, seed=370 + variant, variant=variant)


@pytest.mark.parametrize('variant,k', [(17, 7), (18, 3)])
@pytest.mark.parametrize('h,cin,cout,B', [(46, 48, 128, 3), (46, 185, 256, 2), (9, 16, 128, 5), (47, 32, 100, 1)])
def test_v6_variants(engine, variant, k, h, cin, cout, B):
    # v6: one block per CU, 17 x 32 consecutive pixels of a 46-wide map x 128 channels (last tile partially filled,
    # blocks that start mid-row, maps shorter / taller than 46 rows, cout below the 128-channel block)
    _case(engine, B, cin, h, 46, cout, k, True, False, seed=470 + variant + h, variant=variant)


@pytest.mark.parametrize('variant,k,pool', [(21, 7, False), (22, 3, False), (23, 3, True)])
@pytest.mark.parametrize('h,w,cin,cout,B', [(46, 46, 48, 128, 3), (20, 92, 185, 256, 1)])
def test_v6_nine_tile_blocks(engine, variant, k, pool, h, w, cin, cout, B):
    _case(engine, B, cin, h, w, cout, k, True, pool, seed=560 + variant + h, variant=variant)


@pytest.mark.parametrize('h,w,cin,cout,B', [(92, 92, 48, 128, 2), (31, 92, 128, 256, 1), (20, 184, 32, 128, 1), (7, 368, 16, 128, 1)])
def test_v6_slabs(engine, h, w, cin, cout, B):
    # maps wider than one 46-column slab: the left / right halo columns come from the neighbouring slab
    _case(engine, B, cin, h, w, cout, 3, True, False, seed=520 + h, variant=18)
    if h >= 20:
        _case(engine, B, cin, h, w, cout, 7, False, False, seed=530 + h, variant=17)


@pytest.mark.parametrize('h,w,cin,cout,B', [(46, 46, 48, 128, 3), (92, 92, 32, 256, 1), (24, 184, 64, 128, 2), (2, 46, 16, 128, 1),
                                            (50, 92, 16, 100, 1)])
def test_v6_fused_relu_maxpool(engine, h, w, cin, cout, B):
    # row-pair pixel order: four consecutive MFMA rows = one 2x2 window, pooled in registers
    _case(engine, B, cin, h, w, cout, 3, True, True, seed=540 + h, variant=19)


@pytest.mark.parametrize('cin,h,w,B', [(3, 40, 56, 2), (3, 37, 21, 1), (2, 16, 16, 3), (3, 368, 368, 1)])
def test_c3_packed_k_kernel(engine, cin, h, w, B):
    """conv1_1-shaped layers (<= 3 input channels -> 64): K packed to 14 k-pairs; vs torch, and bit-identical to the generic
    16-channel-chunk kernel (same fp32 FMA chain per output)."""
    _case(engine, B, cin, h, w, 64, 3, True, False, seed=600 + h, variant=20)
    rng = np.random.default_rng(h)
    x = rng.standard_normal((B, cin, h, w)).astype('f')
    wt = rng.standard_normal((64, cin, 3, 3)).astype('f')
    b = rng.standard_normal(64).astype('f')
    engine.set_option('force_variant_k3', 20)
    yc3 = engine.conv2d(x, wt, b, relu=False)
    engine.set_option('force_variant_k3', 14)
    yv5 = engine.conv2d(x, wt, b, relu=False)
    engine.set_option('force_variant_k3', -1)
    assert np.array_equal(yc3, yv5)


@pytest.mark.parametrize('variant', [13, 14, 25, 26])
def test_v5_fused_relu_maxpool(engine, variant):
    _case(engine, 2, 32, 24, 40, 64 if variant in (14, 26) else 128, 3, True, True, seed=390 + variant, variant=variant)


def test_kernel_generations_compute_identical_bits(engine):
    """v1 (weights through LDS), v5 and the default selection compute the same fp32 FMA chains in the same K order."""
    from conftest import pkg
    w = pkg('weights').synthetic_weights(0)
    engine.set_weights(w)
    img = np.random.default_rng(5).integers(0, 256, (2, 184, 184, 3), dtype=np.uint8)
    maps = {}
    engine.set_option('ksplit', 1)          # (split-K changes the summation tree of small launches: compared separately)
    engine.set_option('conv1_wino', 0)      # (the fused conv1 launch of generation 6 in its direct form: the Winograd form is another arithmetic)
    for gen in (1, 5, 6):
        engine.set_option('kernel_gen', gen)
        engine.forward_u8(img)
        maps[gen] = engine.get_maps()
    engine.set_option('kernel_gen', 6)      # library default (v6 only engages on maps a multiple of 46 wide: see below)
    engine.set_option('ksplit', 0)
    engine.set_option('conv1_wino', 1)
    for gen in (5, 6):
        assert np.array_equal(maps[1][0], maps[gen][0]) and np.array_equal(maps[1][1], maps[gen][1])


def test_v6_network_equals_v5_network_bitwise(engine):
    """368x368 input (46x46 maps) at batch 32: the one-block-per-CU v6 kernels take the 7x7 / 3x3 layers whose blocks fill
    whole rounds of the 256 CUs; same K order -> the same bits as v5, and as batch 1 (small tiles)."""
    from conftest import pkg
    native = pkg('native')
    w = pkg('weights').synthetic_weights(0)
    eng = native.Engine(0, max_batch=32, max_h=368, max_w=368)
    eng.set_weights(w)
    img = np.random.default_rng(7).integers(0, 256, (32, 368, 368, 3), dtype=np.uint8)
    eng.set_option('ksplit', 1)             # one K order everywhere (the v5 strip launches of generation 5 could be split)
    eng.set_option('conv_algo', 0)          # ... and the direct kernels at every launch size (batch 32 would take the Winograd kernel)
    eng.set_option('kernel_gen', 5)
    eng.forward_u8(img)
    p5, h5 = eng.get_maps()
    eng.set_option('kernel_gen', 6)
    eng.profile_enable(True)
    eng.forward_u8(img)
    p6, h6 = eng.get_maps()
    names = {e['kernel'] for e in eng.profile()}
    eng.profile_enable(False)
    assert any('_v6_' in k for k in names), names
    assert np.array_equal(p5, p6) and np.array_equal(h5, h6)
    eng.set_option('ksplit', 1)             # unsplit small tiles: the same K order as the batch kernels
    eng.forward_u8(img[5:6])
    p1, h1 = eng.get_maps()
    assert np.array_equal(p1[0], p6[5]) and np.array_equal(h1[0], h6[5])
    eng.set_option('ksplit', 0)             # default: split-K for the single image -- same maps to summation-order noise
    eng.forward_u8(img[5:6])
    q1, g1 = eng.get_maps()
    assert np.abs(q1[0] - p6[5]).max() <= 1e-4 * max(1.0, np.abs(p6[5]).max())
    assert np.abs(g1[0] - h6[5]).max() <= 1e-4 * max(1.0, np.abs(h6[5]).max())
    eng.close()


@pytest.mark.parametrize('variant,k', [(8, 7), (9, 3)])
@pytest.mark.parametrize('hw', [(46, 46), (10, 46), (7, 30), (6, 100)])
def test_row_strip_variants(engine, variant, k, hw):
    # 2 x 46 strips: 92 real pixels padded to 96 MFMA rows (masked); also widths != 46 and odd heights
    _case(engine, 2, 32, hw[0], hw[1], 128, k, True, False, seed=50 + variant + hw[0], variant=variant)


@pytest.mark.parametrize('k', [1, 3, 7])
def test_asymmetric_shapes_and_edges(engine, k):
    # H, W not multiples of the tile, odd sizes, transposition-detecting (H != W, cin != cout)
    _case(engine, 3, 16, 13, 29, 38, k, False, False, seed=10 + k)
    _case(engine, 1, 48, 46, 46, 19, k, True, False, seed=20 + k)


@pytest.mark.parametrize('variant', [1, 2, 6])
def test_fused_relu_maxpool(engine, variant):
    _case(engine, 2, 16, 20, 36, 64 if variant != 1 else 128, 3, True, True, seed=30 + variant, variant=variant)


def test_cin_padding_3_channels(engine):
    _case(engine, 2, 3, 32, 48, 64, 3, True, False, seed=40)       # conv1_1 shape class (cin 3 -> 16)


def test_cin_185_like_concat(engine):
    _case(engine, 1, 185, 46, 46, 128, 7, True, False, seed=41)    # Mconv1 shape class (cin 185 -> 192)


def test_deep_k_512(engine):
    _case(engine, 1, 512, 23, 23, 256, 3, True, False, seed=42)    # conv4_3 shape class


def test_identity_kernel_detects_layout_errors(engine):
    # A = I check with an asymmetric weight: y[:, n] = x[:, perm[n]] exactly
    rng = np.random.default_rng(5)
    cin = cout = 32
    perm = rng.permutation(cin)
    w = np.zeros((cout, cin, 3, 3), 'f')
    w[np.arange(cout), perm, 1, 1] = 1.0
    x = rng.standard_normal((1, cin, 16, 32)).astype('f')
    y = engine.conv2d(x, w, None)
    assert np.array_equal(y, x[:, perm])
    # shifted tap: y(y, x) = x(y + 1, x - 1) with zero padding -> checks ky/kx orientation (cross-correlation)
    w = np.zeros((cout, cin, 3, 3), 'f')
    w[np.arange(cout), np.arange(cin), 2, 0] = 1.0
    y = engine.conv2d(x, w, None)
    ref = np.zeros_like(x)
    ref[:, :, :-1, 1:] = x[:, :, 1:, :-1]
    assert np.array_equal(y, ref)


# ---- randomised shapes (deterministic example set by default; PMX_FUZZ=<n> draws n fresh random examples) -------------------
import os as _os
from hypothesis import given, settings, strategies as st, HealthCheck

_FUZZ = int(_os.environ.get('PMX_FUZZ', '0'))


@settings(max_examples=_FUZZ or 30, derandomize=not _FUZZ, deadline=None, database=None,
          suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])
@given(seed=st.integers(0, 10 ** 6), k=st.sampled_from([1, 3, 7]), B=st.integers(1, 3), h=st.integers(1, 40), w=st.integers(1, 60),
       cin=st.integers(1, 70), cout=st.integers(1, 260), relu=st.booleans(), pool=st.booleans(), gen=st.sampled_from([1, 5, 6]),
       slab=st.sampled_from([0, 0, 46, 92]))
def test_random_shapes_default_selection(engine, seed, k, B, h, w, cin, cout, relu, pool, gen, slab):
    """Whatever conv_pick_variant chooses for a random layer shape must match the torch fp32 reference (odd sizes, single
    rows / columns, partial channel chunks, pooled layers, maps that are / are not a multiple of the 46-column slab)."""
    if slab:
        w = slab
    if pool:
        h, w = max(2, h - h % 2), max(2, w - w % 2)
        if k == 1:
            pool = False
    engine.set_option('kernel_gen', gen)
    try:
        _case(engine, B, cin, h, w, cout, k, relu, pool, seed=seed)
    finally:
        engine.set_option('kernel_gen', 6)


@settings(max_examples=_FUZZ or 12, derandomize=not _FUZZ, deadline=None, database=None,
          suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])
@given(seed=st.integers(0, 10 ** 6), variant=st.sampled_from([17, 18, 19, 21, 22, 23]), B=st.integers(1, 3), h=st.integers(1, 50),
       slabs=st.integers(1, 3), cin=st.integers(1, 70), cout=st.integers(65, 300), relu=st.booleans())
def test_random_shapes_v6(engine, seed, variant, B, h, slabs, cin, cout, relu):
    """The one-block-per-CU kernels forced onto random map heights / slab counts / channel counts (blocks that start
    mid-row, last block nearly empty, single-row maps, pooled row pairs)."""
    pool = variant in (19, 23)
    k = 7 if variant in (17, 21) else 3
    if pool:
        h = max(2, h - h % 2)
    _case(engine, B, cin, h, 46 * slabs, cout, k, relu, pool, seed=seed, variant=variant)


@pytest.mark.parametrize('arch,B,hw', [('posenet', 1, (64, 96)), ('posenet', 5, (184, 120)), ('posenet', 32, (368, 368)),
                                       ('facenet', 2, (96, 96)), ('handnet', 3, (72, 104))])
def test_fused_1x1_pairs_equal_separate_layers_bitwise(native, arch, B, hw):
    """conv5_4 -> conv5_5 and Mconv6 -> Mconv7 (conv6_1 -> conv6_2 for the CPM nets) as ONE launch (conv1x1_pair_kernel: hidden
    map through LDS) == the two single-layer launches, bit for bit: every output walks K in the same order."""
    from conftest import pkg
    eng = native.Engine(0, max_batch=B, max_h=hw[0], max_w=hw[1], arch=arch)
    eng.set_weights(pkg('weights').synthetic_weights(2, arch))
    imgs = np.random.default_rng(B).integers(0, 256, (B,) + hw + (3,), dtype=np.uint8)
    outs = {}
    for fuse in (0, 1):
        eng.set_option('fuse_pairs', fuse)
        eng.profile_reset()
        eng.profile_enable(True)
        eng.forward_u8(imgs)
        names = {e['kernel'] for e in eng.profile()}
        eng.profile_enable(False)
        assert any('pair' in k for k in names) == bool(fuse), names
        outs[fuse] = eng.get_maps()
    eng.close()
    if arch == 'posenet':
        assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    else:
        assert np.array_equal(outs[0], outs[1])


@settings(max_examples=_FUZZ or 20, derandomize=not _FUZZ, deadline=None, database=None,
          suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])
@given(seed=st.integers(0, 10 ** 6), k=st.sampled_from([3, 7]), B=st.integers(1, 2), h=st.integers(2, 24), w=st.integers(2, 40),
       nch=st.integers(2, 9), cout=st.sampled_from([64, 128, 192]), pool=st.booleans(), cuts=st.lists(st.integers(1, 8), min_size=1, max_size=4))
def test_random_split_k_plans_bit_exact(engine, seed, k, B, h, w, nch, cout, pool, cuts):
    """Random explicit slice plans (uneven, any order) on random shapes: the slice kernels + the combine kernel equal the plain-C
    oracle with the same plan, bit for bit."""
    from oracle import conv_fma_ref as R
    # a plan = chunk counts >= 1 summing to nch
    sizes, left = [], nch
    for c in cuts:
        if left <= 1:
            break
        s = min(c, left - 1)
        sizes.append(s)
        left -= s
    sizes.append(left)
    if len(sizes) < 2:
        sizes = [nch - 1, 1]
    if pool:
        h, w = 2 * ((h + 1) // 2), 2 * ((w + 1) // 2)
    cin = 16 * nch - int(seed % 5)          # partial last chunk now and then
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((B, cin, h, w)).astype('f')
    wt = (rng.standard_normal((cout, cin, k, k)) / np.sqrt(cin * k * k)).astype('f')
    b = rng.standard_normal(cout).astype('f')
    engine.set_option('kernel_gen', 5)
    engine.set_option('ksplit_plan', int(''.join(str(v) for v in sizes)))
    y = engine.conv2d(x, wt, b, relu=True, pool=pool)
    engine.set_option('ksplit', 0)
    engine.set_option('kernel_gen', 6)
    ref = R.conv_fma(x, wt, b, relu=True, pool=pool, splitk=sizes)
    assert np.array_equal(y, ref), (sizes, np.abs(y - ref).max())


@pytest.mark.parametrize('arch,B,hw', [('posenet', 1, (368, 368)), ('posenet', 3, (184, 248)), ('posenet', 32, (368, 368)), ('handnet', 2, (368, 368))])
def test_fused_conv1_equals_separate_layers_bitwise(native, arch, B, hw):
    """conv1_1 recomputed on the halo of conv1_2's tiles (conv1_fused_kernel: 3 -> 64 through LDS, then 64 -> 64 + pool) == the two
    launches, bit for bit: both layers keep their K walk.  Image borders, odd tile counts and the batch are all exercised."""
    from conftest import pkg
    eng = native.Engine(0, max_batch=B, max_h=hw[0], max_w=hw[1], arch=arch)
    eng.set_weights(pkg('weights').synthetic_weights(4, arch))
    imgs = np.random.default_rng(B + hw[1]).integers(0, 256, (B,) + hw + (3,), dtype=np.uint8)
    outs = {}
    eng.set_option('conv1_wino', 0)      # (the direct form of the fused launch; its Winograd form is another arithmetic: test_conv1_wino_* below)
    for fuse in (0, 1):
        eng.set_option('fuse_conv1', fuse)
        eng.profile_reset()
        eng.profile_enable(True)
        eng.forward_u8(imgs)
        names = {e['kernel'] for e in eng.profile()}
        eng.profile_enable(False)
        assert any('conv1_fused' in k for k in names) == bool(fuse), names
        outs[fuse] = eng.get_maps()
    eng.close()
    if arch == 'posenet':
        assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    else:
        assert np.array_equal(outs[0], outs[1])


@pytest.mark.parametrize('B,hw', [(7, (88, 104)), (2, (184, 248)), (1, (368, 368)), (2, (16, 24)), (3, (24, 40)), (2, (72, 16)), (1, (56, 120))])
def test_conv1_wino_network_bit_exact_vs_twin(native, B, hw):
    """conv1_1 + conv1_2 as one launch with conv1_2 in Winograd F(2x2, 3x3) on 16 x 16 squares (conv1_wino_kernel: conv1_1 recomputed on
    the 18 x 18 halo, the direct chain; conv1_2 = conv_wino_kernel<3, pool>'s arithmetic at 64 tiles x 64 channels per block) through the
    whole network against the order-defined twin (conv_fma for conv1_1, conv_wino_ref for conv1_2 and every other Winograd layer), bit
    for bit.  Sizes: maps that are no multiple of the 16-pixel squares (clipped squares, masked stores), a batch, one full frame."""
    from conftest import pkg
    from oracle import conv_fma_ref, postprocess_ref
    w = pkg('weights').synthetic_weights(5)
    eng = native.Engine(0, max_batch=B, max_h=hw[0], max_w=hw[1])
    eng.set_weights(w)
    eng.set_option('conv1_wino', 2)       # (2: also where the launch has fewer blocks than CUs)
    imgs = np.random.default_rng(B * 1000 + hw[1]).integers(0, 256, (B,) + hw + (3,), dtype=np.uint8)
    eng.profile_reset(); eng.profile_enable(True)
    eng.forward_u8(imgs)
    prof = eng.profile()
    eng.profile_enable(False)
    assert any(e['kernel'].startswith('conv_wino1_') for e in prof), {e['kernel'] for e in prof}
    paf, heat = eng.get_maps()
    # the same images with the direct fused launch: another fp32 arithmetic, the maps agree to rounding
    eng.set_option('conv1_wino', 0)
    eng.forward_u8(imgs)
    dpaf, dheat = eng.get_maps()
    eng.close()
    plan = conv_fma_ref.splitk_plan(prof)
    assert 'conv1_2' in plan.wino
    nb = min(B, 2)                         # (the twin is a plain-C loop: two images are enough to see the batch stride)
    x = np.concatenate([postprocess_ref.preprocess(im) for im in imgs[:nb]])
    epaf, eheat = conv_fma_ref.forward_fma(w, x, splitk=plan)
    assert np.array_equal(paf[:nb], epaf) and np.array_equal(heat[:nb], eheat)
    scale = max(np.abs(dpaf).max(), np.abs(dheat).max(), 1.0)
    assert max(np.abs(paf - dpaf).max(), np.abs(heat - dheat).max()) <= 1e-4 * scale
    assert not np.array_equal(paf, dpaf)
