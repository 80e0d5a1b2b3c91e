"""CPU: the oracle (oracle/postprocess_ref.py) against the committed golden fixtures, which were produced by the
VERBATIM reference post-process (oracle/make_golden.py).  This is what pins the oracle on the GPU box, where
/root/reference does not exist."""
import numpy as np
import pytest
from scipy.ndimage import gaussian_filter

from conftest import golden_cases, load_golden, conns_by_limb
from oracle import postprocess_ref as P
from oracle import fixtures as Fx


@pytest.mark.parametrize('name', golden_cases())
def test_oracle_reproduces_reference_golden(name):
    g = load_golden(name)
    map_h, map_w = [int(v) for v in g['map_hw']]
    orig_h, orig_w = [int(v) for v in g['orig_hw']]
    out = P.postprocess_from_net_output(g['paf_lo'], g['heat_lo'], map_h, map_w, orig_w=orig_w, orig_h=orig_h)
    # integer / index work: bit-exact
    assert np.array_equal(out['all_peaks'], g['all_peaks'].reshape(-1, 5))
    ref_conns = conns_by_limb(g['connections'])
    for l in range(19):
        mine = np.asarray(out['connections'][l]).reshape(-1, 3)
        assert mine.shape == ref_conns[l].shape, 'limb %d' % l
        assert np.array_equal(mine[:, :2], ref_conns[l][:, :2]), 'limb %d ids' % l
        # PAF scores: tolerance 1e-4 per BASELINE.json (observed <= 1e-12; np.dot vs explicit mul-add)
        assert np.allclose(mine[:, 2], ref_conns[l][:, 2], rtol=0, atol=1e-9)
    assert out['subsets'].shape == g['subsets'].shape
    assert np.array_equal(out['subsets'][:, :18], g['subsets'][:, :18])
    assert np.allclose(out['subsets'][:, 18:], g['subsets'][:, 18:], rtol=0, atol=1e-9)
    poses = np.asarray(out['poses'], dtype=np.float64)
    assert poses.shape == g['poses'].shape
    assert np.array_equal(poses, g['poses'])
    assert np.allclose(out['scores'], g['scores'], rtol=0, atol=1e-9)


@pytest.mark.parametrize('shape', [(320, 320), (40, 56), (21, 21), (12, 9), (5, 7)])
def test_gaussian_restatement_bit_exact_vs_scipy(shape):
    rng = np.random.default_rng(0)
    a = (rng.random(shape) * 2 - 0.5).astype('f')
    assert np.array_equal(gaussian_filter(a, sigma=2.5), P.gaussian_filter_ref(a))


def test_gaussian_taps_match_scipy_internal():
    from scipy.ndimage import _filters
    w = _filters._gaussian_kernel1d(2.5, 0, 10)
    assert np.array_equal(w, P.gaussian_kernel1d(2.5))
    assert len(w) == 21


def test_resize_is_identity_for_same_size_and_corner_aligned():
    rng = np.random.default_rng(1)
    x = rng.standard_normal((3, 9, 11)).astype('f')
    assert np.array_equal(P.resize_images_ref(x, 9, 11), x)
    y = P.resize_images_ref(x, 33, 41)       # (out-1) = 4 * (in-1): every 4th sample is an input sample
    assert np.array_equal(y[:, ::4, ::4], x)
    import torch
    t = torch.nn.functional.interpolate(torch.from_numpy(x)[None], size=(33, 41), mode='bilinear', align_corners=True)[0].numpy()
    assert np.allclose(y, t, atol=2e-6)


def test_np_sum10_is_numpys_order():
    rng = np.random.default_rng(2)
    for _ in range(200):
        v = rng.standard_normal(10) * 10.0 ** rng.integers(-8, 8, 10)
        assert P._np_sum10(v) == v.sum()


def test_fixture_renderer_shapes():
    heat, paf, poses = Fx.synthetic_maps(3, 2, 46, 46, 1.0, 0.9)
    assert heat.shape == (19, 46, 46) and paf.shape == (38, 46, 46) and poses.shape == (2, 18, 3)
    assert heat.dtype == np.float32 and paf.dtype == np.float32
    assert np.all(heat[:18] >= 0) and np.all(heat[:18] <= 1)


def test_order_defined_conv_oracle_matches_torch_and_is_deterministic():
    """oracle/conv_fma_ref.c (the bit-exact twin of the HIP kernels, checked on the GPU) agrees with the BLAS-order torch
    restatement to fp32 summation noise, handles pooled / 1x1 / partial-chunk layers, and is thread-count independent."""
    from oracle import conv_fma_ref as R, network_ref as N
    rng = np.random.default_rng(3)
    for (B, cin, H, W, cout, k, pool) in [(2, 20, 9, 11, 7, 3, False), (1, 5, 8, 10, 3, 3, True), (1, 33, 6, 7, 40, 7, False),
                                          (2, 17, 5, 5, 9, 1, False)]:
        x = rng.standard_normal((B, cin, H, W)).astype('f')
        w = (rng.standard_normal((cout, cin, k, k)) / np.sqrt(cin * k * k)).astype('f')
        b = rng.standard_normal(cout).astype('f')
        y = R.conv_fma(x, w, b, relu=True, pool=pool)
        ref = N.conv2d_ref(x, w, b, relu=True, pool=pool)
        assert y.shape == ref.shape and np.abs(y - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max())
        assert np.array_equal(y, R.conv_fma(x, w, b, relu=True, pool=pool))


def test_winograd_conv_oracle_is_the_same_convolution():
    """oracle/conv_fma_ref.c::conv_wino_ref (twin of the Winograd F(2x2, 3x3) HIP kernel, bit-checked on the GPU): the 3x3 form
    and the 7x7 form (four 3x3 sub-kernels in the frequency domain + 13 direct taps) equal the float64 convolution to fp32
    rounding -- odd sizes, images smaller than the kernel, pooling -- and exactly on data where every product and sum is exact."""
    import torch
    from oracle import conv_fma_ref as R
    rng = np.random.default_rng(4)
    for (B, cin, H, W, cout, k, relu, pool) in [(2, 32, 9, 11, 7, 3, True, False), (1, 64, 8, 10, 3, 3, True, True), (1, 32, 6, 7, 40, 7, False, False),
                                                (1, 40, 13, 5, 9, 7, True, False), (1, 32, 2, 3, 5, 7, True, False)]:
        x = rng.standard_normal((B, cin, H, W)).astype('f')
        w = (rng.standard_normal((cout, cin, k, k)) / np.sqrt(cin * k * k)).astype('f')
        b = rng.standard_normal(cout).astype('f')
        y = R.conv_wino(x, w, b, relu=relu, pool=pool)
        t = torch.nn.functional.conv2d(torch.from_numpy(x).double(), torch.from_numpy(w).double(), torch.from_numpy(b).double(), padding=k // 2)
        t = torch.relu(t) if relu else t
        t = torch.nn.functional.max_pool2d(t, 2, 2) if pool else t
        assert y.shape == tuple(t.shape) and np.abs(y - t.numpy()).max() <= 1e-5 * max(1.0, float(t.abs().max()))
        assert np.array_equal(y, R.conv_wino(x, w, b, relu=relu, pool=pool))
    # small integers, weights that are multiples of 4: G g G^T, every transform and every product is exact -> equality
    x = rng.integers(-3, 4, (1, 32, 10, 12)).astype('f')
    for k in (3, 7):
        w = (4 * rng.integers(-2, 3, (6, 32, k, k))).astype('f')
        b = rng.integers(-5, 6, 6).astype('f')
        t = torch.nn.functional.conv2d(torch.from_numpy(x).double(), torch.from_numpy(w).double(), torch.from_numpy(b).double(), padding=k // 2).numpy()
        assert np.array_equal(R.conv_wino(x, w, b), t.astype('f'))


def test_launch_plan_from_profile_labels():
    """oracle/conv_fma_ref.py::splitk_plan reads the launch plan the kernels report in their profile labels: K slices of split direct
    launches ("/k3-2-2-1"), layers on the Winograd kernel, and those in its unit mode ("/u<g>") -- what forward_fma needs to restate a
    forward bit for bit; and the unit form of the Winograd twin is the same convolution."""
    import torch
    from oracle import conv_fma_ref as R
    prof = [{'layer': 'conv1_1+conv1_2', 'kernel': 'conv1_fused_t8x16_n64'}, {'layer': 'conv2_2', 'kernel': 'conv_wino_f2x2_3x3'},
            {'layer': 'conv4_2', 'kernel': 'conv_wino_f2x2_3x3/u6'}, {'layer': 'conv4_4_CPM', 'kernel': 'conv3x3_v5_t8x8_n64/k3-3-2'},
            {'layer': 'Mconv2_stage3', 'kernel': 'conv_wino_f2x2_7x7/u1'}, {'layer': 'Mconv1_stage2', 'kernel': 'conv7x7_v5_t8x8_n64/k5-3-3-1'},
            {'layer': 'pp_peaks', 'kernel': 'pp_peaks'}]
    plan = R.splitk_plan(prof)
    assert dict(plan) == {'conv4_4_CPM': [3, 3, 2], 'Mconv1_stage2': [5, 3, 3, 1]}
    assert plan.wino == {'conv2_2', 'conv4_2', 'Mconv2_stage3'} and plan.wino_units == {'conv4_2': 6, 'Mconv2_stage3': 1}
    assert R.wino_layers(prof) == plan.wino
    rng = np.random.default_rng(9)
    for (cin, k, g) in [(128, 7, 1), (192, 7, 2), (96, 3, 1), (160, 3, 2)]:
        x = rng.standard_normal((1, cin, 9, 10)).astype('f')
        w = (rng.standard_normal((6, cin, k, k)) / np.sqrt(cin * k * k)).astype('f')
        b = rng.standard_normal(6).astype('f')
        y = R.conv_wino(x, w, b, relu=True, unit_g=g)
        t = torch.relu(torch.nn.functional.conv2d(torch.from_numpy(x).double(), torch.from_numpy(w).double(), torch.from_numpy(b).double(), padding=k // 2)).numpy()
        assert np.abs(y - t).max() <= 1e-5 * max(1.0, np.abs(t).max())
        assert not np.array_equal(y, R.conv_wino(x, w, b, relu=True)), 'the unit form is a different summation'


def test_merged_tail_geometry_covers_every_tail_tile_once():
    """The stream decomposition behind the merged-tail launch (csrc/conv_wino.hip GEOM 3, restated in
    conv_fma_ref.wino_merged_tail_blocks): every tail tile of every image in exactly one block row, at most three images per block, the
    side-by-side halos within the 2 * 32 + 3 * (ks - 1) columns the kernel's LDS tile holds; maps whose tail wraps over two tile rows
    or is shorter than 16 tiles are not merged."""
    from oracle import conv_fma_ref as R
    merge_h = set()
    for ks in (3, 7):
        for H in range(1, 70):
            for B in (1, 2, 3, 5, 7, 31, 32, 33, 64):
                blocks = R.wino_merged_tail_blocks(B, H, 46, ks)
                ntiles = 23 * ((H + 1) // 2)
                nt = ntiles % 32
                if blocks is None:
                    assert B < 2 or nt < 16 or (ntiles - nt) % 23 + nt > 23, (B, H)
                    continue
                merge_h.add(H)
                seen = []
                for segs in blocks:
                    assert 1 <= len(segs) <= 3 and sum(n for _, _, n, _ in segs) <= 32
                    assert segs[-1][3] + 2 * segs[-1][2] + ks - 1 <= 2 * 32 + 3 * (ks - 1)
                    assert all(a[3] + 2 * a[2] + ks - 1 == b[3] for a, b in zip(segs, segs[1:]))          # side by side, no overlap
                    seen += [(img, tt + i) for img, tt, n, _ in segs for i in range(n)]
                assert seen == [(b, t) for b in range(B) for t in range(nt)]                                # stream order, each tile once
                assert all(sum(n for _, _, n, _ in segs) == 32 for segs in blocks[:-1])                      # only the last block is part-filled
    assert {45, 46, 37, 38, 23, 24, 51, 52} <= merge_h and 44 not in merge_h and 40 not in merge_h
    assert [len(s) for s in R.wino_merged_tail_blocks(5, 46, 46, 7)] == [2, 3, 2]                            # 17 + 15 | 2 + 17 + 13 | 4 + 17


def test_launch_plan_of_a_batch_cut_in_two_by_images():
    """A batch whose plain launch would end in a part-filled round of the CUs is cut in two by images (csrc/conv_select.hip::
    wino_split_images); the halves carry "@<first image>+<count>" in their labels and may run different forms: the plan is per image."""
    from oracle import conv_fma_ref as R
    prof = [{'layer': 'conv2_2', 'kernel': 'conv_wino_f2x2_3x3'}, {'layer': 'Mconv2_stage3', 'kernel': 'conv_wino_f2x2_7x7@0+5'},
            {'layer': 'Mconv2_stage3', 'kernel': 'conv_wino_f2x2_7x7/u2@5+3'}, {'layer': 'conv4_2', 'kernel': 'conv_wino_f2x2_3x3r/t6m@0+6'},
            {'layer': 'conv4_2', 'kernel': 'conv_wino_f2x2_3x3r/t6m:units@0+6'}, {'layer': 'conv4_2', 'kernel': 'conv_wino_f2x2_3x3/u3@6+2'}]
    with pytest.raises(ValueError):
        R.splitk_plan(prof)
    p0, p6, p23 = R.splitk_plan(prof, image=0), R.splitk_plan(prof, image=5), R.splitk_plan(prof, image=7)
    assert p0.wino == p6.wino == p23.wino == {'conv2_2', 'Mconv2_stage3', 'conv4_2'}
    assert p0.wino_units == {} and p0.wino_tails == {'conv4_2': 6}
    assert p6.wino_units == {'Mconv2_stage3': 2} and p6.wino_tails == {'conv4_2': 6}
    assert p23.wino_units == {'Mconv2_stage3': 2, 'conv4_2': 3} and p23.wino_tails == {}
    assert R.splitk_plan(prof[:1]).wino == {'conv2_2'}          # (no split: no image needed)

