"""TEST INFRASTRUCTURE: ctypes binding of oracle/conv_fma_ref.c (order-defined fp32 conv, bit-exact twin of the HIP kernels).

Built by `build()` below (gcc, -ffp-contract=off) into oracle/_build/ -- called from __graft_entry__.build() and lazily here."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, 'conv_fma_ref.c')
OUT_DIR = os.path.join(HERE, '_build')
LIB = os.path.join(OUT_DIR, 'libconv_fma_ref.so')


def build(force=False):
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= os.path.getmtime(SRC):
        return LIB
    os.makedirs(OUT_DIR, exist_ok=True)
    subprocess.check_call(['gcc', '-O2', '-mfma', '-fopenmp', '-ffp-contract=off', '-shared', '-fPIC', '-o', LIB, SRC, '-lm'])
    return LIB


_lib = None


def conv_fma(x, w, b, relu=False, pool=False, splitk=1):
    """x (B, cin, H, W), w (cout, cin, k, k), b (cout,) float32 -> y float32, in the HIP kernels' summation order
    (`splitk`: K slices over the 16-channel chunks, added left to right -- what the kernels do for small launches; an int S
    means S near-even slices, larger first (the kernels' forced split), a sequence gives the chunks of every slice)."""
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.conv_fma_ref.restype = None
        _lib.conv_fma_ref.argtypes = [C.c_void_p] * 4 + [C.c_int] * 9 + [C.c_void_p]
    x = np.ascontiguousarray(x, np.float32)
    w = np.ascontiguousarray(w, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    B, cin, H, W = x.shape
    cout, _, ks, _ = w.shape
    y = np.empty((B, cout, H // 2 if pool else H, W // 2 if pool else W), np.float32)
    nch = (cin + 15) // 16
    if isinstance(splitk, (int, np.integer)):
        S = max(1, min(int(splitk), nch, 8))
        sizes = [nch // S + (1 if s < nch % S else 0) for s in range(S)]
    else:
        sizes = [int(v) for v in splitk]
    assert sum(sizes) == nch and len(sizes) <= 8, (sizes, nch)
    arr = np.asarray(sizes, dtype=np.int32)
    _lib.conv_fma_ref(x.ctypes.data, w.ctypes.data, b.ctypes.data, y.ctypes.data, B, cin, H, W, cout, ks, int(relu), int(pool),
                      len(sizes), arr.ctypes.data)
    return y


RUN_TX = 23       # tile columns of a slab of the run geometry (46 pixels)


def wino_run_unit_from(H, W):
    """First Winograd tile (row-major index inside a slab) of the part-filled last block of an (image, slab) in the kernel's run
    geometry: the map is cut into slabs of 46 columns and a block owns 32 consecutive tiles of the ceil(H / 2) x 23 grid of one slab
    (csrc/conv_wino.hip: GEOM 1, maps whose width is a multiple of 46)."""
    assert W % (2 * RUN_TX) == 0
    return ((H + 1) // 2) * RUN_TX // 32 * 32


def wino_merged_tail_blocks(B, H, W, ks):
    """Geometry of the merged-tail launch (csrc/conv_wino.hip: GEOM 3; csrc/pmx_common.h::wino_tail_mergeable): the part-filled last
    blocks of all B images as one stream -- image b's tail tile t sits at position b * nt + t, block j owns the positions [32 j, 32 j + 32).
    Returns None when the tails are not mergeable (then every image keeps its own part-filled block), else a list of blocks, each a list of
    segments (image, first tail tile, tiles, first halo column): a segment's halo is 2 * tiles + ks - 1 columns wide, the segments of a
    block lie side by side.  The arithmetic per tile does not depend on this placement (conv_wino(unit_from=...) restates it)."""
    ntiles = RUN_TX * ((H + 1) // 2)
    t0 = ntiles // 32 * 32
    nt = ntiles - t0
    if not (B >= 2 and W == 2 * RUN_TX and nt >= 16 and t0 % RUN_TX + nt <= RUN_TX):
        return None
    blocks = []
    for j in range((B * nt + 31) // 32):
        segs, col, p = [], 0, 32 * j
        while p < min(32 * j + 32, B * nt):
            img, tt = divmod(p, nt)
            n = min(nt - tt, 32 * j + 32 - p)
            segs.append((img, tt, n, col))
            col += 2 * n + ks - 1
            p += n
        blocks.append(segs)
    return blocks


def conv_wino(x, w, b, relu=False, pool=False, unit_g=0, unit_from=0, run_tx=None):
    """The Winograd F(2x2, 3x3) kernel's arithmetic (csrc/conv_wino.hip::conv_wino_kernel, option "conv_algo"): same shapes as
    conv_fma, 3x3 or 7x7 (four 3x3 sub-kernels in the frequency domain + row 6 / column 6 as 1-D sub-kernels + tap (6, 6)).  Defined order, but not the direct
    kernels' chain: the two agree to ~1e-6 of the map scale.  unit_g > 0 (7x7): the kernel's unit mode for single images -- pass 1 in
    units of unit_g 32-channel chunks, pass 2a, pass 2b, each summed from 0 and added in that order (kernel label ".../u<g>").
    unit_from: only the tiles with row-major index >= unit_from are summed that way (the part-filled last block of every (image, slab)
    in the run geometry, kernel label "...r/t<g>": unit_from = wino_run_unit_from(H, W), the index then runs inside a 23-column slab:
    run_tx, default 23 when unit_from is given on a map whose width is a multiple of 46)."""
    global _lib
    if _lib is None:
        conv_fma(np.zeros((1, 1, 1, 1), 'f'), np.zeros((1, 1, 1, 1), 'f'), np.zeros(1, 'f'))
    _lib.conv_wino_ref2.restype = None
    _lib.conv_wino_ref2.argtypes = [C.c_void_p] * 4 + [C.c_int] * 11
    x = np.ascontiguousarray(x, np.float32)
    w = np.ascontiguousarray(w, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    B, cin, H, W = x.shape
    cout, _, ks, _ = w.shape
    assert ks in (3, 7) and not (pool and ks == 7)
    y = np.empty((B, cout, H // 2 if pool else H, W // 2 if pool else W), np.float32)
    if run_tx is None:
        run_tx = RUN_TX if (unit_from and W % (2 * RUN_TX) == 0) else 0
    _lib.conv_wino_ref2(x.ctypes.data, w.ctypes.data, b.ctypes.data, y.ctypes.data, B, cin, H, W, cout, ks, int(relu), int(pool), int(unit_g),
                        int(unit_from), int(run_tx))
    return y


# ---- the whole network in the kernels' order ---------------------------------------------------------------------------
# Stage inputs are the "cat" buffer of the HIP path: [feature 0..127 | PAF 128..165 | 0, 0 | heat 168..186 | 0 x 5]
# (csrc/pmx_api.hip::concat_map); the reference concatenates (PAF, heat, feature) (models/CocoPoseNet.py:168).  K is walked
# in buffer order, so the Mconv1 weights are permuted (and zero-padded) accordingly.
CAT_C, CAT_PAF, CAT_HEAT = 192, 128, 168


def _cat_weights(W):
    Wp = np.zeros((W.shape[0], CAT_C) + W.shape[2:], np.float32)
    Wp[:, 0:128] = W[:, 57:185]
    Wp[:, CAT_PAF:CAT_PAF + 38] = W[:, 0:38]
    Wp[:, CAT_HEAT:CAT_HEAT + 19] = W[:, 38:57]
    return Wp


class LaunchPlan(dict):
    """{layer label: K slices} of the launches that were split, plus `.wino` = labels of the layers that ran on the Winograd kernel
    and `.wino_units` = {label: chunks per pass-1 unit} of those that ran in its unit mode, `.wino_tails` = {label: chunks per pass-1
    unit} of those that ran the part-filled last block of every image in unit mode (run geometry, label "...r/t<g>")."""
    wino = frozenset()
    wino_units = {}
    wino_tails = {}


def _for_image(profile, image):
    """Profile entries that describe how `image` ran.  A batch whose plain launch would end in a part-filled round of the CUs is cut in two
    by images (csrc/conv_select.hip::wino_split_images); the two halves of such a layer carry "@<first image>+<count>" at the end of their
    kernel label and may run different launch forms -- the plan of a forward is then a plan PER IMAGE.  Returns copies with the suffix
    removed; entries of the other half are dropped."""
    import re
    out, split = [], False
    for e in profile:
        m = re.search(r'@(\d+)\+(\d+)$', e['kernel'])
        if not m:
            out.append(e)
            continue
        split = True
        a, n = int(m.group(1)), int(m.group(2))
        if image is not None and a <= image < a + n:
            out.append(dict(e, kernel=e['kernel'][:m.start()]))
    if split and image is None:
        raise ValueError('this forward cut a batch in two by images: ask for the plan of one image (splitk_plan(profile, image=i))')
    return out


def splitk_plan(profile, image=None):
    """Launch plan of one forward from an engine profile of the SAME forward (native.Engine.profile()): {layer label: chunks of
    every K slice} (the kernel label carries "/k3-2-2-1" where the launch was split) and, as attribute `.wino`, the layers that
    ran on the Winograd kernel (the kernel choice depends on the launch size).  Labels are the layer names without the _L1 / _L2
    branch suffix.  forward_fma(weights, x, splitk=plan) restates exactly that forward.  `image`: which image of the batch the plan is
    for -- needed (only) when the forward cut a batch in two by images (_for_image)."""
    profile = _for_image(profile, image)
    plan = LaunchPlan()
    plan.wino_units, plan.wino_tails = {}, {}
    for e in profile:
        k = e['kernel']
        if '/k' in k:
            plan[e['layer']] = [int(v) for v in k.rsplit('/k', 1)[1].split('-')]
    plan.wino = frozenset(wino_layers(profile))
    import re
    for e in profile:
        if e['kernel'].startswith('conv_wino'):
            m = re.search(r'/([ut])(\d+)', e['kernel'])          # ".../u<g>": unit mode; "...r/t<g>[:units|:combine]": run geometry, tail in units
            if m:
                (plan.wino_units if m.group(1) == 'u' else plan.wino_tails)[e['layer']] = int(m.group(2))
    return plan


def wino_layers(profile):
    """Layer labels the engine ran with the Winograd kernel (kernel label "conv_wino_f2x2_3x3" in an engine profile).  The fused
    conv1_1 + conv1_2 launch with conv1_2 in Winograd form (label "conv_wino1_...", csrc/conv1_wino.hip) counts as `conv1_2`: conv1_1
    keeps the direct chain inside it."""
    out = {e['layer'] for e in profile if e['kernel'].startswith('conv_wino')}
    if 'conv1_1+conv1_2' in out:
        out.discard('conv1_1+conv1_2')
        out.add('conv1_2')
    return out


def forward_fma(weights, x, splitk=None, wino=()):
    """CocoPoseNet forward (models/CocoPoseNet.py:132-262) with every convolution in the HIP kernels' summation order.
    x: (B, 3, H, W) float32 as produced by preprocess; returns (paf (B,38,h,w), heat (B,19,h,w)) of the last stage.
    splitk: {layer label: K slices} of the launch plan the kernels used (splitk_plan); None = no launch was split (large
    batches).  wino: labels of the layers that ran as Winograd F(2x2, 3x3) (wino_layers)."""
    if not wino:
        wino = getattr(splitk, 'wino', ())
    units = getattr(splitk, 'wino_units', {})
    tails = getattr(splitk, 'wino_tails', {})
    splitk = splitk or {}

    def conv(name, h, relu=True, pool=False, cat=False):
        W, b = weights[name]
        label = name[:-3] if name.endswith(('_L1', '_L2')) else name
        if label in wino:
            if label in tails:
                return conv_wino(h, _cat_weights(W) if cat else W, b, relu=relu, pool=pool, unit_g=tails[label],
                                 unit_from=wino_run_unit_from(h.shape[2], h.shape[3]))
            return conv_wino(h, _cat_weights(W) if cat else W, b, relu=relu, pool=pool, unit_g=units.get(label, 0))
        return conv_fma(h, _cat_weights(W) if cat else W, b, relu=relu, pool=pool, splitk=splitk.get(label, 1))
    h = conv('conv1_1', x); h = conv('conv1_2', h, pool=True)
    h = conv('conv2_1', h); h = conv('conv2_2', h, pool=True)
    h = conv('conv3_1', h); h = conv('conv3_2', h); h = conv('conv3_3', h); h = conv('conv3_4', h, pool=True)
    h = conv('conv4_1', h); h = conv('conv4_2', h); h = conv('conv4_3_CPM', h); feat = conv('conv4_4_CPM', h)
    h1, h2 = feat, feat
    for i in range(1, 5):
        h1 = conv('conv5_%d_CPM_L1' % i, h1)
        h2 = conv('conv5_%d_CPM_L2' % i, h2)
    h1 = conv('conv5_5_CPM_L1', h1, relu=False)
    h2 = conv('conv5_5_CPM_L2', h2, relu=False)
    B, _, fh, fw = feat.shape
    for s in range(2, 7):
        cat = np.zeros((B, CAT_C, fh, fw), np.float32)
        cat[:, 0:128] = feat
        cat[:, CAT_PAF:CAT_PAF + 38] = h1
        cat[:, CAT_HEAT:CAT_HEAT + 19] = h2
        h1 = conv('Mconv1_stage%d_L1' % s, cat, cat=True)
        h2 = conv('Mconv1_stage%d_L2' % s, cat, cat=True)
        for i in range(2, 7):
            h1 = conv('Mconv%d_stage%d_L1' % (i, s), h1)
            h2 = conv('Mconv%d_stage%d_L2' % (i, s), h2)
        h1 = conv('Mconv7_stage%d_L1' % s, h1, relu=False)
        h2 = conv('Mconv7_stage%d_L2' % s, h2, relu=False)
    return h1, h2
