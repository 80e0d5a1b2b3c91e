"""Weights of the pose network: layer table, Chainer-NPZ reader/writer, seeded synthetic weights.

Layer names, shapes and order follow the reference `models/CocoPoseNet.py:26-129`; the on-disk format is
the one `chainer.serializers.save_npz` writes for that Chain (`models/convert_model.py:281`) and
`serializers.load_npz` reads at `pose_detector.py:26`: one array `<layer>/W` (float32, OIHW) and one
`<layer>/b` (float32) per layer.

No trained weights can be fetched offline (they are `wget`-ed in the reference README), so tests and
bench use `synthetic_weights(seed)`: He-scaled Gaussian weights, which keep activations O(1) through
all 92 layers; `calibrate_head` then rescales the last (affine) PAF / heat-map layers so that a realistic
(tens, not thousands) number of peaks survives the 0.05 threshold.
"""
import numpy as np

N_LAYERS = 92


def cpm_layer_table(n_maps):
    """FaceNet (71 maps, models/FaceNet.py:12-75) / HandNet (22 maps, models/HandNet.py): VGG-19 stem to conv5_2,
    conv5_3_CPM, one-branch 6-stage CPM on concat((heat, feature))."""
    t = [('conv1_1', 3, 64, 3), ('conv1_2', 64, 64, 3), ('conv2_1', 64, 128, 3), ('conv2_2', 128, 128, 3),
         ('conv3_1', 128, 256, 3), ('conv3_2', 256, 256, 3), ('conv3_3', 256, 256, 3), ('conv3_4', 256, 256, 3),
         ('conv4_1', 256, 512, 3), ('conv4_2', 512, 512, 3), ('conv4_3', 512, 512, 3), ('conv4_4', 512, 512, 3),
         ('conv5_1', 512, 512, 3), ('conv5_2', 512, 512, 3), ('conv5_3_CPM', 512, 128, 3),
         ('conv6_1_CPM', 128, 512, 1), ('conv6_2_CPM', 512, n_maps, 1)]
    for s in range(2, 7):
        t.append(('Mconv1_stage%d' % s, n_maps + 128, 128, 7))
        for i in range(2, 6):
            t.append(('Mconv%d_stage%d' % (i, s), 128, 128, 7))
        t.append(('Mconv6_stage%d' % s, 128, 128, 1))
        t.append(('Mconv7_stage%d' % s, 128, n_maps, 1))
    return t


def layer_table(arch='posenet'):
    """[(name, cin, cout, ksize)] -- posenet: 92 convolutions, reference declaration order."""
    if arch == 'facenet':
        return cpm_layer_table(71)
    if arch == 'handnet':
        return cpm_layer_table(22)
    t = [
        ('conv1_1', 3, 64, 3), ('conv1_2', 64, 64, 3),
        ('conv2_1', 64, 128, 3), ('conv2_2', 128, 128, 3),
        ('conv3_1', 128, 256, 3), ('conv3_2', 256, 256, 3), ('conv3_3', 256, 256, 3),
        ('conv3_4', 256, 256, 3),
        ('conv4_1', 256, 512, 3), ('conv4_2', 512, 512, 3), ('conv4_3_CPM', 512, 256, 3),
        ('conv4_4_CPM', 256, 128, 3),
    ]
    for br, co in (('L1', 38), ('L2', 19)):
        for i in (1, 2, 3):
            t.append(('conv5_%d_CPM_%s' % (i, br), 128, 128, 3))
        t.append(('conv5_4_CPM_' + br, 128, 512, 1))
        t.append(('conv5_5_CPM_' + br, 512, co, 1))
    for s in range(2, 7):
        for br, co in (('L1', 38), ('L2', 19)):
            t.append(('Mconv1_stage%d_%s' % (s, br), 185, 128, 7))
            for i in range(2, 6):
                t.append(('Mconv%d_stage%d_%s' % (i, s, br), 128, 128, 7))
            t.append(('Mconv6_stage%d_%s' % (s, br), 128, 128, 1))
            t.append(('Mconv7_stage%d_%s' % (s, br), 128, co, 1))
    assert len(t) == N_LAYERS
    return t


def n_params(arch='posenet'):
    return sum(co * ci * k * k + co for _, ci, co, k in layer_table(arch))


def synthetic_weights(seed=0, arch='posenet'):
    """{name: (W OIHW float32, b float32)} -- deterministic for a given seed (numpy PCG64)."""
    rng = np.random.default_rng(seed)
    out = {}
    for name, ci, co, k in layer_table(arch):
        std = np.sqrt(2.0 / (ci * k * k))
        W = (rng.standard_normal((co, ci, k, k), dtype=np.float32) * np.float32(std))
        b = (rng.standard_normal(co, dtype=np.float32) * np.float32(0.01))
        out[name] = (np.ascontiguousarray(W, dtype=np.float32), np.ascontiguousarray(b, dtype=np.float32))
    return out


def load_npz(path, arch='posenet'):
    """Chainer NPZ (`<layer>/W`, `<layer>/b`) -> {name: (W, b)}; every layer of the architecture must be present."""
    out = {}
    with np.load(path) as z:
        for name, ci, co, k in layer_table(arch):
            W = np.ascontiguousarray(z[name + '/W'], dtype=np.float32)
            b = np.ascontiguousarray(z[name + '/b'], dtype=np.float32)
            if W.shape != (co, ci, k, k) or b.shape != (co,):
                raise ValueError('%s: expected W %s b %s, file has %s %s'
                                 % (name, (co, ci, k, k), (co,), W.shape, b.shape))
            out[name] = (W, b)
    return out


def save_npz(path, weights):
    flat = {}
    for name, (W, b) in weights.items():
        flat[name + '/W'] = W
        flat[name + '/b'] = b
    np.savez(path, **flat)


def calibrate_head(weights, paf_raw, heat_raw, heat_s=0.1, heat_t=-0.15, paf_s=0.5):
    """Rescale the LAST (linear, 1x1) PAF / heat-map layers of a synthetic weight set so that its outputs on the
    calibration image have, per channel, mean `heat_t` / 0 and spatial std `heat_s` / `paf_s`.

    paf_raw (38, h, w), heat_raw (19, h, w): last-stage outputs of the network with `weights` on one image
    (from any engine).  He-initialised random weights give per-channel offsets that dwarf the spatial variation, so
    without this either no or thousands of peaks pass the 0.05 threshold; after it a COCO-crowd-like ~8 peaks per
    joint type survive.  Exact because the last layers are affine: y' = (y - mean) * s / std + t."""
    out = dict(weights)
    for name, raw, s, t in (('Mconv7_stage6_L2', heat_raw, heat_s, heat_t), ('Mconv7_stage6_L1', paf_raw, paf_s, 0.0)):
        W, b = weights[name]
        C = W.shape[0]
        mean = raw.reshape(C, -1).mean(1).astype(np.float64)
        std = raw.reshape(C, -1).std(1).astype(np.float64)
        g = s / np.maximum(std, 1e-12)
        out[name] = ((W * g[:, None, None, None]).astype(np.float32), ((b - mean) * g + t).astype(np.float32))
    return out
