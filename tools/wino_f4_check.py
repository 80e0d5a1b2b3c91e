#!/usr/bin/env python
"""F(4x4, 3x3) go / no-go for the 3x3 stem, NUMERICS FIRST -- CPU only, no kernel exists.

The 3x3 layers are 9.5 ms of the 37.6 ms step at 16 / 36 of the direct multiplies (F(2x2, 3x3)); F(4x4, 3x3) would issue 36 / 144.
Before any kernel: what does it cost in accuracy, in the arithmetic a matrix-core kernel would have (tools/proto/wino_fmn_proto.c:
float32 transforms in a fixed order, one sequential fmaf chain over the input channels per frequency)?

For every 3x3 layer conv2_1 ... conv5_3 of the network (real activations of one synthetic frame, seeded weights) the max error against a
float64 convolution of the same float32 input, relative to the map scale, for
    direct    the direct kernels' fp32 FMA chain            (oracle/conv_fma_ref.c, bit-identical to the HIP kernels)
    F(2x2)    the Winograd kernel shipped today             (oracle/conv_fma_ref.c::conv_wino_ref, bit-identical to the HIP kernel)
    F(4x4)    the prototype, three variants: Lavin's points {0, +-1, +-2, inf}; the points {0, +-1, +-1/2, inf}; Lavin's points with
              the input transform computed in float64 and rounded once
Go rule (VERDICT): per-layer error <= 2 x the direct chain's.  Output: profiles/r04_wino_f4_check.json."""
import argparse, ctypes as C, importlib, json, os, subprocess, sys
from fractions import Fraction
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = 'chainer_realtime_multi-person_pose_estimation_amd'


def cook_toom(m, r, points):
    """Winograd F(m, r) matrices for the given finite points (+ infinity): returns A^T (m x n), G (n x r), B^T (n x n) as float64
    arrays (exact rationals, converted at the end).  Construction: Toom-Cook with the Vandermonde of the points; B^T from the inverse."""
    n = m + r - 1
    pts = [Fraction(p) for p in points]
    assert len(pts) == n - 1
    # A^T[i][j] = p_j ^ i (last column: infinity -> 1 only in the last row)
    AT = [[(pts[j] ** i if j < n - 1 else (Fraction(1) if i == m - 1 else Fraction(0))) for j in range(n)] for i in range(m)]
    # G[j][k] = p_j ^ k / N_j,  N_j = prod_{l != j} (p_j - p_l); infinity row: (0, .., 1)
    G = []
    for j in range(n - 1):
        N = Fraction(1)
        for l in range(n - 1):
            if l != j:
                N *= (pts[j] - pts[l])
        G.append([pts[j] ** k / N for k in range(r)])
    G.append([Fraction(0)] * (r - 1) + [Fraction(1)])
    # B^T: rows = coefficients of the polynomials  M_j(x) = prod_{l != j} (x - p_l)  (j < n-1),  last row = M(x) = prod_l (x - p_l)
    def polymul(a, b):
        out = [Fraction(0)] * (len(a) + len(b) - 1)
        for i, u in enumerate(a):
            for k, v in enumerate(b):
                out[i + k] += u * v
        return out
    BT = []
    for j in range(n - 1):
        poly = [Fraction(1)]
        for l in range(n - 1):
            if l != j:
                poly = polymul(poly, [-pts[l], Fraction(1)])
        BT.append(poly + [Fraction(0)] * (n - len(poly)))
    poly = [Fraction(1)]
    for l in range(n - 1):
        poly = polymul(poly, [-pts[l], Fraction(1)])
    BT.append(poly)
    f = lambda M_: np.array([[float(v) for v in row] for row in M_], dtype=np.float64)
    return f(AT), f(G), f(BT)


def check_1d(AT, G, BT, m, r):
    rng = np.random.default_rng(0)
    d, g = rng.standard_normal(m + r - 1), rng.standard_normal(r)
    y = AT @ ((G @ g) * (BT @ d))
    ref = np.array([sum(d[i + k] * g[k] for k in range(r)) for i in range(m)])
    return float(np.abs(y - ref).max())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--size', type=int, default=368)
    ap.add_argument('--out', default=os.path.join(ROOT, 'profiles', 'r04_wino_f4_check.json'))
    a = ap.parse_args()
    import torch
    from oracle import conv_fma_ref as R, network_ref as N, postprocess_ref as P
    W_ = importlib.import_module(PKG + '.weights')
    so = os.path.join(ROOT, 'tools', 'proto', '_wino_fmn_proto.so')
    src = os.path.join(ROOT, 'tools', 'proto', 'wino_fmn_proto.c')
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(['gcc', '-O2', '-mfma', '-fopenmp', '-ffp-contract=off', '-shared', '-fPIC', '-o', so, src, '-lm'])
    lib = C.CDLL(so)
    lib.wino_fmn_proto.restype = None
    lib.wino_fmn_proto.argtypes = [C.c_void_p] * 4 + [C.c_int] * 6 + [C.c_void_p, C.c_void_p, C.c_int]

    variants = {}
    for name, pts in (('F4_lavin_0_1_-1_2_-2', [0, 1, -1, 2, -2]), ('F4_half_0_1_-1_1/2_-1/2', [0, 1, -1, Fraction(1, 2), Fraction(-1, 2)])):
        AT, G, BT = cook_toom(4, 3, pts)
        assert check_1d(AT, G, BT, 4, 3) < 1e-12, name
        variants[name] = (AT, G, BT)
    AT2, G2, BT2 = cook_toom(2, 3, [0, 1, -1])
    assert check_1d(AT2, G2, BT2, 2, 3) < 1e-12

    def proto(x, w, b, AT, G, BT, m, tf64=0):
        cout, cin = w.shape[:2]
        n = m + 2
        U = np.einsum('ik,ockl,jl->ijoc', G, w.astype(np.float64), G).astype(np.float32).reshape(n * n, cout, cin)     # G g G^T in double, rounded once
        U = np.ascontiguousarray(U)
        x = np.ascontiguousarray(x[0], np.float32)
        y = np.zeros((cout,) + x.shape[1:], np.float32)
        BTc, ATc = np.ascontiguousarray(BT), np.ascontiguousarray(AT)
        bb = np.ascontiguousarray(b, np.float32)
        lib.wino_fmn_proto(x.ctypes.data, U.ctypes.data, BTc.ctypes.data, ATc.ctypes.data, m, cin, x.shape[1], x.shape[2], cout, 1, bb.ctypes.data,
                           y.ctypes.data, tf64)
        return y[None]

    weights = W_.synthetic_weights(0)
    img = np.random.default_rng(1).integers(0, 256, (a.size, a.size, 3), dtype=np.uint8)
    h = torch.from_numpy(P.preprocess(img)).double()
    F = torch.nn.functional
    layers = ['conv1_1', 'conv1_2', 'conv2_1', 'conv2_2', 'conv3_1', 'conv3_2', 'conv3_3', 'conv3_4', 'conv4_1', 'conv4_2', 'conv4_3_CPM', 'conv4_4_CPM',
              'conv5_1_CPM_L1', 'conv5_2_CPM_L1', 'conv5_3_CPM_L1']
    pool_after = {'conv1_2', 'conv2_2', 'conv3_4'}
    rows = []
    for name in layers:
        Wt, bt = weights[name]
        x32 = h.float().numpy()                                # the layer's real input, as float32 (what the kernels see)
        with torch.no_grad():
            ref = torch.relu(F.conv2d(torch.from_numpy(x32).double(), torch.from_numpy(Wt).double(), torch.from_numpy(bt).double(), padding=1))
        if name not in ('conv1_1', 'conv1_2'):
            scale = float(ref.abs().max())
            r = ref.numpy()
            row = {'layer': name, 'cin': int(Wt.shape[1]), 'cout': int(Wt.shape[0]), 'map': list(x32.shape[2:]), 'scale': scale}
            row['direct'] = float(np.abs(R.conv_fma(x32, Wt, bt, relu=True) - r).max() / scale)
            row['F2_kernel_twin'] = float(np.abs(R.conv_wino(x32, Wt, bt, relu=True) - r).max() / scale)
            row['F2_proto'] = float(np.abs(proto(x32, Wt, bt, AT2, G2, BT2, 2) - r).max() / scale)
            for vn, (AT, G, BT) in variants.items():
                row[vn] = float(np.abs(proto(x32, Wt, bt, AT, G, BT, 4) - r).max() / scale)
            AT, G, BT = variants['F4_lavin_0_1_-1_2_-2']
            row['F4_lavin_tf64'] = float(np.abs(proto(x32, Wt, bt, AT, G, BT, 4, tf64=1) - r).max() / scale)
            row['best_F4_over_direct'] = min(row[k] for k in row if k.startswith('F4_')) / row['direct']
            rows.append(row)
            print('%-16s %4d->%4d %3dx%-3d  direct %.2e  F(2x2) twin %.2e proto %.2e | F(4x4) lavin %.2e  half %.2e  lavin+f64 transform %.2e  | best F4 / direct = %.1f'
                  % (name, row['cin'], row['cout'], x32.shape[2], x32.shape[3], row['direct'], row['F2_kernel_twin'], row['F2_proto'],
                     row['F4_lavin_0_1_-1_2_-2'], row['F4_half_0_1_-1_1/2_-1/2'], row['F4_lavin_tf64'], row['best_F4_over_direct']), flush=True)
        h = ref
        if name in pool_after:
            h = F.max_pool2d(h, 2, 2)
    # a 7x7 layer built from F(4x4, 3x3) sub-kernels: taps (0..5, 0..5) as four 3x3 sub-kernels through the prototype (each rounded on its
    # own and summed -- an upper bound of accumulating them in the transform domain), row 6 / column 6 / tap (6, 6) by the direct chain
    rng = np.random.default_rng(0)
    cin, cout, Hs = 128, 128, 46
    x7 = np.maximum(rng.standard_normal((1, cin, Hs, Hs)), 0).astype('f')
    w7 = (rng.standard_normal((cout, cin, 7, 7)) / np.sqrt(cin * 49)).astype('f')
    b7 = np.zeros(cout, 'f')
    with torch.no_grad():
        ref7 = F.conv2d(torch.from_numpy(x7).double(), torch.from_numpy(w7).double(), padding=3).numpy()[0]
    sc7 = float(np.abs(ref7).max())
    AT, G, BT = variants['F4_lavin_0_1_-1_2_-2']
    xp = np.zeros((cin, Hs + 6, Hs + 6), np.float32)
    xp[:, 3:3 + Hs, 3:3 + Hs] = x7[0]
    acc = np.zeros((cout, Hs, Hs), np.float32)
    for sy in (0, 3):
        for sx in (0, 3):
            sub = np.ascontiguousarray(xp[:, sy:sy + Hs + 2, sx:sx + Hs + 2])[None]
            yy = proto(sub, np.ascontiguousarray(w7[:, :, sy:sy + 3, sx:sx + 3]), b7, AT, G, BT, 4)[0][:, 1:1 + Hs, 1:1 + Hs]
            # (proto applies ReLU; undo is impossible -- run it on the negated input too and combine: relu(v) - relu(-v) = v)
            yn = proto(-sub, np.ascontiguousarray(w7[:, :, sy:sy + 3, sx:sx + 3]), b7, AT, G, BT, 4)[0][:, 1:1 + Hs, 1:1 + Hs]
            acc = acc + (yy - yn)
    wr = np.zeros_like(w7)
    wr[:, :, 6, :] = w7[:, :, 6, :]
    wr[:, :, :, 6] = w7[:, :, :, 6]
    tot = acc + R.conv_fma(x7, wr, b7)[0]
    seven = {'shape': '128 -> 128, 46 x 46, dense post-ReLU Gaussian input', 'direct': float(np.abs(R.conv_fma(x7, w7, b7)[0] - ref7).max() / sc7),
             'F2_kernel_twin': float(np.abs(R.conv_wino(x7, w7, b7)[0] - ref7).max() / sc7), 'F4_composed': float(np.abs(tot - ref7).max() / sc7)}
    seven['F4_over_direct'] = seven['F4_composed'] / seven['direct']
    print('7x7 128->128: direct %.2e  F(2x2) twin %.2e  F(4x4)-composed %.2e  (%.1f x direct)' % (seven['direct'], seven['F2_kernel_twin'], seven['F4_composed'], seven['F4_over_direct']))
    worst = max(r_['best_F4_over_direct'] for r_ in rows)
    out = {'input': '%dx%d synthetic frame, seeded He weights; every layer sees its real (float64-propagated, float32-rounded) input' % (a.size, a.size),
           'metric': 'max |y - y_float64| / max |y_float64| per layer (ReLU applied)', 'rows': rows,
           'go_rule': 'per-layer error of F(4x4, 3x3) <= 2 x the direct fp32 chain', 'worst_best_F4_over_direct': worst, 'go': bool(worst <= 2.0),
           'seven_by_seven_from_F4_sub_kernels': seven}
    json.dump(out, open(a.out, 'w'), indent=1)
    print('worst (best F(4x4) variant) / direct over the layers: %.1f  ->  %s' % (worst, 'GO' if out['go'] else 'NO-GO'))


if __name__ == '__main__':
    main()
