"""TEST INFRASTRUCTURE -- torch-CPU fp32 restatement of the reference network.  NOT product code.

Restates `models/CocoPoseNet.py`:
  * layer table  <- CocoPoseNet.py:26-129 (92 x L.Convolution2D: cross-correlation, OIHW weights,
                    zero pad ksize//2, stride 1, bias)
  * forward      <- CocoPoseNet.py:132-262 (VGG-19 stem with 2x2/2 max-pool after conv1_2, conv2_2,
                    conv3_4; stage 1 two branches; stages 2-6 on concat((paf, heat, feature), axis=1))

PARITY UNPINNED at the bit level: Chainer is not installable offline and its CPU conv is
im2col + BLAS tensordot (summation order BLAS-defined), so no bit-level golden exists for the
convolutions; comparisons against this restatement are tolerance-based (see tests).  The
arithmetic definition (what is summed) is identical to L.Convolution2D.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import numpy as np
import torch
import torch.nn.functional as F


def layer_table():
    """[(name, cin, cout, ksize)] in reference declaration order (CocoPoseNet.py:26-129)."""
    t = [
        ('conv1_1', 3, 64, 3), ('conv1_2', 64, 64, 3),
        ('conv2_1', 64, 128, 3), ('conv2_2', 128, 128, 3),
        ('conv3_1', 128, 256, 3), ('conv3_2', 256, 256, 3), ('conv3_3', 256, 256, 3), ('conv3_4', 256, 256, 3),
        ('conv4_1', 256, 512, 3), ('conv4_2', 512, 512, 3), ('conv4_3_CPM', 512, 256, 3), ('conv4_4_CPM', 256, 128, 3),
    ]
    for br, co in (('L1', 38), ('L2', 19)):
        t += [('conv5_1_CPM_' + br, 128, 128, 3), ('conv5_2_CPM_' + br, 128, 128, 3),
              ('conv5_3_CPM_' + br, 128, 128, 3), ('conv5_4_CPM_' + br, 128, 512, 1),
              ('conv5_5_CPM_' + br, 512, co, 1)]
    for s in range(2, 7):
        for br, co in (('L1', 38), ('L2', 19)):
            t += [('Mconv1_stage%d_%s' % (s, br), 185, 128, 7)]
            t += [('Mconv%d_stage%d_%s' % (i, s, br), 128, 128, 7) for i in range(2, 6)]
            t += [('Mconv6_stage%d_%s' % (s, br), 128, 128, 1), ('Mconv7_stage%d_%s' % (s, br), 128, co, 1)]
    return t


def flops_per_frame(h=368, w=368):
    """2 x MACs of the 92 convs for an h x w input (bias/ReLU/pool excluded)."""
    total = 0
    for name, ci, co, k in layer_table():
        if name.startswith('conv1'):
            s = 1
        elif name.startswith('conv2'):
            s = 2
        elif name.startswith('conv3'):
            s = 4
        else:
            s = 8
        total += 2 * (h // s) * (w // s) * ci * co * k * k
    return total


def forward(weights, x, all_stages=False):
    """weights: {name: (W OIHW float32, b float32)}; x: (B, 3, H, W) float32 (torch or numpy).

    Returns (paf, heat) of the LAST stage as numpy (B,38,h,w), (B,19,h,w) -- the only outputs the
    hot path consumes (pose_detector.py:501-502) -- or all six stages if all_stages."""
    if not torch.is_tensor(x):
        x = torch.from_numpy(np.ascontiguousarray(x))
    x = x.float()

    def conv(name, h, relu=True):
        W, b = weights[name]
        W = torch.as_tensor(W)
        b = torch.as_tensor(b)
        h = F.conv2d(h, W, b, stride=1, padding=W.shape[-1] // 2)
        return F.relu(h) if relu else h

    with torch.no_grad():
        h = conv('conv1_1', x); h = conv('conv1_2', h); h = F.max_pool2d(h, 2, 2)
        h = conv('conv2_1', h); h = conv('conv2_2', h); h = F.max_pool2d(h, 2, 2)
        h = conv('conv3_1', h); h = conv('conv3_2', h); h = conv('conv3_3', h); h = conv('conv3_4', h)
        h = F.max_pool2d(h, 2, 2)
        h = conv('conv4_1', h); h = conv('conv4_2', h); h = conv('conv4_3_CPM', h); h = conv('conv4_4_CPM', h)
        feat = h
        outs = []
        h1, h2 = feat, feat
        for i in range(1, 5):
            h1 = conv('conv5_%d_CPM_L1' % i, h1)
            h2 = conv('conv5_%d_CPM_L2' % i, h2)
        h1 = conv('conv5_5_CPM_L1', h1, relu=False)
        h2 = conv('conv5_5_CPM_L2', h2, relu=False)
        outs.append((h1, h2))
        for s in range(2, 7):
            hc = torch.cat((h1, h2, feat), dim=1)
            h1, h2 = hc, hc
            for i in range(1, 7):
                h1 = conv('Mconv%d_stage%d_L1' % (i, s), h1)
                h2 = conv('Mconv%d_stage%d_L2' % (i, s), h2)
            h1 = conv('Mconv7_stage%d_L1' % s, h1, relu=False)
            h2 = conv('Mconv7_stage%d_L2' % s, h2, relu=False)
            outs.append((h1, h2))
    if all_stages:
        return [(a.numpy(), b.numpy()) for a, b in outs]
    return outs[-1][0].numpy(), outs[-1][1].numpy()


def conv2d_ref(x_nchw, W_oihw, b, relu=False, pool=False):
    """Single-layer torch-CPU reference used by the kernel unit tests (T0)."""
    with torch.no_grad():
        h = F.conv2d(torch.as_tensor(x_nchw).float(), torch.as_tensor(W_oihw).float(),
                     None if b is None else torch.as_tensor(b).float(), padding=W_oihw.shape[-1] // 2)
        if relu:
            h = F.relu(h)
        if pool:
            h = F.max_pool2d(h, 2, 2)
    return h.numpy()
