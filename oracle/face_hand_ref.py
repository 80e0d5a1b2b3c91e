"""TEST INFRASTRUCTURE -- CPU restatement of the reference face / hand key-point path.  Only tests/ may import it.

  cpm_forward        <- models/FaceNet.py:78-160 (= models/HandNet.py with 22 maps): torch-CPU fp32, tolerance-level oracle
                        (Chainer not installable: PARITY UNPINNED at bit level, as for the pose network)
  detect             <- face_detector.py:28-40 / hand_detector.py:28-50: cv2.resize to 368 (restated, oracle/resize_ref.py),
                        x / 256 - 0.5, network, F.resize_images to the crop size (restated), key points
  compute_keypoints  <- face_detector.py:53-68 / hand_detector.py:63-78 CPU branch: gaussian_filter, max, threshold,
                        `np.array(np.where(heatmap == max_value)).flatten()` -> [coords[1], coords[0], max_value]
                        (verbatim semantics, including what it returns for tied maxima); the scalar comparison
                        `max_value > thresh` is evaluated in float64 (NumPy 1.x scalar promotion of the reference's era).
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import postprocess_ref as P
from . import resize_ref as RR


def layer_table(n_maps):
    t = [('conv1_1', 3, 64, 3), ('conv1_2', 64, 64, 3), ('conv2_1', 64, 128, 3), ('conv2_2', 128, 128, 3),
         ('conv3_1', 128, 256, 3), ('conv3_2', 256, 256, 3), ('conv3_3', 256, 256, 3), ('conv3_4', 256, 256, 3),
         ('conv4_1', 256, 512, 3), ('conv4_2', 512, 512, 3), ('conv4_3', 512, 512, 3), ('conv4_4', 512, 512, 3),
         ('conv5_1', 512, 512, 3), ('conv5_2', 512, 512, 3), ('conv5_3_CPM', 512, 128, 3),
         ('conv6_1_CPM', 128, 512, 1), ('conv6_2_CPM', 512, n_maps, 1)]
    for s in range(2, 7):
        t += [('Mconv1_stage%d' % s, n_maps + 128, 128, 7)]
        t += [('Mconv%d_stage%d' % (i, s), 128, 128, 7) for i in range(2, 6)]
        t += [('Mconv6_stage%d' % s, 128, 128, 1), ('Mconv7_stage%d' % s, 128, n_maps, 1)]
    return t


def cpm_forward(weights, x):
    """-> list of the six stage outputs (numpy, (B, n_maps, h/8, w/8))"""
    x = torch.as_tensor(np.ascontiguousarray(x)).float()

    def conv(name, h, relu=True):
        W, b = weights[name]
        W = torch.as_tensor(W)
        h = F.conv2d(h, W, torch.as_tensor(b), padding=W.shape[-1] // 2)
        return F.relu(h) if relu else h
    with torch.no_grad():
        h = conv('conv1_1', x); h = conv('conv1_2', h); h = F.max_pool2d(h, 2, 2)
        h = conv('conv2_1', h); h = conv('conv2_2', h); h = F.max_pool2d(h, 2, 2)
        for n in ('conv3_1', 'conv3_2', 'conv3_3', 'conv3_4'):
            h = conv(n, h)
        h = F.max_pool2d(h, 2, 2)
        for n in ('conv4_1', 'conv4_2', 'conv4_3', 'conv4_4', 'conv5_1', 'conv5_2', 'conv5_3_CPM'):
            h = conv(n, h)
        feat = h
        outs = []
        h = conv('conv6_1_CPM', h)
        h = conv('conv6_2_CPM', h, relu=False)
        outs.append(h)
        for s in range(2, 7):
            h = torch.cat((h, feat), dim=1)
            for i in range(1, 7):
                h = conv('Mconv%d_stage%d' % (i, s), h)
            h = conv('Mconv7_stage%d' % s, h, relu=False)
            outs.append(h)
    return [o.numpy() for o in outs]


def compute_keypoints(heatmaps, thresh):
    """heatmaps (n_maps, H, W) float32 already resized to the crop -> list of [x, y, conf] | None (last channel dropped)"""
    out = []
    for i in range(heatmaps.shape[0] - 1):
        hm = P.gaussian_filter_ref(heatmaps[i])
        max_value = hm.max()
        if float(max_value) > thresh:
            coords = np.array(np.where(hm == max_value)).flatten().tolist()
            out.append([coords[1], coords[0], max_value])
        else:
            out.append(None)
    return out


def detect(model, img, thresh, size=368, hand_type='right'):
    """model(x) -> last-stage maps (1, n_maps, h, w)."""
    img = np.asarray(img)
    if hand_type == 'left':
        img = img[:, ::-1]
    h, w, _ = img.shape
    resized = RR.resize_linear_u8(img, size, size)
    x = np.array(resized[np.newaxis], dtype=np.float32).transpose(0, 3, 1, 2) / 256 - 0.5
    maps = np.asarray(model(x), dtype=np.float32)[0]
    up = P.resize_images_ref(maps, h, w)
    if hand_type == 'left':
        up = up[:, :, ::-1]
    return compute_keypoints(np.ascontiguousarray(up), thresh), up
