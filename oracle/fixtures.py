"""TEST INFRASTRUCTURE -- synthetic skeleton fixtures (heatmaps / PAFs) for the post-process tests.

The reference ships no tests or golden vectors, and no trained weights are
available offline, so realistic post-process inputs are rendered from synthetic skeletons with the
reference's own label formulas:

  render_heatmaps <- coco_data_loader.py:208-229 (`generate_gaussian_heatmap`, `generate_heatmaps`:
                     per joint type max over people of exp(-d^2 / 2 sigma^2); last channel = 1 - max)
  render_pafs     <- coco_data_loader.py:232-268 (`generate_constant_paf`, `generate_pafs`: constant unit
                     vector inside a band of half-width `paf_width` around the limb segment, overlaps averaged)

tests/test_oracle_vs_reference.py checks these restatements against the reference's generator
(imported verbatim in the authoring container).  Used here and on the GPU box (pure NumPy).
"""
import numpy as np

from .postprocess_ref import LIMBS_POINT

# unit skeleton (x, y), nose at origin, body height ~ 1.0; joint order = JointType (entity.py:9-45)
_TEMPLATE = np.array([
    [0.00, 0.00],    # Nose
    [0.00, 0.16],    # Neck
    [-0.16, 0.17],   # RightShoulder
    [-0.21, 0.37],   # RightElbow
    [-0.23, 0.56],   # RightHand
    [0.16, 0.17],    # LeftShoulder
    [0.21, 0.37],    # LeftElbow
    [0.23, 0.56],    # LeftHand
    [-0.10, 0.56],   # RightWaist
    [-0.11, 0.80],   # RightKnee
    [-0.11, 1.02],   # RightFoot
    [0.10, 0.56],    # LeftWaist
    [0.11, 0.80],    # LeftKnee
    [0.11, 1.02],    # LeftFoot
    [-0.035, -0.035],  # RightEye
    [0.035, -0.035],   # LeftEye
    [-0.08, -0.01],  # RightEar
    [0.08, -0.01],   # LeftEar
])


def random_poses(rng, n_people, H, W, height_range=(0.45, 0.8), drop_prob=0.1, jitter=0.02,
                 integer_coords=False):
    """(n_people, 18, 3) float64 poses [x, y, v] (v in {0, 2}) inside an H x W map."""
    poses = []
    for _ in range(n_people):
        hgt = rng.uniform(*height_range) * H
        cx = rng.uniform(0.15 * W, 0.85 * W)
        top = rng.uniform(0.05 * H, max(0.06 * H, 0.92 * H - 1.05 * hgt))
        ang = rng.uniform(-0.35, 0.35)
        rot = np.array([[np.cos(ang), -np.sin(ang)], [np.sin(ang), np.cos(ang)]])
        pts = (_TEMPLATE + rng.normal(0, jitter, _TEMPLATE.shape)) @ rot.T * hgt
        pts[:, 0] += cx
        pts[:, 1] += top + 0.05 * hgt
        vis = (rng.uniform(size=18) > drop_prob).astype(np.float64) * 2
        inside = (pts[:, 0] >= 1) & (pts[:, 0] <= W - 2) & (pts[:, 1] >= 1) & (pts[:, 1] <= H - 2)
        vis[~inside] = 0
        if integer_coords:
            pts = np.round(pts)
        poses.append(np.concatenate([pts, vis[:, None]], axis=1))
    return np.array(poses).reshape(-1, 18, 3)


def render_heatmaps(shape_hw, poses, sigma):
    """-> (19, H, W) float32 (coco_data_loader.py:216-229)."""
    H, W = shape_hw
    gx = np.tile(np.arange(W), (H, 1))
    gy = np.tile(np.arange(H), (W, 1)).transpose()
    maps = np.zeros((19, H, W))
    total = np.zeros((H, W))
    for j in range(18):
        hm = np.zeros((H, W))
        for pose in poses:
            if pose[j, 2] > 0:
                d2 = (gx - pose[j, 0]) ** 2 + (gy - pose[j, 1]) ** 2
                g = np.exp(-0.5 * d2 / sigma ** 2)
                hm[g > hm] = g[g > hm]
                total[g > total] = g[g > total]
        maps[j] = hm
    maps[18] = 1 - total
    return maps.astype('f')


def render_pafs(shape_hw, poses, paf_width):
    """-> (38, H, W) float32 (coco_data_loader.py:232-268)."""
    H, W = shape_hw
    gx = np.tile(np.arange(W), (H, 1))
    gy = np.tile(np.arange(H), (W, 1)).transpose()
    out = np.zeros((38, H, W))
    for li, (ja, jb) in enumerate(LIMBS_POINT):
        paf = np.zeros((2, H, W))
        flags = np.zeros((2, H, W))
        for pose in poses:
            a, b = pose[ja], pose[jb]
            if a[2] > 0 and b[2] > 0:
                if np.array_equal(a[:2], b[:2]):
                    continue
                dist = np.linalg.norm(b[:2] - a[:2])
                unit = (b[:2] - a[:2]) / dist
                rad = np.pi / 2
                rotm = np.array([[np.cos(rad), np.sin(rad)], [-np.sin(rad), np.cos(rad)]])
                vert = np.dot(rotm, unit)
                hor_ip = unit[0] * (gx - a[0]) + unit[1] * (gy - a[1])
                hor_flag = (0 <= hor_ip) & (hor_ip <= dist)
                ver_ip = vert[0] * (gx - a[0]) + vert[1] * (gy - a[1])
                ver_flag = np.abs(ver_ip) <= paf_width
                flag = hor_flag & ver_flag
                limb = np.stack((flag, flag)) * unit[:, None, None]
                lf = limb != 0
                flags += np.broadcast_to(lf[0] | lf[1], limb.shape)
                paf += limb
        paf[flags > 0] /= flags[flags > 0]
        out[2 * li:2 * li + 2] = paf
    return out.astype('f')


def synthetic_maps(seed, n_people, H, W, sigma, paf_width, noise=0.0, **pose_kw):
    """Seeded (heat (19,H,W) f32, paf (38,H,W) f32, poses) triple."""
    rng = np.random.default_rng(seed)
    poses = random_poses(rng, n_people, H, W, **pose_kw)
    heat = render_heatmaps((H, W), poses, sigma)
    paf = render_pafs((H, W), poses, paf_width)
    if noise > 0:
        heat = (heat + rng.normal(0, noise, heat.shape)).astype('f')
        paf = (paf + rng.normal(0, noise, paf.shape)).astype('f')
    return heat, paf, poses
