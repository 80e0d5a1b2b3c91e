"""TEST INFRASTRUCTURE -- near-tie margin census of the peak decisions (SURVEY.md section 4 T3 / section 7 "hard parts").  NOT product code.

The reference decides a peak with five strict float32 comparisons per pixel (reference `pose_detector.py:96-102`):

    smoothed > 0.05  and  smoothed > up  and  > down  and  > left  and  > right           (zero outside the map)

Write the decision as one number, the MARGIN of pixel (y, x) of a smoothed map s:

    m(s)[y, x] = min(s - thr, s - up, s - down, s - left, s - right)          peak  <=>  m > 0

m is 2-Lipschitz in the sup norm of s over the pixel's 5-point neighbourhood: two smoothed maps that differ by at most d there
can only disagree on a pixel whose margins satisfy |m_a| + |m_b| <= 2 d.  The GPU network and the CPU oracle network (and the
reference's own BLAS-ordered convolution) differ by fp32 summation order, ~1e-6 of the map scale -- so "integer peak indices
bit-exact" is a statement about fixtures whose margins are larger than that noise, and every disagreement must be a pixel whose
margin is inside it.  This module measures exactly that:

  margin_map(s)                       m for one smoothed map
  compare_frame(...)                  peak sets of the two sides, every disagreeing pixel with both margins, the test that decided
                                      it and the local |GPU - CPU| of the smoothed maps; the smallest margin among agreed peaks and
                                      among all agreed decisions of the frame
  summarize(frames)                   the census object quoted by bench.py / README / DESIGN (profiles/rNN_parity_census.json)

Used by tools/parity_census.py (the >= 512-frame census run on the GPU box) and tests/test_gpu_census.py.
"""
import numpy as np

from . import postprocess_ref as P

THR = np.float32(P.HEATMAP_PEAK_THRESH)
TESTS = ('threshold', 'up', 'down', 'left', 'right')


def margin_terms(s):
    """(5, H, W) float64: s - thr, s - up, s - down, s - left, s - right with zero outside the map (pose_detector.py:91-94).
    Differences of float32 values are taken in float64, so their SIGN is the sign of the float32 comparison."""
    s32 = np.asarray(s, dtype=np.float32)
    s = s32.astype(np.float64)
    up = np.zeros_like(s); up[1:, :] = s[:-1, :]
    down = np.zeros_like(s); down[:-1, :] = s[1:, :]
    left = np.zeros_like(s); left[:, 1:] = s[:, :-1]
    right = np.zeros_like(s); right[:, :-1] = s[:, 1:]
    return np.stack([s - np.float64(THR), s - up, s - down, s - left, s - right])


def margin_map(s):
    return margin_terms(s).min(axis=0)


def local_diff(g, o, y, x):
    """max |g - o| over the 5-point neighbourhood of (y, x) (the pixels the decision reads)."""
    H, W = g.shape
    d = 0.0
    for dy, dx in ((0, 0), (-1, 0), (1, 0), (0, -1), (0, 1)):
        yy, xx = y + dy, x + dx
        if 0 <= yy < H and 0 <= xx < W:
            d = max(d, abs(float(g[yy, xx]) - float(o[yy, xx])))
    return d


def peak_set(all_peaks):
    p = np.asarray(all_peaks, dtype=np.float64).reshape(-1, 5)
    return {(int(r[0]), int(r[1]), int(r[2])) for r in p}


def compare_frame(gpu_peaks, cpu_peaks, cpu_smoothed, gpu_smoothed_of, gpu_poses, gpu_scores, cpu_poses, cpu_scores):
    """One frame.  gpu_peaks / cpu_peaks: (N, 5) rows (type, x, y, score, id); cpu_smoothed: (18, H, W) float32 of the oracle;
    gpu_smoothed_of(joint) -> (H, W) float32 of the GPU path (fetched only for joints that disagree); poses (n, 18, 3), scores (n,).
    Returns a dict (JSON-serialisable)."""
    gp, cp = np.asarray(gpu_peaks, dtype=np.float64).reshape(-1, 5), np.asarray(cpu_peaks, dtype=np.float64).reshape(-1, 5)
    gs, cs = peak_set(gp), peak_set(cp)
    identical_peaks = gp.shape == cp.shape and np.array_equal(gp[:, [0, 1, 2, 4]], cp[:, [0, 1, 2, 4]])
    out = {'n_peaks_gpu': len(gp), 'n_peaks_cpu': len(cp), 'identical_peaks': bool(identical_peaks), 'mismatches': []}
    # scores of the peaks both sides found
    common = gs & cs
    if common:
        gsc = {(int(r[0]), int(r[1]), int(r[2])): r[3] for r in gp}
        csc = {(int(r[0]), int(r[1]), int(r[2])): r[3] for r in cp}
        out['max_abs_peak_score_diff'] = max(abs(gsc[k] - csc[k]) for k in common)
    else:
        out['max_abs_peak_score_diff'] = 0.0
    # margins of the oracle's decisions: the closest accepted peak, and the closest decision of any pixel
    m_cpu = np.stack([margin_map(cpu_smoothed[j]) for j in range(cpu_smoothed.shape[0])])
    diff_px = gs ^ cs
    agreed = np.ones(m_cpu.shape, dtype=bool)
    for (j, x, y) in diff_px:
        agreed[j, y, x] = False
    acc = [m_cpu[j, y, x] for (j, x, y) in common]
    out['min_margin_of_accepted_peaks'] = float(min(acc)) if acc else None
    out['min_abs_margin_of_agreed_decisions'] = float(np.abs(m_cpu[agreed]).min())
    out['pixels_within_1e-5_of_a_flip'] = int(np.count_nonzero(np.abs(m_cpu) < 1e-5))
    out['pixels_within_1e-6_of_a_flip'] = int(np.count_nonzero(np.abs(m_cpu) < 1e-6))
    cache = {}
    for (j, x, y) in sorted(diff_px):
        if j not in cache:
            cache[j] = np.asarray(gpu_smoothed_of(j), dtype=np.float32)
        g, o = cache[j], cpu_smoothed[j]
        tg, to = margin_terms(g)[:, y, x], margin_terms(o)[:, y, x]
        mg, mo = float(tg.min()), float(to.min())
        side = 'gpu_only' if (j, x, y) in gs else 'cpu_only'
        loser = to if side == 'gpu_only' else tg              # the side that rejected the pixel: which test failed
        out['mismatches'].append({
            'joint': j, 'x': x, 'y': y, 'side': side, 'margin_gpu': mg, 'margin_cpu': mo,
            'margin': max(abs(mg), abs(mo)), 'margin_sum': abs(mg) + abs(mo),
            'decided_by': TESTS[int(np.argmin(loser))], 'local_abs_diff_smoothed': local_diff(g, o, y, x),
            'smoothed_gpu': float(g[y, x]), 'smoothed_cpu': float(o[y, x]),
            'frame_max_abs_diff_smoothed': float(np.abs(g.astype(np.float64) - o.astype(np.float64)).max())})
    # people
    gposes = np.asarray(gpu_poses, dtype=np.float64).reshape(-1, 18, 3)
    cposes = np.asarray(cpu_poses, dtype=np.float64).reshape(-1, 18, 3)
    gsco, csco = np.asarray(gpu_scores, dtype=np.float64).reshape(-1), np.asarray(cpu_scores, dtype=np.float64).reshape(-1)
    out['n_people_gpu'], out['n_people_cpu'] = int(len(gposes)), int(len(cposes))
    out['identical_poses'] = bool(gposes.shape == cposes.shape and np.array_equal(gposes, cposes))
    # matched people = identical (18, 3) key-point rows on both sides (greedy, first match)
    used = set()
    d = 0.0
    matched = 0
    for i, gpo in enumerate(gposes):
        for k, cpo in enumerate(cposes):
            if k not in used and np.array_equal(gpo, cpo):
                used.add(k)
                matched += 1
                d = max(d, abs(float(gsco[i]) - float(csco[k])))
                break
    out['matched_people'] = matched
    out['max_abs_score_diff_matched_people'] = d
    return out


def candidate_margin(paf_xy, a_xy, b_xy, img_len):
    """Decision margin of one candidate connection (reference pose_detector.py:135-156, restated as in
    postprocess_ref.compute_candidate_connections): accepted  <=>  count(ip > 0.05) > 8  and  integ + prior > 0, i.e.
    min(second-smallest ip - 0.05, score) > 0.  Returns (margin, score, which test is the binding one)."""
    vx, vy = b_xy[0] - a_xy[0], b_xy[1] - a_xy[1]
    norm = np.sqrt(vx * vx + vy * vy)
    if norm == 0:
        return None, None, 'zero length'
    ys = np.linspace(a_xy[1], b_xy[1], num=P.N_INTEG_POINTS).round().astype('i')
    xs = np.linspace(a_xy[0], b_xy[0], num=P.N_INTEG_POINTS).round().astype('i')
    ip = paf_xy[0][ys, xs].astype(np.float64) * (vx / norm) + paf_xy[1][ys, xs].astype(np.float64) * (vy / norm)
    score = P._np_sum10(ip) / len(ip) + min(P.LIMB_LENGTH_RATIO * img_len / norm - P.LENGTH_PENALTY_VALUE, 0)
    need = P.N_INTEG_POINTS - P.N_INTEG_POINTS_THRESH               # at most 1 of the 10 samples may fail
    m_count = float(np.sort(ip)[need - 1] - P.INNER_PRODUCT_THRESH)   # the 9th largest sample must exceed the threshold
    return (min(m_count, float(score)), float(score), 'n_valid count' if m_count < score else 'score > 0')


def compare_connections(gpu_conns, cpu_conns, peaks, gpu_paf_lo, cpu_paf_lo, map_h, map_w, img_len):
    """For a frame whose peak sets agree: every connection only one side accepted, with the margin of its acceptance test
    (candidate_margin) on both sides' PAF maps.  gpu_conns / cpu_conns: rows (limb, id_a, id_b, score); *_paf_lo: (38, h, w)
    network outputs (the full-resolution PAF of either side is F.resize_images of it, restated bit-exactly)."""
    g = {(int(r[0]), int(r[1]), int(r[2])): float(r[3]) for r in np.asarray(gpu_conns, dtype=np.float64).reshape(-1, 4)}
    c = {(int(r[0]), int(r[1]), int(r[2])): float(r[3]) for r in np.asarray(cpu_conns, dtype=np.float64).reshape(-1, 4)}
    out = []
    if set(g) == set(c):
        return out
    gp, cp = P.resize_images_ref(gpu_paf_lo, map_h, map_w), P.resize_images_ref(cpu_paf_lo, map_h, map_w)
    pk = np.asarray(peaks, dtype=np.float64).reshape(-1, 5)
    for (l, ia, ib) in sorted(set(g) ^ set(c)):
        a_xy, b_xy = pk[ia, 1:3], pk[ib, 1:3]
        mg, sg, tg = candidate_margin(gp[[2 * l, 2 * l + 1]], a_xy, b_xy, img_len)
        mc, sc_, tc = candidate_margin(cp[[2 * l, 2 * l + 1]], a_xy, b_xy, img_len)
        out.append({'limb': l, 'id_a': ia, 'id_b': ib, 'side': 'gpu_only' if (l, ia, ib) in g else 'cpu_only',
                    'margin_gpu': mg, 'margin_cpu': mc, 'score_gpu': sg, 'score_cpu': sc_, 'binding_test': tg if (mg is not None and abs(mg) <= abs(mc)) else tc,
                    'max_abs_diff_paf_full_res': float(np.abs(gp[[2 * l, 2 * l + 1]].astype(np.float64) - cp[[2 * l, 2 * l + 1]]).max()),
                    'acceptance_flipped': bool(mg is not None and (mg > 0) != (mc > 0))})
    return out


def summarize(frames, label):
    """Census object over a list of compare_frame() results."""
    mism = [m for f in frames for m in f['mismatches']]
    acc = [f['min_margin_of_accepted_peaks'] for f in frames if f['min_margin_of_accepted_peaks'] is not None]
    s = {
        'path': label,
        'frames': len(frames),
        'frames_identical': sum(1 for f in frames if f['identical_peaks'] and f['identical_poses']),
        'frames_with_identical_peak_indices': sum(1 for f in frames if f['identical_peaks']),
        'frames_with_identical_poses': sum(1 for f in frames if f['identical_poses']),
        'frames_with_identical_peaks_but_different_poses': sum(1 for f in frames if f['identical_peaks'] and not f['identical_poses']),
        'peaks_compared': sum(f['n_peaks_cpu'] for f in frames),
        'mismatching_peaks': len(mism),
        'max_margin_of_a_mismatch': max((m['margin'] for m in mism), default=0.0),
        'max_margin_over_local_diff_of_a_mismatch': max((m['margin'] / m['local_abs_diff_smoothed'] for m in mism if m['local_abs_diff_smoothed'] > 0), default=0.0),
        'mismatches_decided_by': {t: sum(1 for m in mism if m['decided_by'] == t) for t in TESTS},
        'min_margin_of_accepted_peaks': min(acc) if acc else None,
        'min_abs_margin_of_agreed_decisions': min(f['min_abs_margin_of_agreed_decisions'] for f in frames) if frames else None,
        'pixels_within_1e-5_of_a_flip_per_frame_mean': float(np.mean([f['pixels_within_1e-5_of_a_flip'] for f in frames])) if frames else None,
        'pixels_within_1e-6_of_a_flip_per_frame_mean': float(np.mean([f['pixels_within_1e-6_of_a_flip'] for f in frames])) if frames else None,
        'max_abs_peak_score_diff': max((f['max_abs_peak_score_diff'] for f in frames), default=0.0),
        'people_cpu': sum(f['n_people_cpu'] for f in frames),
        'matched_people': sum(f['matched_people'] for f in frames),
        'max_abs_score_diff_matched_people': max((f['max_abs_score_diff_matched_people'] for f in frames), default=0.0),
        'max_abs_diff_smoothed_on_mismatching_frames': max((m['frame_max_abs_diff_smoothed'] for m in mism), default=None),
    }
    cd = [dict(c_, frame=f.get('frame')) for f in frames for c_ in f.get('connection_mismatches', [])]
    s['connection_mismatches_on_frames_with_identical_peaks'] = cd
    s['max_margin_of_a_connection_mismatch'] = max((max(abs(c_['margin_gpu']), abs(c_['margin_cpu'])) for c_ in cd if c_['acceptance_flipped']), default=0.0)
    s['all_mismatches_are_near_ties'] = bool(all(m['margin_sum'] <= 2.0 * m['local_abs_diff_smoothed'] * (1 + 1e-9) + 1e-30 for m in mism))
    return s
