#!/usr/bin/env python
"""Near-tie margin census of the default batch path against the CPU oracle.

    python tools/parity_census.py --frames 512 --out profiles/r04_parity_census.json          (GPU box; ~3 min)

For every one of N synthetic 368x368 frames (batches of 32, fresh seeds; the bench's seeded weights + calibrated head):
  GPU   the DEFAULT batch path (pmx_detect_batch at batch 32: run-geometry Winograd kernels with unit-mode tails + the HIP
        post-process) -- and, with --bf16x3, the opt-in bf16x3 mode on the same frames;
  CPU   the oracle: torch-CPU fp32 restatement of the network (oracle/network_ref) + NumPy restatement of the reference post-process
        (oracle/postprocess_ref), both pinned bit-exactly to the verbatim reference in the authoring container.
Per frame: identical peak indices / identical poses; for every peak only one side found, the MARGIN that decided it on both sides
(oracle/census.py: min over the reference's five strict comparisons, pose_detector.py:96-102), which comparison failed, and the
local |GPU - CPU| of the smoothed maps (GPU side: pmx_get_smoothed, the kernel's own map).  Two fp32 networks that differ by
summation order can only disagree on pixels with |margin_gpu| + |margin_cpu| <= 2 x that local difference -- the census asserts it
for every disagreement, and that the scores of all people both sides found agree to 1e-4 (the north_star tolerance).
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = 'chainer_realtime_multi-person_pose_estimation_amd'


def run_census(frames=512, batch=32, size=368, seed0=7000, bf16x3=False, threads=None, log=None, extra_modes=()):
    """extra_modes: ((name, {engine option: value, ...}), ...) -- further engine configurations compared with the SAME oracle frames (e.g.
    ('conv1_direct', {'conv1_wino': 0}), ('direct_kernels', {'conv_algo': 0})): which kernel family a drift of the census comes from."""
    import torch
    from oracle import census, network_ref, postprocess_ref
    native = importlib.import_module(PKG + '.native')
    weights_mod = importlib.import_module(PKG + '.weights')
    if native.needs_build():
        native.build()
    if not threads:
        from bench import usable_cores          # (min of os.cpu_count, the affinity mask and the cgroup CPU quota)
        threads = min(usable_cores(), 32)
    torch.set_num_threads(threads)
    SH, SW = (size, size) if isinstance(size, int) else size         # network input height, width (368 x 496: the 46 x 62 maps of a 4:3 frame)
    B = batch
    map_h, map_w = (320 if SH == 368 else (SH * 320) // 368 // 8 * 8), (320 if SW == 368 else (SW * 320) // 368 // 8 * 8)
    eng = native.Engine(0, max_batch=B, max_h=SH, max_w=SW)
    w = weights_mod.synthetic_weights(0)
    eng.set_weights(w)
    cal = np.random.default_rng(1234).integers(0, 256, (1, SH, SW, 3), dtype=np.uint8)     # as bench.py
    eng.forward_u8(cal)
    paf, heat = eng.get_maps()
    w = weights_mod.calibrate_head(w, paf[0], heat[0])
    eng.set_weights({k: w[k] for k in ('Mconv7_stage6_L1', 'Mconv7_stage6_L2')})
    eng.set_option('keep_smoothed', 1)
    modes = [('f32_default_batch_path', {})] + ([('bf16x3_opt_in', {'precision': 1})] if bf16x3 else []) + [(n, dict(o)) for n, o in extra_modes]
    defaults = {'precision': 0, 'conv1_wino': 1, 'conv_algo': 1, 'ksplit': 0, 'wino_tail': -1, 'wino_geom': -1}      # what a mode's options are reset to
    per_mode = {name: [] for name, _ in modes}
    t_gpu = t_cpu = 0.0
    nb = (frames + B - 1) // B
    for it in range(nb):
        imgs = np.random.default_rng(seed0 + it).integers(0, 256, (B, SH, SW, 3), dtype=np.uint8)
        t0 = time.perf_counter()
        oracle = []
        for i in range(B):
            opaf, oheat = network_ref.forward(w, postprocess_ref.preprocess(imgs[i]))
            o = postprocess_ref.postprocess_from_net_output(opaf[0], oheat[0], map_h, map_w)
            oracle.append(dict({k: o[k] for k in ('all_peaks', 'poses', 'scores', 'smoothed', 'connections')}, paf_lo=opaf[0]))
        t_cpu += time.perf_counter() - t0
        for name, opts in modes:
            for k, v in opts.items():
                assert k in defaults, 'census mode option %r has no reset value' % k
                eng.set_option(k, v)
            t0 = time.perf_counter()
            eng.detect_batch(imgs, map_h, map_w)
            rec = eng.results()
            t_gpu += time.perf_counter() - t0
            assert int(np.bitwise_or.reduce(rec['status'])) == 0, 'status bits set'
            maps_cache = {}
            for i in range(B):
                if len(per_mode[name]) >= frames:
                    break
                n = int(rec[i]['n_people'])
                f = census.compare_frame(eng.peaks(i), oracle[i]['all_peaks'], oracle[i]['smoothed'],
                                         lambda j, i=i: eng.smoothed(i, j), rec[i]['poses'][:n], rec[i]['scores'][:n],
                                         oracle[i]['poses'], oracle[i]['scores'])
                if f['identical_peaks'] and not f['identical_poses']:
                    # same peaks, different people: a connection test (pose_detector.py:155) flipped -- attribute it
                    if name not in maps_cache:
                        maps_cache[name] = eng.get_maps()
                    f['connection_mismatches'] = census.compare_connections(
                        eng.connections(i), np.concatenate([np.column_stack([np.full(len(c_), l), c_]) for l, c_ in enumerate(oracle[i]['connections'])] or [np.zeros((0, 4))]),
                        oracle[i]['all_peaks'], maps_cache[name][0][i], oracle[i]['paf_lo'], map_h, map_w, map_w)
                f['frame'] = it * B + i
                f['seed'] = seed0 + it
                per_mode[name].append(f)
            for k in opts:
                eng.set_option(k, defaults[k])
        if log:
            done = len(per_mode[modes[0][0]])
            log('batch %d/%d: %d frames, %s' % (it + 1, nb, done, ', '.join(
                '%s: %d identical, %d mismatching peaks' % (nm, sum(1 for f in fr if f['identical_peaks'] and f['identical_poses']),
                                                         sum(len(f['mismatches']) for f in fr)) for nm, fr in per_mode.items())))
    eng.close()
    out = {'workload': 'batch%d_%dx%d_synthetic_uint8 (bench.py weights: seeded He + calibrated head), seeds %d..%d' % (B, SH, SW, seed0, seed0 + nb - 1),
           'oracle': 'torch-CPU fp32 network restatement + NumPy restatement of the reference post-process (oracle/network_ref, postprocess_ref)',
           'decision': 'reference pose_detector.py:96-102: smoothed > 0.05 and > up, down, left, right (strict, float32); margin = min of the five differences',
           'seconds': {'gpu_paths': t_gpu, 'cpu_oracle': t_cpu}, 'paths': {}, 'mismatches': {}}
    for name, fr in per_mode.items():
        out['paths'][name] = census.summarize(fr, name)
        out['mismatches'][name] = [dict(m, frame=f['frame'], seed=f['seed']) for f in fr for m in f['mismatches']]
        out['paths'][name]['frames_not_identical'] = [f['frame'] for f in fr if not (f['identical_peaks'] and f['identical_poses'])]
    return out


def check(out, score_tol=1e-4):
    """The census's assertions (also used by tests/test_gpu_census.py)."""
    for name, s in out['paths'].items():
        assert s['all_mismatches_are_near_ties'], (name, 'a disagreeing peak whose margins exceed twice the local map difference')
        assert s['max_abs_score_diff_matched_people'] <= score_tol, (name, s['max_abs_score_diff_matched_people'])
        assert s['max_abs_peak_score_diff'] <= score_tol, (name, s['max_abs_peak_score_diff'])
        assert s['max_margin_of_a_mismatch'] <= score_tol, (name, s['max_margin_of_a_mismatch'])
        # frames with the same peaks but different people: the flipped connection test must be a near-tie too
        assert s['max_margin_of_a_connection_mismatch'] <= score_tol, (name, s['connection_mismatches_on_frames_with_identical_peaks'])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--frames', type=int, default=512)
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--seed0', type=int, default=7000)
    ap.add_argument('--bf16x3', action='store_true')
    ap.add_argument('--threads', type=int, default=0)
    ap.add_argument('--h', type=int, default=368, help='network input height (multiple of 8)')
    ap.add_argument('--w', type=int, default=368, help='network input width: --h 368 --w 496 = the 46 x 62 maps of a 4:3 frame')
    ap.add_argument('--out', default=None)
    ap.add_argument('--mode', action='append', default=[], metavar='NAME:opt=val[,opt=val]',
                    help='a further engine configuration on the same frames, e.g. conv1_direct:conv1_wino=0  direct_kernels:conv_algo=0')
    a = ap.parse_args()
    extra = []
    for m in a.mode:
        name, _, kv = m.partition(':')
        extra.append((name, {k: int(v) for k, v in (x.split('=') for x in kv.split(',') if x)}))
    out = run_census(a.frames, a.batch, (a.h, a.w), a.seed0, a.bf16x3, a.threads or None, log=lambda s: print(s, file=sys.stderr, flush=True),
                     extra_modes=extra)
    check(out)
    txt = json.dumps(out, indent=1)
    if a.out:
        with open(a.out, 'w') as f:
            f.write(txt + '\n')
    print(json.dumps(out['paths'], indent=1))


if __name__ == '__main__':
    main()
