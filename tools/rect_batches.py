#!/usr/bin/env python
"""Landscape / portrait network inputs against the square case at several batch sizes: does the per-pixel rate of the
46 x 62 / 62 x 46 maps (compute_optimal_size of a 4:3 frame, reference pose_detector.py:57-73) depend on the batch size -- i.e. on how
the block count of a launch falls on whole rounds of the 256 CUs?  One table: frames/s per size and batch, rate per pixel relative to
the 368 x 368 step OF THE SAME BATCH SIZE, issued fraction of the dominant kernel.  Device-resident inputs, bench.py's own step
(bench.rect_inputs).  -> profiles/rNN_rect_batches.json"""
import argparse, importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

ap = argparse.ArgumentParser()
ap.add_argument('--batches', default='8,16,24,32'); ap.add_argument('--steps', type=int, default=10); ap.add_argument('--json', default=None)
ap.add_argument('--engine-opt', action='append', default=[], metavar='KEY=VALUE')
a = ap.parse_args()
import torch
native = importlib.import_module(bench.PKG + '.native')
weights_mod = importlib.import_module(bench.PKG + '.weights')
dev = torch.device('cuda:0')
sizes = (('368x368', (368, 368)), ('368x496', (368, 496)), ('496x368', (496, 368)))
out = {'what': __doc__.split('->')[0].strip(), 'batches': {}}
for B in [int(x) for x in a.batches.split(',')]:
    r = bench.rect_inputs(native, weights_mod, torch, dev, 0, B, a.steps, 1.0, 368, sizes=sizes)
    sq = r['368x368']['value']
    row = {}
    for name, (h, w) in sizes:
        o = r[name]
        row[name] = {'frames_per_s': o['value'], 'ms_per_step': o['ms_per_step'], 'rate_per_pixel_vs_square_same_batch': o['value'] * h * w / (sq * 368 * 368),
                     'step_issued_frac': o.get('step_issued_frac'), 'dominant_kernel': (o.get('dominant_kernel') or {}).get('kernel'),
                     'dominant_issued_frac': (o.get('dominant_kernel') or {}).get('issued_frac')}
    out['batches'][str(B)] = row
    print('B=%2d  ' % B + '   '.join('%s %7.1f f/s x%.3f (%s %.3f)' % (n, v['frames_per_s'], v['rate_per_pixel_vs_square_same_batch'],
                                                                       (v['dominant_kernel'] or '?').replace('conv_wino_kernel', 'wino'), v['dominant_issued_frac'] or 0)
                                     for n, v in row.items()), flush=True)
if a.json:
    json.dump(out, open(a.json, 'w'), indent=1)
