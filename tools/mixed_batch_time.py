#!/usr/bin/env python
"""bench.mixed_sizes_mode on its own: a stream of frames of different sizes through PoseDetector.detect_batch (mixed batch) against one
image per call and a uniform batch.  -> profiles/rNN_mixed_batch.json"""
import argparse, importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=32); ap.add_argument('--steps', type=int, default=5); ap.add_argument('--json', default=None)
a = ap.parse_args()
out = bench.mixed_sizes_mode(importlib.import_module(bench.PKG + '.weights'), 0, a.batch, a.steps)
print(json.dumps(out, indent=1))
if a.json:
    json.dump(out, open(a.json, 'w'), indent=1)
