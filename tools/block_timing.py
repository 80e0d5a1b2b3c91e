#!/usr/bin/env python
"""Where does a Winograd block's time go?  Diagnostic build of the library (conv_wino.hip compiled with -DPMX_BLOCK_TIMING: thread 0 of a
block stamps the 100 MHz wall clock at entry / pipeline primed / before the stores / exit (more stamps perturb the loops: six of them made a 7x7 block 1.5x slower),
and the CU it runs on) -> per-block phase durations and the gap between consecutive blocks on one CU, for one layer shape.

    python tools/block_timing.py --ks 7 --cin 128 --batch 64          (one group of a 7x7 layer at batch 32 x 2 branches)
    python tools/block_timing.py --ks 3 --cin 64 --cout 128 --hw 184 --batch 32     (conv2_1)
The product library is untouched: the diagnostic library is built into tools/_build/."""
import argparse, ctypes as C, importlib, json, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = 'chainer_realtime_multi-person_pose_estimation_amd'
ap = argparse.ArgumentParser()
ap.add_argument('--ks', type=int, default=7); ap.add_argument('--cin', type=int, default=128); ap.add_argument('--cout', type=int, default=128)
ap.add_argument('--hw', type=int, default=46); ap.add_argument('--batch', type=int, default=64); ap.add_argument('--pool', type=int, default=0)
ap.add_argument('--json', default=None); ap.add_argument('--raw', default=None, help='save the raw stamps (blocks x 8, 100 MHz ticks; column 7 = CU id) as .npy')
ap.add_argument('--keep-tail', action='store_true', help='leave the wino_tail option alone (batch 1 then runs the unit-mode plan of the product)')
a = ap.parse_args()
native = importlib.import_module(PKG + '.native')
out_dir = os.path.join(ROOT, 'tools', '_build')
os.makedirs(out_dir, exist_ok=True)
lib = os.path.join(out_dir, 'libpose_timing.so')
objs = []
for src, extra in native.SOURCES:
    o = os.path.join(out_dir, src.replace('.hip', '.timing.o'))
    if not os.path.exists(o) or os.path.getmtime(o) < os.path.getmtime(os.path.join(native.CSRC, src)):
        subprocess.check_call([native._hipcc(), '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-DPMX_BLOCK_TIMING'] + extra +
                              ['-c', os.path.join(native.CSRC, src), '-o', o], cwd=native.CSRC)
    objs.append(o)
subprocess.check_call([native._hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', '-o', lib] + objs)
native.LIB_PATH = lib
if os.environ.get('PMX_TIMING_BUILD_ONLY'):
    print('built', lib); sys.exit(0)
L = native.load()
L.pmx_debug_block_times.argtypes = [C.c_void_p, C.c_size_t]
eng = native.Engine(0, max_batch=a.batch, max_h=max(368, a.hw), max_w=max(368, a.hw))
eng.set_option('conv_algo', 1)
if not a.keep_tail:
    eng.set_option('wino_tail', 0)      # one kernel for every block (the part-filled last block of an image runs in the same launch)
rng = np.random.default_rng(0)
x = np.maximum(rng.standard_normal((a.batch, a.cin, a.hw, a.hw)), 0).astype('f')
w = (rng.standard_normal((a.cout, a.cin, a.ks, a.ks)) / np.sqrt(a.cin * a.ks * a.ks)).astype('f')
b = rng.standard_normal(a.cout).astype('f')
y, ms = eng.conv2d(x, w, b, relu=True, pool=bool(a.pool), iters=3)
t = np.zeros(8192 * 8, np.uint64)
assert L.pmx_debug_block_times(t.ctypes.data, t.size) == 0
t = t.reshape(8192, 8).astype(np.int64)
if a.raw:
    np.save(a.raw, t)
# the stamps of the LAST launch that touched each slot: the main launch (lin < its grid) overwrites; keep blocks with a complete stamp set
ok = (t[:, 0] > 0) & (t[:, 6] > t[:, 0]) & (t[:, 6] - t[:, 0] < 10_000_00)
tt = t[ok]
us = lambda v: v * 0.01          # 100 MHz ticks -> microseconds
names = ['prologue: entry -> pipeline primed (first halo in LDS, first window transformed)', 'loops: passes 1, 2a, 2b + output transforms', 'epilogue (bias, ReLU, stores)']
seg = [(0, 2), (2, 5), (5, 6)]
res = {'shape': vars(a), 'layer_ms': ms, 'blocks_with_stamps': int(ok.sum()), 'phases_us_mean_p10_p90': {}}
for nm, (i, j) in zip(names, seg):
    d = us(tt[:, j] - tt[:, i])
    res['phases_us_mean_p10_p90'][nm] = [float(d.mean()), float(np.percentile(d, 10)), float(np.percentile(d, 90))]
tot = us(tt[:, 6] - tt[:, 0])
res['block_us_mean_p10_p90'] = [float(tot.mean()), float(np.percentile(tot, 10)), float(np.percentile(tot, 90))]
# hand-over gaps per CU: sort the blocks of one CU by entry, gap = next entry - previous exit
gaps = []
for cu in np.unique(tt[:, 7]):
    bl = tt[tt[:, 7] == cu]
    bl = bl[np.argsort(bl[:, 0])]
    g = us(bl[1:, 0] - bl[:-1, 6])
    gaps += [v for v in g if -5 < v < 100]
res['cus_seen'] = int(len(np.unique(tt[:, 7])))
res['handover_gap_us_mean_p10_p90'] = [float(np.mean(gaps)), float(np.percentile(gaps, 10)), float(np.percentile(gaps, 90))] if gaps else None
span = us(tt[:, 6].max() - tt[:, 0].min())
res['first_entry_to_last_exit_us'] = float(span)
print(json.dumps(res, indent=1))
if a.json:
    json.dump(res, open(a.json, 'w'), indent=1)
