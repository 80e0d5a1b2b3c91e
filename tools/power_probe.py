#!/usr/bin/env python
"""Power / clock telemetry while single conv layer shapes of the batch-32 network run back to back (pmx_conv2d, ~1.5 s each):
is the VGG stem at 368 / 184 / 92 px power- or clock-limited relative to the 7x7 layers?  Samples the amdgpu hwmon files
(socket power, sclk) every 10 ms from a thread, falls back to `rocm-smi --json`; prints one line per shape.

    python tools/power_probe.py > gpurun_out/power_probe.txt
"""
import glob, importlib, json, os, subprocess, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
native = importlib.import_module('chainer_realtime_multi-person_pose_estimation_amd.native')


def hwmon_files():
    out = {}
    for card in sorted(glob.glob('/sys/class/drm/card*/device')):
        hw = glob.glob(os.path.join(card, 'hwmon', 'hwmon*'))
        if not hw:
            continue
        h = hw[0]
        for key, names in (('power_uW', ('power1_average', 'power1_input')), ('sclk_Hz', ('freq1_input',)), ('mclk_Hz', ('freq2_input',)),
                           ('temp_mC', ('temp1_input',)), ('cap_uW', ('power1_cap',))):
            for n in names:
                p = os.path.join(h, n)
                if os.path.exists(p):
                    out.setdefault(key, p)
                    break
        if 'power_uW' in out or 'sclk_Hz' in out:
            out['card'] = card
            break
    return out


class Sampler(threading.Thread):
    def __init__(self, files):
        threading.Thread.__init__(self, daemon=True)
        self.files, self.rows, self.stop = files, [], False

    def run(self):
        while not self.stop:
            row = {'t': time.perf_counter()}
            for k, p in self.files.items():
                if k == 'card':
                    continue
                try:
                    row[k] = float(open(p).read().split()[0])
                except Exception:
                    pass
            self.rows.append(row)
            time.sleep(0.01)


def smi_once():
    try:
        txt = subprocess.run(['rocm-smi', '--showpower', '--showclocks', '--json'], capture_output=True, text=True, timeout=20).stdout
        return json.loads(txt)
    except Exception as e:
        return {'error': repr(e)}


files = hwmon_files()
print('hwmon files:', files, flush=True)
print('rocm-smi idle:', json.dumps(smi_once())[:600], flush=True)
eng = native.Engine(0, max_batch=32, max_h=368, max_w=368)
if '--bf16x3' in sys.argv:
    eng.set_option('precision', 1)
    print('precision = bf16x3 (v7 kernels where a v6 kernel would run)', flush=True)
rng = np.random.default_rng(0)
SHAPES = [  # name, cin, H, W, cout, k, pool
    ('conv1_2  64->64   368 pool', 64, 368, 368, 64, 3, True), ('conv2_1  64->128  184', 64, 184, 184, 128, 3, False),
    ('conv2_2 128->128  184 pool', 128, 184, 184, 128, 3, True), ('conv3_2 256->256   92', 256, 92, 92, 256, 3, False),
    ('conv4_2 512->512   46', 512, 46, 46, 512, 3, False), ('Mconv2 7x7 128->256 46', 128, 46, 46, 256, 7, False),
    ('Mconv1 7x7 192->256 46', 192, 46, 46, 256, 7, False)]
for dense in (0, 1):
    for name, cin, H, W, cout, k, pool in SHAPES:
        x = rng.standard_normal((32, cin, H, W)).astype('f')
        if not dense:
            x = np.maximum(x, 0)          # post-ReLU-like: half of the activations are zero, as inside the network
        w = (rng.standard_normal((cout, cin, k, k)) / np.sqrt(cin * k * k)).astype('f')
        b = np.zeros(cout, 'f')
        _, ms = eng.conv2d(x, w, b, relu=True, pool=pool, iters=3)
        iters = max(10, int(1500.0 / ms))
        s = Sampler(files)
        s.start()
        t0 = time.perf_counter()
        _, ms = eng.conv2d(x, w, b, relu=True, pool=pool, iters=iters)
        t1 = time.perf_counter()
        s.stop = True
        s.join()
        # the timed loop sits in the middle of the call (upload before, download after): keep the samples of the busy middle
        rows = [r for r in s.rows if t0 + 0.35 * (t1 - t0) <= r['t'] <= t1 - 0.15 * (t1 - t0)]
        def stat(key, scale):
            v = [r[key] * scale for r in rows if key in r]
            return (float(np.mean(v)), float(np.max(v)), float(np.min(v))) if v else (float('nan'),) * 3
        pw, sc = stat('power_uW', 1e-6), stat('sclk_Hz', 1e-6)
        tf = 2.0 * 32 * H * W * cout * cin * k * k / (ms * 1e-3) / 1e12
        print('%-28s %s  %7.3f ms  %6.1f TFLOP/s | power W mean %.0f max %.0f min %.0f | sclk MHz mean %.0f max %.0f min %.0f | %d samples'
              % (name, 'dense ' if dense else 'relu50', ms, tf, pw[0], pw[1], pw[2], sc[0], sc[1], sc[2], len(rows)), flush=True)
if 'cap_uW' in files:
    print('power cap W:', float(open(files['cap_uW']).read()) * 1e-6)
print('rocm-smi after:', json.dumps(smi_once())[:600], flush=True)
