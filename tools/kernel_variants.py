#!/usr/bin/env python
"""A/B builds of the convolution kernels: conv_mfma.hip / conv_wino.hip (and pmx_api.hip, which packs their weights) compiled with extra -D flags, linked with the product's other objects into
tools/_build/libpose_var_<tag>.so (the product library is untouched), and timed through the whole network with the layer profile on.

    python tools/kernel_variants.py build base: noxf:PMX_ABLATE=1 nowl:PMX_ABLATE=2      (here: hipcc cross-compiles)
    python tools/kernel_variants.py time [--steps 5] [--batch 32] [--json out.json]                                                (on the GPU box)
    python tools/kernel_variants.py time-conv [--iters 5] [--json out.json]            (GPU box: single layers through pmx_conv2d -- ablation builds)
"""
import glob, importlib, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = 'chainer_realtime_multi-person_pose_estimation_amd'
OUT = os.path.join(ROOT, 'tools', '_build')


def build(specs):
    native = importlib.import_module(PKG + '.native')
    native.build()
    os.makedirs(OUT, exist_ok=True)
    for spec in specs:
        spec, _, only = spec.partition('@')      # tag:DEF=1,DEF2=2[@file.hip+file2.hip]: the sources the flags apply to (default: the three below)
        tag, _, defs = spec.partition(':')
        flags = [d if d.startswith('-') else '-D' + d for d in defs.split(',') if d]      # (an item that starts with '-' is passed as it is: -mllvm,-pragma-unroll-threshold=N)
        objs = []
        varied = tuple(only.split('+')) if only else ('conv_mfma.hip', 'conv_wino.hip', 'pmx_api.hip')
        for src, extra in native.SOURCES:
            if src in varied:      # (the kernels, and the host side that packs their weights)
                o = os.path.join(OUT, '%s.var_%s.o' % (src[:-4], tag))
                subprocess.check_call([native._hipcc(), '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC'] + flags + extra +
                                      ['-c', os.path.join(native.CSRC, src), '-o', o], cwd=native.CSRC)
            else:
                o = os.path.join(native.CSRC, src.replace('.hip', '.o'))
            objs.append(o)
        lib = os.path.join(OUT, 'libpose_var_%s.so' % tag)
        subprocess.check_call([native._hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', '-o', lib] + objs)
        print('built', lib, ' '.join(flags))


def group(layer, kernel):
    if kernel.endswith(':units') or kernel.endswith(':combine'):
        return 'tails'
    if '7x7' in kernel:
        return '7x7 main'
    if 'f2x2_3x3' in kernel:
        return '3x3 main'
    if kernel.startswith('pp_'):
        return 'post-process'
    if layer.startswith('conv1_1'):
        return 'conv1'
    return 'other'


def time_all(steps, out_json, batch=32):
    res = {}
    for lib in sorted(glob.glob(os.path.join(OUT, 'libpose_var_*.so'))):
        tag = os.path.basename(lib)[len('libpose_var_'):-3]
        pj = os.path.join('/tmp', 'var_%s.json' % tag)
        r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'profile_driver.py'), '--lib', lib, '--batch', str(batch), '--steps', str(steps),
                            '--profile-json', pj], capture_output=True, text=True, timeout=300)
        if r.returncode:
            print(tag, 'FAILED', r.stderr[-400:]); continue
        d = json.load(open(pj))
        g = {}
        for e in d['entries']:
            k = group(e['layer'], e['kernel'])
            g[k] = g.get(k, 0.0) + e['avg_ms']
        g['sum'] = sum(g.values())
        res[tag] = g
        print('%-16s' % tag, ' '.join('%s %.3f' % kv for kv in sorted(g.items())), '|', r.stdout.strip().splitlines()[-2])
        sys.stdout.flush()
    if out_json:
        json.dump(res, open(out_json, 'w'), indent=1)


def time_conv(iters, out_json):
    """Per-layer timing through pmx_conv2d (no post-process: also for ablation builds whose results are wrong)."""
    code = r'''
import importlib, json, sys, numpy as np
sys.path.insert(0, %r)
native = importlib.import_module(%r + '.native')
native.LIB_PATH = sys.argv[1]
eng = native.Engine(0, max_batch=64, max_h=368, max_w=368)
eng.set_option('conv_algo', 1)
rng = np.random.default_rng(0)
res = {}
for name, (B, cin, hw, cout, k, pool) in {'7x7_128_46': (64, 128, 46, 128, 7, 0), '7x7_192_46': (64, 192, 46, 128, 7, 0), 'conv2_2': (32, 128, 184, 128, 3, 1),
                                           'conv3_2': (32, 256, 92, 256, 3, 0), 'conv2_1': (32, 64, 184, 128, 3, 0), 'conv4_2': (32, 512, 46, 512, 3, 0)}.items():
    x = np.maximum(rng.standard_normal((B, cin, hw, hw)), 0).astype('f')
    w = (rng.standard_normal((cout, cin, k, k)) / np.sqrt(cin * k * k)).astype('f')
    b = rng.standard_normal(cout).astype('f')
    y, ms = eng.conv2d(x, w, b, relu=True, pool=bool(pool), iters=int(sys.argv[2]))
    res[name] = ms
print(json.dumps(res))
''' % (ROOT, PKG)
    res = {}
    for lib in sorted(glob.glob(os.path.join(OUT, 'libpose_var_*.so'))):
        tag = os.path.basename(lib)[len('libpose_var_'):-3]
        r = subprocess.run([sys.executable, '-c', code, lib, str(iters)], capture_output=True, text=True, timeout=300)
        if r.returncode:
            print(tag, 'FAILED', r.stderr[-400:]); continue
        res[tag] = json.loads(r.stdout.strip().splitlines()[-1])
        print('%-12s' % tag, ' '.join('%s %.4f' % kv for kv in res[tag].items()))
        sys.stdout.flush()
    if out_json:
        json.dump(res, open(out_json, 'w'), indent=1)


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'build':
        build(sys.argv[2:])
    elif len(sys.argv) > 1 and sys.argv[1] == 'time-conv':
        import argparse
        ap = argparse.ArgumentParser(); ap.add_argument('cmd'); ap.add_argument('--iters', type=int, default=5); ap.add_argument('--json', default=None)
        a = ap.parse_args()
        time_conv(a.iters, a.json)
    elif len(sys.argv) > 1 and sys.argv[1] == 'time':
        import argparse
        ap = argparse.ArgumentParser(); ap.add_argument('cmd'); ap.add_argument('--steps', type=int, default=5); ap.add_argument('--json', default=None); ap.add_argument('--batch', type=int, default=32)
        a = ap.parse_args()
        time_all(a.steps, a.json, a.batch)
    else:
        print(__doc__)
