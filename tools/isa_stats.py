#!/usr/bin/env python
"""Register / spill / instruction statistics of the kernels of one HIP source (device-only -S compile, no GPU needed).

    python tools/isa_stats.py conv_wino.hip [-DFOO=1 ...] [--keep /tmp/x.s]
    ISA_REUSE=1 python tools/isa_stats.py conv_wino.hip --keep /tmp/x.s      (re-read an existing listing)
The source is compiled with the per-file flags of the product build (native.py::SOURCES) + the flags given here; ISA_NO_PRODUCT_FLAGS=1 drops
the former (e.g. to see conv_wino.hip without the raised unroll threshold).
"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'chainer_realtime_multi-person_pose_estimation_amd', 'csrc')


def kernel_stats(listing):
    s = open(listing).read()
    md = s[s.find('amdhsa.kernels'):]
    stats = {}
    for k in re.split(r'\n  - ', md)[1:]:
        m = re.search(r'\.name:\s+(\S+)', k)
        if not m or '.vgpr_count' not in k:
            continue
        g = lambda f: int(re.search(r'\.%s:\s+(\d+)' % f, k).group(1))
        stats[m.group(1)] = dict(vgpr=g('vgpr_count'), agpr=g('agpr_count'), spill=g('vgpr_spill_count'), scratch=g('private_segment_fixed_size'))
    for name in stats:      # instruction counts of the kernel body: total, after the last MFMA, scratch traffic
        m = re.search(r'^%s:[^\n]*\n(.*?)\n\s*s_endpgm' % re.escape(name), s, flags=re.S | re.M)
        if not m:
            continue
        lines = [l.split(';')[0].strip() for l in m.group(1).split('\n')]
        lines = [l for l in lines if l and not l.startswith('.') and not l.endswith(':')]
        last = max((i for i, l in enumerate(lines) if l.startswith('v_mfma')), default=-1)
        tail = lines[last + 1:]
        stats[name].update(instr=len(lines), after_last_mfma=len(tail), mfma=sum(l.startswith('v_mfma') for l in lines),
                           scratch_ld=sum(l.startswith('scratch_load') for l in lines), scratch_st=sum(l.startswith('scratch_store') for l in lines),
                           tail_stores=sum(l.startswith(('buffer_store', 'global_store')) for l in tail))
    return stats


def main():
    args = sys.argv[1:]
    keep = None
    if '--keep' in args:
        i = args.index('--keep'); keep = args[i + 1]; del args[i:i + 2]
    src = args[0]
    flags = [a for a in args[1:] if a.startswith('-')]
    out = keep or '/tmp/isa_stats.s'
    if not os.environ.get('ISA_REUSE') or not os.path.exists(out):
        # the per-file flags of the product build (native.py::SOURCES), unless ISA_NO_PRODUCT_FLAGS=1 asks for the bare compile
        sys.path.insert(0, ROOT)
        import importlib
        native = importlib.import_module('chainer_realtime_multi-person_pose_estimation_amd.native')
        extra = [] if os.environ.get('ISA_NO_PRODUCT_FLAGS') else list(dict(native.SOURCES).get(src, []))
        subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '--cuda-device-only', '-S', '-o', out] + flags + extra +
                              [os.path.join(CSRC, src)], cwd=CSRC, stderr=subprocess.DEVNULL)
    for name, d in kernel_stats(out).items():
        dm = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip().replace('conv_wino_kernel', 'wino').replace('void ', '')
        print('%-44s %s' % (dm[:44], ' '.join('%s=%s' % kv for kv in d.items())))


if __name__ == '__main__':
    main()
