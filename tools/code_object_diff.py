#!/usr/bin/env python3
"""Are the gfx950 kernels of two object files / shared libraries the same machine code?

    python tools/code_object_diff.py old.o new.o

Both files are copied to a scratch directory, their offload bundles are extracted there (`llvm-objdump --offloading` writes next to its
input: never run it inside csrc/), every kernel is disassembled and the per-kernel instruction streams are compared.  Used for refactors
that must not touch the device code (dead-switch removal, moving kernels between translation units): prints the kernels that differ,
the kernels only one side has, exit code 1 on any difference among the kernels both sides have.  No GPU needed.
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

BIN = '/opt/rocm/lib/llvm/bin'


def kernels(path, work):
    d = tempfile.mkdtemp(dir=work)
    local = os.path.join(d, 'in.o')
    shutil.copy(path, local)
    subprocess.run([os.path.join(BIN, 'llvm-objdump'), '--offloading', local], check=True, capture_output=True)
    cos = [f for f in os.listdir(d) if 'amdgcn' in f]
    out = {}
    for co in cos:
        txt = subprocess.run([os.path.join(BIN, 'llvm-objdump'), '-d', '--no-show-raw-insn', os.path.join(d, co)], check=True,
                             capture_output=True, text=True).stdout
        name = None
        for line in txt.splitlines():
            m = re.match(r'^[0-9a-f]+ <(.+)>:$', line)
            if m:
                name = m.group(1)
                out[name] = []
            elif name is not None and line.strip():
                # drop the address column and the `// 0000...` trailers: a kernel that merely moved inside the file is still the same code
                body = re.sub(r'^\s*[0-9a-f]+:\s*', '', line)
                body = re.sub(r'\s*//.*$', '', body).strip()
                if body:
                    out[name].append(body)
    return out


def main():
    a_path, b_path = sys.argv[1:3]
    with tempfile.TemporaryDirectory() as work:
        a, b = kernels(a_path, work), kernels(b_path, work)
    both = sorted(set(a) & set(b))
    differ = [k for k in both if a[k] != b[k]]
    only_a, only_b = sorted(set(a) - set(b)), sorted(set(b) - set(a))
    filt = shutil.which('c++filt') or shutil.which('llvm-cxxfilt', path=BIN)
    demangle = lambda n: (subprocess.run([filt, n], capture_output=True, text=True).stdout.strip() or n) if filt else n
    print('%d kernels / symbols on both sides, %d identical, %d differ' % (len(both), len(both) - len(differ), len(differ)))
    for k in differ:
        print('  DIFFERS  %s  (%d vs %d instructions)' % (demangle(k), len(a[k]), len(b[k])))
    for k in only_a:
        print('  only in %s: %s' % (a_path, demangle(k)))
    for k in only_b:
        print('  only in %s: %s' % (b_path, demangle(k)))
    return 1 if differ else 0


if __name__ == '__main__':
    sys.exit(main())
