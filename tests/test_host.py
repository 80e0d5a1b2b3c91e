"""CPU: host-side logic of the product package and the C-ABI surface (no compute calls without a GPU)."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, pkg


def test_entity_constants_match_reference_golden():
    ent = pkg('entity')
    g = json.load(open(os.path.join(GOLDEN, 'host_fns.json')))['entity']
    assert {j.name: int(j) for j in ent.JointType} == g['JointType']
    assert [[int(a), int(b)] for a, b in ent.params['limbs_point']] == g['limbs_point']
    for k, v in g['params'].items():
        assert ent.params[k] == v, k


def test_c_constants_match_python_constants():
    ent = pkg('entity')
    native = pkg('native')
    hdr = open(os.path.join(os.path.dirname(native.HERE), 'include', 'pose_mi355x.h')).read()
    common = open(os.path.join(native.CSRC, 'pmx_common.h')).read()
    import re

    def define(txt, name):
        return re.search(r'#define\s+%s\s+([^\s/]+)' % name, txt).group(1)
    assert int(define(hdr, 'PMX_N_JOINTS')) == len(ent.JointType) == native.N_JOINTS
    assert int(define(hdr, 'PMX_N_LIMBS')) == len(ent.params['limbs_point']) == native.N_LIMBS
    assert int(define(hdr, 'PMX_INIT_PEAKS_PER_JOINT')) == native.INIT_PEAKS_PER_JOINT
    assert int(define(hdr, 'PMX_INIT_SUBSETS')) == native.INIT_SUBSETS
    assert int(define(hdr, 'PMX_INIT_PEOPLE')) == native.INIT_PEOPLE
    assert float(define(common, 'PMX_HEATMAP_PEAK_THRESH').rstrip('f')) == ent.params['heatmap_peak_thresh']
    assert int(define(common, 'PMX_N_INTEG_POINTS')) == ent.params['n_integ_points']
    assert int(define(common, 'PMX_N_INTEG_POINTS_THRESH')) == ent.params['n_integ_points_thresh']
    assert float(define(common, 'PMX_INNER_PRODUCT_THRESH')) == ent.params['inner_product_thresh']
    assert float(define(common, 'PMX_SUBSET_SCORE_THRESH')) == ent.params['subset_score_thresh']
    assert float(define(common, 'PMX_N_SUBSET_LIMBS_THRESH')) == ent.params['n_subset_limbs_thresh']
    assert float(define(common, 'PMX_GAUSS_SIGMA')) == ent.params['gaussian_sigma']
    limbs = re.search(r'PMX_LIMBS\[PMX_N_LIMBS\]\[2\] = \{(.*?)\};', common, re.S).group(1)
    pairs = [[int(a), int(b)] for a, b in re.findall(r'\{(\d+),\s*(\d+)\}', limbs)]
    assert pairs == [[int(a), int(b)] for a, b in ent.params['limbs_point']]


def test_compute_optimal_size_matches_reference_golden():
    PD = pkg('pose_detector')
    det = PD.PoseDetector.__new__(PD.PoseDetector)     # host helper only; no engine
    for h, w, target, rw, rh in json.load(open(os.path.join(GOLDEN, 'host_fns.json')))['compute_optimal_size']:
        assert det.compute_optimal_size(np.zeros((h, w, 3), 'uint8'), target) == (rw, rh)


def test_preprocess_matches_reference_golden():
    PD = pkg('pose_detector')
    det = PD.PoseDetector.__new__(PD.PoseDetector)
    z = np.load(os.path.join(GOLDEN, 'preprocess.npz'))
    assert np.array_equal(det.preprocess(z['img']), z['x'])


def test_layer_table_and_synthetic_weights():
    W = pkg('weights')
    t = W.layer_table()
    assert len(t) == 92 and sum(co * ci * k * k + co for _, ci, co, k in t) == 52311446
    from oracle import network_ref
    assert sorted(t) == sorted(network_ref.layer_table())
    w1, w2 = W.synthetic_weights(3), W.synthetic_weights(3)
    for k in w1:
        assert np.array_equal(w1[k][0], w2[k][0]) and np.array_equal(w1[k][1], w2[k][1])
    assert w1['Mconv1_stage2_L1'][0].shape == (128, 185, 7, 7) and w1['conv5_5_CPM_L2'][0].shape == (19, 512, 1, 1)


def test_npz_roundtrip(tmp_path):
    W = pkg('weights')
    w = W.synthetic_weights(1)
    p = str(tmp_path / 'w.npz')
    W.save_npz(p, w)
    r = W.load_npz(p)
    assert set(r) == set(w)
    assert np.array_equal(r['conv4_2'][0], w['conv4_2'][0])


def test_resize_linear_u8_oracle_identity_constant_and_torch():
    from oracle import resize_ref as RR
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (24, 32, 3), dtype=np.uint8)
    assert np.array_equal(RR.resize_linear_u8(img, 32, 24), img)
    flat = np.full((50, 70, 3), 137, np.uint8)
    assert np.all(RR.resize_linear_u8(flat, 41, 33) == 137)
    up = RR.resize_linear_u8(img, 64, 48)
    assert up.shape == (48, 64, 3) and up.dtype == np.uint8
    # same geometry as torch's half-pixel bilinear; the 11-bit fixed-point result stays within 1 grey level of it
    import torch
    t = torch.nn.functional.interpolate(torch.from_numpy(img.astype('f').transpose(2, 0, 1))[None], size=(48, 64),
                                        mode='bilinear', align_corners=False)[0].numpy().transpose(1, 2, 0)
    assert np.abs(up.astype('f') - t).max() <= 1.0


def test_library_builds_loads_and_exports_every_declared_symbol(native):
    lib = native.load()
    syms = native.header_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(lib, s), s
    assert set(syms) == set(lib._pmx_sig), set(syms) ^ set(lib._pmx_sig)
    assert b'gfx950' in lib.pmx_version()
    assert native.RESULT_DTYPE.itemsize == 16 + 8 * 64 * (1 + 18 * 3)
    assert native.result_dtype(200).itemsize == 16 + 8 * 200 * (1 + 18 * 3)       # PMX_RECORD_BYTES(people_cap)


def test_product_never_imports_oracle():
    native = pkg('native')
    for fn in os.listdir(native.HERE):
        if fn.endswith('.py'):
            src = open(os.path.join(native.HERE, fn)).read()
            assert 'import oracle' not in src and 'from oracle' not in src, fn


def test_no_device_fails_loudly(native):
    if native.device_count() > 0:
        pytest.skip('GPU present')
    with pytest.raises(native.PmxError):
        native.Engine(0)


def test_draw_person_pose_and_image_io(tmp_path):
    PD = pkg('pose_detector')
    img = np.zeros((120, 160, 3), np.uint8)
    assert PD.draw_person_pose(img, np.empty((0, 18, 3))) is img            # reference :521-522
    pose = np.zeros((1, 18, 3))
    pose[0, 1] = [80, 30, 2]     # neck
    pose[0, 8] = [70, 90, 2]     # right waist   (limb 0: neck -> right waist, colour [0, 255, 0])
    pose[0, 2] = [60, 32, 2]     # right shoulder
    pose[0, 16] = [55, 12, 2]    # right ear     (limb 9 shoulder -> ear is NOT drawn)
    out = PD.draw_person_pose(img, pose)
    assert out is not img and not img.any()
    assert tuple(out[60, 75]) == (0, 255, 0)                                  # a point on the neck-waist segment
    assert tuple(out[30, 80]) == tuple(PD.JOINT_COLORS[1])                    # joint disc drawn over the limb
    assert not out[22, 57].any()                                              # shoulder-ear limb skipped (:542)
    assert (out.reshape(-1, 3).any(axis=1)).sum() > 150
    p = str(tmp_path / 'x.png')
    PD.imwrite_bgr(p, out)
    assert np.array_equal(PD.imread_bgr(p), out)


def test_every_option_key_is_documented_in_the_header():
    """include/pose_mi355x.h documents every key pmx_set_option accepts (the boundary a maintainer reads)."""
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, 'chainer_realtime_multi-person_pose_estimation_amd', 'csrc', 'pmx_api.hip')).read()
    i = src.index('extern "C" int pmx_set_option')
    body = src[i:src.index('\n}\n', i)]
    keys = re.findall(r'strcmp\(key, "([a-z0-9_]+)"\)', body)
    assert len(keys) >= 15
    hdr = open(os.path.join(root, 'include', 'pose_mi355x.h')).read()
    for k in keys:
        documented = ('"%s"' % k) in hdr or (k.startswith('force_variant_k') and '"force_variant_k1|k3|k7"' in hdr)
        assert documented, 'option "%s" is not described in include/pose_mi355x.h' % k


def test_failed_growth_leaves_a_usable_detector(monkeypatch):
    """PoseDetector._make_engine closes the old context before the larger one exists; if creating / loading the larger one
    fails (device out of memory for an oversized image, rejected capacities), the detector must come back with a context at the
    PREVIOUS capacity carrying the saved state, and _cap must still say so."""
    PD = pkg('pose_detector')
    made = []

    class FakeEngine(object):
        def __init__(self, device, max_batch=1, max_h=368, max_w=368, gaussian_sigma=2.5, arch='posenet'):
            if max_h * max_w > 1000 * 1000:
                raise MemoryError('hipMalloc: out of memory')
            self.cap = (max_batch, max_h, max_w)
            self.loaded, self.closed, self.opts = None, False, {}
            made.append(self)

        def state(self):
            return {'layers': {'conv1_1': 'W'}, 'options': dict(self.opts), 'stream': None, 'caps': {}}

        def load_state(self, st):
            self.loaded = st

        def set_weights(self, w):
            self.loaded = {'layers': w}

        def set_option(self, k, v):
            self.opts[k] = v

        def close(self):
            self.closed = True

    monkeypatch.setattr(PD.native, 'Engine', FakeEngine)
    det = PD.PoseDetector(model={'conv1_1': 'W'}, device=0)
    first = det.engine
    assert det._cap == (1, 368, 368) and first.cap == (1, 368, 368)
    det._grow(4, 368, 368)                                   # a growth that works
    assert det._cap == (4, 368, 368) and det.engine.cap == (4, 368, 368) and first.closed and det.engine.loaded['layers'] == {'conv1_1': 'W'}
    good = det.engine
    with pytest.raises(MemoryError):
        det._grow(4, 2000, 2000)                             # a growth that fails
    assert det.engine is not None and det.engine is not good and good.closed
    assert det._cap == (4, 368, 368) and det.engine.cap == (4, 368, 368)          # back at the previous capacity ...
    assert det.engine.loaded['layers'] == {'conv1_1': 'W'}                            # ... with the saved state
    det._grow(2, 368, 368)                                   # and the early return of _grow is still truthful
    assert det.engine.cap == (4, 368, 368)


def test_unpack_results_per_image_error_behaviour():
    """The reference raises IndexError (pose_detector.py:197) for the ONE image it is called on; a batch must not lose the others:
    `return_exceptions=True` keeps the exception in that image's slot, the default raises as the reference does."""
    PD, native = pkg('pose_detector'), pkg('native')
    rec = np.zeros(3, dtype=native.result_dtype(4))
    rec[0]['n_peaks'], rec[0]['n_people'] = 7, 2
    rec[0]['scores'][:2] = [1.5, 2.5]
    rec[1]['n_peaks'], rec[1]['status'] = 9, native.IMG_TRIPLE_MATCH
    rec[2]['n_peaks'] = 0
    with pytest.raises(IndexError):
        PD.unpack_results(rec)
    out = PD.unpack_results(rec, return_exceptions=True)
    assert len(out) == 3 and isinstance(out[1], IndexError)
    assert out[0][0].shape == (2, 18, 3) and list(out[0][1]) == [1.5, 2.5]
    assert out[2][0].shape == (0, 18, 3) and out[2][1].shape == (0,)
    det = PD.PoseDetector.__new__(PD.PoseDetector)
    det.model = None
    with pytest.raises(ValueError):
        PD.PoseDetector.detect_precise_batch(det, [])


def test_needs_build_follows_the_source_digest(native, tmp_path, monkeypatch):
    """The library on disk is 'up to date' iff csrc/.build_stamp holds the digest of the sources + flags it was built from (file times say
    nothing after a checkout or a copy to the GPU box)."""
    assert not native.needs_build()
    assert open(native.STAMP_PATH).read().strip() == native.source_digest()
    stamp = tmp_path / 'stamp'
    stamp.write_text('0' * 64 + '\n')
    monkeypatch.setattr(native, 'STAMP_PATH', str(stamp))
    assert native.needs_build()
    stamp.write_text(native.source_digest() + '\n')
    assert not native.needs_build()


def test_oracle_reads_the_conv1_winograd_launch_as_conv1_2():
    """The fused conv1_1 + conv1_2 launch with conv1_2 in Winograd form is one profile entry; the twin runs conv1_1 on the direct chain and
    conv1_2 on the Winograd chain (oracle/conv_fma_ref.py::wino_layers)."""
    from oracle import conv_fma_ref as R
    prof = [{'layer': 'conv1_1+conv1_2', 'kernel': 'conv_wino1_f2x2_t16x16'}, {'layer': 'conv2_1', 'kernel': 'conv_wino_f2x2_3x3'},
            {'layer': 'conv5_4_CPM+conv5_5_CPM', 'kernel': 'conv1x1_pair_c512_n64'}, {'layer': 'Mconv1_stage2', 'kernel': 'conv_wino_f2x2_7x7r/t2m'}]
    assert R.wino_layers(prof) == {'conv1_2', 'conv2_1', 'Mconv1_stage2'}
    plan = R.splitk_plan(prof)
    assert 'conv1_2' in plan.wino and plan.wino_tails == {'Mconv1_stage2': 2}
    prof[0] = {'layer': 'conv1_1+conv1_2', 'kernel': 'conv1_fused_t8x16_n64'}
    assert 'conv1_2' not in R.wino_layers(prof)


def _kernel_metadata(lib_path, tmp_path):
    """{demangled-ish kernel name: {'scratch': bytes per lane, 'vgpr_spill': n, 'agpr': n}} read from the AMDGPU metadata notes of every
    gfx950 code object bundled into the library (llvm-objdump --offloading + llvm-readelf --notes: seconds, no compile)."""
    import shutil
    import subprocess
    llvm = '/opt/rocm/lib/llvm/bin'
    objdump, readelf = os.path.join(llvm, 'llvm-objdump'), os.path.join(llvm, 'llvm-readelf')
    if not (os.path.exists(objdump) and os.path.exists(readelf)):
        pytest.skip('llvm-objdump / llvm-readelf of the ROCm toolchain not found')
    work = str(tmp_path / 'bundles')
    os.makedirs(work)
    shutil.copy(lib_path, os.path.join(work, 'lib.so'))      # (the tool writes the bundles next to its input)
    subprocess.run([objdump, '--offloading', 'lib.so'], cwd=work, check=True, capture_output=True)
    kernels, cur = {}, {}
    for f in sorted(os.listdir(work)):
        if 'gfx950' not in f:
            continue
        notes = subprocess.run([readelf, '--notes', f], cwd=work, check=True, capture_output=True, text=True).stdout
        for line in notes.splitlines():
            # kernel entries sit at the first list level ("  - .agpr_count: ..." then "    .key: value"); argument lists are nested deeper
            if line.startswith('  - .'):
                cur = {}
                line = '    ' + line[4:]
            if line.startswith('    .') and not line.startswith('     '):
                key, _, val = line.strip().partition(':')
                cur[key] = val.strip()
                if key == '.wavefront_size' and cur.get('.name', '').startswith('_Z'):      # (the last key of an entry)
                    kernels[cur['.name']] = {'scratch': int(cur['.private_segment_fixed_size']), 'vgpr_spill': int(cur.get('.vgpr_spill_count', 0)),
                                             'agpr': int(cur.get('.agpr_count', 0))}
    return kernels


def test_no_kernel_of_the_library_lives_in_scratch(native, tmp_path):
    """Guard against a DECLINED UNROLL (EXPERIMENTS E19): the Winograd kernels keep their accumulators, weight ring and staging in register
    arrays indexed by unrolled loop counters; when the unroller declines a `#pragma unroll` (body over its size limit) those arrays move to
    kilobytes of scratch per lane and the kernel runs at a fraction of its rate -- with bit-identical results, so no parity test notices.
    Every kernel of the built library must stay under 256 bytes of scratch per lane (the largest today: 132, a few spilled registers)."""
    kernels = _kernel_metadata(native.LIB_PATH, tmp_path)
    wino = {k: v for k, v in kernels.items() if 'conv_wino_kernel' in k or 'conv1_wino_kernel' in k}
    assert len(kernels) >= 40 and len(wino) >= 15, (len(kernels), len(wino))
    heavy = {k: v for k, v in kernels.items() if v['scratch'] > 256}
    assert not heavy, heavy
    # one wave per SIMD with the whole accumulator file: every Winograd kernel holds its 16 frequency tiles in 256 AGPRs
    assert all(v['agpr'] == 256 for v in wino.values()), {k: v['agpr'] for k, v in wino.items() if v['agpr'] != 256}


def test_public_header_is_plain_c():
    """include/pose_mi355x.h is the C ABI: it must compile as C99 (and as C++) on its own -- plain pointers, sizes and one POD struct
    (pmx_image), no C++ or HIP types in any signature."""
    import shutil
    import subprocess
    native = pkg('native')
    for cc, lang, std in (('gcc', 'c', '-std=c99'), ('g++', 'c++', '-std=c++11')):
        if shutil.which(cc) is None:
            continue
        r = subprocess.run([cc, '-x', lang, std, '-fsyntax-only', '-Wall', '-Werror', native.HEADER], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
    import re
    code = re.sub(r'/\*.*?\*/', '', open(native.HEADER).read(), flags=re.S)          # declarations without the comments
    assert 'hipStream_t' not in code and 'hip/' not in code and 'torch' not in code.lower() and '#include <' in code
    assert set(re.findall(r'#include <([^>]+)>', code)) <= {'stddef.h', 'stdint.h'}


def test_package_sets_two_hardware_queues_unless_the_user_chose():
    """native.py sets GPU_MAX_HW_QUEUES=2 at import (detect_precise's lanes: profiles/r06_hw_queues.json) and leaves a user's value alone --
    checked in fresh interpreters, because the variable only counts before the HIP runtime initialises."""
    import subprocess
    import sys
    from conftest import ROOT
    code = ("import os, importlib; importlib.import_module('chainer_realtime_multi-person_pose_estimation_amd.native'); "
            "print(os.environ.get('GPU_MAX_HW_QUEUES'))")
    env = {k: v for k, v in os.environ.items() if k != 'GPU_MAX_HW_QUEUES'}
    out = subprocess.run([sys.executable, '-c', code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.strip() == '2', (out.stdout, out.stderr[-400:])
    out = subprocess.run([sys.executable, '-c', code], cwd=ROOT, env=dict(env, GPU_MAX_HW_QUEUES='5'), capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.strip() == '5', (out.stdout, out.stderr[-400:])
