"""ctypes binding of libpose_mi355x.so (C ABI declared in include/pose_mi355x.h) + in-tree build.

There is deliberately NO fallback: if the shared library is missing or no gfx950 device is visible, the
calls raise -- the product path never routes through NumPy/torch or the oracle.
"""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np

# ROCm maps a process's HIP streams onto at most GPU_MAX_HW_QUEUES hardware queues per priority level (default 4); streams that share a
# queue run one after the other.  detect_precise keeps four scales in flight on four prioritised streams next to the context's own, a copy
# stream and whatever the application (torch ...) has created, and how its chains interleave depends on this number: measured on five boxes
# of the pool, the precise leg alone in a process / inside bench.py's process (profiles/r06_hw_queues.json): 1 queue 19.4 / 19.4 ms per
# 482 x 642 image, 2 queues 14.1 / 14.0, 3 queues 16.2 / 14.2, 4 (the default) 16.2 / 19.5, 8 queues 16.0 / 14.0 -- two is the only
# setting that gives the fast interleaving in both kinds of process; the batch, single-image and mixed-batch paths do not depend on it.
# The variable is read when the HIP runtime initialises, i.e. at the first HIP call of the process: set here, at import, unless the user
# has set it.
os.environ.setdefault('GPU_MAX_HW_QUEUES', '2')

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB_PATH = os.path.join(CSRC, 'libpose_mi355x.so')
HEADER = os.path.join(os.path.dirname(HERE), 'include', 'pose_mi355x.h')
SOURCES = [('pmx_api.hip', []), ('pmx_precise.hip', []), ('pmx_multi.hip', []), ('conv_mfma.hip', []), ('conv_wino.hip', ['-mllvm', '-pragma-unroll-threshold=200000']), ('conv1_wino.hip', []), ('conv_select.hip', []), ('prep.hip', ['-ffp-contract=off']),
           ('postproc.hip', ['-ffp-contract=off'])]
# the opt-in bf16x3 kernels (option "precision" = 1; DESIGN.md 4.1.5: frozen, slower than the fp32 Winograd path) are NOT part of the
# default library: PMX_BUILD_BF16X3=1 in the environment of the build adds their translation unit (the stamp then differs, so the
# library is rebuilt when the variable changes)
if os.environ.get('PMX_BUILD_BF16X3', '') not in ('', '0'):
    SOURCES.append(('conv_bf16x3.hip', []))
HEADERS = ['pmx_common.h', 'pmx_ctx.h', 'wino_util.h', 'conv_direct.h', HEADER]

N_JOINTS, N_LIMBS, N_PAF, N_HEAT = 18, 19, 38, 19
# initial capacities of a context (PMX_INIT_* in the header); they grow on demand, results are never truncated
INIT_PEAKS_PER_JOINT, INIT_SUBSETS, INIT_PEOPLE = 128, 128, 64
SNAPSHOT_SLOTS = 4                             # PMX_SNAPSHOT_SLOTS: results_snapshot slots of a context

IMG_PEAK_OVERFLOW, IMG_CAND_OVERFLOW, IMG_SUBSET_OVERFLOW, IMG_TRIPLE_MATCH, IMG_PEOPLE_OVERFLOW = 1, 2, 4, 8, 16


def result_dtype(people_cap):
    """NumPy view of one result record (include/pose_mi355x.h): pmx_image_info | scores[people_cap] | poses[people_cap][18][3]."""
    return np.dtype([
        ('n_people', np.int32), ('n_peaks', np.int32), ('status', np.int32), ('n_subsets_raw', np.int32),
        ('scores', np.float64, (int(people_cap),)), ('poses', np.float64, (int(people_cap), N_JOINTS, 3))])


RESULT_DTYPE = result_dtype(INIT_PEOPLE)       # layout at the initial person capacity


class PmxImage(C.Structure):
    """include/pose_mi355x.h::pmx_image -- one image of a mixed-size batch (pmx_detect_images)."""
    _fields_ = [('bgr', C.c_void_p), ('src_h', C.c_int), ('src_w', C.c_int), ('net_h', C.c_int), ('net_w', C.c_int),
                ('map_h', C.c_int), ('map_w', C.c_int)]


class PmxError(RuntimeError):
    def __init__(self, code, msg):
        RuntimeError.__init__(self, 'libpose_mi355x error %d: %s' % (code, msg))
        self.code = code


def _hipcc():
    for p in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if p and (os.path.isabs(p) and os.path.exists(p) or not os.path.isabs(p)):
            return p
    return 'hipcc'


STAMP_PATH = os.path.join(CSRC, '.build_stamp')


BASE_FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC']
_hipcc_version_cache = {}


def hipcc_version():
    """First lines of `hipcc --version` (part of the build stamp: another compiler = another library)."""
    hipcc = _hipcc()
    if hipcc not in _hipcc_version_cache:
        try:
            out = subprocess.run([hipcc, '--version'], capture_output=True, text=True, timeout=120).stdout
        except (OSError, subprocess.SubprocessError):
            out = ''
        _hipcc_version_cache[hipcc] = '\n'.join(l for l in out.splitlines() if 'version' in l.lower())
    return _hipcc_version_cache[hipcc]


def source_digest():
    """sha256 over every source / header the library is built from, the compile flags (base + per file) and the compiler version:
    what the .so on disk must have been built from."""
    import hashlib
    h = hashlib.sha256()
    h.update((' '.join(BASE_FLAGS) + '\n' + hipcc_version() + '\n').encode())
    for src, extra in SOURCES:
        h.update(('%s %s\n' % (src, ' '.join(extra))).encode())
        h.update(open(os.path.join(CSRC, src), 'rb').read())
    for hd in HEADERS:
        h.update(open(hd if os.path.isabs(hd) else os.path.join(CSRC, hd), 'rb').read())
    return h.hexdigest()


def object_digest(src, extra):
    """sha256 over what ONE object file is compiled from: its source, every header, its flags, the compiler.  build() keeps it next to the
    object (csrc/.<name>.o.stamp) and compiles only the translation units whose digest changed (conv_wino.hip alone is three minutes)."""
    import hashlib
    h = hashlib.sha256()
    h.update((' '.join(BASE_FLAGS + list(extra)) + '\n' + hipcc_version() + '\n').encode())
    for path in _local_includes(os.path.join(CSRC, src)):
        h.update(open(path, 'rb').read())
    return h.hexdigest()


def _local_includes(path, seen=None):
    """`path` and every file it reaches through #include "..." (the project's own headers: quoted includes, resolved against the
    including file's directory), in a fixed order."""
    seen = [] if seen is None else seen
    path = os.path.normpath(path)
    if path in seen or not os.path.exists(path):
        return seen
    seen.append(path)
    for m in re.finditer(r'^\s*#\s*include\s+"([^"]+)"', open(path, errors='replace').read(), flags=re.M):
        _local_includes(os.path.join(os.path.dirname(path), m.group(1)), seen)
    return seen


def needs_build():
    """True unless csrc/libpose_mi355x.so exists AND was built from exactly the sources on disk (content digest in csrc/.build_stamp,
    written by build(): file times say nothing after a checkout or a copy to another box)."""
    if not os.path.exists(LIB_PATH) or not os.path.exists(STAMP_PATH):
        return True
    try:
        return open(STAMP_PATH).read().strip() != source_digest()
    except OSError:
        return True


def build(force=False, verbose=False):
    """Compile the HIP sources for gfx950 into csrc/libpose_mi355x.so (cross-compiles without a GPU)."""
    if not force and not needs_build():
        return LIB_PATH
    # several ranks of one node may get here at once: serialise on a lock file, re-check under the lock
    import fcntl
    with open(os.path.join(CSRC, '.build.lock'), 'w') as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not needs_build():
                return LIB_PATH
            return _build_locked(verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(verbose):
    hipcc = _hipcc()
    objs, cmds = [], []
    # the stamp records what the compiler was GIVEN: digest taken before the (minutes-long) compile, written after a successful link -- a
    # source edited meanwhile leaves a stamp that does not match the tree, i.e. needs_build() stays true
    digest = source_digest()
    base = [hipcc] + BASE_FLAGS
    stamps = []
    for src, extra in SOURCES:
        obj = os.path.join(CSRC, src.replace('.hip', '.o'))
        objs.append(obj)
        stamp, dg = os.path.join(CSRC, '.' + os.path.basename(obj) + '.stamp'), object_digest(src, extra)
        try:
            fresh = os.path.exists(obj) and open(stamp).read().strip() == dg
        except OSError:
            fresh = False
        if fresh and not os.environ.get('PMX_BUILD_ALL'):
            continue
        if os.path.exists(stamp):
            os.remove(stamp)
        cmds.append(base + extra + ['-c', os.path.join(CSRC, src), '-o', obj])
        stamps.append((stamp, dg))
    # the translation units are independent: compile them side by side (conv_wino.hip alone is most of a serial build: 4.6 -> 3.7 min here)
    from concurrent.futures import ThreadPoolExecutor

    def compile_one(cmd):
        if verbose:
            print(' '.join(cmd))
        r = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True)
        if r.returncode:
            raise RuntimeError('%s failed (%d):\n%s' % (' '.join(cmd), r.returncode, (r.stdout + r.stderr)[-4000:]))
    try:
        jobs = max(1, min(max(1, len(cmds)), len(os.sched_getaffinity(0))))
    except (AttributeError, OSError):
        jobs = 2
    with ThreadPoolExecutor(max_workers=jobs) as pool:
        list(pool.map(compile_one, cmds))
    for stamp, dg in stamps:
        with open(stamp, 'w') as f:
            f.write(dg + '\n')
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB_PATH] + objs
    if verbose:
        print(' '.join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    with open(STAMP_PATH, 'w') as f:
        f.write(digest + '\n')
    return LIB_PATH


def header_symbols():
    """Names of every function include/pose_mi355x.h declares."""
    txt = open(HEADER).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(pmx_[a-z0-9_]+)\s*\(', txt)))


_lib = None


def has_bf16x3():
    """Does the loaded library carry the opt-in bf16x3 kernels (conv_bf16x3.hip, built only with PMX_BUILD_BF16X3=1)?"""
    try:
        C.c_void_p.in_dll(load(), 'conv_bf16x3_probe')
        return True
    except ValueError:
        return False


def load():
    """Load the shared library (building is explicit: call build() first).  Raises if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError('%s not found: build it with native.build() / __graft_entry__.build(); '
                           'there is no CPU fallback for the product path' % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    vp, ci, cd = C.c_void_p, C.c_int, C.c_double
    fp = C.POINTER(C.c_float)
    dp = C.POINTER(C.c_double)
    ip = C.POINTER(C.c_int)
    sig = {
        'pmx_version': (C.c_char_p, []),
        'pmx_last_error': (C.c_char_p, []),
        'pmx_device_count': (ci, [ip]),
        'pmx_create': (ci, [C.POINTER(vp), ci, ci, ci, ci]),
        'pmx_create_net': (ci, [C.POINTER(vp), C.c_char_p, ci, ci, ci, ci]),
        'pmx_keypoints': (ci, [vp, ci, ci, ci, cd, vp]),
        'pmx_precise_begin': (ci, [vp, ci, ci]),
        'pmx_precise_add_scale': (ci, [vp, vp, ci, ci]),
        'pmx_precise_finish': (ci, [vp]),
        'pmx_precise_begin_batch': (ci, [vp, ci, ci, ci]),
        'pmx_precise_add_scale_batch': (ci, [vp, vp, ci, ci]),
        'pmx_precise_add_scale_at': (ci, [vp, vp, ci, ci, ci]),
        'pmx_precise_table_stats': (ci, [vp, ip, ip]),
        'pmx_destroy': (None, [vp]),
        'pmx_set_stream': (ci, [vp, vp]),
        'pmx_synchronize': (ci, [vp]),
        'pmx_set_option': (ci, [vp, C.c_char_p, ci]),
        'pmx_set_layer': (ci, [vp, C.c_char_p, vp, vp, ci, ci, ci]),
        'pmx_weights_missing': (ci, [vp, ip]),
        'pmx_forward_u8': (ci, [vp, vp, ci, ci, ci, ci]),
        'pmx_forward_f32': (ci, [vp, vp, ci, ci, ci, ci]),
        'pmx_forward_u8_resized': (ci, [vp, vp, ci, ci, ci, ci, ci, ci]),
        'pmx_get_resized': (ci, [vp, vp, ci, ci, ci]),
        'pmx_get_maps': (ci, [vp, vp, vp]),
        'pmx_set_maps': (ci, [vp, vp, vp, ci, ci, ci]),
        'pmx_set_gaussian': (ci, [vp, vp, ci]),
        'pmx_postprocess': (ci, [vp, ci, ci, ci, cd, vp]),
        'pmx_detect_batch': (ci, [vp, vp, ci, ci, ci, ci, ci, ci, cd, vp]),
        'pmx_detect_images': (ci, [vp, vp, ci]),
        'pmx_forward_u8_images': (ci, [vp, vp, vp, ci, ci]),
        'pmx_postprocess_images': (ci, [vp, vp, ci, vp]),
        'pmx_get_image_maps': (ci, [vp, ci, vp, vp, ci, ci]),
        'pmx_results_layout': (ci, [vp, ip, C.POINTER(C.c_size_t)]),
        'pmx_get_results': (ci, [vp, ci, vp, C.c_size_t]),
        'pmx_results_device_ptr': (ci, [vp, C.POINTER(vp), C.POINTER(C.c_size_t)]),
        'pmx_results_snapshot': (ci, [vp, ci, vp, C.c_size_t]),
        'pmx_snapshot_wait': (ci, [vp, ci, ip, ip, C.POINTER(C.c_size_t), ip]),
        'pmx_set_capacities': (ci, [vp, ci, ci, ci, ci]),
        'pmx_get_capacities': (ci, [vp, ip, ip, ip, ip]),
        'pmx_get_peaks': (ci, [vp, ci, vp, ci, ip]),
        'pmx_get_connections': (ci, [vp, ci, vp, ci, ip]),
        'pmx_get_subsets': (ci, [vp, ci, vp, ci, ip]),
        'pmx_get_smoothed': (ci, [vp, ci, ci, vp, ci, ci]),
        'pmx_timer_start': (ci, [vp]),
        'pmx_timer_stop': (ci, [vp, dp]),
        'pmx_profile_enable': (ci, [vp, ci]),
        'pmx_profile_reset': (ci, [vp]),
        'pmx_profile_count': (ci, [vp, ip]),
        'pmx_profile_entry': (ci, [vp, ci, C.c_char_p, ci, dp, C.POINTER(C.c_int64), dp, dp]),
        'pmx_profile_issued': (ci, [vp, ci, dp]),
        'pmx_conv2d': (ci, [vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, ci, ci, vp, ci, dp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)     # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    lib._pmx_sig = sig
    _lib = lib
    return lib


def device_count():
    n = C.c_int(0)
    load().pmx_device_count(C.byref(n))
    return n.value


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def gaussian_taps(sigma, truncate=4.0):
    """The taps scipy.ndimage.gaussian_filter(sigma) uses (scipy `_gaussian_kernel1d`, order 0), computed
    with NumPy exactly as SciPy does at run time -- the reference calls it at pose_detector.py:86."""
    radius = int(truncate * float(sigma) + 0.5)
    sigma2 = sigma * sigma
    x = np.arange(-radius, radius + 1)
    phi = np.exp(-0.5 / sigma2 * x ** 2)
    return phi / phi.sum(), radius


class Engine(object):
    """One libpose_mi355x context (one GPU, one stream)."""

    N_MAPS = {'posenet': 19, 'facenet': 71, 'handnet': 22}

    def __init__(self, device=0, max_batch=1, max_h=368, max_w=368, gaussian_sigma=2.5, arch='posenet'):
        self.lib = load()
        self._ctx = C.c_void_p()
        self.arch = arch
        self.n_heat = self.N_MAPS[arch]
        self._check(self.lib.pmx_create_net(C.byref(self._ctx), arch.encode(), int(device), int(max_batch), int(max_h), int(max_w)))
        self.device, self.max_batch, self.max_h, self.max_w = device, max_batch, max_h, max_w
        taps, radius = gaussian_taps(gaussian_sigma)
        taps = np.ascontiguousarray(taps, dtype=np.float64)
        self._check(self.lib.pmx_set_gaussian(self._ctx, _ptr(taps), radius))
        self._B = 0
        # host-side record of what was installed, so that a larger context can take over (PoseDetector._grow)
        self._layers, self._options, self._stream_ptr, self._caps_set = {}, {}, None, None

    def _check(self, rc):
        if rc != 0:
            raise PmxError(rc, self.lib.pmx_last_error().decode('utf-8', 'replace'))

    def close(self):
        if getattr(self, '_ctx', None) is not None and self._ctx.value:
            self.lib.pmx_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- options / weights -------------------------------------------------------------------------
    def set_option(self, key, value):
        self._check(self.lib.pmx_set_option(self._ctx, key.encode(), int(value)))
        self._options[key] = int(value)

    def set_stream(self, stream_ptr):
        self._check(self.lib.pmx_set_stream(self._ctx, C.c_void_p(stream_ptr)))
        self._stream_ptr = stream_ptr

    def state(self):
        """Weights (host copies), options, stream and (grown) capacities of this engine -- what a larger context needs to take over."""
        return dict(layers=dict(self._layers), options=dict(self._options), stream=self._stream_ptr, caps=self.capacities())

    def load_state(self, st):
        for name, (W, b) in st['layers'].items():
            self.set_layer(name, W, b)
        for k, v in st['options'].items():
            self.set_option(k, v)
        if st['stream']:
            self.set_stream(st['stream'])
        caps = st['caps']
        if (caps['peaks_per_joint'], caps['subsets'], caps['people']) != (INIT_PEAKS_PER_JOINT, INIT_SUBSETS, INIT_PEOPLE) or caps['candidates']:
            self.set_capacities(caps['peaks_per_joint'], caps['subsets'], caps['people'], caps['candidates'])

    def copy_state_to(self, other):
        """Install this engine's weights, options, stream and (grown) capacities in `other` (a larger context)."""
        other.load_state(self.state())

    def synchronize(self):
        self._check(self.lib.pmx_synchronize(self._ctx))

    def set_layer(self, name, W, b):
        W = np.ascontiguousarray(W, dtype=np.float32)
        b = np.ascontiguousarray(b, dtype=np.float32)
        co, ci, kh, kw = W.shape
        assert kh == kw and b.shape == (co,)
        self._check(self.lib.pmx_set_layer(self._ctx, name.encode(), _ptr(W), _ptr(b), co, ci, kh))
        self._layers[name] = (W, b)

    def set_weights(self, weights):
        for name, (W, b) in weights.items():
            self.set_layer(name, W, b)

    def weights_missing(self):
        n = C.c_int(0)
        self._check(self.lib.pmx_weights_missing(self._ctx, C.byref(n)))
        return n.value

    # ---- network -----------------------------------------------------------------------------------
    def forward_u8(self, imgs=None, device_ptr=None, shape=None):
        """imgs: (B, H, W, 3) uint8 BGR host array, or a device pointer + shape."""
        if device_ptr is not None:
            B, H, W = shape
            self._check(self.lib.pmx_forward_u8(self._ctx, C.c_void_p(device_ptr), B, H, W, 1))
        else:
            imgs = np.ascontiguousarray(imgs, dtype=np.uint8)
            B, H, W, c3 = imgs.shape
            assert c3 == 3
            self._check(self.lib.pmx_forward_u8(self._ctx, _ptr(imgs), B, H, W, 0))
        self._B = B
        self._fhw = (H // 8, W // 8)

    def forward_u8_resized(self, imgs, h, w):
        """imgs (B, H0, W0, 3) uint8 -> cv2.resize(INTER_LINEAR)-equivalent to (h, w) on the device -> forward."""
        imgs = np.ascontiguousarray(imgs, dtype=np.uint8)
        B, H0, W0, c3 = imgs.shape
        assert c3 == 3
        self._check(self.lib.pmx_forward_u8_resized(self._ctx, _ptr(imgs), B, H0, W0, int(h), int(w), 0))
        self._B = B
        self._fhw = (int(h) // 8, int(w) // 8)

    def get_resized(self, h, w):
        out = np.empty((self._B, int(h), int(w), 3), np.uint8)
        self._check(self.lib.pmx_get_resized(self._ctx, _ptr(out), self._B, int(h), int(w)))
        return out

    def resize_u8(self, imgs, h, w):
        """Device resize only needs a context with weights for the forward that follows it in forward_u8_resized; for
        the `model=` seam (no weights) run the resize kernel through the same entry and fetch the resized batch."""
        imgs = np.ascontiguousarray(imgs, dtype=np.uint8)
        B, H0, W0, _ = imgs.shape
        rc = self.lib.pmx_forward_u8_resized(self._ctx, _ptr(imgs), B, H0, W0, int(h), int(w), 0)
        if rc not in (0, 4):          # 4 = PMX_ERR_WEIGHTS: resize done, network skipped (no weights loaded)
            self._check(rc)
        self._B = B
        return self.get_resized(h, w)

    def forward_f32(self, x):
        x = np.ascontiguousarray(x, dtype=np.float32)
        B, c3, H, W = x.shape
        assert c3 == 3
        self._check(self.lib.pmx_forward_f32(self._ctx, _ptr(x), B, H, W, 0))
        self._B = B
        self._fhw = (H // 8, W // 8)

    def get_maps(self):
        """posenet: (paf, heat); facenet / handnet: heat only (B, 71 | 22, h, w)."""
        fh, fw = self._fhw if self._fhw is not None else (1, 1)      # (a mixed batch has no common size: the library refuses, see image_maps)
        heat = np.empty((self._B, self.n_heat, fh, fw), np.float32)
        if self.arch != 'posenet':
            self._check(self.lib.pmx_get_maps(self._ctx, None, _ptr(heat)))
            return heat
        paf = np.empty((self._B, N_PAF, fh, fw), np.float32)
        self._check(self.lib.pmx_get_maps(self._ctx, _ptr(paf), _ptr(heat)))
        return paf, heat

    def set_heat(self, heat):
        """facenet / handnet test seam: install last-stage heat maps (B, 71 | 22, h, w)."""
        heat = np.ascontiguousarray(heat, dtype=np.float32)
        B, c, fh, fw = heat.shape
        assert c == self.n_heat and self.arch != 'posenet'
        self._check(self.lib.pmx_set_maps(self._ctx, None, _ptr(heat), B, fh, fw))
        self._B = B
        self._fhw = (fh, fw)

    def precise_begin(self, orig_h, orig_w, n_images=1):
        self._check(self.lib.pmx_precise_begin_batch(self._ctx, int(n_images), int(orig_h), int(orig_w)))
        self._precise_hw = (int(orig_h), int(orig_w))
        self._precise_n = int(n_images)
        # host buffers handed to add_scale: the C ABI wants them alive and unchanged until a synchronising call and recognises "the same
        # images" by their address -- so the engine keeps the contiguous copy it made of the caller's array for the whole sequence
        # (released by the next precise_begin) and hands the SAME copy to every add_scale that gets the same caller object
        self._precise_src_obj = None
        self._precise_src_arr = None

    def precise_add_scale(self, img_u8, scaled_h, scaled_w, slot=None):
        """img_u8: (H, W, 3) for a batch of one, (n, H, W, 3) for the n images precise_begin announced.  slot: the scale's position in the
        reference's loop (the parts are summed in slot order); None = the next free one."""
        if self._precise_src_obj is img_u8 and self._precise_src_arr is not None:
            img = self._precise_src_arr
        else:
            img = np.ascontiguousarray(img_u8, dtype=np.uint8)
            self._precise_src_obj, self._precise_src_arr = img_u8, img
        assert img.shape[-3:-1] == self._precise_hw and img.size == self._precise_n * self._precise_hw[0] * self._precise_hw[1] * 3, img.shape
        if slot is None:
            self._check(self.lib.pmx_precise_add_scale_batch(self._ctx, _ptr(img), int(scaled_h), int(scaled_w)))
        else:
            self._check(self.lib.pmx_precise_add_scale_at(self._ctx, _ptr(img), int(scaled_h), int(scaled_w), int(slot)))

    def precise_table_stats(self):
        """(cached cubic tables, times the cache was started over) -- include/pose_mi355x.h::pmx_precise_table_stats."""
        n, t = C.c_int(0), C.c_int(0)
        self._check(self.lib.pmx_precise_table_stats(self._ctx, C.byref(n), C.byref(t)))
        return n.value, t.value

    def precise_finish(self):
        self._check(self.lib.pmx_precise_finish(self._ctx))
        self._B = self._precise_n
        self._fhw = self._precise_hw

    def keypoints(self, out_h, out_w, thresh):
        """facenet / handnet: (B, maps - 1, 4) float64 rows (x, y, confidence, valid)."""
        out = np.empty((self._B, self.n_heat - 1, 4), np.float64)
        self._check(self.lib.pmx_keypoints(self._ctx, self._B, int(out_h), int(out_w), float(thresh), _ptr(out)))
        self._map = (int(out_h), int(out_w))
        return out

    def set_maps(self, paf, heat):
        paf = np.ascontiguousarray(paf, dtype=np.float32)
        heat = np.ascontiguousarray(heat, dtype=np.float32)
        B, cp, fh, fw = paf.shape
        assert cp == N_PAF and heat.shape == (B, N_HEAT, fh, fw)
        self._check(self.lib.pmx_set_maps(self._ctx, _ptr(paf), _ptr(heat), B, fh, fw))
        self._B = B
        self._fhw = (fh, fw)

    # ---- mixed-size batches (pmx_detect_images and its two halves) ------------------------------------
    def detect_images(self, imgs, net_hw, map_hw):
        """imgs: list of H x W x 3 uint8 BGR arrays of ANY sizes; net_hw / map_hw: per image (h, w) of the network input / the up-sampled
        maps.  cv2.resize (:493), network and post-process on the device, one launch per layer over all images; records (results()) in
        image order."""
        keep = [np.ascontiguousarray(im, dtype=np.uint8) for im in imgs]
        B = len(keep)
        arr = (PmxImage * B)()
        for i, im in enumerate(keep):
            assert im.ndim == 3 and im.shape[2] == 3
            arr[i].bgr = im.ctypes.data
            arr[i].src_h, arr[i].src_w = im.shape[0], im.shape[1]
            arr[i].net_h, arr[i].net_w = int(net_hw[i][0]), int(net_hw[i][1])
            arr[i].map_h, arr[i].map_w = int(map_hw[i][0]), int(map_hw[i][1])
        self._check(self.lib.pmx_detect_images(self._ctx, arr, B))
        self._B = B
        self._fhw = None
        self._map = None
        self._img_fhw = [(int(h) // 8, int(w) // 8) for h, w in net_hw]
        self._img_map = [(int(h), int(w)) for h, w in map_hw]

    def forward_u8_images(self, imgs):
        """the network alone on uint8 images already at their network sizes (multiples of 8)"""
        keep = [np.ascontiguousarray(im, dtype=np.uint8) for im in imgs]
        flat = np.concatenate([k.reshape(-1) for k in keep])
        hw = np.ascontiguousarray([[k.shape[0], k.shape[1]] for k in keep], dtype=np.int32)
        self._check(self.lib.pmx_forward_u8_images(self._ctx, _ptr(flat), _ptr(hw), len(keep), 0))
        self._B = len(keep)
        self._fhw = None
        self._img_fhw = [(k.shape[0] // 8, k.shape[1] // 8) for k in keep]

    def postprocess_images(self, map_hw, scale_xy=None):
        hw = np.ascontiguousarray(map_hw, dtype=np.int32).reshape(-1, 2)
        sp = None
        if scale_xy is not None:
            scale_xy = np.ascontiguousarray(scale_xy, dtype=np.float64).reshape(len(hw), 2)
            sp = _ptr(scale_xy)
        self._check(self.lib.pmx_postprocess_images(self._ctx, _ptr(hw), len(hw), sp))
        self._map = None
        self._img_map = [(int(h), int(w)) for h, w in hw]

    def image_maps(self, image, fh=None, fw=None):
        """(paf 38 x fh x fw, heat 19 x fh x fw) of ONE image of the current batch (uniform or mixed)."""
        if fh is None:
            fh, fw = self._img_fhw[image] if self._fhw is None else self._fhw
        paf = np.empty((N_PAF, fh, fw), np.float32)
        heat = np.empty((N_HEAT, fh, fw), np.float32)
        self._check(self.lib.pmx_get_image_maps(self._ctx, int(image), _ptr(paf), _ptr(heat), int(fh), int(fw)))
        return paf, heat

    # ---- post-process ------------------------------------------------------------------------------
    def postprocess(self, map_h, map_w, img_len, scale_xy=None):
        sp = None
        if scale_xy is not None:
            scale_xy = np.ascontiguousarray(scale_xy, dtype=np.float64).reshape(self._B, 2)
            sp = _ptr(scale_xy)
        self._check(self.lib.pmx_postprocess(self._ctx, self._B, int(map_h), int(map_w), float(img_len), sp))
        self._map = (int(map_h), int(map_w))

    def detect_batch(self, imgs=None, map_h=320, map_w=320, img_len=None, scale_xy=None, device_ptr=None, shape=None):
        if device_ptr is not None:
            B, H, W = shape
            p, on_dev = C.c_void_p(device_ptr), 1
        else:
            imgs = np.ascontiguousarray(imgs, dtype=np.uint8)
            B, H, W, _ = imgs.shape
            p, on_dev = _ptr(imgs), 0
        sp = None
        if scale_xy is not None:
            scale_xy = np.ascontiguousarray(scale_xy, dtype=np.float64).reshape(B, 2)
            sp = _ptr(scale_xy)
        self._check(self.lib.pmx_detect_batch(self._ctx, p, B, H, W, on_dev, int(map_h), int(map_w),
                                              float(map_w if img_len is None else img_len), sp))
        self._B = B
        self._fhw = (H // 8, W // 8)
        self._map = (int(map_h), int(map_w))

    def set_capacities(self, peaks_per_joint=0, subsets=0, people=0, candidates=0):
        """Pre-size the post-process buffers (0 keeps a value; candidates 0 = LDS store).  Tests shrink them to exercise the
        growth path; crowds can pre-size to avoid the one-off re-run."""
        self._check(self.lib.pmx_set_capacities(self._ctx, int(peaks_per_joint), int(subsets), int(people), int(candidates)))

    def capacities(self):
        v = [C.c_int(0) for _ in range(4)]
        self._check(self.lib.pmx_get_capacities(self._ctx, *[C.byref(x) for x in v]))
        return dict(zip(('peaks_per_joint', 'subsets', 'people', 'candidates'), [x.value for x in v]))

    def results_layout(self):
        """(people_cap, bytes_per_record) of the final records of the last post-process (synchronises; grows the capacities
        and re-runs the post-process first if an image overflowed them)."""
        cap, nbytes = C.c_int(0), C.c_size_t(0)
        self._check(self.lib.pmx_results_layout(self._ctx, C.byref(cap), C.byref(nbytes)))
        assert result_dtype(cap.value).itemsize == nbytes.value
        return cap.value, nbytes.value

    def results(self):
        """Structured array (B,) of result_dtype(people_cap) (synchronises).  ONE device round trip in the common case: the buffer is sized
        for the context's current person capacity (a host-side query) and pmx_get_results checks the status words it has just copied; only
        if an image overflowed a capacity -- the library then grows it and re-runs the post-process, so the record size changes -- the
        call comes back with PMX_ERR_CAPACITY and is repeated with the new layout (pmx_results_layout)."""
        v = [C.c_int(0) for _ in range(4)]
        self._check(self.lib.pmx_get_capacities(self._ctx, *[C.byref(x) for x in v]))
        out = np.empty(self._B, dtype=result_dtype(v[2].value))
        rc = self.lib.pmx_get_results(self._ctx, self._B, _ptr(out), out.nbytes)
        if rc == 5:                                   # PMX_ERR_CAPACITY: the person capacity grew under the call
            cap, _ = self.results_layout()
            out = np.empty(self._B, dtype=result_dtype(cap))
            rc = self.lib.pmx_get_results(self._ctx, self._B, _ptr(out), out.nbytes)
        self._check(rc)
        return out

    def results_device_ptr(self):
        """(device pointer, bytes per record) of the final records (call after results_layout())."""
        p = C.c_void_p()
        n = C.c_size_t()
        self._check(self.lib.pmx_results_device_ptr(self._ctx, C.byref(p), C.byref(n)))
        return p.value, n.value

    def results_snapshot(self, slot, dst_device_ptr, dst_bytes):
        """Enqueue (no sync) a device-to-device copy of the last post-process's records to `dst_device_ptr` and mark it with slot's event;
        the next detect_batch may be enqueued right away (include/pose_mi355x.h: pmx_results_snapshot)."""
        self._check(self.lib.pmx_results_snapshot(self._ctx, int(slot), C.c_void_p(dst_device_ptr), int(dst_bytes)))

    def snapshot_wait(self, slot):
        """Block until slot's snapshot copies are done -> (batch, people_cap, bytes_per_record, overflow): `overflow` = an image needed
        more capacity than the context had (the snapshot is not final: re-run the step through results_layout() / results())."""
        b, cap, st, rec = C.c_int(0), C.c_int(0), C.c_int(0), C.c_size_t(0)
        self._check(self.lib.pmx_snapshot_wait(self._ctx, int(slot), C.byref(b), C.byref(cap), C.byref(rec), C.byref(st)))
        overflow = bool(st.value & (IMG_PEAK_OVERFLOW | IMG_CAND_OVERFLOW | IMG_SUBSET_OVERFLOW | IMG_PEOPLE_OVERFLOW))
        return b.value, cap.value, rec.value, overflow

    def _rows(self, fn, image, width, guess):
        n = C.c_int(0)
        cap = max(int(guess), 1)
        for _ in range(2):
            buf = np.empty((cap, width), np.float64)
            rc = fn(self._ctx, image, _ptr(buf), cap, C.byref(n))
            if rc == 5 and n.value > cap:        # PMX_ERR_CAPACITY: n_rows tells the size needed
                cap = n.value
                continue
            self._check(rc)
            return buf[:n.value].copy()
        self._check(rc)

    def peaks(self, image=0):
        return self._rows(self.lib.pmx_get_peaks, image, 5, N_JOINTS * INIT_PEAKS_PER_JOINT)

    def connections(self, image=0):
        return self._rows(self.lib.pmx_get_connections, image, 4, N_LIMBS * INIT_PEAKS_PER_JOINT)

    def subsets(self, image=0):
        return self._rows(self.lib.pmx_get_subsets, image, 20, INIT_SUBSETS)

    def smoothed(self, image, joint):
        h, w = self._map
        out = np.empty((h, w), np.float32)
        self._check(self.lib.pmx_get_smoothed(self._ctx, image, joint, _ptr(out), h, w))
        return out

    # ---- measurement -------------------------------------------------------------------------------
    def timer_start(self):
        self._check(self.lib.pmx_timer_start(self._ctx))

    def timer_stop(self):
        ms = C.c_double(0)
        self._check(self.lib.pmx_timer_stop(self._ctx, C.byref(ms)))
        return ms.value

    def profile_enable(self, on=True):
        """True / 1: HIP-event pairs around every kernel launch; 2: only around the 7x7 convolutions (dominant kernel)."""
        self._check(self.lib.pmx_profile_enable(self._ctx, int(on)))

    def profile_reset(self):
        self._check(self.lib.pmx_profile_reset(self._ctx))

    def profile(self):
        """[{layer, kernel, total_ms, launches, avg_ms, flop_per_launch, issued_flop_per_launch, bytes_per_launch}]: algorithmic FLOP of the
        convolution and the FLOP the kernel issues to the matrix cores (fewer for the Winograd forms)."""
        n = C.c_int(0)
        self._check(self.lib.pmx_profile_count(self._ctx, C.byref(n)))
        out = []
        for i in range(n.value):
            name = C.create_string_buffer(128)
            ms, fl, by, iss = C.c_double(0), C.c_double(0), C.c_double(0), C.c_double(0)
            ln = C.c_int64(0)
            self._check(self.lib.pmx_profile_entry(self._ctx, i, name, 128, C.byref(ms), C.byref(ln), C.byref(fl), C.byref(by)))
            self._check(self.lib.pmx_profile_issued(self._ctx, i, C.byref(iss)))
            layer, _, kern = name.value.decode().partition('|')
            out.append(dict(layer=layer, kernel=kern, total_ms=ms.value, launches=ln.value,
                            avg_ms=ms.value / max(ln.value, 1), flop_per_launch=fl.value, issued_flop_per_launch=iss.value,
                            bytes_per_launch=by.value))
        return out

    # ---- single-layer test entry ---------------------------------------------------------------------
    def conv2d(self, x, W, b=None, relu=False, pool=False, iters=0):
        x = np.ascontiguousarray(x, dtype=np.float32)
        W = np.ascontiguousarray(W, dtype=np.float32)
        B, ci, H, Wd = x.shape
        co, ci2, k, k2 = W.shape
        assert ci == ci2 and k == k2
        bp = None
        if b is not None:
            b = np.ascontiguousarray(b, dtype=np.float32)
            bp = _ptr(b)
        y = np.empty((B, co, H // 2 if pool else H, Wd // 2 if pool else Wd), np.float32)
        ms = C.c_double(0)
        self._check(self.lib.pmx_conv2d(self._ctx, _ptr(x), _ptr(W), bp, B, ci, H, Wd, co, k, int(relu), int(pool),
                                        _ptr(y), int(iters), C.byref(ms)))
        return (y, ms.value) if iters else y
