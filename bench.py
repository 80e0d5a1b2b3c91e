#!/usr/bin/env python
"""bench.py -- frames/s of the MI355X-native OpenPose hot path on synthetic 368x368 batches.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run, 1 rank/GPU)

With --gpus N > 1 and no WORLD_SIZE in the environment bench.py launches its N ranks itself (re-exec under
torch.distributed.run on 127.0.0.1, one rank per GPU; it refuses when fewer than N GPUs are visible).

A "step" is one pass of the hot path -- PoseDetector.__call__ semantics for every image of one batch:
uint8 BGR NHWC images already resident in HBM -> fused preprocess -> 92-layer CocoPoseNet (47 fp32-MFMA conv
launches) -> upsample + Gaussian + NMS peaks -> PAF scoring + greedy matching -> grouping -> result records
copied to the host (for N > 1: RCCL gather of the device-resident records to rank 0, the only collective, then one copy
to the host there).  Weak scaling: every rank processes its own shard of `--batch` images of the global batch
(BASELINE.json config "Batch 256 sharded 8 x 32"); value = all images / max-rank time.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline      the dominant kernel (the 7x7 Winograd kernel, two branch groups per launch): FLOP per launch the kernel
                ISSUES to the matrix cores (engine profile) / its average launch duration, measured live with HIP events
                on the stream the kernels run on (per-launch event pairs inside the timed region), against the dense
                fp32-MFMA peak (157.3 TFLOP/s, MI355X_MICROARCH.md): `frac` <= 1.  `algorithmic_frac` = the same with the
                FLOP of the convolution the kernel computes.  `traffic` (HBM bytes/launch from PMC counters) is filled
                from profiles/ when a counter pass of this kernel exists, else null.  `step_roofline`: the same pair for
                the whole step.  tools/summarize_profiles.py recomputes both from the rocprofv3 kernel stats.
  cpu_baseline  the oracle (torch-CPU fp32 restatement of the network + NumPy restatement of the reference
                post-process, one image per call as the reference does) timed on this box's host cores, rank 0, N=1
                only, on a bounded sample of the same workload.
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

# Hardware queues of the process (read when the HIP runtime initialises, so set before anything can touch it; stated in the line as
# `hw_queues`).  One rank: the package's own default, 2 -- the setting under which detect_precise's four prioritised lanes interleave best
# both alone and inside this harness (profiles/r06_hw_queues.json).  Several ranks: 8, so that RCCL's streams never share a hardware
# queue with the compute stream (the record gather of step k runs under the convolutions of step k + 1).
def _multi_rank_argv(argv):
    if int(os.environ.get('WORLD_SIZE', '1') or 1) > 1 or '--force-gather' in argv:      # (--force-gather: the one-rank RCCL group, same setting as N > 1)
        return True
    for i, a in enumerate(argv):
        v = a.split('=', 1)[1] if a.startswith('--gpus=') else (argv[i + 1] if a == '--gpus' and i + 1 < len(argv) else None)
        if v is not None and v.isdigit() and int(v) > 1:
            return True
    return False


os.environ.setdefault('GPU_MAX_HW_QUEUES', '8' if _multi_rank_argv(sys.argv[1:]) else '2')
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
PKG = 'chainer_realtime_multi-person_pose_estimation_amd'
FP32_MFMA_PEAK_TFLOPS = 157.3          # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
FLOP_PER_FRAME = 271868013568          # 2 x MACs of the 92 convs at 368 x 368
DOMINANT_LAYERS = ('Mconv2_', 'Mconv3_', 'Mconv4_', 'Mconv5_')   # 7x7 128->128, 20 launches per step
PIPE_DEPTH = 1       # N > 1: the gather of step k is issued after step k + PIPE_DEPTH has been enqueued.  (Measured through a one-rank RCCL group:
                     # depth 1 868 frames/s, depth 2 865 -- the host-side copies / collectives of torch's streams complete late in the step that
                     # was enqueued in front of them whatever the depth, and one step in flight already keeps the GPU fed.)


def rocprof_kernel(label):
    """HIP kernel (as rocprofv3 --kernel-trace names it) behind an engine profile label -- the key the live HIP-event figures and
    the committed rocprofv3 / PMC summaries share.  Labels: csrc/pmx_api.hip::run_conv."""
    import re
    label = re.sub(r'@\d+\+\d+$', '', label)        # ("@<first image>+<count>": one half of a batch cut in two by images -- same kernels)
    m = re.match(r'conv_wino_f2x2_(\d)x\d(r?)(/[ut]\d+m?)?(:units|:combine)?$', label)      # ".../t<g>m": merged tails
    if m:
        ks, run, plan, part = m.group(1), m.group(2), m.group(3) or '', m.group(4) or ''
        if part == ':combine':
            return 'conv_wino_tail_reduce_kernel'
        if part == ':units':
            return 'conv_wino_kernel<%s, 0, 1, %d>' % (ks, 3 if plan.endswith('m') else 1)      # <KS, POOL, UNIT, GEOM>
        if plan.startswith('/u'):
            return 'conv_wino_kernel<%s, 0, 1, 0>' % ks                     # (+ conv_splitk_reduce_kernel inside the same event pair)
        return 'conv_wino_kernel<%s, 0, 0, %d>' % (ks, 1 if run else 0)     # (the pooled 3x3 layers are <3, 1, 0, 0>)
    m = re.match(r'conv(\d)x\d(_v\d)?_t(\d+)x(\d+)_n(\d+)', label)
    if m:
        if m.group(2) == '_v6':          # conv_mfma_v6_kernel<KS, MT, POOL>: 17 x 32 consecutive pixels of a 46-column slab
            return 'conv_mfma_v6_kernel<%s, %s, 0>' % (m.group(1), m.group(3))
        return 'conv_mfma%s_kernel<%s, %s, %s, %s,' % (m.group(2) or '', m.group(1), m.group(3), m.group(4), m.group(5))
    return label


def pmc_traffic(kernel_sig):
    """HBM bytes per launch of the kernel from the newest committed PMC summary (profiles/rNN_pmc_summary.json,
    written by tools/summarize_profiles.py from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the same
    workload; (2 * FETCH_SIZE + WRITE_SIZE) * 1024 per MI355X_MICROARCH.md).  Mean over all launches of the kernel,
    like `achieved`.  None if no summary names the kernel."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_summary.json')))
    for f in reversed(files):
        try:
            d = json.load(open(f))
            for k, e in d['kernels'].items():
                if kernel_sig in k and 'hbm_bytes_per_launch_mean' in e.get('derived', {}):
                    return e['derived']['hbm_bytes_per_launch_mean'], os.path.basename(f)
        except Exception:
            pass
    return None, None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--batch', type=int, default=32, help='images per GPU per step')
    ap.add_argument('--size', type=int, default=368)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-budget', type=float, default=20.0, help='seconds of CPU-baseline sampling')
    ap.add_argument('--no-profile', action='store_true', help='skip the per-launch HIP events (roofline object null)')
    ap.add_argument('--dump-profile', default=None, help='write the per-layer table to this JSON file')
    ap.add_argument('--backend', default='nccl', help="torch.distributed backend for N > 1 ('nccl' = RCCL; 'gloo' only "
                    "for the single-GPU multi-rank smoke test, where all ranks share device 0)")
    ap.add_argument('--dump-records', default=None, help='rank 0 writes the gathered result records of the last step (.npy)')
    ap.add_argument('--no-extras', action='store_true', help='skip the single-image and upload-inclusive measurements')
    ap.add_argument('--bf16x3', action='store_true', help='also run the opt-in split-bf16 mode (frozen after round 5: never the headline, no longer '
                    'part of the default run -- DESIGN.md section 7)')
    ap.add_argument('--engine-opt', action='append', default=[], metavar='KEY=VALUE', help='pmx_set_option on the engine before the run (A/B '
                    'switches such as wino_xcd_groups=0); repeatable; recorded in config.engine_options')
    ap.add_argument('--validate-images', type=int, default=0, help='N > 1 / --force-gather: records per rank that rank 0 re-computes and compares '
                    'after the timed region (0 = the whole shard)')
    ap.add_argument('--force-gather', action='store_true', help='N = 1: still create a one-rank process group and route the records '
                    'through the RCCL gather (dist.RecordPipe), the code path of N > 1')
    return ap.parse_args()


def self_launch(a):
    """`python bench.py --gpus N` without a launcher: start the N ranks here (same command line the driver uses)."""
    import socket
    import subprocess
    import torch
    n_dev = torch.cuda.device_count()
    if a.backend == 'nccl' and n_dev < a.gpus:
        raise SystemExit('bench.py: --gpus %d requested but only %d GPU(s) visible (one rank per GPU over RCCL; '
                         '--backend gloo runs the ranks on the visible GPU(s) as a smoke test)' % (a.gpus, n_dev))
    if n_dev < 1:
        raise SystemExit('bench.py: no GPU visible (there is no CPU path)')
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=%d' % a.gpus,
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd, env=env))


MASK_VARS = ('HIP_VISIBLE_DEVICES', 'ROCR_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES', 'GPU_DEVICE_ORDINAL')


def pick_device(local_rank, n_visible, world, backend, rank=0, env=None):
    """Device index of this rank.  RCCL ("nccl"): one rank per GPU -- device LOCAL_RANK when the rank sees all GPUs of the node; device 0
    when the launcher masks the GPUs per rank (one visible device each AND a visibility variable set in this rank's environment: on an
    unmasked single-GPU box every rank would land on the same GPU); otherwise there is no GPU for it.  Other backends (the gloo smoke
    mode): ranks may share the visible GPU(s)."""
    env = os.environ if env is None else env
    if backend != 'nccl':
        return local_rank % max(1, n_visible)
    if n_visible > local_rank:
        return local_rank
    if n_visible == 1 and world > 1 and any(env.get(v, '') != '' for v in MASK_VARS):
        return 0
    raise SystemExit('bench.py: rank %d has no GPU of its own (%d visible, %d ranks, none of %s set): RCCL needs one GPU per rank'
                     % (rank, n_visible, world, ' / '.join(MASK_VARS)))


def core_limits():
    """What bounds the host cores of this process: os.cpu_count(), the affinity mask, the cgroup CPU quota (containers often
    expose all host CPUs through os.cpu_count() while the quota is far smaller)."""
    lim = {'os_cpu_count': os.cpu_count() or 1, 'affinity': None, 'cgroup_quota': None}
    try:
        lim['affinity'] = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if quota != 'max':
            lim['cgroup_quota'] = max(1, int(float(quota) / float(period)))
    except Exception:
        pass
    return lim


def usable_cores():
    return max(1, min(v for v in core_limits().values() if v))


def cpu_baseline(weights, imgs, map_hw, budget_s=20.0, max_threads=32):
    """Oracle timed on the host: one image per call, as the reference does (pose_detector.py:430,501).
    Bounded sample: one warm-up frame, then as many frames of the batch as fit in ~budget_s (at least one).
    Returns (cpu_baseline object, per-frame oracle results for the keypoint-match check)."""
    import torch
    from oracle import network_ref, postprocess_ref
    lim = core_limits()
    threads = min(usable_cores(), max_threads)     # oneDNN scales poorly beyond a few tens of threads at batch 1
    bound_by = ('the %d-thread cap of this script (oneDNN at batch 1 does not scale further)' % max_threads if threads == max_threads and usable_cores() > max_threads
                else 'the cgroup CPU quota' if lim['cgroup_quota'] == threads else 'the affinity mask' if lim['affinity'] == threads else 'os.cpu_count()')
    torch.set_num_threads(threads)

    def one(i):
        x = postprocess_ref.preprocess(imgs[i % len(imgs)])
        paf, heat = network_ref.forward(weights, x)
        o = postprocess_ref.postprocess_from_net_output(paf[0], heat[0], map_hw[0], map_hw[1])
        return {k: o[k] for k in ('all_peaks', 'poses', 'scores')}
    t0 = time.perf_counter()
    one(0)                      # warm-up (thread pool, oneDNN primitives)
    warm = time.perf_counter() - t0
    results = []
    t0 = time.perf_counter()
    while True:
        results.append(one(len(results)))
        frames = len(results)
        dt = time.perf_counter() - t0
        if dt + dt / frames > budget_s or frames >= 30:
            break
    return ({'value': frames / dt, 'unit': 'frames/s', 'cores': threads, 'kind': 'port',
             'core_limits': lim,
             'sample': '%d frames (after 1 warm-up frame of %.1f s) of the same 368x368 synthetic workload, batch 1 per '
                       'call, torch-CPU fp32 (oneDNN) network restatement + NumPy restatement of the reference '
                       'post-process (both pinned bit-exactly to the verbatim reference in the authoring container; the verbatim '
                       'code cannot travel to this box -- its own timing: BASELINE.md section 3); %d threads = %s (os.cpu_count %s, '
                       'affinity %s, cgroup quota %s); %.1f s'
                       % (frames, warm, threads, bound_by, lim['os_cpu_count'], lim['affinity'], lim['cgroup_quota'], dt)}, results)


def keypoint_match(eng, rec, results, weights_for_match, imgs_for_match, plan=None):
    """The second half of the metric: the GPU path's key points against the oracle's on the frames the CPU baseline
    processed (same images, same weights).  Target (BASELINE.json): integer peak indices identical, scores within 1e-4.
    The two networks differ by ~1e-6 (summation order), so a peak exactly at a tie / threshold could legitimately flip."""
    n = len(results)
    peaks_same = poses_same = 0
    d_peak = d_person = 0.0
    for i, o in enumerate(results):
        gp, op = eng.peaks(i), np.asarray(o['all_peaks'], dtype=np.float64).reshape(-1, 5)
        if gp.shape == op.shape and np.array_equal(gp[:, [0, 1, 2, 4]], op[:, [0, 1, 2, 4]]):
            peaks_same += 1
            if len(gp):
                d_peak = max(d_peak, float(np.abs(gp[:, 3] - op[:, 3]).max()))
        k = int(rec[i]['n_people'])
        g_poses, g_scores = rec[i]['poses'][:k], rec[i]['scores'][:k]
        o_poses = np.asarray(o['poses'], dtype=np.float64).reshape(-1, 18, 3)
        o_scores = np.asarray(o['scores'], dtype=np.float64).reshape(-1)
        if g_poses.shape == o_poses.shape and np.array_equal(g_poses, o_poses):
            poses_same += 1
            if k:
                d_person = max(d_person, float(np.abs(g_scores - o_scores).max()))
    # one full-size frame through the order-defined fp32 oracle (plain C, the kernels' arithmetic: the direct FMA chains, and the
    # Winograd twin for the layers the batch ran on the Winograd kernel): bit-exact maps
    exact = None
    try:
        if weights_for_match is None:
            raise StopIteration
        from oracle import conv_fma_ref, postprocess_ref
        t0 = time.perf_counter()
        paf, heat = eng.get_maps()
        epaf, eheat = conv_fma_ref.forward_fma(weights_for_match, postprocess_ref.preprocess(imgs_for_match[0]), splitk=plan)
        exact = {'frames': 1, 'layers_as_winograd': len(plan.wino), 'layers_with_unit_mode_tails': len(plan.wino_tails),
                 'paf_and_heat_maps_bit_identical': bool(np.array_equal(paf[0], epaf[0]) and np.array_equal(heat[0], eheat[0])),
                 'oracle_seconds': time.perf_counter() - t0}
    except StopIteration:
        exact = None
    except Exception as e:          # the checker must never break the measurement
        exact = {'error': repr(e)}
    return {'frames_compared': n, 'frames_with_identical_peak_indices': peaks_same, 'max_abs_peak_score_diff': d_peak,
            'network_vs_order_defined_oracle': exact,
            'frames_with_identical_poses': poses_same, 'max_abs_person_score_diff': d_person,
            'target': 'peak indices identical, scores within 1e-4 (oracle = torch-CPU fp32 network + NumPy restatement of '
                      'the reference post-process)'}


def dominant_kernel(prof):
    """(HIP kernel, total_ms, launches, algorithmic FLOP, issued FLOP, labels) of the kernel with the largest total time -- grouped by
    HIP kernel, as rocprofv3 groups them (the profile labels of one kernel differ by layer plan, e.g. ".../t2" and ".../t3")."""
    by_kernel = {}
    for p_ in prof:
        e = by_kernel.setdefault(rocprof_kernel(p_['kernel']), [0.0, 0, 0.0, 0.0, set()])
        e[0] += p_['total_ms']; e[1] += p_['launches']; e[2] += p_['flop_per_launch'] * p_['launches']
        e[3] += p_.get('issued_flop_per_launch', p_['flop_per_launch']) * p_['launches']; e[4].add(p_['kernel'])
    name = max(by_kernel, key=lambda k: by_kernel[k][0])
    return (name,) + tuple(by_kernel[name][:4]) + (sorted(by_kernel[name][4]),)


def main():
    a = parse()
    if a.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        self_launch(a)
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    import torch
    import torch.distributed as dist
    native = importlib.import_module(PKG + '.native')
    weights_mod = importlib.import_module(PKG + '.weights')
    dist_mod = importlib.import_module(PKG + '.dist')
    if native.needs_build():
        native.build()          # serialised across ranks by a lock file
    local_rank = pick_device(local_rank, torch.cuda.device_count(), world, a.backend, rank)
    torch.cuda.set_device(local_rank)
    use_group = world > 1 or a.force_gather
    if use_group:
        if world == 1:
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            if 'MASTER_PORT' not in os.environ:
                import socket
                s_ = socket.socket(); s_.bind(('127.0.0.1', 0)); os.environ['MASTER_PORT'] = str(s_.getsockname()[1]); s_.close()
        if a.backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group(a.backend, rank=rank, world_size=world)
        dist.barrier()
    dev = torch.device('cuda', local_rank)
    coll_dev = dev if a.backend == 'nccl' else torch.device('cpu')
    B, S = a.batch, a.size
    map_s = 320 if S == 368 else (S * 320) // 368 // 8 * 8

    eng = native.Engine(local_rank, max_batch=B, max_h=S, max_w=S)
    weights = weights_mod.synthetic_weights(0)
    eng.set_weights(weights)
    # calibrate the synthetic head on one fixed image (same on every rank) so the post-process has a COCO-like load
    cal = np.random.default_rng(1234).integers(0, 256, (1, S, S, 3), dtype=np.uint8)
    eng.forward_u8(cal)
    paf, heat = eng.get_maps()
    weights = weights_mod.calibrate_head(weights, paf[0], heat[0])
    eng.set_weights({k: weights[k] for k in ('Mconv7_stage6_L1', 'Mconv7_stage6_L2')})
    for kv in a.engine_opt:
        k_, v_ = kv.split('=')
        eng.set_option(k_, int(v_))

    # the global batch (B * world images, one seeded stream) is sharded contiguously: rank r owns images [r*B, (r+1)*B),
    # so a 2-rank run of --batch 4 sees exactly the images of a 1-rank run of --batch 8
    lo, hi = dist_mod.shard_range(B * world, rank, world)
    rng = np.random.default_rng(1)
    if lo:
        rng.integers(0, 256, (lo, S, S, 3), dtype=np.uint8)      # skip the images of the lower ranks (same stream position)
    imgs = rng.integers(0, 256, (hi - lo, S, S, 3), dtype=np.uint8)
    d_imgs = torch.from_numpy(imgs).to(dev)          # inputs resident in HBM before the timed region
    torch.cuda.synchronize()
    gather_ms = [0.0]
    wait_ms = [0.0]
    pipe = None
    if use_group:
        # N > 1 (and --force-gather): the records travel through dist.RecordPipe -- ONE fixed-size gather per step, issued one step behind
        # the compute (detect_batch(k + 1) is enqueued before the records of step k are shipped), so ranks never wait for each other
        # inside the loop; the slot size is agreed once, here, from the record size after one plain step
        eng.detect_batch(device_ptr=d_imgs.data_ptr(), shape=(B, S, S), map_h=map_s, map_w=map_s)
        _, rec_bytes0 = eng.results_layout()
        pipe = dist_mod.RecordPipe(B * rec_bytes0, dst=0, device=coll_dev, nslots=PIPE_DEPTH + 2)
    snap_mode = [None] * (PIPE_DEPTH + 2)

    def enqueue(k, ptr=None):
        """detect_batch of step k + a stream-ordered snapshot of its records (into the pipe's send slot when it fits); no host sync"""
        eng.detect_batch(device_ptr=d_imgs.data_ptr() if ptr is None else ptr, shape=(B, S, S), map_h=map_s, map_w=map_s)
        slot_ptr, room = pipe.payload_view(k)
        need = B * native.result_dtype(eng.capacities()['people']).itemsize
        sl = k % (PIPE_DEPTH + 2)
        if slot_ptr is not None and need <= room:
            eng.results_snapshot(sl, slot_ptr, room)
            snap_mode[sl] = None
        else:                       # gloo smoke mode (the pipe lives on the host) or records larger than the slot: through a staging tensor
            stage = torch.empty(need, dtype=torch.uint8, device=dev)
            eng.results_snapshot(sl, stage.data_ptr(), need)
            snap_mode[sl] = stage

    def ship(j):
        """records of step j -> the pipe (one gather); on the root: the steps completed by it [(step, records of all ranks)]"""
        t_w = time.perf_counter()
        sl = j % (PIPE_DEPTH + 2)
        n, cap, rec_bytes, overflow = eng.snapshot_wait(sl)              # (the host runs ahead of the GPU: this waits for it to finish step j)
        t1 = time.perf_counter()
        wait_ms[0] += (t1 - t_w) * 1e3
        if overflow:
            # an image needed more capacity than the context had: the snapshot is not final.  Rare (capacities grow once): drain, run
            # the step again through the growing path, send the records from the host (the in-flight next step is re-checked at its turn)
            eng.synchronize()
            eng.detect_batch(device_ptr=d_imgs.data_ptr(), shape=(B, S, S), map_h=map_s, map_w=map_s)
            r_ = eng.results()
            done = pipe.send(j, j, len(r_), dist_mod._people_cap(r_.dtype), r_.dtype.itemsize, payload=np.frombuffer(r_.tobytes(), dtype=np.uint8))
        elif snap_mode[sl] is None:
            done = pipe.send(j, j, n, cap, rec_bytes)
        else:
            done = pipe.send(j, j, n, cap, rec_bytes, payload=snap_mode[sl][:n * rec_bytes].cpu().numpy())
        gather_ms[0] += (time.perf_counter() - t1) * 1e3
        return done or []

    def run_steps(n):
        """n steps; returns the records of the last one (on rank 0: of all ranks)"""
        if pipe is None:
            for _ in range(n):
                eng.detect_batch(device_ptr=d_imgs.data_ptr(), shape=(B, S, S), map_h=map_s, map_w=map_s)
                rec_ = eng.results()                  # stream sync + D2H of the records
            return rec_
        got = []
        for k in range(n):
            enqueue(k)
            if k >= PIPE_DEPTH:
                got += ship(k - PIPE_DEPTH)
        for j in range(max(0, n - PIPE_DEPTH), n):
            got += ship(j)
        got += pipe.flush() or []
        if rank == 0:
            assert [st for st, _ in got] == list(range(n)), ('steps delivered', [st for st, _ in got])
            return got[-1][1]
        return None

    def step():
        return run_steps(1)

    if a.warmup:
        rec = run_steps(a.warmup)
    profile = not a.no_profile
    if profile:
        # inside the timed region only the dominant kernel's launches carry HIP-event pairs (25 of ~60 launches per step: the
        # events cost ~5 us of idle each); the per-kernel split of a whole step is measured on one extra, untimed step below
        eng.profile_reset()
        eng.profile_enable(2)
    gather_ms[0] = wait_ms[0] = 0.0
    n_coll0 = pipe.collectives if pipe else 0
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rec = run_steps(a.steps)
    eng.synchronize()
    torch.cuda.synchronize()
    dt_local = time.perf_counter() - t0
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    n_coll = (pipe.collectives - n_coll0) if pipe else 0
    per_rank = [B * a.steps / dt_local]
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        pr = [torch.zeros(1, dtype=torch.float64, device=coll_dev) for _ in range(world)]
        dist.all_gather(pr, torch.tensor([per_rank[0]], dtype=torch.float64, device=coll_dev))
        per_rank = [float(v.item()) for v in pr]

    # evidence that N ranks on N distinct GPUs took part: every rank reports (rank, local_rank, device name / uuid / PCI bus id)
    me = device_identity(torch, rank, local_rank)
    rank_info = [me]
    if world > 1:
        rank_info = [None] * world
        dist.all_gather_object(rank_info, me)
        if a.backend == 'nccl':
            # (one node: the PCI address identifies the GPU.  Reported, not asserted: RCCL itself refuses two ranks on one device, and a box
            #  that reports identical addresses for distinct GPUs must not cost the run its result line)
            ids = [r_['pci_bus_id'] or r_['uuid'] for r_ in rank_info]
            if None not in ids and len(set(ids)) != world and rank == 0:
                sys.stderr.write('bench.py: WARNING: %d ranks report %d distinct GPU addresses: %s\n' % (world, len(set(ids)), ids))
    prof = eng.profile() if profile else []
    prof_all = []
    if profile:
        eng.profile_reset()
        eng.profile_enable(1)
        step()                       # untimed: every launch instrumented, for the per-kernel split / --dump-profile
        prof_all = eng.profile()
        eng.profile_enable(False)
    if rank == 0:
        frames = B * world * a.steps
        ms_per_step = dt / a.steps * 1e3
        status_bits = int(np.bitwise_or.reduce(rec['status'])) if len(rec) else 0
        if a.dump_records:
            np.save(a.dump_records, rec)
        out = {
            'metric': 'frames/sec at 368x368 batch (1/2/4/8 GPU) + keypoint-match vs reference',
            'value': frames / dt, 'unit': 'frames/s', 'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup,
            'ms_per_step': ms_per_step, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'batch%d_%dx%d_synthetic_uint8_per_gpu' % (B, S, S), 'per_gpu_batch': B,
                       'global_batch': B * world, 'map': '%dx%d' % (map_s, map_s), 'weights': 'seeded He + calibrated head',
                       'parallelism': ('dp%d (independent images, one %s gather of the result records to rank 0 per step, pipelined one step '
                                       'behind the compute)' % (world, 'RCCL' if a.backend == 'nccl' else a.backend)) if use_group
                                      else 'dp1 (one process, one GPU: no process group, no collective; records by pmx_get_results)',
                       'records_gathered': int(len(rec)),
                       'people_per_frame_mean': float(np.mean(rec['n_people'])) if len(rec) else 0.0,
                       'peaks_per_frame_mean': float(np.mean(rec['n_peaks'])) if len(rec) else 0.0,
                       'status_bits': status_bits, 'engine_options': a.engine_opt or None},
            'achieved_tflops_whole_net': FLOP_PER_FRAME * (S * S / (368.0 * 368.0)) * frames / dt / 1e12,
        }
        if use_group:
            out['per_rank_frames_per_s'] = per_rank
            out['per_rank_frames_per_s_min_max'] = [min(per_rank), max(per_rank)]
            out['gather_ms_per_step_rank0'] = gather_ms[0] / a.steps        # header write + gather + D2H + parse on rank 0 (host time; the GPU runs the next step meanwhile)
            out['host_wait_for_gpu_ms_per_step_rank0'] = wait_ms[0] / a.steps  # the host is one step ahead: time it spent waiting for the step's snapshot event
            out['collectives_per_step'] = n_coll / float(a.steps)             # (1.0 + one flush all_reduce per timed region)
            out['pipeline'] = 'detect_batch(k + %d) enqueued before the records of step k are gathered (dist.RecordPipe, slot %d bytes)' % (PIPE_DEPTH, pipe.slot_bytes)
        out['backend'] = ('rccl (torch.distributed "nccl")' if a.backend == 'nccl' else a.backend) if use_group else None
        out['records_path'] = ('dist.RecordPipe (pmx_results_snapshot into the send slot on the device -> one RCCL gather per step -> one D2H copy on rank 0)'
                               if use_group and a.backend == 'nccl' else 'dist.RecordPipe over %s (host slots)' % a.backend if use_group
                               else 'pmx_get_results (one D2H copy)')
        if use_group:
            # the SCALE record validates itself: the records that came through the gather against rank 0's own re-run of every shard
            try:
                chk = validate_shards(eng, torch, dev, dist_mod, rec, B, S, map_s, world, a.validate_images)
            except Exception as e:          # (a failing check must not cost the run its line: it reports itself as a failed check)
                chk = {'shard_records_match': False, 'records_compared': 0, 'max_abs_score_diff': None, 'error': repr(e)}
            out['shard_records_match'] = chk['shard_records_match']
            out['records_compared'] = chk['records_compared']
            out['max_abs_score_diff'] = chk['max_abs_score_diff']
            out['shard_validation'] = chk
            if not chk['shard_records_match']:
                sys.stderr.write('bench.py: ERROR: gathered records differ from rank 0\'s re-run of the shards: %s\n' % json.dumps(chk))
        out['ranks_seen'] = [r_['rank'] for r_ in rank_info]
        out['devices'] = rank_info
        out['distinct_gpus'] = len(set((r_['pci_bus_id'] or r_['uuid'] or r_['rank']) for r_ in rank_info))
        if use_group and a.backend == 'nccl' and out['distinct_gpus'] != world:
            # not asserted (a box may report one address for distinct GPUs) but the record says so itself: not a scaling figure
            out['scaling_record_valid'] = False
            out['scaling_record_invalid_reason'] = '%d ranks on %d distinct GPU addresses' % (world, out['distinct_gpus'])
        elif use_group:
            out['scaling_record_valid'] = bool(a.backend == 'nccl' and out.get('shard_records_match', False))
            if a.backend != 'nccl':
                out['scaling_record_invalid_reason'] = 'backend %s: ranks share GPUs (code-path check, not a scaling figure)' % a.backend
        roof = None
        if prof:
            # dominant kernel = the HIP kernel (as rocprofv3 groups them) with the largest total time in the timed region: the 7x7
            # Winograd kernel, 25 launches per step (5 x Mconv1 with 185 input channels + 20 x Mconv2-5; at batch 32 the launch of the
            # full 32-tile runs -- the part-filled last block of every image is a separate unit-mode launch with its own entry)
            dom_name, total_ms, launches, total_flop, total_issued, dom_labels = dominant_kernel(prof)
            if launches:
                avg_ms = total_ms / launches
                ach_alg = total_flop / (total_ms * 1e-3) / 1e12
                ach = total_issued / (total_ms * 1e-3) / 1e12
                traffic, traffic_src = pmc_traffic(dom_name)
                sub = [p_ for p_ in prof if rocprof_kernel(p_['kernel']) == dom_name and p_['layer'].startswith(DOMINANT_LAYERS)]
                sub_ach = (sum(p_['issued_flop_per_launch'] * p_['launches'] for p_ in sub) / (sum(p_['total_ms'] for p_ in sub) * 1e-3) / 1e12
                           if sub else None)
                roof = {'kernel': dom_name, 'profile_labels': dom_labels, 'bound': 'mfma', 'achieved': ach, 'peak': FP32_MFMA_PEAK_TFLOPS,
                        'unit': 'TFLOP/s', 'frac': ach / FP32_MFMA_PEAK_TFLOPS, 'issued_frac': ach / FP32_MFMA_PEAK_TFLOPS,
                        'frac_definition': 'issued MFMA FLOP (= algorithmic x 100/196 for the 7x7 Winograd form) / launch duration / peak -- '
                                           'NOT the algorithmic figure of the bench contract, which is `algorithmic_frac` (> 1 for a Winograd '
                                           'kernel); BASELINE.md section 4 states the redefinition',
                        'traffic': traffic, 'traffic_source': traffic_src,
                        'issued_flop_per_launch': total_issued / launches, 'flop_per_launch': total_flop / launches,
                        'executed_flop_fraction': total_issued / total_flop if total_flop else None,
                        'algorithmic_achieved': ach_alg, 'algorithmic_frac': ach_alg / FP32_MFMA_PEAK_TFLOPS,
                        'avg_launch_ms': avg_ms, 'launches_timed': launches, 'achieved_128ch_layers_only': sub_ach,
                        'note': 'all launches of the dominant kernel in the timed region (B=%d, both branch groups per launch), HIP events on '
                                'the launch stream.  `achieved` / `frac` = FLOP the kernel ISSUES to the matrix cores for real outputs '
                                '(per-launch figures from the engine profile: pmx_profile_issued) / mean launch duration / dense fp32-MFMA '
                                'peak: <= 1 by construction.  `algorithmic_*` = the same with the FLOP of the convolution it computes '
                                '(2 * k * k * cin * cout per output pixel); it exceeds the issued figure where the kernel is Winograd '
                                'F(2x2,3x3) (7x7: 100 of 196 products per output tile and channel pair, 3x3: 16 of 36)' % B}
            conv_ms = sum(p['total_ms'] for p in prof_all if p['kernel'].startswith('conv'))
            pp_ms = sum(p['total_ms'] for p in prof_all if p['kernel'].startswith('pp_'))
            out['kernel_time_ms_per_step'] = {'conv': conv_ms, 'postprocess': pp_ms, 'note': 'one extra untimed step with every launch instrumented'}
            # the same pair for the whole step: issued / algorithmic FLOP of every launch of one step (untimed, fully instrumented step
            # gives the launch plan) over the TIMED step duration
            step_issued = sum(p['issued_flop_per_launch'] * p['launches'] for p in prof_all)
            step_alg = sum(p['flop_per_launch'] * p['launches'] for p in prof_all)
            by_form = {}
            for p in prof_all:
                if p['flop_per_launch'] > 0:
                    e = by_form.setdefault(rocprof_kernel(p['kernel']), [0, 0.0])
                    e[0] += p['launches']; e[1] += p['total_ms']
            out['step_roofline'] = {'bound': 'mfma', 'achieved': step_issued / (ms_per_step * 1e-3) / 1e12, 'peak': FP32_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                                    'frac': step_issued / (ms_per_step * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS,
                                    'algorithmic_achieved': step_alg / (ms_per_step * 1e-3) / 1e12,
                                    'algorithmic_frac': step_alg / (ms_per_step * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS,
                                    'issued_flop_per_step': step_issued, 'flop_per_step': step_alg,
                                    'launch_plan': {k: {'launches': v[0], 'ms': v[1]} for k, v in sorted(by_form.items())},
                                    'note': 'whole step (preprocess + 92 layers + post-process + record copy): issued / algorithmic conv FLOP of '
                                            'one step / timed ms_per_step / peak'}
            if a.dump_profile:
                with open(a.dump_profile, 'w') as f:
                    json.dump({'batch': B, 'steps': 1, 'entries': prof_all}, f, indent=1)
        out['roofline'] = roof
        eng.profile_enable(False)
        if world == 1 and not a.no_extras:
            # the side legs (other workloads of the same path, after the timed region): a leg that fails reports its error under its key and
            # must never cost the run its line -- the headline above is already measured
            def leg(key, fn, *args, **kw):
                try:
                    out[key] = fn(*args, **kw)
                except Exception as e:
                    out[key] = {'error': repr(e)}
                    sys.stderr.write('bench.py: leg %s failed: %r\n' % (key, e))
                    try:
                        eng.set_option('conv_algo', 1); eng.set_option('precision', 0)      # (what direct_only / bf16x3_mode switch)
                    except Exception:
                        pass
            leg('value_incl_h2d', upload_inclusive, eng, torch, dev, imgs, a.steps, map_s)
            leg('single_image', single_image, eng, d_imgs, S, map_s)
            leg('direct_kernels_only', direct_only, eng, torch, d_imgs, B, S, map_s, a.steps)
            if a.bf16x3:
                leg('bf16x3', bf16x3_mode, eng, torch, dev, d_imgs, B, S, map_s, a.steps)
            leg('rect_368x496', rect_inputs, native, weights_mod, torch, dev, local_rank, B, a.steps, frames / dt, S)
            leg('mixed_sizes', mixed_sizes_mode, weights_mod, local_rank, B, max(2, a.steps // 4))
            leg('precise', precise_mode, weights_mod, local_rank, with_oracle=not a.no_cpu_baseline)
            if isinstance(out.get('precise'), dict):
                out['precise']['hw_queues'] = 'GPU_MAX_HW_QUEUES=%s' % os.environ.get('GPU_MAX_HW_QUEUES')
            eng.detect_batch(device_ptr=d_imgs.data_ptr(), shape=(B, S, S), map_h=map_s, map_w=map_s)     # restore the batch state
            rec = eng.results()                                                                             # for keypoint_match
        if world == 1 and not a.no_cpu_baseline:
            oracle_results = None
            try:
                out['cpu_baseline'], oracle_results = cpu_baseline(weights, imgs, (map_s, map_s), a.cpu_budget)
                from oracle import conv_fma_ref
                out['keypoint_match'] = keypoint_match(eng, rec, oracle_results, weights, imgs, conv_fma_ref.splitk_plan(prof_all, image=0))
                out['keypoint_match']['census'] = committed_census()
            except Exception as e:          # (the checker and the CPU leg must never cost the run its line)
                out.setdefault('cpu_baseline', {'error': repr(e)})
                out.setdefault('keypoint_match', {'error': repr(e)})
                sys.stderr.write('bench.py: cpu_baseline / keypoint_match failed: %r\n' % (e,))
            if not a.no_extras and a.bf16x3 and oracle_results is not None and 'error' not in out.get('bf16x3', {'error': 1}):
                # the opt-in bf16x3 mode against the same CPU oracle frames (it is compared with the fp32 path above; this is the
                # figure the north_star tolerance applies to)
                eng.set_option('precision', 1)
                try:
                    eng.detect_batch(device_ptr=d_imgs.data_ptr(), shape=(B, S, S), map_h=map_s, map_w=map_s)
                    out['bf16x3']['keypoint_match_vs_cpu_oracle'] = keypoint_match(eng, eng.results(), oracle_results, None, None)
                finally:
                    eng.set_option('precision', 0)
        else:
            out['cpu_baseline'] = None
        print(json.dumps(out))
    eng.close()
    if use_group:
        dist.barrier()
        dist.destroy_process_group()


def compare_records(gathered, recomputed, score_tol=1e-5):
    """Two result-record arrays of the same images (native.result_dtype; the person capacities may differ): people / peak counts and poses
    must be identical, scores within `score_tol`.  Returns what the SCALE line reports."""
    n = min(len(gathered), len(recomputed))
    out = {'records_compared': int(n), 'records_with_identical_counts_and_poses': 0, 'max_abs_score_diff': 0.0, 'bitwise_equal_records': 0,
           'first_mismatch': None}
    for i in range(n):
        a, b = gathered[i], recomputed[i]
        na, nb = int(a['n_people']), int(b['n_people'])
        same = na == nb and int(a['n_peaks']) == int(b['n_peaks']) and int(a['status']) == int(b['status']) and \
            np.array_equal(a['poses'][:na], b['poses'][:nb])
        if same:
            out['records_with_identical_counts_and_poses'] += 1
            d = float(np.max(np.abs(a['scores'][:na] - b['scores'][:nb]))) if na else 0.0
            out['max_abs_score_diff'] = max(out['max_abs_score_diff'], d)
            if d == 0.0:
                out['bitwise_equal_records'] += 1
        elif out['first_mismatch'] is None:
            out['first_mismatch'] = {'record': i, 'n_people': [na, nb], 'n_peaks': [int(a['n_peaks']), int(b['n_peaks'])],
                                     'status': [int(a['status']), int(b['status'])]}
    out['shard_records_match'] = bool(n > 0 and out['records_with_identical_counts_and_poses'] == n and out['max_abs_score_diff'] <= score_tol)
    return out


def validate_shards(eng, torch, dev, dist_mod, rec, B, S, map_s, world, limit=0):
    """Rank 0, after the timed region of an N > 1 run (or --force-gather): are the GATHERED records right?  The global batch is one seeded
    stream, so rank 0 regenerates the images of every rank's shard (all B of them, or the first `limit`), runs them itself -- same
    batch size, hence the same kernel forms as the owning rank used -- fetches the records the plain way (pmx_get_results) and compares
    them with what came through the record pipe: counts and poses exact, scores <= 1e-5 (a shard re-run at ANOTHER batch size may differ
    in the last bits: the path is not batch-invariant; at the same size the kernels are deterministic, so
    `bitwise_equal_records` is expected to equal `records_compared`).  A wrong rank order, a stale slot, a truncated frame or a rank that
    computed on the wrong images all show up here; the rate alone would not notice."""
    total = {'records_compared': 0, 'records_with_identical_counts_and_poses': 0, 'max_abs_score_diff': 0.0, 'bitwise_equal_records': 0,
             'first_mismatch': None, 'ranks_checked': [], 'images_per_rank': None}
    rng = np.random.default_rng(1)
    for r in range(world):
        lo, hi = dist_mod.shard_range(B * world, r, world)
        imgs = rng.integers(0, 256, (hi - lo, S, S, 3), dtype=np.uint8)          # (the stream position of rank r's shard)
        k = hi - lo if not limit else min(limit, hi - lo)
        total['images_per_rank'] = k
        d = torch.from_numpy(imgs).to(dev)
        # the whole shard runs (the kernel forms depend on the batch size); the first k records are compared
        eng.detect_batch(device_ptr=d.data_ptr(), shape=(hi - lo, S, S), map_h=map_s, map_w=map_s)
        mine = eng.results()
        c = compare_records(rec[lo:lo + k], mine[:k])
        total['ranks_checked'].append(r)
        for key in ('records_compared', 'records_with_identical_counts_and_poses', 'bitwise_equal_records'):
            total[key] += c[key]
        total['max_abs_score_diff'] = max(total['max_abs_score_diff'], c['max_abs_score_diff'])
        if c['first_mismatch'] is not None and total['first_mismatch'] is None:
            total['first_mismatch'] = dict(c['first_mismatch'], rank=r)
        del d
    total['shard_records_match'] = bool(total['records_compared'] > 0 and total['max_abs_score_diff'] <= 1e-5 and
                                        total['records_with_identical_counts_and_poses'] == total['records_compared'])
    total['how'] = ('rank 0 regenerated every rank\'s shard of the seeded global batch, ran it at the same batch size and compared pmx_get_results '
                    'with the records gathered through dist.RecordPipe in the last timed step: counts / poses exact, scores <= 1e-5')
    return total


def device_identity(torch, rank, local_rank):
    pr = torch.cuda.get_device_properties(local_rank)
    uuid = getattr(pr, 'uuid', None)
    bus = None
    try:
        bus = '%04x:%02x:%02x' % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
    except Exception:
        try:        # older torch builds: ask the HIP runtime
            import ctypes
            buf = ctypes.create_string_buffer(64)
            if ctypes.CDLL('libamdhip64.so').hipDeviceGetPCIBusId(buf, 64, int(local_rank)) == 0:
                bus = buf.value.decode()
        except Exception:
            pass
    return {'rank': rank, 'local_rank': local_rank, 'name': pr.name, 'uuid': str(uuid) if uuid is not None else None, 'pci_bus_id': bus,
            'compute_units': getattr(pr, 'multi_processor_count', None)}


def upload_inclusive(eng, torch, dev, imgs, steps, map_s):
    """The same K steps with the host->device upload of every batch INSIDE the timed region: pinned staging buffer, the
    upload of batch k+1 on a second stream under the compute of batch k (double-buffered device input).  The first upload
    is not hidden and is counted."""
    B, S = imgs.shape[0], imgs.shape[1]
    pinned = torch.from_numpy(imgs).pin_memory()
    bufs = [torch.empty_like(pinned, device=dev) for _ in range(2)]
    copy_stream = torch.cuda.Stream(device=dev)
    ready = [torch.cuda.Event(), torch.cuda.Event()]

    def upload(k):
        with torch.cuda.stream(copy_stream):
            bufs[k & 1].copy_(pinned, non_blocking=True)
            ready[k & 1].record(copy_stream)

    def run(n):
        upload(0)
        for k in range(n):
            ready[k & 1].synchronize()                # batch k is in HBM
            if k + 1 < n:
                upload(k + 1)                         # overlaps the network of batch k (engine stream)
            eng.detect_batch(device_ptr=bufs[k & 1].data_ptr(), shape=(B, S, S), map_h=map_s, map_w=map_s)
            eng.results()
    run(2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {'value': B * steps / dt, 'unit': 'frames/s', 'ms_per_step': dt / steps * 1e3, 'steps': steps,
            'upload_bytes_per_step': int(imgs.nbytes),
            'how': 'pinned host batch -> HBM on a copy stream, upload of batch k+1 overlapped with the compute of batch k; '
                   'first upload exposed and counted'}


def direct_only(eng, torch, d_imgs, B, S, map_s, steps):
    """The same K steps with option "conv_algo" = 0: every convolution on the direct fp32-MFMA kernels (no Winograd) -- the number
    to hold against the fp32-MFMA roofline of the direct convolution (157.3 TFLOP/s <-> 578 frames/s), and the single-image call."""
    eng.set_option('conv_algo', 0)
    try:
        for _ in range(2):
            eng.detect_batch(device_ptr=d_imgs.data_ptr(), shape=(B, S, S), map_h=map_s, map_w=map_s); eng.results()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            eng.detect_batch(device_ptr=d_imgs.data_ptr(), shape=(B, S, S), map_h=map_s, map_w=map_s); eng.results()
        eng.synchronize()
        dt = time.perf_counter() - t0
        for _ in range(5):
            eng.detect_batch(device_ptr=d_imgs.data_ptr(), shape=(1, S, S), map_h=map_s, map_w=map_s); eng.results()
        t1 = time.perf_counter()
        for _ in range(30):
            eng.detect_batch(device_ptr=d_imgs.data_ptr(), shape=(1, S, S), map_h=map_s, map_w=map_s); eng.results()
        dt1 = (time.perf_counter() - t1) / 30
    finally:
        eng.set_option('conv_algo', 1)
    fps = B * steps / dt
    return {'value': fps, 'unit': 'frames/s', 'ms_per_step': dt / steps * 1e3, 'steps': steps, 'single_image_ms_per_call': dt1 * 1e3,
            'achieved_tflops_whole_net': FLOP_PER_FRAME * (S * S / (368.0 * 368.0)) * fps / 1e12,
            'frac_of_fp32_mfma_peak_whole_net': FLOP_PER_FRAME * (S * S / (368.0 * 368.0)) * fps / 1e12 / FP32_MFMA_PEAK_TFLOPS,
            'note': 'option conv_algo = 0: direct fp32-MFMA convolution everywhere (the headline uses the fp32 Winograd F(2x2,3x3) kernel for the '
                    '3x3 / 7x7 layers: same dtype, same results to fp32 rounding, fewer multiplies)'}


def bf16x3_mode(eng, torch, dev, d_imgs, B, S, map_s, steps, frames=256):
    """NOT the headline: the opt-in "precision" = 1 mode (3x3 / 7x7 layers of large batches on the bf16 matrix cores, every fp32
    value split into three bf16 terms, six products, fp32 accumulate) -- same step, same K, plus its agreement with the fp32 path on
    `frames` fresh frames: peak indices, poses, score deltas."""
    def run(ptr):
        eng.detect_batch(device_ptr=ptr, shape=(B, S, S), map_h=map_s, map_w=map_s)
        return eng.results()
    eng.set_option('precision', 1)
    for _ in range(2):
        run(d_imgs.data_ptr())
    t0 = time.perf_counter()
    for _ in range(steps):
        run(d_imgs.data_ptr())
    dt = time.perf_counter() - t0
    st = dict(frames=0, frames_with_identical_peak_indices=0, frames_with_identical_poses=0, max_abs_peak_score_diff=0.0,
              max_abs_person_score_diff=0.0, max_rel_map_diff=0.0)
    for it in range((frames + B - 1) // B):
        batch = torch.from_numpy(np.random.default_rng(500 + it).integers(0, 256, (B, S, S, 3), dtype=np.uint8)).to(dev)
        res = {}
        for prec in (0, 1):
            eng.set_option('precision', prec)
            rec = run(batch.data_ptr()).copy()
            res[prec] = (rec, [eng.peaks(i) for i in range(B)], eng.get_maps())
        for i in range(B):
            p0, p1 = res[0][1][i], res[1][1][i]
            st['frames'] += 1
            if p0.shape == p1.shape and np.array_equal(p0[:, [0, 1, 2, 4]], p1[:, [0, 1, 2, 4]]):
                st['frames_with_identical_peak_indices'] += 1
                if len(p0):
                    st['max_abs_peak_score_diff'] = max(st['max_abs_peak_score_diff'], float(np.abs(p0[:, 3] - p1[:, 3]).max()))
            r0, r1 = res[0][0][i], res[1][0][i]
            if r0['n_people'] == r1['n_people'] and np.array_equal(r0['poses'], r1['poses']):
                st['frames_with_identical_poses'] += 1
                st['max_abs_person_score_diff'] = max(st['max_abs_person_score_diff'], float(np.abs(r0['scores'] - r1['scores']).max()))
        for m0, m1 in zip(res[0][2], res[1][2]):
            st['max_rel_map_diff'] = max(st['max_rel_map_diff'], float(np.abs(m0 - m1).max() / max(1e-30, np.abs(m0).max())))
    # single image in this mode (v8 small-tile kernels + K slices)
    eng.set_option('precision', 1)
    for _ in range(3):
        eng.detect_batch(device_ptr=d_imgs.data_ptr(), shape=(1, S, S), map_h=map_s, map_w=map_s)
        eng.results()
    t1 = time.perf_counter()
    for _ in range(30):
        eng.detect_batch(device_ptr=d_imgs.data_ptr(), shape=(1, S, S), map_h=map_s, map_w=map_s)
        eng.results()
    one_ms = (time.perf_counter() - t1) / 30 * 1e3
    eng.set_option('precision', 0)
    flop = FLOP_PER_FRAME * (S * S / (368.0 * 368.0))
    return {'value': B * steps / dt, 'single_image_ms_per_call': one_ms, 'unit': 'frames/s', 'ms_per_step': dt / steps * 1e3, 'steps': steps, 'dtype': 'bf16x3/f32acc',
            'fp32_equivalent_tflops': flop * B * steps / dt / 1e12,
            'note': 'opt-in mode, never the headline: fp32 values as hi + mid + lo bf16, products hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid on '
                    'v_mfma_f32_32x32x16_bf16 with fp32 accumulation; conv1_1 and the 1x1 pairs stay on the fp32 MFMA',
            'agreement_with_f32_path': st}


def committed_census():
    """The margin census of the peak decisions (tools/parity_census.py, run on the GPU box, committed under profiles/): the statement
    behind "peak indices identical" -- N of M fresh frames identical, every difference a near-tie below X.  Quoted, not re-run."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_parity_census.json')))
    if not files:
        return None
    try:
        d = json.load(open(files[-1]))
        keys = ('frames', 'frames_identical', 'frames_with_identical_peak_indices', 'frames_with_identical_poses', 'peaks_compared',
                'mismatching_peaks', 'max_margin_of_a_mismatch', 'min_margin_of_accepted_peaks', 'min_abs_margin_of_agreed_decisions',
                'all_mismatches_are_near_ties', 'max_abs_peak_score_diff', 'max_abs_score_diff_matched_people')
        return {'source': os.path.basename(files[-1]), 'workload': d.get('workload'),
                'paths': {name: {k: p_.get(k) for k in keys} for name, p_ in d['paths'].items()}}
    except Exception as e:
        return {'error': repr(e)}


def rect_inputs(native, weights_mod, torch, dev, device_index, B, steps, square_fps, S, sizes=(('368x496', (368, 496)), ('496x368', (496, 368)))):
    """Landscape / portrait network inputs (compute_optimal_size, reference pose_detector.py:57-73, gives 368 x 496 for a 4:3 COCO
    frame): the same step at batch B on 46 x 62 / 62 x 46 feature maps -- frames/s, the rate per pixel relative to the square headline
    (1.0 = the same cost per pixel) and the dominant kernel's issued fraction of the fp32-MFMA peak."""
    out = {}
    for name, (h, w) in sizes:
        eng = native.Engine(device_index, max_batch=B, max_h=h, max_w=w)
        try:
            wts = weights_mod.synthetic_weights(0)
            eng.set_weights(wts)
            cal = np.random.default_rng(1234).integers(0, 256, (1, h, w, 3), dtype=np.uint8)
            eng.forward_u8(cal)
            paf, heat = eng.get_maps()
            wts = weights_mod.calibrate_head(wts, paf[0], heat[0])
            eng.set_weights({k: wts[k] for k in ('Mconv7_stage6_L1', 'Mconv7_stage6_L2')})
            imgs = torch.from_numpy(np.random.default_rng(2).integers(0, 256, (B, h, w, 3), dtype=np.uint8)).to(dev)
            mh, mw = h * 320 // 368 // 8 * 8, w * 320 // 368 // 8 * 8
            for _ in range(2):
                eng.detect_batch(device_ptr=imgs.data_ptr(), shape=(B, h, w), map_h=mh, map_w=mw); eng.results()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                eng.detect_batch(device_ptr=imgs.data_ptr(), shape=(B, h, w), map_h=mh, map_w=mw); rec = eng.results()
            eng.synchronize()
            dt = (time.perf_counter() - t0) / steps
            eng.profile_reset(); eng.profile_enable(1)
            eng.detect_batch(device_ptr=imgs.data_ptr(), shape=(B, h, w), map_h=mh, map_w=mw); eng.results()
            prof = eng.profile()
            eng.profile_enable(False)
            fps = B / dt
            o = {'value': fps, 'unit': 'frames/s', 'ms_per_step': dt * 1e3, 'steps': steps, 'batch': B, 'feature_map': '%dx%d' % (h // 8, w // 8),
                 'rate_per_pixel_vs_square': fps * h * w / (square_fps * S * S), 'people_per_frame_mean': float(np.mean(rec['n_people'])),
                 'achieved_tflops_whole_net': FLOP_PER_FRAME * (h * w / (368.0 * 368.0)) * fps / 1e12}
            if prof:
                nm, total_ms, launches, total_flop, total_issued, labels = dominant_kernel(prof)
                if launches and total_ms > 0:
                    o['dominant_kernel'] = {'kernel': nm, 'profile_labels': labels, 'avg_launch_ms': total_ms / launches, 'launches': launches,
                                            'issued_frac': total_issued / (total_ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS,
                                            'algorithmic_frac': total_flop / (total_ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS}
                step_issued = sum(p_['issued_flop_per_launch'] * p_['launches'] for p_ in prof)
                o['step_issued_frac'] = step_issued / dt / 1e12 / FP32_MFMA_PEAK_TFLOPS
            out[name] = o
        finally:
            eng.close()
    return out


def mixed_sizes_mode(weights_mod, device_index, batch=32, steps=5):
    """A stream of frames of DIFFERENT sizes (COCO-val style: 640 x 480, 480 x 640, 640 x 427, 500 x 375, 640 x 640 ...) through
    PoseDetector.detect_batch: the mixed batch (one launch per layer over all size classes: pmx_detect_images) against the reference's way
    (one image per call, pose_detector.py:484-517) and against a uniform 368 x 496 batch.  Host images in every call: uploads, device
    cv2.resize, network, post-process, records."""
    PD = importlib.import_module(PKG + '.pose_detector')
    native = importlib.import_module(PKG + '.native')
    weights = weights_mod.synthetic_weights(0)
    eng = native.Engine(device_index, max_batch=1, max_h=368, max_w=368)
    try:
        eng.set_weights(weights)
        eng.forward_u8(np.random.default_rng(1234).integers(0, 256, (1, 368, 368, 3), dtype=np.uint8))
        paf, heat = eng.get_maps()
    finally:
        eng.close()
    weights = weights_mod.calibrate_head(weights, paf[0], heat[0])
    rng = np.random.default_rng(7)
    classes = [(480, 640), (640, 480), (427, 640), (375, 500), (640, 640), (426, 640), (480, 640), (333, 500), (500, 375), (640, 427)]
    sizes = [classes[int(rng.integers(0, len(classes)))] for _ in range(batch)]
    imgs = [rng.integers(0, 256, s_ + (3,), dtype=np.uint8) for s_ in sizes]
    det = PD.PoseDetector(weights=weights, device=device_index, max_batch=batch, max_size=(368, 496))
    try:
        net = [det.compute_optimal_size(im, 368)[::-1] for im in imgs]
        npx = sum(h * w for h, w in net)

        def timed(fn, n):
            fn(); det.engine.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                r = fn()
            det.engine.synchronize()
            return (time.perf_counter() - t0) / n, r
        t_mixed, res = timed(lambda: det.detect_batch(imgs), steps)
        t_loop, res1 = timed(lambda: [det(im) for im in imgs], max(1, steps // 2))
        same = sum(1 for x, y in zip(res, res1) if np.asarray(x[0]).shape == np.asarray(y[0]).shape and np.array_equal(np.asarray(x[0]), np.asarray(y[0])))
        uni = [rng.integers(0, 256, (368, 496, 3), dtype=np.uint8) for _ in range(batch)]
        t_uni, _ = timed(lambda: det.detect_batch(uni), steps)
    finally:
        det.engine.close()
    return {'batch': batch, 'original_sizes': sorted(set('%dx%d' % s_ for s_ in sizes)), 'distinct_network_sizes': len(set(net)),
            'network_pixels_in_368x368_frames': npx / (368.0 * 368.0),
            'mixed_batch_ms': t_mixed * 1e3, 'one_image_per_call_ms': t_loop * 1e3, 'speedup_vs_one_image_per_call': t_loop / t_mixed,
            'frames_per_s': batch / t_mixed, 'frames_per_s_one_image_per_call': batch / t_loop,
            'uniform_368x496_batch_ms': t_uni * 1e3,
            'rate_per_pixel_vs_uniform_368x496_batch': (t_uni / (batch * 368 * 496)) / (t_mixed / npx),
            'frames_with_poses_identical_to_the_single_image_call': same, 'people_found': int(sum(len(r[1]) for r in res)),
            'note': 'PoseDetector.detect_batch on host images of %d different sizes (pmx_detect_images: one launch per layer over the size classes) vs '
                    'one __call__ per image (the default single-image kernels: unit mode / split-K)' % len(set(sizes))}


def precise_mode(weights_mod, device_index, with_oracle=True, shape=(482, 642)):
    """BASELINE config 5: PoseDetector(precise=True) (reference pose_detector.py:433-482: four scales 0.5 / 1 / 1.5 / 2, cubic resizes,
    averaged full-resolution maps, post-process at the original resolution) on one 482 x 642 frame with the native network: ms per
    image, the dominant kernel, and the key points against oracle/precise_ref driving the torch-CPU network restatement."""
    PD = importlib.import_module(PKG + '.pose_detector')
    H, W = shape
    img = np.random.default_rng(55).integers(0, 256, (H, W, 3), dtype=np.uint8)
    wts = weights_mod.synthetic_weights(0)
    big = (-(-int(np.ceil(H * 2 * 368 / min(H, W))) // 8) * 8, -(-int(np.ceil(W * 2 * 368 / min(H, W))) // 8) * 8)
    det = PD.PoseDetector(weights=wts, device=device_index, precise=True, max_size=big)
    try:
        # calibrate the synthetic head on the scale-1 input so that the averaged maps carry a crowd-like load
        cal = PD.resize_cubic_u8(img, int(np.ceil(W * 368 / min(H, W))), int(np.ceil(H * 368 / min(H, W))))
        cal, _ = det.pad_image(cal, 8, (104, 117, 123))
        det.engine.forward_u8(cal[None])
        paf0, heat0 = det.engine.get_maps()
        wts = weights_mod.calibrate_head(wts, paf0[0], heat0[0], heat_s=0.2, heat_t=-0.2, paf_s=1.2)
        det._weights = wts
        det.engine.set_weights({k: wts[k] for k in ('Mconv7_stage6_L1', 'Mconv7_stage6_L2')})
        err = None
        try:
            poses, scores = det(img)
        except IndexError as e:           # (the reference raises it too on a third subset match, pose_detector.py:197)
            err, poses, scores = repr(e), np.zeros((0, 18, 3)), np.zeros(0)
        n = 5
        t0 = time.perf_counter()
        for _ in range(n):
            try:
                det._detect_precise_device(img, fetch_maps=False)
            except IndexError:
                pass
        ms = (time.perf_counter() - t0) / n * 1e3
        scales = [(int(np.ceil(H * s_ * 368 / min(H, W))), int(np.ceil(W * s_ * 368 / min(H, W)))) for s_ in (0.5, 1.0, 1.5, 2.0)]
        flop = sum(FLOP_PER_FRAME * ((-(-h_ // 8) * 8) * (-(-w_ // 8) * 8)) / (368.0 * 368.0) for h_, w_ in scales)
        out = {'ms_per_image': ms, 'images_per_s': 1e3 / ms, 'image': '%dx%d' % (H, W), 'scales': [0.5, 1.0, 1.5, 2.0],
               'network_inputs': ['%dx%d' % (-(-h_ // 8) * 8, -(-w_ // 8) * 8) for h_, w_ in scales], 'flop_per_image': flop,
               'algorithmic_tflops': flop / (ms * 1e-3) / 1e12, 'peaks': int(len(det.all_peaks)), 'people': int(len(poses)), 'raised': err}
        det.engine.profile_reset(); det.engine.profile_enable(1)
        try:
            det._detect_precise_device(img, fetch_maps=False)
        except IndexError:
            pass
        prof = det.engine.profile()
        det.engine.profile_enable(False)
        if prof:
            nm, total_ms, launches, total_flop, total_issued, labels = dominant_kernel(prof)
            out['kernel_ms_per_image'] = sum(p_['total_ms'] for p_ in prof)
            if launches and total_ms > 0:
                out['dominant_kernel'] = {'kernel': nm, 'profile_labels': labels, 'total_ms': total_ms, 'launches': launches,
                                          'issued_frac': total_issued / (total_ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS,
                                          'algorithmic_frac': total_flop / (total_ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS}
        # where the time of one image goes, scale by scale: the same four pmx_precise_add_scale calls with the per-launch profile reset
        # in between (conv kernel forms by HIP kernel, the resizes and the rest of a scale's launches are the wall-clock remainder)
        try:
            eng_ = det.engine
            per_scale = []
            eng_.set_option('precise_lanes', 1)      # (one scale at a time on the context's stream: the figures of a scale running alone)
            eng_.precise_begin(H, W, 1)
            for (sh_, sw_) in scales:
                eng_.profile_reset(); eng_.profile_enable(1)
                eng_.synchronize(); t1 = time.perf_counter()
                eng_.precise_add_scale(img[None], sh_, sw_)
                eng_.synchronize(); wall = (time.perf_counter() - t1) * 1e3
                pr = eng_.profile()
                eng_.profile_enable(False)
                forms = {}
                for p_ in pr:
                    if p_['kernel'].startswith('conv'):
                        k_ = rocprof_kernel(p_['kernel'])
                        forms[k_] = forms.get(k_, 0.0) + p_['total_ms']
                issued = sum(p_['issued_flop_per_launch'] * p_['launches'] for p_ in pr)
                conv_ms = sum(forms.values())
                per_scale.append({'network_input': '%dx%d' % (-(-sh_ // 8) * 8, -(-sw_ // 8) * 8), 'wall_ms': wall, 'conv_ms': conv_ms,
                                  'conv_issued_frac': issued / (conv_ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS if conv_ms > 0 else None,
                                  'conv_ms_by_kernel': {k_: round(v_, 4) for k_, v_ in sorted(forms.items(), key=lambda kv: -kv[1])}})
            eng_.precise_finish()
            out['per_scale_running_alone'] = per_scale
            out['scales_in_flight'] = 4
        except Exception as e:          # (diagnostic only: never fail the bench line over it)
            out['per_scale_running_alone'] = {'error': repr(e)}
        finally:
            det.engine.set_option('precise_lanes', 4)
        if with_oracle:
            out['keypoint_match_vs_precise_ref'] = precise_match(det, img, wts, poses, scores)
        # the same path for a batch of same-size frames: every scale runs the n images as one batch (PoseDetector.detect_precise_batch)
        nb = 8
        batch = [img] + [np.random.default_rng(56 + i).integers(0, 256, (H, W, 3), dtype=np.uint8) for i in range(nb - 1)]
        try:
            res_b = det.detect_precise_batch(batch)             # (grows the context to batch 8; first call pays the allocation)
            t0 = time.perf_counter()
            for _ in range(3):
                det.detect_precise_batch(batch)
            ms_b = (time.perf_counter() - t0) / 3 / nb * 1e3
            same = bool(len(res_b[0][0]) == len(poses) and np.array_equal(np.asarray(res_b[0][0]), np.asarray(poses)))
            out['batch8'] = {'ms_per_image': ms_b, 'images_per_s': 1e3 / ms_b, 'algorithmic_tflops': flop / (ms_b * 1e-3) / 1e12,
                             'speedup_vs_one_image_per_call': ms / ms_b, 'first_image_same_poses_as_single_call': same,
                             'people_per_image': [int(len(r_[0])) for r_ in res_b]}
        except IndexError as e:
            out['batch8'] = {'raised': repr(e)}
        return out
    finally:
        det.engine.close()


def precise_match(det, img, wts, poses, scores):
    """detect_precise of the GPU path against oracle/precise_ref (pose_detector.py:433-482 restated; cv2.INTER_CUBIC restated -- parity
    unpinned against a real OpenCV) driving the torch-CPU network restatement: the averaged full-resolution maps within tolerance, the
    peak sets through the margin census (every disagreement a near-tie), the people both sides found within 1e-4."""
    try:
        from oracle import census, network_ref, precise_ref
        t0 = time.perf_counter()
        gpaf, gheat = det.pafs, det.heatmaps
        ref_paf, ref_heat, sizes = precise_ref.averaged_maps(lambda x: network_ref.forward(wts, x), img)
        scale = max(1.0, float(np.abs(ref_heat).max()), float(np.abs(ref_paf).max()))
        o = {'max_abs_diff_averaged_maps_over_scale': float(max(np.abs(gpaf - ref_paf).max(), np.abs(gheat - ref_heat).max()) / scale)}
        try:
            ref = precise_ref.detect_precise_from_maps(ref_paf, ref_heat)
        except IndexError as e:
            o['oracle_raised'] = repr(e)
            return o
        # GPU-side smoothed maps: the post-process is bit-exact given the maps, so the oracle's Gaussian on the GPU's own averaged maps
        from oracle import postprocess_ref
        gsm = {}
        f = census.compare_frame(det.all_peaks, ref['all_peaks'], ref['smoothed'], lambda j: gsm.setdefault(j, postprocess_ref.gaussian_filter_ref(gheat[j])),
                                 poses, scores, ref['poses'], ref['scores'])
        s = census.summarize([f], 'precise')
        for k in ('frames_identical', 'peaks_compared', 'mismatching_peaks', 'max_margin_of_a_mismatch', 'all_mismatches_are_near_ties',
                  'min_margin_of_accepted_peaks', 'max_abs_peak_score_diff', 'people_cpu', 'matched_people', 'max_abs_score_diff_matched_people'):
            o[k] = s[k]
        o['oracle_seconds'] = time.perf_counter() - t0
        return o
    except Exception as e:              # the checker must never break the measurement
        return {'error': repr(e)}


def single_image(eng, d_imgs, S, map_s):
    """BASELINE config 2 (one 368x368 image per call, the reference's own usage): latency with the input resident in HBM,
    and its own roofline -- whole-network algorithmic FLOP / call time, and the dominant kernel at batch 1 (HIP events)."""
    for _ in range(3):
        eng.detect_batch(device_ptr=d_imgs.data_ptr(), shape=(1, S, S), map_h=map_s, map_w=map_s)
        eng.results()
    n = 30
    t1 = time.perf_counter()
    for _ in range(n):
        eng.detect_batch(device_ptr=d_imgs.data_ptr(), shape=(1, S, S), map_h=map_s, map_w=map_s)
        eng.results()
    one_ms = (time.perf_counter() - t1) / n * 1e3
    flop = FLOP_PER_FRAME * (S * S / (368.0 * 368.0))
    out = {'ms_per_call': one_ms, 'frames_per_s': 1e3 / one_ms, 'calls_timed': n,
           'roofline': {'bound': 'mfma', 'algorithmic_achieved': flop / (one_ms * 1e-3) / 1e12, 'peak': FP32_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                        'algorithmic_frac': flop / (one_ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS,
                        'note': 'whole call (network + post-process + result copy), algorithmic FLOP against the fp32-MFMA peak; the issued '
                                'fraction of the dominant kernel is in roofline_dominant_kernel'}}
    eng.profile_reset()
    eng.profile_enable(True)
    for _ in range(10):
        eng.detect_batch(device_ptr=d_imgs.data_ptr(), shape=(1, S, S), map_h=map_s, map_w=map_s)
        eng.results()
    prof = eng.profile()
    eng.profile_enable(False)
    eng.profile_reset()
    if prof:
        name, total_ms, launches, total_flop, total_issued, labels = dominant_kernel(prof)
        if launches and total_ms > 0:
            ach = total_issued / (total_ms * 1e-3) / 1e12
            out['roofline_dominant_kernel'] = {'kernel': name, 'profile_labels': labels, 'bound': 'mfma', 'achieved': ach, 'peak': FP32_MFMA_PEAK_TFLOPS,
                                               'unit': 'TFLOP/s', 'frac': ach / FP32_MFMA_PEAK_TFLOPS,
                                               'algorithmic_frac': total_flop / (total_ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS,
                                               'avg_launch_ms': total_ms / launches, 'launches_timed': launches}
        out['kernel_ms_per_call'] = sum(p['total_ms'] for p in prof) / 10
    return out


if __name__ == '__main__':
    main()
