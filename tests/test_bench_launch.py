"""bench.py's own multi-rank launcher (`--gpus N` without torch.distributed.run around it).

CPU: it refuses loudly when fewer GPUs than ranks are visible.  GPU (`-m gpu`): two ranks on the one GPU of the box
(`--backend gloo`, the single-GPU smoke mode) produce `n_gpus: 2` and exactly the records a single rank computes for the
same global batch (images are sharded contiguously; the gather to rank 0 keeps rank order)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

BENCH = os.path.join(ROOT, 'bench.py')


def _run(args, timeout=900):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    env.pop('WORLD_SIZE', None)
    env.pop('RANK', None)
    return subprocess.run([sys.executable, BENCH] + args, env=env, capture_output=True, text=True, timeout=timeout, cwd=ROOT)


def test_launcher_refuses_without_enough_gpus():
    import torch
    if torch.cuda.device_count() >= 4:
        pytest.skip('4+ GPUs present')
    r = _run(['--gpus', '4', '--steps', '1', '--warmup', '0'], timeout=300)
    assert r.returncode != 0
    assert 'GPU' in (r.stderr + r.stdout)


@pytest.mark.gpu
def test_two_ranks_equal_one_rank(tmp_path):
    common = ['--steps', '3', '--warmup', '1', '--no-cpu-baseline', '--no-profile', '--no-extras']      # (3 steps: the pipeline runs one step behind)
    f2, f1 = str(tmp_path / 'r2.npy'), str(tmp_path / 'r1.npy')
    r2 = _run(['--gpus', '2', '--backend', 'gloo', '--batch', '4', '--dump-records', f2] + common)
    assert r2.returncode == 0, r2.stdout[-2000:] + r2.stderr[-4000:]
    line2 = json.loads([l for l in r2.stdout.splitlines() if l.startswith('{')][-1])
    assert line2['n_gpus'] == 2 and line2['config']['global_batch'] == 8 and line2['config']['records_gathered'] == 8
    assert len(line2['per_rank_frames_per_s']) == 2 and line2['gather_ms_per_step_rank0'] >= 0
    assert line2['collectives_per_step'] == 1.0 and 'RecordPipe' in line2['records_path']      # one gather per step, nothing else
    # the line validates itself: rank 0 re-ran both shards and compared them with what came through the gather
    assert line2['shard_records_match'] is True and line2['records_compared'] == 8 and line2['max_abs_score_diff'] <= 1e-5
    assert line2['shard_validation']['ranks_checked'] == [0, 1] and line2['shard_validation']['bitwise_equal_records'] == 8
    assert line2['scaling_record_valid'] is False and 'gloo' in line2['scaling_record_invalid_reason']      # shared GPU: never a scaling figure
    r1 = _run(['--gpus', '1', '--batch', '8', '--dump-records', f1] + common)
    assert r1.returncode == 0, r1.stdout[-2000:] + r1.stderr[-4000:]
    line1 = json.loads([l for l in r1.stdout.splitlines() if l.startswith('{')][-1])
    assert line1['n_gpus'] == 1 and line1['config']['global_batch'] == 8
    a, b = np.load(f2), np.load(f1)
    assert len(a) == len(b) == 8
    assert np.array_equal(a['n_people'], b['n_people']) and np.array_equal(a['n_peaks'], b['n_peaks'])
    assert int(a['n_peaks'].sum()) > 0
    n = a['n_people']
    for i in range(8):
        # batch 4 and batch 8 may pick different conv kernels (split-K at small launches): poses exact, scores to 1e-5
        assert np.array_equal(a['poses'][i, :n[i]], b['poses'][i, :n[i]])
        assert np.allclose(a['scores'][i, :n[i]], b['scores'][i, :n[i]], rtol=0, atol=1e-5)


@pytest.mark.gpu
def test_eight_ranks_through_the_launcher(tmp_path):
    """Pre-flight for the driver's 8-GPU run (BASELINE config 4): `bench.py --gpus 8` starts its own eight ranks, every rank takes the
    N > 1 path (RecordPipe, one gather per step, pipelined) and rank 0 prints what the SCALE record needs -- here over gloo with the
    eight ranks sharing the one GPU of the test box (the RCCL transport itself needs eight GPUs: only the driver has them)."""
    f8 = str(tmp_path / 'r8.npy')
    r = _run(['--gpus', '8', '--backend', 'gloo', '--batch', '2', '--steps', '3', '--warmup', '1', '--no-cpu-baseline', '--no-profile', '--no-extras',
              '--dump-records', f8], timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])
    assert line['n_gpus'] == 8 and line['ranks_seen'] == list(range(8)) and len(line['devices']) == 8
    assert line['config']['global_batch'] == 16 and line['config']['records_gathered'] == 16
    assert line['collectives_per_step'] == 1.0 and 'RecordPipe' in line['records_path']
    assert len(line['per_rank_frames_per_s']) == 8 and len(line['per_rank_frames_per_s_min_max']) == 2
    assert line['gather_ms_per_step_rank0'] >= 0 and line['scaling'] == 'weak'
    assert line['shard_records_match'] is True and line['records_compared'] == 16 and line['shard_validation']['ranks_checked'] == list(range(8))
    rec = np.load(f8)
    assert len(rec) == 16 and int(rec['n_peaks'].sum()) > 0


@pytest.mark.gpu
def test_one_rank_through_the_rccl_gather_equals_plain_run(tmp_path):
    """`--force-gather`: N = 1 with a one-rank "nccl" (RCCL) process group, records routed through dist.RecordPipe (engine snapshot into the
    device send slot, one RCCL gather per step, pipelined one step behind the compute) -- the branch every rank takes at N > 1 -- must give
    the records of the plain run, and the line must say which path and device it used."""
    common = ['--gpus', '1', '--batch', '4', '--steps', '3', '--warmup', '1', '--no-cpu-baseline', '--no-profile', '--no-extras']
    fg, fp = str(tmp_path / 'g.npy'), str(tmp_path / 'p.npy')
    rg = _run(common + ['--force-gather', '--dump-records', fg])
    assert rg.returncode == 0, rg.stdout[-2000:] + rg.stderr[-4000:]
    lg = json.loads([l for l in rg.stdout.splitlines() if l.startswith('{')][-1])
    assert 'rccl' in lg['backend'] and 'RecordPipe' in lg['records_path'] and 'RCCL gather' in lg['records_path'] and lg['ranks_seen'] == [0]
    assert lg['collectives_per_step'] == 1.0 and 'RCCL' in lg['config']['parallelism']
    assert len(lg['devices']) == 1 and (lg['devices'][0]['uuid'] or lg['devices'][0]['pci_bus_id'])
    assert lg['shard_records_match'] is True and lg['records_compared'] == 4 and lg['scaling_record_valid'] is True
    rp = _run(common + ['--dump-records', fp])
    assert rp.returncode == 0, rp.stdout[-2000:] + rp.stderr[-4000:]
    lp = json.loads([l for l in rp.stdout.splitlines() if l.startswith('{')][-1])
    assert lp['backend'] is None and 'pmx_get_results' in lp['records_path'] and 'RCCL' not in lp['config']['parallelism']
    assert 'shard_records_match' not in lp                                             # (no gather, nothing to validate)
    a, b = np.load(fg), np.load(fp)
    assert a.dtype == b.dtype and a.tobytes() == b.tobytes() and int(a['n_peaks'].sum()) > 0

