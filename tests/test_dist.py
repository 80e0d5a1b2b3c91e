"""CPU, world_size 2 over gloo: batch sharding + final gather of result records (the N>1 path of bench.py)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from conftest import pkg, ROOT, PKG


def test_shard_range_partitions_exactly():
    d = pkg('dist')
    for n in (0, 1, 7, 32, 33, 255, 256):
        for world in (1, 2, 3, 4, 8):
            spans = [d.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_items, q):
    import importlib
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    d = importlib.import_module(PKG + '.dist')
    native = importlib.import_module(PKG + '.native')
    lo, hi = d.shard_range(n_items, rank, world)
    # contexts of other ranks have grown their person capacity (crowd images): the root re-packs everything at the largest capacity
    rec = np.zeros(hi - lo, dtype=native.result_dtype((64, 256, 128)[rank % 3]))
    for i in range(lo, hi):     # deterministic fake "results" keyed by the global image index
        rec[i - lo]['n_people'] = i % 5
        rec[i - lo]['n_peaks'] = 10 * i
        rec[i - lo]['scores'][:3] = [i, i + 0.5, -i]
        rec[i - lo]['poses'][0, 0] = [i, 2 * i, 2]
    allrec = d.gather_records(rec, dst=0)
    if rank == 0:
        ok = len(allrec) == n_items and allrec.dtype == native.result_dtype(256) and all(
            allrec[i]['n_peaks'] == 10 * i and allrec[i]['poses'][0, 0, 1] == 2 * i and allrec[i]['scores'][2] == -i
            for i in range(n_items))
    else:
        ok = allrec is None          # a root gather: only rank 0 holds the result
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('n_items', [8, 7])
def test_gather_records_world2_gloo(n_items):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_items, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]


def test_gather_records_world8_gloo_uneven_shards_mixed_capacities():
    """The shape of BASELINE config 4 (8 ranks), with what a real run can throw at the gather: 255 images do not divide by 8 (shards
    of 32 and 31) and the ranks' record capacities differ (64 / 256 / 128)."""
    world, n_items = 8, 255
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_items, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=120)
    assert sorted(res) == [(r, True) for r in range(world)]


# ---- RecordPipe: the pipelined one-collective-per-step gather (dist.py) -----------------------------------------------------------------
def _fake_records(native, lo, hi, cap, step):
    rec = np.zeros(hi - lo, dtype=native.result_dtype(cap))
    for i in range(lo, hi):
        rec[i - lo]['n_people'] = (i + step) % 5
        rec[i - lo]['n_peaks'] = 10 * i + step
        rec[i - lo]['scores'][:2] = [i + 0.25 * step, -i]
        rec[i - lo]['poses'][1, 3] = [i, step, 2]
    return rec


def _pipe_worker(rank, world, port, n_items, steps, headroom, grow_at, q):
    import importlib
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    d = importlib.import_module(PKG + '.dist')
    native = importlib.import_module(PKG + '.native')
    lo, hi = d.shard_range(n_items, rank, world)
    cap0 = 64
    pipe = d.RecordPipe((hi - lo) * native.result_dtype(cap0).itemsize, dst=0, headroom=headroom)
    got = []
    for k in range(steps):
        # rank 1's person capacity grows 4x at step `grow_at` (a crowd image): its frames no longer fit one slot
        cap = cap0 * 4 if (rank == 1 and grow_at is not None and k >= grow_at) else cap0
        rec = _fake_records(native, lo, hi, cap, k)
        ptr, room = pipe.payload_view(k)
        raw = np.frombuffer(rec.tobytes(), dtype=np.uint8)
        if len(raw) <= room and k % 2 == 0:
            # the steady-state path: the payload is written straight into the send slot (what pmx_results_snapshot does on the device)
            pipe.slots[k % pipe.nslots][d._SLOT_HDR + d._FRAME_HDR:d._SLOT_HDR + d._FRAME_HDR + len(raw)] = torch.from_numpy(raw.copy())
            done = pipe.send(k, k, len(rec), cap, rec.dtype.itemsize)
        else:
            done = pipe.send(k, k, len(rec), cap, rec.dtype.itemsize, payload=raw)
        got += done or []
    got += pipe.flush() or []
    ok = True
    if rank == 0:
        ok = [s for s, _ in got] == list(range(steps))
        for s, allrec in got:
            cap = 256 if (grow_at is not None and s >= grow_at and world > 1) else 64
            ok = ok and len(allrec) == n_items and allrec.dtype == native.result_dtype(cap)
            ok = ok and all(allrec[i]['n_peaks'] == 10 * i + s and allrec[i]['poses'][1, 3, 1] == s and allrec[i]['scores'][1] == -i
                            and allrec[i]['n_people'] == (i + s) % 5 for i in range(n_items))
    else:
        ok = got == []
    q.put((rank, bool(ok), pipe.collectives))
    dist.barrier()
    dist.destroy_process_group()


def _run_pipe(world, n_items, steps, headroom, grow_at):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pipe_worker, args=(r, world, port, n_items, steps, headroom, grow_at, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=120)
    return sorted(res)


def test_record_pipe_steady_state_is_one_gather_per_step():
    """8 ranks, 255 images (shards of 32 / 31), 6 steps: every step arrives complete and in order on the root, records identical to what
    the serial gather would deliver, with exactly one collective per step (no per-step size exchange)."""
    res = _run_pipe(8, 255, 6, 2.0, None)
    assert [(r, ok) for r, ok, _ in res] == [(r, True) for r in range(8)]
    assert all(n == 6 for _, _, n in res), res


def test_record_pipe_capacity_growth_fragments_but_never_truncates():
    """Rank 1's record capacity grows 4x at step 2 (64 -> 256 persons per record): its frames are larger than the agreed slot (headroom
    1.25 x the initial size), cross in several slots, later steps queue behind them, flush() drains the rest -- the root still hands
    out every step, complete, in order, re-packed at the largest capacity; nobody exchanged sizes."""
    res = _run_pipe(2, 9, 5, 1.25, 2)
    assert [(r, ok) for r, ok, _ in res] == [(0, True), (1, True)]
    assert res[0][2] == res[1][2] > 5                 # the same number of gathers on both ranks, more than the steps (fragments)
