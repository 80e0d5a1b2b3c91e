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
