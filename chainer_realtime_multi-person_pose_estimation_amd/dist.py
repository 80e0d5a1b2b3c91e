"""Multi-GPU sharding of the hot path: one process per GPU, images are independent (the reference handles one
image per call, pose_detector.py:430,501), so a batch is split contiguously over ranks with NO data-path
collective; the only exchange is the final GATHER of the result records to rank 0 (RCCL over xGMI when the
backend is "nccl": implemented by RCCL as grouped send/recv into the root; gloo on CPU in the tests).  Nothing in the
reference to mirror: it has no distributed code at all.

Records are moved as raw bytes.  Their size depends on the rank's current person capacity (native.result_dtype), which grows
on demand, and shards may be uneven, so (count, people_cap) of every rank travels first (one small all_gather) and the root
re-packs everything at the largest capacity."""
import numpy as np


def shard_range(n_items, rank, world_size):
    """Contiguous split: rank r owns items [lo, hi); sizes differ by at most one (first ranks get the extra)."""
    base, extra = divmod(n_items, world_size)
    lo = rank * base + min(rank, extra)
    hi = lo + base + (1 if rank < extra else 0)
    return lo, hi


def _people_cap(dtype):
    return int(dtype['scores'].shape[0])


def _repack(records, people_cap):
    """Records at one person capacity -> the same records at a larger one (unused rows zero, as on the device)."""
    from . import native
    if _people_cap(records.dtype) == people_cap:
        return records
    out = np.zeros(len(records), dtype=native.result_dtype(people_cap))
    k = _people_cap(records.dtype)
    for f in ('n_people', 'n_peaks', 'status', 'n_subsets_raw'):
        out[f] = records[f]
    out['scores'][:, :k] = records['scores']
    out['poses'][:, :k] = records['poses']
    return out


def _exchange_meta(n_local, people_cap, group, dev):
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    mine = torch.tensor([n_local, people_cap], dtype=torch.int64, device=dev)
    metas = [torch.zeros(2, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(metas, mine, group=group)
    metas = [[int(v) for v in m.cpu()] for m in metas]
    return [m[0] for m in metas], [m[1] for m in metas]


def _assemble(chunks, counts, caps):
    from . import native
    cap = max(caps)
    parts = []
    for raw, n, c in zip(chunks, counts, caps):
        dt = native.result_dtype(c)
        parts.append(_repack(np.frombuffer(raw, dtype=dt, count=n), cap))
    return np.concatenate(parts) if parts else np.zeros(0, dtype=native.result_dtype(cap))


def gather_records(local_records, dst=0, group=None, device=None):
    """Final gather of per-image result records (host NumPy structured arrays) to rank `dst`.

    Returns the concatenation in rank order on `dst` and None on the other ranks (world size 1: a copy).
    `device`: torch device the collective runs on (cuda:N for nccl, cpu for gloo)."""
    import torch
    import torch.distributed as dist
    from . import native
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local_records.copy()
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    dev = torch.device('cpu') if device is None else torch.device(device)
    counts, caps = _exchange_meta(len(local_records), _people_cap(local_records.dtype), group, dev)
    sizes = [n * native.result_dtype(c).itemsize for n, c in zip(counts, caps)]
    nmax = max(max(sizes), 1)
    buf = torch.zeros(nmax, dtype=torch.uint8, device=dev)
    raw = np.frombuffer(np.ascontiguousarray(local_records).tobytes(), dtype=np.uint8)
    if len(raw):
        buf[:len(raw)] = torch.from_numpy(raw.copy()).to(dev)
    outs = [torch.empty_like(buf) for _ in range(world)] if rank == dst else None
    dist.gather(buf, outs, dst=dst, group=group)
    if rank != dst:
        return None
    return _assemble([outs[r][:sizes[r]].cpu().numpy().tobytes() for r in range(world)], counts, caps)


class _DeviceBytes(object):
    """Zero-copy view of `nbytes` of device memory for torch (``__cuda_array_interface__`` version 2)."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {'shape': (int(nbytes),), 'typestr': '|u1', 'data': (int(ptr), False),
                                         'version': 2, 'strides': None}


def gather_device_records(engine, n_local, dst=0, group=None):
    """RCCL gather to rank `dst` out of the engine's device-resident result records: no host round trip before the collective (one
    device-to-device copy into a torch-owned send buffer), one device-to-host copy of the gathered records on the root after it.

    `engine.results_layout()` first makes the records final (stream sync; capacity growth + re-run if an image needed it).
    The collective runs on torch's RCCL stream.  Returns the records of all ranks on `dst`, None elsewhere."""
    import torch
    import torch.distributed as dist
    from . import native
    people_cap, rec_bytes = engine.results_layout()
    ptr, rec_bytes2 = engine.results_device_ptr()
    assert rec_bytes == rec_bytes2
    dev = torch.device('cuda', engine.device)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    counts, caps = _exchange_meta(n_local, people_cap, group, dev)
    sizes = [n * native.result_dtype(c).itemsize for n, c in zip(counts, caps)]
    nmax = max(max(sizes), 1)
    # send buffer = a torch-owned staging tensor filled by one device-to-device copy out of the engine's records (<= 1 MB per 32
    # frames): RCCL then only ever sees allocations of torch's caching allocator (buffer registration / IPC for the xGMI transport
    # never meets a foreign hipMalloc block), and uneven shards / capacities need no special case
    view = torch.zeros(nmax, dtype=torch.uint8, device=dev)
    if sizes[rank]:
        view[:sizes[rank]].copy_(torch.as_tensor(_DeviceBytes(ptr, sizes[rank]), device=dev))
    big = torch.empty(world * nmax, dtype=torch.uint8, device=dev) if rank == dst else None
    dist.gather(view, list(big.split(nmax)) if rank == dst else None, dst=dst, group=group)
    if rank != dst:
        return None
    host = big.cpu().numpy()
    return _assemble([host[r * nmax:r * nmax + sizes[r]].tobytes() for r in range(world)], counts, caps)
