"""Multi-GPU sharding of the hot path: one process per GPU, images are independent (the reference handles one
image per call, pose_detector.py:430,501), so a batch is split contiguously over ranks with NO data-path
collective; the only exchange is the final gather of the fixed-size result records (RCCL all_gather over xGMI
when the backend is "nccl"; gloo on CPU in the tests).  Nothing in the reference to mirror: it has no
distributed code at all."""
import numpy as np


def shard_range(n_items, rank, world_size):
    """Contiguous split: rank r owns items [lo, hi); sizes differ by at most one (first ranks get the extra)."""
    base, extra = divmod(n_items, world_size)
    lo = rank * base + min(rank, extra)
    hi = lo + base + (1 if rank < extra else 0)
    return lo, hi


def gather_records(local_records, group=None, device=None):
    """all_gather of per-image result records (NumPy structured array, possibly of different length per rank).

    Returns the concatenation in rank order on every rank.  Records are moved as raw bytes; lengths are
    exchanged first so that uneven shards work.  `device`: torch device for the collective (cuda:N for nccl)."""
    import torch
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local_records.copy()
    world = dist.get_world_size(group)
    dev = torch.device('cpu') if device is None else torch.device(device)
    n_local = torch.tensor([len(local_records)], dtype=torch.int64, device=dev)
    counts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(counts, n_local, group=group)
    counts = [int(c.item()) for c in counts]
    itemsize = local_records.dtype.itemsize
    cap = max(counts) if counts else 0
    buf = torch.zeros(max(cap, 1) * itemsize, dtype=torch.uint8, device=dev)
    raw = np.frombuffer(np.ascontiguousarray(local_records).tobytes(), dtype=np.uint8)
    if len(raw):
        buf[:len(raw)] = torch.from_numpy(raw.copy()).to(dev)
    outs = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(outs, buf, group=group)
    parts = []
    for r in range(world):
        b = outs[r][:counts[r] * itemsize].cpu().numpy().tobytes()
        parts.append(np.frombuffer(b, dtype=local_records.dtype, count=counts[r]))
    return np.concatenate(parts) if parts else local_records[:0].copy()


class _DeviceBytes(object):
    """Zero-copy view of `nbytes` of device memory for torch (``__cuda_array_interface__`` version 2)."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {'shape': (int(nbytes),), 'typestr': '|u1', 'data': (int(ptr), False),
                                         'version': 2, 'strides': None}


def gather_device_records(engine, n_local, record_dtype, group=None):
    """RCCL all_gather straight out of the engine's device-resident result records (equal shard sizes on every rank):
    no host round trip before the collective, one device-to-host copy of the gathered records after it.

    The engine runs on its own stream, so it is synchronised first; the collective then runs on torch's RCCL stream."""
    import torch
    import torch.distributed as dist
    ptr, rec_bytes = engine.results_device_ptr()
    assert rec_bytes == np.dtype(record_dtype).itemsize
    engine.synchronize()
    dev = torch.device('cuda', engine.device)
    view = torch.as_tensor(_DeviceBytes(ptr, n_local * rec_bytes), device=dev)
    world = dist.get_world_size(group)
    out = torch.empty(world * n_local * rec_bytes, dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(out, view, group=group)
    return out.cpu().numpy().view(record_dtype)
