"""Multi-GPU sharding of the hot path: one process per GPU, images are independent (the reference handles one
image per call, pose_detector.py:430,501), so a batch is split contiguously over ranks with NO data-path
collective; the only exchange is the final GATHER of the result records to rank 0 (RCCL over xGMI when the
backend is "nccl": implemented by RCCL as grouped send/recv into the root; gloo on CPU in the tests).  Nothing in the
reference to mirror: it has no distributed code at all.

Records are moved as raw bytes by ONE mechanism, RecordPipe: a byte stream of self-describing frames over fixed-size gathers.  A
record's size depends on the rank's current person capacity (native.result_dtype), which grows on demand, and shards may be uneven:
the frame header carries (count, people_cap) and the root re-packs everything at the largest capacity."""
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


def _one_shot(pipe, n_records, people_cap, rec_bytes, payload):
    """One frame through a fresh pipe (every rank calls it): the root gets the concatenated records, the others None."""
    done = pipe.send(0, 0, n_records, people_cap, rec_bytes, payload=payload) or []
    done += pipe.flush() or []
    if pipe.rank != pipe.dst:
        return None
    assert [step for step, _ in done] == [0], [step for step, _ in done]
    return done[0][1]


def gather_records(local_records, dst=0, group=None, device=None):
    """Final gather of per-image result records (host NumPy structured arrays) to rank `dst`: a one-shot RecordPipe (below) -- the
    only record path there is; uneven shards and mixed person capacities are carried by the frame headers.

    Returns the concatenation in rank order on `dst` and None on the other ranks (world size 1: a copy).
    `device`: torch device the collective runs on (cuda:N for nccl, cpu for gloo)."""
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local_records.copy()
    rec = np.ascontiguousarray(local_records)
    pipe = RecordPipe(rec.nbytes, dst=dst, group=group, device=device, headroom=1.0, nslots=1)
    return _one_shot(pipe, len(rec), _people_cap(rec.dtype), rec.dtype.itemsize, rec.tobytes())


class _DeviceBytes(object):
    """Zero-copy view of `nbytes` of device memory for torch (``__cuda_array_interface__`` version 2)."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {'shape': (int(nbytes),), 'typestr': '|u1', 'data': (int(ptr), False),
                                         'version': 2, 'strides': None}


def gather_device_records(engine, n_local, dst=0, group=None):
    """RCCL gather to rank `dst` out of the engine's device-resident result records, as a one-shot RecordPipe: no host round trip
    before the collective (one device-to-device copy into the pipe's torch-owned send slot -- RCCL only ever sees allocations of
    torch's caching allocator), one device-to-host copy of the gathered slots on the root after it.

    `engine.results_layout()` first makes the records final (stream sync; capacity growth + re-run if an image needed it).
    Returns the records of all ranks on `dst`, None elsewhere.  (The steady-state loop of bench.py keeps ONE pipe alive and runs
    it a step behind the compute instead: RecordPipe + pmx_results_snapshot.)"""
    import torch
    people_cap, rec_bytes = engine.results_layout()
    ptr, rec_bytes2 = engine.results_device_ptr()
    assert rec_bytes == rec_bytes2
    dev = torch.device('cuda', engine.device)
    pipe = RecordPipe(n_local * rec_bytes, dst=dst, group=group, device=dev, headroom=1.0, nslots=1)
    nbytes = n_local * rec_bytes
    if nbytes:
        s = pipe.slots[0][_SLOT_HDR + _FRAME_HDR:_SLOT_HDR + _FRAME_HDR + nbytes]
        s.copy_(torch.as_tensor(_DeviceBytes(ptr, nbytes), device=dev))
    return _one_shot(pipe, n_local, people_cap, rec_bytes, None)


# ---- pipelined gather: ONE collective per step, no per-step size exchange, no host sync in the compute path ------------------------
# (Until round 4 a second, serial path lived here: results_layout() (stream sync) -> all_gather of (count, capacity) -> gather -> D2H before
# the next batch was enqueued, i.e. a global barrier per step; gather_records / gather_device_records above are now one-shot pipes.)
# RecordPipe turns the records of a rank into a BYTE STREAM that is moved by one fixed-size `gather` per step (slot size agreed once, at
# construction):
#
#     slot   = [ int64 x 4: magic, valid_bytes, slot_seq, 0 ] [ valid_bytes of the rank's stream ] [ padding ]
#     stream = frame, frame, ...;   frame = [ int64 x 6: magic, step, n_records, people_cap, bytes_per_record, payload_bytes ] [ payload ]
#
# In the steady state the stream holds exactly one frame per step and the frame sits in the send slot ALREADY ON THE DEVICE (the engine
# snapshots its records straight into the slot, pmx_results_snapshot), so a step costs one D2D copy, a 80-byte header write and the
# gather.  Shards may be uneven and capacities may differ (the frame header carries them; the root re-packs at the largest).  If a
# rank's frame does not fit the slot (its person capacity grew beyond the agreed headroom) the frame goes through a host-side outbox and
# crosses in several slots -- later steps queue behind it, nothing is truncated, nobody needs to be told; flush() drains what is left
# (one all_reduce to agree how many more gathers).  The root assembles per-rank byte streams and hands out a step once every rank's
# frame of that step is complete.
_SLOT_MAGIC, _FRAME_MAGIC = 0x504D5853, 0x504D5846
_SLOT_HDR, _FRAME_HDR = 32, 48


class RecordPipe(object):
    def __init__(self, payload_bytes, dst=0, group=None, device=None, headroom=2.0, nslots=3):
        """Collective (every rank of `group` must call it): agrees the slot size = headroom x the largest `payload_bytes` (bytes of one
        step's records of a rank at its current capacity) over the ranks.  `device`: cuda:N (RCCL) or cpu (gloo).  `nslots`: send slots =
        steps the consumer may run behind the compute + 1 (slot of step k: k % nslots)."""
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group, self.dst = torch, dist, group, dst
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.dev = torch.device('cpu') if device is None else torch.device(device)
        t = torch.tensor([int(payload_bytes)], dtype=torch.int64, device=self.dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        cap = int(t.item())
        self.slot_bytes = (_SLOT_HDR + _FRAME_HDR + int(cap * headroom) + 255) // 256 * 256
        self.room = self.slot_bytes - _SLOT_HDR                      # stream bytes one slot carries
        self.slots = [torch.zeros(self.slot_bytes, dtype=torch.uint8, device=self.dev) for _ in range(nslots)]
        self.nslots = nslots
        # headers go host -> device from PINNED staging (one per send slot) with a non-blocking copy: from pageable memory the 80-byte copy
        # blocked the host until the device had drained most of the step that was just enqueued (measured: 29 ms of a 36 ms step)
        self.hdr = [torch.zeros(_SLOT_HDR + _FRAME_HDR, dtype=torch.uint8).pin_memory() if self.dev.type == 'cuda'
                    else torch.zeros(_SLOT_HDR + _FRAME_HDR, dtype=torch.uint8) for _ in range(nslots)]
        self.work = [None] * nslots
        # root: receive buffer, pinned host copy and a "copy done" event per slot -- the device-to-host copy of a gather is issued
        # without waiting and parsed when the NEXT gather is issued (or in flush()): a blocking copy kept the host inside it until the
        # device had drained the step enqueued in front of it, and the GPU then idled while the host enqueued the next one
        root, cuda = self.rank == dst, self.dev.type == 'cuda'
        self.recv = [torch.empty(self.world * self.slot_bytes, dtype=torch.uint8, device=self.dev) for _ in range(nslots)] if root else None
        self.host = [torch.empty(self.world * self.slot_bytes, dtype=torch.uint8).pin_memory() for _ in range(nslots)] if root and cuda else None
        self.ev = [torch.cuda.Event() for _ in range(nslots)] if root and cuda else None
        self.pending = []                                            # slots whose device-to-host copy has been issued, oldest first
        self.outbox = bytearray()                                    # host bytes waiting to cross (slow path only)
        self.inbox = [bytearray() for _ in range(self.world)] if self.rank == dst else None
        self.frames = {}                                             # step -> {rank: records}
        self.ready = []                                              # root: steps completed inside _free_slot, handed out by the next call
        self.seq = 0
        self.collectives = 0

    # -- sending side --------------------------------------------------------------------------------------------------------------
    def payload_view(self, k):
        """(device pointer or None, capacity in bytes) of the place in send slot k % nslots where a frame's payload goes: the engine snapshots
        its records there (pmx_results_snapshot).  Waits until the gather that last used the slot has read it."""
        s = k % self.nslots
        self._free_slot(s)
        t = self.slots[s][_SLOT_HDR + _FRAME_HDR:]
        return (t.data_ptr() if self.dev.type == 'cuda' else None), int(t.numel())

    def _free_slot(self, s):
        """Slot s may be rewritten once the gather that last sent it has completed.  With nslots >= 2 that was one or more steps ago:
        normally a completed-flag check; only if it is still in flight does the host wait (a blocking wait here would keep the host
        from enqueuing the next step until the device has drained the current one)."""
        if self.rank == self.dst and self.pending and s in self.pending:
            # (nslots = consumer depth + 1: the slot being reused still has its device-to-host copy pending.  The steps that parsing it
            #  completes are kept and handed out by the next send / exchange / flush -- dropped here they were lost for good)
            self.ready += self._collect(upto=s)
        w = self.work[s]
        if w is not None:
            if not w.is_completed():
                w.wait()
                if self.dev.type == 'cuda':
                    self.torch.cuda.current_stream(self.dev).synchronize()
            self.work[s] = None

    def _frame_header(self, step, n, cap, rec_bytes, payload_bytes):
        return np.array([_FRAME_MAGIC, step, n, cap, rec_bytes, payload_bytes], dtype=np.int64).tobytes()

    def send(self, k, step, n_records, people_cap, rec_bytes, payload=None):
        """One step of the pipe = one gather.  `payload` None: the frame's payload already sits in send slot k % nslots (payload_view);
        else host bytes (np.uint8 / bytes) of the records.  Returns what exchange() returns."""
        nbytes = int(n_records) * int(rec_bytes)
        s = k % self.nslots
        in_place = payload is None and not self.outbox and _FRAME_HDR + nbytes <= self.room
        if in_place:
            head = np.frombuffer(np.array([_SLOT_MAGIC, _FRAME_HDR + nbytes, self.seq, 0], dtype=np.int64).tobytes()
                                 + self._frame_header(step, n_records, people_cap, rec_bytes, nbytes), dtype=np.uint8)
            self.hdr[s].numpy()[:] = head            # (slot s's previous gather has completed: payload_view waited for it)
            self.slots[s][:_SLOT_HDR + _FRAME_HDR].copy_(self.hdr[s], non_blocking=True)
            return self._gather(s)
        if payload is None:                                          # the frame is in the slot but cannot go from there: pull it to the host
            payload = self.slots[s][_SLOT_HDR + _FRAME_HDR:_SLOT_HDR + _FRAME_HDR + nbytes].cpu().numpy().tobytes()
        payload = bytes(memoryview(np.ascontiguousarray(payload)).cast('B')) if not isinstance(payload, (bytes, bytearray)) else bytes(payload)
        assert len(payload) == nbytes, (len(payload), nbytes)
        self.outbox += self._frame_header(step, n_records, people_cap, rec_bytes, nbytes) + payload
        return self.exchange(k)

    def exchange(self, k=0):
        """Send the next (up to one slot of) outbox bytes -- an empty slot if there is nothing to send."""
        s = k % self.nslots
        self._free_slot(s)                                           # (slow path: the slot may not have gone through payload_view)
        n = min(len(self.outbox), self.room)
        chunk = bytes(self.outbox[:n])
        del self.outbox[:n]
        head = np.array([_SLOT_MAGIC, n, self.seq, 0], dtype=np.int64).tobytes()
        buf = np.frombuffer(head + chunk, dtype=np.uint8)
        self.slots[s][:len(buf)].copy_(self.torch.from_numpy(buf.copy()))
        return self._gather(s)

    def pending_slots(self):
        return (len(self.outbox) + self.room - 1) // self.room

    def flush(self):
        """Collective: drain every rank's outbox (extra gathers, the same number on every rank).  Returns the steps completed meanwhile."""
        t = self.torch.tensor([self.pending_slots()], dtype=self.torch.int64, device=self.dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)
        done = []
        for i in range(int(t.item())):
            done += self.exchange(i) or []
        if self.rank == self.dst:
            done += self.ready + self._collect()
            self.ready = []
        busy = False
        for s in range(self.nslots):
            if self.work[s] is not None:
                busy = busy or not self.work[s].is_completed()
                self.work[s].wait()
                self.work[s] = None
        # (an RCCL work's wait() only orders torch's current stream behind the collective; the engine's stream and the pinned header
        #  staging are not ordered against it, so before the slots may be rewritten the host really waits)
        if busy and self.dev.type == 'cuda':
            self.torch.cuda.current_stream(self.dev).synchronize()
        return done

    # -- the collective + the root's assembler ---------------------------------------------------------------------------------------
    def _gather(self, s):
        self.seq += 1
        self.collectives += 1
        outs = list(self.recv[s].split(self.slot_bytes)) if self.rank == self.dst else None
        self.work[s] = self.dist.gather(self.slots[s], outs, dst=self.dst, group=self.group, async_op=True)
        if self.rank != self.dst:
            return None
        # root: first hand out what earlier gathers delivered (their copies are long done), then issue this one's copy without waiting
        done = self.ready + self._collect()
        self.ready = []
        self.work[s].wait()                                          # (stream-ordered: the slots of all ranks are in recv[s] before the copy)
        if self.host is not None:
            self.host[s].copy_(self.recv[s], non_blocking=True)
            self.ev[s].record()
            self.pending.append(s)
        else:                                                        # gloo: the collective has completed on return from wait()
            self._absorb(self.recv[s].numpy())
            done += self._parse()
        return done

    def _collect(self, upto=None):
        """Root: parse every pending slot (oldest first; stops after slot `upto` if given) -- waits for its copy event, normally long set."""
        done = []
        while self.pending:
            s = self.pending.pop(0)
            self.ev[s].synchronize()
            self._absorb(self.host[s].numpy())
            done += self._parse()
            if upto is not None and s == upto:
                break
        return done

    def _absorb(self, host):
        for r in range(self.world):
            slot = host[r * self.slot_bytes:(r + 1) * self.slot_bytes]
            magic, valid, _, _ = np.frombuffer(slot[:_SLOT_HDR].tobytes(), dtype=np.int64)
            assert magic == _SLOT_MAGIC and 0 <= valid <= self.room, ('corrupt slot from rank %d' % r, int(magic), int(valid))
            self.inbox[r] += slot[_SLOT_HDR:_SLOT_HDR + int(valid)].tobytes()

    def _parse(self):
        from . import native
        for r in range(self.world):
            box = self.inbox[r]
            while len(box) >= _FRAME_HDR:
                magic, step, n, cap, rec_bytes, nbytes = np.frombuffer(bytes(box[:_FRAME_HDR]), dtype=np.int64)
                assert magic == _FRAME_MAGIC and nbytes == n * rec_bytes and rec_bytes == native.result_dtype(int(cap)).itemsize, \
                    ('corrupt frame from rank %d' % r, int(magic), int(n), int(cap), int(rec_bytes), int(nbytes))
                if len(box) < _FRAME_HDR + nbytes:
                    break
                rec = np.frombuffer(bytes(box[_FRAME_HDR:_FRAME_HDR + int(nbytes)]), dtype=native.result_dtype(int(cap)), count=int(n))
                del box[:_FRAME_HDR + int(nbytes)]
                self.frames.setdefault(int(step), {})[r] = rec
        done = []
        for step in sorted(self.frames):
            if len(self.frames[step]) == self.world:
                per_rank = self.frames.pop(step)
                cap = max(_people_cap(per_rank[r].dtype) for r in range(self.world))
                done.append((step, np.concatenate([_repack(per_rank[r], cap) for r in range(self.world)])))
        return done
