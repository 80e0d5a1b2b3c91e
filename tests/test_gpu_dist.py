"""GPU: the RCCL leg of the N>1 path (bench.py, dist.py::gather_device_records) on a one-rank "nccl" group -- the
collective reads the engine's device-resident records in place and must return exactly what pmx_get_results returns."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import pkg, ROOT

pytestmark = pytest.mark.gpu

_CHILD = r'''
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.environ["PMX_ROOT"])
import torch, torch.distributed as dist
PKG = "chainer_realtime_multi-person_pose_estimation_amd"
native = importlib.import_module(PKG + ".native"); W = importlib.import_module(PKG + ".weights"); D = importlib.import_module(PKG + ".dist")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
B = 3
eng = native.Engine(0, max_batch=B, max_h=96, max_w=128)
eng.set_weights(W.synthetic_weights(0))
imgs = np.random.default_rng(0).integers(0, 256, (B, 96, 128, 3), dtype=np.uint8)
eng.forward_u8(imgs)
paf, heat = eng.get_maps()
w = W.calibrate_head(W.synthetic_weights(0), paf[0], heat[0])
eng.set_weights({k: w[k] for k in ("Mconv7_stage6_L1", "Mconv7_stage6_L2")})
eng.detect_batch(imgs, 96, 128)
gathered = D.gather_device_records(eng, B, dst=0)
local = eng.results()
assert gathered.shape == local.shape, (gathered.shape, local.shape)
assert gathered.tobytes() == local.tobytes()
assert int(local["n_peaks"].sum()) > 0
# the pipelined path: snapshot of step k straight into the pipe's device send slot, step k + 1 enqueued BEFORE step k is shipped; a second
# batch makes the two steps differ, and a context shrunk to one person per record exercises the overflow answer of snapshot_wait
pipe = D.RecordPipe(B * local.dtype.itemsize, dst=0, device=torch.device("cuda", 0))
imgs2 = np.ascontiguousarray(imgs[::-1])
# (the engine has ONE record array: the snapshots are what keeps step k while step k + 1 runs)
got = []
for k, im in enumerate((imgs, imgs2, imgs)):
    eng.detect_batch(im, 96, 128)
    ptr, room = pipe.payload_view(k)
    eng.results_snapshot(k & 1, ptr, room)
    if k:
        n, cap, rb, ov = eng.snapshot_wait((k - 1) & 1)
        assert not ov
        got += pipe.send(k - 1, k - 1, n, cap, rb)
n, cap, rb, ov = eng.snapshot_wait(2 & 1)
got += pipe.send(2, 2, n, cap, rb)
got += pipe.flush()
assert [s for s, _ in got] == [0, 1, 2] and pipe.collectives == 3
eng.detect_batch(imgs2, 96, 128); local2 = eng.results()
assert got[0][1].tobytes() == local.tobytes() and got[2][1].tobytes() == local.tobytes() and got[1][1].tobytes() == local2.tobytes()
assert local2.tobytes() != local.tobytes()
# nslots = consumer depth + 1 (what the header documents): the slot a step reuses still has its device-to-host copy pending on the root;
# the steps completed while freeing it must be handed out by the next call, not dropped (round-4 advice: they were lost)
pipe2 = D.RecordPipe(B * local.dtype.itemsize, dst=0, device=torch.device("cuda", 0), nslots=2)
got2, seq = [], (imgs, imgs2, imgs, imgs2, imgs)
for k, im in enumerate(seq):
    eng.detect_batch(im, 96, 128)
    ptr, room = pipe2.payload_view(k)
    eng.results_snapshot(k & 1, ptr, room)             # (payload_view(k) of k >= 2 finds slot k % 2 still pending: step k - 2's copy)
    if k:
        n, cap, rb, ov = eng.snapshot_wait((k - 1) & 1)
        assert not ov
        got2 += pipe2.send(k - 1, k - 1, n, cap, rb)
n, cap, rb, ov = eng.snapshot_wait((len(seq) - 1) & 1)
got2 += pipe2.send(len(seq) - 1, len(seq) - 1, n, cap, rb)
got2 += pipe2.flush()
assert [s for s, _ in got2] == list(range(len(seq))) and pipe2.collectives == len(seq), ([s for s, _ in got2], pipe2.collectives)
assert all(r.tobytes() == (local if k % 2 == 0 else local2).tobytes() for k, (_, r) in enumerate(got2))
if int(local["n_people"].max()) > 1:
    eng.set_capacities(people=1)
    eng.detect_batch(imgs, 96, 128)
    stage = torch.empty(B * native.result_dtype(1).itemsize, dtype=torch.uint8, device="cuda")
    eng.results_snapshot(0, stage.data_ptr(), stage.numel())
    assert eng.snapshot_wait(0)[3] is True          # not final: an image has more people than the record holds ...
    grown = eng.results()                           # ... and the growing path delivers them
    assert D._people_cap(grown.dtype) >= int(local["n_people"].max()) and np.array_equal(grown["n_people"], local["n_people"])
dist.barrier(); dist.destroy_process_group()
print("RCCL_GATHER_OK")
'''


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_rccl_gather_from_device_records(native):
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()), PMX_ROOT=ROOT,
               HSA_ENABLE_IPC_MODE_LEGACY='0')
    r = subprocess.run([sys.executable, '-c', _CHILD], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and 'RCCL_GATHER_OK' in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
