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
