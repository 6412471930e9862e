"""ratsdf.framecast on the device path under backend "nccl" (= RCCL).  A one-GPU box cannot hold two RCCL ranks
("Duplicate GPU detected": RCCL refuses ranks that share a device), so the process group here has ONE rank: the
broadcast still goes through RCCL on the caster's side stream, and everything around it -- ring slots, the events
between the side stream and the engine's stream in both directions, headers through page-locked memory, the engine
integrating straight out of the wire buffer -- is what N ranks run.  (Two ranks on the one device over gloo:
tests/test_bench_launch.py.)  In a child process: the process group is the child's own."""
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent

CODE = r"""
import os, sys
sys.path.insert(0, r'%(root)s/ra-slam_amd'); sys.path.insert(0, r'%(root)s/tests')
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '%(port)d')
import numpy as np, torch, torch.distributed as dist
torch.cuda.init()
import ratsdf
from ratsdf import framecast, multi, synthetic
from ratsdf._abi import Engine
from oracle_binding import load_oracle
from parity import assert_maps_equal
dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
vs, md, C = 0.02, 4.0, 3
stream = synthetic.stream('room', 11, scale=0.25, noise=True, holes=True)   # 3 chunks + a tail of 2: the ring of 2 wraps
H, W = stream[0]['depth'].shape
gpu = ratsdf.TSDFGrid(vs, 6 * vs)
cpu = Engine(load_oracle(), vs, 6 * vs)
ext = torch.cuda.ExternalStream(gpu.stream(), device=dev)
fc = framecast.FrameCaster(H, W, C, ring=2, src=0, device=dev)
packed = [torch.from_numpy(framecast.pack_chunk(stream[c:c + C], md, H, W, C, first_frame_no=c)).to(dev)
          for c in range(0, len(stream), C)]
torch.cuda.synchronize()
posted = 0
for k in range(len(packed)):
    while posted < len(packed) and fc.can_post():
        fc.post(packed[posted]); posted += 1
    ch = fc.take(ext, verify=(k == 0))
    framecast.integrate_chunk(gpu, ch)
    fc.done(ch, ext)
for f in stream:
    cpu.integrate(f['rgb'], f['depth'], f['ht'], f['lt'], md, f['intrinsics'], f['pose'])
w = assert_maps_equal(gpu, cpu)
assert fc.posted == fc.taken == fc.finished == len(packed)
# the directory delta exchange on device tensors under the same backend
ex = multi.DirectoryDeltaExchange(engine=gpu, device=dev, delta_capacity=4096)
for rep in range(3):
    ex.fill_from_engine(gpu)
    ex.all_gather()
    more = synthetic.frame('room', 20 + rep, scale=0.25, noise=True)
    gpu.integrate(more['rgb'], more['depth'], more['ht'], more['lt'], md, more['intrinsics'], more['pose'])
ex.fill_from_engine(gpu); ex.all_gather()
got = ex.result()[0]
_, want = gpu.dump_directory()
key = lambda b: np.sort(b['x'].astype(np.int64) * 2**32 + b['y'].astype(np.int64) * 2**16 + b['z'].astype(np.int64))
assert len(got) == len(want) and np.array_equal(key(got), key(want)), (len(got), len(want))
dist.destroy_process_group()
print('FRAMECAST_NCCL_OK', w['tsdf'], w['prob'])
"""


@pytest.mark.gpu
def test_device_path_under_the_nccl_backend():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    r = subprocess.run([sys.executable, "-c", CODE % dict(root=str(ROOT), port=port)], capture_output=True, text=True,
                       timeout=300, cwd=str(ROOT))
    assert r.returncode == 0 and "FRAMECAST_NCCL_OK" in r.stdout, (r.stdout + r.stderr)[-3000:]
