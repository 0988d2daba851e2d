"""The collectives of panmap_amd/dist.py on the RCCL backend (torch.distributed "nccl"), in a one-rank group: a GPU box of
the pool has one device, so the multi-rank exchange is covered by the two-rank gloo tests (tests/test_dist_cpu.py,
tests/test_bench_gpu.py); this one checks that every collective / dtype the exchange uses is accepted by RCCL itself and
returns what went in."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

WORKER = r'''
import os, sys, torch, numpy as np
import torch.distributed as dist
sys.path.insert(0, os.environ["PMX_ROOT"])
from panmap_amd import dist as pdist
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%s" % os.environ["PMX_PORT"], world_size=1, rank=0, device_id=dev)
assert pdist.exchange_sizes(37, dev) == [37]
h = torch.arange(1000, dtype=torch.int64, device=dev) * 7919 - 3
c = torch.arange(1000, dtype=torch.int64, device=dev) % 5 + 1
allh, sizes = pdist.allgather_histograms(h, c, 900)
assert sizes == [900] and tuple(allh.shape) == (1, 2, 900)
assert torch.equal(allh[0, 0], h[:900]) and torch.equal(allh[0, 1], c[:900])
recs = torch.zeros((50, 32), dtype=torch.uint8, device=dev)
recs[:, 22] = 4                                   # flags: has alignment
recs.view(torch.int32).view(-1, 8)[:, 6] = torch.arange(50, dtype=torch.int32, device=dev) * 3
cig = (torch.arange(150, dtype=torch.int32, device=dev) << 4)
r, arena, n_rec, bases = pdist.gather_alignments(recs, cig, 0)
assert n_rec == [50] and bases == [0] and torch.equal(r, recs) and torch.equal(arena, cig)
dist.barrier()
dist.destroy_process_group()
print("nccl one-rank collectives OK")
'''


def test_rccl_accepts_the_exchange_collectives(tmp_path):
    env = dict(os.environ, PMX_ROOT=ROOT, PMX_PORT=str(29000 + os.getpid() % 2000), HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", WORKER], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
