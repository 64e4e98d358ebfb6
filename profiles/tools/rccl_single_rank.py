#!/usr/bin/env python3
"""The table gather of bench.py --gpus N through RCCL with a world of one rank (all a one-GPU box allows): the same
init_process_group / all_gather / gather calls as on a node, staged on the device.  Prints OK or raises."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
import numpy as np
import torch
import torch.distributed as dist
from imageprocessing_amd import sharding

dist.init_process_group("nccl", device_id=torch.device("cuda:0"))
torch.cuda.set_device(0)
g = sharding.TableGatherer(device="cuda:0")
rng = np.random.default_rng(0)
for size in (1000, 5_000_000, 300):
    blob = rng.integers(0, 256, size, dtype=np.uint8)
    got = g.gather(blob)
    assert len(got) == 1 and np.array_equal(got[0], blob)
t = torch.tensor([3.5], device="cuda:0", dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
torch.cuda.synchronize()
dist.destroy_process_group()
print("OK rccl single rank")
