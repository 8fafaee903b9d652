#!/usr/bin/env python3
"""What makes the cfg2 step read 80 us right after an RCCL barrier and 70 us without one?  One rank, backend nccl.
    WORLD_SIZE=1 RANK=0 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29555 python3 tools/distgap.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29555")
os.environ.setdefault("WORLD_SIZE", "1"); os.environ.setdefault("RANK", "0"); os.environ.setdefault("LOCAL_RANK", "0")
import torch
import torch.distributed as dist

torch.cuda.set_device(0)
from pypwt_amd import BatchedWavelets

p = BatchedWavelets(1, 4096, 4096, "db4", 4, device=0)
p.fill_hash(20242, 255.0)


def step():
    p.forward(); p.inverse()


def timed(n=50, before=None):
    p.synchronize()
    if before:
        before()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    p.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


def preheat():
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.2:
        for _ in range(20):
            step()
        p.synchronize()


preheat()
print("before init_process_group: 50 steps  %.2f us" % timed())
dist.init_process_group(backend="nccl", rank=0, world_size=1)
preheat()
print("after init, no barrier:              %.2f us" % timed())
dist.barrier(device_ids=[0]); torch.cuda.synchronize()
preheat()
print("after first barrier, preheated:      %.2f us" % timed())
for k in range(3):
    preheat()
    print("barrier + cuda sync right before:    %.2f us" % timed(before=lambda: (dist.barrier(device_ids=[0]), torch.cuda.synchronize())))
preheat()
print("torch.cuda.synchronize only before:  %.2f us" % timed(before=torch.cuda.synchronize))
preheat()
print("sleep 2 ms before:                   %.2f us" % timed(before=lambda: time.sleep(0.002)))
preheat()
print("sleep 20 ms before:                  %.2f us" % timed(before=lambda: time.sleep(0.02)))
preheat()
t = torch.zeros(1, dtype=torch.float64, device="cuda")
print("all_reduce(MAX) + item before:       %.2f us" % timed(before=lambda: (dist.all_reduce(t, op=dist.ReduceOp.MAX), t.item())))
preheat()
print("200 steps after a barrier:           %.2f us" % timed(200, before=lambda: (dist.barrier(device_ids=[0]), torch.cuda.synchronize())))
print("threads in this process: %d" % len(os.listdir("/proc/self/task")))
dist.destroy_process_group()
