#!/usr/bin/env python3
"""Why is the cfg2 step 80 us under bench.py's N > 1 path (torch imported first) and 70 us without torch?
Times the same 200-step loop in child processes that differ only in what is loaded, and prints which
libamdhip64 each one mapped.   python3 tools/torchcoexist.py"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import sys, time, os
mode = sys.argv[1]
sys.path.insert(0, %r)
if mode == "torch_first":
    import torch
if mode == "torch_first_cuda_init":
    import torch; torch.cuda.init(); torch.zeros(1, device="cuda")
from pypwt_amd import BatchedWavelets
from pypwt_amd import _lib
_lib.load()
if mode == "lib_first_then_torch":
    import torch
if mode == "lib_first_then_torch_cuda_init":
    import torch; torch.cuda.init(); torch.zeros(1, device="cuda")
p = BatchedWavelets(1, 4096, 4096, "db4", 4, device=0)
p.fill_hash(20242, 255.0)
def step():
    p.forward(); p.inverse()
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.3:
    for _ in range(20): step()
    p.synchronize()
best = 1e9
for rep in range(5):
    p.synchronize(); t0 = time.perf_counter()
    for _ in range(200): step()
    p.synchronize(); best = min(best, (time.perf_counter() - t0) / 200)
# host enqueue cost of one step (no sync)
p.synchronize(); t0 = time.perf_counter()
for _ in range(200): step()
enq = (time.perf_counter() - t0) / 200
p.synchronize()
hips = sorted(set(l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l or "libhsa-runtime" in l))
print("%%-32s step %%6.2f us   host enqueue %%6.2f us   %%s" %% (mode, best * 1e6, enq * 1e6, hips))
''' % ROOT

for mode in ("no_torch", "torch_first", "torch_first_cuda_init", "lib_first_then_torch", "lib_first_then_torch_cuda_init"):
    r = subprocess.run([sys.executable, "-c", CODE, mode], capture_output=True, text=True, timeout=600)
    print(r.stdout.strip() or ("%s FAILED: %s" % (mode, r.stderr[-800:])))
