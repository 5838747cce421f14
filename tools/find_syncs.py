#!/usr/bin/env python3
"""Does a training step contain a host<->device synchronisation? (torch sync-debug mode) + host loop timing"""
import os, sys, time, torch, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from graphnet_amd.parallel import FlatGradAllReduce
from graphnet_amd.synthetic import synthetic_icecube86_batch
m = bench.build_model("bf16").to("cuda")
opt = torch.optim.Adam(m.parameters(), lr=1e-3, eps=1e-3, fused=True)
sync = FlatGradAllReduce(m.parameters())
b = synthetic_icecube86_batch(1024, seed=20241016).to("cuda")
def step():
    sync.zero_grad(); loss = m.shared_step(b); loss.backward(); sync(); opt.step()
for _ in range(5): step()
torch.cuda.synchronize()
torch.cuda.set_sync_debug_mode("warn")
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    step()
    torch.cuda.set_sync_debug_mode("default")
print("sync warnings:", len(w))
for x in w[:10]: print("  ", str(x.message)[:200], x.filename, x.lineno)
torch.cuda.synchronize()
for trial in range(3):
    t0 = time.perf_counter()
    for _ in range(20): step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"host enqueue of 20 steps: {1e3*(t1-t0)/20:.2f} ms/step; until GPU done: {1e3*(t2-t0)/20:.2f} ms/step")
