#!/usr/bin/env python3
"""Where does the HOST time of one training step go? (cProfile over eager steps, B = 1024)"""
import cProfile, io, os, pstats, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from graphnet_amd.parallel import FlatGradAllReduce
from graphnet_amd.synthetic import synthetic_icecube86_batch
m = bench.build_model("bf16").to("cuda")
opt = torch.optim.Adam(m.parameters(), lr=1e-3, eps=1e-3)
sync = FlatGradAllReduce(m.parameters())
b = synthetic_icecube86_batch(1024, seed=20241016).to("cuda")
def step():
    sync.zero_grad(); loss = m.shared_step(b); loss.backward(); sync(); opt.step()
for _ in range(5): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(20): step()
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28); print(s.getvalue()[:6000])
