#!/usr/bin/env python3
"""Micro-driver for profiling one EdgeConv kernel family in isolation (rocprofv3 --pmc runs).
usage: python3 tools/prof_edge.py [fwd|bwd|dw2|all] [events] [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphnet_amd import ops  # noqa: E402
from graphnet_amd.synthetic import synthetic_icecube86_batch  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "fwd"
events = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 5
dev = "cuda"
mode, dt = ops.MODE_BF16, torch.bfloat16
b = synthetic_icecube86_batch(events, seed=20241016).to(dev)
N = b.x.shape[0]
ptr32, batch32 = b.ptr.to(torch.int32), b.batch.to(torch.int32)
g = ops.knn_graph(b.x, [0, 1, 2], batch32, ptr32, 8)
if os.environ.get("PROF_LOCAL") == "1":      # latency experiment: every neighbour is the centre itself (gathers hit in cache)
    g.nbr.copy_(torch.arange(N, dtype=torch.int32, device=dev)[:, None].expand(N, 8))
torch.manual_seed(0)
F, H1, H2 = 256, 336, 256
H1p = ops.round_up(H1, 32)
x = torch.randn(N, F, device=dev).to(dt)
W1 = torch.randn(H1, 2 * F, device=dev) * 0.05
W2 = torch.randn(H2, H1, device=dev) * 0.05
b2 = torch.randn(H2, device=dev) * 0.1
Wpq = torch.zeros(2 * H1p, F, device=dev)
Wpq[:H1] = W1[:, :F] - W1[:, F:]
Wpq[H1p:H1p + H1] = W1[:, F:]
PQ = ops.linear_fwd(mode, [(x, F)], ops.pack_weight(Wpq, [F], dt, ops.gemm_kunit(mode)), 2 * H1p, out_lowp=True)
W2p = ops.pack_weight(W2, [H1], dt)
W2Tp = ops.pack_weight(W2.t().contiguous(), [H2], dt)
gout = torch.randn(N, H2, device=dev).to(dt)
out, mask = ops.edgeconv_fwd(mode, g, PQ, H1p, W2p, b2, H2, H1=H1)
ops.edgeconv_dw2(mode, g, PQ, H1p, H1, H2, gout, mask)
dPQ = torch.empty(N, 2 * H1p, dtype=dt, device=dev)
dpre = torch.empty(g.rows, H1p, dtype=dt, device=dev)
torch.cuda.synchronize()
ops.enable_timers(True)
for _ in range(iters):
    if what in ("fwd", "all"):
        ops.edgeconv_fwd(mode, g, PQ, H1p, W2p, b2, H2, H1=H1)
    if what in ("dw2", "all"):
        ops.edgeconv_dw2(mode, g, PQ, H1p, H1, H2, gout, mask)
    if what in ("bwd", "all"):
        ops.edgeconv_bwd(mode, g, PQ, H1p, H2, gout, mask, W2Tp, dpre, dPQ[:, :H1p])
    if what in ("dq", "all"):
        ops.edgeconv_dq_gather(mode, g, dpre, H1p, dPQ[:, H1p:])
print({k: (n, round(ms / n, 4)) for k, (n, ms) in ops.timer_summary().items()}, "N", N)

