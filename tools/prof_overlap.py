#!/usr/bin/env python3
"""Does the HBM-bound source gather run UNDER the compute-bound dW2 kernel when both are launched at once on two
streams?  (dW2: 8 waves x 192 VGPRs per CU leave 128 registers per SIMD; the gather needs 80.)
usage: python3 tools/prof_overlap.py [events] [iters]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphnet_amd import ops  # noqa: E402
from graphnet_amd.synthetic import synthetic_icecube86_batch  # noqa: E402

events = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = "cuda"
mode, dt = ops.MODE_BF16, torch.bfloat16
b = synthetic_icecube86_batch(events, seed=20241016).to(dev)
N = b.x.shape[0]
ptr32, batch32 = b.ptr.to(torch.int32), b.batch.to(torch.int32)
g = ops.knn_graph(b.x, [0, 1, 2], batch32, ptr32, 8)
g.build_reverse()
torch.manual_seed(0)
F, H1, H2 = 256, 336, 256
H1p = ops.round_up(H1, 32)
PQ = (torch.randn(N, 2 * H1p, device=dev) * 0.5).to(dt)
W2 = torch.randn(H2, H1, device=dev) * 0.05
b2 = torch.randn(H2, device=dev) * 0.1
W2p = ops.pack_weight(W2, [H1], dt)
W2Tp = ops.pack_weight(W2.t().contiguous(), [H2], dt)
gout = torch.randn(N, H2, device=dev).to(dt)
out, mask = ops.edgeconv_fwd(mode, g, PQ, H1p, W2p, b2, H2, H1=H1)
ops.edgeconv_dw2(mode, g, PQ, H1p, H1, H2, gout, mask)
dPQ = torch.empty(N, 2 * H1p, dtype=dt, device=dev)
dpre = (torch.randn(g.rows, H1p, device=dev) * 0.1).to(dt)
torch.cuda.synchronize()
side = torch.cuda.Stream()
main = torch.cuda.current_stream()


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / iters


def dw2():
    ops.edgeconv_dw2(mode, g, PQ, H1p, H1, H2, gout, mask)


def dq():
    ops.edgeconv_dq_gather(mode, g, dpre, H1p, dPQ[:, H1p:])


def bwd():
    ops.edgeconv_bwd(mode, g, PQ, H1p, H2, gout, mask, W2Tp, dpre, dPQ[:, :H1p])


def both(first, second):
    def f():
        side.wait_stream(main)
        first()                                   # main stream
        with torch.cuda.stream(side):
            second()
        main.wait_stream(side)
    return f


res = {"N": N, "dw2_ms": timed(dw2), "dq_ms": timed(dq), "bwd_ms": timed(bwd)}
res["dw2_then_dq_serial_ms"] = timed(lambda: (dw2(), dq()))
res["dw2_main_dq_side_ms"] = timed(both(dw2, dq))
res["dq_main_dw2_side_ms"] = timed(both(dq, dw2))
res["bwd_main_dw2_side_ms"] = timed(both(bwd, dw2))
print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in res.items()})
