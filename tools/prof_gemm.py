#!/usr/bin/env python3
"""Micro-driver: per-node GEMMs of the DynEdge path in isolation. usage: prof_gemm.py [which] [iters]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphnet_amd import ops
which = sys.argv[1] if len(sys.argv) > 1 else "all"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev, mode, dt = "cuda", 1, torch.bfloat16
M = 158439
ku = ops.gemm_kunit(mode)
torch.manual_seed(0)
x = torch.randn(M, 256, device=dev).to(dt)
xs = [torch.randn(M, 32, device=dev).to(dt)] + [torch.randn(M, 256, device=dev).to(dt) for _ in range(4)]
Wpq = ops.pack_weight(torch.randn(704, 256, device=dev), [256], dt, ku)
Wpost = ops.pack_weight(torch.randn(336, 32 + 1024, device=dev), [32, 256, 256, 256, 256], dt, ku)
dY = torch.randn(M, 704, device=dev).to(dt)
torch.cuda.synchronize()
def t(fn, n=iters):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(True), torch.cuda.Event(True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
if which in ("pq", "all"):
    us = t(lambda: ops.linear_fwd(mode, [(x, 256)], Wpq, 704, out_lowp=True))
    print(f"PQ  gemm  [{M}x256]x[704x256]^T bf16-out: {us:8.1f} us  {2*M*256*704/us/1e6:7.1f} TF/s")
    us = t(lambda: ops.linear_fwd(mode, [(x, 256)], Wpq, 704))
    print(f"PQ  gemm  f32-out:                       {us:8.1f} us  {2*M*256*704/us/1e6:7.1f} TF/s")
if which in ("post", "all"):
    us = t(lambda: ops.linear_fwd(mode, [(a, w) for a, w in zip(xs, [32, 256, 256, 256, 256])], Wpost, 336, relu=True))
    print(f"post gemm [{M}x1056]x[336]:               {us:8.1f} us  {2*M*1056*336/us/1e6:7.1f} TF/s")
if which in ("wgrad", "all"):
    us = t(lambda: ops.linear_wgrad(mode, dY, 704, [(x, 256)]))
    print(f"wgrad     [704x{M}]x[{M}x256]:            {us:8.1f} us  {2*M*256*704/us/1e6:7.1f} TF/s")
if which in ("dx", "all"):
    Wdx = ops.pack_weight(torch.randn(256, 704, device=dev), [704], dt, ku)
    acc = torch.zeros(M, 256, dtype=dt, device=dev)
    us = t(lambda: ops.linear_fwd(mode, [(dY, 704)], Wdx, 256, out=acc, accum=True))
    print(f"dx   gemm [{M}x704]x[256] accum:          {us:8.1f} us  {2*M*704*256/us/1e6:7.1f} TF/s")
if which in ("colsum", "all"):
    us = t(lambda: ops.colsum(dY.float(), 336))
    print(f"colsum    [{M}x336 of 704]:               {us:8.1f} us  {M*336*4/us/1e6:7.2f} TB/s")
