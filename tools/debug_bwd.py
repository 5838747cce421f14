#!/usr/bin/env python3
"""Isolate edge_bwd_v2: crafted maskB / hbits, identity-like W2 -> dpre reveals the dm tile."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphnet_amd import ops
from graphnet_amd.synthetic import synthetic_icecube86_batch
dev, mode, dt = "cuda", 1, torch.bfloat16
H1, H2 = 128, 256
H1p = 128
b = synthetic_icecube86_batch(4, seed=12).to(dev)
g = ops.knn_graph(b.x, [0, 1, 2], b.batch.to(torch.int32), b.ptr.to(torch.int32), 8)
N = g.N
nbytes = int(ops._lib.lib().gn_edgeconv_saved_bytes(N, 8, H1p, H2))
saved = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
off_maskB = ((N * 8 + N) * (H2 // 32) * 4 + 255) // 256 * 256
off_hbits = off_maskB + (N * H2 + 255) // 256 * 256
torch.manual_seed(1)
maskB = torch.randint(0, 256, (N, H2), dtype=torch.uint8, device=dev)
saved[off_maskB: off_maskB + N * H2] = maskB.reshape(-1)
saved[off_hbits: off_hbits + N * 8 * (H1p // 8)] = 255
W2 = torch.zeros(H2, H1, device=dev)
W2[torch.arange(H1), torch.arange(H1)] = 1.0            # dh[:, n] = dm[:, n], n < 128
W2Tp = ops.pack_weight(W2.t().contiguous(), [H2], dt)
gout = torch.randn(N, H2, device=dev)
PQ = torch.zeros(N, 2 * H1p, dtype=dt, device=dev)
dP = torch.zeros(N, H1p, device=dev)
dpre = torch.zeros(g.rows, H1p, dtype=dt, device=dev)
ops.edgeconv_bwd(mode, g, PQ, H1p, H2, gout, saved, W2Tp, dpre, dP)
torch.cuda.synchronize()
rows = torch.arange(N * 8, device=dev)
ic, sl = rows // 8, rows % 8
nbrv = g.nbr.long()[ic, sl] >= 0
bit = ((maskB[ic].int() >> sl.unsqueeze(1).int()) & 1).float()
ref = (gout[ic] * bit).to(dt).float()[:, :H1]
got = dpre[: N * 8].float()
err = (got - ref).abs().amax(1)
print("rows with error (first 64):", [int(i) for i in torch.nonzero(err[:64] > 1e-6).flatten().tolist()])
for r_ in (1, 8):
    print("row", r_, "got", got[r_, :8].tolist(), "\n      ref", ref[r_, :8].tolist(), "\n      gbf", gout[ic[r_], :8].to(dt).float().tolist(), "bits", bit[r_, :8].tolist())
    # which slot's bits would explain it?
    for s in range(8):
        alt = (gout[ic[r_]] * ((maskB[ic[r_]].int() >> s) & 1).float()).to(dt).float()[:H1]
        if float((alt - got[r_]).abs().max()) < 1e-6:
            print("      -> matches slot", s, "of its own centre")
    for c in range(min(N, 8)):
        for s in range(8):
            alt = (gout[c] * ((maskB[c].int() >> s) & 1).float()).to(dt).float()[:H1]
            if float((alt - got[r_]).abs().max()) < 1e-6:
                print("      -> matches centre", c, "slot", s)
