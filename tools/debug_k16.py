#!/usr/bin/env python3
"""v2 (persistent) vs v1 (tiled) edge kernels at k=16: where do dpre / dPQ / out differ?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphnet_amd import ops
from graphnet_amd.synthetic import synthetic_icecube86_batch
kk = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev, mode, dt = "cuda", 1, torch.bfloat16
b = synthetic_icecube86_batch(40, seed=12).to(dev)
x3 = b.x.clone(); x3[3:3 + kk + 6, :3] = x3[2, :3]
g = ops.knn_graph(x3, [0, 1, 2], b.batch.to(torch.int32), b.ptr.to(torch.int32), kk)
N, F, H1, H2 = g.N, 256, 336, 256
H1p = 352
torch.manual_seed(12)
x = torch.randn(N, F, device=dev)
W1 = torch.randn(H1, 2 * F, device=dev) * 0.05; W2 = torch.randn(H2, H1, device=dev) * 0.05; b2 = torch.randn(H2, device=dev) * 0.1
Wpq = torch.zeros(2 * H1p, F, device=dev); Wpq[:H1] = W1[:, :F] - W1[:, F:]; Wpq[H1p:H1p + H1] = W1[:, F:]
PQ = ops.linear_fwd(mode, [(x, F)], ops.pack_weight(Wpq, [F], dt, ops.gemm_kunit(mode)), 2 * H1p, out_lowp=True)
W2p, W2Tp = ops.pack_weight(W2, [H1], dt), ops.pack_weight(W2.t().contiguous(), [H2], dt)
gout = torch.randn(N, H2, device=dev).to(dt)
res = {}
for tag, flag in (("v2", "0"), ("v1", "1")):
    os.environ["GN_DISABLE_V2"] = flag
    out, saved = ops.edgeconv_fwd(mode, g, PQ, H1p, W2p, b2, H2)
    dW2, db2 = ops.edgeconv_dw2(mode, g, PQ, H1p, H1, H2, gout, saved)
    dPQ = torch.zeros(N, 2 * H1p, dtype=dt, device=dev); dpre = torch.zeros(g.rows, H1p, dtype=dt, device=dev)
    ops.edgeconv_bwd(mode, g, PQ, H1p, H2, gout, saved, W2Tp, dpre, dPQ[:, :H1p])
    torch.cuda.synchronize()
    res[tag] = dict(out=out.float(), dpre=dpre.float(), dP=dPQ[:, :H1p].float(), dW2=dW2)
for k in ("out", "dpre", "dP", "dW2"):
    a, c = res["v2"][k], res["v1"][k]
    d = (a - c).abs()
    i = int(d.argmax()); r, col = divmod(i, a.shape[1])
    print(k, "max|d|", float(d.max()), "at", r, col, "v2", float(a[r, col]), "v1", float(c[r, col]), "max|v1|", float(c.abs().max()),
          "n(|d|>1e-2*max)", int((d > 1e-2 * c.abs().max()).sum()))
    if k == "dpre":
        S = g.S
        print("   row -> centre", r // S, "slot", r % S, "nbr", g.nbr[r // S].tolist() if r < N * S else "ovf")
