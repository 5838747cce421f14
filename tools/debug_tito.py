import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphnet_amd import ops
from graphnet_amd.synthetic import synthetic_icecube86_batch
DEV = "cuda"; mode = ops.MODE_BF16
k = int(sys.argv[1]) if len(sys.argv) > 1 else 8
b = synthetic_icecube86_batch(20, seed=4).to(DEV)
N = int(b.x.shape[0])
ptr32, batch32 = b.ptr.to(torch.int32), b.batch.to(torch.int32)
g = ops.exact_table(ops.knn_graph(b.x, [0, 1, 2], batch32, ptr32, k))
S = g.S
H1 = H1p = d = 256
torch.manual_seed(11)
PQ16 = (torch.randn(N, 2 * H1p, device=DEV) * 0.7).bfloat16()
W2 = torch.randn(d, H1, device=DEV) * 0.06
b2 = torch.randn(d, device=DEV) * 0.2
gout = torch.randn(N, d, device=DEV)
out16, saved = ops.edgeconv_max_fwd(g, PQ16, H1p, ops.pack_weight(W2, [H1], torch.bfloat16), b2, d)
gmax, _, _ = ops.rownorm_act_bwd(gout, out16, d, "leaky_relu", cpad=d, lowp="only")
dW2_f, db2_f = ops.edgeconv_max_dw2(g, PQ16, H1p, H1, d, gmax, saved)
dPQ_f = torch.zeros((N, 2 * H1p), dtype=torch.bfloat16, device=DEV)
dpre = torch.zeros((g.rows, H1p), dtype=torch.bfloat16, device=DEV)
ops.edgeconv_max_bwd(g, H1p, d, gmax, saved, ops.pack_weight(W2.t().contiguous(), [d], torch.bfloat16), dpre, dPQ_f[:, :H1p])
ic, jc = ops.edge_rows(g)
a1 = ops.edge_gather_pre(PQ16.float(), H1p, ic, jc, act="leaky_relu", lowp=True)
z2 = ops.linear_fwd(mode, [(a1, H1p)], ops.pack_weight(W2, [H1], torch.bfloat16, ops.gemm_kunit(mode)), d, bias=b2, out_cols=d)
conv, aux = ops.slot_reduce(z2, d, g, "max", post_act="leaky_relu")
dz2, _, _ = ops.rownorm_act_bwd(gout, z2, d, "leaky_relu", valid=jc, gidx=ic, argrow=aux[1], cpad=d, lowp="only")
da1 = ops.linear_fwd(mode, [(dz2, d)], ops.pack_weight(W2.t().contiguous(), [d], torch.bfloat16, ops.gemm_kunit(mode)), H1, out_cols=H1p)
dpre_u, _, _ = ops.rownorm_act_bwd(da1, a1, H1, "leaky_relu", valid=jc, cpad=H1p)
torch.cuda.synchronize()
R = N * S
A, B = dpre[:R].float(), dpre_u[:R]
err = (A - B).abs()
print("S", S, "rows", R, "max|ref|", float(B.abs().max()), "max err", float(err.max()))
colerr = err.max(0).values.cpu()
print("col-block max err:", [round(float(colerr[i*32:(i+1)*32].max()), 3) for i in range(8)])
sloterr = err.reshape(N, S, H1p).amax(dim=(0, 2)).cpu()
print("slot max err:", [round(float(v), 3) for v in sloterr])
# is the fused dpre consistent with slope 1 / 0.01 swapped somewhere?
pos = (a1[:R].float() > 0)
ratio = torch.where(B.abs() > 1e-3, A / B, torch.ones_like(A))
print("ratio stats where h>0:", float(ratio[pos].median()), " where h<=0:", float(ratio[~pos & (B.abs() > 1e-3)].median()))
bad = (err > 0.05 * B.abs().max()).nonzero()
print("bad count", len(bad), "of", A.numel(), "first:", bad[:8].tolist())
for r_, c_ in bad[:6].tolist():
    print(" row", r_, "slot", r_ % S, "col", c_, "fused", float(A[r_, c_]), "ref", float(B[r_, c_]), "da1", float(da1[r_, c_]), "a1", float(a1[r_, c_]))
