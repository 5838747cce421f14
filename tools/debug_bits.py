import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from graphnet_amd import ops
from oracle.tito_oracle import keep_mask
DEV = "cuda"
H, dh = 8, 32
torch.manual_seed(1)
sizes = [1, 63, 64, 65, 200, 1300, 2]
ptr = [0]
for s in sizes: ptr.append(ptr[-1] + s)
N, d = ptr[-1], H * dh
xb = (torch.randn(N, 3 * d) * 1.5).to(torch.bfloat16).to(DEV)
w = torch.randn(N, d).to(torch.bfloat16).to(DEV)
drop = (987654329, ops.drop_thresh(0.25))
ptr_d = torch.tensor(ptr, dtype=torch.int32, device=DEV)
plan = ops.attention_plan(ptr_d)
out1, lse1 = ops.attention_fwd(xb, H, ptr_d, plan, drop=drop)
d1 = ops.attention_bwd(xb, H, ptr_d, plan, out1, lse1, w, drop=drop)
lay = ops.attention_drop_layout(ptr_d)
print("evoff", lay[0].tolist(), lay[1])
out2, lse2, bits = ops.attention_fwd_saved(xb, H, ptr_d, plan, drop, lay)
d2 = ops.attention_bwd_saved(xb, H, ptr_d, plan, out2, lse2, w, drop[1], bits, lay)
print("fwd equal", torch.equal(out1, out2))
for name, sl in (("dq", slice(0, d)), ("dk", slice(d, 2 * d)), ("dv", slice(2 * d, 3 * d))):
    bad = (d1[:, sl] != d2[:, sl]).any(dim=1).nonzero().flatten().tolist()
    print(name, "rows differing:", len(bad), bad[:20])
evoff = lay[0].cpu().numpy()
br = bits[0].cpu().numpy().view(np.uint32).reshape(H, lay[1])
bc = bits[1].cpu().numpy().view(np.uint32).reshape(H, lay[1])
for e in range(len(sizes)):
    n, W = sizes[e], (sizes[e] + 31) // 32
    rows = np.arange(ptr[e], ptr[e + 1], dtype=np.uint32)
    for head in (0, 7):
        cols = (rows * np.uint32(H) + np.uint32(head)).astype(np.uint32)
        keep = keep_mask(drop[0], rows[:, None], cols[None, :], drop[1])
        badr = badc = 0
        for qb in range(W):
            for kb in range(W):
                tr = br[head, (evoff[e] + qb * W + kb) * 32:][:32]
                tc = bc[head, (evoff[e] + kb * W + qb) * 32:][:32]
                for c in range(32):
                    q = 32 * qb + c
                    if q < n:
                        ks = np.arange(32 * kb, min(32 * kb + 32, n))
                        got = (tr[c] >> (ks - 32 * kb).astype(np.uint32)) & 1
                        badr += int((got.astype(bool) != keep[q, ks]).sum())
                    k = 32 * kb + c
                    if k < n:
                        qs = np.arange(32 * qb, min(32 * qb + 32, n))
                        got = (tc[c] >> (qs - 32 * qb).astype(np.uint32)) & 1
                        badc += int((got.astype(bool) != keep[qs, k]).sum())
        print("event", e, "n", n, "head", head, "bad row bits", badr, "bad col bits", badc)
