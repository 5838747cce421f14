"""GPU parity tests of the DynEdgeTITO path against ``oracle/tito_oracle.py``, whose encoder
layer is pinned against torch's own TransformerEncoder in ``tests/test_oracle_pins.py``.

fp32 mode within 1e-4 relative (of the tensor's max magnitude); bf16 operand mode within 3e-2 (stated per assert).
"""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda"


def rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def _ragged_attention_reference(qkv: torch.Tensor, ptr, H: int) -> torch.Tensor:
    """fp64 restatement of the attention core of ``tito_oracle.self_attention_ragged`` (no projections)."""
    N, d3 = qkv.shape
    d = d3 // 3
    dh = d // H
    q, k, v = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]
    outs = []
    for e in range(len(ptr) - 1):
        a, b = int(ptr[e]), int(ptr[e + 1])
        n = b - a
        qe = q[a:b].reshape(n, H, dh).transpose(0, 1)
        ke = k[a:b].reshape(n, H, dh).transpose(0, 1)
        ve = v[a:b].reshape(n, H, dh).transpose(0, 1)
        p = torch.softmax(qe @ ke.transpose(1, 2) / math.sqrt(dh), dim=-1)
        outs.append((p @ ve).transpose(0, 1).reshape(n, d))
    return torch.cat(outs, 0)


@pytest.mark.parametrize("H,dh", [(8, 32), (4, 16), (8, 8), (2, 64)])
def test_ragged_attention_forward_backward(H, dh):
    """Events of 1, 63, 64, 65, 200 and 1300 pulses (tile tails, single-key softmax, a multi-tile event)."""
    from graphnet_amd import ops
    torch.manual_seed(H * 100 + dh)
    sizes = [1, 63, 64, 65, 200, 1300, 2]
    ptr = [0]
    for s in sizes:
        ptr.append(ptr[-1] + s)
    N, d = ptr[-1], H * dh
    qkv = torch.randn(N, 3 * d, dtype=torch.float64) * 1.5
    qkv.requires_grad_(True)
    want = _ragged_attention_reference(qkv, ptr, H)
    w = torch.randn(N, d, dtype=torch.float64)
    (want * w).sum().backward()
    ptr_d = torch.tensor(ptr, dtype=torch.int32, device=DEV)
    plan = ops.knn_plan(ptr_d, N)
    x = qkv.detach().float().to(DEV)
    out, lse2 = ops.attention_fwd(x, H, ptr_d, plan)
    assert rel_err(out, want.detach()) < 1e-5          # fp32 flash accumulation against fp64
    dqkv = ops.attention_bwd(x, H, ptr_d, plan, out, lse2, w.float().to(DEV))
    for name, sl in (("dq", slice(0, d)), ("dk", slice(d, 2 * d)), ("dv", slice(2 * d, 3 * d))):
        assert rel_err(dqkv[:, sl], qkv.grad[:, sl]) < 2e-5, name


def test_ragged_attention_rejects_unsupported_head_width():
    from graphnet_amd import ops
    ptr_d = torch.tensor([0, 4], dtype=torch.int32, device=DEV)
    plan = ops.knn_plan(ptr_d, 4)
    with pytest.raises(RuntimeError, match="head width"):
        ops.attention_fwd(torch.zeros(4, 3 * 24, device=DEV), 2, ptr_d, plan)
