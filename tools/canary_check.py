#!/usr/bin/env python3
"""Out-of-bounds hunt without faulting the GPU: every tensor the step allocates gets a guard band behind it
(0xA5 bytes); after each eager training step all bands are verified.  A kernel that writes past its output shows
up as a damaged band, with the shape of the tensor in front of it."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from graphnet_amd.parallel import FlatGradAllReduce
from graphnet_amd.synthetic import synthetic_icecube86_batch

GUARD = 4096
_real_empty, _real_zeros = torch.empty, torch.zeros
_guards = []

def _guarded(alloc, zero):
    def f(*size, dtype=None, device=None, **kw):
        if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)):
            size = tuple(size[0])
        dev = torch.device(device) if device is not None else torch.device("cpu")
        if dev.type != "cuda" or kw:
            if device is not None: kw["device"] = device
            if dtype is not None: kw["dtype"] = dtype
            return alloc(*size, **kw)
        dtype_ = dtype or torch.float32
        numel = 1
        for s in size: numel *= int(s)
        nbytes = numel * _real_empty((), dtype=dtype_).element_size()
        nbytes_al = (nbytes + 15) // 16 * 16
        raw = _real_empty(nbytes_al + GUARD, dtype=torch.uint8, device=dev)
        raw[nbytes:].fill_(0xA5)
        t = raw[:nbytes].view(dtype_).view(*size) if numel else _real_empty(*size, dtype=dtype_, device=dev)
        if zero and numel: t.zero_()
        if numel: _guards.append((raw, nbytes, tuple(size), str(dtype_)))
        return t
    return f

def check(tag):
    torch.cuda.synchronize()
    bad = 0
    for raw, nbytes, shape, dt in _guards:
        tail = raw[nbytes:]
        if not bool((tail == 0xA5).all()):
            idx = int((tail != 0xA5).nonzero()[0])
            print(f"[{tag}] GUARD DAMAGED behind tensor shape={shape} {dt}: first bad byte at +{idx}, "
                  f"{int((tail != 0xA5).sum())} bytes changed")
            bad += 1
    n = len(_guards)
    _guards.clear()
    return bad, n

m = bench.build_model("bf16").to("cuda")
opt = torch.optim.Adam(m.parameters(), lr=1e-3, eps=1e-3, fused=True)
sync = FlatGradAllReduce(m.parameters())
b = synthetic_icecube86_batch(int(sys.argv[1]) if len(sys.argv) > 1 else 1024, seed=20241016).to("cuda")
def step():
    sync.zero_grad(); loss = m.shared_step(b); loss.backward(); sync(); opt.step(); return loss
for _ in range(3): step()
torch.cuda.synchronize()
torch.empty, torch.zeros = _guarded(_real_empty, False), _guarded(_real_zeros, True)
total_bad = 0
for i in range(int(sys.argv[2]) if len(sys.argv) > 2 else 30):
    loss = step()
    bad, n = check(f"step {i}")
    total_bad += bad
    if i % 10 == 0: print(f"step {i}: {n} guarded tensors, loss {float(loss):.4f}, damaged {bad}", flush=True)
print("TOTAL damaged guard bands:", total_bad)
