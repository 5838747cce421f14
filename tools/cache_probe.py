#!/usr/bin/env python3
"""Does a producer->consumer pair run faster when the intermediate buffer is small enough to stay in the 256 MB
Infinity Cache?  write (fill) + read (sum) of a buffer of S MB, repeated so that the total bytes are constant."""
import torch
dev = "cuda"
total = 8 << 30
for mb in (32, 64, 128, 192, 256, 384, 512, 1024, 4096):
    n = (mb << 20) // 2
    x = torch.empty(n, dtype=torch.bfloat16, device=dev)
    src = torch.randn(n, dtype=torch.float32, device=dev).to(torch.bfloat16)
    out = torch.empty(n, dtype=torch.bfloat16, device=dev)
    reps = max(2, total // (mb << 20))
    for _ in range(3):
        x.copy_(src); out.copy_(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        x.mul_(1.0001)          # read + write x (stands for: producer writes, consumer reads the same buffer)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    print(f"buffer {mb:5d} MB: in-place rmw {reps * 2 * (mb << 20) / ms / 1e9:.2f} TB/s")
