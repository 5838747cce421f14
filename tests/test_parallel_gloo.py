"""CPU, world_size 2 (gloo): the flat-gradient all-reduce and event sharding used for N > 1."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from graphnet_amd.parallel import FlatGradAllReduce, broadcast_parameters
    torch.manual_seed(100 + rank)                       # different init per rank ...
    net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.ReLU(), torch.nn.Linear(5, 1))
    broadcast_parameters(net)                           # ... made identical
    sync = FlatGradAllReduce(net.parameters())
    opt = torch.optim.Adam(net.parameters(), lr=1e-2)
    torch.manual_seed(7)
    data = torch.randn(8, 6)
    shard = data[rank::world]                           # disjoint event shards
    for _ in range(3):
        sync.zero_grad()
        loss = net(shard).pow(2).mean()
        loss.backward()
        sync()
        opt.step()
    flat = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    q.put((rank, flat.numpy().copy(), sync.flat.numpy().copy()))
    dist.destroy_process_group()


def test_flat_grad_allreduce_world2_matches_single_process():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    import numpy as np
    assert np.array_equal(res[0][1], res[1][1]), "post-step weights must be bitwise identical across ranks"
    assert np.array_equal(res[0][2], res[1][2])
    # single-process reference: mean of the two shard gradients each step
    torch.manual_seed(100)
    net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.ReLU(), torch.nn.Linear(5, 1))
    opt = torch.optim.Adam(net.parameters(), lr=1e-2)
    torch.manual_seed(7)
    data = torch.randn(8, 6)
    for _ in range(3):
        opt.zero_grad()
        loss = 0.5 * (net(data[0::2]).pow(2).mean() + net(data[1::2]).pow(2).mean())
        loss.backward()
        opt.step()
    flat = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    assert torch.allclose(flat, torch.from_numpy(res[0][1]), rtol=1e-5, atol=1e-6)


def test_shard_events_by_pulses_balances_load():
    from graphnet_amd.parallel import shard_events_by_pulses
    import numpy as np
    n = np.random.default_rng(0).integers(8, 2000, size=257)
    shards = shard_events_by_pulses(n, 8)
    assert sorted(i for s in shards for i in s) == list(range(257))
    loads = [int(n[s].sum()) for s in shards]
    assert max(loads) - min(loads) <= int(n.max())
