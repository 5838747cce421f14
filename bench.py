#!/usr/bin/env python3
"""bench.py — events/s for DynEdge fwd+bwd(+Adam, +gradient all-reduce) on synthetic IceCube-86
pulse graphs (BASELINE.json metric; workload = configs[1]), one process per GPU.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one pass of the hot path over one batch of B events already resident in HBM:
layer-1 k-NN build, global variables, 4x(P|Q GEMM, fused EdgeConv, re-kNN), post MLP, pooling,
readout, energy head, LogCosh, full backward, one flat RCCL all-reduce (N > 1), Adam step.
Rank 0 prints ONE JSON line (contract in the task statement) with `roofline` and `cpu_baseline`.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0      # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_F32_TFLOPS = 157.3


def build_model(dtype: str):
    import graphnet_amd as g
    torch.manual_seed(20241016)                       # identical initial weights on every rank
    m = g.StandardModel(
        graph_definition=g.KNNGraph(g.IceCube86()),
        backbone=g.DynEdge(7, global_pooling_schemes=["min", "max", "mean", "sum"]),
        tasks=[g.EnergyReconstruction(hidden_size=128, loss_function=g.LogCoshLoss(),
                                      transform_prediction_and_target=torch.log10)],
        optimizer_kwargs={"lr": 1e-3, "eps": 1e-3},
    )
    m.backbone.set_backend(dtype=dtype)
    return m


def measured_peaks(dev) -> dict:
    """Yardsticks measured in this run: a large library bf16 GEMM (hipBLASLt through
    torch.matmul) and a device-to-device copy.  Not used as denominators: `roofline.peak` stays the
    MI355X_MICROARCH.md figure (2.5 PFLOP/s dense bf16, 8 TB/s HBM)."""
    n = 8192
    a = torch.randn(n, n, device=dev, dtype=torch.bfloat16)
    b = torch.randn(n, n, device=dev, dtype=torch.bfloat16)
    torch.matmul(a, b)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(10):
        torch.matmul(a, b)
    e1.record()
    torch.cuda.synchronize()
    gemm = 10 * 2.0 * n ** 3 / (e0.elapsed_time(e1) * 1e-3) / 1e12
    src = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
    dst = torch.empty_like(src)
    dst.copy_(src)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(10):
        dst.copy_(src)
    e1.record()
    torch.cuda.synchronize()
    copy = 10 * 2.0 * (1 << 30) / (e0.elapsed_time(e1) * 1e-3) / 1e9          # read + write bytes
    return {"library_bf16_gemm_tflops": gemm, "hbm_copy_gbs": copy,
            "note": "8192^3 bf16 torch.matmul; 1 GiB device copy (read+write bytes)"}


def algorithmic_flops(n_nodes: int, n_edges: int, n_events: int, F0: int = 19, n_pool: int = 4) -> float:
    """Reference-formulation forward GEMM FLOPs, MAC = 2 FLOP."""
    return 2.0 * (n_edges * (2 * F0 * 128 + 128 * 256) + 3 * n_edges * (512 * 336 + 336 * 256)
                  + n_nodes * ((F0 + 1024) * 336 + 336 * 256) + n_events * (n_pool * 256 * 128 + 128))


def cpu_baseline(events: int, steps: int):
    """The oracle (plain-torch restatement of the PyG formulation) timed on the host cores."""
    from graphnet_amd.synthetic import synthetic_icecube86_batch
    from oracle import dynedge_oracle as orc
    try:
        cores = len(os.sched_getaffinity(0))          # the cores this job may actually use
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, int(os.environ.get("GN_CPU_BASELINE_THREADS", "32"))))
    torch.set_num_threads(cores)
    torch.manual_seed(20241016)
    m = orc.StandardModelOracle(7, global_pooling_schemes=["min", "max", "mean", "sum"], literal_distribute=True)
    opt = torch.optim.Adam(m.parameters(), lr=1e-3, eps=1e-3)
    b = synthetic_icecube86_batch(events, seed=20241016)

    def step():
        ei = orc.knn_graph(b.x, 8, b.batch, [0, 1, 2])          # loader-side k-NN is part of the path
        loss = m.loss(b.x, ei, b.batch, b.n_pulses, b.energy)
        opt.zero_grad()
        loss.backward()
        opt.step()

    step()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    dt = time.perf_counter() - t0
    return {"value": events * steps / dt, "unit": "events/s", "cores": cores, "kind": "port",
            "sample": f"{steps} fwd+bwd+Adam steps of {events} synthetic IceCube-86 events (fp32, torch CPU, "
                      f"{cores} threads; PyG not installed: in-repo restatement of the PyG formulation)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--events", type=int, default=4096,
                    help="events per GPU per step (SURVEY 8d: B in {256, 1024, 4096}, report best and B=1024: "
                         "the headline is B=4096, B=1024 is reported in other_batch_size)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", action="store_true", help="replay the step as a hipGraph (experimental) instead of eager launches")
    ap.add_argument("--profile-steps", type=int, default=10, help="eager steps with per-op HIP events (roofline)")
    ap.add_argument("--extra-events", type=int, default=1024,
                    help="also report events/s at this many events per GPU (SURVEY 8d: B in {256,1024,4096}); 0 = skip")
    ap.add_argument("--overlap", action="store_true",
                    help="graph building on a second HIP stream (opt-in: dispatcher-dependent, see DESIGN.md)")
    ap.add_argument("--cpu-events", type=int, default=64)
    ap.add_argument("--cpu-steps", type=int, default=3)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # Rehearsal knobs (one-GPU box only, never set by the driver): GN_BENCH_REHEARSE=1 puts every rank on
    # cuda:0 and exchanges gradients over gloo, to exercise the multi-rank control flow without a second GPU.
    rehearse = os.environ.get("GN_BENCH_REHEARSE") == "1"
    if rehearse:
        local = 0
    torch.cuda.set_device(local)                         # before the process group: RCCL binds to the current device
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from graphnet_amd import _lib, ops
    from graphnet_amd.parallel import FlatGradAllReduce, broadcast_parameters
    from graphnet_amd.synthetic import synthetic_icecube86_batch
    _lib.lib()                                           # fail loudly if the HIP library is missing

    model = build_model(args.dtype).to(dev)
    if args.overlap:
        model.backbone.set_backend(overlap=True)
    broadcast_parameters(model)
    sync = FlatGradAllReduce(model.parameters())
    # Adam(lr 1e-3, eps 1e-3) as in the reference example (easy_model.py:215-235).  Eager launches use torch's
    # fused multi-tensor implementation (one launch instead of ~10 per step); the experimental hipGraph path
    # needs the capturable variant.
    if args.graph:
        opt = torch.optim.Adam(model.parameters(), lr=1e-3, eps=1e-3, capturable=True)
    else:
        try:
            opt = torch.optim.Adam(model.parameters(), lr=1e-3, eps=1e-3, fused=True)
        except (RuntimeError, TypeError):
            opt = torch.optim.Adam(model.parameters(), lr=1e-3, eps=1e-3)
    batch = synthetic_icecube86_batch(args.events, seed=20241016 + rank).to(dev)   # disjoint shards (weak scaling)
    n_nodes = int(batch.x.shape[0])

    def eager_step():
        sync.zero_grad()
        loss = model.shared_step(batch)
        loss.backward()
        sync()
        opt.step()
        return loss

    launch = "eager"
    step = eager_step
    if args.graph:
        try:
            from graphnet_amd.graphed import GraphedTrainStep
            graphed = GraphedTrainStep(model, opt, sync)
            graphed(batch)                                    # capture (+ its own eager warm-up)
            step = lambda: graphed(batch)
            launch = "hipgraph"
        except Exception as exc:                              # pragma: no cover - reported, never silent
            print(f"[bench] hipGraph capture failed, running eager: {exc!r}", file=sys.stderr)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # Untimed ramp-up before the W warm-up steps: the first ~0.3 s of sustained load run at ~25 % lower
    # throughput (GPU clock / power-state ramp and allocator growth: 10.0 ms/step with 5 warm-up steps vs
    # 7.5 ms/step with 40), so a short warm-up would time the ramp instead of the steady state.
    # A FIXED number of steps: every step contains the gradient all-reduce, so all ranks must run the same count.
    for i in range(80):
        step()
        if i % 10 == 9:
            torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    fence()
    dt = time.perf_counter() - t0
    # per-kernel durations: HIP events on the launch stream around every C-ABI op.  A graph replay has
    # no per-kernel events, so the same kernels are timed in a few eager steps right after the timed
    # region (same process, same buffers); rocprofv3 --kernel-trace of this command must agree.
    fence()
    th = time.perf_counter()
    eager_step()                                              # host time to ENQUEUE one step (no sync)
    host_issue_ms = 1e3 * (time.perf_counter() - th)
    fence()
    ops.enable_timers(True)
    for _ in range(max(1, args.profile_steps)):
        eager_step()
    fence()
    timers = ops.timer_summary()
    detail = ops.timer_summary(detail=True)
    ops.enable_timers(False)
    prof_steps = max(1, args.profile_steps)
    # second batch size (same model, same step), reported beside the headline value
    extra = None
    if args.extra_events and args.extra_events != args.events:
        big = synthetic_icecube86_batch(args.extra_events, seed=20241016 + 1000 + rank).to(dev)
        saved_batch = batch

        def big_step():
            sync.zero_grad()
            loss_b = model.shared_step(big)
            loss_b.backward()
            sync()
            opt.step()
        for i in range(30):
            big_step()
            if i % 10 == 9:
                torch.cuda.synchronize()
        fence()
        tb = time.perf_counter()
        for _ in range(20):
            big_step()
        fence()
        tbig = torch.tensor([time.perf_counter() - tb], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(tbig, op=dist.ReduceOp.MAX)
        extra = {"events_per_gpu": args.extra_events, "pulses_per_gpu": int(big.x.shape[0]), "steps": 20,
                 "value": args.extra_events * world * 20 / float(tbig.item()), "unit": "events/s",
                 "ms_per_step": 1e3 * float(tbig.item()) / 20}
        del big
        batch = saved_batch
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    if rank == 0:
        total_events = args.events * world * args.steps
        # edges of the layer-1 graph (degree k or k+1) ~ edges of every layer
        with torch.no_grad():
            from graphnet_amd import ops as _o
            t = _o.knn_graph(batch.x, [0, 1, 2], batch.batch.to(torch.int32), batch.ptr.to(torch.int32), 8)
            n_edges = int((t.nbr >= 0).sum().item()) + int(t.ovf_cnt.item())
        flops_fwd = algorithmic_flops(n_nodes, n_edges, args.events)
        peak = PEAK_BF16_TFLOPS if args.dtype == "bf16" else PEAK_F32_TFLOPS
        # dominant op group (by HIP-event time on the launch stream) priced with ITS OWN algorithmic work:
        # MFMA groups in FLOPs (2*rows*K*N of the contraction it performs), HBM groups in bytes.
        E, N = float(n_edges), float(n_nodes)
        conv = [(19, 128, 256), (256, 336, 256), (256, 336, 256), (256, 336, 256)]       # (F_in, H1, H2)
        F0 = 19
        flops = {
            "edgeconv_fwd": sum(2 * E * h1 * h2 for _, h1, h2 in conv),
            "edgeconv_bwd": sum(2 * E * h1 * h2 for _, h1, h2 in conv),
            "edgeconv_dw2": sum(2 * E * h1 * h2 for _, h1, h2 in conv),
            # P|Q GEMMs, post MLP (2 layers), their input-gradient GEMMs, conv input gradients
            "linear_fwd": sum(2 * N * f * 2 * h1 for f, h1, _ in conv) + 2 * N * ((F0 + 1024) * 336 + 336 * 256)
                          + 2 * N * (336 * 256 + (F0 + 1024) * 336) + sum(2 * N * 2 * h1 * f for f, h1, _ in conv[1:]),
            "linear_wgrad": sum(2 * N * f * 2 * h1 for f, h1, _ in conv) + 2 * N * ((F0 + 1024) * 336 + 336 * 256),
        }
        act = 2 if args.dtype == "bf16" else 4     # bytes per stored activation element
        hbm_bytes = {   # algorithmic bytes per step of the HBM-bound groups
            "edgeconv_dq_gather": sum((E + N) * act * ((h1 + 31) // 32 * 32) for _, h1, _ in conv),
            "knn_graph": 4 * (N * 3 * 4 + N * 9 * 4),
            "colsum": sum(N * 4 * h1 for _, h1, _ in conv) + N * 4 * (336 + 256),
            "segment_pool_fwd": N * 256 * 4, "segment_pool_bwd": N * 256 * (4 + act),
        }
        dom = max(timers.items(), key=lambda kv: kv[1][1]) if timers else ("none", (1, 0.0))
        name, (launches, ms) = dom
        per_launch_ms = ms / max(launches, 1)
        lps = launches / prof_steps
        kernel_shape = None
        if name in flops:
            bound, unit = "mfma", "TFLOP/s"
            # the group mixes layer shapes (conv 1 is 128 wide, conv 2-4 are 336 wide): price the ONE kernel shape
            # that takes the most time with its own contraction, launch by launch - this is the duration a
            # rocprofv3 kernel trace reports for that kernel
            shapes = {k: v for k, v in detail.items() if k.startswith(name + "[")}
            if shapes:
                kernel_shape, (launches, ms) = max(shapes.items(), key=lambda kv: kv[1][1])
                h1p, h2 = (int(v) for v in kernel_shape[len(name) + 1:-1].split("x"))
                h1 = next(h for _, h, _ in conv if (h + 31) // 32 * 32 == h1p)
                per_launch_ms = ms / max(launches, 1)
                lps = launches / prof_steps
                achieved = 2 * E * h1 * h2 / (per_launch_ms * 1e-3) / 1e12
            else:
                achieved = flops[name] / lps / (per_launch_ms * 1e-3) / 1e12
        else:
            bound, unit, peak = "hbm", "GB/s", 8000.0
            achieved = hbm_bytes.get(name, 0.0) / lps / (per_launch_ms * 1e-3) / 1e9
        # measured HBM traffic per launch of that group (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate
        # passes, FETCH_SIZE doubled as the gfx950 guide prescribes), recorded in profiles/r01_traffic.json
        traffic = None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic.json")))
            if tj.get(name, {}).get("events_per_gpu", 1024) == args.events:      # measured on this workload only
                traffic = tj.get(name, {}).get("bytes_per_launch")
        except Exception:
            pass
        out = {
            "metric": "events/sec DynEdge fwd+bwd, IceCube-86 k=8",
            "value": total_events / dt, "unit": "events/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "configs[1]: DynEdge energy regression, synthetic IceCube-86 pulses "
                                   "(~150/event, 7 features), k=8, fwd+bwd+Adam",
                       "events_per_gpu": args.events, "pulses_per_gpu": n_nodes, "edges_per_layer": n_edges,
                       "parallelism": f"dp{world} (event shards, one flat RCCL all-reduce)"},
            "roofline": {"bound": bound, "kernel": kernel_shape or name, "achieved": achieved, "peak": peak, "unit": unit,
                         "frac": achieved / peak, "traffic": traffic,
                         "launch_ms": per_launch_ms, "launches_per_step": lps},
            "group_tflops": {k: flops[k] / (timers[k][1] / prof_steps * 1e-3) / 1e12 for k in flops if k in timers},
            "path_roofline": {"algorithmic_tflop_per_step": 3.0 * flops_fwd * world / 1e12,
                              "achieved_tflops": 3.0 * flops_fwd * world * args.steps / dt / 1e12,
                              "frac_of_peak": 3.0 * flops_fwd * args.steps / dt / 1e12 / peak},
            "phase_ms_per_step": {k: v[1] / prof_steps for k, v in sorted(timers.items(), key=lambda kv: -kv[1][1])},
            "launch": launch, "host_issue_ms": host_issue_ms,
            "final_loss": float(loss.detach()),
            "measured_peaks": measured_peaks(dev),
            "other_batch_size": extra,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.cpu_events, args.cpu_steps)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()                  # rank 0 is still measuring its yardsticks: leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
