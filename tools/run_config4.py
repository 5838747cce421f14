#!/usr/bin/env python3
"""BASELINE configs[3]: DynEdgeTITO (EdgeConvTito + transformer encoder per layer, default sizes: 4 x (256, 256),
8 heads, FFN 2048), mixed 50-3000 pulses/event (log-uniform), k=8 static edges.  Times fwd+bwd+Adam.
usage: run_config4.py [B] [fp32|bf16] [steps] [--dropout P] [--cpu-baseline]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import graphnet_amd as g
from graphnet_amd import ops
from graphnet_amd.synthetic import synthetic_icecube86_batch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dtype = sys.argv[2] if len(sys.argv) > 2 else "bf16"
steps = int(sys.argv[3]) if len(sys.argv) > 3 and sys.argv[3].isdigit() else 20
p_drop = float(sys.argv[sys.argv.index("--dropout") + 1]) if "--dropout" in sys.argv else 0.1   # torch / reference default
torch.manual_seed(0)
b = synthetic_icecube86_batch(B, seed=5, count_range=(50, 3000)).to("cuda")
m = g.StandardModel(
    graph_definition=g.KNNGraph(g.IceCube86(), nb_nearest_neighbours=8),
    backbone=g.DynEdgeTITO(7, global_pooling_schemes=["max"], dropout=p_drop),
    tasks=[g.EnergyReconstruction(hidden_size=128, loss_function=g.LogCoshLoss(),
                                  transform_prediction_and_target=torch.log10)],
    optimizer_kwargs={"lr": 1e-4, "eps": 1e-3},
).to("cuda")
m.backbone.set_backend(dtype=dtype)
opt = torch.optim.Adam(m.parameters(), lr=1e-4, eps=1e-3, fused=True)
def step():
    opt.zero_grad(set_to_none=True)
    loss = m.shared_step(b)
    loss.backward()
    opt.step()
    return loss
for _ in range(30): l = step()          # clock ramp-up + allocator growth
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps): l = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
ops.enable_timers(True)                 # per-op HIP events (adds host work: not part of the timed steps above)
for _ in range(steps): step()
torch.cuda.synchronize()
n = b.n_pulses.double()
print(f"config4 dropout={p_drop} B={B} N={b.x.shape[0]} (pulses/event {int(n.min())}..{int(n.max())}, sum n^2 = {float((n*n).sum()):.3g}) {dtype}: "
      f"{1e3*dt:.1f} ms/step  {B/dt:.1f} events/s  {b.x.shape[0]/dt/1e6:.2f} Mpulses/s  loss {float(l):.4f}")
summary = ops.timer_summary()
print({k: round(ms / steps, 3) for k, (n_, ms) in summary.items()})
print("per kernel shape (launches per step, ms per launch):",
      {k: (n_ / steps, round(ms / max(n_, 1), 4)) for k, (n_, ms) in sorted(ops.timer_summary(detail=True).items(), key=lambda kv: -kv[1][1])})
# one JSON line in the shape of bench.py's (BASELINE configs[3] has no bench.py leg): throughput + the roofline of the
# dominant kernel family, the ragged self attention, priced at 4 * sum(n_i^2) * d_model FLOP per layer forward
# (Q K^T and P V, 2 FLOP per MAC) and 2.5x that backward (dQ, dK, dV, dP recomputation), against the dense bf16 peak
import json
sn2, dmod, layers = float((n * n).sum()), 256, 4
f_fwd, f_bwd = 4.0 * sn2 * dmod * layers, 10.0 * sn2 * dmod * layers
t_fwd = summary.get("attention_fwd", (0, 0.0))[1] / steps * 1e-3
t_bwd = summary.get("attention_bwd", (0, 0.0))[1] / steps * 1e-3
print(json.dumps({
    "metric": "events/sec DynEdgeTITO fwd+bwd+Adam, mixed 50-3000 pulses/event, k=8 static edges", "value": B / dt,
    "unit": "events/s", "n_gpus": 1, "steps": steps, "ms_per_step": 1e3 * dt, "dtype": dtype, "data": "synthetic",
    "config": {"workload": "configs[3]: DynEdgeTITO 4 x (256, 256), 8 heads, FFN 2048, IceCube-86 geometry, "
                           "pulses per event log-uniform 50..3000", "events_per_gpu": B, "pulses_per_gpu": int(b.x.shape[0]),
               "sum_n2": sn2, "dropout": p_drop},
    "roofline": {"bound": "mfma", "kernel": "attn_fwd_mfma + attn_bwd_dq_mfma + attn_bwd_dkv_mfma (4 layers)",
                 "achieved": (f_fwd + f_bwd) / max(t_fwd + t_bwd, 1e-9) / 1e12, "peak": 2500.0, "unit": "TFLOP/s",
                 "frac": (f_fwd + f_bwd) / max(t_fwd + t_bwd, 1e-9) / 1e12 / 2500.0,
                 "fwd_tflops": f_fwd / max(t_fwd, 1e-9) / 1e12, "bwd_tflops": f_bwd / max(t_bwd, 1e-9) / 1e12,
                 "launch_ms_fwd": 1e3 * t_fwd / layers, "launch_ms_bwd": 1e3 * t_bwd / layers, "traffic": None},
    "phase_ms_per_step": {k: ms / steps for k, (n_, ms) in sorted(summary.items(), key=lambda kv: -kv[1][1])}}))
if "--cpu-baseline" in sys.argv:     # the oracle (CPU restatement, test infrastructure) timed on a bounded sample
    from oracle import dynedge_oracle, tito_oracle
    nb = 8
    bc = synthetic_icecube86_batch(nb, seed=5, count_range=(50, 3000))
    ref = tito_oracle.DynEdgeTITOOracle(7, global_pooling_schemes=["max"])
    ei = dynedge_oracle.knn_graph(bc.x, 8, bc.batch, [0, 1, 2])
    t0 = time.perf_counter()
    y = ref(bc.x, ei, bc.batch, bc.n_pulses)
    y.sum().backward()
    dtc = time.perf_counter() - t0
    print(f"cpu oracle (torch CPU, {torch.get_num_threads()} threads): {nb} events / {bc.x.shape[0]} pulses fwd+bwd in {dtc:.2f} s "
          f"= {nb/dtc:.2f} events/s, {bc.x.shape[0]/dtc/1e3:.1f} kpulses/s")
