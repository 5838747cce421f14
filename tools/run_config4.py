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
print({k: round(ms / steps, 3) for k, (n_, ms) in ops.timer_summary().items()})
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
