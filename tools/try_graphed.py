#!/usr/bin/env python3
"""Experiment: hipGraph replay of the training step vs the eager loop (same batch, same seeds)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import graphnet_amd as g
from graphnet_amd.graphed import GraphedTrainStep
from graphnet_amd.parallel import FlatGradAllReduce
from graphnet_amd.synthetic import synthetic_icecube86_batch
DEV = "cuda"
n_ev = int(sys.argv[1]) if len(sys.argv) > 1 else 24
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
b = synthetic_icecube86_batch(n_ev, seed=15).to(DEV)
def make():
    torch.manual_seed(3)
    m = g.StandardModel(graph_definition=g.KNNGraph(g.IceCube86()),
                        backbone=g.DynEdge(7, global_pooling_schemes=["min", "max", "mean", "sum"]),
                        tasks=[g.EnergyReconstruction(hidden_size=128, loss_function=g.LogCoshLoss(),
                                                      transform_prediction_and_target=torch.log10)]).to(DEV)
    opt = torch.optim.Adam(m.parameters(), lr=1e-3, eps=1e-3, capturable=True)
    return m, opt, FlatGradAllReduce(m.parameters())
m1, o1, s1 = make()
l1 = []
for _ in range(steps):
    s1.zero_grad(); loss = m1.shared_step(b); loss.backward(); s1(); o1.step(); l1.append(float(loss))
m2, o2, s2 = make()
W = 2                                      # eager warm-up steps inside the first call (real optimizer steps:
                                           # they also create the Adam state BEFORE the capture)
step = GraphedTrainStep(m2, o2, s2, warmup=W)
if len(sys.argv) > 3 and sys.argv[3] == "nosync":      # replays enqueued back to back, no host sync in between
    lt = [step(b).detach().clone() for _ in range(steps - W)]
    torch.cuda.synchronize()
    l2 = [float(t) for t in lt]
else:
    l2 = [float(step(b)) for _ in range(steps - W)]
torch.cuda.synchronize()
print("eager  ", l1[W:]); print("graphed", l2)
if len(sys.argv) > 3 and sys.argv[3] == "eager_after":
    for _ in range(2):                     # eager steps on the graphed model after the capture (what bench.py does)
        s2.zero_grad(); loss = m2.shared_step(b); loss.backward(); s2(); o2.step()
    torch.cuda.synchronize()
    print("eager steps after the graph: ok", float(loss))
print("weights equal:", all(torch.equal(p, q) for p, q in zip(m1.parameters(), m2.parameters())))
