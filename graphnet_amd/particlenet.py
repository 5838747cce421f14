"""graphnet_amd/particlenet.py — ``ParticleNeT`` on the HIP kernels.

Mirrors ``/root/reference/src/graphnet/models/gnn/particlenet.py`` (ctor l.27-170, layers l.172-213, forward
l.228-244): DynEdgeConv blocks whose edge MLP is ``[Linear, BatchNorm1d, act] x L`` with **mean** aggregation and a
k-NN re-clustering after every block, global pooling, read-out MLP with dropout.  Same constructor, same sub-module
names (state-dict keys ``_conv_layers.{l}.nn.{idx}.*`` incl. the BatchNorm running statistics, ``_readout.*``).

Kernels: the first Linear of every block through the per-node split ``W [x_i | x_j - x_i] = (Wa - Wb) x_i + Wb x_j``
(one MFMA GEMM over nodes + a gather), the other Linears as MFMA GEMMs over edge rows, BatchNorm1d as masked
column statistics over the edge rows (empty slots take no part, fixed-order reductions) fused with the activation
(``csrc/generic.hip: bn_*``), mean aggregation and its backward on the slot kernels, k-NN on ``knn_kernel``.
"""
from __future__ import annotations

from typing import Any, List, Optional, Sequence, Tuple, Union

import torch
from torch import Tensor

from . import ops
from .gnn import GNN, _ACT_NAMES, _ksegs, _maybe, _subset_cols
from .tito import _PostPoolFunction, _wt


def _n_valid(g: ops.NeighbourTable, jc: Tensor) -> Tensor:
    nv = getattr(g, "_n_valid", None)
    if nv is None:
        nv = (jc >= 0).sum().to(torch.int32).reshape(1)
        g._n_valid = nv
    return nv


class _EdgeMLPFunction(torch.autograd.Function):
    """EdgeConv with an L-layer MLP ``[Linear, (BatchNorm1d), act] x L`` and aggr add / mean / max.
    ``params``: per layer W, b and, with ``cfg["bn"]``, gamma, beta.  ``cfg["bn_stats"][l]``: (mean, rstd) to use
    (eval mode: running statistics) or None (training: batch statistics, returned in ``cfg["bn_out"]``)."""

    @staticmethod
    def forward(ctx, cfg: dict, x: Tensor, *params: Tensor) -> Tensor:  # type: ignore[override]
        mode, g, aggr, bn, acts = cfg["mode"], cfg["graph"], cfg["aggr"], cfg["bn"], cfg["acts"]
        lp = mode == ops.MODE_BF16
        step = 4 if bn else 2
        L = len(params) // step
        dev = x.device
        N, Fin = int(x.shape[0]), int(x.shape[1])
        xin = torch.zeros((N, ops.round_up(Fin, 32)), dtype=torch.float32, device=dev)
        xin[:, :Fin] = x
        W0, b0 = params[0], params[1]
        H0 = int(W0.shape[0])
        H0p = ops.round_up(H0, 32)
        Wa, Wb = W0[:, :Fin], W0[:, Fin:]
        Wpq = torch.zeros((2 * H0p, Fin), dtype=torch.float32, device=dev)
        Wpq[:H0] = Wa - Wb
        Wpq[H0p:H0p + H0] = Wb
        bpq = torch.zeros(2 * H0p, dtype=torch.float32, device=dev)
        bpq[:H0] = b0
        PQ = ops.linear_fwd(mode, _ksegs([(xin, Fin)]), _wt(mode, Wpq, [Fin]), 2 * H0p, bias=bpq)
        ic, jc = ops.edge_rows(g)
        nv = _n_valid(g, jc)
        z = ops.edge_gather_pre(PQ, H0p, ic, jc)
        saved, bn_out = [], []
        a = None
        for l in range(L):
            pl = params[step * l: step * (l + 1)]
            H = int(pl[0].shape[0])
            Hp = ops.round_up(H, 32)
            last = l + 1 == L
            if bn:
                given = cfg["bn_stats"][l]
                if given is None:
                    mean, rstd, varu = ops.bn_stats(z, H, jc, nv, cfg["eps"][l])
                    bn_out.append((mean, varu))
                else:
                    mean, rstd = given
                    bn_out.append(None)
                a = ops.bn_act_fwd(z, H, jc, mean, rstd, pl[2], pl[3], acts[l], cpad=Hp, lowp=lp and not last)
                saved.append((z, mean, rstd, a))
            else:
                a, _ = ops.rownorm_act_fwd(z, H, acts[l], valid=jc, cpad=Hp, lowp="only" if lp and not last else "no")
                saved.append((z, None, None, a))
            if not last:
                Wn, bnx = params[step * (l + 1)], params[step * (l + 1) + 1]
                Hn = int(Wn.shape[0])
                z = ops.linear_fwd(mode, [(a, Hp)], _wt(mode, Wn, [H]), Hn, bias=bnx.contiguous(),
                                   out_cols=ops.round_up(Hn, 32))
        H_last = int(params[step * (L - 1)].shape[0])
        out, aux = ops.slot_reduce(a, H_last, g, aggr)
        cfg["bn_out"] = bn_out
        ctx.cfg, ctx.params, ctx.saved, ctx.aux, ctx.xin, ctx.Fin = cfg, params, saved, aux, xin, Fin
        return out

    @staticmethod
    def backward(ctx, gout: Tensor):  # type: ignore[override]
        cfg, params, saved, aux, xin, Fin = ctx.cfg, ctx.params, ctx.saved, ctx.aux, ctx.xin, ctx.Fin
        mode, g, aggr, bn, acts = cfg["mode"], cfg["graph"], cfg["aggr"], cfg["bn"], cfg["acts"]
        lp = mode == ops.MODE_BF16
        step = 4 if bn else 2
        L = len(params) // step
        dev = xin.device
        N = int(xin.shape[0])
        ic, jc = ops.edge_rows(g)
        nv = _n_valid(g, jc)
        grads: List[Optional[Tensor]] = [None] * len(params)
        H_last = int(params[step * (L - 1)].shape[0])
        g_a = ops.slot_reduce_bwd(gout.contiguous().to(torch.float32), H_last, g, aggr, aux, cpad=ops.round_up(H_last, 32))
        for l in reversed(range(L)):
            pl = params[step * l: step * (l + 1)]
            H = int(pl[0].shape[0])
            Hp = ops.round_up(H, 32)
            z, mean, rstd, _a = saved[l]
            low = lp and l > 0                      # dz of layer 0 feeds the fp32 slot sums / source gather
            if bn:
                training = cfg["bn_stats"][l] is None
                dz, grads[step * l + 2], grads[step * l + 3] = ops.bn_act_bwd(
                    g_a, z, H, jc, mean, rstd, pl[2], pl[3], acts[l], nv, training=training, cpad=Hp, lowp=low)
            else:
                dz, _, _ = ops.rownorm_act_bwd(g_a, z, H, acts[l], valid=jc, cpad=Hp, lowp="only" if low else "no")
            if l > 0:
                Hprev = int(params[step * (l - 1)].shape[0])
                Hpp = ops.round_up(Hprev, 32)
                a_prev = saved[l - 1][3]
                dW, grads[step * l + 1] = ops.linear_wgrad(mode, dz, H, [(a_prev, Hpp)], with_bias=True)
                grads[step * l] = dW[:, :Hprev]
                g_a = ops.linear_fwd(mode, _ksegs([(dz, H)]), _wt(mode, pl[0].t(), [H]), Hprev, out_cols=Hpp)
            else:
                dPQ = torch.empty((N, 2 * Hp), dtype=torch.float32, device=dev)
                dPQ[:, :Hp] = ops.slot_sum(dz, Hp, g)
                ops.edgeconv_dq_gather(ops.MODE_F32, g, dz, Hp, dPQ[:, Hp:])
                dWpq, dbpq = ops.linear_wgrad(mode, dPQ, 2 * Hp, _ksegs([(xin, Fin)]), with_bias=True)
                dWpq = dWpq[:, :Fin]
                dWp, dWq = dWpq[:H], dWpq[Hp:Hp + H]
                grads[0] = torch.cat([dWp, dWq - dWp], dim=1)
                grads[1] = dbpq[:H]
        dx = None
        if ctx.needs_input_grad[1]:
            W0 = params[0]
            H0 = int(W0.shape[0])
            H0p = ops.round_up(H0, 32)
            Wa, Wb = W0[:, :Fin], W0[:, Fin:]
            WpqT = torch.zeros((Fin, 2 * H0p), dtype=torch.float32, device=dev)
            WpqT[:, :H0] = (Wa - Wb).t()
            WpqT[:, H0p:H0p + H0] = Wb.t()
            dx = ops.linear_fwd(mode, [(dPQ, 2 * H0p)], _wt(mode, WpqT, [2 * H0p]), Fin,
                                out_cols=ops.round_up(Fin, 8))[:, :Fin]
        return (None, dx) + tuple(grads)


class _ConvBlock(torch.nn.Module):
    """Parameter holder of one DynEdgeConv block: attribute ``nn`` as in ``components/layers.py:20-50``."""

    def __init__(self, nn: torch.nn.Sequential, aggr: str, nb_neighbors: int, features_subset: Any):
        super().__init__()
        self.nn = nn
        self.aggr = aggr
        self.nb_neighbors = nb_neighbors
        self.features_subset = features_subset
        mods = list(nn)
        self._lin = [m for m in mods if isinstance(m, torch.nn.Linear)]
        self._bns = [m for m in mods if isinstance(m, torch.nn.BatchNorm1d)]
        acts = [m for m in mods if type(m) in _ACT_NAMES]
        if len(acts) != len(self._lin) or len(self._bns) not in (0, len(self._lin)) or \
                len(mods) != len(self._lin) + len(self._bns) + len(acts):
            raise NotImplementedError("graphnet_amd: edge MLP must be [Linear, (BatchNorm1d), act] x L; no fallback")
        if max(l.out_features for l in self._lin) > 512:
            raise NotImplementedError("graphnet_amd: edge MLP widths above 512 are not supported by the row kernels")
        self._acts = [_ACT_NAMES[type(a)] for a in acts]

    def forward(self, x: Tensor, table: ops.NeighbourTable, mode: int) -> Tensor:
        bn = bool(self._bns)
        params: List[Tensor] = []
        for i, lin in enumerate(self._lin):
            params += [lin.weight, lin.bias]
            if bn:
                params += [self._bns[i].weight, self._bns[i].bias]
        use_batch_stats = self.training or any(b.running_mean is None for b in self._bns)
        stats = []
        for b in self._bns:
            stats.append(None if use_batch_stats else (b.running_mean, torch.rsqrt(b.running_var + b.eps)))
        cfg = {"mode": mode, "graph": table, "aggr": self.aggr, "bn": bn, "acts": self._acts, "bn_stats": stats,
               "eps": [b.eps for b in self._bns]}
        out = _EdgeMLPFunction.apply(cfg, x, *params)
        if bn and self.training:                    # running statistics (torch.nn.BatchNorm1d: momentum 0.1, unbiased var)
            with torch.no_grad():
                for b, st in zip(self._bns, cfg["bn_out"]):
                    if st is None or b.running_mean is None:
                        continue
                    b.num_batches_tracked += 1
                    mom = b.momentum if b.momentum is not None else 1.0 / float(b.num_batches_tracked)
                    b.running_mean.mul_(1.0 - mom).add_(st[0], alpha=mom)
                    b.running_var.mul_(1.0 - mom).add_(st[1], alpha=mom)
        return out


class ParticleNeT(GNN):
    """ParticleNeT (dynamical edge convolutional) model on MI355X; constructor as ``gnn/particlenet.py:27-46``."""

    def __init__(self, nb_inputs: int, *, nb_neighbours: int = 16,
                 features_subset: Optional[Union[List[int], slice]] = None, dynamic: bool = True,
                 dynedge_layer_sizes: Optional[List[Tuple[int, ...]]] = [(64, 64, 64), (128, 128, 128), (256, 256, 256)],
                 readout_layer_sizes: Optional[List[int]] = [256],
                 global_pooling_schemes: Optional[Union[str, List[str]]] = "mean",
                 activation_layer: Optional[str] = "relu", add_batchnorm_layer: bool = True,
                 dropout_readout: float = 0.1, skip_readout: bool = False):
        if features_subset is None:
            features_subset = slice(0, 3)
        if dynedge_layer_sizes is None:
            dynedge_layer_sizes = [(64, 64, 64), (128, 128, 128), (256, 256, 256)]
        sizes = [tuple(s) for s in dynedge_layer_sizes]
        assert len(sizes) and all(len(s) > 0 and all(v > 0 for v in s) for s in sizes)
        self._dynedge_layer_sizes = sizes
        if readout_layer_sizes is None:
            readout_layer_sizes = [256]
        assert isinstance(readout_layer_sizes, list) and len(readout_layer_sizes) and all(v > 0 for v in readout_layer_sizes)
        self._readout_layer_sizes = readout_layer_sizes
        if isinstance(global_pooling_schemes, str):
            global_pooling_schemes = [global_pooling_schemes]
        if isinstance(global_pooling_schemes, list):
            for s in global_pooling_schemes:
                assert s in ops.POOL_CODES, f"Global pooling scheme {s} not supported."
        else:
            assert global_pooling_schemes is None
        self._global_pooling_schemes = global_pooling_schemes
        if activation_layer is None or activation_layer.lower() == "relu":
            act: torch.nn.Module = torch.nn.ReLU()
        elif activation_layer.lower() == "gelu":
            act = torch.nn.GELU()
        else:
            raise ValueError(f"Activation layer {activation_layer} not supported.")
        super().__init__(nb_inputs, self._readout_layer_sizes[-1])
        self._activation = act
        self._nb_inputs = nb_inputs
        self._nb_neighbours = nb_neighbours
        self._features_subset = features_subset
        self._dynamic = dynamic
        self._add_batchnorm_layer = add_batchnorm_layer
        self._dropout_readout = dropout_readout
        self._skip_readout = skip_readout
        self._compute_mode = ops.MODE_BF16
        self._knn_strict = False
        self._graph_columns = [0, 1, 2]
        # layers (particlenet.py:172-213)
        self._conv_layers = torch.nn.ModuleList()
        lat = nb_inputs
        for s in sizes:
            layers: List[torch.nn.Module] = []
            ls = [lat] + list(s)
            for ix, (nb_in, nb_out) in enumerate(zip(ls[:-1], ls[1:])):
                if ix == 0:
                    nb_in *= 2
                layers.append(torch.nn.Linear(nb_in, nb_out))
                if add_batchnorm_layer:
                    layers.append(torch.nn.BatchNorm1d(nb_out))
                layers.append(act)
            self._conv_layers.append(_ConvBlock(torch.nn.Sequential(*layers), "mean", nb_neighbours, features_subset))
            lat = ls[-1]
        lat = lat * (len(global_pooling_schemes) if global_pooling_schemes else 1)
        ro: List[torch.nn.Module] = []
        ls = [lat] + list(self._readout_layer_sizes)
        for a, b in zip(ls[:-1], ls[1:]):
            ro += [torch.nn.Linear(a, b), act, torch.nn.Dropout(dropout_readout)]
        self._readout = torch.nn.Sequential(*ro)

    def set_backend(self, *, dtype: str = "bf16", knn_mode: str = "compat",
                    graph_columns: Optional[Sequence[int]] = None) -> "ParticleNeT":
        self._compute_mode = {"fp32": ops.MODE_F32, "bf16": ops.MODE_BF16}[dtype]
        self._knn_strict = {"compat": False, "strict": True}[knn_mode]
        if graph_columns is not None:
            self._graph_columns = list(graph_columns)
        return self

    def forward(self, data: Any, return_trace: bool = False) -> Tensor:
        """Apply learnable forward pass (``particlenet.py:228-244``)."""
        x = data.x
        if not x.is_cuda:
            raise RuntimeError("graphnet_amd.ParticleNeT runs on an MI355X (HIP) device only; move the batch to 'cuda'.")
        x = x.to(torch.float32)
        N = int(x.shape[0])
        n_pulses = data.n_pulses.to(torch.int32)
        ptr = _maybe(data, "ptr")
        if ptr is None:
            ptr = torch.zeros(n_pulses.shape[0] + 1, dtype=torch.int64, device=x.device)
            ptr[1:] = torch.cumsum(torch.bincount(data.batch, minlength=n_pulses.shape[0]), 0)
        ptr32 = ptr.to(torch.int32)
        batch = _maybe(data, "batch")
        batch32 = batch.to(torch.int32) if batch is not None else ops.ptr_to_batch(ptr32, N)
        table = _maybe(data, "nbr_table")
        if not isinstance(table, ops.NeighbourTable):
            ei = _maybe(data, "edge_index")
            table = ops.table_from_edge_index(ei, N, self._nb_neighbours) if ei is not None else \
                ops.knn_graph(x, self._graph_columns, batch32, ptr32, self._nb_neighbours, strict=self._knn_strict)
        plan = ops.knn_plan(ptr32, N)
        graphs, conv_out = [table], []
        for conv in self._conv_layers:
            x = conv(x, table, self._compute_mode)
            conv_out.append(x)
            if self._dynamic:                       # components/layers.py:63-67
                cols = _subset_cols(self._features_subset, int(x.shape[1]))
                table = ops.knn_graph(x.detach(), cols, batch32, ptr32, self._nb_neighbours, strict=self._knn_strict,
                                      plan=plan)
                graphs.append(table)
        if not self._skip_readout:
            if self._global_pooling_schemes:
                pcfg = {"mode": self._compute_mode, "ptr": ptr32, "batch": batch32, "pools": self._global_pooling_schemes}
                x = _PostPoolFunction.apply(pcfg, x)
            x = self._readout(x)
        if return_trace:
            return x, {"graphs": graphs, "conv_out": conv_out}
        return x
