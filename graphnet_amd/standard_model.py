"""Model composition: ``StandardModel``, tasks, losses, LR schedule, training step.

Host-side mirror of the reference's L3/L4 glue on the DynEdge path — plain torch, no kernels:
``models/standard_model.py:24-119``, ``models/task/task.py:22-337``,
``models/task/reconstruction.py:101-112``, ``training/loss_functions.py:23-112``,
``training/callbacks.py:25-78``, ``models/easy_model.py:215-256`` (optimizer / training step).
Lightning is not a dependency: :meth:`StandardModel.fit` is a small explicit loop with the same
step semantics (forward, summed task losses, backward, optimizer step, per-step LR schedule).
"""
from __future__ import annotations

from typing import Any, Callable, Dict, List, Optional, Sequence, Type, Union

import numpy as np
import torch
from torch import Tensor
from torch.nn import Linear
from torch.optim import Adam
from torch.optim.lr_scheduler import _LRScheduler

from .data import Data
from .model import Model


def eps_like(tensor: Tensor) -> float:
    """``utilities/maths.py:6-8``."""
    return torch.finfo(tensor.dtype).eps


# ------------------------------------------------------------------------------ losses
class LossFunction(Model):
    """``training/loss_functions.py:23-60``."""

    def forward(self, prediction: Tensor, target: Tensor, weights: Optional[Tensor] = None,
                return_elements: bool = False) -> Tensor:
        elements = self._forward(prediction, target)
        if weights is not None:
            elements = elements * weights
        assert elements.size(dim=0) == target.size(dim=0), "`_forward` should return elementwise loss terms."
        return elements if return_elements else torch.mean(elements)

    def _forward(self, prediction: Tensor, target: Tensor) -> Tensor:  # pragma: no cover
        raise NotImplementedError


class MSELoss(LossFunction):
    def _forward(self, prediction: Tensor, target: Tensor) -> Tensor:
        assert prediction.dim() == 2
        if target.dim() != prediction.dim():
            target = target.squeeze(1)
        assert prediction.size() == target.size()
        return torch.mean((prediction - target) ** 2, dim=-1)


class LogCoshLoss(LossFunction):
    """``training/loss_functions.py:93-112``: x + softplus(-2x) - log 2."""

    @classmethod
    def _log_cosh(cls, x: Tensor) -> Tensor:
        return x + torch.nn.functional.softplus(-2.0 * x) - np.log(2.0)

    def _forward(self, prediction: Tensor, target: Tensor) -> Tensor:
        return self._log_cosh(prediction - target)


# ------------------------------------------------------------------------------ tasks
class Task(Model):
    """``models/task/task.py:22-222`` (transform handling reduced to what the path uses)."""

    nb_inputs: int = 1
    default_target_labels: List[str] = []
    default_prediction_labels: List[str] = []

    def __init__(self, *, target_labels: Optional[Union[str, List[str]]] = None,
                 prediction_labels: Optional[Union[str, List[str]]] = None,
                 transform_prediction_and_target: Optional[Callable] = None,
                 transform_target: Optional[Callable] = None, transform_inference: Optional[Callable] = None,
                 transform_support: Optional[tuple] = None, loss_weight: Optional[str] = None):
        super().__init__()
        if target_labels is None:
            target_labels = self.default_target_labels
        if isinstance(target_labels, str):
            target_labels = [target_labels]
        if prediction_labels is None:
            prediction_labels = self.default_prediction_labels
        if isinstance(prediction_labels, str):
            prediction_labels = [prediction_labels]
        self._target_labels = target_labels
        self._prediction_labels = prediction_labels
        self._inference = False
        self._loss_weight = loss_weight
        self._regularisation_loss: Any = 0
        ident = lambda x: x  # noqa: E731
        self._transform_prediction_training: Callable = ident
        self._transform_prediction_inference: Callable = ident
        self._transform_target: Callable = ident
        assert not ((transform_prediction_and_target is not None) and (transform_target is not None)), (
            "Please specify at most one of `transform_prediction_and_target` and `transform_target`")
        if (transform_target is not None) != (transform_inference is not None):
            raise AssertionError("Please specify both `transform_inference` and `transform_target`")
        if transform_target is not None:
            self._transform_target = transform_target
            self._transform_prediction_inference = transform_inference
        elif transform_prediction_and_target is not None:
            self._transform_prediction_training = transform_prediction_and_target
            self._transform_target = transform_prediction_and_target

    def inference(self) -> None:
        self._inference = True

    def train_eval(self) -> None:
        self._inference = False

    def _transform_prediction(self, prediction: Tensor) -> Tensor:
        if self._inference:
            return self._transform_prediction_inference(prediction)
        return self._transform_prediction_training(prediction)


class StandardLearnedTask(Task):
    """``task.py:225-337``: learned affine head + supervised loss."""

    def __init__(self, hidden_size: int, loss_function: LossFunction, **task_kwargs: Any):
        super().__init__(**task_kwargs)
        self._loss_function = loss_function
        self._affine = Linear(hidden_size, self.nb_inputs)

    def _forward(self, x: Tensor) -> Tensor:  # pragma: no cover
        raise NotImplementedError

    def forward(self, x: Tensor) -> Tensor:
        self._regularisation_loss = 0
        x = self._affine(x)
        x = self._forward(x=x)
        return self._transform_prediction(x)

    def compute_loss(self, pred: Tensor, data: Any) -> Tensor:
        target = torch.stack([data[label] for label in self._target_labels], dim=1)
        target = self._transform_target(target)
        weights = data[self._loss_weight] if self._loss_weight is not None else None
        return self._loss_function(pred, target, weights=weights) + self._regularisation_loss


class IdentityTask(StandardLearnedTask):
    def __init__(self, nb_outputs: int, target_labels: Union[List[str], Any], *args: Any, **kwargs: Any):
        self.nb_inputs = nb_outputs  # type: ignore[misc]
        super().__init__(*args, target_labels=target_labels, **kwargs)

    def _forward(self, x: Tensor) -> Tensor:
        return x


class EnergyReconstruction(StandardLearnedTask):
    """``task/reconstruction.py:101-112``."""

    default_target_labels = ["energy"]
    default_prediction_labels = ["energy_pred"]
    nb_inputs = 1

    def _forward(self, x: Tensor) -> Tensor:
        return torch.nn.functional.softplus(x, beta=0.05) + eps_like(x)


# ------------------------------------------------------------------------------ LR schedule
class PiecewiseLinearLR(_LRScheduler):
    """``training/callbacks.py:25-78``."""

    def __init__(self, optimizer, milestones: List[int], factors: List[float], last_epoch: int = -1):
        if milestones != sorted(milestones):
            raise ValueError("Milestones must be increasing")
        if len(milestones) != len(factors):
            raise ValueError("Only multiplicative factor must be specified for each milestone.")
        self.milestones = milestones
        self.factors = factors
        super().__init__(optimizer, last_epoch)

    def _get_factor(self) -> np.ndarray:
        return np.interp(self.last_epoch, self.milestones, self.factors)

    def get_lr(self) -> List[float]:
        return [base_lr * self._get_factor() for base_lr in self.base_lrs]


# ------------------------------------------------------------------------------ StandardModel
class StandardModel(Model):
    """backbone + tasks (``models/standard_model.py:24-119``)."""

    def __init__(self, *, graph_definition: Any, backbone: Model = None, tasks: Union[Task, List[Task]] = None,
                 optimizer_class: Type[torch.optim.Optimizer] = Adam, optimizer_kwargs: Optional[Dict] = None,
                 scheduler_class: Optional[type] = None, scheduler_kwargs: Optional[Dict] = None,
                 scheduler_config: Optional[Dict] = None, gnn: Optional[Model] = None) -> None:
        super().__init__()
        if isinstance(tasks, Task):
            tasks = [tasks]
        assert isinstance(tasks, (list, tuple)) and all(isinstance(t, Task) for t in tasks)
        if backbone is None and isinstance(gnn, Model):
            backbone = gnn                        # deprecated keyword of the reference (l.50-60)
        elif backbone is None:
            raise TypeError("__init__() missing 1 required keyword argument:'backbone'")
        assert isinstance(backbone, Model)
        self._graph_definition = graph_definition
        self.backbone = backbone
        self._tasks = torch.nn.ModuleList(tasks)
        self._optimizer_class = optimizer_class
        self._optimizer_kwargs = optimizer_kwargs or dict()
        self._scheduler_class = scheduler_class
        self._scheduler_kwargs = scheduler_kwargs or dict()
        self._scheduler_config = scheduler_config or dict()

    @property
    def target_labels(self) -> List[str]:
        return [label for task in self._tasks for label in task._target_labels]

    @property
    def prediction_labels(self) -> List[str]:
        return [label for task in self._tasks for label in task._prediction_labels]

    def forward(self, data: Union[Any, List[Any]]) -> List[Tensor]:
        if not isinstance(data, (list, tuple)):
            data = [data]
        x = torch.cat([self.backbone(d) for d in data], dim=0)
        return [task(x) for task in self._tasks]

    def compute_loss(self, preds: List[Tensor], data: List[Any], verbose: bool = False) -> Tensor:
        data_merged = {}
        for label in list(set(self.target_labels)):
            data_merged[label] = torch.cat([d[label] for d in data], dim=0)
        for task in self._tasks:
            if task._loss_weight is not None:
                data_merged[task._loss_weight] = torch.cat([d[task._loss_weight] for d in data], dim=0)
        losses = [task.compute_loss(pred, data_merged) for task, pred in zip(self._tasks, preds)]
        assert all(loss.dim() == 0 for loss in losses), "Please reduce loss for each task separately"
        return torch.sum(torch.stack(losses))

    def shared_step(self, batch: Union[Any, List[Any]], batch_idx: int = 0) -> Tensor:
        if not isinstance(batch, (list, tuple)):
            batch = [batch]
        preds = self(batch)
        return self.compute_loss(preds, batch)

    def configure_optimizers(self):
        """``easy_model.py:215-235``."""
        optimizer = self._optimizer_class(self.parameters(), **self._optimizer_kwargs)
        scheduler = None
        if self._scheduler_class is not None:
            scheduler = self._scheduler_class(optimizer, **self._scheduler_kwargs)
        return optimizer, scheduler

    def inference(self) -> None:
        for task in self._tasks:
            task.inference()

    def train(self, mode: bool = True) -> "StandardModel":
        super().train(mode)
        if mode:
            for task in self._tasks:
                task.train_eval()
        return self

    def fit(self, train_batches: Sequence[Any], max_epochs: int = 1, device: str = "cuda",
            grad_sync: Optional[Callable[[], None]] = None, log_every: int = 0) -> List[float]:
        """Minimal explicit training loop with the step semantics of ``easy_model.py:237-256``.
        ``grad_sync`` (e.g. ``FlatGradAllReduce.__call__``) runs between backward and step."""
        self.to(device)
        self.train()
        optimizer, scheduler = self.configure_optimizers()
        history: List[float] = []
        step = 0
        for _epoch in range(max_epochs):
            for batch in train_batches:
                batch = batch.to(device) if isinstance(batch, Data) else batch
                loss = self.shared_step(batch, step)
                optimizer.zero_grad(set_to_none=True)
                loss.backward()
                if grad_sync is not None:
                    grad_sync()
                optimizer.step()
                if scheduler is not None:
                    scheduler.step()
                step += 1
                if log_every and step % log_every == 0:
                    history.append(float(loss.detach()))
        return history

    @torch.no_grad()
    def predict(self, batches: Sequence[Any], device: str = "cuda") -> List[Tensor]:
        """``easy_model.py:289-319``: inference-mode transforms, concatenated per task."""
        self.to(device)
        self.eval()
        self.inference()
        outs = [self(b.to(device) if isinstance(b, Data) else b) for b in batches]
        self.train()
        return [torch.cat([o[i] for o in outs], dim=0) for i in range(len(self._tasks))]
