"""Task heads on the pooled [B, C] graph embedding (mirror of the reference's
``models/decoders.py:15-320``).  B x 128 MLPs: negligible compute, plain torch.nn modules."""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F


def _act(name: str) -> nn.Module:
    return {"relu": nn.ReLU, "gelu": nn.GELU, "elu": nn.ELU}.get(name, nn.ReLU)()


def _mlp(dims: List[int], act: nn.Module, dropout: float, use_batch_norm: bool) -> List[nn.Module]:
    layers: List[nn.Module] = []
    for a, b in zip(dims[:-1], dims[1:]):
        layers.append(nn.Linear(a, b))
        if use_batch_norm:
            layers.append(nn.BatchNorm1d(b))
        layers += [act, nn.Dropout(dropout)]
    return layers


class ClassificationHead(nn.Module):
    def __init__(self, input_dim: int, num_classes: int, hidden_dims: Optional[List[int]] = None, dropout: float = 0.1,
                 activation: str = "gelu", use_batch_norm: bool = True, class_weights: Optional[torch.Tensor] = None,
                 label_smoothing: float = 0.0):
        super().__init__()
        self.num_classes, self.label_smoothing = num_classes, label_smoothing
        hidden_dims = [input_dim // 2] if hidden_dims is None else hidden_dims
        self.activation = _act(activation)
        dims = [input_dim] + list(hidden_dims)
        self.classifier = nn.Sequential(*_mlp(dims, self.activation, dropout, use_batch_norm), nn.Linear(dims[-1], num_classes))
        if class_weights is not None:
            self.register_buffer("class_weights", class_weights)
        else:
            self.class_weights = None

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.classifier(x)

    def compute_loss(self, logits: torch.Tensor, targets: torch.Tensor) -> torch.Tensor:
        if self.label_smoothing > 0:
            logp = F.log_softmax(logits, dim=-1)
            soft = torch.zeros_like(logp).scatter_(1, targets.unsqueeze(1), 1 - self.label_smoothing)
            soft = soft + self.label_smoothing / self.num_classes
            return -(soft * logp).sum(dim=-1).mean()
        return F.cross_entropy(logits, targets, weight=self.class_weights)

    def predict(self, x: torch.Tensor, return_probs: bool = False) -> torch.Tensor:
        with torch.no_grad():
            logits = self.forward(x)
            return F.softmax(logits, dim=-1) if return_probs else torch.argmax(logits, dim=-1)


class RegressionHead(nn.Module):
    def __init__(self, input_dim: int, num_targets: int, hidden_dims: Optional[List[int]] = None, dropout: float = 0.1,
                 activation: str = "gelu", use_batch_norm: bool = True, output_activation: Optional[str] = None,
                 predict_uncertainty: bool = False):
        super().__init__()
        self.num_targets, self.predict_uncertainty = num_targets, predict_uncertainty
        hidden_dims = [input_dim // 2] if hidden_dims is None else hidden_dims
        self.activation = _act(activation)
        self.output_activation = {"sigmoid": nn.Sigmoid, "tanh": nn.Tanh, "softplus": nn.Softplus}.get(output_activation, nn.Identity)()
        dims = [input_dim] + list(hidden_dims)
        self.feature_layers = nn.Sequential(*_mlp(dims, self.activation, dropout, use_batch_norm))
        self.mean_head = nn.Linear(dims[-1], num_targets)
        self.var_head = nn.Linear(dims[-1], num_targets) if predict_uncertainty else None

    def forward(self, x: torch.Tensor):
        f = self.feature_layers(x)
        mean = self.output_activation(self.mean_head(f))
        if self.predict_uncertainty:
            log_var = self.var_head(f)
            return {"mean": mean, "var": torch.exp(log_var), "log_var": log_var}
        return mean

    def compute_loss(self, predictions, targets: torch.Tensor, loss_type: str = "mse") -> torch.Tensor:
        if isinstance(predictions, dict):
            mean, var = predictions["mean"], predictions["var"]
            if loss_type == "gaussian_nll":
                return (0.5 * (torch.log(var) + (targets - mean) ** 2 / var)).mean()
            predictions = mean
        if loss_type == "mse":
            return F.mse_loss(predictions, targets)
        if loss_type == "mae":
            return F.l1_loss(predictions, targets)
        if loss_type == "huber":
            return F.huber_loss(predictions, targets)
        raise ValueError(f"Unknown loss type: {loss_type}")

    def predict(self, x: torch.Tensor) -> torch.Tensor:
        with torch.no_grad():
            out = self.forward(x)
            return out["mean"] if isinstance(out, dict) else out
