"""Oversampling with replacement for the imbalanced QSAR assays (reference ``data.py:136-166``).

The reference weights every training molecule by the inverse size of its class and draws ``len(dataset)`` indices
per epoch with ``torch.utils.data.WeightedRandomSampler`` seeded from the run's seed.  Same weights, same sampler
class, same generator seeding here -- so the same labels and seed give the same index stream -- computed from a label
tensor in one pass instead of a Python loop over ``Data`` objects.
"""
from __future__ import annotations

import torch
from torch.utils.data import WeightedRandomSampler


def oversampling_weights(labels: torch.Tensor) -> torch.Tensor:
    """``1 / #inactive`` for label 0, ``1 / #active`` otherwise (``data.py:146-151``), float32 like the reference's."""
    y = torch.as_tensor(labels).reshape(-1)
    active = y != 0
    n_active = int(active.sum())
    n_inactive = y.numel() - n_active
    w_active = torch.tensor(1. / n_active if n_active else float("inf"))
    w_inactive = torch.tensor(1. / n_inactive if n_inactive else float("inf"))
    return torch.where(active.cpu(), w_active, w_inactive)


def oversampling_sampler(labels: torch.Tensor, seed: int) -> WeightedRandomSampler:
    """The sampler ``train_dataloader`` builds when ``enable_oversampling_with_replacement`` is set (``data.py:153-159``)."""
    weights = oversampling_weights(labels)
    generator = torch.Generator()
    generator.manual_seed(seed)
    return WeightedRandomSampler(weights=weights, num_samples=len(weights), generator=generator)
