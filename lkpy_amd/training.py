"""
Training interfaces -- mirror of ``lenskit.training`` (src/lenskit/training.py:40-378):
``TrainingOptions`` (retrain, device, rng, environment, ``LK_DEVICE``), ``Trainable``,
``UsesTrainer`` (epoch loop with per-epoch timing), ``ModelTrainer``.
"""

from __future__ import annotations

import logging
import os
from abc import ABC, abstractmethod
from dataclasses import dataclass, field
from time import perf_counter
from typing import Protocol, runtime_checkable

import numpy as np

_log = logging.getLogger(__name__)


@dataclass(frozen=True)
class TrainingOptions:
    retrain: bool = True
    device: str | None = None
    rng: object = None
    environment: dict[str, str] = field(default_factory=dict)

    def random_generator(self) -> np.random.Generator:
        "``random_generator(seed)`` -> ``np.random.default_rng`` (src/lenskit/random.py:181-185)."
        if isinstance(self.rng, np.random.Generator):
            return self.rng
        return np.random.default_rng(self.rng)

    def configured_device(self, *, gpu_default: bool = True) -> str:
        "device -> LK_DEVICE -> 'cuda' (src/lenskit/training.py:125-149)."
        if self.device is not None:
            return self.device
        if dev := os.environ.get("LK_DEVICE", None):
            return dev
        return "cuda"

    def env_var(self, name: str, default: str | None = None) -> str | None:
        if name in self.environment:
            return self.environment[name]
        return os.environ.get(name, default)


@runtime_checkable
class Trainable(Protocol):
    def is_trained(self) -> bool: ...

    def train(self, data, options: TrainingOptions) -> None: ...


class ModelTrainer(ABC):
    @abstractmethod
    def train_epoch(self) -> dict[str, float] | None: ...

    def finalize(self) -> None:
        pass


class UsesTrainer(ABC):
    "``UsesTrainer.train`` (src/lenskit/training.py:301-334): epoch loop over a ModelTrainer."

    trained_epochs: int = 0

    @property
    def expected_training_epochs(self) -> int | None:
        cfg = getattr(self, "config", None)
        return getattr(cfg, "epochs", None) if cfg is not None else None

    def is_trained(self):
        return self.trained_epochs > 0

    def train(self, data, options: TrainingOptions = TrainingOptions()) -> None:
        if self.trained_epochs > 0 and not options.retrain:
            return
        self.trained_epochs = 0
        n = self.expected_training_epochs
        assert n is not None, "no training epochs configured"
        trainer = self.create_trainer(data, options)
        start = perf_counter()
        for i in range(1, n + 1):
            metrics = trainer.train_epoch() or {}
            now = perf_counter()
            # the reference's own epochs/sec source (training.py:320-329)
            _log.info("finished epoch %d time=%.4fs %s", i, now - start, metrics)
            self.trained_epochs += 1
            start = now
        trainer.finalize()

    @abstractmethod
    def create_trainer(self, data, options: TrainingOptions) -> ModelTrainer: ...
