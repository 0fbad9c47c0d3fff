from dataclasses import dataclass
from enum import Enum, auto
from typing import Any, Callable, Dict, List, Optional


@dataclass
class TrainingCallbackAttributes:
    optimizers: Any = None
    grad_scaler: Any = None
    pipeline: Any = None
    trainer: Any = None


class TrainingCallbackLocation(Enum):
    BEFORE_TRAIN_ITERATION = auto()
    AFTER_TRAIN_ITERATION = auto()
    AFTER_TRAIN = auto()


class TrainingCallback:
    def __init__(self, where_to_run: List[TrainingCallbackLocation], func: Callable, update_every_num_iters: Optional[int] = None,
                 iters=None, args: Optional[List] = None, kwargs: Optional[Dict] = None):
        self.where_to_run, self.func, self.update_every_num_iters, self.iters = where_to_run, func, update_every_num_iters, iters
        self.args, self.kwargs = args or [], kwargs or {}

    def run_callback(self, step: int) -> None:
        if self.update_every_num_iters is not None:
            if step % self.update_every_num_iters == 0:
                self.func(*self.args, **self.kwargs, step=step)
        elif self.iters is not None:
            if step in self.iters:
                self.func(*self.args, **self.kwargs, step=step)

    def run_callback_at_location(self, step: int, location: TrainingCallbackLocation) -> None:
        if location in self.where_to_run:
            self.run_callback(step)
