"""The two step schedules of the hot path that are the reference's OWN code (closures inside ``FruitModel``,
``crop_nerf/fruit_nerf/fruit_nerf.py``), as plain functions -- used by ``FruitModel.get_training_callbacks`` and
``FruitTrainer`` and pinned by vectors the reference's closures produced (``tests/golden/reference_functions.npz``,
``tests/test_reference_golden.py``)."""

from __future__ import annotations

import numpy as np


def proposal_update_schedule(step: int, proposal_warmup: int, proposal_update_every: int) -> float:
    """``update_schedule`` (``fruit_nerf.py:144-149``): the number of steps after which the proposal networks get a gradient
    again -- 1 at the start, ``proposal_update_every`` after ``proposal_warmup`` steps, linear in between."""
    return float(np.clip(np.interp(step, [0, proposal_warmup], [0, proposal_update_every]), 1, proposal_update_every))


def proposal_weights_anneal(step: int, max_num_iters: int, slope: float) -> float:
    """``set_anneal`` / ``bias`` (``fruit_nerf.py:206-216``; arXiv 2111.12077 eq. 18): the exponent applied to the proposal
    weights before PDF sampling -- ``bias(clip(step / N, 0, 1), slope)`` with ``bias(x, b) = b x / ((b - 1) x + 1)``."""
    train_frac = float(np.clip(step / max_num_iters, 0, 1))
    return slope * train_frac / ((slope - 1) * train_frac + 1)
