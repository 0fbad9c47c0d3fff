"""Weighted asynchronous label propagation as the merger uses it (``crop_nerf/segmentation/lpa.py:5-99``, networkx's
``asyn_lpa_communities`` with two changes: the node order and the tie breaks come from Python's global ``random`` module,
and a node whose best neighbouring label has a non-positive total weight keeps its own label -- affinities are signed).

This is the one module of the reference that imports in the build container (it needs networkx only), so this
restatement is PINNED: ``tests/golden/merger_small.npz`` holds communities produced by the reference's own function under
fixed ``random.seed`` values (``tests/golden/make_golden_merger.py``), and this one must reproduce them draw for draw."""

from __future__ import annotations

import random
from typing import Dict, Hashable, Iterable, List, Optional, Set


def _votes(G, node, labels: Dict[Hashable, int], weight: Optional[str]) -> Dict[int, float]:
    """Total (weighted) vote per label among the neighbours of ``node``, in neighbour order."""
    tally: Dict[int, float] = {}
    if weight is None:
        for nb in G[node]:
            tally[labels[nb]] = tally.get(labels[nb], 0) + 1
    else:
        for _, nb, w in G.edges(node, data=weight, default=1):
            tally[labels[nb]] = tally.get(labels[nb], 0.0) + w
    return tally


def asyn_lpa_communities(G, weight: Optional[str] = None, seed=None) -> Iterable[Set]:
    """Communities of ``G`` (sets of nodes).  ``seed`` is accepted and ignored, as in the reference: call
    ``random.seed`` beforehand for a reproducible run."""
    labels = {node: i for i, node in enumerate(G)}
    changed = True
    while changed:
        changed = False
        order = list(G)
        random.shuffle(order)
        for node in order:
            if not G[node]:  # isolated
                continue
            tally = _votes(G, node, labels, weight)
            top = max(tally.values())
            winners: List[int] = [lab for lab, v in tally.items() if v == top] if top > 0 else [labels[node]]
            if labels[node] not in winners:
                labels[node] = random.choice(winners)
                changed = True
    groups: Dict[int, Set] = {}
    for node, lab in labels.items():
        groups.setdefault(lab, set()).add(node)
    return iter(groups.values())
