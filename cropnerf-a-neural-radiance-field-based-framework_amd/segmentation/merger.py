"""Graph stage of the merger -- mirror of ``crop_nerf/segmentation/merger.py:335-355`` (``calc_affinity``) and ``:26-74``
(``get_component``): from the per-camera label and reliability of every sub-cluster of one super-cluster to the affinity
matrix, its partition (maximal cliques / bridge removal / signed label propagation) and the fruit count.

The stage before it (``:219-333``: areas, bounding boxes and majority labels of the projected sub-clusters, via OpenCV
contours on the PNGs and the GroundedSAM instance-label frames) is not mirrored: neither OpenCV nor those label frames
exist here, and contour tracing cannot be pinned without them.  Host code on a handful of nodes; networkx as upstream."""

from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import numpy as np

from . import label_propagation


def calc_affinity(cluster_prop: Dict[int, Dict[str, np.ndarray]]) -> np.ndarray:
    """``:335-355``: a_ij = sum over cameras where both sub-clusters carry the SAME non-background instance label of
    r_i r_j, minus the same sum over cameras where they carry DIFFERENT non-background labels."""
    n = len(cluster_prop)
    lab = np.stack([np.asarray(cluster_prop[i]["label"]) for i in range(n)])
    rel = np.stack([np.asarray(cluster_prop[i]["reliability"], dtype=np.float64) for i in range(n)])
    seen = lab != 0
    affinity = np.zeros((n, n))
    for i in range(n):
        for j in range(i + 1, n):
            both = seen[i] & seen[j]
            same = both & (lab[i] == lab[j])
            diff = both & (lab[i] != lab[j])
            affinity[i, j] = affinity[j, i] = rel[i][same] @ rel[j][same] - rel[i][diff] @ rel[j][diff]
    return affinity


def get_component(affinity: np.ndarray, algo: str) -> Tuple[int, np.ndarray]:
    """``:26-74`` without the drawing: (number of components, label per node; 0 = dropped singleton in 'bridge')."""
    import networkx as nx

    if algo in ("clique", "bridge"):
        affinity = np.where(affinity > 0, 1, 0)
    G = nx.from_numpy_array(affinity)
    labels = np.zeros(G.order())
    components: List[Sequence[int]] = []
    next_label = 1
    if algo == "clique":  # peel maximal cliques, largest first
        while G.order() > 0:
            clique = max(nx.find_cliques(G), key=len)
            components.append(clique)
            G.remove_nodes_from(clique)
            labels[clique] = next_label
            next_label += 1
    elif algo == "bridge":  # cut the bridges of every component with more than two nodes; singletons are dropped
        for comp in [G.subgraph(c).copy() for c in nx.connected_components(G)]:
            if len(comp) > 2:
                for e in list(nx.bridges(comp)):
                    comp.remove_edge(*e)
            for c in nx.connected_components(comp):
                if len(c) == 1:
                    labels[list(c)] = 0
                    continue
                components.append(c)
                labels[list(c)] = next_label
                next_label += 1
    elif algo == "community":
        for c in label_propagation.asyn_lpa_communities(G, weight="weight"):
            c = list(c)
            components.append(c)
            labels[c] = next_label
            next_label += 1
    else:
        raise ValueError(f"unknown graph partition {algo!r} (clique | bridge | community)")
    return len(components), labels


def count_fruit(cluster_props: Sequence[Dict[int, Dict[str, np.ndarray]]], graph_partition: str = "clique"
                ) -> Tuple[int, List[np.ndarray]]:
    """``main`` (``:389-437``) for the graph stage: total count over the super-clusters and the per-node labels, shifted so
    that they are unique over all super-clusters."""
    total, all_labels = 0, []
    for props in cluster_props:
        n, labels = get_component(calc_affinity(props), graph_partition)
        all_labels.append(labels + total)
        total += n
    return total, all_labels
