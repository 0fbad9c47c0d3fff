"""Graph stage of the merger: the signed label propagation is pinned by vectors that the reference's
own ``lpa.py`` produced (tests/golden/merger_small.npz, made by tests/golden/make_golden_merger.py); affinity and the
clique / bridge partitions by known answers."""

import os
import random

import networkx as nx
import numpy as np
import pytest

from cropnerf_amd.segmentation import label_propagation as LP
from cropnerf_amd.segmentation import merger as MG

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "merger_small.npz")


def test_label_propagation_reproduces_the_reference_vectors():
    gold = np.load(GOLD)
    n_cases = int(gold["num_cases"])
    assert n_cases == 42
    for c in range(n_cases):
        a, seed, want = gold[f"case{c}/affinity"], int(gold[f"case{c}/seed"]), gold[f"case{c}/labels"]
        random.seed(seed)
        labels = np.zeros(len(a), dtype=np.int64)
        for k, community in enumerate(LP.asyn_lpa_communities(nx.from_numpy_array(a), weight="weight")):
            labels[list(community)] = k + 1
        assert np.array_equal(labels, want), f"case {c} (n = {len(a)}, seed {seed})"


def test_label_propagation_unweighted_and_isolated_nodes():
    random.seed(0)
    G = nx.Graph()
    G.add_nodes_from(range(5))
    G.add_edges_from([(0, 1), (1, 2), (0, 2)])          # a triangle, two isolated nodes
    comms = sorted(sorted(c) for c in LP.asyn_lpa_communities(G))
    assert comms == [[0, 1, 2], [3], [4]]
    # a negative edge never pulls two nodes together: both keep their own label
    H = nx.Graph()
    H.add_edge(0, 1, weight=-2.0)
    assert sorted(sorted(c) for c in LP.asyn_lpa_communities(H, weight="weight")) == [[0], [1]]


def _props(labels, rel=None):
    labels = np.asarray(labels)
    rel = np.ones_like(labels, dtype=float) if rel is None else np.asarray(rel, dtype=float)
    return {i: {"label": labels[i], "reliability": rel[i]} for i in range(len(labels))}


def test_calc_affinity_known_answers():
    # three sub-clusters seen by four cameras; 0 = background (ignored)
    lab = [[5, 5, 0, 7], [5, 6, 3, 7], [0, 0, 3, 8]]
    rel = [[1.0, 0.5, 1.0, 2.0], [1.0, 1.0, 1.0, 1.0], [1.0, 1.0, 0.5, 1.0]]
    a = MG.calc_affinity(_props(lab, rel))
    assert a.shape == (3, 3) and np.allclose(a, a.T) and np.allclose(np.diag(a), 0)
    assert a[0, 1] == pytest.approx(1.0 * 1.0 + 2.0 * 1.0 - 0.5 * 1.0)     # same at cams 0, 3; different at cam 1
    assert a[0, 2] == pytest.approx(-2.0 * 1.0)                            # only cam 3 is seen by both: 7 vs 8
    assert a[1, 2] == pytest.approx(1.0 * 0.5 - 1.0 * 1.0)                 # same at cam 2, different at cam 3


def test_get_component_partitions():
    # nodes 0-1-2 form a triangle, 3 hangs off 2 by one edge (a bridge), 4 is negative to everyone
    a = np.zeros((5, 5))
    for i, j, w in [(0, 1, 1.0), (1, 2, 2.0), (0, 2, 0.5), (2, 3, 0.3), (3, 4, -1.0), (0, 4, -0.2)]:
        a[i, j] = a[j, i] = w
    n, labels = MG.get_component(a, "clique")
    assert n == 3 and len(set(labels[:3])) == 1 and len({labels[0], labels[3], labels[4]}) == 3 and labels.min() >= 1
    n, labels = MG.get_component(a, "bridge")
    assert n == 1 and labels[:3].tolist() == [1, 1, 1] and labels[3] == 0 and labels[4] == 0   # singletons are dropped
    random.seed(35)
    n, labels = MG.get_component(a, "community")
    assert labels[0] == labels[1] == labels[2] and labels[4] not in (labels[0], labels[3]) and n == len(set(labels))
    with pytest.raises(ValueError):
        MG.get_component(a, "spectral")
    total, all_labels = MG.count_fruit([_props([[1, 1], [1, 1]]), _props([[1, 2], [2, 1], [0, 0]])], "clique")
    assert total == 1 + 3 and all_labels[1].min() == 2     # second super-cluster's labels are shifted past the first's


def test_depth_projection_merger_graph_stage_reproduces_the_reference():
    """``segmentation/depth_projection_based_merger.py``: its ``calc_affinity`` (``:275-297``) and its own ``get_component``
    (``:23-61``: edge weights kept for clique / bridge, networkx's label propagation for community), executed from the
    reference's source on seeded cluster properties (``tests/golden/make_golden_reference.py: depth_merger_cases``) -- the
    mirror gives the same affinities and, on the row-normalised affinity of its ``main`` (``:330``), the same partitions."""
    from cropnerf_amd.segmentation import depth_projection_based_merger as DM

    gold = np.load(os.path.join(os.path.dirname(GOLD), "reference_functions.npz"))
    n_cases = int(gold["num_dpm"])
    assert n_cases >= 4
    for c in range(n_cases):
        labels, rel = gold[f"dpm{c}/labels"], gold[f"dpm{c}/reliability"]
        prop = {i: {"label": labels[i], "reliability": rel[i]} for i in range(len(labels))}
        aff = DM.calc_affinity(prop)
        assert np.array_equal(aff, gold[f"dpm{c}/affinity"]), c
        norm = DM.normalise_affinity(aff)
        assert np.isfinite(norm).all()
        for algo in ("clique", "bridge", "community"):
            random.seed(100 + c)
            k, lab = DM.get_component(norm.copy(), algo)
            assert k == int(gold[f"dpm{c}/{algo}_count"]), (c, algo)
            assert np.array_equal(np.asarray(lab), gold[f"dpm{c}/{algo}_labels"]), (c, algo)
    # a row whose maximum is 0 divides by zero, as in the reference (its own comment: "buggy!!! handle case when max 0")
    bad = DM.normalise_affinity(np.array([[0.0, -1.0], [-1.0, 0.0]]))
    assert not np.isfinite(bad).all()
    with pytest.raises(ValueError):
        DM.get_component(np.eye(2), "spectral")
