#!/usr/bin/env python
"""Golden vectors of the merger's signed label propagation, produced by the REFERENCE'S OWN function.

    python tests/golden/make_golden_merger.py          ->  tests/golden/merger_small.npz

``crop_nerf/segmentation/lpa.py`` needs only networkx, so -- unlike the rest of the reference -- it imports in the build
container.  This script imports it from /root/reference (it is not copied anywhere), runs it on seeded signed affinity
matrices under fixed ``random.seed`` values, and stores inputs and resulting node labels.  The fixture travels; the
reference does not."""

import os
import random
import sys

import networkx as nx
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference/crop_nerf/segmentation")
import lpa as reference_lpa  # noqa: E402


def main():
    rng = np.random.default_rng(7)
    out = {}
    case = 0
    for n in (2, 3, 5, 6, 8, 10, 12):
        for density in (0.4, 0.8):
            a = rng.normal(size=(n, n)) * (rng.uniform(size=(n, n)) < density)
            a = np.triu(a, 1)
            a = a + a.T
            for seed in (0, 1, 35):
                random.seed(seed)
                G = nx.from_numpy_array(a)
                labels = np.zeros(n, dtype=np.int64)
                for k, community in enumerate(reference_lpa.asyn_lpa_communities(G, weight="weight")):
                    labels[list(community)] = k + 1
                out[f"case{case}/affinity"] = a
                out[f"case{case}/seed"] = np.array(seed)
                out[f"case{case}/labels"] = labels
                case += 1
    out["num_cases"] = np.array(case)
    path = os.path.join(HERE, "merger_small.npz")
    np.savez_compressed(path, **out)
    print(path, case, "cases")


if __name__ == "__main__":
    main()
