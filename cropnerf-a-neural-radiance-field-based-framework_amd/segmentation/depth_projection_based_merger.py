"""The merger of the depth-based projection -- mirror of ``crop_nerf/segmentation/depth_projection_based_merger.py``: the same
image and graph stages as ``merger.py`` over the tree that ``scripts/depth_based_semantic_projection.py`` writes
(``artifacts/<recording>/depth_projection/super_cluster_<i>/cam_<j>/{wo_occ,visible}_cluster_<c>.png``), with this variant's
own choices:

* reliability without ``--area_normalize`` = label-overlap area / un-occluded area (``:260-263``; ``merger.py`` uses ones);
* the affinity matrix is divided row-wise by ``|max of the row|`` before it is partitioned (``:330``; a row whose maximum is 0
  divides by zero there, as the reference's own comment in ``merger.py:398`` notes -- reproduced, with numpy's warning silenced);
* ``get_component`` (``:23-61``) keeps the WEIGHTS for the ``clique`` / ``bridge`` partitions (an edge wherever the normalised
  affinity is non-zero, negative ones included) and uses networkx's own ``asyn_lpa_communities`` for ``community``, the default
  here;
* every camera is used (``--frame_sampling_interval 1``) and at most four super-clusters (``:355``).

The image stage is ``merger.process_super_cluster`` (contours on the device, ``cn_contour_largest``)."""

from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np

from . import merger

calc_affinity = merger.calc_affinity  # ``:275-297``: the same function as ``merger.py:335-355``


def get_component(affinity: np.ndarray, algo: str) -> Tuple[int, np.ndarray]:
    """``:23-61``: (number of components, label per node)."""
    import networkx as nx

    G = nx.from_numpy_array(affinity)
    labels = np.zeros(G.order())
    components: List[Sequence[int]] = []
    next_label = 1
    if algo == "clique":
        while G.order() > 0:
            clique = max(nx.find_cliques(G), key=len)
            components.append(clique)
            G.remove_nodes_from(clique)
            labels[clique] = next_label
            next_label += 1
    elif algo == "bridge":
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
        for c in nx.algorithms.community.asyn_lpa_communities(G, weight="weight"):
            c = list(c)
            components.append(c)
            labels[c] = next_label
            next_label += 1
    else:
        raise ValueError(f"unknown graph partition {algo!r} (clique | bridge | community)")
    return len(components), labels


def normalise_affinity(affinity: np.ndarray) -> np.ndarray:
    """``:330``: ``affinity / np.abs(affinity.max(axis=1, keepdims=True))``."""
    with np.errstate(divide="ignore", invalid="ignore"):
        return affinity / np.abs(affinity.max(axis=1, keepdims=True))


def count_from_projection_dir(projection_dir, n_super_clusters: int, n_sub_clusters: int, graph_partition: str = "community",
                              binary_threshold: int = 100, frame_sampling_interval: int = 1, area_normalize: bool = False,
                              visible_img_prefix: str = "visible_cluster", wo_occ_img_prefix: str = "wo_occ_cluster",
                              device="cuda", seed: int = 35):
    """``main`` (``:299-395``) without the viewers: (total count, per-super-cluster counts, labels shifted to be unique,
    affinities).  ``seed``: the community partition draws from networkx's random state."""
    import os

    total, counts, labels_all, affinities = 0, [], [], []
    for i in range(n_super_clusters):
        wo, vis, lab = merger.load_projection_tree(os.path.join(str(projection_dir), f"super_cluster_{i}"), n_sub_clusters,
                                                   visible_img_prefix, wo_occ_img_prefix)
        prop = merger.process_super_cluster(wo, vis, lab, binary_threshold, frame_sampling_interval, area_normalize, device,
                                            plain_reliability="overlap")
        aff = calc_affinity(prop)
        np.random.seed(seed)
        import random

        random.seed(seed)
        n, labels = get_component(normalise_affinity(aff), graph_partition)
        counts.append(n)
        labels_all.append(np.asarray(labels) + total)
        affinities.append(aff)
        total += n
    return total, counts, labels_all, affinities


def main(argv=None) -> int:
    """``python -m cropnerf_amd.segmentation.depth_projection_based_merger --base_dir D --recording_name R`` -- the reference's
    arguments (``:299-316``); reads ``D/artifacts/R/depth_projection`` and ``D/artifacts/R/pcd/all_super_cluster_info.npy``."""
    import argparse
    import os

    ap = argparse.ArgumentParser()
    ap.add_argument("--base_dir", type=str, required=True)
    ap.add_argument("--recording_name", type=str, required=True)
    ap.add_argument("--visible_img_prefix", type=str, default="visible_cluster")
    ap.add_argument("--wo_occ_img_prefix", type=str, default="wo_occ_cluster")
    ap.add_argument("--area_normalize", type=lambda v: str(v).lower() in ("1", "true", "yes"), default=False)
    ap.add_argument("--graph_partition", type=str, default="community")
    ap.add_argument("--super_cluster_idx", type=int, default=-1)
    ap.add_argument("--binary_threshold", type=int, default=100)
    ap.add_argument("--frame_sampling_interval", type=int, default=1)
    a = ap.parse_args(argv)
    projection_dir = os.path.join(a.base_dir, "artifacts", a.recording_name, "depth_projection")
    pcd = np.load(os.path.join(a.base_dir, "artifacts", a.recording_name, "pcd", "all_super_cluster_info.npy"), allow_pickle=True)
    if a.super_cluster_idx != -1:
        raise SystemExit("a single --super_cluster_idx: rename that directory to super_cluster_0 or use the Python API")
    n_sc, k = min(4, len(pcd)), pcd[0]["aabb"].shape[0]  # :355
    total, counts, _, _ = count_from_projection_dir(projection_dir, n_sc, k, a.graph_partition, a.binary_threshold,
                                                    a.frame_sampling_interval, a.area_normalize, a.visible_img_prefix,
                                                    a.wo_occ_img_prefix)
    for i, n in enumerate(counts):
        print(f"{i}_th super cluster has: {n}.")
    print(f"Total bool: {total}")  # sic (:377)
    return total


if __name__ == "__main__":
    main()
