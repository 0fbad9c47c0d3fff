"""The merger -- mirror of ``crop_nerf/segmentation/merger.py``.

Image stage (``:219-333``, ``process_super_cluster``): per sub-cluster and camera, the area / bounding box of the
un-occluded projection, and inside that box the "area" and majority instance label of the visible projection -- OpenCV
contour calls on PNG files in the reference, ``cn_contour_largest`` (``csrc/contour.hip``) on a stack of images here: the
a15 projections can go in straight from device memory (``process_super_cluster``), or from the reference's PNG tree
(``load_projection_tree``).  OpenCV's semantics are restated in ``oracle/contours.py`` (unpinned: no OpenCV here).

Graph stage (``:335-355`` ``calc_affinity``, ``:26-74`` ``get_component``): from the per-camera label and reliability of every
sub-cluster of one super-cluster to the affinity matrix, its partition (maximal cliques / bridge removal / signed label
propagation) and the fruit count.  Host code on a handful of nodes; networkx as upstream; pinned by vectors the reference's
own functions produced (``tests/golden/reference_functions.npz``)."""

from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import numpy as np

from . import label_propagation

EPS = 1e-6  # merger.py:22


def quantise_projection(img) -> "torch.Tensor":
    """What ``torchvision.utils.save_image`` + ``cv2.imread`` + ``COLOR_BGR2GRAY`` leave of a projection image
    (``fruit_nerf.py:302-304,315``): ``x * 255 + 0.5`` clamped to [0, 255] as uint8 (the three channels are equal, and
    OpenCV's fixed-point gray conversion maps equal channels onto themselves).  [..., H, W] or [..., H, W, C] float."""
    import torch

    t = img if isinstance(img, torch.Tensor) else torch.as_tensor(np.asarray(img))
    if t.dtype == torch.uint8:
        return t
    if t.dim() >= 3 and t.shape[-1] in (1, 3):
        t = t[..., 0]
    return t.to(torch.float32).mul(255).add_(0.5).clamp_(0, 255).to(torch.uint8)


def process_super_cluster(wo_occ, visible, label_frames, binary_thresh: int = 100, frame_sampling_interval: int = 10,
                          area_normalize: bool = False, device="cuda", plain_reliability: str = "ones"
                          ) -> Dict[int, Dict[str, np.ndarray]]:
    """``process_super_cluster`` (``:273-333``) on arrays: ``wo_occ`` / ``visible`` [n_cams, k, H, W] (uint8 gray, or the
    float projection images, quantised as the PNG round trip does), ``label_frames`` [n_cams, H, W] uint8 instance labels.
    Every ``frame_sampling_interval``-th camera is used (the reference samples its directory listing the same way).  Returns
    the reference's ``cluster_prop``: per sub-cluster ``visible_area``, ``wo_occ_area``, ``wo_occ_area_norm``, ``label``,
    ``label_overlap_area``, ``reliability`` (arrays over the cameras, ``eps`` / 0 where nothing was seen).
    ``plain_reliability``: the reliability without ``area_normalize`` -- ``"ones"`` (``merger.py:320``) or ``"overlap"``
    = label-overlap area / un-occluded area (``depth_projection_based_merger.py:263``)."""
    import torch

    from .. import ops

    wo = quantise_projection(wo_occ).to(device)
    vis = quantise_projection(visible).to(device)
    lab = (label_frames if isinstance(label_frames, torch.Tensor) else torch.as_tensor(np.asarray(label_frames))).to(
        device=device, dtype=torch.uint8).contiguous()
    n_cams, k, H, W = wo.shape
    cams = torch.arange(0, n_cams, frame_sampling_interval, device=device)
    wo_s = wo[cams].reshape(-1, H, W).contiguous()  # job j = (sampled camera j // k, sub-cluster j % k)
    vis_s = vis[cams].reshape(-1, H, W).contiguous()
    a = ops.contour_largest(wo_s, binary_thresh)
    seen = a["area"] >= 10  # get_wo_occlusion_projection_area: area < 10 -> eps, no box
    x, y, w, h = a["bbox"].unbind(1)
    roi = torch.stack([x, y, x + w, y + h], 1)
    roi = torch.where(seen[:, None], roi, torch.zeros_like(roi)).to(torch.int32).contiguous()  # empty box: nothing to find
    lidx = cams.repeat_interleave(k).to(torch.int32).contiguous()
    b = ops.contour_largest(vis_s, binary_thresh, roi=roi, labels=lab, label_index=lidx)
    vis_ok = seen & (b["vertex_count"] >= 10)  # get_visible_projection_area: no contour / fewer than 10 mask pixels -> eps, 0, eps
    wo_area = torch.where(seen, a["area"].double(), torch.full_like(a["area"], EPS, dtype=torch.float64))
    vis_area = torch.where(vis_ok, b["vertex_count"].double(), torch.full_like(wo_area, EPS))
    label = torch.where(vis_ok, b["label"], torch.zeros_like(b["label"]))
    overlap = torch.where(vis_ok, torch.where(label == 0, torch.zeros_like(wo_area), b["label_count"].double()),
                          torch.full_like(wo_area, EPS))
    shape = (cams.numel(), k)
    wo_area, vis_area, label, overlap = (t.reshape(shape).cpu().numpy() for t in (wo_area, vis_area, label, overlap))
    ci = cams.cpu().numpy()
    prop: Dict[int, Dict[str, np.ndarray]] = {}
    for cid in range(k):
        va, wa, oa, ol = EPS * np.ones(n_cams), EPS * np.ones(n_cams), EPS * np.ones(n_cams), np.zeros(n_cams)
        va[ci], wa[ci], oa[ci], ol[ci] = vis_area[:, cid], wo_area[:, cid], overlap[:, cid], label[:, cid]
        wn = wa / wa.max()
        rel = wn * (oa / wa) if area_normalize else (oa / wa if plain_reliability == "overlap" else np.ones_like(wa))
        prop[cid] = {"visible_area": va, "wo_occ_area": wa, "wo_occ_area_norm": wn, "label": ol, "label_overlap_area": oa,
                     "reliability": rel}
    return prop


def load_projection_tree(super_cluster_dir, n_sub_clusters: int, visible_img_prefix: str = "visible_cluster",
                         wo_occ_img_prefix: str = "wo_occ_cluster"):
    """The PNG tree ``get_outputs_for_projections`` writes (``fruit_nerf.py:268-316``) and the reference's merger reads:
    ``<dir>/cam_<j>/{wo_occ,visible}_cluster_<c>.png`` + ``<dir>/cam_<j>/label_frame*.png`` -> (wo_occ, visible
    [n_cams, k, H, W] uint8, label_frames [n_cams, H, W] uint8), cameras ordered by their number."""
    import glob
    import os

    from PIL import Image

    cams = sorted((int(os.path.basename(d).split("_")[-1]), d) for d in glob.glob(os.path.join(str(super_cluster_dir), "cam_*")))
    if not cams:
        raise FileNotFoundError(f"no cam_* directories under {super_cluster_dir}")

    def gray(path):
        return np.asarray(Image.open(path).convert("RGB"))[..., 0]

    wo, vis, lab = [], [], []
    for _, d in cams:
        wo.append(np.stack([gray(os.path.join(d, f"{wo_occ_img_prefix}_{c}.png")) for c in range(n_sub_clusters)]))
        vis.append(np.stack([gray(os.path.join(d, f"{visible_img_prefix}_{c}.png")) for c in range(n_sub_clusters)]))
        frames = glob.glob(os.path.join(d, "label_frame*.png"))
        lab.append(np.asarray(Image.open(frames[0]).convert("L")) if frames else np.zeros(wo[-1].shape[1:], np.uint8))
    return np.stack(wo), np.stack(vis), np.stack(lab)


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


def count_from_projection_dir(projection_dir, n_super_clusters: int, n_sub_clusters: int, graph_partition: str = "clique",
                              binary_threshold: int = 100, frame_sampling_interval: int = 10, area_normalize: bool = False,
                              visible_img_prefix: str = "visible_cluster", wo_occ_img_prefix: str = "wo_occ_cluster",
                              device="cuda"):
    """``main`` (``:359-437``) without the viewers: for every ``super_cluster_<i>`` directory of the projection tree, image
    stage -> affinity -> partition; returns (total count, per-super-cluster counts, labels shifted to be unique, affinities)."""
    import os
    import random

    total, counts, labels_all, affinities = 0, [], [], []
    for i in range(n_super_clusters):
        wo, vis, lab = load_projection_tree(os.path.join(str(projection_dir), f"super_cluster_{i}"), n_sub_clusters,
                                            visible_img_prefix, wo_occ_img_prefix)
        prop = process_super_cluster(wo, vis, lab, binary_threshold, frame_sampling_interval, area_normalize, device)
        aff = calc_affinity(prop)
        random.seed(35)  # merger.py:23 graph_seed (the community partition draws from `random`)
        n, labels = get_component(aff, graph_partition)
        counts.append(n)
        labels_all.append(np.asarray(labels) + total)
        affinities.append(aff)
        total += n
    return total, counts, labels_all, affinities


def main(argv=None) -> int:
    """``python -m cropnerf_amd.segmentation.merger --base_dir D --recording_name R`` -- the reference's arguments
    (``:359-378``); reads ``D/artifacts/R/projection`` and ``D/artifacts/R/pcd/all_super_cluster_info.npy``."""
    import argparse
    import os

    ap = argparse.ArgumentParser()
    ap.add_argument("--base_dir", type=str, required=True)
    ap.add_argument("--recording_name", type=str, required=True)
    ap.add_argument("--visible_img_prefix", type=str, default="visible_cluster")
    ap.add_argument("--wo_occ_img_prefix", type=str, default="wo_occ_cluster")
    ap.add_argument("--area_normalize", type=lambda v: str(v).lower() in ("1", "true", "yes"), default=False)
    ap.add_argument("--graph_partition", type=str, default="clique")
    ap.add_argument("--super_cluster_idx", type=int, default=-1)
    ap.add_argument("--binary_threshold", type=int, default=100)
    ap.add_argument("--frame_sampling_interval", type=int, default=10)
    a = ap.parse_args(argv)
    projection_dir = os.path.join(a.base_dir, "artifacts", a.recording_name, "projection")
    pcd = np.load(os.path.join(a.base_dir, "artifacts", a.recording_name, "pcd", "all_super_cluster_info.npy"), allow_pickle=True)
    n_sc, k = (min(17, len(pcd)) if a.super_cluster_idx == -1 else 1), pcd[0]["aabb"].shape[0]  # :398-399,421
    if a.super_cluster_idx != -1:
        raise SystemExit("a single --super_cluster_idx: rename that directory to super_cluster_0 or use the Python API")
    total, counts, _, _ = count_from_projection_dir(projection_dir, n_sc, k, a.graph_partition, a.binary_threshold,
                                                    a.frame_sampling_interval, a.area_normalize, a.visible_img_prefix,
                                                    a.wo_occ_img_prefix)
    for i, n in enumerate(counts):
        print(f"{i}_th super cluster has: {n}.")
    print(f"Total bool: {total}")  # sic (:437)
    return total


if __name__ == "__main__":
    main()
