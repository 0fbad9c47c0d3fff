#!/usr/bin/env python
"""Golden vectors produced by EXECUTING the reference's own functions (build container only).

    python tests/golden/make_golden_reference.py      ->  tests/golden/reference_functions.npz

Most of the reference cannot be imported here (its modules import nerfstudio / open3d / cv2 at the top, and
``depth_based_semantic_projection.py`` runs a hard-coded job at import time).  A handful of its functions on or next to
the hot path are nevertheless pure numpy / torch / networkx code.  This script reads the reference's source files with
``ast``, takes the *definitions* of exactly those functions out of the module, executes them unchanged in a namespace
that holds only the libraries they use, and runs them on seeded inputs:

    fruit_nerf/scripts/depth_based_semantic_projection.py   get_projection_mat :31-43, get_projection :45-49,
                                                            update_buffer :84-105
    fruit_nerf/data/fruit_datamanager.py                    get_corners_of_aabb :42-69, sample_surface_points :71-121
    segmentation/merger.py                                  get_component :26-74, calc_affinity :335-355
                                                            (with segmentation/lpa.py, which imports as is)
    fruit_nerf/components/ray_samplers.py                   UniformSamplerWithNoise.generate_ray_samples :54-104 (round 3): the
                                                            method's definition, run as a plain function on an object that holds
                                                            the five attributes its constructor sets (:39-51) and a ray bundle
                                                            whose get_ray_samples() RECORDS its keyword arguments -- the bins,
                                                            the jitter arithmetic and spacing_to_euclidean_fn are the reference's
    fruit_nerf/components/ray_generators.py                 class OrthographicRayGenerator :22-66 (round 3): the class definition,
                                                            on torch's real nn.Module, with RayBundle a record of its keywords

    fruit_nerf/fruit_nerf.py                                update_schedule :144-149 and set_anneal / bias :206-216 (round 4): the two
                                                            nested closures of the hot path that are the reference's own code, executed
                                                            on an object that holds the config fields they read

    (round 5) STATEMENT BLOCKS of functions that cannot run as a whole (they need a trained pipeline, progress bars, open3d):
    the torch / numpy statements between two source lines, compiled from the function's own AST and executed on seeded
    tensors in a namespace that holds only what they name --
    fruit_nerf/data/cotton_nerf_dataparser.py               :248-254   the class list and colour table of `Semantics`
    fruit_nerf/fruit_nerf.py                                :594-597   get_outputs: sigmoid -> heaviside(. - 0.9, 0) -> colormap
                                                            :488-492   get_export_outputs: the per-sample label
                                                            :178, :603-608  the BCE loss object and get_loss_dict's two terms
                                                                        (`self.rgb_loss` = upstream's alias of torch.nn.MSELoss)
    fruit_nerf/export/exporter_utils.py                     :96-153    sample_volume's per-call masks, gathers and colours, with
                                                                        a pipeline object that hands back seeded model outputs
    fruit_nerf/export/exporter_utils_nerfacto.py            :156-180   generate_point_cloud: point = o + d * depth, fruit mask
                                                            :221-225   the re-orientation of the normals (o3d's Vector3dVector is
                                                                        a container: the identity here)
    segmentation/segmenter.py                               cluster_kmeans :28-45 (scikit-learn IS installed), on an object with
                                                            a `points` array
    segmentation/depth_projection_based_merger.py           get_component :23-61, calc_affinity :275-297 (whole functions)

Nothing of the reference is copied into the repository: the fixture holds inputs and outputs only.  The oracle
(``oracle/zbuffer.py``, ``oracle/rays.py``), the host mirrors (``cropnerf_amd/segmentation/merger.py``,
``fruit_nerf/data/fruit_datamanager.py``) and the HIP kernels (``cn_depth_project``, ``cn_zbuffer_update*``,
``cn_surface_grid``) are tested against it (``tests/test_reference_golden.py``).
"""

import ast
import os
import random
import sys

import networkx as nx
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/crop_nerf"


def extract(path, names, namespace):
    """exec the FunctionDef nodes `names` of the module at `path` (and nothing else of it) inside `namespace`."""
    with open(path, encoding="utf-8") as f:
        tree = ast.parse(f.read(), filename=path)
    found = []
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            mod = ast.Module(body=[node], type_ignores=[])
            exec(compile(mod, path, "exec"), namespace)
            found.append(node.name)
    missing = set(names) - set(found)
    if missing:
        raise RuntimeError(f"{path}: functions not found: {sorted(missing)}")
    return namespace


def extract_class(path, class_name, namespace, method=None):
    """exec the ClassDef `class_name` of the module at `path` inside `namespace`; with `method`, only that method's
    FunctionDef, as a module-level function."""
    with open(path, encoding="utf-8") as f:
        tree = ast.parse(f.read(), filename=path)
    for node in tree.body:
        if isinstance(node, ast.ClassDef) and node.name == class_name:
            if method is None:
                exec(compile(ast.Module(body=[node], type_ignores=[]), path, "exec"), namespace)
                return namespace
            for sub in node.body:
                if isinstance(sub, ast.FunctionDef) and sub.name == method:
                    exec(compile(ast.Module(body=[sub], type_ignores=[]), path, "exec"), namespace)
                    return namespace
    raise RuntimeError(f"{path}: {class_name}.{method} not found")


class Record:
    """A container of keyword arguments (stands where nerfstudio's RayBundle / RaySamples dataclasses would: no arithmetic)."""

    def __init__(self, **kw):
        self.__dict__.update(kw)


def orbit_c2w(rng):
    """A camera-to-world matrix looking roughly at the origin from a random direction (float64, like the reference's)."""
    d = rng.normal(size=3)
    eye = d / np.linalg.norm(d) * rng.uniform(0.6, 1.2)
    z = eye / np.linalg.norm(eye)  # camera looks down -z
    up = np.array([0.0, 0.0, 1.0])
    x = np.cross(up, z)
    x /= np.linalg.norm(x)
    y = np.cross(z, x)
    c2w = np.eye(4)
    c2w[:3, 0], c2w[:3, 1], c2w[:3, 2], c2w[:3, 3] = x, y, z, eye
    return c2w


def sparse(img):
    idx = np.argwhere(img != 0)
    return idx.astype(np.int32), img[img != 0]


def projection_cases(out):
    ns = extract(f"{REF}/fruit_nerf/scripts/depth_based_semantic_projection.py",
                 {"get_projection_mat", "get_projection", "update_buffer"}, {"np": np})
    rng = np.random.default_rng(20)
    fx = fy = 1442.4757
    cx, cy = 961.9397, 723.2478  # fruit_nerf/utils/transforms.json
    for case in range(3):
        c2w = orbit_c2w(rng)
        P = ns["get_projection_mat"](fx, fy, cx, cy, c2w)
        # three clouds: a wide "tree" (large=True splat), then two compact clusters tested against it (large=False)
        # (coordinates on a 2^-12 lattice: exactly representable, and the fixture compresses)
        q = lambda a: np.round(a * 4096.0) / 4096.0
        tree = q(rng.normal(size=(2500, 3)) * 0.25)
        c1 = q(rng.normal(size=(600, 3)) * 0.03 + rng.normal(size=3) * 0.15)
        c2 = q(rng.normal(size=(600, 3)) * 0.03 + rng.normal(size=3) * 0.15)
        z_buffer = np.ones((1440, 1920), dtype=np.float32) * 1e10
        img = np.zeros((1440, 1920), dtype=np.uint8)
        k = f"proj{case}"
        out[f"{k}/c2w"], out[f"{k}/intrinsics"], out[f"{k}/P"] = c2w, np.array([fx, fy, cx, cy]), P
        for name, pts, label, large in (("tree", tree, 60, True), ("c1", c1, 120, False), ("c2", c2, 200, False)):
            im = ns["get_projection"](P, pts)
            z_buffer, img, (vx, vy) = ns["update_buffer"](z_buffer, im, img, label, large)
            out[f"{k}/{name}/points"], out[f"{k}/{name}/im"] = pts, im
            out[f"{k}/{name}/label"], out[f"{k}/{name}/large"] = np.array(label), np.array(large)
            out[f"{k}/{name}/visible_xs"], out[f"{k}/{name}/visible_ys"] = np.asarray(vx), np.asarray(vy)
            idx, val = sparse(img)
            out[f"{k}/{name}/img_idx"], out[f"{k}/{name}/img_val"] = idx, val
            zi = np.argwhere(z_buffer < 1e9).astype(np.int32)
            out[f"{k}/{name}/z_idx"], out[f"{k}/{name}/z_val"] = zi, z_buffer[z_buffer < 1e9]
    out["num_proj"] = np.array(3)


def datamanager_cases(out):
    ns = extract(f"{REF}/fruit_nerf/data/fruit_datamanager.py", {"get_corners_of_aabb", "sample_surface_points"},
                 {"torch": torch})
    cases = [
        (torch.tensor([[-1.0, -1.0, -0.682], [1.0, 1.0, 1.318]]), 4),  # SURVEY 8(c) KAT 11
        (torch.tensor([[-1.0, -1.0, -1.0], [1.0, 1.0, 1.0]]), 7),
        (torch.tensor([[-0.5, -0.75, -0.25], [0.5, 0.25, 0.75]]), 6),
        (torch.tensor([[-0.3, -0.2, 0.1], [0.4, 0.6, 0.9]]), 9),
    ]
    for i, (aabb, n) in enumerate(cases):
        corners = ns["get_corners_of_aabb"](aabb, "cpu")
        pts, plane = ns["sample_surface_points"](corners, n, "cpu")
        out[f"dm{i}/aabb"], out[f"dm{i}/n"] = aabb.numpy(), np.array(n)
        out[f"dm{i}/corners"], out[f"dm{i}/points"], out[f"dm{i}/plane"] = corners.numpy(), pts.numpy(), plane.numpy()
    out["num_dm"] = np.array(len(cases))


def merger_cases(out):
    import matplotlib

    matplotlib.use("Agg")
    from matplotlib import cm

    sys.path.insert(0, f"{REF}/segmentation")
    import lpa  # the reference's own module (networkx only)

    cmap = matplotlib.colors.ListedColormap(cm.tab20.colors + cm.tab20c.colors, name="tab40")  # merger.py:20
    ns = extract(f"{REF}/segmentation/merger.py", {"get_component", "calc_affinity"},
                 {"np": np, "nx": nx, "lpa": lpa, "cmap": cmap, "plt": None})
    rng = np.random.default_rng(21)
    case = 0
    for n_clusters in (2, 3, 4, 6, 9):
        for n_cams in (5, 24):
            prop = {}
            for i in range(n_clusters):
                label = rng.integers(0, 4, size=n_cams)  # 0 = not seen
                reliability = rng.uniform(0, 1, size=n_cams) * (label != 0)
                prop[i] = {"label": label, "reliability": reliability}
            aff = ns["calc_affinity"](prop)
            out[f"mg{case}/labels"] = np.stack([prop[i]["label"] for i in range(n_clusters)])
            out[f"mg{case}/reliability"] = np.stack([prop[i]["reliability"] for i in range(n_clusters)])
            out[f"mg{case}/affinity"] = aff
            for algo in ("clique", "bridge", "community"):
                random.seed(35)  # merger.py:23 graph_seed; lpa draws from `random`
                count, labels = ns["get_component"](aff.copy(), algo)
                out[f"mg{case}/{algo}/count"], out[f"mg{case}/{algo}/labels"] = np.array(count), np.asarray(labels)
            case += 1
    out["num_mg"] = np.array(case)


def sampler_cases(out):
    """UniformSamplerWithNoise.generate_ray_samples (components/ray_samplers.py:54-104) in eval and in training with both
    jitter kinds.  torch.rand is seeded right before the call; the same seed regenerates the jitter for the fixture."""
    from types import SimpleNamespace
    from typing import Optional

    ns = extract_class(f"{REF}/fruit_nerf/components/ray_samplers.py", "UniformSamplerWithNoise",
                       {"torch": torch, "Optional": Optional, "RayBundle": Record, "RaySamples": Record},
                       method="generate_ray_samples")
    fn = ns["generate_ray_samples"]
    g = torch.Generator().manual_seed(40)
    case = 0
    for num_samples, num_rays, training, single in ((64, 7, False, False), (192, 5, False, False), (3000, 3, False, False),
                                                    (48, 9, True, True), (48, 9, True, False), (100, 4, True, True)):
        nears = torch.rand(num_rays, 1, generator=g) * 0.5
        fars = nears + 0.5 + torch.rand(num_rays, 1, generator=g) * 2.0
        rec = {}

        def get_ray_samples(**kw):
            rec.update(kw)
            return Record(**kw)

        bundle = Record(origins=torch.zeros(num_rays, 3), nears=nears, fars=fars, get_ray_samples=get_ray_samples)
        # what UniformSamplerWithNoise.__init__ (:39-51) hands to SpacedSampler: identity spacing functions
        self_ = SimpleNamespace(num_samples=None, train_stratified=True, single_jitter=single, training=training,
                                spacing_fn=lambda x: x, spacing_fn_inv=lambda x: x)
        seed = 1000 + case
        torch.manual_seed(seed)
        fn(self_, bundle, num_samples)
        torch.manual_seed(seed)
        t_rand = torch.rand((num_rays, 1 if single else num_samples + 1)) if training else torch.zeros(0)
        k = f"us{case}"
        out[f"{k}/num_samples"], out[f"{k}/training"], out[f"{k}/single_jitter"] = np.array(num_samples), np.array(training), np.array(single)
        out[f"{k}/nears"], out[f"{k}/fars"], out[f"{k}/t_rand"] = nears.numpy(), fars.numpy(), t_rand.numpy()
        for name in ("bin_starts", "bin_ends", "spacing_starts", "spacing_ends"):
            v = rec[name]
            out[f"{k}/{name}"] = v.expand(num_rays, num_samples, 1).numpy().copy()
        x = torch.linspace(0, 1, 5)[None, :].expand(num_rays, 5)
        out[f"{k}/s2e_at_quarters"] = rec["spacing_to_euclidean_fn"](x).numpy()
        case += 1
    out["num_us"] = np.array(case)


def ortho_cases(out):
    """OrthographicRayGenerator (components/ray_generators.py:22-66): constructor + forward(count) for first, middle and
    ragged last batches."""
    from torch import Tensor, nn

    ns = extract_class(f"{REF}/fruit_nerf/components/ray_generators.py", "OrthographicRayGenerator",
                       {"torch": torch, "nn": nn, "Tensor": Tensor, "RayBundle": Record})
    cls = ns["OrthographicRayGenerator"]
    g = torch.Generator().manual_seed(41)
    case = 0
    for n_pts, batch, plane in ((37, 8, [[0.0, 0.0, 2.0]]), (64, 16, [[0.0, 1.5, 0.0]]), (10, 512, [[0.3, -0.4, 1.2]])):
        pts = torch.rand(n_pts, 3, generator=g) * 2 - 1
        plane_t = torch.tensor(plane)
        gen = cls(pts, plane_t, batch, "cpu", None)
        counts = sorted({1, 2, (n_pts + batch - 1) // batch})
        for count in counts:
            if batch * (count - 1) >= n_pts:
                continue
            rb = gen(count)
            k = f"og{case}"
            out[f"{k}/points"], out[f"{k}/plane"], out[f"{k}/batch"], out[f"{k}/count"] = pts.numpy(), plane_t.numpy(), np.array(batch), np.array(count)
            for name in ("origins", "directions", "pixel_area", "nears", "fars"):
                out[f"{k}/{name}"] = getattr(rb, name).numpy()
            case += 1
    out["num_og"] = np.array(case)


def extract_nested(path, class_name, method, inner, namespace):
    """exec the FunctionDef `inner` that is nested (at any depth) inside `class_name.method` of the module at `path`, as a
    module-level function of `namespace` -- the names it closes over (`self`, `N`, `np`) are then looked up there."""
    with open(path, encoding="utf-8") as f:
        tree = ast.parse(f.read(), filename=path)
    for node in tree.body:
        if isinstance(node, ast.ClassDef) and node.name == class_name:
            for sub in node.body:
                if isinstance(sub, ast.FunctionDef) and sub.name == method:
                    for n in ast.walk(sub):
                        if isinstance(n, ast.FunctionDef) and n.name == inner:
                            exec(compile(ast.Module(body=[n], type_ignores=[]), path, "exec"), namespace)
                            return namespace
    raise RuntimeError(f"{path}: {class_name}.{method}.{inner} not found")


def schedule_cases(out):
    """The two closures of the hot path that are the reference's own code (fruit_nerf/fruit_nerf.py): `update_schedule`
    (:144-149, nested in populate_modules: after how many steps the proposal networks get a gradient) and `set_anneal` with its
    inner `bias` (:206-216, nested in get_training_callbacks: the proposal-weight annealing exponent, arXiv 2111.12077 eq. 18).
    Executed unchanged on an object that holds the config fields they read (nerfacto's defaults) and a sampler that records
    what `set_anneal` hands it."""
    path = f"{REF}/fruit_nerf/fruit_nerf.py"
    cfg = Record(proposal_warmup=5000, proposal_update_every=5, proposal_weights_anneal_slope=10.0,
                 proposal_weights_anneal_max_num_iters=1000)
    seen = []
    fake = Record(config=cfg, proposal_sampler=Record(set_anneal=lambda a: seen.append(float(a))), step=0)
    ns = {"np": np, "self": fake}
    extract_nested(path, "FruitModel", "populate_modules", "update_schedule", ns)
    steps = np.array([0, 1, 9, 10, 999, 1000, 1001, 2499, 2500, 4999, 5000, 5001, 39999], dtype=np.int64)
    out["sched_steps"] = steps
    out["sched_values"] = np.array([float(ns["update_schedule"](int(s))) for s in steps], dtype=np.float64)
    out["sched_config"] = np.array([cfg.proposal_warmup, cfg.proposal_update_every], dtype=np.float64)
    ns2 = {"np": np, "self": fake, "N": cfg.proposal_weights_anneal_max_num_iters}
    extract_nested(path, "FruitModel", "get_training_callbacks", "set_anneal", ns2)
    asteps = np.array([0, 1, 2, 10, 100, 250, 500, 999, 1000, 1001, 5000], dtype=np.int64)
    for s in asteps:
        ns2["set_anneal"](int(s))
    assert len(seen) == len(asteps)
    out["anneal_steps"] = asteps
    out["anneal_values"] = np.array(seen, dtype=np.float64)
    out["anneal_config"] = np.array([cfg.proposal_weights_anneal_slope, cfg.proposal_weights_anneal_max_num_iters], dtype=np.float64)


def statement_block(path, first, last):
    """A code object of the statements of the module at `path` that lie between source lines `first` and `last` (inclusive),
    taken from the innermost statement list that holds them -- the reference's own AST nodes, nothing edited."""
    with open(path, encoding="utf-8") as f:
        tree = ast.parse(f.read(), filename=path)

    def pick(body):
        sel = [n for n in body if first <= n.lineno and n.end_lineno <= last]
        if sel:
            return sel
        for n in body:
            if n.lineno <= first and last <= n.end_lineno:
                for field in ("body", "orelse", "finalbody"):
                    sub = getattr(n, field, None)
                    if isinstance(sub, list) and sub and isinstance(sub[0], ast.stmt):
                        got = pick(sub)
                        if got:
                            return got
        return []

    sel = pick(tree.body)
    if not sel:
        raise RuntimeError(f"{path}: no statements between lines {first} and {last}")
    return compile(ast.Module(body=sel, type_ignores=[]), path, "exec")


def colormap_cases(out):
    """a13: `Semantics.colors` as the dataparser builds it (:248-254) and the colormap statements of get_outputs (:594-597) and
    get_export_outputs (:488-492) on seeded logits -- including the threshold logit ln 9 itself and its float32 neighbours."""
    ns = {"torch": torch}
    exec(statement_block(f"{REF}/fruit_nerf/data/cotton_nerf_dataparser.py", 248, 254), ns)
    colors = ns["colors"]
    g = torch.Generator().manual_seed(31)
    ln9 = torch.tensor(9.0).log()
    edge = torch.stack([ln9, torch.nextafter(ln9, torch.tensor(10.0)), torch.nextafter(ln9, torch.tensor(0.0)),
                        ln9 + 1e-4, ln9 - 1e-4, torch.tensor(0.0), torch.tensor(-30.0), torch.tensor(30.0)])
    logits = torch.cat([edge, torch.randn(248, generator=g) * 3.0 + 2.0])[:, None]
    ns2 = {"torch": torch, "outputs": {"semantics": logits.clone()}, "self": Record(colormap=colors.clone(), device="cpu")}
    exec(statement_block(f"{REF}/fruit_nerf/fruit_nerf.py", 594, 597), ns2)
    out["cm_classes"] = np.array(ns["classes"])
    out["cm_colors"] = colors.numpy()
    out["cm_logits"] = logits.numpy()
    out["cm_colormap"] = ns2["outputs"]["semantics_colormap"].numpy()
    ns3 = {"torch": torch, "outputs": {"semantics": torch.randn(16, 24, generator=g) * 3.0 + 2.0}}
    out["cm_export_logits"] = ns3["outputs"]["semantics"].numpy().copy()
    exec(statement_block(f"{REF}/fruit_nerf/fruit_nerf.py", 488, 492), ns3)
    out["cm_export_labels"] = ns3["outputs"]["semantics_colormap"].numpy()


def loss_cases(out):
    """a18: the loss object the reference constructs itself (:178) and the two data terms of get_loss_dict (:603-608; the
    interlevel term and the camera optimiser's are upstream code).  The image carries an alpha channel: `image[:, :3]`."""
    g = torch.Generator().manual_seed(32)
    R = 192
    fake = Record(config=Record(semantic_loss_weight=1.0), device="cpu", rgb_loss=torch.nn.MSELoss())
    exec(statement_block(f"{REF}/fruit_nerf/fruit_nerf.py", 178, 178), {"torch": torch, "self": fake})
    batch = {"image": torch.rand(R, 4, generator=g), "fruit_mask": (torch.rand(R, 1, generator=g) > 0.6).float()}
    outputs = {"rgb": torch.rand(R, 3, generator=g), "semantics": torch.randn(R, 1, generator=g) * 4.0}
    ns = {"torch": torch, "self": fake, "batch": batch, "outputs": outputs, "loss_dict": {}}
    exec(statement_block(f"{REF}/fruit_nerf/fruit_nerf.py", 603, 608), ns)
    out["loss_image"], out["loss_mask"] = batch["image"].numpy(), batch["fruit_mask"].numpy()
    out["loss_rgb"], out["loss_sem"] = outputs["rgb"].numpy(), outputs["semantics"].numpy()
    out["loss_values"] = np.array([float(ns["loss_dict"]["rgb_loss"]), float(ns["loss_dict"]["semantics_loss"])], dtype=np.float64)


def sample_volume_cases(out):
    """a16: one pass of sample_volume's loop body (:96-153) -- thresholds `semantic >= 3`, `density >= 70`,
    `semantics_colormap >= 0.999`, the three point sets and their four-channel colours -- on seeded model outputs with values ON
    the thresholds."""
    g = torch.Generator().manual_seed(33)
    R, S = 24, 40
    sem = torch.randn(R, S, generator=g) * 3.0 + 1.5
    den = torch.rand(R, S, generator=g) * 140.0
    sem[0, :4] = torch.tensor([3.0, 2.9999998, 3.0000002, 3.0])
    den[0, :4] = torch.tensor([70.0, 70.0, 69.99999, 70.00001])
    lab = torch.heaviside(torch.sigmoid(sem) - 0.9, torch.tensor(0.0)).to(torch.long)  # (what get_export_outputs hands on)
    outputs = {"point_location": torch.rand(R, S, 3, generator=g) * 2 - 1, "semantics": sem, "semantics_colormap": lab,
               "density": den, "rgb": torch.rand(R, S, 3, generator=g)}
    pipeline = Record(datamanager=Record(next_sample_volume=lambda step: (Record(), None)), model=lambda rb: outputs)
    lists = {k: [] for k in ("points_sem", "points_only_sem", "points_den", "points_sem_colormap", "color_semantics",
                             "color_only_semantics", "color_semantics_colormap", "densities")}
    ns = {"torch": torch, "pipeline": pipeline, "rgb_flag": True, **lists}
    exec(statement_block(f"{REF}/fruit_nerf/export/exporter_utils.py", 96, 153), ns)
    for k in ("point_location", "semantics", "semantics_colormap", "density", "rgb"):
        out[f"sv_in_{k}"] = outputs[k].numpy()
    for name, pk, ck in (("semantic_colormap", "points_sem_colormap", "color_semantics_colormap"),
                         ("semantic", "points_sem", "color_semantics"), ("density", "points_den", "densities")):
        assert len(ns[pk]) == 1 and len(ns[ck]) == 1
        out[f"sv_{name}_points"] = ns[pk][0].numpy()
        out[f"sv_{name}_colors"] = ns[ck][0].numpy()


def pointcloud_cases(out):
    """a17: generate_point_cloud's per-call statements (:156-180: point = origin + direction * depth, kept where
    `semantics_colormap[:, 0] > 0`; no crop box, no model normals) and the re-orientation of the normals (:221-225)."""
    g = torch.Generator().manual_seed(34)
    R = 300
    d = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1)
    rb = Record(origins=torch.randn(R, 3, generator=g) * 0.3, directions=d)
    depth = torch.rand(R, 1, generator=g) * 2.0 + 0.1
    cmap = (torch.rand(R, 1, generator=g) > 0.7).float().repeat(1, 3)
    rgba = torch.rand(R, 4, generator=g)
    lists = {"points": [], "rgbs": [], "view_directions": [], "normals": []}
    ns = {"torch": torch, "ray_bundle": rb, "depth": depth, "outputs": {"semantics_colormap": cmap}, "rgba": rgba,
          "only_semantics": True, "normal": None, "crop_obb": None, "progress": Record(advance=lambda *a: None), "task": 0, **lists}
    exec(statement_block(f"{REF}/fruit_nerf/export/exporter_utils_nerfacto.py", 156, 180), ns)
    out["pc_origins"], out["pc_directions"], out["pc_depth"] = rb.origins.numpy(), d.numpy(), depth.numpy()
    out["pc_colormap"], out["pc_rgba"] = cmap.numpy(), rgba.numpy()
    out["pc_points"], out["pc_rgbs"] = ns["points"][0].numpy(), ns["rgbs"][0].numpy()
    out["pc_view_directions"] = ns["view_directions"][0].numpy()
    # re-orientation: unit normals (float64, as open3d holds them) against the kept view directions
    n = ns["points"][0].shape[0]
    nrm = torch.nn.functional.normalize(torch.randn(n, 3, generator=g, dtype=torch.float64), dim=-1).numpy()
    nrm[:3] = ns["view_directions"][0][:3].double().numpy() * np.array([[1.0], [-1.0], [0.0]])  # along, against, zero
    pcd = Record(normals=nrm.copy())
    o3d = Record(utility=Record(Vector3dVector=lambda a: a))
    ns2 = {"torch": torch, "np": np, "o3d": o3d, "pcd": pcd, "reorient_normals": True,
           "view_directions": ns["view_directions"][0].clone()}
    exec(statement_block(f"{REF}/fruit_nerf/export/exporter_utils_nerfacto.py", 221, 225), ns2)
    out["pc_normals_in"] = nrm
    out["pc_normals_out"] = np.asarray(pcd.normals)


def kmeans_cases(out):
    """f2: cluster_kmeans (segmentation/segmenter.py:28-45; consider_normals=False) -- the KMeans configuration the segmenter
    splits every super-cluster with -- on three seeded clouds."""
    from sklearn.cluster import KMeans

    ns = extract(f"{REF}/segmentation/segmenter.py", {"cluster_kmeans"}, {"np": np, "KMeans": KMeans})
    rng = np.random.default_rng(35)
    case = 0
    for k, n in ((2, 400), (3, 900), (5, 2500)):
        centres = rng.uniform(-0.5, 0.5, size=(k, 3))
        pts = np.concatenate([c + rng.normal(scale=0.04 + 0.02 * j, size=(n // k, 3)) for j, c in enumerate(centres)])
        labels = ns["cluster_kmeans"](Record(points=pts), k=k)
        out[f"km_{case}_points"], out[f"km_{case}_k"], out[f"km_{case}_labels"] = pts, np.array(k), np.asarray(labels)
        case += 1
    out["num_km"] = np.array(case)


def depth_merger_cases(out):
    """f3/f4: get_component of the depth-projection merger (segmentation/depth_projection_based_merger.py:23-61 -- unlike
    merger.py's it keeps the edge WEIGHTS for the clique / bridge partitions and uses networkx's own label propagation) on
    row-normalised affinities as its main() builds them (:330), and calc_affinity (:275-297) on seeded cluster properties."""
    path = f"{REF}/segmentation/depth_projection_based_merger.py"
    ns = extract(path, {"get_component", "calc_affinity"}, {"np": np, "nx": nx})
    rng = np.random.default_rng(36)
    case = 0
    for n in (3, 5, 8):
        for rep in range(3):
            # sub-clusters of two or three fruits: every sub-cluster mostly carries its fruit's label, sometimes a neighbour's or
            # background (0), so that every row of the affinity has a positive maximum, as real projections give
            fruit = rng.integers(1, 4, size=n)
            fruit[:2] = (1, 1)
            labels = np.where(rng.uniform(size=(n, 12)) < 0.75, fruit[:, None], rng.integers(0, 4, size=(n, 12))).astype(np.float64)
            rel = rng.uniform(0.2, 1.0, size=(n, 12))
            prop = {i: {"label": labels[i], "reliability": rel[i]} for i in range(n)}
            aff = ns["calc_affinity"](prop)
            with np.errstate(divide="ignore", invalid="ignore"):
                norm = aff / np.abs(aff.max(axis=1, keepdims=True))
            if not np.isfinite(norm).all():
                continue
            out[f"dpm{case}/labels"], out[f"dpm{case}/reliability"], out[f"dpm{case}/affinity"] = labels, rel, aff
            for algo in ("clique", "bridge", "community"):
                random.seed(100 + case)
                k, lab = ns["get_component"](norm.copy(), algo)
                out[f"dpm{case}/{algo}_count"] = np.array(k)
                out[f"dpm{case}/{algo}_labels"] = np.asarray(lab)
            case += 1
    out["num_dpm"] = np.array(case)


def main():
    out = {}
    projection_cases(out)
    datamanager_cases(out)
    merger_cases(out)
    sampler_cases(out)
    ortho_cases(out)
    schedule_cases(out)
    colormap_cases(out)
    loss_cases(out)
    sample_volume_cases(out)
    pointcloud_cases(out)
    kmeans_cases(out)
    depth_merger_cases(out)
    path = os.path.join(HERE, "reference_functions.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes", len(out), "arrays")


if __name__ == "__main__":
    main()
