"""P-fit: train the fruit_nerf method on the analytic plant with the HIP trainer, then report PSNR and
fruit-mask IoU of held-out views against the closed-form ground truth.  With --oracle the trained parameters are also
rendered by the CPU oracle at low resolution (PSNR HIP vs oracle, and both against ground truth).

    python tools/fit_scene.py [--iters 2000] [--res 200] [--oracle] [--save run_dir]
"""
import argparse, json, os, sys, time
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cropnerf_amd import config as PC, synthetic
from cropnerf_amd.fruit_nerf.data.fruit_datamanager import FruitDataManagerConfig
from cropnerf_amd.fruit_nerf.fruit_pipeline import FruitPipeline, FruitPipelineConfig
from cropnerf_amd.fruit_nerf.trainer import FruitTrainer
from cropnerf_amd.rays import Cameras, SceneBox


def psnr(a, b):
    return float(-10.0 * torch.log10(((a - b) ** 2).mean()))


def fit(iters=2000, res=200, n_train=60, rays=4096, log2_T=19, seed=0, device="cuda", log_every=250, pose_noise=0.0):
    focal = 1111.1 * res / 800.0
    c2w, intr = synthetic.orbit_cameras(n_train + 4, height=res, width=res, focal=focal)
    total = n_train + 4
    held = [total // 16, (5 * total) // 16, (9 * total) // 16, (13 * total) // 16]  # spread around the orbit
    train_ids = [i for i in range(n_train + 4) if i not in held]
    cams_all = Cameras(c2w, intr[:, 0], intr[:, 1], intr[:, 2], intr[:, 3], res, res)
    images, masks = synthetic.analytic_dataset(cams_all, device)
    sel = torch.tensor(train_ids)
    c2w_train = c2w[sel].clone()
    if pose_noise > 0:  # perturbed training poses: the camera_opt group has something to recover
        g = torch.Generator().manual_seed(seed + 7)
        c2w_train[:, :, 3] += pose_noise * torch.randn(len(sel), 3, generator=g)
    cams_train = Cameras(c2w_train, intr[sel, 0], intr[sel, 1], intr[sel, 2], intr[sel, 3], res, res)
    cfg = PC.FruitNerfModelConfig(log2_hashmap_size=log2_T)
    params = PC.init_params(cfg.field_spec(len(sel)), cfg.proposal_specs(), seed=seed, grid_scale=1e-3, device=device)
    pipe = FruitPipeline(FruitPipelineConfig(FruitDataManagerConfig(rays, rays), cfg), device, cams_train,
                         SceneBox(torch.tensor(synthetic.SCENE_AABB)), test_mode="val", params=params,
                         images=images[sel], fruit_masks=masks[sel], seed=seed)
    model = pipe.model
    model.training = True
    tr = FruitTrainer(model)
    log = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(iters):
        rb, batch = pipe.datamanager.next_train(it)
        out = tr.train_iteration(rb, batch)
        if it % log_every == 0 or it == iters - 1:
            ld = {k: float(v) for k, v in out["loss_dict"].items()}
            log.append({"iter": it, **{k: round(v, 6) for k, v in ld.items()},
                        "psnr": round(float(out["metrics_dict"]["psnr"]), 2)})
    torch.cuda.synchronize()
    train_s = time.perf_counter() - t0
    model.eval()
    cams_dev = cams_all.to(device)
    views = []
    for h in held:
        rb = cams_dev.generate_rays(h, keep_shape=True)
        rb.camera_indices = torch.zeros_like(rb.camera_indices)  # held-out view: pose tweak / embedding of camera 0
        out = model.get_outputs_for_camera_ray_bundle(rb)
        pred = out["rgb"].to(device)
        lab = out["semantics_colormap"][..., :1].to(device) > 0.5
        gt_m = masks[h] > 0.5
        inter, union = float((lab & gt_m).sum()), float((lab | gt_m).sum())
        views.append({"camera": h, "psnr": round(psnr(pred, images[h]), 2), "fruit_iou": round(inter / max(union, 1.0), 3)})
    res_d = {"iters": iters, "rays_per_iter": rays, "image": [res, res], "train_views": len(sel), "train_seconds": round(train_s, 2),
             "ms_per_iter": round(1e3 * train_s / iters, 3), "log": log, "held_out": views,
             "held_out_psnr_mean": round(sum(v["psnr"] for v in views) / len(views), 2),
             "held_out_fruit_iou_mean": round(sum(v["fruit_iou"] for v in views) / len(views), 3),
             "camera_opt_translation": float(model.params["camera_optimizer.pose_adjustment"][:, :3].norm()),
             "camera_opt_rotation": float(model.params["camera_optimizer.pose_adjustment"][:, 3:].norm())}
    return res_d, pipe, (cams_all, images, masks, held)


def oracle_check(pipe, data, res_small=48, device="cuda"):
    """Render one held-out view at low resolution with the HIP model and with the CPU oracle (test infrastructure)."""
    from oracle import model as OM, rays as ORY
    from oracle import field as OF

    cams_all, images, masks, held = data
    h = held[0]
    focal = float(cams_all.fx[h]) * res_small / cams_all.width
    c2w = cams_all.camera_to_worlds[h:h + 1].cpu()
    intr = torch.tensor([[focal, focal, res_small / 2.0, res_small / 2.0]])
    cam = Cameras(c2w, intr[:, 0], intr[:, 1], intr[:, 2], intr[:, 3], res_small, res_small).to(device)
    rb = cam.generate_rays(0, keep_shape=True)
    gt, _, _ = synthetic.analytic_render(rb.origins, rb.directions)
    hip = pipe.model.get_outputs_for_camera_ray_bundle(rb)["rgb"]
    m = pipe.model
    cpu_params = {k: v.detach().cpu() for k, v in m.params.items()}
    fspec = OF.FieldSpec(grid=OF.GridSpec(16, 16, m.config.max_res, m.config.log2_hashmap_size, 2),
                         num_images=m.num_train_data)
    pspecs = [OF.ProposalSpec(OF.GridSpec(a["num_levels"], 16, a["max_res"], a["log2_hashmap_size"], 2))
              for a in m.config.proposal_net_args_list]
    om = OM.OracleModel(cpu_params, OM.ModelConfig(field=fspec, proposals=pspecs), torch.tensor(synthetic.SCENE_AABB),
                        test_mode="val")
    om.anneal = m._anneal  # the sampler keeps the last annealing exponent of training (fruit_nerf.py:206-216)
    ys, xs = torch.meshgrid(torch.arange(res_small), torch.arange(res_small), indexing="ij")
    orb = ORY.pinhole_rays(c2w, intr, torch.zeros(res_small * res_small, dtype=torch.long), ys.reshape(-1), xs.reshape(-1))
    ref = om.forward(orb)["rgb"].reshape(res_small, res_small, 3)
    gt = gt.cpu()
    return {"image": [res_small, res_small], "psnr_hip_vs_oracle": round(psnr(hip.cpu(), ref), 2),
            "psnr_hip_vs_gt": round(psnr(hip.cpu(), gt), 3), "psnr_oracle_vs_gt": round(psnr(ref, gt), 3)}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=2000)
    ap.add_argument("--res", type=int, default=200)
    ap.add_argument("--rays", type=int, default=4096)
    ap.add_argument("--log2-T", type=int, default=19)
    ap.add_argument("--pose-noise", type=float, default=0.0)
    ap.add_argument("--oracle", action="store_true")
    ap.add_argument("--save", default=None)
    a = ap.parse_args()
    out, pipe, data = fit(a.iters, a.res, rays=a.rays, log2_T=a.log2_T, pose_noise=a.pose_noise)
    if a.oracle:
        out["oracle"] = oracle_check(pipe, data)
    if a.save:
        from cropnerf_amd.fruit_nerf.checkpoint import save_run
        save_run(a.save, pipe.model.config, pipe.datamanager.cameras.to("cpu"), pipe.model.scene_box, pipe.model.params, step=a.iters)
    print(json.dumps(out))
