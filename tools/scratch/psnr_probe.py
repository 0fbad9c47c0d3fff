import sys, os, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import fit_scene
from cropnerf_amd import synthetic
res, pipe, (cams_all, images, masks, held) = fit_scene.fit(2000, 200)
m = pipe.model; m.eval()
dev = "cuda"
cams = pipe.datamanager.cameras
tr_imgs = pipe.datamanager.images
def psnr(a,b): return float(-10*torch.log10(((a-b)**2).mean()))
out = {}
for name, cfg_avg in (("mean_embedding", True), ("zeros_embedding", False)):
    m.config.use_average_appearance_embedding = cfg_avg
    vals = []
    for i in (0, 10, 20, 30):
        rb = cams.generate_rays(i, keep_shape=True)
        pred = m.get_outputs_for_camera_ray_bundle(rb)["rgb"].to(dev)
        vals.append(round(psnr(pred, tr_imgs[i]), 2))
    out["train_views_eval_mode_" + name] = vals
# train views with training-mode embedding (per camera)
m.training = True
vals = []
for i in (0, 10, 20, 30):
    rb = cams.generate_rays(i, keep_shape=True).flatten()
    o = m(rb)
    vals.append(round(psnr(o["rgb"].reshape(200,200,3), tr_imgs[i]), 2))
out["train_views_train_mode"] = vals
out["held_out"] = res["held_out"]
print(out)
