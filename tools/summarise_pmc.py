"""Fold the rocprofv3 CSVs written by tools/collect_pmc.sh into profiles/<tag>_pmc_render_fused.json (per-launch
averages of every counter for the dominant render kernel + its dispatch info) and profiles/<tag>_kernel_stats.csv."""
import csv, glob, json, os, shutil, sys

tag = sys.argv[1]
suffix = sys.argv[2] if len(sys.argv) > 2 else "pmc_render_fused"  # variants of the render: "pmc_variant" (bench.py reads *_pmc_render*)
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"pmc_{tag}")
KERNELS = ("render_split_kernel", "render_fused_kernel<false, false>", "render_f16_kernel")
out = {}
for f in glob.glob(os.path.join(src, "*", "**", "*counter_collection.csv"), recursive=True):
    acc = {}
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if not any(k in row["Kernel_Name"] for k in KERNELS):
                continue
            out.setdefault("kernel", row["Kernel_Name"])
            out.setdefault("dispatch", {k: row[k] for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count",
                                                               "LDS_Block_Size", "Scratch_Size", "Grid_Size",
                                                               "Workgroup_Size") if k in row})
            a = acc.setdefault(row["Counter_Name"], [0.0, set()])
            a[0] += float(row["Counter_Value"])
            a[1].add(row["Dispatch_Id"])
    for name, (total, ids) in acc.items():
        out[name] = {"launches": len(ids), "avg_per_launch": total / max(len(ids), 1)}
stats = glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True)
os.makedirs(os.path.join(root, "profiles"), exist_ok=True)
if stats:
    shutil.copy(stats[0], os.path.join(root, "profiles", f"{tag}_kernel_stats.csv"))
if "TCC_HIT_sum" in out and "TCC_MISS_sum" in out:
    h, m = out["TCC_HIT_sum"]["avg_per_launch"], out["TCC_MISS_sum"]["avg_per_launch"]
    out["l2_hit_rate"] = h / max(h + m, 1.0)
import subprocess

try:  # the commit the counters were measured on (bench.py prints it next to `traffic`)
    out["commit"] = os.environ.get("CN_PROFILE_COMMIT") or subprocess.run(
        ["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or None
except Exception:  # noqa: BLE001
    out["commit"] = os.environ.get("CN_PROFILE_COMMIT")
out["note"] = ("per-launch averages over the launches of the named kernel in `bench.py --steps 3 --warmup 1 "
               "--no-cpu-baseline --no-secondary`, one rocprofv3 --pmc pass per counter group; FETCH_SIZE / WRITE_SIZE in KiB")
with open(os.path.join(root, "profiles", f"{tag}_{suffix}.json"), "w") as fh:
    json.dump(out, fh, indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "note"}, indent=1)[:1500])
