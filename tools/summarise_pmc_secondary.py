"""Fold the CSVs of tools/collect_pmc_secondary.sh into profiles/<tag>_pmc_secondary.json: per kernel (proposal sampler,
training kernels) the per-launch average of every counter, plus L2 hit rate and matrix-pipe / VALU shares of the SIMD cycles."""
import csv, glob, json, os, sys

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"pmc2_{tag}")
WANT = ("proposal_sample_kernel", "field_backward_mfma_kernel", "proposal_backward_kernel", "proposal_density_kernel",
        "render_split_kernel<true>", "adam_step_kernel", "train_render_backward_kernel", "interlevel_backward_kernel")
out = {}
for f in glob.glob(os.path.join(src, "*", "**", "*counter_collection.csv"), recursive=True):
    acc = {}
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = next((w for w in WANT if w in row["Kernel_Name"]), None)
            if k is None:
                continue
            a = acc.setdefault((k, row["Counter_Name"]), [0.0, set()])
            a[0] += float(row["Counter_Value"])
            a[1].add(row["Dispatch_Id"])
            out.setdefault(k, {}).setdefault("dispatch", {c: row[c] for c in ("VGPR_Count", "Accum_VGPR_Count", "LDS_Block_Size",
                                                                               "Scratch_Size", "Grid_Size", "Workgroup_Size") if c in row})
    for (k, name), (total, ids) in acc.items():
        out[k][name] = {"launches": len(ids), "avg_per_launch": total / max(len(ids), 1)}
for k, v in out.items():
    g = lambda n: v.get(n, {}).get("avg_per_launch")
    if g("TCC_HIT_sum") is not None and g("TCC_MISS_sum") is not None:
        v["l2_hit_rate"] = g("TCC_HIT_sum") / max(g("TCC_HIT_sum") + g("TCC_MISS_sum"), 1.0)
    if g("SQ_BUSY_CYCLES") and g("SQ_VALU_MFMA_BUSY_CYCLES") is not None:
        v["note_sq"] = "SQ_* are summed over shader engines / SIMDs as rocprofv3 reports them; compare ratios between kernels, not absolutes"
out["_note"] = ("per-launch averages; FETCH_SIZE / WRITE_SIZE in KiB; proposal kernel from `bench.py --mode proposal`, training kernels from "
                "`tools/train_probe.py` (default method, 4096 rays); one rocprofv3 --pmc run per counter group")
path = os.path.join(root, "profiles", f"{tag}_pmc_secondary.json")
json.dump(out, open(path, "w"), indent=1)
for k, v in out.items():
    if k.startswith("_"):
        continue
    print(k, {n: (round(x["avg_per_launch"]) if isinstance(x, dict) and "avg_per_launch" in x else x) for n, x in v.items() if n not in ("dispatch", "note_sq")})
