"""Fold the CSVs of tools/collect_pmc_workloads.sh into profiles/<tag>_pmc_workloads.json: per secondary workload of bench.py
the memory-side bytes (FETCH_SIZE + WRITE_SIZE of this library's kernels, KiB -> bytes, raw: the guide's x2 for FETCH_SIZE is
calibrated for 16-byte streaming reads, these are 8-byte gathers and 4-byte atomics) per unit of work, with the split by kernel;
and profiles/<tag>_pmc_train_atomics.json: float-atomic requests (TCC_ATOMIC) per training iteration and kernel at each batch
size, measured (none scaled)."""
import csv, glob, json, os, re, subprocess, sys

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"pmcw_{tag}")
try:
    commit = os.environ.get("CN_PROFILE_COMMIT") or subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"],
                                                                   capture_output=True, text=True).stdout.strip() or None
except Exception:  # noqa: BLE001
    commit = None


def short(name):
    name = name.replace("void ", "")
    m = re.match(r"(?:cn::(?:mf::|pw::)?)?([A-Za-z0-9_]+)", name)
    return m.group(1) if m else name[:40]


def fold(d):
    """{counter: {kernel: (total, launches)}} for the cn:: kernels of one pass, and the units line of its log"""
    acc = {}
    # (gpurun merges a call's files INTO gpurun_out/: an earlier collection's CSVs of the same pass stay beside the new one --
    #  only the newest file of a pass is that pass)
    files = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
    for f in files[-1:]:
        for row in csv.DictReader(open(f)):
            if "cn::" not in row["Kernel_Name"]:
                continue
            a = acc.setdefault(row["Counter_Name"], {}).setdefault(short(row["Kernel_Name"]), [0.0, set()])
            a[0] += float(row["Counter_Value"])
            a[1].add(row["Dispatch_Id"])
    units = None
    log = os.path.join(d, "run.log")
    if os.path.exists(log):
        for line in open(log):
            if line.startswith("PMC_UNITS "):
                units = json.loads(line[len("PMC_UNITS "):])
    return acc, units


out = {"commit": commit, "command": "rocprofv3 --pmc <group> -- python3 tools/pmc_workloads.py <workload>  (one pass per group)",
       "note": "bytes = (FETCH_SIZE + WRITE_SIZE) x 1024 summed over this library's kernels in the whole process, divided by the units "
               "the process handled; memory-side (beyond the L2) traffic, Infinity-Cache hits included", "workloads": {}}
atom = {"commit": commit, "command": "TRAIN_RAYS=<rays> TRAIN_FIELD_SAMPLES=<S> rocprofv3 --pmc TCC_ATOMIC_sum TCC_EA0_ATOMIC_sum -- "
        "python3 tools/pmc_workloads.py train", "configs": {}}
names = sorted({os.path.basename(d).rsplit("_", 1)[0] for d in glob.glob(os.path.join(src, "*_*")) if os.path.isdir(d)}
               | {re.sub(r"_(FETCH_SIZE|WRITE_SIZE|TCC_ATOMIC_sum)$", "", os.path.basename(d)) for d in glob.glob(os.path.join(src, "*"))})
for name in sorted({re.sub(r"_(FETCH_SIZE|WRITE_SIZE|TCC_ATOMIC_sum)$", "", os.path.basename(d)) for d in glob.glob(os.path.join(src, "*")) if os.path.isdir(d)}):
    entry, per_kernel, units = {}, {}, None
    for group in ("FETCH_SIZE", "WRITE_SIZE"):
        acc, u = fold(os.path.join(src, f"{name}_{group}"))
        units = units or u
        tot = 0.0
        for kern, (total, ids) in acc.get(group, {}).items():
            per_kernel.setdefault(kern, {})[group + "_bytes"] = total * 1024
            per_kernel[kern]["launches"] = len(ids)
            tot += total * 1024
        entry[group + "_bytes"] = tot
    if not units or not units.get("units"):
        continue
    n = units["units"]
    entry.update(units=n, unit=units["unit"], bytes_per_unit=(entry["FETCH_SIZE_bytes"] + entry["WRITE_SIZE_bytes"]) / n,
                 per_kernel={k: {kk: (vv / n if kk.endswith("_bytes") else vv) for kk, vv in v.items()} for k, v in per_kernel.items()})
    entry["per_kernel_note"] = "bytes per unit of the workload, launches in the whole process"
    out["workloads"][name] = entry
    if name.startswith("train_"):
        acc, u = fold(os.path.join(src, f"{name}_TCC_ATOMIC_sum"))
        if u and acc:
            k = {}
            for counter, kernels in acc.items():
                for kern, (total, ids) in kernels.items():
                    if total > 0:
                        k.setdefault(kern, {})[counter + "_per_iteration"] = total / u["units"]
                        k[kern]["launches_per_iteration"] = len(ids) / u["units"]
            atom["configs"][name[len("train_"):]] = {"iterations": u["units"], "unit": u["unit"], "kernels": k,
                                                      "requests_per_iteration": sum(v.get("TCC_ATOMIC_sum_per_iteration", 0.0) for v in k.values())}
json.dump(out, open(os.path.join(root, "profiles", f"{tag}_pmc_workloads.json"), "w"), indent=1)
json.dump(atom, open(os.path.join(root, "profiles", f"{tag}_pmc_train_atomics.json"), "w"), indent=1)
for n, e in out["workloads"].items():
    print(n, f"{e['bytes_per_unit']:.1f} B per unit ({e['unit']}); fetch {e['FETCH_SIZE_bytes'] / 1e9:.3f} GB write {e['WRITE_SIZE_bytes'] / 1e9:.3f} GB over {e['units']} units")
for n, e in atom["configs"].items():
    print("atomics", n, f"{e['requests_per_iteration']:.4g} requests per iteration", {k: round(v.get('TCC_ATOMIC_sum_per_iteration', 0)) for k, v in e['kernels'].items()})
