"""rocprofv3 output of tools/pmc_target.py -> <out>/r04_pmc_<name>_summary.json and <out>/r04_traffic_<name>.json (one entry of
profiles/r04_traffic.json: `python tools/pmc_summary.py --merge profiles` folds the entries into it).

    python tools/pmc_summary.py <name> <units_per_sweep> <warm> <timed> <dir with the passes> [out dir, default profiles/]
The directory holds one sub-directory per pass: `trace/` (rocprofv3 --kernel-trace --stats: *_kernel_trace.csv) and `pmc*/`
(*_counter_collection.csv, one --pmc pass each).  Per kernel (template arguments kept, namespaces dropped) the dispatches are in
launch order; the last timed / (warm + timed) of them belong to the timed sweeps (every sweep launches the same kernels).
FETCH_SIZE / WRITE_SIZE are KiB; FETCH_SIZE is doubled as /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950 (128-byte
read requests are tallied at 64 bytes).  hbm_bytes = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024."""
import collections
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"\bphm::", "", name)
    m = re.match(r"(?:void\s+)?([\w:]+(?:<[^(]*>)?)\s*\(", name)
    return (m.group(1) if m else name).replace(" ", "")


ROUND = "r04"


def merge(out):
    tf = os.path.join(out, f"{ROUND}_traffic.json")
    t = json.load(open(tf)) if os.path.exists(tf) else {}
    for f in sorted(glob.glob(os.path.join(out, f"{ROUND}_traffic_*.json"))):
        t.update(json.load(open(f)))
        os.remove(f)
    json.dump(t, open(tf, "w"), indent=1)
    print("merged", sorted(t))


def main():
    if sys.argv[1] == "--merge":
        return merge(sys.argv[2])
    name, units, warm, timed, d = sys.argv[1], float(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    out_dir = sys.argv[6] if len(sys.argv) > 6 else os.path.join(ROOT, "profiles")
    frac = timed / float(warm + timed)
    kernels = collections.defaultdict(dict)
    for f in sorted(glob.glob(os.path.join(d, "trace*", "**", "*kernel_trace.csv"), recursive=True)):
        per = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            per[short(r["Kernel_Name"])].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
        for k, v in per.items():
            v.sort()
            keep = v[len(v) - int(round(len(v) * frac)):] if len(v) >= warm + timed else v
            kernels[k]["launches_per_sweep"] = len(keep) / float(timed)
            kernels[k]["ns_per_sweep"] = sum(x[1] for x in keep) / float(timed)
            kernels[k]["avg_launch_ns"] = sum(x[1] for x in keep) / max(1, len(keep))
    for f in sorted(glob.glob(os.path.join(d, "pmc*", "**", "*counter_collection.csv"), recursive=True)):
        per = collections.defaultdict(lambda: collections.defaultdict(dict))
        for r in csv.DictReader(open(f)):
            dd = per[short(r["Kernel_Name"])][r["Counter_Name"]]
            dd[int(r["Dispatch_Id"])] = dd.get(int(r["Dispatch_Id"]), 0.0) + float(r["Counter_Value"])
        for k, cs in per.items():
            for c, dd in cs.items():
                ids = sorted(dd)
                keep = ids[len(ids) - int(round(len(ids) * frac)):] if len(ids) >= warm + timed else ids
                kernels[k][c] = sum(dd[i] for i in keep) / float(timed)
    total = 0.0
    for k, e in kernels.items():
        if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
            e["hbm_bytes_per_sweep"] = e["FETCH_SIZE"] * 2048.0 + e["WRITE_SIZE"] * 1024.0
            e["hbm_bytes_per_unit"] = e["hbm_bytes_per_sweep"] / units
            total += e["hbm_bytes_per_sweep"]
    out = {"name": name, "units_per_sweep": units, "warm_sweeps": warm, "timed_sweeps": timed, "per": "sweep (counter sums over the timed sweeps / timed)",
           "all_kernels_hbm_bytes_per_unit": total / units, "kernels": {k: kernels[k] for k in sorted(kernels)}}
    json.dump(out, open(os.path.join(out_dir, f"{ROUND}_pmc_{name}_summary.json"), "w"), indent=1)
    t = {name: {"source": f"profiles/{ROUND}_pmc_{name}_summary.json", "units_per_sweep_in_pmc_run": units,
                "all_kernels_bytes_per_unit": total / units,
                "kernels": {k: {kk: e[kk] for kk in ("hbm_bytes_per_unit", "launches_per_sweep", "ns_per_sweep", "SQ_INSTS_VALU", "SQ_WAIT_ANY", "SQ_WAVE_CYCLES",
                                                     "SQ_INSTS_VALU_MFMA_MOPS_F64") if kk in e} for k, e in kernels.items() if "hbm_bytes_per_unit" in e}}}
    json.dump(t, open(os.path.join(out_dir, f"{ROUND}_traffic_{name}.json"), "w"), indent=1)
    for k in sorted(kernels, key=lambda k: -kernels[k].get("ns_per_sweep", 0)):
        e = kernels[k]
        print(f"{k[:70]:70s} {e.get('launches_per_sweep', 0):6.1f} launches  {e.get('ns_per_sweep', 0) / 1e6:8.3f} ms  {e.get('hbm_bytes_per_unit', float('nan')):8.1f} B/unit")
    print(f"all kernels: {total / units:.1f} B per unit")


if __name__ == "__main__":
    main()
