"""profiles/r02_pmc_C*_summary.json (tools/summarise_pmc.py) -> profiles/r02_traffic.json, the per-unit HBM traffic figures that
bench.py scales to its own launch size for `roofline.traffic`.  A unit = one branch x replica x sweep; the PMC runs are
`bench.py --config C --steps 3 --warmup 3 --no-cpu --no-extras` under rocprofv3 --pmc (separate FETCH_SIZE / WRITE_SIZE passes,
FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950).
Usage: python tools/make_traffic_json.py C:units_per_sweep[:ipl] ...   e.g.  3:327647232 4:65404928 5:163807232 2:785645568:8"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = {}
for arg in sys.argv[1:]:
    parts = arg.split(":")
    cfg, units = int(parts[0]), float(parts[1])
    ipl = int(parts[2]) if len(parts) > 2 else 1
    j = json.load(open(os.path.join(ROOT, "profiles", f"r02_pmc_C{cfg}_summary.json")))
    k = j["kernels"]
    e = {"source": f"profiles/r02_pmc_C{cfg}_summary.json", "units_per_sweep_in_pmc_run": units}
    for name in ("tiles_branch_kernel", "wt_branch_kernel"):
        if name in k and "hbm_bytes_per_launch" in k[name]:
            e["branch_kernel_bytes_per_unit"] = k[name]["hbm_bytes_per_launch"] / units
    for name in ("tiles_up_kernel", "wt_up_kernel"):
        if name in k and "hbm_bytes_per_sweep" in k[name]:
            e["up_kernels_bytes_per_unit"] = k[name]["hbm_bytes_per_sweep"] / units
    if "mcmc_sweep_kernel" in k and "hbm_bytes" in k["mcmc_sweep_kernel"]:
        m = k["mcmc_sweep_kernel"]
        e["sweep_kernel_bytes_per_unit"] = m.get("sweep_bytes_per_unit")
        e["pruning_bytes_per_unit"] = m.get("pruning_bytes_per_unit")
    e["all_kernels_bytes_per_unit"] = sum(v.get("hbm_bytes_per_sweep", 0.0) for v in k.values()) / units
    out[f"C{cfg}"] = e
# C2 (mcmc_sweep_kernel<4>, unchanged since round 1): the round-1 PMC passes (profiles/r01_final_pmc_summary.json): 248.5 B per unit for
# the fused sweep; pruning-only dispatches (5.84e6 KiB x 2 fetched + 12.28e6 KiB written) / 7.856e8 units = 31 B per unit
r01 = os.path.join(ROOT, "profiles", "r01_traffic.json")
if os.path.exists(r01):
    t = json.load(open(r01))
    out["C2"] = {"source": "profiles/r01_final_pmc_summary.json (kernel unchanged this round)", "sweep_kernel_bytes_per_unit": t["hbm_bytes_per_unit"],
                 "pruning_bytes_per_unit": (5.84e6 * 2 + 12.28e6) * 1024 / 7.856e8, "all_kernels_bytes_per_unit": t["hbm_bytes_per_unit"]}
json.dump(out, open(os.path.join(ROOT, "profiles", "r02_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
