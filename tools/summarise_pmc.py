"""Summarise rocprofv3 --pmc passes of `bench.py --steps 16 --warmup 8 --no-cpu` into profiles/r01_final_pmc_summary.json.

Usage: python tools/summarise_pmc.py gpurun_out/pmc_a/x_counter_collection.csv [more csv ...] > summary.json
Per counter: the value of every dispatch of the sweep / pruning kernels, in launch order.  FETCH_SIZE / WRITE_SIZE are in
KiB; FETCH_SIZE is doubled as /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950 (wide coalesced reads are
reported at half their size).  The timed dispatch is the LAST 8-sweep launch of mcmc_sweep_kernel."""
import collections
import csv
import json
import sys

E, SWEEPS = 1998, 8
out = collections.OrderedDict()
disp = {}
for path in sys.argv[1:]:
    per = collections.defaultdict(lambda: collections.OrderedDict())
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        if "mcmc_sweep_kernel" not in k and "mcmc_pruning_kernel" not in k:
            continue
        d = per[r["Counter_Name"]]
        d[r["Dispatch_Id"]] = d.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
        if "mcmc_sweep_kernel" in k:
            disp = {x: r[x] for x in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "SGPR_Count", "Scratch_Size")}
            disp["kernel"] = k[:80]
    for c, d in per.items():
        out[c] = [d[i] for i in sorted(d, key=int)]
sweeps = [i for i, v in enumerate(out.get("SQ_WAVES", out[next(iter(out))])) if True]
res = {k: v for k, v in out.items()}
res["dispatch"] = disp
S = int(disp.get("Grid_Size", 0))
units = E * S * SWEEPS
per_launch = {"units": units}
def timed(c):      # the last of the leading 8-sweep sweep-kernel dispatches = the largest values; take index 2 (warm-up, then two timed)
    v = out.get(c)
    return None if not v else v[2] if len(v) > 2 else v[-1]
if timed("FETCH_SIZE") is not None and timed("WRITE_SIZE") is not None:
    rd, wr = timed("FETCH_SIZE") * 1024 * 2, timed("WRITE_SIZE") * 1024
    per_launch.update(hbm_read_bytes_corrected=rd, hbm_write_bytes=wr, hbm_bytes_per_unit=(rd + wr) / units)
if timed("SQ_INSTS_VALU") is not None:
    per_launch["valu_insts_per_branch_wave"] = timed("SQ_INSTS_VALU") / (E * (S / 64) * SWEEPS)
res["per_launch"] = per_launch
res["note"] = ("rocprofv3 --pmc passes (separate runs, --pmc only) of: python3 bench.py --steps 16 --warmup 8 --no-cpu ; 3 dispatches of 8 "
               "sweeps (first = warm-up) followed by 8 single-sweep pruning-only dispatches; FETCH_SIZE/WRITE_SIZE in KiB; FETCH_SIZE doubled per "
               "MI355X_MICROARCH.md (gfx950 reports half of wide coalesced reads).")
json.dump(res, sys.stdout, indent=1)
