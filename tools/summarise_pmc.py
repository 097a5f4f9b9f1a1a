"""rocprofv3 --pmc passes of `bench.py --config C --no-cpu --no-extras` -> per-kernel counter sums and the HBM traffic figures
bench.py reports as roofline.traffic (profiles/r02_traffic.json).

Usage: python tools/summarise_pmc.py C STEPS WARMUP pass1_counter_collection.csv [pass2.csv ...]   (one CSV per --pmc pass)
FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE is doubled as /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950
(wide coalesced reads are tallied at half their size).  Counters are summed per kernel over the TIMED sweeps (the dispatches
after the warm-up sweeps); per-launch / per-sweep figures divide by the number of timed launches / sweeps."""
import collections
import csv
import json
import sys


def short(name):
    for key in ("tiles_branch_kernel", "tiles_up_kernel", "tiles_down_kernel", "tiles_root_kernel", "tiles_chunk_kernel", "tiles_stats_kernel",
                "wt_branch_kernel", "wt_up_kernel", "wt_down_kernel", "wt_root_kernel", "wt_stats_kernel", "mcmc_sweep_kernel",
                "exp_sample_kernel", "exp_wide_kernel", "stats_reduce_kernel"):
        if key in name:
            return key
    return None


def main():
    cfg, steps, warm = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    sums = collections.defaultdict(lambda: collections.defaultdict(float))      # kernel -> counter -> sum over timed dispatches
    launches = collections.defaultdict(int)
    for path in sys.argv[4:]:
        per = collections.defaultdict(lambda: collections.defaultdict(dict))    # kernel -> counter -> dispatch -> value
        for r in csv.DictReader(open(path)):
            k = short(r["Kernel_Name"])
            if k is None:
                continue
            d = per[k][r["Counter_Name"]]
            d[int(r["Dispatch_Id"])] = d.get(int(r["Dispatch_Id"]), 0.0) + float(r["Counter_Value"])
        for k, cs in per.items():
            for c, d in cs.items():
                ids = sorted(d)
                n_total = len(ids)
                # dispatches of this kernel are spread evenly over warm-up and timed sweeps (+ the engine probe for sizing,
                # which runs no sweep); the pruning-only repetitions of the replica mapping come last
                per_sweep = n_total / float(steps + warm) if k != "mcmc_sweep_kernel" else None
                if per_sweep is not None and abs(per_sweep - round(per_sweep)) < 1e-9:
                    first = int(round(per_sweep)) * warm
                    timed = ids[first:]
                else:
                    timed = ids
                sums[k][c] = sum(d[i] for i in timed)
                launches[k] = len(timed)
    out = {"config": cfg, "steps": steps, "warmup": warm, "kernels": {}}
    for k, cs in sums.items():
        e = {"timed_launches": launches[k]}
        e.update({c: v for c, v in cs.items()})
        if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
            e["hbm_bytes"] = cs["FETCH_SIZE"] * 1024 * 2 + cs["WRITE_SIZE"] * 1024
            e["hbm_bytes_per_launch"] = e["hbm_bytes"] / max(1, launches[k])
            e["hbm_bytes_per_sweep"] = e["hbm_bytes"] / steps
        out["kernels"][k] = e
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
