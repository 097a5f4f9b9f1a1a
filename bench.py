#!/usr/bin/env python3
"""Headline benchmark: stochastic-map realisations/s (branch x site paths sampled per second).

Contract: `python bench.py --gpus N --steps K --warmup W`; for N > 1 the driver launches one rank per GPU with
torch.distributed.run.  A *step* is one full MCMC sweep (pruning + node sampling + branch path resampling +
sufficient statistics) over every replica resident on the GPU.

Workload at N = 1 (and, replica-sharded, at N > 1): BASELINE.json configs[2] = C3, the configuration the north-star target
is quoted on -- sumstatMCMC_bigtree, 4-state Q = make2sQ(.1,.1,.2,.2,10), 10 000-tip synthetic tree
(src/phylomap.cpp:942-986).  It fits one GPU (16 384 replicas with one wave per (tile of 64 replicas, branch)).
Replicas (independent chains / sites) are sharded across ranks with no data-path collective; the only exchange is one
RCCL all-reduce of the K x cols statistics at the end.  Inputs are resident in HBM before the timed region starts.
`--scaling weak` (default): every rank runs the same number of replicas, rank 0's choice broadcast to all;
`--scaling strong`: the replicas one GPU holds are split over the ranks.

`roofline` is SURVEY.md 8(d)'s whole-sweep figure, E * S * B_alg / (time of all kernels of one sweep), with one block per
kernel of the sweep nested under it (`kernels`; `dominant_kernel` names the largest).  The same JSON line carries, at N = 1,
the rate of ONE chain on the headline configuration (what the R call gets), one block per other BASELINE configuration
(C2 streaming layout, C4 dense 61 states in the n + n^2 counting layout, C5 sparse 20 states, sumstatEXP on C1 and on a
1 000-tip tree), the batched expm rates, and the CPU oracle timed on this box's host cores.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (about 6.3 TB/s achievable)
MEASURED_TRIAD_GBS = 5751.9   # tools/device_peaks/device_peaks.hip on an MI355X of this pool (read-only 6382, copy 4956 GB/s; FP64 vector 65.5 TFLOP/s)
MFMA_F64_PEAK_TFLOPS = 78.6   # MI355X_MICROARCH.md: dense FP64 matrix peak (= the vector FP64 peak on this chip)
N_SIMD, NOMINAL_GHZ, VALU_CLOCKS = 1024, 2.4, 4.9   # 256 CUs x 4 SIMDs; a back-to-back wave64 VALU instruction issues in 4.7-5.4 nominal clocks (profiles/r01_valu_rates.log)


# ----------------------------------------------------------------------------------------------------------------------
# CPU baseline: the oracle (oracle/phm_oracle.c, a restatement of src/phylomap.cpp), timed BEFORE anything touches the GPU
# (the all-cores figure forks worker processes; a process that has initialised HIP must not fork workers on this pool).
# ----------------------------------------------------------------------------------------------------------------------
def _oracle_run(args):
    cfg, n_it, faithful, replica = args
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    from phylomap_amd import synth, treeorder
    z, Q, pid, Omega = synth.config_problem(cfg)
    nen, nodelist, root = treeorder.pruningwiseedgeorder(z), treeorder.makenodelist(z), treeorder.myreorder(z)
    B = np.eye(Q.shape[0]) + Q / Omega
    t0 = time.perf_counter()
    _, rc = O.maketreelistMCMC(z, Q, pid, B, Omega, nen, nodelist, root, n_it, variant=O.BIGTREE, seed=1, replica=replica,
                               faithful_search=faithful)
    dt = time.perf_counter() - t0
    assert rc == 0
    return dt


def cpu_baseline(cfg, target_s=3.0, all_cores=False):
    """The oracle on `cfg`'s tree and Q: one core (edge lookup table), one core with the reference's O(E) edge search per
    node (src/phylomap.cpp:643), and -- all_cores -- one independent chain per host core."""
    from phylomap_amd import synth
    z = synth.config_problem(cfg)[0]
    E = z["edge"].shape[0]
    n_probe = max(2, int(2e5 // E))
    probe = _oracle_run((cfg, n_probe, False, 0))
    n_it = max(n_probe, int(target_s / (probe / n_probe)))
    dt = _oracle_run((cfg, n_it, False, 0))
    probe_f = _oracle_run((cfg, 2, True, 0))
    n_f = max(2, int(min(target_s, 3.0) / (probe_f / 2)))
    dtf = _oracle_run((cfg, n_f, True, 0))
    out = {"value": E * n_it / dt, "unit": "branch-site realisations/s", "cores": 1, "kind": "port",
           "sample_short": f"C{cfg} tree+Q, 1 chain, {n_it} sweeps",
           "sample": f"same C{cfg} tree and Q, 1 chain, {n_it} sweeps (edge lookup table); with the reference's O(E) edge "
                     f"search per node (src/phylomap.cpp:643): {E * n_f / dtf:.4g}/s over {n_f} sweeps",
           "faithful_value": E * n_f / dtf}
    if all_cores:
        import multiprocessing as mp
        cores = os.cpu_count() or 1
        n_all = max(2, n_it // 2)
        t0 = time.perf_counter()
        with mp.get_context("fork").Pool(cores) as pool:
            pool.map(_oracle_run, [(cfg, n_all, False, r) for r in range(cores)])
        wall = time.perf_counter() - t0           # includes each worker's problem set-up: a lower bound on the rate
        out["all_cores"] = {"value": cores * E * n_all / wall, "cores": cores,
                            "sample": f"{cores} independent chains (one process per core), {n_all} sweeps each, wall clock "
                                      f"including process start and tree generation"}
    return out


def load_traffic():
    """HBM bytes per unit and per kernel from this round's rocprofv3 --pmc passes (tools/pmc_target.py -> tools/pmc_summary.py
    -> profiles/r04_traffic.json, entries not re-profiled this round from r03_traffic.json); bench.py scales them to its own launch sizes."""
    t = {}
    for rnd in ("r03", "r04"):                     # a configuration re-profiled this round replaces last round's entry
        f = os.path.join(ROOT, "profiles", f"{rnd}_traffic.json")
        if os.path.exists(f):
            t.update(json.load(open(f)))
    return t


def kernel_traffic(entry, *needles):
    """(bytes per unit, VALU instructions per sweep, kernel names) summed over the kernels of a traffic entry whose name holds a needle"""
    tot, valu, names = 0.0, 0.0, []
    for k, e in (entry or {}).get("kernels", {}).items():
        if any(nd in k for nd in needles):
            tot += e.get("hbm_bytes_per_unit", 0.0)
            valu += e.get("SQ_INSTS_VALU", 0.0)
            names.append(k)
    return (tot if names else None), (valu if names else None), names


def _r(x, digits=5):
    """floats to `digits` significant digits (the contract line is read by people and a parser, not re-computed from)"""
    if isinstance(x, float):
        return float(f"{x:.{digits}g}")
    if isinstance(x, str):
        return x[:64]
    return x


def _pick(d, keys):
    """the scalar (number / short label) entries of `d` named in `keys`, floats rounded"""
    return {k: _r(d[k]) for k in keys if d and k in d and not isinstance(d[k], (dict, list))}


def compact_line(out):
    """The ONE stdout line of the bench contract, built from the full result `out` (which goes to gpurun_out/bench_detail_n*.json).
    Numbers only, no prose, < 4 KB (tests/test_host_cpu.py::test_bench_line_is_compact): the round-3 line was 21 KB and the
    driver could not parse it."""
    line = {k: _r(out[k], 9 if k in ("value", "ms_per_step") else 5) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
                                                                                    "higher_is_better", "scaling", "vs_baseline", "dtype", "data") if k in out}
    cfg = out.get("config", {})
    line["config"] = {k: cfg[k] for k in ("workload", "n_states", "n_tips", "branches", "replicas_total", "max_iters_provisioned", "parallelism") if k in cfg}
    rk = ("bound", "alg_bytes_per_unit", "units_per_launch", "launches", "avg_launch_ms", "achieved", "peak", "unit", "frac", "traffic")
    rl = out.get("roofline") or {}
    line["roofline"] = _pick(rl, rk)
    if rl.get("dominant_kernel"):
        line["roofline"]["dominant_kernel"] = _pick(rl["dominant_kernel"], ("kernel",) + rk + ("valu_issue_frac",))
    ks = rl.get("kernels") or {}
    if ks:
        line["roofline"]["phases"] = {ph: _pick(k, ("avg_launch_ms", "alg_bytes_per_unit", "frac")) for ph, k in ks.items() if ph != "reductions"}
    cb = out.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = _pick(cb, ("value", "unit", "cores", "kind", "faithful_value"))
        line["cpu_baseline"]["sample"] = cb.get("sample_short", "")
        if "all_cores" in cb:
            line["cpu_baseline"]["all_cores"] = _pick(cb["all_cores"], ("value", "cores"))
        line["speedup_vs_cpu_1core"] = _r(out.get("speedup_vs_cpu_1core"))
    sc = out.get("single_chain")
    if sc:
        line["single_chain"] = {"ms_per_sweep": _r(sc["ms_per_sweep"]), "realisations_per_s": _r(sc["realisations_per_s"]),
                                "speedup": _r(sc.get("speedup_vs_cpu_1core"))}
    for k in ("recoveries", "replicas_per_gib", "hbm_bytes_resident"):
        if k in out:
            line[k] = _r(out[k])
    if out.get("stated_length"):
        line["stated_length"] = {k: _r(v) for k, v in out["stated_length"].items() if not isinstance(v, (dict, list, str))}
    others = {}
    for key, blk in (out.get("configs") or {}).items():
        o = {"value": _r(blk.get("realisations_per_s"), 4), "ms": _r(blk.get("ms_per_sweep"), 4)}
        brl = blk.get("roofline") or {}
        if "frac" in brl:
            o["frac"] = _r(brl["frac"], 3)
        dk = brl.get("dominant_kernel")
        if dk:
            o["dom"] = {"kernel": (dk.get("kernel") or "")[:36], "ms": _r(dk.get("avg_launch_ms"), 4), "frac": _r(dk.get("frac"), 3)}
        if "replicas" in blk:
            o["replicas"] = blk["replicas"]
        others[key] = {k: v for k, v in o.items() if v is not None}
    if others:
        line["others"] = others
    ex = out.get("expm_per_s")
    if ex:
        line["expm_per_s"] = {k: _r(v) for k, v in ex.items() if not isinstance(v, (dict, list, str))}
    line["detail"] = "gpurun_out/bench_detail_n%d.json" % out.get("n_gpus", 1)
    # hard bound: shed the secondary blocks (they stay in the detail file) before the line can outgrow the parser
    for shed in ("dom", "others", "expm_per_s", "stated_length"):
        if len(json.dumps(line)) <= 3800:
            break
        if shed == "dom":
            for o in line.get("others", {}).values():
                o.pop("dom", None)
        else:
            line.pop(shed, None)
    return line


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--burnin", type=int, default=12, help="untimed sweeps before the warm-up: the chain leaves its initial paths (two half-length segments per "
                    "branch, n on C5), whose sweeps cost less (more) than stationary ones; SURVEY 8(d) discards the first 10 %% of a run likewise")
    ap.add_argument("--replicas", type=int, default=0, help="replicas per GPU (weak) / in total (strong); 0: sized from rank 0's free HBM")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--config", type=int, default=3, help="BASELINE configuration of the headline line (3 = C3, the north-star's; 2 = C2)")
    ap.add_argument("--ipl", type=int, default=8, help="sweeps fused per kernel launch (replica mapping)")
    ap.add_argument("--storage", type=int, default=2, help="replica mapping, dwell streams: 2 = two buffers (fastest), 1 = one ring (half the HBM)")
    ap.add_argument("--mapping", default="", choices=["", "replicas", "tiles", "branches", "auto"],
                    help="how a sweep is laid over the lanes (DESIGN.md 4b); default: tiles for C3/C4/C5, replicas for C2")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="headline line only (skip the C2/C4/C5/EXP/expm blocks)")
    ap.add_argument("--force-collective", action="store_true", help="exercise the statistics hand-over to torch at N=1")
    args = ap.parse_args()

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit(f"bench.py --gpus {args.gpus} was started with WORLD_SIZE={world}: launch N > 1 with "
                         f"`python -m torch.distributed.run --nnodes=1 --nproc-per-node {args.gpus} --master-addr 127.0.0.1 "
                         f"--master-port P bench.py --gpus {args.gpus} ...` (one rank per GPU)")

    cpu = {}
    if rank == 0 and world == 1 and not args.no_cpu:          # before any HIP call: the all-cores figure forks workers
        cpu[args.config] = cpu_baseline(args.config, target_s=4.0, all_cores=True)
        if not args.no_extras:
            for c in (2, 4, 5):
                if c != args.config:
                    cpu[c] = cpu_baseline(c, target_s=2.0)

    import torch
    import torch.distributed as dist
    from phylomap_amd import _lib, api, parallel, synth

    rank, world, local_rank = parallel.env_rank()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the engine has no CPU fallback)")
    # Rehearsal switches (never set by the driver): several ranks on ONE card with a gloo process group, to exercise the
    # sharding / barrier / reduction logic where only one GPU is available.  The real N > 1 path is RCCL ("nccl").
    backend = os.environ.get("PHM_DIST_BACKEND", "nccl")
    if os.environ.get("PHM_ALL_RANKS_ON_DEVICE0") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    parallel.init_process_group(backend, device_index=local_rank)
    stream = torch.cuda.current_stream().cuda_stream
    traffic = load_traffic()
    dist_dev = torch.device("cuda", local_rank) if backend == "nccl" else torch.device("cpu")

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def sized_replicas(z, Q, pid, Omega, mapping, cap, frac=0.80, storage=0, variant=_lib.PHM_MCMC_BIGTREE, max_iters=1):
        """replicas that fit `frac` of the free HBM, in whole groups of tiles, at most `cap`"""
        free_b, _ = torch.cuda.mem_get_info()
        # the slots of the (tile, item) mappings are provisioned from the run length (S x E x sweeps draws): the probe tile gets the tail of the full job
        tail = max(1e-16, min(1e-9, 0.05 / (float(cap) * z["edge"].shape[0] * max_iters))) if mapping == "tiles" else 0.0
        probe = _lib.Engine(z, Q, pid, Omega, max_iters, variant=variant, seed=1, n_replicas=64, reduce=True,
                            device=local_rank, storage=storage, mapping=mapping, cap_tail=tail)
        per_tile = probe.info().device_bytes
        probe.close()
        S = int(min(cap, (frac * free_b) // per_tile * 64))
        return max(64, S // 1024 * 1024 if S >= 1024 else S // 64 * 64)

    def measure(cfg, mapping, S, K, W, seed, ipl=8, storage=0, collective=False, offset=0, variant=_lib.PHM_MCMC_BIGTREE, tkey=None, max_iters=0,
                problem=None):
        """K timed sweeps of configuration `cfg` after args.burnin + W untimed sweeps; returns (block, wall seconds).
        max_iters > K + W: the engine is provisioned for a run of that length (slots, statistics rows), of which K + W sweeps run."""
        W = W + args.burnin
        z, Q, pid, Omega = problem or synth.config_problem(cfg)
        n, E = Q.shape[0], z["edge"].shape[0]
        tiled = mapping == "tiles"
        eng = _lib.Engine(z, Q, pid, Omega, max(K + W, max_iters), variant=variant, seed=seed, n_replicas=S, replica_offset=offset,
                          reduce=True, device=local_rank, iters_per_launch=ipl, storage=storage, mapping=mapping,
                          phase_timing=tiled)
        cols = eng.cols
        eng.run(W, stream)
        eng.sync()
        seg0 = eng.info().seg_read
        barrier()
        t0 = time.perf_counter()
        eng.run(K, stream)                                   # the hot path: K sweeps
        total = None
        if collective:                                       # the only collective: K x cols f64 over RCCL/xGMI
            red_ptr = eng.reduced_stats_device(W, K, stream)  # fixed-order reduction over this GPU's replicas
            try:
                class _Dev:
                    __cuda_array_interface__ = {"shape": (K, cols), "typestr": "<f8", "data": (red_ptr, False), "version": 3}
                total = torch.as_tensor(_Dev(), device=torch.device("cuda", local_rank))
            except (TypeError, ValueError, RuntimeError):    # no zero-copy view: one host round trip instead
                eng.sync()
                total = torch.from_numpy(np.ascontiguousarray(eng.stats(W, K))).to(torch.device("cuda", local_rank))
            if backend != "nccl":
                total = total.cpu()
            parallel.allreduce_stats(total)
            torch.cuda.synchronize()
            total = total.cpu().numpy().copy()               # snapshot before the engine is asked for anything else
        eng.sync()
        barrier()
        dt = time.perf_counter() - t0

        info = eng.info()
        assert info.recoveries == 0, "a capacity recovery (rebuild + replay) happened inside the timed region"
        units = E * S * K                                    # branch x replica paths sampled by this rank
        seg = (info.seg_read - seg0) / units                 # measured mean (m_b + m'_b)
        b_alg = 16 * n + 12 * seg + 26                       # algorithmic bytes per branch x replica x sweep
        kernel_s = info.last_run_ms / 1e3
        stats = eng.stats(W, K)
        # sanity: dwell row sums = S x tree length; the matrix handed to RCCL is the same matrix (summed over ranks)
        assert np.allclose(stats[:, :n].sum(1), S * z["edge.length"].sum(), rtol=1e-9)
        if total is not None and world == 1:
            assert np.array_equal(total, stats)
        tr = traffic.get(tkey or f"C{cfg}", {})
        per = E * S                                          # units of one sweep
        blk = {"workload": tkey or f"C{cfg}", "n_states": n, "n_tips": int(z["states"].size), "branches": E, "replicas": S, "mapping": mapping,
               "ms_per_sweep": dt / K * 1e3, "realisations_per_s": units / dt, "hbm_gib_resident": info.device_bytes / 2 ** 30,
               "replicas_per_gib": S / (info.device_bytes / 2 ** 30), "mean_segments_read_plus_written": seg, "recoveries": int(info.recoveries),
               "max_iters_provisioned": int(info.max_iters)}
        sweep = {"bound": "hbm", "definition": "SURVEY 8(d): E * S * B_alg / (HIP-event time of all kernels of one sweep), B_alg = 16 n + 12 mean(m + m') + 26",
                 "alg_bytes_per_unit": b_alg, "units_per_launch": per, "launches": K, "avg_launch_ms": info.last_run_ms / K,
                 "achieved": units * b_alg / kernel_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                 "frac": units * b_alg / kernel_s / 1e9 / HBM_PEAK_GBS, "kernel_launches_per_sweep": info.last_run_launches / K,
                 "frac_of_measured_triad": units * b_alg / kernel_s / 1e9 / MEASURED_TRIAD_GBS,
                 "traffic": (tr["all_kernels_bytes_per_unit"] * per) if tr.get("all_kernels_bytes_per_unit") else None,
                 "traffic_source": tr.get("source")}
        if tiled:
            up_ms, down_ms, br_ms, st_ms = [v / K for v in eng.phase_ms()]

            def kblock(needles, ms, alg_bytes, what):
                bpu, valu, names = kernel_traffic(tr, *needles)
                e = {"kernel": ", ".join(names) if names else "/".join(needles), "what": what, "bound": "hbm", "alg_bytes_per_unit": alg_bytes,
                     "units_per_launch": per, "launches": K, "avg_launch_ms": ms,
                     "achieved": per * alg_bytes / (ms / 1e3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": per * alg_bytes / (ms / 1e3) / 1e9 / HBM_PEAK_GBS, "traffic": (bpu * per) if bpu else None}
                if bpu:
                    e["counter_traffic_frac"] = bpu * per / (ms / 1e3) / 1e9 / HBM_PEAK_GBS
                if valu:      # VALU issue: instructions of the PMC run, scaled to this run's units, at the measured issue rate
                    scale = per / tr["units_per_sweep_in_pmc_run"]
                    e["valu_issue_frac"] = valu * scale * VALU_CLOCKS / (N_SIMD * NOMINAL_GHZ * 1e9 * ms / 1e3)
                return e
            branch = kblock(("_branch_kernel",), br_ms, 12 * seg + 8, "resamplebranchstates + shortener + virtual jumps + dwell sums (src/phylomap.cpp:264-413, :44-73, :745-757); one launch per sweep")
            prune = kblock(("_up_", "_up2_", "sparse_up"), up_ms, 12 * n + 12, "makePLrcpp* (src/phylomap.cpp:503-529), all height levels of one sweep")
            draws = kblock(("_down", "_root_kernel"), down_ms, 4 * n + 6, "sampleinternalnodes* + updatenodestates (src/phylomap.cpp:618-657, :460-475), all depth levels")
            red = kblock(("_stats_kernel", "_chunk_kernel"), st_ms, 0.0, "statistics rows (fixed-order reductions)")
            if n > 4 and eng.info().sparse_chains & 1 == 0:   # 5..64 states, dense B: the pruning chains run on the matrix cores
                n_int = E // 2 - 1                           # branches whose child is an internal node
                flops = 2.0 * n * n * max(seg / 2.0 - 1.0, 0.0) * n_int * S      # 2 n^2 (m - 1) per internal-child branch
                prune["mfma"] = {"bound": "mfma", "achieved": flops / (up_ms / 1e3) / 1e12, "peak": MFMA_F64_PEAK_TFLOPS, "unit": "TFLOP/s",
                                 "frac": flops / (up_ms / 1e3) / 1e12 / MFMA_F64_PEAK_TFLOPS, "alg_flops_per_sweep": flops,
                                 "note": "algorithmic flops 2 n^2 (mean m - 1) per internal-child branch; the kernel issues max-over-16-replicas steps of 64-padded tiles"}
            kernels = {"branch": branch, "pruning": prune, "node_draws": draws, "reductions": red}
            dom = max(("branch", "pruning", "node_draws"), key=lambda k: kernels[k]["avg_launch_ms"])
            blk["phases_ms_per_sweep"] = {"pruning_levels": up_ms, "node_draws": down_ms, "branch_kernel": br_ms, "reductions": st_ms}
            blk["roofline"] = dict(sweep, dominant_kernel=dict(kernels[dom], phase=dom), kernels=kernels)
            blk["pruning_sweep"] = prune
        else:
            per_l = E * S * ipl
            bpu, valu, names = kernel_traffic(tr, "mcmc_sweep_kernel")
            blk["roofline"] = dict(sweep, kernel=f"mcmc_sweep_kernel<{n}> (the whole sweep fused, {ipl} sweeps per launch)",
                                   launches=info.last_run_launches, avg_launch_ms=info.last_run_ms / max(1, info.last_run_launches),
                                   units_per_launch=per_l, measured_triad_peak=MEASURED_TRIAD_GBS)
            if valu:
                blk["roofline"]["valu_issue_frac"] = valu * (per / tr["units_per_sweep_in_pmc_run"]) * VALU_CLOCKS / (N_SIMD * NOMINAL_GHZ * 1e9 * info.last_run_ms / K / 1e3)
            if n <= 4 and mapping == "replicas":
                prune_ms = eng.time_pruning(8, stream) / 8.0
                blk["pruning_sweep"] = {"kernel": f"mcmc_sweep_kernel<{n}> (up sweep only)", "ms_per_sweep": prune_ms,
                                        "alg_bytes_per_unit": 12 * n + 12, "bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                        "achieved": E * S * (12 * n + 12) / (prune_ms / 1e3) / 1e9,
                                        "frac": E * S * (12 * n + 12) / (prune_ms / 1e3) / 1e9 / HBM_PEAK_GBS}
        eng.close()
        return blk, dt

    def one_chain(cfg, sweeps):
        """the reference's own calling pattern: ONE chain (every R/sumstat*.R call), on the mapping the library picks for it"""
        z, Q, pid, Om = synth.config_problem(cfg)
        E = z["edge"].shape[0]
        res = {}
        for S1 in (1, 8):                                 # 8: a handful of chains on the same mapping (options(phylomap.hip.replicas = 8))
            one = _lib.Engine(z, Q, pid, Om, sweeps + 40, variant=_lib.PHM_MCMC_BIGTREE, seed=1, n_replicas=S1, device=local_rank)
            one.run(40); one.sync()                       # the paths reach their stationary length (two segments per branch at the start)
            t1 = time.perf_counter(); one.run(sweeps); one.sync(); d1 = time.perf_counter() - t1
            mp = {1: "one lane per replica", 2: "branch mapping: subtree clusters (8 lanes per node), transition maps + walk, 8 lanes per branch (n <= 4) / "
                                                "one wave per (replica, branch) (n > 4)", 3: "one lane per replica, wave per (tile, item)"}[one.info().mapping]
            one.close()
            res[S1] = (mp, d1 / sweeps * 1e3, S1 * E * sweeps / d1)
        return {"workload": f"C{cfg}, ONE chain: what the R call sumstatMCMC_bigtree(z, Q, pid, Omega, N) gets (R/sumstatMCMC_bigtree.R:21-29)",
                "mapping": res[1][0], "ms_per_sweep": res[1][1], "realisations_per_s": res[1][2],
                "eight_chains": {"ms_per_sweep": res[8][1], "realisations_per_s": res[8][2]}}

    # ---- headline ------------------------------------------------------------------------------------------------
    K, W = args.steps, args.warmup
    cfg = args.config
    mapping = args.mapping or ("replicas" if cfg == 2 else "tiles")
    zc, Qc, pidc, Omc = synth.config_problem(cfg)
    sto = args.storage if mapping == "replicas" else 0
    S_req = args.replicas
    if S_req <= 0:
        cap = {2: 393216, 3: 16384, 4: 65536, 5: 16384}.get(cfg, 16384)
        S_req = sized_replicas(zc, Qc, pidc, Omc, mapping, cap, storage=sto, max_iters=K + W + args.burnin)
    if world > 1:      # one decision for the whole job: rank 0's (ranks sizing from their own free HBM could disagree)
        t = torch.tensor([S_req], dtype=torch.int64, device=dist_dev)
        dist.broadcast(t, src=0)
        S_req = int(t.item())
    if args.scaling == "strong":
        offset, S = parallel.split_replicas(S_req, world, rank)
        S_total = S_req
    else:
        offset, S = parallel.weak_shard(S_req, rank)
        S_total = S_req * world
    if world > 1:      # fail loudly if the ranks do not tile [0, S_total) exactly
        mine = torch.tensor([offset, S], dtype=torch.int64, device=dist_dev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        spans = sorted((int(a[0]), int(a[1])) for a in allr)
        pos = 0
        for o_, s_ in spans:
            if o_ != pos or s_ < 1:
                raise SystemExit(f"bench.py: ranks disagree on the replica sharding: {spans} does not tile [0, {S_total})")
            pos += s_
        if pos != S_total:
            raise SystemExit(f"bench.py: ranks disagree on the replica sharding: {spans} does not tile [0, {S_total})")
    head, dt = measure(cfg, mapping, S, K, W, 0x5EED0000 + cfg, ipl=args.ipl, storage=sto,
                       collective=(world > 1 or args.force_collective), offset=offset)
    tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    out = None
    if rank == 0:
        n, E = head["n_states"], head["branches"]
        names = {2: "C2: sumstatMCMC sweep (row-normalised PL)", 3: "C3: sumstatMCMC_bigtree sweep", 4: "C4: dense 61-state sweep (row-normalised PL)",
                 5: "C5: sparse 20-state sweep (row-normalised PL)"}
        out = {
            "metric": "stochastic-map realisations/sec (branches x sites sampled/s)",
            "value": E * S_total * K / dt, "unit": "branch-site realisations/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": dt / K * 1e3,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{names.get(cfg, f'C{cfg}')}, {n}-state Q, {head['n_tips']}-tip synthetic tree, Omega*mean(t_b)=4",
                       "n_states": n, "n_tips": head["n_tips"], "branches": E, "replicas_total": S_total, "replicas_rank0": S,
                       "mapping": {"tiles": "one wave per (tile of 64 replicas, branch)", "replicas": "one lane per replica, one wave per tile walks the tree"}.get(mapping, mapping),
                       "parallelism": f"replica-sharded x{world} ({args.scaling}), one RCCL all-reduce of the statistics"},
            "roofline": head["roofline"],
            "phases_ms_per_sweep": head.get("phases_ms_per_sweep"),
            "pruning_sweep": head.get("pruning_sweep"),
            "hbm_bytes_resident": int(head["hbm_gib_resident"] * 2 ** 30),
            "replicas_per_gib": head["replicas_per_gib"],
            "recoveries": head["recoveries"],
        }
        out["config"]["max_iters_provisioned"] = head["max_iters_provisioned"]
        if cfg in cpu:
            out["cpu_baseline"] = cpu[cfg]
            out["speedup_vs_cpu_1core"] = out["value"] / cpu[cfg]["value"]
            out["speedup_vs_cpu_1core_faithful"] = out["value"] / cpu[cfg]["faithful_value"]
            out["speedup_vs_cpu_all_cores"] = out["value"] / cpu[cfg]["all_cores"]["value"]
    if rank == 0 and world == 1 and not args.no_extras and cfg == 3:
        # The headline at the configuration's STATED length (BASELINE configs[2]: 10 000 iterations): an engine provisioned for
        # max_iters = 10 000 -- slot tail 0.05 / (S E 10 000) per draw, 10 000 statistics rows per tile -- of which the same
        # burn-in + warm-up + K sweeps run.  Fewer replicas fit per GiB than in the 37-sweep engine above; both are reported.
        N_stated = 10000
        S10 = sized_replicas(zc, Qc, pidc, Omc, mapping, 16384, storage=sto, max_iters=N_stated)
        blk10, _ = measure(cfg, mapping, S10, K, W, 0x5EED0000 + cfg, ipl=args.ipl, storage=sto, max_iters=N_stated)
        out["stated_length"] = {"max_iters": N_stated, "replicas": S10, "replicas_per_gib": blk10["replicas_per_gib"],
                                "hbm_gib_resident": blk10["hbm_gib_resident"], "ms_per_sweep": blk10["ms_per_sweep"],
                                "realisations_per_s": blk10["realisations_per_s"], "roofline_frac": blk10["roofline"]["frac"],
                                "recoveries": blk10["recoveries"]}
    if rank == 0 and world == 1:
        sc = one_chain(cfg, 400)
        if cfg in cpu:
            sc["speedup_vs_cpu_1core"] = sc["realisations_per_s"] / cpu[cfg]["value"]
            sc["speedup_vs_cpu_1core_faithful"] = sc["realisations_per_s"] / cpu[cfg]["faithful_value"]
        out["single_chain"] = sc
        # the drop-in's own shape: ONE reference-shaped call (host buffers in, host matrix out) with the replica axis switched on --
        # upload, engine set-up, N sweeps, reduction, download all inside the clock; never part of `value`
        if not args.no_extras:
            Nc, Sc = 500, 4096
            t1 = time.perf_counter()
            ss = api.sumstatMCMC_bigtree(zc, Qc, pidc, Omc, Nc, seed=1, n_replicas=Sc, reduce=True, device=local_rank)
            wall = time.perf_counter() - t1
            assert ss.shape == (Nc, head["n_states"] + head["n_states"] * (head["n_states"] - 1))
            out["one_shot_call"] = {"workload": f"C{cfg}: one phm_maketreelistMCMC_bigtree call, {Sc} chains summed, N = {Nc} sweeps, host buffers in / host matrix out "
                                                f"(upload, engine set-up, sweeps, reduction, download inside the clock)",
                                    "wall_s": wall, "realisations_per_s_incl_setup_and_copies": head["branches"] * Sc * Nc / wall}

    # ---- the other BASELINE configurations, one block each (N = 1 only) -----------------------------------------------
    if rank == 0 and world == 1 and not args.no_extras:
        blocks = {}
        # (cfg, mapping, cap, K, W, storage, variant, key): C4 in the n + n^2 counting layout (shortenerbf: self pairs counted) SURVEY 8 scopes it in
        plan = [(2, "replicas", 393216, 16, 8, 2, _lib.PHM_MCMC_BIGTREE, "C2"), (4, "tiles", 65536, 6, 6, 0, _lib.PHM_MCMC_BF, "C4bf"),
                (5, "tiles", 16384, 6, 8, 0, _lib.PHM_MCMC_BIGTREE, "C5")]
        if cfg == 2:
            plan[0] = (3, "tiles", 16384, 12, 6, 0, _lib.PHM_MCMC_BIGTREE, "C3")
        # C5's tree size with an UNSTRUCTURED sparse Q (a degree-6 neighbour graph: BASELINE configs[4] says "amino-acid Q"): pruning in
        # the kernel generated for the matrix's pattern (phm_rtc.h) instead of on the matrix cores
        Qn = synth.neighbour_Q(20, 6)
        Omn = 1.25 * float(np.max(np.abs(np.diag(Qn))))
        pidn = np.full(20, 0.05)
        problems = {"C5_unstructured": (synth.make_tree(5000, Qn, Omn, 0x5EED0005, pidn, init_segments=20), Qn, pidn, Omn)}
        plan.append((5, "tiles", 16384, 6, 8, 0, _lib.PHM_MCMC_BIGTREE, "C5_unstructured"))
        for c, mp, capS, k, w, sto_c, var, key in plan:
            z, Q, pid, Om = problems.get(key) or synth.config_problem(c)
            Sx = sized_replicas(z, Q, pid, Om, mp, capS, frac=0.80, storage=sto_c, variant=var, max_iters=k + w + args.burnin)
            blk, _ = measure(c, mp, Sx, k, w, 0x5EED0000 + c, ipl=8, storage=sto_c, variant=var, tkey=key, problem=problems.get(key))
            if var == _lib.PHM_MCMC_BF:
                blk["layout"] = "n dwell sums + n x n counts incl. self pairs (shortenerbf, src/phylomap.cpp:997-1028) + root state"
            if c in cpu:
                blk["cpu_baseline"] = cpu[c]
                blk["speedup_vs_cpu_1core"] = blk["realisations_per_s"] / cpu[c]["value"]
            blocks[key] = blk
        # the reference's own calling pattern -- ONE chain -- and an alignment-sized job, on the C2 tree
        blocks["C2_single_chain"] = one_chain(2, 400)
        blocks["C4_single_chain"] = one_chain(4, 60)      # 61 / 20 states: one wave per (replica, branch), state per lane (phm_wbranch.hip)
        blocks["C5_single_chain"] = one_chain(5, 100)
        for key, c in (("C2_single_chain", 2), ("C4_single_chain", 4), ("C5_single_chain", 5)):
            if c in cpu:
                blocks[key]["speedup_vs_cpu_1core"] = blocks[key]["realisations_per_s"] / cpu[c]["value"]
        z, Q, pid, Om = synth.config_problem(2)
        E2 = z["edge"].shape[0]
        mid = _lib.Engine(z, Q, pid, Om, 44, variant=_lib.PHM_MCMC_BIGTREE, seed=1, n_replicas=4096, reduce=True, device=local_rank, mapping="tiles")
        mid.run(4); mid.sync()
        t1 = time.perf_counter(); mid.run(40); mid.sync(); d1 = time.perf_counter() - t1
        mid.close()
        blocks["C2_4096_sites"] = {"mapping": "one wave per (tile, branch)", "ms_per_sweep": d1 / 40 * 1e3, "realisations_per_s": E2 * 4096 * 40 / d1}
        for Sm in (256, 1024):                         # an alignment's sites on the 10 000-tip tree: few tiles (tree passes over level clusters)
            z, Q, pid, Om = synth.config_problem(3)
            mid = _lib.Engine(z, Q, pid, Om, 52, variant=_lib.PHM_MCMC_BIGTREE, seed=1, n_replicas=Sm, reduce=True, device=local_rank, mapping="tiles")
            mid.run(12); mid.sync()
            t1 = time.perf_counter(); mid.run(40); mid.sync(); d1 = time.perf_counter() - t1
            nl = mid.info().last_run_launches // 40
            mid.close()
            blocks[f"C3_{Sm}_sites"] = {"mapping": "one wave per (tile, branch)", "ms_per_sweep": d1 / 40 * 1e3, "launches_per_sweep": nl,
                                        "realisations_per_s": z["edge"].shape[0] * Sm * 40 / d1}

        # The reference's own flagship run: the 3 951-tip squamate tree at Omega = 10 (vignettes/Squamate_DIC_model_selection.Rnw:76-120):
        # ~111 segments per branch, 2 280 on the longest -- one chain (what the R call gets) and 1 024 chains (automatic mapping)
        sq = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden", "squamate", "seed101_tips.npz")
        if os.path.exists(sq):
            d = np.load(sq)
            Tq = len(d["states"])
            zq = {"edge": d["edge"], "Nnode": Tq - 1, "edge.length": d["edge_length"], "states": d["states"]}
            zq["maps"] = [np.full(100, l / 100) if c > Tq else np.full(2, l / 2) for (p_, c), l in zip(d["edge"], d["edge_length"])]
            zq["mapnames"] = [np.ones(100, dtype=np.int32) if c > Tq else np.array([1, d["states"][c - 1]], dtype=np.int32) for (p_, c) in d["edge"]]
            Qq = np.array([[-0.001, 0.001], [0.006, -0.006]])
            for key, Sq, Kq in (("squamate_one_chain", 1, 100), ("squamate_1024_chains", 1024, 12)):
                mid = _lib.Engine(zq, Qq, [.5, .5], 10.0, Kq + 24, variant=_lib.PHM_MCMC_BIGTREE, seed=1, n_replicas=Sq, reduce=Sq > 1, device=local_rank)
                mid.run(24); mid.sync()
                i0 = mid.info()
                t1 = time.perf_counter(); mid.run(Kq); mid.sync(); d1 = time.perf_counter() - t1
                i1 = mid.info()
                assert i1.recoveries == 0
                blocks[key] = {"ms_per_sweep": d1 / Kq * 1e3, "realisations_per_s": d["edge"].shape[0] * Sq * Kq / d1,
                               "segments_per_s": (i1.seg_read - i0.seg_read) / d1, "mapping": {v: k for k, v in _lib.MAPPING.items()}.get(i1.mapping, i1.mapping)}
                mid.close()

        # sumstatEXP (src/phylomap.cpp:3001-3051): C1 as stated, and the 1 000-tip 4-state tree with the rescaled pruning pass
        L = _lib.load()
        for key, c, N, resc in (("EXP_C1", 1, 1 << 16, False), ("EXP_1000_tips", 2, 1 << 16, True)):
            z, Q, pid, Om = synth.config_problem(c)
            n, E = Q.shape[0], z["edge"].shape[0]
            api.sumstatEXP(z, Q, pid, 64, seed=1, rescale=resc, device=local_rank)
            t1 = time.perf_counter()
            st = api.sumstatEXP(z, Q, pid, N, seed=2, rescale=resc, device=local_rank)
            wall = time.perf_counter() - t1
            kms = float(L.phm_last_kernel_ms())
            assert np.allclose(st[:, :n].sum(1), z["edge.length"].sum(), rtol=1e-9)
            nj = float(st[:, n:].sum()) / (N * E)                    # real jumps per branch
            alg = 16 * n + 2 + 12 * (1 + nj)                          # P row 8n + PL row 8n + state 2 B, segments written 12 B each
            t1 = time.perf_counter()
            api.sumstatEXP(z, Q, pid, 1000, seed=3, rescale=resc, device=local_rank)      # the sample count R users call it with
            call_1000 = time.perf_counter() - t1
            tr = traffic.get(key, {})
            bpu, valu, knames = kernel_traffic(tr, "exp_tiles_", "exp_sample", "exp_wide")
            rl = {"bound": "valu", "kernel": ", ".join(knames) if knames else "exp_tiles_branch_kernel (+ root, node levels, finish)", "launches": 1, "avg_launch_ms": kms,
                  "alg_bytes_per_unit": alg, "units_per_launch": E * N, "achieved": E * N * alg / (kms / 1e3) / 1e9,
                  "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": E * N * alg / (kms / 1e3) / 1e9 / HBM_PEAK_GBS,
                  "traffic": (bpu * E * N) if bpu else None, "traffic_source": tr.get("source")}
            if valu:      # the bound claimed: VALU issue = instructions (PMC run, scaled by units) x measured issue clocks / kernel time
                rl["valu_issue_frac"] = valu * (E * N / tr["units_per_sweep_in_pmc_run"]) * VALU_CLOCKS / (N_SIMD * NOMINAL_GHZ * 1e9 * kms / 1e3)
                rl["note"] = ("VALU-issue bound: SQ_INSTS_VALU of the sampler's kernels x 4.9 nominal clocks / (1 024 SIMDs x 2.4 GHz) over the kernel time; "
                              "the 300-row B^k e_j table and P(t_b) stay in L2, HBM sees the statistics and the jump-time scratch only")
            blocks[key] = {"workload": f"sumstatEXP, {n}-state Q, {z['states'].size}-tip tree, N = {N} i.i.d. samples" + (", rescaled pruning pass" if resc else ""),
                           "mapping": "one wave per (tile of 64 samples, branch)", "whole_call_ms_at_N_1000": call_1000 * 1e3,
                           "realisations_per_s": E * N / (kms / 1e3), "realisations_per_s_incl_setup_and_copies": E * N / wall, "roofline": rl}
        out["configs"] = blocks

        # secondary metric of BASELINE.json: expm(Q t)/s (batched transition matrices, kernel time)
        n = Qc.shape[0]
        t = np.random.default_rng(0).exponential(4.0 / Omc, 1 << 20)
        Q4 = synth.config_Q(2)
        _, ms_p = api.expm_pade(Q4, t[:1 << 18], device=local_rank)
        _, ms_p = api.expm_pade(Q4, t[:1 << 18], device=local_rank)
        lefts, rights, d = api.eigen_decompose(Q4)
        _, ms_e = api.expm_eigen(lefts, rights, d, t, device=local_rank)
        _, ms_e = api.expm_eigen(lefts, rights, d, t, device=local_rank)
        out["expm_per_s"] = {"n_states": 4, "pade_route": (1 << 18) / (ms_p / 1e3), "eigen_route": t.size / (ms_e / 1e3)}
        # the same metric on the dense 61-state shape of C4, where the products fill MFMA f64 tiles
        Q61 = synth.config_Q(4)
        Q61 = (Q61 + Q61.T) / 2                      # matexp handles a real spectrum only (R/sumstatEXP.R:26-29)
        np.fill_diagonal(Q61, 0.0)
        np.fill_diagonal(Q61, -Q61.sum(1))
        l61, r61, d61 = api.eigen_decompose(Q61)
        t61 = t[: 1 << 16]
        api.expm_eigen(l61, r61, d61, t61, device=local_rank, mfma=True)
        _, ms_m = api.expm_eigen(l61, r61, d61, t61, device=local_rank, mfma=True)
        _, ms_x = api.expm_eigen(l61, r61, d61, t61, device=local_rank)
        api.expm_pade(Q61, t61[:1 << 13], device=local_rank, mfma=True)
        _, ms_pm = api.expm_pade(Q61, t61[:1 << 13], device=local_rank, mfma=True)
        out["expm_per_s"].update({"n61_eigen_route_mfma": t61.size / (ms_m / 1e3), "n61_eigen_route_exact": t61.size / (ms_x / 1e3),
                                  "n61_pade_route_mfma": (1 << 13) / (ms_pm / 1e3),
                                  "n61_mfma_tflops": 2 * 61 ** 3 * t61.size / (ms_m / 1e3) / 1e12, "mfma_f64_peak_tflops": MFMA_F64_PEAK_TFLOPS,
                                  "roofline": {"bound": "mfma", "kernel": "expm_eigen_mfma_kernel", "achieved": 2 * 61 ** 3 * t61.size / (ms_m / 1e3) / 1e12,
                                               "peak": MFMA_F64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": 2 * 61 ** 3 * t61.size / (ms_m / 1e3) / 1e12 / MFMA_F64_PEAK_TFLOPS}})
    if rank == 0:
        # the full detail goes to a FILE (never to stdout / stderr: the driver keeps a few KB of their tail); stdout carries ONE compact line
        ddir = os.path.join(ROOT, "gpurun_out")
        try:
            os.makedirs(ddir, exist_ok=True)
            with open(os.path.join(ddir, f"bench_detail_n{world}.json"), "w") as f:
                json.dump(out, f, indent=1)
        except OSError:
            pass
        print(json.dumps(compact_line(out)), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
