#!/usr/bin/env python3
"""Headline benchmark: stochastic-map realisations/s (branch x site paths sampled per second).

Contract: `python bench.py --gpus N --steps K --warmup W`; for N > 1 the driver launches one rank per GPU with
torch.distributed.run.  A *step* is one full MCMC sweep (pruning + node sampling + branch path resampling +
sufficient statistics) over every replica resident on the GPU.  Workload at N = 1: BASELINE.json configs[1]
(C2: 4-state Q = make2sQ(.1,.1,.2,.2,10), 1000-tip synthetic tree, sumstatMCMC), run with row-normalised
partial likelihoods (the `_bigtree` arithmetic) because the plain variant underflows at 1000 tips in the
reference as well.  Replicas (independent chains / sites) are sharded across ranks with no
data-path collective; the only exchange is one RCCL all-reduce of the K x cols statistics at the end.
Inputs are resident in HBM before the timed region starts.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (about 6.3 TB/s achievable)
MEASURED_TRIAD_GBS = 5751.9   # tools/device_peaks/device_peaks.hip on an MI355X of this pool (read-only 6382, copy 4956 GB/s; FP64 vector 65.5 TFLOP/s)


def cpu_baseline(z, Q, pid, Omega, target_s=12.0, what="C2"):
    """The CPU oracle (oracle/phm_oracle.c, a restatement of src/phylomap.cpp) timed on one host core."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    from phylomap_amd import treeorder
    nen, nodelist, root = treeorder.pruningwiseedgeorder(z), treeorder.makenodelist(z), treeorder.myreorder(z)
    B = np.eye(Q.shape[0]) + Q / Omega
    E = z["edge"].shape[0]

    def run(n_it, faithful):
        t0 = time.perf_counter()
        _, rc = O.maketreelistMCMC(z, Q, pid, B, Omega, nen, nodelist, root, n_it, variant=O.BIGTREE, seed=1,
                                   faithful_search=faithful)
        dt = time.perf_counter() - t0
        assert rc == 0
        return dt

    n_probe = 100 if E < 5000 else 20
    probe = run(n_probe, False)
    n_it = max(n_probe, int(target_s / (probe / n_probe)))
    dt = run(n_it, False)
    n_f = max(50, n_it // 4) if E < 5000 else max(10, n_it // 16)
    dtf = run(n_f, True)
    return {"value": E * n_it / dt, "unit": "branch-site realisations/s", "cores": 1, "kind": "port",
            "sample": f"same {what} tree and Q, 1 chain, {n_it} sweeps (edge lookup table); "
                      f"with the reference's O(E) edge search per node (src/phylomap.cpp:643): "
                      f"{E * n_f / dtf:.4g}/s over {n_f} sweeps",
            "faithful_value": E * n_f / dtf}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=240)
    ap.add_argument("--warmup", type=int, default=24)
    ap.add_argument("--replicas", type=int, default=0, help="replicas per GPU (0: sized from free HBM, max 393216 = 6 waves per SIMD)")
    ap.add_argument("--config", type=int, default=2)
    ap.add_argument("--ipl", type=int, default=8, help="sweeps fused per kernel launch")
    ap.add_argument("--storage", type=int, default=2, help="dwell streams: 2 = two buffers (fastest), 1 = one ring (half the HBM)")
    ap.add_argument("--mapping", default="replicas", choices=["replicas", "tiles", "branches", "auto"],
                    help="how a sweep is laid over the lanes (DESIGN.md 4b); the headline configuration streams with one lane per replica")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-c3", action="store_true", help="skip the extra 10 000-tip measurement")
    ap.add_argument("--force-collective", action="store_true", help="exercise the statistics hand-over to torch at N=1")
    args = ap.parse_args()

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

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

    z, Q, pid, Omega = synth.config_problem(args.config)
    n = Q.shape[0]
    E = z["edge"].shape[0]
    cols = n + n * (n - 1)
    K, W = args.steps, args.warmup

    S = args.replicas
    if S <= 0:
        free_b, _ = torch.cuda.mem_get_info()
        probe = _lib.Engine(z, Q, pid, Omega, 1, variant=_lib.PHM_MCMC_BIGTREE, seed=1, n_replicas=64, reduce=True,
                            device=local_rank, storage=args.storage if args.mapping == "replicas" else 0, mapping=args.mapping)
        per_tile = probe.info().device_bytes
        probe.close()
        S = int(min(393216, (0.80 * free_b) // per_tile * 64))
        S = max(64, S // 16384 * 16384 if S >= 16384 else S // 64 * 64)

    eng = _lib.Engine(z, Q, pid, Omega, K + W, variant=_lib.PHM_MCMC_BIGTREE, seed=0x5EED0000 + args.config,
                      n_replicas=S, replica_offset=parallel.weak_shard(S, rank)[0], reduce=True, device=local_rank,
                      iters_per_launch=args.ipl, storage=args.storage if args.mapping == "replicas" else 0,
                      mapping=args.mapping)   # default: one lane per replica, the streaming layout of the headline configuration
    stream = torch.cuda.current_stream().cuda_stream

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    eng.run(W, stream)
    eng.sync()
    seg0 = eng.info().seg_read

    barrier()
    t0 = time.perf_counter()
    eng.run(K, stream)                                   # the hot path: K sweeps, N-loop on the device
    red_ptr = eng.reduced_stats_device(W, K, stream)     # fixed-order reduction over this GPU's replicas
    if world > 1 or args.force_collective:               # the only collective: K x cols f64 over RCCL/xGMI
        try:
            class _Dev:
                __cuda_array_interface__ = {"shape": (K, cols), "typestr": "<f8", "data": (red_ptr, False), "version": 3}
            total = torch.as_tensor(_Dev(), device=torch.device("cuda", local_rank))
        except (TypeError, ValueError, RuntimeError):    # no zero-copy view: one 30 KB host round trip instead
            eng.sync()
            total = torch.from_numpy(np.ascontiguousarray(eng.stats(W, K))).to(torch.device("cuda", local_rank))
        if backend != "nccl":
            total = total.cpu()
        parallel.allreduce_stats(total)
        torch.cuda.synchronize()
    eng.sync()
    barrier()
    dt = time.perf_counter() - t0

    tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    info = eng.info()
    units_rank = E * S * K                                # branch x replica paths sampled by this rank
    seg = (info.seg_read - seg0) / units_rank             # measured mean (m_b + m'_b)
    b_alg = 16 * n + 12 * seg + 26                        # algorithmic bytes per branch x replica x sweep
    kernel_s = info.last_run_ms / 1e3
    achieved = units_rank * b_alg / kernel_s / 1e9

    # the pruning sweep alone (12n+12 B per branch); timed for the n <= 4 replica kernel only
    prune_ms = eng.time_pruning(8, stream) / 8.0 if (n <= 4 and args.mapping == "replicas") else None
    stats = eng.stats(W, K)
    # sanity: dwell row sums = S x tree length; the tensor handed to RCCL is the same matrix (x world)
    assert np.allclose(stats[:, :n].sum(1), S * z["edge.length"].sum(), rtol=1e-9)
    if world > 1 or args.force_collective:
        tot = total.cpu().numpy()
        assert np.allclose(tot[:, :n].sum(1), world * S * z["edge.length"].sum(), rtol=1e-9)
        if world == 1:
            assert np.array_equal(tot, stats)

    traffic = None
    tfile = os.path.join(ROOT, "profiles", "r01_traffic.json")
    if os.path.exists(tfile):          # HBM bytes per unit from the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes
        tj = json.load(open(tfile))
        if tj.get("config") == args.config:
            traffic = tj["hbm_bytes_per_unit"] * E * S * args.ipl

    out = None
    if rank == 0:
        value = units_rank * world / dt
        out = {
            "metric": "stochastic-map realisations/sec (branches x sites sampled/s)",
            "value": value, "unit": "branch-site realisations/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": dt / K * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"C{args.config}: sumstatMCMC sweep (row-normalised PL), {n}-state Q, "
                                   f"{z['states'].size}-tip synthetic tree, Omega*mean(t_b)=4",
                       "n_states": n, "n_tips": int(z["states"].size), "branches": E,
                       "replicas_per_gpu": S, "sweeps_per_launch": args.ipl,
                       "dwell_storage": "two buffers" if args.storage == 2 else "ring",
                       "parallelism": f"replica-sharded x{world}, one RCCL all-reduce of the statistics"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "mcmc_sweep_kernel<4>", "launches": info.last_run_launches,
                         "avg_launch_ms": info.last_run_ms / max(1, info.last_run_launches),
                         "alg_bytes_per_unit": b_alg, "mean_segments_read_plus_written": seg,
                         "units_per_launch": E * S * args.ipl,
                         # attainable bandwidth measured on this pool with a stream triad (tools/device_peaks, profiles/r01_device_peaks.log)
                         "measured_triad_peak": MEASURED_TRIAD_GBS, "frac_of_measured_triad": achieved / MEASURED_TRIAD_GBS},
            "pruning_sweep": None if prune_ms is None else {
                "kernel": "mcmc_sweep_kernel<4> (up sweep only)", "ms_per_sweep": prune_ms, "alg_bytes_per_unit": 12 * n + 12,
                "achieved": E * S * (12 * n + 12) / (prune_ms / 1e3) / 1e9, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": E * S * (12 * n + 12) / (prune_ms / 1e3) / 1e9 / HBM_PEAK_GBS},
            "hbm_bytes_resident": int(info.device_bytes),
        }
    eng.close()

    if rank == 0 and n <= 4:
        # the reference's own calling pattern -- ONE chain -- on the same tree: the branch-parallel mapping (phm_narrow.hip)
        one = _lib.Engine(z, Q, pid, Omega, 104, variant=_lib.PHM_MCMC_BIGTREE, seed=1, n_replicas=1, device=local_rank,
                          mapping="branches")
        one.run(4); one.sync()
        t1 = time.perf_counter(); one.run(100); one.sync(); d1 = time.perf_counter() - t1
        one.close()
        out["single_chain"] = {"mapping": "one lane per branch", "ms_per_sweep": d1 / 100 * 1e3, "realisations_per_s": E * 100 / d1}

    if rank == 0 and n <= 4:
        # an alignment-sized job (4 096 sites) on the same tree: one wave per (tile of 64 replicas, branch) (phm_tiles.hip)
        mid = _lib.Engine(z, Q, pid, Omega, 44, variant=_lib.PHM_MCMC_BIGTREE, seed=1, n_replicas=4096, reduce=True,
                          device=local_rank, mapping="tiles")
        mid.run(4); mid.sync()
        t1 = time.perf_counter(); mid.run(40); mid.sync(); d1 = time.perf_counter() - t1
        mid.close()
        out["replicas_4096"] = {"mapping": "one wave per (tile, branch)", "ms_per_sweep": d1 / 40 * 1e3,
                                "realisations_per_s": E * 4096 * 40 / d1}

    if rank == 0 and world == 1 and n <= 4 and args.config == 2 and not args.no_c3:
        # BASELINE.json's stated target is quoted on the 10 000-tip 4-state tree (C3): the same sweep with one wave per
        # (tile, branch), as many replicas as fit (at most 16 384), next to the CPU oracle on that same tree.
        z3, Q3, pid3, Om3 = synth.config_problem(3)
        E3 = z3["edge"].shape[0]
        free_b, _ = torch.cuda.mem_get_info()
        probe = _lib.Engine(z3, Q3, pid3, Om3, 1, variant=_lib.PHM_MCMC_BIGTREE, seed=1, n_replicas=64, reduce=True,
                            device=local_rank, mapping="tiles")
        per_tile = probe.info().device_bytes
        probe.close()
        S3 = int(min(16384, (0.80 * free_b) // per_tile * 64))
        S3 = max(64, S3 // 1024 * 1024 if S3 >= 1024 else S3 // 64 * 64)
        big = _lib.Engine(z3, Q3, pid3, Om3, 24, variant=_lib.PHM_MCMC_BIGTREE, seed=0x5EED0003, n_replicas=S3, reduce=True,
                          device=local_rank, mapping="tiles")
        big.run(8); big.sync()
        t1 = time.perf_counter(); big.run(16); big.sync(); d1 = time.perf_counter() - t1
        st3 = big.stats(8, 16)
        assert np.allclose(st3[:, :4].sum(1), S3 * z3["edge.length"].sum(), rtol=1e-9)
        gib3 = big.info().device_bytes / 2 ** 30
        big.close()
        out["ten_k_tip_tree"] = {"workload": "C3: sumstatMCMC_bigtree sweep, 4-state Q, 10000-tip synthetic tree",
                                 "mapping": "one wave per (tile, branch)", "replicas": S3, "ms_per_sweep": d1 / 16 * 1e3,
                                 "realisations_per_s": E3 * S3 * 16 / d1, "hbm_gib_resident": gib3}
        if not args.no_cpu:
            c3 = cpu_baseline(z3, Q3, pid3, Om3, target_s=3.0, what="C3")
            out["ten_k_tip_tree"]["cpu_baseline"] = c3
            out["ten_k_tip_tree"]["speedup_vs_cpu_1core"] = out["ten_k_tip_tree"]["realisations_per_s"] / c3["value"]
            out["ten_k_tip_tree"]["speedup_vs_cpu_1core_faithful"] = out["ten_k_tip_tree"]["realisations_per_s"] / c3["faithful_value"]

    if rank == 0:
        # secondary metric of BASELINE.json: expm(Q t)/s (batched 4x4 transition matrices, kernel time)
        t = np.random.default_rng(0).exponential(4.0 / Omega, 1 << 20)
        npade = (1 << 18) if n <= 8 else (1 << 13)
        _, ms_p = api.expm_pade(Q, t[:npade], device=local_rank)
        _, ms_p = api.expm_pade(Q, t[:npade], device=local_rank)
        out["expm_per_s"] = {"n_states": n, "pade_route": npade / (ms_p / 1e3)}
        try:                                          # the eigen route (matexp, R/sumstatEXP.R:26-29) needs a real spectrum
            lefts, rights, d = api.eigen_decompose(Q)
            nt = t.size if n <= 8 else (1 << 15)
            _, ms_e = api.expm_eigen(lefts, rights, d, t[:nt], device=local_rank)
            _, ms_e = api.expm_eigen(lefts, rights, d, t[:nt], device=local_rank)
            out["expm_per_s"]["eigen_route"] = nt / (ms_e / 1e3)
        except ValueError:
            out["expm_per_s"]["eigen_route"] = None
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
        out["expm_per_s"]["n61_eigen_route_mfma"] = t61.size / (ms_m / 1e3)
        out["expm_per_s"]["n61_eigen_route_exact"] = t61.size / (ms_x / 1e3)
        out["expm_per_s"]["n61_mfma_tflops"] = 2 * 61 ** 3 * t61.size / (ms_m / 1e3) / 1e12
        out["expm_per_s"]["mfma_f64_peak_tflops"] = 78.6
        if world == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(z, Q, pid, Omega)
            out["speedup_vs_cpu_1core"] = out["value"] / out["cpu_baseline"]["value"]
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
