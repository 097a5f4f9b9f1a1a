"""The N > 1 path with the HIP engine on every rank.  A one-GPU box cannot host two RCCL ranks, so two
freshly spawned processes each create an engine on device 0 with `replica_offset = rank * S` (what bench.py --gpus N does
with one rank per GPU), run the sweeps on the GPU and sum their statistics with a gloo all-reduce -- the sharding, the
global replica ids of the Philox streams and the single collective are the real ones, only the transport differs.
The sum must equal ONE engine running all 2 S replicas: counts exactly, dwell sums to 1e-12.

This file sorts first so that the children are started before the pytest process itself has touched the GPU (a process that
has initialised HIP must not start other programs on this pool); conftest.py decides on skipping without a HIP call."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N_ITERS, PER_RANK, SEED = 6, 96, 4711


def _worker(rank, world, port, mapping, cfg_tips, q, cfg=2, total=0):
    """total = 0: weak sharding, PER_RANK replicas each; total > 0: `total` replicas split as evenly as possible (strong)"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch
    import torch.distributed as dist
    from phylomap_amd import _lib, parallel, synth
    r, w, _ = parallel.init_process_group("gloo")
    assert (r, w) == (rank, world)
    z, Q, pid, Omega = synth.config_problem(cfg, n_tips=cfg_tips)
    off, cnt = parallel.split_replicas(total, world, rank) if total else parallel.weak_shard(PER_RANK, rank)
    eng = _lib.Engine(z, Q, pid, Omega, N_ITERS, variant=_lib.PHM_MCMC_BIGTREE, seed=SEED, n_replicas=cnt, replica_offset=off,
                      reduce=True, device=0, mapping=mapping)
    eng.run(N_ITERS); eng.sync()
    t = torch.from_numpy(np.ascontiguousarray(eng.stats(0, N_ITERS)))
    eng.close()
    parallel.allreduce_stats(t)
    dist.barrier()
    if rank == 0:
        q.put(t.numpy().copy())
    dist.destroy_process_group()


@pytest.mark.parametrize("mapping", ["tiles", "replicas"])
def test_two_ranks_with_hip_engines_equal_one_engine(mapping):
    import torch.multiprocessing as mp
    tips = 300
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, mapping, tips, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    # one engine with all 2 S replicas (the parent touches the GPU only now)
    from phylomap_amd import _lib, synth
    z, Q, pid, Omega = synth.config_problem(2, n_tips=tips)
    eng = _lib.Engine(z, Q, pid, Omega, N_ITERS, variant=_lib.PHM_MCMC_BIGTREE, seed=SEED, n_replicas=2 * PER_RANK, reduce=True,
                      device=0, mapping=mapping)
    eng.run(N_ITERS); eng.sync()
    want = eng.stats(0, N_ITERS)
    eng.close()
    np.testing.assert_array_equal(got[:, 4:], want[:, 4:])
    np.testing.assert_allclose(got[:, :4], want[:, :4], rtol=1e-12, atol=0)
    np.testing.assert_allclose(got[:, :4].sum(1), 2 * PER_RANK * z["edge.length"].sum(), rtol=1e-11)


@pytest.mark.parametrize("cfg,tips,mapping,total", [(2, 300, "tiles", 201), (5, 150, "tiles", 203), (4, 60, "tiles", 130), (5, 150, "branches", 14)])
def test_four_ranks_with_unequal_shards_equal_one_engine(cfg, tips, mapping, total):
    """Strong split over FOUR ranks (parallel.split_replicas: 51 + 50 + 50 + 50 of 201 replicas, i.e. shards that are neither
    equal nor whole tiles), for the n <= 4 (tile, branch) mapping and for the 5..64-state mappings (20 states banded: band
    kernels; 61 states dense: matrix cores; a wave per (replica, branch)).  Four HIP engines share device 0, the statistics meet
    in one gloo all-reduce; the result must be ONE engine running all the replicas: counts exactly, dwell sums to 1e-12."""
    import torch.multiprocessing as mp
    world = 4
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, mapping, tips, q, cfg, total)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    from phylomap_amd import _lib, synth
    z, Q, pid, Omega = synth.config_problem(cfg, n_tips=tips)
    n = Q.shape[0]
    eng = _lib.Engine(z, Q, pid, Omega, N_ITERS, variant=_lib.PHM_MCMC_BIGTREE, seed=SEED, n_replicas=total, reduce=True,
                      device=0, mapping=mapping)
    eng.run(N_ITERS); eng.sync()
    want = eng.stats(0, N_ITERS)
    eng.close()
    np.testing.assert_array_equal(got[:, n:], want[:, n:])
    np.testing.assert_allclose(got[:, :n], want[:, :n], rtol=1e-12, atol=0)
    np.testing.assert_allclose(got[:, :n].sum(1), total * z["edge.length"].sum(), rtol=1e-11)


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_n2_rehearsal_on_one_card(scaling):
    """bench.py --gpus 2 as the driver launches it (torch.distributed.run, one rank per GPU), rehearsed on ONE card: both
    ranks on device 0, gloo instead of RCCL (PHM_ALL_RANKS_ON_DEVICE0 / PHM_DIST_BACKEND, never set by the driver).  Checks
    the rank-0 broadcast of the replica count, the sharding check, the barrier-to-barrier timing and the JSON line."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, PHM_ALL_RANKS_ON_DEVICE0="1", PHM_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--burnin", "2",
           "--replicas", "192", "--scaling", scaling, "--no-cpu", "--no-extras"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    j = json.loads(line)
    assert j["n_gpus"] == 2 and j["scaling"] == scaling and j["steps"] == 3
    total = 384 if scaling == "weak" else 192
    assert j["config"]["replicas_total"] == total
    assert abs(j["value"] - 19998 * total * 3 / (j["ms_per_step"] * 3 / 1e3)) < 1e-6 * j["value"]
    assert j["roofline"]["frac"] > 0 and "dominant_kernel" in j["roofline"]
    # a launch whose --gpus disagrees with WORLD_SIZE must refuse to run
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--no-cpu", "--no-extras"],
                         capture_output=True, text=True, timeout=120, env=env, cwd=root)
    assert bad.returncode != 0 and "WORLD_SIZE" in (bad.stderr + bad.stdout)
