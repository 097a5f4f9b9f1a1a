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


def _worker(rank, world, port, mapping, cfg_tips, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch
    import torch.distributed as dist
    from phylomap_amd import _lib, parallel, synth
    r, w, _ = parallel.init_process_group("gloo")
    assert (r, w) == (rank, world)
    z, Q, pid, Omega = synth.config_problem(2, n_tips=cfg_tips)
    off, cnt = parallel.weak_shard(PER_RANK, rank)
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
