"""world_size-2 gloo test of the N>1 path: replica sharding + the single statistics all-reduce.
Each rank stands in for a GPU and produces its shard's statistics with the CPU oracle (allowed in tests)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_lib as O
from phylomap_amd import parallel, synth, treeorder

N_ITERS, PER_RANK, SEED = 6, 3, 77


def _problem():
    z, Q, pid, Omega = synth.config_problem(2, n_tips=12)
    return z, Q, pid, Omega, treeorder.pruningwiseedgeorder(z), treeorder.makenodelist(z), treeorder.myreorder(z)


def _shard_stats(offset, count):
    z, Q, pid, Omega, nen, nodelist, root = _problem()
    total = np.zeros((N_ITERS, 16))
    for r in range(offset, offset + count):
        out, rc = O.maketreelistMCMC(z, Q, pid, np.eye(4) + Q / Omega, Omega, nen, nodelist, root, N_ITERS,
                                     variant=O.BIGTREE, seed=SEED, replica=r)
        assert rc == 0
        total += out
    return total


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    r, w, _ = parallel.init_process_group("gloo")
    assert (r, w) == (rank, world)
    off, cnt = parallel.weak_shard(PER_RANK, rank)
    t = torch.from_numpy(_shard_stats(off, cnt))
    parallel.allreduce_stats(t)
    dist.barrier()
    if rank == 0:
        q.put(t.numpy().copy())
    dist.destroy_process_group()


def test_two_rank_sharding_and_allreduce():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = _shard_stats(0, 2 * PER_RANK)                   # one process running every replica
    np.testing.assert_array_equal(got[:, 4:], want[:, 4:])
    np.testing.assert_allclose(got[:, :4], want[:, :4], rtol=1e-12)


def test_shard_arithmetic():
    assert parallel.weak_shard(100, 3) == (300, 100)
    cover = []
    for r in range(3):
        o, c = parallel.split_replicas(10, 3, r)
        cover += list(range(o, o + c))
    assert cover == list(range(10))
    assert parallel.env_rank()[1] >= 1
