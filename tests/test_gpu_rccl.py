"""The one collective of the multi-GPU path on the real backend: RCCL ("nccl") all-reduce of the statistics matrix, taken
zero-copy from the engine's device buffer exactly as bench.py does at N > 1.  A one-GPU box can only form a world of one
rank, which still loads RCCL, creates the communicator on the device and runs the kernel; the world_size-2 logic is covered
on CPU with gloo (tests/test_distributed_cpu.py)."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_rccl_allreduce_of_device_statistics():
    import torch
    import torch.distributed as dist
    from phylomap_amd import _lib, parallel, synth

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    z, Q, pid, Omega = synth.config_problem(1)
    n, K = Q.shape[0], 4
    cols = n + n * (n - 1)
    eng = _lib.Engine(z, Q, pid, Omega, K, variant=_lib.PHM_MCMC_BIGTREE, seed=7, n_replicas=128, reduce=True, device=0,
                      mapping="replicas")
    stream = torch.cuda.current_stream().cuda_stream
    eng.run(K, stream)
    ptr = eng.reduced_stats_device(0, K, stream)
    eng.sync()
    want = eng.stats(0, K)

    class _Dev:
        __cuda_array_interface__ = {"shape": (K, cols), "typestr": "<f8", "data": (ptr, False), "version": 3}
    total = torch.as_tensor(_Dev(), device=torch.device("cuda", 0))
    assert total.data_ptr() == ptr                      # a view of the engine's buffer, no copy

    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        dist.all_reduce(total, op=dist.ReduceOp.SUM)   # parallel.allreduce_stats skips a world of one; call RCCL itself
        t = torch.tensor([1.5], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.barrier()
        torch.cuda.synchronize()
    finally:
        dist.destroy_process_group()
    assert float(t.item()) == 1.5
    assert np.array_equal(total.cpu().numpy(), want)
    # the buffer handed to RCCL belongs to the caller: reading the statistics again (which re-runs the tile reduction into the
    # engine's own buffer) must not overwrite an all-reduced matrix (bench.py's N > 1 sanity check)
    total.mul_(3.0)
    torch.cuda.synchronize()
    again = eng.stats(0, K)
    assert np.array_equal(again, want)
    assert np.array_equal(total.cpu().numpy(), 3.0 * want)
    assert parallel.weak_shard(128, 3) == (384, 128)
    eng.close()
