"""Do two half-size engines on two HIP streams overlap (tree passes of one under the branch kernel of the other)?
python tools/probe_two_streams.py cfg S_total"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from phylomap_amd import _lib, synth
cfg, S = int(sys.argv[1]), int(sys.argv[2])
z, Q, pid, Om = synth.config_problem(cfg)
E = z["edge"].shape[0]
K, W = 16, 6
def run(parts, shift):
    engs = [_lib.Engine(z, Q, pid, Om, K + W + 2, variant=_lib.PHM_MCMC_BIGTREE, seed=1, n_replicas=S // parts, replica_offset=i * (S // parts),
                        reduce=True, mapping="tiles") for i in range(parts)]
    streams = [torch.cuda.Stream() for _ in range(parts)]
    for e, s in zip(engs, streams): e.run(W, s.cuda_stream)
    if shift and parts > 1: engs[1].run(1, streams[1].cuda_stream)      # de-phase the second engine by half a sweep's worth of launches
    torch.cuda.synchronize()
    t = time.time()
    for i in range(K):                        # interleave the enqueues sweep by sweep
        for e, s in zip(engs, streams): e.run(1, s.cuda_stream)
    torch.cuda.synchronize()
    dt = time.time() - t
    for e in engs: e.sync(); e.close()
    return E * S * K / dt / 1e9
print(f"C{cfg} S={S}: one engine {run(1, False):.3f} G/s; two engines on two streams {run(2, False):.3f} G/s; four {run(4, False):.3f} G/s", flush=True)
