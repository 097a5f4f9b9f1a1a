"""Two resident engines on ONE device, each with half of the replicas and its own stream, against one engine with all of them:
do the latency-bound top levels of one half's tree passes hide behind the other half's big kernels?
python tools/probes/probe_two_engines.py [cfg] [replicas] [sweeps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from phylomap_amd import _lib, synth
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
S = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
N = int(sys.argv[3]) if len(sys.argv) > 3 else 40
z, Q, pid, Om = synth.config_problem(cfg)
E = z["edge"].shape[0]
for parts in (1, 2, 4):
    per = S // parts
    engs = [_lib.Engine(z, Q, pid, Om, 2 * N + 8, variant=_lib.PHM_MCMC_BIGTREE, seed=3, n_replicas=per, replica_offset=k * per, reduce=True, mapping="tiles")
            for k in range(parts)]
    streams = [torch.cuda.Stream() for _ in range(parts)]
    def sweep(n):
        for _ in range(n):                            # one sweep per engine in turn: the queues of the streams fill side by side
            for e, s in zip(engs, streams):
                e.run(1, s.cuda_stream)
        for e in engs:
            e.sync()
    sweep(8)
    t0 = time.time(); sweep(N); dt = time.time() - t0
    rec = sum(e.info().recoveries for e in engs)
    print(f"C{cfg} S={S} in {parts} engine(s): {1e3 * dt / N:.3f} ms per sweep = {S * E * N / dt / 1e9:.3f} G/s; recoveries {rec}", flush=True)
    for e in engs:
        e.close()
