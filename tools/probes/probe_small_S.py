"""ms per sweep of the 5..64-state mappings at small replica counts (tunes the automatic choice): python tools/probe_small_S.py cfg"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from phylomap_amd import _lib, synth
cfg = int(sys.argv[1])
z, Q, pid, Om = synth.config_problem(cfg)
E = z["edge"].shape[0]
for S in (1, 2, 4, 8, 16, 32, 64, 128, 256):
    for mapping in ("branches", "tiles"):
        N = 12
        eng = _lib.Engine(z, Q, pid, Om, N + 6, variant=_lib.PHM_MCMC_BIGTREE, seed=1, n_replicas=S, mapping=mapping, reduce=True)
        eng.run(6); eng.sync()
        t = time.time(); eng.run(N); eng.sync(); dt = time.time() - t
        print(f"C{cfg} {mapping:9s} S={S:4d}: {1e3*dt/N:8.3f} ms/sweep  {S*E*N/dt/1e6:10.2f} M units/s", flush=True)
        eng.close()
