"""Where the wall time of ONE reference-shaped call with the replica axis switched on goes (C3, 4 096 chains, 500 sweeps):
engine set-up, sweeps, read-back."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from phylomap_amd import _lib, synth
z, Q, pid, Om = synth.config_problem(3)
S, N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096, 500
for rep in range(2):
    t0 = time.time()
    eng = _lib.Engine(z, Q, pid, Om, N, variant=_lib.PHM_MCMC_BIGTREE, seed=1, n_replicas=S, reduce=True)
    t1 = time.time()
    eng.run(N); eng.sync()
    t2 = time.time()
    st = eng.stats(0, N)
    t3 = time.time()
    eng.close()
    t4 = time.time()
    print(f"S={S}: create {t1 - t0:.2f} s, {N} sweeps {t2 - t1:.2f} s, read {t3 - t2:.3f} s, destroy {t4 - t3:.2f} s", flush=True)
