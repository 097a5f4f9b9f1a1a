"""One chain on the C2 tree with the branch mapping: wall time per sweep (run under rocprofv3 --kernel-trace to see launches)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from phylomap_amd import _lib, synth
z, Q, pid, Om = synth.config_problem(2)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 400
eng = _lib.Engine(z, Q, pid, Om, N + 8, variant=_lib.PHM_MCMC_BIGTREE, seed=1, n_replicas=1, mapping="branches", reduce=True)
eng.run(8); eng.sync()
t = time.time(); eng.run(N); eng.sync(); dt = time.time() - t
print(f"single chain: {1e3 * dt / N:.4f} ms per sweep")
eng.close()
