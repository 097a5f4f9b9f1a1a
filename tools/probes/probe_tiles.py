"""Per-kernel split of the wave-per-(tile, branch) mapping at one replica count (run under rocprofv3 --kernel-trace)."""
import sys, time, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
from phylomap_amd import _lib, synth
S = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
mapping = sys.argv[2] if len(sys.argv) > 2 else "tiles"
cfg = int(sys.argv[3]) if len(sys.argv) > 3 else 2
z, Q, pid, Om = synth.config_problem(cfg)
eng = _lib.Engine(z, Q, pid, Om, 24, variant=_lib.PHM_MCMC_BIGTREE, seed=1, n_replicas=S, mapping=mapping, reduce=True)
eng.run(4); eng.sync()
t = time.time(); eng.run(20); eng.sync(); dt = time.time() - t
E = z["edge"].shape[0]
print(f"C{cfg} {mapping} S={S}: {1e3*dt/20:.3f} ms/sweep {S*E*20/dt/1e9:.3f} G units/s, {eng.info().device_bytes/2**30:.1f} GiB")
