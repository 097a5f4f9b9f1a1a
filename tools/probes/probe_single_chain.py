"""One chain with the branch mapping (what a plain R call gets): wall time per sweep on the C2 / C3 / C1 trees
(run under rocprofv3 --kernel-trace to see the launches).  python tools/probes/probe_single_chain.py [sweeps] [cfg ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from phylomap_amd import _lib, synth
FORM = {'pruning_form': int(os.environ['PHM_PROBE_FORM'])} if 'PHM_PROBE_FORM' in os.environ else {}   # 1 = level barriers, 2 = dependency-driven clusters
N = int(sys.argv[1]) if len(sys.argv) > 1 else 400
for cfg in ([int(a) for a in sys.argv[2:]] or [3, 2]):
    z, Q, pid, Om = synth.config_problem(cfg)
    E = z["edge"].shape[0]
    for S in ([int(os.environ["PHM_PROBE_S"])] if "PHM_PROBE_S" in os.environ else (1, 8)):
        eng = _lib.Engine(z, Q, pid, Om, N + 8, variant=_lib.PHM_MCMC_BIGTREE, seed=1, n_replicas=S, mapping="branches", reduce=S > 1, **FORM)
        eng.run(8); eng.sync()
        t = time.time(); eng.run(N); eng.sync(); dt = time.time() - t
        print(f"C{cfg} {S} chain(s), branch mapping: {1e3 * dt / N:.4f} ms per sweep = {S * E * N / dt / 1e6:.1f} M realisations/s", flush=True)
        eng.close()
