"""ms per sweep and per phase (pruning, node draws, branch kernel, reductions) of a (tile, item) mapping:
python tools/probes/probe_phases.py cfg S [iters [sparse_chains [variant]]]   (variant: bigtree | bf | ks; PHM_LIB selects another build of the library)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from phylomap_amd import _lib, synth
cfg = int(sys.argv[1]); S = int(sys.argv[2]); N = int(sys.argv[3]) if len(sys.argv) > 3 else 8
sparse = int(sys.argv[4]) if len(sys.argv) > 4 else 0      # phm_options.sparse_chains: 0 auto, 1 band kernels required, 2 dense
var = {"bigtree": _lib.PHM_MCMC_BIGTREE, "bf": _lib.PHM_MCMC_BF, "ks": _lib.PHM_MCMC_KS}[sys.argv[5] if len(sys.argv) > 5 else "bigtree"]
z, Q, pid, Om = synth.config_problem(cfg)
if os.environ.get("PHM_PROBE_Q") == "neighbour":      # C5's tree size with an UNSTRUCTURED sparse Q (degree-6 neighbour graph: phm_rtc.h)
    import numpy as np
    Q = synth.neighbour_Q(Q.shape[0], 6)
    Om = 1.25 * float(np.max(np.abs(np.diag(Q))))
    z = synth.make_tree(z["states"].size, Q, Om, 0x5EED0005, pid, init_segments=Q.shape[0])
if "PHM_PROBE_N" in os.environ:                       # the configuration's tree size with a dense Q of another state count
    import numpy as np
    nn = int(os.environ["PHM_PROBE_N"])
    Q = synth.dense_Q(nn, 0.005, 0.015, seed=nn)
    Om = 1.25 * float(np.max(np.abs(np.diag(Q))))
    pid = np.full(nn, 1.0 / nn)
    z = synth.make_tree(z["states"].size, Q, Om, 0x5EED0000 + nn, pid, init_segments=2)
E = z["edge"].shape[0]
eng = _lib.Engine(z, Q, pid, Om, N + 10, variant=var, seed=1, n_replicas=S, reduce=True, mapping="tiles", phase_timing=True, sparse_chains=sparse,
                  pruning_form=int(os.environ.get("PHM_PROBE_FORM", "0")))      # PHM_PROBE_FORM: phm_debug_options.pruning_form
def sync():
    try:
        eng.sync()
    except _lib.PhmError as ex:                        # a timing-only build of an experiment may leave the chains in a state the checks refuse
        print("   (sync:", str(ex)[:80], ")")
eng.run(10); sync()
i0 = eng.info()
eng.run(N); sync()
i1 = eng.info()
ph = [x / N for x in eng.phase_ms()]
seg = (i1.seg_read - i0.seg_read) / (E * S * N)
print(f"C{cfg} S={S}: {i1.last_run_ms / N:.2f} ms/sweep = {E * S * N / (i1.last_run_ms / 1e3) / 1e9:.3f} G/s; phases up/down/branch/red = " + " / ".join(f"{x:.2f}" for x in ph) + f"; mean(m+m')={seg:.2f}")
eng.close()
