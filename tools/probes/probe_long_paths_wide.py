"""5 .. 64 states on paths of ~100 segments per branch: the three mappings at a few chain counts (is the automatic choice sane there?).
python tools/probes/probe_long_paths_wide.py [n_states] [tips] [Omega*t]"""
import sys, time, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from phylomap_amd import _lib, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
T = int(sys.argv[2]) if len(sys.argv) > 2 else 500
lam = float(sys.argv[3]) if len(sys.argv) > 3 else 100.0
Q = synth.dense_Q(n, 0.02, 0.08, seed=n)
Om = 1.25 * float(np.max(np.abs(np.diag(Q))))
pid = np.full(n, 1.0 / n)
z = synth.make_tree(T, Q, Om, 11, pid, init_segments=2)
f = lam / 4.0
z = dict(z, maps=[mp * f for mp in z["maps"]]); z["edge.length"] = z["edge.length"] * f
MAPS = {v: k for k, v in _lib.MAPPING.items()}
for S in (1, 8, 64, 256):
    row = []
    for mapping in ("branches", "tiles", "auto"):
        N = 10
        try:
            eng = _lib.Engine(z, Q, pid, Om, N + 12, variant=_lib.PHM_MCMC_BIGTREE, seed=1, n_replicas=S, mapping=mapping, reduce=True)
            eng.run(12); eng.sync()
            t = time.time(); eng.run(N); eng.sync(); dt = time.time() - t
            info = eng.info()
            row.append(f"{mapping} {1e3 * dt / N:.3f} ms" + (f" (-> {MAPS.get(info.mapping, info.mapping)})" if mapping == "auto" else "") + (f" rec {info.recoveries}" if info.recoveries else ""))
            eng.close()
        except _lib.PhmError as ex:
            row.append(f"{mapping} FAILED {str(ex)[:60]}")
    print(f"n={n} tips={T} Omega*t={lam:g} S={S}: " + "; ".join(row), flush=True)
