"""The reference's squamate tree (Omega = 10: ~111 segments per branch) at a few chain counts, branch mapping against (tile, branch)
mapping and the automatic choice: python tools/probes/probe_squamate_crossover.py"""
import sys, time, os, numpy as np
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, root)
from phylomap_amd import _lib
d = np.load(root + '/tests/golden/squamate/seed101_tips.npz')
E, T = d["edge"].shape[0], len(d["states"])
z = {"edge": d["edge"], "Nnode": T - 1, "edge.length": d["edge_length"], "states": d["states"]}
z["maps"] = [np.full(100, l / 100) if c > T else np.full(2, l / 2) for (p, c), l in zip(d["edge"], d["edge_length"])]
z["mapnames"] = [np.ones(100, dtype=np.int32) if c > T else np.array([1, d["states"][c - 1]], dtype=np.int32) for (p, c) in d["edge"]]
Q = np.array([[-0.001, 0.001], [0.006, -0.006]])
MAPS = {v: k for k, v in _lib.MAPPING.items()}
for S in (8, 16, 32, 64, 128, 256):
    row = []
    for mapping in ("branches", "tiles", "auto"):
        N = 30
        eng = _lib.Engine(z, Q, [.5, .5], 10.0, N + 20, variant=_lib.PHM_MCMC_BIGTREE, seed=1, n_replicas=S, mapping=mapping, reduce=True)
        eng.run(20); eng.sync()
        t = time.time(); eng.run(N); eng.sync(); dt = time.time() - t
        info = eng.info()
        row.append(f"{mapping} {1e3 * dt / N:.3f} ms" + (f" (-> {MAPS.get(info.mapping, info.mapping)})" if mapping == "auto" else "") + (f" rec {info.recoveries}" if info.recoveries else ""))
        eng.close()
    print(f"squamate S={S}: " + "; ".join(row), flush=True)
