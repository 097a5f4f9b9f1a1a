import sys, time, os, numpy as np
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, root)
from phylomap_amd import _lib
FORM = {'pruning_form': int(os.environ['PHM_PROBE_FORM'])} if 'PHM_PROBE_FORM' in os.environ else {}   # 1 = level barriers, 2 = dependency-driven clusters
d = np.load(root + '/tests/golden/squamate/seed101_tips.npz')
E, T = d["edge"].shape[0], len(d["states"])
z = {"edge": d["edge"], "Nnode": T - 1, "edge.length": d["edge_length"], "states": d["states"]}
z["maps"] = [np.full(100, l / 100) if c > T else np.full(2, l / 2) for (p, c), l in zip(d["edge"], d["edge_length"])]
z["mapnames"] = [np.ones(100, dtype=np.int32) if c > T else np.array([1, d["states"][c - 1]], dtype=np.int32) for (p, c) in d["edge"]]
Q = np.array([[-0.001, 0.001], [0.006, -0.006]])
for S in (1, 8):
    N = 60
    eng = _lib.Engine(z, Q, [.5, .5], 10.0, N + 20, variant=_lib.PHM_MCMC_BIGTREE, seed=1, n_replicas=S, mapping="branches", reduce=S > 1, **FORM)
    eng.run(20); eng.sync()
    t = time.time(); eng.run(N); eng.sync(); dt = time.time() - t
    st = eng.stats(20, N)
    print(f"squamate tree (3 951 tips, Omega = 10), branch mapping, {S} chain(s): {1e3 * dt / N:.3f} ms per sweep; recoveries {eng.info().recoveries}; "
          f"mean segments per branch {float(np.asarray(st)[..., :2].sum()) and 0 or 0}", flush=True)
    eng.close()
