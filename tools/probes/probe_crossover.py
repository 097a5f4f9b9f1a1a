import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from phylomap_amd import _lib, synth
for cfg in (3, 2, 1):
    z, Q, pid, Om = synth.config_problem(cfg)
    E = z["edge"].shape[0]
    for S in (16, 64, 96, 128, 192, 256, 384, 512, 1024):
        row = []
        for mapping in ("branches", "tiles"):
            N = 20
            eng = _lib.Engine(z, Q, pid, Om, N + 8, variant=_lib.PHM_MCMC_BIGTREE, seed=1, n_replicas=S, mapping=mapping, reduce=True)
            eng.run(8); eng.sync()
            t = time.time(); eng.run(N); eng.sync(); dt = time.time() - t
            row.append(1e3 * dt / N)
            eng.close()
        print(f"C{cfg} S={S:5d}: branches {row[0]:8.3f} ms  tiles {row[1]:8.3f} ms per sweep", flush=True)
