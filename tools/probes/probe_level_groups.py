"""(tile, branch) mapping, n <= 4: tree passes with one launch per level (level_groups=1) against clusters cut by height band (2) or by
subtree size (3), by replica count: python tools/probes/probe_level_groups.py [cfg ...]   (PHM_PROBE_S="64,256": replica counts)"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from phylomap_amd import _lib, synth
Ss = [int(x) for x in os.environ.get("PHM_PROBE_S", "64,256,1024,2048,4096,8192,16384").split(",")]
for cfg in [int(a) for a in sys.argv[1:]] or [3, 2]:
    z, Q, pid, Om = synth.config_problem(cfg)
    E = z["edge"].shape[0]
    for S in Ss:
        row = []
        for lg in (1, 2, 3):
            N = 20
            eng = _lib.Engine(z, Q, pid, Om, N + 12, variant=_lib.PHM_MCMC_BIGTREE, seed=1, n_replicas=S, mapping="tiles", reduce=True, level_groups=lg,
                              phase_timing=True)
            eng.run(12); eng.sync()
            t = time.time(); eng.run(N); eng.sync(); dt = time.time() - t
            ph = [x / N for x in eng.phase_ms()]
            row.append((1e3 * dt / N, ph, eng.info().last_run_launches // N))
            eng.close()
        print(f"C{cfg} S={S:6d}: " + "   ".join(f"{nm} {r[0]:7.3f} ms ({r[2]} launches; up/down/branch/red {r[1][0]:.3f}/{r[1][1]:.3f}/{r[1][2]:.3f}/{r[1][3]:.3f})"
                                                 for nm, r in zip(("per level", "bands", "subtrees"), row)) + f"   best {E * S / min(r[0] for r in row) / 1e6:.2f} G/s", flush=True)
