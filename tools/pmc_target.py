"""The program the rocprofv3 passes of this round run (kernel trace, and one --pmc pass per counter group):
    python3 tools/pmc_target.py <C2|C3|C4|C4bf|C5|C5dense|C5_unstructured|EXP_C1|EXP_1000_tips> [replicas | samples]
W warm-up sweeps, then K timed sweeps of the same engine bench.py measures (same mapping, same options, reduce = 1); prints one JSON
line with the units (branch x replica) of one sweep.  tools/pmc_summary.py keeps, per kernel, the dispatches of the timed sweeps."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from phylomap_amd import _lib, api, synth  # noqa: E402

W, K = 14, 3      # the first sweeps start from the initial paths (2 or n segments per branch) and cost less / more than stationary ones
DEFAULT_S = {"C2": 393216, "C3": 16384, "C4": 65536, "C4bf": 65536, "C5": 16384, "C5dense": 16384, "C5_unstructured": 16384}


def main():
    name = sys.argv[1]
    if name.startswith("EXP"):
        cfg, resc = (1, False) if name == "EXP_C1" else (2, True)
        N = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 16
        z, Q, pid, Om = synth.config_problem(cfg)
        for _ in range(W):
            api.sumstatEXP(z, Q, pid, N, seed=1, rescale=resc)
        for _ in range(K):
            api.sumstatEXP(z, Q, pid, N, seed=2, rescale=resc)
        print(json.dumps({"name": name, "units_per_sweep": int(z["edge"].shape[0]) * N, "warm": W, "timed": K}))
        return
    cfg = int(name[1])
    S = int(sys.argv[2]) if len(sys.argv) > 2 else DEFAULT_S[name]
    z, Q, pid, Om = synth.config_problem(cfg)
    if name == "C5_unstructured":      # C5's size with a degree-6 neighbour graph instead of the band (the kernel generated for the pattern)
        import numpy as np
        Q = synth.neighbour_Q(20, 6)
        Om = 1.25 * float(np.max(np.abs(np.diag(Q))))
        z = synth.make_tree(5000, Q, Om, 0x5EED0005, pid, init_segments=20)
    variant = _lib.PHM_MCMC_BF if name == "C4bf" else _lib.PHM_MCMC_BIGTREE
    opt = dict(mapping="replicas", storage=2, iters_per_launch=1) if name == "C2" else dict(mapping="tiles")
    if name == "C5dense":
        opt["sparse_chains"] = 2
    if "PHM_PMC_FORM" in os.environ:      # phm_debug_options.pruning_form (kernel variants side by side)
        opt["pruning_form"] = int(os.environ["PHM_PMC_FORM"])
    eng = _lib.Engine(z, Q, pid, Om, W + K, variant=variant, seed=0x5EED0000 + cfg, n_replicas=S, reduce=True, **opt)
    eng.run(W); eng.sync()
    eng.run(K); eng.sync()
    assert eng.info().recoveries == 0
    eng.close()
    print(json.dumps({"name": name, "units_per_sweep": int(z["edge"].shape[0]) * S, "warm": W, "timed": K}))


if __name__ == "__main__":
    main()
