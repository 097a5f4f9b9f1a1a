"""Generates tests/golden/*.npz with the CPU oracle (oracle/phm_oracle.c).

The reference ships no tests or golden vectors and cannot be run here (no R toolchain), so these fixtures are
SELF-GENERATED regression pins of the oracle ("parity unpinned" w.r.t. the R package): inputs (tree, Q, pid,
Omega, seed) and the expected sufficient-statistic matrices.  Run from the repo root:
    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(HERE))

import oracle_lib as O  # noqa: E402
from phylomap_amd import api, synth, treeorder  # noqa: E402

Q3 = np.array([[-.3, .2, .1], [.05, -.15, .1], [.2, .2, -.4]])
CASES = {
    "n2_t16": (synth.config_Q(1), 16, 101),
    "n3_t16": (Q3, 16, 102),
    "n4_t16": (synth.config_Q(2), 16, 103),
    "n4_t64": (synth.config_Q(2), 64, 104),
    "n20_t12": (synth.config_Q(5), 12, 105),
}


def pack_tree(z):
    lens = np.array([len(m) for m in z["maps"]], dtype=np.int32)
    return {"edge": z["edge"], "edge_length": z["edge.length"], "states": z["states"], "map_len": lens,
            "maps": np.concatenate(z["maps"]), "mapnames": np.concatenate(z["mapnames"]).astype(np.int32)}


def unpack_tree(d):
    off = np.concatenate([[0], np.cumsum(d["map_len"])])
    E = d["edge"].shape[0]
    return {"edge": d["edge"], "Nnode": d["states"].size - 1, "edge.length": d["edge_length"], "states": d["states"],
            "maps": [d["maps"][off[i]:off[i + 1]] for i in range(E)],
            "mapnames": [d["mapnames"][off[i]:off[i + 1]] for i in range(E)],
            "node.states": np.ones((E, 2), dtype=np.int32)}


def main():
    for name, (Q, tips, seed) in CASES.items():
        n = Q.shape[0]
        Omega = 1.25 * float(np.max(np.abs(np.diag(Q))))
        pid = np.full(n, 1.0 / n)
        z = synth.make_tree(tips, Q, Omega, seed, pid, init_segments=(n if n == 20 else 2))
        nen, nodelist, root = treeorder.pruningwiseedgeorder(z), treeorder.makenodelist(z), treeorder.myreorder(z)
        B = np.eye(n) + Q / Omega
        d = pack_tree(z)
        d.update(Q=Q, pid=pid, Omega=Omega, seed=np.int64(seed), nen=nen, nodelist=nodelist, root=np.int32(root))
        N = 24
        for key, var in (("mcmc", O.PLAIN), ("bigtree", O.BIGTREE), ("sparse", O.SPARSE)):
            for r in (0, 5):
                out, rc = O.maketreelistMCMC(z, Q, pid, B, Omega, nen, nodelist, root, N, variant=var, seed=seed, replica=r)
                assert rc == 0, (name, key, rc)
                d[f"{key}_r{r}"] = np.ascontiguousarray(out)
        lefts, rights, dm = api.eigen_decompose(Q)
        out, rc = O.maketreelistEXP(z, Q, pid, nen, nodelist, root, 48, lefts, rights, dm, seed=seed)
        assert rc == 0
        d.update(lefts=lefts, rights=rights, d=dm, exp_r0=np.ascontiguousarray(out))
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **d)
        print(name, "ok")


if __name__ == "__main__":
    main()
