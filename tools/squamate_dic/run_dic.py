"""The reference's one published numerical result, re-run: DIC of the 2-state and the 4-state hidden-rates model on the
3 951-tip squamate tree with tip data simulated under seed 101 (vignettes/Squamate_DIC_model_selection.Rnw:76-120:
"the 4-state model with a DIC of 2536.056 compared to the 2-state's DIC of 2538.272").

  python tools/squamate_dic/run_dic.py --model 2 --engine hip    --N 10000     # the product (needs an MI355X)
  python tools/squamate_dic/run_dic.py --model 4 --engine oracle --N 10000     # the CPU oracle

The published values are Monte-Carlo averages over R's RNG stream; a different stream reproduces them up to Monte-Carlo
error only (DIC = 2 mean(-2 log p(y|Q_i)) - D(mean Q), burn-in 1000 of 10000, make2stateDICbig / make4stateDICbig,
R/sourceme.R:445-517, restated below).  The tip data ARE reproduced exactly (simulate_tips.py, pinned by "n01 is 21").
TEST TOOLING; prints one JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
from scipy.linalg import expm

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, HERE)

PUBLISHED = {2: 2538.272, 4: 2536.056}             # vignettes/Squamate_DIC_model_selection.Rnw:120


def loglik(z, Q, pid, nen, root, masks):
    """D's likelihood: pruning with P = expm(Q t_b) and per-node rescaling (R/sourceme.R:452-470, :486-508)."""
    n = Q.shape[0]
    edge, T = z["edge"], int(z["Nnode"]) + 1
    P = {}
    PL = np.zeros((2 * T - 1, n))
    for i, s in enumerate(z["states"]):
        if masks:
            PL[i, (s - 1) % 2::2] = 1.0
        else:
            PL[i, s - 1] = 1.0
    S = 0.0
    for i in range(T - 1):
        a, b = nen[2 * i] - 1, nen[2 * i + 1] - 1
        for r in (a, b):
            if r not in P:
                P[r] = expm(Q * z["edge.length"][r])
        v = (P[a] @ PL[edge[a, 1] - 1]) * (P[b] @ PL[edge[b, 1] - 1])
        S += np.log(v.sum())
        PL[edge[a, 0] - 1] = v / v.sum()
    return float(np.log(PL[root - 1] @ pid) + S)


def dic(mat, z, pid, nen, root, model):
    from phylomap_amd import synth
    if model == 2:
        l01, l10 = mat[:, 6].mean(), mat[:, 7].mean()
        Q = np.array([[-l01, l01], [l10, -l10]])
    else:
        Q = synth.make2sQ(mat[:, 20].mean(), mat[:, 21].mean(), [mat[:, 22].mean()], [mat[:, 23].mean()], [mat[:, 24].mean()])
    D = -2.0 * loglik(z, Q, np.asarray(pid), nen, root, model == 4)
    pD = float(np.mean(-2.0 * mat[:, -1])) - D
    return D + 2.0 * pD, D, pD


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", type=int, choices=(2, 4), default=2)
    ap.add_argument("--engine", choices=("hip", "oracle"), default="hip")
    ap.add_argument("--N", type=int, default=10000)
    ap.add_argument("--seed", type=int, default=101)
    ap.add_argument("--tree", default="/root/reference/inst/extdata/Squamate/phylomap_compatible_squamate_tree.RData")
    ap.add_argument("--tips", default=os.path.join(ROOT, "tests", "golden", "squamate", "seed101_tips.npz"),
                    help="tree + simulated tips as written by --save-tips (used when the RDS file is not available)")
    ap.add_argument("--save-tips", action="store_true")
    ap.add_argument("--save", default=None, help="write the N x cols matrix to this .npy")
    a = ap.parse_args()

    from phylomap_amd import api, synth, treeorder
    if os.path.exists(a.tree):
        import simulate_tips as st
        from phylomap_amd import rds
        z0 = rds.read_phylomap_tree(a.tree)
        Q2 = np.array([[-0.001, 0.001], [0.006, -0.006]])           # matrix(c(-0.001, 0.006, 0.001, -0.006), nrow = 2)
        tips, n01, *_ = st.sample2statehistory(z0, Q2, [.5, .5], 101, root_tie_first=2)
        assert n01 == 21, "R/simulate_2_state_tree.R:11 says n01 is 21 for seed 101"
        z = st.with_simulated_tips(z0, tips)
        if a.save_tips:
            np.savez_compressed(a.tips, edge=z["edge"], edge_length=z["edge.length"], states=z["states"])
            print("wrote", a.tips)
            return
    else:
        d = np.load(a.tips)
        E, T = d["edge"].shape[0], len(d["states"])
        z = {"edge": d["edge"], "Nnode": T - 1, "edge.length": d["edge_length"], "states": d["states"]}
        z["maps"] = [np.full(100, l / 100) if c > T else np.full(2, l / 2) for (p, c), l in zip(d["edge"], d["edge_length"])]
        z["mapnames"] = [np.ones(100, dtype=np.int32) if c > T else np.array([1, d["states"][c - 1]], dtype=np.int32)
                         for (p, c) in d["edge"]]
    nen, nodelist, root = treeorder.pruningwiseedgeorder(z), treeorder.makenodelist(z), treeorder.myreorder(z)
    Omega = 10.0
    if a.model == 2:
        Q, pid, prior = np.array([[-0.001, 0.001], [0.006, -0.006]]), [.5, .5], [.55, 1, .55, 1]
    else:
        Q, pid, prior = synth.make2sQ(0.001, 0.001, [0.001], [0.03], [16]), [.25] * 4, [1, 10, 2, 10, 20, 2]
        z = dict(z)
        # a 4-state run on 2-state data: states 1/2 are the observed trait; initial paths stay in the visible regime
    t0 = time.time()
    if a.engine == "hip":
        fn = api.sumstatMCMC2sDICt if a.model == 2 else api.sumstatMCMCksDICt
        mat = fn(z, Q, pid, Omega, a.N, prior, seed=a.seed)
    else:
        import oracle_lib as O
        n = Q.shape[0]
        mat, rc = O.maketreelistMCMC(z, Q, pid, np.eye(n) + Q / Omega, Omega, nen, nodelist, root, a.N,
                                     variant=O.BF if a.model == 2 else O.KS, seed=a.seed, prior=prior, dic=True)
        assert rc == 0, rc
    secs = time.time() - t0
    if a.save:
        np.save(a.save, mat)
    burn = min(int(np.ceil(a.N / 10)), 1000)                        # R/sourceme.R:599
    val, D, pD = dic(mat[burn:], z, pid, nen, root, a.model)
    ll = mat[burn:, -1]
    # Monte-Carlo standard error of 2*mean(-2 ll) by batch means (20 batches)
    nb = 20
    bm = np.array([b.mean() for b in np.array_split(-2.0 * ll, nb)])
    mcse = 2.0 * bm.std(ddof=1) / np.sqrt(nb)
    print(json.dumps({"model": a.model, "engine": a.engine, "N": a.N, "burn_in": burn, "seed": a.seed, "DIC": round(val, 3),
                      "published_DIC": PUBLISHED[a.model], "diff": round(val - PUBLISHED[a.model], 3),
                      "mc_standard_error": round(float(mcse), 3), "D_at_mean": round(D, 3), "pD": round(pD, 3),
                      "mean_loglik": round(float(ll.mean()), 3), "seconds": round(secs, 1),
                      "ms_per_iteration": round(1e3 * secs / a.N, 2)}))


if __name__ == "__main__":
    main()
