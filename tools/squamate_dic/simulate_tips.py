"""Tip data of the reference's one published DIC comparison, regenerated without R.

vignettes/Squamate_DIC_model_selection.Rnw:93 calls ``simulate_2_state_tree(seed=101, atree, Q2, pid2)``
(R/simulate_2_state_tree.R:8-33 -> sample2statehistory, R/sourceme.R:382-414 -> samplethebranch :346-378).  That is:
set.seed(101); one ``sample(1:2, 1, prob=pid)`` for the root (one unif_rand); then, branch by branch in the order
``apply(reorder(tree, "postorder")$edge, 2, rev)``, exponential holding times ``rexp(1, rate)`` until the branch is used
up.  The R stream (set.seed scrambling, Mersenne-Twister, unif_rand, Ahrens-Dieter exp_rand) is the oracle's R-stream
mode; ape's postorder is restated from its published C routine (neworder_phylo / bar_reorder: a node's child edges are
written last-to-first from the END of the order, then each internal child is visited in turn).

Known answer held by the reference: the comment at R/simulate_2_state_tree.R:11 -- "n01 is 21" for seed 101.
TEST TOOLING (uses the oracle's R stream); not part of the product path.
"""
from __future__ import annotations

import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def r_stream(seed: int, n_unif: int, n_exp: int):
    import oracle_lib as O
    u, e = np.zeros(n_unif), np.zeros(n_exp)
    O.lib().orc_rstream_selftest(C.c_uint32(seed), n_unif, n_exp, u.ctypes.data_as(C.POINTER(C.c_double)),
                                 e.ctypes.data_as(C.POINTER(C.c_double)))
    return u, e


def reversed_postorder(edge: np.ndarray, n_tips: int):
    """Row indices (0-based) of ``apply(reorder(tree, "postorder")$edge, 2, rev)``: the order in which ape's bar_reorder
    WRITES edges (it fills the postorder from the end), parents before children."""
    E = edge.shape[0]
    kids = {}
    for r in range(E):
        kids.setdefault(int(edge[r, 0]), []).append(r)
    order, stack = [], [n_tips + 1]
    # recursion: write the node's edges last-to-first, then visit internal children first-to-last (depth first)
    def visit(node):
        rows = kids[node]
        order.extend(reversed(rows))
        for r in rows:
            if edge[r, 1] > n_tips:
                visit(int(edge[r, 1]))
    sys.setrecursionlimit(100000)
    visit(n_tips + 1)
    assert len(order) == E
    return order


def sample2statehistory(z, Q, pid, seed, root_tie_first=1):
    """Returns (tip states 1-based, n01, n10, t0, t1).  ``root_tie_first``: which state R's revsort leaves first when
    pid = (.5, .5) (a tie; the root then takes that state when unif_rand() <= .5)."""
    edge, lens = z["edge"], z["edge.length"]
    T = int(z["Nnode"]) + 1
    u, e = r_stream(seed, 1, 200000)
    nodestates = np.zeros(2 * T - 1, dtype=np.int32)
    p = np.asarray(pid, dtype=float)
    first = root_tie_first if p[0] == p[1] else (1 if p[0] > p[1] else 2)
    pf = p[first - 1] / p.sum()
    nodestates[T] = first if u[0] <= pf else 3 - first             # ProbSampleReplace: rU <= cumulative p of the first sorted entry
    rate = [-Q[0, 0], -Q[1, 1]]
    k = 0
    n01 = n10 = 0
    t = [0.0, 0.0]
    for r in reversed_postorder(edge, T):
        state = int(nodestates[edge[r, 0] - 1])
        bl, dab, i = lens[r], 0.0, 0
        while dab < bl and i < 10000:                                # samplethebranch :356-373
            seg = (1.0 / rate[state - 1]) * e[k]; k += 1             # rexp(1, rate) = exp_rand() / rate
            dab += seg
            if dab < bl:
                t[state - 1] += seg
                n01 += state == 1
                n10 += state == 2
                state = 3 - state                                    # :363
            if dab > bl:
                t[state - 1] += seg - (dab - bl)
            i += 1
        nodestates[edge[r, 1] - 1] = state
    return nodestates[:T].copy(), n01, n10, t[0], t[1], k


def with_simulated_tips(z, tips):
    """The tree as simulate_2_state_tree returns it (R/simulate_2_state_tree.R:15-31): tip branches re-initialised with two
    half-length segments (state 1, then the tip state), internal branches keep their 100 equal segments in state 1."""
    T = int(z["Nnode"]) + 1
    out = dict(z)
    out["states"] = np.asarray(tips, dtype=np.int32)
    maps, names = list(z["maps"]), [np.ones(len(m), dtype=np.int32) for m in z["maps"]]
    for r in range(z["edge"].shape[0]):
        child = int(z["edge"][r, 1])
        if child <= T:
            maps[r] = np.array([z["edge.length"][r] / 2] * 2)
            names[r] = np.array([1, tips[child - 1]], dtype=np.int32)
    out["maps"], out["mapnames"] = maps, names
    return out


if __name__ == "__main__":
    from phylomap_amd import rds
    path = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/inst/extdata/Squamate/phylomap_compatible_squamate_tree.RData"
    z = rds.read_phylomap_tree(path)
    Q2 = np.array([[-0.001, 0.001], [0.006, -0.006]])               # matrix(c(-0.001, 0.006, 0.001, -0.006), nrow = 2)
    for tie in (1, 2):
        tips, n01, n10, t0, t1, used = sample2statehistory(z, Q2, [.5, .5], 101, tie)
        print(f"root tie -> state {tie} first: n01={n01} n10={n10} t0={t0:.3f} t1={t1:.3f} tips in state 2: {(tips == 2).sum()} exp draws {used}")
