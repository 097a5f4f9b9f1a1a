"""sumstatMCMCmt on a list of 600-tip trees: an engine per tree (the automatic choice for big trees) against the one engine over the
list (a whole tree in one lane).  (The multi-tree drivers prune without rescaling, as the reference's makePLrcppmt does: like the
reference they underflow on trees of thousands of tips.)  python tools/probes/probe_multi_tree.py [n_trees [tips]]"""
import sys, time, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from phylomap_amd import api, synth
nt = int(sys.argv[1]) if len(sys.argv) > 1 else 8
tips = int(sys.argv[2]) if len(sys.argv) > 2 else 600
Q = np.array([[-.1, .1], [.1, -.1]]); Omega, pid, prior = 2.0, np.array([.5, .5]), [.55, 1, .56, 1.01]
trees = synth.make_treelist(nt, tips, Q, 0.5, 314, pid)
for mapping, N in (("auto", 100), ("branches", 100), ("replicas", 20)):
    t = {}
    for n in (N, 2 * N):
        t0 = time.time(); api.sumstatMCMCmt(trees, Q, pid, Omega, n, prior, seed=5, mapping=mapping); t[n] = time.time() - t0
    print(f"{nt} trees of {tips} tips, sumstatMCMCmt, mapping={mapping}: {1e3 * (t[2 * N] - t[N]) / N:.3f} ms per iteration (all trees swept)", flush=True)
