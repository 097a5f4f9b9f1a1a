"""Wall time per iteration of the rate-updating drivers (one chain, sweep on the GPU + Gibbs / MH updates on the host every iteration):
sumstatMCMCbf on a 1 000-tip two-state tree, sumstatMCMCks on a 1 000-tip four-state hidden-rates tree; and the fixed-Q sweep beside them."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from phylomap_amd import api, synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
Q2 = np.array([[-.1, .1], [.1, -.1]]); pid2 = np.array([.5, .5])
z2 = synth.make_tree(1000, Q2, 10.0, 15, pid2)
Q4 = synth.make2sQ(.1, .1, [.2], [.2], [10.0]); pid4 = np.full(4, .25)
z4 = synth.make_tree(1000, Q4, 10.0, 16, pid4)
z4 = dict(z4, states=((z4["states"] - 1) % 2 + 1).astype(np.int32))
T = 1000
for b, (p_, c_) in enumerate(z4["edge"]):
    if c_ <= T:
        z4["mapnames"][b][-1] = z4["states"][c_ - 1]
for name, fn, args in (("sumstatMCMCbf  (2 states)", api.sumstatMCMCbf, (z2, Q2, pid2, 10.0, N, [.55, 1, .56, 1.01])),
                       ("sumstatMCMCks  (4 states)", api.sumstatMCMCks, (z4, Q4, pid4, 10.0, N, [1, 10, 2, 10, 20, 2])),
                       ("sumstatMCMC2sDICt (2 states, + log p(y|Q) by matrix exponentials every iteration)", api.sumstatMCMC2sDICt, (z2, Q2, pid2, 10.0, N, [.55, 1, .56, 1.01])),
                       ("sumstatMCMC_bigtree fixed Q (2 states)", api.sumstatMCMC_bigtree, (z2, Q2, pid2, 10.0, N))):
    fn(*args[:4], 20, *args[5:], seed=1)
    t = time.time(); out = fn(*args, seed=2); dt = time.time() - t
    print(f"{name}: {1e3 * dt / N:.3f} ms per iteration (N = {N}, whole call {dt:.2f} s)", flush=True)
