"""Write a small synthetic case as CSV files for tools/r_parity/run_reference.R (see README.md)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from phylomap_amd import synth  # noqa: E402

out = sys.argv[1] if len(sys.argv) > 1 else "r_parity_case"
n_tips = int(sys.argv[2]) if len(sys.argv) > 2 else 12      # 12: the parity case; 1000 / 10000: BASELINE C2 / C3 sizes for time_reference.R
os.makedirs(out, exist_ok=True)
Q = synth.make2sQ(.1, .1, .2, .2, 10)
Omega = 1.25 * float(np.max(np.abs(np.diag(Q))))
pid = np.full(4, 0.25)
z = synth.make_tree(n_tips, Q, Omega, 20260101, pid)
np.savetxt(os.path.join(out, "edge.csv"), z["edge"], fmt="%d", delimiter=",")
np.savetxt(os.path.join(out, "edge_length.csv"), z["edge.length"], fmt="%.17g")
np.savetxt(os.path.join(out, "states.csv"), z["states"], fmt="%d")
np.savetxt(os.path.join(out, "Q.csv"), Q, fmt="%.17g", delimiter=",")
np.savetxt(os.path.join(out, "pid.csv"), pid, fmt="%.17g")
with open(os.path.join(out, "maps.csv"), "w") as f:          # one branch per line: dwell times ; states
    for d, s in zip(z["maps"], z["mapnames"]):
        f.write(" ".join("%.17g" % v for v in d) + ";" + " ".join(str(int(v)) for v in s) + "\n")
with open(os.path.join(out, "params.csv"), "w") as f:
    f.write("Omega,N,seed\n%.17g,%d,%d\n" % (Omega, 25, 101))
# The rate-updating drivers (Rf_rgamma on R's stream): sumstatMCMCbf on the same tree with two-state data, sumstatMCMCks on the
# four-state case above; Omega = 10 and the priors of the reference's tutorial (vignettes/phylomap_tutorial.Rnw:204-305)
Q2 = np.array([[-0.1, 0.1], [0.1, -0.1]])
st2 = np.asarray(synth.simulate_tips(z["edge"], z["edge.length"], Q2, np.full(2, 0.5), 20260102), dtype=np.int32)      # the same tree, two-state tips
T = st2.size
z2 = {"states": st2, "maps": [], "mapnames": []}
for r in range(z["edge"].shape[0]):                                  # initial paths as R/simulate_2_state_tree.R:19-24: two half-length segments
    child = int(z["edge"][r, 1])
    z2["maps"].append(np.full(2, z["edge.length"][r] / 2))
    z2["mapnames"].append(np.array([1, int(st2[child - 1]) if child <= T else 1], dtype=np.int32))
np.savetxt(os.path.join(out, "states2.csv"), z2["states"], fmt="%d")
np.savetxt(os.path.join(out, "Q2.csv"), Q2, fmt="%.17g", delimiter=",")
with open(os.path.join(out, "maps2.csv"), "w") as f:
    for d, s in zip(z2["maps"], z2["mapnames"]):
        f.write(" ".join("%.17g" % v for v in d) + ";" + " ".join(str(int(v)) for v in s) + "\n")
np.savetxt(os.path.join(out, "prior_bf.csv"), [0.55, 1, 0.56, 1.01], fmt="%.17g")
np.savetxt(os.path.join(out, "prior_ks.csv"), [1, 10, 2, 10, 20, 2], fmt="%.17g")
with open(os.path.join(out, "params_q.csv"), "w") as f:
    f.write("Omega,N,seed\n%.17g,%d,%d\n" % (10.0, 25, 101))
print("wrote", out)
