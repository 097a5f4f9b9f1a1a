"""Write a small synthetic case as CSV files for tools/r_parity/run_reference.R (see README.md)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from phylomap_amd import synth  # noqa: E402

out = sys.argv[1] if len(sys.argv) > 1 else "r_parity_case"
os.makedirs(out, exist_ok=True)
Q = synth.make2sQ(.1, .1, .2, .2, 10)
Omega = 1.25 * float(np.max(np.abs(np.diag(Q))))
pid = np.full(4, 0.25)
z = synth.make_tree(12, Q, Omega, 20260101, pid)
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
print("wrote", out)
