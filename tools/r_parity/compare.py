"""Compare the R package's output (run_reference.R) with the oracle in R-stream mode (see README.md)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402

d = sys.argv[1]
edge = np.loadtxt(os.path.join(d, "edge.csv"), delimiter=",", dtype=np.int32)
el = np.loadtxt(os.path.join(d, "edge_length.csv"))
states = np.loadtxt(os.path.join(d, "states.csv"), dtype=np.int32)
Q = np.loadtxt(os.path.join(d, "Q.csv"), delimiter=",")
pid = np.loadtxt(os.path.join(d, "pid.csv"))
par = np.genfromtxt(os.path.join(d, "params.csv"), delimiter=",", names=True)
maps, names = [], []
for line in open(os.path.join(d, "maps.csv")):
    a, b = line.strip().split(";")
    maps.append(np.array([float(v) for v in a.split()]))
    names.append(np.array([int(v) for v in b.split()], dtype=np.int32))
z = {"edge": edge, "Nnode": states.size - 1, "edge.length": el, "states": states, "maps": maps, "mapnames": names}
nen = np.loadtxt(os.path.join(d, "nen.csv"), dtype=np.int32)
nodelist = np.atleast_1d(np.loadtxt(os.path.join(d, "nodelist.csv"), dtype=np.int32))
root = int(np.loadtxt(os.path.join(d, "root.csv")))
Omega, N, seed = float(par["Omega"]), int(par["N"]), int(par["seed"])
n = Q.shape[0]
ok = True
for name, var in (("sumstatMCMC", O.PLAIN), ("sumstatMCMC_bigtree", O.BIGTREE), ("SPARSEsumstatMCMC", O.SPARSE)):
    want = np.loadtxt(os.path.join(d, name + ".csv"), delimiter=",")
    got, rc = O.maketreelistMCMC(z, Q, pid, np.eye(n) + Q / Omega, Omega, nen, nodelist, root, N, variant=var, seed=seed, rstream=True)
    counts = np.array_equal(got[:, n:], want[:, n:])
    dwell = np.allclose(got[:, :n], want[:, :n], rtol=1e-10, atol=0)
    print(f"{name}: rc={rc} counts {'EXACT' if counts else 'DIFFER'}, dwell {'within 1e-10' if dwell else 'DIFFER'}")
    ok = ok and counts and dwell and rc == 0
# sumstatEXP: set.seed + unif_rand in the order of treesampleEXP / newunifSample (src/phylomap.cpp:2977-2996, :93-208), the
# sorted RcppArmadillo::sample for the node states, sampleOnce for the interior states, Rf_dpois as dpois_raw (R <= 4.0.x;
# R >= 4.1 evaluates it through ebd0: last-ulp differences that move a jump count only when a uniform lands within ~1e-16
# of a partial sum)
if os.path.exists(os.path.join(d, "sumstatEXP.csv")):
    want = np.loadtxt(os.path.join(d, "sumstatEXP.csv"), delimiter=",")
    lefts, rights, dd = (np.loadtxt(os.path.join(d, f + ".csv"), delimiter=",") for f in ("lefts", "rights", "dd"))
    got, rc = O.maketreelistEXP(z, Q, pid, nen, nodelist, root, N, lefts, rights, dd, seed=seed, rstream=True, recompute=True)
    counts = np.array_equal(got[:, n:], want[:, n:])
    dwell = np.allclose(got[:, :n], want[:, :n], rtol=1e-10, atol=0)
    rv = open(os.path.join(d, "R_version.txt")).read().strip() if os.path.exists(os.path.join(d, "R_version.txt")) else "?"
    print(f"sumstatEXP (R {rv}): rc={rc} counts {'EXACT' if counts else 'DIFFER'}, dwell {'within 1e-10' if dwell else 'DIFFER'}")
    ok = ok and counts and dwell and rc == 0
# the rate-updating drivers: tree sweep + Rf_rgamma (Ahrens-Dieter GD / GS, norm_rand by inversion) + runif on the same stream
if os.path.exists(os.path.join(d, "sumstatMCMCbf.csv")):
    parq = np.genfromtxt(os.path.join(d, "params_q.csv"), delimiter=",", names=True)
    OmQ, NQ, seedQ = float(parq["Omega"]), int(parq["N"]), int(parq["seed"])
    maps2, names2 = [], []
    for line in open(os.path.join(d, "maps2.csv")):
        a, b = line.strip().split(";")
        maps2.append(np.array([float(v) for v in a.split()]))
        names2.append(np.array([int(v) for v in b.split()], dtype=np.int32))
    z2 = dict(z, states=np.loadtxt(os.path.join(d, "states2.csv"), dtype=np.int32), maps=maps2, mapnames=names2)
    Q2 = np.loadtxt(os.path.join(d, "Q2.csv"), delimiter=",")
    for name, var, zz, QQ, pp, prior in (("sumstatMCMCbf", O.BF, z2, Q2, np.array([0.5, 0.5]), np.loadtxt(os.path.join(d, "prior_bf.csv"))),
                                         ("sumstatMCMCks", O.KS, z, Q, pid, np.loadtxt(os.path.join(d, "prior_ks.csv")))):
        want = np.loadtxt(os.path.join(d, name + ".csv"), delimiter=",")
        nq = QQ.shape[0]
        got, rc = O.maketreelistMCMC(zz, QQ, pp, np.eye(nq) + QQ / OmQ, OmQ, nen, nodelist, root, NQ, variant=var, seed=seedQ, prior=prior, rstream=True)
        got = got[:, :want.shape[1]]
        counts = np.array_equal(got[:, nq:nq + nq * nq], want[:, nq:nq + nq * nq])
        rest = np.allclose(np.delete(got, np.s_[nq:nq + nq * nq], axis=1), np.delete(want, np.s_[nq:nq + nq * nq], axis=1), rtol=1e-10, atol=0)
        print(f"{name}: rc={rc} counts {'EXACT' if counts else 'DIFFER'}, dwell / rates / root state {'within 1e-10' if rest else 'DIFFER'}")
        ok = ok and counts and rest and rc == 0
sys.exit(0 if ok else 1)
