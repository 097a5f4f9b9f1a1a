"""Randomised cross-check of every mapping against the CPU oracle (run on a GPU box): random state counts, tree sizes, edge
orders, initial path lengths, replica counts and variants.  Counts must be bit-exact, dwell sums within 1e-10."""
import sys
import numpy as np
import os
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _ROOT); sys.path.insert(0, os.path.join(_ROOT, "tests"))
import oracle_lib as O
from phylomap_amd import api, synth, treeorder

VAR = [("sumstatMCMC", O.PLAIN), ("sumstatMCMC_bigtree", O.BIGTREE), ("SPARSEsumstatMCMC", O.SPARSE)]


def run(seed, n_cases):
    """Returns the number of mismatching (case, mapping, replica) triples."""
    rs = np.random.default_rng(seed)
    bad = 0
    for case in range(n_cases):
        n = int(rs.choice([2, 3, 4, 4, 5, 6, 8, 12, 17, 20, 33, 48, 61]))      # 1 .. 4 row blocks of the matrix-core pruning kernels
        Q = synth.dense_Q(n, 0.02, 0.3, seed=int(rs.integers(1 << 30)))
        band = int(rs.choice([0, 0, 1, 2])) if n >= 5 else 0      # round 3: tridiagonal / pentadiagonal rate matrices (band kernels, n <= 32)
        if band:
            idx = np.arange(n)
            Q[np.abs(idx[:, None] - idx[None, :]) > band] = 0.0
            np.fill_diagonal(Q, 0.0); np.fill_diagonal(Q, -Q.sum(1))
        elif rs.random() < 0.3:                                    # some exactly-zero rates (sparse threshold path)
            mask = rs.random((n, n)) < 0.3
            np.fill_diagonal(mask, False)
            Q[mask] = 0.0
            np.fill_diagonal(Q, 0.0); np.fill_diagonal(Q, -Q.sum(1))
            if np.any(np.diag(Q) == 0):
                continue
        Omega = float(rs.uniform(1.0, 2.5)) * float(np.max(np.abs(np.diag(Q))))
        pid = rs.dirichlet(np.ones(n))
        tips = int(rs.integers(2, 40))
        segs = int(rs.choice([1, 2, 2, 3, 9, 9, 140]))             # 140: paths beyond the windows of the one-chain branch kernel, its wave-wide walk and
                                                                   # the pruning clusters without level barriers (round 4, n <= 4)
        z = synth.make_tree(tips, Q, Omega * float(rs.uniform(0.3, 3.0)), int(rs.integers(1 << 30)), pid, init_segments=max(segs, 2) if tips > 2 else 2)
        if rs.random() < 0.5 and tips > 2:                       # shuffled edge rows
            perm = rs.permutation(len(z["maps"]))
            z = dict(z, edge=z["edge"][perm], **{"edge.length": z["edge.length"][perm]}, maps=[z["maps"][i] for i in perm],
                     mapnames=[z["mapnames"][i] for i in perm])
        nen, nodelist, root = treeorder.pruningwiseedgeorder(z), treeorder.makenodelist(z), treeorder.myreorder(z)
        fn, var = VAR[int(rs.integers(3))]
        if rs.random() < 0.25:                                   # round 3: the n + n^2 counting layout with observed tips, any n
            fn, var = "sumstatMCMCbf_sweep", O.BF
        elif n % 2 == 0 and rs.random() < 0.4:                     # hidden-rates sweep: only the parity of a tip state is observed
            fn, var = "sumstatMCMCks_sweep", O.KS
            z = dict(z, states=((z["states"] - 1) % 2 + 1).astype(np.int32), mapnames=[m.copy() for m in z["mapnames"]])
            T = len(z["states"])
            for b_, (p_, c_) in enumerate(z["edge"]):
                if c_ <= T:
                    z["mapnames"][b_][-1] = z["states"][c_ - 1]
        S, N, seed = int(rs.choice([1, 2, 3, 70, 130])), int(rs.integers(3, 12)), int(rs.integers(1 << 40))
        B = np.eye(n) + Q / Omega
        reps = sorted(set([0, min(S - 1, 1), min(S - 1, 65), S - 1]))          # replicas of different tiles when there are several
        wants = [O.maketreelistMCMC(z, Q, pid, B, Omega, nen, nodelist, root, N, variant=var, seed=seed, replica=r) for r in reps]
        for mapping in ["replicas", "branches", "tiles"]:
            form = int(rs.integers(3)) if mapping != "replicas" else 0   # pruning kernel of the 5..64-state tile mapping / cluster form of the n <= 4 branch mapping
            sc = int(rs.choice([0, 0, 2])) if mapping == "tiles" else 0  # band / pattern-generated kernels where the matrix offers zeros, or dense kernels
            lg = int(rs.integers(4)) if mapping == "tiles" else 0        # round 4: tree passes per level / clusters by height / by subtree size (n <= 4)
            extra = {"devices": [0, 0] if S < 70 else [0, 0, 0]} if (S > 1 and rs.random() < 0.25) else {}      # round 4: replicas sharded inside the call
            try:
                got = getattr(api, fn)(z, Q, pid, Omega, N, seed=seed, n_replicas=S, mapping=mapping, pruning_form=form, sparse_chains=sc, level_groups=lg, **extra)
                if S == 1:
                    got = got[None]
                err = None
            except Exception as ex:      # noqa: BLE001
                got, err = None, ex
            # the replica mapping for 5..64 states keeps a branch's states in an LDS scratch of 128 slots (phm_wide.h): longer paths are refused
            refused = mapping == "replicas" and n > 4 and max(len(mp) for mp in z["maps"]) > 128
            for r, (want, rc) in zip(reps, wants):
                if rc != 0 or refused:
                    ok = err is not None
                elif err is not None:
                    ok = False
                else:
                    ncnt = n * n if var in (O.KS, O.BF) else n * (n - 1)
                    ok = (np.array_equal(got[r][:, n:n + ncnt], want[:, n:n + ncnt]) and np.allclose(got[r][:, :n], want[:, :n], rtol=1e-10, atol=0)
                          and np.allclose(got[r][:, n + ncnt:], want[:, n + ncnt:], rtol=1e-12, atol=0, equal_nan=True))
                if not ok:
                    bad += 1
                    if got is not None and rc == 0:
                        d = np.argwhere(~np.isclose(got[r], want, rtol=1e-10, atol=0, equal_nan=True))
                        print("   differing (row, col):", d[:6].tolist(), "got", [got[r][tuple(x)] for x in d[:3]], "want", [want[tuple(x)] for x in d[:3]])
                    print(f"MISMATCH case {case}: n={n} band={band} tips={tips} S={S} N={N} {fn} mapping={mapping} form={form} sparse_chains={sc} level_groups={lg} {extra} replica={r} oracle_rc={rc} err={err}")
    return bad


if __name__ == "__main__":
    n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    bad = run(int(sys.argv[1]) if len(sys.argv) > 1 else 0, n_cases)
    print(f"{n_cases} cases done, {bad} mismatches")
    sys.exit(1 if bad else 0)
