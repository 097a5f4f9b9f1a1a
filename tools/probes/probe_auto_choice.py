"""A sweep over tree shapes, path lengths, state counts and chain counts: the automatic choice of mapping against the branch and the
(tile, item) mappings named explicitly -- prints the cases where the automatic choice is more than 1.5x slower than the better of the two.
python tools/probes/probe_auto_choice.py [seed [cases]]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from phylomap_amd import _lib, synth

MAPS = {v: k for k, v in _lib.MAPPING.items()}


def ladder(T, rs):
    edges = []
    for k in range(T - 1):
        node = T + 1 + k
        edges.append((node, k + 1))
        edges.append((node, node + 1) if k < T - 2 else (node, T))
    return np.asarray(edges, dtype=np.int32)


def tree(shape, T, Q, pid, Omega, lam, rs):
    n = Q.shape[0]
    if shape == "yule":
        z = synth.make_tree(T, Q, Omega, int(rs.integers(1 << 30)), pid, init_segments=2)
        f = lam / 4.0
        z = dict(z, maps=[mp * f for mp in z["maps"]]); z["edge.length"] = z["edge.length"] * f
        return z
    edge = ladder(T, rs)
    lens = rs.exponential(lam / Omega, size=edge.shape[0])
    states = np.asarray(synth.simulate_tips(edge, lens, Q, pid, int(rs.integers(1 << 30))), dtype=np.int32)
    maps = [np.full(2, l / 2) for l in lens]
    mapnames = [np.array([1, states[c - 1] if c <= T else 1], dtype=np.int32) for (p_, c) in edge]
    return {"edge": edge, "Nnode": T - 1, "edge.length": lens, "states": states, "maps": maps, "mapnames": mapnames,
            "node.states": np.ones((edge.shape[0], 2), dtype=np.int32)}


def run(z, Q, pid, Omega, S, mapping, N=4):
    eng = _lib.Engine(z, Q, pid, Omega, N + 4, variant=_lib.PHM_MCMC_BIGTREE, seed=1, n_replicas=S, mapping=mapping, reduce=S > 1)
    eng.run(4); eng.sync()
    t = time.time(); eng.run(N); eng.sync(); dt = (time.time() - t) / N
    m = MAPS.get(eng.info().mapping, "?")
    eng.close()
    return 1e3 * dt, m


def run_lg(z, Q, pid, Omega, S, lg, N=4):
    eng = _lib.Engine(z, Q, pid, Omega, N + 4, variant=_lib.PHM_MCMC_BIGTREE, seed=1, n_replicas=S, mapping="tiles", reduce=True, level_groups=lg)
    eng.run(4); eng.sync()
    t = time.time(); eng.run(N); eng.sync(); dt = (time.time() - t) / N
    nl = eng.info().last_run_launches // N
    eng.close()
    return 1e3 * dt, nl


def passes(rs, cases):
    """third argument "passes": the (tile, item) mapping's tree passes -- automatic form against a launch per level (1), clusters by height
    band (2), clusters by subtree size (3)"""
    bad = 0
    for c in range(cases):
        n = int(rs.choice([2, 4, 4, 8]))
        shape = str(rs.choice(["yule", "yule", "ladder"]))
        T = int(rs.choice([60, 300, 1200, 3000, 10000]))
        lam = float(rs.choice([0.5, 4.0, 4.0, 30.0]))
        S = int(rs.choice([64, 256, 1024, 4096, 16384]))
        if 2.0 * T * (1 + lam) * S > 4e8 or (shape == "ladder" and T > 3000):
            continue
        Q = synth.dense_Q(n, 0.02, 0.3, seed=n) if n > 4 else synth.config_Q(1 if n == 2 else 2)
        if n > 4:                                      # tridiagonal: the band pruning kernel (the one with a cluster form)
            idx = np.arange(n); Q[np.abs(idx[:, None] - idx[None, :]) > 1] = 0.0; np.fill_diagonal(Q, 0.0); np.fill_diagonal(Q, -Q.sum(1))
        Omega = 1.25 * float(np.max(np.abs(np.diag(Q))))
        pid = np.full(n, 1.0 / n)
        z = tree(shape, T, Q, pid, Omega, lam, rs)
        if n > 4:
            z = dict(z, maps=[np.full(n, mp.sum() / n) for mp in z["maps"]],
                     mapnames=[np.array([1] * (n - 1) + [int(mn[-1])], dtype=np.int32) for mn in z["mapnames"]])
        try:
            r = {lg: run_lg(z, Q, pid, Omega, S, lg) for lg in (0, 1, 2, 3)}
        except _lib.PhmError as ex:
            print(f"case {c}: n={n} {shape} tips={T} Omega*t={lam:g} S={S}: {str(ex)[:90]}", flush=True)
            continue
        best = min(r[lg][0] for lg in (1, 2, 3))
        flag = "  <-- automatic form off" if r[0][0] > 1.3 * best and r[0][0] - best > 0.05 else ""
        bad += bool(flag)
        print(f"case {c}: n={n} {shape} tips={T} Omega*t={lam:g} S={S}: auto {r[0][0]:.3f} ms ({r[0][1]} launches); per level {r[1][0]:.3f}; bands {r[2][0]:.3f}; "
              f"subtrees {r[3][0]:.3f}{flag}", flush=True)
    print(f"{bad} cases where the automatic form of the tree passes is more than 1.3x slower than the best explicit one")


def main():
    rs = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
    cases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    if len(sys.argv) > 3 and sys.argv[3] == "passes":
        return passes(rs, cases)
    bad = 0
    for c in range(cases):
        n = int(rs.choice([2, 3, 4, 4, 8, 20, 33, 61]))
        shape = str(rs.choice(["yule", "yule", "ladder"]))
        T = int(rs.choice([60, 300, 1200, 3000]))
        lam = float(rs.choice([0.5, 4.0, 4.0, 30.0, 200.0]))
        S = int(rs.choice([1, 8, 32, 64, 200, 512, 4096, 16384]))
        if 2.0 * T * (1 + lam) * S * (1 + n / 8.0) > 3e8 or (n > 4 and shape == "ladder" and T > 1200):      # keep a case within a second or so
            continue
        Q = synth.dense_Q(n, 0.02, 0.3, seed=n) if n > 4 or n == 3 else synth.config_Q(1 if n == 2 else 2)
        Omega = 1.25 * float(np.max(np.abs(np.diag(Q))))
        pid = np.full(n, 1.0 / n)
        z = tree(shape, T, Q, pid, Omega, lam, rs)
        try:
            ta, ma = run(z, Q, pid, Omega, S, "auto")
            tb, _ = run(z, Q, pid, Omega, S, "branches")
            tt, _ = run(z, Q, pid, Omega, S, "tiles")
        except _lib.PhmError as ex:
            print(f"case {c}: n={n} {shape} tips={T} Omega*t={lam:g} S={S}: {str(ex)[:90]}", flush=True)
            continue
        best = min(tb, tt)
        flag = "  <-- automatic choice off" if ta > 1.5 * best and ta - best > 0.05 else ""
        bad += bool(flag)
        print(f"case {c}: n={n} {shape} tips={T} Omega*t={lam:g} S={S}: auto {ta:.3f} ms ({ma}); branches {tb:.3f}; tiles {tt:.3f}{flag}", flush=True)
    print(f"{bad} cases where the automatic choice is more than 1.5x slower than the better explicit mapping")


if __name__ == "__main__":
    main()
