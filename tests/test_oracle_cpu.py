"""CPU tests of the oracle: known-answer vectors, the independent Python restatement, golden fixtures and the
reference's own validation idea (EXP == MCMC == SPARSE in distribution, vignettes/phylomap_tutorial.Rnw:113-134)."""
import glob
import os

import numpy as np
import pytest
from scipy.linalg import expm

import oracle_lib as O
import pyref
from phylomap_amd import api, synth, treeorder

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _orders(z):
    return treeorder.pruningwiseedgeorder(z), treeorder.makenodelist(z), treeorder.myreorder(z)


# ---- RNG and elementary functions ---------------------------------------------------------------------
def test_philox_random123_known_answers():
    """Random123's kat_vectors for philox4x32 with 10 rounds (its default) and with 7 (the sampler's streams)."""
    assert O.philox([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert O.philox([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert O.philox([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]
    assert O.philox([0, 0, 0, 0], [0, 0], 7) == [0x5f6fb709, 0x0d893f64, 0x4f121f81, 0x4f730a48]
    assert O.philox([0xffffffff] * 4, [0xffffffff] * 2, 7) == [0x5207ddc2, 0x45165e59, 0x4d8ee751, 0x8c52f662]
    assert O.philox([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0], 7) == \
        [0x4dfccaba, 0x190a87f0, 0xc47362ba, 0xb6b5242a]
    ctr = np.array([[1, 2, 3, 4], [0xdeadbeef, 5, 6, 7]], dtype=np.uint32)
    got = synth.philox4x32_10(ctr, (11, 22))                 # the synthetic-input generator keeps 10 rounds
    for i in range(2):
        assert list(got[i]) == O.philox([int(v) for v in ctr[i]], [11, 22])
        for rounds in (7, 10):
            assert list(pyref.philox(tuple(int(v) for v in ctr[i]), (11, 22), rounds)) == O.philox([int(v) for v in ctr[i]], [11, 22], rounds)
    assert pyref.STREAM_ROUNDS == 7


def test_u01_is_open_interval_and_exact():
    L = O.lib()
    assert L.orc_u01(0) == 2.0 ** -33                          # (x + 0.5) 2^-32: never 0, never 1, exact in a double
    assert L.orc_u01(0xffffffff) == 1.0 - 2.0 ** -33
    assert L.orc_u01(0x12345678) == pyref.u01(0x12345678) == (0x12345678 + 0.5) / 2.0 ** 32
    # stream layout: draw d -> word d & 3 of Philox block d >> 2
    o = O.philox([3, 7, 9, 2], [5, 6], 7)
    for w in range(4):
        assert L.orc_stream_u(5, 6, 2, 9, 7, 12 + w) == L.orc_u01(o[w])


def test_log_exp_accuracy_and_python_twin():
    L = O.lib()
    rs = np.random.default_rng(0)
    xs = np.concatenate([np.exp(rs.uniform(-40, 0, 20000)), rs.uniform(0, 1, 20000), [2.0 ** -53, 1 - 2.0 ** -53, 0.5, 1.0]])
    mine = np.array([L.orc_log(float(v)) for v in xs])
    ref = np.log(xs)
    ok = ref != 0
    assert np.max(np.abs(mine[ok] - ref[ok]) / np.spacing(np.abs(ref[ok]))) <= 2.0
    assert L.orc_log(1.0) == 0.0
    for v in xs[:2000]:
        assert L.orc_log(float(v)) == pyref.plog(float(v))
    xe = np.concatenate([rs.uniform(-700, 5, 20000), rs.uniform(-1, 1, 20000), [0.0, -0.0, -745.0, -800.0]])
    mine = np.array([L.orc_exp(float(v)) for v in xe])
    ref = np.exp(xe)
    ok = ref > 1e-300
    assert np.max(np.abs(mine[ok] - ref[ok]) / np.spacing(ref[ok])) <= 2.0
    assert L.orc_exp(0.0) == 1.0 and L.orc_exp(-800.0) == 0.0
    for v in xe[:2000]:
        assert L.orc_exp(float(v)) == pyref.pexp(float(v))


def test_r_stream_primitives():
    """R-stream mode (tools/r_parity): set.seed scrambling + Mersenne-Twister + unif_rand + Ahrens-Dieter exp_rand
    reproduce R's well-known outputs."""
    import ctypes as C
    L = O.lib()
    def draws(seed, nu, ne):
        u, e = np.zeros(max(nu, 1)), np.zeros(max(ne, 1))
        L.orc_rstream_selftest(C.c_uint32(seed), nu, ne, u.ctypes.data_as(C.POINTER(C.c_double)), e.ctypes.data_as(C.POINTER(C.c_double)))
        return u[:nu], e[:ne]
    np.testing.assert_allclose(draws(42, 5, 0)[0], [0.9148060, 0.9370754, 0.2861395, 0.8304476, 0.6417455], atol=5e-8)
    np.testing.assert_allclose(draws(1, 3, 0)[0], [0.2655087, 0.3721239, 0.5728534], atol=5e-8)
    np.testing.assert_allclose(draws(123, 3, 0)[0], [0.2875775, 0.7883051, 0.4089769], atol=5e-8)
    np.testing.assert_allclose(draws(42, 0, 5)[1], [0.1983368, 0.6608953, 0.2834910, 0.0381919, 0.4731766], atol=5e-8)
    np.testing.assert_allclose(draws(1, 0, 3)[1], [0.7551818, 1.1816428, 0.1457067], atol=5e-8)
    # the Ahrens-Dieter table is q[k-1] = sum_{j<=k} ln2^j / j!
    import math
    acc, q = 0.0, []
    for j in range(1, 17):
        acc += math.log(2.0) ** j / math.factorial(j)
        q.append(acc)
    assert abs(q[0] - 0.6931471805599453) < 1e-16 and abs(q[1] - 0.9333736875190459) < 1e-15 and abs(q[15] - 1.0) < 1e-15


def test_r_stream_mode_runs_the_same_sampler():
    """Sequential R stream instead of Philox: a different realisation of the same chain (invariants hold, statistics agree)."""
    z, Q, pid, Omega = synth.config_problem(2, n_tips=25)
    nen, nodelist, root = _orders(z)
    B = np.eye(4) + Q / Omega
    a, rc = O.maketreelistMCMC(z, Q, pid, B, Omega, nen, nodelist, root, 400, variant=O.BIGTREE, seed=7, rstream=True)
    assert rc == 0
    np.testing.assert_allclose(a[:, :4].sum(1), z["edge.length"].sum(), rtol=1e-12)
    b, rc = O.maketreelistMCMC(z, Q, pid, B, Omega, nen, nodelist, root, 400, variant=O.BIGTREE, seed=7)
    ja, jb = a[100:, 4:].sum(1), b[100:, 4:].sum(1)
    assert abs(ja.mean() - jb.mean()) < 6 * 4 * np.hypot(ja.std(), jb.std()) / np.sqrt(300)
    a2, _ = O.maketreelistMCMC(z, Q, pid, B, Omega, nen, nodelist, root, 400, variant=O.BIGTREE, seed=7, rstream=True)
    np.testing.assert_array_equal(a, a2)


def test_r_dpois_restatement_and_exp_r_stream_mode():
    """Rf_dpois (src/phylomap.cpp:107,128) restated as R's saddle-point dpois_raw (stirlerr + bd0): against
    exp(-lam) lam^k / k! in exact rational / 50-digit arithmetic, and sumstatEXP in R-stream mode (set.seed + unif_rand in the
    reference's draw order, dpois_raw) as a second realisation of the same sampler."""
    import ctypes as C
    import math
    import mpmath
    mpmath.mp.dps = 50
    L = O.lib()
    L.orc_r_dpois.restype = C.c_double
    L.orc_r_dpois.argtypes = [C.c_double, C.c_double]
    worst = 0.0
    for lam in (0.01, 0.3, 1.0, 2.5, 4.0, 7.75, 19.0, 33.3, 60.0):
        for k in (0, 1, 2, 3, 5, 8, 13, 15, 16, 21, 34, 36, 55, 81, 150, 300):
            want = mpmath.exp(-mpmath.mpf(lam)) * mpmath.mpf(lam) ** k / mpmath.factorial(k)
            got = L.orc_r_dpois(float(k), lam)
            if want < 1e-300:
                continue
            ulp = abs(mpmath.mpf(got) - want) / (mpmath.mpf(2) ** (mpmath.floor(mpmath.log(want, 2)) - 52))
            # dpois_raw = exp(-stirlerr - bd0) / sqrt(2 pi k): a few ulp near the mode, and the rounding of the exponent's argument
            # (relative error ~ |log p| eps) far out in the tails -- a wrong table entry or coefficient would show as 1e3 .. 1e12 ulp
            # (and bd0's x log(x / lambda) + lambda - x cancels when |x - lambda| > 0.1 (x + lambda): the weakness R 4.1's ebd0 removed)
            worst = max(worst, float(ulp) / (8.0 + 4.0 * (abs(float(mpmath.log(want))) + (k * abs(math.log(k / lam)) if k else 0.0))))
    assert worst <= 1.0, worst
    assert L.orc_r_dpois(0.0, 0.0) == 1.0 and L.orc_r_dpois(3.0, 0.0) == 0.0 and L.orc_r_dpois(0.0, 2.0) == math.exp(-2.0)
    z, Q, pid, Omega = synth.config_problem(1, n_tips=30)
    nen, nodelist, root = _orders(z)
    lefts, rights, d = api.eigen_decompose(Q)
    a, rc = O.maketreelistEXP(z, Q, pid, nen, nodelist, root, 600, lefts, rights, d, seed=11, rstream=True)
    assert rc == 0
    np.testing.assert_allclose(a[:, :2].sum(1), z["edge.length"].sum(), rtol=1e-12)
    b, rc = O.maketreelistEXP(z, Q, pid, nen, nodelist, root, 600, lefts, rights, d, seed=11)
    assert rc == 0
    ja, jb = a[:, 2:].sum(1), b[:, 2:].sum(1)
    assert abs(ja.mean() - jb.mean()) < 5 * np.hypot(ja.std(), jb.std()) / np.sqrt(600)
    a2, _ = O.maketreelistEXP(z, Q, pid, nen, nodelist, root, 600, lefts, rights, d, seed=11, rstream=True)
    np.testing.assert_array_equal(a, a2)


# ---- per-function known answers (hand-derived) ---------------------------------------------------------
def test_shortener_merges_and_counts():
    # states 0,0,2,2,1,0 with n=3: merged 0(1.5) 2(3.0) 1(4.0) 0(8.0); transitions 0->2, 2->1, 1->0
    d, s, row = O.shortener([1.0, 0.5, 1.0, 2.0, 4.0, 8.0], [0, 0, 2, 2, 1, 0], 3)
    np.testing.assert_array_equal(d, [1.5, 3.0, 4.0, 8.0])
    np.testing.assert_array_equal(s, [0, 2, 1, 0])
    want = np.zeros(9)
    want[3 + 0 * 2 + 1] = 1      # 0->2 : n + from(n-1) + (to-1)   (src/phylomap.cpp:65)
    want[3 + 2 * 2 + 1] = 1      # 2->1 : n + from(n-1) + to       (:66)
    want[3 + 1 * 2 + 0] = 1      # 1->0
    np.testing.assert_array_equal(row, want)


def test_mattospmat_threshold():
    B = np.array([[0.5, 1e-7, 2e-7], [-0.1, 0.0, 1.0], [1e-8, 0.3, 0.7]])
    out = np.zeros((3, 3))
    import ctypes as C
    O.lib().orc_matTospmat(B.ctypes.data_as(C.POINTER(C.c_double)), 3, out.ctypes.data_as(C.POINTER(C.c_double)))
    np.testing.assert_array_equal(out, [[0.5, 0.0, 2e-7], [0.0, 0.0, 1.0], [0.0, 0.3, 0.7]])


def _three_tip_tree(states, m=(2, 2, 2, 2)):
    # ((t1,t2)n5,t3)n4 ; cladewise rows: 4->5, 5->1, 5->2, 4->3
    edge = np.array([[4, 5], [5, 1], [5, 2], [4, 3]], dtype=np.int32)
    lens = np.array([1.0, 2.0, 3.0, 4.0])
    maps = [np.full(k, lens[i] / k) for i, k in enumerate(m)]
    mapnames = [np.ones(k, dtype=np.int32) for k in m]
    return {"edge": edge, "Nnode": 2, "edge.length": lens, "states": np.asarray(states, dtype=np.int32),
            "maps": maps, "mapnames": mapnames, "node.states": np.ones((4, 2), dtype=np.int32)}


def test_pruning_three_tips_by_hand():
    z = _three_tip_tree([1, 2, 2], m=(3, 2, 1, 2))
    B = np.array([[0.9, 0.1], [0.2, 0.8]])
    nen, nodelist, root = _orders(z)
    assert root == 4 and list(nodelist) == [5]
    PL, rc = O.makePL(z, 2, B, nen, [3, 2, 1, 2], False)
    assert rc == 0
    e1, e2 = np.array([1.0, 0.0]), np.array([0.0, 1.0])
    pl5 = (B @ e1) * e2                                   # t1 through B^1, t2 through B^0
    pl4 = (B @ (B @ pl5)) * (B @ e2)                      # n5 through B^2, t3 through B^1
    np.testing.assert_allclose(PL[4], pl5, rtol=1e-15)
    np.testing.assert_allclose(PL[3], pl4, rtol=1e-15)
    PLn, _ = O.makePL(z, 2, B, nen, [3, 2, 1, 2], True)
    np.testing.assert_allclose(PLn[4], pl5 / pl5.sum(), rtol=1e-15)
    assert abs(PLn[3].sum() - 1) < 1e-15


def test_sweep_with_scripted_tape_by_hand():
    """One sumstatMCMC sweep on the 3-tip tree with scripted draws, checked against a hand walk-through."""
    z = _three_tip_tree([1, 1, 2], m=(1, 3, 2, 2))
    Q = np.array([[-1.0, 1.0], [1.0, -1.0]])
    Omega = 2.0
    B = np.eye(2) + Q / Omega                             # all entries 0.5
    pid = np.array([0.5, 0.5])
    nen, nodelist, root = _orders(z)
    # uniforms in reference order: root, node 5, then branch-state draws (only branch 2 has m=3 -> one draw)
    tape_u = [0.1, 0.1, 0.9]
    # exponentials: every merged segment draws until it overshoots; make each first draw overshoot
    tape_e = [100.0] * 16
    out, rc = O.maketreelistMCMC(z, Q, pid, B, Omega, nen, nodelist, root, 1, tape_u=tape_u, tape_e=tape_e)
    assert rc == 0
    # root: p = pid*PL[root] with u = .1 -> state 0 (first index whose cumulative share >= .1);
    # node 5 likewise state 0.  branch 1 (5->t1, m=3): middle state ~ B[0,:]*B e_0 = (.25,.25), u=.9 -> state 1,
    # so the path 0,1,0 has two transitions; every other branch has none except 4->t3 (0 -> 1).
    row = out[0]
    assert row[2] == 2.0 and row[3] == 1.0                # 0->1 twice (branch 1 and branch 4->t3), 1->0 once
    np.testing.assert_allclose(row[:2].sum(), 10.0, rtol=1e-15)
    # dwell in state 1: middle third of branch 5->t1 (2/3) + second half of 4->t3 (2.0)
    np.testing.assert_allclose(row[1], 2.0 / 3 + 2.0, rtol=1e-15)


def test_matexp_and_pade_against_scipy():
    for Q in (synth.config_Q(1), synth.config_Q(2), synth.config_Q(5)):
        lefts, rights, d = api.eigen_decompose(Q)
        for t in (0.0, 0.3, 2.5, 40.0):
            P = O.matexp(lefts, rights, np.diag(d), t)
            np.testing.assert_allclose(P, expm(Q * t), atol=1e-11)
            P2, rc = O.expmat_pade(Q * t)
            assert rc == 0
            np.testing.assert_allclose(P2, expm(Q * t), atol=1e-12)
    Q = synth.config_Q(4)
    P2, rc = O.expmat_pade(Q * 1.7)
    np.testing.assert_allclose(P2, expm(Q * 1.7), atol=1e-12)
    P = np.array(pyref.matexp(*[a.tolist() for a in api.eigen_decompose(synth.config_Q(2))[:2]],
                              np.diag(api.eigen_decompose(synth.config_Q(2))[2]).tolist(), 0.7))
    l, r, d = api.eigen_decompose(synth.config_Q(2))
    np.testing.assert_array_equal(P, O.matexp(l, r, np.diag(d), 0.7))


# ---- independent restatement: bit-for-bit -----------------------------------------------------------------
@pytest.mark.parametrize("n,variant,ov", [(2, "plain", O.PLAIN), (4, "plain", O.PLAIN), (4, "bigtree", O.BIGTREE),
                                          (4, "sparse", O.SPARSE), (3, "bigtree", O.BIGTREE),
                                          # n > 4: fused k-ordered chains, interleaved normalisation sums
                                          (5, "plain", O.PLAIN), (7, "bigtree", O.BIGTREE), (6, "sparse", O.SPARSE)])
def test_mcmc_oracle_equals_python_restatement(n, variant, ov):
    Q = {2: synth.config_Q(1), 3: np.array([[-.3, .2, .1], [.05, -.15, .1], [.2, .2, -.4]]), 4: synth.config_Q(2),
         5: synth.dense_Q(5, 0.02, 0.08, seed=5), 6: synth.dense_Q(6, 0.02, 0.08, seed=6), 7: synth.dense_Q(7, 0.02, 0.08, seed=7)}[n]
    Omega = 1.25 * float(np.max(np.abs(np.diag(Q))))
    pid = np.full(n, 1.0 / n)
    z = synth.make_tree(10, Q, Omega, 700 + n)
    nen, nodelist, root = _orders(z)
    N, seed, rep = 6, 99, 3
    got, rc = O.maketreelistMCMC(z, Q, pid, np.eye(n) + Q / Omega, Omega, nen, nodelist, root, N, variant=ov, seed=seed, replica=rep)
    assert rc == 0
    want = pyref.sumstatMCMC(z, Q.tolist(), pid.tolist(), Omega, N, [int(v) for v in nen], [int(v) for v in nodelist],
                             root, seed, rep, variant)
    np.testing.assert_array_equal(got, np.array(want))
    # the O(E) edge search of the reference (src/phylomap.cpp:643) changes cost, never results
    got2, _ = O.maketreelistMCMC(z, Q, pid, np.eye(n) + Q / Omega, Omega, nen, nodelist, root, N, variant=ov, seed=seed,
                                 replica=rep, faithful_search=True)
    np.testing.assert_array_equal(got, got2)


@pytest.mark.parametrize("n", [2, 4, 6])
def test_ks_sweep_oracle_equals_python_restatement(n):
    Q = {2: synth.config_Q(1), 4: synth.make2sQ(.1, .1, .2, .2, 10), 6: synth.make2sQ(.1, .3, [.2, .4], [.5, .6], [2, 3])}[n]
    Omega = 1.25 * float(np.max(np.abs(np.diag(Q))))
    pid = np.full(n, 1.0 / n)
    z = synth.make_tree(9, Q, Omega, 650 + n)
    z["states"] = ((z["states"] - 1) % 2 + 1).astype(np.int32)
    nen, nodelist, root = _orders(z)
    N, seed, rep = 5, 12, 2
    got, rc = O.maketreelistMCMC(z, Q, pid, np.eye(n) + Q / Omega, Omega, nen, nodelist, root, N, variant=O.KS, seed=seed, replica=rep)
    assert rc == 0
    want = np.array(pyref.sumstatMCMC(z, Q.tolist(), pid.tolist(), Omega, N, [int(v) for v in nen], [int(v) for v in nodelist],
                                      root, seed, rep, "ks"))
    np.testing.assert_array_equal(got, want)
    k = n // 2 - 1
    np.testing.assert_allclose(got[:, :n].sum(1), z["edge.length"].sum(), rtol=1e-12)
    assert got.shape[1] == n + n * n + 2 + 3 * k + 1                       # man/sumstatMCMCks.Rd:19
    assert np.all(got[:, n + n * n] == Q[0, 1]) and np.all(got[:, n + n * n + 1] == Q[1, 0])
    assert np.all((got[:, -1] >= 0) & (got[:, -1] < n))
    with pytest.raises(AssertionError):
        Qodd = np.array([[-.3, .2, .1], [.05, -.15, .1], [.2, .2, -.4]])
        _, rc = O.maketreelistMCMC(z, Qodd, np.full(3, 1 / 3), np.eye(3) + Qodd / .5, .5, nen, nodelist, root, 2, variant=O.KS)
        assert rc == 0


@pytest.mark.parametrize("n", [2, 3, 5, 7])
def test_bf_sweep_oracle_equals_python_restatement_for_any_state_count(n):
    """treesamplebf (src/phylomap.cpp:1169-1179) is n-generic: observed tips, n x n counts incl. self pairs (shortenerbf
    :1010-1014), row-normalised pruning (:1085); only the driver's 9 columns and the rate updates are two-state."""
    Q = {2: synth.config_Q(1), 3: np.array([[-.3, .2, .1], [.05, -.15, .1], [.2, .2, -.4]]),
         5: synth.dense_Q(5, 0.02, 0.08, seed=5), 7: synth.dense_Q(7, 0.02, 0.08, seed=7)}[n]
    Omega = 1.25 * float(np.max(np.abs(np.diag(Q))))
    pid = np.full(n, 1.0 / n)
    z = synth.make_tree(9, Q, Omega, 660 + n)
    nen, nodelist, root = _orders(z)
    N, seed, rep = 5, 13, 1
    got, rc = O.maketreelistMCMC(z, Q, pid, np.eye(n) + Q / Omega, Omega, nen, nodelist, root, N, variant=O.BF, seed=seed, replica=rep)
    assert rc == 0 and got.shape == (N, n + n * n + 3)
    want = np.array(pyref.sumstatMCMC(z, Q.tolist(), pid.tolist(), Omega, N, [int(v) for v in nen], [int(v) for v in nodelist],
                                      root, seed, rep, "bf"))
    np.testing.assert_array_equal(got, want)
    np.testing.assert_allclose(got[:, :n].sum(1), z["edge.length"].sum(), rtol=1e-12)
    cnt = got[:, n:n + n * n].reshape(N, n, n)
    # every consecutive pair of segments is counted once: segments - branches, summed over the tree
    big, rc = O.maketreelistMCMC(z, Q, pid, np.eye(n) + Q / Omega, Omega, nen, nodelist, root, N, variant=O.BIGTREE, seed=seed, replica=rep)
    assert rc == 0
    # the bf sweep draws what the _bigtree sweep draws (same pruning, same streams): its off-diagonal counts are _bigtree's
    off = np.array([[cnt[i][a][c] for a in range(n) for c in range(n) if a != c] for i in range(N)])
    np.testing.assert_array_equal(off, big[:, n:])
    np.testing.assert_array_equal(got[:, :n], big[:, :n])
    assert np.all(got[:, n + n * n] == Q[0, 1]) and np.all(got[:, n + n * n + 1] == Q[1, 0])
    assert np.all((got[:, -1] >= 0) & (got[:, -1] < n))
    # the rate-updating driver stays two-state
    if n != 2:
        _, rc = O.maketreelistMCMC(z, Q, pid, np.eye(n) + Q / Omega, Omega, nen, nodelist, root, 2, variant=O.BF, prior=[1, 1, 1, 1])
        assert rc & O.ERR_BAD_INPUT


def test_dic_loglikelihood_against_scipy():
    """The log p(y|Q) column of the DIC drivers (src/phylomap.cpp:3239-3251): Pade expm + scaled pruning vs scipy."""
    from scipy.linalg import expm
    Q = np.array([[-.1, .1], [.1, -.1]])
    Omega, pid, prior = 10.0, np.array([.3, .7]), [.55, 1, .56, 1.01]
    z = synth.make_tree(40, Q, 2.0, 61, pid)
    nen, nodelist, root = _orders(z)
    out, rc = O.maketreelistMCMC(z, Q, pid, np.eye(2) + Q / Omega, Omega, nen, nodelist, root, 6, variant=O.BF, seed=3,
                                 prior=prior, dic=True)
    assert rc == 0 and out.shape == (6, 10)
    E, T = z["edge"], 40
    for it in range(6):
        Qi = np.array([[-out[it, 6], out[it, 6]], [out[it, 7], -out[it, 7]]])      # the Q recorded for this sweep
        PL = np.zeros((2 * T - 1, 2))
        PL[np.arange(T), z["states"] - 1] = 1
        for i in range(T - 1):
            ea, eb = nen[2 * i] - 1, nen[2 * i + 1] - 1
            PL[E[ea, 0] - 1] = (expm(Qi * z["edge.length"][ea]) @ PL[E[ea, 1] - 1]) * (expm(Qi * z["edge.length"][eb]) @ PL[E[eb, 1] - 1])
        np.testing.assert_allclose(out[it, 9], np.log(PL[root - 1] @ pid), rtol=1e-11)


@pytest.mark.parametrize("n", [2, 4])
def test_exp_oracle_equals_python_restatement(n):
    Q = {2: synth.config_Q(1), 4: synth.config_Q(2)}[n]
    Omega = 1.25 * float(np.max(np.abs(np.diag(Q))))
    pid = np.full(n, 1.0 / n)
    z = synth.make_tree(9, Q, Omega, 800 + n)
    nen, nodelist, root = _orders(z)
    lefts, rights, d = api.eigen_decompose(Q)
    N, seed = 12, 5
    got, rc = O.maketreelistEXP(z, Q, pid, nen, nodelist, root, N, lefts, rights, d, seed=seed, replica=1)
    assert rc == 0
    want = pyref.sumstatEXP(z, Q.tolist(), pid.tolist(), N, [int(v) for v in nen], [int(v) for v in nodelist], root,
                            lefts.tolist(), rights.tolist(), np.diag(d).tolist(), seed, 1)
    np.testing.assert_array_equal(got, np.array(want))
    got2, _ = O.maketreelistEXP(z, Q, pid, nen, nodelist, root, N, lefts, rights, d, seed=seed, replica=1, recompute=True,
                                faithful_search=True)
    np.testing.assert_array_equal(got, got2)      # re-exponentiating every iteration (:2980) changes nothing
    # rescaled pruning (not in the reference: row / sum after every node): oracle == Python twin bit for bit, and on a tree this
    # small the rescaled and the plain sampler pick the same states -- identical counts, identical dwell times
    got3, rc = O.maketreelistEXP(z, Q, pid, nen, nodelist, root, N, lefts, rights, d, seed=seed, replica=1, rescale=True)
    assert rc == 0
    want3 = pyref.sumstatEXP(z, Q.tolist(), pid.tolist(), N, [int(v) for v in nen], [int(v) for v in nodelist], root,
                             lefts.tolist(), rights.tolist(), np.diag(d).tolist(), seed, 1, rescale=True)
    np.testing.assert_array_equal(got3, np.array(want3))
    np.testing.assert_array_equal(got3, got)


def test_forced_normalisation_of_the_plain_and_sparse_drivers():
    """ORC_FORCE_NORMALISE: the plain driver with its rows rescaled IS the _bigtree driver; the SPARSE driver keeps its thresholded
    chain matrix; on a small tree (no underflow) the rescaled drivers draw the same histories as the un-rescaled ones."""
    Q = synth.config_Q(2)
    Omega = 1.25 * float(np.max(np.abs(np.diag(Q))))
    pid = np.full(4, 0.25)
    z = synth.make_tree(30, Q, Omega, 4321, pid)
    nen, nodelist, root = _orders(z)
    B = np.eye(4) + Q / Omega
    run = lambda v: O.maketreelistMCMC(z, Q, pid, B, Omega, nen, nodelist, root, 12, variant=v, seed=6)
    big, rc = run(O.BIGTREE); assert rc == 0
    forced, rc = run(O.PLAIN | O.FORCE_NORMALISE); assert rc == 0
    np.testing.assert_array_equal(big, forced)
    plain, rc = run(O.PLAIN); assert rc == 0
    np.testing.assert_array_equal(plain[:, 4:], forced[:, 4:])
    sp, rc = run(O.SPARSE); assert rc == 0
    spf, rc = run(O.SPARSE | O.FORCE_NORMALISE); assert rc == 0
    np.testing.assert_array_equal(sp[:, 4:], spf[:, 4:])
    z5, Q5, pid5, Om5 = synth.config_problem(5)
    n5 = _orders(z5)
    _, rc = O.maketreelistMCMC(z5, Q5, pid5, np.eye(20) + Q5 / Om5, Om5, *n5, 2, variant=O.SPARSE, seed=1)
    assert rc & O.ERR_ZERO_PROB                        # 5 000 tips: the SPARSE driver underflows ...
    out, rc = O.maketreelistMCMC(z5, Q5, pid5, np.eye(20) + Q5 / Om5, Om5, *n5, 2, variant=O.SPARSE | O.FORCE_NORMALISE, seed=1)
    assert rc == 0                                     # ... rescaled it runs
    np.testing.assert_allclose(out[:, :20].sum(1), z5["edge.length"].sum(), rtol=1e-12)


def test_exp_rescaled_pruning_survives_a_thousand_tips():
    """makePLexp has no rescaling (src/phylomap.cpp:2899-2906): at 1 000 tips the plain sampler's root vector underflows
    (RcppArmadillo::sample would throw); with the rows rescaled the same sampler runs and conserves the tree length."""
    z, Q, pid, Omega = synth.config_problem(2)
    nen, nodelist, root = _orders(z)
    lefts, rights, d = api.eigen_decompose(Q)
    _, rc = O.maketreelistEXP(z, Q, pid, nen, nodelist, root, 2, lefts, rights, d, seed=3)
    assert rc & O.ERR_ZERO_PROB
    out, rc = O.maketreelistEXP(z, Q, pid, nen, nodelist, root, 4, lefts, rights, d, seed=3, rescale=True)
    assert rc == 0
    np.testing.assert_allclose(out[:, :4].sum(1), z["edge.length"].sum(), rtol=1e-12)


# ---- golden fixtures --------------------------------------------------------------------------------------
@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "*.npz"))))
def test_oracle_reproduces_golden(path):
    from golden.make_golden import unpack_tree
    g = np.load(path)
    z = unpack_tree(g)
    Q, pid, Omega, seed = g["Q"], g["pid"], float(g["Omega"]), int(g["seed"])
    n = Q.shape[0]
    B = np.eye(n) + Q / Omega
    np.testing.assert_array_equal(treeorder.pruningwiseedgeorder(z), g["nen"])
    for key, var in (("mcmc", O.PLAIN), ("bigtree", O.BIGTREE), ("sparse", O.SPARSE)):
        for r in (0, 5):
            out, rc = O.maketreelistMCMC(z, Q, pid, B, Omega, g["nen"], g["nodelist"], int(g["root"]), 24, variant=var,
                                         seed=seed, replica=r)
            assert rc == 0
            np.testing.assert_array_equal(out, g[f"{key}_r{r}"])
    out, rc = O.maketreelistEXP(z, Q, pid, g["nen"], g["nodelist"], int(g["root"]), 48, g["lefts"], g["rights"], g["d"], seed=seed)
    assert rc == 0
    np.testing.assert_array_equal(out, g["exp_r0"])


# ---- invariants and the reference's statistical cross-check ------------------------------------------------
def test_invariants():
    z, Q, pid, Omega = synth.config_problem(2, n_tips=60)
    nen, nodelist, root = _orders(z)
    out, rc, dump = O.maketreelistMCMC(z, Q, pid, np.eye(4) + Q / Omega, Omega, nen, nodelist, root, 30, variant=O.BIGTREE,
                                       seed=4, dump=True)
    assert rc == 0
    np.testing.assert_allclose(out[:, :4].sum(1), z["edge.length"].sum(), rtol=1e-12)
    assert np.all(out[:, 4:] == np.round(out[:, 4:])) and np.all(out >= 0)
    np.testing.assert_array_equal(dump.node_states[:60], z["states"])                      # tips never change (:612)
    for b in range(len(dump.seg_count)):                                                   # endpoints = node states (:468-472)
        m = dump.seg_count[b]
        assert dump.seg_state[b, 0] + 1 == dump.node_states[z["edge"][b, 0] - 1] or m == 1
        assert dump.seg_state[b, m - 1] + 1 == dump.node_states[z["edge"][b, 1] - 1]
        np.testing.assert_allclose(dump.seg_dwell[b, :m].sum(), z["edge.length"][b], rtol=1e-12)
    # positive rescaling of PL rows does not change the sampler (:525 vs :510): plain == bigtree on a small tree
    a, _ = O.maketreelistMCMC(z, Q, pid, np.eye(4) + Q / Omega, Omega, nen, nodelist, root, 5, variant=O.PLAIN, seed=4)
    np.testing.assert_array_equal(a[:, 4:], out[:5, 4:])


def test_exp_mcmc_sparse_agree_in_distribution():
    """vignettes/phylomap_tutorial.Rnw:113-134: total jump counts from EXP, MCMC and SPARSE must agree."""
    Q = synth.tridiagonal_Q(6, 0.05)
    Omega = 0.2
    pid = np.full(6, 1 / 6)
    z = synth.make_tree(14, Q, 1.0, 31, pid, init_segments=6)
    nen, nodelist, root = _orders(z)
    lefts, rights, d = api.eigen_decompose(Q)
    B = np.eye(6) + Q / Omega
    ex, rc = O.maketreelistEXP(z, Q, pid, nen, nodelist, root, 6000, lefts, rights, d, seed=1)
    assert rc == 0
    chains = {}
    for name, var in (("mcmc", O.PLAIN), ("sparse", O.SPARSE)):
        out, rc = O.maketreelistMCMC(z, Q, pid, B, Omega, nen, nodelist, root, 12000, variant=var, seed=2)
        assert rc == 0
        chains[name] = out[2000:]
    je = ex[:, 6:].sum(1)
    se = je.std() / np.sqrt(je.size)
    for name, out in chains.items():
        jm = out[:, 6:].sum(1)
        # chain is autocorrelated: allow a generous 6 sigma of the (inflated) standard error
        sm = 4 * jm.std() / np.sqrt(jm.size)
        assert abs(jm.mean() - je.mean()) < 6 * np.hypot(se, sm), (name, jm.mean(), je.mean())
        np.testing.assert_allclose(out[:, :6].mean(0), ex[:, :6].mean(0), rtol=0.1, atol=0.5)


def test_multitree_oracle_structure():
    """orc_maketreelistMCMCmt (maketreelistMCMCmt src/phylomap.cpp:2267-2365): with one tree in the list the first sweep is
    the plain single-tree sweep (un-normalised pruning, replica word 0) with self pairs counted as well; with several
    trees the kept row belongs to the tree named in the last column, and the recorded rates are the ones the previous
    iteration's update left behind (recordQmtNS :2169-2173 runs before the sweep)."""
    Q = synth.config_Q(1)
    Omega, pid, prior = 0.5, [.5, .5], [.55, 1, .56, 1.01]
    trees = synth.make_treelist(5, 12, Q, Omega, 77)
    orders = [_orders(z) for z in trees]
    nen_m, nl_m, roots = np.stack([o[0] for o in orders]), np.stack([o[1] for o in orders]), [o[2] for o in orders]
    B = np.eye(2) + Q / Omega
    one, rc = O.maketreelistMCMCmt(trees[:1], Q, pid, B, Omega, nen_m[:1], nl_m[:1], roots[:1], 1, prior, seed=5)
    plain, rc2 = O.maketreelistMCMC(trees[0], Q, pid, B, Omega, nen_m[0], nl_m[0], roots[0], 1, variant=O.PLAIN, seed=5)
    assert rc == 0 and rc2 == 0
    np.testing.assert_array_equal(one[0, [0, 1, 3, 4]], plain[0])
    assert one[0, 6] == Q[0, 1] and one[0, 7] == Q[1, 0] and one[0, 8] == 0

    N = 60
    out, rc = O.maketreelistMCMCmt(trees, Q, pid, B, Omega, nen_m, nl_m, roots, N, prior, seed=5)
    assert rc == 0 and out.shape == (N, 9)
    picks = out[:, 8].astype(int)
    assert set(picks) == set(range(5))                                    # sampleOnce over unit weights reaches every tree
    lengths = np.array([z["edge.length"].sum() for z in trees])
    np.testing.assert_allclose(out[:, 0] + out[:, 1], lengths[picks], rtol=1e-12)
    # the rates recorded at iteration i+1 are what phm_qupdate_apply-style updates make of row i
    import ctypes as C
    for i in range(N - 1):
        Qi = np.array([[-out[i, 6], out[i, 6]], [out[i, 7], -out[i, 7]]])
        rc = O.lib().orc_qupdate_apply(O.MT, 2, Qi.ctypes.data_as(C.POINTER(C.c_double)), C.c_double(Omega),
                                       np.asarray(prior, dtype=float).ctypes.data_as(C.POINTER(C.c_double)),
                                       np.ascontiguousarray(out[i, :6]).ctypes.data_as(C.POINTER(C.c_double)),
                                       C.c_uint32(5), C.c_uint32(0), C.c_uint32(i))
        assert rc == 0
        assert Qi[0, 1] == out[i + 1, 6] and Qi[1, 0] == out[i + 1, 7]
    assert 3 < len(np.unique(out[:, 6])) < N                              # some proposals accepted, some rejected (:2207)

    # hidden rates over a list (maketreelistMCMCksmt :2722-2844): shapes, tree column, conserved tree length
    Qk = synth.make2sQ(.1, .1, .2, .2, 10)
    tk = synth.make_treelist(3, 10, Qk, 8.0, 9)
    for z in tk:
        z["states"] = ((z["states"] - 1) % 2 + 1).astype(np.int32)
    ok = [_orders(z) for z in tk]
    outk, rc = O.maketreelistMCMCmt(tk, Qk, np.full(4, .25), np.eye(4) + Qk / 25.0, 25.0, np.stack([o[0] for o in ok]),
                                    np.stack([o[1] for o in ok]), [o[2] for o in ok], 15, [1, 10, 1.5, 11, 2, 10, 20, 2],
                                    variant=O.KSMT, seed=3)
    assert rc == 0 and outk.shape == (15, 4 + 16 + 2 + 3 + 1)
    lk = np.array([z["edge.length"].sum() for z in tk])
    np.testing.assert_allclose(outk[:, :4].sum(1), lk[outk[:, -1].astype(int)], rtol=1e-12)
    # n must be 2k+2 (:2729)
    _, rc = O.maketreelistMCMCmt(tk, Qk[:3, :3], [1 / 3] * 3, np.eye(3), 25.0, np.stack([o[0] for o in ok]),
                                 np.stack([o[1] for o in ok]), [o[2] for o in ok], 2, [1] * 8, variant=O.KSMT)
    assert rc & O.ERR_BAD_INPUT


SQUAMATE_RDS = "/root/reference/inst/extdata/Squamate/phylomap_compatible_squamate_tree.RData"


@pytest.mark.skipif(not os.path.exists(SQUAMATE_RDS), reason="the reference's data file is only present in the build container")
def test_known_answer_squamate_seed_101_tip_simulation():
    """The reference's own known answer for its data pipeline: R/simulate_2_state_tree.R:11 notes "n01 is 21" for
    set.seed(101) on the shipped 3 951-tip squamate tree.  Reproducing it takes the RDS reader, the R random stream of the
    oracle (set.seed scrambling, Mersenne-Twister, unif_rand, Ahrens-Dieter exp_rand) and ape's postorder -- all restated
    here without R.  The committed fixture (tree + these tips) is what tools/squamate_dic/run_dic.py feeds the DIC drivers."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "squamate_dic"))
    import simulate_tips as st
    from phylomap_amd import rds
    z = rds.read_phylomap_tree(SQUAMATE_RDS)
    assert z["edge"].shape == (7900, 2) and z["Nnode"] == 3950 and all(len(m) == 100 for m in z["maps"][:50])
    Q2 = np.array([[-0.001, 0.001], [0.006, -0.006]])
    tips, n01, n10, t0, t1, used = st.sample2statehistory(z, Q2, [.5, .5], 101, root_tie_first=2)
    assert n01 == 21
    np.testing.assert_allclose(t0 + t1, z["edge.length"].sum(), rtol=1e-12)
    fx = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "squamate", "seed101_tips.npz"))
    np.testing.assert_array_equal(fx["states"], tips)
    np.testing.assert_array_equal(fx["edge"], z["edge"])
    np.testing.assert_array_equal(fx["edge_length"], z["edge.length"])


def test_rds_reader_round_trip_of_hand_built_stream(tmp_path):
    """phylomap_amd/rds.py on a stream assembled by hand from the published format (list with names, integer matrix with
    dim, named real vector, character vector)."""
    import gzip
    import struct
    from phylomap_amd import rds

    def i32(v): return struct.pack(">i", v)
    def chars(s_): return i32(9 | (1 << 18)) + i32(len(s_)) + s_.encode()            # CHARSXP, UTF-8 flag in gp
    def sym(s_): return i32(1) + chars(s_)
    def strvec(v): return i32(16) + i32(len(v)) + b"".join(chars(x) for x in v)
    def attrs(pairs):                                                                 # pairlist with tags
        out = b""
        for k, v in pairs:
            out += i32(2 | 0x400) + sym(k) + v
        return out + i32(254)
    edge = i32(13 | 0x200) + i32(4) + b"".join(i32(v) for v in (3, 3, 1, 2)) + attrs([("dim", i32(13) + i32(2) + i32(2) + i32(2))])
    maps1 = i32(14 | 0x200) + i32(2) + struct.pack(">dd", 0.25, 0.5) + attrs([("names", strvec(["1", "2"]))])
    lst = i32(19 | 0x200) + i32(3) + edge + (i32(19) + i32(1) + maps1) + strvec(["a", "b"]) + \
        attrs([("names", strvec(["edge", "maps", "tip.label"]))])
    path = tmp_path / "x.rds"
    path.write_bytes(gzip.compress(b"X\n" + i32(2) + i32(0x030102) + i32(0x020300) + lst))
    x = rds.read_rds(str(path))
    np.testing.assert_array_equal(x["edge"], [[3, 1], [3, 2]])
    np.testing.assert_array_equal(x["maps"][0], [0.25, 0.5])
    assert x["maps"][0].names == ["1", "2"] and x["tip.label"] == ["a", "b"]
    # version 3 with ALTREP items: a compact integer sequence 1:5, as R >= 3.5 writes `1:n`, and a deferred
    # as.character(c(7, 8)); item = flags 238, info pairlist (class symbol, package symbol, type), state, attributes
    def altrep(cls, state, type_code):
        info = i32(2) + sym(cls) + i32(2) + sym("base") + i32(2) + (i32(13) + i32(1) + i32(type_code)) + i32(254)
        return i32(238) + info + state + i32(254)
    seq = altrep("compact_intseq", i32(14) + i32(3) + struct.pack(">ddd", 5.0, 1.0, 1.0), 13)
    dstr = altrep("deferred_string", i32(2) + (i32(14) + i32(2) + struct.pack(">dd", 7.0, 8.0)) + i32(2) + (i32(13) + i32(1) + i32(0)) + i32(254), 16)
    lst3 = i32(19 | 0x200) + i32(2) + seq + dstr + attrs([("names", strvec(["idx", "lab"]))])
    enc = b"UTF-8"
    path3 = tmp_path / "y.rds"
    path3.write_bytes(b"X\n" + i32(3) + i32(0x040300) + i32(0x030500) + i32(len(enc)) + enc + lst3)
    y = rds.read_rds(str(path3))
    np.testing.assert_array_equal(y["idx"], [1, 2, 3, 4, 5])
    assert y["idx"].dtype == np.int32 and y["lab"] == ["7", "8"]


def test_exponential_variate_table_log_accuracy_and_agreement():
    """orc_neglog_u32 / pyref.neglog_u32 / phm_device.h neglog_u32: -log((k + 0.5) 2^-32) through the 128-entry table of
    tools/gen_log_table.py.  The C and the Python evaluation agree bit for bit; against 50-digit arithmetic the result is
    within 1.5 ulp everywhere probed, including both ends (U near 0: 23.2; U within 2^-8 of 1: the direct series)."""
    import ctypes as C
    import decimal
    import random
    L = O.lib()
    L.orc_neglog_u32.restype = C.c_double
    L.orc_neglog_u32.argtypes = [C.c_uint32]
    decimal.getcontext().prec = 50
    random.seed(7)
    ks = [0, 1, 2, 2 ** 31 - 1, 2 ** 31, 2 ** 31 + 1, 2 ** 32 - 2, 2 ** 32 - 1, 4278190079, 4278190080, 4278190081]
    ks += [random.getrandbits(32) for _ in range(4000)] + [2 ** 32 - 1 - 4099 * i for i in range(1500)] + [8191 * i for i in range(500)]
    worst = 0.0
    for k in ks:
        a = L.orc_neglog_u32(k)
        assert a == pyref.neglog_u32(k)
        exact = -((decimal.Decimal(k) + decimal.Decimal("0.5")) / decimal.Decimal(2 ** 32)).ln()
        worst = max(worst, float(abs((decimal.Decimal(a) - exact) / exact)))
    assert worst < 1.5 * 2.0 ** -52
    assert L.orc_neglog_u32(0) == pyref.neglog_u32(0) and abs(L.orc_neglog_u32(0) - 33 * np.log(2.0)) < 1e-14
    assert 0 < L.orc_neglog_u32(2 ** 32 - 1) < 1.2e-10


def test_exponential_variates_are_exponential():
    """Kolmogorov-Smirnov check of -log((k + 0.5) 2^-32) over a Philox stream against Exp(1), and of the uniforms against U(0,1)."""
    import ctypes as C
    from scipy import stats
    L = O.lib()
    L.orc_neglog_u32.restype = C.c_double
    L.orc_neglog_u32.argtypes = [C.c_uint32]
    L.orc_stream_word.restype = C.c_uint32
    L.orc_stream_word.argtypes = [C.c_uint32] * 6
    words = [L.orc_stream_word(11, 22, 3, 4, 2 << 30 | 5, d) for d in range(40000)]
    e = np.array([L.orc_neglog_u32(k) for k in words])
    u = np.array([L.orc_stream_u(11, 22, 3, 4, 2 << 30 | 5, d) for d in range(40000)])
    assert stats.kstest(e, "expon").pvalue > 1e-3
    assert stats.kstest(u, "uniform").pvalue > 1e-3
    np.testing.assert_allclose(e, -np.log(u), rtol=1e-13)                  # the variate IS -log of the same draw's uniform
    assert abs(e.mean() - 1.0) < 0.02 and abs(e.var() - 1.0) < 0.05


def test_r_parity_procedure_file_plumbing(tmp_path):
    """tools/r_parity/: export_case.py -> (Rscript run_reference.R, needs R) -> compare.py.  Without R the files the R script
    would write are produced by the oracle's own R-stream mode, which checks every file format and the comparison itself
    (all four drivers incl. sumstatEXP must then report EXACT); it pins nothing about R."""
    import subprocess
    import sys
    root_dir = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = str(tmp_path / "case")
    subprocess.run([sys.executable, os.path.join(root_dir, "tools", "r_parity", "export_case.py"), d], check=True, capture_output=True)
    edge = np.loadtxt(os.path.join(d, "edge.csv"), delimiter=",", dtype=np.int32)
    states = np.loadtxt(os.path.join(d, "states.csv"), dtype=np.int32)
    Q = np.loadtxt(os.path.join(d, "Q.csv"), delimiter=",")
    pid = np.loadtxt(os.path.join(d, "pid.csv"))
    maps, names = [], []
    for line in open(os.path.join(d, "maps.csv")):
        a, b = line.strip().split(";")
        maps.append(np.array([float(v) for v in a.split()]))
        names.append(np.array([int(v) for v in b.split()], dtype=np.int32))
    z = {"edge": edge, "Nnode": states.size - 1, "edge.length": np.loadtxt(os.path.join(d, "edge_length.csv")), "states": states,
         "maps": maps, "mapnames": names}
    nen, nodelist, root = _orders(z)
    np.savetxt(os.path.join(d, "nen.csv"), nen, fmt="%d")
    np.savetxt(os.path.join(d, "nodelist.csv"), nodelist, fmt="%d")
    np.savetxt(os.path.join(d, "root.csv"), [root], fmt="%d")
    par = np.genfromtxt(os.path.join(d, "params.csv"), delimiter=",", names=True)
    Omega, N, seed, n = float(par["Omega"]), int(par["N"]), int(par["seed"]), Q.shape[0]
    for name, var in (("sumstatMCMC", O.PLAIN), ("sumstatMCMC_bigtree", O.BIGTREE), ("SPARSEsumstatMCMC", O.SPARSE)):
        got, rc = O.maketreelistMCMC(z, Q, pid, np.eye(n) + Q / Omega, Omega, nen, nodelist, root, N, variant=var, seed=seed, rstream=True)
        assert rc == 0
        np.savetxt(os.path.join(d, name + ".csv"), got, delimiter=",", fmt="%.17g")
    lefts, rights, dd = api.eigen_decompose(Q)
    for nm, a in (("lefts", lefts), ("rights", rights), ("dd", dd)):
        np.savetxt(os.path.join(d, nm + ".csv"), a, delimiter=",", fmt="%.17g")
    got, rc = O.maketreelistEXP(z, Q, pid, nen, nodelist, root, N, lefts, rights, dd, seed=seed, rstream=True, recompute=True)
    assert rc == 0
    np.savetxt(os.path.join(d, "sumstatEXP.csv"), got, delimiter=",", fmt="%.17g")
    r = subprocess.run([sys.executable, os.path.join(root_dir, "tools", "r_parity", "compare.py"), d], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("EXACT") == 4 and "sumstatEXP" in r.stdout
    rs = open(os.path.join(root_dir, "tools", "r_parity", "run_reference.R")).read()
    assert "sumstatEXP(z, Q, pid, par$N)" in rs and "eigen(Q)" in rs


def test_r_stream_normal_and_gamma_variates():
    """R-stream mode's Rf_rgamma (Ahrens-Dieter GD / GS with norm_rand by INVERSION, restated from R's published nmath sources; what the
    rate updates of bf / ks / mt / DIC draw from, src/phylomap.cpp:1202,1235,1463,...).  No R here, so: the quantile function of
    norm_rand (Wichura's AS 241) against scipy to 1e-15 -- any wrong digit in its 46 coefficients would show at 1e-8 or worse --, the
    variates against their distributions (Kolmogorov-Smirnov) for shapes on both sides of every branch of the algorithm, and
    the stream bookkeeping: rnorm(1) consumes exactly two unif_rand()."""
    import ctypes as C
    from scipy import stats
    L = O.lib()
    L.orc_r_qnorm.restype = C.c_double
    L.orc_r_qnorm.argtypes = [C.c_double]
    ps = np.concatenate([np.linspace(1e-6, 1 - 1e-6, 4001), 10.0 ** -np.arange(7, 300, 7.0), [0.075, 0.925, 0.5, 1 - 1e-12]])
    got = np.array([L.orc_r_qnorm(float(p)) for p in ps])
    want = stats.norm.ppf(ps)
    np.testing.assert_allclose(got, want, rtol=2e-15, atol=2e-15)

    def draws(seed, nn, ng, shape=1.0, scale=1.0):
        a, g = np.zeros(max(nn, 1)), np.zeros(max(ng, 1))
        L.orc_rstream_gamma_selftest(C.c_uint32(seed), nn, ng, C.c_double(shape), C.c_double(scale),
                                     a.ctypes.data_as(C.POINTER(C.c_double)), g.ctypes.data_as(C.POINTER(C.c_double)))
        return a[:nn], g[:ng]
    # rnorm(1) = qnorm((floor(2^27 u1) + u2) / 2^27) of the first two uniforms of the stream
    u = np.zeros(2); e = np.zeros(1)
    L.orc_rstream_selftest(C.c_uint32(42), 2, 0, u.ctypes.data_as(C.POINTER(C.c_double)), e.ctypes.data_as(C.POINTER(C.c_double)))
    z1 = draws(42, 1, 0)[0][0]
    assert z1 == L.orc_r_qnorm((np.floor(134217728.0 * u[0]) + u[1]) / 134217728.0)
    zs = draws(7, 20000, 0)[0]
    assert stats.kstest(zs, "norm").pvalue > 1e-3 and abs(zs.mean()) < 0.03 and abs(zs.std() - 1) < 0.02
    for shape, scale in ((0.3, 2.0), (0.95, 1.0), (1.0, 0.5), (2.5, 1.0), (3.686, 1.0), (7.0, 0.1), (13.022, 1.0), (40.0, 0.25), (300.5, 1 / 77.0)):
        g = draws(11, 0, 20000, shape, scale)[1]
        assert np.all(g > 0)
        assert stats.kstest(g, "gamma", args=(shape, 0, scale)).pvalue > 1e-3, (shape, scale)


def test_r_stream_mode_drives_the_rate_updating_drivers():
    """With Rf_rgamma restated, R-stream mode runs sumstatMCMCbf / sumstatMCMCks end to end on R's own stream (tools/r_parity compares
    them with the package where R exists): deterministic in the seed, dwell rows sum to the tree length, rates stay below Omega,
    and the posterior means of the rates agree with the Philox-mode runs of the same driver (same sampler, another stream)."""
    Q2 = np.array([[-0.1, 0.1], [0.1, -0.1]])
    Omega = 10.0
    pid = np.array([0.5, 0.5])
    z = synth.make_tree(40, Q2, 1.0, 0x77, pid)
    nen, nodelist, root = treeorder.pruningwiseedgeorder(z), treeorder.makenodelist(z), treeorder.myreorder(z)
    prior = [0.55, 1, 0.56, 1.01]
    N = 1500
    means = {True: [], False: []}
    first = None
    for rstream in (True, False):
        for seed in range(1, 7):
            got, rc = O.maketreelistMCMC(z, Q2, pid, np.eye(2) + Q2 / Omega, Omega, nen, nodelist, root, N, variant=O.BF, seed=seed, prior=prior, rstream=rstream)
            assert rc == 0
            np.testing.assert_allclose(got[:, :2].sum(1), z["edge.length"].sum(), rtol=1e-12)
            assert np.all(got[:, 6:8] > 0) and np.all(got[:, 6:8] < Omega)
            means[rstream].append(got[300:, 6:8].mean(0))
            if rstream and seed == 1:
                first = got
    again, rc = O.maketreelistMCMC(z, Q2, pid, np.eye(2) + Q2 / Omega, Omega, nen, nodelist, root, N, variant=O.BF, seed=1, prior=prior, rstream=True)
    np.testing.assert_array_equal(again, first)                     # set.seed(1) reproduces the run
    # the chain mixes slowly on 40 tips (run means range over 0.2 .. 0.9 in either mode): the two modes agree within that spread
    mr, mp = np.array(means[True]), np.array(means[False])
    se = np.sqrt(mr.var(0, ddof=1) / 6 + mp.var(0, ddof=1) / 6)
    assert np.all(np.abs(mr.mean(0) - mp.mean(0)) < 3.0 * se), (mr.mean(0), mp.mean(0), se)
    Q4 = synth.make2sQ(.1, .1, .2, .2, 10)
    z4 = synth.make_tree(30, Q4, 1.0, 0x78, np.full(4, .25))
    nen, nodelist, root = treeorder.pruningwiseedgeorder(z4), treeorder.makenodelist(z4), treeorder.myreorder(z4)
    got, rc = O.maketreelistMCMC(z4, Q4, np.full(4, .25), np.eye(4) + Q4 / Omega, Omega, nen, nodelist, root, 50, variant=O.KS, seed=3,
                                 prior=[1, 10, 2, 10, 20, 2], rstream=True)
    assert rc == 0
    np.testing.assert_allclose(got[:, :4].sum(1), z4["edge.length"].sum(), rtol=1e-12)
