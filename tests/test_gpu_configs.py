"""Every BASELINE.json configuration at its STATED size on the GPU, against the CPU oracle.

C1 100 tips / 2 states / sumstatEXP; C2 1 000 tips / 4 states / sumstatMCMC; C3 10 000 tips / 4 states /
sumstatMCMC_bigtree (src/phylomap.cpp:942-986); C4 500 tips / dense 61-state Q; C5 5 000 tips / sparse 20-state Q
(SPARSEsumstatMCMC, src/phylomap.cpp:822-870).  Per configuration: >= 8 sweeps, replicas {0, 63, last} against
`O.maketreelistMCMC` (counts array_equal, dwell <= 1e-10), a chain-state dump of one replica, and every mapping that
serves the state count.  The golden fixtures of tests/golden/ are replayed through the C-ABI as well."""
import os

import numpy as np
import pytest

import oracle_lib as O
from phylomap_amd import _lib, api, synth, treeorder

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
_CACHE = {}


def _config(cfg):
    """(z, Q, pid, Omega, nen, nodelist, root) of a BASELINE configuration; built once per session."""
    if cfg not in _CACHE:
        z, Q, pid, Omega = synth.config_problem(cfg)
        _CACHE[cfg] = (z, Q, pid, Omega, treeorder.pruningwiseedgeorder(z), treeorder.makenodelist(z), treeorder.myreorder(z))
    return _CACHE[cfg]


def _check_rows(got, want, n, exact_dwell):
    np.testing.assert_array_equal(got[:, n:], want[:, n:])            # transition counts: bit-exact
    if exact_dwell:
        np.testing.assert_array_equal(got[:, :n], want[:, :n])
    else:
        np.testing.assert_allclose(got[:, :n], want[:, :n], rtol=1e-10, atol=0)


def _check_dump(d, dump):
    np.testing.assert_array_equal(d["seg_count"], dump.seg_count)
    np.testing.assert_array_equal(d["node_states"], dump.node_states)
    np.testing.assert_array_equal(d["PL"], dump.PL)
    for b in range(len(dump.seg_count)):
        np.testing.assert_array_equal(d["seg_dwell"][b, :dump.seg_count[b]], dump.seg_dwell[b, :dump.seg_count[b]])


def _run_config(cfg, variant, orc_variant, mapping, S, N, seed, exact_dwell, dump_replica=None, **opt):
    z, Q, pid, Omega, nen, nodelist, root = _config(cfg)
    n = Q.shape[0]
    B = np.eye(n) + Q / Omega
    eng = _lib.Engine(z, Q, pid, Omega, N, variant=variant, seed=seed, n_replicas=S, mapping=mapping, **opt)
    eng.run(N); eng.sync()
    st = eng.stats(0, N)
    assert st.shape == (S, N, n + n * (n - 1))
    for r in sorted({0, min(63, S - 1), S - 1}):
        want, rc = O.maketreelistMCMC(z, Q, pid, B, Omega, nen, nodelist, root, N, variant=orc_variant, seed=seed, replica=r)
        assert rc == 0
        _check_rows(st[r], want, n, exact_dwell)
    np.testing.assert_allclose(st[:, :, :n].sum(2), z["edge.length"].sum(), rtol=1e-11)      # every replica, every sweep
    if dump_replica is not None:
        _, rc, dump = O.maketreelistMCMC(z, Q, pid, B, Omega, nen, nodelist, root, N, variant=orc_variant, seed=seed,
                                         replica=dump_replica, dump=True)
        assert rc == 0
        _check_dump(eng.dump(dump_replica), dump)
    eng.close()
    return st


def test_c1_sumstatEXP_100_tips():
    """C1: simulate_2_state_tree-shaped problem, ~100 tips, sumstatEXP (src/phylomap.cpp:3001-3051)."""
    z, Q, pid, Omega, nen, nodelist, root = _config(1)
    lefts, rights, d = api.eigen_decompose(Q)
    N = 1000                                        # C1 runs N = 1 000 samples
    got = api.sumstatEXP(z, Q, pid, N, seed=11)
    want, rc = O.maketreelistEXP(z, Q, pid, nen, nodelist, root, N, lefts, rights, d, seed=11)
    assert rc == 0 and got.shape == (N, 4)
    np.testing.assert_array_equal(got[:, 2:], want[:, 2:])
    np.testing.assert_allclose(got[:, :2], want[:, :2], rtol=1e-10, atol=0)
    np.testing.assert_allclose(got[:, :2].sum(1), z["edge.length"].sum(), rtol=1e-12)
    # the same tree through the MCMC drivers (the configuration's tree is small enough for the plain variant)
    for mapping in ("replicas", "branches", "tiles"):
        gm = api.sumstatMCMC(z, Q, pid, Omega, 16, seed=5, n_replicas=2, mapping=mapping)
        for r in range(2):
            wm, rc = O.maketreelistMCMC(z, Q, pid, np.eye(2) + Q / Omega, Omega, nen, nodelist, root, 16, seed=5, replica=r)
            assert rc == 0
            _check_rows(gm[r], wm, 2, mapping == "replicas")


@pytest.mark.parametrize("mapping", ["replicas", "tiles", "branches"])
def test_c2_sumstatMCMC_1000_tips(mapping):
    S = {"replicas": 128, "tiles": 128, "branches": 3}[mapping]
    _run_config(2, _lib.PHM_MCMC_BIGTREE, O.BIGTREE, mapping, S, 10, 2024, mapping == "replicas", dump_replica=S - 1)


def test_c3_bigtree_10000_tips_replica_mapping():
    """C3 as stated (4 states, 10 000 tips, _bigtree): the streaming layout, one ring per tile."""
    _run_config(3, _lib.PHM_MCMC_BIGTREE, O.BIGTREE, "replicas", 128, 8, 31337, True, dump_replica=127, storage=1)


def test_c3_bigtree_10000_tips_tiles_mapping_with_grouped_branches():
    """1 024 replicas = 16 tiles x 19 998 branches: tiles_branch_kernel walks 4 branches per wave (group > 1); every replica
    is compared with the replica mapping, replicas {0, 63, 1023} with the oracle, one replica's chain state dumped."""
    z, Q, pid, Omega = _config(3)[:4]
    S, N, seed = 1024, 8, 31337
    st = _run_config(3, _lib.PHM_MCMC_BIGTREE, O.BIGTREE, "tiles", S, N, seed, False, dump_replica=1000)
    eng = _lib.Engine(z, Q, pid, Omega, N, variant=_lib.PHM_MCMC_BIGTREE, seed=seed, n_replicas=S, mapping="replicas", storage=1)
    eng.run(N); eng.sync()
    ref = eng.stats(0, N)
    eng.close()
    np.testing.assert_array_equal(st[:, :, 4:], ref[:, :, 4:])
    np.testing.assert_allclose(st[:, :, :4], ref[:, :, :4], rtol=1e-10, atol=0)


def test_c3_one_chain_branch_mapping():
    """the reference's own calling pattern on C3: ONE chain (R/sumstatMCMC_bigtree.R), lane = branch"""
    _run_config(3, _lib.PHM_MCMC_BIGTREE, O.BIGTREE, "branches", 1, 8, 99, False, dump_replica=0)


WIDE_MAPPINGS = ["replicas", "branches", "tiles"]


@pytest.mark.parametrize("mapping", WIDE_MAPPINGS)
def test_c4_dense_61_states_500_tips(mapping):
    """C4: dense 61-state Q on a 500-tip tree (row-normalised pruning: 61-state partial likelihoods underflow without)."""
    S = {"replicas": 128, "branches": 66, "tiles": 128}[mapping]
    _run_config(4, _lib.PHM_MCMC_BIGTREE, O.BIGTREE, mapping, S, 8, 4242, False, dump_replica=S - 1)


def _check_ks_rows(got, want, n):
    """n dwell sums (1e-10), n x n counts incl. self pairs (exact), parameter columns and root state (exact)"""
    assert got.shape == want.shape
    np.testing.assert_array_equal(got[:, n:], want[:, n:])
    np.testing.assert_allclose(got[:, :n], want[:, :n], rtol=1e-10, atol=0)


@pytest.mark.parametrize("mapping,S", [("replicas", 70), ("branches", 66), ("tiles", 128), ("tiles", 8512)])
def test_c4_61_states_in_the_n_plus_n2_counting_layout(mapping, S):
    """C4 as SURVEY section 8 scopes it: the `ks layout` = n dwell sums + n x n counts with self pairs (shortenerbf,
    src/phylomap.cpp:997-1028; sampleabranchbf :1031-1074; treesamplebf :1169-1179) on the dense 61-state Q -- odd n, so
    observed tips (the bf sweep; the hidden-rates masks need even n).  8 512 replicas = 133 tiles: the branch kernel's
    KS template with grouped branches and 61 x 61 counters per lane."""
    z, Q, pid, Omega, nen, nodelist, root = _config(4)
    n, N, seed = 61, (3 if S > 1000 else 6), 6100 + S
    got = api.sumstatMCMCbf_sweep(z, Q, pid, Omega, N, seed=seed, n_replicas=S, mapping=mapping)
    assert got.shape == (S, N, n + n * n + 3)
    for r in sorted({0, 17, min(63, S - 1), min(S // 2 + 50, S - 1), S - 1}):
        want, rc = O.maketreelistMCMC(z, Q, pid, np.eye(n) + Q / Omega, Omega, nen, nodelist, root, N, variant=O.BF, seed=seed, replica=r)
        assert rc == 0
        _check_ks_rows(got[r], want, n)
    np.testing.assert_allclose(got[:, :, :n].sum(2), z["edge.length"].sum(), rtol=1e-11)
    cnt = got[:, :, n:n + n * n]
    assert np.all(cnt == np.round(cnt)) and cnt.sum() > 0
    if S <= 128:      # summed over replicas on the device = the sum of the per-replica rows
        red = api.sumstatMCMCbf_sweep(z, Q, pid, Omega, N, seed=seed, n_replicas=S, mapping=mapping, reduce=True)
        np.testing.assert_array_equal(red[:, n:n + n * n], got.sum(0)[:, n:n + n * n])
        np.testing.assert_allclose(red[:, :n], got.sum(0)[:, :n], rtol=1e-12)


def _hidden_rates_62():
    """62 states = 31 rate regimes of a binary trait (make2sQ, R/sourceme.R:229-246) on the C4 tree; only the trait is observed"""
    if "hr62" not in _CACHE:
        k = 30
        rk = 0.02 + 0.01 * (np.arange(k) % 5)
        lk = 0.03 + 0.01 * (np.arange(k) % 3)
        gam = 0.5 + 0.25 * (np.arange(k) % 7)
        Q = synth.make2sQ(0.05, 0.08, rk, lk, gam)
        n = Q.shape[0]
        Omega = 1.25 * float(np.max(np.abs(np.diag(Q))))
        pid = np.full(n, 1.0 / n)
        z = synth.make_tree(500, Q, Omega, 0x5EED0004, pid)
        z["states"] = ((z["states"] - 1) % 2 + 1).astype(np.int32)
        for b, (p_, c_) in enumerate(z["edge"]):
            if c_ <= 500:
                z["mapnames"][b][-1] = z["states"][c_ - 1]
        _CACHE["hr62"] = (z, Q, pid, Omega, treeorder.pruningwiseedgeorder(z), treeorder.makenodelist(z), treeorder.myreorder(z))
    return _CACHE["hr62"]


@pytest.mark.parametrize("mapping,S", [("replicas", 70), ("branches", 66), ("tiles", 128), ("tiles", 8512)])
def test_hidden_rates_62_states_500_tips_ks_sweep(mapping, S):
    """sumstatMCMCks' tree sweep (src/phylomap.cpp:1422-1432) at the size C4 names: n = 2k+2 = 62, parity tip masks
    (:1838-1845), tips re-sampled (:1384-1397), n x n counts, recordQks columns (2 + 3k = 92), root state."""
    z, Q, pid, Omega, nen, nodelist, root = _hidden_rates_62()
    n, k, N, seed = 62, 30, (3 if S > 1000 else 6), 6200 + S
    got = api.sumstatMCMCks_sweep(z, Q, pid, Omega, N, seed=seed, n_replicas=S, mapping=mapping)
    assert got.shape == (S, N, n + n * n + 2 + 3 * k + 1)
    for r in sorted({0, 17, min(63, S - 1), min(S // 2 + 50, S - 1), S - 1}):
        want, rc = O.maketreelistMCMC(z, Q, pid, np.eye(n) + Q / Omega, Omega, nen, nodelist, root, N, variant=O.KS, seed=seed, replica=r)
        assert rc == 0
        _check_ks_rows(got[r], want, n)
    np.testing.assert_allclose(got[:, :, :n].sum(2), z["edge.length"].sum(), rtol=1e-11)


@pytest.mark.parametrize("mapping", WIDE_MAPPINGS)
def test_c5_sparse_20_states_5000_tips(mapping):
    """C5: tridiagonal 20-state Q on a 5 000-tip tree.  SPARSEsumstatMCMC has no rescaling (src/phylomap.cpp:490-501), so at
    this size the reference's arithmetic underflows -- oracle and GPU must both say so -- and the sweep itself is checked
    with the row-normalised pruning on the same sparse matrix (no entry of this B is below the 1e-7 threshold of :811, so
    the thresholded and the dense matrix coincide and every other step is the SPARSE driver's)."""
    z, Q, pid, Omega, nen, nodelist, root = _config(5)
    _, rc = O.maketreelistMCMC(z, Q, pid, np.eye(20) + Q / Omega, Omega, nen, nodelist, root, 2, variant=O.SPARSE, seed=1)
    assert rc & O.ERR_ZERO_PROB
    with pytest.raises(_lib.PhmError) as e:
        api.SPARSEsumstatMCMC(z, Q, pid, Omega, 2, seed=1, n_replicas=2, mapping=mapping)
    assert e.value.status == 5
    S = {"replicas": 128, "branches": 66, "tiles": 128}[mapping]
    _run_config(5, _lib.PHM_MCMC_BIGTREE, O.BIGTREE, mapping, S, 8, 5151, False, dump_replica=S - 1)
    # ... and the SPARSE driver itself at the stated size, with its pruning pass rescaled (phm_options.reserved[3]; oracle:
    # SPARSE | FORCE_NORMALISE): thresholded chain matrix, dense forward rows, rows divided by their sum
    _run_config(5, _lib.PHM_MCMC_SPARSE, O.SPARSE | O.FORCE_NORMALISE, mapping, 70, 8, 6161, False, dump_replica=69, rescale=True)


@pytest.mark.parametrize("S", [200, 7232])
def test_c5_band_kernels_equal_the_dense_kernels_bit_for_bit(S):
    """C5's B is tridiagonal: the lane-per-replica mapping then prunes with per-lane FMAs over the band (wt_up_band_kernel) and
    draws forward states over the band of a row -- the use SPARSEmakePLrcpp / SPARSEresamplebranchstates make of sp_mat
    (src/phylomap.cpp:490-501, :218-261).  A skipped term is an exact zero, so everything must equal the dense kernels
    (matrix cores + running-sum tables) to the last bit: statistics, partial likelihoods, node states, paths."""
    z, Q, pid, Omega, nen, nodelist, root = _config(5)
    n, N, seed = 20, 5, 5500 + S
    res = {}
    for sc in (1, 2):
        eng = _lib.Engine(z, Q, pid, Omega, N, variant=_lib.PHM_MCMC_BIGTREE, seed=seed, n_replicas=S, mapping="tiles", sparse_chains=sc)
        eng.run(N); eng.sync()
        assert eng.info().sparse_chains == (3 if sc == 1 else 0)
        res[sc] = (eng.stats(0, N), eng.dump(S - 1))
        eng.close()
    np.testing.assert_array_equal(res[1][0], res[2][0])
    for key in ("seg_count", "node_states", "PL", "seg_dwell"):
        np.testing.assert_array_equal(res[1][1][key], res[2][1][key])
    for r in (0, S - 1):
        want, rc = O.maketreelistMCMC(z, Q, pid, np.eye(n) + Q / Omega, Omega, nen, nodelist, root, N, variant=O.BIGTREE, seed=seed, replica=r)
        assert rc == 0
        _check_rows(res[1][0][r], want, n, False)
    # a dense matrix has no band to offer
    z4, Q4, pid4, Om4 = _config(4)[:4]
    with pytest.raises(_lib.PhmError) as e:
        _lib.Engine(z4, Q4, pid4, Om4, 1, variant=_lib.PHM_MCMC_BIGTREE, n_replicas=64, mapping="tiles", sparse_chains=1)
    assert e.value.status == 2


def _neighbour_problem(tips):
    """C5's tree size with an unstructured sparse Q (degree-6 neighbour graph, not banded)"""
    Q = synth.neighbour_Q(20, 6)
    Omega = 1.25 * float(np.max(np.abs(np.diag(Q))))
    pid = np.full(20, 0.05)
    z = synth.make_tree(tips, Q, Omega, 0x5EED0005, pid, init_segments=20)
    return z, Q, pid, Omega, treeorder.pruningwiseedgeorder(z), treeorder.makenodelist(z), treeorder.myreorder(z)


@pytest.mark.parametrize("mapping", WIDE_MAPPINGS)
def test_c5_unstructured_sparse_20_states_5000_tips(mapping):
    """BASELINE configs[4] says "sparse 20-state amino-acid Q": a neighbour structure, not a band.  SPARSEmakePLrcpp / spmmmmvFORpl
    (src/phylomap.cpp:490-501, :451-457) walk the non-zeros of ANY sp_mat; here the lane-per-replica mapping prunes with a kernel
    generated for the pattern of the chain matrix (phm_rtc.h).  5 000 tips, every mapping against the oracle."""
    z, Q, pid, Omega, nen, nodelist, root = _neighbour_problem(5000)
    n, N, seed = 20, 4, 909
    hb = max(abs(i - j) for i in range(n) for j in range(n) if Q[i, j] != 0.0)
    assert hb > 2 and np.count_nonzero(Q) == n * 7                       # not banded; 6 neighbours + the diagonal
    S = {"replicas": 66, "branches": 6, "tiles": 130}[mapping]
    got = api.sumstatMCMC_bigtree(z, Q, pid, Omega, N, seed=seed, n_replicas=S, mapping=mapping)
    for r in (0, S - 1):
        want, rc = O.maketreelistMCMC(z, Q, pid, np.eye(n) + Q / Omega, Omega, nen, nodelist, root, N, variant=O.BIGTREE, seed=seed, replica=r)
        assert rc == 0
        _check_rows(got[r], want, n, False)


@pytest.mark.parametrize("S", [70, 4100])
def test_c5_unstructured_generated_kernel_equals_the_dense_kernels_bit_for_bit(S):
    """sparse_chains = 1 (generated kernel required) against 2 (matrix cores): a skipped term is an exact zero, so statistics, partial
    likelihoods, node states and paths agree to the last bit; the SPARSE driver (thresholded matrix, rescaled) likewise; a rate update
    that keeps the pattern reuses the kernel; a dense matrix is refused."""
    z, Q, pid, Omega, nen, nodelist, root = _neighbour_problem(600)
    n, N, seed = 20, 5, 4400 + S
    for variant, orc in ((_lib.PHM_MCMC_BIGTREE, O.BIGTREE), (_lib.PHM_MCMC_SPARSE, O.SPARSE | O.FORCE_NORMALISE)):
        res = {}
        for sc in (1, 2):
            eng = _lib.Engine(z, Q, pid, Omega, N, variant=variant, seed=seed, n_replicas=S, mapping="tiles", sparse_chains=sc,
                              rescale=variant == _lib.PHM_MCMC_SPARSE)
            eng.run(2); eng.sync()
            assert eng.info().sparse_chains == (5 if sc == 1 else 0)
            eng.set_model(Q * 0.9)                       # same zeros, other values: the kernel of the pattern with new coefficients
            eng.run(N - 2); eng.sync()
            assert eng.info().sparse_chains == (5 if sc == 1 else 0)
            res[sc] = (eng.stats(0, N), eng.dump(S - 1))
            eng.close()
        np.testing.assert_array_equal(res[1][0], res[2][0])
        for key in ("seg_count", "node_states", "PL", "seg_dwell"):
            np.testing.assert_array_equal(res[1][1][key], res[2][1][key])
        want, rc = O.maketreelistMCMC(z, Q, pid, np.eye(n) + Q / Omega, Omega, nen, nodelist, root, 2, variant=orc, seed=seed, replica=S - 1)
        assert rc == 0
        _check_rows(res[1][0][S - 1][:2], want, n, False)
    # the automatic choice takes the generated kernel too
    eng = _lib.Engine(z, Q, pid, Omega, 1, variant=_lib.PHM_MCMC_BIGTREE, n_replicas=S, mapping="tiles")
    assert eng.info().sparse_chains & 4
    eng.close()
    z4, Q4, pid4, Om4 = _config(4)[:4]
    with pytest.raises(_lib.PhmError) as e:
        _lib.Engine(z4, Q4, pid4, Om4, 1, variant=_lib.PHM_MCMC_BIGTREE, n_replicas=64, mapping="tiles", sparse_chains=1)
    assert e.value.status == 2


@pytest.mark.parametrize("mapping", WIDE_MAPPINGS)
def test_c5_sparse_variant_on_the_largest_tree_it_survives(mapping):
    """the SPARSE driver proper (thresholded chain matrix, dense forward rows, no rescaling) on a 200-tip tree"""
    z, Q, pid, Omega = synth.config_problem(5, n_tips=200)
    nen, nodelist, root = treeorder.pruningwiseedgeorder(z), treeorder.makenodelist(z), treeorder.myreorder(z)
    N, S, seed = 8, 70, 77
    got = api.SPARSEsumstatMCMC(z, Q, pid, Omega, N, seed=seed, n_replicas=S, mapping=mapping)
    for r in (0, 63, S - 1):
        want, rc = O.maketreelistMCMC(z, Q, pid, np.eye(20) + Q / Omega, Omega, nen, nodelist, root, N, variant=O.SPARSE,
                                      seed=seed, replica=r)
        assert rc == 0
        _check_rows(got[r], want, 20, False)


@pytest.mark.parametrize("name", ["n2_t16", "n3_t16", "n4_t16", "n4_t64", "n20_t12"])
def test_golden_fixtures_through_the_c_abi(name):
    """tests/golden/*.npz (inputs + expected matrices, generated by tests/golden/make_golden.py) replayed on the GPU"""
    from golden.make_golden import unpack_tree
    g = np.load(os.path.join(HERE, "golden", name + ".npz"))
    z = unpack_tree(g)
    Q, pid, Omega, seed = g["Q"], g["pid"], float(g["Omega"]), int(g["seed"])
    n = Q.shape[0]
    mappings = ["replicas", "branches", "tiles"]
    for key, fn in (("mcmc", api.sumstatMCMC), ("bigtree", api.sumstatMCMC_bigtree), ("sparse", api.SPARSEsumstatMCMC)):
        for mapping in mappings:
            got = fn(z, Q, pid, Omega, 24, seed=seed, n_replicas=6, mapping=mapping)
            for r in (0, 5):
                _check_rows(got[r], g[f"{key}_r{r}"], n, mapping == "replicas" and n <= 4)
    if "exp_r0" in g.files:
        got = api.sumstatEXP(z, Q, pid, int(g["exp_r0"].shape[0]), seed=seed, eig=(g["lefts"], g["rights"], g["d"]))
        np.testing.assert_array_equal(got[:, n:], g["exp_r0"][:, n:])
        np.testing.assert_allclose(got[:, :n], g["exp_r0"][:, :n], rtol=1e-10, atol=0)


def test_exp_rescaled_pruning_at_c2_size_and_equal_to_plain_on_small_trees():
    """sumstatEXP with the pruning pass rescaled (phm_options.reserved[3]; not in the reference, whose makePLexp
    src/phylomap.cpp:2899-2906 underflows beyond a few hundred tips): 1 000-tip C2 tree against the oracle; plain raises like
    RcppArmadillo::sample would; on the 100-tip C1 tree both samplers draw the same histories."""
    z, Q, pid, Omega, nen, nodelist, root = _config(2)
    lefts, rights, d = api.eigen_decompose(Q)
    with pytest.raises(_lib.PhmError) as e:
        api.sumstatEXP(z, Q, pid, 8, seed=3)
    assert e.value.status == 5
    N = 96
    got = api.sumstatEXP(z, Q, pid, N, seed=3, rescale=True)
    want, rc = O.maketreelistEXP(z, Q, pid, nen, nodelist, root, N, lefts, rights, d, seed=3, rescale=True)
    assert rc == 0
    np.testing.assert_array_equal(got[:, 4:], want[:, 4:])
    np.testing.assert_allclose(got[:, :4], want[:, :4], rtol=1e-10, atol=0)
    np.testing.assert_allclose(got[:, :4].sum(1), z["edge.length"].sum(), rtol=1e-12)
    z1, Q1, pid1 = _config(1)[:3]
    a = api.sumstatEXP(z1, Q1, pid1, 512, seed=9)
    b = api.sumstatEXP(z1, Q1, pid1, 512, seed=9, rescale=True)
    np.testing.assert_array_equal(a[:, 2:], b[:, 2:])
    np.testing.assert_allclose(a[:, :2], b[:, :2], rtol=1e-10, atol=0)
    # 20 states (exp_wide_kernel) on a tree the plain pass cannot finish
    z5, Q5, pid5, _, nen5, nl5, root5 = _config(5)
    Qs = (Q5 + Q5.T) / 2
    l5, r5, d5 = api.eigen_decompose(Qs)
    got5 = api.sumstatEXP(z5, Qs, pid5, 64, seed=4, rescale=True, eig=(l5, r5, d5))
    want5, rc = O.maketreelistEXP(z5, Qs, pid5, nen5, nl5, root5, 64, l5, r5, d5, seed=4, rescale=True)
    assert rc == 0
    np.testing.assert_array_equal(got5[:, 20:], want5[:, 20:])
    np.testing.assert_allclose(got5[:, :20], want5[:, :20], rtol=1e-10, atol=0)


# 16 / 10 tiles: a workgroup per 16-replica block, a wave per 16-state row block; 113 tiles: the same at 61 states, sorted blocks
# per (node, tile) at 20; forms 1 / 2: each pruning kernel at a tile count the automatic choice gives to the other
@pytest.mark.parametrize("cfg,S,form", [(4, 1024, 0), (5, 640, 0), (4, 7232, 0), (5, 7232, 0), (4, 7232, 1), (5, 7232, 2), (4, 192, 1), (4, 7232, 3)])
def test_wide_lane_per_replica_mapping_with_many_tiles(cfg, S, form):
    """phm_wtiles.hip beyond a couple of tiles: persistent pruning waves striding over (node, tile) items, branch groups per
    workgroup, per-tile accumulators -- C4 / C5 at their stated sizes with 16 / 10 tiles, replicas from different tiles and
    different 16-column MFMA blocks against the oracle, tree-length invariant on every replica and sweep."""
    z, Q, pid, Omega, nen, nodelist, root = _config(cfg)
    n = Q.shape[0]
    N, seed = 6, 900 + cfg
    eng = _lib.Engine(z, Q, pid, Omega, N, variant=_lib.PHM_MCMC_BIGTREE, seed=seed, n_replicas=S, mapping="tiles", pruning_form=form)
    eng.run(N); eng.sync()
    st = eng.stats(0, N)
    for r in (0, 17, 64 + 33, S // 2 + 50, S - 1):
        want, rc = O.maketreelistMCMC(z, Q, pid, np.eye(n) + Q / Omega, Omega, nen, nodelist, root, N, variant=O.BIGTREE, seed=seed, replica=r)
        assert rc == 0
        _check_rows(st[r], want, n, False)
    np.testing.assert_allclose(st[:, :, :n].sum(2), z["edge.length"].sum(), rtol=1e-11)
    assert np.all(st[:, :, n:] == np.round(st[:, :, n:]))
    # the reduced output (sum over replicas per sweep) equals the sum of the per-replica rows
    red = _lib.Engine(z, Q, pid, Omega, N, variant=_lib.PHM_MCMC_BIGTREE, seed=seed, n_replicas=S, mapping="tiles", reduce=True, pruning_form=form)
    red.run(N); red.sync()
    tot = red.stats(0, N)
    red.close()
    eng.close()
    np.testing.assert_array_equal(tot[:, n:], st.sum(0)[:, n:])
    np.testing.assert_allclose(tot[:, :n], st.sum(0)[:, :n], rtol=1e-12)
