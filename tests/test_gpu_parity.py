"""GPU parity: the HIP path (through the C-ABI) against the CPU oracle on the same seeded inputs.

Bar (BASELINE.json north_star): transition counts bit-exact under a fixed seed, dwell times within 1e-10
relative.  On cladewise trees the summation orders coincide and the dwell columns are bit-identical too."""
import numpy as np
import pytest

import oracle_lib as O
from phylomap_amd import _lib, api, synth, treeorder

pytestmark = pytest.mark.gpu


def _orders(z):
    return treeorder.pruningwiseedgeorder(z), treeorder.makenodelist(z), treeorder.myreorder(z)


def _problem(n, tips, seed, omega_factor=1.25):
    Q = {2: synth.config_Q(1), 3: np.array([[-.3, .2, .1], [.05, -.15, .1], [.2, .2, -.4]]), 4: synth.config_Q(2)}[n]
    Omega = omega_factor * float(np.max(np.abs(np.diag(Q))))
    pid = np.full(n, 1.0 / n)
    z = synth.make_tree(tips, Q, Omega, seed, pid)
    return z, Q, pid, Omega


MAPPINGS = ["replicas", "branches", "tiles"]     # lane per chain in one wave per tile / lane per branch / wave per (tile, branch)


def _same(got, want, n, mapping, ks=False):
    """The bar of BASELINE.json: counts bit-exact, dwell times within 1e-10 relative.  The replica mapping adds the dwell
    times in the reference's order (bit-identical on cladewise trees); the branch mapping adds them per branch and then
    reduces, and whatever is derived from those sums (updated rates, log-likelihoods) inherits the rounding."""
    got, want = np.asarray(got), np.asarray(want)
    assert got.shape == want.shape
    if mapping == "replicas":
        np.testing.assert_array_equal(got, want)
        return
    ncnt = n * n if ks else n * (n - 1)
    np.testing.assert_array_equal(got[..., n:n + ncnt], want[..., n:n + ncnt])
    np.testing.assert_allclose(got[..., :n], want[..., :n], rtol=1e-10, atol=0)
    np.testing.assert_allclose(got[..., n + ncnt:], want[..., n + ncnt:], rtol=1e-9, atol=0)


VARIANTS = [("sumstatMCMC", O.PLAIN), ("sumstatMCMC_bigtree", O.BIGTREE), ("SPARSEsumstatMCMC", O.SPARSE)]


@pytest.mark.parametrize("mapping", MAPPINGS)
@pytest.mark.parametrize("n", [2, 3, 4])
@pytest.mark.parametrize("fn,variant", VARIANTS)
def test_mcmc_matches_oracle_small(n, fn, variant, mapping):
    z, Q, pid, Omega = _problem(n, 24, 1234 + n)
    nen, nodelist, root = _orders(z)
    B = np.eye(n) + Q / Omega
    N, S, seed = 40, 3, 0xC0FFEE
    got = getattr(api, fn)(z, Q, pid, Omega, N, seed=seed, n_replicas=S, mapping=mapping)
    assert got.shape == (S, N, n + n * (n - 1))
    for r in range(S):
        want, rc = O.maketreelistMCMC(z, Q, pid, B, Omega, nen, nodelist, root, N, variant=variant, seed=seed, replica=r)
        assert rc == 0
        _same(got[r], want, n, mapping)      # counts bit-exact; dwell bit-exact (replicas: same summation order) / 1e-10
        np.testing.assert_allclose(got[r][:, :n].sum(1), z["edge.length"].sum(), rtol=1e-12)


def test_mcmc_single_chain_is_drop_in_shape():
    z, Q, pid, Omega = _problem(4, 16, 77)
    nen, nodelist, root = _orders(z)
    got = api.sumstatMCMC(z, Q, pid, Omega, 25, seed=5)
    want, rc = O.maketreelistMCMC(z, Q, pid, np.eye(4) + Q / Omega, Omega, nen, nodelist, root, 25, seed=5)
    assert rc == 0 and got.shape == (25, 16)
    _same(got, want, 4, "branches")                    # one chain: the automatic choice is the branch mapping
    np.testing.assert_array_equal(api.sumstatMCMC(z, Q, pid, Omega, 25, seed=5, mapping="replicas"), want)


@pytest.mark.parametrize("storage", [1, 2, 0, -1])   # 1: one ring per tile; 2: two buffers; 0: branch mapping; -1: tiles mapping
def test_mcmc_chain_state_matches_oracle(storage):
    z, Q, pid, Omega = _problem(4, 40, 99)
    nen, nodelist, root = _orders(z)
    N, seed = 15, 42
    mapping = "replicas" if storage > 0 else ("branches" if storage == 0 else "tiles")
    storage = max(storage, 0)
    eng = _lib.Engine(z, Q, pid, Omega, N, variant=_lib.PHM_MCMC_BIGTREE, seed=seed, n_replicas=70, storage=storage, mapping=mapping)
    eng.run(7); eng.run(N - 7); eng.sync()
    for r in (0, 63, 69):
        want, rc, dump = O.maketreelistMCMC(z, Q, pid, np.eye(4) + Q / Omega, Omega, nen, nodelist, root, N,
                                            variant=O.BIGTREE, seed=seed, replica=r, dump=True)
        assert rc == 0
        got = eng.dump(r)
        np.testing.assert_array_equal(got["seg_count"], dump.seg_count)
        np.testing.assert_array_equal(got["node_states"], dump.node_states)
        for b in range(len(dump.seg_count)):
            m = dump.seg_count[b]
            np.testing.assert_array_equal(got["seg_dwell"][b, :m], dump.seg_dwell[b, :m])
        np.testing.assert_array_equal(got["PL"], dump.PL)
        _same(eng.stats(0, N)[r], want, 4, mapping)
    eng.close()


@pytest.mark.parametrize("storage", [1, 2, 0, -1])    # 0: the branch mapping (CSR slots; no ring / buffer choice); -1: tiles
def test_mcmc_long_initial_paths_take_the_general_branch_path(storage):
    """100 equal segments per branch (R/Squamate_tree_setup.R:57): exercises the > 64-segment code path and the
    hand-over to the packed two-pass path once the chain has shrunk the paths."""
    Q = synth.config_Q(2)
    Omega = 1.25 * float(np.max(np.abs(np.diag(Q))))
    pid = np.full(4, 0.25)
    z = synth.make_tree(14, Q, Omega, 21, pid, init_segments=100)
    nen, nodelist, root = _orders(z)
    mapping = "replicas" if storage > 0 else ("branches" if storage == 0 else "tiles")
    got = api.sumstatMCMC(z, Q, pid, Omega, 12, seed=13, n_replicas=2, storage=max(storage, 0), mapping=mapping)
    for r in range(2):
        want, rc = O.maketreelistMCMC(z, Q, pid, np.eye(4) + Q / Omega, Omega, nen, nodelist, root, 12, seed=13, replica=r)
        assert rc == 0
        _same(got[r], want, 4, mapping)


@pytest.mark.parametrize("fn,variant,n", [("sumstatMCMC", O.PLAIN, 2), ("sumstatMCMC_bigtree", O.BIGTREE, 4), ("sumstatMCMCks_sweep", O.KS, 4)])
def test_tiles_mapping_on_paths_of_hundreds_of_segments(fn, variant, n):
    """(tile, branch) mapping, 2 .. 4 states, a branch expected to hold more than 48 segments: the branch kernel without its limit of
    64 merged segments per branch and lane (tiles_branch_kernel<NS, KS, true>: their states in a byte per (row, lane) instead of two
    registers; 13x on the reference's squamate tree).  300-segment paths to start with, one branch of 1 500, several sweeps, two tiles."""
    Q = synth.config_Q(1) if n == 2 else synth.config_Q(2)
    Omega = 1.25 * float(np.max(np.abs(np.diag(Q))))
    pid = np.full(n, 1.0 / n)
    z = synth.make_tree(12, Q, Omega, 33, pid, init_segments=300)
    if variant == O.KS:
        z = dict(z, states=((z["states"] - 1) % 2 + 1).astype(np.int32))
        for b, (p_, c_) in enumerate(z["edge"]):
            if c_ <= 12:
                z["mapnames"][b][-1] = z["states"][c_ - 1]
    b = int(np.argmax(z["edge.length"]))
    end = z["mapnames"][b][-1]
    z["maps"][b] = np.full(1500, 60.0 * z["edge.length"][b] / 1500)       # a long branch: ~ 60 x 4 virtual jumps per sweep as well
    z["mapnames"][b] = np.array([1] * 1499 + [end], dtype=np.int32)
    z["edge.length"] = np.array([mp.sum() for mp in z["maps"]])
    nen, nodelist, root = _orders(z)
    S, N = 70, 5
    got = getattr(api, fn)(z, Q, pid, Omega, N, seed=19, n_replicas=S, mapping="tiles")
    for r in (0, 63, 64, S - 1):
        want, rc = O.maketreelistMCMC(z, Q, pid, np.eye(n) + Q / Omega, Omega, nen, nodelist, root, N, variant=variant, seed=19, replica=r)
        assert rc == 0
        _same(got[r], want, n, "tiles", ks=variant == O.KS)


def test_tiles_mapping_general_loop_when_a_lane_outgrows_64_segments_unexpectedly():
    """Expected path lengths below 48 on every branch keep the two-pass form with its states packed in registers; a lane that still
    draws more than 64 segments (Poisson(45): 0.4 % of the draws -- a handful in 130 chains x 12 sweeps) sends its wave through
    the lane-sequential general loop for that branch."""
    Q = synth.config_Q(1)
    Omega = 1.25 * float(np.max(np.abs(np.diag(Q))))
    pid = np.full(2, 0.5)
    z = synth.make_tree(6, Q, Omega, 8, pid, init_segments=2)
    b = int(np.argmax(z["edge.length"]))
    z["maps"][b] = z["maps"][b] * (45.0 / (Omega * z["edge.length"][b]))
    z["edge.length"] = np.array([mp.sum() for mp in z["maps"]])
    nen, nodelist, root = _orders(z)
    S, N = 130, 12
    got = api.sumstatMCMC(z, Q, pid, Omega, N, seed=23, n_replicas=S, mapping="tiles")
    for r in (0, 5, 63, 64, 100, S - 1):
        want, rc = O.maketreelistMCMC(z, Q, pid, np.eye(2) + Q / Omega, Omega, nen, nodelist, root, N, seed=23, replica=r)
        assert rc == 0
        _same(got[r], want, 2, "tiles")


def _two_tip_tree(states, lens=(1.5, 0.7), segs=(1, 1)):
    edge = np.array([[3, 1], [3, 2]], dtype=np.int32)
    maps = [np.full(segs[i], lens[i] / segs[i]) for i in range(2)]
    mapnames = [np.array([1] * (segs[i] - 1) + [states[i]], dtype=np.int32) for i in range(2)]
    return {"edge": edge, "Nnode": 1, "edge.length": np.asarray(lens, dtype=float), "states": np.asarray(states, dtype=np.int32),
            "maps": maps, "mapnames": mapnames, "node.states": np.ones((2, 2), dtype=np.int32)}


@pytest.mark.parametrize("segs", [(1, 1), (1, 3), (2, 2)])
def test_edge_case_smallest_tree_and_single_segment_branches(segs):
    """Two tips, one internal node (no nodelist), branches that start with ONE segment: updatenodestates then writes the
    single segment twice and the child wins (src/phylomap.cpp:468-472)."""
    Q = synth.config_Q(2)
    Omega = 1.5
    pid = np.array([.1, .2, .3, .4])
    z = _two_tip_tree([3, 3] if segs == (1, 1) else [2, 4], segs=segs)    # B^0 = I: single-segment branches need equal tips
    nen, nodelist, root = _orders(z)
    assert len(nodelist) == 0 and root == 3
    for fn, var in VARIANTS:
        for mapping in MAPPINGS:
            got = getattr(api, fn)(z, Q, pid, Omega, 30, seed=8, n_replicas=2, mapping=mapping)
            for r in range(2):
                want, rc = O.maketreelistMCMC(z, Q, pid, np.eye(4) + Q / Omega, Omega, nen, nodelist, root, 30, variant=var, seed=8, replica=r)
                assert rc == 0
                _same(got[r], want, 4, mapping)
    lefts, rights, d = api.eigen_decompose(Q)
    want, rc = O.maketreelistEXP(z, Q, pid, nen, nodelist, root, 100, lefts, rights, d, seed=8)
    assert rc == 0
    np.testing.assert_array_equal(api.sumstatEXP(z, Q, pid, 100, seed=8, mapping="replicas"), want)
    got = api.sumstatEXP(z, Q, pid, 100, seed=8)          # automatic: one wave per (tile, branch), dwell sums in fixed point
    np.testing.assert_array_equal(got[:, 4:], want[:, 4:])
    np.testing.assert_allclose(got[:, :4], want[:, :4], rtol=1e-10, atol=0)


def test_edge_case_zero_length_segments_and_omega_on_the_boundary():
    """A zero-length input segment stops the virtual-jump loop for the rest of that branch (list iterators are not
    advanced, src/phylomap.cpp:397,405-406); Omega == |q_ii| for one state gives rate 0 = no virtual jumps there."""
    Q = np.array([[-1.0, 1.0], [0.25, -0.25]])
    Omega = 1.0                                            # Omega + q_00 == 0
    pid = np.array([.5, .5])
    z = synth.make_tree(12, Q, Omega, 55, pid)
    z["maps"] = [np.array([m[0], 0.0, m[1]]) if i % 3 == 0 else m for i, m in enumerate(z["maps"])]
    z["mapnames"] = [np.array([n_[0], n_[0], n_[1]], dtype=np.int32) if i % 3 == 0 else n_ for i, n_ in enumerate(z["mapnames"])]
    nen, nodelist, root = _orders(z)
    for mapping in MAPPINGS:
        got = api.sumstatMCMC(z, Q, pid, Omega, 25, seed=21, n_replicas=3, mapping=mapping)
        for r in range(3):
            want, rc = O.maketreelistMCMC(z, Q, pid, np.eye(2) + Q / Omega, Omega, nen, nodelist, root, 25, seed=21, replica=r)
            assert rc == 0
            _same(got[r], want, 2, mapping)


def test_edge_case_impossible_data_raises_like_the_sampler():
    """Tip data that the jump structure cannot produce (absorbing state, one segment per branch) give an all-zero weight
    vector: RcppArmadillo::sample throws; oracle and GPU report PHM_ERR_ZERO_PROB."""
    Q = np.array([[-0.5, 0.5], [0.0, 0.0]])                # state 2 is absorbing
    Omega = 1.0
    z = _two_tip_tree([1, 2], segs=(1, 1))                 # B^0 = I cannot connect a root state to both tips
    nen, nodelist, root = _orders(z)
    _, rc = O.maketreelistMCMC(z, Q, np.array([.5, .5]), np.eye(2) + Q / Omega, Omega, nen, nodelist, root, 3, seed=1)
    assert rc & O.ERR_ZERO_PROB
    for mapping in MAPPINGS:
        with pytest.raises(_lib.PhmError) as e:
            api.sumstatMCMC(z, Q, np.array([.5, .5]), Omega, 3, seed=1, mapping=mapping)
        assert e.value.status == 5


def test_mcmc_non_cladewise_edge_order():
    """Edge rows shuffled: the engine derives its own sweeps; counts stay exact, dwell within 1e-10."""
    z, Q, pid, Omega = _problem(4, 30, 5)
    perm = np.random.default_rng(3).permutation(len(z["maps"]))
    z2 = dict(z)
    z2["edge"] = z["edge"][perm]
    z2["edge.length"] = z["edge.length"][perm]
    z2["maps"] = [z["maps"][i] for i in perm]
    z2["mapnames"] = [z["mapnames"][i] for i in perm]
    z2["node.states"] = z["node.states"][perm]
    nen, nodelist, root = _orders(z2)
    want, rc = O.maketreelistMCMC(z2, Q, pid, np.eye(4) + Q / Omega, Omega, nen, nodelist, root, 30, seed=9)
    assert rc == 0
    for mapping in MAPPINGS:
        got = api.sumstatMCMC(z2, Q, pid, Omega, 30, seed=9, mapping=mapping)
        np.testing.assert_array_equal(got[:, 4:], want[:, 4:])
        np.testing.assert_allclose(got[:, :4], want[:, :4], rtol=1e-10)


def test_mcmc_config2_shape_matches_oracle():
    """BASELINE config C2 (4 states, 1000 tips), a few iterations, first and last lane of a tile."""
    z, Q, pid, Omega = synth.config_problem(2)
    nen, nodelist, root = _orders(z)
    N, seed = 12, 2024
    eng = _lib.Engine(z, Q, pid, Omega, N, variant=_lib.PHM_MCMC_BIGTREE, seed=seed, n_replicas=128, storage=1, mapping="replicas")
    eng.run(N); eng.sync()
    st = eng.stats(0, N)
    for r in (0, 63, 127):
        want, rc = O.maketreelistMCMC(z, Q, pid, np.eye(4) + Q / Omega, Omega, nen, nodelist, root, N,
                                      variant=O.BIGTREE, seed=seed, replica=r)
        assert rc == 0
        np.testing.assert_array_equal(st[r], want)
    eng.close()


def test_plain_variant_underflows_like_the_reference():
    """sumstatMCMC does not rescale partial likelihoods (src/phylomap.cpp:510 vs :525): on a 1000-tip tree the
    root vector underflows to zero and RcppArmadillo::sample throws; oracle and GPU both report it."""
    z, Q, pid, Omega = synth.config_problem(2)
    nen, nodelist, root = _orders(z)
    _, rc = O.maketreelistMCMC(z, Q, pid, np.eye(4) + Q / Omega, Omega, nen, nodelist, root, 2, seed=1)
    assert rc & O.ERR_ZERO_PROB
    for mapping in MAPPINGS:
        with pytest.raises(_lib.PhmError) as e:
            api.sumstatMCMC(z, Q, pid, Omega, 2, seed=1, mapping=mapping)
        assert e.value.status == 5


@pytest.mark.parametrize("mapping", MAPPINGS)
def test_mcmc_per_site_tips_and_reduce(mapping):
    z, Q, pid, Omega = _problem(4, 20, 8)
    nen, nodelist, root = _orders(z)
    S, N, seed = 5, 20, 11
    rs = np.random.default_rng(0)
    sites = rs.integers(1, 5, size=(S, 20)).astype(np.int32)
    eng = _lib.Engine(z, Q, pid, Omega, N, seed=seed, n_replicas=S, tips_per_replica=True, states=sites, mapping=mapping)
    eng.run(N); eng.sync()
    per = eng.stats(0, N)
    eng.close()
    total = np.zeros((N, 16))
    for r in range(S):
        zr = dict(z); zr["states"] = sites[r]
        want, rc = O.maketreelistMCMC(zr, Q, pid, np.eye(4) + Q / Omega, Omega, nen, nodelist, root, N, seed=seed, replica=r)
        assert rc == 0
        _same(per[r], want, 4, mapping)
        total += want
    eng = _lib.Engine(z, Q, pid, Omega, N, seed=seed, n_replicas=S, tips_per_replica=True, states=sites, reduce=True, mapping=mapping)
    eng.run(N); eng.sync()
    red = eng.stats(0, N)
    eng.close()
    np.testing.assert_array_equal(red[:, 4:], total[:, 4:])
    np.testing.assert_allclose(red[:, :4], total[:, :4], rtol=1e-12)


def test_wide_kernel_per_site_tips_on_thinly_filled_tiles():
    """n > 4 with few replicas: the engine places fewer than 64 replicas on a tile (the wave takes them in turn) -- per-site
    tip vectors, per-replica statistics and the reduced sum must still belong to the right replica."""
    n = 6
    Q = synth.dense_Q(n, 0.02, 0.08, seed=60)
    Omega = 1.25 * float(np.max(np.abs(np.diag(Q))))
    pid = np.full(n, 1.0 / n)
    z = synth.make_tree(14, Q, Omega, 61, pid, init_segments=n)
    nen, nodelist, root = _orders(z)
    S, N, seed = 5, 12, 19
    sites = np.random.default_rng(1).integers(1, n + 1, size=(S, 14)).astype(np.int32)
    eng = _lib.Engine(z, Q, pid, Omega, N, variant=_lib.PHM_MCMC_BIGTREE, seed=seed, n_replicas=S, tips_per_replica=True, states=sites,
                      mapping="replicas")
    assert eng.info().n_replicas_padded == 64 * S                         # one replica per tile
    eng.run(N); eng.sync()
    per = eng.stats(0, N)
    total = np.zeros_like(per[0])
    for r in range(S):
        zr = dict(z); zr["states"] = sites[r]
        want, rc, dump = O.maketreelistMCMC(zr, Q, pid, np.eye(n) + Q / Omega, Omega, nen, nodelist, root, N, variant=O.BIGTREE,
                                            seed=seed, replica=r, dump=True)
        assert rc == 0
        np.testing.assert_array_equal(per[r], want)
        np.testing.assert_array_equal(eng.dump(r)["node_states"], dump.node_states)
        total += want
    eng.close()
    for mapping in ("replicas", "branches", "tiles"):      # branches: one wave per (replica, branch) with per-replica tip vectors; tiles: lanes = replicas
        eng = _lib.Engine(z, Q, pid, Omega, N, variant=_lib.PHM_MCMC_BIGTREE, seed=seed, n_replicas=S, tips_per_replica=True, states=sites,
                          reduce=True, mapping=mapping)
        eng.run(N); eng.sync()
        red = eng.stats(0, N)
        eng.close()
        np.testing.assert_array_equal(red[:, n:], total[:, n:])
        np.testing.assert_allclose(red[:, :n], total[:, :n], rtol=1e-12)


@pytest.mark.parametrize("mapping", ["replicas", "branches", "tiles"])     # wave per 64-replica tile, replicas in turn / wave per (replica, branch) / lane per replica, wave per (tile, item)
@pytest.mark.parametrize("n,fn,variant", [(20, "sumstatMCMC", O.PLAIN), (20, "SPARSEsumstatMCMC", O.SPARSE),
                                          (20, "sumstatMCMC_bigtree", O.BIGTREE), (5, "sumstatMCMC", O.PLAIN),
                                          (40, "sumstatMCMC_bigtree", O.BIGTREE),      # three 16-state row blocks (MT = 3 kernels)
                                          (61, "sumstatMCMC_bigtree", O.BIGTREE), (64, "sumstatMCMC", O.PLAIN)])
def test_wide_kernel_matches_oracle(n, fn, variant, mapping):
    """5..64 states (C4: dense 61-state Q; C5: sparse 20-state tridiagonal Q): states-over-lanes kernel."""
    if n == 20:
        Q = synth.config_Q(5)
    elif n == 61:
        Q = synth.config_Q(4)
    else:
        Q = synth.dense_Q(n, 0.02, 0.08, seed=n)
    Omega = 1.25 * float(np.max(np.abs(np.diag(Q))))
    pid = np.full(n, 1.0 / n)
    tips = 14 if n < 61 else 9
    z = synth.make_tree(tips, Q, Omega, 400 + n, pid, init_segments=(n if n == 20 else 2))
    nen, nodelist, root = _orders(z)
    N, S, seed = 10, 3, 31
    got = getattr(api, fn)(z, Q, pid, Omega, N, seed=seed, n_replicas=S, mapping=mapping)
    for r in range(S):
        want, rc = O.maketreelistMCMC(z, Q, pid, np.eye(n) + Q / Omega, Omega, nen, nodelist, root, N, variant=variant,
                                      seed=seed, replica=r)
        assert rc == 0
        _same(got[r], want, n, mapping)
    red = getattr(api, fn)(z, Q, pid, Omega, N, seed=seed, n_replicas=S, reduce=True, mapping=mapping)
    np.testing.assert_array_equal(red[:, n:], got.sum(0)[:, n:])
    np.testing.assert_allclose(red[:, :n], got.sum(0)[:, :n], rtol=1e-12)


@pytest.mark.parametrize("n,form", [(n, f) for n in (12, 24, 40, 61) for f in (1, 2)] + [(17, 3), (20, 3), (24, 3), (32, 3), (33, 3), (40, 3), (48, 3), (61, 3)])
def test_wide_tiles_pruning_kernels_agree(n, form):
    """The pruning pass of the lane-per-replica mapping has three kernels chosen by tile count (a wave per (node, tile) with sorted
    blocks; for few tiles a workgroup per 16-replica block with a wave per row block, or a wave per block when n <= 16): each
    forced (pruning_form) on a problem the automatic choice gives to another, 1 / 2 / 3 / 4 row blocks, against the oracle.  Form 3
    (33 .. 64 states): the matrix fragments in LDS and two waves per SIMD, the first child's vectors parked in the parent's row."""
    Q = synth.dense_Q(n, 0.02, 0.08, seed=n)
    Omega = 1.25 * float(np.max(np.abs(np.diag(Q))))
    pid = np.full(n, 1.0 / n)
    z = synth.make_tree(11, Q, Omega, 700 + n, pid, init_segments=3)
    nen, nodelist, root = _orders(z)
    N, S, seed = 6, 150, 77
    eng = _lib.Engine(z, Q, pid, Omega, N, variant=_lib.PHM_MCMC_BIGTREE, seed=seed, n_replicas=S, mapping="tiles", pruning_form=form)
    eng.run(N); eng.sync()
    got = eng.stats(0, N)
    eng.close()
    for r in (0, 15, 16, 63, 64, 129, S - 1):
        want, rc = O.maketreelistMCMC(z, Q, pid, np.eye(n) + Q / Omega, Omega, nen, nodelist, root, N, variant=O.BIGTREE, seed=seed, replica=r)
        assert rc == 0
        _same(got[r], want, n, "tiles")


@pytest.mark.parametrize("n,band", [(32, 1), (24, 2), (32, 31)])
def test_wide_tiles_branch_kernel_lds_budget_at_the_top_of_the_small_range(n, band):
    """n <= 32: an eight-wave workgroup of the branch kernel keeps the rows of B, the dwell table and (for a banded B with at most 96
    possible transitions) the transition counters in LDS -- up to 86 KB at 32 states, beyond the default 64 KB per workgroup;
    a dense B (992 possible transitions) counts in global memory instead."""
    rs = np.random.default_rng(n * 10 + band)
    Q = np.zeros((n, n))
    for i in range(n):
        for j in range(max(0, i - band), min(n, i + band + 1)):
            if i != j:
                Q[i, j] = rs.uniform(0.02, 0.08)
    np.fill_diagonal(Q, -Q.sum(1))
    Omega = 1.25 * float(np.max(np.abs(np.diag(Q))))
    pid = np.full(n, 1.0 / n)
    z = synth.make_tree(12, Q, Omega, 800 + n + band, pid, init_segments=(n if band < n - 1 else 2))      # a banded B reaches state j from i in |i - j| / band steps
    nen, nodelist, root = _orders(z)
    N, S, seed = 6, 70, 55
    got = api.sumstatMCMC_bigtree(z, Q, pid, Omega, N, seed=seed, n_replicas=S, mapping="tiles")
    for r in (0, 63, 64, S - 1):
        want, rc = O.maketreelistMCMC(z, Q, pid, np.eye(n) + Q / Omega, Omega, nen, nodelist, root, N, variant=O.BIGTREE, seed=seed, replica=r)
        assert rc == 0
        _same(got[r], want, n, "tiles")


def test_wide_tiles_branch_kernel_with_B_rows_in_lds_beyond_32_states():
    """Beyond 32 states the branch kernel of the lane-per-replica mapping stages the rows of B in LDS once a wave walks four or more
    branches (tiles x branches >= 4 x 65 536): 40 states on a 3 000-tip tree with 45 tiles, replicas of three tiles against the oracle."""
    n = 40
    Q = synth.dense_Q(n, 0.01, 0.04, seed=404)
    Omega = 1.25 * float(np.max(np.abs(np.diag(Q))))
    pid = np.full(n, 1.0 / n)
    z = synth.make_tree(3000, Q, Omega, 4040, pid, init_segments=2)
    nen, nodelist, root = _orders(z)
    N, S, seed = 4, 2880, 909
    eng = _lib.Engine(z, Q, pid, Omega, N, variant=_lib.PHM_MCMC_BIGTREE, seed=seed, n_replicas=S, mapping="tiles")
    eng.run(N); eng.sync()
    got = eng.stats(0, N)
    eng.close()
    for r in (0, 64 + 17, S - 1):
        want, rc = O.maketreelistMCMC(z, Q, pid, np.eye(n) + Q / Omega, Omega, nen, nodelist, root, N, variant=O.BIGTREE, seed=seed, replica=r)
        assert rc == 0
        _same(got[r], want, n, "tiles")
    np.testing.assert_allclose(got[:, :, :n].sum(2), z["edge.length"].sum(), rtol=1e-11)


@pytest.mark.parametrize("n,band", [(9, 1), (33, 1), (64, 1), (16, 2), (48, 7)])
@pytest.mark.parametrize("fn,variant", [("sumstatMCMC", O.PLAIN), ("SPARSEsumstatMCMC", O.SPARSE)])
def test_wide_branch_mapping_sparse_rows_match_oracle(n, band, fn, variant):
    """Banded rate matrices: the (replica, branch) kernels apply the chain matrix in ELLPACK form and draw the forward states
    over the non-zero slots of a row (phm_wbranch.hip) -- the exact zeros they skip must not move a single count."""
    rs = np.random.default_rng(1000 * n + band)
    Q = np.zeros((n, n))
    for i in range(n):
        for j in range(max(0, i - band), min(n, i + band + 1)):
            if i != j:
                Q[i, j] = rs.uniform(0.01, 0.05)
    if band == 1:
        Q[3, 4] = 0.0                                   # a row with fewer non-zeros than its neighbours (padding slots)
    np.fill_diagonal(Q, -Q.sum(1))
    Omega = 1.25 * float(np.max(np.abs(np.diag(Q))))
    pid = rs.dirichlet(np.ones(n))
    z = synth.make_tree(12, Q, Omega, 77 + n, pid, init_segments=7)
    nen, nodelist, root = _orders(z)
    N, S, seed = 8, 3, 5
    got = getattr(api, fn)(z, Q, pid, Omega, N, seed=seed, n_replicas=S, mapping="branches")
    for r in range(S):
        want, rc = O.maketreelistMCMC(z, Q, pid, np.eye(n) + Q / Omega, Omega, nen, nodelist, root, N, variant=variant,
                                      seed=seed, replica=r)
        assert rc == 0
        _same(got[r], want, n, "branches")


def test_wide_kernel_chain_state_and_golden():
    from golden.make_golden import unpack_tree
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "n20_t12.npz"))
    z = unpack_tree(g)
    Q, pid, Omega, seed = g["Q"], g["pid"], float(g["Omega"]), int(g["seed"])
    got = api.SPARSEsumstatMCMC(z, Q, pid, Omega, 24, seed=seed, n_replicas=6, mapping="replicas")
    for mp in ("branches", "tiles"):
        gotb = api.SPARSEsumstatMCMC(z, Q, pid, Omega, 24, seed=seed, n_replicas=6, mapping=mp)
        _same(gotb[0], g["sparse_r0"], 20, mp)
        _same(gotb[5], g["sparse_r5"], 20, mp)
    np.testing.assert_array_equal(got[0], g["sparse_r0"])
    np.testing.assert_array_equal(got[5], g["sparse_r5"])
    want, rc, dump = O.maketreelistMCMC(z, Q, pid, np.eye(20) + Q / Omega, Omega, g["nen"], g["nodelist"], int(g["root"]), 9,
                                        variant=O.BIGTREE, seed=seed, replica=1, dump=True)
    for mapping in ("replicas", "branches", "tiles"):
        eng = _lib.Engine(z, Q, pid, Omega, 9, variant=_lib.PHM_MCMC_BIGTREE, seed=seed, n_replicas=2, mapping=mapping)
        eng.run(9); eng.sync()
        d = eng.dump(1)
        eng.close()
        np.testing.assert_array_equal(d["seg_count"], dump.seg_count)
        np.testing.assert_array_equal(d["node_states"], dump.node_states)
        np.testing.assert_array_equal(d["PL"], dump.PL)
        for b in range(len(dump.seg_count)):
            np.testing.assert_array_equal(d["seg_dwell"][b, :dump.seg_count[b]], dump.seg_dwell[b, :dump.seg_count[b]])


@pytest.mark.parametrize("n,mapping", [(2, "replicas"), (4, "replicas"), (2, "branches"), (4, "branches"), (2, "tiles"), (4, "tiles"), (6, "replicas"), (6, "branches"), (6, "tiles"), (8, "tiles")])
def test_ks_sweep_matches_oracle(n, mapping):
    """Tree sweep of sumstatMCMCks with Q fixed (hidden-rates Q = make2sQ, binary trait observed): n<=4 kernel and,
    for k=2 (n=6), the wide kernel."""
    Q = {2: synth.config_Q(1), 4: synth.make2sQ(.1, .1, .2, .2, 10), 6: synth.make2sQ(.1, .3, [.2, .4], [.5, .6], [2, 3]),
         8: synth.make2sQ(.1, .3, [.2, .4, .3], [.5, .6, .2], [2, 3, 1.5])}[n]
    Omega = 1.25 * float(np.max(np.abs(np.diag(Q))))
    pid = np.full(n, 1.0 / n)
    z = synth.make_tree(20, Q, Omega, 900 + n, pid)
    z["states"] = ((z["states"] - 1) % 2 + 1).astype(np.int32)          # only the binary trait is observed
    for b, (p_, c_) in enumerate(z["edge"]):
        if c_ <= 20:
            z["mapnames"][b][-1] = z["states"][c_ - 1]
    nen, nodelist, root = _orders(z)
    N, S, seed = 15, 3, 77
    got = api.sumstatMCMCks_sweep(z, Q, pid, Omega, N, seed=seed, n_replicas=S, mapping=mapping)
    k = n // 2 - 1
    assert got.shape == (S, N, n + n * n + 2 + 3 * k + 1)
    for r in range(S):
        want, rc = O.maketreelistMCMC(z, Q, pid, np.eye(n) + Q / Omega, Omega, nen, nodelist, root, N, variant=O.KS,
                                      seed=seed, replica=r)
        assert rc == 0
        _same(got[r], want, n, mapping, ks=True)
    red = api.sumstatMCMCks_sweep(z, Q, pid, Omega, N, seed=seed, n_replicas=S, reduce=True, mapping=mapping)
    np.testing.assert_array_equal(red[:, n:n + n * n], got.sum(0)[:, n:n + n * n])
    np.testing.assert_array_equal(red[:, n + n * n:-1], got[0][:, n + n * n:-1])     # parameter columns: plain values
    np.testing.assert_array_equal(red[:, -1], got.sum(0)[:, -1])


@pytest.mark.parametrize("storage", [1, 2, 0])        # 0: the branch mapping (what a default single-chain call gets)
def test_sumstatMCMCbf_with_rate_updates_matches_oracle(storage):
    """R/sumstatMCMCbf.R: sweep on the GPU, Gibbs update of (l01, l10) on the host, every iteration."""
    Q = np.array([[-.1, .1], [.1, -.1]])
    Omega, pid, prior = 10.0, np.array([.5, .5]), [.55, 1, .56, 1.01]        # vignettes/phylomap_tutorial.Rnw:204-212
    z = synth.make_tree(40, Q, 0.5, 15, pid)
    nen, nodelist, root = _orders(z)
    Q0 = Q.copy()
    mapping = "replicas" if storage else "branches"
    got = api.sumstatMCMCbf(z, Q, pid, Omega, 30, prior, seed=99, storage=storage, mapping=mapping)
    assert np.array_equal(Q, Q0)                                             # inputs are never written (:1212-1217 does)
    want, rc = O.maketreelistMCMC(z, Q, pid, np.eye(2) + Q / Omega, Omega, nen, nodelist, root, 30, variant=O.BF, seed=99, prior=prior)
    assert rc == 0 and got.shape == (30, 9)
    _same(got, want, 2, mapping, ks=True)
    assert len(np.unique(got[:, 6])) > 20 and np.all(got[:, 6] > 0)          # l01 really moves


@pytest.mark.parametrize("n", [4, 6])
def test_sumstatMCMCks_with_rate_updates_matches_oracle(n):
    """R/sumstatMCMCks.R: hidden-rates model, 2+3k parameters updated after every sweep (n = 6 runs the wide kernel)."""
    Q = synth.make2sQ(.1, .1, .2, .2, 10) if n == 4 else synth.make2sQ(.1, .3, [.2, .4], [.5, .6], [2, 3])
    Omega, pid = 25.0, np.full(n, 1.0 / n)
    prior = [1, 10, 2, 10, 20, 2]                                            # vignettes/phylomap_tutorial.Rnw:248-256
    # n = 4: ~100 segments per branch (general path of the lane kernel); n = 6: the wide kernel holds <= 128 per branch
    z = synth.make_tree(30, Q, 1.0 if n == 4 else Omega / 3, 16, pid)
    z["states"] = ((z["states"] - 1) % 2 + 1).astype(np.int32)
    nen, nodelist, root = _orders(z)
    want, rc = O.maketreelistMCMC(z, Q, pid, np.eye(n) + Q / Omega, Omega, nen, nodelist, root, 25, variant=O.KS, seed=7, prior=prior)
    assert rc == 0
    for mapping in (MAPPINGS if n == 4 else ["replicas"]):                   # n = 4, one chain: "branches" is the default
        got = api.sumstatMCMCks(z, Q, pid, Omega, 25, prior, seed=7, mapping=mapping)
        _same(got, want, n, mapping, ks=True)
    k = n // 2 - 1
    assert len(np.unique(got[:, n + n * n])) > 10                            # l01 moves
    # multi-site: S sites share Q; updates see the summed statistics
    got2 = api.sumstatMCMCks(z, Q, pid, Omega, 10, prior, seed=7, n_replicas=3)
    np.testing.assert_allclose(got2[:, :n].sum(1), 3 * z["edge.length"].sum(), rtol=1e-12)


@pytest.mark.parametrize("which", ["2s", "ks"])
def test_dic_drivers_match_oracle(which):
    """sumstatMCMC2sDICt / sumstatMCMCksDICt: the bf / ks drivers plus log p(y|Q) by matrix exponentiation every iteration
    (Pade expm of every branch on the device, pruning with scale factors)."""
    from scipy.linalg import expm
    if which == "2s":
        Q, prior, var, fn = np.array([[-.1, .1], [.1, -.1]]), [.55, 1, .56, 1.01], O.BF, api.sumstatMCMC2sDICt
    else:
        Q, prior, var, fn = synth.make2sQ(.1, .1, .2, .2, 10), [1, 10, 2, 10, 20, 2], O.KS, api.sumstatMCMCksDICt
    n = Q.shape[0]
    Omega, pid = 25.0, np.full(n, 1.0 / n)
    z = synth.make_tree(25, Q, Omega / 3, 44, pid)
    z["states"] = ((z["states"] - 1) % 2 + 1).astype(np.int32)
    nen, nodelist, root = _orders(z)
    want, rc = O.maketreelistMCMC(z, Q, pid, np.eye(n) + Q / Omega, Omega, nen, nodelist, root, 20, variant=var, seed=5, prior=prior, dic=True)
    assert rc == 0
    for mapping in MAPPINGS:
        got = fn(z, Q, pid, Omega, 20, prior, seed=5, mapping=mapping)
        _same(got, want, n, mapping, ks=True)
    # the first log-likelihood against an independent Felsenstein pass with scipy's expm
    T, E = 25, z["edge"]
    PL = np.zeros((2 * T - 1, n))
    for i, s_ in enumerate(z["states"]):
        if which == "2s":
            PL[i, s_ - 1] = 1
        else:
            PL[i, (0 if s_ % 2 == 1 else 1)::2] = 1
    for i in range(T - 1):
        ea, eb = nen[2 * i] - 1, nen[2 * i + 1] - 1
        PL[E[ea, 0] - 1] = (expm(Q * z["edge.length"][ea]) @ PL[E[ea, 1] - 1]) * (expm(Q * z["edge.length"][eb]) @ PL[E[eb, 1] - 1])
    np.testing.assert_allclose(got[0, -1], np.log(PL[root - 1] @ pid), rtol=1e-11)


def _list_orders(treelist):
    orders = [_orders(z) for z in treelist]
    return np.stack([o[0] for o in orders]), np.stack([o[1] for o in orders]), [o[2] for o in orders]


def test_sumstatMCMCmt_list_of_trees_matches_oracle():
    """R/sumstatMCMCmt.R: every iteration sweeps EVERY tree of the list (here the one engine over the list: one launch, tree j on
    replica tile j), keeps a uniformly drawn one, and updates (l01, l10) from it with the acceptance-testing mt updates."""
    Q = np.array([[-.1, .1], [.1, -.1]])
    Omega, pid, prior = 2.0, np.array([.5, .5]), [.55, 1, .56, 1.01]
    trees = synth.make_treelist(7, 16, Q, 0.5, 314, pid)
    nen_m, nodelist_m, roots = _list_orders(trees)
    N = 40
    got = api.sumstatMCMCmt(trees, Q, pid, Omega, N, prior, seed=2718, mapping="replicas")      # the list engine: bit-identical dwell sums
    want, rc = O.maketreelistMCMCmt(trees, Q, pid, np.eye(2) + Q / Omega, Omega, nen_m, nodelist_m, roots, N, prior, variant=O.MT, seed=2718)
    assert rc == 0 and got.shape == (N, 9)
    np.testing.assert_array_equal(got, want)
    assert len(np.unique(got[:, 8])) >= 5 and got[:, 8].max() <= 6          # tree_number, 0-based (:2350)
    assert len(np.unique(got[:, 6])) > 10                                     # l01 moves (and some proposals are rejected)
    assert np.any(got[1:, 6] == got[:-1, 6])
    # one tree in the list: the single-tree sweep without normalisation; total time = that tree's length
    one = api.sumstatMCMCmt(trees[:1], Q, pid, Omega, 5, prior, seed=3)                         # automatic: an engine per tree
    np.testing.assert_allclose(one[:, :2].sum(1), trees[0]["edge.length"].sum(), rtol=1e-12)
    assert np.all(one[:, 8] == 0)


@pytest.mark.parametrize("n", [4, 6])
def test_sumstatMCMCksmt_list_of_trees_matches_oracle(n):
    """R/sumstatMCMCksmt.R: hidden-rates model over a list of trees (n = 6 runs the wide kernel); 8 prior entries."""
    Q = synth.make2sQ(.1, .1, .2, .2, 10) if n == 4 else synth.make2sQ(.1, .3, [.2, .4], [.5, .6], [2, 3])
    Omega, pid = 25.0, np.full(n, 1.0 / n)
    prior = [1, 10, 1.5, 11, 2, 10, 20, 2]
    trees = synth.make_treelist(5, 14, Q, Omega / 3, 2024, pid)
    for z in trees:
        z["states"] = ((z["states"] - 1) % 2 + 1).astype(np.int32)
    nen_m, nodelist_m, roots = _list_orders(trees)
    N = 20
    got = api.sumstatMCMCksmt(trees, Q, pid, Omega, N, prior, seed=11, mapping="replicas")      # the list engine: bit-identical dwell sums
    want, rc = O.maketreelistMCMCmt(trees, Q, pid, np.eye(n) + Q / Omega, Omega, nen_m, nodelist_m, roots, N, prior, variant=O.KSMT, seed=11)
    k = n // 2 - 1
    assert rc == 0 and got.shape == (N, n + n * n + 2 + 3 * k + 1)
    np.testing.assert_array_equal(got, want)
    assert len(np.unique(got[:, -1])) >= 3


@pytest.mark.parametrize("variant,n,tips,mapping", [("mt", 2, 16, "branches"), ("ksmt", 4, 14, "branches"), ("ksmt", 6, 14, "branches"),
                                                    ("mt", 2, 300, "auto"), ("mt", 2, 300, "replicas")])
def test_multi_tree_drivers_with_an_engine_per_tree(variant, n, tips, mapping):
    """Lists of BIG trees: the one engine over the list walks a whole tree in one lane (seconds per sweep on the reference's squamate
    tree), so the multi-tree drivers then run an engine per tree in the branch mapping -- tree j's chain on replica 64 j either way,
    the same draws.  Forced on the small lists of the tests above, chosen automatically on 300-tip trees, against the oracle: counts,
    tree numbers and root states exact, dwell sums and what the rate updates make of them to rounding."""
    if variant == "mt":
        Q = np.array([[-.1, .1], [.1, -.1]]); Omega, pid, prior = 2.0, np.array([.5, .5]), [.55, 1, .56, 1.01]
        trees = synth.make_treelist(4 if tips > 100 else 7, tips, Q, 0.5, 314, pid)
        fn, var = api.sumstatMCMCmt, O.MT
    else:
        Q = synth.make2sQ(.1, .1, .2, .2, 10) if n == 4 else synth.make2sQ(.1, .3, [.2, .4], [.5, .6], [2, 3])
        Omega, pid, prior = 25.0, np.full(n, 1.0 / n), [1, 10, 1.5, 11, 2, 10, 20, 2]
        trees = synth.make_treelist(5, tips, Q, Omega / 3, 2024, pid)
        for z in trees:
            z["states"] = ((z["states"] - 1) % 2 + 1).astype(np.int32)
        fn, var = api.sumstatMCMCksmt, O.KSMT
    nen_m, nodelist_m, roots = _list_orders(trees)
    N = 15
    got = fn(trees, Q, pid, Omega, N, prior, seed=77, mapping=mapping)
    want, rc = O.maketreelistMCMCmt(trees, Q, pid, np.eye(n) + Q / Omega, Omega, nen_m, nodelist_m, roots, N, prior, variant=var, seed=77)
    assert rc == 0 and got.shape == want.shape
    np.testing.assert_array_equal(got[:, n:n + n * n], want[:, n:n + n * n])
    np.testing.assert_array_equal(got[:, -1], want[:, -1])
    np.testing.assert_allclose(got, want, rtol=1e-9, atol=0)


def test_engine_over_a_list_of_trees_runs_each_tree_like_its_own_engine():
    """phm_engine_create_multi: chains of tree j on their own tiles, Philox replica word 64*tiles_per_tree*j + c; every
    (tree, chain) pair reproduces the oracle's single-tree run, and the chain state dump follows the right topology."""
    z0, Q, pid, Omega = _problem(3, 20, 99)
    trees = synth.make_treelist(4, 20, Q, Omega, 99, pid)
    B = np.eye(3) + Q / Omega
    N, S = 12, 3
    eng = _lib.Engine(trees, Q, pid, Omega, N, variant=_lib.PHM_MCMC_BIGTREE, seed=5, n_replicas=S)
    eng.run(N)
    eng.sync()
    got = eng.stats(0, N)
    assert got.shape == (4 * S, N, 9)
    for j, z in enumerate(trees):
        nen, nodelist, root = _orders(z)
        for c in range(S):
            want, rc, dump = O.maketreelistMCMC(z, Q, pid, B, Omega, nen, nodelist, root, N, variant=O.BIGTREE, seed=5,
                                                replica=64 * j + c, dump=True)
            assert rc == 0
            np.testing.assert_array_equal(got[j * S + c][:, 3:], want[:, 3:])
            np.testing.assert_allclose(got[j * S + c][:, :3], want[:, :3], rtol=1e-10)
            d = eng.dump(j * S + c)
            np.testing.assert_array_equal(d["seg_count"], dump.seg_count)
            np.testing.assert_array_equal(d["node_states"], dump.node_states)
    eng.close()
    with pytest.raises(_lib.PhmError):                                        # lists of unequal trees are refused
        _lib.Engine([trees[0], synth.make_tree(21, Q, Omega, 1, pid)], Q, pid, Omega, 2)


@pytest.mark.parametrize("mapping", MAPPINGS)
def test_replica_offset_shards_like_one_device(mapping):
    """Multi-GPU sharding is by global replica id: ranks that take replicas [2, 4) reproduce those of a single device."""
    z, Q, pid, Omega = _problem(2, 18, 4)
    a = api.sumstatMCMC(z, Q, pid, Omega, 10, seed=3, n_replicas=4, mapping=mapping)
    b = api.sumstatMCMC(z, Q, pid, Omega, 10, seed=3, n_replicas=2, replica_offset=2, mapping=mapping)
    np.testing.assert_array_equal(a[2:], b)


EXP_MAPPINGS = ["replicas", "tiles"]      # one wave per tile of 64 samples walks the tree / one wave per (tile, branch)


@pytest.mark.parametrize("mapping", EXP_MAPPINGS)
@pytest.mark.parametrize("n", [2, 4])
def test_exp_matches_oracle(n, mapping):
    z, Q, pid, Omega = _problem(n, 30, 321 + n)
    nen, nodelist, root = _orders(z)
    lefts, rights, d = api.eigen_decompose(Q)
    N, seed = 200, 17
    got = api.sumstatEXP(z, Q, pid, N, seed=seed, mapping=mapping)
    want, rc = O.maketreelistEXP(z, Q, pid, nen, nodelist, root, N, lefts, rights, d, seed=seed)
    assert rc == 0
    np.testing.assert_array_equal(got[:, n:], want[:, n:])
    np.testing.assert_allclose(got[:, :n], want[:, :n], rtol=1e-10)
    if mapping == "replicas":                 # the walking wave adds the dwell times in the reference's order
        np.testing.assert_array_equal(got, want)
    np.testing.assert_array_equal(api.sumstatEXP(z, Q, pid, N, seed=seed), got if mapping == "tiles" else api.sumstatEXP(z, Q, pid, N, seed=seed, mapping="tiles"))


@pytest.mark.parametrize("mapping", EXP_MAPPINGS)
@pytest.mark.parametrize("n", [5, 20, 61])
def test_exp_wide_matches_oracle(n, mapping):
    """sumstatEXP for 5..64 states (the tutorial's 20-state tridiagonal Q, vignettes/phylomap_tutorial.Rnw:72-134)."""
    if n == 20:
        Q = synth.tridiagonal_Q(20, 0.03)
    else:
        Q = synth.dense_Q(n, 0.005, 0.03, seed=n)
        Q = (Q + Q.T) / 2
        np.fill_diagonal(Q, 0.0)
        np.fill_diagonal(Q, -Q.sum(1))
    pid = np.full(n, 1.0 / n)
    z = synth.make_tree(12 if n == 61 else 30, Q, 1.25 * float(np.max(np.abs(np.diag(Q)))), 70 + n, pid)
    nen, nodelist, root = _orders(z)
    lefts, rights, d = api.eigen_decompose(Q)
    N = 150
    got = api.sumstatEXP(z, Q, pid, N, seed=23, mapping=mapping)
    want, rc = O.maketreelistEXP(z, Q, pid, nen, nodelist, root, N, lefts, rights, d, seed=23)
    assert rc == 0
    np.testing.assert_array_equal(got[:, n:], want[:, n:])
    if mapping == "replicas":
        np.testing.assert_array_equal(got[:, :n], want[:, :n])
    else:
        np.testing.assert_allclose(got[:, :n], want[:, :n], rtol=1e-10, atol=0)
    np.testing.assert_allclose(got[:, :n].sum(1), z["edge.length"].sum(), rtol=1e-12)


def test_tutorial_three_way_agreement_on_gpu():
    """The reference's own validation (vignettes/phylomap_tutorial.Rnw:72-134): 20-state tridiagonal Q (rate .003), 50 tips,
    Omega = 0.2; total jump counts from sumstatEXP, sumstatMCMC and SPARSEsumstatMCMC must agree in distribution."""
    Q = synth.tridiagonal_Q(20, 0.003)
    Omega = 0.2
    pid = np.full(20, 1 / 20)
    z = synth.make_tree(50, Q, 4.0 / 30.0, 321, pid, init_segments=20)      # mean branch length 30: a few jumps per tree
    ex = api.sumstatEXP(z, Q, pid, 20000, seed=1)
    je = ex[:, 20:].sum(1)
    means = {"EXP": je.mean()}
    for name, fn in (("MCMC", api.sumstatMCMC), ("SPARSE", api.SPARSEsumstatMCMC)):
        out = fn(z, Q, pid, Omega, 360, seed=2, n_replicas=64)                # 64 chains, first 120 sweeps dropped
        jm = out[:, 120:, 20:].sum(2)
        means[name] = jm.mean()
        chain_means = jm.mean(1)
        se = np.hypot(chain_means.std() / np.sqrt(64), je.std() / np.sqrt(je.size))
        assert abs(means[name] - means["EXP"]) < 6 * se, means
        np.testing.assert_allclose(out[:, 120:, :20].mean((0, 1)), ex[:, :20].mean(0), rtol=0.15, atol=1.0)
    assert means["EXP"] > 1.0


@pytest.mark.parametrize("n", [2, 4, 20, 61])
def test_expm_routes(n):
    from scipy.linalg import expm
    Q = {2: synth.config_Q(1), 4: synth.config_Q(2), 20: synth.config_Q(5), 61: synth.config_Q(4)}[n]
    if n == 61:    # matexp handles a real spectrum only (R/sumstatEXP.R:26-29): use a reversible Q there
        Q = (Q + Q.T) / 2
        np.fill_diagonal(Q, 0.0)
        np.fill_diagonal(Q, -Q.sum(1))
    t = np.concatenate([[0.0, 1e-6], np.random.default_rng(n).exponential(3.0, 37)])
    P2, _ = api.expm_pade(Q, t)
    for b in range(t.size):
        want, rc = O.expmat_pade(Q * t[b])
        assert rc == 0
        np.testing.assert_array_equal(P2[b], want)
        np.testing.assert_allclose(P2[b], expm(Q * t[b]), atol=1e-12)
    lefts, rights, d = api.eigen_decompose(Q)
    P1, _ = api.expm_eigen(lefts, rights, d, t)
    for b in range(t.size):
        np.testing.assert_array_equal(P1[b], O.matexp(lefts, rights, np.diag(d), t[b]))
        np.testing.assert_allclose(P1[b], expm(Q * t[b]), atol=1e-10)
    np.testing.assert_allclose(P1, P2, atol=1e-10)


@pytest.mark.parametrize("n", [20, 61, 64])
def test_expm_eigen_on_matrix_cores(n):
    """K1 as an MFMA f64 batched GEMM: agrees with the exact kernel to rounding, with scipy to 1e-10, and is
    bit-for-bit a k-ordered fma chain (oracle model orc_matexp_fma)."""
    from scipy.linalg import expm
    Q = synth.config_Q(5) if n == 20 else synth.dense_Q(n, 0.005, 0.015, seed=n)
    Q = (Q + Q.T) / 2
    np.fill_diagonal(Q, 0.0)
    np.fill_diagonal(Q, -Q.sum(1))
    lefts, rights, d = api.eigen_decompose(Q)
    t = np.concatenate([[0.0, 1e-6], np.random.default_rng(n).exponential(3.0, 70)])
    P_exact, _ = api.expm_eigen(lefts, rights, d, t)
    P_mfma, ms = api.expm_eigen(lefts, rights, d, t, mfma=True)
    np.testing.assert_allclose(P_mfma, P_exact, rtol=0, atol=1e-13)
    for b in (0, 1, 5, 71):
        np.testing.assert_allclose(P_mfma[b], expm(Q * t[b]), atol=1e-10)
        model = O.matexp_fma(lefts, rights, np.diag(d), t[b])
        assert np.array_equal(P_mfma[b], model), np.abs(P_mfma[b] - model).max()


@pytest.mark.parametrize("n", [20, 61, 64])
def test_expm_pade_on_matrix_cores(n):
    from scipy.linalg import expm
    Q = synth.config_Q(5) if n == 20 else synth.dense_Q(n, 0.005, 0.015, seed=n)
    t = np.concatenate([[0.0, 1e-6, 200.0], np.random.default_rng(n).exponential(3.0, 40)])
    P_exact, _ = api.expm_pade(Q, t)
    P_mfma, _ = api.expm_pade(Q, t, mfma=True)
    np.testing.assert_allclose(P_mfma, P_exact, rtol=0, atol=2e-13)
    for b in (0, 1, 2, 7, 42):
        np.testing.assert_allclose(P_mfma[b], expm(Q * t[b]), atol=1e-12)
        np.testing.assert_allclose(P_mfma[b].sum(1), 1.0, atol=1e-12)


def test_expm_pade_on_matrix_cores_hands_small_pivots_to_the_pivoted_kernel():
    """A matrix whose block elimination meets a pivot below the threshold is flagged by the kernel and recomputed by the pivoted
    one: with the threshold raised above every pivot the call returns the exact kernel's matrices bit for bit."""
    Q = synth.dense_Q(61, 0.005, 0.015, seed=61)
    t = np.concatenate([[0.0, 200.0], np.random.default_rng(3).exponential(3.0, 30)])
    P_exact, _ = api.expm_pade(Q, t)
    try:
        _lib.set_debug_options(pade_pivot_min=1e300)
        P_fallback, _ = api.expm_pade(Q, t, mfma=True)
        np.testing.assert_array_equal(P_fallback, P_exact)
        _lib.set_debug_options(pade_pivot_min=1.2)      # some matrices (the ones scaled most) go each way
        P_mixed, _ = api.expm_pade(Q, t, mfma=True)
    finally:
        _lib.set_debug_options()
    np.testing.assert_allclose(P_mixed, P_exact, rtol=0, atol=2e-13)


@pytest.mark.parametrize("mapping", MAPPINGS)
def test_full_size_invariants(mapping):
    """BASELINE sizes, size-independent properties: dwell row sums = tree length; counts are integers;
    a second engine with the same seed reproduces the first bit for bit (both mappings reduce in a fixed order)."""
    z, Q, pid, Omega = synth.config_problem(2)
    N = 30
    outs = []
    for _ in range(2):
        eng = _lib.Engine(z, Q, pid, Omega, N, variant=_lib.PHM_MCMC_BIGTREE, seed=7, n_replicas=1024, reduce=True, mapping=mapping)
        eng.run(N); eng.sync()
        outs.append(eng.stats(0, N))
        info = eng.info()
        eng.close()
    np.testing.assert_array_equal(outs[0], outs[1])
    np.testing.assert_allclose(outs[0][:, :4].sum(1), 1024 * z["edge.length"].sum(), rtol=1e-11)
    assert np.all(outs[0][:, 4:] == np.round(outs[0][:, 4:]))
    assert info.seg_read > 0
    if mapping == "branches":          # the two mappings sample the same histories: counts equal, dwell sums to rounding
        eng = _lib.Engine(z, Q, pid, Omega, N, variant=_lib.PHM_MCMC_BIGTREE, seed=7, n_replicas=1024, reduce=True, mapping="replicas")
        eng.run(N); eng.sync()
        lanes = eng.stats(0, N)
        eng.close()
        np.testing.assert_array_equal(outs[0][:, 4:], lanes[:, 4:])
        np.testing.assert_allclose(outs[0][:, :4], lanes[:, :4], rtol=1e-10)


@pytest.mark.parametrize("variant", ["bigtree", "ks"])
def test_tiles_mapping_with_grouped_branches_matches_the_replica_mapping(variant):
    """With many tiles a wave of the (tile, branch) kernel walks several consecutive branches and writes one partial dwell sum
    for the group (phm_tiles.hip); 16 384 replicas on the C2 tree put 7 branches in a group.  The replica mapping (checked
    against the oracle elsewhere) samples the same histories: every count equal, dwell sums equal to rounding -- per replica."""
    z, Q, pid, Omega = synth.config_problem(2)
    S, N = 16384, 4
    v = _lib.PHM_MCMC_BIGTREE if variant == "bigtree" else _lib.PHM_MCMC_KS
    n = Q.shape[0]
    outs = {}
    for mapping in ("tiles", "replicas"):
        eng = _lib.Engine(z, Q, pid, Omega, N, variant=v, seed=11, n_replicas=S, reduce=False, mapping=mapping)
        eng.run(N); eng.sync()
        outs[mapping] = eng.stats(0, N)
        eng.close()
    a, b = outs["tiles"], outs["replicas"]
    assert a.shape == b.shape and a.shape[0] == S
    np.testing.assert_array_equal(a[:, :, n:], b[:, :, n:])
    np.testing.assert_allclose(a[:, :, :n], b[:, :, :n], rtol=1e-10, atol=0)
    np.testing.assert_allclose(a[:, :, :n].sum(2), z["edge.length"].sum(), rtol=1e-11)


def test_errors_are_loud():
    z, Q, pid, Omega = _problem(4, 12, 1)
    with pytest.raises(_lib.PhmError) as e:
        api.sumstatMCMC(z, Q, pid, 0.5 * Omega, 5)          # Omega below |q_ii|
    assert e.value.status == 1
    zbad = dict(z); zbad["edge"] = z["edge"].copy(); zbad["edge"][3, 0] = zbad["edge"][0, 0]
    with pytest.raises((_lib.PhmError, ValueError)):
        api.sumstatMCMC(zbad, Q, pid, Omega, 5)
    Q65 = synth.dense_Q(65, 0.001, 0.002, seed=1)
    z65 = synth.make_tree(6, Q65, 1.0, 3)
    with pytest.raises(_lib.PhmError) as e:
        api.sumstatMCMC(z65, Q65, np.full(65, 1 / 65), 1.0, 2)
    assert e.value.status == 2


def test_alignment_sites_through_the_r_level_api():
    """api.sumstatMCMC_bigtree(..., sites=): S sites of an alignment on one tree, one chain per site, per-site statistics."""
    z, Q, pid, Omega = _problem(4, 30, 17)
    nen, nodelist, root = _orders(z)
    sites = np.random.default_rng(5).integers(1, 5, size=(130, 30)).astype(np.int32)
    got = api.sumstatMCMC_bigtree(z, Q, pid, Omega, 6, sites=sites, seed=9)          # 130 sites: wave per (tile, branch)
    assert got.shape == (130, 6, 16)
    for r in (0, 64, 129):
        zr = dict(z); zr["states"] = sites[r]
        want, rc = O.maketreelistMCMC(zr, Q, pid, np.eye(4) + Q / Omega, Omega, nen, nodelist, root, 6, variant=O.BIGTREE, seed=9, replica=r)
        assert rc == 0
        _same(got[r], want, 4, "tiles")
    red = api.sumstatMCMC_bigtree(z, Q, pid, Omega, 6, sites=sites, seed=9, reduce=True)
    np.testing.assert_array_equal(red[:, 4:], got.sum(0)[:, 4:])
    np.testing.assert_allclose(red[:, :4], got.sum(0)[:, :4], rtol=1e-12)


def test_random_cases_every_mapping_against_the_oracle():
    """tools/fuzz_mappings.py: random state counts (2..12), trees, edge orders, initial paths, replica counts and variants,
    every mapping of the sweep against the oracle (counts bit-exact, dwell <= 1e-10)."""
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_mappings.py")
    spec = importlib.util.spec_from_file_location("fuzz_mappings", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.run(20261003, 120) == 0


def test_replica_mapping_for_5_to_64_states_refuses_paths_beyond_its_scratch():
    """phm_wide.hip keeps the states of one branch in 128 LDS slots per replica: a longer caller-supplied path is refused up front
    with the way out in the message (the automatic choice for one tree never takes this mapping), and works on the other two."""
    Q = synth.dense_Q(6, 0.02, 0.3, seed=2)
    Omega = 1.5 * float(np.max(np.abs(np.diag(Q))))
    pid = np.full(6, 1.0 / 6)
    z = synth.make_tree(12, Q, Omega, 5, pid, init_segments=140)
    with pytest.raises(_lib.PhmError, match="at most 128 segments per branch"):
        api.sumstatMCMC(z, Q, pid, Omega, 3, seed=1, n_replicas=2, mapping="replicas")
    nen, nodelist, root = treeorder.pruningwiseedgeorder(z), treeorder.makenodelist(z), treeorder.myreorder(z)
    want, rc = O.maketreelistMCMC(z, Q, pid, np.eye(6) + Q / Omega, Omega, nen, nodelist, root, 3, seed=1, replica=1)
    assert rc == 0
    for mapping in ("tiles", "branches", "auto"):
        got = api.sumstatMCMC(z, Q, pid, Omega, 3, seed=1, n_replicas=2, mapping=mapping)
        np.testing.assert_array_equal(got[1][:, 6:], want[:, 6:])
        np.testing.assert_allclose(got[1][:, :6], want[:, :6], rtol=1e-10, atol=0)


@pytest.mark.parametrize("mapping", MAPPINGS)
def test_capacity_overflow_is_recovered_like_an_unbounded_list(mapping):
    """The reference's std::list paths are unbounded (src/phylomap.cpp:18-21); a fixed HBM layout is not.  cap_tail = 0.9
    provisions far too little on purpose: the engine rebuilds itself with doubled slots and replays the iterations run so far
    (bit-identical: every random number is addressed, not consumed), so the run completes and still matches the oracle.  With
    the recovery switched off the overflow is reported (PHM_ERR_CAPACITY), never written out of bounds."""
    z, Q, pid, Omega = _problem(4, 40, 3)
    Om = 6.0 * Omega
    nen, nodelist, root = _orders(z)
    kw = dict(variant=_lib.PHM_MCMC_BIGTREE, seed=1, n_replicas=70, mapping=mapping, cap_tail=0.9, storage=1 if mapping == "replicas" else 0)
    with pytest.raises(_lib.PhmError) as e:
        eng = _lib.Engine(z, Q, pid, Om, 30, recover=False, **kw)
        eng.run(30)
        eng.sync()
    assert e.value.status == 6
    eng = _lib.Engine(z, Q, pid, Om, 30, **kw)
    eng.run(12); eng.sync()                      # overflows and recovers here ...
    eng.run(18); eng.sync()                      # ... and keeps going on the rebuilt engine
    got = eng.stats(0, 30)
    for r in (0, 63, 69):
        want, rc, dump = O.maketreelistMCMC(z, Q, pid, np.eye(4) + Q / Om, Om, nen, nodelist, root, 30, variant=O.BIGTREE, seed=1, replica=r,
                                            dump=True)
        assert rc == 0
        _same(got[r], want, 4, mapping)
        np.testing.assert_array_equal(eng.dump(r)["seg_count"], dump.seg_count)
    eng.close()
    ok = api.sumstatMCMC_bigtree(z, Q, pid, Om, 5, seed=1, n_replicas=70, mapping=mapping)      # default capacity: no recovery needed
    np.testing.assert_array_equal(ok[0][:, 4:], got[0][:5, 4:])


def test_failed_capacity_recovery_leaves_a_dead_handle_not_freed_memory():
    """the recovery frees the engine's buffers before the replacement exists.  When the replacement cannot be built
    (here: phm_debug_options.fail_recovery; in the field: doubled slots that do not fit) the handle must refuse every further call
    instead of sweeping on freed HBM."""
    z, Q, pid, Omega = _problem(4, 40, 3)
    Om = 6.0 * Omega
    eng = _lib.Engine(z, Q, pid, Om, 30, variant=_lib.PHM_MCMC_BIGTREE, seed=1, n_replicas=70, mapping="tiles", cap_tail=0.9, fail_recovery=1)
    eng.run(12)
    with pytest.raises(_lib.PhmError) as e:
        eng.sync()                               # overflow -> recovery -> the replacement "does not fit"
    assert e.value.status == 6
    for call in (lambda: eng.run(1), eng.sync, lambda: eng.stats(0, 1), lambda: eng.dump(0), lambda: eng.set_model(Q), eng.info):
        with pytest.raises(_lib.PhmError) as e:
            call()
        assert e.value.status == 6 and "could not be rebuilt" in str(e.value)
    eng.close()                                  # destroying a dead handle is fine
    ok = api.sumstatMCMC_bigtree(z, Q, pid, Om, 5, seed=1, n_replicas=70, mapping="tiles", cap_tail=0.9)      # and the library is unharmed
    assert ok.shape == (70, 5, 16)


def test_capacity_recovery_replays_the_rate_updates():
    """sumstatMCMCbf changes Q after every sweep: a recovery has to replay the model changes at their iterations."""
    z, Q, pid, Omega = _problem(2, 30, 21)
    nen, nodelist, root = _orders(z)
    prior = np.array([.55, 1, .56, 1.01])
    Om = 40.0 * Omega
    want, rc = O.maketreelistMCMC(z, Q, pid, np.eye(2) + Q / Om, Om, nen, nodelist, root, 25, variant=O.BF, seed=4, prior=prior)
    assert rc == 0
    for mapping in ("branches", "tiles"):
        got = api.sumstatMCMCbf(z, Q, pid, Om, 25, prior, seed=4, mapping=mapping, cap_tail=0.9)
        _same(got, want, 2, mapping, ks=True)


@pytest.mark.parametrize("mapping", ["branches", "tiles"])
def test_capacity_overflow_is_recovered_for_wide_state_spaces(mapping):
    n = 6
    Q = synth.dense_Q(n, 0.02, 0.08, seed=66)
    Omega = 1.25 * float(np.max(np.abs(np.diag(Q))))
    pid = np.full(n, 1.0 / n)
    z = synth.make_tree(20, Q, Omega, 67, pid)
    nen, nodelist, root = _orders(z)
    Om = 8.0 * Omega
    with pytest.raises(_lib.PhmError) as e:
        eng = _lib.Engine(z, Q, pid, Om, 30, variant=_lib.PHM_MCMC_BIGTREE, seed=1, n_replicas=5, mapping=mapping, cap_tail=0.9, recover=False)
        eng.run(30)
        eng.sync()
    assert e.value.status == 6
    got = api.sumstatMCMC_bigtree(z, Q, pid, Om, 30, seed=1, n_replicas=5, mapping=mapping, cap_tail=0.9)
    for r in (0, 4):
        want, rc = O.maketreelistMCMC(z, Q, pid, np.eye(n) + Q / Om, Om, nen, nodelist, root, 30, variant=O.BIGTREE, seed=1, replica=r)
        assert rc == 0
        _same(got[r], want, n, mapping)


def test_capacity_overflow_of_the_state_per_lane_tile_kernel_is_reported():
    """mapping 1 with n > 4 (phm_wide.hip) keeps a fixed 128-segment scratch per branch: not recoverable by regrowth"""
    n = 6
    Q = synth.dense_Q(n, 0.02, 0.08, seed=66)
    Omega = 1.25 * float(np.max(np.abs(np.diag(Q))))
    pid = np.full(n, 1.0 / n)
    z = synth.make_tree(20, Q, Omega, 67, pid)
    with pytest.raises(_lib.PhmError) as e:
        eng = _lib.Engine(z, Q, pid, 8.0 * Omega, 30, variant=_lib.PHM_MCMC_BIGTREE, seed=1, n_replicas=5, mapping="replicas", cap_tail=0.9)
        eng.run(30)
        eng.sync()
    assert e.value.status == 6
