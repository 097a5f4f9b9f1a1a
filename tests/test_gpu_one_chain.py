"""The one-chain mapping (phm_narrow.hip: what a plain R call gets) on the tree shapes its latency-shaped kernels special-case:
subtree clusters in several tiers, a walk with hundreds of levels, node states beyond the LDS copy, windows of the branch kernel.
Counts bit-exact against the CPU oracle, dwell sums within 1e-10 (src/phylomap.cpp:775-785 one sweep; :503-529, :591-663, :264-413)."""
import numpy as np
import pytest

import oracle_lib as O
from phylomap_amd import _lib, api, synth, treeorder

pytestmark = pytest.mark.gpu


def _orders(z):
    return treeorder.pruningwiseedgeorder(z), treeorder.makenodelist(z), treeorder.myreorder(z)


def _tree_from_edges(edge, lens, Q, pid, seed, init_segments=2):
    """The tail of synth.make_tree for a hand-made topology (cladewise edge rows, tips 1..T, root T + 1)."""
    T = edge.shape[0] // 2 + 1
    states = np.asarray(synth.simulate_tips(edge, lens, Q, pid, seed), dtype=np.int32)
    maps, mapnames = [], []
    node_states = np.ones((edge.shape[0], 2), dtype=np.int32)
    for r in range(edge.shape[0]):
        child = int(edge[r, 1])
        end = int(states[child - 1]) if child <= T else 1
        maps.append(np.full(init_segments, lens[r] / init_segments))
        mapnames.append(np.array([1] * (init_segments - 1) + [end], dtype=np.int32))
        node_states[r, 1] = end
    return {"edge": edge, "Nnode": T - 1, "edge.length": lens, "states": states, "maps": maps, "mapnames": mapnames,
            "node.states": node_states}


def _ladder(T, mean_len, seed):
    """Caterpillar: internal node k hangs tip k and internal node k + 1; the last one hangs two tips.  Depth T - 1."""
    rs = np.random.default_rng(seed)
    edges = []
    for k in range(T - 1):
        node = T + 1 + k
        edges.append((node, k + 1))                                  # a tip
        edges.append((node, node + 1) if k < T - 2 else (node, T))   # the next rung / the last tip
    edge = np.asarray(edges, dtype=np.int32)
    # cladewise: a node's rows follow the row that leads to it -- here they already do (tip row, then the rung and its subtree)
    return edge, rs.exponential(mean_len, size=edge.shape[0])


def _same(got, want, n, ks=False):
    ncnt = n * n if ks else n * (n - 1)
    np.testing.assert_array_equal(got[..., n:n + ncnt], want[..., n:n + ncnt])
    np.testing.assert_allclose(got[..., :n], want[..., :n], rtol=1e-10, atol=0)
    np.testing.assert_array_equal(got[..., n + ncnt:], want[..., n + ncnt:])


@pytest.mark.parametrize("fn,variant", [("sumstatMCMC_bigtree", O.BIGTREE), ("sumstatMCMCks_sweep", O.KS)])
def test_ladder_tree_many_cluster_tiers_and_a_walk_of_699_levels(fn, variant):
    """700 tips in a caterpillar: three tiers of subtree clusters (256 + 256 + 187 nodes, each a 256-level chain), 699 walk levels
    of one edge each, and the ks sweep's re-sampled tips on every rung."""
    Q = synth.config_Q(2)
    Omega = 1.25 * float(np.max(np.abs(np.diag(Q))))
    pid = np.full(4, 0.25)
    edge, lens = _ladder(700, 1.0 / Omega, 5)
    z = _tree_from_edges(edge, lens, Q, pid, seed=9)
    if variant == O.KS:                              # only the parity of a tip state is observed
        z = dict(z, states=((z["states"] - 1) % 2 + 1).astype(np.int32))
        for b, (p_, c_) in enumerate(edge):
            if c_ <= 700:
                z["mapnames"][b][-1] = z["states"][c_ - 1]
    nen, nodelist, root = _orders(z)
    S, N = 3, 6
    got = getattr(api, fn)(z, Q, pid, Omega, N, seed=77, n_replicas=S, mapping="branches")
    for r in range(S):
        want, rc = O.maketreelistMCMC(z, Q, pid, np.eye(4) + Q / Omega, Omega, nen, nodelist, root, N, variant=variant, seed=77, replica=r)
        assert rc == 0
        _same(got[r], want, 4, ks=variant == O.KS)


def test_one_chain_on_70000_tips_node_states_beyond_the_lds_copy():
    """69 999 internal nodes: the walk keeps the node states in the global array (the LDS copy holds 61 440), the pruning sweep
    has three tiers; three sweeps of one chain against the oracle, and the tree-length invariant."""
    z, Q, pid, Omega = synth.config_problem(3, n_tips=70000)
    nen, nodelist, root = _orders(z)
    N = 3
    eng = _lib.Engine(z, Q, pid, Omega, N, variant=_lib.PHM_MCMC_BIGTREE, seed=4, n_replicas=1, mapping="branches")
    eng.run(N); eng.sync()
    info = eng.info()
    assert info.mapping == _lib.MAPPING["branches"] and info.recoveries == 0
    got = eng.stats(0, N)[0]
    eng.close()
    want, rc = O.maketreelistMCMC(z, Q, pid, np.eye(4) + Q / Omega, Omega, nen, nodelist, root, N, variant=O.BIGTREE, seed=4, replica=0)
    assert rc == 0
    _same(got, want, 4)
    np.testing.assert_allclose(got[:, :4].sum(1), float(z["edge.length"].sum()), rtol=1e-11)


@pytest.mark.parametrize("n", [2, 3])
def test_long_paths_beyond_the_branch_kernel_windows_two_and_three_states(n):
    """300 segments per branch on 40 tips: every wave of the branch kernel overflows its map window (192 change points) and its
    variate window (256), for 2 and 3 states (a quad of the pruning kernel then has idle lanes)."""
    Q = {2: synth.config_Q(1), 3: np.array([[-.3, .2, .1], [.05, -.15, .1], [.2, .2, -.4]])}[n]
    Omega = 1.25 * float(np.max(np.abs(np.diag(Q))))
    pid = np.full(n, 1.0 / n)
    z = synth.make_tree(40, Q, Omega, 31, pid, init_segments=300)
    nen, nodelist, root = _orders(z)
    got = api.sumstatMCMC(z, Q, pid, Omega, 5, seed=3, n_replicas=2, mapping="branches")
    for r in range(2):
        want, rc = O.maketreelistMCMC(z, Q, pid, np.eye(n) + Q / Omega, Omega, nen, nodelist, root, 5, seed=3, replica=r)
        assert rc == 0
        _same(got[r], want, n)


@pytest.mark.parametrize("fn,variant,n", [("sumstatMCMC", O.PLAIN, 2), ("sumstatMCMC_bigtree", O.BIGTREE, 4), ("sumstatMCMCks_sweep", O.KS, 4)])
def test_one_branch_of_3000_segments_the_wave_wide_walk(fn, variant, n):
    """The branches that get a wavefront of their own are walked 64 old segments at a time (narrow_branch_wide: a scan of composed
    transition maps, ballots for the counts, the left-to-right sums as a loop over lanes).  One branch of 3 000 segments (47 blocks,
    the last one ragged), one of 65 (one full block and a single lane), one whose path holds a zero-length segment (the reference's
    iterators stop advancing there, src/phylomap.cpp:397, :405-406), the rest short; several sweeps, so that later sweeps read what
    the wide walk wrote."""
    Q = synth.config_Q(1) if n == 2 else synth.config_Q(2)
    Omega = 1.25 * float(np.max(np.abs(np.diag(Q))))
    pid = np.full(n, 1.0 / n)
    z = synth.make_tree(40, Q, Omega, 17, pid, init_segments=3)
    T = 40
    if variant == O.KS:
        z = dict(z, states=((z["states"] - 1) % 2 + 1).astype(np.int32))
        for b, (p_, c_) in enumerate(z["edge"]):
            if c_ <= T:
                z["mapnames"][b][-1] = z["states"][c_ - 1]
    order = np.argsort(-z["edge.length"])
    for b, m in zip(order[:3], (3000, 65, 130)):
        end = z["mapnames"][b][-1]
        z["maps"][b] = np.full(m, z["edge.length"][b] / m)
        z["mapnames"][b] = np.array([1] * (m - 1) + [end], dtype=np.int32)
    b = order[2]
    z["maps"][b][70] = 0.0                                           # the walk stops cutting here
    z["edge.length"] = np.array([mp.sum() for mp in z["maps"]])
    nen, nodelist, root = _orders(z)
    S, N = 2, 4
    got = getattr(api, fn)(z, Q, pid, Omega, N, seed=21, n_replicas=S, mapping="branches")
    for r in range(S):
        want, rc = O.maketreelistMCMC(z, Q, pid, np.eye(n) + Q / Omega, Omega, nen, nodelist, root, N, variant=variant, seed=21, replica=r)
        assert rc == 0
        _same(got[r], want, n, ks=variant == O.KS)


@pytest.mark.parametrize("case", ["ladder", "C2", "long"])
def test_pruning_clusters_with_and_without_level_barriers_same_bits(case):
    """The subtree clusters of the pruning sweep come in two forms (phm_narrow.hip): a barrier per tree level, or nodes handed out in
    height order to whichever eight lanes are free, each quad starting when its child's vector is there (chosen when some branch is
    expected to hold >= 96 segments).  Same instructions per node: identical statistics, on a caterpillar (every node waits for the
    one below: 256-deep dependencies inside a cluster, three tiers), on C2 and on paths of 300 segments; the dependency-driven form
    against the oracle as well."""
    if case == "ladder":
        Q = synth.config_Q(2); n = 4
        Omega = 1.25 * float(np.max(np.abs(np.diag(Q))))
        pid = np.full(4, 0.25)
        edge, lens = _ladder(700, 1.0 / Omega, 5)
        z = _tree_from_edges(edge, lens, Q, pid, seed=9)
    elif case == "C2":
        z, Q, pid, Omega = synth.config_problem(2); n = 4
    else:
        Q = np.array([[-.3, .2, .1], [.05, -.15, .1], [.2, .2, -.4]]); n = 3
        Omega = 1.25 * float(np.max(np.abs(np.diag(Q))))
        pid = np.full(3, 1.0 / 3)
        z = synth.make_tree(60, Q, Omega, 31, pid, init_segments=300)
    S, N = 3, 5
    got = {f: api.sumstatMCMC_bigtree(z, Q, pid, Omega, N, seed=8, n_replicas=S, mapping="branches", pruning_form=f) for f in (1, 2)}
    np.testing.assert_array_equal(got[1], got[2])
    nen, nodelist, root = _orders(z)
    want, rc = O.maketreelistMCMC(z, Q, pid, np.eye(n) + Q / Omega, Omega, nen, nodelist, root, N, variant=O.BIGTREE, seed=8, replica=S - 1)
    assert rc == 0
    _same(got[2][S - 1], want, n)


def test_automatic_mapping_follows_tree_size():
    """profiles/r04_probe_crossover.log: the branch mapping up to ~20 chains on 10 000 tips, ~90 on 1 000 tips, ~450 on 100 (the (tile, branch)
    mapping with a single tile got 2.5x faster in round 4: level clusters, counter copies)."""
    for cfg, S, want in [(3, 16, "branches"), (3, 32, "tiles"), (2, 64, "branches"), (2, 128, "tiles"), (1, 256, "branches"), (1, 1024, "tiles")]:
        z, Q, pid, Omega = synth.config_problem(cfg)
        eng = _lib.Engine(z, Q, pid, Omega, 2, variant=_lib.PHM_MCMC_BIGTREE, seed=1, n_replicas=S)
        assert eng.info().mapping == _lib.MAPPING[want], (cfg, S)
        eng.close()


def test_automatic_mapping_keeps_the_branch_mapping_on_paths_of_hundreds_of_segments():
    """Beyond 64 segments on a branch the (tile, branch) kernel leaves its two-pass form for a lane-sequential loop, ~20 us per segment of
    one wave (the reference's squamate tree at Omega = 10: 46-51 ms per sweep from 8 to 256 chains against 0.7-10 ms on the branch
    mapping, profiles/r04_probe_squamate_crossover.log): the automatic choice stays with the branch mapping up to floor / slope chains."""
    Q = synth.config_Q(1)
    Omega = 1.25 * float(np.max(np.abs(np.diag(Q))))
    z = synth.make_tree(200, Q, Omega, 3, np.full(2, 0.5), init_segments=2)
    long_z = dict(z, maps=[mp * 30.0 for mp in z["maps"]])               # Omega t_b = 120 on average
    long_z["edge.length"] = z["edge.length"] * 30.0
    for tree, S, want in [(z, 300, "tiles"), (long_z, 300, "branches"), (long_z, 20000, "tiles")]:      # floor / slope = 432 chains on long_z
        eng = _lib.Engine(tree, Q, np.full(2, 0.5), Omega, 2, variant=_lib.PHM_MCMC_BIGTREE, seed=1, n_replicas=S)
        assert eng.info().mapping == _lib.MAPPING[want], (S, want)
        eng.close()


def test_one_chain_in_several_run_calls_and_single_row_reads():
    """run(2), read the last row (served from the row the statistics kernel left in host memory), run(3), read everything: the
    sweeps of one call hand their statistics to the next sweep's first launch, the last one of a call to a launch of its own."""
    z, Q, pid, Omega = synth.config_problem(2)
    nen, nodelist, root = _orders(z)
    want, rc = O.maketreelistMCMC(z, Q, pid, np.eye(4) + Q / Omega, Omega, nen, nodelist, root, 5, variant=O.BIGTREE, seed=12, replica=0)
    assert rc == 0
    eng = _lib.Engine(z, Q, pid, Omega, 5, variant=_lib.PHM_MCMC_BIGTREE, seed=12, n_replicas=1)
    assert eng.info().mapping == _lib.MAPPING["branches"]
    eng.run(2); eng.sync()
    _same(eng.stats(1, 1)[0], want[1:2], 4)
    eng.run(3); eng.sync()
    _same(eng.stats(4, 1)[0], want[4:5], 4)
    _same(eng.stats(0, 5)[0], want, 4)
    eng.close()


def test_segment_counter_of_the_branch_mapping_counts_every_sweep_once():
    """phm_info.seg_read (the measured m + m' behind bench.py's B_alg) is the same whether the rows are summed over replicas on the
    device (reduce: every sweep adds its own row) or not (the row is added by the next sweep's first launch), and whether the sweeps
    come in one call or one by one (the reduce path counted all but the last sweep of a call twice)."""
    z, Q, pid, Omega = synth.config_problem(2)
    K, S = 6, 3
    seen = {}
    for reduce in (False, True):
        for step in (K, 1):
            eng = _lib.Engine(z, Q, pid, Omega, K, variant=_lib.PHM_MCMC_BIGTREE, seed=5, n_replicas=S, reduce=reduce, mapping="branches")
            for _ in range(K // step):
                eng.run(step)
            eng.sync()
            seen[(reduce, step)] = eng.info().seg_read
            eng.close()
    assert len(set(seen.values())) == 1, seen
    E = z["edge"].shape[0]
    assert 4 * E * S * K < seen[(True, K)] < 16 * E * S * K          # mean m + m' is about 10 at Omega * mean(t_b) = 4
