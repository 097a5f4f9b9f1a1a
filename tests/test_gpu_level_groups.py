"""The (tile, branch) mapping with FEW tiles (10^2 .. 10^3 replicas: the sites of an alignment): its two tree passes run over subtree
clusters of tree levels, a workgroup per (cluster, tile) walking the cluster's levels with workgroup barriers and one launch per tier, instead of one
launch per tree level (phm_tiles.hip; makePLrcpp* src/phylomap.cpp:503-529, sampleinternalnodes* :618-657, updatenodestates :460-475).
Same draws, same arithmetic: the two forms must agree bit for bit, and with the oracle."""
import numpy as np
import pytest

import oracle_lib as O
from phylomap_amd import _lib, api, synth
from test_gpu_one_chain import _ladder, _orders, _same, _tree_from_edges

pytestmark = pytest.mark.gpu


def _both(z, Q, pid, Omega, N, S, variant, **kw):
    out = []
    for lg in (1, 2, 3):      # phm_debug_options.level_groups: 1 = a launch per level, 2 = clusters by height band, 3 = by subtree size
        eng = _lib.Engine(z, Q, pid, Omega, N, variant=variant, seed=19, n_replicas=S, mapping="tiles", level_groups=lg, **kw)
        eng.run(N); eng.sync()
        out.append((eng.stats(0, N), eng.dump(S - 1), eng.info().last_run_launches))
        eng.close()
    return out


@pytest.mark.parametrize("cfg,tips,S", [(3, None, 130), (2, None, 300), (1, None, 64), (2, 3, 70), (2, 2, 5)])
def test_clusters_equal_levels_bit_for_bit(cfg, tips, S):
    z, Q, pid, Omega = synth.config_problem(cfg, n_tips=tips)
    n = Q.shape[0]
    (lv, dlv, nl), (cl, dcl, nc), (sz, dsz, ns) = _both(z, Q, pid, Omega, 5, S, _lib.PHM_MCMC_BIGTREE)
    np.testing.assert_array_equal(cl, lv)
    np.testing.assert_array_equal(sz, lv)
    for k in ("seg_count", "node_states", "PL"):
        np.testing.assert_array_equal(dcl[k], dlv[k])
        np.testing.assert_array_equal(dsz[k], dlv[k])
    if cfg == 3:
        assert nc <= 12 * 5 and nl >= 60 * 5, (nc, nl)      # 10 000 tips: four tiers of eight levels -> 4 + 4 + 3 launches per sweep instead of 66


def test_clusters_on_a_ladder_tree_many_tiers_against_the_oracle():
    """700-tip caterpillar: 699 height levels, 88 tiers of eight levels; hidden-rates sweep (tips re-drawn) and plain sweep."""
    edge, lens = _ladder(700, 0.5, 3)
    for fn, variant, Q in (("sumstatMCMC_bigtree", O.BIGTREE, synth.config_Q(2)), ("sumstatMCMCks_sweep", O.KS, synth.make2sQ(.1, .1, .2, .2, 10))):
        n = Q.shape[0]
        Omega = 1.25 * float(np.max(np.abs(np.diag(Q))))
        pid = np.full(n, 1.0 / n)
        z = _tree_from_edges(edge, lens, Q, pid, 11)
        nen, nodelist, root = _orders(z)
        got = getattr(api, fn)(z, Q, pid, Omega, 4, seed=8, n_replicas=66, mapping="tiles", level_groups=2)
        for r in (0, 65):
            want, rc = O.maketreelistMCMC(z, Q, pid, np.eye(n) + Q / Omega, Omega, nen, nodelist, root, 4, variant=variant, seed=8, replica=r)
            assert rc == 0
            _same(got[r], want, n, ks=variant == O.KS)


def test_automatic_choice_follows_the_tile_count():
    z, Q, pid, Omega = synth.config_problem(2)          # 999 internal nodes: clusters up to 65 tiles
    for S, few in ((64, True), (4160, True), (4161, False)):
        eng = _lib.Engine(z, Q, pid, Omega, 2, variant=_lib.PHM_MCMC_BIGTREE, seed=1, n_replicas=S, mapping="tiles", reduce=True)
        eng.run(1); eng.sync()
        assert (eng.info().last_run_launches < 20) == few, (S, eng.info().last_run_launches)
        eng.close()


@pytest.mark.parametrize("n,band", [(8, 1), (20, 1), (12, 2), (32, 2), (5, 1), (20, 0), (32, 0)])
def test_deep_tree_with_5_to_32_states_tree_passes_over_subtree_clusters(n, band):
    """phm_wtiles.hip on a ladder (one launch per tree level and pass is all a sweep would do: 4 002 launches, 19.9 of 21.9 ms at
    2 000 tips, 8 states and 4 096 replicas): the band pruning kernel and the node draws for n <= 32 over subtree clusters, a launch
    per tier -- chosen automatically on such a tree, forced by level_groups = 2, switched off by 1.  Same per-node code: bit for bit the
    same statistics, and the oracle's.  band = 0: a dense Q (pruning on the matrix cores stays per level, the node draws use clusters)."""
    Q = synth.dense_Q(n, 0.02, 0.3, seed=n)
    if band:
        idx = np.arange(n)
        Q[np.abs(idx[:, None] - idx[None, :]) > band] = 0.0
        np.fill_diagonal(Q, 0.0); np.fill_diagonal(Q, -Q.sum(1))
    Omega = 1.25 * float(np.max(np.abs(np.diag(Q))))
    pid = np.full(n, 1.0 / n)
    edge, lens = _ladder(300, 2.0 / Omega, 5)
    z = _tree_from_edges(edge, lens, Q, pid, seed=9, init_segments=n if band else 2)      # a banded B reaches state j from i in |i - j| / band steps
    nen, nodelist, root = _orders(z)
    S, N = 70, 4
    got, launches = {}, {}
    for lg in (1, 2, 0):
        eng = _lib.Engine(z, Q, pid, Omega, N, variant=_lib.PHM_MCMC_BIGTREE, seed=19, n_replicas=S, mapping="tiles", level_groups=lg)
        eng.run(N); eng.sync()
        got[lg], launches[lg] = eng.stats(0, N), eng.info().last_run_launches
        eng.close()
    np.testing.assert_array_equal(got[1], got[2])
    np.testing.assert_array_equal(got[1], got[0])
    for r in (0, 63, 64, S - 1):
        want, rc = O.maketreelistMCMC(z, Q, pid, np.eye(n) + Q / Omega, Omega, nen, nodelist, root, N, variant=O.BIGTREE, seed=19, replica=r)
        assert rc == 0
        _same(got[0][r], want, n)

