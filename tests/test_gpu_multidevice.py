"""Replica sharding over the GPUs of a node INSIDE the one-shot C-ABI calls (phm_options.n_devices / devices[]):
the reference's caller is one R function -> .Call -> one C++ driver (R/sumstatMCMC_bigtree.R:21-29 -> src/phylomap.cpp:942-986), so
the only way it can reach GPUs 1..7 is below the boundary.  The boxes of this pool have ONE GPU: the device list repeats ordinal 0
(several engines and host threads on one card) -- sharding by global replica id, the per-device engines, the fold of the per-tile sums
in device order and the error paths are the real ones, only the ordinals differ on an 8-GPU node.
Bar: transition counts exactly those of one device, dwell sums to 1e-12 (bit-identical where stated)."""
import numpy as np
import pytest

from phylomap_amd import _lib, api, synth

pytestmark = pytest.mark.gpu

N = 6


def _cols(n):
    return slice(0, n), slice(n, None)


@pytest.mark.parametrize("devices", [[0, 0], [0, 0, 0, 0]])
@pytest.mark.parametrize("cfg,tips,mapping,S", [
    (2, 300, "tiles", 201),        # n <= 4, wave per (tile, branch); shards that are neither equal nor whole tiles
    (2, 300, "tiles", 512),        # whole tiles on every device
    (2, 300, "replicas", 330),     # fused lane-per-replica kernel
    (2, 300, "branches", 6),       # a handful of chains in latency form: one or two per device
    (5, 150, "tiles", 203),        # 20 states, banded B: band kernels
    (4, 60, "tiles", 130),         # 61 states: pruning on the matrix cores
    (5, 150, "branches", 14),      # 20 states, wave per (replica, branch)
])
def test_sharded_one_shot_call_equals_one_device(cfg, tips, mapping, S, devices):
    z, Q, pid, Omega = synth.config_problem(cfg, n_tips=tips)
    n = Q.shape[0]
    dw, cn = _cols(n)
    for reduce in (True, False):
        want = api.sumstatMCMC_bigtree(z, Q, pid, Omega, N, seed=77, n_replicas=S, reduce=reduce, mapping=mapping, device=0)
        got = api.sumstatMCMC_bigtree(z, Q, pid, Omega, N, seed=77, n_replicas=S, reduce=reduce, mapping=mapping, devices=devices)
        assert got.shape == want.shape
        np.testing.assert_array_equal(got[..., cn], want[..., cn])
        np.testing.assert_allclose(got[..., dw], want[..., dw], rtol=1e-12, atol=0)
        if reduce:
            np.testing.assert_allclose(got[:, dw].sum(1), S * z["edge.length"].sum(), rtol=1e-11)


def test_whole_tile_shards_of_the_replica_mapping_fold_bit_identically():
    """The fused lane-per-replica kernel sums a tile's lanes by butterfly and the tiles in order; with whole tiles per device the
    fold carried from device to device adds the same terms in the same order as one device: equal to the last bit."""
    z, Q, pid, Omega = synth.config_problem(2, n_tips=300)
    S = 64 * 12
    want = api.sumstatMCMC_bigtree(z, Q, pid, Omega, N, seed=5, n_replicas=S, reduce=True, mapping="replicas", device=0)
    for devices in ([0, 0], [0, 0, 0], [0, 0, 0, 0]):
        got = api.sumstatMCMC_bigtree(z, Q, pid, Omega, N, seed=5, n_replicas=S, reduce=True, mapping="replicas", devices=devices)
        np.testing.assert_array_equal(got, want)


def test_sites_are_sharded_with_their_tip_vectors():
    """tips_per_replica: device d gets the tip vectors of its own replica range"""
    z, Q, pid, Omega = synth.config_problem(2, n_tips=120)
    rs = np.random.default_rng(3)
    S = 150
    sites = rs.integers(1, 5, size=(S, 120))
    want = api.sumstatMCMC_bigtree(z, Q, pid, Omega, N, seed=9, sites=sites, device=0)
    got = api.sumstatMCMC_bigtree(z, Q, pid, Omega, N, seed=9, sites=sites, devices=[0, 0, 0])
    np.testing.assert_array_equal(got[..., 4:], want[..., 4:])
    np.testing.assert_allclose(got[..., :4], want[..., :4], rtol=1e-12, atol=0)
    assert not np.array_equal(want[0, :, 4:], want[149, :, 4:])


@pytest.mark.parametrize("fn", ["sumstatMCMC", "SPARSEsumstatMCMC", "sumstatMCMCks_sweep", "sumstatMCMCbf_sweep"])
def test_every_fixed_q_entry_point_takes_the_device_list(fn):
    if fn == "sumstatMCMCks_sweep":
        Q = synth.make2sQ(.1, .1, .2, .2, 10)
    else:
        Q = synth.config_Q(2)
    n = Q.shape[0]
    Omega = 1.25 * float(np.max(np.abs(np.diag(Q))))
    pid = np.full(n, 1.0 / n)
    z = synth.make_tree(60, Q, Omega, 0xABC, pid)
    want = getattr(api, fn)(z, Q, pid, Omega, N, seed=3, n_replicas=100, reduce=True, device=0)
    got = getattr(api, fn)(z, Q, pid, Omega, N, seed=3, n_replicas=100, reduce=True, devices=2 * [0])
    np.testing.assert_array_equal(got[:, n:], want[:, n:])
    np.testing.assert_allclose(got[:, :n], want[:, :n], rtol=1e-12, atol=0)


@pytest.mark.parametrize("cfg,N_s", [(1, 1000), (2, 333)])
def test_sumstatEXP_samples_are_sharded_row_for_row(cfg, N_s):
    """the N i.i.d. samples (src/phylomap.cpp:3045-3048) are addressed by their index: a contiguous range per device"""
    z, Q, pid, _ = synth.config_problem(cfg, n_tips=100 if cfg == 1 else 200)
    eig = api.eigen_decompose(Q)
    want = api.sumstatEXP(z, Q, pid, N_s, eig=eig, seed=11, rescale=True, device=0)
    for devices in ([0, 0], [0, 0, 0, 0, 0]):
        got = api.sumstatEXP(z, Q, pid, N_s, eig=eig, seed=11, rescale=True, devices=devices)
        np.testing.assert_array_equal(got, want)


def test_rate_updating_driver_with_sites_on_several_devices():
    """sumstatMCMCks with S sites sharing Q: every device sweeps its sites, the host adds the rows in device order before the rate
    update (src/phylomap.cpp:1859-1868 with the statistics summed over sites)"""
    Q = synth.make2sQ(.1, .1, .2, .2, 10)
    Omega = 1.25 * float(np.max(np.abs(np.diag(Q))))
    pid = np.full(4, .25)
    z = synth.make_tree(80, Q, Omega, 0x77, pid)
    prior = [1, 10, 2, 10, 20, 2]
    want = api.sumstatMCMCks(z, Q, pid, Omega, 12, prior, seed=21, n_replicas=140, device=0)
    got = api.sumstatMCMCks(z, Q, pid, Omega, 12, prior, seed=21, n_replicas=140, devices=[0, 0, 0])
    np.testing.assert_array_equal(got[:, 4:20], want[:, 4:20])
    np.testing.assert_allclose(got, want, rtol=1e-10, atol=0)


def test_one_chain_stays_on_the_first_device_of_the_list():
    z, Q, pid, Omega = synth.config_problem(2, n_tips=200)
    want = api.sumstatMCMC_bigtree(z, Q, pid, Omega, N, seed=1, device=0)
    got = api.sumstatMCMC_bigtree(z, Q, pid, Omega, N, seed=1, devices=[0, 0, 0, 0])
    np.testing.assert_array_equal(got, want)


def test_fewer_chains_or_samples_than_devices():
    """three chains over four devices: one each on three of them; five EXP samples over eight list entries"""
    z, Q, pid, Omega = synth.config_problem(2, n_tips=100)
    want = api.sumstatMCMC_bigtree(z, Q, pid, Omega, N, seed=2, n_replicas=3, device=0)
    got = api.sumstatMCMC_bigtree(z, Q, pid, Omega, N, seed=2, n_replicas=3, devices=[0, 0, 0, 0])
    np.testing.assert_array_equal(got[..., 4:], want[..., 4:])
    np.testing.assert_allclose(got[..., :4], want[..., :4], rtol=1e-12, atol=0)
    eig = api.eigen_decompose(Q)
    np.testing.assert_array_equal(api.sumstatEXP(z, Q, pid, 5, eig=eig, seed=4, devices=8 * [0]), api.sumstatEXP(z, Q, pid, 5, eig=eig, seed=4, device=0))


def test_bad_device_lists_are_refused():
    z, Q, pid, Omega = synth.config_problem(2, n_tips=50)
    with pytest.raises(_lib.PhmError) as e:
        api.sumstatMCMC_bigtree(z, Q, pid, Omega, 2, n_replicas=128, devices=[0, 99])
    assert e.value.status == 3 and "ordinal" in str(e.value)
    with pytest.raises(ValueError):
        _lib.make_options(devices=list(range(9)))
    o = _lib.make_options(n_replicas=128)
    o.n_devices = 9
    import ctypes as C
    ft = _lib.FlatTree(z)
    Qf = np.asfortranarray(Q)
    nen, nodelist, root = _lib.tree_orders(z)
    out = np.zeros((2, 16), order="F")
    st = _lib.load().phm_maketreelistMCMC_bigtree(C.byref(ft.c), 4, _lib._p(Qf, C.c_double), _lib._p(np.ascontiguousarray(pid), C.c_double), None,
                                                  float(Omega), _lib._p(nen, C.c_int32), _lib._p(nodelist, C.c_int32), root, 2, C.byref(o),
                                                  _lib._p(out, C.c_double))
    assert st == 1


def test_an_error_on_one_device_is_reported_with_its_device():
    """an overflow that may not be recovered (no_recovery) inside a worker thread reaches the caller as the call's status"""
    z, Q, pid, Omega = synth.config_problem(2, n_tips=40)
    with pytest.raises(_lib.PhmError) as e:
        api.sumstatMCMC_bigtree(z, Q, pid, 6.0 * Omega, 30, seed=1, n_replicas=140, mapping="tiles", cap_tail=0.9, recover=False, devices=[0, 0])
    assert e.value.status == 6 and "device 0" in str(e.value)


def test_c5_sparse_driver_with_the_option_set_the_shim_passes():
    """SPARSEsumstatMCMC on the 5 000-tip C5 tree as an R user reaches it (shim/phylomap_shim.cpp request_from_R):
    options(phylomap.hip.replicas = 128, phylomap.hip.rescale = TRUE, phylomap.hip.devices = 2) -> n_replicas, reduce, rescale_pruning,
    n_devices; without the rescaling the driver underflows there as the reference does (src/phylomap.cpp:490-501)."""
    z, Q, pid, Omega = synth.config_problem(5)
    n = Q.shape[0]
    with pytest.raises(_lib.PhmError) as e:
        api.SPARSEsumstatMCMC(z, Q, pid, Omega, 2, seed=1, n_replicas=128, reduce=True)
    assert e.value.status == 5
    one = api.SPARSEsumstatMCMC(z, Q, pid, Omega, 3, seed=1, n_replicas=128, reduce=True, rescale=True, device=0)
    two = api.SPARSEsumstatMCMC(z, Q, pid, Omega, 3, seed=1, n_replicas=128, reduce=True, rescale=True, devices=[0, 0])
    np.testing.assert_array_equal(two[:, n:], one[:, n:])
    np.testing.assert_allclose(two[:, :n], one[:, :n], rtol=1e-12, atol=0)
    np.testing.assert_allclose(one[:, :n].sum(1), 128 * z["edge.length"].sum(), rtol=1e-11)
