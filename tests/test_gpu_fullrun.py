"""C3 at its STATED length: sumstatMCMC_bigtree on the 10 000-tip 4-state tree for N = 10 000 sweeps (BASELINE.json configs[2],
src/phylomap.cpp:942-986), 1 024 replicas on the (tile, branch) mapping.  Replica 0 is compared with the CPU oracle over all
10 000 sweeps (counts bit-exact, dwell <= 1e-10), the tree-length invariant is checked on every replica and sweep, and the run
must not have gone through a capacity recovery.  The oracle (about a minute of one CPU core) runs in a thread beside the GPU."""
import threading

import numpy as np
import pytest

import oracle_lib as O
from phylomap_amd import _lib, synth, treeorder

pytestmark = pytest.mark.gpu


def test_c3_full_length_10000_sweeps_1024_replicas():
    z, Q, pid, Omega = synth.config_problem(3)
    nen, nodelist, root = treeorder.pruningwiseedgeorder(z), treeorder.makenodelist(z), treeorder.myreorder(z)
    n, N, S, seed = 4, 10000, 1024, 20261004
    box = {}

    def oracle():
        box["want"], box["rc"] = O.maketreelistMCMC(z, Q, pid, np.eye(n) + Q / Omega, Omega, nen, nodelist, root, N,
                                                    variant=O.BIGTREE, seed=seed, replica=0)

    th = threading.Thread(target=oracle)
    th.start()                                       # ctypes releases the GIL for the duration of the call
    eng = _lib.Engine(z, Q, pid, Omega, N, variant=_lib.PHM_MCMC_BIGTREE, seed=seed, n_replicas=S, mapping="tiles")
    for _ in range(N // 500):                        # the launch queue stays short; nothing is read back in between
        eng.run(500)
        eng.sync()
    info = eng.info()
    assert info.iters_done == N and info.recoveries == 0 and info.mapping == _lib.MAPPING["tiles"]
    tree_len = float(z["edge.length"].sum())
    first = None
    for i0 in range(0, N, 1000):                     # 1 000 sweeps x 1 024 replicas x 16 columns = 131 MB per piece
        st = eng.stats(i0, 1000)
        np.testing.assert_allclose(st[:, :, :n].sum(2), tree_len, rtol=1e-11)
        assert np.all(st[:, :, n:] == np.round(st[:, :, n:])) and np.all(st[:, :, n:] >= 0)
        first = st[0].copy() if first is None else np.concatenate([first, st[0]])
    eng.close()
    th.join()
    assert box["rc"] == 0
    np.testing.assert_array_equal(first[:, n:], box["want"][:, n:])
    np.testing.assert_allclose(first[:, :n], box["want"][:, :n], rtol=1e-10, atol=0)
    # the chain moves: the late sweeps are not a copy of the early ones, and the jump counts fluctuate
    assert not np.array_equal(first[:100, n:], first[-100:, n:])
    assert first[1000:, n:].sum(1).std() > 0


@pytest.mark.parametrize("cfg", [2, 4, 5])
def test_c2_c4_c5_at_their_stated_length_of_1000_sweeps(cfg):
    """N = 1 000 for C2, C4 and C5.  128 replicas on the (tile, item) mapping for 1 000 sweeps; replicas 0 and 127
    against the oracle over all of them.  C2: sumstatMCMC (row-normalised: the plain arithmetic underflows at 1 000 tips in the
    reference too); C4: the 61-state sweep in the n + n^2 counting layout (treesamplebf); C5: tridiagonal 20 states, band kernels."""
    z, Q, pid, Omega = synth.config_problem(cfg)
    nen, nodelist, root = treeorder.pruningwiseedgeorder(z), treeorder.makenodelist(z), treeorder.myreorder(z)
    n, N, S, seed = Q.shape[0], 1000, 128, 7000 + cfg
    variant, ov = (_lib.PHM_MCMC_BF, O.BF) if cfg == 4 else (_lib.PHM_MCMC_BIGTREE, O.BIGTREE)
    box = {}

    def oracle(r):
        box[r] = O.maketreelistMCMC(z, Q, pid, np.eye(n) + Q / Omega, Omega, nen, nodelist, root, N, variant=ov, seed=seed, replica=r)

    ths = [threading.Thread(target=oracle, args=(r,)) for r in (0, S - 1)]
    for th in ths:
        th.start()
    eng = _lib.Engine(z, Q, pid, Omega, N, variant=variant, seed=seed, n_replicas=S, mapping="tiles")
    eng.run(N); eng.sync()
    info = eng.info()
    assert info.recoveries == 0 and info.iters_done == N
    assert info.sparse_chains == (3 if cfg == 5 else 0)
    ncnt = n * n if cfg == 4 else n * (n - 1)
    tree_len = float(z["edge.length"].sum())
    got = {}
    for i0 in range(0, N, 250):
        st = eng.stats(i0, 250)
        np.testing.assert_allclose(st[:, :, :n].sum(2), tree_len, rtol=1e-11)
        for r in (0, S - 1):
            got[r] = st[r].copy() if i0 == 0 else np.concatenate([got[r], st[r]])
    eng.close()
    for th in ths:
        th.join()
    for r in (0, S - 1):
        want, rc = box[r]
        assert rc == 0
        np.testing.assert_array_equal(got[r][:, n:n + ncnt], want[:, n:n + ncnt])
        np.testing.assert_allclose(got[r][:, :n], want[:, :n], rtol=1e-10, atol=0)
        np.testing.assert_array_equal(got[r][:, n + ncnt:], want[:, n + ncnt:])      # bf: Q[0,1], Q[1,0], root state
