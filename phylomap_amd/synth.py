"""Synthetic inputs for the stochastic-mapping hot path.

The reference's own generators (R/simulate_2_state_tree.R, R/simulate_4_state_tree.R) are
hard-wired to the 3 951-tip squamate tree, so BASELINE.json's named shapes are produced here:
a deterministic Yule-like tree builder, the rate matrices the reference's vignettes use, a
forward simulator for tip data (restating R/sourceme.R:346-410) and the two-half-segment
initial paths of R/simulate_2_state_tree.R:19-24.

A phylomap tree object ``z`` is a dict with the fields src/phylomap.cpp:896-910,3034 read:
``edge`` (E x 2 int32, 1-based, parent/child), ``Nnode``, ``edge.length`` (E), ``states``
(T, 1-based), ``maps`` / ``mapnames`` (lists of per-branch arrays), ``node.states`` (E x 2).
"""
from __future__ import annotations

import numpy as np

_M0, _M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_W0, _W1 = 0x9E3779B9, 0xBB67AE85
_MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(ctr: np.ndarray, key) -> np.ndarray:
    """Vectorised Philox4x32-10. ``ctr`` is (..., 4) uint32, ``key`` two ints. Returns (..., 4) uint32."""
    c = [ctr[..., i].astype(np.uint64) for i in range(4)]
    k0, k1 = int(key[0]) & 0xFFFFFFFF, int(key[1]) & 0xFFFFFFFF
    for _ in range(10):
        p0 = _M0 * c[0]
        p1 = _M1 * c[2]
        c = [(p1 >> np.uint64(32)) ^ c[1] ^ np.uint64(k0), p1 & _MASK,
             (p0 >> np.uint64(32)) ^ c[3] ^ np.uint64(k1), p0 & _MASK]
        k0 = (k0 + _W0) & 0xFFFFFFFF
        k1 = (k1 + _W1) & 0xFFFFFFFF
    return np.stack(c, axis=-1).astype(np.uint32)


class PhiloxStream:
    """Sequential uniforms in (0,1) from one Philox key; draw d uses block d//2 (same u01 map as the engine)."""

    def __init__(self, seed: int, stream: int = 0):
        self.key = (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
        self.stream = stream
        self._buf = np.empty(0)
        self._pos = 0
        self._block = 0

    def _refill(self, nblocks=4096):
        ctr = np.zeros((nblocks, 4), dtype=np.uint32)
        ctr[:, 0] = np.arange(self._block, self._block + nblocks, dtype=np.uint64).astype(np.uint32)
        ctr[:, 1] = self.stream
        w = philox4x32_10(ctr, self.key).astype(np.uint64)
        lo = (w[:, 1] << np.uint64(32)) | w[:, 0]
        hi = (w[:, 3] << np.uint64(32)) | w[:, 2]
        x = np.stack([lo, hi], axis=1).reshape(-1)
        k = ((x >> np.uint64(12)) << np.uint64(1)) | np.uint64(1)
        self._buf = k.astype(np.float64) * 2.0 ** -53
        self._pos = 0
        self._block += nblocks

    def uniform(self) -> float:
        if self._pos >= self._buf.size:
            self._refill()
        v = self._buf[self._pos]
        self._pos += 1
        return float(v)

    def exponential(self) -> float:
        return -float(np.log(self.uniform()))


# ----------------------------------------------------------------------------------------------
# rate matrices
# ----------------------------------------------------------------------------------------------
def make2sQ(l01, l10, rkappas, lkappas, gammas) -> np.ndarray:
    """Hidden-rates Q of size 2k+2 (restates R/sourceme.R:229-246)."""
    rkappas, lkappas, gammas = (np.atleast_1d(np.asarray(a, dtype=float)) for a in (rkappas, lkappas, gammas))
    k = rkappas.size
    n = 2 * k + 2
    Q = np.zeros((n, n))
    Q[0, 1] = l01
    Q[1, 0] = l10
    for i in range(1, k + 1):
        Q[2 * i - 2, 2 * i] = rkappas[i - 1]
        Q[2 * i - 1, 2 * i + 1] = rkappas[i - 1]
        Q[2 * i, 2 * i - 2] = lkappas[i - 1]
        Q[2 * i + 1, 2 * i - 1] = lkappas[i - 1]
        Q[2 * i, 2 * i + 1] = gammas[i - 1] * l01
        Q[2 * i + 1, 2 * i] = gammas[i - 1] * l10
    np.fill_diagonal(Q, 0.0)
    np.fill_diagonal(Q, -Q.sum(axis=1))
    return Q


def tridiagonal_Q(n=20, rate=0.003) -> np.ndarray:
    """vignettes/phylomap_tutorial.Rnw:72-76."""
    Q = np.zeros((n, n))
    for j in range(n - 1):
        Q[j, j + 1] = rate
        Q[j + 1, j] = rate
    np.fill_diagonal(Q, -Q.sum(axis=1))
    return Q


def neighbour_Q(n=20, degree=6, rate=0.03, seed=0x5EED0005) -> np.ndarray:
    """An UNSTRUCTURED sparse rate matrix: a symmetric neighbour graph in which every state exchanges with `degree` others (the
    shape of an amino-acid model restricted to single-nucleotide neighbours: BASELINE configs[4] says "sparse 20-state amino-acid Q").
    A ring lattice (neighbours at distance 1 .. degree / 2) under a random relabelling of the states, so the matrix is not banded;
    rates rate * U(0.5, 1.5), symmetric."""
    rs = PhiloxStream(seed, stream=9)
    perm = list(range(n))
    for i in range(n - 1, 0, -1):                        # Fisher-Yates on the generator's stream
        j = int(rs.uniform() * (i + 1))
        perm[i], perm[j] = perm[j], perm[i]
    Q = np.zeros((n, n))
    for k in range(n):
        for d in range(1, degree // 2 + 1):
            a, b = perm[k], perm[(k + d) % n]
            Q[a, b] = Q[b, a] = rate * (0.5 + rs.uniform())
    np.fill_diagonal(Q, -Q.sum(axis=1))
    return Q


def dense_Q(n=61, lo=0.005, hi=0.015, seed=0x5EED0004) -> np.ndarray:
    rs = PhiloxStream(seed, stream=7)
    Q = np.zeros((n, n))
    for i in range(n):
        for j in range(n):
            if i != j:
                Q[i, j] = lo + (hi - lo) * rs.uniform()
    np.fill_diagonal(Q, -Q.sum(axis=1))
    return Q


def config_Q(config: int) -> np.ndarray:
    """Q for BASELINE.json configs C1..C5."""
    if config == 1:
        return np.array([[-0.1, 0.1], [0.1, -0.1]])
    if config in (2, 3):
        return make2sQ(0.1, 0.1, 0.2, 0.2, 10.0)
    if config == 4:
        return dense_Q(61)
    if config == 5:
        return tridiagonal_Q(20, 0.03)
    raise ValueError(config)


# ----------------------------------------------------------------------------------------------
# trees
# ----------------------------------------------------------------------------------------------
def random_tree(n_tips: int, mean_length: float, seed: int):
    """Random bifurcating tree: join two uniformly chosen active lineages until one remains.

    Returns (edge, edge_length): ``edge`` is E x 2 int32, 1-based, tips 1..T, root T+1, internal nodes
    numbered in pre-order, rows in cladewise (pre-order) order as ape emits them."""
    rs = PhiloxStream(seed, stream=1)
    T = n_tips
    left = {}
    right = {}
    active = list(range(T))          # provisional ids: tips 0..T-1, joins T..2T-2
    nxt = T
    while len(active) > 1:
        i = int(rs.uniform() * len(active))
        a = active[i]
        active[i] = active[-1]
        active.pop()
        j = int(rs.uniform() * len(active))
        b = active[j]
        active[j] = nxt
        left[nxt], right[nxt] = a, b
        nxt += 1
    root = active[0]
    # renumber internal nodes in pre-order, emit edges cladewise
    new_id = {}
    counter = T + 1
    edges = []
    stack = [(root, 0)]
    while stack:
        node, parent_new = stack.pop()
        if node < T:
            nid = node + 1
        else:
            nid = counter
            counter += 1
        new_id[node] = nid
        if parent_new:
            edges.append((parent_new, nid))
        if node >= T:
            stack.append((right[node], nid))
            stack.append((left[node], nid))
    edge = np.asarray(edges, dtype=np.int32)
    lens = np.array([mean_length * rs.exponential() for _ in range(edge.shape[0])])
    return edge, lens


def simulate_tips(edge, edge_length, Q, pid, seed: int):
    """Forward-simulate one history from the root (restates R/sourceme.R:346-410); returns 1-based tip states."""
    rs = PhiloxStream(seed, stream=2)
    n = Q.shape[0]
    E = edge.shape[0]
    T = E // 2 + 1
    node_state = np.zeros(2 * T, dtype=np.int64)
    cum = np.cumsum(pid) / np.sum(pid)
    root = T + 1
    node_state[root] = int(np.searchsorted(cum, rs.uniform(), side="left")) if n > 1 else 0
    node_state[root] = min(node_state[root], n - 1)
    for r in range(E):                      # cladewise rows: parents are always sampled first
        s = int(node_state[edge[r, 0]])
        t = 0.0
        while True:
            rate = -Q[s, s]
            if rate <= 0:
                break
            t += rs.exponential() / rate
            if t >= edge_length[r]:
                break
            w = Q[s].copy()
            w[s] = 0.0
            c = np.cumsum(w) / w.sum()
            s = min(int(np.searchsorted(c, rs.uniform(), side="left")), n - 1)
        node_state[edge[r, 1]] = s
    return (node_state[1:T + 1] + 1).astype(np.int32)


def make_tree(n_tips: int, Q, Omega: float, seed: int, pid=None, states=None, init_segments: int = 2):
    """Full phylomap tree object.  Initial paths: ``init_segments`` equal pieces per branch, all in state 1 except
    the last piece of a tip branch (two half-length pieces as in R/simulate_2_state_tree.R:19-24; 100 equal pieces
    in R/Squamate_tree_setup.R:57).  With a banded Q the initial segment count must let B^(m-1) connect the
    observed tip states, e.g. init_segments = n for a tridiagonal Q."""
    Q = np.asarray(Q, dtype=float)
    n = Q.shape[0]
    pid = np.full(n, 1.0 / n) if pid is None else np.asarray(pid, dtype=float)
    tau = 4.0 / Omega                                   # Omega * mean branch length = 4
    edge, lens = random_tree(n_tips, tau, seed)
    if states is None:
        states = simulate_tips(edge, lens, Q, pid, seed)
    states = np.asarray(states, dtype=np.int32)
    E = edge.shape[0]
    maps, mapnames = [], []
    node_states = np.ones((E, 2), dtype=np.int32)
    for r in range(E):
        child = int(edge[r, 1])
        end = int(states[child - 1]) if child <= n_tips else 1
        maps.append(np.full(init_segments, lens[r] / init_segments))
        mapnames.append(np.array([1] * (init_segments - 1) + [end], dtype=np.int32))
        node_states[r, 1] = end
    return {"edge": edge, "Nnode": n_tips - 1, "edge.length": lens, "states": states,
            "maps": maps, "mapnames": mapnames, "node.states": node_states}


def make_treelist(n_trees: int, n_tips: int, Q, Omega: float, seed: int, pid=None, init_segments: int = 2):
    """A list of trees as sumstatMCMCmt takes it (R/sumstatMCMCmt.R:29): the same taxa with the same observed tip
    states on ``n_trees`` different random topologies / branch lengths (a posterior sample of trees)."""
    first = make_tree(n_tips, Q, Omega, seed, pid=pid, init_segments=init_segments)
    rest = [make_tree(n_tips, Q, Omega, seed + 7919 * j, pid=pid, states=first["states"], init_segments=init_segments)
            for j in range(1, n_trees)]
    return [first] + rest


def config_problem(config: int, n_tips: int | None = None):
    """(z, Q, pid, Omega) for BASELINE.json config C1..C5; seed = 0x5EED0000 + config."""
    tips = {1: 100, 2: 1000, 3: 10000, 4: 500, 5: 5000}[config] if n_tips is None else n_tips
    Q = config_Q(config)
    Omega = 1.25 * float(np.max(np.abs(np.diag(Q))))
    n = Q.shape[0]
    pid = np.full(n, 1.0 / n)
    z = make_tree(tips, Q, Omega, 0x5EED0000 + config, pid, init_segments=(n if config == 5 else 2))
    return z, Q, pid, Omega
