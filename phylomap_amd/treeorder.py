"""Host-side traversal orders: the Python mirror of the R helper preamble every ``sumstat*`` wrapper
carries (R/sumstatMCMC.R:1-18): ``pruningwiseedgeorder``, ``makenodelist``, ``myreorder``.

In R these call ``ape::reorder(x, "pruningwise")`` and then match rows with an interpreted E x E
double loop (R/sumstatMCMC.R:4-8).  ape is a third-party dependency that is not under
/root/reference; its pruningwise order is restated here from its published algorithm
(``neworder_pruningwise``: repeated scans of the cladewise edge table that collect every node whose
child edges are all "ready") in O(E): a node is collected in scan number height(node), and within
one scan in the order in which the scan meets its last child edge.  UNVERIFIED against a real ape
install (none available in the build environment); tests pin it against a literal restatement of the
scan algorithm and against the structural properties src/phylomap.cpp:503-514,638-657 rely on
(sibling edges adjacent, children before parents, root pair last).
"""
from __future__ import annotations

import numpy as np


def _children(edge: np.ndarray, n_nodes: int):
    kids = [[] for _ in range(n_nodes + 1)]
    for r in range(edge.shape[0]):
        kids[int(edge[r, 0])].append(r)
    return kids


def find_root(edge: np.ndarray) -> int:
    is_child = np.zeros(int(edge.max()) + 1, dtype=bool)
    is_child[edge[:, 1]] = True
    roots = [int(v) for v in np.unique(edge[:, 0]) if not is_child[v]]
    if len(roots) != 1:
        raise ValueError("tree must have exactly one root")
    return roots[0]


def cladewise_positions(edge: np.ndarray) -> np.ndarray:
    """Position of every edge row in ape's cladewise (pre-order, children in row order) ordering."""
    E = edge.shape[0]
    kids = _children(edge, int(edge.max()))
    pos = np.empty(E, dtype=np.int64)
    stack = list(reversed(kids[find_root(edge)]))
    k = 0
    while stack:
        r = stack.pop()
        pos[r] = k
        k += 1
        stack.extend(reversed(kids[int(edge[r, 1])]))
    if k != E:
        raise ValueError("edge table is not a connected rooted tree")
    return pos


def pruningwise_rows(edge: np.ndarray) -> np.ndarray:
    """0-based rows of ``edge`` in pruningwise order (see module docstring)."""
    edge = np.asarray(edge)
    E = edge.shape[0]
    n_nodes = int(edge.max())
    kids = _children(edge, n_nodes)
    pos = cladewise_positions(edge)
    root = find_root(edge)
    # heights by reverse cladewise order (children appear after their parent edge)
    height = np.zeros(n_nodes + 1, dtype=np.int64)
    for r in np.argsort(-pos):
        p, c = int(edge[r, 0]), int(edge[r, 1])
        height[p] = max(height[p], height[c] + 1)
    internal = [v for v in range(1, n_nodes + 1) if kids[v] and v != root]
    internal.sort(key=lambda v: (height[v], max(pos[r] for r in kids[v])))
    out = []
    for v in internal + [root]:
        out.extend(sorted(kids[v], key=lambda r: pos[r]))
    return np.asarray(out, dtype=np.int64)


def pruningwiseedgeorder(z) -> np.ndarray:
    """R/sumstatMCMC.R:1-10: for each pruningwise position, the 1-based row of ``z$edge``."""
    return (pruningwise_rows(z["edge"]) + 1).astype(np.int32)


def makenodelist(z) -> np.ndarray:
    """R/sumstatMCMC.R:11-17: parents at pruningwise rows E-2i, i = 1..Nnode-1 (root side first)."""
    edge = np.asarray(z["edge"])
    rows = pruningwise_rows(edge)
    E = edge.shape[0]
    return np.asarray([edge[rows[E - 2 * i - 1], 0] for i in range(1, int(z["Nnode"]))], dtype=np.int32)


def myreorder(z) -> int:
    """R/sumstatMCMC.R:18: the parent in the last pruningwise row, i.e. the root node id."""
    edge = np.asarray(z["edge"])
    return int(edge[pruningwise_rows(edge)[-1], 0])
