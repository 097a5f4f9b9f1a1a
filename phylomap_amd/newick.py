"""Newick trees -> phylomap tree objects, without R.

The reference prepares its real-data input in R (R/Squamate_tree_setup.R:8-89): ``ape::read.tree`` on
``inst/extdata/Squamate/squamate.phy``, ``drop.tip`` for the taxa without usable trait data, then every branch is given
an initial path (100 equal segments in state 1, the last segment of a tip branch in the tip's state) and the result is
saved with ``saveRDS``.  This module restates that input side: a Newick parser with ape's node numbering (tips 1..T in
order of appearance, internal nodes T+1.. in pre-order, edges in cladewise order), tip pruning that suppresses the
resulting single-child nodes (branch lengths added up), and the initial-path construction.
"""
from __future__ import annotations

import numpy as np


def _parse(text: str):
    """-> nested nodes {"label", "length", "children"} of the first tree in ``text``."""
    s = text.strip()
    end = s.find(";")
    if end >= 0:
        s = s[:end]
    pos = 0

    def node():
        nonlocal pos
        children = []
        if pos < len(s) and s[pos] == "(":
            pos += 1
            while True:
                children.append(node())
                if pos >= len(s):
                    raise ValueError("unbalanced parentheses in Newick string")
                if s[pos] == ",":
                    pos += 1
                    continue
                if s[pos] == ")":
                    pos += 1
                    break
                raise ValueError(f"unexpected character {s[pos]!r} at {pos}")
        start = pos
        while pos < len(s) and s[pos] not in ",():":
            pos += 1
        label = s[start:pos].strip()
        length = None
        if pos < len(s) and s[pos] == ":":
            pos += 1
            start = pos
            while pos < len(s) and s[pos] not in ",()":
                pos += 1
            length = float(s[start:pos])
        return {"label": label, "length": length, "children": children}

    root = node()
    if pos != len(s):
        raise ValueError(f"trailing text at {pos} in Newick string")
    return root


def _emit(root) -> dict:
    """Nested nodes -> ape-style arrays (iterative pre-order; the squamate tree is ~100 levels deep, others may be deeper)."""
    tips = []
    stack = [root]
    while stack:                                   # tips in order of appearance
        nd = stack.pop()
        if not nd["children"]:
            tips.append(nd)
        else:
            stack.extend(reversed(nd["children"]))
    T = len(tips)
    for i, nd in enumerate(tips):
        nd["id"] = i + 1
    edge, length, node_label = [], [], []
    next_id = T + 1
    root["id"] = next_id
    next_id += 1
    node_label.append(root["label"])
    stack = [root]
    # pre-order: a node's id is assigned when it is first reached; its edges are emitted child by child, each child's
    # subtree before the next sibling (cladewise order)
    work = [(root, 0)]
    while work:
        nd, k = work.pop()
        if k < len(nd["children"]):
            ch = nd["children"][k]
            work.append((nd, k + 1))
            if ch["children"]:
                ch["id"] = next_id
                next_id += 1
                node_label.append(ch["label"])
            edge.append((nd["id"], ch["id"]))
            length.append(np.nan if ch["length"] is None else ch["length"])
            if ch["children"]:
                work.append((ch, 0))
    return {"edge": np.asarray(edge, dtype=np.int32).reshape(-1, 2), "edge.length": np.asarray(length, dtype=np.float64),
            "Nnode": next_id - T - 1, "tip.label": [nd["label"] for nd in tips], "node.label": node_label}


def read_newick(text: str) -> dict:
    """``ape::read.tree(text=...)``: dict with ``edge`` (E x 2, 1-based, cladewise), ``edge.length``, ``Nnode``,
    ``tip.label``, ``node.label``."""
    return _emit(_parse(text))


def drop_tips(text_or_tree, drop_labels) -> dict:
    """``ape::drop.tip``: remove the named tips, remove internal nodes left without tips, suppress nodes left with a single
    child (their branch lengths add up; a single-child root is dropped together with its edge), renumber."""
    root = _parse(text_or_tree) if isinstance(text_or_tree, str) else _from_arrays(text_or_tree)
    drop = set(drop_labels)

    # post-order pruning without recursion
    order, stack = [], [root]
    while stack:
        nd = stack.pop()
        order.append(nd)
        stack.extend(nd["children"])
    for nd in reversed(order):
        if not nd["children"]:
            nd["keep"] = nd.get("was_tip", True) and nd["label"] not in drop
            continue
        kept = [c for c in nd["children"] if c["keep"]]
        flat = []
        for c in kept:                                     # suppress single-child internal nodes below this one
            while c["children"] and len(c["children"]) == 1:
                only = c["children"][0]
                only["length"] = (only["length"] or 0.0) + (c["length"] or 0.0)
                c = only
            flat.append(c)
        nd["children"] = flat
        nd["keep"] = len(flat) > 0
        nd["was_tip"] = False
    while root["children"] and len(root["children"]) == 1:  # a root left with one child disappears with its edge
        root = root["children"][0]
        root["length"] = None
    if not root.get("keep", True) or not root["children"]:
        raise ValueError("no tips left")
    return _emit(root)


def _from_arrays(tree) -> dict:
    edge = np.asarray(tree["edge"])
    T = len(tree["tip.label"])
    nodes = {}
    for i, lab in enumerate(tree["tip.label"]):
        nodes[i + 1] = {"label": lab, "length": None, "children": []}
    labels = tree.get("node.label") or [""] * int(tree["Nnode"])
    for j in range(int(tree["Nnode"])):
        nodes[T + 1 + j] = {"label": labels[j] if j < len(labels) else "", "length": None, "children": []}
    for (p, c), l in zip(edge, tree["edge.length"]):
        nodes[int(c)]["length"] = float(l)
        nodes[int(p)]["children"].append(nodes[int(c)])
    children = set(int(c) for c in edge[:, 1])
    roots = [k for k in nodes if k not in children]
    return nodes[roots[0]]


def as_phylomap(tree: dict, states, segments: int = 100) -> dict:
    """The initial phylomap object of R/Squamate_tree_setup.R:54-85: every branch ``segments`` equal pieces in state 1, the
    last piece of a tip branch in that tip's state; ``states`` 1-based per tip (in ``tip.label`` order)."""
    states = np.asarray(states).round().astype(np.int32)
    T = len(tree["tip.label"])
    if states.shape != (T,):
        raise ValueError("one state per tip expected")
    maps, mapnames = [], []
    node_states = np.ones((tree["edge"].shape[0], 2), dtype=np.int32)
    for r, ((p, c), l) in enumerate(zip(tree["edge"], tree["edge.length"])):
        maps.append(np.full(segments, l / segments))
        names = np.ones(segments, dtype=np.int32)
        if c <= T:
            names[-1] = states[c - 1]
            node_states[r, 1] = states[c - 1]
        mapnames.append(names)
    out = dict(tree)
    out.update({"states": states, "maps": maps, "mapnames": mapnames, "node.states": node_states})
    return out
