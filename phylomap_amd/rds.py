"""Minimal reader for R's XDR serialisation (``saveRDS``), enough for phylomap tree objects.

The reference ships its real-data input as an RDS file (``inst/extdata/Squamate/phylomap_compatible_squamate_tree.RData``,
written by ``saveRDS`` at R/Squamate_tree_setup.R:85) and reads trees with ``readRDS``
(vignettes/Squamate_DIC_model_selection.Rnw:78).  R is not available to this project, so this module restates the
published format (R Internals, "Serialization Formats"; version 2, XDR): gzip stream, ``X\\n`` header, three ints, then
one item = flags word (type in the low byte, has-attr / has-tag / is-object bits) followed by the payload.

Supported items: NULL, symbols, pairlists (attributes), character / logical / integer / real vectors, generic
vectors (lists), back references, and the ALTREP forms version-3 files use for ordinary vectors (compact integer / real
sequences such as ``1:n``, wrapped vectors, deferred ``as.character`` of a number vector).  R lists come back as ``dict`` (when named) or ``list``; atomic vectors as numpy
arrays; a ``dim`` attribute reshapes (column-major).  Names of atomic vectors are kept in ``RVector.names``.
"""
from __future__ import annotations

import gzip
import struct

import numpy as np

NILVALUE, REFSXP, GLOBALENV, EMPTYENV, BASEENV, ALTREP_SXP = 254, 255, 253, 242, 241, 238
SYMSXP, LISTSXP, CHARSXP, LGLSXP, INTSXP, REALSXP, STRSXP, VECSXP = 1, 2, 9, 10, 13, 14, 16, 19
NA_INTEGER = -2147483648


class RVector(np.ndarray):
    """numpy array carrying the R ``names`` attribute (phylomap stores segment states as names(maps[[b]]))."""
    names = None

    def __array_finalize__(self, obj):
        self.names = getattr(obj, "names", None)


class _Reader:
    def __init__(self, data: bytes):
        self.b, self.i, self.refs = data, 0, []

    def take(self, n):
        out = self.b[self.i:self.i + n]
        if len(out) != n:
            raise ValueError("truncated RDS stream")
        self.i += n
        return out

    def int(self):
        return struct.unpack(">i", self.take(4))[0]

    def length(self):
        n = self.int()
        if n == -1:                                    # long vector: two more ints
            hi, lo = struct.unpack(">II", self.take(8))
            n = (hi << 32) | lo
        return n

    def item(self):
        flags = self.int()
        t, has_attr, has_tag = flags & 0xFF, bool(flags & 0x200), bool(flags & 0x400)
        if t == NILVALUE:
            return None
        if t in (GLOBALENV, EMPTYENV, BASEENV):
            return None
        if t == REFSXP:
            idx = flags >> 8
            if idx == 0:
                idx = self.int()
            return self.refs[idx - 1]
        if t == SYMSXP:
            name = self.item()
            self.refs.append(name)
            return name
        if t == CHARSXP:
            n = self.int()
            return None if n == -1 else self.take(n).decode("utf-8", "replace")
        if t == ALTREP_SXP:                            # version 3: info pairlist (class, package, type), state, attributes
            info, state, attr = self.item(), self.item(), self.item()
            cls = info[0][1] if info else None
            attrs = dict(attr or [])
            if cls in ("compact_intseq", "compact_realseq"):      # state = c(length, first, increment)
                n, first, incr = (float(v) for v in np.asarray(state).reshape(-1)[:3])
                val = first + incr * np.arange(int(n))
                return _finish(val.astype(np.int32 if cls == "compact_intseq" else np.float64), attrs)
            if cls in ("wrap_integer", "wrap_real", "wrap_logical", "wrap_string", "wrap_list", "wrap_complex", "wrap_raw"):
                val = state[0] if isinstance(state, list) else state      # state = list(x, meta)
                return _finish(val, attrs) if attrs else val
            if cls == "deferred_string":                # state = (number vector . scientific flag): as.character(x), unevaluated
                arg = state[0][1] if isinstance(state, list) and state and isinstance(state[0], tuple) else state
                arr = np.asarray(arg).reshape(-1)
                val = [str(int(v)) if float(v).is_integer() else repr(float(v)) for v in arr]
                return _finish(val, attrs)
            raise ValueError(f"unsupported ALTREP class {cls!r} (re-save with saveRDS(x, file, version = 2))")
        if t == LISTSXP:                               # pairlist: [attr] [tag] car cdr -> list of (tag, value)
            out = []
            while True:
                if has_attr:
                    self.item()
                tag = self.item() if has_tag else None
                out.append((tag, self.item()))
                flags = self.int()
                t, has_attr, has_tag = flags & 0xFF, bool(flags & 0x200), bool(flags & 0x400)
                if t == NILVALUE:
                    return out
                if t != LISTSXP:
                    raise ValueError(f"pairlist continues with unsupported type {t}")
        if t in (LGLSXP, INTSXP):
            n = self.length()
            val = np.frombuffer(self.take(4 * n), dtype=">i4").astype(np.int32)
        elif t == REALSXP:
            n = self.length()
            val = np.frombuffer(self.take(8 * n), dtype=">f8").astype(np.float64)
        elif t == STRSXP:
            n = self.length()
            val = [self.item() for _ in range(n)]
        elif t == VECSXP:
            n = self.length()
            val = [self.item() for _ in range(n)]
        else:
            raise ValueError(f"unsupported SEXP type {t} at byte {self.i}")
        attrs = dict(self.item() or []) if has_attr else {}
        return _finish(val, attrs)


def _finish(val, attrs):
    names, dim = attrs.get("names"), attrs.get("dim")
    if isinstance(val, np.ndarray):
        if dim is not None:
            return val.reshape(tuple(int(d) for d in dim), order="F")
        if names is not None:
            out = val.view(RVector)
            out.names = list(names)
            return out
        return val
    if isinstance(val, list) and names is not None and len(names) == len(val) and all(names) and len(set(names)) == len(names):
        return dict(zip(names, val))
    return val


def read_rds(path: str):
    """Python value of the R object stored by ``saveRDS(obj, path)`` (XDR, versions 2 and 3, gzip or plain)."""
    raw = open(path, "rb").read()
    if raw[:2] == b"\x1f\x8b":
        raw = gzip.decompress(raw)
    if raw[:2] != b"X\n":
        raise ValueError("not an XDR-serialised R object (expected the 'X\\n' header)")
    r = _Reader(raw)
    r.take(2)
    version = r.int()
    r.int(); r.int()                                   # writer version, minimal reader version
    if version == 3:
        r.take(r.int())                                # native encoding name
    elif version != 2:
        raise ValueError(f"unsupported serialisation version {version}")
    return r.item()


def read_phylomap_tree(path: str) -> dict:
    """A tree saved by the reference's tooling as the dict the ``sumstat*`` mirrors take: ``edge`` (E x 2, int32),
    ``Nnode``, ``edge.length``, ``states`` (int32; the shipped file stores doubles), ``maps`` and ``mapnames`` (lists of
    arrays; ``mapnames`` is rebuilt from names(maps[[b]]) when absent, as R/simulate_2_state_tree.R:26-27 does)."""
    x = read_rds(path)
    if not isinstance(x, dict) or "edge" not in x or "maps" not in x:
        raise ValueError("the RDS object is not a phylomap tree (need edge and maps)")
    maps = [np.asarray(m, dtype=np.float64) for m in x["maps"]]
    if x.get("mapnames") is not None:
        mapnames = [np.asarray(m).round().astype(np.int32) for m in x["mapnames"]]
    else:
        mapnames = [np.array([int(s) for s in m.names], dtype=np.int32) for m in x["maps"]]
    z = {"edge": np.asarray(x["edge"]).round().astype(np.int32), "Nnode": int(np.asarray(x["Nnode"]).reshape(-1)[0]),
         "edge.length": np.asarray(x["edge.length"], dtype=np.float64),
         "states": np.asarray(x["states"]).round().astype(np.int32), "maps": maps, "mapnames": mapnames}
    if "node.states" in x:
        z["node.states"] = np.asarray(x["node.states"]).round().astype(np.int32)
    if "tip.label" in x:
        z["tip.label"] = list(x["tip.label"])
    return z
