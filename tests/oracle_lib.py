"""ctypes front end to the CPU oracle (oracle/phm_oracle.c).  TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.path.join(_ROOT, "oracle", "_build", "libphm_oracle.so")

ERR_ZERO_PROB, ERR_UNIF_CAP, ERR_BAD_INPUT, ERR_TAPE, ERR_SAMPLEONCE = 1, 2, 4, 8, 16
PLAIN, BIGTREE, SPARSE, KS, BF, MT, KSMT = 0, 1, 2, 3, 4, 5, 6
FORCE_NORMALISE = 32      # OR-ed into PLAIN / SPARSE: rescaled pruning pass (not in the reference)


class Rng(C.Structure):
    _fields_ = [("mode", C.c_int32), ("seed_lo", C.c_uint32), ("seed_hi", C.c_uint32), ("replica", C.c_uint32),
                ("tape_u", C.POINTER(C.c_double)), ("n_u", C.c_int64), ("pos_u", C.c_int64),
                ("tape_e", C.POINTER(C.c_double)), ("n_e", C.c_int64), ("pos_e", C.c_int64)]


class Tree(C.Structure):
    _fields_ = [("n_tips", C.c_int32), ("n_node", C.c_int32), ("n_edge", C.c_int32),
                ("edge", C.POINTER(C.c_int32)), ("edge_length", C.POINTER(C.c_double)),
                ("states", C.POINTER(C.c_int32)), ("map_off", C.POINTER(C.c_int32)),
                ("maps", C.POINTER(C.c_double)), ("mapnames", C.POINTER(C.c_int32))]


class Dump(C.Structure):
    _fields_ = [("node_states", C.POINTER(C.c_int32)), ("seg_count", C.POINTER(C.c_int32)),
                ("seg_dwell", C.POINTER(C.c_double)), ("seg_state", C.POINTER(C.c_int32)),
                ("seg_cap", C.c_int32), ("PL", C.POINTER(C.c_double))]


def _ptr(a, t):
    return a.ctypes.data_as(C.POINTER(t))


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            subprocess.check_call(["make", "-C", os.path.join(_ROOT, "oracle")], stdout=subprocess.DEVNULL)
        L = C.CDLL(_SO)
        L.orc_u01.restype = C.c_double
        L.orc_u01.argtypes = [C.c_uint32]
        L.orc_log.restype = C.c_double
        L.orc_log.argtypes = [C.c_double]
        L.orc_exp.restype = C.c_double
        L.orc_exp.argtypes = [C.c_double]
        L.orc_stream_u.restype = C.c_double
        L.orc_stream_u.argtypes = [C.c_uint32] * 6
        _lib = L
    return _lib


class FlatTree:
    """Keeps the numpy buffers behind an orc_tree alive."""

    def __init__(self, z):
        edge = np.asarray(z["edge"], dtype=np.int32)
        self.E = edge.shape[0]
        self.T = len(z["states"])
        self.edge = np.asfortranarray(edge).reshape(-1, order="F").copy()
        el = z.get("edge.length")
        self.edge_length = None if el is None else np.ascontiguousarray(el, dtype=np.float64)
        self.states = np.ascontiguousarray(z["states"], dtype=np.int32)
        lens = [len(m) for m in z["maps"]]
        self.map_off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        self.maps = np.concatenate([np.asarray(m, dtype=np.float64) for m in z["maps"]])
        self.mapnames = np.concatenate([np.asarray(m, dtype=np.int32) for m in z["mapnames"]]).astype(np.int32)
        self.c = Tree(self.T, int(z["Nnode"]), self.E, _ptr(self.edge, C.c_int32),
                      _ptr(self.edge_length, C.c_double) if self.edge_length is not None else None,
                      _ptr(self.states, C.c_int32), _ptr(self.map_off, C.c_int32),
                      _ptr(self.maps, C.c_double), _ptr(self.mapnames, C.c_int32))


def make_rng(seed=1, replica=0, tape_u=None, tape_e=None, rstream=False):
    r = Rng()
    keep = []
    if rstream:
        r.mode = 2
        r.seed_lo, r.seed_hi, r.replica = seed & 0xFFFFFFFF, 0, 0
        return r, keep
    if tape_u is not None or tape_e is not None:
        r.mode = 1
        tu = np.ascontiguousarray(tape_u if tape_u is not None else [], dtype=np.float64)
        te = np.ascontiguousarray(tape_e if tape_e is not None else [], dtype=np.float64)
        r.tape_u, r.n_u, r.pos_u = _ptr(tu, C.c_double), tu.size, 0
        r.tape_e, r.n_e, r.pos_e = _ptr(te, C.c_double), te.size, 0
        keep = [tu, te]
    else:
        r.mode = 0
        r.seed_lo, r.seed_hi, r.replica = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF, replica
    return r, keep


class DumpBuf:
    def __init__(self, ft: FlatTree, n: int, seg_cap=512):
        self.node_states = np.zeros(2 * ft.T - 1, dtype=np.int32)
        self.seg_count = np.zeros(ft.E, dtype=np.int32)
        self.seg_dwell = np.zeros((ft.E, seg_cap))
        self.seg_state = np.zeros((ft.E, seg_cap), dtype=np.int32)
        self.PL = np.zeros((2 * ft.T - 1, n))
        self.c = Dump(_ptr(self.node_states, C.c_int32), _ptr(self.seg_count, C.c_int32),
                      _ptr(self.seg_dwell, C.c_double), _ptr(self.seg_state, C.c_int32), seg_cap,
                      _ptr(self.PL, C.c_double))


def philox(ctr, key, rounds=10):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    lib().orc_philox4x32(c, k, int(rounds), o)
    return list(o)


def maketreelistMCMC(z, Q, pid, B, Omega, nen, nodelist, root, N, variant=PLAIN, seed=1, replica=0,
                     faithful_search=False, tape_u=None, tape_e=None, dump=False, prior=None, rstream=False, dic=False):
    Q = np.asarray(Q, dtype=np.float64)
    n = Q.shape[0]
    ft = FlatTree(z)
    Qc, Bc = np.asfortranarray(Q), np.asfortranarray(np.asarray(B, dtype=np.float64))
    pid = np.ascontiguousarray(pid, dtype=np.float64)
    nen = np.ascontiguousarray(nen, dtype=np.int32)
    nodelist = np.ascontiguousarray(nodelist, dtype=np.int32)
    cols = n + n * (n - 1)
    if variant & ~FORCE_NORMALISE == KS:
        cols = n + n * n + 2 + 3 * (n // 2 - 1) + 1
    if variant & ~FORCE_NORMALISE == BF:
        cols = n + n * n + 3
    if dic:
        cols += 1
    out = np.zeros((N, cols), order="F")
    rng, keep = make_rng(seed, replica, tape_u, tape_e, rstream)
    if prior is not None:
        prior = np.ascontiguousarray(prior, dtype=np.float64)
        rc = lib().orc_maketreelistMCMC_qupdate(C.byref(ft.c), n, _ptr(Qc, C.c_double), _ptr(pid, C.c_double),
                                                _ptr(Bc, C.c_double), C.c_double(Omega), _ptr(nen, C.c_int32),
                                                _ptr(nodelist, C.c_int32), int(root), int(N), int(variant) | (16 if dic else 0),
                                                _ptr(prior, C.c_double), int(faithful_search), C.byref(rng),
                                                _ptr(out, C.c_double), None)
        return out, rc
    db = DumpBuf(ft, n) if dump else None
    rc = lib().orc_maketreelistMCMC(C.byref(ft.c), n, _ptr(Qc, C.c_double), _ptr(pid, C.c_double),
                                    _ptr(Bc, C.c_double), C.c_double(Omega), _ptr(nen, C.c_int32),
                                    _ptr(nodelist, C.c_int32), int(root), int(N), int(variant),
                                    int(faithful_search), C.byref(rng), _ptr(out, C.c_double),
                                    C.byref(db.c) if db else None)
    del keep
    return (out, rc, db) if dump else (out, rc)


def maketreelistMCMCmt(treelist, Q, pid, B, Omega, nen_m, nodelist_m, roots, N, prior, variant=MT, seed=1, replica=0,
                       faithful_search=False):
    """orc_maketreelistMCMCmt: nen_m / nodelist_m one row per tree (row-major), roots one entry per tree."""
    Q = np.asarray(Q, dtype=np.float64)
    n = Q.shape[0]
    fts = [FlatTree(z) for z in treelist]
    arr = (C.POINTER(Tree) * len(fts))(*[C.pointer(ft.c) for ft in fts])
    Qc, Bc = np.asfortranarray(Q), np.asfortranarray(np.asarray(B, dtype=np.float64))
    pid = np.ascontiguousarray(pid, dtype=np.float64)
    nen_m = np.ascontiguousarray(nen_m, dtype=np.int32)
    nodelist_m = np.ascontiguousarray(nodelist_m, dtype=np.int32)
    roots = np.ascontiguousarray(roots, dtype=np.int32)
    prior = np.ascontiguousarray(prior, dtype=np.float64)
    cols = n + n * n + 2 + (3 * (n // 2 - 1) if variant == KSMT else 0) + 1
    out = np.zeros((N, cols), order="F")
    rng, keep = make_rng(seed, replica)
    rc = lib().orc_maketreelistMCMCmt(arr, len(fts), n, _ptr(Qc, C.c_double), _ptr(pid, C.c_double), _ptr(Bc, C.c_double),
                                      C.c_double(Omega), _ptr(nen_m, C.c_int32), _ptr(nodelist_m, C.c_int32),
                                      _ptr(roots, C.c_int32), int(N), int(variant), _ptr(prior, C.c_double),
                                      int(faithful_search), C.byref(rng), _ptr(out, C.c_double))
    del keep
    return out, rc


def maketreelistEXP(z, Q, pid, nen, nodelist, root, N, lefts, rights, d, seed=1, replica=0,
                    faithful_search=False, recompute=False, tape_u=None, dump=False, rescale=False, rstream=False):
    Q = np.asarray(Q, dtype=np.float64)
    n = Q.shape[0]
    ft = FlatTree(z)
    Qc = np.asfortranarray(Q)
    Lc, Rc, Dc = (np.asfortranarray(np.asarray(a, dtype=np.float64)) for a in (lefts, rights, d))
    pid = np.ascontiguousarray(pid, dtype=np.float64)
    nen = np.ascontiguousarray(nen, dtype=np.int32)
    nodelist = np.ascontiguousarray(nodelist, dtype=np.int32)
    cols = n + n * (n - 1)
    out = np.zeros((N, cols), order="F")
    rng, keep = make_rng(seed, replica, tape_u, None, rstream)
    db = DumpBuf(ft, n) if dump else None
    rc = lib().orc_maketreelistEXP(C.byref(ft.c), n, _ptr(Qc, C.c_double), _ptr(pid, C.c_double),
                                   _ptr(nen, C.c_int32), _ptr(nodelist, C.c_int32), int(root), int(N),
                                   _ptr(Lc, C.c_double), _ptr(Rc, C.c_double), _ptr(Dc, C.c_double),
                                   int(faithful_search), int(bool(recompute)) | (2 if rescale else 0), C.byref(rng), _ptr(out, C.c_double),
                                   C.byref(db.c) if db else None)
    del keep
    return (out, rc, db) if dump else (out, rc)


def matexp(L, R, dvals, t):
    L = np.ascontiguousarray(L, dtype=np.float64)
    R = np.ascontiguousarray(R, dtype=np.float64)
    dv = np.ascontiguousarray(dvals, dtype=np.float64)
    n = L.shape[0]
    P = np.zeros((n, n))
    lib().orc_matexp(_ptr(L, C.c_double), _ptr(R, C.c_double), _ptr(dv, C.c_double), n, C.c_double(t),
                     _ptr(P, C.c_double))
    return P


def matexp_fma(L, R, dvals, t):
    L = np.ascontiguousarray(L, dtype=np.float64)
    R = np.ascontiguousarray(R, dtype=np.float64)
    dv = np.ascontiguousarray(dvals, dtype=np.float64)
    n = L.shape[0]
    P = np.zeros((n, n))
    lib().orc_matexp_fma(_ptr(L, C.c_double), _ptr(R, C.c_double), _ptr(dv, C.c_double), n, C.c_double(t),
                         _ptr(P, C.c_double))
    return P


def expmat_pade(A):
    A = np.ascontiguousarray(A, dtype=np.float64)
    n = A.shape[0]
    out = np.zeros((n, n))
    rc = lib().orc_expmat_pade(_ptr(A, C.c_double), n, _ptr(out, C.c_double))
    return out, rc


def shortener(d, s, n):
    d = np.ascontiguousarray(d, dtype=np.float64).copy()
    s = np.ascontiguousarray(s, dtype=np.int32).copy()
    row = np.zeros(n + n * (n - 1))
    m = lib().orc_shortener(_ptr(d, C.c_double), _ptr(s, C.c_int32), len(d), n, _ptr(row, C.c_double))
    return d[:m], s[:m], row


def makePL(z, n, Bchain, nen, seg_count, normalise):
    ft = FlatTree(z)
    Bc = np.ascontiguousarray(Bchain, dtype=np.float64)
    nen = np.ascontiguousarray(nen, dtype=np.int32)
    sc = np.ascontiguousarray(seg_count, dtype=np.int32)
    PL = np.zeros((2 * ft.T - 1, n))
    rc = lib().orc_makePL(C.byref(ft.c), n, _ptr(Bc, C.c_double), _ptr(nen, C.c_int32), _ptr(sc, C.c_int32),
                          int(normalise), _ptr(PL, C.c_double))
    return PL, rc
