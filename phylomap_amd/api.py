"""R-level API of phylomap, mirrored in Python (the reference's R toolchain is not available here).

Same names, argument order and meaning as the R wrappers: ``sumstatMCMC(z,Q,pid,Omega,N)``
(R/sumstatMCMC.R:21-29), ``sumstatMCMC_bigtree`` (R/sumstatMCMC_bigtree.R), ``SPARSEsumstatMCMC``
(R/SPARSEsumstatMCMC.R:21-29) and ``sumstatEXP(z,Q,pid,N)`` (R/sumstatEXP.R:21-33).  Each computes the
derived arguments exactly where the R wrapper does (``nen``, ``nodelist``, ``root``, ``B``; EXP:
``eigen``/``solve``) and then calls the C-ABI entry point that replaces the corresponding ``.Call``
(R/RcppExports.R:4-22).  Returns the N x (n + n(n-1)) matrix of man/sumstatMCMC.Rd:18.

Extra keyword arguments (``seed``, ``n_replicas``, ...) map onto ``phm_options``; with the defaults a
call is a drop-in for the R function (one chain, one tip vector).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .treeorder import makenodelist, myreorder, pruningwiseedgeorder  # noqa: F401  (Python twins of phm_tree_orders)


def _mcmc(fn_name, z, Q, pid, Omega, N, sites=None, **opt):
    """``sites``: optional S x n_tips matrix of 1-based tip states -- S sites of an alignment on the same tree, one chain each
    (``n_replicas = S``, ``tips_per_replica``); the initial paths of ``z`` must be compatible with every site (e.g. internal
    segments in a state from which every tip state is reachable)."""
    L = _lib.load()
    if sites is not None:
        sites = np.ascontiguousarray(np.asarray(sites).round(), dtype=np.int32)
        opt = dict(opt, n_replicas=sites.shape[0], tips_per_replica=True)
    Q = np.asfortranarray(np.asarray(Q, dtype=np.float64))
    n = Q.shape[0]
    nen, nodelist, root = _lib.tree_orders(z)                                 # R/sumstatMCMC.R:22-24, native O(E)
    B = np.asfortranarray(np.eye(n) + Q / Omega)                              # :25
    pid = np.ascontiguousarray(pid, dtype=np.float64)
    ft = _lib.FlatTree(z, sites)
    o = _lib.make_options(**opt)
    S = max(1, int(o.n_replicas))
    cols = n + n * (n - 1)
    if fn_name == "phm_maketreelistMCMCks_sweep":
        cols = n + n * n + 2 + 3 * (n // 2 - 1) + 1                         # man/sumstatMCMCks.Rd:19
    if fn_name == "phm_maketreelistMCMCbf_sweep":
        cols = n + n * n + 3                                                  # dwell, n x n counts, Q[0,1], Q[1,0], root state
    single = bool(o.reduce) or S == 1
    out = np.zeros((N, cols), order="F") if single else np.zeros((S, cols, N))
    st = getattr(L, fn_name)(C.byref(ft.c), n, _lib._p(Q, C.c_double), _lib._p(pid, C.c_double),
                             _lib._p(B, C.c_double), float(Omega), _lib._p(nen, C.c_int32),
                             _lib._p(nodelist, C.c_int32), root, int(N), C.byref(o), _lib._p(out, C.c_double))
    _lib.check(st)
    return out if single else out.transpose(0, 2, 1)


def sumstatMCMC(z, Q, pid, Omega, N, **opt):
    """R/sumstatMCMC.R:21-29 -> phm_maketreelistMCMC."""
    return _mcmc("phm_maketreelistMCMC", z, Q, pid, Omega, N, **opt)


def sumstatMCMC_bigtree(z, Q, pid, Omega, N, **opt):
    """R/sumstatMCMC_bigtree.R -> phm_maketreelistMCMC_bigtree (row-normalised partial likelihoods)."""
    return _mcmc("phm_maketreelistMCMC_bigtree", z, Q, pid, Omega, N, **opt)


def SPARSEsumstatMCMC(z, Q, pid, Omega, N, **opt):
    """R/SPARSEsumstatMCMC.R:21-29 -> phm_SPARSEmaketreelistMCMC."""
    return _mcmc("phm_SPARSEmaketreelistMCMC", z, Q, pid, Omega, N, **opt)


def sumstatMCMCks_sweep(z, Q, pid, Omega, N, **opt):
    """The tree sweep of ``sumstatMCMCks`` (R/sumstatMCMCks.R, src/phylomap.cpp:1802-1872) with Q held FIXED:
    hidden-rates Q of even size (``synth.make2sQ``), tips observed only up to parity and re-sampled every sweep,
    n x n transition counters including self pairs, result layout of man/sumstatMCMCks.Rd:19.  ``sumstatMCMCks`` below
    adds the per-iteration Gibbs/MH updates of Q (src/phylomap.cpp:1862-1866) and is the drop-in for the R function."""
    return _mcmc("phm_maketreelistMCMCks_sweep", z, Q, pid, Omega, N, **opt)


def sumstatMCMCbf_sweep(z, Q, pid, Omega, N, **opt):
    """The tree sweep of ``sumstatMCMCbf`` (treesamplebf, src/phylomap.cpp:1169-1179) with Q held FIXED, for ANY number of
    states: tips observed, row-normalised pruning, every consecutive pair of segment states counted -- self pairs, i.e.
    virtual jumps, included (shortenerbf :1010-1014) -- into n x n counters.  Columns: n dwell sums, n*n counts (row-major
    from, to), Q[0,1], Q[1,0], root state (0-based); at n = 2 that is the layout of R/sumstatMCMCbf.R:33."""
    return _mcmc("phm_maketreelistMCMCbf_sweep", z, Q, pid, Omega, N, **opt)


def _qupdate(fn_name, z, Q, pid, Omega, N, prior, cols, **opt):
    L = _lib.load()
    Q = np.asfortranarray(np.asarray(Q, dtype=np.float64))
    n = Q.shape[0]
    nen, nodelist, root = _lib.tree_orders(z)
    B = np.asfortranarray(np.eye(n) + Q / Omega)
    pid = np.ascontiguousarray(pid, dtype=np.float64)
    prior = np.ascontiguousarray(prior, dtype=np.float64)
    ft = _lib.FlatTree(z)
    o = _lib.make_options(**opt)
    out = np.zeros((N, cols), order="F")
    st = getattr(L, fn_name)(C.byref(ft.c), n, _lib._p(Q, C.c_double), _lib._p(pid, C.c_double), _lib._p(B, C.c_double),
                             float(Omega), _lib._p(nen, C.c_int32), _lib._p(nodelist, C.c_int32), root, int(N),
                             _lib._p(prior, C.c_double), int(prior.size), C.byref(o), _lib._p(out, C.c_double))
    _lib.check(st)
    return out


def sumstatMCMCbf(z, Q, pid, Omega, N, prior, **opt):
    """R/sumstatMCMCbf.R:19-35 -> phm_maketreelistMCMCbf: two-state model, rates re-drawn after every sweep.
    Columns: time 0, time 1, n00, n01, n10, n11, l01, l10, root_state (R/sumstatMCMCbf.R:33).  ``Q`` is not modified."""
    return _qupdate("phm_maketreelistMCMCbf", z, Q, pid, Omega, N, prior, 9, **opt)


def sumstatMCMCks(z, Q, pid, Omega, N, prior, **opt):
    """R/sumstatMCMCks.R:19-33 -> phm_maketreelistMCMCks: hidden-rates model with k regimes (n = 2k+2), all 2+3k rate
    parameters updated after every sweep.  Layout man/sumstatMCMCks.Rd:19.  ``Q`` is not modified."""
    n = np.asarray(Q).shape[0]
    return _qupdate("phm_maketreelistMCMCks", z, Q, pid, Omega, N, prior, n + n * n + 2 + 3 * (n // 2 - 1) + 1, **opt)


def sumstatMCMC2sDICt(z, Q, pid, Omega, N, prior, **opt):
    """R/sumstatMCMC2sDICt.R -> phm_maketreelistMCMC2sDICt: ``sumstatMCMCbf`` plus log p(y|Q) (matrix exponentiation) per
    iteration; columns time 0, time 1, n00, n01, n10, n11, l01, l10, root_state, log(p(y|Q))."""
    return _qupdate("phm_maketreelistMCMC2sDICt", z, Q, pid, Omega, N, prior, 10, **opt)


def sumstatMCMCksDICt(z, Q, pid, Omega, N, prior, **opt):
    """R/sumstatMCMCksDICt.R -> phm_maketreelistMCMCksDICt: ``sumstatMCMCks`` plus log p(y|Q) per iteration (last column)."""
    n = np.asarray(Q).shape[0]
    return _qupdate("phm_maketreelistMCMCksDICt", z, Q, pid, Omega, N, prior, n + n * n + 2 + 3 * (n // 2 - 1) + 2, **opt)


def _qupdate_mt(fn_name, treelist, Q, pid, Omega, N, prior, cols, **opt):
    L = _lib.load()
    Q = np.asfortranarray(np.asarray(Q, dtype=np.float64))
    n = Q.shape[0]
    orders = [_lib.tree_orders(z) for z in treelist]                 # R/sumstatMCMCmt.R:37-46, one row per tree
    nen_m = np.asfortranarray(np.stack([o[0] for o in orders]), dtype=np.int32)
    nodelist_m = np.asfortranarray(np.stack([o[1] for o in orders]), dtype=np.int32)
    roots = np.ascontiguousarray([o[2] for o in orders], dtype=np.int32)
    B = np.asfortranarray(np.eye(n) + Q / Omega)
    pid = np.ascontiguousarray(pid, dtype=np.float64)
    prior = np.ascontiguousarray(prior, dtype=np.float64)
    ftl = _lib.FlatTreeList(treelist)
    o = _lib.make_options(**opt)
    out = np.zeros((N, cols), order="F")
    st = getattr(L, fn_name)(ftl.c, ftl.n, n, _lib._p(Q, C.c_double), _lib._p(pid, C.c_double), _lib._p(B, C.c_double),
                             float(Omega), _lib._p(nen_m, C.c_int32), _lib._p(nodelist_m, C.c_int32),
                             _lib._p(roots, C.c_int32), int(N), _lib._p(prior, C.c_double), int(prior.size), C.byref(o),
                             _lib._p(out, C.c_double))
    _lib.check(st)
    return out


def sumstatMCMCmt(treelist, Q, pid, Omega, N, prior, **opt):
    """R/sumstatMCMCmt.R:29-52 -> phm_maketreelistMCMCmt: two-state model over a list of trees (same tips, different
    topologies / branch lengths); every iteration sweeps every tree, keeps one drawn uniformly and updates the rates from
    it.  Columns: time_0, time_1, n00, n01, n10, n11, l01, l10, tree_number (0-based, as the reference stores it)."""
    return _qupdate_mt("phm_maketreelistMCMCmt", treelist, Q, pid, Omega, N, prior, 9, **opt)


def sumstatMCMCksmt(treelist, Q, pid, Omega, N, prior, **opt):
    """R/sumstatMCMCksmt.R -> phm_maketreelistMCMCksmt: hidden-rates model (n = 2k+2) over a list of trees; ``prior`` has
    8 entries (l01, l10, kappa, gamma shape/rate pairs).  Columns as ``sumstatMCMCks`` with tree_number last."""
    n = np.asarray(Q).shape[0]
    return _qupdate_mt("phm_maketreelistMCMCksmt", treelist, Q, pid, Omega, N, prior, n + n * n + 2 + 3 * (n // 2 - 1) + 1, **opt)


def eigen_decompose(Q):
    """R/sumstatEXP.R:26-29: lefts = eigen(Q)$vectors, rights = solve(lefts), d = diag(values) (real spectrum only)."""
    vals, vecs = np.linalg.eig(np.asarray(Q, dtype=np.float64))
    if np.max(np.abs(np.imag(vals))) > 0:
        raise ValueError("Q has complex eigenvalues; the reference's matexp handles a real spectrum only")
    lefts = np.real(vecs)
    rights = np.linalg.solve(lefts, np.eye(lefts.shape[0]))
    return lefts, rights, np.diag(np.real(vals))


def sumstatEXP(z, Q, pid, N, eig=None, **opt):
    """R/sumstatEXP.R:21-33 -> phm_maketreelistEXP.  ``eig`` = (lefts, rights, d) overrides the eigendecomposition
    R/sumstatEXP.R:26-29 computes (LAPACK results differ between machines in the last bits)."""
    L = _lib.load()
    Q = np.asfortranarray(np.asarray(Q, dtype=np.float64))
    n = Q.shape[0]
    nen, nodelist, root = _lib.tree_orders(z)
    lefts, rights, d = (np.asfortranarray(np.asarray(a, dtype=np.float64)) for a in (eigen_decompose(Q) if eig is None else eig))
    pid = np.ascontiguousarray(pid, dtype=np.float64)
    ft = _lib.FlatTree(z)
    o = _lib.make_options(**opt)
    out = np.zeros((N, n + n * (n - 1)), order="F")
    st = L.phm_maketreelistEXP(C.byref(ft.c), n, _lib._p(Q, C.c_double), _lib._p(pid, C.c_double),
                               _lib._p(nen, C.c_int32), _lib._p(nodelist, C.c_int32), root, int(N),
                               _lib._p(lefts, C.c_double), _lib._p(rights, C.c_double), _lib._p(d, C.c_double),
                               C.byref(o), _lib._p(out, C.c_double))
    _lib.check(st)
    return out


def expm_eigen(lefts, rights, d, t, device=-1, mfma=False):
    """Batched P_b = |L diag(exp(d t_b)) R| (matexp, src/phylomap.cpp:2964-2968). Returns (P[n_t,n,n], kernel_ms).
    ``mfma=True`` (16 < n <= 64) runs the product on the matrix cores (last-bit differences)."""
    L = _lib.load()
    lefts, rights, d = (np.asfortranarray(np.asarray(a, dtype=np.float64)) for a in (lefts, rights, d))
    t = np.ascontiguousarray(t, dtype=np.float64)
    n = lefts.shape[0]
    out = np.zeros((t.size, n, n))
    ms = C.c_double(0.0)
    fn = L.phm_expm_eigen_mfma if mfma else L.phm_expm_eigen
    _lib.check(fn(n, _lib._p(lefts, C.c_double), _lib._p(rights, C.c_double), _lib._p(d, C.c_double),
                                _lib._p(t, C.c_double), int(t.size), int(device), _lib._p(out, C.c_double), C.byref(ms)))
    return out, ms.value


def expm_pade(Q, t, device=-1, mfma=False):
    """Batched expmat(Q t_b), Pade(6) scaling-and-squaring. Returns (P[n_t,n,n], kernel_ms).
    ``mfma=True`` (16 < n <= 64): every matrix product on the matrix cores (agrees to rounding, not bit for bit)."""
    L = _lib.load()
    Q = np.asfortranarray(np.asarray(Q, dtype=np.float64))
    t = np.ascontiguousarray(t, dtype=np.float64)
    n = Q.shape[0]
    out = np.zeros((t.size, n, n))
    ms = C.c_double(0.0)
    fn = L.phm_expm_pade_mfma if mfma else L.phm_expm_pade
    _lib.check(fn(n, _lib._p(Q, C.c_double), _lib._p(t, C.c_double), int(t.size), int(device),
                               _lib._p(out, C.c_double), C.byref(ms)))
    return out, ms.value
