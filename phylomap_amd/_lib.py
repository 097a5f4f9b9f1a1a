"""ctypes binding of the C-ABI (include/phylomap_hip.h).  The library is built in-tree by
``__graft_entry__.build()`` / ``make``; there is no fallback when it is missing."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PHM_LIB", os.path.join(_HERE, "libphylomap_hip.so"))   # PHM_LIB: developer override

PHM_OK = 0
STATUS = {0: "PHM_OK", 1: "PHM_ERR_BAD_INPUT", 2: "PHM_ERR_UNSUPPORTED", 3: "PHM_ERR_NO_DEVICE", 4: "PHM_ERR_OOM",
          5: "PHM_ERR_ZERO_PROB", 6: "PHM_ERR_CAPACITY", 7: "PHM_ERR_UNIF_CAP", 8: "PHM_ERR_STATE"}
PHM_MCMC, PHM_MCMC_BIGTREE, PHM_MCMC_SPARSE, PHM_MCMC_KS, PHM_MCMC_BF, PHM_MCMC_MT, PHM_MCMC_KSMT = 0, 1, 2, 3, 4, 5, 6

EXPORTS = [
    "phm_version", "phm_struct_size", "phm_device_count", "phm_last_error", "phm_status_string",
    "phm_maketreelistMCMC", "phm_maketreelistMCMC_bigtree", "phm_SPARSEmaketreelistMCMC", "phm_maketreelistEXP",
    "phm_maketreelistMCMCks_sweep", "phm_maketreelistMCMCbf_sweep", "phm_maketreelistMCMCbf", "phm_maketreelistMCMCks", "phm_maketreelistMCMC2sDICt", "phm_maketreelistMCMCksDICt", "phm_engine_set_model", "phm_qupdate_apply",
    "phm_expm_eigen", "phm_expm_eigen_mfma", "phm_expm_pade", "phm_expm_pade_mfma",
    "phm_engine_create", "phm_engine_run", "phm_engine_sync", "phm_engine_read_stats", "phm_engine_dump",
    "phm_engine_info", "phm_engine_destroy", "phm_engine_reduced_stats_device",
    "phm_engine_time_pruning", "phm_tree_orders",
    "phm_engine_create_multi", "phm_maketreelistMCMCmt", "phm_maketreelistMCMCksmt", "phm_engine_phase_ms",
    "phm_last_kernel_ms", "phm_set_debug_options", "phm_sparse_kernel_source",
]


class PhmError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"{STATUS.get(status, status)}: {message}")
        self.status = status


class Tree(C.Structure):
    _fields_ = [("n_tips", C.c_int32), ("n_node", C.c_int32), ("n_edge", C.c_int32),
                ("edge", C.POINTER(C.c_int32)), ("edge_length", C.POINTER(C.c_double)),
                ("states", C.POINTER(C.c_int32)), ("map_off", C.POINTER(C.c_int32)),
                ("maps", C.POINTER(C.c_double)), ("mapnames", C.POINTER(C.c_int32))]


class Model(C.Structure):
    _fields_ = [("n_states", C.c_int32), ("Q", C.POINTER(C.c_double)), ("pid", C.POINTER(C.c_double)),
                ("B", C.POINTER(C.c_double)), ("Omega", C.c_double), ("variant", C.c_int32)]


PHM_MAX_DEVICES = 8


class Options(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("n_replicas", C.c_int32), ("replica_offset", C.c_int32),
                ("reduce", C.c_int32), ("tips_per_replica", C.c_int32), ("device", C.c_int32),
                ("iters_per_launch", C.c_int32), ("cap_tail", C.c_double), ("mapping", C.c_int32), ("storage", C.c_int32),
                ("rescale_pruning", C.c_int32), ("no_recovery", C.c_int32), ("sparse_chains", C.c_int32),
                ("n_devices", C.c_int32), ("devices", C.c_int32 * PHM_MAX_DEVICES), ("reserved", C.c_int32 * 3)]


class DebugOptions(C.Structure):
    """phm_debug_options: measurement / test aids, per thread (phm_set_debug_options)."""
    _fields_ = [("pruning_form", C.c_int32), ("phase_timing", C.c_int32), ("fail_recovery", C.c_int32),
                ("branch_group", C.c_int32), ("level_groups", C.c_int32), ("q_timing", C.c_int32),
                ("pade_pivot_min", C.c_double), ("reserved", C.c_int32 * 4)]


class Info(C.Structure):
    _fields_ = [("n_states", C.c_int32), ("n_edge", C.c_int32), ("n_replicas", C.c_int32),
                ("n_replicas_padded", C.c_int32), ("n_cols", C.c_int32), ("max_iters", C.c_int32),
                ("device_bytes", C.c_int64), ("rows_per_replica", C.c_int64), ("seg_read", C.c_int64),
                ("seg_written", C.c_int64), ("last_run_ms", C.c_double), ("last_run_launches", C.c_int32),
                ("iters_done", C.c_int32), ("recoveries", C.c_int32), ("mapping", C.c_int32), ("sparse_chains", C.c_int32),
                ("reserved", C.c_int32)]


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t)) if a is not None else None


_lib = None


def load():
    """Load libphylomap_hip.so; raises if it has not been built (no CPU fallback exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "or `make`; phylomap_amd has no CPU fallback")
        L = C.CDLL(LIB_PATH)
        L.phm_last_error.restype = C.c_char_p
        L.phm_status_string.restype = C.c_char_p
        L.phm_status_string.argtypes = [C.c_int32]
        L.phm_last_kernel_ms.restype = C.c_double
        L.phm_engine_create.argtypes = [C.POINTER(Tree), C.POINTER(Model), C.POINTER(Options), C.c_int32,
                                        C.POINTER(C.c_void_p)]
        L.phm_engine_create_multi.argtypes = [C.POINTER(Tree), C.c_int32, C.POINTER(Model), C.POINTER(Options), C.c_int32,
                                              C.POINTER(C.c_void_p)]
        L.phm_engine_run.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
        L.phm_engine_sync.argtypes = [C.c_void_p]
        L.phm_engine_read_stats.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_double)]
        L.phm_engine_dump.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_double), C.c_int32,
                                      C.POINTER(C.c_int32), C.POINTER(C.c_double)]
        L.phm_engine_info.argtypes = [C.c_void_p, C.POINTER(Info)]
        L.phm_tree_orders.argtypes = [C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                      C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.phm_engine_time_pruning.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.POINTER(C.c_double)]
        L.phm_engine_reduced_stats_device.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p,
                                                      C.POINTER(C.c_void_p)]
        L.phm_engine_phase_ms.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        L.phm_engine_destroy.argtypes = [C.c_void_p]
        L.phm_engine_destroy.restype = None
        mc = [C.POINTER(Tree), C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double),
              C.c_double, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int32, C.c_int32, C.POINTER(Options),
              C.POINTER(C.c_double)]
        L.phm_maketreelistMCMC.argtypes = mc
        L.phm_maketreelistMCMC_bigtree.argtypes = mc
        L.phm_SPARSEmaketreelistMCMC.argtypes = mc
        L.phm_maketreelistMCMCks_sweep.argtypes = mc
        L.phm_maketreelistMCMCbf_sweep.argtypes = mc
        mcq = mc[:10] + [C.POINTER(C.c_double), C.c_int32] + mc[10:]
        L.phm_maketreelistMCMCbf.argtypes = mcq
        L.phm_maketreelistMCMCks.argtypes = mcq
        L.phm_maketreelistMCMC2sDICt.argtypes = mcq
        L.phm_maketreelistMCMCksDICt.argtypes = mcq
        mt = [C.POINTER(Tree), C.c_int32, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double),
              C.c_double, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int32,
              C.POINTER(C.c_double), C.c_int32, C.POINTER(Options), C.POINTER(C.c_double)]
        L.phm_maketreelistMCMCmt.argtypes = mt
        L.phm_maketreelistMCMCksmt.argtypes = mt
        L.phm_engine_set_model.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        L.phm_qupdate_apply.argtypes = [C.c_int32, C.c_int32, C.POINTER(C.c_double), C.c_double, C.POINTER(C.c_double),
                                        C.c_int32, C.POINTER(C.c_double), C.c_uint64, C.c_uint32]
        L.phm_maketreelistEXP.argtypes = [C.POINTER(Tree), C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                          C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int32, C.c_int32,
                                          C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                          C.POINTER(Options), C.POINTER(C.c_double)]
        L.phm_expm_eigen.argtypes = [C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                     C.POINTER(C.c_double), C.c_int32, C.c_int32, C.POINTER(C.c_double),
                                     C.POINTER(C.c_double)]
        L.phm_expm_eigen_mfma.argtypes = L.phm_expm_eigen.argtypes
        L.phm_expm_pade_mfma.argtypes = [C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int32, C.c_int32,
                                         C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.phm_expm_pade.argtypes = [C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int32, C.c_int32,
                                    C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.phm_set_debug_options.argtypes = [C.POINTER(DebugOptions)]
        L.phm_sparse_kernel_source.argtypes = [C.c_int32, C.POINTER(C.c_double), C.c_char_p, C.c_int32]
        for which, mirror in enumerate((Options, Info, Tree, Model, DebugOptions)):      # the ctypes mirrors must match the C layout
            if L.phm_struct_size(which) != C.sizeof(mirror):
                raise RuntimeError(f"{LIB_PATH}: {mirror.__name__} is {L.phm_struct_size(which)} bytes in the library, "
                                   f"{C.sizeof(mirror)} in phylomap_amd/_lib.py (stale build? run make)")
        _lib = L
    return _lib


def check(status):
    if status != PHM_OK:
        raise PhmError(status, load().phm_last_error().decode())


class FlatTree:
    """Flattens the R-list-shaped tree object ``z`` into the plain arrays of ``phm_tree`` (and keeps them alive)."""

    def __init__(self, z, states=None):
        edge = np.asarray(z["edge"], dtype=np.int32)
        if edge.ndim != 2 or edge.shape[1] != 2:
            raise ValueError("z['edge'] must be E x 2")
        self.E = edge.shape[0]
        st = z["states"] if states is None else states
        self.states = np.ascontiguousarray(np.asarray(st).round(), dtype=np.int32)   # R may hand doubles (:910)
        self.T = int(np.asarray(z["states"]).shape[-1])
        self.edge = np.asfortranarray(edge).reshape(-1, order="F").copy()
        el = z.get("edge.length")
        self.edge_length = None if el is None else np.ascontiguousarray(el, dtype=np.float64)
        lens = [len(m) for m in z["maps"]]
        self.map_off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        self.maps = np.ascontiguousarray(np.concatenate([np.asarray(m, dtype=np.float64) for m in z["maps"]]))
        self.mapnames = np.ascontiguousarray(np.concatenate([np.asarray(m) for m in z["mapnames"]]), dtype=np.int32)
        self.c = Tree(self.T, int(z["Nnode"]), self.E, _p(self.edge, C.c_int32), _p(self.edge_length, C.c_double),
                      _p(self.states.reshape(-1), C.c_int32), _p(self.map_off, C.c_int32), _p(self.maps, C.c_double),
                      _p(self.mapnames, C.c_int32))


class FlatTreeList:
    """A list of phylomap trees (R/sumstatMCMCmt.R:33) as one contiguous array of ``phm_tree``."""

    def __init__(self, treelist):
        if len(treelist) < 1:
            raise ValueError("treelist is empty")
        self.trees = [FlatTree(z) for z in treelist]
        self.n = len(self.trees)
        self.c = (Tree * self.n)(*[t.c for t in self.trees])
        self.E, self.T = self.trees[0].E, self.trees[0].T


def tree_orders(z):
    """(nen, nodelist, root) computed natively in O(E): phm_tree_orders, the replacement of R/sumstatMCMC.R:1-18."""
    edge = np.asarray(z["edge"], dtype=np.int32)
    E = edge.shape[0]
    flat = np.asfortranarray(edge).reshape(-1, order="F").copy()
    T = int(z["Nnode"]) + 1
    nen = np.zeros(E, dtype=np.int32)
    nodelist = np.zeros(max(int(z["Nnode"]) - 1, 1), dtype=np.int32)
    root = C.c_int32(0)
    check(load().phm_tree_orders(T, E, _p(flat, C.c_int32), _p(nen, C.c_int32), _p(nodelist, C.c_int32), C.byref(root)))
    return nen, nodelist[: int(z["Nnode"]) - 1], int(root.value)


def sparse_kernel_source(M):
    """HIP source of the pruning kernel the library generates for the non-zero pattern of the chain matrix ``M`` (phm_rtc.h)."""
    Mf = np.asfortranarray(np.asarray(M, dtype=np.float64))
    L = load()
    need = L.phm_sparse_kernel_source(Mf.shape[0], _p(Mf, C.c_double), None, 0)
    if need < 0:
        check(1)
    buf = C.create_string_buffer(need)
    L.phm_sparse_kernel_source(Mf.shape[0], _p(Mf, C.c_double), buf, need)
    return buf.value.decode()


MAPPING = {"auto": 0, "replicas": 1, "branches": 2, "tiles": 3}


def set_debug_options(**kw):
    """Install this thread's phm_debug_options (measurement / test aids; no arguments = defaults): pruning_form, phase_timing,
    fail_recovery, branch_group, level_groups, q_timing, pade_pivot_min."""
    d = DebugOptions()
    for k, v in kw.items():
        setattr(d, k, float(v) if k == "pade_pivot_min" else int(v))
    check(load().phm_set_debug_options(C.byref(d)))


_DEBUG_KEYS = ("pruning_form", "phase_timing", "fail_recovery", "branch_group", "level_groups", "q_timing", "pade_pivot_min")


def make_options(seed=0, n_replicas=1, replica_offset=0, reduce=False, tips_per_replica=False, device=-1,
                 iters_per_launch=0, cap_tail=0.0, storage=0, mapping="auto", rescale=False, recover=True,
                 sparse_chains=0, devices=None, **debug):
    """``mapping``: how a sweep is laid over the lanes -- "replicas" (one lane per chain: the throughput layout for many
    replicas), "branches" (one chain or a handful in latency form, n <= 4; one wave per (replica, branch) for 5..64 states: few chains on a large
    tree), "tiles" (lanes = replicas, one wave per tile of 64 replicas and branch: 10^2 .. 10^5 replicas) or "auto".
    ``sparse_chains``: 5..64 states, "tiles": 0 automatic, 1 chains over the non-zeros of B only, 2 dense (matrix cores).
    ``devices``: an int D (GPUs 0..D-1) or a list of HIP ordinals -- the one-shot calls shard the replicas (sumstatEXP: the
    samples) over them (phm_options.n_devices).  Remaining keywords (``phase_timing``, ``pruning_form``, ...) are
    phm_debug_options and are installed for this thread as a side effect (defaults when none is given)."""
    unknown = set(debug) - set(_DEBUG_KEYS)
    if unknown:
        raise TypeError(f"unknown option(s) {sorted(unknown)}")
    set_debug_options(**debug)
    o = Options()
    o.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    o.n_replicas, o.replica_offset, o.reduce = int(n_replicas), int(replica_offset), int(bool(reduce))
    o.storage = int(storage)          # 0 automatic, 1 ring, 2 two buffers
    o.mapping = MAPPING[mapping] if isinstance(mapping, str) else int(mapping)
    o.rescale_pruning = int(bool(rescale))      # sumstatEXP / sumstatMCMC / SPARSEsumstatMCMC: rescaled pruning pass
    o.no_recovery = 0 if recover else 1         # capacity recovery (rebuild with doubled slots + replay)
    o.sparse_chains = int(sparse_chains)
    o.tips_per_replica, o.device, o.iters_per_launch, o.cap_tail = int(bool(tips_per_replica)), int(device), int(iters_per_launch), float(cap_tail)
    if devices is not None:
        devs = list(range(int(devices))) if np.isscalar(devices) else [int(d) for d in devices]
        if len(devs) > PHM_MAX_DEVICES:
            raise ValueError(f"at most {PHM_MAX_DEVICES} devices")
        o.n_devices = len(devs)
        for i, d in enumerate(devs):
            o.devices[i] = d
    return o


class Engine:
    """Resident engine: tree, model and chain state stay in HBM between run() calls."""

    def __init__(self, z, Q, pid, Omega, max_iters, variant=PHM_MCMC, B=None, states=None, **opt):
        L = load()
        self.Q = np.asfortranarray(np.asarray(Q, dtype=np.float64))
        self.n = self.Q.shape[0]
        self.pid = np.ascontiguousarray(pid, dtype=np.float64)
        self.B = None if B is None else np.asfortranarray(np.asarray(B, dtype=np.float64))
        self.opt = make_options(**opt)
        self.model = Model(self.n, _p(self.Q, C.c_double), _p(self.pid, C.c_double), _p(self.B, C.c_double),
                           float(Omega), int(variant))
        self.h = C.c_void_p()
        n_trees = 1
        if isinstance(z, (list, tuple)):      # a list of trees sharing the model: n_replicas chains per tree
            if states is not None:
                raise ValueError("a list of trees carries its own tip states")
            self.ft = FlatTreeList(z)
            n_trees = self.ft.n
            check(L.phm_engine_create_multi(self.ft.c, n_trees, C.byref(self.model), C.byref(self.opt), int(max_iters),
                                            C.byref(self.h)))
        else:
            self.ft = FlatTree(z, states)
            check(L.phm_engine_create(C.byref(self.ft.c), C.byref(self.model), C.byref(self.opt), int(max_iters),
                                      C.byref(self.h)))
        self.cols = self.n + self.n * (self.n - 1)
        if int(variant) in (PHM_MCMC_KS, PHM_MCMC_KSMT):
            self.cols = self.n + self.n * self.n + 2 + 3 * (self.n // 2 - 1) + 1
        if int(variant) in (PHM_MCMC_BF, PHM_MCMC_MT):
            self.cols = self.n + self.n * self.n + 3
        self.S = max(1, int(self.opt.n_replicas)) * n_trees      # tree-major: replica j * n_replicas + c
        self.reduce = bool(self.opt.reduce)

    def run(self, n_iters, stream=None):
        check(load().phm_engine_run(self.h, int(n_iters), C.c_void_p(stream) if stream else None))

    def set_model(self, Q):
        """Replace the rate matrix between sweeps (B = I + Q/Omega is recomputed; the chain state is kept)."""
        Qc = np.asfortranarray(np.asarray(Q, dtype=np.float64))
        check(load().phm_engine_set_model(self.h, _p(Qc, C.c_double)))

    def sync(self):
        check(load().phm_engine_sync(self.h))

    def info(self):
        i = Info()
        check(load().phm_engine_info(self.h, C.byref(i)))
        return i

    def stats(self, iter0, n):
        """reduce: (n, cols); else (S, n, cols).  Memory is R's column-major per matrix."""
        if self.reduce:
            out = np.zeros((n, self.cols), order="F")
            check(load().phm_engine_read_stats(self.h, int(iter0), int(n), _p(out, C.c_double)))
            return out
        buf = np.zeros((self.S, self.cols, n))
        check(load().phm_engine_read_stats(self.h, int(iter0), int(n), _p(buf, C.c_double)))
        return buf.transpose(0, 2, 1)

    def reduced_stats_device(self, iter0, n, stream=None):
        """Device pointer (int) to the (n, cols) row-major reduced statistics; see phm_engine_reduced_stats_device."""
        ptr = C.c_void_p()
        check(load().phm_engine_reduced_stats_device(self.h, int(iter0), int(n), C.c_void_p(stream) if stream else None,
                                                     C.byref(ptr)))
        return ptr.value

    def phase_ms(self):
        """(pruning levels, node draws, branch kernel, reductions): HIP-event ms of the last run, summed over its sweeps."""
        out = (C.c_double * 4)()
        check(load().phm_engine_phase_ms(self.h, out))
        return [float(v) for v in out]

    def time_pruning(self, n_iters, stream=None):
        """HIP-event milliseconds for n_iters repetitions of the pruning sweep alone (chain state untouched)."""
        ms = C.c_double(0.0)
        check(load().phm_engine_time_pruning(self.h, int(n_iters), C.c_void_p(stream) if stream else None, C.byref(ms)))
        return ms.value

    def dump(self, replica=0, seg_cap=512):
        E, T, n = self.ft.E, self.ft.T, self.n
        seg_count = np.zeros(E, dtype=np.int32)
        seg_dwell = np.zeros((E, seg_cap))
        node_states = np.zeros(2 * T - 1, dtype=np.int32)
        PL = np.zeros((2 * T - 1, n))
        check(load().phm_engine_dump(self.h, int(replica), _p(seg_count, C.c_int32), _p(seg_dwell, C.c_double),
                                     int(seg_cap), _p(node_states, C.c_int32), _p(PL, C.c_double)))
        return {"seg_count": seg_count, "seg_dwell": seg_dwell, "node_states": node_states, "PL": PL}

    def close(self):
        if self.h:
            load().phm_engine_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
