"""CPU tests of the host logic and of the C-ABI surface (no compute calls: there is no GPU here)."""
import os
import re

import numpy as np
import pytest

from phylomap_amd import _lib, api, synth, treeorder

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _ape_pruningwise_scan(edge, ntip):
    """Literal restatement of ape's published neworder_pruningwise scan (repeated passes over the cladewise edge
    table collecting nodes whose child edges are all ready).  Input must already be cladewise."""
    e1, e2 = [int(v) for v in edge[:, 0]], [int(v) for v in edge[:, 1]]
    E = len(e1)
    deg = {}
    for p in e1:
        deg[p] = deg.get(p, 0) + 1
    ready = [c <= ntip for c in e2]
    order = []
    root_deg = deg[ntip + 1]
    guard = 0
    while len(order) < E - root_deg:
        guard += 1
        assert guard < 10 * E
        n, node = 0, None
        for i in range(E):
            if not ready[i]:
                continue
            if n == 0:
                node, n = e1[i], 1
            elif e1[i] == node:
                n += 1
            else:
                node, n = e1[i], 1
            if n == deg[node] and node != ntip + 1:
                for j in range(i + 1):
                    if e2[j] == node:
                        ready[j] = True
                    if e1[j] == node and ready[j]:
                        order.append(j)
                        ready[j] = False
                n = 0
    order += [i for i in range(E) if ready[i]]
    return np.asarray(order)


@pytest.mark.parametrize("tips,seed", [(2, 1), (3, 2), (7, 3), (40, 4), (257, 5)])
def test_pruningwise_order_properties_and_scan_twin(tips, seed):
    edge, _ = synth.random_tree(tips, 1.0, seed)
    z = {"edge": edge, "Nnode": tips - 1}
    rows = treeorder.pruningwise_rows(edge)
    E = edge.shape[0]
    assert sorted(rows) == list(range(E))
    done = set(range(1, tips + 1))
    for i in range(tips - 1):                              # sibling pairs, children before parents (:503-514)
        a, b = rows[2 * i], rows[2 * i + 1]
        assert edge[a, 0] == edge[b, 0] and edge[a, 1] in done and edge[b, 1] in done
        done.add(int(edge[a, 0]))
    assert treeorder.myreorder(z) == tips + 1 == edge[rows[-1], 0]
    nl = treeorder.makenodelist(z)
    assert len(nl) == tips - 2 and len(set(nl)) == len(nl)
    seen = {tips + 1}
    parent = {int(c): int(p) for p, c in edge}
    for v in nl:                                           # parents before children (:640-657)
        assert parent[int(v)] in seen
        seen.add(int(v))
    np.testing.assert_array_equal(rows, _ape_pruningwise_scan(edge, tips))


@pytest.mark.parametrize("tips,seed,shuffle", [(2, 1, False), (9, 2, False), (300, 3, False), (50, 4, True), (2000, 5, True)])
def test_native_tree_orders_equal_python_twin(tips, seed, shuffle):
    edge, _ = synth.random_tree(tips, 1.0, seed)
    if shuffle:
        edge = edge[np.random.default_rng(seed).permutation(edge.shape[0])]
    z = {"edge": edge, "Nnode": tips - 1}
    nen, nodelist, root = _lib.tree_orders(z)
    np.testing.assert_array_equal(nen, treeorder.pruningwiseedgeorder(z))
    np.testing.assert_array_equal(nodelist, treeorder.makenodelist(z))
    assert root == treeorder.myreorder(z)


def test_pruningwise_on_shuffled_rows_maps_back():
    edge, _ = synth.random_tree(30, 1.0, 9)
    perm = np.random.default_rng(0).permutation(edge.shape[0])
    e2 = edge[perm]
    rows = treeorder.pruningwise_rows(e2)
    for i in range(29):
        assert e2[rows[2 * i], 0] == e2[rows[2 * i + 1], 0]
    assert e2[rows[-1], 0] == 31


def test_make2sQ_matches_definition():
    Q = synth.make2sQ(.1, .1, .2, .2, 10)
    want = np.array([[-.3, .1, .2, 0], [.1, -.3, 0, .2], [.2, 0, -1.2, 1.0], [0, .2, 1.0, -1.2]])
    np.testing.assert_allclose(Q, want, atol=1e-15)
    Q6 = synth.make2sQ(.1, .3, [.2, .4], [.5, .6], [2, 3])
    assert Q6.shape == (6, 6) and np.allclose(Q6.sum(1), 0)
    assert Q6[2, 4] == .4 and Q6[4, 2] == .6 and Q6[4, 5] == 3 * .1 and Q6[5, 4] == 3 * .3


def test_synthetic_configs_are_deterministic():
    z1, Q, pid, Om = synth.config_problem(2, n_tips=50)
    z2, *_ = synth.config_problem(2, n_tips=50)
    np.testing.assert_array_equal(z1["edge"], z2["edge"])
    np.testing.assert_array_equal(z1["edge.length"], z2["edge.length"])
    np.testing.assert_array_equal(z1["states"], z2["states"])
    assert z1["edge"][0, 0] == 51 and z1["edge"].shape == (98, 2)
    assert abs(Om - 1.25 * 1.2) < 1e-15
    assert all(len(m) == 2 for m in z1["maps"])


@pytest.mark.parametrize("n,mt", [(2, 0), (4, 0), (6, 0), (8, 0), (2, 1), (4, 1), (8, 1)])
def test_rate_matrix_updates_product_equals_oracle(n, mt):
    """The host glue of sumstatMCMCbf / sumstatMCMCks (phm_qupdate.cpp) against the oracle's transcription of
    updatel01/l10 (src/phylomap.cpp:1189-1253) and the five ks updates (:1435-1785), and of their multi-tree twins
    (updatel01mtNS :2192-2262, update*mt :2371-2705): same Philox stream, same Marsaglia-Tsang gamma, same libm ->
    bit-identical Q after every update."""
    import ctypes as C
    import oracle_lib as O
    rs = np.random.default_rng(n + 100 * mt)
    variant = (_lib.PHM_MCMC_MT if mt else _lib.PHM_MCMC_BF) if n == 2 else (_lib.PHM_MCMC_KSMT if mt else _lib.PHM_MCMC_KS)
    k = n // 2 - 1
    Q = synth.config_Q(1) if n == 2 else synth.make2sQ(.1, .3, rs.uniform(.1, .4, k), rs.uniform(.1, .4, k), rs.uniform(1, 5, k))
    prior = np.array([.55, 1, .56, 1.01]) if n == 2 else np.array([1., 10, 2, 10, 20, 2])
    if mt and n > 2:
        prior = np.array([1., 10, 1.5, 11, 2, 10, 20, 2])
    Omega = 12.0
    changed = 0
    for it in range(40):
        row = np.concatenate([rs.uniform(5, 60, n), rs.integers(0, 40, n * n).astype(float)])
        Qp = np.asfortranarray(Q.copy())
        _lib.check(_lib.load().phm_qupdate_apply(variant, n, _lib._p(Qp, C.c_double), Omega, _lib._p(prior, C.c_double),
                                                 prior.size, _lib._p(row, C.c_double), 1234567890123, it))
        Qo = np.ascontiguousarray(Q.copy())
        rc = O.lib().orc_qupdate_apply(int(variant), n, Qo.ctypes.data_as(C.POINTER(C.c_double)), C.c_double(Omega),
                                       prior.ctypes.data_as(C.POINTER(C.c_double)), row.ctypes.data_as(C.POINTER(C.c_double)),
                                       C.c_uint32(1234567890123 & 0xFFFFFFFF), C.c_uint32(1234567890123 >> 32), C.c_uint32(it))
        assert rc == 0
        np.testing.assert_array_equal(np.asarray(Qp), Qo)
        np.testing.assert_allclose(Qo.sum(1), 0.0, atol=1e-12)          # still a rate matrix
        assert np.all(Qo - np.diag(np.diag(Qo)) >= 0)
        changed += int(not np.array_equal(Qo, Q))
        Q = Qo
    assert changed >= 5                                                  # proposals do get accepted (random rows reject often)


def test_cabi_exports_every_declared_symbol():
    import ctypes as C
    hdr = open(os.path.join(ROOT, "include", "phylomap_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(phm_[A-Za-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 17
    L = _lib.load()
    for name in declared:
        assert hasattr(L, name), f"{name} declared in include/phylomap_hip.h but not exported"
    assert sorted(_lib.EXPORTS) == declared
    assert L.phm_version() == 300
    assert [L.phm_struct_size(i) for i in range(6)] == [C.sizeof(_lib.Options), C.sizeof(_lib.Info), C.sizeof(_lib.Tree), C.sizeof(_lib.Model),
                                                        C.sizeof(_lib.DebugOptions), -1]
    # the public options struct carries no developer aids and the library reads no environment variable
    opt_fields = [f[0] for f in _lib.Options._fields_]
    assert "n_devices" in opt_fields and "devices" in opt_fields
    assert not {"pruning_form", "phase_timing", "capacity_boost_log2"} & set(opt_fields)
    for f in os.listdir(os.path.join(ROOT, "phylomap_amd", "csrc")):
        assert "getenv" not in open(os.path.join(ROOT, "phylomap_amd", "csrc", f)).read(), f
    assert L.phm_status_string(6).decode() == "branch capacity exceeded"


def test_no_cpu_fallback():
    """Without a GPU every compute entry point must fail loudly, never compute on the host."""
    if _lib.load().phm_device_count() > 0:
        pytest.skip("GPU present")
    z, Q, pid, Om = synth.config_problem(1, n_tips=8)
    for fn in (api.sumstatMCMC, api.sumstatMCMC_bigtree, api.SPARSEsumstatMCMC):
        with pytest.raises(_lib.PhmError) as e:
            fn(z, Q, pid, Om, 3)
        assert e.value.status == 3
    with pytest.raises(_lib.PhmError):
        api.sumstatEXP(z, Q, pid, 3)
    with pytest.raises(_lib.PhmError):
        api.expm_pade(Q, [1.0])


def test_input_validation_happens_before_the_device():
    z, Q, pid, Om = synth.config_problem(1, n_tips=8)
    with pytest.raises(_lib.PhmError) as e:
        api.sumstatMCMC(z, Q, pid, 0.01, 3)                 # Omega < |q_ii|
    assert e.value.status == 1
    zb = dict(z)
    zb["states"] = z["states"].copy()
    zb["states"][0] = 7
    with pytest.raises(_lib.PhmError) as e:
        api.sumstatMCMC(zb, Q, pid, Om, 3)
    assert e.value.status == 1
    zb = dict(z)
    zb["edge"] = z["edge"].copy()
    zb["edge"][2, 0] = zb["edge"][0, 0]                     # three children on one node
    with pytest.raises((_lib.PhmError, ValueError)):
        api.sumstatMCMC(zb, Q, pid, Om, 3)


def test_newick_parser_numbers_nodes_like_ape_and_prunes():
    """phylomap_amd/newick.py: tips 1..T in order of appearance, internal nodes in pre-order, cladewise edge rows; dropping
    tips suppresses single-child nodes and adds their branch lengths."""
    from phylomap_amd import newick
    t = newick.read_newick("((A:1,B:2)X:3,(C:4,(D:5,E:6)Y:7)Z:8)R;")
    assert t["tip.label"] == ["A", "B", "C", "D", "E"] and t["Nnode"] == 4 and t["node.label"] == ["R", "X", "Z", "Y"]
    np.testing.assert_array_equal(t["edge"], [[6, 7], [7, 1], [7, 2], [6, 8], [8, 3], [8, 9], [9, 4], [9, 5]])
    np.testing.assert_array_equal(t["edge.length"], [3, 1, 2, 8, 4, 7, 5, 6])
    p = newick.drop_tips("((A:1,B:2)X:3,(C:4,(D:5,E:6)Y:7)Z:8)R;", ["B", "E"])
    assert p["tip.label"] == ["A", "C", "D"]
    np.testing.assert_array_equal(p["edge"], [[4, 1], [4, 5], [5, 2], [5, 3]])
    np.testing.assert_array_equal(p["edge.length"], [4, 8, 4, 12])            # A: 1 + 3, D: 5 + 7
    q = newick.drop_tips(t, ["A", "B"])                                         # the root loses a child and disappears with its edge
    assert q["tip.label"] == ["C", "D", "E"] and q["Nnode"] == 2
    np.testing.assert_array_equal(q["edge.length"], [4, 7, 5, 6])
    z = newick.as_phylomap(p, [1, 2, 1], segments=4)
    assert [len(m) for m in z["maps"]] == [4] * 4 and z["mapnames"][2].tolist() == [1, 1, 1, 2] and z["mapnames"][1].tolist() == [1] * 4
    eng_orders = _lib.tree_orders(z)                                            # the result is a tree the engine accepts
    assert eng_orders[2] == 4


@pytest.mark.skipif(not os.path.exists("/root/reference/inst/extdata/Squamate/squamate.phy"), reason="reference data only in the build container")
def test_newick_reader_reproduces_the_reference_prepared_squamate_tree():
    """inst/extdata/Squamate/squamate.phy through phylomap_amd/newick.py equals the tree the reference prepared in R
    (R/Squamate_tree_setup.R -> phylomap_compatible_squamate_tree.RData): same tip labels, the same ape node numbering and
    edge order, the same branch lengths, the same 100-segment initial paths."""
    from phylomap_amd import newick, rds
    t = newick.read_newick(open("/root/reference/inst/extdata/Squamate/squamate.phy").read())
    z = rds.read_rds("/root/reference/inst/extdata/Squamate/phylomap_compatible_squamate_tree.RData")
    assert t["tip.label"] == list(z["tip.label"]) and t["Nnode"] == 3950
    np.testing.assert_array_equal(t["edge"], np.asarray(z["edge"]).astype(np.int32))
    np.testing.assert_allclose(t["edge.length"], np.asarray(z["edge.length"]), rtol=0, atol=1e-12)
    mine = newick.as_phylomap(t, np.asarray(z["states"]), segments=100)
    for b in (0, 1, 17, 7899):
        np.testing.assert_allclose(mine["maps"][b], np.asarray(z["maps"][b]), rtol=1e-15)
        np.testing.assert_array_equal(mine["mapnames"][b], np.asarray(z["mapnames"][b]).round().astype(np.int32))


def test_shim_type_checks_against_a_mock_of_the_rcpp_surface():
    """R / Rcpp are not installed here, so shim/phylomap_shim.cpp cannot be built; it is at least parsed and type-checked
    against a declaration-only mock of the Rcpp API it uses (tests/mock_rcpp/Rcpp.h) together with the real C-ABI header,
    and it must export every `.Call` symbol of src/RcppExports.cpp plus phylomap_tree_orders."""
    import re
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = os.path.join(root, "shim", "phylomap_shim.cpp")
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-I" + os.path.join(root, "tests", "mock_rcpp"),
                        "-I" + os.path.join(root, "include"), src], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    text = open(src).read()
    exported = set(re.findall(r"RcppExport SEXP (phylomap_\w+)\(", text))
    assert exported == {"phylomap_SPARSEmaketreelistMCMC", "phylomap_maketreelistMCMC", "phylomap_maketreelistMCMC_bigtree",
                        "phylomap_maketreelistEXP", "phylomap_maketreelistMCMCbf", "phylomap_maketreelistMCMCks",
                        "phylomap_maketreelistMCMCmt", "phylomap_maketreelistMCMCksmt", "phylomap_maketreelistMCMC2sDICt",
                        "phylomap_maketreelistMCMCksDICt", "phylomap_tree_orders"}
    rfile = open(os.path.join(root, "shim", "R", "phylomap_tree_orders.R")).read()
    for name in ("pruningwiseedgeorder", "makenodelist", "myreorder"):
        assert re.search(rf"^{name} <- function\(x\)", rfile, re.M)
    # the replica axis from R: the option names the shim reads are the ones the R helper file and INTEGRATION.md use
    shim = open(src).read()
    helper = open(os.path.join(root, "shim", "R", "phylomap_hip_options.R")).read()
    integ = open(os.path.join(root, "INTEGRATION.md")).read()
    for opt in ("phylomap.hip.replicas", "phylomap.hip.reduce", "phylomap.hip.device", "phylomap.hip.devices", "phylomap.hip.rescale",
                "phylomap.hip.mapping", "phylomap.hip.cap_tail"):
        assert f'"{opt}"' in shim and opt in helper and opt in integ
    assert 'containsElementNamed("sites")' in shim and "z$sites <- sites" in helper


def test_bench_line_is_compact():
    """The ONE stdout line of bench.py stays under 4 KB whatever the run measured (the 21 KB line could not be parsed);
    built here from a stub of the full result with every block present and worst-case long kernel names."""
    import json
    import sys
    sys.path.insert(0, ROOT)
    import bench
    kern = {"kernel": "wt_branch_kernel<true,false,true,0>, " * 3, "what": "prose " * 60, "bound": "hbm", "alg_bytes_per_unit": 127.69658182798261,
            "units_per_launch": 327647232, "launches": 20, "avg_launch_ms": 16.0381112575531, "achieved": 2608.750550479929, "peak": 8000.0,
            "unit": "GB/s", "frac": 0.32609381880999116, "traffic": 60180509013.333336, "counter_traffic_frac": 0.469, "valu_issue_frac": 1.0018}
    rl = dict(kern, definition="prose " * 40, dominant_kernel=dict(kern, phase="branch"),
              kernels={"branch": kern, "pruning": kern, "node_draws": kern, "reductions": kern})
    blk = {"workload": "w" * 200, "realisations_per_s": 1.2345678e10, "ms_per_sweep": 21.123456, "replicas": 16384, "roofline": rl,
           "cpu_baseline": {"value": 1.0, "sample": "s" * 300}}
    out = {"metric": "stochastic-map realisations/sec (branches x sites sampled/s)", "value": 1.512345678e10, "unit": "branch-site realisations/s",
           "n_gpus": 8, "steps": 20, "warmup": 5, "ms_per_step": 21.66412345, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f64", "data": "synthetic",
           "config": {"workload": "C3: sumstatMCMC_bigtree sweep, 4-state Q, 10000-tip synthetic tree, Omega*mean(t_b)=4", "n_states": 4, "n_tips": 10000,
                      "branches": 19998, "replicas_total": 131072, "replicas_rank0": 16384, "mapping": "m" * 100,
                      "parallelism": "replica-sharded x8 (weak), one RCCL all-reduce of the statistics"},
           "roofline": rl, "phases_ms_per_sweep": {"a": 1.0}, "pruning_sweep": kern, "hbm_bytes_resident": 158743271040, "replicas_per_gib": 110.8,
           "recoveries": 0,
           "cpu_baseline": {"value": 3172571.2, "unit": "branch-site realisations/s", "cores": 1, "kind": "port", "sample": "s" * 400,
                            "sample_short": "C3 tree+Q, 1 chain, 638 sweeps", "faithful_value": 760875.6, "all_cores": {"value": 2.38e7, "cores": 256, "sample": "s" * 200}},
           "speedup_vs_cpu_1core": 4767.2,
           "single_chain": {"workload": "w" * 200, "mapping": "m" * 200, "ms_per_sweep": 0.0785, "realisations_per_s": 2.5e8, "speedup_vs_cpu_1core": 80.2},
           "stated_length": {"max_iters": 10000, "replicas": 4096, "replicas_per_gib": 28.1, "realisations_per_s": 1.4e10, "ms_per_sweep": 5.9, "recoveries": 0},
           "one_shot_call": {"workload": "w" * 300, "wall_s": 2.9},
           "configs": {k: blk for k in ("C2", "C4bf", "C5", "C5_unstructured", "C2_single_chain", "C4_single_chain", "C5_single_chain", "C2_4096_sites",
                                        "C3_256_sites", "C3_1024_sites", "EXP_C1", "EXP_1000_tips")},
           "expm_per_s": {"n_states": 4, "pade_route": 1.08e8, "eigen_route": 9.39e9, "n61_eigen_route_mfma": 8.7e7, "n61_eigen_route_exact": 6.3e6,
                          "n61_pade_route_mfma": 7.4e6, "n61_mfma_tflops": 39.5, "mfma_f64_peak_tflops": 78.6, "roofline": kern}}
    line = bench.compact_line(out)
    text = json.dumps(line)
    assert len(text) < 4096, len(text)
    assert "\n" not in text
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline"):
        assert k in line, k
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "dominant_kernel"):
        assert k in line["roofline"], k
    assert line["cpu_baseline"]["kind"] == "port" and line["cpu_baseline"]["cores"] == 1
    assert "what" not in text and "prose" not in text


def test_generated_sparse_pruning_kernel_compiles_for_gfx950():
    """The pruning kernel for an unstructured sparse chain matrix is generated at model upload (phm_rtc.cpp) and compiled by hipRTC on
    the GPU box; here its source is generated for the degree-6 neighbour matrix of tests/test_gpu_configs.py and handed to hipcc
    (which cross-compiles without a GPU): one fused multiply-add per non-zero, columns ascending inside a row, nothing else."""
    import shutil
    import subprocess
    import tempfile
    Q = synth.neighbour_Q(20, 6)
    B = np.eye(20) + Q / (1.25 * np.max(np.abs(np.diag(Q))))
    src = _lib.sparse_kernel_source(B)
    assert src.count("__builtin_fma(") == np.count_nonzero(B) == 140
    for i in range(20):
        line = [ln for ln in src.splitlines() if f" y[{i}] = a; }}" in ln][0]
        cols = [int(c) for c in re.findall(r"x\[(\d+)\]", line)]
        assert cols == sorted(np.nonzero(B[i])[0].tolist())
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    with tempfile.TemporaryDirectory() as d:
        f = os.path.join(d, "k.hip")
        open(f, "w").write(src)
        r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17", "-c", f, "-o", os.path.join(d, "k.o")],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
