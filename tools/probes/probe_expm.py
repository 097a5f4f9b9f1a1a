"""expm throughput probe: python tools/probe_expm.py n n_t"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from phylomap_amd import api, synth
n = int(sys.argv[1]); nt = int(sys.argv[2])
Q = synth.dense_Q(n, 0.005, 0.015, seed=n); Q = (Q + Q.T) / 2; np.fill_diagonal(Q, 0); np.fill_diagonal(Q, -Q.sum(1))
l, r, d = api.eigen_decompose(Q)
t = np.random.default_rng(0).exponential(3.0, nt)
for name, fn in (("eigen exact", lambda: api.expm_eigen(l, r, d, t)), ("eigen mfma", lambda: api.expm_eigen(l, r, d, t, mfma=True)),
                 ("pade exact", lambda: api.expm_pade(Q, t[: max(1, nt // 8)])),
                 ("pade mfma", lambda: api.expm_pade(Q, t[: max(1, nt // 8)], mfma=True))):
    if name.endswith("mfma") and not (16 < n <= 64): continue
    fn(); _, ms = fn()
    cnt = nt if "eigen" in name else max(1, nt // 8)
    fl = 2 * n ** 3 * cnt / (ms / 1e3) / 1e12
    print(f"n={n} {name}: {cnt/(ms/1e3):.4g} matrices/s  ({ms:.2f} ms, ~{fl:.2f} TFLOP/s per GEMM-equivalent)")
