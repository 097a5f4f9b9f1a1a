"""sumstatEXP throughput probe: python tools/probe_exp.py config n_tips N"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from phylomap_amd import api, synth
cfg = int(sys.argv[1]); tips = int(sys.argv[2]); N = int(sys.argv[3])
z, Q, pid, Omega = synth.config_problem(cfg, n_tips=tips)
E = z["edge"].shape[0]
api.sumstatEXP(z, Q, pid, 64, seed=1)
t0 = time.time(); out = api.sumstatEXP(z, Q, pid, N, seed=1); dt = time.time() - t0
print(f"C{cfg} tips={tips} N={N}: {dt:.3f}s wall (incl. upload/download) -> {E*N/dt/1e9:.3f} G branch-sample/s; jumps/sample={out[:, Q.shape[0]:].sum(1).mean():.1f}")
