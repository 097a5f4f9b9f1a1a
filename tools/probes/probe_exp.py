"""sumstatEXP throughput by sample count and mapping: python tools/probe_exp.py [cfg]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from phylomap_amd import _lib, api, synth
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
z, Q, pid, Om = synth.config_problem(cfg)
E = z["edge"].shape[0]
L = _lib.load()
resc = cfg != 1
for N in (1000, 4096, 16384, 65536):
    for mapping in ("replicas", "tiles"):
        api.sumstatEXP(z, Q, pid, 64, seed=1, rescale=resc, mapping=mapping)
        t = time.time(); st = api.sumstatEXP(z, Q, pid, N, seed=2, rescale=resc, mapping=mapping); wall = time.time() - t
        ms = L.phm_last_kernel_ms()
        print(f"C{cfg} N={N:6d} {mapping:9s}: kernel {ms:8.3f} ms  {E*N/(ms/1e3)/1e9:7.3f} G/s   whole call {wall*1e3:8.1f} ms", flush=True)
