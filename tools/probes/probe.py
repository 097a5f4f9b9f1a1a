"""Throughput probe for the MCMC engine (not part of the product): python tools/probe.py S iters ipl [config] [mapping]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from phylomap_amd import _lib, synth

S = int(sys.argv[1]); iters = int(sys.argv[2]); ipl = int(sys.argv[3]); cfg = int(sys.argv[4]) if len(sys.argv) > 4 else 2; mapping = sys.argv[5] if len(sys.argv) > 5 else 'auto'
z, Q, pid, Omega = synth.config_problem(cfg)
E = z["edge"].shape[0]
t0 = time.time()
eng = _lib.Engine(z, Q, pid, Omega, iters + 10, variant=_lib.PHM_MCMC_BIGTREE, seed=1, n_replicas=S, reduce=True, iters_per_launch=ipl, mapping=mapping)
t1 = time.time()
eng.run(10); eng.sync()
i0 = eng.info()
t2 = time.time()
eng.run(iters); eng.sync()
t3 = time.time()
i1 = eng.info()
units = E * S * iters
seg = (i1.seg_read - i0.seg_read) / units
n = Q.shape[0]
balg = 16 * n + 12 * seg + 26
print(f"cfg=C{cfg} mapping={mapping} S={S} iters={iters} ipl={ipl} create={t1-t0:.2f}s bytes={i1.device_bytes/2**30:.2f}GiB rows/rep={i1.rows_per_replica}")
print(f"  wall={t3-t2:.3f}s kernel_ms={i1.last_run_ms:.1f} launches={i1.last_run_launches} -> {units/(i1.last_run_ms/1e3)/1e9:.3f} G branch-samples/s; mean(m+m')={seg:.2f} B_alg={balg:.0f} B -> {units*balg/(i1.last_run_ms/1e3)/1e12:.3f} TB/s algorithmic")
