import sys, time, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
from phylomap_amd import _lib, synth
z,Q,pid,Om=synth.config_problem(2)
for mapping,S in (("branches",1),("replicas",1),("branches",16),("branches",64),("tiles",64),("branches",256),("tiles",256),("replicas",256),("branches",1024),("tiles",1024),("replicas",1024),("branches",4096),("tiles",4096),("replicas",4096),("tiles",16384),("replicas",16384),("tiles",65536),("replicas",65536),("tiles",131072),("replicas",131072),("replicas",196608),("replicas",327680),("replicas",393216),("branches",32),("tiles",32)):
    N=40 if mapping=="branches" or S>1 else 10
    try:
        eng=_lib.Engine(z,Q,pid,Om,N+2,variant=_lib.PHM_MCMC_BIGTREE,seed=1,n_replicas=S,mapping=mapping,reduce=True,storage=2 if mapping=="replicas" else 0)
        eng.run(2); eng.sync()
        t=time.time(); eng.run(N); eng.sync(); dt=time.time()-t
        print(f"C2 {mapping:9s} S={S:5d}: {1e3*dt/N:8.3f} ms/sweep  {S*1998*N/dt/1e6:10.2f} M units/s  kernel_ms={eng.info().last_run_ms/N:.3f}", flush=True)
        eng.close()
    except Exception as ex:
        print(mapping,S,'ERR',ex)
