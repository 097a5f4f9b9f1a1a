import sys, time, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tools/squamate_dic')
from phylomap_amd import _lib, api
d=np.load('/root/repo/tests/golden/squamate/seed101_tips.npz')
E,T=d["edge"].shape[0],len(d["states"])
z={"edge":d["edge"],"Nnode":T-1,"edge.length":d["edge_length"],"states":d["states"]}
z["maps"]=[np.full(100,l/100) if c>T else np.full(2,l/2) for (p,c),l in zip(d["edge"],d["edge_length"])]
z["mapnames"]=[np.ones(100,dtype=np.int32) if c>T else np.array([1,d["states"][c-1]],dtype=np.int32) for (p,c) in d["edge"]]
Q=np.array([[-0.001,0.001],[0.006,-0.006]])
for mapping,S,N in (("replicas",1,3),("replicas",64,3),("tiles",64,6),("branches",1,20)):
    t=time.time()
    eng=_lib.Engine(z,Q,[.5,.5],10.0,N,variant=_lib.PHM_MCMC_BIGTREE,seed=1,n_replicas=S,mapping=mapping,reduce=True)
    eng.run(N); eng.sync(); dt=time.time()-t
    print(f"squamate {mapping} S={S}: {eng.info().last_run_ms/N:.1f} ms/sweep (kernel), total {dt:.1f}s", flush=True)
    eng.close()
