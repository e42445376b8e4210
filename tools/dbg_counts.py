import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bench
from graphem_rapids_amd import _native
n,D,k,S,edges,pos=bench.make_workload(sys.argv[1] if len(sys.argv)>1 else "rr1m")
eng=_native.Engine(n,D,edges,1.0,0.2,0.5,k,S)
eng.set_positions(pos)
rng=np.random.default_rng(0)
for it in range(8):
    sampled=rng.permutation(len(edges))[:S].astype(np.int32)
    eng.knn_midpoints(sampled)
    a,b,c=eng.knn_last_counts()
    p=eng.get_positions()
    print(it,"subset cnt mean/max",a.mean(),a.max(),"final cnt mean/max",b.mean(),b.max(),"ovf",c.sum(),"| pos absmax",np.abs(p).max(),"median |x|",np.median(np.abs(p)))
    eng.step(sampled)
