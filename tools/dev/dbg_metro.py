import sys; sys.path.insert(0,'.')
import numpy as np, torch
from rlsolver_amd import ops_mcpg_tsp as mops
from rlsolver_amd.ops_mcpg_tsp import PackedChains
DEV=torch.device('cuda:0')
n,C=300,192
rng=np.random.RandomState(n)
xs=torch.from_numpy((rng.rand(n,C)<0.5).astype(np.float32)).to(DEV)
probs=torch.from_numpy((rng.rand(n)*0.6+0.2).astype(np.float32)).to(DEV)
for T in (1,2,8,9,30):
    a32=torch.zeros(T,dtype=torch.int64,device=DEV); apk=torch.zeros(T,dtype=torch.int64,device=DEV)
    o32=xs.clone(); mops.mcpg_metro_rounds(o32,probs,T,seed=77,accepts=a32)
    opk=PackedChains.pack(xs); mops.mcpg_metro_rounds(opk,probs,T,seed=77,accepts=apk)
    d=(opk.unpack()!=o32)
    print(T,'state diff',int(d.sum()),'acc32',a32.tolist()[:10],'accpk',apk.tolist()[:10])
    if d.any():
        idx=d.nonzero()[:5]; print(idx.tolist())
