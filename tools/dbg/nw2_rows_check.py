import os, sys, numpy as np
ROOT=os.environ.get("GRAFT_REPO_ROOT","/root/repo")
sys.path[:0]=[os.path.join(ROOT,"dojo.jl_amd","host"),os.path.join(ROOT,"oracle"),os.path.join(ROOT,"tests"),ROOT]
import dojo_amd as d
from dojo_amd import api
from oracle import Oracle
from random_mechanisms import random_mechanism
for seed,nb in ((21,20),(22,24),(23,30),(24,17)):
    spec,z,u=random_mechanism(seed,nb=nb)
    B=8
    Z=np.stack([z]*B); U=np.stack([u*(1+0.1*i) for i in range(B)])
    o=Oracle(spec)
    res={}
    for rows in ("0",None):
        if rows is None: os.environ.pop("DOJO_ROWS",None)
        else: os.environ["DOJO_ROWS"]=rows
        gm=api.BatchedMechanism(spec,B,dtype="f64")
        zz=Z.copy()
        for k in range(3):
            zz,st,it=gm.step(zz,U,with_gradient=(k==2))
        dz,du=gm.gradients(); res[rows]=(zz.copy(),st.copy(),it.copy(),dz.copy()); gm.close()
    zo=Z.copy()
    for k in range(3):
        zo,sto,ito,dzo,duo=o.step_batch(zo,U,with_grad=(k==2),nthreads=4)
    a,b=res["0"],res[None]
    print("seed %d nb %d: quad vs rows state %.1e iters equal %s | rows vs oracle state %.1e iters equal %s grad %.1e status %s"%(seed,nb,np.abs(a[0]-b[0]).max(),np.array_equal(a[2],b[2]),np.abs(b[0]-zo).max(),np.array_equal(b[2],ito),np.abs(b[3]-dzo).max()/max(1,np.abs(dzo).max()),b[1].tolist()))
