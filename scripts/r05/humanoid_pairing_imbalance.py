import numpy as np, sys, time
sys.path.insert(0,'/root/repo')
from gymnasium_amd.envs.mujoco import compiler as cp
from oracle import mujoco as omj
om=omj.OracleModel(cp.compile_model("humanoid"))
m=om.m; rng=np.random.default_rng(0)
R=192
ds=[om.make_data() for _ in range(R)]
t0=time.time()
for d in ds:
    d.reset()
    q=m.qpos0.copy(); q[7:]+=rng.uniform(-.01,.01,size=m.nq-7); v=rng.normal(size=m.nv)*0.01
    d.set_state(q,v,rng.uniform(-.4,.4,size=m.nu))
W=[]
for step in range(180):   # env steps of 5 substeps with fresh random ctrl
    for d in ds:
        d.set_state(None,None,rng.uniform(-.4,.4,size=m.nu))
        d.step(5)
    if step>=160:
        w=[]
        for d in ds:
            d.forward()
            w.append((d.get("ncon"), d.get("solver_iter"), d.get("nefc")))
        W.append(w)
print("time",time.time()-t0)
W=np.array(W)  # [T, R, 3]
ncon, it, nefc = W[...,0], W[...,1], W[...,2]
print("ncon mean %.2f  sweeps mean %.1f  nefc mean %.1f"%(ncon.mean(), it.mean(), nefc.mean()))
work = it*ncon
def pair_cost(w): return np.maximum(w[:, 0::2], w[:, 1::2]).mean()
print("pair max / mean (unsorted):", pair_cost(work)/work.mean())
# sorted by PREVIOUS step's work
gains=[]
for t in range(1,len(work)):
    order=np.argsort(work[t-1]); ws=work[t][order]
    gains.append(np.maximum(ws[0::2],ws[1::2]).mean()/work[t].mean())
print("pair max / mean when paired by previous step's work:", np.mean(gains))
order=np.argsort(work,axis=1); ws=np.take_along_axis(work,order,1)
print("ideal (sorted by own work):", pair_cost(ws)/work.mean())
print("corr work t vs t+1:", np.corrcoef(work[:-1].ravel(), work[1:].ravel())[0,1])
f=155/388
for name,r in (("unsorted",pair_cost(work)/work.mean()),("prev-sorted",np.mean(gains))):
    print(name,"relative pass cost", f+(1-f)*r)
