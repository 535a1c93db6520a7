#!/usr/bin/env python3
"""CPU emulation of a TWO-STAGE cold start (next round's candidate, profiles/r5_landscape.md): the landscape picks the 64 highest peaks, the smallest
eigenpair of the Dirichlet Laplacian on each peak's 2-hop ball ranks them, the best ball's ground state (extended by zero) starts the Lanczos solve.
Steps on the reference's own 20 iterates (tests/golden), configs[3]: unweighted 4 682, landscape 3 974, ball 3 606.  usage: ball_start_emulation.py c4   (at configs[1] the 2-hop ball of a dense iterate is most of the graph: use 1 hop there)"""
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as sla, sys, os
from scipy.linalg import eigh_tridiagonal
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench
cfg=sys.argv[1]; nit=20
wl = bench.make_workload(cfg)
n=wl["n"]; ci,cj=wl["ci"],wl["cj"]; m=len(ci); k=wl["k"]
def lap(x):
    i=np.concatenate([wl["fi"],ci]); j=np.concatenate([wl["fj"],cj]); w=np.concatenate([wl["fw"],x*wl["cw"]])
    keep = w>1e-10
    i,j,w=i[keep],j[keep],w[keep]
    A=sp.coo_matrix((np.concatenate([w,w]),(np.concatenate([i,j]),np.concatenate([j,i]))),shape=(n,n)).tocsr()
    d=np.asarray(A.sum(1)).ravel()
    return (sp.diags(d)-A).tocsr(), d
def lanczos_steps(L, u0, tol=1e-8, maxit=8000, want=False):
    norm_inf = abs(L).sum(1).max()
    u=u0-u0.mean(); v=u/np.linalg.norm(u)
    al=[]; be=[]; V=[v] if want else None
    vprev=np.zeros(n); b=0.0
    thr = tol*norm_inf/(0.8*np.sqrt(n))
    for j in range(maxit):
        w=L@v - b*vprev
        a=v@w; w-=a*v
        w-=w.mean()
        b2=np.linalg.norm(w)
        al.append(a)
        if j>=8 and (j%2==1):
            ev,S=eigh_tridiagonal(np.array(al),np.array(be),select='i',select_range=(0,0))
            if b2*abs(S[-1,0])<thr:
                if want:
                    y=np.array(V).T@S[:,0]; return j+1, ev[0], y
                return j+1, ev[0], None
        be.append(b2); vprev=v; v=w/b2; b=b2
        if want: V.append(v)
    return maxit,None,None
gv=np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests", "golden", {"c2":"er10k_vertices.npz","c4":"er100k_arpack.npz"}[cfg]))
bits=gv["ref_s_bits"]
z=np.random.RandomState(7).normal(size=(n,))
x=wl["x0"].copy(); M=64; tot={}
for t in range(20):
    L,d=lap(x)
    A=(sp.diags(d)-L).tocsr()
    u=1.0/d
    for kk in range(3): u=(1.0+A@u)/d
    C=np.argsort(-u)[:M]
    best=None
    for c in C:
        ball=np.array([c])
        for h in range(2): ball=np.unique(np.concatenate([ball,A[ball].indices]))
        Lb=L[ball][:,ball].toarray()
        ev,V=np.linalg.eigh(Lb)
        if best is None or ev[0]<best[0]: best=(ev[0],ball,V[:,0])
    phi=np.zeros(n); phi[best[1]]=best[2]
    w128=(u/u.max())**128
    res={}
    res["unweighted"]=lanczos_steps(L,z)[0]
    res["landscape"]=lanczos_steps(L,z*w128)[0]
    res["ball"]=lanczos_steps(L,phi+1e-3*z/np.linalg.norm(z))[0]
    res["ball+land"]=lanczos_steps(L,phi+ (z*w128)/np.linalg.norm(z*w128))[0]
    for k2,v in res.items(): tot[k2]=tot.get(k2,0)+v
    print(t,"ball size",len(best[1]),"local eigenvalue %.4f"%best[0],res,flush=True)
    x=x+2.0/(t+2)*(np.unpackbits(bits[t])[:m].astype(np.float64)-x)
print("TOTAL",tot)
