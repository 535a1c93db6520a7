#!/usr/bin/env python3
"""CPU emulation: block Lanczos (block B, full re-orthogonalisation) whose start block are the top landscape peaks, on the reference's own iterates of configs[1] / [3]
(same stop rule as tools/experiments/start_landscape_emulation.py).  Measured, configs[1], 20 iterates: B = 1: 3 360 steps, B = 2: 2 702 block steps, B = 3: 2 406 --
less than a block step costs more.  usage: block_landscape_emulation.py c2|c4 B [iterates]"""
import numpy as np, scipy.sparse as sp, sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench
cfg=sys.argv[1]; B=int(sys.argv[2]); nit=int(sys.argv[3]) if len(sys.argv)>3 else 20
wl=bench.make_workload(cfg)
n,ci,cj,k=wl["n"],wl["ci"],wl["cj"],wl["k"]; m=len(ci)
gv=np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests", "golden", {"c2":"er10k_vertices.npz","c4":"er100k_arpack.npz"}[cfg]))
bits=gv["ref_s_bits"]
def block_lanczos(L, U0, tol=1e-8, maxit=400):
    norm_inf=abs(L).sum(1).max(); thr=tol*norm_inf
    U0=U0-U0.mean(0); Q,_=np.linalg.qr(U0); b=Q.shape[1]
    Vs=[Q]; T=np.zeros((0,0)); Bprev=None; Qprev=None
    blocks_A=[]; blocks_B=[]
    for j in range(maxit):
        W=L@Q
        if Qprev is not None: W-=Qprev@Bprev.T
        Aj=Q.T@W; W-=Q@Aj
        W-=W.mean(0)
        for V in Vs: W-=V@(V.T@W)      # full reorthogonalisation (clean counts)
        Qn,Bn=np.linalg.qr(W)
        blocks_A.append(Aj); blocks_B.append(Bn)
        J=len(blocks_A); T=np.zeros((J*b,J*b))
        for i in range(J):
            T[i*b:(i+1)*b,i*b:(i+1)*b]=blocks_A[i]
            if i+1<J:
                T[(i+1)*b:(i+2)*b,i*b:(i+1)*b]=blocks_B[i]; T[i*b:(i+1)*b,(i+1)*b:(i+2)*b]=blocks_B[i].T
        if j>=3:
            ev,S=np.linalg.eigh(T)
            s=S[:,0]
            # residual estimate ||B_J s_last||_2 * ~0.8 sqrt(n) for the 1-norm
            est=np.linalg.norm(Bn@s[-b:])
            if est*0.8*np.sqrt(n)<thr: return j+1, ev[0]
        Qprev,Bprev=Q,Bn; Q=Qn; Vs.append(Q)
    return maxit,None
z=np.random.RandomState(7).normal(size=(n,4))
x=wl["x0"].copy(); tot=0; res=[]
for t in range(nit):
    i=np.concatenate([wl["fi"],ci]); j=np.concatenate([wl["fj"],cj]); w=np.concatenate([wl["fw"],x*wl["cw"]])
    keep=w>1e-10; i,j,w=i[keep],j[keep],w[keep]
    A=sp.coo_matrix((np.concatenate([w,w]),(np.concatenate([i,j]),np.concatenate([j,i]))),shape=(n,n)).tocsr()
    d=np.asarray(A.sum(1)).ravel(); L=(sp.diags(d)-A).tocsr()
    u=1.0/d
    for kk in range(3): u=(1.0+A@u)/d
    wt=(u/u.max())**128
    # start block: column 0 the weighted draw; the others the weighted draw with the mass of the earlier columns' main spots masked out
    cols=[]; mask=np.ones(n)
    for c in range(B):
        v=z[:,c%4]*wt*mask
        cols.append(v)
        top=np.abs(v)>0.2*np.abs(v).max()
        # mask the top spot and its neighbours for the next column
        nb=(A@top.astype(float))>0
        mask=mask*(~(top|nb))
    U0=np.array(cols).T
    st,lam=block_lanczos(L,U0)
    res.append(st); tot+=st
    print(t,"block steps",st,"lam2",lam,flush=True)
    x=x+2.0/(t+2)*(np.unpackbits(bits[t])[:m].astype(np.float64)-x)
print("TOTAL block steps",tot,res)
