#!/usr/bin/env python3
"""lambda_2 .. lambda_4 of the reference's own 20 iterates of configs[1] / configs[3] (ARPACK on the CPU) next to the Lanczos steps the library needed on
them (tools/land_teacher.py): what is left of a solve after the landscape start is the gap (lambda_3 - lambda_2) / (lambda_max - lambda_2) -- profiles/r5_landscape.md.
usage: gap_probe.py c2|c4"""
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as sla, sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench
cfg=sys.argv[1]
wl=bench.make_workload(cfg)
n,ci,cj,k=wl["n"],wl["ci"],wl["cj"],wl["k"]; m=len(ci)
gv=np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests", "golden", {"c2":"er10k_vertices.npz","c4":"er100k_arpack.npz"}[cfg]))
bits=gv["ref_s_bits"]
steps_w={"c4":[72, 218, 224, 168, 152, 147, 350, 166, 216, 136, 232, 207, 242, 286, 125, 182, 180, 166, 276, 168],
         "c2":[141, 221, 193, 151, 132, 151, 239, 91, 135, 190, 149, 219, 194, 134, 219, 261, 176, 163, 156, 173]}[cfg]
steps_u={"c4":[103, 336, 326, 232, 217, 172, 353, 193, 263, 176, 264, 211, 289, 293, 156, 210, 185, 198, 283, 172],
         "c2":[155, 285, 250, 201, 176, 185, 290, 156, 163, 212, 152, 228, 214, 147, 222, 248, 210, 187, 170, 193]}[cfg]
x=wl["x0"].copy()
for t in range(20):
    i=np.concatenate([wl["fi"],ci]); j=np.concatenate([wl["fj"],cj]); w=np.concatenate([wl["fw"],x*wl["cw"]])
    keep=w>1e-10; i,j,w=i[keep],j[keep],w[keep]
    A=sp.coo_matrix((np.concatenate([w,w]),(np.concatenate([i,j]),np.concatenate([j,i]))),shape=(n,n)).tocsr()
    d=np.asarray(A.sum(1)).ravel(); L=(sp.diags(d)-A).tocsr()
    lmax=2*d.max()
    ev=sla.eigsh(L,k=5,which="SA",tol=1e-10,return_eigenvectors=False,ncv=80)
    ev=np.sort(ev)
    gap=(ev[2]-ev[1])/(lmax-ev[1])
    print("%2d lam2 %.5f lam3 %.5f lam4 %.5f  (lam3-lam2)/(lmax-lam2) %.2e  1/sqrt %.0f  steps weighted %d unweighted %d"%(t,ev[1],ev[2],ev[3],gap,1/np.sqrt(gap),steps_w[t],steps_u[t]),flush=True)
    x=x+2.0/(t+2)*(np.unpackbits(bits[t])[:m].astype(np.float64)-x)
