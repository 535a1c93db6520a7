import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np
from mac_amd import _lib
from test_gpu_parity import make_er
from conftest import load_golden
g=load_golden("er10k_x0"); n=10000
ci,cj=make_er(n,0.01,0); m=len(ci)
fi=np.arange(n-1,dtype=np.int32); fj=fi+1
P=_lib.Problem(n,fi,fj,np.ones(n-1),ci,cj,np.ones(m))
x0=np.zeros(m); x0[g["x0_idx"]]=1.0
P.set_x(x0); P.set_start(np.random.RandomState(7).normal(size=(4,n)).T[:,0].copy())
k=m//10; xp=x0
for it in range(5):
    f,dual,gn=P.fw_step(k,it)
    grad=P.gradient(); s=P.lp_topk(k)
    P.fw_commit(); x=P.get_x()
    exp=xp+(2.0/(it+2.0))*(s-xp)
    bad=np.nonzero(x!=exp)[0]
    print(it,f,dual,gn,P.stats.lanczos_steps,P.stats.support,P.stats.gpu_ms,"bad",len(bad))
    for b in bad[:5]:
        print("   ",b,repr(x[b]),repr(exp[b]),repr(xp[b]),s[b],repr(grad[b]))
    kth=np.sort(grad)[-k]; print("   ties at kth:",np.sum(grad==kth), "s sum", s.sum())
    xp=x
