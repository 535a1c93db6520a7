import sys, time, numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla, networkx as nx
sys.path.insert(0, "/root/repo")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 8
p = 2.0e6 / (100000 * (100000 - 1) / 2) if n == 100000 else 0.01
G = nx.fast_gnp_random_graph(n, p, seed=0)
e = np.array([(min(a, b), max(a, b)) for a, b in G.edges() if abs(a - b) != 1], dtype=np.int64)
m = len(e); k = m // 10
x = np.zeros(m); x[np.random.default_rng(0).choice(m, k, replace=False)] = 1.0
ci, cj = e[:, 0], e[:, 1]
fi = np.arange(n - 1); fj = fi + 1
def lap(x):
    idx = x > 1e-10
    i = np.concatenate([fi, ci[idx]]); j = np.concatenate([fj, cj[idx]]); w = np.concatenate([np.ones(n - 1), x[idx]])
    A = sp.coo_matrix((np.concatenate([w, w]), (np.concatenate([i, j]), np.concatenate([j, i]))), shape=(n, n)).tocsr()
    d = np.asarray(A.sum(axis=1)).ravel()
    return (sp.diags(d) - A).tocsr(), d
def lanczos_steps(L, linf, tol=1e-8, maxit=2000):
    # plain Lanczos w/o reorth on 1-perp, check true residual every 8 steps via Ritz vector (full V kept)
    rs = np.random.RandomState(7); v = rs.normal(size=n); v -= v.mean(); v /= np.linalg.norm(v)
    V = [v]; al = []; be = []
    vp = np.zeros(n); b = 0.0
    for j in range(maxit):
        w = L @ v - b * vp; a = v @ w; w -= a * v; w -= w.mean(); al.append(a)
        b = np.linalg.norm(w); be.append(b); vp = v; v = w / b; V.append(v)
        if (j + 1) % 8 == 0 and j > 30:
            T = np.diag(al) + np.diag(be[:-1], 1) + np.diag(be[:-1], -1)
            th, S = np.linalg.eigh(T); s = S[:, 0]
            est = abs(be[-1] * s[-1]) * np.sqrt(n) * 0.8
            if est / linf < tol * 3:
                y = np.array(V[:-1]).T @ s; y -= y.mean(); y /= np.linalg.norm(y)
                rho = y @ (L @ y); r = np.abs(L @ y - rho * y).sum() / linf
                if r < tol: return j + 1, rho
    return maxit, None
def lobpcg(L, d, linf, tol=1e-8, maxit=2000, prec="jacobi", shift=0.0):
    rs = np.random.RandomState(7); xv = rs.normal(size=n); xv -= xv.mean(); xv /= np.linalg.norm(xv)
    Lx = L @ xv; pv = None; Lp = None
    for it in range(maxit):
        rho = xv @ Lx
        r = Lx - rho * xv
        if np.abs(r).sum() / linf < tol: return it, rho
        if prec == "jacobi": w = r / (d - shift * rho)
        else: w = r.copy()
        w -= w.mean()
        w -= (w @ xv) * xv; w /= np.linalg.norm(w)
        Lw = L @ w
        if pv is None: S = np.stack([xv, w], 1); LS = np.stack([Lx, Lw], 1)
        else: S = np.stack([xv, w, pv], 1); LS = np.stack([Lx, Lw, Lp], 1)
        Gm = S.T @ S; Hm = S.T @ LS
        import scipy.linalg as sl
        th, C = sl.eigh(Hm, Gm)
        c = C[:, 0]
        pn = S[:, 1:] @ c[1:]; Lpn = LS[:, 1:] @ c[1:]
        xv = c[0] * xv + pn; Lx = c[0] * Lx + Lpn
        nr = np.linalg.norm(xv); xv /= nr; Lx /= nr
        pnn = np.linalg.norm(pn); pv = pn / pnn; Lp = Lpn / pnn
    return maxit, None
for it in range(iters):
    L, d = lap(x); linf = 2 * d.max()
    t0 = time.time(); nl, rl = lanczos_steps(L, linf); t1 = time.time()
    nj, rj = lobpcg(L, d, linf); t2 = time.time()
    nj2, rj2 = lobpcg(L, d, linf, shift=1.0); t3 = time.time()
    print(f"it {it} nnz {L.nnz} lanczos {nl} ({rl}) lobpcg-jacobi {nj} ({rj}) shifted {nj2} ({rj2}) times {t1-t0:.1f} {t2-t1:.1f} {t3-t2:.1f}", flush=True)
    # FW step with an accurate vector (Lanczos ritz value only; use eigsh for v)
    wv, Vv = spla.eigsh(L, k=2, which="SA", tol=1e-10, ncv=64, v0=np.random.RandomState(7).normal(size=n))
    v = Vv[:, np.argsort(wv)[1]]
    g = (v[ci] - v[cj]) ** 2
    s = np.zeros(m); s[np.argpartition(g, -k)[-k:]] = 1.0
    x = x + 2.0 / (it + 2) * (s - x)
