import sys, time, numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla, networkx as nx, scipy.linalg as sl
n = 100000
p = 2.0e6 / (n * (n - 1) / 2)
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
def block_lobpcg(L, d, linf, q, tol=1e-8, maxit=1000):
    rs = np.random.RandomState(7); X = rs.normal(size=(n, q)); X -= X.mean(axis=0); X, _ = np.linalg.qr(X)
    P = None
    for it in range(maxit):
        LX = L @ X
        H = X.T @ LX; th, C = np.linalg.eigh(H); X = X @ C; LX = LX @ C
        R = LX - X * th
        r0 = np.abs(R[:, 0]).sum() / linf
        if r0 < tol: return it, th[0]
        W = R / d[:, None]; W -= W.mean(axis=0)
        S = np.hstack([X, W] + ([P] if P is not None else []))
        S, _ = np.linalg.qr(S)
        LS = L @ S
        Hs = S.T @ LS; ths, Cs = np.linalg.eigh(Hs)
        Xn = S @ Cs[:, :q]
        P = Xn - X @ (X.T @ Xn)
        X = Xn
    return maxit, None
for it in range(8):
    L, d = lap(x); linf = 2 * d.max()
    out = []
    for q in (1, 2, 4):
        t0 = time.time(); ni, lam = block_lobpcg(L, d, linf, q); out.append((q, ni, round(time.time() - t0, 1)))
    print(f"it {it} nnz {L.nnz} block LOBPCG-Jacobi iterations (q, iters, s): {out} lam {lam}", flush=True)
    wv, Vv = spla.eigsh(L, k=2, which="SA", tol=1e-10, ncv=64, v0=np.random.RandomState(7).normal(size=n))
    v = Vv[:, np.argsort(wv)[1]]
    g = (v[ci] - v[cj]) ** 2
    s = np.zeros(m); s[np.argpartition(g, -k)[-k:]] = 1.0
    x = x + 2.0 / (it + 2) * (s - x)
