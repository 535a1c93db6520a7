// band.h -- host-side analysis of the BLOCK Lanczos matrix (blocklan.h): symmetric, N = J * b rows, half-bandwidth w = b
// (diagonal blocks A_j full, sub-diagonal blocks B_j upper triangular).  Smallest eigenpair by Rayleigh-quotient iteration from
// the previous chunk's vector, verified by an inertia count; the q lowest pairs by shifted subspace iteration.  Plain C++,
// O(N w^2) per factorisation: it runs while the GPU executes the next chunk, like tridiag.h does for the scalar recurrence.
// Storage: lower band, h[k * N + i] = H(i + k, i) for k = 0..w.
#pragma once
#include <algorithm>
#include <cmath>
#include <vector>

namespace machip {
namespace band {

struct Factor {
    int N = 0, w = 0;
    std::vector<double> L;   // unit lower band, L[k * N + j] = L(j + k, j), k = 1..w
    std::vector<double> D;
    int negative = 0;        // number of negative pivots = eigenvalues below the shift
};

// (H - shift I) = L D L^T without pivoting.  The shifts used here sit at or below the smallest eigenvalue, where the shifted matrix
// is positive (semi-)definite up to one direction and the factorisation needs no pivoting; a vanishing pivot is perturbed.
inline void ldl(const double* h, int N, int w, double shift, double tiny, Factor& F) {
    F.N = N; F.w = w; F.L.assign((size_t)(w + 1) * N, 0.0); F.D.assign((size_t)N, 0.0); F.negative = 0;
    double* L = F.L.data(); double* D = F.D.data();
    for (int j = 0; j < N; ++j) {
        double d = h[j] - shift;
        const int k0 = std::max(0, j - w);
        for (int k = k0; k < j; ++k) { const double l = L[(size_t)(j - k) * N + k]; d -= l * l * D[k]; }
        if (std::fabs(d) < tiny) d = -tiny;
        D[j] = d;
        F.negative += d < 0.0;
        const int i1 = std::min(N - 1, j + w);
        for (int i = j + 1; i <= i1; ++i) {
            double v = h[(size_t)(i - j) * N + j];
            for (int k = std::max(0, i - w); k < j; ++k) v -= L[(size_t)(i - k) * N + k] * L[(size_t)(j - k) * N + k] * D[k];
            L[(size_t)(i - j) * N + j] = v / d;
        }
    }
}
inline void solve(const Factor& F, double* x) {
    const int N = F.N, w = F.w; const double* L = F.L.data(); const double* D = F.D.data();
    for (int j = 0; j < N; ++j) {
        const double xj = x[j];
        const int i1 = std::min(N - 1, j + w);
        for (int i = j + 1; i <= i1; ++i) x[i] -= L[(size_t)(i - j) * N + j] * xj;
    }
    for (int j = 0; j < N; ++j) x[j] /= D[j];
    for (int j = N - 1; j >= 0; --j) {
        double s = x[j];
        const int i1 = std::min(N - 1, j + w);
        for (int i = j + 1; i <= i1; ++i) s -= L[(size_t)(i - j) * N + j] * x[i];
        x[j] = s;
    }
}
inline void matvec(const double* h, int N, int w, const double* x, double* y) {
    for (int i = 0; i < N; ++i) y[i] = h[i] * x[i];
    for (int k = 1; k <= w; ++k) {
        const double* hk = h + (size_t)k * N;
        for (int i = 0; i + k < N; ++i) { y[i + k] += hk[i] * x[i]; y[i] += hk[i] * x[i + k]; }
    }
}
inline double dot(const double* a, const double* b, int N) { double s = 0.0; for (int i = 0; i < N; ++i) s += a[i] * b[i]; return s; }
inline void normalize(double* s, int N) {
    const double n2 = dot(s, s, N);
    const double inv = n2 > 0.0 ? 1.0 / std::sqrt(n2) : 0.0;
    for (int i = 0; i < N; ++i) s[i] *= inv;
}
inline double scale_of(const double* h, int N, int w) {
    double s = 0.0;
    for (int i = 0; i < N; ++i) {
        double r = std::fabs(h[i]);
        for (int k = 1; k <= w; ++k) { if (i + k < N) r += std::fabs(h[(size_t)k * N + i]); if (i - k >= 0) r += std::fabs(h[(size_t)k * N + i - k]); }
        s = std::max(s, r);
    }
    return s > 0.0 ? s : 1.0;
}

struct Smallest {
    double theta = 0.0;
    std::vector<double> s;     // unit eigenvector, length N
    int factorisations = 0;    // diagnostic
};

// Smallest eigenpair.  guess / guess_len: the previous call's vector (zero-padded here), theta_guess its eigenvalue.
// rough = true (far from convergence: the caller only forecasts from the result): ONE factorisation at a shift below the spectrum
// (the matrix is a projection of a positive semi-definite one) and a few inverse-iteration solves -- Rayleigh-quotient iteration from
// the padded previous vector lands on an interior eigenvalue while the smallest Ritz value is still falling fast, and the bisection
// that repairs it costs 60 factorisations per chunk (measured: the host became the bottleneck of the first third of a solve).
inline void smallest_eigpair(const double* h, int N, int w, const double* guess, int guess_len, double theta_guess, Smallest& out,
                             Factor& F, std::vector<double>& wk, bool rough = false) {
    out.s.assign((size_t)N, 0.0);
    out.factorisations = 0;
    const double scale = scale_of(h, N, w);
    const double tiny = 1e-300 + 1e-30 * scale;
    double* s = out.s.data();
    wk.resize((size_t)N);
    auto rq = [&](const double* v) { matvec(h, N, w, v, wk.data()); return dot(v, wk.data(), N) / dot(v, v, N); };
    auto bisect = [&]() {
        double lo = -scale, hi = scale;
        for (int it = 0; it < 200; ++it) {
            const double mid = 0.5 * (lo + hi);
            if (mid <= lo || mid >= hi) break;
            ldl(h, N, w, mid, tiny, F); ++out.factorisations;
            if (F.negative > 0) hi = mid; else lo = mid;
            if (hi - lo <= 4e-16 * std::max(std::fabs(lo), std::fabs(hi)) + 1e-300) break;
        }
        return 0.5 * (lo + hi);
    };
    double theta = 0.0;
    const bool have_guess = guess_len > 0 && guess_len <= N;
    if (have_guess) {
        for (int i = 0; i < guess_len; ++i) s[i] = guess[i];
        // (the new rows are coupled to the old ones through the last block only: give them a small generic component so that the
        // first solve can move weight there)
        for (int i = guess_len; i < N; ++i) s[i] = 1e-3 / (1.0 + (i - guess_len));
        normalize(s, N);
    } else {
        for (int i = 0; i < N; ++i) s[i] = 1.0 / (1.0 + (i % 17));
        normalize(s, N);
    }
    if (rough) {
        ldl(h, N, w, -1e-9 * scale, tiny, F); ++out.factorisations;
        if (F.negative == 0) {
            for (int it = 0; it < 6; ++it) { solve(F, s); normalize(s, N); }
            out.theta = rq(s);
            return;
        }
    }
    if (have_guess) theta = std::min(theta_guess, rq(s));
    else theta = bisect();
    auto refine = [&](double th, int maxit) {
        for (int it = 0; it < maxit; ++it) {
            ldl(h, N, w, th, tiny, F); ++out.factorisations;
            solve(F, s);
            normalize(s, N);
            const double nt = rq(s);
            const bool done = std::fabs(nt - th) <= 8e-16 * scale;
            th = nt;
            if (done) break;
        }
        return th;
    };
    theta = refine(theta, 8);
    // verify it is the smallest: nothing below theta - delta
    const double delta = 1e-10 * scale;
    ldl(h, N, w, theta - delta, tiny, F); ++out.factorisations;
    if (F.negative != 0) {
        theta = bisect();
        for (int i = 0; i < N; ++i) s[i] = 1.0 / (1.0 + (i % 17));
        normalize(s, N);
        ldl(h, N, w, theta - 1e-13 * scale, tiny, F); ++out.factorisations;
        for (int it = 0; it < 3; ++it) { solve(F, s); normalize(s, N); }
        theta = std::min(theta, rq(s));
    }
    out.theta = theta;
}

// Jacobi eigen-decomposition of a small symmetric matrix (q <= 8): a (q x q, row-major) -> eigenvalues ascending in e, vectors in
// the columns of v.
inline void small_eig(double* a, int q, double* e, double* v) {
    for (int i = 0; i < q; ++i) for (int j = 0; j < q; ++j) v[i * q + j] = i == j ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 30; ++sweep) {
        double off = 0.0;
        for (int i = 0; i < q; ++i) for (int j = i + 1; j < q; ++j) off += a[i * q + j] * a[i * q + j];
        if (off < 1e-300) break;
        for (int p = 0; p < q; ++p) for (int r = p + 1; r < q; ++r) {
            const double apr = a[p * q + r];
            if (apr == 0.0) continue;
            const double tau = (a[r * q + r] - a[p * q + p]) / (2.0 * apr);
            const double t = (tau >= 0 ? 1.0 : -1.0) / (std::fabs(tau) + std::sqrt(1.0 + tau * tau));
            const double c = 1.0 / std::sqrt(1.0 + t * t), sn = t * c;
            for (int k = 0; k < q; ++k) {
                const double akp = a[k * q + p], akr = a[k * q + r];
                a[k * q + p] = c * akp - sn * akr; a[k * q + r] = sn * akp + c * akr;
            }
            for (int k = 0; k < q; ++k) {
                const double apk = a[p * q + k], ark = a[r * q + k];
                a[p * q + k] = c * apk - sn * ark; a[r * q + k] = sn * apk + c * ark;
            }
            for (int k = 0; k < q; ++k) {
                const double vkp = v[k * q + p], vkr = v[k * q + r];
                v[k * q + p] = c * vkp - sn * vkr; v[k * q + r] = sn * vkp + c * vkr;
            }
        }
    }
    for (int i = 0; i < q; ++i) e[i] = a[i * q + i];
    for (int i = 0; i < q; ++i) for (int j = i + 1; j < q; ++j) if (e[j] < e[i]) {
        std::swap(e[i], e[j]);
        for (int k = 0; k < q; ++k) std::swap(v[k * q + i], v[k * q + j]);
    }
}

// The q lowest eigenpairs by subspace iteration on (H - sigma I)^-1, sigma a little below theta1 (the smallest eigenvalue, known).
// S: column-major N x q, column 0 = the known eigenvector on entry; theta: q values out.  The extra columns only have to be GOOD
// START VECTORS for the next solve's block, so a few iterations suffice.
inline void lowest_block(const double* h, int N, int w, int q, double theta1, const double* s1, std::vector<double>& theta,
                         std::vector<double>& S, Factor& F, int iterations = 3) {
    q = std::min(q, N);
    theta.assign((size_t)q, theta1);
    S.assign((size_t)N * q, 0.0);
    const double scale = scale_of(h, N, w);
    const double tiny = 1e-300 + 1e-30 * scale;
    for (int i = 0; i < N; ++i) S[i] = s1[i];
    for (int c = 1; c < q; ++c) for (int i = 0; i < N; ++i) S[(size_t)c * N + i] = 1.0 / (1.0 + ((i * 7 + c * 3) % 11)) * ((i / (c + 1)) & 1 ? -1.0 : 1.0);
    const double sigma = theta1 - 0.05 * std::max(std::fabs(theta1), 1e-12 * scale);
    ldl(h, N, w, sigma, tiny, F);
    std::vector<double> HS((size_t)N * q), G((size_t)q * q), E((size_t)q), Y((size_t)q * q), T((size_t)N * q);
    for (int it = 0; it < iterations; ++it) {
        for (int c = (it == 0 ? 1 : 0); c < q; ++c) solve(F, S.data() + (size_t)c * N);
        for (int c = 0; c < q; ++c) {     // modified Gram-Schmidt
            double* sc = S.data() + (size_t)c * N;
            for (int p = 0; p < c; ++p) { const double* sp = S.data() + (size_t)p * N; const double d = dot(sp, sc, N); for (int i = 0; i < N; ++i) sc[i] -= d * sp[i]; }
            normalize(sc, N);
        }
        for (int c = 0; c < q; ++c) matvec(h, N, w, S.data() + (size_t)c * N, HS.data() + (size_t)c * N);
        for (int a = 0; a < q; ++a) for (int b = 0; b < q; ++b) G[(size_t)a * q + b] = dot(S.data() + (size_t)a * N, HS.data() + (size_t)b * N, N);
        for (int a = 0; a < q; ++a) for (int b = a + 1; b < q; ++b) G[(size_t)a * q + b] = G[(size_t)b * q + a] = 0.5 * (G[(size_t)a * q + b] + G[(size_t)b * q + a]);
        small_eig(G.data(), q, E.data(), Y.data());
        for (int c = 0; c < q; ++c) {
            double* tc = T.data() + (size_t)c * N;
            for (int i = 0; i < N; ++i) { double v = 0.0; for (int a = 0; a < q; ++a) v += S[(size_t)a * N + i] * Y[(size_t)a * q + c]; tc[i] = v; }
            theta[(size_t)c] = E[(size_t)c];
        }
        S.swap(T);
    }
}

}  // namespace band
}  // namespace machip
