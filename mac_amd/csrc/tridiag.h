// tridiag.h -- host-side analysis of the Lanczos tridiagonal T_J (J x J, symmetric):
// smallest eigenpairs by Sturm-count bisection + Rayleigh-quotient / inverse iteration.
// Plain C++ (no HIP): O(J) per pass, it runs on the host while the GPU executes the next
// chunk of Lanczos steps.  a[0..J) diagonal, b[1..J) sub-diagonal (b[i] couples i-1, i).
#pragma once
#include <algorithm>
#include <cmath>
#include <vector>

namespace machip {
namespace tri {

// number of eigenvalues of T_J strictly below x
inline int sturm_count(const double* a, const double* b, int J, double x, double tiny) {
    int cnt = 0;
    double d = a[0] - x;
    if (d == 0.0) d = -tiny;
    cnt += d < 0.0;
    for (int i = 1; i < J; ++i) {
        d = (a[i] - x) - b[i] * b[i] / d;
        if (d == 0.0) d = -tiny;
        cnt += d < 0.0;
    }
    return cnt;
}

inline double scale_of(const double* a, const double* b, int J) {
    double s = 0.0;
    for (int i = 0; i < J; ++i) s = std::max(s, std::fabs(a[i]) + (i ? std::fabs(b[i]) : 0.0));
    return s > 0.0 ? s : 1.0;
}

// Solve (T - shift I) z = rhs by Gaussian elimination with partial pivoting (tridiagonal).
// rhs is overwritten with z.  Near-singular pivots are perturbed (inverse iteration wants that).
inline void shifted_solve(const double* a, const double* b, int J, double shift, double* rhs,
                          std::vector<double>& wk, double tiny) {
    wk.resize((size_t)4 * J);
    double* d = wk.data();        // main diagonal (working)
    double* du = d + J;           // first super-diagonal
    double* du2 = du + J;         // second super-diagonal (fill-in from pivoting)
    double* dl = du2 + J;         // sub-diagonal
    for (int i = 0; i < J; ++i) {
        d[i] = a[i] - shift;
        du[i] = (i + 1 < J) ? b[i + 1] : 0.0;
        dl[i] = (i + 1 < J) ? b[i + 1] : 0.0;
        du2[i] = 0.0;
    }
    for (int i = 0; i + 1 < J; ++i) {
        if (std::fabs(d[i]) >= std::fabs(dl[i])) {
            if (d[i] == 0.0) d[i] = tiny;
            const double f = dl[i] / d[i];
            d[i + 1] -= f * du[i];
            rhs[i + 1] -= f * rhs[i];
            // du2[i] stays 0
        } else {   // swap rows i and i+1
            const double f = d[i] / dl[i];
            d[i] = dl[i];
            const double t = d[i + 1];
            d[i + 1] = du[i] - f * t;
            du[i] = t;
            if (i + 2 < J) {
                du2[i] = du[i + 1];
                du[i + 1] = -f * du[i + 1];
            }
            const double r = rhs[i];
            rhs[i] = rhs[i + 1];
            rhs[i + 1] = r - f * rhs[i + 1];
        }
    }
    if (std::fabs(d[J - 1]) < tiny) d[J - 1] = d[J - 1] < 0 ? -tiny : tiny;
    rhs[J - 1] /= d[J - 1];
    if (J >= 2) rhs[J - 2] = (rhs[J - 2] - du[J - 2] * rhs[J - 1]) / d[J - 2];
    for (int i = J - 3; i >= 0; --i)
        rhs[i] = (rhs[i] - du[i] * rhs[i + 1] - du2[i] * rhs[i + 2]) / d[i];
}

inline double rayleigh(const double* a, const double* b, int J, const double* s) {
    double num = 0.0, den = 0.0;
    for (int i = 0; i < J; ++i) {
        double t = a[i] * s[i];
        if (i) t += b[i] * s[i - 1];
        if (i + 1 < J) t += b[i + 1] * s[i + 1];
        num += s[i] * t;
        den += s[i] * s[i];
    }
    return num / den;
}

inline void normalize(double* s, int J) {
    double n2 = 0.0;
    for (int i = 0; i < J; ++i) n2 += s[i] * s[i];
    const double inv = n2 > 0.0 ? 1.0 / std::sqrt(n2) : 0.0;
    for (int i = 0; i < J; ++i) s[i] *= inv;
}

// k-th smallest eigenvalue (k = 0 is the smallest) to full precision by bisection.
inline double kth_eig_bisect(const double* a, const double* b, int J, int k, double scale) {
    double lo = a[0], hi = a[0];
    for (int i = 0; i < J; ++i) {
        const double r = (i ? std::fabs(b[i]) : 0.0) + (i + 1 < J ? std::fabs(b[i + 1]) : 0.0);
        lo = std::min(lo, a[i] - r);
        hi = std::max(hi, a[i] + r);
    }
    const double tiny = 1e-300 + 1e-30 * scale;
    for (int it = 0; it < 200; ++it) {
        const double mid = 0.5 * (lo + hi);
        if (mid <= lo || mid >= hi) break;
        if (sturm_count(a, b, J, mid, tiny) > k) hi = mid; else lo = mid;
        if (hi - lo <= 2e-16 * std::max(std::fabs(lo), std::fabs(hi))) break;
    }
    return 0.5 * (lo + hi);
}

struct Smallest {
    double theta = 0.0;
    std::vector<double> s;   // unit eigenvector, length J
    int passes = 0;          // O(J) passes spent (diagnostic)
};

// Smallest eigenpair of T_J.  `guess`/`guess_len`: eigenvector of a leading sub-block from
// the previous call (zero-padded here) with its eigenvalue `theta_guess`; pass guess_len = 0
// for a cold start.
inline void smallest_eigpair(const double* a, const double* b, int J, const double* guess,
                             int guess_len, double theta_guess, Smallest& out,
                             std::vector<double>& wk) {
    out.s.assign((size_t)J, 0.0);
    out.passes = 0;
    if (J == 1) { out.theta = a[0]; out.s[0] = 1.0; return; }
    const double scale = scale_of(a, b, J);
    const double tiny = 1e-300 + 1e-30 * scale;
    double* s = out.s.data();
    double theta;
    bool have_guess = guess_len > 0 && guess_len <= J;
    if (have_guess) {
        for (int i = 0; i < guess_len; ++i) s[i] = guess[i];
        normalize(s, J);
        theta = std::min(theta_guess, rayleigh(a, b, J, s));
    } else {
        theta = kth_eig_bisect(a, b, J, 0, scale);
        out.passes += 60;
        for (int i = 0; i < J; ++i) s[i] = 1.0 / (1.0 + i);   // generic start
        normalize(s, J);
    }
    auto refine = [&](double th) {
        for (int it = 0; it < 8; ++it) {
            shifted_solve(a, b, J, th, s, wk, tiny);
            normalize(s, J);
            const double nt = rayleigh(a, b, J, s);
            out.passes += 4;
            const bool done = std::fabs(nt - th) <= 8e-16 * scale;
            th = nt;
            if (done) break;
        }
        return th;
    };
    theta = refine(theta);
    // verify it is the smallest: no eigenvalue below theta - delta
    const double delta = 1e-10 * scale;
    out.passes += 1;
    if (sturm_count(a, b, J, theta - delta, tiny) != 0) {
        theta = kth_eig_bisect(a, b, J, 0, scale);
        out.passes += 60;
        for (int i = 0; i < J; ++i) s[i] = 1.0 / (1.0 + i);
        normalize(s, J);
        // fixed-shift inverse iteration (shift just below the eigenvalue), then one RQ polish
        for (int it = 0; it < 3; ++it) {
            shifted_solve(a, b, J, theta - 1e-13 * scale, s, wk, tiny);
            normalize(s, J);
            out.passes += 3;
        }
        theta = std::min(theta, rayleigh(a, b, J, s));
    }
    out.theta = theta;
}

// q smallest eigenpairs (for the Ritz block X): bisection + inverse iteration with
// Gram-Schmidt against the earlier ones.  S is column-major J x q.
inline void smallest_block(const double* a, const double* b, int J, int q, std::vector<double>& theta,
                           std::vector<double>& S, std::vector<double>& wk) {
    q = std::min(q, J);
    theta.assign((size_t)q, 0.0);
    S.assign((size_t)J * q, 0.0);
    const double scale = scale_of(a, b, J);
    const double tiny = 1e-300 + 1e-30 * scale;
    for (int k = 0; k < q; ++k) {
        theta[k] = kth_eig_bisect(a, b, J, k, scale);
        double* s = S.data() + (size_t)k * J;
        for (int i = 0; i < J; ++i) s[i] = 1.0 / (1.0 + ((i * 7 + k * 3) % 11));
        for (int it = 0; it < 4; ++it) {
            for (int p = 0; p < k; ++p) {
                const double* sp = S.data() + (size_t)p * J;
                double dot = 0.0;
                for (int i = 0; i < J; ++i) dot += sp[i] * s[i];
                for (int i = 0; i < J; ++i) s[i] -= dot * sp[i];
            }
            normalize(s, J);
            shifted_solve(a, b, J, theta[k] + (k + 1) * 1e-14 * scale, s, wk, tiny);
            normalize(s, J);
        }
        for (int p = 0; p < k; ++p) {
            const double* sp = S.data() + (size_t)p * J;
            double dot = 0.0;
            for (int i = 0; i < J; ++i) dot += sp[i] * s[i];
            for (int i = 0; i < J; ++i) s[i] -= dot * sp[i];
        }
        normalize(s, J);
    }
}

}  // namespace tri
}  // namespace machip
