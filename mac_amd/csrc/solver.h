// solver.h -- device-resident Lanczos solver for the Fiedler pair of a CSR Laplacian.
//
// Algorithm (replaces nx:151-256 TraceMIN + SuperLU, which the reference calls at
// mac/utils/fiedler.py:42): plain Lanczos on L restricted to 1-perp (the mean is projected out
// of every new vector, as nx:209-213 does), no re-orthogonalisation, the whole basis kept in HBM
// (288 GB makes that free) so the Ritz vector is one tall-skinny pass at the end.  The host only
// sees O(J) scalars per chunk of steps and runs the tridiagonal analysis (tridiag.h).
// Convergence is declared by the reference's own rule evaluated explicitly on the device:
//     || L v - lambda v ||_1 / || L ||_inf < tol          (nx:232, nx:246)
#pragma once
#include <vector>

#include "kernels.h"
#include "tridiag.h"

namespace machip {

template <class T>
inline int dev_alloc(T** p, size_t count) {
    *p = nullptr;
    if (count == 0) count = 1;
    HIP_TRY(hipMalloc((void**)p, count * sizeof(T)));
    return MACHIP_OK;
}

enum SpmvVariant { kAuto = 0, kStream = 1, kVec = 2 };

struct SpmvPlan {
    int variant = kVec;   // kStream or kVec
    int width = 8;        // TPR for stream, G for vec
    int grid = 1;
};

inline int env_int(const char* name, int dflt) {
    const char* s = getenv(name);
    return (s && *s) ? atoi(s) : dflt;
}

inline SpmvPlan plan_spmv(int n, long nnz, int forced_variant) {
    SpmvPlan pl;
    const double mean = n > 0 ? (double)nnz / (double)n : 1.0;
    int variant = forced_variant;
    if (variant == kAuto) {
        const char* e = getenv("MACHIP_SPMV");
        if (e && !strcmp(e, "stream")) variant = kStream;
        else if (e && !strcmp(e, "vec")) variant = kVec;
        else variant = mean < 24.0 ? kStream : kVec;
    }
    pl.variant = variant;
    if (variant == kStream) {
        int tpr = 16;
        while (tpr > 1 && ((long)n * tpr / kBlock > kMaxGrid || tpr > std::max(2.0, mean))) tpr >>= 1;
        tpr = env_int("MACHIP_TPR", tpr);
        pl.width = tpr;
        const int R = kBlock / tpr;
        pl.grid = (int)std::min<long>(kMaxGrid, ((long)n + R - 1) / R);
    } else {
        int g = 4;
        while (g < 64 && g < mean * 0.75) g <<= 1;
        g = env_int("MACHIP_G", g);
        pl.width = g;
        const int gpb = kBlock / g;
        pl.grid = (int)std::min<long>(kMaxGrid, ((long)n + gpb - 1) / gpb);
    }
    if (pl.grid < 1) pl.grid = 1;
    return pl;
}

template <class Op>
inline void launch_spmv(const SpmvPlan& pl, hipStream_t s, const CsrView& A, const double* x, const Op& op) {
    if (pl.variant == kStream) {
        switch (pl.width) {
            case 1: k_spmv_stream<1, Op><<<pl.grid, kBlock, 0, s>>>(A, x, op); break;
            case 2: k_spmv_stream<2, Op><<<pl.grid, kBlock, 0, s>>>(A, x, op); break;
            case 4: k_spmv_stream<4, Op><<<pl.grid, kBlock, 0, s>>>(A, x, op); break;
            case 8: k_spmv_stream<8, Op><<<pl.grid, kBlock, 0, s>>>(A, x, op); break;
            default: k_spmv_stream<16, Op><<<pl.grid, kBlock, 0, s>>>(A, x, op); break;
        }
    } else {
        switch (pl.width) {
            case 2: k_spmv_vec<2, Op><<<pl.grid, kBlock, 0, s>>>(A, x, op); break;
            case 4: k_spmv_vec<4, Op><<<pl.grid, kBlock, 0, s>>>(A, x, op); break;
            case 8: k_spmv_vec<8, Op><<<pl.grid, kBlock, 0, s>>>(A, x, op); break;
            case 16: k_spmv_vec<16, Op><<<pl.grid, kBlock, 0, s>>>(A, x, op); break;
            case 32: k_spmv_vec<32, Op><<<pl.grid, kBlock, 0, s>>>(A, x, op); break;
            default: k_spmv_vec<64, Op><<<pl.grid, kBlock, 0, s>>>(A, x, op); break;
        }
    }
}

struct Solver {
    int n = 0;
    hipStream_t stream = nullptr;
    size_t vcap = 0;            // Lanczos vectors that fit in V
    // Krylov state
    double *u = nullptr, *w = nullptr, *V = nullptr, *alpha = nullptr, *beta = nullptr, *l1 = nullptr;
    double *part_u = nullptr, *part_a = nullptr;
    LanState* st = nullptr;
    // explicit-check / result state
    double *y_raw = nullptr, *w2 = nullptr, *yvec = nullptr, *ypart = nullptr, *sdev = nullptr;
    double *part_c = nullptr, *part_a2 = nullptr, *part_r = nullptr, *scratch3 = nullptr, *rq_dev = nullptr;
    LanState* st2 = nullptr;
    double* start = nullptr;    // persistent cold-start vector
    bool have_start = false, have_prev = false;
    int ks_max = 16;
    // pinned host staging
    double* h_pin = nullptr;
    size_t h_pin_cap = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // host copies of T
    std::vector<double> ha, hb, hl1;
    std::vector<double> wk, guess;
    tri::Smallest sm;
    int J_last = 0;             // dimension of the Krylov space behind the current yvec

    int init(int n_, hipStream_t s) {
        n = n_;
        stream = s;
        size_t budget = (size_t)env_int("MACHIP_VBUDGET_MB", 4096) * (size_t)(1 << 20);
        vcap = budget / (sizeof(double) * (size_t)std::max(n, 1));
        vcap = std::max<size_t>(std::min<size_t>(vcap, 16384), 64);
        vcap = (size_t)env_int("MACHIP_VCAP", (int)vcap);
        ST_TRY(dev_alloc(&u, n)); ST_TRY(dev_alloc(&w, n));
        ST_TRY(dev_alloc(&V, (size_t)n * vcap));
        ST_TRY(dev_alloc(&alpha, vcap + 2)); ST_TRY(dev_alloc(&beta, vcap + 2)); ST_TRY(dev_alloc(&l1, vcap + 2));
        ST_TRY(dev_alloc(&part_u, 3 * kMaxGrid)); ST_TRY(dev_alloc(&part_a, kMaxGrid));
        ST_TRY(dev_alloc(&st, 1)); ST_TRY(dev_alloc(&st2, 1));
        ST_TRY(dev_alloc(&y_raw, n)); ST_TRY(dev_alloc(&w2, n)); ST_TRY(dev_alloc(&yvec, n));
        ST_TRY(dev_alloc(&ypart, (size_t)n * ks_max)); ST_TRY(dev_alloc(&sdev, vcap + 2));
        ST_TRY(dev_alloc(&part_c, 3 * kMaxGrid)); ST_TRY(dev_alloc(&part_a2, kMaxGrid));
        ST_TRY(dev_alloc(&part_r, kMaxGrid)); ST_TRY(dev_alloc(&scratch3, 8)); ST_TRY(dev_alloc(&rq_dev, 1));
        ST_TRY(dev_alloc(&start, n));
        h_pin_cap = 3 * (vcap + 2) + 4 * kMaxGrid + 64;
        HIP_TRY(hipHostMalloc((void**)&h_pin, h_pin_cap * sizeof(double), hipHostMallocDefault));
        HIP_TRY(hipEventCreate(&ev0)); HIP_TRY(hipEventCreate(&ev1));
        return MACHIP_OK;
    }
    void destroy() {
        double* ptrs[] = {u, w, V, alpha, beta, l1, part_u, part_a, y_raw, w2, yvec, ypart, sdev,
                          part_c, part_a2, part_r, scratch3, rq_dev, start};
        for (double* p : ptrs) if (p) (void)hipFree(p);
        if (st) (void)hipFree(st);
        if (st2) (void)hipFree(st2);
        if (h_pin) (void)hipHostFree(h_pin);
        if (ev0) (void)hipEventDestroy(ev0);
        if (ev1) (void)hipEventDestroy(ev1);
    }

    int vgrid() const { return (int)std::min<long>(kMaxGrid, ((long)n + kBlock - 1) / kBlock); }

    LanView view(const SpmvPlan& pl) const {
        LanView L;
        L.n = n; L.st = st; L.u = u; L.w = w; L.V = V; L.alpha = alpha; L.beta = beta; L.l1 = l1;
        L.part_u = part_u; L.P_u = vgrid(); L.part_a = part_a; L.P_a = pl.grid;
        return L;
    }
    LanView check_view(const SpmvPlan& pl) const {   // "column 0" machinery for the explicit check
        LanView L;
        L.n = n; L.st = st2; L.u = y_raw; L.w = w2; L.V = yvec; L.alpha = scratch3; L.beta = scratch3 + 2;
        L.l1 = scratch3 + 4; L.part_u = part_c; L.P_u = vgrid(); L.part_a = part_a2; L.P_a = pl.grid;
        return L;
    }

    // Enqueue `steps` Lanczos steps + the tail kernel.
    void enqueue_steps(const CsrView& A, const SpmvPlan& pl, const LanView& L, int steps) {
        OpLanczos op;
        op.L = L;
        const int g2 = vgrid();
        for (int s = 0; s < steps; ++s) {
            launch_spmv(pl, stream, A, L.u, op);
            k_lan_update<<<g2, kBlock, 0, stream>>>(L);
        }
        k_lan_tail<<<1, kBlock, 0, stream>>>(L);
    }

    // y = V[:, :J] s  -> normalised into yvec; w2 = L yvec; returns (rq, ||w2 - rq yvec||_1).
    int explicit_check(const CsrView& A, const SpmvPlan& pl, int J, const double* s_host, double* rq,
                       double* res_l1) {
        memcpy(h_pin, s_host, sizeof(double) * (size_t)J);
        HIP_TRY(hipMemcpyAsync(sdev, h_pin, sizeof(double) * (size_t)J, hipMemcpyHostToDevice, stream));
        const int g2 = vgrid();
        const int KS = std::max(1, std::min(ks_max, J / 8));
        k_ritz_partial<<<dim3(g2, KS), kBlock, 0, stream>>>(V, n, J, sdev, ypart);
        k_ritz_combine<<<g2, kBlock, 0, stream>>>(ypart, n, KS, y_raw, part_c);
        return check_vector(A, pl, rq, res_l1);
    }
    // Same, for a vector already in y_raw with its sums in part_c.
    int check_vector(const CsrView& A, const SpmvPlan& pl, double* rq, double* res_l1) {
        const int g2 = vgrid();
        k_set_state<<<1, 64, 0, stream>>>(st2, 0);
        OpLanczos op;
        op.L = check_view(pl);
        launch_spmv(pl, stream, A, y_raw, op);
        k_resid_l1<<<g2, kBlock, 0, stream>>>(w2, yvec, n, part_a2, pl.grid, part_r, rq_dev);
        double* hp = h_pin + 3 * (vcap + 2);
        HIP_TRY(hipMemcpyAsync(hp, part_r, sizeof(double) * (size_t)g2, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipMemcpyAsync(hp + kMaxGrid, rq_dev, sizeof(double), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        double s = 0.0;
        for (int i = 0; i < g2; ++i) s += hp[i];
        *res_l1 = s;
        *rq = hp[kMaxGrid];
        return MACHIP_OK;
    }

    // start_mode: 0 = stored cold-start vector (or device pseudo-random if none), 1 = previous
    // Fiedler vector (warm start).
    int solve(const CsrView& A, long nnz, double lnorm, double tol, int max_steps, int start_mode,
              int forced_variant, double* lambda2, machip_solve_stats* stats) {
        const SpmvPlan pl = plan_spmv(n, nnz, forced_variant);
        const int g2 = vgrid();
        HIP_TRY(hipEventRecord(ev0, stream));
        if (max_steps <= 0) max_steps = 200000;
        long steps_total = 0, spmv_total = 0, restarts = 0;
        int status = MACHIP_NOT_CONVERGED;
        double lam = 0.0, res = 0.0;
        const double tiny_l = (lnorm > 0 ? lnorm : 1.0);

        if (n == 1) { *lambda2 = 0.0; return fail(MACHIP_BAD_ARG, "graph with a single node has no Fiedler pair"); }

        // ---- start vector ----
        if (start_mode == 1 && have_prev) {
            HIP_TRY(hipMemcpyAsync(u, yvec, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, stream));
        } else if (have_start) {
            HIP_TRY(hipMemcpyAsync(u, start, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, stream));
        } else {
            k_fill_start<<<g2, kBlock, 0, stream>>>(u, n, 0x1234567ull);
        }
        bool fresh = true;   // u holds the (re)start vector, sums not yet taken

        const int min_chunk = std::max(1, env_int("MACHIP_CHUNK", 16));
        const double trigger_slack = 1.5;   // run the explicit check a little early rather than late
        double last_check_est = 1e300;

        while (steps_total < max_steps) {
            // ---- (re)start a Krylov sequence from u ----
            if (fresh) {
                k_vec_sums<<<g2, kBlock, 0, stream>>>(u, n, part_u);
                k_set_state<<<1, 64, 0, stream>>>(st, 0);
                fresh = false;
            }
            const LanView L = view(pl);
            int J = 0;
            ha.clear(); hb.assign(1, 0.0); hl1.assign(1, 0.0);
            guess.clear();
            double theta_prev = 0.0;
            bool converged = false, need_restart = false;
            const int jcap = (int)std::min<size_t>(vcap - 1, (size_t)std::max(2, n - 1) + 8);
            last_check_est = 1e300;
            while (!converged) {
                int chunk = std::max(min_chunk, J / 8);
                chunk = std::min(chunk, jcap - J);
                chunk = (int)std::min<long>(chunk, max_steps - steps_total);
                if (chunk <= 0) { need_restart = true; break; }
                enqueue_steps(A, pl, L, chunk);
                // scalars of this chunk: alpha[J..J+chunk), beta[J..J+chunk], l1[J..J+chunk]
                double* hp = h_pin;
                HIP_TRY(hipMemcpyAsync(hp, alpha + J, sizeof(double) * (size_t)chunk, hipMemcpyDeviceToHost, stream));
                HIP_TRY(hipMemcpyAsync(hp + vcap + 2, beta + J, sizeof(double) * (size_t)(chunk + 1), hipMemcpyDeviceToHost, stream));
                HIP_TRY(hipMemcpyAsync(hp + 2 * (vcap + 2), l1 + J, sizeof(double) * (size_t)(chunk + 1), hipMemcpyDeviceToHost, stream));
                HIP_TRY(hipStreamSynchronize(stream));
                ha.resize((size_t)J + chunk); hb.resize((size_t)J + chunk + 1); hl1.resize((size_t)J + chunk + 1);
                for (int i = 0; i < chunk; ++i) ha[(size_t)J + i] = hp[i];
                for (int i = 0; i <= chunk; ++i) {
                    hb[(size_t)J + i] = hp[vcap + 2 + i];
                    hl1[(size_t)J + i] = hp[2 * (vcap + 2) + i];
                }
                steps_total += chunk; spmv_total += chunk;
                const int Jold = J;
                J += chunk;
                // ---- breakdown: beta_j ~ 0 means span(v_0..v_{j-1}) is invariant ----
                int Jeff = J;
                bool broke = false;
                for (int j = std::max(1, Jold); j <= J; ++j) {
                    if (!(hb[(size_t)j] > 1e-13 * tiny_l)) { Jeff = j; broke = true; break; }
                }
                if (hb[0] <= 0.0 || !(hb[0] == hb[0])) {
                    return fail(MACHIP_BAD_ARG, "start vector is constant, zero or not finite");
                }
                // ---- host: smallest Ritz pair of T_Jeff ----
                tri::smallest_eigpair(ha.data(), hb.data(), Jeff, guess.data(), (int)guess.size(), theta_prev, sm, wk);
                guess = sm.s;
                theta_prev = sm.theta;
                const double rho = broke ? 0.0 : std::fabs(hb[(size_t)Jeff] * sm.s[(size_t)Jeff - 1]);
                const double est = rho * hl1[(size_t)Jeff];   // predicted ||r||_1
                const bool at_cap = (J >= jcap) || (steps_total >= max_steps);
                const bool trig = broke || est < trigger_slack * tol * lnorm;
                if ((trig && est < 0.5 * last_check_est) || broke || at_cap) {
                    double rq = 0.0, r1 = 0.0;
                    ST_TRY(explicit_check(A, pl, Jeff, sm.s.data(), &rq, &r1));
                    J_last = Jeff;
                    spmv_total += 1;
                    last_check_est = std::max(est, 1e-300);
                    lam = rq;
                    res = lnorm > 0 ? r1 / lnorm : r1;
                    if (res < tol) { converged = true; status = MACHIP_OK; break; }
                    if (broke || at_cap) { need_restart = true; break; }
                }
            }
            if (converged) break;
            if (need_restart) {
                if (steps_total >= max_steps) break;
                // restart from the best Ritz vector found so far (it sits normalised in yvec)
                HIP_TRY(hipMemcpyAsync(u, yvec, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, stream));
                fresh = true;
                ++restarts;
                if (restarts > 64) break;
            }
        }
        have_prev = true;
        HIP_TRY(hipEventRecord(ev1, stream));
        HIP_TRY(hipEventSynchronize(ev1));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, ev0, ev1));
        *lambda2 = lam;
        if (stats) {
            stats->lanczos_steps = steps_total;
            stats->spmv_total = spmv_total;
            stats->vec_passes = steps_total * 7;   // K1: u, v_{j-1} read; w, v_j written. K2: w, v_j, v_{j-1} read; u written
            stats->restarts = restarts;
            stats->nnz = nnz;
            stats->residual = res;
            stats->lnorm = lnorm;
            stats->gpu_ms = ms;
        }
        if (status == MACHIP_OK && lam < 1e-12 * tiny_l) {
            return fail(MACHIP_DISCONNECTED, "lambda_2 ~ 0: the graph is not connected");
        }
        if (status != MACHIP_OK) return fail(MACHIP_NOT_CONVERGED, "Lanczos hit the step cap before the residual test passed");
        return MACHIP_OK;
    }

    // q Ritz vectors (column-major n x q) of the last Krylov sequence -> host buffer.  Column 0 is
    // the converged Fiedler vector in yvec; the others are the next Ritz vectors, orthonormalised
    // on the host; copies produced by Lanczos "ghost" Ritz values are skipped.
    int ritz_block(int q, double* X_host) {
        const int Jeff = std::max(1, std::min(J_last, (int)ha.size()));
        const int ncand = std::min(Jeff, q + 6);
        std::vector<double> th, S;
        tri::smallest_block(ha.data(), hb.data(), Jeff, ncand, th, S, wk);
        const int g2 = vgrid();
        HIP_TRY(hipMemcpyAsync(X_host, yvec, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        int acc = 1;
        std::vector<double> col((size_t)n);
        for (int c = 1; c < (int)th.size() && acc < q; ++c) {
            memcpy(h_pin, S.data() + (size_t)c * Jeff, sizeof(double) * (size_t)Jeff);
            HIP_TRY(hipMemcpyAsync(sdev, h_pin, sizeof(double) * (size_t)Jeff, hipMemcpyHostToDevice, stream));
            const int KS = std::max(1, std::min(ks_max, Jeff / 8));
            k_ritz_partial<<<dim3(g2, KS), kBlock, 0, stream>>>(V, n, Jeff, sdev, ypart);
            k_ritz_combine<<<g2, kBlock, 0, stream>>>(ypart, n, KS, y_raw, part_c);
            HIP_TRY(hipMemcpyAsync(col.data(), y_raw, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, stream));
            HIP_TRY(hipStreamSynchronize(stream));
            double mean = 0.0, n0 = 0.0;
            for (int i = 0; i < n; ++i) mean += col[(size_t)i];
            mean /= n;
            for (int i = 0; i < n; ++i) { col[(size_t)i] -= mean; n0 += col[(size_t)i] * col[(size_t)i]; }
            for (int pass = 0; pass < 2; ++pass)
                for (int p = 0; p < acc; ++p) {
                    const double* xp = X_host + (size_t)p * n;
                    double dot = 0.0;
                    for (int i = 0; i < n; ++i) dot += xp[i] * col[(size_t)i];
                    for (int i = 0; i < n; ++i) col[(size_t)i] -= dot * xp[i];
                }
            double n2 = 0.0;
            for (int i = 0; i < n; ++i) n2 += col[(size_t)i] * col[(size_t)i];
            if (!(n2 > 1e-6 * std::max(n0, 1e-300))) continue;   // ghost copy of an accepted vector
            const double inv = 1.0 / std::sqrt(n2);
            double* dst = X_host + (size_t)acc * n;
            for (int i = 0; i < n; ++i) dst[i] = col[(size_t)i] * inv;
            ++acc;
        }
        // Krylov space exhausted (tiny or highly symmetric graph): complete the block with
        // deterministic vectors orthogonal to 1 and to the accepted columns, so X is always an
        // orthonormal n x q block like the reference's (nx:238).
        for (unsigned long long seed = 1; acc < q && seed < 64; ++seed) {
            double mean = 0.0;
            for (int i = 0; i < n; ++i) {
                unsigned long long z = seed * 0xD1B54A32D192ED03ull + 0x9E3779B97F4A7C15ull * (unsigned long long)(i + 1);
                z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
                z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
                z = z ^ (z >> 31);
                col[(size_t)i] = (double)(z >> 11) * (2.0 / 9007199254740992.0) - 1.0;
                mean += col[(size_t)i];
            }
            mean /= n;
            for (int i = 0; i < n; ++i) col[(size_t)i] -= mean;
            for (int pass = 0; pass < 2; ++pass)
                for (int p = 0; p < acc; ++p) {
                    const double* xp = X_host + (size_t)p * n;
                    double dot = 0.0;
                    for (int i = 0; i < n; ++i) dot += xp[i] * col[(size_t)i];
                    for (int i = 0; i < n; ++i) col[(size_t)i] -= dot * xp[i];
                }
            double n2 = 0.0;
            for (int i = 0; i < n; ++i) n2 += col[(size_t)i] * col[(size_t)i];
            if (!(n2 > 1e-12)) continue;
            const double inv = 1.0 / std::sqrt(n2);
            double* dst = X_host + (size_t)acc * n;
            for (int i = 0; i < n; ++i) dst[i] = col[(size_t)i] * inv;
            ++acc;
        }
        for (; acc < q; ++acc) {
            double* dst = X_host + (size_t)acc * n;
            for (int i = 0; i < n; ++i) dst[i] = 0.0;
        }
        return MACHIP_OK;
    }
};

}  // namespace machip
