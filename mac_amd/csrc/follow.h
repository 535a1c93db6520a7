// follow.h -- the host's reading of a Lanczos sequence through its streamed records (round 5).
//
// Every fused Lanczos step hands (alpha_{j-1}, ||v_{j-1}||_1, beta_j) to the host as it derives them (PipeView::pubstep,
// kernels.h).  A Follower analyses the tridiagonal T_a at a sequence of points a_0 < a_1 < ... that is a function of the
// records alone: after the analysis at a -- smallest Ritz pair of T_a (tridiag.h), residual estimate
// est_a = beta_a |s_a| ||v_{a-1}||_1 (what the reference's test ||L y - rho y||_1 will measure, nx:232, 246), slope of ln(est)
// over a window that shrinks with the forecast distance -- the next point is a third of the predicted distance to the target
// away, at most one chunk.  The sequence ENDS at the first point whose estimate is below the target.  Nothing here depends on
// how far the GPU has run ahead, on chunk sizes or on timing: two solves that produce the same records end at the same point
// with the same Ritz pair, whatever fed their queues (solver.h: the timing-driven feeder of the unpartitioned solve, the
// chunk-at-a-time feeder of the row-partitioned ones).  Plain C++ (no HIP): machip_host_follow_records exports it for the
// `-m "not gpu"` tests.
#pragma once
#include <algorithm>
#include <climits>
#include <cmath>
#include <deque>
#include <utility>
#include <vector>
#include "tridiag.h"

namespace machip {

struct FollowCfg {
    int n = 0;             // rows of the matrix (||v||_1 falls back to sqrt(n) when a record carries none)
    int jcap = INT_MAX;    // the sequence cannot grow beyond this many steps (basis capacity): the last point
    int chunk0 = 32;       // farthest apart two points may be (twice that from step 4 096 on: the O(J) analysis must keep up)
    int win_max = 64;      // widest slope window, in steps
    int margin = 0;        // steps added to the forecast
    double tiny_l = 1.0;   // ||L||_inf (1 when unknown): scale of the breakdown test
};

struct Follower {
    FollowCfg c;
    double e_target;                     // a check needs the estimate below this
    // the solver keeps these (the explicit check takes the Ritz coefficients, the Ritz block the tridiagonal)
    std::vector<double>&ha, &hb, &hl1, &guess;
    tri::Smallest& sm;
    std::vector<double>& wk;
    std::deque<std::pair<int, double>> hist;   // (point, ln est)
    int next_a;                          // the next analysis point
    int Jold = 0;                        // records below this point are in ha / hb / hl1
    int T = INT_MAX;                     // forecast: no step >= T is wanted (INT_MAX: no forecast yet)
    int Jeff = 0;                        // order of the analysed tridiagonal (< the point after a breakdown)
    double to_go = 1e18, est = 1e300, theta_prev = 0.0;
    bool broke = false;

    Follower(const FollowCfg& cfg, double target, std::vector<double>& a, std::vector<double>& b, std::vector<double>& l1,
             std::vector<double>& g, tri::Smallest& s, std::vector<double>& w)
        : c(cfg), e_target(target), ha(a), hb(b), hl1(l1), guess(g), sm(s), wk(w), next_a(std::min(16, cfg.jcap)) {}

    // Analyse T_a.  tri3: interleaved (alpha_j, beta_j, ||v_j||_1) triples, valid through beta_a.  Returns false when the start
    // vector was constant, zero or not finite (beta_0).  Afterwards: Jeff, est, broke, sm (the Ritz pair), to_go, T, next_a.
    bool analyse(const volatile double* tri3, int a) {
        const int J = a;
        ha.resize((size_t)J); hb.resize((size_t)J + 1); hl1.resize((size_t)J + 1);
        for (int j = std::max(0, Jold - 1); j < J; ++j) { ha[(size_t)j] = tri3[3 * (size_t)j]; hl1[(size_t)j] = tri3[3 * (size_t)j + 2]; }
        for (int j = Jold; j <= J; ++j) hb[(size_t)j] = tri3[3 * (size_t)j + 1];
        if (Jold == 0 && (hb[0] <= 0.0 || !(hb[0] == hb[0]))) return false;
        // breakdown: beta_j ~ 0 means span(v_0 .. v_{j-1}) is invariant
        Jeff = J; broke = false;
        for (int j = std::max(1, Jold); j <= J; ++j)
            if (!(hb[(size_t)j] > 1e-13 * c.tiny_l)) { Jeff = j; broke = true; break; }
        Jold = J;
        tri::smallest_eigpair(ha.data(), hb.data(), Jeff, guess.data(), (int)guess.size(), theta_prev, sm, wk);
        guess = sm.s;
        theta_prev = sm.theta;
        const double rho = broke ? 0.0 : std::fabs(hb[(size_t)Jeff] * sm.s[(size_t)Jeff - 1]);
        const double l1v = hl1[(size_t)Jeff - 1] > 0 ? hl1[(size_t)Jeff - 1] : std::sqrt((double)c.n);
        est = rho * l1v;      // predicted ||r||_1 (r = rho v_J; ||v_J||_1 ~ ||v_{J-1}||_1)
        // forecast: the convergence accelerates, so a long window under-estimates the current rate -- it shrinks with the distance
        const double lt = std::log(std::max(e_target, 1e-300));
        if (!broke && est > 0.0) {
            hist.emplace_back(J, std::log(est));
            const int win = (to_go < 1e17) ? std::max(12, std::min(c.win_max, (int)(2.0 * to_go))) : c.win_max;
            while (hist.size() > 2 && hist[1].first <= J - win) hist.pop_front();
            to_go = 1e18;
            if (hist.front().first < J) {
                const double slope = (hist.front().second - hist.back().second) / (double)(J - hist.front().first);
                if (slope > 1e-7) to_go = std::max(0.0, (hist.back().second - lt) / slope);
            }
        }
        T = to_go < 1e17 ? (int)std::min<double>(2e9, (double)J + std::ceil(to_go) + c.margin) : INT_MAX;
        const int far = J >= 4096 ? 2 * c.chunk0 : (J < 64 ? 16 : c.chunk0);
        const int stride = to_go < 1e17 ? std::max(1, std::min(far, (int)(to_go / 3.0))) : far;
        next_a = std::min(J + stride, c.jcap);
        return true;
    }
    bool triggered() const { return broke || est < e_target; }
    // a check at this point failed (the estimate flattered the residual): further down before the next one
    void lower_target(double t) { e_target = std::min(e_target, t); }
};

}  // namespace machip
