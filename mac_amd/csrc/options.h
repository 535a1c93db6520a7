// options.h -- per-handle option table (round 5).
//
// Rounds 1-4 selected kernels, launch shapes and thresholds through ~65 process-wide MACHIP_* environment variables, read with
// getenv() at plan time on every solve: a drop-in library whose behaviour depended on the host's environment, and tests that
// switched code paths by mutating os.environ between calls.  Now every knob is an entry of this table:
//   * a handle carries its own copy (machip_set_option / machip_get_option, include/machip.h); evaluation lanes inherit their
//     owner's; a value of MACHIP_OPTION_AUTO means "the measured default" (each call site states it);
//   * the PROCESS defaults -- what a new handle starts from -- are read from the environment ONCE, when the library is first used
//     (MACHIP_<NAME IN CAPITALS>: developer use -- sweeps, A/B runs of tools/), and can be changed with machip_set_option(NULL, ..);
//   * the reference's only selector is one string, `fiedler_method` (mac/utils/fiedler.py:38-42): machip_set_solver stays its mirror.
#pragma once
#include <climits>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>

namespace machip {

// name list: lower case here, MACHIP_<UPPER> in the environment
#define MACHIP_OPTION_LIST(X)                                                                                                      \
    /* eigen-solver driver */                                                                                                      \
    X(solver) X(graph) X(debug) X(chunk) X(chunk_near) X(sched) X(near_x10) X(trigger_pct) X(classic_n) X(persist) X(pchunk)      \
    X(spec_epilogue) X(spec_slack_pct) X(f32_switch_e9) X(tailless) X(stream) X(stream_trigger_pct) X(stream_retry_pct) X(stream_look) X(stream_margin) X(stream_window) X(stream_far) X(start_land) X(start_pow) X(start_floor_e6) X(start_land_pr_permille) X(start_land_min_mean10)                                                                \
    /* launch shape of the fused step / stand-alone products */                                                                    \
    X(spmv) X(g) X(unroll) X(block) X(maxgrid) X(defer) X(tpr) X(ell)                                                              \
    /* column-panel step */                                                                                                        \
    X(panel) X(panel_min_n) X(panel_min_mean10) X(panel_multi_min_mean10) X(panel_np) X(panel_nb) X(panel_b2) X(panel_g2)         \
    X(panel_cells) X(panel_maxcells) X(panel_band) X(panel_u) X(panel_u_amp) X(pan32) X(pan32_switch_e9) X(panel_ops) X(panel_rev)                                                                               \
    /* preconditioned / exact modes */                                                                                             \
    X(woodbury) X(wb_max) X(wb_hard) X(wb_lane_max) X(exact_big) X(exact_switch) X(gj_look_min) X(lob_density_pct) X(lob_chunk)   \
    X(lob_patience) X(lob_small_s) X(lob_fuse)                                                                                     \
    /* select / assembly / lanes / communicators (the handle-creation ones are read when the handle, or its first lane, is made) */ \
    X(sel_small) X(sel_fuse) X(asm_g) X(asm_maxgrid) X(vbudget_mb) X(vcap) X(lanes) X(lane_vbudget_mb) X(lane_queues) X(shard_eig) X(ipc_panel) \
    X(rccl_timeout_s)                                                                                                              \
    /* experiments (compiled in with -DMACHIP_EXPERIMENTS only: tools/) */                                                         \
    X(cheb_deg) X(cheb_after) X(cheb_chunk) X(cheb_depth) X(panel_fused) X(panel_spin_us) X(lob_pan2)                              \
    X(blocklan) X(blocklan_min_steps) X(blk_chunk) X(blk_chunk_near)

enum OptId {
#define X(n) kOpt_##n,
    MACHIP_OPTION_LIST(X)
#undef X
    kNumOpts
};

constexpr long kOptAuto = LONG_MIN;      // == MACHIP_OPTION_AUTO of include/machip.h

inline const char* const* option_names() {
    static const char* const names[] = {
#define X(n) #n,
        MACHIP_OPTION_LIST(X)
#undef X
        nullptr};
    return names;
}
inline int option_id(const char* name) {
    if (!name) return -1;
    const char* const* nm = option_names();
    for (int i = 0; i < kNumOpts; ++i) if (!strcmp(nm[i], name)) return i;
    return -1;
}

struct Options {
    long v[kNumOpts];
    int get(OptId id, int dflt) const { return v[id] == kOptAuto ? dflt : (int)v[id]; }
    bool is_set(OptId id) const { return v[id] != kOptAuto; }
};

// words some variables used to take (kept for the environment defaults)
inline long option_parse(int id, const char* s) {
    if (id == kOpt_solver) {
        if (!strcmp(s, "auto")) return 0;
        if (!strcmp(s, "lanczos")) return 1;
        if (!strcmp(s, "lobpcg")) return 2;
        if (!strcmp(s, "jacobi")) return 3;
    }
    if (id == kOpt_spmv) {
        if (!strcmp(s, "stream")) return 1;
        if (!strcmp(s, "vec")) return 2;
    }
    if (id == kOpt_lane_queues) {
        if (!strcmp(s, "shared")) return 1;
        if (!strcmp(s, "cumask")) return 0;
    }
    return atol(s);
}

// process defaults: the environment, read once; machip_set_option(NULL, ...) edits them afterwards
inline Options& default_options() {
    static Options D;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* const* nm = option_names();
        for (int i = 0; i < kNumOpts; ++i) {
            D.v[i] = kOptAuto;
            std::string e = "MACHIP_";
            for (const char* c = nm[i]; *c; ++c) e += (char)(*c >= 'a' && *c <= 'z' ? *c - 32 : *c);
            const char* s = getenv(e.c_str());
            if (s && *s) D.v[i] = option_parse(i, s);
        }
    });
    return D;
}

#define OPT(name, dflt) (opt.get(kOpt_##name, (dflt)))

}  // namespace machip
