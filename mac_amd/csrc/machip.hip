// machip.hip -- C ABI of libmachip.so (see include/machip.h).  gfx950 only.
#include <rccl/rccl.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <future>
#include <memory>
#include <mutex>
#include <numeric>
#include <thread>
#include <vector>

#include "solver.h"

namespace machip {
thread_local std::string g_err;
}
using namespace machip;

#define NCCL_TRY(expr)                                                                       \
    do {                                                                                     \
        ncclResult_t r__ = (expr);                                                           \
        if (r__ != ncclSuccess)                                                              \
            return fail(MACHIP_RCCL_ERROR, std::string(#expr) + ": " + ncclGetErrorString(r__)); \
    } while (0)

// First-contact watchdog (round 4): a call into RCCL that may block for ever -- ncclCommInitRank with a peer that never
// arrives, the first collective on a fabric that is not what was assumed -- runs on a helper thread; the caller waits with a
// limit (MACHIP_RCCL_TIMEOUT_S, default 600) and turns a stall into MACHIP_RCCL_ERROR with the rank in the message instead of
// a hung job.  (A helper that never returns is abandoned with its state: the process is about to report failure anyway.)
static int run_with_watchdog(const std::function<int()>& fn, double limit_s, const std::string& what) {
    auto state = std::make_shared<std::promise<int>>();
    std::future<int> fut = state->get_future();
    std::thread([state, fn] { int r = MACHIP_HIP_ERROR; try { r = fn(); } catch (...) {} state->set_value(r); }).detach();
    if (fut.wait_for(std::chrono::duration<double>(limit_s)) != std::future_status::ready)
        return fail(MACHIP_RCCL_ERROR, what + " did not return within " + std::to_string((int)limit_s) + " s (a peer rank is missing, or the fabric / bootstrap interface is not reachable; MACHIP_RCCL_TIMEOUT_S raises the limit)");
    return fut.get();
}
static double rccl_timeout_s() { const Options& opt = default_options(); const int t = OPT(rccl_timeout_s, 600); return t > 0 ? (double)t : 600.0; }

// In-process communicator (machip_comm_init_local): the ranks are handles of ONE process driven by one host
// thread each (one GPU per handle, or several handles on one GPU); the all-gather is peer-to-peer device copies
// between the handles' gradient buffers, bracketed by a host barrier.  Same shard arithmetic and the same call
// site as the RCCL communicator.
struct machip_problem;
struct LocalGroup {
    int nranks = 0;
    std::vector<double*> g;     // every rank's gradient buffer (device)
    // row-partitioned eigen-solve (solver.h ShardGroup): rank 0 leads, its results are handed to the peers
    std::vector<machip_problem*> members;
    ShardGroup sg;
    bool shard_eig = false;
    int lead_status = MACHIP_OK;
    double lead_lam = 0.0;
    machip_solve_stats lead_stats{};
    std::string lead_err;
    ~LocalGroup() {
        for (ShardRank& r : sg.rk) for (hipEvent_t e : r.ev) if (e) (void)hipEventDestroy(e);
        if (sg.fork) (void)hipEventDestroy(sg.fork);
    }
    std::mutex mu;
    std::condition_variable cv;
    int waiting = 0;
    unsigned long generation = 0;
    bool broken = false;
    // returns false when the group was broken (a rank failed or was destroyed): never blocks forever
    bool barrier() {
        std::unique_lock<std::mutex> lk(mu);
        if (broken) return false;
        const unsigned long gen = generation;
        if (++waiting == nranks) { waiting = 0; ++generation; cv.notify_all(); return true; }
        cv.wait(lk, [&] { return generation != gen || broken; });
        return !broken;
    }
    void abort() {
        std::lock_guard<std::mutex> lk(mu);
        broken = true;
        cv.notify_all();
    }
};

struct machip_problem {
    int device = 0;
    hipStream_t stream = nullptr;
    int n = 0;
    long m = 0, m_pad = 0;
    double tol_sel = 1e-10;
    // pattern (device)
    int *prow = nullptr, *pcol = nullptr, *pk = nullptr;
    double* pw = nullptr;
    long P = 0;
    int asm_G = 16, asm_rpb = 1, asm_grid = 1;
    // candidates (device)
    int *ci = nullptr, *cj = nullptr;
    double* cw = nullptr;
    double *x = nullptr, *x_next = nullptr, *g = nullptr, *s = nullptr, *scratch_m = nullptr;
    // assembled CSR (device)
    int *cnt = nullptr, *blk_sum = nullptr, *rowptr = nullptr, *col = nullptr;
    double *val = nullptr, *blk_lnorm = nullptr;
    unsigned long long *xb = nullptr, *xb_next = nullptr;     // support bitmaps of x / x_next (kernels.h k_x_bits)
    bool xb_valid = false, xb_next_valid = false;
    long nnz = 0, support = 0;
    int maxlen = 0;           // longest row of the assembled L(x)
    double lnorm = 0.0;
    bool assembled = false, have_vec = false, csr_only = false;
    bool band_dups_ok = true;    // no row holds more than 7 slots of columns r - 1 / r + 1 (the panel form packs that count in 3 bits)
    // select / FW scalars
    unsigned int* hist = nullptr;
    SelState* sel = nullptr;
    double* part_fw = nullptr;
    // pinned host
    int* h_int = nullptr;        // mapped pinned memory; d_hint / d_hdbl = the device's view of it
    double* h_dbl = nullptr;
    int* d_hint = nullptr;
    double* d_hdbl = nullptr;
    Solver sol;
    // multi-GPU
    ncclComm_t comm = nullptr;
    std::shared_ptr<LocalGroup> lgroup;
    bool first_collective_pending = false;   // the communicator's first ncclAllGather is awaited with a time limit
    hipEvent_t ev_g0 = nullptr, ev_g1 = nullptr, ev_g2 = nullptr;   // around the last gradient kernel / exchange of a communicator (machip_comm_timing)
    bool ev_g_valid = false;
    std::unique_ptr<IpcGroup> ipcg;      // inter-process communicator with a row-partitioned eigen-solve (machip_comm_init_ipc)
    int rank = 0, nranks = 1;
    // evaluation lanes (machip_eval_batch): lightweight copies that share the pattern and the candidate arrays
    bool is_lane = false, lane_fw_ready = false;
    std::vector<machip_problem*> lanes;
    unsigned long start_version = 0;          // bumped whenever sol.start changes
    unsigned long seen_start_version = ~0ul;  // (lane) the owner's start_version this lane last copied

    CsrView csr() const { return CsrView{n, rowptr, col, val}; }
    PatternView pattern() const { return PatternView{n, prow, pcol, pk, pw}; }
};

// ROCm maps a process's HIP streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); kernels of streams that share a queue
// run one after another.  The concurrent budget sweep keeps up to 12 single-CU solves in flight (machip_fw_sweep, one stream per
// lane): on 4 shared queues it saturates at ~2.8x, with a queue per lane at 5-7x (intel 1 764 -> 4 155 it/s).  Round 3 raised
// the process-wide variable when the library was loaded -- a side effect on every other HIP user of the process, and silently
// ineffective once HIP was initialised.  Round 4: a lane's stream is created WITH A CU MASK (all CUs enabled): the runtime
// gives a CU-masked stream a hardware queue of its own (the mask is a property of the queue) instead of one from the shared
// pool -- per stream, no environment, no effect on anybody else.  MACHIP_LANE_QUEUES=shared takes plain streams.
static int create_lane_stream(const Options& opt, int device, hipStream_t* out) {
    if (OPT(lane_queues, 0) == 0) {      // 0: CU-masked stream with a hardware queue of its own; 1: plain stream on the shared pool
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) {
            const int words = (prop.multiProcessorCount + 31) / 32;
            std::vector<uint32_t> mask((size_t)words, 0xffffffffu);
            if (prop.multiProcessorCount % 32) mask.back() = (1u << (prop.multiProcessorCount % 32)) - 1u;
            if (hipExtStreamCreateWithCUMask(out, (uint32_t)words, mask.data()) == hipSuccess) return MACHIP_OK;
            (void)hipGetLastError();     // (no such queue available: a plain stream serves, more slowly)
        }
    }
    HIP_TRY(hipStreamCreateWithFlags(out, hipStreamNonBlocking));
    return MACHIP_OK;
}

namespace {

std::mutex g_csr_mu;                        // machip_fiedler_csr's cached handle
machip_problem* g_csr_cache = nullptr;

struct Slot {
    int col;
    int k;
    double w;
};

int build_pattern(machip_problem* p, int64_t nf, const int32_t* fi, const int32_t* fj, const double* fw,
                  int64_t m, const int32_t* ci, const int32_t* cj, const double* cw) {
    const Options& opt = p->sol.opt;
    const int n = p->n;
    std::vector<long> deg((size_t)n + 1, 0);
    auto chk = [&](int a, int b) { return a >= 0 && a < n && b >= 0 && b < n; };
    for (int64_t e = 0; e < nf; ++e) {
        if (!chk(fi[e], fj[e])) return fail(MACHIP_BAD_ARG, "fixed edge endpoint out of range");
        if (fi[e] != fj[e]) { deg[(size_t)fi[e]]++; deg[(size_t)fj[e]]++; }
    }
    for (int64_t e = 0; e < m; ++e) {
        if (!chk(ci[e], cj[e])) return fail(MACHIP_BAD_ARG, "candidate edge endpoint out of range");
        if (ci[e] != cj[e]) { deg[(size_t)ci[e]]++; deg[(size_t)cj[e]]++; }
    }
    std::vector<long> off((size_t)n + 1, 0);
    for (int r = 0; r < n; ++r) off[(size_t)r + 1] = off[(size_t)r] + deg[(size_t)r];
    std::vector<Slot> slots((size_t)off[(size_t)n]);
    std::vector<long> cur(off.begin(), off.end() - 1);
    for (int64_t e = 0; e < nf; ++e) {
        const int a = fi[e], b = fj[e];
        if (a == b) continue;   // a self-loop contributes +w -w = 0 (graphs.py:27-46)
        slots[(size_t)cur[(size_t)a]++] = Slot{b, -1, fw[e]};
        slots[(size_t)cur[(size_t)b]++] = Slot{a, -1, fw[e]};
    }
    for (int64_t e = 0; e < m; ++e) {
        const int a = ci[e], b = cj[e];
        if (a == b) continue;
        slots[(size_t)cur[(size_t)a]++] = Slot{b, (int)e, cw[e]};
        slots[(size_t)cur[(size_t)b]++] = Slot{a, (int)e, cw[e]};
    }
    // sort each row by (col, fixed first, k); merge duplicate fixed slots (coo->csr sums them)
    std::vector<int> prow((size_t)n + 1, 0);
    std::vector<int> pcol, pk;
    std::vector<double> pw;
    pcol.reserve(slots.size()); pk.reserve(slots.size()); pw.reserve(slots.size());
    for (int r = 0; r < n; ++r) {
        Slot* b = slots.data() + off[(size_t)r];
        Slot* e = slots.data() + off[(size_t)r + 1];
        std::sort(b, e, [](const Slot& x, const Slot& y) {
            if (x.col != y.col) return x.col < y.col;
            return x.k < y.k;
        });
        for (Slot* q = b; q < e; ++q) {
            if (q->k < 0 && !pcol.empty() && (long)pcol.size() > prow[(size_t)r] && pcol.back() == q->col &&
                pk.back() < 0) {
                pw.back() += q->w;
            } else {
                pcol.push_back(q->col); pk.push_back(q->k); pw.push_back(q->w);
            }
        }
        if (pcol.size() > 2000000000ull) return fail(MACHIP_BAD_ARG, "pattern exceeds int32 indexing");
        {
            int band_slots = 0;
            for (size_t q = (size_t)prow[(size_t)r]; q < pcol.size(); ++q) band_slots += (pcol[q] == r - 1 || pcol[q] == r + 1);
            if (band_slots > 7) p->band_dups_ok = false;
        }
        prow[(size_t)r + 1] = (int)pcol.size();
    }
    p->P = (long)pcol.size();
    ST_TRY(dev_alloc(&p->prow, (size_t)n + 1));
    ST_TRY(dev_alloc(&p->pcol, (size_t)p->P)); ST_TRY(dev_alloc(&p->pk, (size_t)p->P)); ST_TRY(dev_alloc(&p->pw, (size_t)p->P));
    // (every copy of this library names the handle's own stream: the legacy stream would serialise with -- and, while a lane
    // captures a chunk graph, fail against -- every blocking stream of the process; the lanes' CU-masked streams are blocking)
    HIP_TRY(hipMemcpyAsync(p->prow, prow.data(), sizeof(int) * ((size_t)n + 1), hipMemcpyHostToDevice, p->stream));
    if (p->P) {
        HIP_TRY(hipMemcpyAsync(p->pcol, pcol.data(), sizeof(int) * (size_t)p->P, hipMemcpyHostToDevice, p->stream));
        HIP_TRY(hipMemcpyAsync(p->pk, pk.data(), sizeof(int) * (size_t)p->P, hipMemcpyHostToDevice, p->stream));
        HIP_TRY(hipMemcpyAsync(p->pw, pw.data(), sizeof(double) * (size_t)p->P, hipMemcpyHostToDevice, p->stream));
    }
    HIP_TRY(hipStreamSynchronize(p->stream));      // (the host vectors go out of scope)
    // assembly launch shape: G lanes per row ~ half the mean pattern degree
    const double mean = n ? (double)p->P / n : 1.0;
    int G = 4;
    while (G < 64 && G < mean * 0.75) G <<= 1;
    p->asm_G = OPT(asm_g, G);
    const int gpb = kBlock / p->asm_G;
    // (round 5: up to 4 096 workgroups -- with 1 024 a G-lane group walked 13 rows one after the other at configs[3], each a chain of
    // dependent loads: prow -> pk -> x[pk]; the pass was latency-bound at a quarter of the chip's memory-level parallelism)
    long nblk = std::min<long>(std::min(kAsmGrid, std::max(1, OPT(asm_maxgrid, kAsmGrid))), ((long)n + gpb - 1) / gpb);
    if (nblk < 1) nblk = 1;
    long rpb = ((long)n + nblk - 1) / nblk;
    rpb = (rpb + gpb - 1) / gpb * gpb;
    p->asm_rpb = (int)rpb;
    p->asm_grid = (int)(((long)n + rpb - 1) / rpb);
    if (p->asm_grid < 1) p->asm_grid = 1;
    // k_asm_fill scans its rows' counts in (rpb + 1) ints of dynamic LDS: stay inside the 64 KB every launch may use
    if ((rpb + 1) * (long)sizeof(int) > 64 * 1024)
        return fail(MACHIP_BAD_ARG, "num_nodes above the supported 16.7 million (row-offset scan of the assembly kernel)");
    return MACHIP_OK;
}

template <int G>
void launch_asm(machip_problem* p, const PanSpec& S) {
    const PatternView P = p->pattern();
    const unsigned int* xb32 = reinterpret_cast<const unsigned int*>(p->xb);
    k_asm_count<G><<<p->asm_grid, kBlock, 0, p->stream>>>(P, xb32, p->asm_rpb, p->cnt, p->blk_sum, p->d_hint);
    const size_t lds = sizeof(int) * ((size_t)p->asm_rpb + 1);
    k_asm_fill<G><<<p->asm_grid, kBlock, lds, p->stream>>>(P, xb32, p->x, p->asm_rpb, p->cnt, p->blk_sum,
                                                           p->rowptr, p->col, p->val, p->d_hdbl, S);   // (row-sum maxima: host only)
}

int assemble(machip_problem* p) {
    if (p->csr_only) return fail(MACHIP_BAD_ARG, "handle wraps a caller CSR; nothing to assemble");
    // the fill pass writes the column-panel form's per-row tables on the side where that form can run at this n (solver.h)
    PanSpec S;
    ST_TRY(p->sol.pan_spec_prepare(&S));
    if (!p->xb_valid) {     // x came from the host (or a copy): the support bitmap is taken here; k_fw_final leaves the next iterate's
        const int grid = std::max(1, (int)std::min<long>(kMaxGrid, (p->m + kBlock - 1) / kBlock));
        k_x_bits<<<grid, kBlock, 0, p->stream>>>(p->x, p->m, p->tol_sel, p->xb);
        p->xb_valid = true;
    }
    switch (p->asm_G) {
        case 4: launch_asm<4>(p, S); break;
        case 8: launch_asm<8>(p, S); break;
        case 16: launch_asm<16>(p, S); break;
        case 32: launch_asm<32>(p, S); break;
        default: launch_asm<64>(p, S); break;
    }
    HIP_TRY(hipGetLastError());       // a refused launch must not leave stale row offsets behind a MACHIP_OK
    const int gb = p->asm_grid;
    // (both kernels wrote the host's copies into mapped pinned memory.  Round 5 tried a completion word published by the fill pass's
    // last workgroup -- system-scope fence + ticket per workgroup -- for the host to spin on instead of this wait: 3 125 fences made
    // the launch 312 us instead of 82)
    HIP_TRY(hipStreamSynchronize(p->stream));
    long nnz = 0, supp = 0;
    int maxlen = 0;
    double ln = 0.0;
    for (int b = 0; b < gb; ++b) {
        nnz += p->h_int[b]; supp += p->h_int[kAsmGrid + b]; maxlen = std::max(maxlen, p->h_int[2 * kAsmGrid + b]);
        ln = std::max(ln, p->h_dbl[b]);
    }
    p->nnz = nnz; p->support = supp / 2; p->lnorm = ln; p->maxlen = maxlen;     // (two pattern slots per active candidate)
    p->assembled = true;
    return MACHIP_OK;
}

int alloc_common(machip_problem* p, int vbudget_mb = 0) {
    ST_TRY(p->sol.init(p->n, p->stream, vbudget_mb));
    HIP_TRY(hipHostMalloc((void**)&p->h_int, sizeof(int) * 3 * kAsmGrid, hipHostMallocMapped));
    HIP_TRY(hipHostMalloc((void**)&p->h_dbl, sizeof(double) * std::max(4 * kMaxGrid, kAsmGrid), hipHostMallocMapped));
    HIP_TRY(hipHostGetDevicePointer((void**)&p->d_hint, p->h_int, 0));
    HIP_TRY(hipHostGetDevicePointer((void**)&p->d_hdbl, p->h_dbl, 0));
    return MACHIP_OK;
}

// k-th largest of keys[0..m): threshold, remaining rank and tie rule into *st.
// hist0_done: the first digit's histogram has been counted by the producer of the keys (k_grad<true>) behind a k_sel_init.
int select_on(machip_problem* p, const double* keys, long k, SelState* st, int prefer_high, bool hist0_done = false) {
    const Options& opt = p->sol.opt;
    const long m = p->m;
    if (k < 0) k = 0;
    if (k > m) k = m;
    if (m <= kSelSmallMax && OPT(sel_small, 1)) {      // short key vectors: the whole select in one launch
        k_sel_small<<<1, 1024, 0, p->stream>>>(keys, m, (long long)k, st, prefer_high);
        HIP_TRY(hipGetLastError());
        return MACHIP_OK;
    }
    if (!hist0_done) k_sel_init<<<1, 1024, 0, p->stream>>>(st, (long long)k, p->hist, 6 * kBins);
    if (k > 0) {
        // few, large workgroups: a dense digit costs one returning device-scope atomic per non-empty bin and workgroup (kernels.h)
        // (measured at configs[3], 16 MB of keys: 512 x 256 threads 17 us per dense pass, 128 x 1 024 threads 10 us)
        constexpr int B = 1024, U = 4;
        const int grid = (int)std::max<long>(1, std::min<long>(128, (m + (long)B * U - 1) / ((long)B * U)));
        if (hist0_done) k_sel_close0<<<1, 1024, 0, p->stream>>>(p->hist, st);
        for (int pass = hist0_done ? 1 : 0; pass < 6; ++pass) k_sel_pass<U, B><<<grid, B, 0, p->stream>>>(keys, m, pass, p->hist, st);
    }
    k_sel_ties<<<1, 1024, 0, p->stream>>>(keys, m, st, prefer_high);
    HIP_TRY(hipGetLastError());
    return MACHIP_OK;
}

int select_topk(machip_problem* p, long k, bool hist0_done = false) { return select_on(p, p->g, k, p->sel, 0, hist0_done); }

// Contiguous candidate ranges: shard = ceil(m / R); rank r owns [r shard, (r+1) shard) clipped to m;
// the gathered vector is padded to R shard entries.  The ONE place this arithmetic lives (machip_shard_plan exports it).
void shard_plan(long m, int nranks, int rank, long* lo, long* hi, long* shard) {
    const long sh = nranks > 0 ? (m + nranks - 1) / nranks : m;
    *shard = sh;
    *lo = std::min(m, sh * rank);
    *hi = std::min(m, *lo + sh);
}

int local_allgather(machip_problem* p, long shard) {
    LocalGroup& G = *p->lgroup;
    HIP_TRY(hipStreamSynchronize(p->stream));                 // my shard is in my buffer
    if (!G.barrier()) return fail(MACHIP_RCCL_ERROR, "in-process communicator was shut down by another rank");
    for (int r = 0; r < G.nranks; ++r)
        if (r != p->rank)
            HIP_TRY(hipMemcpyAsync(p->g + shard * r, G.g[(size_t)r] + shard * r, sizeof(double) * (size_t)shard,
                                   hipMemcpyDeviceToDevice, p->stream));
    HIP_TRY(hipStreamSynchronize(p->stream));
    if (!G.barrier()) return fail(MACHIP_RCCL_ERROR, "in-process communicator was shut down by another rank");   // nobody overwrites a shard a peer is still reading
    return MACHIP_OK;
}

// fuse_k >= 0 (single rank, long candidate lists): the first digit pass of the top-fuse_k select that follows is counted by the
// gradient kernel itself (k_grad<true>; *fused tells the caller so)
int compute_gradient(machip_problem* p, bool have_vec_now = false, long fuse_k = -1, bool* fused = nullptr) {
    const Options& opt = p->sol.opt;
    if (!p->have_vec && !have_vec_now) return fail(MACHIP_BAD_ARG, "no Fiedler vector on the device: call machip_fiedler first");
    const long m = p->m;
    long lo = 0, hi = m, shard = m;
    if (p->nranks > 1) shard_plan(m, p->nranks, p->rank, &lo, &hi, &shard);
    const bool ipc_gather = p->nranks > 1 && !p->comm && p->ipcg;     // no RCCL communicator (ranks sharing a GPU): shards by peer writes
    // IPC gather: my shard is written into every peer's gradient -- not before every peer is done with its gradient of the
    // previous call.  Channel 3 is that handshake: "I have arrived here" (everything this stream did with g before is complete),
    // then wait until every peer has arrived too.  Without an eigen-solve that is partitioned between the ranks nothing else
    // orders a fast rank against a slow one (round-4 advisor finding).
    if (ipc_gather) {
        k_ipc_pubwait<<<1, 64, 0, p->stream>>>(p->ipcg->view, 3);
    }
    const bool timed = p->nranks > 1;      // (communicators only: three event records per iteration)
    if (timed) {
        if (!p->ev_g0) { HIP_TRY(hipEventCreate(&p->ev_g0)); HIP_TRY(hipEventCreate(&p->ev_g1)); HIP_TRY(hipEventCreate(&p->ev_g2)); }
        p->ev_g_valid = false;
        HIP_TRY(hipEventRecord(p->ev_g0, p->stream));
    }
    if (hi > lo) {
        const int grid = (int)std::min<long>(kMaxGrid * 4, (hi - lo + kBlock - 1) / kBlock);
        PeerVecs all;
        if (ipc_gather) { all.n = p->ipcg->nranks; for (int q = 0; q < all.n; ++q) all.v[q] = p->ipcg->g[q]; }
        const bool fuse = fuse_k > 0 && p->nranks <= 1 && m > kSelSmallMax && OPT(sel_fuse, 1) != 0;
        if (fused) *fused = fuse;
        if (fuse) {
            k_sel_init<<<1, 1024, 0, p->stream>>>(p->sel, (long long)std::min(fuse_k, m), p->hist, (6 + kSelRep) * kBins);
            k_grad<true><<<grid, kBlock, 0, p->stream>>>(p->ci, p->cj, p->cw, p->sol.yvec, lo, hi, p->g, all, p->hist + 6 * kBins);
        } else k_grad<false><<<grid, kBlock, 0, p->stream>>>(p->ci, p->cj, p->cw, p->sol.yvec, lo, hi, p->g, all);
        HIP_TRY(hipGetLastError());
    }
    if (timed) HIP_TRY(hipEventRecord(p->ev_g1, p->stream));
    if (ipc_gather) {
        k_ipc_pubwait<<<1, 64, 0, p->stream>>>(p->ipcg->view, 2);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(p->ev_g2, p->stream));
        p->ev_g_valid = true;
        return MACHIP_OK;
    }
    if (p->nranks > 1) {
        if (p->comm) {
            NCCL_TRY(ncclAllGather(p->g + shard * p->rank, p->g, (size_t)shard, ncclDouble, p->comm, p->stream));
            if (p->first_collective_pending) {
                // the FIRST collective of a communicator is awaited with a limit: a fabric that cannot carry it must surface as an
                // error naming the rank, not as a job that never ends (later collectives are not polled)
                p->first_collective_pending = false;
                const auto t0 = std::chrono::steady_clock::now();
                const double lim = rccl_timeout_s();
                for (;;) {
                    const hipError_t q = hipStreamQuery(p->stream);
                    if (q == hipSuccess) break;
                    if (q != hipErrorNotReady) return fail(MACHIP_HIP_ERROR, std::string("first ncclAllGather: ") + hipGetErrorString(q));
                    ncclResult_t ar = ncclSuccess;
                    if (ncclCommGetAsyncError(p->comm, &ar) == ncclSuccess && ar != ncclSuccess && ar != ncclInProgress)
                        return fail(MACHIP_RCCL_ERROR, std::string("first ncclAllGather failed asynchronously: ") + ncclGetErrorString(ar));
                    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > lim) {
                        (void)ncclCommAbort(p->comm);
                        p->comm = nullptr;
                        return fail(MACHIP_RCCL_ERROR, "first ncclAllGather of rank " + std::to_string(p->rank) + " of " + std::to_string(p->nranks) +
                                    " did not complete within " + std::to_string((int)lim) + " s: communicator aborted");
                    }
                    std::this_thread::sleep_for(std::chrono::microseconds(200));
                }
            }
        }
        else if (p->lgroup) { const int st = local_allgather(p, shard); if (st != MACHIP_OK) { p->lgroup->abort(); return st; } }
        else return fail(MACHIP_BAD_ARG, "nranks > 1 without a communicator");
        HIP_TRY(hipEventRecord(p->ev_g2, p->stream));
        p->ev_g_valid = true;
    }
    return MACHIP_OK;
}

int run_fiedler(machip_problem* p, double tol, int max_steps, const double* x0, int warm_start,
                double* lambda2, machip_solve_stats* stats) {
    if (!p->assembled) ST_TRY(assemble(p));
    if (x0) {
        HIP_TRY(hipMemcpyAsync(p->sol.start, x0, sizeof(double) * (size_t)p->n, hipMemcpyHostToDevice, p->stream));
        HIP_TRY(hipStreamSynchronize(p->stream));
        p->sol.have_start = true;
        ++p->start_version;          // evaluation lanes must pick the new vector up (machip_eval_batch)
    }
    machip_solve_stats local;
    memset(&local, 0, sizeof(local));
    p->sol.maxlen_hint = p->maxlen;
    p->sol.support_hint = (p->csr_only && !p->sol.chain_like) ? -1 : p->support;
    p->sol.start_guess = x0 != nullptr;      // a caller's own start vector is taken as it is (no landscape weighting)
    const int st = p->sol.solve(p->csr(), p->nnz, p->lnorm, tol, max_steps, (x0 == nullptr && warm_start) ? 1 : 0,
                                kAuto, lambda2, &local);
    p->sol.start_guess = false;
    local.support = p->support;
    if (stats) *stats = local;
    p->have_vec = (st == MACHIP_OK || st == MACHIP_NOT_CONVERGED || st == MACHIP_DISCONNECTED);
    return st;
}

// Collective eigen-solve of an in-process communicator whose step is row-partitioned: every rank has assembled the same
// L(x); rank 0's solver drives all ranks' streams (solver.h, ShardGroup) and hands lambda_2, the statistics and the
// Fiedler vector to the peers.
int group_fiedler(machip_problem* p, double tol, int max_steps, int warm_start, double* lambda2, machip_solve_stats* stats) {
    LocalGroup& G = *p->lgroup;
    if (!G.barrier()) return fail(MACHIP_RCCL_ERROR, "in-process communicator was shut down by another rank");   // everyone's L(x) is assembled
    if (p->rank == 0) {
        for (int r = 0; r < G.nranks; ++r) {
            machip_problem* q = G.members[(size_t)r];
            ShardRank& k = G.sg.rk[(size_t)r];
            k.device = q->device; k.stream = q->stream; k.Z0 = q->sol.Z0; k.Z1 = q->sol.Z1; k.part = q->sol.part; k.V = q->sol.V;
            k.ypart = q->sol.ypart; k.sdev = q->sol.sdev; k.yvec = q->sol.yvec; k.A = q->csr();
        }
        machip_solve_stats local;
        memset(&local, 0, sizeof(local));
        p->sol.shard = &G.sg;
        int st = run_fiedler(p, tol, max_steps, nullptr, warm_start, &G.lead_lam, &local);
        p->sol.shard = nullptr;
        if (st == MACHIP_OK || st == MACHIP_NOT_CONVERGED || st == MACHIP_DISCONNECTED) {
            const std::string keep = g_err;
            for (int r = 1; r < G.nranks && (st == MACHIP_OK || st == MACHIP_NOT_CONVERGED || st == MACHIP_DISCONNECTED); ++r)
                if (hipMemcpyAsync(G.sg.rk[(size_t)r].yvec, p->sol.yvec, sizeof(double) * (size_t)p->n, hipMemcpyDeviceToDevice, p->stream) != hipSuccess)
                    st = fail(MACHIP_HIP_ERROR, "copy of the Fiedler vector to a peer failed");
            if (hipStreamSynchronize(p->stream) != hipSuccess) st = fail(MACHIP_HIP_ERROR, "stream error after the row-partitioned solve");
            else g_err = keep;
        }
        G.lead_status = st; G.lead_stats = local; G.lead_err = g_err;
    }
    if (!G.barrier()) return fail(MACHIP_RCCL_ERROR, "in-process communicator was shut down by another rank");
    const int st = G.lead_status;
    *lambda2 = G.lead_lam;
    if (stats) { *stats = G.lead_stats; stats->support = p->support; }
    if (p->rank != 0) {
        g_err = G.lead_err;
        p->have_vec = (st == MACHIP_OK || st == MACHIP_NOT_CONVERGED || st == MACHIP_DISCONNECTED);
    }
    return st;
}

}  // namespace

// ------------------------------------------------------------------------------------------
extern "C" {

int machip_version(void) { return MACHIP_ABI_VERSION; }
int machip_sizeof_stats(void) { return (int)sizeof(machip_solve_stats); }

int machip_device_count(void) {
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess) return 0;
    return c;
}

const char* machip_last_error(void) { return g_err.c_str(); }

int machip_create(int device, int64_t n, int64_t n_fixed, const int32_t* fi, const int32_t* fj, const double* fw,
                  int64_t m, const int32_t* ci, const int32_t* cj, const double* cw, double min_sel_tol,
                  machip_problem** out) {
    if (!out) return fail(MACHIP_BAD_ARG, "out is NULL");
    *out = nullptr;
    if (n < 2 || n > 2000000000ll) return fail(MACHIP_BAD_ARG, "num_nodes must be in [2, 2e9]");
    if (n_fixed < 0 || m < 0 || m > 2000000000ll) return fail(MACHIP_BAD_ARG, "bad edge counts");
    if ((n_fixed && (!fi || !fj || !fw)) || (m && (!ci || !cj || !cw))) return fail(MACHIP_BAD_ARG, "NULL edge array");
    // mac/solvers/mac.py:46-52
    if (n_fixed + m < n - 1) return fail(MACHIP_BAD_ARG, "fewer than n-1 edges: no spanning tree possible");
    if ((double)(n_fixed + m) > 0.5 * (double)n * (double)(n - 1)) return fail(MACHIP_BAD_ARG, "more edges than the complete graph");
    if (machip_device_count() <= 0) return fail(MACHIP_NO_DEVICE, "no HIP device visible");
    HIP_TRY(hipSetDevice(device));
    machip_problem* p = new machip_problem();
    p->device = device;
    p->n = (int)n;
    p->m = (long)m;
    p->m_pad = (long)m;
    p->tol_sel = min_sel_tol;
    int st = MACHIP_OK;
    auto body = [&]() -> int {
        HIP_TRY(hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking));
        ST_TRY(build_pattern(p, n_fixed, fi, fj, fw, m, ci, cj, cw));
        const size_t mp = (size_t)m + 64 * 8;   // slack so a later all-gather padding fits
        ST_TRY(dev_alloc(&p->ci, mp)); ST_TRY(dev_alloc(&p->cj, mp)); ST_TRY(dev_alloc(&p->cw, mp));
        ST_TRY(dev_alloc(&p->x, mp)); ST_TRY(dev_alloc(&p->x_next, mp)); ST_TRY(dev_alloc(&p->g, mp)); ST_TRY(dev_alloc(&p->s, mp));
        if (m) {
            HIP_TRY(hipMemcpyAsync(p->ci, ci, sizeof(int) * (size_t)m, hipMemcpyHostToDevice, p->stream));
            HIP_TRY(hipMemcpyAsync(p->cj, cj, sizeof(int) * (size_t)m, hipMemcpyHostToDevice, p->stream));
            HIP_TRY(hipMemcpyAsync(p->cw, cw, sizeof(double) * (size_t)m, hipMemcpyHostToDevice, p->stream));
        }
        HIP_TRY(hipMemsetAsync(p->x, 0, sizeof(double) * mp, p->stream));
        HIP_TRY(hipMemsetAsync(p->g, 0, sizeof(double) * mp, p->stream));
        HIP_TRY(hipStreamSynchronize(p->stream));      // (the caller's arrays are borrowed for the duration of the call only)
        const size_t cap = (size_t)p->P + (size_t)n + 8;
        ST_TRY(dev_alloc(&p->cnt, (size_t)n + 1)); ST_TRY(dev_alloc(&p->blk_sum, 3 * kAsmGrid));
        ST_TRY(dev_alloc(&p->rowptr, (size_t)n + 1)); ST_TRY(dev_alloc(&p->col, cap)); ST_TRY(dev_alloc(&p->val, cap));
        ST_TRY(dev_alloc(&p->blk_lnorm, kAsmGrid));
        ST_TRY(dev_alloc(&p->xb, (size_t)m / 64 + 2)); ST_TRY(dev_alloc(&p->xb_next, (size_t)m / 64 + 2));
        ST_TRY(dev_alloc(&p->hist, (6 + kSelRep) * kBins)); ST_TRY(dev_alloc(&p->sel, 2)); ST_TRY(dev_alloc(&p->part_fw, 2 * kMaxGrid));
        ST_TRY(alloc_common(p));
        p->sol.csr_cap = cap;
        p->sol.pat = p->pattern();
        p->sol.pan_allowed = p->band_dups_ok;     // panel.h walks rows as this library assembles them (diagonal first)
        // pose-graph shape: do the fixed edges contain (nearly) the whole chain (i, i+1)?  Decides
        // whether the preconditioned eigen-solver mode is considered (solver.h, precond.h).
        {
            std::vector<char> has((size_t)n, 0);
            for (int64_t e = 0; e < n_fixed; ++e) {
                const int a = std::min(fi[e], fj[e]), b = std::max(fi[e], fj[e]);
                if (a >= 0 && b < n && b == a + 1 && fw[e] > 0.0) has[(size_t)a] = 1;
            }
            long cnt = 0;
            for (int64_t i = 0; i + 1 < n; ++i) cnt += has[(size_t)i];
            p->sol.chain_like = cnt >= (long)(0.98 * (double)(n - 1));
            p->sol.chain_edges = cnt;
        }
        return MACHIP_OK;
    };
    st = body();
    if (st != MACHIP_OK) { machip_destroy(p); return st; }
    *out = p;
    return MACHIP_OK;
}

void machip_destroy(machip_problem* p) {
    if (!p) return;
    for (machip_problem* q : p->lanes) machip_destroy(q);
    p->lanes.clear();
    (void)hipSetDevice(p->device);
    if (p->stream) (void)hipStreamSynchronize(p->stream);
    if (p->is_lane) { p->prow = p->pcol = p->pk = nullptr; p->pw = nullptr; p->ci = p->cj = nullptr; p->cw = nullptr; }   // borrowed
    if (p->comm) (void)ncclCommDestroy(p->comm);
    if (p->lgroup) p->lgroup->abort();      // peers blocked in the group's barrier return an error instead of hanging
    if (p->ipcg) {
        IpcGroup& G = *p->ipcg;
        if (!G.clean_exit) { k_ipc_abort<<<1, 64, 0, p->stream>>>(G.view); (void)hipStreamSynchronize(p->stream); }   // peers' waits end with an error, not a time-out
        p->sol.ipc = nullptr;
        p->ipcg.reset();                   // (~IpcGroup closes the mappings and frees the flag / error words)
    }
    p->sol.destroy();
    void* ptrs[] = {p->prow, p->pcol, p->pk, p->pw, p->ci, p->cj, p->cw, p->x, p->x_next, p->g, p->s, p->scratch_m, p->cnt,
                    p->blk_sum, p->rowptr, p->col, p->val, p->blk_lnorm, p->xb, p->xb_next, p->hist, p->sel, p->part_fw};
    for (void* q : ptrs) if (q) (void)hipFree(q);
    for (hipEvent_t e : {p->ev_g0, p->ev_g1, p->ev_g2}) if (e) (void)hipEventDestroy(e);
    if (p->h_int) (void)hipHostFree(p->h_int);
    if (p->h_dbl) (void)hipHostFree(p->h_dbl);
    if (p->stream) (void)hipStreamDestroy(p->stream);
    delete p;
}

int machip_set_x(machip_problem* p, const double* x) {
    if (!p || !x || p->csr_only) return fail(MACHIP_BAD_ARG, "machip_set_x: bad handle or NULL x");
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(hipMemcpyAsync(p->x, x, sizeof(double) * (size_t)p->m, hipMemcpyHostToDevice, p->stream));
    HIP_TRY(hipStreamSynchronize(p->stream));
    p->assembled = false; p->xb_valid = false;
    return MACHIP_OK;
}

int machip_get_x(machip_problem* p, double* x) {
    if (!p || !x || p->csr_only) return fail(MACHIP_BAD_ARG, "machip_get_x: bad handle or NULL x");
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(hipMemcpyAsync(x, p->x, sizeof(double) * (size_t)p->m, hipMemcpyDeviceToHost, p->stream));
    HIP_TRY(hipStreamSynchronize(p->stream));
    return MACHIP_OK;
}

int machip_assemble(machip_problem* p, int64_t* nnz_out) {
    if (!p) return fail(MACHIP_BAD_ARG, "NULL handle");
    HIP_TRY(hipSetDevice(p->device));
    ST_TRY(assemble(p));
    if (nnz_out) *nnz_out = p->nnz;
    return MACHIP_OK;
}

int machip_get_laplacian(machip_problem* p, int32_t* indptr, int32_t* indices, double* data) {
    if (!p || !indptr || !indices || !data) return fail(MACHIP_BAD_ARG, "NULL argument");
    HIP_TRY(hipSetDevice(p->device));
    if (!p->assembled) ST_TRY(assemble(p));
    HIP_TRY(hipMemcpyAsync(indptr, p->rowptr, sizeof(int) * ((size_t)p->n + 1), hipMemcpyDeviceToHost, p->stream));
    HIP_TRY(hipMemcpyAsync(indices, p->col, sizeof(int) * (size_t)p->nnz, hipMemcpyDeviceToHost, p->stream));
    HIP_TRY(hipMemcpyAsync(data, p->val, sizeof(double) * (size_t)p->nnz, hipMemcpyDeviceToHost, p->stream));
    HIP_TRY(hipStreamSynchronize(p->stream));
    return MACHIP_OK;
}

int machip_fiedler(machip_problem* p, double tol, int max_steps, const double* x0, int warm_start,
                   double* lambda2, double* v_out, double* X_out, int q, machip_solve_stats* stats) {
    if (!p || !lambda2) return fail(MACHIP_BAD_ARG, "NULL argument");
    if (X_out && (q < 1 || q > 4)) return fail(MACHIP_BAD_ARG, "q must be in [1,4]");
    HIP_TRY(hipSetDevice(p->device));
    const int st = run_fiedler(p, tol, max_steps, x0, warm_start, lambda2, stats);
    if (st != MACHIP_OK && st != MACHIP_NOT_CONVERGED && st != MACHIP_DISCONNECTED) return st;
    const std::string keep = g_err;
    if (v_out) {
        HIP_TRY(hipMemcpyAsync(v_out, p->sol.yvec, sizeof(double) * (size_t)p->n, hipMemcpyDeviceToHost, p->stream));
        HIP_TRY(hipStreamSynchronize(p->stream));
    }
    if (X_out) ST_TRY(p->sol.ritz_block(q, X_out));
    g_err = keep;
    return st;
}

int machip_set_start(machip_problem* p, const double* x0) {
    if (!p || !x0) return fail(MACHIP_BAD_ARG, "NULL argument");
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(hipMemcpyAsync(p->sol.start, x0, sizeof(double) * (size_t)p->n, hipMemcpyHostToDevice, p->stream));
    HIP_TRY(hipStreamSynchronize(p->stream));
    p->sol.have_start = true;
    ++p->start_version;
    return MACHIP_OK;
}

namespace {
// Collective entry points of an in-process communicator: whatever makes a rank leave early (assembly, eigen-solve,
// a HIP error) must release the peers waiting in the group's barrier instead of leaving them blocked.
struct GroupGuard {
    machip_problem* p;
    bool ok = false;
    ~GroupGuard() {
        if (!ok && p && p->lgroup) p->lgroup->abort();
        if (!ok && p && p->ipcg) { k_ipc_abort<<<1, 64, 0, p->stream>>>(p->ipcg->view); (void)hipStreamSynchronize(p->stream); }
    }
};
}  // namespace

int machip_gradient(machip_problem* p, double* g_out) {
    if (!p || p->csr_only) return fail(MACHIP_BAD_ARG, "bad handle");
    GroupGuard guard{p};
    HIP_TRY(hipSetDevice(p->device));
    if (!p->have_vec) { guard.ok = true; return fail(MACHIP_BAD_ARG, "no Fiedler vector on the device: call machip_fiedler first"); }   // (argument error before any barrier: the group stays usable)
    ST_TRY(compute_gradient(p));
    if (g_out) HIP_TRY(hipMemcpyAsync(g_out, p->g, sizeof(double) * (size_t)p->m, hipMemcpyDeviceToHost, p->stream));
    HIP_TRY(hipStreamSynchronize(p->stream));
    guard.ok = true;
    return MACHIP_OK;
}

int machip_lp_topk(machip_problem* p, int64_t k, double* s_out) {
    if (!p || p->csr_only) return fail(MACHIP_BAD_ARG, "bad handle");
    HIP_TRY(hipSetDevice(p->device));
    ST_TRY(select_topk(p, (long)k));
    const int grid = (int)std::min<long>(kMaxGrid, (p->m + kBlock - 1) / kBlock);
    k_fw_final<<<std::max(grid, 1), kBlock, 0, p->stream>>>(p->g, nullptr, p->m, p->sel, 0.0, nullptr, p->s, p->part_fw);
    if (s_out) HIP_TRY(hipMemcpyAsync(s_out, p->s, sizeof(double) * (size_t)p->m, hipMemcpyDeviceToHost, p->stream));
    HIP_TRY(hipStreamSynchronize(p->stream));
    return MACHIP_OK;
}

int machip_fw_step(machip_problem* p, int64_t k, int iter, double tol, int max_steps, int warm_start,
                   double* f, double* dual, double* gnorm, machip_solve_stats* stats) {
    if (!p || p->csr_only || !f || !dual || !gnorm) return fail(MACHIP_BAD_ARG, "bad argument");
    GroupGuard guard{p};
    HIP_TRY(hipSetDevice(p->device));
    ST_TRY(assemble(p));
    double lam = 0.0;
    const int grid = std::max(1, (int)std::min<long>(kMaxGrid, (p->m + kBlock - 1) / kBlock));
    const double gamma = 2.0 / ((double)iter + 2.0);   // frankwolfe.py:7-8
    // gradient -> top-K -> bookkeeping, on the vector in yvec
    int epi_status = MACHIP_OK;
    auto epilogue = [&](bool speculative) {
        bool fused = false;
        int st = compute_gradient(p, speculative, (long)k, &fused);
        if (st == MACHIP_OK) st = select_topk(p, (long)k, fused);
        if (st == MACHIP_OK) {
            k_fw_final<<<grid, kBlock, 0, p->stream>>>(p->g, p->x, p->m, p->sel, gamma, p->x_next, nullptr, p->d_hdbl, p->tol_sel, p->xb_next);   // partials: host only
            p->xb_next_valid = true;
            if (hipGetLastError() != hipSuccess) st = fail(MACHIP_HIP_ERROR, "k_fw_final launch failed");
        }
        epi_status = st;
    };
    bool done = false;
    // A soft status (step cap reached, disconnected selection) reaches every rank of a communicator alike -- replicated solves
    // are deterministic, the row-partitioned one hands the leader's status round -- and nobody is left in a barrier: the group
    // stays usable (a retry with a larger max_steps, the next iteration).  Only hard errors shut it down.
    auto soft = [](int st) { return st == MACHIP_NOT_CONVERGED || st == MACHIP_DISCONNECTED; };
    if (p->lgroup && p->lgroup->shard_eig) {
        const int st = group_fiedler(p, tol, max_steps, warm_start, &lam, stats);
        if (st != MACHIP_OK) { guard.ok = soft(st); return st; }
    } else {
        // single rank: the epilogue rides behind the solve's explicit check (solver.h, after_check) -- no exchange step in it
        const bool spec = p->nranks <= 1 && p->sol.opt.get(kOpt_spec_epilogue, 1) != 0;
        if (spec) p->sol.after_check = [&] { epilogue(true); };
        const int st = run_fiedler(p, tol, max_steps, nullptr, warm_start, &lam, stats);
        p->sol.after_check = nullptr;
        if (st != MACHIP_OK) { guard.ok = soft(st); return st; }
        done = spec && p->sol.hook_seq == p->sol.final_check_seq && epi_status == MACHIP_OK;     // (the passing check's wait covered it)
    }
    if (!done) {
        epilogue(false);
        if (epi_status != MACHIP_OK) return epi_status;
        HIP_TRY(hipStreamSynchronize(p->stream));
    }
    if (p->ipcg) ST_TRY(p->sol.ipc_check_err("gradient exchange"));
    double d = 0.0, q2 = 0.0;
    for (int b = 0; b < grid; ++b) { d += p->h_dbl[b]; q2 += p->h_dbl[kMaxGrid + b]; }
    *f = lam;
    *dual = lam + d;
    *gnorm = std::sqrt(q2);
    guard.ok = true;
    return MACHIP_OK;
}

int machip_fw_run(machip_problem* p, int64_t k, int first_iter, int max_iters, double gap_tol, double grad_tol, double tol, int max_steps,
                  int warm_start, double* upper_inout, double* f_traj, double* dual_traj, double* gnorm_traj, machip_solve_stats* stats,
                  int* modes, int* iters_done) {
    if (!p || p->csr_only || max_iters < 0 || first_iter < 0 || !upper_inout || !iters_done) return fail(MACHIP_BAD_ARG, "machip_fw_run: bad argument");
    double u = *upper_inout;
    int done = 0;
    *iters_done = 0;
    for (int i = 0; i < max_iters; ++i) {
        double f = 0.0, dual = 0.0, gn = 0.0;
        machip_solve_stats st;
        memset(&st, 0, sizeof(st));
        const int rc = machip_fw_step(p, k, first_iter + i, tol, max_steps, (warm_start && (first_iter + i) > 0) ? 1 : 0, &f, &dual, &gn, &st);
        if (rc != MACHIP_OK) { *upper_inout = u; *iters_done = done; return rc; }
        u = std::min(u, dual);                                      // frankwolfe.py:62
        if (f_traj) f_traj[i] = f;
        if (dual_traj) dual_traj[i] = dual;
        if (gnorm_traj) gnorm_traj[i] = gn;
        if (stats) stats[i] = st;
        if (modes) { modes[2 * i] = p->sol.last_mode; modes[2 * i + 1] = (int)p->sol.last_wb_s; }
        done = i + 1;
        if (gn < grad_tol) break;                                   // frankwolfe.py:65-68 (x stays the current iterate)
        if ((u - f) < gap_tol * std::fabs(f)) break;                // frankwolfe.py:70-74
        ST_TRY(machip_fw_commit(p));                                // frankwolfe.py:76
    }
    *upper_inout = u; *iters_done = done;
    return MACHIP_OK;
}

int machip_round_nearest(machip_problem* p, int64_t k, int decimals, double* rounded_out) {
    if (!p || p->csr_only || !rounded_out) return fail(MACHIP_BAD_ARG, "bad argument");
    if (decimals > 15) return fail(MACHIP_BAD_ARG, "decimals must be <= 15");
    HIP_TRY(hipSetDevice(p->device));
    const long m = p->m;
    if (m == 0) return MACHIP_OK;
    if (!p->scratch_m) ST_TRY(dev_alloc(&p->scratch_m, (size_t)m + 64));
    const int grid = std::max(1, (int)std::min<long>(kMaxGrid, (m + kBlock - 1) / kBlock));
    double* r = p->s;                       // rounded selection weights (or x itself)
    const bool tb = decimals >= 0;
    if (tb) k_round_keys<<<grid, kBlock, 0, p->stream>>>(p->x, m, std::pow(10.0, decimals), r);
    else HIP_TRY(hipMemcpyAsync(r, p->x, sizeof(double) * (size_t)m, hipMemcpyDeviceToDevice, p->stream));
    ST_TRY(select_on(p, r, (long)k, p->sel, 1));
    int two = 0;
    if (tb && k > 0 && k < m) {
        SelState h;
        HIP_TRY(hipMemcpyAsync(&h, p->sel, sizeof(SelState), hipMemcpyDeviceToHost, p->stream));
        HIP_TRY(hipStreamSynchronize(p->stream));
        if (h.kk < h.cnt_eq) {              // more candidates tie at the k-th rounded value than fit
            k_tie_keys<<<grid, kBlock, 0, p->stream>>>(r, p->cw, m, p->sel, p->scratch_m);
            ST_TRY(select_on(p, p->scratch_m, (long)h.kk, p->sel + 1, 1));
            two = 1;
        }
    }
    k_round_mark<<<grid, kBlock, 0, p->stream>>>(r, p->scratch_m, m, p->sel, p->sel + 1, two, r);
    HIP_TRY(hipMemcpyAsync(rounded_out, r, sizeof(double) * (size_t)m, hipMemcpyDeviceToHost, p->stream));
    HIP_TRY(hipStreamSynchronize(p->stream));
    return MACHIP_OK;
}

int machip_fw_commit(machip_problem* p) {
    if (!p || p->csr_only) return fail(MACHIP_BAD_ARG, "bad handle");
    std::swap(p->x, p->x_next);
    std::swap(p->xb, p->xb_next);
    p->xb_valid = p->xb_next_valid; p->xb_next_valid = false;
    p->assembled = false;
    return MACHIP_OK;
}

int machip_fiedler_csr(int device, int64_t n, const int32_t* indptr, const int32_t* indices, const double* data,
                       double tol, int max_steps, const double* x0, double* lambda2, double* v_out, double* X_out,
                       int q, machip_solve_stats* stats) {
    if (!indptr || !indices || !data || !lambda2) return fail(MACHIP_BAD_ARG, "NULL argument");
    if (n < 2 || n > 2000000000ll) return fail(MACHIP_BAD_ARG, "n must be in [2, 2e9]");
    if (X_out && (q < 1 || q > 4)) return fail(MACHIP_BAD_ARG, "q must be in [1,4]");
    // The row pointers are validated as a whole BEFORE anything is read through them (this entry point takes arbitrary
    // caller matrices): first entry 0, monotone, hence every row's range inside [0, indptr[n]) -- and all of that on the
    // host, before the first device call, so a malformed matrix is a BAD_ARG on any machine.
    const long nnz = indptr[n];
    if (indptr[0] != 0 || nnz < 0) return fail(MACHIP_BAD_ARG, "bad indptr: indptr[0] must be 0 and indptr[n] >= 0 (int32 row pointers: nnz < 2^31)");
    for (int64_t r = 0; r < n; ++r)
        if (indptr[r + 1] < indptr[r]) return fail(MACHIP_BAD_ARG, "indptr not monotone");
    double lnorm = 0.0;
    int maxlen_csr = 0;
    long chain_cnt = 0;   // super-diagonal entries: a pose-graph Laplacian carries the chain (i, i+1)
    for (int64_t r = 0; r < n; ++r) {
        double s = 0.0;
        for (long pp = indptr[r]; pp < (long)indptr[r + 1]; ++pp) {
            if (indices[pp] < 0 || indices[pp] >= n) return fail(MACHIP_BAD_ARG, "column index out of range");
            s += std::fabs(data[pp]);
            if (indices[pp] == r + 1 && data[pp] != 0.0) ++chain_cnt;
        }
        lnorm = std::max(lnorm, s);   // nx:232
        maxlen_csr = std::max(maxlen_csr, (int)(indptr[r + 1] - indptr[r]));
    }
    if (machip_device_count() <= 0) return fail(MACHIP_NO_DEVICE, "no HIP device visible");
    HIP_TRY(hipSetDevice(device));
    // One CSR-only handle is kept between calls (stream, ~25 device buffers, pinned mirrors, chunk graphs: creating them
    // costs more than a small solve -- a Madow loop or a sweep through find_fiedler_pair paid it per call).  It is reused
    // when device and n match and the matrix fits; every solve starts from a clean solver state, so a cached handle
    // computes exactly what a fresh one does.  A second thread calling concurrently gets a fresh handle of its own.
    std::unique_lock<std::mutex> lk(g_csr_mu, std::try_to_lock);
    machip_problem* p = nullptr;
    bool cached = false;
    if (lk.owns_lock() && g_csr_cache && g_csr_cache->device == device && g_csr_cache->n == (int)n && (long)g_csr_cache->sol.csr_cap >= nnz) {
        p = g_csr_cache;
        cached = true;
    } else {
        if (lk.owns_lock() && g_csr_cache) { machip_destroy(g_csr_cache); g_csr_cache = nullptr; }
        p = new machip_problem();
        p->device = device; p->n = (int)n; p->csr_only = true;
    }
    auto body = [&]() -> int {
        if (!cached) {
            const size_t cap = (size_t)nnz + (size_t)nnz / 4 + 64;
            HIP_TRY(hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking));
            ST_TRY(dev_alloc(&p->rowptr, (size_t)n + 1)); ST_TRY(dev_alloc(&p->col, cap)); ST_TRY(dev_alloc(&p->val, cap));
            ST_TRY(alloc_common(p));
            p->sol.csr_cap = cap;
        }
        HIP_TRY(hipMemcpyAsync(p->rowptr, indptr, sizeof(int) * ((size_t)n + 1), hipMemcpyHostToDevice, p->stream));
        if (nnz) {
            HIP_TRY(hipMemcpyAsync(p->col, indices, sizeof(int) * (size_t)nnz, hipMemcpyHostToDevice, p->stream));
            HIP_TRY(hipMemcpyAsync(p->val, data, sizeof(double) * (size_t)nnz, hipMemcpyHostToDevice, p->stream));
        }
        HIP_TRY(hipStreamSynchronize(p->stream));
        p->nnz = nnz; p->lnorm = lnorm; p->maxlen = maxlen_csr; p->assembled = true; p->have_vec = false;
        // clean solver state: nothing learnt from, or left over by, an earlier matrix
        Solver& S = p->sol;
        S.have_prev = false; S.warm_skip = 0; S.warm_fails = 0; S.last_pr = 0.0; S.have_start = false; S.hist_lan_steps = -1; S.hist_lob_iters = -1; S.hist_exact_iters = -1; S.last_steps = 0;
        S.last_steps_lowp = 0; S.J_last = 0; S.last_was_lob = false; S.last_seq_f32 = false; S.solver_mode = 0; S.precision = 0;
        S.opt = default_options();       // (a cached handle too starts from the process defaults of THIS call: include/machip.h says so -- advisor finding on round 5)
        // same solver selection as a MAC handle gets (solver.h): chain-like matrices may run the
        // single-workgroup / preconditioned modes; "support" = off-chain edges
        p->sol.chain_like = chain_cnt >= (long)(0.98 * (double)(n - 1));
        p->sol.chain_edges = chain_cnt;
        p->support = std::max<long>(0, (nnz - n) / 2 - chain_cnt);
        return MACHIP_OK;
    };
    int st = body();
    if (st == MACHIP_OK) st = machip_fiedler(p, tol, max_steps, x0, 0, lambda2, v_out, X_out, q, stats);
    const std::string keep = g_err;
    const bool hard = st != MACHIP_OK && st != MACHIP_NOT_CONVERGED && st != MACHIP_DISCONNECTED;
    if (lk.owns_lock() && !hard) g_csr_cache = p;          // keep it for the next call
    else {
        if (cached) g_csr_cache = nullptr;
        machip_destroy(p);
    }
    g_err = keep;
    return st;
}

void machip_release_cache(void) {
    std::lock_guard<std::mutex> lk(g_csr_mu);
    if (g_csr_cache) { machip_destroy(g_csr_cache); g_csr_cache = nullptr; }
}

int machip_spmv(machip_problem* p, const double* v, double* y, int variant) {
    if (!p || !v || !y) return fail(MACHIP_BAD_ARG, "NULL argument");
    HIP_TRY(hipSetDevice(p->device));
    if (!p->assembled) ST_TRY(assemble(p));
    HIP_TRY(hipMemcpyAsync(p->sol.y_raw, v, sizeof(double) * (size_t)p->n, hipMemcpyHostToDevice, p->stream));
    const SpmvPlan pl = plan_spmv(p->sol.opt, p->n, p->nnz, variant);
    OpPlain op{p->sol.w2};
    launch_spmv(pl, p->stream, p->csr(), p->sol.y_raw, op);
    HIP_TRY(hipMemcpyAsync(y, p->sol.w2, sizeof(double) * (size_t)p->n, hipMemcpyDeviceToHost, p->stream));
    HIP_TRY(hipStreamSynchronize(p->stream));
    return MACHIP_OK;
}

int machip_landscape(machip_problem* p, int sweeps, double* u_out) {
    if (!p || !u_out || sweeps < 0 || sweeps > 16) return fail(MACHIP_BAD_ARG, "bad argument (0 <= sweeps <= 16)");
    HIP_TRY(hipSetDevice(p->device));
    if (!p->assembled) ST_TRY(assemble(p));
    const SpmvPlan pl = p->sol.landscape_plan(p->nnz);
    const double* f = p->sol.landscape_field(p->csr(), pl, sweeps);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(u_out, f, sizeof(double) * (size_t)p->n, hipMemcpyDeviceToHost, p->stream));
    HIP_TRY(hipStreamSynchronize(p->stream));
    return MACHIP_OK;
}

int machip_profile_spmv(machip_problem* p, int reps, double* avg_us, double* bytes_per_launch) {
    if (!p || !avg_us || reps < 1) return fail(MACHIP_BAD_ARG, "bad argument");
    HIP_TRY(hipSetDevice(p->device));
    if (!p->assembled) ST_TRY(assemble(p));
    Solver& S = p->sol;
    const SpmvPlan pl = plan_pipe(p->sol.opt, p->n, p->nnz, p->maxlen);
    // The dominant kernel of the path: one fused Lanczos step (k_pipe_*).  Re-launching step 1 of
    // a scratch sequence is idempotent (same Z read, same Z/V column written), so `reps`
    // back-to-back launches time exactly the kernel the solve runs, on the same L(x).
    k_fill_start<<<S.vgrid(), kBlock, 0, p->stream>>>(S.u, p->n, 77ull);
    const PipeView L = S.pview(pl);
    k_pipe_init<<<pl.grid, kBlock, 0, p->stream>>>(L, S.u, (int)(++S.epoch));
    launch_pipe(pl, p->stream, p->csr(), L, 0);
    launch_pipe(pl, p->stream, p->csr(), L, 1);
    k_pipe_tail<<<1, 64, 0, p->stream>>>(L, 2);      // jA = 2
    for (int i = 0; i < 3; ++i) launch_pipe(pl, p->stream, p->csr(), L, 0);
    HIP_TRY(hipEventRecord(S.ev0, p->stream));
    for (int i = 0; i < reps; ++i) launch_pipe(pl, p->stream, p->csr(), L, 0);
    HIP_TRY(hipEventRecord(S.ev1, p->stream));
    HIP_TRY(hipEventSynchronize(S.ev1));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, S.ev0, S.ev1));
    *avg_us = 1e3 * (double)ms / reps;
    if (bytes_per_launch) {
        // SURVEY 8(d) B_spmv = nnz*(8+4) + (n+1)*4 + n*8 [x] + n*8 [y], with the fused vector
        // work on top: the gathered operand is the 16-byte record Z[c] (counted once per row),
        // own-row Z read 16 B, next Z written 16 B, Lanczos vector v_j written 8 B.
        *bytes_per_launch = (double)p->nnz * 12.0 + ((double)p->n + 1.0) * 4.0 + (double)p->n * 56.0;
    }
    return MACHIP_OK;
}

int machip_comm_unique_id(void* id128) {
    if (!id128) return fail(MACHIP_BAD_ARG, "NULL id buffer");
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is expected to be 128 bytes");
    ncclUniqueId id;
    NCCL_TRY(ncclGetUniqueId(&id));
    memcpy(id128, &id, sizeof(id));
    return MACHIP_OK;
}

int machip_comm_init(machip_problem* p, int rank, int nranks, const void* id128) {
    if (!p || p->csr_only || !id128 || nranks < 1 || rank < 0 || rank >= nranks) return fail(MACHIP_BAD_ARG, "bad argument");
    if (nranks > 64) return fail(MACHIP_BAD_ARG, "at most 64 ranks");
    if (p->comm || p->lgroup) return fail(MACHIP_BAD_ARG, "machip_comm_init: handle already belongs to a communicator");
    HIP_TRY(hipSetDevice(p->device));
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    {
        const int dev = p->device;
        auto comm_out = std::make_shared<ncclComm_t>(nullptr);
        auto err_out = std::make_shared<std::string>();
        const int st = run_with_watchdog([=]() -> int {
            if (hipSetDevice(dev) != hipSuccess) { *err_out = "hipSetDevice failed"; return MACHIP_HIP_ERROR; }
            const ncclResult_t r = ncclCommInitRank(comm_out.get(), nranks, id, rank);
            if (r != ncclSuccess) { *err_out = std::string("ncclCommInitRank: ") + ncclGetErrorString(r); return MACHIP_RCCL_ERROR; }
            return MACHIP_OK;
        }, rccl_timeout_s(), "ncclCommInitRank (rank " + std::to_string(rank) + " of " + std::to_string(nranks) + ")");
        if (st != MACHIP_OK) return err_out->empty() ? st : fail((machip_status)st, *err_out);
        p->comm = *comm_out;
    }
    p->rank = rank; p->nranks = nranks;
    p->first_collective_pending = true;
    long lo, hi, shard;
    shard_plan(p->m, nranks, rank, &lo, &hi, &shard);
    p->m_pad = shard * nranks;   // <= m + 63 < allocation slack
    return MACHIP_OK;
}

// ---- communicator between processes with a row-partitioned eigen-solve (round 4) ----
// Blob a rank hands to its peers: IPC handles of its record buffers, partial sums, Ritz staging vector, gradient and flag words.
namespace {
struct IpcBlob {
    unsigned long long magic;
    int n; long m; int pid; int device;
    hipIpcMemHandle_t Z0, Z1, part, yraw, g, flags;
};
constexpr unsigned long long kIpcMagic = 0x6d61636869706331ull;
constexpr size_t kIpcFlagWords = (size_t)kIpcChannels * kMaxPeers + kIpcChannels + 8;
}  // namespace

int machip_ipc_blob_bytes(void) { return (int)sizeof(IpcBlob); }

int machip_ipc_export(machip_problem* p, void* blob, int blob_bytes) {
    if (!p || p->csr_only || !blob || blob_bytes < (int)sizeof(IpcBlob)) return fail(MACHIP_BAD_ARG, "machip_ipc_export: bad argument");
    if (p->lgroup || p->ipcg) return fail(MACHIP_BAD_ARG, "machip_ipc_export: handle already belongs to a communicator");
    HIP_TRY(hipSetDevice(p->device));
    auto G = std::make_unique<IpcGroup>();
    ST_TRY(dev_alloc(&G->flags_mem, kIpcFlagWords));
    HIP_TRY(hipMemsetAsync(G->flags_mem, 0, sizeof(unsigned long long) * kIpcFlagWords, p->stream));
    HIP_TRY(hipStreamSynchronize(p->stream));
    HIP_TRY(hipHostMalloc((void**)&G->h_err, 64, hipHostMallocMapped));
    *G->h_err = 0;
    IpcBlob b;
    memset(&b, 0, sizeof(b));
    b.magic = kIpcMagic; b.n = p->n; b.m = p->m; b.pid = (int)getpid(); b.device = p->device;
    HIP_TRY(hipIpcGetMemHandle(&b.Z0, p->sol.Z0)); HIP_TRY(hipIpcGetMemHandle(&b.Z1, p->sol.Z1));
    HIP_TRY(hipIpcGetMemHandle(&b.part, p->sol.part)); HIP_TRY(hipIpcGetMemHandle(&b.yraw, p->sol.y_raw));
    HIP_TRY(hipIpcGetMemHandle(&b.g, p->g)); HIP_TRY(hipIpcGetMemHandle(&b.flags, G->flags_mem));
    memcpy(blob, &b, sizeof(b));
    p->ipcg = std::move(G);
    return MACHIP_OK;
}

int machip_comm_init_ipc(machip_problem* p, int rank, int nranks, const void* blobs, double timeout_s) {
    if (!p || p->csr_only || !blobs || nranks < 2 || nranks > kMaxPeers || rank < 0 || rank >= nranks)
        return fail(MACHIP_BAD_ARG, "machip_comm_init_ipc: 2..8 ranks, a blob per rank");
    if (!p->ipcg || p->ipcg->nranks) return fail(MACHIP_BAD_ARG, "machip_comm_init_ipc: call machip_ipc_export first (once)");
    if (p->lgroup) return fail(MACHIP_BAD_ARG, "machip_comm_init_ipc: handle already belongs to an in-process communicator");
    HIP_TRY(hipSetDevice(p->device));
    IpcGroup& G = *p->ipcg;
    const IpcBlob* B = static_cast<const IpcBlob*>(blobs);
    for (int q = 0; q < nranks; ++q)
        if (B[q].magic != kIpcMagic || B[q].n != p->n || B[q].m != p->m) return fail(MACHIP_BAD_ARG, "machip_comm_init_ipc: blob of a different problem / library");
    if (B[rank].pid != (int)getpid()) return fail(MACHIP_BAD_ARG, "machip_comm_init_ipc: blob[rank] is not this process's");
    auto open = [&](const hipIpcMemHandle_t& h, void** out, int peer_device) -> int {
        if (peer_device != p->device) {     // distinct devices: the peer's memory is reached over xGMI
            int can = 0;
            HIP_TRY(hipDeviceCanAccessPeer(&can, p->device, peer_device));
            if (!can) return fail(MACHIP_RCCL_ERROR, "machip_comm_init_ipc: no peer access between the ranks' devices");
            const hipError_t e = hipDeviceEnablePeerAccess(peer_device, 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) return fail(MACHIP_HIP_ERROR, std::string("hipDeviceEnablePeerAccess: ") + hipGetErrorString(e));
            (void)hipGetLastError();
        }
        const hipError_t e = hipIpcOpenMemHandle(out, h, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) { (void)hipGetLastError(); return fail(MACHIP_RCCL_ERROR, std::string("hipIpcOpenMemHandle: ") + hipGetErrorString(e) + " (HSA_ENABLE_IPC_MODE_LEGACY=0 set?)"); }
        G.opened.push_back(*out);
        return MACHIP_OK;
    };
    // (a failure below leaves a half-mapped group behind: it is dropped -- mappings closed by ~IpcGroup -- and a retry starts from
    // machip_ipc_export again)
    struct Rollback { machip_problem* p; bool ok = false; ~Rollback() { if (!ok) p->ipcg.reset(); } } rollback{p};
    for (int q = 0; q < nranks; ++q) {
        if (q == rank) {
            G.Z0[q] = p->sol.Z0; G.Z1[q] = p->sol.Z1; G.part[q] = p->sol.part; G.yraw[q] = p->sol.y_raw; G.g[q] = p->g;
            G.view.peer_flags[q] = G.flags_mem;
            continue;
        }
        void* t = nullptr;
        ST_TRY(open(B[q].Z0, &G.Z0[q], B[q].device)); ST_TRY(open(B[q].Z1, &G.Z1[q], B[q].device));
        ST_TRY(open(B[q].part, &t, B[q].device)); G.part[q] = (double*)t;
        ST_TRY(open(B[q].yraw, &t, B[q].device)); G.yraw[q] = (double*)t;
        ST_TRY(open(B[q].g, &t, B[q].device)); G.g[q] = (double*)t;
        ST_TRY(open(B[q].flags, &t, B[q].device)); G.view.peer_flags[q] = (unsigned long long*)t;
    }
    G.nranks = nranks; G.rank = rank;
    G.view.n = nranks; G.view.rank = rank;
    G.view.flags = G.flags_mem;
    G.view.done = G.flags_mem + (size_t)kIpcChannels * kMaxPeers;
    HIP_TRY(hipHostGetDevicePointer((void**)&G.view.err, G.h_err, 0));
    if (!(timeout_s > 0.0)) timeout_s = 10.0;
    G.view.timeout_ticks = (long long)(timeout_s * 1e8);
    p->rank = rank; p->nranks = nranks;
    long lo, hi, shard;
    shard_plan(p->m, nranks, rank, &lo, &hi, &shard);
    p->m_pad = shard * nranks;
    if (p->sol.opt.get(kOpt_shard_eig, 1) != 0) p->sol.ipc = &G;      // (0: the eigen-solve stays replicated, only the gradient is exchanged)
    rollback.ok = true;
    return MACHIP_OK;
}

int machip_comm_close_ipc(machip_problem* p) {
    if (!p || !p->ipcg) return fail(MACHIP_BAD_ARG, "machip_comm_close_ipc: no inter-process communicator");
    p->ipcg->clean_exit = true;
    return MACHIP_OK;
}

int machip_comm_init_local(machip_problem** handles, int nranks) {
    if (!handles || nranks < 1 || nranks > 64) return fail(MACHIP_BAD_ARG, "machip_comm_init_local: 1..64 handles");
    for (int r = 0; r < nranks; ++r) {
        machip_problem* p = handles[r];
        if (!p || p->csr_only || p->comm || p->lgroup) return fail(MACHIP_BAD_ARG, "machip_comm_init_local: NULL, CSR-only or already attached handle");
        if (p->m != handles[0]->m || p->n != handles[0]->n) return fail(MACHIP_BAD_ARG, "machip_comm_init_local: handles of different problems");
    }
    auto G = std::make_shared<LocalGroup>();
    G->nranks = nranks;
    for (int r = 0; r < nranks; ++r) G->g.push_back(handles[r]->g);
    // row-partitioned eigen-solve (MACHIP_SHARD_EIG=0: every rank runs the whole solve itself, as in round 2)
    G->shard_eig = nranks > 1 && nranks <= kMaxPeers && handles[0]->sol.opt.get(kOpt_shard_eig, 1) != 0;
    if (G->shard_eig) {
        G->sg.rk.resize((size_t)nranks);
        for (int r = 0; r < nranks; ++r) {
            G->members.push_back(handles[r]);
            HIP_TRY(hipSetDevice(handles[r]->device));
            for (hipEvent_t& e : G->sg.rk[(size_t)r].ev) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            if (handles[r]->device != handles[0]->device) G->sg.same_device = false;
            for (int q = 0; q < nranks; ++q)          // peers write each other's operand copies
                if (handles[q]->device != handles[r]->device) {
                    const hipError_t e = hipDeviceEnablePeerAccess(handles[q]->device, 0);
                    if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) return fail(MACHIP_HIP_ERROR, std::string("hipDeviceEnablePeerAccess: ") + hipGetErrorString(e));
                    (void)hipGetLastError();
                }
        }
        HIP_TRY(hipSetDevice(handles[0]->device));
        HIP_TRY(hipEventCreateWithFlags(&G->sg.fork, hipEventDisableTiming));
    }
    for (int r = 0; r < nranks; ++r) {
        machip_problem* p = handles[r];
        long lo, hi, shard;
        shard_plan(p->m, nranks, r, &lo, &hi, &shard);
        p->m_pad = shard * nranks;
        p->rank = r; p->nranks = nranks; p->lgroup = G;
    }
    return MACHIP_OK;
}

int machip_selftest_watchdog(int work_ms, int limit_ms) {
    // (test hook, no GPU needed: the watchdog around the RCCL first-contact calls, exercised with a sleeping stand-in)
    return run_with_watchdog([=]() -> int { std::this_thread::sleep_for(std::chrono::milliseconds(work_ms)); return MACHIP_OK; },
                             1e-3 * (double)limit_ms, "watchdog self-test (" + std::to_string(work_ms) + " ms of work)");
}

int machip_peer_access(int device_a, int device_b) {
    int can = 0;
    if (device_a == device_b) return 1;
    if (hipDeviceCanAccessPeer(&can, device_a, device_b) != hipSuccess) { (void)hipGetLastError(); return -1; }
    return can;
}

int machip_comm_mode(machip_problem* p) {
    if (!p) return -1;
    if (p->ipcg && p->ipcg->nranks) return p->sol.ipc ? (p->sol.last_seq_sharded ? 5 : 4) : (p->comm ? 1 : 6);
    if (p->comm) return 1;
    if (p->lgroup) return p->lgroup->shard_eig ? 3 : 2;
    return 0;
}

int machip_panel_plan(int64_t n, int64_t nnz, int maxlen, int* out12) {
    if (n < 1 || n > 2000000000ll || nnz < 0 || !out12) return fail(MACHIP_BAD_ARG, "machip_panel_plan: bad argument");
    const Options& opt = default_options();
    const PanPlan pp = plan_panel(opt, (int)n, (long)nnz, maxlen, true, -1, false, OPT(stream, 1) != 0);
    out12[0] = pp.on ? 1 : 0; out12[1] = pp.NP; out12[2] = pp.C; out12[3] = pp.NB; out12[4] = pp.NTB; out12[5] = pp.TWW; out12[6] = pp.RPT; out12[7] = pp.grid2;
    out12[8] = pp.u ? 1 : 0; out12[9] = pp.LPT; out12[10] = pp.TWT; out12[11] = pp.cells;
    return MACHIP_OK;
}

#ifdef PAN_CLOCKS
// developer build (MACHIP_BUILD_FLAGS=-DPAN_CLOCKS, tools/pan_clocks.py): the 16 wall-clock stamps (100 MHz) every workgroup of the
// LAST column-panel matrix launch left
extern "C" int machip_debug_pan_clocks(machip_problem* p, long long* out, int groups) {
    if (!p || !out || groups < 1 || groups > kMaxGrid || !p->sol.panv.clk) return fail(MACHIP_BAD_ARG, "machip_debug_pan_clocks: no clock buffer");
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(hipStreamSynchronize(p->stream));
    HIP_TRY(hipMemcpy(out, p->sol.panv.clk, sizeof(long long) * 16 * (size_t)groups, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemset(p->sol.panv.clk, 0, sizeof(long long) * 16 * (size_t)kMaxGrid));      // (the next read sees one launch shape only)
    return MACHIP_OK;
}
#endif

int machip_shard_plan(int64_t m, int nranks, int rank, int64_t* lo, int64_t* hi, int64_t* shard) {
    if (m < 0 || nranks < 1 || rank < 0 || rank >= nranks || !lo || !hi || !shard) return fail(MACHIP_BAD_ARG, "machip_shard_plan: bad argument");
    long a, b, c;
    shard_plan((long)m, nranks, rank, &a, &b, &c);
    *lo = a; *hi = b; *shard = c;
    return MACHIP_OK;
}

namespace {
int make_lane(machip_problem* p, machip_problem** out) {
    machip_problem* q = new machip_problem();
    q->is_lane = true;
    q->device = p->device; q->n = p->n; q->m = p->m; q->m_pad = p->m; q->tol_sel = p->tol_sel;
    q->prow = p->prow; q->pcol = p->pcol; q->pk = p->pk; q->pw = p->pw; q->P = p->P;
    q->asm_G = p->asm_G; q->asm_rpb = p->asm_rpb; q->asm_grid = p->asm_grid;
    q->ci = p->ci; q->cj = p->cj; q->cw = p->cw;
    auto body = [&]() -> int {
        ST_TRY(create_lane_stream(p->sol.opt, q->device, &q->stream));
        ST_TRY(dev_alloc(&q->x, (size_t)q->m + 64));
        const size_t cap = (size_t)q->P + (size_t)q->n + 8;
        ST_TRY(dev_alloc(&q->cnt, (size_t)q->n + 1)); ST_TRY(dev_alloc(&q->blk_sum, 3 * kAsmGrid));
        ST_TRY(dev_alloc(&q->rowptr, (size_t)q->n + 1)); ST_TRY(dev_alloc(&q->col, cap)); ST_TRY(dev_alloc(&q->val, cap));
        ST_TRY(dev_alloc(&q->blk_lnorm, kAsmGrid));
        ST_TRY(dev_alloc(&q->xb, (size_t)q->m / 64 + 2)); ST_TRY(dev_alloc(&q->xb_next, (size_t)q->m / 64 + 2));
        // every lane keeps its own Krylov basis: a share of the handle's budget each (MACHIP_LANE_VBUDGET_MB, default
        // MACHIP_VBUDGET_MB / 8 = 512 MB: 16 lanes together hold twice what the handle itself holds; a sequence longer than
        // the lane's share restarts earlier than it would on the handle -- include/machip.h states the guarantee accordingly)
        const Options& opt = p->sol.opt;
        q->sol.opt = opt;                     // a lane runs on its owner's options
        ST_TRY(alloc_common(q, std::max(16, OPT(lane_vbudget_mb, std::max(16, OPT(vbudget_mb, 4096) / 8)))));
        q->sol.csr_cap = cap;
        q->sol.pat = q->pattern();
        q->sol.pan_allowed = p->sol.pan_allowed;
        q->sol.chain_like = p->sol.chain_like; q->sol.chain_edges = p->sol.chain_edges;
        return MACHIP_OK;
    };
    const int st = body();
    if (st != MACHIP_OK) { machip_destroy(q); return st; }
    *out = q;
    return MACHIP_OK;
}

// FW state of a lane (gradient, LP vertex, next iterate, select scratch): only the sweep needs it
int ensure_lane_fw(machip_problem* q) {
    if (q->lane_fw_ready) return MACHIP_OK;
    const size_t mp = (size_t)q->m + 64 * 8;
    // each pointer guarded on its own: a failed allocation half way leaves nothing to leak when the next call retries
    if (!q->x_next) ST_TRY(dev_alloc(&q->x_next, mp));
    if (!q->g) ST_TRY(dev_alloc(&q->g, mp));
    if (!q->s) ST_TRY(dev_alloc(&q->s, mp));
    if (!q->hist) ST_TRY(dev_alloc(&q->hist, (6 + kSelRep) * kBins));
    if (!q->sel) ST_TRY(dev_alloc(&q->sel, 2));
    if (!q->part_fw) ST_TRY(dev_alloc(&q->part_fw, 2 * kMaxGrid));
    HIP_TRY(hipMemsetAsync(q->g, 0, sizeof(double) * mp, q->stream));     // (never the legacy stream: another lane's thread may be capturing a graph)
    HIP_TRY(hipStreamSynchronize(q->stream));
    q->lane_fw_ready = true;
    return MACHIP_OK;
}

// Lanes [0, *nl_out) of the handle, created on demand, on the handle's solver mode / precision / start vector.
int prepare_lanes(machip_problem* p, int B, bool fw, int* nl_out) {
    // Lanes: 16 where a solve is the single-workgroup kernel (one CU each; every lane's stream has a hardware queue of its own,
    // see create_lane_stream at the top of this file), 4 where the solves launch chip-filling step kernels -- more than
    // four hardware queues of those at once collapse (city10000 sweep: 647 it/s with 4 lanes, 277 with 8, 240 with 12).
    const bool small = p->sol.chain_like && persist_fits(p->n, std::max(0l, (long)p->P - 2 * p->sol.chain_edges));   // (P: off-diagonal slots of the union pattern)
    const Options& opt = p->sol.opt;
    // Round 5 (profiles/r5_bench_c4s.json, r5_bench_c2s.json): problems whose every step fills the chip for 7-17 us (ER sizes: union
    // pattern beyond ~400 000 slots) gain from ONE more problem in flight -- 2 lanes 1.27x (configs[3]) / 1.48x (configs[1]) the
    // one-at-a-time rate, 4 lanes 1.03x / 0.95x: two problems already cover each other's launch gaps, more only thrash the L2s.
    // Later in round 5 (tools/r5_eager3.sh): that was the captured chunks -- a lane's hipGraph executables take hardware queues away from
    // the other lanes.  Lanes launch eagerly now (Solver::use_graph) and 4 lanes win at the ER sizes too: configs[3] 267 (2 lanes) ->
    // 293 it/s (4), configs[1] 943 -> 1 180.  Beyond 4 lanes the queues are shared whatever the launch form (city10000 sweep: 1 099 it/s
    // with 4 lanes, 733 with 8, 474 with 12).
    int nl = std::max(1, std::min(B, std::min(16, OPT(lanes, small ? 16 : 4))));
    while ((int)p->lanes.size() < nl) {
        machip_problem* q = nullptr;
        const int st = make_lane(p, &q);
        if (st != MACHIP_OK) {
            // out of device memory for another lane: carry on with the lanes there are (the batch then takes longer, it
            // does not fail); without any lane the error stands
            if (p->lanes.empty()) return st;
            (void)hipGetLastError();
            nl = (int)p->lanes.size();
            break;
        }
        p->lanes.push_back(q);
    }
    HIP_TRY(hipStreamSynchronize(p->stream));
    for (int l = 0; l < nl; ++l) {
        machip_problem* q = p->lanes[(size_t)l];
        q->sol.solver_mode = p->sol.solver_mode; q->sol.precision = p->sol.precision;
        q->sol.opt = p->sol.opt;
        q->sol.throughput_lane = true;
        if (p->sol.have_start && q->seen_start_version != p->start_version) {     // per lane: a batch may use fewer lanes than exist
            HIP_TRY(hipMemcpyAsync(q->sol.start, p->sol.start, sizeof(double) * (size_t)p->n, hipMemcpyDeviceToDevice, q->stream));
            HIP_TRY(hipStreamSynchronize(q->stream));
            q->sol.have_start = true;
            q->seen_start_version = p->start_version;
        }
    }
    if (fw) for (int l = 0; l < nl; ++l) ST_TRY(ensure_lane_fw(p->lanes[(size_t)l]));
    *nl_out = nl;
    return MACHIP_OK;
}
}  // namespace

int machip_eval_batch(machip_problem* p, int B, const double* X, double tol, int max_steps, double* lambda2, int* status) {
    if (!p || p->csr_only || p->is_lane || B < 0 || (B && (!X || !lambda2))) return fail(MACHIP_BAD_ARG, "machip_eval_batch: bad argument");
    if (B == 0) return MACHIP_OK;
    HIP_TRY(hipSetDevice(p->device));
    int nl = 0;
    ST_TRY(prepare_lanes(p, B, false, &nl));
    std::atomic<int> next{0};
    std::vector<int> lane_status((size_t)nl, MACHIP_OK);
    std::vector<std::string> lane_err((size_t)nl);
    auto work = [&](int l) {
        machip_problem* q = p->lanes[(size_t)l];
        if (hipSetDevice(q->device) != hipSuccess) { lane_status[(size_t)l] = MACHIP_HIP_ERROR; lane_err[(size_t)l] = "hipSetDevice failed"; return; }
        for (int b = next.fetch_add(1); b < B; b = next.fetch_add(1)) {
            int st = MACHIP_OK;
            double lam = 0.0;
            if (hipMemcpyAsync(q->x, X + (size_t)b * (size_t)q->m, sizeof(double) * (size_t)q->m, hipMemcpyHostToDevice, q->stream) != hipSuccess ||
                hipStreamSynchronize(q->stream) != hipSuccess) {
                st = fail(MACHIP_HIP_ERROR, "machip_eval_batch: copy of x failed");
            } else {
                q->assembled = false; q->xb_valid = false;
                st = assemble(q);
                // clean solver state per entry (no step-count history from whatever this lane solved before: the automatic
                // mode choice reads it): an entry's result does not depend on the lane that takes it or on its place in the batch
                q->sol.have_prev = false; q->sol.warm_skip = 0; q->sol.warm_fails = 0; q->sol.last_pr = 0.0; q->sol.hist_lan_steps = -1; q->sol.hist_lob_iters = -1; q->sol.hist_exact_iters = -1; q->sol.last_steps = 0; q->sol.last_steps_lowp = 0;
                if (st == MACHIP_OK) st = run_fiedler(q, tol, max_steps, nullptr, 0, &lam, nullptr);
            }
            lambda2[b] = lam;
            if (status) status[b] = st;
            if (st != MACHIP_OK && st != MACHIP_NOT_CONVERGED && st != MACHIP_DISCONNECTED) {
                lane_status[(size_t)l] = st; lane_err[(size_t)l] = g_err;     // hard failure: stop this lane
                return;
            }
        }
    };
    std::vector<std::thread> th;
    for (int l = 1; l < nl; ++l) th.emplace_back(work, l);
    work(0);
    for (auto& t : th) t.join();
    for (int l = 0; l < nl; ++l)
        if (lane_status[(size_t)l] != MACHIP_OK) return fail((machip_status)lane_status[(size_t)l], lane_err[(size_t)l]);
    return MACHIP_OK;
}

int machip_fw_sweep(machip_problem* p, int B, const int64_t* ks, const double* X0, int max_iters, double gap_tol,
                    double grad_tol, double tol, int max_steps, int warm_start, int round_decimals, double* X_out,
                    double* R_out, double* upper, double* f_traj, int* iters, int* status) {
    if (!p || p->csr_only || p->is_lane || B < 0 || max_iters < 0 || (B && (!ks || !X0 || !X_out || !upper || !iters || !status)))
        return fail(MACHIP_BAD_ARG, "machip_fw_sweep: bad argument");
    if (p->nranks > 1) return fail(MACHIP_BAD_ARG, "machip_fw_sweep: not on a handle that belongs to a communicator");
    if (B == 0) return MACHIP_OK;
    HIP_TRY(hipSetDevice(p->device));
    int nl = 0;
    ST_TRY(prepare_lanes(p, B, true, &nl));
    const size_t m = (size_t)p->m;
    std::atomic<int> next{0};
    std::vector<int> lane_status((size_t)nl, MACHIP_OK);
    std::vector<std::string> lane_err((size_t)nl);
    auto work = [&](int l) {
        machip_problem* q = p->lanes[(size_t)l];
        for (int b = next.fetch_add(1); b < B; b = next.fetch_add(1)) {
            // every problem starts from a clean solver state: its result does not depend on which lane takes it or on
            // what that lane solved before (= a fresh handle running the reference's loop, frankwolfe.py:53-76)
            Solver& S = q->sol;
            S.have_prev = false; S.warm_skip = 0; S.warm_fails = 0; S.last_pr = 0.0; S.hist_lan_steps = -1; S.hist_lob_iters = -1; S.hist_exact_iters = -1; S.last_steps = 0; S.last_steps_lowp = 0;
            int st = machip_set_x(q, X0 + (size_t)b * m);
            double u = std::numeric_limits<double>::infinity();
            int done = 0;
            for (int i = 0; i < max_iters && st == MACHIP_OK; ++i) {
                double f = 0.0, dual = 0.0, gn = 0.0;
                st = machip_fw_step(q, ks[b], i, tol, max_steps, warm_start && i > 0, &f, &dual, &gn, nullptr);
                if (st != MACHIP_OK) break;
                u = std::min(u, dual);
                if (f_traj) f_traj[(size_t)b * (size_t)max_iters + (size_t)i] = f;
                done = i + 1;
                if (gn < grad_tol) break;                            // frankwolfe.py:65-68
                if ((u - f) < gap_tol * std::fabs(f)) break;         // frankwolfe.py:70-74
                st = machip_fw_commit(q);
            }
            if (st == MACHIP_OK) st = machip_get_x(q, X_out + (size_t)b * m);
            if (st == MACHIP_OK && R_out) st = machip_round_nearest(q, ks[b], round_decimals, R_out + (size_t)b * m);
            upper[b] = u; iters[b] = done; status[b] = st;
            if (st != MACHIP_OK && st != MACHIP_NOT_CONVERGED && st != MACHIP_DISCONNECTED) {
                lane_status[(size_t)l] = st; lane_err[(size_t)l] = g_err;     // hard failure: stop this lane
                return;
            }
        }
    };
    std::vector<std::thread> th;
    for (int l = 1; l < nl; ++l) th.emplace_back(work, l);
    work(0);
    for (auto& t : th) t.join();
    for (int l = 0; l < nl; ++l)
        if (lane_status[(size_t)l] != MACHIP_OK) return fail((machip_status)lane_status[(size_t)l], lane_err[(size_t)l]);
    return MACHIP_OK;
}

int machip_set_option(machip_problem* p, const char* name, int64_t value) {
    const int id = option_id(name);
    if (id < 0) return fail(MACHIP_BAD_ARG, std::string("machip_set_option: unknown option '") + (name ? name : "(null)") + "'");
    const long v = value == INT64_MIN ? kOptAuto : (long)value;
    if (!p) { default_options().v[id] = v; return MACHIP_OK; }
    p->sol.opt.v[id] = v;
    for (machip_problem* q : p->lanes) q->sol.opt.v[id] = v;
    return MACHIP_OK;
}

int machip_get_option(machip_problem* p, const char* name, int64_t* value) {
    const int id = option_id(name);
    if (id < 0 || !value) return fail(MACHIP_BAD_ARG, "machip_get_option: unknown option or NULL output");
    const long v = p ? p->sol.opt.v[id] : default_options().v[id];
    *value = v == kOptAuto ? INT64_MIN : (int64_t)v;
    return MACHIP_OK;
}

const char* machip_option_name(int i) { return (i >= 0 && i < kNumOpts) ? option_names()[i] : nullptr; }

int machip_comm_timing(machip_problem* p, double* grad_us, double* exchange_us) {
    if (!p || !grad_us || !exchange_us) return fail(MACHIP_BAD_ARG, "machip_comm_timing: NULL argument");
    *grad_us = 0.0; *exchange_us = 0.0;
    if (!p->ev_g_valid) return fail(MACHIP_BAD_ARG, "machip_comm_timing: no sharded gradient has been computed on this handle");
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(hipEventSynchronize(p->ev_g2));
    float a = 0.f, b = 0.f;
    HIP_TRY(hipEventElapsedTime(&a, p->ev_g0, p->ev_g1));
    HIP_TRY(hipEventElapsedTime(&b, p->ev_g1, p->ev_g2));
    *grad_us = 1e3 * (double)a; *exchange_us = 1e3 * (double)b;
    return MACHIP_OK;
}

int machip_solve_mode(machip_problem* p, int64_t* closures) {
    if (!p) return -1;
    if (closures) *closures = p->sol.last_wb_s;
    return p->sol.last_mode;
}

int machip_comm_drop(machip_problem* p) {
    if (!p) return fail(MACHIP_BAD_ARG, "machip_comm_drop: NULL handle");
    if (p->lgroup) return fail(MACHIP_BAD_ARG, "machip_comm_drop: in-process communicators end with their handles");
    HIP_TRY(hipSetDevice(p->device));
    (void)hipStreamSynchronize(p->stream);
    if (p->ipcg) { p->sol.ipc = nullptr; p->ipcg.reset(); }
    if (p->comm) { (void)ncclCommAbort(p->comm); p->comm = nullptr; }     // (abort, not destroy: the peers may never have arrived)
    p->first_collective_pending = false; p->ev_g_valid = false;
    p->rank = 0; p->nranks = 1; p->m_pad = p->m;
    return MACHIP_OK;
}

int machip_comm_drop_ipc(machip_problem* p) {
    if (!p || !p->ipcg) return fail(MACHIP_BAD_ARG, "machip_comm_drop_ipc: no inter-process communicator");
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(hipStreamSynchronize(p->stream));
    p->sol.ipc = nullptr;
    p->ipcg.reset();
    // (an RCCL communicator attached before the IPC leg stays in charge of the gradient shards: rank, size and padding are ITS values --
    // advisor finding on round 5: the reset below used to turn such a handle into a replicated one with a dangling communicator)
    if (!p->comm) { p->rank = 0; p->nranks = 1; p->m_pad = p->m; }
    return MACHIP_OK;
}

int machip_set_solver(machip_problem* p, int mode) {
    if (!p || mode < 0 || mode > 2) return fail(MACHIP_BAD_ARG, "machip_set_solver: mode must be 0 (auto), 1 (Lanczos) or 2 (preconditioned)");
    p->sol.solver_mode = mode;
    return MACHIP_OK;
}

int machip_set_precision(machip_problem* p, int precision) {
    if (!p || precision < 0 || precision > 1) return fail(MACHIP_BAD_ARG, "machip_set_precision: 0 (f64) or 1 (f32 iterate + f64 refinement)");
    p->sol.precision = precision;
    return MACHIP_OK;
}

int machip_synchronize(machip_problem* p) {
    if (!p) return fail(MACHIP_BAD_ARG, "NULL handle");
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(hipStreamSynchronize(p->stream));
    return MACHIP_OK;
}

int machip_membench(int device, int64_t bytes, int reps, double* read_gbs, double* triad_gbs) {
    if (bytes < (1 << 20) || reps < 1 || !read_gbs || !triad_gbs) return fail(MACHIP_BAD_ARG, "machip_membench: bytes >= 1 MiB, reps >= 1, non-NULL outputs");
    if (machip_device_count() <= 0) return fail(MACHIP_NO_DEVICE, "no HIP device visible");
    HIP_TRY(hipSetDevice(device));
    const long cnt2 = (long)(bytes / 16);
    double2 *a = nullptr, *b = nullptr, *c = nullptr;
    double* out = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipStream_t ms_ = nullptr;
    auto body = [&]() -> int {
        HIP_TRY(hipStreamCreateWithFlags(&ms_, hipStreamNonBlocking));      // (never the legacy stream)
        ST_TRY(dev_alloc(&a, (size_t)cnt2)); ST_TRY(dev_alloc(&b, (size_t)cnt2)); ST_TRY(dev_alloc(&c, (size_t)cnt2)); ST_TRY(dev_alloc(&out, 8));
        HIP_TRY(hipMemsetAsync(a, 0, (size_t)cnt2 * 16, ms_)); HIP_TRY(hipMemsetAsync(b, 0, (size_t)cnt2 * 16, ms_)); HIP_TRY(hipMemsetAsync(c, 0, (size_t)cnt2 * 16, ms_));
        HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
        const int grid = 256 * 16;           // 16 workgroups of 256 threads per CU
        float ms = 0.f;
        k_mb_read<<<grid, kBlock, 0, ms_>>>(a, cnt2, out);
        HIP_TRY(hipEventRecord(e0, ms_));
        for (int r = 0; r < reps; ++r) k_mb_read<<<grid, kBlock, 0, ms_>>>(r & 1 ? b : a, cnt2, out);
        HIP_TRY(hipEventRecord(e1, ms_)); HIP_TRY(hipEventSynchronize(e1));
        HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
        *read_gbs = (double)cnt2 * 16.0 * reps / (ms * 1e-3) / 1e9;
        k_mb_triad<<<grid, kBlock, 0, ms_>>>(a, b, c, 3.0, cnt2);
        HIP_TRY(hipEventRecord(e0, ms_));
        for (int r = 0; r < reps; ++r) k_mb_triad<<<grid, kBlock, 0, ms_>>>(a, b, c, 3.0, cnt2);
        HIP_TRY(hipEventRecord(e1, ms_)); HIP_TRY(hipEventSynchronize(e1));
        HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
        *triad_gbs = (double)cnt2 * 48.0 * reps / (ms * 1e-3) / 1e9;      // two reads + one write per element
        HIP_TRY(hipGetLastError());
        return MACHIP_OK;
    };
    const int st = body();
    const std::string keep = g_err;
    if (ms_) (void)hipStreamSynchronize(ms_);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    for (void* q : {(void*)a, (void*)b, (void*)c, (void*)out}) if (q) (void)hipFree(q);
    if (ms_) (void)hipStreamDestroy(ms_);
    g_err = keep;
    return st;
}

// Host-only helper exported for CPU tests of the tridiagonal analysis (not part of the
// reference surface): smallest eigenpair of the J x J symmetric tridiagonal (a, b[1..J)).
int machip_host_tridiag_smallest(const double* a, const double* b, int J, double* theta, double* s) {
    if (!a || !b || J < 1 || !theta || !s) return fail(MACHIP_BAD_ARG, "bad argument");
    tri::Smallest sm;
    std::vector<double> wk;
    tri::smallest_eigpair(a, b, J, nullptr, 0, 0.0, sm, wk);
    *theta = sm.theta;
    for (int i = 0; i < J; ++i) s[i] = sm.s[(size_t)i];
    return MACHIP_OK;
}

// Host-only: the deterministic reading of a Lanczos sequence (follow.h) over a finished record array -- tri3 = interleaved
// (alpha_j, beta_j, ||v_j||_1) triples valid through beta_J.  points[0..*npoints) = the analysis points visited (at most cap are
// stored), *jeff = order of the tridiagonal the sequence ends on (-1: the estimate never fell below e_target within J),
// *est = the estimate at the last point.  For the `-m "not gpu"` tests of the rule the solver ends its solves by.
int machip_host_follow_records(const double* tri3, int J, int n, double e_target, double tiny_l, int jcap, int* points, int cap,
                               int* npoints, int* jeff, double* est) {
    if (!tri3 || J < 1 || n < 2 || !npoints || !jeff || !est || !(e_target > 0.0) || cap < 0 || (cap > 0 && !points))
        return fail(MACHIP_BAD_ARG, "bad argument");
    std::vector<double> ha, hb, hl1, guess, wk;
    tri::Smallest sm;
    FollowCfg fc;
    fc.n = n; fc.jcap = jcap > 0 ? (jcap & ~1) : INT_MAX; fc.tiny_l = tiny_l > 0 ? tiny_l : 1.0;
    Follower F(fc, e_target, ha, hb, hl1, guess, sm, wk);
    *npoints = 0; *jeff = -1; *est = 0.0;
    while (F.next_a <= J) {
        const int a = F.next_a;
        if (!F.analyse(tri3, a)) return fail(MACHIP_BAD_ARG, "start vector is constant, zero or not finite");
        if (*npoints < cap) points[*npoints] = a;
        ++*npoints;
        *est = F.est;
        if (F.triggered()) { *jeff = F.Jeff; break; }
        if (a >= fc.jcap) break;
    }
    return MACHIP_OK;
}

}  // extern "C"
