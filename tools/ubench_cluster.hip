// ubench_cluster.hip -- how fast can W workgroups of ONE XCD run a chain of dependent "Lanczos-like" steps inside one launch?
// Each step: wait until every workgroup has stamped step s (its six partial sums travel with the stamps: one L2 round trip),
// reduce them, gather E records per workgroup through L2 (sc1 loads: bypass the per-CU vector cache), write the own rows of the
// next record array, wait for the stores to reach L2, stamp s + 1.  Build: hipcc -O3 --offload-arch=gfx950 -o ubench_cluster ubench_cluster.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

struct alignas(16) Pair { double v; long long stamp; };
struct alignas(16) Rec { double t, v; };
#ifndef STORE_SC1
#define STORE_SC1 0
#endif
constexpr bool kStoreSc1 = STORE_SC1;
constexpr int kSlot = 8;     // pairs per workgroup slot (6 used)

__device__ __forceinline__ long long wall() { return (long long)__builtin_amdgcn_s_memrealtime(); }   // 100 MHz
__device__ __forceinline__ int xcc_id() { return (int)__builtin_amdgcn_s_getreg(20 | (3 << 11)) & 15; }

__device__ __forceinline__ Pair load_pair_sc1(const Pair* p) {
    Pair r;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(*(__attribute__((ext_vector_type(4))) int*)&r) : "v"(p) : "memory");
    return r;
}
__device__ __forceinline__ void store_pair_sc1(Pair* p, Pair r) {
    if (kStoreSc1) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(*(__attribute__((ext_vector_type(4))) int*)&r) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(p), "v"(*(__attribute__((ext_vector_type(4))) int*)&r) : "memory");
}

template <int BLOCK, int EPT>      // EPT gathers per thread per step
__global__ __launch_bounds__(BLOCK) void k_cluster(Pair* slots, Rec* Z0, Rec* Z1, const int* __restrict__ cols, int n, int W, int S,
                                                   int stride, int* info, long long* clk, int mode) {
    if ((int)blockIdx.x % stride) return;
    const int rank = (int)blockIdx.x / stride;
    if (rank >= W) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    __shared__ double scoef[8];
    __shared__ double sred[BLOCK / 64][8];
    __shared__ int stop;
    if (tid == 0) { info[rank] = xcc_id(); stop = 0; }
    int mycol[EPT];
    for (int e = 0; e < EPT; ++e) mycol[e] = cols[((size_t)rank * BLOCK + tid) * EPT + e];
    const int rows = (n + W - 1) / W, r0 = rank * rows, r1 = min(n, r0 + rows);
    __syncthreads();
    const long long t0 = wall();
    for (int s = 0; s < S; ++s) {
        const Rec* __restrict__ Zc = (s & 1) ? Z1 : Z0;
        Rec* __restrict__ Zn = (s & 1) ? Z0 : Z1;
        if (wv == 0) {      // wait + partial sums in one round trip per poll
            double a[6] = {0, 0, 0, 0, 0, 0};
            long long tw = 0;
            for (;;) {
                bool ok = true;
                if (lane < W) {
                    __attribute__((ext_vector_type(4))) int raw[6];
#pragma unroll
                    for (int q = 0; q < 6; ++q)
                        asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(raw[q]) : "v"(slots + (size_t)lane * kSlot + q) : "memory");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                    for (int q = 0; q < 6; ++q) {
                        const Pair p = *(Pair*)&raw[q];
                        ok = ok && p.stamp == (long long)s;
                        a[q] = p.v;
                    }
                }
                if (__all(ok)) break;
                if (++tw > (1ll << 22)) { stop = 1; break; }     // a peer is not resident
            }
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                double v = lane < W ? a[q] : 0.0;
                for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
                if (lane == 0) scoef[q] = v;
            }
        }
        __syncthreads();
        if (stop) break;
        const double c0 = scoef[0] * 1e-3, c1 = scoef[1] * 1e-3;
        double st = 0.0, sv = 0.0;
        if (mode == 0) {
#pragma unroll
            for (int e = 0; e < EPT; ++e) {
                const Rec r = Zc[mycol[e]];       // plain loads (may be stale in the vector cache: timing reference only)
                st += r.t; sv += r.v;
            }
        } else {
            __attribute__((ext_vector_type(4))) int raw[EPT];
#pragma unroll
            for (int e = 0; e < EPT; ++e)
                asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(raw[e]) : "v"(Zc + mycol[e]) : "memory");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int e = 0; e < EPT; ++e) { const Rec r = *(Rec*)&raw[e]; st += r.t; sv += r.v; }
        }
        // own rows: a thread per row (rows <= BLOCK here or loop)
        double acc = 0.0;
        for (int r = r0 + tid; r < r1; r += BLOCK) {
            Rec o; o.t = st * 1e-3 + c0; o.v = sv * 1e-3 + c1;
            Pair pr; pr.v = o.t; pr.stamp = *(long long*)&o.v;
            store_pair_sc1((Pair*)(Zn + r), pr);
            acc += o.t;
        }
        for (int o = 32; o; o >>= 1) acc += __shfl_xor(acc, o);
        if (lane == 0) sred[wv][0] = acc;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // my stores have reached L2
        __syncthreads();
        if (tid < 6) {
            double v = 0.0;
            for (int w = 0; w < BLOCK / 64; ++w) v += sred[w][0];
            Pair p; p.v = v * 1e-6 + tid; p.stamp = (long long)(s + 1);
            store_pair_sc1(slots + (size_t)rank * kSlot + tid, p);
        }
    }
    if (tid == 0) { clk[rank] = wall() - t0; if (stop) info[64 + rank] = 1; }
}

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 10000;
    const int S = argc > 2 ? atoi(argv[2]) : 2000;
    constexpr int BLOCK = 512, EPT = 8;
    Pair* slots; Rec *Z0, *Z1; int* cols; int* info; long long* clk;
    CK(hipMalloc(&slots, sizeof(Pair) * kSlot * 64)); CK(hipMalloc(&Z0, sizeof(Rec) * n)); CK(hipMalloc(&Z1, sizeof(Rec) * n));
    CK(hipMalloc(&cols, sizeof(int) * 64 * BLOCK * EPT)); CK(hipMalloc(&info, sizeof(int) * 128)); CK(hipMalloc(&clk, sizeof(long long) * 64));
    std::vector<int> hc((size_t)64 * BLOCK * EPT);
    srand(1);
    for (auto& c : hc) c = rand() % n;
    CK(hipMemcpy(cols, hc.data(), sizeof(int) * hc.size(), hipMemcpyHostToDevice));
    std::vector<Rec> hz((size_t)n, Rec{1.0, 0.5});
    CK(hipMemcpy(Z0, hz.data(), sizeof(Rec) * n, hipMemcpyHostToDevice)); CK(hipMemcpy(Z1, hz.data(), sizeof(Rec) * n, hipMemcpyHostToDevice));
    for (int stride : {8, 1}) for (int mode : {1, 0}) for (int W : {2, 4, 8, 16, 32}) {
        CK(hipMemset(slots, 0, sizeof(Pair) * kSlot * 64)); CK(hipMemset(info, 0, sizeof(int) * 128));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0));
        k_cluster<BLOCK, EPT><<<W * stride, BLOCK>>>(slots, Z0, Z1, cols, n, W, S, stride, info, clk, mode);
        CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        int hi[128]; long long hk[64];
        CK(hipMemcpy(hi, info, sizeof(hi), hipMemcpyDeviceToHost)); CK(hipMemcpy(hk, clk, sizeof(hk), hipMemcpyDeviceToHost));
        int same = 1, stopped = 0;
        for (int r = 0; r < W; ++r) { same &= hi[r] == hi[0]; stopped |= hi[64 + r]; }
        printf("stride %d mode %s W %2d: %.3f us/step (kernel %.1f us; in-kernel %.3f us/step) xcc %d same=%d stopped=%d entries/step %d\n", stride, mode ? "sc1" : "plain", W,
               1e3 * ms / S, 1e3 * ms, 10.0 * hk[0] / 1e3 / S * 1.0, hi[0], same, stopped, W * BLOCK * EPT);
    }
    return 0;
}
