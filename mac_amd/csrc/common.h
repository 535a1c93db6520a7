// common.h -- error plumbing and wave64 / workgroup reduction helpers (gfx950).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>

#include "../../include/machip.h"

namespace machip {

constexpr int kWave = 64;     // CDNA4 wavefront
constexpr int kBlock = 256;   // 4 waves per workgroup everywhere
constexpr int kMaxGrid = 1024;  // partial-sum arrays are sized for this many workgroups

extern thread_local std::string g_err;

inline int fail(machip_status st, const std::string& msg) {
    g_err = msg;
    return (int)st;
}

#define HIP_TRY(expr)                                                                        \
    do {                                                                                     \
        hipError_t e__ = (expr);                                                             \
        if (e__ != hipSuccess)                                                               \
            return machip::fail(MACHIP_HIP_ERROR, std::string(#expr) + ": " +                \
                                                      hipGetErrorString(e__));               \
    } while (0)

#define ST_TRY(expr)                                                                         \
    do {                                                                                     \
        int s__ = (expr);                                                                    \
        if (s__ != MACHIP_OK) return s__;                                                    \
    } while (0)

// ---- device-side reductions -------------------------------------------------------------
// Sum over the 64 lanes of a wave; every lane gets the total (butterfly, fixed order).
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, kWave);
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, kWave));
    return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, kWave);
    return v;
}
// Sum over the lanes of a G-lane group (G power of two <= 64); all lanes get it.
template <int G>
__device__ __forceinline__ double group_sum(double v) {
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, kWave);
    return v;
}
template <int G>
__device__ __forceinline__ int group_sum_i(int v) {
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, kWave);
    return v;
}

// Workgroup (256 threads = 4 waves) sum; every thread gets the total.  `sm` = 4 doubles of LDS.
__device__ __forceinline__ double block_sum(double v, double* sm) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();  // protect sm against a previous use
    if ((threadIdx.x & 63) == 0) sm[w] = v;
    __syncthreads();
    return (sm[0] + sm[1]) + (sm[2] + sm[3]);
}
__device__ __forceinline__ double block_max(double v, double* sm) {
    v = wave_max(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm[w] = v;
    __syncthreads();
    return fmax(fmax(sm[0], sm[1]), fmax(sm[2], sm[3]));
}
__device__ __forceinline__ int block_sum_i(int v, int* sm) {
    v = wave_sum_i(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm[w] = v;
    __syncthreads();
    return (sm[0] + sm[1]) + (sm[2] + sm[3]);
}
// Every workgroup re-reduces an array of `cnt` per-workgroup partials in the same fixed
// order (deterministic, identical on every workgroup): thread t takes t, t+256, ...
__device__ __forceinline__ double reduce_partials(const double* __restrict__ part, int cnt,
                                                  double* sm) {
    double a = 0.0;
    for (int i = threadIdx.x; i < cnt; i += kBlock) a += part[i];
    return block_sum(a, sm);
}

}  // namespace machip
