// common.h -- shared host/device helpers of libssdk (gfx950 only; wave = 64).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/ssdk.h"

namespace ssdk {

constexpr int kWave = 64;

void set_error(const char* fmt, ...);
// Deterministic mode (ssdk_set_deterministic): every reduction of the training kernels runs in an order fixed by the launch, not by the
// hardware -- no fp32 atomics; see include/ssdk.h.
bool deterministic();

#define SSDK_REQUIRE(cond, code, ...)     \
    do {                                  \
        if (!(cond)) {                    \
            ::ssdk::set_error(__VA_ARGS__); \
            return (code);                \
        }                                 \
    } while (0)

// Launch check: reports configuration errors of the launch just issued (no device sync).
#define SSDK_CHECK_LAUNCH(name)                                                        \
    do {                                                                               \
        hipError_t e_ = hipGetLastError();                                             \
        if (e_ != hipSuccess) {                                                        \
            ::ssdk::set_error("%s: launch failed: %s", name, hipGetErrorString(e_));   \
            return (int)e_;                                                            \
        }                                                                              \
    } while (0)

#define SSDK_CHECK_HIP(expr)                                                             \
    do {                                                                                 \
        hipError_t e_ = (expr);                                                          \
        if (e_ != hipSuccess) {                                                          \
            ::ssdk::set_error("%s failed: %s", #expr, hipGetErrorString(e_));            \
            return (int)e_;                                                              \
        }                                                                                \
    } while (0)

__host__ __device__ static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// Bump allocator over a caller-provided workspace; every carve is 256-byte aligned.
struct Carver {
    char* base;
    size_t off;
    explicit Carver(void* p) : base(static_cast<char*>(p)), off(0) {}
    template <typename T>
    T* take(size_t n) {
        T* r = base ? reinterpret_cast<T*>(base + off) : nullptr;
        off += align_up(n * sizeof(T), 256);
        return r;
    }
};

#ifdef __HIPCC__

// torch.max / torch.min semantics (NaN propagates) -- bf/utils/box_utils.py:61-69 use torch.max/min.
__device__ __forceinline__ float tmaxf(float a, float b) { return (a > b || a != a) ? a : b; }
__device__ __forceinline__ float tminf(float a, float b) { return (a < b || a != a) ? a : b; }
// Tensor.clamp_(0): NaN stays NaN.
__device__ __forceinline__ float clamp0(float v) { return v < 0.0f ? 0.0f : v; }
// bf/utils/box_utils.py:38-46 area
__device__ __forceinline__ float area4(float x1, float y1, float x2, float y2) { return clamp0(x2 - x1) * clamp0(y2 - y1); }

// IoU of two corner boxes, op for op bf/utils/box_utils.py:49-101 (no FMA: this TU is built -ffp-contract=off).
__device__ __forceinline__ float iou_corner(float4 a, float area_a, float4 b, float area_b) {
    const float inter = area4(tmaxf(a.x, b.x), tmaxf(a.y, b.y), tminf(a.z, b.z), tminf(a.w, b.w));
    return inter / (area_a + area_b - inter);
}

// bf/utils/box_utils.py:16-23 to_corners
__device__ __forceinline__ float4 to_corners(float4 c) {
    return make_float4(c.x - c.z / 2.0f, c.y - c.w / 2.0f, c.x + c.z / 2.0f, c.y + c.w / 2.0f);
}

__device__ __forceinline__ int lane_id() { return threadIdx.x & (kWave - 1); }

// ---- wave-level reductions over 64 lanes (butterfly; every lane ends with the result) ----------------------
template <typename T, typename Op>
__device__ __forceinline__ T wave_allreduce(T v, Op op) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = op(v, __shfl_xor(v, m, kWave));
    return v;
}
struct OpAddF { __device__ __forceinline__ float operator()(float a, float b) const { return a + b; } };
struct OpAddD { __device__ __forceinline__ double operator()(double a, double b) const { return a + b; } };
struct OpAddI { __device__ __forceinline__ int operator()(int a, int b) const { return a + b; } };
struct OpMaxF { __device__ __forceinline__ float operator()(float a, float b) const { return fmaxf(a, b); } };
struct OpMaxU64 {
    __device__ __forceinline__ unsigned long long operator()(unsigned long long a, unsigned long long b) const { return a > b ? a : b; }
};

// Block-wide sum into thread 0 (deterministic for a fixed block size).  `red` = LDS float[blockDim.x / 64].
template <typename T>
__device__ __forceinline__ T block_sum(T v, T* red) {
    struct Add { __device__ __forceinline__ T operator()(T a, T b) const { return a + b; } };
    v = wave_allreduce(v, Add());
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane_id() == 0) red[w] = v;
    __syncthreads();
    T s = 0;
    if (threadIdx.x == 0)
        for (int i = 0; i < nw; ++i) s += red[i];
    return s;
}

// ---- scans ------------------------------------------------------------------------------------------------------------
// inclusive scan over the 64 lanes of a wave
template <typename T, typename Op>
__device__ __forceinline__ T wave_inclusive_scan(T v, Op op) {
    const int l = lane_id();
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
        const T t = __shfl_up(v, d, kWave);
        if (l >= d) v = op(t, v);
    }
    return v;
}
// inclusive scan over a workgroup of up to 1024 threads (thread order); `tmp` = LDS T[17]; *agg = the total.  Every thread calls it.
template <typename T, typename Op>
__device__ __forceinline__ T block_inclusive_scan(T v, Op op, T* tmp, T* agg) {
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6, l = lane_id();
    v = wave_inclusive_scan(v, op);
    __syncthreads();
    if (l == kWave - 1) tmp[w] = v;
    __syncthreads();
    T before = v;   // (value unused for wave 0)
    bool have = false;
    for (int i = 0; i < w; ++i) { before = have ? op(before, tmp[i]) : tmp[i]; have = true; }
    T total = tmp[0];
    for (int i = 1; i < nw; ++i) total = op(total, tmp[i]);
    if (agg) *agg = total;
    return have ? op(before, v) : v;
}

// Global -> LDS copy of n4 float4 (both 16-byte aligned) by a 256-thread workgroup with 8 loads per thread in flight.
// Loads AND stores are branch-free: indices past the end are clamped to the last element, which is then written several
// times with the same value.  (A `for (t < n4) lds[t] = src[t]` loop compiles to load / s_waitcnt vmcnt(0) / ds_write per
// trip -- one memory round trip per 4 KB -- and a predicated store makes the optimiser sink each load into its branch again.)
__device__ __forceinline__ void stage_tile_f4(float4* __restrict__ lds, const float4* __restrict__ src, int n4) {
    for (int base = 0; base < n4; base += 8 * 256) {
        float4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = src[min(base + (int)threadIdx.x + k * 256, n4 - 1)];
#pragma unroll
        for (int k = 0; k < 8; ++k) lds[min(base + (int)threadIdx.x + k * 256, n4 - 1)] = v[k];
    }
}

// Zero-fill as a KERNEL.  hipMemsetAsync is not used anywhere in the library: captured into a HIP graph, its memset node left a 16-byte
// counter block non-zero from the second replay on (ROCm 7.2, ssd_mb2_voc's loss workspace; the multibox loss came out divided by a
// garbage positive count), and as a stream operation it costs a kernel launch anyway.
static __global__ void __launch_bounds__(256) zero_fill_kernel(unsigned char* __restrict__ p, size_t n) {
    const size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x, step = (size_t)gridDim.x * blockDim.x;
    if (((uintptr_t)p & 15) == 0) {
        const uint4 z = make_uint4(0u, 0u, 0u, 0u);
        for (size_t i = i0; i < (n >> 4); i += step) reinterpret_cast<uint4*>(p)[i] = z;
        for (size_t i = (n & ~(size_t)15) + i0; i < n; i += step) p[i] = 0;
    } else {
        for (size_t i = i0; i < n; i += step) p[i] = 0;
    }
}
static inline hipError_t zero_async(void* p, size_t bytes, hipStream_t s) {
    if (!p || !bytes) return hipSuccess;
    const size_t want = (bytes / 16 + 255) / 256;
    const unsigned grid = (unsigned)(want < 1 ? 1 : (want > 2048 ? 2048 : want));
    hipLaunchKernelGGL(zero_fill_kernel, dim3(grid), dim3(256), 0, s, (unsigned char*)p, bytes);
    return hipGetLastError();
}

#endif  // __HIPCC__

}  // namespace ssdk
