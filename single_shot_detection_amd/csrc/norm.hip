// norm.hip -- BatchNorm2d (+ optional ReLU) on NHWC rows, forward and backward.
//
// Reference: bf/modules/conv.py:30-36 (Conv2dBn: conv -> BatchNorm2d -> ReLU, the SSD pyramid tail built by
// detection/detector_builder.py:57-109) and detection/modules/predictors.py:60-76 (RetinaNet tower: conv -> ReLU ->
// a BatchNorm2d per level).  torch.nn.BatchNorm2d semantics: training mode normalises with the biased batch variance
// and updates running_mean / running_var (unbiased) with `momentum`; eval mode uses the running statistics.
//
// All three kernels are HBM-bound streams over a [rows = B*H*W][C] matrix (channels contiguous, so a wave reads whole
// lines): bn_reduce_kernel accumulates two per-channel sums (64 rows per workgroup in registers, then one fp64 atomic
// per channel), bn_apply_kernel turns them into mean / rstd itself (and updates the running statistics), bn_apply_kernel /
// bn_bwd_apply_kernel are the elementwise passes with the ReLU fused.
#include "common.h"

namespace ssdk {

constexpr int kBnRows = 64;
// Rows per bn_reduce workgroup.  Measured over the SSD-300 tail (rocprofv3, all eight layers; SSDK_BN_ROWS overrides): 64 rows is the best
// single value (8.3 us per launch on average; 128: 10.0, 256: 14.3 -- fewer, longer chains of dependent loads; 32: 10.4, 16: 16.3 --
// the big maps then queue 4x the same-address fp64 atomics per channel).  Picking 16 / 32 rows for the small maps only (2.7 - 2.9 us
// against 4.1) did not move the step time beyond run-to-run noise and is not done.
static inline int bn_rows_per_block(long long rows) {
    if (const char* e = getenv("SSDK_BN_ROWS")) return atoi(e);
    // Large maps (RetinaNet's 63 x 63 level at batch 32: 127 k rows): every workgroup ends with one fp64 atomic per channel and sum, all on
    // the same 2C addresses, and with 64 rows 1 984 workgroups queue there.  bn_reduce<0> per launch, averaged over the tower's five levels
    // (rocprofv3, tools/bnrows_retina.sh): 64 rows everywhere 53.4 us; 256 rows everywhere 30.2 (the small levels then take 17 instead of
    // 6 us); 64 rows until the launch has N workgroups, then more rows per workgroup: N = 1536: 42.3, 1024: 36.6, 768: 32.3, 512: 29.1,
    // 384: 24.8, 256: 23.0, 192: 23.1 (backward statistics alike: 60.7 -> 28.9).  retina_rn50_500_coco step 52.97 -> 50.17 ms.
    // (round 4, tools/ab_step.sh SSDK_BN_WGS 256 128 64 on one box: M2Det 31.45 / 31.28 / 31.73 ms per step, RetinaNet 46.57 / 46.53 / 46.82, SSD-300
    // unchanged: 128 -- the mid-size maps of the M2Det neck, 16 k rows, spent ~20 of their 36 us in the 256-way atomics)
    const long long target = getenv("SSDK_BN_WGS") ? atoll(getenv("SSDK_BN_WGS")) : 128;
    long long r = (rows + target - 1) / target;
    r = (r + 15) / 16 * 16;
    return (int)(r < kBnRows ? kBnRows : (r > 4096 ? 4096 : r));
}

// MODE 0: s0 = sum x, s1 = sum x^2.   MODE 1 (backward): s0 = sum dy', s1 = sum dy' * xhat, dy' = relu ? dy * (y > 0) : dy
// Workgroup = 64 rows; wave w takes rows w, w+4, ...; lane l owns float4 columns l, l+64, ... (whole 1 KB lines per wave).
template <int MODE>
__global__ void __launch_bounds__(256) bn_reduce_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ dy,
                                                        long long rows, int rows_per_block, int C, const float* __restrict__ mean,
                                                        const float* __restrict__ rstd, int relu, double* __restrict__ sums) {
    __shared__ float4 s_part[2][4][64];
    const long long r0 = (long long)blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int C4 = C >> 2;
    // Rows of at most 32 float4 (C <= 128: the 128-channel smoothing layers of the M2Det neck, 48 of its 131 norms): a wave instruction
    // takes 64 / C4 consecutive rows instead of one row on half (a quarter, ...) of its lanes; the lanes that share a column are folded
    // with cross-lane adds at the end.
    const bool packed = C4 <= 32 && (64 % C4) == 0;
    const int rpw = packed ? 64 / C4 : 1;          // rows per wave instruction
    const int sub = packed ? lane / C4 : 0;        // this lane's row inside them
    if (blockIdx.x == 0 && threadIdx.x == 0) sums[2 * C] = (double)rows;   // (slot 2C: the count behind the sums)
    for (int cbase = 0; cbase < C4; cbase += 64) {
        const int c4 = packed ? lane - sub * C4 : cbase + lane;
        float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
        if (c4 < C4) {
            float4 m4 = a0, rs4 = a0;
            if (MODE == 1) { m4 = reinterpret_cast<const float4*>(mean)[c4]; rs4 = reinterpret_cast<const float4*>(rstd)[c4]; }
            // four row groups per trip, all their loads issued before the first is consumed (a row per trip made every wave wait one
            // memory round trip per row: 10-24 us on the small pyramid maps); rows past the end re-read the last row, weight 0
            for (long long v = wave; r0 + v * rpw < r1; v += 16) {
                float4 xv[4], gv[4], yv[4];
                float wgt[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const long long ru = r0 + (v + 4 * u) * rpw + sub;
                    wgt[u] = ru < r1 ? 1.0f : 0.0f;
                    const long long rr = ru < r1 ? ru : r1 - 1;
                    xv[u] = reinterpret_cast<const float4*>(x + rr * C)[c4];
                    if (MODE == 1) {
                        gv[u] = reinterpret_cast<const float4*>(dy + rr * C)[c4];
                        if (relu) yv[u] = reinterpret_cast<const float4*>(y + rr * C)[c4];
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (wgt[u] == 0.0f) continue;
                    const float4 xu = xv[u];
                    if (MODE == 0) {
                        a0.x += xu.x; a0.y += xu.y; a0.z += xu.z; a0.w += xu.w;
                        a1.x += xu.x * xu.x; a1.y += xu.y * xu.y; a1.z += xu.z * xu.z; a1.w += xu.w * xu.w;
                    } else {
                        float4 g = gv[u];
                        if (relu) {
                            if (!(yv[u].x > 0.f)) g.x = 0.f;
                            if (!(yv[u].y > 0.f)) g.y = 0.f;
                            if (!(yv[u].z > 0.f)) g.z = 0.f;
                            if (!(yv[u].w > 0.f)) g.w = 0.f;
                        }
                        a0.x += g.x; a0.y += g.y; a0.z += g.z; a0.w += g.w;
                        a1.x += g.x * ((xu.x - m4.x) * rs4.x); a1.y += g.y * ((xu.y - m4.y) * rs4.y);
                        a1.z += g.z * ((xu.z - m4.z) * rs4.z); a1.w += g.w * ((xu.w - m4.w) * rs4.w);
                    }
                }
            }
        }
        if (packed) {   // (uniform) lanes c4, c4 + C4, c4 + 2 C4, ... hold the same column: fold them into lane c4
            for (int d = C4; d < 64; d <<= 1) {
                a0.x += __shfl_xor(a0.x, d, 64); a0.y += __shfl_xor(a0.y, d, 64); a0.z += __shfl_xor(a0.z, d, 64); a0.w += __shfl_xor(a0.w, d, 64);
                a1.x += __shfl_xor(a1.x, d, 64); a1.y += __shfl_xor(a1.y, d, 64); a1.z += __shfl_xor(a1.z, d, 64); a1.w += __shfl_xor(a1.w, d, 64);
            }
        }
        __syncthreads();
        s_part[0][wave][lane] = a0;
        s_part[1][wave][lane] = a1;
        __syncthreads();
        if (wave < 2 && (packed ? lane < C4 : c4 < C4)) {  // wave 0 folds the s0 partials, wave 1 the s1 partials
            float4 t = s_part[wave][0][lane];
            for (int w = 1; w < 4; ++w) { const float4 u = s_part[wave][w][lane]; t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; }
            double* dst = sums + (size_t)wave * C + (size_t)(packed ? lane : c4) * 4;
            atomicAdd(dst + 0, (double)t.x); atomicAdd(dst + 1, (double)t.y); atomicAdd(dst + 2, (double)t.z); atomicAdd(dst + 3, (double)t.w);
        }
    }
}

// sums != NULL (training): mean / rstd come straight from the fp64 sums of bn_reduce_kernel<0> (no separate finalize
// launch); workgroup 0 also publishes save_mean / save_rstd for the backward and updates the running statistics.
// (inv_n = 1 / rows and unbias = rows / (rows - 1) once per thread, multiplications and one reciprocal square root per channel: the
// form with three fp64 divisions and a square root per channel was ~600 instructions in front of a thread's first load -- every thread of
// every apply launch derives its four channels' constants this way)
__device__ __forceinline__ void stats_from_sums(const double* sums, double inv_n, double unbias, int C, int c, float eps, float& m, float& rs, float& var_out) {
    const double mu = sums[c] * inv_n;
    double var = sums[C + c] * inv_n - mu * mu;
    if (var < 0.0) var = 0.0;
    m = (float)mu;
    rs = (float)rsqrt(var + (double)eps);
    var_out = (float)(var * unbias);
}

__global__ void __launch_bounds__(256) bn_apply_kernel(const float4* __restrict__ x, long long n4, int C4, const float4* __restrict__ mean,
                                                       const float4* __restrict__ rstd, const float4* __restrict__ gamma,
                                                       const float4* __restrict__ beta, int relu, float4* __restrict__ y,
                                                       const double* __restrict__ sums, long long rows, float eps, float momentum,
                                                       float* __restrict__ running_mean, float* __restrict__ running_var,
                                                       float* __restrict__ save_mean, float* __restrict__ save_rstd,
                                                       long long* __restrict__ num_batches_tracked, int count_on_device,
                                                       double* __restrict__ zero_after) {
    const int C = C4 * 4;
    if (zero_after && blockIdx.x == gridDim.x - 1)   // side job: zero-fill another sums buffer (nobody reads or writes it during this launch)
        for (int c = threadIdx.x; c < 2 * C + 2; c += blockDim.x) zero_after[c] = 0.0;
    if (sums && count_on_device) rows = (long long)sums[2 * C];   // synchronised statistics: the row count of all ranks travels with the sums
    if (num_batches_tracked && blockIdx.x == 0 && threadIdx.x == 0) *num_batches_tracked += 1;
    const double inv_n = sums ? 1.0 / (double)rows : 0.0, unbias = rows > 1 ? (double)rows / ((double)rows - 1.0) : 1.0;
    if (sums && blockIdx.x == 0) {
        for (int c = threadIdx.x; c < C; c += blockDim.x) {
            float m, rs, uv;
            stats_from_sums(sums, inv_n, unbias, C, c, eps, m, rs, uv);
            save_mean[c] = m;
            save_rstd[c] = rs;
            if (running_mean) running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * m;
            if (running_var) running_var[c] = (1.0f - momentum) * running_var[c] + momentum * uv;
        }
    }
    if (!sums && !mean && blockIdx.x == 0)   // evaluation mode: what the backward of a frozen norm needs
        for (int c = threadIdx.x; c < C; c += blockDim.x) {
            save_mean[c] = running_mean[c];
            save_rstd[c] = 1.0f / sqrtf(running_var[c] + eps);
        }
    // A thread's four channels do not change over its grid-stride trips when the stride is a multiple of the row length (always for
    // power-of-two C): mean / rstd -- four fp64 divisions and square roots from the sums -- gamma and beta are then worked out ONCE per
    // thread, not once per float4.
    const long long stride = (long long)gridDim.x * blockDim.x;
    const bool fixed_c = stride % C4 == 0;
    float4 m = make_float4(0.f, 0.f, 0.f, 0.f), rs = m, ga = m, be = m;
    auto load_consts = [&](int c) {
        if (sums) {
            float uv;
            stats_from_sums(sums, inv_n, unbias, C, 4 * c + 0, eps, m.x, rs.x, uv);
            stats_from_sums(sums, inv_n, unbias, C, 4 * c + 1, eps, m.y, rs.y, uv);
            stats_from_sums(sums, inv_n, unbias, C, 4 * c + 2, eps, m.z, rs.z, uv);
            stats_from_sums(sums, inv_n, unbias, C, 4 * c + 3, eps, m.w, rs.w, uv);
        } else if (mean) {
            m = mean[c];
            rs = rstd[c];
        } else {   // evaluation mode: the running statistics (torch: 1 / sqrt(running_var + eps))
            m = reinterpret_cast<const float4*>(running_mean)[c];
            const float4 rv = reinterpret_cast<const float4*>(running_var)[c];
            rs = make_float4(1.0f / sqrtf(rv.x + eps), 1.0f / sqrtf(rv.y + eps), 1.0f / sqrtf(rv.z + eps), 1.0f / sqrtf(rv.w + eps));
        }
        ga = gamma ? gamma[c] : make_float4(1.f, 1.f, 1.f, 1.f);
        be = beta ? beta[c] : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    const long long i0 = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (fixed_c && i0 < n4) load_consts((int)(i0 % C4));
    for (long long i = i0; i < n4; i += stride) {
        if (!fixed_c) load_consts((int)(i % C4));
        const float4 v = x[i];
        float4 o = make_float4((v.x - m.x) * rs.x * ga.x + be.x, (v.y - m.y) * rs.y * ga.y + be.y, (v.z - m.z) * rs.z * ga.z + be.z,
                               (v.w - m.w) * rs.w * ga.w + be.w);
        if (relu) o = make_float4(fmaxf(o.x, 0.f), fmaxf(o.y, 0.f), fmaxf(o.z, 0.f), fmaxf(o.w, 0.f));
        y[i] = o;
    }
}

// dx = gamma * rstd * (dy' - [training] (sum dy')/n - xhat * (sum dy' xhat)/n)
// sums: the sums the statistics are taken over (all ranks' when synchronised) with their row count in sums[2C] when count_on_device;
// sums_local: this rank's own sums -- the affine parameters' gradients (summed over ranks later by the gradient exchange, like any other).
__global__ void __launch_bounds__(256) bn_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ dy,
                                                           long long rows, int C, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           const float* __restrict__ gamma, const double* __restrict__ sums,
                                                           const double* __restrict__ sums_local, int count_on_device, int relu, int training,
                                                           float* __restrict__ dx, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                           double* __restrict__ zero_after) {
    // relu bit 0: the norm's own fused ReLU (dy counts where y > 0); bit 1: the norm's INPUT is the output of a ReLU whose gradient is
    // taken here as well (dx = 0 where x <= 0: conv -> ReLU -> BatchNorm of the RetinaNet tower, the producing convolution then skips its
    // own ReLU-gradient pass)
    const bool own_relu = (relu & 1) != 0, mask_input = (relu & 2) != 0;
    const long long total4 = rows * C / 4;   // (C % 4 == 0, 16-byte aligned buffers: checked by the host)
    const int C4 = C >> 2;
    if (zero_after && blockIdx.x == gridDim.x - 1)
        for (int c = threadIdx.x; c < 2 * C + 2; c += blockDim.x) zero_after[c] = 0.0;
    const double inv_n = 1.0 / (count_on_device ? sums[2 * C] : (double)rows);
    // A thread's four channels stay the same over its grid-stride trips whenever the stride is a multiple of the row length (C / 4 divides
    // 256 * gridDim for every power-of-two C): their five per-channel constants are then loaded and converted ONCE.  (Per element they
    // were five scattered loads and two fp64 multiplies: the float4 form of this kernel was slower than the scalar one until they moved.)
    const long long stride = (long long)gridDim.x * blockDim.x;
    const bool fixed_c = stride % C4 == 0;
    float k_mean[4], k_rs[4], k_ga[4], k_m1[4], k_m2[4];
    auto load_consts = [&](int c) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            k_mean[k] = mean[c + k];
            k_rs[k] = rstd[c + k];
            k_ga[k] = gamma ? gamma[c + k] : 1.0f;
            k_m1[k] = training ? (float)(sums[c + k] * inv_n) : 0.0f;
            k_m2[k] = training ? (float)(sums[C + c + k] * inv_n) : 0.0f;
        }
    };
    const long long i0 = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (fixed_c && i0 < total4) load_consts((int)(i0 % C4) * 4);
    for (long long i = i0; i < total4; i += stride) {
        if (!fixed_c) load_consts((int)(i % C4) * 4);
        const float4 g4 = reinterpret_cast<const float4*>(dy)[i];
        const float4 x4 = reinterpret_cast<const float4*>(x)[i];
        float4 y4 = make_float4(1.f, 1.f, 1.f, 1.f);
        if (own_relu) y4 = reinterpret_cast<const float4*>(y)[i];
        const float gg[4] = {g4.x, g4.y, g4.z, g4.w}, xx[4] = {x4.x, x4.y, x4.z, x4.w}, yy[4] = {y4.x, y4.y, y4.z, y4.w};
        float out[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float g = gg[k];
            if (own_relu && !(yy[k] > 0.0f)) g = 0.0f;
            const float xhat = (xx[k] - k_mean[k]) * k_rs[k];
            float v = g;
            if (training) v -= k_m1[k] + xhat * k_m2[k];
            out[k] = (mask_input && !(xx[k] > 0.0f)) ? 0.0f : k_ga[k] * k_rs[k] * v;
        }
        reinterpret_cast<float4*>(dx)[i] = make_float4(out[0], out[1], out[2], out[3]);
    }
    if (blockIdx.x == 0)
        for (int c = threadIdx.x; c < C; c += blockDim.x) {
            if (dbeta) dbeta[c] = (float)sums_local[c];
            if (dgamma) dgamma[c] = (float)sums_local[C + c];
        }
}

static inline int stream_blocks(long long items, int per_block) {
    long long b = (items + per_block - 1) / per_block;
    return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}
// Grid of the BatchNorm apply kernels (256 threads, one float4 per thread and trip): a thread keeps its four channels over all its
// grid-stride trips -- and works out their constants once -- when the stride, 256 * blocks, is a multiple of the row length C4.  For a
// power-of-two C every grid does; for the others (the seven stacked 128-channel reducers of the M2Det neck: C4 = 224) the block count is
// rounded down to a multiple of C4 / gcd(C4, 256): at 4 096 blocks such a launch re-derived mean / rstd from the fp64 sums for every
// float4 (16 x 896 x 64 x 64: forward 282 us against 188 at the rate of the 512-channel map, backward 484 against 330).
static inline int apply_blocks(long long n4, int C4) {
    const int b = stream_blocks(n4, 256);
    int g = C4, r = 256;
    while (r) { const int t = g % r; g = r; r = t; }   // gcd(C4, 256)
    const int m = C4 / g;
    return b >= m ? b / m * m : b;
}

}  // namespace ssdk

using namespace ssdk;

// a `sums` buffer: 2C sums, the row count at [2C], one double of padding (an even number of doubles: the zero-fill of an odd number
// is two fill kernels in the runtime, one for the 16-byte aligned part and one for the tail)
extern "C" size_t ssdk_batchnorm_workspace_bytes(int channels) { return align_up(((size_t)2 * channels + 2) * sizeof(double), 256); }

static int bn_stats(const float* x, long long rows, int channels, double* sums, bool zero_first, void* stream) {
    SSDK_REQUIRE(x && sums && rows > 0 && channels > 0, SSDK_E_INVALID, "ssdk_batchnorm_stats: bad arguments");
    SSDK_REQUIRE(channels % 4 == 0 && ((uintptr_t)x & 15) == 0, SSDK_E_UNSUPPORTED, "ssdk_batchnorm_stats: channels %% 4 != 0 or x not 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    if (zero_first) SSDK_CHECK_HIP(zero_async(sums, sizeof(double) * (2 * (size_t)channels + 2), s));
    const int rpb = bn_rows_per_block(rows);
    hipLaunchKernelGGL(bn_reduce_kernel<0>, dim3((unsigned)((rows + rpb - 1) / rpb)), dim3(256), 0, s, x, (const float*)nullptr,
                       (const float*)nullptr, rows, rpb, channels, (const float*)nullptr, (const float*)nullptr, 0, sums);
    SSDK_CHECK_LAUNCH("bn_reduce_kernel");
    return SSDK_OK;
}

extern "C" int ssdk_batchnorm_stats(const float* x, long long rows, int channels, double* sums, void* stream) {
    return bn_stats(x, rows, channels, sums, true, stream);
}

// sums += this call's partial sums (the caller holds a buffer of zeros, or is accumulating over several calls); sums[2C] = rows
extern "C" int ssdk_batchnorm_stats_accumulate(const float* x, long long rows, int channels, double* sums, void* stream) {
    return bn_stats(x, rows, channels, sums, false, stream);
}

static int bn_apply(const float* x, long long rows, int channels, const float* gamma, const float* beta, float* running_mean,
                    float* running_var, int64_t* num_batches_tracked, float momentum, float eps, int relu, float* y, float* save_mean,
                    float* save_rstd, const double* sums, int count_in_sums, double* zero_after, void* stream) {
    SSDK_REQUIRE(x && y && save_mean && save_rstd && sums && rows > 0 && channels > 0, SSDK_E_INVALID, "ssdk_batchnorm_apply: bad arguments");
    SSDK_REQUIRE(channels % 4 == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0, SSDK_E_UNSUPPORTED,
                 "ssdk_batchnorm_apply: channels %% 4 != 0 or buffers not 16-byte aligned");
    const long long n4 = rows * channels / 4;
    hipLaunchKernelGGL(bn_apply_kernel, dim3(apply_blocks(n4, channels / 4)), dim3(256), 0, (hipStream_t)stream, (const float4*)x, n4, channels / 4,
                       (const float4*)save_mean, (const float4*)save_rstd, (const float4*)gamma, (const float4*)beta, relu, (float4*)y, sums, rows,
                       eps, momentum, running_mean, running_var, save_mean, save_rstd, (long long*)num_batches_tracked, count_in_sums, zero_after);
    SSDK_CHECK_LAUNCH("bn_apply_kernel");
    return SSDK_OK;
}

extern "C" int ssdk_batchnorm_apply(const float* x, long long rows, int channels, const float* gamma, const float* beta,
                                    float* running_mean, float* running_var, int64_t* num_batches_tracked, float momentum, float eps,
                                    int relu, float* y, float* save_mean, float* save_rstd, const double* sums, int count_in_sums,
                                    void* stream) {
    return bn_apply(x, rows, channels, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps, relu, y, save_mean, save_rstd,
                    sums, count_in_sums, nullptr, stream);
}

// The layer keeps two sums buffers of its own: `sums` holds zeros on entry (the layer's previous backward call zeroed it), `zero_after`
// -- the buffer the backward will accumulate into -- is zero-filled by the apply launch.  No fill launch in front of the statistics.
extern "C" int ssdk_batchnorm_fwd_chained(const float* x, long long rows, int channels, const float* gamma, const float* beta,
                                          float* running_mean, float* running_var, int64_t* num_batches_tracked, float momentum, float eps,
                                          int relu, float* y, float* save_mean, float* save_rstd, double* sums, double* zero_after,
                                          void* stream) {
    SSDK_REQUIRE(sums && sums != zero_after, SSDK_E_INVALID, "ssdk_batchnorm_fwd_chained: sums missing or equal to zero_after");
    const int rc = bn_stats(x, rows, channels, sums, false, stream);
    if (rc) return rc;
    return bn_apply(x, rows, channels, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps, relu, y, save_mean, save_rstd,
                    sums, 0, zero_after, stream);
}

// ... and when the statistics are already in `sums` (accumulated by the producing convolution's epilogue, ssdk_conv_desc::stats): the apply
// half alone, with the same side job of zero-filling the layer's other buffer.
extern "C" int ssdk_batchnorm_apply_chained(const float* x, long long rows, int channels, const float* gamma, const float* beta,
                                            float* running_mean, float* running_var, int64_t* num_batches_tracked, float momentum, float eps,
                                            int relu, float* y, float* save_mean, float* save_rstd, const double* sums, double* zero_after,
                                            void* stream) {
    SSDK_REQUIRE(sums && sums != zero_after, SSDK_E_INVALID, "ssdk_batchnorm_apply_chained: sums missing or equal to zero_after");
    return bn_apply(x, rows, channels, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps, relu, y, save_mean, save_rstd,
                    sums, 0, zero_after, stream);
}

extern "C" int ssdk_batchnorm_fwd(const float* x, long long rows, int channels, const float* gamma, const float* beta,
                                  float* running_mean, float* running_var, int64_t* num_batches_tracked, float momentum, float eps,
                                  int training, int relu, float* y, float* save_mean, float* save_rstd, void* workspace,
                                  size_t workspace_bytes, void* stream) {
    SSDK_REQUIRE(x && y && save_mean && save_rstd && rows > 0 && channels > 0, SSDK_E_INVALID, "ssdk_batchnorm_fwd: bad arguments");
    SSDK_REQUIRE(channels % 4 == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0, SSDK_E_UNSUPPORTED,
                 "ssdk_batchnorm_fwd: channels %% 4 != 0 or buffers not 16-byte aligned");
    SSDK_REQUIRE(training || (running_mean && running_var), SSDK_E_INVALID, "ssdk_batchnorm_fwd: eval mode needs running statistics");
    hipStream_t s = (hipStream_t)stream;
    if (training) {   // = ssdk_batchnorm_stats + ssdk_batchnorm_apply on this process's rows alone
        SSDK_REQUIRE(workspace && workspace_bytes >= ssdk_batchnorm_workspace_bytes(channels), SSDK_E_WORKSPACE, "ssdk_batchnorm_fwd: workspace too small");
        const int rc = ssdk_batchnorm_stats(x, rows, channels, (double*)workspace, stream);
        if (rc) return rc;
        return ssdk_batchnorm_apply(x, rows, channels, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps, relu, y,
                                    save_mean, save_rstd, (const double*)workspace, 0, stream);
    }
    // evaluation mode: ONE launch -- the apply kernel takes mean / rstd from the running statistics itself (mean == NULL) and leaves
    // save_mean / save_rstd for a backward through the frozen norm (a launch of their own before: 8 per SSD-300 evaluation step)
    SSDK_REQUIRE((((uintptr_t)running_mean | (uintptr_t)running_var) & 15) == 0, SSDK_E_UNSUPPORTED, "ssdk_batchnorm_fwd: running statistics not 16-byte aligned");
    const long long n4 = rows * channels / 4;
    hipLaunchKernelGGL(bn_apply_kernel, dim3(apply_blocks(n4, channels / 4)), dim3(256), 0, s, (const float4*)x, n4, channels / 4, (const float4*)nullptr,
                       (const float4*)nullptr, (const float4*)gamma, (const float4*)beta, relu, (float4*)y, (const double*)nullptr, rows, eps,
                       momentum, running_mean, running_var, save_mean, save_rstd, (long long*)nullptr, 0, (double*)nullptr);
    SSDK_CHECK_LAUNCH("bn_apply_kernel");
    return SSDK_OK;
}

static int bn_bwd_stats(const float* x, const float* y, const float* dy, long long rows, int channels, const float* save_mean,
                        const float* save_rstd, int relu, double* sums, bool zero_first, void* stream) {
    relu &= 1;   // (bit 1 -- the norm's input is a ReLU output -- concerns dx only: the sums are over the gradient of the norm's OUTPUT)
    SSDK_REQUIRE(x && dy && sums && save_mean && save_rstd && rows > 0 && channels > 0 && (!relu || y), SSDK_E_INVALID,
                 "ssdk_batchnorm_bwd_stats: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    if (zero_first) SSDK_CHECK_HIP(zero_async(sums, sizeof(double) * (2 * (size_t)channels + 2), s));
    const int rpb = bn_rows_per_block(rows);
    hipLaunchKernelGGL(bn_reduce_kernel<1>, dim3((unsigned)((rows + rpb - 1) / rpb)), dim3(256), 0, s, x, y, dy, rows, rpb, channels, save_mean,
                       save_rstd, relu, sums);
    SSDK_CHECK_LAUNCH("bn_reduce_kernel");
    return SSDK_OK;
}

extern "C" int ssdk_batchnorm_bwd_stats(const float* x, const float* y, const float* dy, long long rows, int channels, const float* save_mean,
                                        const float* save_rstd, int relu, double* sums, void* stream) {
    return bn_bwd_stats(x, y, dy, rows, channels, save_mean, save_rstd, relu, sums, true, stream);
}

static int bn_bwd_apply(const float* x, const float* y, const float* dy, long long rows, int channels, const float* gamma,
                        const float* save_mean, const float* save_rstd, int relu, int training, const double* sums,
                        const double* sums_local, const double* total_rows, float* dx, float* dgamma, float* dbeta, double* zero_after,
                        void* stream) {
    SSDK_REQUIRE(x && dy && dx && sums && save_mean && save_rstd && rows > 0 && channels > 0 && (!(relu & 1) || y), SSDK_E_INVALID,
                 "ssdk_batchnorm_bwd_apply: bad arguments");
    SSDK_REQUIRE(channels % 4 == 0 && (((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx | (uintptr_t)y) & 15) == 0, SSDK_E_UNSUPPORTED,
                 "ssdk_batchnorm_bwd_apply: channels %% 4 != 0 or buffers not 16-byte aligned");
    SSDK_REQUIRE(!total_rows || total_rows == sums + 2 * (size_t)channels, SSDK_E_INVALID,
                 "ssdk_batchnorm_bwd_apply: total_rows must be the slot behind the sums (sums + 2 * channels) or NULL");
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(apply_blocks(rows * channels / 4, channels / 4)), dim3(256), 0, (hipStream_t)stream, x, y, dy, rows, channels,
                       save_mean, save_rstd, gamma, sums, sums_local ? sums_local : sums, total_rows ? 1 : 0, relu, training, dx, dgamma, dbeta,
                       zero_after);
    SSDK_CHECK_LAUNCH("bn_bwd_apply_kernel");
    return SSDK_OK;
}

extern "C" int ssdk_batchnorm_bwd_apply(const float* x, const float* y, const float* dy, long long rows, int channels, const float* gamma,
                                        const float* save_mean, const float* save_rstd, int relu, int training, const double* sums,
                                        const double* sums_local, const double* total_rows, float* dx, float* dgamma, float* dbeta,
                                        void* stream) {
    return bn_bwd_apply(x, y, dy, rows, channels, gamma, save_mean, save_rstd, relu, training, sums, sums_local, total_rows, dx, dgamma, dbeta,
                        nullptr, stream);
}

// backward of ssdk_batchnorm_fwd_chained: `sums` (zeroed by the forward's apply launch) takes the backward sums, `zero_after` -- the
// layer's forward buffer -- is zero-filled for the next forward.
extern "C" int ssdk_batchnorm_bwd_chained(const float* x, const float* y, const float* dy, long long rows, int channels, const float* gamma,
                                          const float* save_mean, const float* save_rstd, int relu, float* dx, float* dgamma, float* dbeta,
                                          double* sums, double* zero_after, void* stream) {
    SSDK_REQUIRE(sums && sums != zero_after && dx, SSDK_E_INVALID, "ssdk_batchnorm_bwd_chained: bad arguments");
    const int rc = bn_bwd_stats(x, y, dy, rows, channels, save_mean, save_rstd, relu, sums, false, stream);
    if (rc) return rc;
    return bn_bwd_apply(x, y, dy, rows, channels, gamma, save_mean, save_rstd, relu, 1, sums, nullptr, nullptr, dx, dgamma, dbeta, zero_after, stream);
}

extern "C" int ssdk_batchnorm_bwd(const float* x, const float* y, const float* dy, long long rows, int channels, const float* gamma,
                                  const float* save_mean, const float* save_rstd, int relu, int training, float* dx, float* dgamma,
                                  float* dbeta, void* workspace, size_t workspace_bytes, void* stream) {
    SSDK_REQUIRE(x && dy && dx && save_mean && save_rstd && rows > 0 && channels > 0 && (!(relu & 1) || y), SSDK_E_INVALID, "ssdk_batchnorm_bwd: bad arguments");
    SSDK_REQUIRE(workspace && workspace_bytes >= ssdk_batchnorm_workspace_bytes(channels), SSDK_E_WORKSPACE, "ssdk_batchnorm_bwd: workspace too small");
    const int rc = ssdk_batchnorm_bwd_stats(x, y, dy, rows, channels, save_mean, save_rstd, relu, (double*)workspace, stream);
    if (rc) return rc;
    return ssdk_batchnorm_bwd_apply(x, y, dy, rows, channels, gamma, save_mean, save_rstd, relu, training, (const double*)workspace, nullptr, nullptr,
                                    dx, dgamma, dbeta, stream);
}

// dx = (y > 0) ? dy : 0  -- undoes a ReLU that was fused into a convolution epilogue (predictors.py:67-68)
namespace ssdk {
__global__ void __launch_bounds__(256) relu_bwd_kernel(const float4* __restrict__ y, const float4* __restrict__ dy, long long n4, float4* __restrict__ dx) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        const float4 a = y[i], g = dy[i];
        dx[i] = make_float4(a.x > 0.f ? g.x : 0.f, a.y > 0.f ? g.y : 0.f, a.z > 0.f ? g.z : 0.f, a.w > 0.f ? g.w : 0.f);
    }
}
}  // namespace ssdk

extern "C" int ssdk_relu_bwd(const float* y, const float* dy, long long n, float* dx, void* stream) {
    SSDK_REQUIRE(y && dy && dx && n > 0 && n % 4 == 0, SSDK_E_INVALID, "ssdk_relu_bwd: bad arguments (n %% 4 == 0)");
    hipLaunchKernelGGL(relu_bwd_kernel, dim3(stream_blocks(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, (const float4*)y, (const float4*)dy, n / 4, (float4*)dx);
    SSDK_CHECK_LAUNCH("relu_bwd_kernel");
    return SSDK_OK;
}

// ---- FPN top-down step: out = fine + nearest_upsample(coarse) -------------------------------------
// Reference: bf/modules/features.py:106-107  features[i] += F.interpolate(features[i+1], size=features[i].size()[2:],
// mode='nearest').  torch 'nearest': src = min(floor(dst * (float)in / out), in - 1).
namespace ssdk {
__device__ __forceinline__ int nearest_src(int dst, float scale, int in) {
    const int s = (int)floorf((float)dst * scale);
    return s < in - 1 ? s : in - 1;
}

__global__ void __launch_bounds__(256) upsample_add_kernel(const float4* __restrict__ fine, const float4* __restrict__ coarse, int B, int Hf,
                                                           int Wf, int Hc, int Wc, int C4, float4* __restrict__ out) {
    const long long total = (long long)B * Hf * Wf * C4;
    const float sh = (float)Hc / (float)Hf, sw = (float)Wc / (float)Wf;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4);
        long long p = i / C4;
        const int x = (int)(p % Wf); p /= Wf;
        const int y = (int)(p % Hf);
        const int b = (int)(p / Hf);
        const float4 a = fine ? fine[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 u = coarse[(((long long)b * Hc + nearest_src(y, sh, Hc)) * Wc + nearest_src(x, sw, Wc)) * C4 + c];
        out[i] = make_float4(a.x + u.x, a.y + u.y, a.z + u.z, a.w + u.w);
    }
}

// dcoarse[yc][xc] = sum of dout over the fine pixels whose nearest source is (yc, xc)  (gather form: deterministic)
__global__ void __launch_bounds__(256) upsample_add_bwd_kernel(const float4* __restrict__ dout, int B, int Hf, int Wf, int Hc, int Wc, int C4,
                                                               float4* __restrict__ dcoarse) {
    const long long total = (long long)B * Hc * Wc * C4;
    const float sh = (float)Hc / (float)Hf, sw = (float)Wc / (float)Wf;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4);
        long long p = i / C4;
        const int xc = (int)(p % Wc); p /= Wc;
        const int yc = (int)(p % Hc);
        const int b = (int)(p / Hc);
        // candidate fine rows/cols: around yc / sh
        const int y0 = max(0, (int)floorf((float)yc / sh) - 2), y1 = min(Hf - 1, (int)ceilf((float)(yc + 1) / sh) + 2);
        const int x0 = max(0, (int)floorf((float)xc / sw) - 2), x1 = min(Wf - 1, (int)ceilf((float)(xc + 1) / sw) + 2);
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int y = y0; y <= y1; ++y) {
            if (nearest_src(y, sh, Hc) != yc) continue;
            for (int x = x0; x <= x1; ++x) {
                if (nearest_src(x, sw, Wc) != xc) continue;
                const float4 g = dout[(((long long)b * Hf + y) * Wf + x) * C4 + c];
                s.x += g.x; s.y += g.y; s.z += g.z; s.w += g.w;
            }
        }
        dcoarse[i] = s;
    }
}
}  // namespace ssdk

extern "C" int ssdk_upsample_nearest_add_fwd(const float* fine, const float* coarse, int batch, int hf, int wf, int hc, int wc, int channels,
                                             float* out, void* stream) {
    SSDK_REQUIRE(coarse && out && batch > 0 && hf > 0 && wf > 0 && hc > 0 && wc > 0 && channels > 0 && channels % 4 == 0, SSDK_E_INVALID,
                 "ssdk_upsample_nearest_add_fwd: bad arguments (channels %% 4 == 0)");
    const long long n4 = (long long)batch * hf * wf * channels / 4;
    hipLaunchKernelGGL(upsample_add_kernel, dim3(stream_blocks(n4, 256)), dim3(256), 0, (hipStream_t)stream, (const float4*)fine, (const float4*)coarse,
                       batch, hf, wf, hc, wc, channels / 4, (float4*)out);
    SSDK_CHECK_LAUNCH("upsample_add_kernel");
    return SSDK_OK;
}

extern "C" int ssdk_upsample_nearest_add_bwd(const float* dout, int batch, int hf, int wf, int hc, int wc, int channels, float* dcoarse,
                                             void* stream) {
    SSDK_REQUIRE(dout && dcoarse && batch > 0 && hf > 0 && wf > 0 && hc > 0 && wc > 0 && channels > 0 && channels % 4 == 0, SSDK_E_INVALID,
                 "ssdk_upsample_nearest_add_bwd: bad arguments (channels %% 4 == 0)");
    const long long n4 = (long long)batch * hc * wc * channels / 4;
    hipLaunchKernelGGL(upsample_add_bwd_kernel, dim3(stream_blocks(n4, 256)), dim3(256), 0, (hipStream_t)stream, (const float4*)dout, batch, hf, wf,
                       hc, wc, channels / 4, (float4*)dcoarse);
    SSDK_CHECK_LAUNCH("upsample_add_bwd_kernel");
    return SSDK_OK;
}


// ---- SFAM (M2Det scale-wise feature aggregation): squeeze-excite gate ---------------------------------
// Reference: bf/modules/features.py:286-298  x = adaptive_avg_pool2d(f, 1); x = fc2(relu(fc1(x))); out = f * sigmoid(x).
// The two 1x1 "fc" convolutions run on the GEMM kernels; these are the pool and the gate (with their backward).
namespace ssdk {
// mean over the HW pixels of each image: x [B][HW][C] -> out [B][C].  One workgroup per (image, 16-float4 column block): thread
// (column q of 16, row phase r of 16); a wave instruction reads four rows x 256 contiguous bytes, four row groups are in flight per
// thread, the 16 phases are folded through LDS.  (Round 3's form -- 64 columns per workgroup, one row in flight per wave -- put M2Det's
// SFAM pool of a [16, 1024, 64, 64] map, 268 MB, on 64 workgroups: ~600 us; this one uses 256.)
constexpr int kPoolCols = 16;   // float4 columns per workgroup
__global__ void __launch_bounds__(256) avgpool_kernel(const float4* __restrict__ x, int HW, int C4, float4* __restrict__ out) {
    __shared__ float4 s_part[16][kPoolCols];
    const int b = blockIdx.y, q = threadIdx.x & (kPoolCols - 1), r = threadIdx.x / kPoolCols;
    const int c4 = blockIdx.x * kPoolCols + q;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c4 < C4) {
        const float4* base = x + (long long)b * HW * C4 + c4;
        int p = r;
        for (; p + 48 < HW; p += 64) {   // rows p, p + 16, p + 32, p + 48: four loads in flight
            const float4 v0 = base[(long long)p * C4], v1 = base[(long long)(p + 16) * C4], v2 = base[(long long)(p + 32) * C4], v3 = base[(long long)(p + 48) * C4];
            a.x += (v0.x + v1.x) + (v2.x + v3.x); a.y += (v0.y + v1.y) + (v2.y + v3.y);
            a.z += (v0.z + v1.z) + (v2.z + v3.z); a.w += (v0.w + v1.w) + (v2.w + v3.w);
        }
        for (; p < HW; p += 16) {
            const float4 v = base[(long long)p * C4];
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
    }
    s_part[r][q] = a;
    __syncthreads();
    if (r == 0 && c4 < C4) {
        float4 t = s_part[0][q];
        for (int w = 1; w < 16; ++w) { const float4 u = s_part[w][q]; t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; }
        const float inv = 1.0f / (float)HW;
        out[(long long)b * C4 + c4] = make_float4(t.x * inv, t.y * inv, t.z * inv, t.w * inv);
    }
}
__global__ void __launch_bounds__(256) avgpool_bwd_kernel(const float4* __restrict__ dout, int B, int HW, int C4, float4* __restrict__ dx) {
    const long long total = (long long)B * HW * C4;
    const float inv = 1.0f / (float)HW;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4);
        const int b = (int)(i / ((long long)HW * C4));
        const float4 g = dout[(long long)b * C4 + c];
        dx[i] = make_float4(g.x * inv, g.y * inv, g.z * inv, g.w * inv);
    }
}
__device__ __forceinline__ float sigm(float v) { return 1.0f / (1.0f + __expf(-v)); }
// out = x * sigmoid(z[b][c])
__global__ void __launch_bounds__(256) gate_kernel(const float4* __restrict__ x, const float4* __restrict__ z, int B, int HW, int C4,
                                                   float4* __restrict__ out) {
    const long long total = (long long)B * HW * C4;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4);
        const int b = (int)(i / ((long long)HW * C4));
        const float4 v = x[i], q = z[(long long)b * C4 + c];
        out[i] = make_float4(v.x * sigm(q.x), v.y * sigm(q.y), v.z * sigm(q.z), v.w * sigm(q.w));
    }
}
// dx = dout * sigmoid(z);  dz[b][c] = sigmoid'(z) * sum_hw dout * x   (one workgroup per (image, 16-float4 column block), as avgpool_kernel)
__global__ void __launch_bounds__(256) gate_bwd_kernel(const float4* __restrict__ x, const float4* __restrict__ z, const float4* __restrict__ dout,
                                                       int HW, int C4, float4* __restrict__ dx, float4* __restrict__ dz) {
    __shared__ float4 s_part[16][kPoolCols];
    const int b = blockIdx.y, q = threadIdx.x & (kPoolCols - 1), r = threadIdx.x / kPoolCols;
    const int c4 = blockIdx.x * kPoolCols + q;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), sg = a;
    if (c4 < C4) {
        const float4 zq = z[(long long)b * C4 + c4];
        sg = make_float4(sigm(zq.x), sigm(zq.y), sigm(zq.z), sigm(zq.w));
        const long long base = (long long)b * HW * C4 + c4;
        int p = r;
        for (; p + 16 < HW; p += 32) {   // two rows (of both operands) in flight
            const long long i0 = base + (long long)p * C4, i1 = base + (long long)(p + 16) * C4;
            const float4 g0 = dout[i0], v0 = x[i0], g1 = dout[i1], v1 = x[i1];
            dx[i0] = make_float4(g0.x * sg.x, g0.y * sg.y, g0.z * sg.z, g0.w * sg.w);
            dx[i1] = make_float4(g1.x * sg.x, g1.y * sg.y, g1.z * sg.z, g1.w * sg.w);
            a.x += g0.x * v0.x + g1.x * v1.x; a.y += g0.y * v0.y + g1.y * v1.y; a.z += g0.z * v0.z + g1.z * v1.z; a.w += g0.w * v0.w + g1.w * v1.w;
        }
        for (; p < HW; p += 16) {
            const long long i = base + (long long)p * C4;
            const float4 g = dout[i], v = x[i];
            dx[i] = make_float4(g.x * sg.x, g.y * sg.y, g.z * sg.z, g.w * sg.w);
            a.x += g.x * v.x; a.y += g.y * v.y; a.z += g.z * v.z; a.w += g.w * v.w;
        }
    }
    s_part[r][q] = a;
    __syncthreads();
    if (r == 0 && c4 < C4) {
        float4 t = s_part[0][q];
        for (int w = 1; w < 16; ++w) { const float4 u = s_part[w][q]; t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; }
        dz[(long long)b * C4 + c4] = make_float4(t.x * sg.x * (1.f - sg.x), t.y * sg.y * (1.f - sg.y), t.z * sg.z * (1.f - sg.z), t.w * sg.w * (1.f - sg.w));
    }
}
}  // namespace ssdk

static int check_bhwc(const char* fn, int batch, int hw, int channels) {
    SSDK_REQUIRE(batch > 0 && batch <= 65535 && hw > 0 && channels > 0 && channels % 4 == 0, SSDK_E_INVALID, "%s: batch=%d hw=%d channels=%d (%% 4 == 0)", fn, batch, hw, channels);
    return SSDK_OK;
}

extern "C" int ssdk_global_avgpool_fwd(const float* x, int batch, int hw, int channels, float* out, void* stream) {
    int rc = check_bhwc("ssdk_global_avgpool_fwd", batch, hw, channels);
    if (rc) return rc;
    SSDK_REQUIRE(x && out, SSDK_E_INVALID, "ssdk_global_avgpool_fwd: null pointer");
    hipLaunchKernelGGL(avgpool_kernel, dim3(cdiv(channels / 4, ssdk::kPoolCols), batch), dim3(256), 0, (hipStream_t)stream, (const float4*)x, hw, channels / 4, (float4*)out);
    SSDK_CHECK_LAUNCH("avgpool_kernel");
    return SSDK_OK;
}
extern "C" int ssdk_global_avgpool_bwd(const float* dout, int batch, int hw, int channels, float* dx, void* stream) {
    int rc = check_bhwc("ssdk_global_avgpool_bwd", batch, hw, channels);
    if (rc) return rc;
    SSDK_REQUIRE(dout && dx, SSDK_E_INVALID, "ssdk_global_avgpool_bwd: null pointer");
    hipLaunchKernelGGL(avgpool_bwd_kernel, dim3(stream_blocks((long long)batch * hw * channels / 4, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float4*)dout, batch, hw, channels / 4, (float4*)dx);
    SSDK_CHECK_LAUNCH("avgpool_bwd_kernel");
    return SSDK_OK;
}
extern "C" int ssdk_sigmoid_gate_fwd(const float* x, const float* z, int batch, int hw, int channels, float* out, void* stream) {
    int rc = check_bhwc("ssdk_sigmoid_gate_fwd", batch, hw, channels);
    if (rc) return rc;
    SSDK_REQUIRE(x && z && out, SSDK_E_INVALID, "ssdk_sigmoid_gate_fwd: null pointer");
    hipLaunchKernelGGL(gate_kernel, dim3(stream_blocks((long long)batch * hw * channels / 4, 256)), dim3(256), 0, (hipStream_t)stream, (const float4*)x,
                       (const float4*)z, batch, hw, channels / 4, (float4*)out);
    SSDK_CHECK_LAUNCH("gate_kernel");
    return SSDK_OK;
}
extern "C" int ssdk_sigmoid_gate_bwd(const float* x, const float* z, const float* dout, int batch, int hw, int channels, float* dx, float* dz,
                                     void* stream) {
    int rc = check_bhwc("ssdk_sigmoid_gate_bwd", batch, hw, channels);
    if (rc) return rc;
    SSDK_REQUIRE(x && z && dout && dx && dz, SSDK_E_INVALID, "ssdk_sigmoid_gate_bwd: null pointer");
    hipLaunchKernelGGL(gate_bwd_kernel, dim3(cdiv(channels / 4, ssdk::kPoolCols), batch), dim3(256), 0, (hipStream_t)stream, (const float4*)x, (const float4*)z,
                       (const float4*)dout, hw, channels / 4, (float4*)dx, (float4*)dz);
    SSDK_CHECK_LAUNCH("gate_bwd_kernel");
    return SSDK_OK;
}

// ---- SFAM over PIECES ------------------------------------------------------------------------------------------------------------
// The reference concatenates the eight TUM outputs of a scale (features.py:385, torch.cat of [B, 128, H, W] maps) and hands the
// [B, 1024, H, W] map to the gate (:286-298).  The concatenated map is never needed as such: the pool and the gate read the pieces where
// they are and the gate writes the one map the heads read; the backward pass writes each piece's gradient as its own contiguous map --
// dout * sigmoid(z) + dpool / HW in ONE pass (torch.cat's backward made eight strided slices that every consumer first copied, and
// autograd added the pool's and the gate's gradient maps with one more pass).
namespace ssdk {
constexpr int kMaxPieces = 8;
struct Pieces {
    const float4* p[kMaxPieces];
    int n, Cp4;   // pieces, float4 columns per piece
};
struct PiecesOut {
    float4* p[kMaxPieces];
    int n, Cp4;
};
// mean over the pixels of every image: pieces [B][HW][Cp] -> out [B][n * Cp]   (avgpool_kernel's layout of the work)
__global__ void __launch_bounds__(256) pool_pieces_kernel(Pieces ps, int HW, float4* __restrict__ out) {
    __shared__ float4 s_part[16][kPoolCols];
    const int C4 = ps.n * ps.Cp4;
    const int b = blockIdx.y, q = threadIdx.x & (kPoolCols - 1), r = threadIdx.x / kPoolCols;
    const int c4 = blockIdx.x * kPoolCols + q;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c4 < C4) {
        const int k = c4 / ps.Cp4, w = c4 - k * ps.Cp4;
        const int Cp4 = ps.Cp4;
        const float4* base = ps.p[k] + (long long)b * HW * Cp4 + w;
        int p = r;
        for (; p + 48 < HW; p += 64) {
            const float4 v0 = base[(long long)p * Cp4], v1 = base[(long long)(p + 16) * Cp4], v2 = base[(long long)(p + 32) * Cp4], v3 = base[(long long)(p + 48) * Cp4];
            a.x += (v0.x + v1.x) + (v2.x + v3.x); a.y += (v0.y + v1.y) + (v2.y + v3.y);
            a.z += (v0.z + v1.z) + (v2.z + v3.z); a.w += (v0.w + v1.w) + (v2.w + v3.w);
        }
        for (; p < HW; p += 16) {
            const float4 v = base[(long long)p * Cp4];
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
    }
    s_part[r][q] = a;
    __syncthreads();
    if (r == 0 && c4 < C4) {
        float4 t = s_part[0][q];
        for (int w = 1; w < 16; ++w) { const float4 u = s_part[w][q]; t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; }
        const float inv = 1.0f / (float)HW;
        out[(long long)b * C4 + c4] = make_float4(t.x * inv, t.y * inv, t.z * inv, t.w * inv);
    }
}
// out [B][HW][n * Cp] = piece * sigmoid(z[b][c])
__global__ void __launch_bounds__(256) gate_pieces_kernel(Pieces ps, const float4* __restrict__ z, int B, int HW, float4* __restrict__ out) {
    const int C4 = ps.n * ps.Cp4, Cp4 = ps.Cp4;
    const long long total = (long long)B * HW * C4;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long row = i / C4;
        const int c = (int)(i - row * C4);
        const int b = (int)(row / HW);
        const int k = c / Cp4, w = c - k * Cp4;
        const float4 v = ps.p[k][row * Cp4 + w], q = z[(long long)b * C4 + c];
        out[i] = make_float4(v.x * sigm(q.x), v.y * sigm(q.y), v.z * sigm(q.z), v.w * sigm(q.w));
    }
}
// dz[b][c] = sigmoid'(z) * sum_hw dout * piece   (gate_bwd_kernel's reduction; the data gradient waits for dpool: gate_bwd_pieces_kernel)
__global__ void __launch_bounds__(256) gate_bwd_reduce_pieces_kernel(Pieces ps, const float4* __restrict__ z, const float4* __restrict__ dout, int HW,
                                                                     float4* __restrict__ dz) {
    __shared__ float4 s_part[16][kPoolCols];
    const int C4 = ps.n * ps.Cp4, Cp4 = ps.Cp4;
    const int b = blockIdx.y, q = threadIdx.x & (kPoolCols - 1), r = threadIdx.x / kPoolCols;
    const int c4 = blockIdx.x * kPoolCols + q;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c4 < C4) {
        const int k = c4 / Cp4, w = c4 - k * Cp4;
        const float4* xb = ps.p[k] + (long long)b * HW * Cp4 + w;
        const float4* gb = dout + (long long)b * HW * C4 + c4;
        int p = r;
        for (; p + 16 < HW; p += 32) {
            const float4 g0 = gb[(long long)p * C4], v0 = xb[(long long)p * Cp4], g1 = gb[(long long)(p + 16) * C4], v1 = xb[(long long)(p + 16) * Cp4];
            a.x += g0.x * v0.x + g1.x * v1.x; a.y += g0.y * v0.y + g1.y * v1.y; a.z += g0.z * v0.z + g1.z * v1.z; a.w += g0.w * v0.w + g1.w * v1.w;
        }
        for (; p < HW; p += 16) {
            const float4 g = gb[(long long)p * C4], v = xb[(long long)p * Cp4];
            a.x += g.x * v.x; a.y += g.y * v.y; a.z += g.z * v.z; a.w += g.w * v.w;
        }
    }
    s_part[r][q] = a;
    __syncthreads();
    if (r == 0 && c4 < C4) {
        float4 t = s_part[0][q];
        for (int w = 1; w < 16; ++w) { const float4 u = s_part[w][q]; t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; }
        const float4 zq = z[(long long)b * C4 + c4];
        const float4 sg = make_float4(sigm(zq.x), sigm(zq.y), sigm(zq.z), sigm(zq.w));
        dz[(long long)b * C4 + c4] = make_float4(t.x * sg.x * (1.f - sg.x), t.y * sg.y * (1.f - sg.y), t.z * sg.z * (1.f - sg.z), t.w * sg.w * (1.f - sg.w));
    }
}
// dpiece_k [B][HW][Cp] = dout[.., k * Cp + c] * sigmoid(z) + dpool[b][k * Cp + c] / HW   (the gate's and the pool's gradient in one pass)
__global__ void __launch_bounds__(256) gate_bwd_pieces_kernel(PiecesOut ds, const float4* __restrict__ z, const float4* __restrict__ dout,
                                                              const float4* __restrict__ dpool, int B, int HW) {
    const int C4 = ds.n * ds.Cp4, Cp4 = ds.Cp4;
    const long long total = (long long)B * HW * C4;
    const float inv = 1.0f / (float)HW;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long row = i / C4;
        const int c = (int)(i - row * C4);
        const int b = (int)(row / HW);
        const int k = c / Cp4, w = c - k * Cp4;
        const float4 g = dout[i], q = z[(long long)b * C4 + c], dp = dpool[(long long)b * C4 + c];
        ds.p[k][row * Cp4 + w] = make_float4(g.x * sigm(q.x) + dp.x * inv, g.y * sigm(q.y) + dp.y * inv, g.z * sigm(q.z) + dp.z * inv, g.w * sigm(q.w) + dp.w * inv);
    }
}
}  // namespace ssdk

static int check_pieces(const char* fn, const float* const* pieces, int n_pieces, int batch, int hw, int piece_channels) {
    SSDK_REQUIRE(pieces && n_pieces > 0 && n_pieces <= ssdk::kMaxPieces, SSDK_E_INVALID, "%s: n_pieces=%d (1..%d)", fn, n_pieces, ssdk::kMaxPieces);
    SSDK_REQUIRE(batch > 0 && batch <= 65535 && hw > 0 && piece_channels > 0 && piece_channels % 4 == 0, SSDK_E_INVALID,
                 "%s: batch=%d hw=%d piece_channels=%d (%% 4 == 0)", fn, batch, hw, piece_channels);
    for (int k = 0; k < n_pieces; ++k)
        SSDK_REQUIRE(pieces[k] && ((uintptr_t)pieces[k] & 15) == 0, SSDK_E_INVALID, "%s: piece %d is null or not 16-byte aligned", fn, k);
    return SSDK_OK;
}
extern "C" int ssdk_sfam_pool_fwd(const float* const* pieces, int n_pieces, int batch, int hw, int piece_channels, float* pooled, void* stream) {
    int rc = check_pieces("ssdk_sfam_pool_fwd", pieces, n_pieces, batch, hw, piece_channels);
    if (rc) return rc;
    SSDK_REQUIRE(pooled, SSDK_E_INVALID, "ssdk_sfam_pool_fwd: null output");
    ssdk::Pieces ps{};
    ps.n = n_pieces; ps.Cp4 = piece_channels / 4;
    for (int k = 0; k < n_pieces; ++k) ps.p[k] = (const float4*)pieces[k];
    hipLaunchKernelGGL(pool_pieces_kernel, dim3(cdiv(n_pieces * piece_channels / 4, ssdk::kPoolCols), batch), dim3(256), 0, (hipStream_t)stream, ps, hw, (float4*)pooled);
    SSDK_CHECK_LAUNCH("pool_pieces_kernel");
    return SSDK_OK;
}
extern "C" int ssdk_sfam_gate_fwd(const float* const* pieces, int n_pieces, int batch, int hw, int piece_channels, const float* z, float* out,
                                  void* stream) {
    int rc = check_pieces("ssdk_sfam_gate_fwd", pieces, n_pieces, batch, hw, piece_channels);
    if (rc) return rc;
    SSDK_REQUIRE(z && out, SSDK_E_INVALID, "ssdk_sfam_gate_fwd: null pointer");
    ssdk::Pieces ps{};
    ps.n = n_pieces; ps.Cp4 = piece_channels / 4;
    for (int k = 0; k < n_pieces; ++k) ps.p[k] = (const float4*)pieces[k];
    hipLaunchKernelGGL(gate_pieces_kernel, dim3(stream_blocks((long long)batch * hw * n_pieces * piece_channels / 4, 256)), dim3(256), 0, (hipStream_t)stream, ps,
                       (const float4*)z, batch, hw, (float4*)out);
    SSDK_CHECK_LAUNCH("gate_pieces_kernel");
    return SSDK_OK;
}
extern "C" int ssdk_sfam_gate_bwd_reduce(const float* const* pieces, int n_pieces, int batch, int hw, int piece_channels, const float* z,
                                         const float* dout, float* dz, void* stream) {
    int rc = check_pieces("ssdk_sfam_gate_bwd_reduce", pieces, n_pieces, batch, hw, piece_channels);
    if (rc) return rc;
    SSDK_REQUIRE(z && dout && dz, SSDK_E_INVALID, "ssdk_sfam_gate_bwd_reduce: null pointer");
    ssdk::Pieces ps{};
    ps.n = n_pieces; ps.Cp4 = piece_channels / 4;
    for (int k = 0; k < n_pieces; ++k) ps.p[k] = (const float4*)pieces[k];
    hipLaunchKernelGGL(gate_bwd_reduce_pieces_kernel, dim3(cdiv(n_pieces * piece_channels / 4, ssdk::kPoolCols), batch), dim3(256), 0, (hipStream_t)stream, ps,
                       (const float4*)z, (const float4*)dout, hw, (float4*)dz);
    SSDK_CHECK_LAUNCH("gate_bwd_reduce_pieces_kernel");
    return SSDK_OK;
}
extern "C" int ssdk_sfam_gate_bwd_apply(float* const* dpieces, int n_pieces, int batch, int hw, int piece_channels, const float* z, const float* dout,
                                        const float* dpool, void* stream) {
    int rc = check_pieces("ssdk_sfam_gate_bwd_apply", (const float* const*)dpieces, n_pieces, batch, hw, piece_channels);
    if (rc) return rc;
    SSDK_REQUIRE(z && dout && dpool, SSDK_E_INVALID, "ssdk_sfam_gate_bwd_apply: null pointer");
    ssdk::PiecesOut ds{};
    ds.n = n_pieces; ds.Cp4 = piece_channels / 4;
    for (int k = 0; k < n_pieces; ++k) ds.p[k] = (float4*)dpieces[k];
    hipLaunchKernelGGL(gate_bwd_pieces_kernel, dim3(stream_blocks((long long)batch * hw * n_pieces * piece_channels / 4, 256)), dim3(256), 0, (hipStream_t)stream, ds,
                       (const float4*)z, (const float4*)dout, (const float4*)dpool, batch, hw);
    SSDK_CHECK_LAUNCH("gate_bwd_pieces_kernel");
    return SSDK_OK;
}

// ---- depthwise convolution (bf/modules/conv.py:39-85) ----------------------------------------------------------------------
// HBM-bound stencils on NHWC maps: a thread owns 4 consecutive channels of one output pixel (16-byte loads / stores, the k*k
// weights of its channels in registers).  Backward-weights: every thread accumulates its pixels' products for its 4 channels,
// a workgroup reduces over its pixels through LDS and adds one partial per (channel, tap) with an atomic.
namespace ssdk {

constexpr int kDwMaxTaps = 25;   // up to 5 x 5

__global__ void __launch_bounds__(256) dw_fwd_kernel(const float4* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias, int B, int Hin,
                                                     int Win, int C4, int ks, int stride, int pad, int Hout, int Wout, float4* __restrict__ y) {
    const long long total = (long long)B * Hout * Wout * C4;
    const int taps = ks * ks;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        long long p = i / C4;
        const int xo = (int)(p % Wout); p /= Wout;
        const int yo = (int)(p % Hout);
        const int b = (int)(p / Hout);
        float4 acc = bias ? *reinterpret_cast<const float4*>(bias + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float* wc = w + (long long)c4 * 4 * taps;
        for (int ky = 0; ky < ks; ++ky) {
            const int iy = yo * stride - pad + ky;
            if (iy < 0 || iy >= Hin) continue;
            for (int kx = 0; kx < ks; ++kx) {
                const int ix = xo * stride - pad + kx;
                if (ix < 0 || ix >= Win) continue;
                const float4 v = x[((long long)(b * Hin + iy) * Win + ix) * C4 + c4];
                const int t = ky * ks + kx;
                acc.x = fmaf(wc[t], v.x, acc.x);
                acc.y = fmaf(wc[taps + t], v.y, acc.y);
                acc.z = fmaf(wc[2 * taps + t], v.z, acc.z);
                acc.w = fmaf(wc[3 * taps + t], v.w, acc.w);
            }
        }
        y[i] = acc;
    }
}

// dx[b,iy,ix,c] = sum over taps with (iy + pad - ky) % stride == 0 ... of w[c][tap] * dy[b,(iy+pad-ky)/stride,(ix+pad-kx)/stride,c]
__global__ void __launch_bounds__(256) dw_dgrad_kernel(const float4* __restrict__ dy, const float* __restrict__ w, int B, int Hin, int Win, int C4, int ks,
                                                       int stride, int pad, int Hout, int Wout, float4* __restrict__ dx, int accumulate) {
    const long long total = (long long)B * Hin * Win * C4;
    const int taps = ks * ks;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        long long p = i / C4;
        const int ix = (int)(p % Win); p /= Win;
        const int iy = (int)(p % Hin);
        const int b = (int)(p / Hin);
        float4 acc = accumulate ? dx[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        const float* wc = w + (long long)c4 * 4 * taps;
        for (int ky = 0; ky < ks; ++ky) {
            const int ty = iy + pad - ky;
            if (ty < 0 || ty % stride || ty / stride >= Hout) continue;
            for (int kx = 0; kx < ks; ++kx) {
                const int tx = ix + pad - kx;
                if (tx < 0 || tx % stride || tx / stride >= Wout) continue;
                const float4 v = dy[((long long)(b * Hout + ty / stride) * Wout + tx / stride) * C4 + c4];
                const int t = ky * ks + kx;
                acc.x = fmaf(wc[t], v.x, acc.x);
                acc.y = fmaf(wc[taps + t], v.y, acc.y);
                acc.z = fmaf(wc[2 * taps + t], v.z, acc.z);
                acc.w = fmaf(wc[3 * taps + t], v.w, acc.w);
            }
        }
        dx[i] = acc;
    }
}

// grid (C4 blocks of 64 channel-quads, pixel chunks): thread (cq, ps) = channel quad cq of the block, pixel phase ps of 4
__global__ void __launch_bounds__(256) dw_wgrad_kernel(const float4* __restrict__ x, const float4* __restrict__ dy, int B, int Hin, int Win, int C4, int ks,
                                                       int stride, int pad, int Hout, int Wout, int pixels_per_block, float* __restrict__ dw,
                                                       float* __restrict__ db) {
    __shared__ float s_red[4][64][4];
    const int taps = ks * ks;
    const int cq = threadIdx.x & 63, ps = threadIdx.x >> 6;
    const int c4 = blockIdx.x * 64 + cq;
    const long long M = (long long)B * Hout * Wout;
    const long long p0 = (long long)blockIdx.y * pixels_per_block, p1 = p0 + pixels_per_block < M ? p0 + pixels_per_block : M;
    float4 acc[kDwMaxTaps];
    float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int t = 0; t < kDwMaxTaps; ++t) acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c4 < C4) {
        for (long long p = p0 + ps; p < p1; p += 4) {
            const int xo = (int)(p % Wout);
            const int yo = (int)((p / Wout) % Hout);
            const int b = (int)(p / ((long long)Wout * Hout));
            const float4 g = dy[p * C4 + c4];
            bsum.x += g.x; bsum.y += g.y; bsum.z += g.z; bsum.w += g.w;
#pragma unroll
            for (int t = 0; t < kDwMaxTaps; ++t) {
                if (t >= taps) break;
                const int iy = yo * stride - pad + t / ks, ix = xo * stride - pad + t % ks;
                if (iy < 0 || iy >= Hin || ix < 0 || ix >= Win) continue;
                const float4 v = x[((long long)(b * Hin + iy) * Win + ix) * C4 + c4];
                acc[t].x = fmaf(g.x, v.x, acc[t].x);
                acc[t].y = fmaf(g.y, v.y, acc[t].y);
                acc[t].z = fmaf(g.z, v.z, acc[t].z);
                acc[t].w = fmaf(g.w, v.w, acc[t].w);
            }
        }
    }
    // reduce the 4 pixel phases through LDS, one tap at a time; phase 0 adds the block's partial
    for (int t = -1; t < taps; ++t) {
        const float4 v = t < 0 ? bsum : acc[t < 0 ? 0 : t];
        __syncthreads();
        s_red[ps][cq][0] = v.x; s_red[ps][cq][1] = v.y; s_red[ps][cq][2] = v.z; s_red[ps][cq][3] = v.w;
        __syncthreads();
        if (ps == 0 && c4 < C4) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float sum = s_red[0][cq][e] + s_red[1][cq][e] + s_red[2][cq][e] + s_red[3][cq][e];
                if (t < 0) { if (db) atomicAdd(db + c4 * 4 + e, sum); }
                else atomicAdd(dw + (long long)(c4 * 4 + e) * taps + t, sum);
            }
        }
    }
}

__global__ void zero_small_kernel(float* a, long long na, float* b, long long nb) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < na + nb; i += (long long)gridDim.x * blockDim.x) {
        if (i < na) a[i] = 0.0f;
        else b[i - na] = 0.0f;
    }
}

}  // namespace ssdk

static int dw_check(const char* fn, int batch, int hin, int win, int channels, int ksize, int stride, int pad) {
    SSDK_REQUIRE(batch > 0 && hin > 0 && win > 0 && channels > 0 && channels % 4 == 0 && ksize >= 1 && ksize * ksize <= ssdk::kDwMaxTaps && stride >= 1 && pad >= 0 &&
                     hin + 2 * pad >= ksize && win + 2 * pad >= ksize,
                 SSDK_E_INVALID, "%s: batch=%d H=%d W=%d C=%d (%%4) k=%d stride=%d pad=%d", fn, batch, hin, win, channels, ksize, stride, pad);
    return SSDK_OK;
}

extern "C" int ssdk_depthwise_conv2d_fwd(const float* x, const float* w, const float* bias, int batch, int hin, int win, int channels, int ksize,
                                         int stride, int pad, float* y, void* stream) {
    int rc = dw_check("ssdk_depthwise_conv2d_fwd", batch, hin, win, channels, ksize, stride, pad);
    if (rc) return rc;
    SSDK_REQUIRE(x && w && y && (((uintptr_t)x | (uintptr_t)y | (uintptr_t)bias) & 15) == 0, SSDK_E_INVALID, "ssdk_depthwise_conv2d_fwd: null or unaligned pointer");
    const int ho = (hin + 2 * pad - ksize) / stride + 1, wo = (win + 2 * pad - ksize) / stride + 1;
    const long long total = (long long)batch * ho * wo * (channels / 4);
    const unsigned blocks = (unsigned)((total + 255) / 256 < 65535 ? (total + 255) / 256 : 65535);
    hipLaunchKernelGGL(ssdk::dw_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const float4*>(x), w, bias, batch, hin, win,
                       channels / 4, ksize, stride, pad, ho, wo, reinterpret_cast<float4*>(y));
    SSDK_CHECK_LAUNCH("dw_fwd_kernel");
    return SSDK_OK;
}

extern "C" int ssdk_depthwise_conv2d_bwd(const float* x, const float* w, const float* dy, int batch, int hin, int win, int channels, int ksize,
                                         int stride, int pad, float* dx, float* dw, float* db, int accumulate, void* stream) {
    int rc = dw_check("ssdk_depthwise_conv2d_bwd", batch, hin, win, channels, ksize, stride, pad);
    if (rc) return rc;
    SSDK_REQUIRE(x && w && dy && (dx || dw) && (((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx) & 15) == 0, SSDK_E_INVALID,
                 "ssdk_depthwise_conv2d_bwd: null or unaligned pointer");
    hipStream_t s = (hipStream_t)stream;
    const int ho = (hin + 2 * pad - ksize) / stride + 1, wo = (win + 2 * pad - ksize) / stride + 1, c4 = channels / 4;
    if (dx) {
        const long long total = (long long)batch * hin * win * c4;
        const unsigned blocks = (unsigned)((total + 255) / 256 < 65535 ? (total + 255) / 256 : 65535);
        hipLaunchKernelGGL(ssdk::dw_dgrad_kernel, dim3(blocks), dim3(256), 0, s, reinterpret_cast<const float4*>(dy), w, batch, hin, win, c4, ksize, stride, pad, ho,
                           wo, reinterpret_cast<float4*>(dx), accumulate);
        SSDK_CHECK_LAUNCH("dw_dgrad_kernel");
    }
    if (dw) {
        if (!accumulate) {
            hipLaunchKernelGGL(ssdk::zero_small_kernel, dim3(8), dim3(256), 0, s, dw, (long long)channels * ksize * ksize, db, db ? (long long)channels : 0LL);
            SSDK_CHECK_LAUNCH("zero_small_kernel");
        }
        const long long M = (long long)batch * ho * wo;
        int chunks = (int)((M + 255) / 256);   // >= 256 pixels per block ...
        if (chunks > 512) chunks = 512;         // ... and at most 512 partials per (channel, tap)
        if (ssdk::deterministic()) chunks = 1;  // (one workgroup per 64 channel quads walks all pixels: its single "atomic" per element adds to zero)
        const int ppb = (int)((M + chunks - 1) / chunks);
        hipLaunchKernelGGL(ssdk::dw_wgrad_kernel, dim3((unsigned)((c4 + 63) / 64), (unsigned)chunks), dim3(256), 0, s, reinterpret_cast<const float4*>(x),
                           reinterpret_cast<const float4*>(dy), batch, hin, win, c4, ksize, stride, pad, ho, wo, ppb, dw, db);
        SSDK_CHECK_LAUNCH("dw_wgrad_kernel");
    }
    return SSDK_OK;
}
