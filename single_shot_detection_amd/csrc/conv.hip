// conv.hip -- multi-scale head convolutions as implicit GEMM on the CDNA4 matrix cores.
//
// Reference: detection/detector_builder.py:111-137 (get_heads: per level a 3x3/pad-1 score conv with nb*C outputs
// and a 3x3/pad-1 loc conv with nb*4 outputs, both with bias) applied in detection/detector.py:50-66, each followed
// by permute(0,2,3,1).contiguous().view(B,-1) and a cat over levels -- 2*L library convolutions plus 2*L
// permute copies plus 2 concatenations per forward.
//
// Here each level is ONE GEMM  C[m][n] = sum_k A[m][k] * W[n][k]  with
//     m = (image, y, x) output pixel, n = output channel of the FUSED score|loc head, k = (tap, input channel),
// computed with v_mfma_f32_32x32x2_f32 (exact fp32: parity mode -- the loss must match the reference to 1e-4) and
// an epilogue that adds the bias and stores straight into the concatenated [B, A*C] / [B, A*4] buffers at the
// level's offset (the NHWC flatten of detector.py:52-63 IS the natural output order of this GEMM, so permute,
// contiguous and cat disappear).  Activations are NHWC (channels-last) so that a K-slice of an A row is one
// contiguous 128-byte line; weights are [n][tap][cin] (= torch channels_last memory of the OIHW parameter).
//
// Tiling (wave = 64 lanes): workgroup = 4 waves = 128 output pixels x (32*tn) channels, tn <= 8 chosen per level so
// that the N tiles are balanced (N = 340 -> 6 + 5 tiles, N = 510 -> 8 + 8); wave w owns pixel rows 32w..32w+31 and
// all tn column tiles: tn accumulators of 16 VGPRs.  K is walked in slices of 32: the next slice is prefetched
// global -> registers while the current one is multiplied out of LDS (rows padded to 36 floats: conflict-free
// ds_read_b128; one b128 read feeds four MFMAs of a tile).  The backward-data pass is the same kernel with the
// taps mirrored, dY (= the dscores|dlocs slices) as the A operand and the per-tap transposed weights as W.
// The backward-weights pass (igemm_wgrad_kernel) contracts over pixels instead: both operands are read in their
// natural row-major form ([pixel][channel]), K = pixels is split across workgroups and partial tiles are
// accumulated with fp32 atomics shaped as two 128-byte segments per wave instruction.
#include "common.h"

namespace ssdk {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kBM = 128;       // output pixels per workgroup
constexpr int kBK = 32;        // K slice
constexpr int kLdsStride = 36; // floats per LDS row (32 + 4 pad)
constexpr int kMaxTN = 8;      // 32-wide column tiles per workgroup
constexpr int kConvThreads = 256;

struct RowSeg {        // a [pixel][channel] operand made of up to two channel segments (dscores | dlocs)
    const float* p0;   // segment 0 base (level offset already applied)
    const float* p1;   // segment 1 base or null
    long long b0, b1;  // per-image stride in floats
    int c0, c1;        // channels per segment
    int s0, s1;        // per-pixel stride in floats
};

struct GemmFwd {
    RowSeg a;            // A operand rows: input pixels (Hin x Win per image)
    int B, Hout, Wout, Hin, Win;
    int ksize, stride, pad, mirror;  // mirror != 0: backward-data (taps flipped, pad' = ksize-1-pad)
    const float* w0;     // W rows for n <  n0: [n0][taps*Cc]
    const float* w1;     // W rows for n >= n0: [n1][taps*Cc]
    const float* bias0;
    const float* bias1;
    int n0, n1;
    float* o0;           // output segment 0: element (image b, pixel p, channel n) at o0 + b*ob0 + p*os0 + n
    float* o1;
    long long ob0, ob1;
    int os0, os1;
    int tiles_n;         // 32-wide column tiles in total
    int n_blocks;        // workgroups along N
    int m_tiles;
    int relu;
};

template <int VEC>
struct VecT;
template <>
struct VecT<4> { typedef float4 type; };
template <>
struct VecT<2> { typedef float2 type; };
template <>
struct VecT<1> { typedef float type; };

template <int VEC>
__device__ __forceinline__ typename VecT<VEC>::type vzero();
template <>
__device__ __forceinline__ float4 vzero<4>() { return make_float4(0.f, 0.f, 0.f, 0.f); }
template <>
__device__ __forceinline__ float2 vzero<2>() { return make_float2(0.f, 0.f); }
template <>
__device__ __forceinline__ float vzero<1>() { return 0.f; }

// pointer to channel c of pixel row `pix` of image b, or null when c is past the last channel
__device__ __forceinline__ const float* seg_ptr(const RowSeg& r, int b, long long pix, int c) {
    if (c < r.c0) return r.p0 + (long long)b * r.b0 + pix * r.s0 + c;
    c -= r.c0;
    if (c < r.c1) return r.p1 + (long long)b * r.b1 + pix * r.s1 + c;
    return nullptr;
}

// ---- forward / backward-data ------------------------------------------------------------------------------------
// MIRROR = false: forward convolution; MIRROR = true: backward-data (separate instantiations so that profiles list the
// forward GEMMs and the dgrad GEMMs as different kernels).
template <int VEC, bool MIRROR>
__global__ void __launch_bounds__(kConvThreads) igemm_fwd_kernel(GemmFwd g) {
    typedef typename VecT<VEC>::type vec_t;
    constexpr int kVecPerRow = kBK / VEC;                 // vector loads per 32-float row slice
    constexpr int kRowsPerPass = kConvThreads / kVecPerRow;
    constexpr int kAPasses = kBM / kRowsPerPass;
    constexpr int kBPassesMax = kMaxTN * 32 / kRowsPerPass;

    __shared__ __attribute__((aligned(16))) float s_a[kBM * kLdsStride];
    __shared__ __attribute__((aligned(16))) float s_b[kMaxTN * 32 * kLdsStride];

    // workgroups that share an M tile get the same blockIdx % 8 (same XCD under round-robin placement: they re-read
    // the same activation rows from one L2).  Speed only; any placement is correct.
    const int id = blockIdx.x;
    const int per_chunk = 8 * g.n_blocks;
    const int chunk = id / per_chunk, within = id % per_chunk;
    const int m_tile = chunk * 8 + (within & 7);
    const int n_block = within >> 3;
    if (m_tile >= g.m_tiles) return;

    const int Cc = g.a.c0 + g.a.c1;
    const int taps = g.ksize * g.ksize;
    const int chunks = (Cc + kBK - 1) / kBK;
    const int n_slices = taps * chunks;
    const long long K = (long long)taps * Cc;
    const int N = g.n0 + g.n1;
    const int M = g.B * g.Hout * g.Wout;

    // balanced split of the column tiles over the n blocks
    const int base_t = g.tiles_n / g.n_blocks, rem_t = g.tiles_n % g.n_blocks;
    const int tn = base_t + (n_block < rem_t ? 1 : 0);
    const int tile0 = n_block * base_t + min(n_block, rem_t);
    const int n_begin = tile0 * 32;

    const int tid = threadIdx.x;
    const int lrow = tid / kVecPerRow, lcol = (tid % kVecPerRow) * VEC;

    // per-thread A rows: pixel coordinates of the rows this thread stages
    int a_b[kAPasses], a_y[kAPasses], a_x[kAPasses];
#pragma unroll
    for (int p = 0; p < kAPasses; ++p) {
        const int m = m_tile * kBM + lrow + p * kRowsPerPass;
        if (m < M) {
            const int hw = g.Hout * g.Wout;
            a_b[p] = m / hw;
            const int r = m % hw;
            a_y[p] = r / g.Wout;
            a_x[p] = r % g.Wout;
        } else {
            a_b[p] = -1; a_y[p] = 0; a_x[p] = 0;
        }
    }

    vec_t ra[kAPasses], rb[kBPassesMax];
    const int b_passes = (tn * 32 + kRowsPerPass - 1) / kRowsPerPass;

    auto load_slice = [&](int slice) {
        const int tap = slice / chunks, c = (slice % chunks) * kBK + lcol;
        const int ky = tap / g.ksize, kx = tap % g.ksize;
#pragma unroll
        for (int p = 0; p < kAPasses; ++p) {
            vec_t v = vzero<VEC>();
            if (a_b[p] >= 0) {
                int iy, ix;
                bool ok = true;
                if (!MIRROR) {  // input pixel = out*stride - pad + k
                    iy = a_y[p] * g.stride - g.pad + ky;
                    ix = a_x[p] * g.stride - g.pad + kx;
                } else {          // backward-data: the rows are OUTPUT-gradient pixels, (y + pad - k) / stride
                    const int ty = a_y[p] + g.pad - ky, tx = a_x[p] + g.pad - kx;
                    ok = (ty % g.stride == 0) && (tx % g.stride == 0) && ty >= 0 && tx >= 0;
                    iy = ty / g.stride;
                    ix = tx / g.stride;
                }
                if (ok && iy >= 0 && iy < g.Hin && ix >= 0 && ix < g.Win) {
                    const float* src = seg_ptr(g.a, a_b[p], (long long)iy * g.Win + ix, c);
                    if (src) v = *reinterpret_cast<const vec_t*>(src);
                }
            }
            ra[p] = v;
        }
#pragma unroll
        for (int p = 0; p < kBPassesMax; ++p) {
            vec_t v = vzero<VEC>();
            if (p < b_passes) {
                const int n = n_begin + lrow + p * kRowsPerPass;
                if (n < N && c < Cc && lrow + p * kRowsPerPass < tn * 32) {
                    const float* wrow = n < g.n0 ? g.w0 + (long long)n * K : g.w1 + (long long)(n - g.n0) * K;
                    v = *reinterpret_cast<const vec_t*>(wrow + (long long)tap * Cc + c);
                }
            }
            rb[p] = v;
        }
    };
    auto store_slice = [&]() {
#pragma unroll
        for (int p = 0; p < kAPasses; ++p)
            *reinterpret_cast<vec_t*>(&s_a[(lrow + p * kRowsPerPass) * kLdsStride + lcol]) = ra[p];
#pragma unroll
        for (int p = 0; p < kBPassesMax; ++p)
            if (p < b_passes) *reinterpret_cast<vec_t*>(&s_b[(lrow + p * kRowsPerPass) * kLdsStride + lcol]) = rb[p];
    };

    f32x16 acc[kMaxTN];
#pragma unroll
    for (int j = 0; j < kMaxTN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.0f;

    const int lane = tid & 63, wave = tid >> 6;
    const int r32 = lane & 31, h = lane >> 5;
    const float* a_rd = s_a + (wave * 32 + r32) * kLdsStride + 4 * h;
    const float* b_rd = s_b + r32 * kLdsStride + 4 * h;

    load_slice(0);
    store_slice();
    __syncthreads();
    for (int slice = 0; slice < n_slices; ++slice) {
        if (slice + 1 < n_slices) load_slice(slice + 1);  // in flight while the MFMAs below run
#pragma unroll
        for (int gk = 0; gk < kBK / 8; ++gk) {
            const f32x4 av = *reinterpret_cast<const f32x4*>(a_rd + gk * 8);
            f32x4 bv[kMaxTN];
#pragma unroll
            for (int j = 0; j < kMaxTN; ++j)
                if (j < tn) bv[j] = *reinterpret_cast<const f32x4*>(b_rd + j * 32 * kLdsStride + gk * 8);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int j = 0; j < kMaxTN; ++j)
                    if (j < tn) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk], bv[j][kk], acc[j], 0, 0, 0);
        }
        __syncthreads();
        if (slice + 1 < n_slices) {
            store_slice();
            __syncthreads();
        }
    }

    // epilogue: C/D map of the 32x32 MFMA: column = lane & 31, row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)
    const int hw = g.Hout * g.Wout;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int m = m_tile * kBM + wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (m >= M) continue;
        const int b = m / hw, pix = m % hw;
#pragma unroll
        for (int j = 0; j < kMaxTN; ++j) {
            if (j >= tn) continue;
            const int n = n_begin + j * 32 + r32;
            if (n >= N) continue;
            float v = acc[j][e];
            if (n < g.n0) {
                if (g.bias0) v += g.bias0[n];
                if (g.relu) v = fmaxf(v, 0.0f);
                g.o0[(long long)b * g.ob0 + (long long)pix * g.os0 + n] = v;
            } else {
                const int n1 = n - g.n0;
                if (g.bias1) v += g.bias1[n1];
                if (g.relu) v = fmaxf(v, 0.0f);
                g.o1[(long long)b * g.ob1 + (long long)pix * g.os1 + n1] = v;
            }
        }
    }
}

// ---- backward-weights ------------------------------------------------------------------------------------------
struct GemmWgrad {
    RowSeg dy;           // output-gradient rows [pixel of Hout x Wout][n], n = dy.c0 + dy.c1 channels
    RowSeg x;            // input rows [pixel of Hin x Win][c]
    int B, Hout, Wout, Hin, Win, ksize, stride, pad;
    float* dw0;          // [n0][taps*Cc] (+=)
    float* dw1;          // [n1][taps*Cc] (+=)
    int n0, n1;
    int k_splits;        // workgroups along the pixel (K) dimension
    int n_tiles;         // 128-row tiles over N
    int c_tiles32;       // 32-wide tiles over Cc
    int c_blocks;        // workgroups along Cc
};

// LDS: dY slice [32 pixels][128 n] and X slice [32 pixels][32*tn c]; MFMA A operand = dY^T, B operand = X.
template <int VEC_DY>
__global__ void __launch_bounds__(kConvThreads) igemm_wgrad_kernel(GemmWgrad g) {
    typedef typename VecT<VEC_DY>::type dvec_t;
    constexpr int kDyVecPerRow = 128 / VEC_DY;
    constexpr int kDyRowsPerPass = kConvThreads / kDyVecPerRow > 0 ? kConvThreads / kDyVecPerRow : 1;
    constexpr int kDyPasses = 32 / kDyRowsPerPass;
    __shared__ __attribute__((aligned(16))) float s_dy[32 * 128];
    __shared__ __attribute__((aligned(16))) float s_x[32 * kMaxTN * 32];

    const int Cc = g.x.c0 + g.x.c1;
    const int N = g.n0 + g.n1;
    const int taps = g.ksize * g.ksize;
    int id = blockIdx.x;
    const int ksp = id % g.k_splits; id /= g.k_splits;
    const int cb = id % g.c_blocks; id /= g.c_blocks;
    const int nt = id % g.n_tiles; id /= g.n_tiles;
    const int tap = id;
    const int ky = tap / g.ksize, kx = tap % g.ksize;

    const int base_t = g.c_tiles32 / g.c_blocks, rem_t = g.c_tiles32 % g.c_blocks;
    const int tn = base_t + (cb < rem_t ? 1 : 0);
    const int c_begin = (cb * base_t + min(cb, rem_t)) * 32;
    const int n_begin = nt * 128;

    const int hw = g.Hout * g.Wout;
    const int M = g.B * hw;
    const int slices_total = (M + 31) / 32;
    const int per = (slices_total + g.k_splits - 1) / g.k_splits;
    const int s_begin = ksp * per, s_end = min(slices_total, s_begin + per);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r32 = lane & 31, h = lane >> 5;

    f32x16 acc[kMaxTN];
#pragma unroll
    for (int j = 0; j < kMaxTN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.0f;

    // staging maps: dY slice = 32 rows x 128 floats; X slice = 32 rows x (32*tn) floats as float4
    const int dy_row = tid / kDyVecPerRow, dy_col = (tid % kDyVecPerRow) * VEC_DY;
    const int x_vec_per_row = tn * 8;  // float4 per row

    for (int s = s_begin; s < s_end; ++s) {
        __syncthreads();
        // dY slice
#pragma unroll
        for (int p = 0; p < kDyPasses; ++p) {
            const int row = dy_row + p * kDyRowsPerPass;
            const int m = s * 32 + row;
            dvec_t v = vzero<VEC_DY>();
            if (m < M) {
                const int b = m / hw, pix = m % hw;
                const float* src = seg_ptr(g.dy, b, pix, n_begin + dy_col);
                if (src) v = *reinterpret_cast<const dvec_t*>(src);
            }
            *reinterpret_cast<dvec_t*>(&s_dy[row * 128 + dy_col]) = v;
        }
        // X slice (shifted by the tap)
        for (int t = tid; t < 32 * x_vec_per_row; t += kConvThreads) {
            const int row = t / x_vec_per_row, c = c_begin + (t % x_vec_per_row) * 4;
            const int m = s * 32 + row;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m < M && c < Cc) {
                const int b = m / hw, pix = m % hw;
                const int iy = (pix / g.Wout) * g.stride - g.pad + ky, ix = (pix % g.Wout) * g.stride - g.pad + kx;
                if (iy >= 0 && iy < g.Hin && ix >= 0 && ix < g.Win) {
                    const float* src = seg_ptr(g.x, b, (long long)iy * g.Win + ix, c);
                    if (src) v = *reinterpret_cast<const float4*>(src);
                }
            }
            *reinterpret_cast<float4*>(&s_x[row * (kMaxTN * 32) + (t % x_vec_per_row) * 4]) = v;
        }
        __syncthreads();
#pragma unroll 4
        for (int k2 = 0; k2 < 32; k2 += 2) {
            const float av = s_dy[(k2 + h) * 128 + wave * 32 + r32];
#pragma unroll
            for (int j = 0; j < kMaxTN; ++j)
                if (j < tn) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, s_x[(k2 + h) * (kMaxTN * 32) + j * 32 + r32], acc[j], 0, 0, 0);
        }
    }
    const long long K = (long long)taps * Cc;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int n = n_begin + wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (n >= N) continue;
        float* row = n < g.n0 ? g.dw0 + (long long)n * K : g.dw1 + (long long)(n - g.n0) * K;
#pragma unroll
        for (int j = 0; j < kMaxTN; ++j) {
            if (j >= tn) continue;
            const int c = c_begin + j * 32 + r32;
            if (c < Cc) atomicAdd(row + (long long)tap * Cc + c, acc[j][e]);
        }
    }
}

// dbias[n] += sum over pixels of dY[pixel][n]
__global__ void __launch_bounds__(256) colsum_kernel(RowSeg dy, int B, int HW, float* __restrict__ db0, float* __restrict__ db1,
                                                     int rows_per_block) {
    const int N = dy.c0 + dy.c1;
    const long long M = (long long)B * HW;
    const long long m0 = (long long)blockIdx.x * rows_per_block, m1 = min(M, m0 + rows_per_block);
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
        float s = 0.0f;
        for (long long m = m0; m < m1; ++m) s += *seg_ptr(dy, (int)(m / HW), m % HW, n);
        if (n < dy.c0) atomicAdd(db0 + n, s); else atomicAdd(db1 + (n - dy.c0), s);
    }
}

// Wd[c][tap][n] = W[n][tap][c]  (per-tap transpose: the backward-data GEMM wants K = (tap, n) contiguous per c)
__global__ void __launch_bounds__(256) transpose_taps_kernel(const float* __restrict__ w0, const float* __restrict__ w1, int n0, int n1,
                                                             int taps, int Cc, float* __restrict__ wd) {
    __shared__ float tile[32][33];
    const int N = n0 + n1;
    const int tap = blockIdx.z;
    const int nb = blockIdx.x * 32, cb = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const int n = nb + r, c = cb + tx;
        float v = 0.0f;
        if (n < N && c < Cc) v = n < n0 ? w0[((long long)n * taps + tap) * Cc + c] : w1[((long long)(n - n0) * taps + tap) * Cc + c];
        tile[r][tx] = v;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int c = cb + r, n = nb + tx;
        if (n < N && c < Cc) wd[((long long)c * taps + tap) * N + n] = tile[tx][r];
    }
}

static int pick_vec(const RowSeg& r, int also_mult) {
    auto ok = [&](int v) {
        if (r.c0 % v || (r.c1 && r.c1 % v) || r.s0 % v || (r.p1 && r.s1 % v) || r.b0 % v || (r.p1 && r.b1 % v)) return false;
        if (((uintptr_t)r.p0 & (v * 4 - 1)) || (r.p1 && ((uintptr_t)r.p1 & (v * 4 - 1)))) return false;
        if (also_mult % v) return false;
        return true;
    };
    return ok(4) ? 4 : (ok(2) ? 2 : 1);
}

}  // namespace ssdk

using namespace ssdk;

static RowSeg make_seg(const float* p0, long long b0, int c0, int s0, const float* p1, long long b1, int c1, int s1) {
    RowSeg r;
    r.p0 = p0; r.b0 = b0; r.c0 = c0; r.s0 = s0;
    r.p1 = p1; r.b1 = b1; r.c1 = p1 ? c1 : 0; r.s1 = s1;
    return r;
}

static int launch_fwd(GemmFwd& g, int vec_hint_k, hipStream_t s) {
    const int N = g.n0 + g.n1;
    const int M = g.B * g.Hout * g.Wout;
    g.tiles_n = cdiv(N, 32);
    g.n_blocks = cdiv(g.tiles_n, kMaxTN);
    g.m_tiles = cdiv(M, kBM);
    const int chunks8 = cdiv(g.m_tiles, 8);
    const int grid = chunks8 * 8 * g.n_blocks;
    const int Cc = g.a.c0 + g.a.c1;
    int vec = pick_vec(g.a, Cc);
    // weights rows: [n][taps*Cc] -> need Cc % vec == 0 and aligned bases
    while (vec > 1 && ((((uintptr_t)g.w0) & (vec * 4 - 1)) || (g.w1 && (((uintptr_t)g.w1) & (vec * 4 - 1))) || Cc % vec)) vec >>= 1;
    (void)vec_hint_k;
    if (g.mirror) {
        if (vec == 4) hipLaunchKernelGGL((igemm_fwd_kernel<4, true>), dim3(grid), dim3(kConvThreads), 0, s, g);
        else if (vec == 2) hipLaunchKernelGGL((igemm_fwd_kernel<2, true>), dim3(grid), dim3(kConvThreads), 0, s, g);
        else hipLaunchKernelGGL((igemm_fwd_kernel<1, true>), dim3(grid), dim3(kConvThreads), 0, s, g);
    } else {
        if (vec == 4) hipLaunchKernelGGL((igemm_fwd_kernel<4, false>), dim3(grid), dim3(kConvThreads), 0, s, g);
        else if (vec == 2) hipLaunchKernelGGL((igemm_fwd_kernel<2, false>), dim3(grid), dim3(kConvThreads), 0, s, g);
        else hipLaunchKernelGGL((igemm_fwd_kernel<1, false>), dim3(grid), dim3(kConvThreads), 0, s, g);
    }
    SSDK_CHECK_LAUNCH("igemm_fwd_kernel");
    return SSDK_OK;
}

static int check_conv_geom(const char* fn, int batch, int h, int w, int cin, int n_score, int n_loc) {
    SSDK_REQUIRE(batch > 0 && h > 0 && w > 0 && cin > 0 && n_score > 0 && n_loc >= 0, SSDK_E_INVALID,
                 "%s: batch=%d H=%d W=%d Cin=%d n_score=%d n_loc=%d", fn, batch, h, w, cin, n_score, n_loc);
    SSDK_REQUIRE((long long)batch * h * w < (1LL << 31) - kBM, SSDK_E_INVALID, "%s: too many pixels", fn);
    return SSDK_OK;
}

extern "C" int ssdk_head_conv_fwd(const float* x, int batch, int h, int w, int cin, const float* w_score, const float* b_score,
                                  int n_score, const float* w_loc, const float* b_loc, int n_loc, float* scores,
                                  long long scores_batch_stride, long long scores_offset, float* locs,
                                  long long locs_batch_stride, long long locs_offset, void* stream) {
    int rc = check_conv_geom("ssdk_head_conv_fwd", batch, h, w, cin, n_score, n_loc);
    if (rc) return rc;
    SSDK_REQUIRE(x && w_score && scores && (n_loc == 0 || (w_loc && locs)), SSDK_E_INVALID, "ssdk_head_conv_fwd: null pointer");
    GemmFwd g{};
    g.a = make_seg(x, (long long)h * w * cin, cin, cin, nullptr, 0, 0, 0);
    g.B = batch; g.Hout = h; g.Wout = w; g.Hin = h; g.Win = w;
    g.ksize = 3; g.stride = 1; g.pad = 1; g.mirror = 0;
    g.w0 = w_score; g.w1 = w_loc; g.bias0 = b_score; g.bias1 = b_loc; g.n0 = n_score; g.n1 = n_loc;
    g.o0 = scores + scores_offset; g.ob0 = scores_batch_stride; g.os0 = n_score;
    g.o1 = locs ? locs + locs_offset : nullptr; g.ob1 = locs_batch_stride; g.os1 = n_loc;
    g.relu = 0;
    return launch_fwd(g, 0, (hipStream_t)stream);
}

extern "C" size_t ssdk_head_conv_bwd_workspace_bytes(int cin, int n_score, int n_loc) {
    return align_up((size_t)9 * cin * (size_t)(n_score + n_loc) * sizeof(float), 256);
}

extern "C" int ssdk_head_conv_bwd(const float* x, int batch, int h, int w, int cin, const float* w_score, int n_score,
                                  const float* w_loc, int n_loc, const float* dscores, long long scores_batch_stride,
                                  long long scores_offset, const float* dlocs, long long locs_batch_stride,
                                  long long locs_offset, float* dx, float* dw_score, float* db_score, float* dw_loc,
                                  float* db_loc, void* workspace, size_t workspace_bytes, void* stream) {
    int rc = check_conv_geom("ssdk_head_conv_bwd", batch, h, w, cin, n_score, n_loc);
    if (rc) return rc;
    SSDK_REQUIRE(x && w_score && dscores && (n_loc == 0 || (w_loc && dlocs)), SSDK_E_INVALID, "ssdk_head_conv_bwd: null pointer");
    hipStream_t s = (hipStream_t)stream;
    const int N = n_score + n_loc;
    RowSeg dy = make_seg(dscores + scores_offset, scores_batch_stride, n_score, n_score, n_loc ? dlocs + locs_offset : nullptr,
                         locs_batch_stride, n_loc, n_loc);
    if (dx) {  // backward-data: dX[m][c] = sum_(tap,n) dY[m + pad - tap][n] * W[n][tap][c]
        SSDK_REQUIRE(workspace && workspace_bytes >= ssdk_head_conv_bwd_workspace_bytes(cin, n_score, n_loc), SSDK_E_WORKSPACE,
                     "ssdk_head_conv_bwd: workspace too small");
        float* wd = (float*)workspace;
        hipLaunchKernelGGL(transpose_taps_kernel, dim3(cdiv(N, 32), cdiv(cin, 32), 9), dim3(256), 0, s, w_score, w_loc, n_score,
                           n_loc, 9, cin, wd);
        SSDK_CHECK_LAUNCH("transpose_taps_kernel");
        GemmFwd g{};
        g.a = dy;
        g.B = batch; g.Hout = h; g.Wout = w; g.Hin = h; g.Win = w;
        g.ksize = 3; g.stride = 1; g.pad = 1; g.mirror = 1;
        g.w0 = wd; g.w1 = nullptr; g.bias0 = nullptr; g.bias1 = nullptr; g.n0 = cin; g.n1 = 0;
        g.o0 = dx; g.ob0 = (long long)h * w * cin; g.os0 = cin; g.o1 = nullptr; g.ob1 = 0; g.os1 = 0;
        g.relu = 0;
        rc = launch_fwd(g, 0, s);
        if (rc) return rc;
    }
    if (dw_score) {
        SSDK_REQUIRE(n_loc == 0 || dw_loc, SSDK_E_INVALID, "ssdk_head_conv_bwd: dw_loc missing");
        SSDK_CHECK_HIP(hipMemsetAsync(dw_score, 0, sizeof(float) * (size_t)n_score * 9 * cin, s));
        if (n_loc) SSDK_CHECK_HIP(hipMemsetAsync(dw_loc, 0, sizeof(float) * (size_t)n_loc * 9 * cin, s));
        GemmWgrad g{};
        g.dy = dy;
        g.x = make_seg(x, (long long)h * w * cin, cin, cin, nullptr, 0, 0, 0);
        SSDK_REQUIRE(cin % 4 == 0 && ((uintptr_t)x & 15) == 0, SSDK_E_UNSUPPORTED, "ssdk_head_conv_bwd: Cin must be a multiple of 4 and x 16-byte aligned");
        g.B = batch; g.Hout = h; g.Wout = w; g.Hin = h; g.Win = w; g.ksize = 3; g.stride = 1; g.pad = 1;
        g.dw0 = dw_score; g.dw1 = dw_loc; g.n0 = n_score; g.n1 = n_loc;
        g.n_tiles = cdiv(N, 128);
        g.c_tiles32 = cdiv(cin, 32);
        g.c_blocks = cdiv(g.c_tiles32, kMaxTN);
        const int out_tiles = 9 * g.n_tiles * g.c_blocks;
        const int slices = cdiv(batch * h * w, 32);
        int ks = cdiv(2048, out_tiles);
        if (ks > slices) ks = slices;
        if (ks < 1) ks = 1;
        g.k_splits = ks;
        const int vec = pick_vec(dy, 4) ;
        const int grid = out_tiles * ks;
        // the dY slice loader reads 128 consecutive n starting at a multiple of 128: segment boundary must not split a vector
        if (vec == 4) hipLaunchKernelGGL(igemm_wgrad_kernel<4>, dim3(grid), dim3(kConvThreads), 0, s, g);
        else if (vec == 2) hipLaunchKernelGGL(igemm_wgrad_kernel<2>, dim3(grid), dim3(kConvThreads), 0, s, g);
        else hipLaunchKernelGGL(igemm_wgrad_kernel<1>, dim3(grid), dim3(kConvThreads), 0, s, g);
        SSDK_CHECK_LAUNCH("igemm_wgrad_kernel");
    }
    if (db_score) {
        SSDK_CHECK_HIP(hipMemsetAsync(db_score, 0, sizeof(float) * (size_t)n_score, s));
        if (n_loc && db_loc) SSDK_CHECK_HIP(hipMemsetAsync(db_loc, 0, sizeof(float) * (size_t)n_loc, s));
        const int rows_per_block = 64;
        const long long M = (long long)batch * h * w;
        hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)((M + rows_per_block - 1) / rows_per_block)), dim3(256), 0, s, dy, batch, h * w,
                           db_score, db_loc, rows_per_block);
        SSDK_CHECK_LAUNCH("colsum_kernel");
    }
    return SSDK_OK;
}
