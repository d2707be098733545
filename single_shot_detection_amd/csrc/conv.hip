// conv.hip -- multi-scale head convolutions as implicit GEMM on the CDNA4 matrix cores.
//
// Reference: detection/detector_builder.py:111-137 (get_heads: per level a 3x3/pad-1 score conv with nb*C outputs
// and a 3x3/pad-1 loc conv with nb*4 outputs, both with bias) applied in detection/detector.py:50-66, each followed
// by permute(0,2,3,1).contiguous().view(B,-1) and a cat over levels -- 2*L library convolutions plus 2*L
// permute copies plus 2 concatenations per forward, and tail levels (5x5 ... 1x1) that cannot fill the chip.
//
// Here every level is ONE GEMM  C[m][n] = sum_k A[m][k] * W[n][k]  with
//     m = (image, y, x) output pixel, n = output channel of the FUSED score|loc head, k = (tap, input channel),
// and ALL levels run in ONE grouped launch (a work list ordered by decreasing work per workgroup), so the small maps
// fill the gaps of the big ones.  The arithmetic is v_mfma_f32_32x32x2_f32: exact fp32 (parity mode -- the loss must
// match the reference to 1e-4).  The epilogue adds the bias and stores straight into the concatenated [B, A*C] /
// [B, A*4] buffers at the level's offset: the NHWC flatten of detector.py:52-63 IS the natural output order of this
// GEMM, so permute, contiguous and cat disappear.  Activations are NHWC (channels-last): a K-slice of an A row is one
// contiguous 128-byte line; weights are [n][tap][cin] (= torch channels_last memory of the OIHW parameter).
//
// Tiling (wave = 64 lanes): workgroup = 4 waves = 128 output pixels x (32*tn) channels, tn <= 4 balanced per level
// (N = 344 -> 4+4+3 tiles, N = 512 -> 4+4+4+4); wave w owns pixel rows 32w..32w+31 and all tn column tiles (tn
// accumulators of 16 registers).  K is walked in slices of 32, channel chunk outer, tap inner.
//
// Two implementations of the same GEMM:
//   * igemm_dma_kernel (the hot one): the slices travel global -> LDS by LDS-DMA (buffer_load ... lds), two LDS stages, one
//     barrier per slice, XOR-swizzled unpadded rows, zero fill by out-of-range buffer offsets -- see the comment in front of
//     dma_tile().  Forward, dense stride-1 dgrad (mirrored taps), scatter-form dgrad; 146 VGPRs, 64 KB LDS, 2 workgroups per CU.
//   * igemm_fwd_kernel: the same pipeline with the next slice staged through registers (global_load -> ds_write_b128, LDS rows
//     padded to 36 floats).  Fallback for channel counts that are not multiples of 32, operands of 2 GiB and more, and the
//     dense dgrad of strided convolutions; ~11 % more cycles than the DMA form.
//
// Backward: pack_dy_kernel gathers each level's slice of dscores|dlocs into 16-byte aligned, zero padded rows
// [pixel][Npad] (the [B, A*C] rows are only 4/8-byte aligned when nb*C is odd), sums the bias gradients on the way and lists
// the pixel rows / the single anchors that carry a gradient.  Backward-data is the forward kernel with mirrored taps (dense)
// or its scatter form over the listed rows (sparse, and every strided convolution).  Backward-weights contracts over pixels
// (igemm_wgrad_dma_kernel; igemm_wgrad_kernel is its register-staged fallback): both operands are read in their natural
// [pixel][channel] form, K = pixels is split across workgroups and partial tiles are accumulated with fp32 atomics.
// Which of the three backward forms (dense / pixel rows / anchor rows) a level takes is decided on the device from the row
// counts (decide_sparse_kernel); see DESIGN.md 4.
#include <stdlib.h>

#include <algorithm>

#include <type_traits>

#include <mutex>

#include "common.h"

namespace ssdk {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kBM = 128;        // output pixels per workgroup
constexpr int kBK = 32;         // K slice
constexpr int kLdsStride = 36;  // floats per LDS row (32 + 4 pad)
constexpr int kMaxTN = 4;       // 32-wide column tiles per workgroup
constexpr int kConvThreads = 256;
constexpr int kMaxProblems = 8;
#ifndef SSDK_CONV_WAVES
#define SSDK_CONV_WAVES 2  // workgroups per CU the GEMM kernels are register-budgeted for: 2 -> <=256 VGPRs, no spills
                           // (measured: the 3-workgroup budget of 168 VGPRs spills the prefetch registers and is 10% slower)
#endif

struct ConvProblem {
    // A operand rows [pixel][channel], one segment
    const float* a;
    long long a_bstride;  // per-image stride (floats)
    int a_pstride;        // per-pixel stride (floats)
    int Cc;               // channels (K per tap)
    int B, Hout, Wout, Hin, Win, ksize, stride, pad;
    // W rows: n < n0 -> w0[n][taps*Cc], else w1[n-n0][taps*Cc]
    const float* w0;
    const float* w1;
    const float* bias0;
    const float* bias1;
    int n0, n1;
    // output: element (image b, pixel p, channel n) at o0 + b*ob0 + p*os0 + n (n < n0) / o1 + b*ob1 + p*os1 + (n-n0)
    float* o0;
    float* o1;
    long long ob0, ob1;
    int os0, os1;
    int tiles_n, n_blocks, m_tiles;
    int m_tiles256;   // M tiles of the 8-wave (256-pixel) tiling
    int k_splits;     // > 1: the K slices are divided over k_splits workgroups that atomically add into a zeroed output
    int block_begin;  // first workgroup of this problem in the grouped grid
    int relu;
    // device-side mode switch (sparse backward): the launch is a no-op for this problem unless *mode == want_mode
    const int* mode;
    int want_mode;
    // SCATTER instantiation: rows are the entries of row_list (pixel ids with a non-zero gradient row), *row_count of them
    const int* row_list;
    const int* row_count;
    int sc_cin;  // scatter: output channels per tap (n = tap * sc_cin + c)
    // column index space: [0, n0) = rows of w0, [n0, n0_pad) unused, [n0_pad, n0_pad + n1) = rows of w1.  n0_pad = n0 except for
    // the LDS-DMA kernel, which rounds it up to 8 so that every 8-row DMA piece reads ONE weight tensor (one descriptor)
    int n0_pad;
    unsigned w0_bytes, w1_bytes;   // LDS-DMA kernel: sizes of the two weight tensors (buffer descriptors)
    int forced;   // n_blocks / k_splits were set by the caller: launch_group keeps them
    // The last 32-column tile holds at most 16 columns (N = 104 of the 21-class heads: 3 tiles + 8 columns): it is computed as a 16-column
    // tile by v_mfma_f32_16x16x1_4b_f32 at half the cycles of a 32 x 32 x 2 (dma_tile, forward LDS-DMA form only; set by launch_group)
    int half_last;
    // BatchNorm statistics of the output, fused into the epilogue (forward, one output, not split over K): per-column sums of the
    // stored values and of their squares are ADDED into stats[0 .. n0) / stats[n0 .. 2 n0) (fp64), stats[2 n0] = rows.  NULL: none.
    double* stats;
};

struct ConvGroup {
    int count;
    int total_blocks;
    const int* vtab;   // compact tile list of a sparse launch (see igemm_dma_kernel), NULL otherwise
    ConvProblem p[kMaxProblems];
};

// Stream-K partition of a grouped forward launch (plain heads forward only): the launch's K slices -- every (M tile, column block,
// K slice), weighted by the column tiles of the block -- are cut into `nwg` equal contiguous ranges, one per persistent workgroup, so
// that all workgroups finish together instead of leaving the last of 2.9 rounds of whole tiles 10 % full.  A tile that straddles a
// range boundary is computed in two K parts: the second part (start of the next workgroup's range, i.e. computed right after launch)
// parks its accumulators in `partial`, the first part (END of the previous workgroup's range) adds them and runs the epilogue -- no
// atomics on the outputs, no zero-fill, and by the time the owner looks the other half has been there for a millisecond.
struct StreamK {
    int nwg;
    long long total_units;
    long long unit_begin[kMaxProblems + 1];   // prefix over the problems in launch order; unit = one K slice of HALF a 32-column tile (a tile = 2, a 16-column remainder tile = 1)
    float* partial;                           // [nwg][4 waves][kMaxTN][4][64 lanes][4]: a workgroup's accumulators as they lie in registers
    unsigned* flags;                          // [nwg]: flags[s] == epoch <=> workgroup s parked its partial tile
    unsigned* timeouts;                       // DEV counter of partners given up on (stays 0; ssdk_heads_fwd_timeouts reads it)
    unsigned* host_err;                       // the same event as a sticky word in pinned HOST memory (NULL: none): the next ssdk_heads_fwd fails
    unsigned epoch;
};

template <int VEC>
struct VecT;
template <>
struct VecT<4> { typedef f32x4 type; };
template <>
struct VecT<1> { typedef float type; };
template <int VEC>
__device__ __forceinline__ typename VecT<VEC>::type vzero();
template <>
__device__ __forceinline__ f32x4 vzero<4>() { return f32x4{0.f, 0.f, 0.f, 0.f}; }
template <>
__device__ __forceinline__ float vzero<1>() { return 0.f; }

// A pointer that is the same in every lane, pinned into an SGPR pair (so that `base + lane_offset` becomes the
// saddr + 32-bit voffset addressing form and never a per-lane 64-bit select or a reload from the kernarg segment).
// The result is typed as a GLOBAL (address space 1) pointer: a generic pointer rebuilt from integers compiles to
// flat_load, which counts against lgkmcnt as well as vmcnt, so every `s_waitcnt lgkmcnt(0)` in front of the MFMAs (meant
// for the LDS reads) would also wait for the slice prefetch that was just issued.
typedef const float __attribute__((address_space(1))) * gptr_t;
__device__ __forceinline__ gptr_t uniform_ptr(const float* p) {
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (gptr_t)(((unsigned long long)hi << 32) | lo);
}
template <int VEC>
struct GVecT;
template <>
struct GVecT<4> { typedef const f32x4 __attribute__((address_space(1))) * type; };
template <>
struct GVecT<1> { typedef const float __attribute__((address_space(1))) * type; };


// Epilogue shared by the GEMM kernels: bias, optional ReLU, store (or atomic add for split-K / scatter).
template <bool SCATTER, int NT>
__device__ __forceinline__ void conv_epilogue(const ConvProblem& g, const f32x16 (&acc)[NT], int m_base, int wave, int r32, int h, int tn,
                                              int n_begin, int M, int N, int hw, int ksp, float* s_red = nullptr, int waves = 4, int half_j = -1) {
    // epilogue: C/D map of the 32x32 MFMA: column = lane & 31, row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)
    if (SCATTER) {
        // row = a pixel (or one anchor of a pixel) with a non-zero output gradient; column n = tap * Cin + c: add the product into
        // the input-gradient pixel that tap connects it to (yo - pad + ky, xo - pad + kx).  32 consecutive lanes = 32 consecutive c
        // of one tap.  When Cin % 32 == 0 a 32-column tile lies inside one tap: tap and first channel are per-tile scalars (the
        // four integer divisions per ELEMENT of the general form were most of the time of a sparse backward workgroup).
        const bool tile_uniform = (g.sc_cin & 31) == 0;
        int dyj[NT], dxj[NT], cbj[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int n0j = n_begin + j * 32;
            const int tap = n0j / g.sc_cin;
            dyj[j] = tap / g.ksize - g.pad;
            dxj[j] = tap % g.ksize - g.pad;
            cbj[j] = n0j % g.sc_cin;
        }
        if (tile_uniform) {
            // Lane l works out row (l & 31) of the wave ONCE: the element offset of its pixel in dX and which of the taps land inside
            // the map; every accumulator element then fetches its row's pair with two cross-lane reads and costs a bit test, an
            // add and the atomic.  (Deriving pixel, tap and bounds per element was ~2,600 instructions per lane and tile -- more
            // than the tile's MFMAs take.)
            long long roff = 0;
            unsigned rmask = 0u;
            {
                const int m = m_base + wave * 32 + r32;
                if (m < M) {
                    const int pid = g.row_list ? g.row_list[m] : m;
                    const int b = pid / hw, r = pid - b * hw;
                    const int yq = r / g.Wout, yo = yq * g.stride, xo = (r - yq * g.Wout) * g.stride;
                    roff = (long long)b * g.ob0 + ((long long)yo * g.Win + xo) * g.os0;
                    for (int ky = 0; ky < g.ksize; ++ky)
                        for (int kx = 0; kx < g.ksize; ++kx) {
                            const int ty = yo - g.pad + ky, tx = xo - g.pad + kx;
                            if (ty >= 0 && ty < g.Hin && tx >= 0 && tx < g.Win) rmask |= 1u << (ky * g.ksize + kx);
                        }
                }
            }
            long long dj[NT];
            unsigned bitj[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                dj[j] = ((long long)dyj[j] * g.Win + dxj[j]) * g.os0 + cbj[j] + r32;
                bitj[j] = (j < tn && n_begin + j * 32 + r32 < N) ? 1u << ((dyj[j] + g.pad) * g.ksize + dxj[j] + g.pad) : 0u;
            }
            const unsigned lo = (unsigned)(unsigned long long)roff, hi = (unsigned)((unsigned long long)roff >> 32);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int src = (e & 3) + 8 * (e >> 2) + 4 * h;
                const unsigned mask_e = (unsigned)__shfl((int)rmask, src, kWave);
                const unsigned long long off_e = (unsigned long long)(unsigned)__shfl((int)lo, src, kWave) |
                                                 ((unsigned long long)(unsigned)__shfl((int)hi, src, kWave) << 32);
                float* const row = g.o0 + (long long)off_e;
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    if (mask_e & bitj[j]) atomicAdd(row + dj[j], acc[j][e]);
            }
            return;
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int m = m_base + wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
            if (m >= M) continue;
            const int pid = g.row_list ? g.row_list[m] : m;
            const int b = pid / hw, r = pid % hw;
            const int yo = (r / g.Wout) * g.stride, xo = (r % g.Wout) * g.stride;
            float* const img = g.o0 + (long long)b * g.ob0;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                if (j >= tn) continue;
                const int n = n_begin + j * 32 + r32;
                if (n >= N) continue;
                int ty, tx, c;
                if (tile_uniform) {
                    ty = yo + dyj[j]; tx = xo + dxj[j]; c = cbj[j] + r32;
                } else {
                    const int tap = n / g.sc_cin;
                    c = n % g.sc_cin;
                    ty = yo - g.pad + tap / g.ksize; tx = xo - g.pad + tap % g.ksize;
                }
                if (ty >= 0 && ty < g.Hin && tx >= 0 && tx < g.Win) atomicAdd(img + ((long long)ty * g.Win + tx) * g.os0 + c, acc[j][e]);
            }
        }
        return;
    }
    // Everything that is fixed per column (which output, column inside it, bias) or per row (element offset of the pixel in both
    // outputs) is worked out once -- the row part by lane (row & 31), fetched with cross-lane reads -- so an element costs a
    // select, an add and its store.  (Phase stamps: re-deriving them per element from the problem record made the epilogue of a
    // 128 x 128 tile 16 us, twice its K loop on the pyramid tail's 1x1 convolutions.)
    const int n0 = g.n0, n0_pad = g.n0_pad;
    const bool split = g.k_splits > 1, relu = g.relu != 0;
    float* const o0 = g.o0;
    float* const o1 = g.o1;
    long long off0 = 0, off1 = 0;   // lane l: row (l & 31) of this wave
    {
        const int m = m_base + wave * 32 + r32;
        if (m < M) {
            const int b = m / hw, pix = m - b * hw;
            off0 = (long long)b * g.ob0 + (long long)pix * g.os0;
            off1 = o1 ? (long long)b * g.ob1 + (long long)pix * g.os1 : 0;
        } else {
            off0 = -1;   // (no element offset is negative)
        }
    }
    int colj[NT];
    float biasj[NT];
    unsigned is1 = 0u, valid = 0u;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int n = n_begin + j * 32 + r32;
        const bool second = n >= n0;
        const bool ok = j < tn && j != half_j && n < N && !(second && n < n0_pad);   // (n0 .. n0_pad: padding columns between the two heads)
        colj[j] = second ? n - n0_pad : n;
        const float* bias = second ? g.bias1 : g.bias0;
        biasj[j] = (ok && bias && ksp == 0) ? bias[colj[j]] : 0.0f;
        if (second) is1 |= 1u << j;
        if (ok) valid |= 1u << j;
    }
    const unsigned lo0 = (unsigned)(unsigned long long)off0, hi0 = (unsigned)((unsigned long long)off0 >> 32);
    const unsigned lo1 = (unsigned)(unsigned long long)off1, hi1 = (unsigned)((unsigned long long)off1 >> 32);
    // The bias values are in their registers HERE: left to itself the compiler waits for the loads at their first use inside the
    // element loop -- behind branches it can no longer tell them from the stores / atomics issued since, so every element waited
    // vmcnt(0), i.e. for the previous element's store to complete (seen in the one-column-tile instantiation: 17 waits per tile).
#pragma unroll
    for (int j = 0; j < NT; ++j) asm volatile("" : "+v"(biasj[j]));
    double* const stats = (!split && s_red) ? g.stats : nullptr;   // (uniform over the workgroup)
    float cs1[NT], cs2[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) cs1[j] = cs2[j] = 0.0f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int src = (e & 3) + 8 * (e >> 2) + 4 * h;
        const long long r0 = (long long)((unsigned long long)(unsigned)__shfl((int)lo0, src, kWave) | ((unsigned long long)(unsigned)__shfl((int)hi0, src, kWave) << 32));
        if (r0 < 0) continue;
        float* const p0 = o0 + r0;
        float* p1 = p0;
        if (o1) p1 = o1 + (long long)((unsigned long long)(unsigned)__shfl((int)lo1, src, kWave) | ((unsigned long long)(unsigned)__shfl((int)hi1, src, kWave) << 32));
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            if (!((valid >> j) & 1u)) continue;
            float* const dst = (((is1 >> j) & 1u) ? p1 : p0) + colj[j];
            const float v = acc[j][e] + biasj[j];
            if (split) {
                atomicAdd(dst, v);
            } else {
                const float out = relu ? fmaxf(v, 0.0f) : v;
                *dst = out;
                cs1[j] += out;
                cs2[j] += out * out;
            }
        }
    }
    if (half_j >= 0) {
        // the 16-column remainder tile: C/D map of the four 16 x 16 blocks of v_mfma_f32_16x16x1_4b_f32 (tools/mfma_16x16x1_layout.hip):
        // register 4 * b + r of lane l = block b, row 4 * (l >> 4) + r, column l & 15; blocks b and b + 2 are the two K halves of the
        // wave's rows 16 * b .. 16 * b + 15
        const int lane = h * 32 + r32;
        const int n = n_begin + half_j * 32 + (lane & 15);
        const bool second = n >= n0;
        const bool ok = n < N && !(second && n < n0_pad);
        const int col = second ? n - n0_pad : n;
        const float* bias = second ? g.bias1 : g.bias0;
        float bv = (ok && bias && ksp == 0) ? bias[col] : 0.0f;
        asm volatile("" : "+v"(bv));   // (as above: the load is waited for here, not in front of every store)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int src = 16 * b + 4 * (lane >> 4) + r;   // row of the wave = the lane that worked out its offsets
                const long long r0 = (long long)((unsigned long long)(unsigned)__shfl((int)lo0, src, kWave) | ((unsigned long long)(unsigned)__shfl((int)hi0, src, kWave) << 32));
                long long r1 = 0;
                if (o1) r1 = (long long)((unsigned long long)(unsigned)__shfl((int)lo1, src, kWave) | ((unsigned long long)(unsigned)__shfl((int)hi1, src, kWave) << 32));
                if (r0 < 0 || !ok) continue;
                float* const dst = (second ? o1 + r1 : o0 + r0) + col;
                float v = 0.0f;
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    if (j == half_j) v = acc[j][4 * b + r] + acc[j][4 * (b + 2) + r];
                v += bv;
                if (split) atomicAdd(dst, v);
                else *dst = relu ? fmaxf(v, 0.0f) : v;
            }
    }
    if (stats) {
        // BatchNorm statistics of what was just stored: a lane holds the sums of its column over 16 rows of the wave's 32; the two lane
        // halves together are the wave's 32 rows; the waves meet in LDS (the K loop's staging buffers are free by now), then ONE fp64
        // atomic per column and sum for the whole 32 * waves rows of the tile
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            cs1[j] += __shfl_xor(cs1[j], 32, kWave);
            cs2[j] += __shfl_xor(cs2[j], 32, kWave);
        }
        __syncthreads();
        if (h == 0) {
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                s_red[((wave * NT + j) * 32 + r32) * 2 + 0] = cs1[j];
                s_red[((wave * NT + j) * 32 + r32) * 2 + 1] = cs2[j];
            }
        }
        __syncthreads();
        const int t = wave * 64 + h * 32 + r32;   // thread id inside the workgroup
        for (int q = t; q < NT * 32; q += waves * 64) {
            const int j = q >> 5, c = q & 31, n = n_begin + j * 32 + c;
            if (j < tn && n < n0) {
                float a = 0.0f, b = 0.0f;
                for (int w = 0; w < waves; ++w) { a += s_red[((w * NT + j) * 32 + c) * 2 + 0]; b += s_red[((w * NT + j) * 32 + c) * 2 + 1]; }
                atomicAdd(stats + n, (double)a);
                atomicAdd(stats + n0 + n, (double)b);
            }
        }
        if (m_base == 0 && n_begin == 0 && t == 0) stats[2 * n0] = (double)M;
    }
}

#ifdef SSDK_CONV_TRACE
// experiment only: per-slice timestamps of a few waves (never built into the shipped library)
__device__ unsigned long long g_trace[32 * 4 * 64 * 8];
extern "C" int ssdk_debug_read_trace(unsigned long long* host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_trace), sizeof(unsigned long long) * n);
}
#define TRACE(slot)                                                                                              \
    if (trace_on && (tid & 63) == 0 && slice - slice_begin < 64)                                                   \
    {                                                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                       \
        g_trace[((trace_blk * 4 + wave) * 64 + (slice - slice_begin)) * 8 + (slot)] = __builtin_readcyclecounter(); \
        __builtin_amdgcn_sched_barrier(0);                                                                       \
    }
#else
#define TRACE(slot)
#endif

// ---- forward / backward-data, LDS-DMA staging (the hot instantiations) -----------------------------------------------
// Same GEMM, same tiling and the same epilogue as igemm_fwd_kernel below, but the K slices travel global -> LDS directly
// (buffer_load_dwordx4 ... lds), with no staging registers, no ds_write pass and almost no per-slice vector arithmetic:
//   * measured on the register-staged kernel (in-kernel cycle stamps): of the two waves that share a SIMD
//     the one in wave slot 1 loses vector-issue arbitration to its partner's MFMA stream, and the ~100 VALU instructions
//     per slice of address arithmetic / zero-fill selects cost it ~4,000 cycles per slice (12-14k cycles per slice
//     against 8-9k for the slot-0 wave).  Here a slice costs 8 VALU + 8 DMA issues.
//   * a DMA piece is one wave instruction = 64 lanes x 16 B = 8 rows x 128 B, written lane-linearly into LDS.  Rows
//     cannot be padded, so bank conflicts are avoided by an XOR swizzle on the SOURCE side: LDS chunk position q of row
//     r holds source chunk q ^ ((r >> 1) & 7); the fragment reads apply the same XOR (conflict-free ds_read_b128).
//   * padding taps, rows past M / N: the lane's voffset gets bit 31 set -> out of range of the buffer -> the DMA
//     writes zeros (checked on the device: tools/lds_dma_test.hip).  The host only selects this kernel when every valid
//     byte offset is below 2^31 and Cc % 32 == 0.
//   * wave w stages the A rows it multiplies itself (32w..32w+31) and W rows 32w..32w+31 for everybody; one barrier
//     per slice publishes slice s+1 and retires the reads of slice s (two LDS stages of 32 KB).
typedef __attribute__((address_space(3))) void* lds_ptr_t;
constexpr unsigned kOobBit = 0x80000000u;

// WAVES = 4: 128-pixel tiles, two workgroups per CU.  WAVES = 8: 256-pixel tiles, ONE workgroup per CU whose waves w and
// w + 4 share a SIMD: the barrier keeps the two in step, so the slot-1 wave cannot fall behind its partner the way it
// does between two independent workgroups (cycle stamps: 12.0k vs 8.7k cycles per slice), and the W slice is staged once
// for 256 pixels.  The host picks WAVES = 8 when the 256-pixel tiling still fills the chip.
#ifdef SSDK_CONV_PHASE
// experiment only (never built into the shipped library; tools/phase_conv.py): 100 MHz wall-clock stamps of the first 64
// workgroups at tile start / K loop start / epilogue start / end
__device__ unsigned long long g_phase[64 * 4];
extern "C" int ssdk_debug_read_phase(unsigned long long* host) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_phase), sizeof(unsigned long long) * 64 * 4); }
#define PHASE(i) if (threadIdx.x == 0 && blockIdx.x < 64) g_phase[blockIdx.x * 4 + (i)] = wall_clock64();
#else
#define PHASE(i)
#endif
// sk_mode (stream-K): 0 = a whole tile (or a split-K part `ksp`); 1 = K slices [sk_s0, sk_s1) of the tile, the accumulators are parked
// in sk_buf and *sk_flag set to sk_epoch; 2 = K slices [sk_s0, sk_s1), then the accumulators parked by another workgroup are added
// (once *sk_flag == sk_epoch) and the epilogue runs.
// Test hook (ssdk_debug_streamk_fault): the stream-K workgroup whose range index equals g_sk_drop_wg never raises its "parked" flag (a
// workgroup that never ran), and an owner gives up after g_sk_spin_limit polls instead of 2^22.  -1 / 0: off (the defaults).
__device__ int g_sk_drop_wg = -1;
__device__ unsigned g_sk_spin_limit = 0;

template <bool MIRROR, bool GENERIC, bool SCATTER, int WAVES, int BK, int MAXTN>
__device__ __forceinline__ void dma_tile(const ConvProblem& g, int pi, int m_tile, int n_block, int ksp, int sk_mode = 0, int sk_s0 = 0, int sk_s1 = 0,
                                         float* sk_buf = nullptr, unsigned* sk_flag = nullptr, unsigned sk_epoch = 0, unsigned* sk_timeouts = nullptr,
                                         int sk_parts = 1, unsigned* sk_host_err = nullptr, bool sk_poisoned = false, bool sk_drop = false) {
    PHASE(0)
    // BK = K slice: 32 floats (128-byte rows, 8 rows per DMA piece, 64 KB of LDS: 2 workgroups per CU) or 16 floats (64-byte rows,
    // 16 rows per piece, 32 KB: 3 workgroups per CU at <= 170 VGPRs, a barrier every 32 MFMAs instead of 64)
    constexpr int BM = 32 * WAVES;                         // output pixels per workgroup
    constexpr int kRowsPerPiece = 1024 / (BK * 4);         // rows of one 1 KB DMA piece
    constexpr int kChunks = BK / 4;                        // 16-byte chunks per row
    constexpr int kAPieces = 32 / kRowsPerPiece;           // A pieces each wave stages per slice (its own 32 rows)
    // MAXTN = 32-column tiles per workgroup: 4 (128 columns, 64 KB of LDS) or 6 (192 columns, 80 KB: two workgroups fill the CU's 160 KB)
    constexpr int kWPieces = (MAXTN * 32 / kRowsPerPiece) / WAVES;   // W pieces each wave stages per slice
    constexpr int kBK = BK;                                // (shadows the file-level constant inside this function)
    // FOUR separate LDS objects (two stages x two operands), not one array: the compiler orders a ds_read behind an
    // in-flight LDS-DMA (s_waitcnt vmcnt(0) in front of the read) unless alias scopes prove they touch different objects,
    // and only distinct __shared__ variables get such scopes.
    __shared__ __attribute__((aligned(1024))) float s_a0[BM * kBK];
    __shared__ __attribute__((aligned(1024))) float s_a1[BM * kBK];
    __shared__ __attribute__((aligned(1024))) float s_b0[MAXTN * 32 * kBK];
    __shared__ __attribute__((aligned(1024))) float s_b1[MAXTN * 32 * kBK];
    // MAXTN == 1 (one 32-column tile per workgroup: the launches smaller than the chip, whose K loop runs at the latency of the DMA --
    // ~0.75 us per slice against 0.43 us of MFMA, tools/phase_conv.py): THREE stages of 16 + 4 KB, the DMA issued two slices ahead
    constexpr int kStages = MAXTN == 1 ? 3 : 2;
    __shared__ __attribute__((aligned(1024))) float s_a2[kStages == 3 ? BM * kBK : 4];
    __shared__ __attribute__((aligned(1024))) float s_b2[kStages == 3 ? MAXTN * 32 * kBK : 4];

    const int Cc = g.Cc;
    const int ks = g.ksize;
    const int taps = SCATTER ? 1 : ks * ks;
    const int chunks = Cc / kBK;
    const int n_slices_all = taps * chunks;
    const int per_split = (n_slices_all + g.k_splits - 1) / g.k_splits;
    const int slice_begin = sk_mode ? sk_s0 : ksp * per_split;
    const int n_slices = sk_mode ? sk_s1 : min(n_slices_all, slice_begin + per_split);
    if (slice_begin >= n_slices) return;
    const int K = taps * Cc;
    const int N = g.n0_pad + g.n1;
    const int hw = g.Hout * g.Wout;
    int M = (SCATTER && g.row_list) ? *g.row_count : g.B * hw;   // one past the last existing row
    int m_base = m_tile * BM;                                                     // first row of this workgroup
    const float* const w0p = g.w0;
    if (SCATTER && m_base >= M) return;

    const int base_t = g.tiles_n / g.n_blocks, rem_t = g.tiles_n % g.n_blocks;
    const int tn = base_t + (n_block < rem_t ? 1 : 0);
    const int n_begin = (n_block * base_t + min(n_block, rem_t)) * 32;
    // the launch's last tile as a 16-column tile (ConvProblem::half_last): tile tn - 1 of the last column block
    const bool half = !MIRROR && !SCATTER && WAVES == 4 && BK == 32 && MAXTN == 4 && g.half_last && n_block == g.n_blocks - 1;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r32 = lane & 31, h = lane >> 5;

    // ---- DMA source offsets (per lane, constant over the K loop) ---------------------------------------------------
    // element offsets are taken relative to a shifted base so that every part is non-negative:
    //   forward:  pixel part starts at (y*stride - pad, x*stride - pad) >= -pad*(Win+1) pixels;  tap part (ky*Win + kx) >= 0
    //   mirror:   pixel part (y + pad, x + pad) >= 0;  tap part -(ky*Win + kx) >= -((ks-1)*(Win+1))
    const int a_ps = g.a_pstride, win_ps = g.Win * a_ps;
    const int shift = SCATTER ? 0 : (MIRROR ? -(ks - 1) * (win_ps + a_ps) : -g.pad * (win_ps + a_ps));   // elements, <= 0
    unsigned a_vo[4], a_nmask[4], w_vo[6];   // (kAPieces / kWPieces entries used)
    bool w_seg1[6];
#pragma unroll
    for (int i = 0; i < kAPieces; ++i) {
        const int row = kRowsPerPiece * i + lane / kChunks;        // row inside the wave's 32
        // swizzle: chunk position q of row r holds source chunk q ^ f(r); f = (r >> 1) & 7 for 128-byte rows, (r >> 2) & 3 for
        // 64-byte rows (conflict-free ds_read_b128 over the 16-lane groups of the instruction; row + 32 * wave has the same f)
        const int src_chunk = (lane % kChunks) ^ (BK == 32 ? (row >> 1) & 7 : (row >> 2) & 3);
        const int m = m_base + wave * 32 + row;
        a_vo[i] = kOobBit;
        a_nmask[i] = 0;
        if (m < M) {
            if (SCATTER) {
                // (pixel-sparse rows index the dense packed dY through row_list)
                a_vo[i] = (unsigned)((g.row_list ? g.row_list[m] : m) * a_ps + src_chunk * 4) * 4u;
            } else {
                const int b = m / hw, r = m % hw;
                const int y = r / g.Wout, x = r % g.Wout;
                unsigned mask = 0;
                int by, bx;
                // (the taps that exist = existing kernel rows x existing kernel columns: 2 * ks compares, no division by ks -- the tap-by-tap
                // loop with t / ks and t % ks was 3.5 us of a 3 x 3 tile's 6 us prologue, tools/phase_conv.py)
                unsigned colbits = 0;
                if (!MIRROR) {
                    by = y * g.stride - g.pad;
                    bx = x * g.stride - g.pad;
                    for (int kx = 0; kx < ks; ++kx) colbits |= ((unsigned)(bx + kx) < (unsigned)g.Win ? 1u : 0u) << kx;
                    for (int ky = 0; ky < ks; ++ky)
                        if ((unsigned)(by + ky) < (unsigned)g.Hin) mask |= colbits << (ky * ks);
                    by += g.pad;   // relative to the shifted base
                    bx += g.pad;
                } else {
                    by = y + g.pad;
                    bx = x + g.pad;
                    for (int kx = 0; kx < ks; ++kx) colbits |= ((unsigned)(bx - kx) < (unsigned)g.Win ? 1u : 0u) << kx;   // stride 1 only
                    for (int ky = 0; ky < ks; ++ky)
                        if ((unsigned)(by - ky) < (unsigned)g.Hin) mask |= colbits << (ky * ks);
                }
                a_vo[i] = (unsigned)(b * (int)g.a_bstride + by * win_ps + bx * a_ps + src_chunk * 4) * 4u;
                a_nmask[i] = ~mask;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < kWPieces; ++i) {
        const int wrow = (wave * kWPieces + i) * kRowsPerPiece + lane / kChunks;   // row of the 128-row W slice
        const int src_chunk = (lane % kChunks) ^ (BK == 32 ? (wrow >> 1) & 7 : (wrow >> 2) & 3);
        const int n = n_begin + wrow;
        // (n0_pad is a multiple of the rows of a piece: a piece never mixes the two weight tensors)
        w_vo[i] = (wrow < tn * 32 && n < N && !(n >= g.n0 && n < g.n0_pad))
                      ? (unsigned)((n < g.n0 ? n : n - g.n0_pad) * K) * 4u + (unsigned)src_chunk * 16u
                      : kOobBit;
        w_seg1[i] = n_begin + (wave * kWPieces + i) * kRowsPerPiece >= g.n0_pad;   // uniform: which tensor this piece reads
    }

    // buffer descriptors (wave-uniform): A window starts `shift` elements before the tensor, W window spans both segments
    const long long a_records = ((long long)g.B * g.a_bstride - shift) * 4;
    const __amdgpu_buffer_rsrc_t rsrc_a =
        __builtin_amdgcn_make_buffer_rsrc((void*)uniform_ptr(g.a + shift), 0, (int)(a_records > 0x7FFFFFFFLL ? 0x7FFFFFFFLL : a_records), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w0 = __builtin_amdgcn_make_buffer_rsrc((void*)uniform_ptr(w0p), 0, (int)g.w0_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w1 = __builtin_amdgcn_make_buffer_rsrc((void*)uniform_ptr(g.w1 ? g.w1 : g.w0), 0, (int)(g.w1 ? g.w1_bytes : 0), 0x00020000);

    // position of the next slice to stage: channel chunk outer, tap inner (the taps of a chunk re-read rows still in L1/L2)
    int ld_tap = slice_begin % taps, ld_chunk = slice_begin / taps;
    int ld_ky = ld_tap / ks, ld_kx = ld_tap % ks;
    // staging of one slice = 4 A pieces + 4 W pieces per wave; slice_offsets() fixes the (uniform) source offsets of the
    // slice at the counters and advances them, stage_piece() issues piece i.  The pieces are issued one A + one W per
    // 16-MFMA group rather than back to back: a DMA instruction costs its wave 60-185 cycles of issue, and eight in a
    // row at the top of the slice is a ~1,500-cycle hole in that wave's MFMA stream (measured with cycle stamps).
    unsigned so_a = 0, so_w = 0, tap_bit = 0;
    auto slice_offsets = [&](bool advance) {
        const int tap_off = MIRROR ? (ks - 1 - ld_ky) * win_ps + (ks - 1 - ld_kx) * a_ps : ld_ky * win_ps + ld_kx * a_ps;
        so_a = (unsigned)(tap_off + ld_chunk * kBK) * 4u;
        so_w = (unsigned)(ld_tap * Cc + ld_chunk * kBK) * 4u;
        tap_bit = (unsigned)ld_tap;
        const int inc = advance ? 1 : 0;   // past the last slice: stage the same slice again (into the stage nobody reads)
        ld_tap += inc;
        ld_kx += inc;
        const bool wrap_x = ld_kx == ks, wrap_t = ld_tap == taps;
        ld_kx = (wrap_x || wrap_t) ? 0 : ld_kx;
        ld_ky = wrap_t ? 0 : ld_ky + (wrap_x ? 1 : 0);
        ld_tap = wrap_t ? 0 : ld_tap;
        ld_chunk += wrap_t ? 1 : 0;
    };
    auto stage_piece = [&](int DST, int i) {   // DST is a literal at every call site (folds after inlining)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lds_ptr_t)((DST == 2 ? s_a2 : DST ? s_a1 : s_a0) + (wave * 32 + kRowsPerPiece * i) * kBK), 16,
                                                 a_vo[i] | ((a_nmask[i] >> tap_bit) << 31), so_a, 0, 0);
    };
    auto stage_w_piece = [&](int DST, int i) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(w_seg1[i] ? rsrc_w1 : rsrc_w0, (lds_ptr_t)((DST == 2 ? s_b2 : DST ? s_b1 : s_b0) + ((wave * ((MAXTN * 32 / (1024 / (BK * 4))) / WAVES) + i) * (1024 / (BK * 4))) * kBK), 16, w_vo[i],
                                                 so_w, 0, 0);
    };

    f32x16 acc[MAXTN];
#pragma unroll
    for (int j = 0; j < MAXTN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.0f;

    // fragment reads: lane (r, h) reads source chunk 2*gk + h of its row, stored at position (2*gk + h) ^ ((r >> 1) & 7)
    const int pos0 = h ^ (BK == 32 ? (r32 >> 1) & 7 : (r32 >> 2) & 3);
    const int a_row = (wave * 32 + r32) * kBK, b_row = r32 * kBK;
    // 16-column remainder tile: lane l reads W row (l & 15) of that tile, the same K chunk 2 * gk + h as its A fragment -- block l >> 4 of
    // the four-block MFMA then multiplies rows 16 * ((l >> 4) & 1) + (l & 15) of the wave (what the lane's A fragment holds) with K half h
    const int pos0h = h ^ (((lane & 15) >> 1) & 7);
    const int bh_row = (lane & 15) * kBK;

    slice_offsets(true);
#pragma unroll
    for (int i = 0; i < kAPieces; ++i) stage_piece(0, i);
#pragma unroll
    for (int i = 0; i < kWPieces; ++i) stage_w_piece(0, i);
    __syncthreads();   // (waits for the DMA: it is a pending LDS write of this wave)

    auto k_loop = [&](auto tn_c, auto half_c) {
        constexpr int TN = decltype(tn_c)::value;     // whole 32-column tiles
        constexpr bool HALF = decltype(half_c)::value && TN < MAXTN;   // + the 16-column tile in acc[TN]
        constexpr int TH = HALF ? TN : 0;
#ifdef SSDK_CONV_TRACE
        const bool trace_on = pi == 0 && (blockIdx.x % 31) == 0 && blockIdx.x / 31 < 32;
        const int trace_blk = blockIdx.x / 31;
        if (trace_on && lane == 0)
            g_trace[((trace_blk * 4 + wave) * 64) * 8 + 7] = (unsigned long long)__builtin_amdgcn_s_getreg((15 << 11) | 4) |
                                                             ((unsigned long long)__builtin_amdgcn_s_getreg((3 << 11) | 20) << 32) |
                                                             ((unsigned long long)TN << 40) | ((unsigned long long)m_tile << 44);
#endif
        auto body = [&](auto st_c, int slice) {
            constexpr int ST = decltype(st_c)::value;
            TRACE(0)
            slice_offsets(slice + 2 < n_slices);   // the slice staged now is slice + 1
            // fragments of group gk + 1 are read, and DMA piece gk is issued, BEFORE the 16 MFMAs of group gk (pinned with a
            // scheduling barrier: left alone the scheduler sinks the DMAs to the end of the slice, right in front of the wait)
            f32x4 av[2], bv[2][TN > 0 ? TN : 1], bh[2];
            auto read_frags = [&](int buf, int gk) {
                const int pos = (pos0 ^ (2 * gk)) * 4;
                av[buf] = *reinterpret_cast<const f32x4*>(&(ST ? s_a1 : s_a0)[a_row + pos]);
#pragma unroll
                for (int j = 0; j < TN; ++j) bv[buf][j] = *reinterpret_cast<const f32x4*>(&(ST ? s_b1 : s_b0)[b_row + j * 32 * kBK + pos]);
                if (HALF) bh[buf] = *reinterpret_cast<const f32x4*>(&(ST ? s_b1 : s_b0)[bh_row + TH * 32 * kBK + (pos0h ^ (2 * gk)) * 4]);
            };
            read_frags(0, 0);
#pragma unroll
            for (int gk = 0; gk < kBK / 8; ++gk) {
                if (gk + 1 < kBK / 8) read_frags((gk + 1) & 1, gk + 1);
                if (gk < kAPieces) stage_piece(ST ^ 1, gk);
                if (gk < kWPieces) stage_w_piece(ST ^ 1, gk);
                if (gk + kBK / 8 < kWPieces) stage_w_piece(ST ^ 1, gk + kBK / 8);   // (192-column workgroups: 6 W pieces per wave)
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[gk & 1][kk], bv[gk & 1][j][kk], acc[j], 0, 0, 0);
                    if (HALF) acc[TH] = __builtin_amdgcn_mfma_f32_16x16x1f32(av[gk & 1][kk], bh[gk & 1][kk], acc[TH], 0, 0, 0);
                }
#ifdef SSDK_CONV_TRACE
                if (gk == 0) { TRACE(1) }
                if (gk == 1) { TRACE(2) }
                if (gk == 2) { TRACE(3) }
                if (gk == 3) { TRACE(4) }
#endif
            }
            __syncthreads();
            TRACE(5)
        };
        for (int slice = slice_begin; slice < n_slices; slice += 2) {
            body(std::integral_constant<int, 0>{}, slice);
            if (slice + 1 < n_slices) body(std::integral_constant<int, 1>{}, slice + 1);
        }
    };
    PHASE(1)
    if (half) {
        if constexpr (!MIRROR && !SCATTER && WAVES == 4 && BK == 32 && MAXTN == 4) {
            switch (tn - 1) {
                case 3: k_loop(std::integral_constant<int, 3>{}, std::true_type{}); break;
                case 2: k_loop(std::integral_constant<int, 2>{}, std::true_type{}); break;
                case 1: k_loop(std::integral_constant<int, 1>{}, std::true_type{}); break;
                default: k_loop(std::integral_constant<int, 0>{}, std::true_type{}); break;
            }
        }
    } else {
        if constexpr (MAXTN == 1) {
            // Three stages: slice i is read from stage i % 3 while the DMA of slice i + 2 goes into stage (i + 2) % 3 (last read in iteration
            // i - 1, behind that iteration's barrier).  The end of an iteration waits for the wave's own DMAs of slice i + 1 only -- the
            // kAPieces + kWPieces issued in this iteration stay in flight (vmcnt counts down in issue order) -- then the barrier.
            constexpr int kPer = kAPieces + kWPieces;
            static_assert(kPer < 16, "vmcnt immediate below");
            if (slice_begin + 1 < n_slices) {   // slice_begin + 1 into stage 1 (slice_begin is in stage 0, landed: the barrier above)
                slice_offsets(slice_begin + 2 < n_slices);
#pragma unroll
                for (int i = 0; i < kAPieces; ++i) stage_piece(1, i);
#pragma unroll
                for (int i = 0; i < kWPieces; ++i) stage_w_piece(1, i);
            }
            auto body3 = [&](auto st_c, int slice) {
                constexpr int ST = decltype(st_c)::value;
                constexpr int NX = (ST + 2) % 3;
                const float* const sa = ST == 2 ? s_a2 : ST ? s_a1 : s_a0;
                const float* const sb = ST == 2 ? s_b2 : ST ? s_b1 : s_b0;
                const bool more = slice + 2 < n_slices;   // (uniform) is there a slice to stage?  Past the end nothing is issued: the launch would end waiting for it
                if (more) slice_offsets(slice + 3 < n_slices);   // the slice staged now is slice + 2
                f32x4 av[2], bv[2];
                auto read_frags = [&](int buf, int gk) {
                    const int pos = (pos0 ^ (2 * gk)) * 4;
                    av[buf] = *reinterpret_cast<const f32x4*>(&sa[a_row + pos]);
                    bv[buf] = *reinterpret_cast<const f32x4*>(&sb[b_row + pos]);
                };
                read_frags(0, 0);
#pragma unroll
                for (int gk = 0; gk < kBK / 8; ++gk) {
                    if (gk + 1 < kBK / 8) read_frags((gk + 1) & 1, gk + 1);
                    if (gk == 0 && more) {
#pragma unroll
                        for (int i = 0; i < kAPieces; ++i) stage_piece(NX, i);
#pragma unroll
                        for (int i = 0; i < kWPieces; ++i) stage_w_piece(NX, i);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[gk & 1][kk], bv[gk & 1][kk], acc[0], 0, 0, 0);
                }
                // s_waitcnt vmcnt(n) lgkmcnt(0): simm16 = vmcnt[3:0] | expcnt (7: no wait) << 4 | lgkmcnt << 8 | vmcnt[5:4] << 14;
                // n = what this iteration issued (everything older -- slice + 1 -- has then landed)
                if (more) __builtin_amdgcn_s_waitcnt(kPer | (7 << 4));
                else __builtin_amdgcn_s_waitcnt(0 | (7 << 4));
                __builtin_amdgcn_s_barrier();
            };
            int slice = slice_begin;
            for (; slice + 2 < n_slices; slice += 3) {
                body3(std::integral_constant<int, 0>{}, slice);
                body3(std::integral_constant<int, 1>{}, slice + 1);
                body3(std::integral_constant<int, 2>{}, slice + 2);
            }
            if (slice < n_slices) {
                body3(std::integral_constant<int, 0>{}, slice);
                if (slice + 1 < n_slices) body3(std::integral_constant<int, 1>{}, slice + 1);
            }
        } else {
            switch (tn) {
                case 6: if constexpr (MAXTN >= 6) { k_loop(std::integral_constant<int, 6>{}, std::false_type{}); } break;
                case 5: if constexpr (MAXTN >= 6) { k_loop(std::integral_constant<int, 5>{}, std::false_type{}); } break;
                case 4: k_loop(std::integral_constant<int, 4>{}, std::false_type{}); break;
                case 3: k_loop(std::integral_constant<int, 3>{}, std::false_type{}); break;
                case 2: k_loop(std::integral_constant<int, 2>{}, std::false_type{}); break;
                default: k_loop(std::integral_constant<int, 1>{}, std::false_type{}); break;
            }
        }
    }
    PHASE(2)
    if (!GENERIC && !SCATTER && WAVES == 4 && sk_mode) {   // (forward and mirrored-tap backward-data: igemm_streamk_kernel<MIRROR>)
        // accumulator image: [wave][column tile j][e / 4][lane] float4 -- every store / load instruction moves 1 KB contiguous
        f32x4* img = reinterpret_cast<f32x4*>(sk_buf) + (size_t)wave * MAXTN * 4 * 64 + lane;
        if (sk_mode == 1) {
#pragma unroll
            for (int j = 0; j < MAXTN; ++j)
                if (j < tn)
#pragma unroll
                    for (int q = 0; q < 4; ++q) img[(j * 4 + q) * 64] = f32x4{acc[j][4 * q], acc[j][4 * q + 1], acc[j][4 * q + 2], acc[j][4 * q + 3]};
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");   // every wave: its stores have reached memory the other XCDs see ...
            __syncthreads();
            if (tid == 0 && !sk_drop) __hip_atomic_store(sk_flag, sk_epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);   // ... before the flag does
            return;
        }
        // the rest of the tile was parked by the next sk_parts workgroups (one when a range is longer than a tile, several when a
        // small launch cuts a long K chain into many ranges); they all parked first thing after launch
#pragma unroll 1
        for (int part = 0; part < sk_parts; ++part) {
            if (tid == 0) {
                unsigned spins = 0;
                bool lost = false;
                const unsigned limit = g_sk_spin_limit ? g_sk_spin_limit : (1u << 22);
                while (__hip_atomic_load(sk_flag + part, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != sk_epoch) {
                    __builtin_amdgcn_s_sleep(8);
                    if (++spins > limit) { lost = true; break; }   // (never hangs: a lost partner costs the tile, not the GPU)
                }
                if (lost) {
                    // LOUD: the device counter (ssdk_heads_fwd_timeouts), a sticky word in pinned host memory that makes every later
                    // ssdk_heads_fwd on this process fail, and the tile itself is poisoned below -- an incomplete sum is never stored
                    atomicAdd(sk_timeouts, 1u);
                    if (sk_host_err) __hip_atomic_store(sk_host_err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
                s_a0[0] = lost ? __builtin_nanf("") : 0.0f;   // (the staging LDS is free: the K loop ended behind a barrier)
            }
            __syncthreads();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // (every wave: its reads below come from memory, not from lines cached earlier)
            const float poison = s_a0[0];
            const f32x4* pimg = img + (size_t)part * (4 * MAXTN * 4 * 64);
#pragma unroll
            for (int j = 0; j < MAXTN; ++j)
                if (j < tn)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 v = pimg[(j * 4 + q) * 64];
                        acc[j][4 * q] += v[0]; acc[j][4 * q + 1] += v[1]; acc[j][4 * q + 2] += v[2]; acc[j][4 * q + 3] += v[3];
                    }
            if (poison != poison) {
#pragma unroll
                for (int j = 0; j < MAXTN; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[j][e] = poison;
            }
            __syncthreads();   // (s_a0[0] is rewritten for the next part)
            // the consumed flag goes back to 0: a launch replayed from a captured HIP graph carries the SAME epoch every time, and must
            // not take the previous replay's flag for this one's (a partner writes its flag once per launch, so nothing races here).
            // A flag that was NOT consumed (timeout) is left alone: the late partner may still be writing it.
            if (tid == 0 && poison == poison) __hip_atomic_store(sk_flag + part, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (!GENERIC && !SCATTER && WAVES == 4 && sk_poisoned) {
        // an EARLIER launch on this workspace lost a partner (igemm_streamk_kernel read the counter at entry): its flag may still be up,
        // and a launch replayed from a HIP graph carries the same epoch -- it would add that launch's stale partial tile without
        // noticing.  From the first loss on, every tile of every launch on the workspace is NaN.
        const float nanv = __builtin_nanf("");
#pragma unroll
        for (int j = 0; j < MAXTN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][e] = nanv;
    }
    conv_epilogue<SCATTER>(g, acc, m_base, wave, r32, h, tn, n_begin, M, N, hw, ksp, (!MIRROR && !SCATTER) ? s_a0 : nullptr, WAVES,
                           half ? tn - 1 : -1);   // (a stream-K owner too: the fix-up ends behind a barrier)
    PHASE(3)
}


// vtab (pixel-sparse backward only): the row tiles that exist, counted on the device after the rows were listed --
//   [0] number of virtual workgroups, [1 + p] first virtual workgroup of problem p (p = 0 .. count).  Without it the grid must cover
//   the worst case (every pixel row listed), and a workgroup that only finds out that it has nothing to do still occupies one of the two
//   64 KB LDS slots of a CU for ~2 us.
constexpr int kVtabInts = 1 + (kMaxProblems + 1);

template <bool MIRROR, bool GENERIC, bool SCATTER, int WAVES, int BK = 32, int MAXTN = kMaxTN>
__global__ void __launch_bounds__(64 * WAVES, WAVES == 4 ? (BK == 16 ? 3 : SSDK_CONV_WAVES) : 1) igemm_dma_kernel(ConvGroup grp) {
    if (SCATTER && grp.vtab) {
        const int* vt = grp.vtab;
        const int total = vt[0];
        for (int vb = blockIdx.x; vb < total; vb += gridDim.x) {
            int pi = 0;
#pragma unroll 1
            for (int i = 1; i < grp.count; ++i)
                if (vb >= vt[1 + i]) pi = i;
            const ConvProblem& g = grp.p[pi];
            const int local = vb - vt[1 + pi];
            const int n_block = local % g.n_blocks, t = local / g.n_blocks;
            dma_tile<MIRROR, GENERIC, SCATTER, WAVES, BK, MAXTN>(g, pi, t, n_block, 0);
            __syncthreads();   // the next tile's first DMA overwrites LDS stage 0
        }
        return;
    }
    int pi = 0;
#pragma unroll 1
    for (int i = 1; i < grp.count; ++i)
        if ((int)blockIdx.x >= grp.p[i].block_begin) pi = i;
    const ConvProblem& g = grp.p[pi];

    if (g.mode && *g.mode != g.want_mode) return;
    const int id_all = blockIdx.x - g.block_begin;
    const int ksp = id_all % g.k_splits;
    const int id = id_all / g.k_splits;
    const int per_chunk = 8 * g.n_blocks;   // (both tilings pad the M tiles to a multiple of 8)
    const int chunk = id / per_chunk, within = id % per_chunk;
    const int m_tile = chunk * 8 + (within & 7);
    const int n_block = within >> 3;
    if (m_tile >= (WAVES == 4 ? g.m_tiles : g.m_tiles256)) return;
    dma_tile<MIRROR, GENERIC, SCATTER, WAVES, BK, MAXTN>(g, pi, m_tile, n_block, ksp);
}

// Stream-K form of igemm_dma_kernel<MIRROR, false, false, 4> (see StreamK): `nwg` persistent workgroups, each walks its range of units.
// MIRROR = true (round 5): the stride-1 backward-data launches -- the RetinaNet towers' grouped data gradient is 2 664 tiles on 512 slots
// like its forward launch.
// Order inside a problem: column block major, M tile minor, K slice innermost; and the workgroups of one XCD (round-robin placement: equal
// blockIdx % 8) take NEIGHBOURING ranges, so that at any moment an XCD works inside one or two column blocks: their weight rows
// (2.3 MB per block of the 37 x 37 level) stay in its 4 MB L2 -- with M-tile-major order and ranges dealt out in blockIdx order every
// XCD needed all of a level's weights at once and FETCH_SIZE tripled.
template <bool MIRROR>
__global__ void __launch_bounds__(kConvThreads, SSDK_CONV_WAVES) igemm_streamk_kernel(ConvGroup grp, StreamK sk) {
    const int s = (sk.nwg % 8 == 0) ? (int)(blockIdx.x % 8) * (sk.nwg / 8) + (int)(blockIdx.x / 8) : (int)blockIdx.x;
    // boundary b of the range split, snapped down to a whole K slice of the tile it falls in (both neighbours compute the same value)
    auto locate = [&](long long u, int& pi, int& m_tile, int& n_block, int& slice, int& tn) -> long long {
        pi = 0;
#pragma unroll 1
        for (int i = 1; i < grp.count; ++i)
            if (u >= sk.unit_begin[i]) pi = i;
        const ConvProblem& g = grp.p[pi];
        const int slices = g.ksize * g.ksize * (g.Cc / kBK);
        long long local = u - sk.unit_begin[pi];
        const int base_t = g.tiles_n / g.n_blocks, rem_t = g.tiles_n % g.n_blocks;
        int nb = 0;
        long long acc_u = 0;
        // `tn` here = the weight of a K slice of the column block in units of half a tile (see StreamK::unit_begin)
        auto weight = [&](int b) { return 2 * (base_t + (b < rem_t ? 1 : 0)) - ((g.half_last && b == g.n_blocks - 1) ? 1 : 0); };
        tn = weight(0);
        while (nb + 1 < g.n_blocks && local >= acc_u + (long long)g.m_tiles * slices * tn) {
            acc_u += (long long)g.m_tiles * slices * tn;
            ++nb;
            tn = weight(nb);
        }
        n_block = nb;
        local -= acc_u;
        m_tile = (int)(local / ((long long)slices * tn));
        slice = (int)(local - (long long)m_tile * slices * tn) / tn;
        return sk.unit_begin[pi] + acc_u + ((long long)m_tile * slices + slice) * tn;   // snapped position
    };
    int pi, m_tile, n_block, slice, tn;
    long long u = s == 0 ? 0 : locate(sk.total_units * s / sk.nwg, pi, m_tile, n_block, slice, tn);
    int pj, mj, nj, sj, tj;
    const long long u_end = s + 1 == sk.nwg ? sk.total_units : locate(sk.total_units * (s + 1) / sk.nwg, pj, mj, nj, sj, tj);
    float* const my_buf = sk.partial + (size_t)s * (4 * kMaxTN * 4 * 64 * 4);
    float* const next_buf = my_buf + (size_t)(4 * kMaxTN * 4 * 64 * 4);
    // sticky: a fix-up wait of an earlier launch on this workspace ran out (see dma_tile) -- HIP-graph replays never pass through
    // ssdk_heads_fwd's host-side check, so the kernel itself refuses to produce numbers from then on
    const bool poisoned = __hip_atomic_load(sk.timeouts, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
    const bool drop = g_sk_drop_wg == s;
    while (u < u_end) {
        locate(u, pi, m_tile, n_block, slice, tn);
        const ConvProblem& g = grp.p[pi];
        const int slices = g.ksize * g.ksize * (g.Cc / kBK);
        const long long left = (u_end - u) / tn;
        const int s1 = (int)min((long long)slices, (long long)slice + left);
        if (s1 <= slice) break;   // (cannot happen: boundaries are whole slices; a guard against walking on the spot)
        if (slice == 0 && s1 == slices) {
            dma_tile<MIRROR, false, false, 4, kBK, kMaxTN>(g, pi, m_tile, n_block, 0, 0, 0, 0, nullptr, nullptr, 0u, nullptr, 1, nullptr, poisoned);
        } else if (slice > 0) {   // second K part of a tile whose first part closes the previous workgroup's range: park the partial sums
            dma_tile<MIRROR, false, false, 4, kBK, kMaxTN>(g, pi, m_tile, n_block, 0, 1, slice, s1, my_buf, sk.flags + s, sk.epoch, sk.timeouts, 1, nullptr,
                                                          false, drop);
        } else {                  // first K part: the rest was computed by the following workgroup(s) right after launch
            const long long tile_end = u + (long long)slices * tn;
            int parts = 1;
            while (s + 1 + parts < sk.nwg && sk.total_units * (s + 1 + parts) / sk.nwg < tile_end) ++parts;
            dma_tile<MIRROR, false, false, 4, kBK, kMaxTN>(g, pi, m_tile, n_block, 0, 2, 0, s1, next_buf, sk.flags + s + 1, sk.epoch, sk.timeouts,
                                                          parts, sk.host_err, poisoned);
        }
        u += (long long)(s1 - slice) * tn;
        __syncthreads();   // the next tile's first DMA overwrites LDS stage 0
    }
}

// ---- forward / backward-data ------------------------------------------------------------------------------------
// MIRROR = false: forward convolution; MIRROR = true: backward-data (separate instantiations so that profiles list the
// forward GEMMs and the dgrad GEMMs as different kernels).
// STRIDED (backward-data of a strided convolution only): the source pixel of a tap is (y + pad - k) / stride when
// divisible, so the tap offset is no longer uniform over the rows; it is recomputed per row and slice.
template <int VEC, bool MIRROR, bool STRIDED = false, bool GENERIC = false, bool SCATTER = false>
__global__ void __launch_bounds__(kConvThreads, SSDK_CONV_WAVES) igemm_fwd_kernel(ConvGroup grp) {
    typedef typename VecT<VEC>::type vec_t;
    constexpr int kVecPerRow = kBK / VEC;  // vector loads per 32-float row slice
    constexpr int kRowsPerPass = kConvThreads / kVecPerRow;
    constexpr int kAPasses = kBM / kRowsPerPass;
    constexpr int kBPasses = kMaxTN * 32 / kRowsPerPass;

    // two LDS stages of the A and W slices (72 KB per workgroup, two workgroups per CU fit the 160 KB)
    __shared__ __attribute__((aligned(16))) float s_a[2][kBM * kLdsStride];
    __shared__ __attribute__((aligned(16))) float s_b[2][kMaxTN * 32 * kLdsStride];

    int pi = 0;
#pragma unroll 1
    for (int i = 1; i < grp.count; ++i)
        if ((int)blockIdx.x >= grp.p[i].block_begin) pi = i;
    const ConvProblem& g = grp.p[pi];

    // workgroups that share an M tile get the same (id % 8) inside the problem (same XCD under round-robin placement
    // when block_begin % 8 == 0: they re-read the same activation rows from one L2).  Speed only.
    if (g.mode && *g.mode != g.want_mode) return;
    const int id_all = blockIdx.x - g.block_begin;
    const int ksp = id_all % g.k_splits;
    const int id = id_all / g.k_splits;
    const int per_chunk = 8 * g.n_blocks;
    const int chunk = id / per_chunk, within = id % per_chunk;
    const int m_tile = chunk * 8 + (within & 7);
    const int n_block = within >> 3;
    if (m_tile >= g.m_tiles) return;

    const int Cc = g.Cc;
    const int taps = SCATTER ? 1 : g.ksize * g.ksize;   // scatter: the taps live in N, K is just the channels
    const int chunks = (Cc + kBK - 1) / kBK;
    const int n_slices_all = taps * chunks;
    const int per_split = (n_slices_all + g.k_splits - 1) / g.k_splits;
    const int slice_begin = ksp * per_split;
    const int n_slices = min(n_slices_all, slice_begin + per_split);   // end of this workgroup's slice range
    if (slice_begin >= n_slices) return;
    const long long K = (long long)taps * Cc;
    const int N = g.n0 + g.n1;
    const int hw = g.Hout * g.Wout;
    const int M = (SCATTER && g.row_list) ? *g.row_count : g.B * hw;
    if (SCATTER && m_tile * kBM >= M) return;

    const int base_t = g.tiles_n / g.n_blocks, rem_t = g.tiles_n % g.n_blocks;
    const int tn = base_t + (n_block < rem_t ? 1 : 0);
    const int n_begin = (n_block * base_t + min(n_block, rem_t)) * 32;

    const int tid = threadIdx.x;
    const int lrow = tid / kVecPerRow, lcol = (tid % kVecPerRow) * VEC;

    // per-thread A rows: 32-bit element offset of the row's base pixel and a 9-bit validity mask per tap
    int a_off[kAPasses];
    unsigned a_mask[kAPasses];
    int a_yx[kAPasses];  // STRIDED only: (y + pad) << 16 | (x + pad)
#pragma unroll
    for (int p = 0; p < kAPasses; ++p) {
        const int m = m_tile * kBM + lrow + p * kRowsPerPass;
        a_off[p] = 0;
        a_mask[p] = 0;
        a_yx[p] = 0;
        if (SCATTER) {
            if (m < M) {
                a_mask[p] = 1u;
                a_off[p] = (g.row_list ? g.row_list[m] : m) * g.a_pstride;
            }
        } else if (m < M) {
            const int b = m / hw, r = m % hw;
            const int y = r / g.Wout, x = r % g.Wout;
            unsigned mask = 0;
            int by, bx;
            if (!MIRROR) {  // input pixel = out*stride - pad + k
                by = y * g.stride - g.pad;
                bx = x * g.stride - g.pad;
                for (int t = 0; t < taps; ++t) {
                    const int iy = by + t / g.ksize, ix = bx + t % g.ksize;
                    if (iy >= 0 && iy < g.Hin && ix >= 0 && ix < g.Win) mask |= 1u << t;
                }
            } else {  // backward-data: rows are output-gradient pixels (y + pad - k) / stride
                by = y + g.pad;
                bx = x + g.pad;
                for (int t = 0; t < taps; ++t) {
                    const int ty = by - t / g.ksize, tx = bx - t % g.ksize;
                    if (ty >= 0 && tx >= 0 && ty % g.stride == 0 && tx % g.stride == 0 && ty / g.stride < g.Hin && tx / g.stride < g.Win)
                        mask |= 1u << t;
                }
            }
            a_mask[p] = mask;
            a_yx[p] = (by << 16) | bx;
            a_off[p] = STRIDED ? b * (int)g.a_bstride : b * (int)g.a_bstride + (by * g.Win + bx) * g.a_pstride;
        }
    }
    // per-thread W rows: 32-bit element offset inside the row's segment; bit 30 set = second segment (w1); -1 = no row
    int w_off[kBPasses];
#pragma unroll
    for (int p = 0; p < kBPasses; ++p) {
        const int rr = lrow + p * kRowsPerPass;
        const int n = n_begin + rr;
        w_off[p] = (rr < tn * 32 && n < N) ? (n < g.n0 ? n * (int)K : ((n - g.n0) * (int)K) | (1 << 30)) : -1;
    }

    // Prefetch of one K slice into registers.  Branch-free: a lane whose element does not exist reads element 0 of the
    // operand instead (always mapped) and the value is replaced by zero when it is written to LDS, so the eight loads
    // issue back to back and are all in flight during the MFMAs of the current slice.
    typedef typename GVecT<VEC>::type gvec_t;
    const gptr_t a_base = uniform_ptr(g.a);
    const gptr_t w0_base = uniform_ptr(g.w0);
    const gptr_t w1_base = uniform_ptr(g.w1 ? g.w1 : g.w0);
    const int win_ps = g.Win * g.a_pstride, a_ps = g.a_pstride, ks = g.ksize, strd = g.stride;
    vec_t ra[kAPasses], rb[kBPasses];
    unsigned live = 0;
    // slice -> (channel chunk, tap): chunk outer, tap inner -- the nine taps of one chunk re-read (shifted) pixel rows that
    // are still in L1/L2.  The position of the NEXT slice to load is kept in scalar counters (no divisions in the loop).
    int ld_tap = slice_begin % taps, ld_chunk = slice_begin / taps;
    int ld_ky = ld_tap / ks, ld_kx = ld_tap % ks;
    auto load_slice = [&]() {
        const int tap = ld_tap, c = ld_chunk * kBK + lcol;
        const int ky = ld_ky, kx = ld_kx;
        const int tap_off = MIRROR ? -(ky * win_ps + kx * a_ps) : (ky * win_ps + kx * a_ps);
        const bool c_ok = c < Cc;
        live = 0;
#pragma unroll
        for (int p = 0; p < kAPasses; ++p) {
            const bool ok = c_ok && ((a_mask[p] >> tap) & 1u);
            int off;
            if (STRIDED) {
                const int iy = ((a_yx[p] >> 16) - ky) / strd, ix = ((a_yx[p] & 0xFFFF) - kx) / strd;
                off = a_off[p] + iy * win_ps + ix * a_ps + c;
            } else {
                off = a_off[p] + tap_off + c;
            }
            ra[p] = *(gvec_t)(a_base + (size_t)(unsigned)(ok ? off : 0));
            live |= (ok ? 1u : 0u) << p;
        }
        const int wk = tap * Cc + c;
#pragma unroll
        for (int p = 0; p < kBPasses; ++p) {
            const bool ok = c_ok && w_off[p] >= 0;
            const unsigned off = ok ? (unsigned)((w_off[p] & ~(1 << 30)) + wk) : 0u;
            // two loads would double the traffic; the segment is a per-row constant, so select the (uniform) base per lane
            const gptr_t base = (w_off[p] & (1 << 30)) ? w1_base : w0_base;
            rb[p] = *(gvec_t)(base + (size_t)off);
            live |= (ok ? 1u : 0u) << (kAPasses + p);
        }
        // advance to the next slice
        // (selects, not branches: the K loop body must stay one basic block)
        ++ld_tap;
        ++ld_kx;
        const bool wrap_x = ld_kx == ks, wrap_t = ld_tap == taps;
        ld_kx = (wrap_x || wrap_t) ? 0 : ld_kx;   // (scatter form: taps == 1 although ksize > 1)
        ld_ky = wrap_t ? 0 : ld_ky + (wrap_x ? 1 : 0);
        ld_tap = wrap_t ? 0 : ld_tap;
        ld_chunk += wrap_t ? 1 : 0;
    };
    auto store_slice = [&](int stage) {
#pragma unroll
        for (int p = 0; p < kAPasses; ++p)
            *reinterpret_cast<vec_t*>(&s_a[stage][(lrow + p * kRowsPerPass) * kLdsStride + lcol]) = ((live >> p) & 1u) ? ra[p] : vzero<VEC>();
#pragma unroll
        for (int p = 0; p < kBPasses; ++p)
            *reinterpret_cast<vec_t*>(&s_b[stage][(lrow + p * kRowsPerPass) * kLdsStride + lcol]) = ((live >> (kAPasses + p)) & 1u) ? rb[p] : vzero<VEC>();
    };

    f32x16 acc[kMaxTN];
#pragma unroll
    for (int j = 0; j < kMaxTN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.0f;

    const int lane = tid & 63, wave = tid >> 6;
    const int r32 = lane & 31, h = lane >> 5;
    const float* a_rd = &s_a[0][0] + (wave * 32 + r32) * kLdsStride + 4 * h;
    const float* b_rd = &s_b[0][0] + r32 * kLdsStride + 4 * h;

    // Pipeline (one barrier per slice): while slice s is multiplied out of LDS stage s&1, the registers holding slice s+1
    // (loaded during slice s-1) are written to the other stage and re-issued for slice s+2 -- the LDS writes, the address
    // arithmetic and the global loads all sit between MFMAs instead of in a bubble between two barriers.
    load_slice();
    store_slice(0);
    load_slice();
    __syncthreads();
    // The K loop is specialised on the number of column tiles (dispatched once per workgroup): with a run-time bound
    // every MFMA would sit behind its own scalar branch.
    auto k_loop = [&](auto tn_c) {
        constexpr int TN = decltype(tn_c)::value;
        int stage = 0;
#ifdef SSDK_CONV_TRACE
        const bool trace_on = pi == 0 && (blockIdx.x % 31) == 0 && blockIdx.x / 31 < 32;
        const int trace_blk = blockIdx.x / 31;
        if (trace_on && (tid & 63) == 0)   // slot 3 of slice 0: where the wave runs (HW_ID, XCC_ID) and what it computes
            g_trace[((trace_blk * 4 + wave) * 64) * 8 + 7] = (unsigned long long)__builtin_amdgcn_s_getreg((15 << 11) | 4) |
                                                             ((unsigned long long)__builtin_amdgcn_s_getreg((3 << 11) | 20) << 32) |
                                                             ((unsigned long long)TN << 40) | ((unsigned long long)m_tile << 44);
#endif
        for (int slice = slice_begin; slice < n_slices; ++slice) {
            TRACE(0)
            const float* a_cur = a_rd + stage * (kBM * kLdsStride);
            const float* b_cur = b_rd + stage * (kMaxTN * 32 * kLdsStride);
#pragma unroll
            for (int gk = 0; gk < kBK / 8; ++gk) {
                // lane (r, h) reads k = gk*8 + 4h .. +3 of its row: MFMA kk pairs k = gk*8+kk (h=0) with gk*8+4+kk (h=1)
                const f32x4 av = *reinterpret_cast<const f32x4*>(a_cur + gk * 8);
                f32x4 bv[TN];
#pragma unroll
                for (int j = 0; j < TN; ++j) bv[j] = *reinterpret_cast<const f32x4*>(b_cur + j * 32 * kLdsStride + gk * 8);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk], bv[j][kk], acc[j], 0, 0, 0);
                // unconditional (straight-line code the scheduler can spread between the MFMAs): past the last slice the
                // loads degenerate to element 0 of the operands (c >= Cc) and the stage written is never read
                if (gk == 0) store_slice(stage ^ 1);
#ifdef SSDK_CONV_TRACE
                if (gk == 0) { TRACE(1) }
                if (gk == 2) { TRACE(3) }
                if (gk == 3) { TRACE(4) }
#endif
                if (gk == 1) {
                    load_slice();
                    // keep the prefetch HERE: left alone the scheduler sinks the loads to the end of the slice (it reuses the
                    // staging registers for fragment reads), which exposes their latency at the top of the next slice
                    __builtin_amdgcn_sched_barrier(0);
                    TRACE(2)
                }
            }
            __syncthreads();
            TRACE(5)
            stage ^= 1;
        }
    };
    switch (tn) {
        case 4: k_loop(std::integral_constant<int, 4>{}); break;
        case 3: k_loop(std::integral_constant<int, 3>{}); break;
        case 2: k_loop(std::integral_constant<int, 2>{}); break;
        default: k_loop(std::integral_constant<int, 1>{}); break;
    }

    conv_epilogue<SCATTER>(g, acc, m_tile * kBM, wave, r32, h, tn, n_begin, M, N, hw, ksp, (!MIRROR && !SCATTER) ? &s_a[0][0] : nullptr, 4);
}

// ---- backward-weights ------------------------------------------------------------------------------------------
struct WgradProblem {
    const float* dy;  // packed [M][Npad]
    const float* x;   // [B][Hin*Win][Cc]
    int Npad, Cc;
    int B, Hout, Wout, Hin, Win, ksize, stride, pad;
    float* dw0;       // [n0][taps*Cc] (+=)
    float* dw1;       // [n1][taps*Cc] (+=)
    int n0, n1;
    int k_splits, n_tiles, c_tiles32, c_blocks;
    int block_begin;
    const int* mode;       // no-op unless *mode == want_mode (NULL: always run)
    int want_mode;
    const int* row_list;   // sparse: contract only over these pixel ids (*row_count of them)
    const int* row_count;
    // segmented (the anchor rows of the ordered pipeline: segment = anchor type): `segs` independent problems that share x: segment s has seg_count[s] rows of dy at
    // dy + s * seg_cap * Npad, their pixel ids at row_list + s * seg_cap, and adds into dw0 + s * dw0_seg / dw1 + s * dw1_seg
    const int* seg_count;
    int segs, seg_cap;
    long long dw0_seg, dw1_seg;
    unsigned dy_bytes, x_bytes;   // LDS-DMA kernel: buffer descriptor sizes (dy: one segment when segmented)
    // deterministic mode: != 0 -> K split `ksp` STORES its partial tile into copy ksp of a [N][taps*Cc] image (dw0 = the first copy,
    // dw1 = dw0 + n0 * taps * Cc, consecutive copies det_stride floats apart) instead of adding into dw with atomics;
    // reduce_partials_kernel then adds the copies in split order
    long long det_stride;
    // != 0: the workgroup STORES its tile (into copy ksp when det_stride != 0, into dw itself when k_splits == 1) instead of adding it with
    // atomics, and a split whose row range is empty stores zeros: the outputs need no zero-fill and every copy is always complete
    int ordered;
};
struct WgradGroup {
    int count;
    int total_blocks;
    WgradProblem p[kMaxProblems];
};

// LDS: dY slice [32 pixels][128 n] and X slice [32 pixels][128 c]; MFMA A operand = dY^T, B operand = X.
__global__ void __launch_bounds__(kConvThreads, SSDK_CONV_WAVES) igemm_wgrad_kernel(WgradGroup grp) {
    __shared__ __attribute__((aligned(16))) float s_dy[32 * 128];
    __shared__ __attribute__((aligned(16))) float s_x[32 * 128];

    int pi = 0;
#pragma unroll 1
    for (int i = 1; i < grp.count; ++i)
        if ((int)blockIdx.x >= grp.p[i].block_begin) pi = i;
    const WgradProblem& g = grp.p[pi];

    if (g.mode && *g.mode != g.want_mode) return;
    const int Cc = g.Cc;
    const int N = g.n0 + g.n1;
    const int taps = g.ksize * g.ksize;
    int id = blockIdx.x - g.block_begin;
    const int ksp = id % g.k_splits; id /= g.k_splits;
    const int cb = id % g.c_blocks; id /= g.c_blocks;
    const int nt = id % g.n_tiles; id /= g.n_tiles;
    const int tap = id % taps;
    const int seg = id / taps;
    const int ky = tap / g.ksize, kx = tap % g.ksize;
    const float* dy_p = g.dy;
    const int* rows_p = g.row_list;
    float* dw0_p = g.dw0;
    float* dw1_p = g.dw1;
    if (g.seg_count) {
        dy_p += (long long)seg * g.seg_cap * g.Npad;
        rows_p += (long long)seg * g.seg_cap;
        dw0_p += (long long)seg * g.dw0_seg;
        if (dw1_p) dw1_p += (long long)seg * g.dw1_seg;
    }

    const int base_t = g.c_tiles32 / g.c_blocks, rem_t = g.c_tiles32 % g.c_blocks;
    const int tn = base_t + (cb < rem_t ? 1 : 0);
    const int c_begin = (cb * base_t + min(cb, rem_t)) * 32;
    const int n_begin = nt * 128;

    const int hw = g.Hout * g.Wout;
    const int M = g.seg_count ? g.seg_count[seg] : (g.row_list ? *g.row_count : g.B * hw);   // rows of the contraction (compacted when sparse)
    const int slices_total = (M + 31) / 32;
    const int per = (slices_total + g.k_splits - 1) / g.k_splits;
    const int s_begin = ksp * per, s_end = min(slices_total, s_begin + per);
    if (s_begin >= s_end) return;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r32 = lane & 31, h = lane >> 5;
    const bool wave_live = n_begin + wave * 32 < N;

    f32x16 acc[kMaxTN];
#pragma unroll
    for (int j = 0; j < kMaxTN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.0f;

    // staging: 32 rows x 32 float4 per operand = 1024 float4 -> 4 per thread; thread t: row = t/32 + 8p, col4 = t%32
    const int srow = tid >> 5, scol = (tid & 31) * 4;
    float4 rdy[4], rx[4];
    auto load_slice = [&](int s) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int mi = s * 32 + srow + p * 8;
            float4 vd = make_float4(0.f, 0.f, 0.f, 0.f), vx = vd;
            if (mi < M) {
                const int m = rows_p ? rows_p[mi] : mi;   // pixel id of row mi
                // (segmented: the dy rows are stored compacted, row mi of the segment; otherwise dy is indexed by pixel)
                if (n_begin + scol < g.Npad) vd = *reinterpret_cast<const float4*>(dy_p + (long long)(g.seg_count ? mi : m) * g.Npad + n_begin + scol);
                const int c = c_begin + scol;
                if (scol < tn * 32 && c < Cc) {
                    const int b = m / hw, pix = m % hw;
                    const int iy = (pix / g.Wout) * g.stride - g.pad + ky, ix = (pix % g.Wout) * g.stride - g.pad + kx;
                    if (iy >= 0 && iy < g.Hin && ix >= 0 && ix < g.Win)
                        vx = *reinterpret_cast<const float4*>(g.x + ((long long)b * g.Hin * g.Win + (long long)iy * g.Win + ix) * Cc + c);
                }
            }
            rdy[p] = vd;
            rx[p] = vx;
        }
    };
    auto store_slice = [&]() {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            *reinterpret_cast<float4*>(&s_dy[(srow + p * 8) * 128 + scol]) = rdy[p];
            *reinterpret_cast<float4*>(&s_x[(srow + p * 8) * 128 + scol]) = rx[p];
        }
    };

    load_slice(s_begin);
    store_slice();
    __syncthreads();
    auto k_loop = [&](auto tn_c) {   // specialised on the number of column tiles: no branch per MFMA
        constexpr int TN = decltype(tn_c)::value;
        for (int s = s_begin; s < s_end; ++s) {
            if (s + 1 < s_end) load_slice(s + 1);
            if (wave_live) {
#pragma unroll 4
                for (int k2 = 0; k2 < 32; k2 += 2) {
                    const float av = s_dy[(k2 + h) * 128 + wave * 32 + r32];
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, s_x[(k2 + h) * 128 + j * 32 + r32], acc[j], 0, 0, 0);
                }
            }
            __syncthreads();
            if (s + 1 < s_end) {
                store_slice();
                __syncthreads();
            }
        }
    };
    switch (tn) {
        case 4: k_loop(std::integral_constant<int, 4>{}); break;
        case 3: k_loop(std::integral_constant<int, 3>{}); break;
        case 2: k_loop(std::integral_constant<int, 2>{}); break;
        default: k_loop(std::integral_constant<int, 1>{}); break;
    }
    if (!wave_live) return;
    const long long K = (long long)taps * Cc;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int n = n_begin + wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (n >= N) continue;
        float* row = (n < g.n0 ? dw0_p + (long long)n * K : dw1_p + (long long)(n - g.n0) * K) + (long long)ksp * g.det_stride;
#pragma unroll
        for (int j = 0; j < kMaxTN; ++j) {
            if (j >= tn) continue;
            const int c = c_begin + j * 32 + r32;
            if (c < Cc) {
                if (g.det_stride) row[(long long)tap * Cc + c] = acc[j][e];
                else atomicAdd(row + (long long)tap * Cc + c, acc[j][e]);
            }
        }
    }
}

// ---- backward-weights, LDS-DMA staging ----------------------------------------------------------------------------------
// dW[n][tap][c] += sum over pixels m of dY[m][n] * X[m shifted by tap][c].  Workgroup = one tap, 128 n, up to 128 c, a range
// of 32-pixel slices (split-K, fp32 atomics at the end).  Both operands are staged as they lie in memory, [pixel][channel]:
// a DMA piece (64 lanes x 16 B) is 2 pixels x 512 B, 32 pieces per slice, no swizzle needed because the fragment reads walk
// along a pixel row.  MFMA A operand = dY^T, B operand = X:
//   * lane (r, h) reads dY[pixel 2s + h][4r .. 4r + 3] with ONE ds_read_b128 and uses the four values as the A operands of four
//     MFMAs -- MFMA q then produces the output rows n = 4 i + q (a permutation undone in the epilogue) -- and X[pixel][32 w + r]
//     with one ds_read_b32: 2 LDS instructions per 4 MFMAs (the first wgrad kernel issued 5 ds_read_b32 per 4 MFMAs);
//   * wave w owns output columns c = 32 w .. 32 w + 31 of the workgroup's c block and all 128 n (64 accumulator registers);
//   * the per-pixel source offsets (row gather for the sparse forms, image / y / x decomposition, border test for this tap) are
//     computed by lanes 0..7 of each wave -- one pixel each, two slices ahead -- and handed to the piece lanes with ds_bpermute.
__device__ __forceinline__ void fast_divmod(int m, int d, float rcp, int& q, int& r) {   // exact for 0 <= m < 2^24
    q = (int)((float)m * rcp);
    r = m - q * d;
    if (r < 0) { --q; r += d; }
    if (r >= d) { ++q; r -= d; }
}

typedef __bf16 wg_bf16x8 __attribute__((ext_vector_type(8)));
// FAST = the opt-in split-bf16 mode (ssdk_heads_bwd_fast / ssdk_conv2d_bwd_fast): both operands are split into bf16 pieces in registers
// after the LDS reads and every fp32 product becomes a_hi b_hi + a_hi b_mid + a_mid b_hi on v_mfma_f32_32x32x16_bf16 (fp32 accumulate):
// 24 MFMAs of 8 passes per 32-pixel slice instead of 64 of 16.  Staging, tiling, K split and epilogue are the fp32 kernel's.
// ROW32 (anchor rows: Npad = Jpad is 32 .. 128): MFMA q owns output rows n_begin + 32 q .. + 31 -- lane (r, h) reads dY[pixel][r + 32 q] with a
// ds_read_b32 per q -- and only the ceil(rows / 32) MFMAs that have rows are issued (Jpad = 96: 3 of 4; 21-class heads, Jpad = 32: 1 of 4).
// The default map (MFMA q owns rows 4 i + q: one ds_read_b128 feeds four MFMAs) always issues four.
template <bool FAST, bool ROW32 = false, int NQ = 4>   // NQ (ROW32 only): MFMAs per K step that have output rows = Npad / 32, the same for every problem of the launch
__device__ __forceinline__ void wgrad_dma_body(const WgradGroup& grp) {
    __shared__ __attribute__((aligned(1024))) float s_dy0[32 * 128];
    __shared__ __attribute__((aligned(1024))) float s_dy1[32 * 128];
    __shared__ __attribute__((aligned(1024))) float s_x0[32 * 128];
    __shared__ __attribute__((aligned(1024))) float s_x1[32 * 128];

    int pi = 0;
#pragma unroll 1
    for (int i = 1; i < grp.count; ++i)
        if ((int)blockIdx.x >= grp.p[i].block_begin) pi = i;
    const WgradProblem& g = grp.p[pi];
    if (g.mode && *g.mode != g.want_mode) return;
    const int Cc = g.Cc;
    const int N = g.n0 + g.n1;
    const int taps = g.ksize * g.ksize;
    int id = blockIdx.x - g.block_begin;
    const int ksp = id % g.k_splits; id /= g.k_splits;
    const int cb = id % g.c_blocks; id /= g.c_blocks;
    const int nt = id % g.n_tiles; id /= g.n_tiles;
    const int tap = id % taps;
    const int seg = id / taps;
    const int ky = tap / g.ksize, kx = tap % g.ksize;
    const float* dy_p = g.dy;
    const int* rows_p = g.row_list;
    float* dw0_p = g.dw0;
    float* dw1_p = g.dw1;
    if (g.seg_count) {
        dy_p += (long long)seg * g.seg_cap * g.Npad;
        rows_p += (long long)seg * g.seg_cap;
        dw0_p += (long long)seg * g.dw0_seg;
        if (dw1_p) dw1_p += (long long)seg * g.dw1_seg;
    }
    const int base_t = g.c_tiles32 / g.c_blocks, rem_t = g.c_tiles32 % g.c_blocks;
    const int tn = base_t + (cb < rem_t ? 1 : 0);
    const int c_begin = (cb * base_t + min(cb, rem_t)) * 32;
    const int n_begin = nt * 128;
    const int hw = g.Hout * g.Wout;
    const int M = g.seg_count ? g.seg_count[seg] : (g.row_list ? *g.row_count : g.B * hw);
    const int slices_total = (M + 31) / 32;
    const int per = (slices_total + g.k_splits - 1) / g.k_splits;
    const int s_begin = ksp * per, s_end = min(slices_total, s_begin + per);
    if (s_begin >= s_end && !g.ordered) return;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r32 = lane & 31, h = lane >> 5;
    const bool wave_live = wave < tn;

    const __amdgpu_buffer_rsrc_t rsrc_dy = __builtin_amdgcn_make_buffer_rsrc((void*)uniform_ptr(dy_p), 0, (int)g.dy_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc((void*)uniform_ptr(g.x), 0, (int)g.x_bytes, 0x00020000);
    // column part of this lane's 16 bytes inside the 512-byte row window of a piece (bit 31 = outside the operand: zeros)
    const int col4 = (lane & 31) * 4;
    const unsigned dy_col = (n_begin + col4 < g.Npad) ? (unsigned)(n_begin + col4) * 4u : kOobBit;
    const unsigned x_col = (col4 < tn * 32 && c_begin + col4 < Cc) ? (unsigned)(c_begin + col4) * 4u : kOobBit;
    const float rcp_hw = 1.0f / (float)hw, rcp_w = 1.0f / (float)g.Wout;
    const int dy_row_bytes = g.Npad * 4, x_pix_bytes = Cc * 4;

    // pixel lanes: lane l < 8 of wave w owns pixel 8 w + l of a slice
    auto load_id = [&](int sl) -> int {      // pixel id of my pixel in slice sl (only meaningful on the pixel lanes)
        const int mi = sl * 32 + wave * 8 + (lane & 7);
        return (rows_p && sl < s_end && mi < M) ? rows_p[mi] : mi;
    };
    auto pixel_offsets = [&](int sl, int m, unsigned& dyo, unsigned& xo) {
        const int mi = sl * 32 + wave * 8 + (lane & 7);
        dyo = kOobBit;
        xo = kOobBit;
        if (sl < s_end && mi < M) {
            dyo = (unsigned)(g.seg_count ? mi : m) * (unsigned)dy_row_bytes;
            int b, pix, y, x;
            fast_divmod(m, hw, rcp_hw, b, pix);
            fast_divmod(pix, g.Wout, rcp_w, y, x);
            const int iy = y * g.stride - g.pad + ky, ix = x * g.stride - g.pad + kx;
            if (iy >= 0 && iy < g.Hin && ix >= 0 && ix < g.Win) xo = (unsigned)((b * g.Hin + iy) * g.Win + ix) * (unsigned)x_pix_bytes;
        }
    };
    auto issue = [&](int stage, unsigned dyo, unsigned xo) {   // the wave's 4 dY pieces and 4 X pieces of one slice
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int src = 2 * i + (lane >> 5);
            const unsigned d = (unsigned)__shfl((int)dyo, src, kWave), x = (unsigned)__shfl((int)xo, src, kWave);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_dy, (lds_ptr_t)((stage ? s_dy1 : s_dy0) + (wave * 8 + 2 * i) * 128), 16, (dy_col & kOobBit) ? kOobBit : d + dy_col, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (lds_ptr_t)((stage ? s_x1 : s_x0) + (wave * 8 + 2 * i) * 128), 16, (x_col & kOobBit) ? kOobBit : x + x_col, 0, 0, 0);
        }
    };
    // (row offsets below 2^31 -- checked on the host -- plus a column part < 2^20: an invalid pixel keeps bit 31 set after the add)

    f32x16 acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[q][e] = 0.0f;

    unsigned dyo1, xo1, dyo2, xo2;
    {
        unsigned dyo0, xo0;
        pixel_offsets(s_begin, load_id(s_begin), dyo0, xo0);
        issue(0, dyo0, xo0);
        pixel_offsets(s_begin + 1, load_id(s_begin + 1), dyo1, xo1);
    }
    int id2 = load_id(s_begin + 2);
    __syncthreads();

    auto body = [&](auto st_c, int sl) {
        constexpr int ST = decltype(st_c)::value;
        issue(ST ^ 1, dyo1, xo1);                                  // slice sl + 1 lands while slice sl is multiplied
        pixel_offsets(sl + 2, id2, dyo2, xo2);
        id2 = load_id(sl + 3);
        if (wave_live && !FAST && ROW32) {
            const float* ady = (ST ? s_dy1 : s_dy0) + h * 128 + r32;
            const float* bx = (ST ? s_x1 : s_x0) + h * 128 + wave * 32 + r32;
            float av[2][4], bv[2];
#pragma unroll
            for (int q = 0; q < NQ; ++q) av[0][q] = ady[32 * q];
            bv[0] = bx[0];
#pragma unroll
            for (int k2 = 0; k2 < 16; ++k2) {
                if (k2 + 1 < 16) {
#pragma unroll
                    for (int q = 0; q < NQ; ++q) av[(k2 + 1) & 1][q] = ady[(k2 + 1) * 256 + 32 * q];
                    bv[(k2 + 1) & 1] = bx[(k2 + 1) * 256];
                }
#pragma unroll
                for (int q = 0; q < NQ; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[k2 & 1][q], bv[k2 & 1], acc[q], 0, 0, 0);
            }
        }
        if (wave_live && !FAST && !ROW32) {
            const float* ady = (ST ? s_dy1 : s_dy0) + h * 128 + 4 * r32;
            const float* bx = (ST ? s_x1 : s_x0) + h * 128 + wave * 32 + r32;
            f32x4 av[2];
            float bv[2];
            av[0] = *reinterpret_cast<const f32x4*>(ady);
            bv[0] = bx[0];
#pragma unroll
            for (int k2 = 0; k2 < 16; ++k2) {
                if (k2 + 1 < 16) {
                    av[(k2 + 1) & 1] = *reinterpret_cast<const f32x4*>(ady + (k2 + 1) * 256);
                    bv[(k2 + 1) & 1] = bx[(k2 + 1) * 256];
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[k2 & 1][q], bv[k2 & 1], acc[q], 0, 0, 0);
            }
        }
        if (wave_live && FAST) {
            // lane (r32, h), K step s2: pixels 16 s2 + 8 h .. + 7 of the slice; A_q[row 4 r32 + q][those pixels] from eight 16-byte reads,
            // B[those pixels][column wave * 32 + r32] from eight 4-byte reads
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const float* ady = (ST ? s_dy1 : s_dy0) + (16 * s2 + 8 * h) * 128 + 4 * r32;
                const float* bx = (ST ? s_x1 : s_x0) + (16 * s2 + 8 * h) * 128 + wave * 32 + r32;
                f32x4 av[8];
                float bv[8];
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    av[t] = *reinterpret_cast<const f32x4*>(ady + t * 128);
                    bv[t] = bx[t * 128];
                }
                wg_bf16x8 b_hi, b_mid;
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const __bf16 hb = (__bf16)bv[t];
                    b_hi[t] = hb;
                    b_mid[t] = (__bf16)(bv[t] - (float)hb);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    wg_bf16x8 a_hi, a_mid;
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
                        const __bf16 ha = (__bf16)av[t][q];
                        a_hi[t] = ha;
                        a_mid[t] = (__bf16)(av[t][q] - (float)ha);
                    }
                    acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_mid, b_hi, acc[q], 0, 0, 0);
                    acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, b_mid, acc[q], 0, 0, 0);
                    acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, b_hi, acc[q], 0, 0, 0);
                }
            }
        }
        __syncthreads();
        dyo1 = dyo2;
        xo1 = xo2;
    };
    for (int sl = s_begin; sl < s_end; sl += 2) {
        body(std::integral_constant<int, 0>{}, sl);
        if (sl + 1 < s_end) body(std::integral_constant<int, 1>{}, sl + 1);
    }
    if (!wave_live) return;
    const long long K = (long long)taps * Cc;
    const int c = c_begin + wave * 32 + r32;
    if (c >= Cc) return;
    // (everything but n * K hoisted out of the 64 elements: the two row bases already carry tap, channel and the -n0 shift)
    const int n0 = g.n0;
    float* const base0 = dw0_p + (long long)tap * Cc + c + (long long)ksp * g.det_stride;
    float* const base1 = dw1_p ? dw1_p + (long long)tap * Cc + c - (long long)n0 * K + (long long)ksp * g.det_stride : base0;
    const bool ordered = g.det_stride != 0 || g.ordered;   // this split's own copy (or the output itself), plain stores
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int i = (e & 3) + 8 * (e >> 2) + 4 * h;
        const int nrow = n_begin + (ROW32 ? i : 4 * i);
        const long long nK = (long long)nrow * K;
#pragma unroll
        for (int q = 0; q < (ROW32 ? NQ : 4); ++q) {
            const int n = nrow + (ROW32 ? 32 * q : q);   // MFMA q, row i
            if (n >= N) continue;
            float* const dst = (n < n0 ? base0 : base1) + nK + (long long)(ROW32 ? 32 * q : q) * K;
            if (ordered) *dst = acc[q][e];
            else atomicAdd(dst, acc[q][e]);
        }
    }
}
__global__ void __launch_bounds__(kConvThreads, SSDK_CONV_WAVES) igemm_wgrad_dma_kernel(WgradGroup grp) { wgrad_dma_body<false>(grp); }
__global__ void __launch_bounds__(kConvThreads, SSDK_CONV_WAVES) igemm_wgrad_bf16x3_kernel(WgradGroup grp) { wgrad_dma_body<true>(grp); }
template <int NQ> __global__ void __launch_bounds__(kConvThreads, SSDK_CONV_WAVES) igemm_wgrad_rows_kernel(WgradGroup grp) { wgrad_dma_body<false, true, NQ>(grp); }

// ---- dY pack (+ bias gradient) --------------------------------------------------------------------------------------
// out[m][n] (n < Npad) = n < n0 ? ds[b*sb + p*n0 + n] : (n < n0+n1 ? dl[b*lb + p*n1 + n-n0] : 0);  db += column sums
// Also appends the ids of the rows that are not entirely zero to row_list (order inside a block preserved, blocks in
// completion order) and counts them: the loss gradient is non-zero only on sampled anchors, i.e. on a few % of the pixels.
constexpr int kPackRows = 32;   // rows per workgroup (<= 64: row masks are 64-bit; 64 rows take the same time)
constexpr int kMaxAnchorTypes_ = 16;   // (= kMaxAnchorTypes; the last counter of a level doubles as its "not sparse" flag, so nb <= 15)
constexpr int kPackColIters = 12;  // Npad <= 768 (RetinaNet: 9 * (80 + 4) = 756)
struct PackLevel {
    const float* ds; const float* dl;
    int n0, n1, Npad, HW;
    float* out; float* db0; float* db1;
    int* row_list; int* row_count;
    int block_begin;
    // ordered pipeline (gather_rows_kernel, pack_store_kernel): anchor type k of a pixel owns C score columns [k*C, (k+1)*C) and 4 loc
    // columns; every anchor with a gradient is one row [Jpad] of ga[k]: C scores, 4 locs, zero padding; apix = the rows' pixel ids,
    // acount = rows per type (both written by anchor_plan_kernel / anchor_fill_kernel)
    int nb, C, Jpad, cap;
    int LQ;            // loc columns per anchor: 4, or 0 for a level that is a single head (SharedConvPredictor: score and loc towers apart)
    float* ga; const int* apix; const int* acount;
    const int* mode;   // the level's backward form as anchor_plan_kernel chose it (2 = anchor rows: `out` is not needed; else `ga` is not)
    // gather_rows_kernel STORES the column sums of chunk c of type k into dbp[(k * chunks_cap + c) * Jpad + j] (anchor_dbias_kernel adds
    // them in chunk order)
    float* dbp;
    int chunks_cap;
    // legacy pipeline, caller's guarantee (ssdk_heads_bwd_ex): rmask[b * a_total + a_off + pixel * rnb + k] == 0 -> the score and loc gradient
    // rows of that anchor are entirely zero; a pixel whose rnb anchors are all 0 is not read at all (NULL: every row is read)
    const unsigned char* rmask;
    int a_total, a_off, rnb;
};
struct PackGroup {
    int count, B;
    long long sb, lb;
    PackLevel lv[kMaxProblems];
};
// Main pass of pack_dy_kernel for one wave: rows wave, wave + 4, ... of the kPackRows-row block, ITERS 64-column trips per row.
// Per lane and trip the column's source (score / loc / padding), its offset inside the source row and the bit of its anchor
// type are fixed for the block, so a row costs one select + one add per load and no branches; two rows are in flight.
// (The first version re-derived source and bounds per element behind divergent branches: ~25 instructions per load, 2 TB/s.)
// Returns the mask of this wave's rows that are not entirely zero; leaves the wave's column sums in `sum`.
template <int ITERS, bool STORE, bool COUNT>
__device__ __forceinline__ unsigned long long pack_rows(const PackLevel& L, const float* __restrict__ ds, const float* __restrict__ dl, long long sb,
                                              long long lb, int B, int m0, int M, float* __restrict__ sum) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // (scalar: row bases stay in SGPRs)
    const int n0 = L.n0, n1 = L.n1, N = n0 + n1, Npad = L.Npad, HW = L.HW;
    float acc[ITERS];
#pragma unroll
    for (int k = 0; k < ITERS; ++k) acc[k] = 0.0f;
    unsigned long long mine = 0ull;
    constexpr int U = ITERS <= 8 ? 2 : 1;   // rows of the wave in flight
    for (int r0 = wave; r0 < kPackRows; r0 += 4 * U) {
        if (m0 + r0 >= M) break;
        float v[U][ITERS];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int m = min(m0 + r0 + 4 * u, M - 1);   // (a row past the end re-reads the last one; it is not consumed)
            const int b = m / HW, p = m - b * HW;
            if (L.rmask) {   // the pixel's anchors are all known to carry no gradient: its row is zeros, nothing is loaded (wave-uniform)
                const unsigned char* mp = L.rmask + (long long)b * L.a_total + L.a_off + (long long)p * L.rnb;
                const unsigned char f = lane < L.rnb ? mp[lane] : (unsigned char)0;
                if (!__ballot(f != 0)) {
#pragma unroll
                    for (int k = 0; k < ITERS; ++k) v[u][k] = 0.0f;
                    continue;
                }
            }
            const float* srow = ds + (long long)b * sb + (long long)p * n0;
            const float* lrow = dl ? dl + (long long)b * lb + (long long)p * n1 : srow;
#pragma unroll
            for (int k = 0; k < ITERS; ++k) {
                // branch-free: one load per trip from a selected base (a load behind a branch makes the compiler wait for
                // every earlier load at the join); padding columns re-read element 0 and are zeroed
                const int n = k * 64 + lane;
                const float* base = n >= n0 ? lrow : srow;
                const float x = base[n < n0 ? n : (n < N ? n - n0 : 0)];
                v[u][k] = n < N ? x : 0.0f;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int r = r0 + 4 * u, m = m0 + r;
            if (m >= M) continue;
            float* orow = L.out + (long long)m * Npad;
            bool any = false;   // this lane saw a non-zero value of the row
#pragma unroll
            for (int k = 0; k < ITERS; ++k) {
                const float x = v[u][k];
                if (STORE && k * 64 + lane < Npad) orow[k * 64 + lane] = x;
                if (COUNT) {
                    acc[k] += x;
                    any = any || x != 0.0f;
                }
            }
            if (!COUNT) continue;
            if (__ballot(any)) mine |= 1ull << r;
        }
    }
    if (COUNT) {
#pragma unroll
        for (int k = 0; k < ITERS; ++k) sum[k * 64 + lane] = acc[k];
    }
    return mine;
}

// (legacy pipeline) packs the rows, sums the bias gradients, lists the pixel rows that carry a gradient
template <int ITERS>
__global__ void __launch_bounds__(256) pack_dy_kernel(PackGroup grp) {
    int pi = 0;
#pragma unroll 1
    for (int i = 1; i < grp.count; ++i)
        if ((int)blockIdx.x >= grp.lv[i].block_begin) pi = i;
    const PackLevel& L = grp.lv[pi];
    const float* __restrict__ ds = L.ds;
    const float* __restrict__ dl = L.dl;
    const long long sb = grp.sb, lb = grp.lb;
    const int n0 = L.n0, n1 = L.n1, B = grp.B, HW = L.HW;
    float* __restrict__ db0 = L.db0;
    float* __restrict__ db1 = L.db1;
    int* __restrict__ row_list = L.row_list;
    int* __restrict__ row_count = L.row_count;
    const int block = blockIdx.x - L.block_begin;
    // wave w packs rows w, w+4, ... of the kPackRows-row block; lane l owns columns l, l+64, ... : every load and store of a
    // wave instruction is 256 contiguous bytes
    __shared__ unsigned long long s_flag;
    __shared__ int s_base;
    __shared__ float s_sum[4][kPackColIters * 64];
    const int M = B * HW;
    const int m0 = block * kPackRows;
    const int N = n0 + n1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_flag = 0ull;
    __syncthreads();
    const unsigned long long mine = pack_rows<ITERS, true, true>(L, ds, dl, sb, lb, B, m0, M, s_sum[wave]);
    if (lane == 0 && mine) atomicOr(&s_flag, mine);
    __syncthreads();
    for (int n = threadIdx.x; n < N; n += 256) {
        const float t = s_sum[0][n] + s_sum[1][n] + s_sum[2][n] + s_sum[3][n];
        if (t != 0.0f) {
            if (n < n0) { if (db0) atomicAdd(db0 + n, t); }
            else if (db1) atomicAdd(db1 + (n - n0), t);
        }
    }
    const unsigned long long flags = s_flag;
    if (threadIdx.x == 0) s_base = flags ? atomicAdd(row_count, __popcll(flags)) : 0;
    __syncthreads();
    if (threadIdx.x < kPackRows && ((flags >> threadIdx.x) & 1ull))
        row_list[s_base + __popcll(flags & ((1ull << threadIdx.x) - 1ull))] = m0 + (int)threadIdx.x;
}

// ---- gather_rows_kernel (ordered pipeline) walks the anchor lists anchor_fill_kernel wrote: a wave per listed anchor copies its C + 4
// gradient values into the anchor-row matrix (levels that took the anchor form) and sums them per 32-row chunk for the bias gradients
// (every level: the bias gradient IS the sum over the marked anchors).
constexpr int kGatherRows = 32;   // listed anchors per workgroup of gather_rows_kernel: 8 per wave, 4 in flight
constexpr int kGtabInts = 2 + kMaxProblems * kMaxAnchorTypes_;   // [0] total chunks, [1 + i * 16 + k] first chunk of (level i, type k), then the end
// chunk table of gather_rows_kernel from the anchor counts, by one wave: lane l owns entries 2 l and 2 l + 1 of the 8 levels x 16 types
__device__ __forceinline__ void build_gather_table(const int* __restrict__ acounts, const int* nb, int n, int* __restrict__ gtab) {
    const int lane = threadIdx.x & 63;
    int cnt[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int e = 2 * lane + u, i = e / kMaxAnchorTypes_, k = e % kMaxAnchorTypes_;
        cnt[u] = (i < n && k < nb[i]) ? (acounts[e] + kGatherRows - 1) / kGatherRows : 0;
    }
    const int mine = cnt[0] + cnt[1];
    int incl = mine;
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
        const int t = __shfl_up(incl, d, kWave);
        if (lane >= d) incl += t;
    }
    const int base = incl - mine;
    gtab[1 + 2 * lane] = base;
    gtab[2 + 2 * lane] = base + cnt[0];
    if (lane == kWave - 1) { gtab[1 + kMaxProblems * kMaxAnchorTypes_] = incl; gtab[0] = incl; }
}
__global__ void __launch_bounds__(256) gather_rows_kernel(PackGroup grp, const int* __restrict__ gtab) {
    __shared__ float s_sum[4][256];   // per wave: column sums of its rows (Jpad <= 256)
    __shared__ int s_tab[kGtabInts];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int t = threadIdx.x; t < kGtabInts; t += 256) s_tab[t] = gtab[t];
    __syncthreads();
    const int total = s_tab[0];
    for (int chunk = blockIdx.x; chunk < total; chunk += gridDim.x) {
        int e = 0;   // the last (level, type) whose first chunk is <= chunk and that has chunks
#pragma unroll 1
        for (int q = 1; q < kMaxProblems * kMaxAnchorTypes_; ++q)
            if (s_tab[1 + q] <= chunk && s_tab[2 + q] > s_tab[1 + q]) e = q;
        const int pi = e / kMaxAnchorTypes_, k = e % kMaxAnchorTypes_;
        const PackLevel& L = grp.lv[pi];
        const int c = chunk - s_tab[1 + e];
        const int n = L.acount[k];
        const bool store = *L.mode == 2;      // the anchor-row matrix is read only by the anchor form
        const int C = L.C, Jpad = L.Jpad;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};   // this lane's columns lane, lane + 64, ...
        const int r0 = c * kGatherRows + wave * 8, r_end = min(n, (c + 1) * kGatherRows);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            int mrow[4];
            float v[4][4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int r = r0 + half * 4 + u;
                mrow[u] = r < r_end ? L.apix[(long long)k * L.cap + r] : -1;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int m = mrow[u] < 0 ? 0 : mrow[u];
                const int b = m / L.HW, p = m - b * L.HW;
                const float* srow = L.ds + (long long)b * grp.sb + ((long long)p * L.nb + k) * C;
                const float* lrow = L.LQ ? L.dl + (long long)b * grp.lb + ((long long)p * L.nb + k) * 4 : srow;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int j = lane + 64 * q;
                    v[u][q] = (mrow[u] >= 0 && j < C + L.LQ && j < Jpad) ? (j < C ? srow[j] : lrow[j - C]) : 0.0f;
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (mrow[u] < 0) continue;
                float* grow = L.ga + ((long long)k * L.cap + (r0 + half * 4 + u)) * Jpad;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int j = lane + 64 * q;
                    if (store && j < Jpad) grow[j] = v[u][q];
                    acc[q] += v[u][q];
                }
            }
        }
        __syncthreads();   // (s_sum of the previous chunk has been read)
#pragma unroll
        for (int q = 0; q < 4; ++q) s_sum[wave][lane + 64 * q] = acc[q];
        __syncthreads();
        for (int j = threadIdx.x; j < C + L.LQ; j += 256) {
            const float t = s_sum[0][j] + s_sum[1][j] + s_sum[2][j] + s_sum[3][j];
            L.dbp[((long long)k * L.chunks_cap + c) * Jpad + j] = t;   // (stored, not added: anchor_dbias_kernel adds the chunks in order)
        }
    }
}

// second half of the pack: the dense rows of the levels whose chosen form reads them (mode 0 / 1); a fixed grid walks the row blocks
template <int ITERS>
__global__ void __launch_bounds__(256) pack_store_kernel(PackGroup grp, int total_blocks) {
    __shared__ float s_dummy[kPackColIters * 64];
    // (all the modes are loaded before any is tested: `any || *mode != 2` short-circuits into one dependent L2 round trip per level --
    // 12 us for a launch that, under hard-negative mining, has nothing to do)
    int modes[kMaxProblems];
#pragma unroll
    for (int i = 0; i < kMaxProblems; ++i) modes[i] = i < grp.count ? *grp.lv[i].mode : 2;
    bool any = false;
#pragma unroll
    for (int i = 0; i < kMaxProblems; ++i) any |= modes[i] != 2;
    if (!any) return;
    // level by level, only the levels that read the dense rows (walking ALL row blocks and testing the mode of each cost a dependent L2
    // round trip per block: 12 us to pack the 128 rows of SSD-300's last level)
#pragma unroll
    for (int i = 0; i < kMaxProblems; ++i) {
        if (i >= grp.count || modes[i] == 2) continue;
        const PackLevel& L = grp.lv[i];
        const int nblk = (i + 1 < grp.count ? grp.lv[i + 1].block_begin : total_blocks) - L.block_begin;
        const int M = grp.B * L.HW;
        for (int blk = blockIdx.x; blk < nblk; blk += gridDim.x)
            pack_rows<ITERS, true, false>(L, L.ds, L.dl, grp.sb, grp.lb, grp.B, blk * kPackRows, M, s_dummy);
    }
}

// (legacy pipeline) mode[i] = 1 (pixel-sparse backward) when fewer than 70 % of the level's pixel rows carry a gradient, else 0 (dense).
// The sparse form never does more multiplies than the dense one; what it adds is the scatter's atomic traffic
// (rows * 9 * Cin * 4 bytes), which costs about a third of the dense GEMM time at full density.
constexpr int kMaxAnchorTypes = 16;
static_assert(kMaxAnchorTypes_ == kMaxAnchorTypes, "anchor types");
struct LevelTotals { int v[kMaxProblems]; int force; };
__global__ void decide_sparse_kernel(const int* __restrict__ counts, LevelTotals totals, int n, int* __restrict__ mode) {
    const int i = threadIdx.x;
    if (i >= n) return;
    int m = ((long long)counts[i] * 10 < (long long)totals.v[i] * 7) ? 1 : 0;
    if (totals.force == 0 || totals.force == 1) m = totals.force;
    mode[i] = m;
}

// builds the tile list of a pixel-sparse scatter launch (igemm_dma_kernel, vtab): one thread, <= 8 problems
struct VtabArgs {
    int count;
    struct { const int* counts; const int* mode; int want; int n_blocks; } p[kMaxProblems];
    int* vtab;
};
__global__ void build_vtab_kernel(VtabArgs a) {
    if (threadIdx.x || blockIdx.x) return;
    int* vt = a.vtab;
    int v = 0;
    for (int i = 0; i < a.count; ++i) {
        vt[1 + i] = v;
        const bool on = !a.p[i].mode || *a.p[i].mode == a.p[i].want;
        if (on) v += (*a.p[i].counts + kBM - 1) / kBM * a.p[i].n_blocks;
    }
    vt[1 + a.count] = v;
    vt[0] = v;
}

// =====================================================================================================================
// Ordered anchor-row backward (the sparse form of the heads' backward; no atomics anywhere, the same bits on every run).
//
// Under hard-negative mining ~4 % of the anchors carry a gradient (detection/sampler.py:12-25, detection/losses/multibox_loss.py:60-90).
// Row = one such anchor: its C score gradients and 4 box gradients (Jpad = C + 4 rounded up to 32), multiplied with the weight rows of its
// anchor TYPE k only (detector_builder.py:111-137: output channel k * C + c of the score head, k * 4 + q of the loc head).
//   anchor_mask_kernel    (only when the caller has no row mask) which anchors carry a gradient, from dscores / dlocs themselves;
//   anchor_count_kernel   per 256-pixel block and anchor type: marked anchors;
//   anchor_plan_kernel    ONE workgroup: exclusive scan of the block counts per (level, type) -- rows are numbered in PIXEL ORDER, whatever
//                         order the blocks ran in --, the level's backward form (2 = anchor rows when they fit the T buffer, else 0 = dense),
//                         row bases of the types inside the level's T buffer, the tile list of the row GEMM, the chunk table of
//                         gather_rows_kernel;
//   anchor_fill_kernel    pixel id of every row (apix) and, inverted, the T row of every (type, pixel) (aidx; -1: no gradient);
//   gather_rows_kernel    copies the rows' C + 4 values into the compact matrix ga[type][row][Jpad]; per 32-row chunk column sums
//                         (stored, not added: anchor_dbias_kernel adds them in chunk order);
//   anchor_rowgemm_kernel T[row][tap * Cin + c] = sum_j ga[row][j] * W_type[j][tap][c] -- every contribution row is STORED once (plain
//                         stores run at 4-5 x the rate of float atomics, MI355X_MICROARCH.md "Global float atomics");
//   anchor_dx_kernel      dX[pixel][c] = sum over the (tap, type) pairs whose source anchor exists, in that fixed order, of its T row
//                         (the inverted index is aidx: at most 9 * nb lookups per pixel); writes EVERY pixel, so dX needs no zero-fill;
//   igemm_wgrad_dma_kernel<.., ROW32> over the rows (K = rows in pixel order, split over workgroups by a host-side rule; every split stores
//                         its own copy, reduce_partials_kernel adds the copies in split order).
constexpr int kAT = kMaxAnchorTypes;
constexpr int kAnchorBlk = 256;                       // pixels per workgroup of the count / fill kernels
constexpr int kRgRows = 128;                          // rows per workgroup of the row GEMM
constexpr int kRgStepCols = 64;                       // T columns per step (two 32-column MFMA tiles)
constexpr int kRgChunkCols = 512;                     // T columns per work item (8 steps)
// device-side plan (ints): [0] row GEMM items, [1 + i] first item of level i (i = 0 .. count), then per level kPlanStride ints:
// [0 .. kAT] first T row of type k (entry nb: rows of the level), [kAT + 1 .. 2 kAT + 1] first 128-row tile of type k
constexpr int kPlanHead = 1 + kMaxProblems + 1;
constexpr int kPlanStride = 2 * (kAT + 1);
constexpr int kPlanQueue = kPlanHead + kMaxProblems * kPlanStride;   // head of the row GEMM's work queue (zeroed by anchor_plan_kernel)
constexpr int kPlanInts = kPlanQueue + 1;

struct AnchorLevel {
    const unsigned char* rmask;   // mask byte of (image b, pixel p, type k): rmask[b * a_total + a_off + p * nb + k]
    int a_total, a_off;
    int nb, HW, cap;              // cap = B * HW pixels = rows a type can have
    int blk_begin, nblk;          // workgroups of the count / fill grids
    int* blk;                     // [nblk][kAT]: marked anchors per block and type -> (anchor_plan_kernel) first row of the block
    int* apix;                    // [nb][cap] pixel id of row r of type k
    int* aidx;                    // [nb][cap] T row of (type k, pixel m), -1: none
    int tcap;                     // rows the level's T buffer holds
    int items_per_tile;           // column chunks of the row GEMM per row tile (0: no data gradient wanted)
};
struct AnchorGroup {
    int count, force;
    int* acounts;   // [kMaxProblems][kAT]
    int* plan;      // [kPlanInts]
    int* mode;      // [kMaxProblems]
    int* gtab;      // chunk table of gather_rows_kernel
    float* zeros;   // 64 floats of zeros (source of the row GEMM's padding lanes)
    AnchorLevel lv[kMaxProblems];
};

__device__ __forceinline__ unsigned anchor_bits(const AnchorLevel& L, int m) {
    unsigned bits = 0;
    if (m < L.cap) {
        const int b = m / L.HW, p = m - b * L.HW;
        const unsigned char* mp = L.rmask + (long long)b * L.a_total + L.a_off + (long long)p * L.nb;
        for (int k = 0; k < L.nb; ++k) bits |= mp[k] ? 1u << k : 0u;
    }
    return bits;
}

__global__ void __launch_bounds__(kAnchorBlk) anchor_count_kernel(AnchorGroup grp) {
    int pi = 0;
#pragma unroll 1
    for (int i = 1; i < grp.count; ++i)
        if ((int)blockIdx.x >= grp.lv[i].blk_begin) pi = i;
    const AnchorLevel& L = grp.lv[pi];
    __shared__ int s_w[4][kAT];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int blk = blockIdx.x - L.blk_begin;
    const unsigned bits = anchor_bits(L, blk * kAnchorBlk + threadIdx.x);
#pragma unroll 1
    for (int k = 0; k < L.nb; ++k) {
        const unsigned long long bk = __ballot((bits >> k) & 1u);
        if (lane == 0) s_w[wave][k] = __popcll(bk);
    }
    __syncthreads();
    if ((int)threadIdx.x < L.nb) L.blk[blk * kAT + threadIdx.x] = s_w[0][threadIdx.x] + s_w[1][threadIdx.x] + s_w[2][threadIdx.x] + s_w[3][threadIdx.x];
}

__global__ void __launch_bounds__(1024) anchor_plan_kernel(AnchorGroup grp) {
    __shared__ int s_cnt[kMaxProblems * kAT];
    __shared__ int s_nb[kMaxProblems];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x < 64) grp.zeros[threadIdx.x] = 0.0f;
    if (threadIdx.x < kMaxProblems) s_nb[threadIdx.x] = (int)threadIdx.x < grp.count ? grp.lv[threadIdx.x].nb : 0;
    // (pair p = wave + 16 i is level p % 8, type p / 8: the few types a level has are spread over the waves -- with level-major pairs
    // the waves 0 .. nb - 1 scanned every level one after the other and the rest idled)
#pragma unroll 1
    for (int q = wave; q < kMaxProblems * kAT; q += 16) {
        const int li = q % kMaxProblems, k = q / kMaxProblems, p = li * kAT + k;
        int run = 0;
        if (li < grp.count && k < grp.lv[li].nb) {
            const AnchorLevel& L = grp.lv[li];
#pragma unroll 1
            for (int c0 = 0; c0 < L.nblk; c0 += kWave) {
                const int i = c0 + lane;
                const int v = i < L.nblk ? L.blk[i * kAT + k] : 0;
                const int incl = wave_inclusive_scan(v, OpAddI());
                if (i < L.nblk) L.blk[i * kAT + k] = run + incl - v;
                run += __shfl(incl, kWave - 1, kWave);
            }
        }
        if (lane == 0) { s_cnt[p] = run; grp.acounts[p] = run; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int items = 0;
        for (int li = 0; li < grp.count; ++li) {
            const AnchorLevel& L = grp.lv[li];
            int* pl = grp.plan + kPlanHead + li * kPlanStride;
            long long rows = 0;
            for (int k = 0; k < L.nb; ++k) rows += s_cnt[li * kAT + k];
            int m = rows <= (long long)L.tcap ? 2 : 0;
            if (grp.force == 0) m = 0;
            int base = 0, tiles = 0;
            for (int k = 0; k <= L.nb; ++k) {
                pl[k] = base;
                pl[kAT + 1 + k] = tiles;
                if (k < L.nb && m == 2) { base += s_cnt[li * kAT + k]; tiles += (s_cnt[li * kAT + k] + kRgRows - 1) / kRgRows; }
            }
            grp.plan[1 + li] = items;
            items += tiles * L.items_per_tile;
            grp.mode[li] = m;
        }
        for (int li = grp.count; li <= kMaxProblems; ++li) grp.plan[1 + li] = items;
        grp.plan[0] = items;
        grp.plan[kPlanQueue] = 0;
    }
    if (wave == 0 && grp.gtab) build_gather_table(s_cnt, s_nb, grp.count, grp.gtab);
}

__global__ void __launch_bounds__(kAnchorBlk) anchor_fill_kernel(AnchorGroup grp) {
    int pi = 0;
#pragma unroll 1
    for (int i = 1; i < grp.count; ++i)
        if ((int)blockIdx.x >= grp.lv[i].blk_begin) pi = i;
    const AnchorLevel& L = grp.lv[pi];
    __shared__ int s_w[4][kAT];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int blk = blockIdx.x - L.blk_begin;
    const int m = blk * kAnchorBlk + threadIdx.x;
    const unsigned bits = anchor_bits(L, m);
#pragma unroll 1
    for (int k = 0; k < L.nb; ++k) {
        const unsigned long long bk = __ballot((bits >> k) & 1u);
        if (lane == 0) s_w[wave][k] = __popcll(bk);
    }
    __syncthreads();
    const int* tbase = grp.plan + kPlanHead + pi * kPlanStride;
    const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll 1
    for (int k = 0; k < L.nb; ++k) {
        const unsigned long long bk = __ballot((bits >> k) & 1u);
        int off = L.blk[blk * kAT + k];
        for (int w = 0; w < wave; ++w) off += s_w[w][k];
        off += __popcll(bk & below);
        const bool on = (bits >> k) & 1u;
        if (on) L.apix[(long long)k * L.cap + off] = m;
        if (m < L.cap) L.aidx[(long long)k * L.cap + m] = on ? tbase[k] + off : -1;
    }
}

// Which anchors carry a gradient, from the gradient itself (callers without a row mask): out[(b * HW + p) * nb + k] = any of the C score
// values or 4 box values of anchor (p, k) of image b is non-zero.  A wave per pixel row, lanes over the nb * (C + 4) columns.
struct MaskLevel { const float* ds; const float* dl; unsigned char* out; int nb, C, HW, blk_begin, LQ; };
struct MaskGroup { int count, B; long long sb, lb; MaskLevel lv[kMaxProblems]; };
constexpr int kMaskRows = 16;   // pixel rows per workgroup
__global__ void __launch_bounds__(256) anchor_mask_kernel(MaskGroup grp) {
    int pi = 0;
#pragma unroll 1
    for (int i = 1; i < grp.count; ++i)
        if ((int)blockIdx.x >= grp.lv[i].blk_begin) pi = i;
    const MaskLevel& L = grp.lv[pi];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int M = grp.B * L.HW, n0 = L.nb * L.C, n1 = L.nb * L.LQ;
    const float inv_c = 1.0f / (float)L.C;
    const int m0 = (blockIdx.x - L.blk_begin) * kMaskRows;
#pragma unroll 1
    for (int r = wave; r < kMaskRows; r += 4) {
        const int m = m0 + r;
        if (m >= M) break;
        const int b = m / L.HW, p = m - b * L.HW;
        const float* srow = L.ds + (long long)b * grp.sb + (long long)p * n0;
        const float* lrow = n1 ? L.dl + (long long)b * grp.lb + (long long)p * n1 : srow;
        unsigned mine = 0u;
        for (int n = lane; n < n0; n += kWave) {
            const int a = (int)(((float)n + 0.5f) * inv_c);   // n / C, exact for n < 2^20
            mine |= srow[n] != 0.0f ? 1u << a : 0u;
        }
        for (int n = lane; n < n1; n += kWave) mine |= lrow[n] != 0.0f ? 1u << (n >> 2) : 0u;
        unsigned char v = 0;
        for (int k = 0; k < L.nb; ++k) {
            const bool any = __ballot((mine >> k) & 1u) != 0ull;
            if (lane == k) v = any ? 1 : 0;
        }
        if (lane < L.nb) L.out[(long long)m * L.nb + lane] = v;
    }
}

// db of every (level, type): the column sums gather_rows_kernel stored per 32-row chunk, added in chunk order
struct DbiasLevel { const float* part; float* db0; float* db1; int nb, C, Jpad, chunks_cap, LQ; };
struct DbiasGroup { int count; const int* acounts; DbiasLevel lv[kMaxProblems]; };
__global__ void __launch_bounds__(128) anchor_dbias_kernel(DbiasGroup grp) {
    const int li = blockIdx.x / kAT, k = blockIdx.x % kAT;
    if (li >= grp.count) return;
    const DbiasLevel& L = grp.lv[li];
    if (k >= L.nb) return;
    const int chunks = (grp.acounts[li * kAT + k] + 31) / 32;
    const float* p = L.part + (long long)k * L.chunks_cap * L.Jpad;
    for (int j = threadIdx.x; j < L.C + L.LQ; j += blockDim.x) {
        float s = 0.0f;
        int c = 0;
        for (; c + 8 <= chunks; c += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = p[(long long)(c + u) * L.Jpad + j];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; c < chunks; ++c) s += p[(long long)c * L.Jpad + j];
        if (j < L.C) { if (L.db0) L.db0[k * L.C + j] = s; }
        else if (L.db1) L.db1[k * L.LQ + (j - L.C)] = s;
    }
}

// ---- the row GEMM -------------------------------------------------------------------------------------------------------
// T[row][n] = sum_j ga[row][j] * Wk[j][n], n = tap * Cin + c in [0, 9 Cin), Wk[j] = row k * C + j of the score weights (j < C), row
// k * 4 + j - C of the loc weights (j < C + 4), zeros above: the weights are read where they lie ([n][tap][cin] IS [j][tap * Cin + c]).
// K = Jpad <= 128 is short, so the A operand never leaves registers: a wave holds its 32 rows x Jpad (Jpad / 2 registers) while the
// workgroup walks its columns in steps of 64; the B slice [Jpad][64] of a step travels global -> LDS by global_load_lds_dwordx4 (one
// piece = 4 rows j x 256 B; a lane whose row or column does not exist reads the zero line), two stages.  MFMA step s = 4 g + e takes
// K index 8 g + 4 h + e from lane half h: the lane's A values are float4 number 2 g + h of its row, its B values lie in piece 2 g + h.
// LDS image of a step: the piece PAIR g at float g * 544, its odd piece 288 floats (256 + 32) further on, so that the two halves of a
// ds_read_b32 (same column, rows 4 apart) hit banks 32 apart (and the odd piece ends where the next pair begins).
struct RowGemmLevel {
    const float* ga; const float* ws; const float* wl;
    float* T;
    int nb, C, cap, K9, n_chunks, LQ;
};
struct RowGemmGroup { int count; int* plan; const int* acounts; const float* zeros; int probe; RowGemmLevel lv[kMaxProblems]; };   // probe: measurement knob (1: T stores dropped)

#ifdef SSDK_RG_STAMPS
// experiment only (never built into the shipped library; tools/rg_stamps.py): per workgroup, 100 MHz wall-clock sums of the phases of its items
__device__ unsigned long long g_rg_stamp[1024 * 8];
extern "C" int ssdk_debug_read_rg_stamps(unsigned long long* host) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_rg_stamp), sizeof(unsigned long long) * 1024 * 8); }
extern "C" int ssdk_debug_zero_rg_stamps() { static unsigned long long z[1024 * 8]; return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_rg_stamp), z, sizeof(z)); }
#define RG_STAMP(var) unsigned long long var = 0; if (threadIdx.x == 0) { __builtin_amdgcn_sched_barrier(0); var = wall_clock64(); __builtin_amdgcn_sched_barrier(0); }
#define RG_ADD(slot, a, b) if (threadIdx.x == 0 && blockIdx.x < 1024) g_rg_stamp[blockIdx.x * 8 + (slot)] += (b) - (a);
#else
#define RG_STAMP(var)
#define RG_ADD(slot, a, b)
#endif
template <int KQ>   // Jpad = 32 KQ
__global__ void __launch_bounds__(256, KQ <= 3 ? 3 : 2) anchor_rowgemm_kernel(RowGemmGroup grp) {
    constexpr int Jpad = 32 * KQ, G = Jpad / 8, kPieces = Jpad / 4, kPerWave = kPieces / 4;
    constexpr int kPair = 544, kOdd = 288;
    constexpr int kStageFloats = G * kPair;
    __shared__ __attribute__((aligned(1024))) float s_b0[kStageFloats];
    __shared__ __attribute__((aligned(1024))) float s_b1[kStageFloats];
    __shared__ int s_item;
    const int* plan = grp.plan;
    const int* acounts = grp.acounts;
    const int total = plan[0];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31, h = lane >> 5;
    // Items are handed out by a queue (one returning atomic per ~20 us item), heaviest levels first: with a fixed stride the workgroups
    // that started with the largest tiles also got the second round (PMC: 1.9 of 3 wave slots busy on average).  Which workgroup
    // computes an item does not change a bit of it.
    int item = blockIdx.x;
#pragma unroll 1
    while (item < total) {
        RG_STAMP(t_item)
        int li = 0;
#pragma unroll 1
        for (int i = 1; i < grp.count; ++i)
            if (item >= plan[1 + i]) li = i;
        const RowGemmLevel& L = grp.lv[li];
        const int local = item - plan[1 + li];
        const int chunk = local % L.n_chunks, tile = local / L.n_chunks;
        const int* pl = plan + kPlanHead + li * kPlanStride;
        int k = 0;
#pragma unroll 1
        for (int q = 1; q < L.nb; ++q)
            if (tile >= pl[kAT + 1 + q]) k = q;
        const int r0 = (tile - pl[kAT + 1 + k]) * kRgRows;
        const int cnt = acounts[li * kAT + k];
        const int K9 = L.K9, C = L.C;

        const int col_begin = chunk * kRgChunkCols, col_end = min(K9, col_begin + kRgChunkCols);
        const int nsteps = (col_end - col_begin + kRgStepCols - 1) / kRgStepCols;
        // rows of this wave that exist (uniform; through readfirstlane so that the store descriptor below is built in scalar registers --
        // the counts come from vector loads, and a descriptor the compiler takes for divergent costs a waterfall loop per store)
        const int nvalid = __builtin_amdgcn_readfirstlane(min(32, max(0, cnt - r0 - wave * 32)));

        // A: this lane's row, float4 number 2 g + h
        f32x4 a[G];
        {
            const int r = r0 + wave * 32 + n;
            const float* arow = L.ga + ((long long)k * L.cap + (r < cnt ? r : 0)) * Jpad + 4 * h;
#pragma unroll
            for (int g = 0; g < G; ++g) {
                a[g] = *reinterpret_cast<const f32x4*>(arow + 8 * g);
                if (r >= cnt) a[g] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        // B staging: piece p = wave + 4 i holds rows j = 4 p .. 4 p + 3; lane l: row 4 p + l / 16, float4 column l % 16
        const float* src[kPerWave];
        const int colq = (lane & 15) * 4;
#pragma unroll
        for (int i = 0; i < kPerWave; ++i) {
            const int j = 4 * (wave + 4 * i) + (lane >> 4);
            src[i] = j < C ? L.ws + ((long long)k * C + j) * K9 : (j < C + L.LQ ? L.wl + ((long long)k * L.LQ + (j - C)) * K9 : nullptr);
        }
        // The LDS-DMA is issued from an asm statement (M0 = the piece's LDS address, saved and restored inside the statement:
        // cdna_hip_programming.md, "M0 ... is compiler-reserved"): as a builtin the compiler sees an LDS write in flight and puts
        // s_waitcnt vmcnt(0) in front of the next step's fragment reads -- which also waits for the T stores of the step before.  Hidden
        // from its bookkeeping, the pieces are waited for by the counted waits below; a wait the compiler inserts for its own loads can
        // only be longer for them, never shorter (vmcnt counts down in issue order).
        const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(__attribute__((address_space(3))) float*)s_b0);
        const unsigned lds1 = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(__attribute__((address_space(3))) float*)s_b1);
        auto stage = [&](int DST, int st) {   // DST is a literal at the call sites
            const int col = col_begin + st * kRgStepCols + colq;
#pragma unroll
            for (int i = 0; i < kPerWave; ++i) {
                const int p = wave + 4 * i;
                const float* s = (src[i] && col < col_end) ? src[i] + col : grp.zeros;
                const unsigned dst = __builtin_amdgcn_readfirstlane((DST ? lds1 : lds0) + (unsigned)((p >> 1) * kPair + (p & 1) * kOdd) * 4u);
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(s), "s"(dst) : "memory");
            }
        };
        // T rows of this wave through a buffer descriptor that ends behind its last existing row (built before the first piece is
        // issued: its operands come from vector loads, and a wait for those would wait for the pieces as well)
        const __amdgpu_buffer_rsrc_t rsrc_t = __builtin_amdgcn_make_buffer_rsrc(
            (void*)uniform_ptr(L.T + (long long)(pl[k] + r0 + wave * 32) * K9), 0, __builtin_amdgcn_readfirstlane(grp.probe == 1 ? 0 : nvalid * K9 * 4), 0x00020000);
        __syncthreads();   // (the previous item's last step has been read)
        stage(0, 0);
        __builtin_amdgcn_s_waitcnt(0 | (7 << 4));   // vmcnt(0): this wave's pieces have landed (and the previous item's stores are done)
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int g = 0; g < G; ++g) asm volatile("" : "+v"(a[g]));   // (the A rows are waited for HERE, where nothing else is in flight)
        if (nsteps > 1) stage(1, 1);
        RG_STAMP(t_pro)
        RG_ADD(0, t_item, t_pro)
#ifdef SSDK_RG_STAMPS
        if (threadIdx.x == 0 && blockIdx.x < 1024) { g_rg_stamp[blockIdx.x * 8 + 5] += 1; g_rg_stamp[blockIdx.x * 8 + 6] += (unsigned long long)nsteps; }
#endif

        // (the stores of rows that do not exist are dropped by the descriptor's range check -- no predicates, and EVERY wave with rows
        // issues exactly 32 stores per step, 16 when only one column tile of the step exists: what the counted wait below relies on)
        const unsigned lane_off = (unsigned)(4 * h * K9 + n) * 4u;
        auto body = [&](auto st_c, int st) {
            constexpr int ST = decltype(st_c)::value;
            const int col0 = col_begin + st * kRgStepCols;
            f32x16 acc[2];
            RG_STAMP(t_a)
            if (nvalid > 0) {
                const float* sb = (ST ? s_b1 : s_b0) + h * kOdd + n;
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[ct][e] = 0.0f;
                // the B values of group g + 1 are read BEFORE the eight MFMAs of group g (pinned: left alone the compiler reads a pair,
                // waits, issues its two MFMAs, reads the next pair into the same registers ...)
                float b[2][4][2];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    b[0][e][0] = sb[e * 64];
                    b[0][e][1] = sb[e * 64 + 32];
                }
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    if (g + 1 < G) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            b[(g + 1) & 1][e][0] = sb[(g + 1) * kPair + e * 64];
                            b[(g + 1) & 1][e][1] = sb[(g + 1) * kPair + e * 64 + 32];
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g][e], b[g & 1][e][0], acc[0], 0, 0, 0);
                        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g][e], b[g & 1][e][1], acc[1], 0, 0, 0);
                    }
                }
            }
            RG_STAMP(t_b)
            RG_ADD(1, t_a, t_b)
            if (st + 1 < nsteps) {
                // The next step's pieces must have landed: vmcnt(0).  That also waits for the T stores of the PREVIOUS step, which have had
                // this step's MFMAs to complete (a counted wait that leaves them in flight measured the same: 122.9 vs 124.3 us).
                __builtin_amdgcn_s_waitcnt(0 | (7 << 4));   // s_waitcnt simm16 = vmcnt[3:0] | expcnt << 4 | lgkmcnt << 8 | vmcnt[5:4] << 14
                __builtin_amdgcn_s_barrier();
                RG_STAMP(t_c)
                RG_ADD(2, t_b, t_c)
                if (st + 2 < nsteps) stage(ST, st + 2);   // into the stage every wave has just finished reading
            }
            RG_STAMP(t_d)
            if (nvalid > 0) {
                // C/D map of the 32 x 32 MFMA: column = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    if (col0 + ct * 32 >= col_end) continue;   // (uniform: 9 Cin is a multiple of 32; only an item's last step can be half)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int row = (e & 3) + 8 * (e >> 2);
                        const float v = acc[ct][e];   // (a named float: __builtin_bit_cast on the vector element stored element 0 sixteen times)
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rsrc_t, lane_off + (unsigned)(row * K9 + col0 + ct * 32) * 4u, 0, 0);
                    }
                }
            }
            RG_STAMP(t_e)
            RG_ADD(3, t_d, t_e)
        };
#pragma unroll 1
        for (int st = 0; st < nsteps; st += 2) {
            body(std::integral_constant<int, 0>{}, st);
            if (st + 1 < nsteps) body(std::integral_constant<int, 1>{}, st + 1);
        }
        if (tid == 0) s_item = (int)gridDim.x + atomicAdd(grp.plan + kPlanQueue, 1);
        __syncthreads();
        item = s_item;
        RG_STAMP(t_end)
        RG_ADD(4, t_item, t_end)
    }
}

// ---- the sum pass ----------------------------------------------------------------------------------------------------------
// dX[m][c] = sum over taps t = (ky, kx) in order, over types k in order, of T[aidx[k][m']][t * Cin + c], m' = the output pixel that tap
// connects to input pixel m: (y + 1 - ky, x + 1 - kx) (3 x 3, pad 1, stride 1).  A wave per pixel; the hits go through LDS.
struct DxLevel { const int* aidx; const float* T; float* dx; const int* mode; int nb, H, W, cin, cap, blk_begin; };
struct DxGroup { int count; DxLevel lv[kMaxProblems]; };
constexpr int kDxHits = 9 * kAT;
__global__ void __launch_bounds__(256) anchor_dx_kernel(DxGroup grp) {
    int pi = 0;
#pragma unroll 1
    for (int i = 1; i < grp.count; ++i)
        if ((int)blockIdx.x >= grp.lv[i].blk_begin) pi = i;
    const DxLevel& L = grp.lv[pi];
    if (*L.mode != 2) return;
    __shared__ unsigned s_hit[4][kDxHits];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m = (blockIdx.x - L.blk_begin) * 4 + wave;
    if (m >= L.cap) return;   // (whole waves; no workgroup barrier below)
    const int HW = L.H * L.W;
    const int b = m / HW, p = m - b * HW, y = p / L.W, x = p - y * L.W;
    const int nq = 9 * L.nb, K9 = 9 * L.cin;
    const float inv_nb = 1.0f / (float)L.nb;
    int nh = 0;
    const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll 1
    for (int q0 = 0; q0 < nq; q0 += kWave) {
        const int q = q0 + lane;
        const int t = (int)(((float)q + 0.5f) * inv_nb), k = q - t * L.nb;   // q / nb, exact for q < 2^20
        const int ky = t / 3, kx = t - 3 * ky;
        const int yo = y + 1 - ky, xo = x + 1 - kx;
        int row = -1;
        if (q < nq && (unsigned)yo < (unsigned)L.H && (unsigned)xo < (unsigned)L.W) row = L.aidx[(long long)k * L.cap + b * HW + yo * L.W + xo];
        const unsigned long long hit = __ballot(row >= 0);
        if (row >= 0) s_hit[wave][nh + __popcll(hit & below)] = (unsigned)row * (unsigned)K9 + (unsigned)(t * L.cin);
        nh += __popcll(hit);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    float* const out = L.dx + (long long)m * L.cin;
    for (int c0 = lane * 4; c0 < L.cin; c0 += 4 * kWave) {
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
        int i = 0;
        for (; i + 4 <= nh; i += 4) {
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(L.T + s_hit[wave][i] + c0), v1 = *reinterpret_cast<const f32x4*>(L.T + s_hit[wave][i + 1] + c0);
            const f32x4 v2 = *reinterpret_cast<const f32x4*>(L.T + s_hit[wave][i + 2] + c0), v3 = *reinterpret_cast<const f32x4*>(L.T + s_hit[wave][i + 3] + c0);
            acc += v0; acc += v1; acc += v2; acc += v3;
        }
        for (; i < nh; ++i) acc += *reinterpret_cast<const f32x4*>(L.T + s_hit[wave][i] + c0);
        *reinterpret_cast<f32x4*>(out + c0) = acc;
    }
}

// ---- strided convolutions' data gradient without atomics: contribution rows + per-pixel sum --------------------------------------------
// T[out pixel][tap * Cin + c] = dy[out pixel][:] . W[:, tap, c] is an ordinary 1 x 1 GEMM over the output pixels (no multiply is spent on
// (pixel, tap) pairs that do not exist); dX[(y, x)][c] = the sum, in tap order, of the T rows of the output pixels a tap connects to (y, x):
// yo = (y + pad - ky) / stride where that divides (at most ceil(k / stride)^2 of the k^2 taps).  Every dX pixel is written: no zero-fill.
struct StridedDxJob { const float* T; float* dx; int B, hin, win, cin, ho, wo, ks, stride, pad, blk_begin; };
struct StridedDxGroup { int count; StridedDxJob j[kMaxProblems]; };
__global__ void __launch_bounds__(256) strided_dx_kernel(StridedDxGroup grp) {
    int ji = 0;
#pragma unroll 1
    for (int i = 1; i < grp.count; ++i)
        if ((int)blockIdx.x >= grp.j[i].blk_begin) ji = i;
    const StridedDxJob& J = grp.j[ji];
    const int c4n = J.cin / 4;
    const long long total = (long long)J.B * J.hin * J.win * c4n;
    const long long e = (long long)(blockIdx.x - J.blk_begin) * 256 + threadIdx.x;
    if (e >= total) return;
    const int c4 = (int)(e % c4n);
    const long long pix = e / c4n;
    const int x = (int)(pix % J.win), y = (int)((pix / J.win) % J.hin), b = (int)(pix / ((long long)J.win * J.hin));
    const long long K9 = (long long)J.ks * J.ks * J.cin;
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int ky = 0; ky < J.ks; ++ky) {
        const int ty = y + J.pad - ky;
        if (ty < 0 || ty % J.stride) continue;
        const int yo = ty / J.stride;
        if (yo >= J.ho) continue;
        for (int kx = 0; kx < J.ks; ++kx) {
            const int tx = x + J.pad - kx;
            if (tx < 0 || tx % J.stride) continue;
            const int xo = tx / J.stride;
            if (xo >= J.wo) continue;
            acc += *reinterpret_cast<const f32x4*>(J.T + (((long long)b * J.ho + yo) * J.wo + xo) * K9 + (long long)(ky * J.ks + kx) * J.cin + 4 * c4);
        }
    }
    *reinterpret_cast<f32x4*>(J.dx + pix * J.cin + 4 * c4) = acc;
}

// Deterministic mode: out[e] (+)= sum_k src[k * stride + e], k = 0 .. n_src - 1 IN THAT ORDER (the partial tiles of a K-split weight
// gradient, the per-workgroup column sums of a bias gradient): the fixed-order second half of what the atomics do in any order.
struct ReduceJob {
    float* dst; const float* src;
    long long elems, stride;
    int n_src, accumulate, block_begin, want_mode;
    const int* mode;   // the job runs only when *mode == want_mode (NULL: always)
};
constexpr int kMaxReduceJobs = 36;   // (56 bytes each: the group stays inside the 4 KB of kernel arguments with room to spare)
struct ReduceGroup { int count; ReduceJob j[kMaxReduceJobs]; };
__global__ void __launch_bounds__(256) reduce_partials_kernel(ReduceGroup grp) {
    int ji = 0;
#pragma unroll 1
    for (int i = 1; i < grp.count; ++i)
        if ((int)blockIdx.x >= grp.j[i].block_begin) ji = i;
    const ReduceJob& J = grp.j[ji];
    if (J.mode && *J.mode != J.want_mode) return;
    const long long e4 = ((long long)(blockIdx.x - J.block_begin) * 256 + threadIdx.x) * 4;
    if (e4 >= J.elems) return;
    if (e4 + 4 <= J.elems && ((J.stride | (long long)((uintptr_t)J.src >> 2) | (long long)((uintptr_t)J.dst >> 2)) & 3) == 0) {
        f32x4 a = J.accumulate ? *reinterpret_cast<const f32x4*>(J.dst + e4) : f32x4{0.f, 0.f, 0.f, 0.f};
        const float* p = J.src + e4;
        int k = 0;
        for (; k + 4 <= J.n_src; k += 4) {   // four copies in flight, added in order
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(p), v1 = *reinterpret_cast<const f32x4*>(p + J.stride);
            const f32x4 v2 = *reinterpret_cast<const f32x4*>(p + 2 * J.stride), v3 = *reinterpret_cast<const f32x4*>(p + 3 * J.stride);
            a += v0; a += v1; a += v2; a += v3;
            p += 4 * J.stride;
        }
        for (; k < J.n_src; ++k) { a += *reinterpret_cast<const f32x4*>(p); p += J.stride; }
        *reinterpret_cast<f32x4*>(J.dst + e4) = a;
    } else {
        for (long long e = e4; e < min(J.elems, e4 + 4); ++e) {
            float a = J.accumulate ? J.dst[e] : 0.0f;
            for (int k = 0; k < J.n_src; ++k) a += J.src[(long long)k * J.stride + e];
            J.dst[e] = a;
        }
    }
}

// part[blockIdx.x][n] = sum over this workgroup's rows of dy[row][n]  (deterministic mode's first half of colsum_kernel)
__global__ void __launch_bounds__(256) colsum_partial_kernel(const float* __restrict__ dy, long long M, int N, int ld, float* __restrict__ part,
                                                             int rows_per_block) {
    const long long m0 = (long long)blockIdx.x * rows_per_block, m1 = min(M, m0 + rows_per_block);
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
        float s = 0.0f;
        long long m = m0;
        for (; m + 8 <= m1; m += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = dy[(m + u) * ld + n];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; m < m1; ++m) s += dy[m * ld + n];
        part[(long long)blockIdx.x * N + n] = s;
    }
}

// db[n] += sum over rows of dy[row][n]   (dense [M][N] rows)
__global__ void __launch_bounds__(256) colsum_kernel(const float* __restrict__ dy, long long M, int N, float* __restrict__ db,
                                                     int rows_per_block) {
    const long long m0 = (long long)blockIdx.x * rows_per_block, m1 = min(M, m0 + rows_per_block);
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
        float s = 0.0f;
        long long m = m0;
        for (; m + 8 <= m1; m += 8) {   // eight loads in flight (a row per trip is one memory round trip per row)
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = dy[(m + u) * N + n];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; m < m1; ++m) s += dy[m * N + n];
        atomicAdd(db + n, s);
    }
}

// Wd[c][tap][n] (n < Npad, zero padded) = W[n][tap][c]: the backward-data GEMM wants K = (tap, n) contiguous per c
__global__ void __launch_bounds__(256) transpose_taps_kernel(const float* __restrict__ w0, const float* __restrict__ w1, int n0, int n1,
                                                             int Npad, int taps, int Cc, float* __restrict__ wd) {
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
        if (n < Npad && c < Cc) wd[((long long)c * taps + tap) * Npad + n] = tile[tx][r];
    }
}

// Wt[tap][c][n] (n < Npad, zero padded) = W[n][tap][c]: rows of the scatter GEMM (n' = tap * Cc + c), K = n contiguous
__global__ void __launch_bounds__(256) transpose_tapmajor_kernel(const float* __restrict__ w0, const float* __restrict__ w1, int n0, int n1,
                                                                 int Npad, int taps, int Cc, float* __restrict__ wt) {
    __shared__ float tile[32][33];
    const int N = n0 + n1;
    const int tap = blockIdx.z;
    const int nb = blockIdx.x * 32, cb = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const int n = nb + r, c = cb + tx;
        float v = 0.0f;
        if (n < N && c < Cc) v = n < n0 ? w0[((long long)n * taps + tap) * Cc + c] : w1[((long long)(n - n0) * taps + tap) * Cc + c];
        tile[r][tx] = v;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int c = cb + r, n = nb + tx;
        if (n < Npad && c < Cc) wt[((long long)tap * Cc + c) * Npad + n] = tile[tx][r];
    }
}

// All weight re-layouts of one heads backward in ONE launch (18 launches of ~5 us each before): a job table, 32x32 tiles.
//   kind 0: out[c][tap][n]  (n < Npad)  = W[n][tap][c]   (dense dgrad)          kind 1: out[tap][c][n] = W[n][tap][c]  (scatter dgrad)
struct TransposeJob {
    const float* w0; const float* w1; float* out;
    int kind, n0, n1, Npad, taps, Cc;
    const int* mode;   // the layout is produced only when *mode == kind (backward mode of the level: 0 dense, 1 pixel rows, 2 anchor rows)
    int tiles_x, tiles_y, block_begin;   // tiles_x * tiles_y * kTrDepth blocks
};
constexpr int kMaxTransposeJobs = 3 * kMaxProblems;   // (ssdk_conv2d_transpose_weights takes up to this many weights per launch)
constexpr int kTrDepth = 3;   // kinds 0, 1: workgroups per (n, c) tile, each walks taps / kTrDepth taps
struct TransposeGroup { int count; TransposeJob j[kMaxTransposeJobs]; };
__global__ void __launch_bounds__(256) transpose_group_kernel(TransposeGroup grp) {
    __shared__ float tile[32][33];
    int ji = 0;
#pragma unroll 1
    for (int i = 1; i < grp.count; ++i)
        if ((int)blockIdx.x >= grp.j[i].block_begin) ji = i;
    const TransposeJob& J = grp.j[ji];
    if (J.mode && *J.mode != J.kind) return;
    int id = blockIdx.x - J.block_begin;
    const int bx = id % J.tiles_x; id /= J.tiles_x;
    const int by = id % J.tiles_y; id /= J.tiles_y;
    const int bz = id;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    // kinds 0, 1: a workgroup walks a third of the taps of its 32 x 32 (n, c) tile.  (A workgroup per tap: 9x the workgroups, which
    // in the usual sparse step are launched only to find their layout is not wanted; one workgroup for all taps: 18 dependent
    // memory round trips, the long pole of the launch whenever a small level does want the layout.)
    const int N = J.n0 + J.n1, nb = bx * 32, cb = by * 32, taps = J.taps, Cc = J.Cc;
    const int per = (taps + kTrDepth - 1) / kTrDepth;
    for (int tap = bz * per; tap < min(taps, (bz + 1) * per); ++tap) {
        for (int r = ty; r < 32; r += 8) {
            const int n = nb + r, c = cb + tx;
            float v = 0.0f;
            if (n < N && c < Cc) v = n < J.n0 ? J.w0[((long long)n * taps + tap) * Cc + c] : J.w1[((long long)(n - J.n0) * taps + tap) * Cc + c];
            tile[r][tx] = v;
        }
        __syncthreads();
        for (int r = ty; r < 32; r += 8) {
            const int c = cb + r, n = nb + tx;
            if (n < J.Npad && c < Cc) J.out[(J.kind == 0 ? ((long long)c * taps + tap) : ((long long)tap * Cc + c)) * J.Npad + n] = tile[tx][r];
        }
        __syncthreads();
    }
}

// zeroes up to 32 buffers in one launch (gradient accumulators that the atomics add into)
constexpr int kMaxZero = 32;
struct ZeroArgs {
    int count;
    float* ptr[kMaxZero];
    unsigned long long n[kMaxZero];  // floats, multiples of 4 use the float4 path
};
__global__ void __launch_bounds__(256) zero_many_kernel(ZeroArgs a) {
    for (int i = blockIdx.y; i < a.count; i += gridDim.y) {
        float* p = a.ptr[i];
        const unsigned long long n = a.n[i];
        if ((n & 3ull) == 0 && ((uintptr_t)p & 15) == 0) {
            const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
            for (unsigned long long k = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; k < (n >> 2); k += (unsigned long long)gridDim.x * blockDim.x)
                reinterpret_cast<float4*>(p)[k] = z;
        } else {
            for (unsigned long long k = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; k < n; k += (unsigned long long)gridDim.x * blockDim.x) p[k] = 0.0f;
        }
    }
}


// =====================================================================================================================
// Split-bf16 FAST MODE of the forward head GEMM (opt-in; the fp32 kernel above stays the default and the parity path).
// Reference analogue: apex AMP O1 (bf/training/env.py:87-95) runs these convolutions in half precision.  Here every fp32 operand is
// split into bf16 pieces a = a_hi + a_mid (+ a_lo, dropped) and the product is the sum of the three largest cross terms
//     a b ~= a_hi b_hi + a_hi b_mid + a_mid b_hi        (relative error of a product ~ 2^-16, accumulated in fp32 on the matrix cores)
// on v_mfma_f32_32x32x16_bf16: 3 x 32 cycles per 32 x 32 x 16 block against 8 x 64 cycles of v_mfma_f32_32x32x2_f32 for the same
// block, i.e. 5.3x fewer matrix-pipe cycles.  The weights are split once per call by split_weights_kernel into two bf16 matrices laid out
// in the launch's column index space ([n][9 * Cin], score rows then loc rows); the activations stay fp32 in HBM and LDS (same LDS-DMA
// staging, same XOR swizzle as dma_tile) and are split in registers after the fragment read.  Tiling, epilogue and the concatenated
// output layout are those of the fp32 kernel (the C / D register map of the 32 x 32 MFMAs does not depend on the input type).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct FastProblem {
    const __bf16* w_hi;   // [n_rows][K] bf16, K = 9 * Cin
    const __bf16* w_mid;
    unsigned w_bytes;     // n_rows * K * 2
};
struct FastGroup {
    FastProblem p[kMaxProblems];
};

struct SplitJob {
    const float* w0;
    const float* w1;
    int n0, n1, n_rows, K;   // rows [0, n0) = w0, [n0, n0 + n1) = w1, the rest zeros
    __bf16* hi;
    __bf16* mid;
    int block_begin;
    // backward-data: the source rows are [taps][tap_len] (the re-laid-out weights [cin][tap][cout]) and the planes get them with the
    // taps in REVERSE order -- the data gradient of a stride-1 convolution is the forward convolution of dy with the mirrored kernel
    int flip_taps, tap_len;
};
struct SplitGroup {
    int count;
    SplitJob j[kMaxProblems];
};

__global__ void __launch_bounds__(256) split_weights_kernel(SplitGroup sg) {
    int ji = 0;
#pragma unroll 1
    for (int i = 1; i < sg.count; ++i)
        if ((int)blockIdx.x >= sg.j[i].block_begin) ji = i;
    const SplitJob& J = sg.j[ji];
    const long long q4 = (long long)J.n_rows * J.K / 4;
    const long long q = (long long)(blockIdx.x - J.block_begin) * 256 + threadIdx.x;
    if (q >= q4) return;
    const long long e = q * 4;
    const int n = (int)(e / J.K);
    long long k = e - (long long)n * J.K;
    if (J.flip_taps) {   // (tap_len % 4 == 0: the four elements stay inside one tap)
        const int t = (int)(k / J.tap_len);
        k = (long long)(J.flip_taps - 1 - t) * J.tap_len + (k - (long long)t * J.tap_len);
    }
    f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
    if (n < J.n0) v = *reinterpret_cast<const f32x4*>(J.w0 + (long long)n * J.K + k);
    else if (n < J.n0 + J.n1) v = *reinterpret_cast<const f32x4*>(J.w1 + (long long)(n - J.n0) * J.K + k);
    __bf16 h[4], m[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        h[t] = (__bf16)v[t];
        m[t] = (__bf16)(v[t] - (float)h[t]);
    }
    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
    *reinterpret_cast<bf16x4*>(J.hi + e) = bf16x4{h[0], h[1], h[2], h[3]};
    *reinterpret_cast<bf16x4*>(J.mid + e) = bf16x4{m[0], m[1], m[2], m[3]};
}

__device__ __forceinline__ void split8(const f32x4& lo4, const f32x4& hi4, bf16x8& a_hi, bf16x8& a_mid) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const __bf16 h0 = (__bf16)lo4[t], h1 = (__bf16)hi4[t];
        a_hi[t] = h0;
        a_hi[4 + t] = h1;
        a_mid[t] = (__bf16)(lo4[t] - (float)h0);
        a_mid[4 + t] = (__bf16)(hi4[t] - (float)h1);
    }
}

__device__ __forceinline__ void fast_tile(const ConvProblem& g, const FastProblem& fp, int m_tile, int n_block) {
    constexpr int BM = 128;
    __shared__ __attribute__((aligned(1024))) float s_a0[BM * kBK];
    __shared__ __attribute__((aligned(1024))) float s_a1[BM * kBK];
    __shared__ __attribute__((aligned(1024))) __bf16 s_w0[2 * kMaxTN * 32 * kBK];   // [plane][128 columns][32 k]
    __shared__ __attribute__((aligned(1024))) __bf16 s_w1[2 * kMaxTN * 32 * kBK];

    const int Cc = g.Cc, ks = g.ksize, taps = ks * ks, chunks = Cc / kBK, n_slices = taps * chunks;
    const int K = taps * Cc;
    const int N = g.n0_pad + g.n1;
    const int hw = g.Hout * g.Wout;
    const int M = g.B * hw;
    const int m_base = m_tile * BM;
    const int base_t = g.tiles_n / g.n_blocks, rem_t = g.tiles_n % g.n_blocks;
    const int tn = base_t + (n_block < rem_t ? 1 : 0);
    const int n_begin = (n_block * base_t + min(n_block, rem_t)) * 32;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r32 = lane & 31, h = lane >> 5;

    // ---- A: as dma_tile (forward): per-lane source offsets of the wave's four pieces, padding taps / rows past M read out of range
    const int a_ps = g.a_pstride, win_ps = g.Win * a_ps;
    const int shift = -g.pad * (win_ps + a_ps);
    unsigned a_vo[4], a_nmask[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = 8 * i + lane / 8;
        const int src_chunk = (lane % 8) ^ ((row >> 1) & 7);
        const int m = m_base + wave * 32 + row;
        a_vo[i] = kOobBit;
        a_nmask[i] = 0;
        if (m < M) {
            const int b = m / hw, r = m % hw;
            const int y = r / g.Wout, x = r % g.Wout;
            unsigned mask = 0;
            int by = y * g.stride - g.pad, bx = x * g.stride - g.pad;
            unsigned colbits = 0;   // (existing kernel rows x existing kernel columns, as dma_tile)
            for (int kx = 0; kx < ks; ++kx) colbits |= ((unsigned)(bx + kx) < (unsigned)g.Win ? 1u : 0u) << kx;
            for (int ky = 0; ky < ks; ++ky)
                if ((unsigned)(by + ky) < (unsigned)g.Hin) mask |= colbits << (ky * ks);
            by += g.pad;
            bx += g.pad;
            a_vo[i] = (unsigned)(b * (int)g.a_bstride + by * win_ps + bx * a_ps + src_chunk * 4) * 4u;
            a_nmask[i] = ~mask;
        }
    }
    // ---- W: 16 pieces per slice (2 planes x 8 pieces of 16 columns x 64 B); wave w issues pieces 4w .. 4w + 3
    unsigned w_vo[4];
    bool w_mid[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int pid = wave * 4 + i;
        const int col = (pid & 7) * 16 + lane / 4;                   // column of the 128-column block
        const int src_chunk = (lane % 4) ^ ((col >> 2) & 3);          // 64-byte rows: chunk position q of column c holds chunk q ^ ((c >> 2) & 3)
        const int n = n_begin + col;
        w_vo[i] = (col < tn * 32 && n < N) ? (unsigned)n * (unsigned)K * 2u + (unsigned)src_chunk * 16u : kOobBit;
        w_mid[i] = pid >= 8;                                          // (uniform)
    }
    const long long a_records = ((long long)g.B * g.a_bstride - shift) * 4;
    const __amdgpu_buffer_rsrc_t rsrc_a =
        __builtin_amdgcn_make_buffer_rsrc((void*)uniform_ptr(g.a + shift), 0, (int)(a_records > 0x7FFFFFFFLL ? 0x7FFFFFFFLL : a_records), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_wh = __builtin_amdgcn_make_buffer_rsrc((void*)uniform_ptr(reinterpret_cast<const float*>(fp.w_hi)), 0, (int)fp.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_wm = __builtin_amdgcn_make_buffer_rsrc((void*)uniform_ptr(reinterpret_cast<const float*>(fp.w_mid)), 0, (int)fp.w_bytes, 0x00020000);

    int ld_tap = 0, ld_chunk = 0, ld_ky = 0, ld_kx = 0;
    unsigned so_a = 0, so_w = 0, tap_bit = 0;
    auto slice_offsets = [&](bool advance) {
        so_a = (unsigned)(ld_ky * win_ps + ld_kx * a_ps + ld_chunk * kBK) * 4u;
        so_w = (unsigned)(ld_tap * Cc + ld_chunk * kBK) * 2u;
        tap_bit = (unsigned)ld_tap;
        const int inc = advance ? 1 : 0;
        ld_tap += inc;
        ld_kx += inc;
        const bool wrap_x = ld_kx == ks, wrap_t = ld_tap == taps;
        ld_kx = (wrap_x || wrap_t) ? 0 : ld_kx;
        ld_ky = wrap_t ? 0 : ld_ky + (wrap_x ? 1 : 0);
        ld_tap = wrap_t ? 0 : ld_tap;
        ld_chunk += wrap_t ? 1 : 0;
    };
    auto stage_a = [&](int DST, int i) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lds_ptr_t)((DST ? s_a1 : s_a0) + (wave * 32 + 8 * i) * kBK), 16,
                                                 a_vo[i] | ((a_nmask[i] >> tap_bit) << 31), so_a, 0, 0);
    };
    auto stage_w = [&](int DST, int i) {
        const int pid = wave * 4 + i;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(w_mid[i] ? rsrc_wm : rsrc_wh, (lds_ptr_t)((DST ? s_w1 : s_w0) + pid * 16 * kBK), 16, w_vo[i], so_w, 0, 0);
    };

    f32x16 acc[kMaxTN];
#pragma unroll
    for (int j = 0; j < kMaxTN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.0f;

    const int fa = (r32 >> 1) & 7;                 // swizzle of this lane's A row
    const int a_row = (wave * 32 + r32) * kBK;     // floats
    slice_offsets(true);
#pragma unroll
    for (int i = 0; i < 4; ++i) { stage_a(0, i); stage_w(0, i); }
    __syncthreads();

    auto k_loop = [&](auto tn_c) {
        constexpr int TN = decltype(tn_c)::value;
        auto body = [&](auto st_c, int slice) {
            constexpr int ST = decltype(st_c)::value;
            slice_offsets(slice + 2 < n_slices);
            const float* sa = ST ? s_a1 : s_a0;
            const __bf16* sw = ST ? s_w1 : s_w0;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                // lane (r, h): A[row r][k = 16 s2 + 8 h .. + 7] = chunks 4 s2 + 2 h and + 1 of its 128-byte row
                const int c0 = 4 * s2 + 2 * h;
                const f32x4 alo = *reinterpret_cast<const f32x4*>(&sa[a_row + ((c0 ^ fa) * 4)]);
                const f32x4 ahi = *reinterpret_cast<const f32x4*>(&sa[a_row + (((c0 + 1) ^ fa) * 4)]);
                bf16x8 wh[TN], wm[TN];
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int col = j * 32 + r32;
                    const int pos = ((2 * s2 + h) ^ ((col >> 2) & 3)) * 8;   // bf16 elements
                    wh[j] = *reinterpret_cast<const bf16x8*>(&sw[col * kBK + pos]);
                    wm[j] = *reinterpret_cast<const bf16x8*>(&sw[(kMaxTN * 32 + col) * kBK + pos]);
                }
                // the next slice's pieces between the two halves of this one's MFMAs (two A + two W per half)
                stage_a(ST ^ 1, 2 * s2);
                stage_w(ST ^ 1, 2 * s2);
                stage_a(ST ^ 1, 2 * s2 + 1);
                stage_w(ST ^ 1, 2 * s2 + 1);
                bf16x8 a_hi, a_mid;
                split8(alo, ahi, a_hi, a_mid);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_mid, wh[j], acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, wm[j], acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, wh[j], acc[j], 0, 0, 0);
                }
            }
            __syncthreads();
        };
        for (int slice = 0; slice < n_slices; slice += 2) {
            body(std::integral_constant<int, 0>{}, slice);
            if (slice + 1 < n_slices) body(std::integral_constant<int, 1>{}, slice + 1);
        }
    };
    switch (tn) {
        case 4: k_loop(std::integral_constant<int, 4>{}); break;
        case 3: k_loop(std::integral_constant<int, 3>{}); break;
        case 2: k_loop(std::integral_constant<int, 2>{}); break;
        default: k_loop(std::integral_constant<int, 1>{}); break;
    }
    conv_epilogue<false>(g, acc, m_base, wave, r32, h, tn, n_begin, M, N, hw, 0, nullptr, 4);
}

__global__ void __launch_bounds__(kConvThreads, SSDK_CONV_WAVES) igemm_bf16x3_kernel(ConvGroup grp, FastGroup fg) {
    int pi = 0;
#pragma unroll 1
    for (int i = 1; i < grp.count; ++i)
        if ((int)blockIdx.x >= grp.p[i].block_begin) pi = i;
    const ConvProblem& g = grp.p[pi];
    if (g.mode && *g.mode != g.want_mode) return;
    const int id = blockIdx.x - g.block_begin;
    const int per_chunk = 8 * g.n_blocks;
    const int chunk = id / per_chunk, within = id % per_chunk;
    const int m_tile = chunk * 8 + (within & 7);
    const int n_block = within >> 3;
    if (m_tile >= g.m_tiles) return;
    fast_tile(g, fg.p[pi], m_tile, n_block);
}

}  // namespace ssdk

using namespace ssdk;

// ---- host side ---------------------------------------------------------------------------------------------------

struct ZeroList {
    ZeroArgs a;
    bool overflow;   // more than kMaxZero buffers between two launches: a caller bug, reported by launch() (never dropped silently)
    ZeroList() : overflow(false) { a.count = 0; }
    void add(float* p, size_t n) {
        if (!p || !n) return;
        if (a.count >= kMaxZero) { overflow = true; return; }
        a.ptr[a.count] = p;
        a.n[a.count] = n;
        ++a.count;
    }
    int launch(hipStream_t s) {
        SSDK_REQUIRE(!overflow, SSDK_E_INVALID, "ZeroList: more than %d buffers queued for one zeroing launch", kMaxZero);
        if (!a.count) return SSDK_OK;
        // workgroups per buffer sized to the largest one (four float4 stores per thread; the heads' 45 MB of weight gradients took
        // 22 us at 256 workgroups), the small buffers' surplus workgroups find nothing to do
        unsigned long long largest = 0;
        for (int i = 0; i < a.count; ++i) largest = std::max(largest, a.n[i]);
        const unsigned long long want = (largest / 4 + 256 * 4 - 1) / (256 * 4);
        const unsigned gx = (unsigned)std::min<unsigned long long>(2048, std::max<unsigned long long>(32, want));
        hipLaunchKernelGGL(zero_many_kernel, dim3(gx, a.count), dim3(256), 0, s, a);
        SSDK_CHECK_LAUNCH("zero_many_kernel");
        a.count = 0;
        return SSDK_OK;
    }
};

static void finish_problem(ConvProblem& g) {
    g.n0_pad = g.n0;
    const int N = g.n0 + g.n1;
    g.tiles_n = cdiv(N, 32);
    g.n_blocks = cdiv(g.tiles_n, kMaxTN);
    g.m_tiles = cdiv(g.B * g.Hout * g.Wout, kBM);
    g.m_tiles256 = cdiv(g.B * g.Hout * g.Wout, 256);
    g.k_splits = 1;
}
static long long problem_block_work(const ConvProblem& g) {
    const int chunks = cdiv(g.Cc, kBK);
    return (long long)g.ksize * g.ksize * chunks * cdiv(g.tiles_n, g.n_blocks) / g.k_splits;
}
// small GEMMs (pyramid tail): too few output tiles to fill 256 CUs -> split K, add partial tiles atomically (the
// caller zeroes the output first).  Not with a fused ReLU (needs the complete sum).
static bool maybe_split_k_any(ConvProblem& g);
static bool maybe_split_k(ConvProblem& g) {
    if (!deterministic()) return maybe_split_k_any(g);
    // deterministic mode: a K split adds its partial tiles with fp32 atomics in hardware order -- never taken (the column-block choices
    // that come without a split are kept)
    ConvProblem t = g;
    maybe_split_k_any(t);
    if (t.k_splits == 1) g = t;
    return false;
}
static bool maybe_split_k_any(ConvProblem& g) {
    const int blocks = cdiv(g.m_tiles, 8) * 8 * g.n_blocks;
    const int slices = g.ksize * g.ksize * cdiv(g.Cc, kBK);
    static const bool old_rules = getenv("SSDK_CONV_OLD_SPLIT") != nullptr;   // (measurement knob: the rules before round 4's sweep)
    // A K chain of eight slices is not worth cutting: a split saves at most ~3 us of it and costs a zero-fill launch, an atomic epilogue
    // and -- for a convolution in front of a BatchNorm -- the statistics pass that a complete tile does in its epilogue
    // (tools/conv_decomp_sweep.py m2det: 1 x 1 256 -> 256 at 16 x 16, batch 16: 19.4 -> 13.6 us with 32-column workgroups and no split)
    static const int no_split_upto = []() { const char* e = getenv("SSDK_CONV_NO_SPLIT_UPTO"); return e ? atoi(e) : 8; }();   // (measurement knob: 16 and 24 measured within noise of 8 on SSD-300 / SSD-512 / M2Det)
    if (g.relu || blocks >= 256 || slices < (old_rules ? 8 : no_split_upto + 1)) return false;
    // Many row tiles, few column blocks (the SSD-300 tail's 1 x 1 512 -> 256 at 18 x 18, batch 32: 81 x 2 tiles of 128 x 128): 64-column
    // workgroups fill the chip WITHOUT splitting K -- no atomic epilogue (3 x the output through 1.3 TB/s of atomics), no zero-fill
    // launch: 57 -> 44 us (tools/conv_decomp_sweep.py; the other tail layers stay within 15 % of their best split)
    if (g.tiles_n >= 4 && g.tiles_n % 2 == 0 && (long long)g.m_tiles * (g.tiles_n / 2) >= 256 && slices <= 32) {
        g.n_blocks = g.tiles_n / 2;
        g.forced = 1;
        return false;
    }
    // Few row tiles and a deep K (the 3 x 3 / 2 layers on 16 x 16 .. 4 x 4 maps: M <= 1 024 rows, 36 .. 72 slices): 32-column workgroups
    // first, then only as many K splits as bring the launch to ~256 workgroups with at least 8 slices each -- the rule below split the
    // M2Det TUM's 256 -> 256 layer at 16 x 16 eighteen ways (49 us; 26 us with 8 column blocks x 4 splits), tools/conv_decomp_sweep.py
    // (round 4: up to 32 row tiles when tiles x column tiles still fit one per CU -- the M2Det TUM's 256 -> 256 layer at 32 x 32 -> 16 x 16,
    // batch 16, took the rule below: 2 column blocks x 8 splits, 73.5 us; 8 column blocks x 2 splits: 52.6)
    static const int wide = []() { const char* e = getenv("SSDK_CONV_SPLIT_WIDE"); return e ? atoi(e) : 256; }();   // (measurement knob)
    if (g.m_tiles >= 3 && (g.m_tiles <= 8 || (!old_rules && g.m_tiles <= 32 && g.m_tiles * g.tiles_n <= wide)) && slices >= 32) {   // (one or two row tiles: the rule below measured as good or better)
        g.n_blocks = g.tiles_n;
        int ks = std::max(2, 256 / std::max(1, g.m_tiles * g.n_blocks));
        ks = std::min(ks, slices / 8);
        if (!old_rules && g.m_tiles > 8 && g.m_tiles * g.n_blocks * ks > 512) ks = 512 / (g.m_tiles * g.n_blocks);   // (never more than two workgroups per CU)
        if (!old_rules && g.m_tiles > 8 && ks < 2) { g.forced = 1; return false; }   // one 32-column workgroup per tile, whole K: no atomics, statistics in the epilogue
        if (ks >= 2) {
            g.k_splits = ks;
            g.forced = 1;
            return true;
        }
        g.n_blocks = cdiv(g.tiles_n, kMaxTN);
    }
    int ks = cdiv(512, blocks);
    if (ks > slices / 4) ks = slices / 4;
    if (ks < 2) return false;
    g.k_splits = ks;
    return true;
}
// Atomic epilogues (split K, scatter) leave a CU at about one 256-byte wave instruction per 50 ns (MI355X_MICROARCH.md, Global
// float atomics): the 256 of a 128-column tile take 13 us -- phase stamps of the pyramid tail's convolutions showed 3 us of
// prologue, 8 us of K loop and 13 us of epilogue.  While the launch is smaller than the chip, halve the columns per workgroup
// instead: twice the workgroups, each with half the atomics, on CUs that were idle.
static void narrow_for_atomics(ConvProblem& g) {
    while (g.n_blocks < g.tiles_n && (long long)g.m_tiles * g.n_blocks * g.k_splits <= 256) g.n_blocks = std::min(g.tiles_n, g.n_blocks * 2);
}

// decides the kernel (LDS-DMA or register staged; 128- or 256-pixel tiles), orders the problems by decreasing work per
// workgroup (longest first), assigns block ranges, launches
constexpr int kStreamKWgs = 512;   // two 64 KB-LDS workgroups per CU x 256 CUs: the most a stream-K launch uses, and the size its workspace is laid out for
struct StreamKWs {
    float* partial;
    unsigned* flags;   // [kStreamKWgs + 2]: a flag per workgroup, then the timeout counter at [kStreamKWgs] (a FIXED place: a launch capped
                       // below kStreamKWgs workgroups -- ssdk_heads_fwd_ex -- shares the workspace with uncapped ones)
    int nwg;
};
// does a column space of N end in a tile of at most 16 columns?
static inline bool half_tile_of(int N) { return N % 32 != 0 && N % 32 <= 16; }
constexpr long long kStreamKMinRange = 24 * kMaxTN;
constexpr int kStreamKMinWgs = 256;
// generic convolutions (pyramid tail, tower, necks): the launches stream-K helps are ONE to two rounds of tiles on 256 CUs (the SSD-300
// tail's 1 x 1 512 -> 256 at 18 x 18: 162 tiles of 128 x 128, split over K three ways with an atomic epilogue before), so their ranges are
// shorter than the heads': 6 K slices of a 128-column block (SSDK_SK_MINRANGE: measurement knob)
static long long streamk_generic_min_range() {
    const char* e = getenv("SSDK_SK_MINRANGE");
    const long long v = e ? atoll(e) : 0;
    return v > 0 ? v : 6 * kMaxTN;
}
// would launch_group run these forward problems in stream-K form? (decided before the caller splits K: a split launch never does)
static bool streamk_would_take(const ConvProblem* probs, int count, bool generic, bool mirror = false) {
    // generic convolutions: OFF unless asked for (SSDK_CONV_STREAMK_GENERIC=1).  Measured on the SSD-300 tail at batch 32
    // (tools/r03_sk_sweep.sh): the 1 x 1 512 -> 256 layer 56 -> 92-115 us and the 3 x 3 / 2 256 -> 512 layer 85 -> 140 us with ranges of
    // 8 .. 32 units -- a launch of ONE round has no tail to even out, and every workgroup then parks and fixes up a 64 KB partial tile
    // Round 4: launches of TWO rounds of tiles and more do take it (the RetinaNet tower's grouped launches: 2 664 tiles of 128 x 128 on 512
    // slots -- the last, partly filled round is what stream-K evens out: 46.40 -> 45.94 ms per step); SSDK_CONV_STREAMK_GENERIC=1: every
    // generic launch that qualifies like a heads launch, =0: none
    static const int generic_mode = []() { const char* e = getenv("SSDK_CONV_STREAMK_GENERIC"); return !e ? -1 : (atoi(e) ? 1 : 0); }();
    // Round 5: the mirrored-tap data gradient can take it too (igemm_streamk_kernel<true>), but does not by default: on the RetinaNet towers'
    // grouped launch (2 688 tiles on 512 slots, 8 launches per step) it measured 1 475 us per launch against 1 478 us for the whole-tile
    // launch and 46.13 / 46.11 against 46.12 / 46.22 ms per step -- nothing to show for the spin-waits.  SSDK_CONV_STREAMK_BWD=1 turns it on.
    static const bool bwd_on = []() { const char* e = getenv("SSDK_CONV_STREAMK_BWD"); return e && atoi(e) != 0; }();
    if (getenv("SSDK_CONV_NO_STREAMK") || (generic && generic_mode == 0) || (mirror && !bwd_on)) return false;
    long long units = 0, blocks = 0;
    for (int i = 0; i < count; ++i) {
        const ConvProblem& g = probs[i];
        if (g.Cc % kBK || g.mode) return false;
        const int N = (g.n1 > 0 ? cdiv(g.n0, 8) * 8 : g.n0) + g.n1, tiles_n = cdiv(N, 32);
        const int half = (!mirror && !g.stats && half_tile_of(N) && !getenv("SSDK_CONV_NO_HALF_TILE")) ? 1 : 0;   // (as launch_group will set half_last)
        units += (long long)g.m_tiles * g.ksize * g.ksize * (g.Cc / kBK) * (2 * tiles_n - half);           // half-tile units, as StreamK counts
        blocks += (long long)cdiv(g.m_tiles, 8) * 8 * cdiv(tiles_n, kMaxTN);
    }
    const long long min_range = generic ? streamk_generic_min_range() : kStreamKMinRange;
    const long long nwg = std::min<long long>(512, units / (2 * min_range) / 8 * 8);
    // (two rounds and more, the last one at most three quarters full: a launch of whole rounds -- the M2Det neck's 2 048- and 3 584-tile
    // layers -- has no tail to even out and measured 0.1 ms slower per step with the fix-up traffic)
    if (generic && generic_mode < 0 && (blocks < 2 * kStreamKWgs || blocks % kStreamKWgs == 0 || blocks % kStreamKWgs > 3 * kStreamKWgs / 4)) return false;
    return nwg >= kStreamKMinWgs && blocks <= 16 * nwg;
}
static unsigned g_streamk_epoch = 0;   // (a launch counter: tells this launch's flags from an earlier launch's in the same workspace)
// A stream-K owner that gave up on a parked partner says so in a word of pinned, device-visible HOST memory (one per process, allocated
// on first use outside stream capture, never freed): reading it costs the host nothing, so every ssdk_heads_fwd checks it first and fails
// from then on.  The kernel also poisons the tile with NaN and counts the event in the workspace (ssdk_heads_fwd_timeouts).
static unsigned* g_sk_host_err = nullptr;
static bool g_sk_host_err_tried = false;
static unsigned* streamk_host_err_word(hipStream_t s) {
    if (!g_sk_host_err && !__atomic_load_n(&g_sk_host_err_tried, __ATOMIC_ACQUIRE)) {
        hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &st) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        if (st != hipStreamCaptureStatusNone) return nullptr;   // (an allocation is not capturable; the warm-up calls in front of a capture made it)
        static std::mutex mu;
        std::lock_guard<std::mutex> lock(mu);
        if (!g_sk_host_err && !g_sk_host_err_tried) {
            void* p = nullptr;
            if (hipHostMalloc(&p, 64, hipHostMallocMapped | hipHostMallocPortable) == hipSuccess && p) {
                *static_cast<volatile unsigned*>(p) = 0u;
                g_sk_host_err = static_cast<unsigned*>(p);
            } else {
                (void)hipGetLastError();
            }
            __atomic_store_n(&g_sk_host_err_tried, true, __ATOMIC_RELEASE);
        }
    }
    return g_sk_host_err;
}

static int launch_group(ConvProblem* probs, int count, bool mirror, hipStream_t s, bool generic = false, bool scatter = false, int* vtab = nullptr,
                        const StreamKWs* skws = nullptr) {
    bool vec4 = true, strided = false;
    for (int i = 0; i < count; ++i) {
        const ConvProblem& g = probs[i];
        if (g.Cc % 4 || g.a_pstride % 4 || g.a_bstride % 4 || ((uintptr_t)g.a & 15) || ((uintptr_t)g.w0 & 15) || (g.w1 && ((uintptr_t)g.w1 & 15))) vec4 = false;
        strided = strided || g.stride != 1;
    }
    // LDS-DMA kernel: 16-byte rows, whole 32-channel chunks, stride-1 taps when mirrored, all byte offsets below 2^31
    bool dma = vec4 && !(mirror && strided) && !getenv("SSDK_CONV_NO_DMA");
    for (int i = 0; i < count && dma; ++i) {
        ConvProblem& g = probs[i];
        const long long span_a = ((long long)g.B * g.a_bstride + (long long)(g.ksize + g.pad) * ((long long)g.Win + 1) * g.a_pstride) * 4;
        const long long w0_bytes = (long long)g.n0 * (scatter ? 1 : g.ksize * g.ksize) * g.Cc * 4, w1_bytes = (long long)g.n1 * g.ksize * g.ksize * g.Cc * 4;
        if (g.Cc % kBK || span_a >= (1LL << 31) - 4096 || w0_bytes >= (1LL << 31) - 4096 || w1_bytes >= (1LL << 31) - 4096) { dma = false; break; }
        g.w0_bytes = (unsigned)w0_bytes;
        g.w1_bytes = (unsigned)w1_bytes;
    }
    // 16-float K slices (3 workgroups per CU): opt-in experiment for the plain forward launch
    const bool bk16 = dma && !mirror && !generic && !scatter && getenv("SSDK_CONV_BK16");
    // 192-column workgroups (6 column tiles) for the plain forward launch: opt-in experiment
    const bool tn6 = dma && !mirror && !generic && !scatter && !bk16 && getenv("SSDK_CONV_TN6");
    for (int i = 0; i < count; ++i) {   // column space of the chosen kernel (see ConvProblem::n0_pad)
        ConvProblem& g = probs[i];
        g.n0_pad = (dma && g.n1 > 0) ? cdiv(g.n0, bk16 ? 16 : 8) * (bk16 ? 16 : 8) : g.n0;
        g.tiles_n = cdiv(g.n0_pad + g.n1, 32);
        g.half_last = (dma && !mirror && !scatter && !bk16 && !tn6 && !g.stats && half_tile_of(g.n0_pad + g.n1) && !getenv("SSDK_CONV_W8") &&
                       !getenv("SSDK_CONV_NO_HALF_TILE")) ? 1 : 0;
        if (!g.forced) {
            g.n_blocks = cdiv(g.tiles_n, tn6 ? 6 : kMaxTN);
            if (!vtab && (scatter || g.k_splits > 1)) narrow_for_atomics(g);
        }
    }
    // A launch smaller than the chip, whatever its epilogue: halve the columns per workgroup while the workgroups still fit one per CU.
    // A 128 x 128 x 32 slice is 1.7 us of MFMA on one CU, so the 1 x 1 data gradients of the pyramid tail (K = 128: 4 slices, 6 .. 100
    // workgroups of 128 columns) spent 9 us in the K loop and 6.6 us storing four column tiles on a chip that was 60-98 % idle
    // (tools/phase_conv.py bwd); at 32 columns a slice costs its DMA latency (~0.85 us) instead.  The column partition does not change
    // any sum's order: same bits.
    // (not for the heads' forward launch: its stream-K partition is sized from the 128-column blocks)
    if (!vtab && !skws && !getenv("SSDK_CONV_NO_NARROW")) {
        static const long long fill = []() { const char* e = getenv("SSDK_CONV_NARROW_TO"); const long long v = e ? atoll(e) : 0; return v > 0 ? v : 256LL; }();   // (measurement knob)
        long long total = 0;
        for (int i = 0; i < count; ++i) total += (long long)probs[i].m_tiles * probs[i].n_blocks * probs[i].k_splits;
        for (bool again = true; again && total < fill;) {
            again = false;
            for (int i = 0; i < count; ++i) {
                ConvProblem& g = probs[i];
                if (g.forced || g.n_blocks >= g.tiles_n) continue;
                const int nb = std::min(g.tiles_n, g.n_blocks * 2);
                const long long grown = total + (long long)g.m_tiles * (nb - g.n_blocks) * g.k_splits;
                if (grown > fill) continue;
                g.n_blocks = nb;
                total = grown;
                again = true;
            }
        }
    }
    // 8-wave / 256-pixel tiling: measured 3 % (B=128) to 14 % (B=32) MORE cycles than two 4-wave workgroups per CU on the
    // SSD-300 heads (one barrier stalls all eight waves of the CU at once) -- kept as an opt-in experiment only
    bool w8 = false;
    if (dma && !scatter && getenv("SSDK_CONV_W8")) {
        long long blocks256 = 0;
        for (int i = 0; i < count; ++i) blocks256 += (long long)probs[i].m_tiles256 * probs[i].n_blocks * probs[i].k_splits;
        w8 = blocks256 >= 384;
    }
    ConvGroup grp;
    int order[kMaxProblems];
    for (int i = 0; i < count; ++i) order[i] = i;
    for (int i = 0; i < count; ++i)
        for (int j = i + 1; j < count; ++j)
            if (problem_block_work(probs[order[j]]) > problem_block_work(probs[order[i]])) { int t = order[i]; order[i] = order[j]; order[j] = t; }
    int begin = 0;
    for (int i = 0; i < count; ++i) {
        ConvProblem& g = probs[order[i]];
        g.block_begin = begin;
        begin += cdiv(w8 ? g.m_tiles256 : g.m_tiles, 8) * 8 * g.n_blocks * g.k_splits;
        grp.p[i] = g;
    }
    grp.count = count;
    grp.total_blocks = begin;
    grp.vtab = nullptr;
    if (dma && w8) {
        if (mirror) hipLaunchKernelGGL((igemm_dma_kernel<true, false, false, 8>), dim3(begin), dim3(512), 0, s, grp);
        else if (generic) hipLaunchKernelGGL((igemm_dma_kernel<false, true, false, 8>), dim3(begin), dim3(512), 0, s, grp);
        else hipLaunchKernelGGL((igemm_dma_kernel<false, false, false, 8>), dim3(begin), dim3(512), 0, s, grp);
    } else if (dma && scatter && vtab) {
        // sparse backward: only the row tiles that exist (listed on the device), walked by a fixed grid of 8 workgroups per CU
        VtabArgs va{};
        va.count = count;
        va.vtab = vtab;
        for (int i = 0; i < count; ++i) {
            const ConvProblem& g = grp.p[i];
            va.p[i].counts = g.row_count;
            va.p[i].mode = g.mode; va.p[i].want = g.want_mode; va.p[i].n_blocks = g.n_blocks;
            SSDK_REQUIRE(va.p[i].counts && g.k_splits == 1, SSDK_E_INVALID, "launch_group: a tile list needs device-side row counts and no split-K");
        }
        hipLaunchKernelGGL(build_vtab_kernel, dim3(1), dim3(64), 0, s, va);
        SSDK_CHECK_LAUNCH("build_vtab_kernel");
        grp.vtab = vtab;
        hipLaunchKernelGGL((igemm_dma_kernel<false, false, true, 4>), dim3(2048), dim3(kConvThreads), 0, s, grp);
    } else if (dma && skws && !scatter && !bk16 && !tn6 && !getenv("SSDK_CONV_NO_STREAMK")) {
        // stream-K only where it pays: a launch of a few rounds of whole tiles (its last round is then a large share of the time), and
        // every range at least as long as the longest tile (a tile is cut at most once)
        StreamK sk{};
        sk.nwg = skws->nwg;
        long long max_tile = 0;
        for (int i = 0; i < count; ++i) {
            const ConvProblem& g = grp.p[i];
            const long long slices = (long long)g.ksize * g.ksize * (g.Cc / kBK);
            sk.unit_begin[i] = sk.total_units;
            sk.total_units += (long long)g.m_tiles * slices * (2 * g.tiles_n - g.half_last);
            max_tile = std::max(max_tile, slices * cdiv(g.tiles_n, g.n_blocks));
            if (g.k_splits != 1 || g.mode) sk.nwg = 0;
        }
        sk.unit_begin[count] = sk.total_units;
        // Not for launches of many rounds (the tail is then a small share and whole tiles need no fix-up).  Otherwise as many workgroups as
        // leave each a range of at least kStreamKMinRange units (24 K slices of a 128-column block): a tile longer than a range is cut
        // several times and its owner adds all the parked parts.  Below 256 workgroups the split-K path of the caller does as well (measured on ssd_mb2_voc).
        if (sk.nwg > 0 && begin > 16 * sk.nwg) sk.nwg = 0;
        const long long min_range = generic ? streamk_generic_min_range() : kStreamKMinRange;
        if (sk.nwg > 0) sk.nwg = (int)std::min<long long>(sk.nwg, sk.total_units / (2 * min_range) / 8 * 8);   // (min_range counts whole tiles)
        (void)max_tile;
        const bool worth = sk.nwg >= kStreamKMinWgs;
        if (worth) {
            sk.partial = skws->partial;
            sk.flags = skws->flags;
            sk.timeouts = skws->flags + kStreamKWgs;   // (behind the flags of the largest launch)
            sk.host_err = streamk_host_err_word(s);
            sk.epoch = __atomic_add_fetch(&g_streamk_epoch, 1u, __ATOMIC_RELAXED);
            if (sk.epoch == 0) sk.epoch = __atomic_add_fetch(&g_streamk_epoch, 1u, __ATOMIC_RELAXED);   // (0 is what a fresh workspace holds)
            if (mirror) hipLaunchKernelGGL(igemm_streamk_kernel<true>, dim3(sk.nwg), dim3(kConvThreads), 0, s, grp, sk);
            else hipLaunchKernelGGL(igemm_streamk_kernel<false>, dim3(sk.nwg), dim3(kConvThreads), 0, s, grp, sk);
        } else if (mirror) {
            hipLaunchKernelGGL((igemm_dma_kernel<true, false, false, 4>), dim3(begin), dim3(kConvThreads), 0, s, grp);
        } else if (generic) {
            hipLaunchKernelGGL((igemm_dma_kernel<false, true, false, 4>), dim3(begin), dim3(kConvThreads), 0, s, grp);
        } else {
            hipLaunchKernelGGL((igemm_dma_kernel<false, false, false, 4>), dim3(begin), dim3(kConvThreads), 0, s, grp);
        }
    } else if (dma) {
        // every workgroup of the launch owns ONE 32-column tile (the launches narrowed above, the atomic-epilogue launches of the small
        // maps): the three-stage instantiation -- its K loop does not wait for a DMA issued one slice earlier but two
        bool one_tile = !bk16 && !tn6 && !getenv("SSDK_CONV_NO_3STAGE");
        for (int i = 0; i < count && one_tile; ++i) one_tile = grp.p[i].n_blocks == grp.p[i].tiles_n && !grp.p[i].half_last;
        if (one_tile && scatter) hipLaunchKernelGGL((igemm_dma_kernel<false, false, true, 4, 32, 1>), dim3(begin), dim3(kConvThreads), 0, s, grp);
        else if (one_tile && mirror) hipLaunchKernelGGL((igemm_dma_kernel<true, false, false, 4, 32, 1>), dim3(begin), dim3(kConvThreads), 0, s, grp);
        else if (one_tile && generic) hipLaunchKernelGGL((igemm_dma_kernel<false, true, false, 4, 32, 1>), dim3(begin), dim3(kConvThreads), 0, s, grp);
        else if (scatter) hipLaunchKernelGGL((igemm_dma_kernel<false, false, true, 4>), dim3(begin), dim3(kConvThreads), 0, s, grp);
        else if (mirror) hipLaunchKernelGGL((igemm_dma_kernel<true, false, false, 4>), dim3(begin), dim3(kConvThreads), 0, s, grp);
        else if (generic) hipLaunchKernelGGL((igemm_dma_kernel<false, true, false, 4>), dim3(begin), dim3(kConvThreads), 0, s, grp);
        else if (bk16) hipLaunchKernelGGL((igemm_dma_kernel<false, false, false, 4, 16>), dim3(begin), dim3(kConvThreads), 0, s, grp);
        else if (tn6) hipLaunchKernelGGL((igemm_dma_kernel<false, false, false, 4, 32, 6>), dim3(begin), dim3(kConvThreads), 0, s, grp);
        else hipLaunchKernelGGL((igemm_dma_kernel<false, false, false, 4>), dim3(begin), dim3(kConvThreads), 0, s, grp);
    } else if (scatter) {
        SSDK_REQUIRE(vec4, SSDK_E_UNSUPPORTED, "scatter dgrad needs 16-byte aligned rows");
        hipLaunchKernelGGL((igemm_fwd_kernel<4, false, false, false, true>), dim3(begin), dim3(kConvThreads), 0, s, grp);
    } else if (mirror && strided) {
        if (vec4) hipLaunchKernelGGL((igemm_fwd_kernel<4, true, true>), dim3(begin), dim3(kConvThreads), 0, s, grp);
        else hipLaunchKernelGGL((igemm_fwd_kernel<1, true, true>), dim3(begin), dim3(kConvThreads), 0, s, grp);
    } else if (mirror) {
        if (vec4) hipLaunchKernelGGL((igemm_fwd_kernel<4, true>), dim3(begin), dim3(kConvThreads), 0, s, grp);
        else hipLaunchKernelGGL((igemm_fwd_kernel<1, true>), dim3(begin), dim3(kConvThreads), 0, s, grp);
    } else if (generic) {  // same code, separate instantiation: profiles list the extras/tower convs apart from the heads
        if (vec4) hipLaunchKernelGGL((igemm_fwd_kernel<4, false, false, true>), dim3(begin), dim3(kConvThreads), 0, s, grp);
        else hipLaunchKernelGGL((igemm_fwd_kernel<1, false, false, true>), dim3(begin), dim3(kConvThreads), 0, s, grp);
    } else {
        if (vec4) hipLaunchKernelGGL((igemm_fwd_kernel<4, false>), dim3(begin), dim3(kConvThreads), 0, s, grp);
        else hipLaunchKernelGGL((igemm_fwd_kernel<1, false>), dim3(begin), dim3(kConvThreads), 0, s, grp);
    }
    SSDK_CHECK_LAUNCH("igemm_fwd_kernel");
    return SSDK_OK;
}

static int check_level(const char* fn, int batch, const ssdk_head_level& lv) {
    SSDK_REQUIRE(batch > 0 && lv.h > 0 && lv.w > 0 && lv.cin > 0 && lv.n_score > 0 && lv.n_loc >= 0, SSDK_E_INVALID,
                 "%s: batch=%d H=%d W=%d Cin=%d n_score=%d n_loc=%d", fn, batch, lv.h, lv.w, lv.cin, lv.n_score, lv.n_loc);
    SSDK_REQUIRE((long long)batch * lv.h * lv.w < (1LL << 31) - kBM, SSDK_E_INVALID, "%s: too many pixels", fn);
    SSDK_REQUIRE(lv.x && lv.w_score && (lv.n_loc == 0 || lv.w_loc), SSDK_E_INVALID, "%s: null pointer", fn);
    return SSDK_OK;
}

static inline int npad_of(const ssdk_head_level& lv) { return cdiv(lv.n_score + lv.n_loc, 32) * 32; }
// anchor types per pixel (0: unknown -- no loc head, or a layout the anchor-granular backward does not cover)
// A single head (n_loc == 0: the score tower's or the loc tower's convolution of a SharedConvPredictor level, detector.py:50-66) has no loc
// part to read the count from: its caller may pass it in locs_offset (meaningless otherwise for such a level; 0 = unknown).
static inline int anchor_types_of(const ssdk_head_level& lv) {
    if (lv.n_loc < 0 || lv.n_loc % 4) return 0;
    const long long nb = lv.n_loc > 0 ? lv.n_loc / 4 : lv.locs_offset;
    return (nb > 0 && nb < kMaxAnchorTypes && lv.n_score % nb == 0 && lv.cin % kBK == 0) ? (int)nb : 0;
}
static inline int loc_cols_of(const ssdk_head_level& lv) { return lv.n_loc > 0 ? 4 : 0; }
static inline int jpad_of(const ssdk_head_level& lv) {   // (only where anchor_types_of(lv) != 0)
    return cdiv(lv.n_score / anchor_types_of(lv) + loc_cols_of(lv), 32) * 32;
}

extern "C" size_t ssdk_heads_fwd_workspace_bytes(void) {
    return align_up((size_t)(kStreamKWgs + 1) * (4 * kMaxTN * 4 * 64 * 4) * sizeof(float), 256) + align_up((size_t)(kStreamKWgs + 2) * sizeof(unsigned), 256);
}

extern "C" int ssdk_heads_fwd(const ssdk_head_level* levels, int n_levels, int batch, float* scores,
                              long long scores_batch_stride, float* locs, long long locs_batch_stride, void* workspace,
                              size_t workspace_bytes, void* stream) {
    return ssdk_heads_fwd_ex(levels, n_levels, batch, scores, scores_batch_stride, locs, locs_batch_stride, 0, workspace, workspace_bytes, stream);
}

extern "C" int ssdk_heads_fwd_ex(const ssdk_head_level* levels, int n_levels, int batch, float* scores,
                                 long long scores_batch_stride, float* locs, long long locs_batch_stride, int max_workgroups,
                                 void* workspace, size_t workspace_bytes, void* stream) {
    SSDK_REQUIRE(max_workgroups >= 0, SSDK_E_INVALID, "ssdk_heads_fwd_ex: max_workgroups=%d", max_workgroups);
    SSDK_REQUIRE(levels && n_levels > 0 && n_levels <= kMaxProblems, SSDK_E_INVALID, "ssdk_heads_fwd: n_levels=%d (1..%d)", n_levels, kMaxProblems);
    SSDK_REQUIRE(scores, SSDK_E_INVALID, "ssdk_heads_fwd: null scores");
    SSDK_REQUIRE(!g_sk_host_err || *static_cast<volatile unsigned*>(g_sk_host_err) == 0u, SSDK_E_STREAMK_TIMEOUT,
                 "ssdk_heads_fwd: an earlier stream-K launch of this process gave up waiting for a parked partial tile (its output tile "
                 "was filled with NaN); the card is oversubscribed or a workgroup never ran -- see ssdk_heads_fwd_timeouts");
    ConvProblem probs[kMaxProblems];
    for (int i = 0; i < n_levels; ++i) {
        const ssdk_head_level& lv = levels[i];
        int rc = check_level("ssdk_heads_fwd", batch, lv);
        if (rc) return rc;
        SSDK_REQUIRE(lv.n_loc == 0 || locs, SSDK_E_INVALID, "ssdk_heads_fwd: null locs");
        ConvProblem g{};
        g.a = lv.x; g.a_bstride = (long long)lv.h * lv.w * lv.cin; g.a_pstride = lv.cin; g.Cc = lv.cin;
        g.B = batch; g.Hout = lv.h; g.Wout = lv.w; g.Hin = lv.h; g.Win = lv.w; g.ksize = 3; g.stride = 1; g.pad = 1;
        g.w0 = lv.w_score; g.w1 = lv.n_loc ? lv.w_loc : nullptr; g.bias0 = lv.b_score; g.bias1 = lv.b_loc; g.n0 = lv.n_score; g.n1 = lv.n_loc;
        g.o0 = scores + lv.scores_offset; g.ob0 = scores_batch_stride; g.os0 = lv.n_score;
        g.o1 = lv.n_loc ? locs + lv.locs_offset : nullptr; g.ob1 = locs_batch_stride; g.os1 = lv.n_loc;
        g.relu = 0;
        finish_problem(g);
        probs[i] = g;
    }
    // Small batches: a level has a handful of row tiles, each with a K chain of 9 * Cin / 32 slices (1.5 us apiece) -- ssd_mb2_voc at batch
    // 2 spent 534 us in 24 workgroups.  While the launch is too small for stream-K (at most one workgroup per slot), the K slices of such
    // a level are divided over several workgroups that add into the zeroed outputs, like the pyramid tail's convolutions.
    long long blocks = 0;
    for (int i = 0; i < n_levels; ++i) blocks += (long long)cdiv(probs[i].m_tiles, 8) * 8 * probs[i].n_blocks;
    bool dma_ok = true;   // (what launch_group asks of its LDS-DMA kernel; stream-K is a form of it)
    for (int i = 0; i < n_levels; ++i) dma_ok = dma_ok && probs[i].Cc % kBK == 0;
    const bool have_ws = workspace && workspace_bytes >= ssdk_heads_fwd_workspace_bytes();
    const bool streamk_takes_it = have_ws && dma_ok && streamk_would_take(probs, n_levels, false);
    if (blocks <= kStreamKWgs && !streamk_takes_it && !getenv("SSDK_HEADS_NO_SPLITK")) {
        bool any = false;
        for (int i = 0; i < n_levels; ++i) any = maybe_split_k(probs[i]) || any;
        if (any) {
            ZeroList zl;
            for (int which = 0; which < 2; ++which) {
                // the split levels' segments of one image's row, merged where they touch (levels are usually laid out back to back)
                long long off[kMaxProblems], len[kMaxProblems];
                int ns = 0;
                for (int i = 0; i < n_levels; ++i) {
                    const ssdk_head_level& lv = levels[i];
                    const long long n = (long long)lv.h * lv.w * (which ? lv.n_loc : lv.n_score);
                    if (probs[i].k_splits > 1 && n > 0) { off[ns] = which ? lv.locs_offset : lv.scores_offset; len[ns] = n; ++ns; }
                }
                for (int a = 0; a < ns; ++a)
                    for (int b = a + 1; b < ns; ++b)
                        if (off[b] < off[a]) { std::swap(off[a], off[b]); std::swap(len[a], len[b]); }
                int m = 0;
                for (int a = 0; a < ns; ++a) {
                    if (m && off[m - 1] + len[m - 1] == off[a]) len[m - 1] += len[a];
                    else { off[m] = off[a]; len[m] = len[a]; ++m; }
                }
                float* const base = which ? locs : scores;
                const long long stride = which ? locs_batch_stride : scores_batch_stride;
                for (int b = 0; b < batch; ++b)
                    for (int a = 0; a < m; ++a) {
                        if (zl.a.count == kMaxZero) { const int rc = zl.launch((hipStream_t)stream); if (rc) return rc; }
                        zl.add(base + (long long)b * stride + off[a], (size_t)len[a]);
                    }
            }
            const int rc = zl.launch((hipStream_t)stream);
            if (rc) return rc;
        }
    }
    StreamKWs sk{};
    if (workspace && workspace_bytes >= ssdk_heads_fwd_workspace_bytes()) {
        Carver c(workspace);
        sk.partial = c.take<float>((size_t)(kStreamKWgs + 1) * (4 * kMaxTN * 4 * 64 * 4));
        sk.flags = c.take<unsigned>((size_t)kStreamKWgs + 2);
        sk.nwg = kStreamKWgs;
        // a caller that runs other kernels BESIDE this launch (the pyramid tail on a second stream) leaves them LDS slots: the persistent
        // workgroups of the stream-K form otherwise hold every slot of the chip until the launch ends
        if (max_workgroups > 0) sk.nwg = std::max(kStreamKMinWgs, std::min(kStreamKWgs, max_workgroups / 8 * 8));
    }
    return launch_group(probs, n_levels, false, (hipStream_t)stream, false, false, nullptr, sk.nwg ? &sk : nullptr);
}

extern "C" int ssdk_streamk_poisoned(void) {
    return (g_sk_host_err && *static_cast<volatile unsigned*>(g_sk_host_err) != 0u) ? 1 : 0;
}

// Recovery (a trainer that restarts from a checkpoint inside the process): clears the workspace's flags and timeout counter and the
// process-wide sticky host word.  Synchronises the stream first: nothing of an earlier launch may still be polling or parking.
// HIP graphs captured before the fault replay healthy launches again (the kernel reads the counter at entry).
extern "C" int ssdk_streamk_reset(void* workspace, size_t workspace_bytes, void* stream) {
    SSDK_REQUIRE(workspace && workspace_bytes >= ssdk_heads_fwd_workspace_bytes(), SSDK_E_WORKSPACE,
                 "ssdk_streamk_reset: the workspace of ssdk_heads_fwd (ssdk_heads_fwd_workspace_bytes() bytes) is required");
    SSDK_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
    Carver c(workspace);
    c.take<float>((size_t)(kStreamKWgs + 1) * (4 * kMaxTN * 4 * 64 * 4));
    unsigned* flags = c.take<unsigned>((size_t)kStreamKWgs + 2);
    SSDK_CHECK_HIP(zero_async(flags, ((size_t)kStreamKWgs + 2) * sizeof(unsigned), (hipStream_t)stream));
    SSDK_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
    if (g_sk_host_err) *static_cast<volatile unsigned*>(g_sk_host_err) = 0u;
    return SSDK_OK;
}

// (fault injection is honoured only when the process asked for it: SSDK_ENABLE_FAULT_INJECTION=1 in the environment -- the two device
// globals it sets are read by every stream-K launch)
extern "C" int ssdk_debug_streamk_fault(int drop_workgroup, unsigned spin_limit) {
    const char* en = getenv("SSDK_ENABLE_FAULT_INJECTION");
    SSDK_REQUIRE(en && atoi(en) != 0, SSDK_E_UNSUPPORTED, "ssdk_debug_streamk_fault: fault injection is off (set SSDK_ENABLE_FAULT_INJECTION=1: tests only)");
    hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(g_sk_drop_wg), &drop_workgroup, sizeof(int));
    if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(g_sk_spin_limit), &spin_limit, sizeof(unsigned));
    SSDK_REQUIRE(e == hipSuccess, (int)e, "ssdk_debug_streamk_fault: %s", hipGetErrorString(e));
    return SSDK_OK;
}

extern "C" int ssdk_heads_fwd_timeouts(const void* workspace, size_t workspace_bytes, void* stream, unsigned* timeouts_host) {
    SSDK_REQUIRE(workspace && timeouts_host && workspace_bytes >= ssdk_heads_fwd_workspace_bytes(), SSDK_E_WORKSPACE,
                 "ssdk_heads_fwd_timeouts: workspace of ssdk_heads_fwd_workspace_bytes() bytes and a host word are required");
    Carver c(const_cast<void*>(workspace));
    c.take<float>((size_t)(kStreamKWgs + 1) * (4 * kMaxTN * 4 * 64 * 4));
    const unsigned* flags = c.take<unsigned>((size_t)kStreamKWgs + 2);
    SSDK_CHECK_HIP(hipMemcpyAsync(timeouts_host, flags + kStreamKWgs, sizeof(unsigned), hipMemcpyDeviceToHost, (hipStream_t)stream));
    SSDK_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
    if (g_sk_host_err && *static_cast<volatile unsigned*>(g_sk_host_err) && *timeouts_host == 0u) *timeouts_host = 1u;   // (another workspace's)
    return SSDK_OK;
}


// ---- fast mode of the forward heads (opt-in): bf16 x 3 split operands on v_mfma_f32_32x32x16_bf16, fp32 accumulate ----------------
static inline int fast_rows_of(const ssdk_head_level& lv) { return cdiv(lv.n_score + lv.n_loc, 32) * 32; }

extern "C" size_t ssdk_heads_fwd_fast_workspace_bytes(const ssdk_head_level* levels, int n_levels) {
    if (!levels || n_levels <= 0 || n_levels > kMaxProblems) return 0;
    Carver c(nullptr);
    for (int i = 0; i < n_levels; ++i) {
        const size_t elems = (size_t)fast_rows_of(levels[i]) * 9 * (size_t)levels[i].cin;
        c.take<__bf16>(elems);
        c.take<__bf16>(elems);
    }
    return c.off;
}

extern "C" int ssdk_heads_fwd_fast(const ssdk_head_level* levels, int n_levels, int batch, float* scores, long long scores_batch_stride,
                                   float* locs, long long locs_batch_stride, int terms, void* workspace, size_t workspace_bytes, void* stream) {
    SSDK_REQUIRE(levels && n_levels > 0 && n_levels <= kMaxProblems, SSDK_E_INVALID, "ssdk_heads_fwd_fast: n_levels=%d (1..%d)", n_levels, kMaxProblems);
    SSDK_REQUIRE(terms == 3, SSDK_E_UNSUPPORTED, "ssdk_heads_fwd_fast: terms=%d (3 = a_hi b_hi + a_hi b_mid + a_mid b_hi is the one form built)", terms);
    SSDK_REQUIRE(scores, SSDK_E_INVALID, "ssdk_heads_fwd_fast: null scores");
    SSDK_REQUIRE(workspace && workspace_bytes >= ssdk_heads_fwd_fast_workspace_bytes(levels, n_levels), SSDK_E_WORKSPACE, "ssdk_heads_fwd_fast: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    Carver c(workspace);
    ConvProblem probs[kMaxProblems];
    FastProblem fps[kMaxProblems];
    SplitGroup sg{};
    int split_blocks = 0;
    for (int i = 0; i < n_levels; ++i) {
        const ssdk_head_level& lv = levels[i];
        int rc = check_level("ssdk_heads_fwd_fast", batch, lv);
        if (rc) return rc;
        SSDK_REQUIRE(lv.n_loc == 0 || locs, SSDK_E_INVALID, "ssdk_heads_fwd_fast: null locs");
        SSDK_REQUIRE(lv.cin % kBK == 0 && ((uintptr_t)lv.x & 15) == 0 && ((uintptr_t)lv.w_score & 15) == 0 && (!lv.n_loc || ((uintptr_t)lv.w_loc & 15) == 0),
                     SSDK_E_UNSUPPORTED, "ssdk_heads_fwd_fast: level %d needs Cin %% 32 == 0 and 16-byte aligned operands (Cin=%d)", i, lv.cin);
        const int K = 9 * lv.cin, rows = fast_rows_of(lv);
        const long long span_a = ((long long)batch * lv.h * lv.w * lv.cin + (long long)(3 + 1) * ((long long)lv.w + 1) * lv.cin) * 4;
        SSDK_REQUIRE(span_a < (1LL << 31) - 4096 && (long long)rows * K * 2 < (1LL << 31) - 4096, SSDK_E_UNSUPPORTED, "ssdk_heads_fwd_fast: level %d too large for 32-bit buffer offsets", i);
        ConvProblem g{};
        g.a = lv.x; g.a_bstride = (long long)lv.h * lv.w * lv.cin; g.a_pstride = lv.cin; g.Cc = lv.cin;
        g.B = batch; g.Hout = lv.h; g.Wout = lv.w; g.Hin = lv.h; g.Win = lv.w; g.ksize = 3; g.stride = 1; g.pad = 1;
        g.w0 = lv.w_score; g.w1 = lv.n_loc ? lv.w_loc : nullptr; g.bias0 = lv.b_score; g.bias1 = lv.b_loc; g.n0 = lv.n_score; g.n1 = lv.n_loc;
        g.o0 = scores + lv.scores_offset; g.ob0 = scores_batch_stride; g.os0 = lv.n_score;
        g.o1 = lv.n_loc ? locs + lv.locs_offset : nullptr; g.ob1 = locs_batch_stride; g.os1 = lv.n_loc;
        g.relu = 0;
        finish_problem(g);   // n0_pad = n0: the split weights are ONE matrix in the column index space, no padding between the two heads
        probs[i] = g;
        __bf16* hi = c.take<__bf16>((size_t)rows * K);
        __bf16* mid = c.take<__bf16>((size_t)rows * K);
        fps[i].w_hi = hi; fps[i].w_mid = mid; fps[i].w_bytes = (unsigned)((size_t)rows * K * 2);
        SplitJob& J = sg.j[sg.count++];
        J.w0 = lv.w_score; J.w1 = lv.n_loc ? lv.w_loc : nullptr; J.n0 = lv.n_score; J.n1 = lv.n_loc; J.n_rows = rows; J.K = K; J.hi = hi; J.mid = mid;
        J.block_begin = split_blocks;
        split_blocks += (int)(((long long)rows * K / 4 + 255) / 256);
    }
    hipLaunchKernelGGL(split_weights_kernel, dim3(split_blocks), dim3(256), 0, s, sg);
    SSDK_CHECK_LAUNCH("split_weights_kernel");
    // grouped launch, problems ordered by decreasing work per workgroup (as launch_group)
    int order[kMaxProblems];
    for (int i = 0; i < n_levels; ++i) order[i] = i;
    for (int i = 0; i < n_levels; ++i)
        for (int j = i + 1; j < n_levels; ++j)
            if (problem_block_work(probs[order[j]]) > problem_block_work(probs[order[i]])) { int t = order[i]; order[i] = order[j]; order[j] = t; }
    ConvGroup grp;
    FastGroup fg;
    int begin = 0;
    for (int i = 0; i < n_levels; ++i) {
        ConvProblem& g = probs[order[i]];
        g.block_begin = begin;
        begin += cdiv(g.m_tiles, 8) * 8 * g.n_blocks;
        grp.p[i] = g;
        fg.p[i] = fps[order[i]];
    }
    grp.count = n_levels;
    grp.total_blocks = begin;
    grp.vtab = nullptr;
    hipLaunchKernelGGL(igemm_bf16x3_kernel, dim3(begin), dim3(kConvThreads), 0, s, grp, fg);
    SSDK_CHECK_LAUNCH("igemm_bf16x3_kernel");
    return SSDK_OK;
}

// ---- the same split-bf16 GEMM for the generic convolutions (RetinaNet's towers, neck and tail convolutions with Cin % 32 == 0) ----------
// out_dim / check_conv are defined further down with ssdk_conv2d_fwd
static int check_conv(const char* fn, int batch, const ssdk_conv_desc& d);
static inline int out_dim(int in, int k, int s, int p);

extern "C" size_t ssdk_conv2d_fwd_fast_workspace_bytes(const ssdk_conv_desc* descs, int n) {
    if (!descs || n <= 0 || n > kMaxProblems) return 0;
    Carver c(nullptr);
    for (int i = 0; i < n; ++i) {   // (an upper bound: descriptors that share a weight tensor share its two planes)
        const size_t elems = (size_t)cdiv(descs[i].cout, 32) * 32 * descs[i].ksize * descs[i].ksize * (size_t)descs[i].cin;
        c.take<__bf16>(elems);
        c.take<__bf16>(elems);
    }
    return c.off;
}

extern "C" int ssdk_conv2d_fwd_fast(const ssdk_conv_desc* descs, int n, int batch, int terms, void* workspace, size_t workspace_bytes, void* stream) {
    SSDK_REQUIRE(descs && n > 0 && n <= kMaxProblems, SSDK_E_INVALID, "ssdk_conv2d_fwd_fast: n=%d (1..%d)", n, kMaxProblems);
    SSDK_REQUIRE(terms == 3, SSDK_E_UNSUPPORTED, "ssdk_conv2d_fwd_fast: terms=%d (3 is the one form built)", terms);
    SSDK_REQUIRE(workspace && workspace_bytes >= ssdk_conv2d_fwd_fast_workspace_bytes(descs, n), SSDK_E_WORKSPACE, "ssdk_conv2d_fwd_fast: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    Carver c(workspace);
    ConvProblem probs[kMaxProblems];
    FastProblem fps[kMaxProblems];
    SplitGroup sg{};
    int split_blocks = 0;
    for (int i = 0; i < n; ++i) {
        const ssdk_conv_desc& d = descs[i];
        int rc = check_conv("ssdk_conv2d_fwd_fast", batch, d);
        if (rc) return rc;
        SSDK_REQUIRE(d.y, SSDK_E_INVALID, "ssdk_conv2d_fwd_fast: null output");
        SSDK_REQUIRE(d.cin % kBK == 0 && ((uintptr_t)d.x & 15) == 0 && ((uintptr_t)d.w & 15) == 0, SSDK_E_UNSUPPORTED,
                     "ssdk_conv2d_fwd_fast: descriptor %d needs Cin %% 32 == 0 and 16-byte aligned operands (Cin=%d)", i, d.cin);
        const int ho = out_dim(d.hin, d.ksize, d.stride, d.pad), wo = out_dim(d.win, d.ksize, d.stride, d.pad);
        const int K = d.ksize * d.ksize * d.cin, rows = cdiv(d.cout, 32) * 32;
        const long long span_a = ((long long)batch * d.hin * d.win * d.cin + (long long)(d.ksize + d.pad) * ((long long)d.win + 1) * d.cin) * 4;
        SSDK_REQUIRE(span_a < (1LL << 31) - 4096 && (long long)rows * K * 2 < (1LL << 31) - 4096, SSDK_E_UNSUPPORTED,
                     "ssdk_conv2d_fwd_fast: descriptor %d too large for 32-bit buffer offsets", i);
        ConvProblem g{};
        g.a = d.x; g.a_bstride = (long long)d.hin * d.win * d.cin; g.a_pstride = d.cin; g.Cc = d.cin;
        g.B = batch; g.Hout = ho; g.Wout = wo; g.Hin = d.hin; g.Win = d.win; g.ksize = d.ksize; g.stride = d.stride; g.pad = d.pad;
        g.w0 = d.w; g.w1 = nullptr; g.bias0 = d.bias; g.bias1 = nullptr; g.n0 = d.cout; g.n1 = 0;
        g.o0 = d.y; g.ob0 = (long long)ho * wo * d.cout; g.os0 = d.cout; g.o1 = nullptr; g.ob1 = 0; g.os1 = 0;
        g.relu = d.relu;
        finish_problem(g);
        probs[i] = g;
        int same = -1;   // a tower layer's weights are shared by all pyramid levels: split them once
        for (int j = 0; j < i && same < 0; ++j)
            if (descs[j].w == d.w && descs[j].cout == d.cout && descs[j].cin == d.cin && descs[j].ksize == d.ksize) same = j;
        if (same >= 0) { fps[i] = fps[same]; continue; }
        __bf16* hi = c.take<__bf16>((size_t)rows * K);
        __bf16* mid = c.take<__bf16>((size_t)rows * K);
        fps[i].w_hi = hi; fps[i].w_mid = mid; fps[i].w_bytes = (unsigned)((size_t)rows * K * 2);
        SplitJob& J = sg.j[sg.count++];
        J.w0 = d.w; J.w1 = nullptr; J.n0 = d.cout; J.n1 = 0; J.n_rows = rows; J.K = K; J.hi = hi; J.mid = mid;
        J.block_begin = split_blocks;
        split_blocks += (int)(((long long)rows * K / 4 + 255) / 256);
    }
    hipLaunchKernelGGL(split_weights_kernel, dim3(split_blocks), dim3(256), 0, s, sg);
    SSDK_CHECK_LAUNCH("split_weights_kernel");
    int order[kMaxProblems];
    for (int i = 0; i < n; ++i) order[i] = i;
    for (int i = 0; i < n; ++i)
        for (int j = i + 1; j < n; ++j)
            if (problem_block_work(probs[order[j]]) > problem_block_work(probs[order[i]])) { int t = order[i]; order[i] = order[j]; order[j] = t; }
    ConvGroup grp;
    FastGroup fg;
    int begin = 0;
    for (int i = 0; i < n; ++i) {
        ConvProblem& g = probs[order[i]];
        g.block_begin = begin;
        begin += cdiv(g.m_tiles, 8) * 8 * g.n_blocks;
        grp.p[i] = g;
        fg.p[i] = fps[order[i]];
    }
    grp.count = n;
    grp.total_blocks = begin;
    grp.vtab = nullptr;
    hipLaunchKernelGGL(igemm_bf16x3_kernel, dim3(begin), dim3(kConvThreads), 0, s, grp, fg);
    SSDK_CHECK_LAUNCH("igemm_bf16x3_kernel");
    for (int i = 0; i < n; ++i) {   // BatchNorm statistics of the outputs (ssdk_conv_desc::stats): a pass of their own, as after a split-K launch
        const ssdk_conv_desc& d = descs[i];
        if (!d.stats) continue;
        SSDK_REQUIRE(d.cout % 4 == 0 && ((uintptr_t)d.y & 15) == 0, SSDK_E_UNSUPPORTED, "ssdk_conv2d_fwd_fast: stats need cout %% 4 == 0 and a 16-byte aligned output");
        const long long rows = (long long)batch * out_dim(d.hin, d.ksize, d.stride, d.pad) * out_dim(d.win, d.ksize, d.stride, d.pad);
        const int rc = ssdk_batchnorm_stats_accumulate(d.y, rows, d.cout, d.stats, stream);
        if (rc) return rc;
    }
    return SSDK_OK;
}

// Fast-mode (split-bf16) launch of forward-form problems whose weight planes `fps` are produced by the split jobs of `sg` (both filled
// by the caller): one split launch, one grouped GEMM launch ordered by work per workgroup.
static int launch_fast_group(ConvProblem* probs, FastProblem* fps, int n, SplitGroup& sg, int split_blocks, hipStream_t s) {
    if (sg.count) {
        hipLaunchKernelGGL(split_weights_kernel, dim3(split_blocks), dim3(256), 0, s, sg);
        SSDK_CHECK_LAUNCH("split_weights_kernel");
    }
    int order[kMaxProblems];
    for (int i = 0; i < n; ++i) order[i] = i;
    for (int i = 0; i < n; ++i)
        for (int j = i + 1; j < n; ++j)
            if (problem_block_work(probs[order[j]]) > problem_block_work(probs[order[i]])) { int t = order[i]; order[i] = order[j]; order[j] = t; }
    ConvGroup grp;
    FastGroup fg;
    int begin = 0;
    for (int i = 0; i < n; ++i) {
        ConvProblem& g = probs[order[i]];
        g.block_begin = begin;
        begin += cdiv(g.m_tiles, 8) * 8 * g.n_blocks;
        grp.p[i] = g;
        fg.p[i] = fps[order[i]];
    }
    grp.count = n;
    grp.total_blocks = begin;
    grp.vtab = nullptr;
    hipLaunchKernelGGL(igemm_bf16x3_kernel, dim3(begin), dim3(kConvThreads), 0, s, grp, fg);
    SSDK_CHECK_LAUNCH("igemm_bf16x3_kernel");
    return SSDK_OK;
}
// fast mode: launches below this many FLOPs stay fp32 (SSDK_FAST_MIN_FLOPS: tests set 0 to put small shapes through the bf16 kernel)
static double fast_min_flops() {
    const char* e = getenv("SSDK_FAST_MIN_FLOPS");
    return e ? atof(e) : 1.0e9;
}
// can a stride-1 data gradient -- the forward-form convolution of `a` [batch][h][w][ch] with a ksize x ksize kernel, `wrows` weight
// rows of K = ksize^2 * ch -- take the fast kernel? (whole 32-float K slices, 16-byte rows, 32-bit buffer offsets)
static bool fast_conv_ok(const float* a, int batch, int h, int w, int ch, int ksize, int pad, int wrows) {
    const long long span_a = ((long long)batch * h * w * ch + (long long)(ksize + pad) * ((long long)w + 1) * ch) * 4;
    const long long K = (long long)ksize * ksize * ch;
    return ch % kBK == 0 && ((uintptr_t)a & 15) == 0 && span_a < (1LL << 31) - 4096 && (long long)cdiv(wrows, 32) * 32 * K * 2 < (1LL << 31) - 4096;
}

constexpr int kColsumBlocks = 256;   // deterministic bias gradients: at most this many per-workgroup partial column sums per tensor

// workspace of the legacy pipeline (levels without anchor structure, or SSDK_HEADS_BWD_MODE=1): dense or pixel-row backward
struct HeadsBwdWs {
    float* dyp[kMaxProblems];
    float* wd[kMaxProblems];
    float* wt[kMaxProblems];
    int* row_list[kMaxProblems];
    int* vtab;    // tile list of the pixel-sparse scatter launch
    int* counts;  // [kMaxProblems] non-zero gradient rows per level
    int* totals;  // [kMaxProblems] pixel rows per level
    int* mode;    // [kMaxProblems] 1 = pixel-sparse backward, 0 = dense
    // deterministic mode only: per level the K-split copies of the weight gradient [k_splits][N][9*Cin] and the per-workgroup column sums
    // of the bias gradient [kColsumBlocks][N]
    float* dw_part[kMaxProblems];
    int dw_splits[kMaxProblems];
    float* db_part[kMaxProblems];
    // fast mode only (ssdk_heads_bwd_fast): the two bf16 planes of the mirrored kernel [cin][tap'][Npad] per level
    __bf16* fast_hi[kMaxProblems];
    __bf16* fast_mid[kMaxProblems];
};

static void size_wgrad_splits(WgradGroup& wg, int n, int density_div);
// the dense weight-gradient problem of one head level (dy / row lists / outputs are filled in by the caller)
static WgradProblem heads_wgrad_problem(const ssdk_head_level& lv, int batch) {
    WgradProblem g{};
    g.x = lv.x; g.Npad = cdiv(lv.n_score + lv.n_loc, 32) * 32; g.Cc = lv.cin;
    g.B = batch; g.Hout = lv.h; g.Wout = lv.w; g.Hin = lv.h; g.Win = lv.w; g.ksize = 3; g.stride = 1; g.pad = 1;
    g.dw0 = lv.dw_score; g.dw1 = lv.dw_loc; g.n0 = lv.n_score; g.n1 = lv.n_loc;
    g.n_tiles = cdiv(lv.n_score + lv.n_loc, 128);
    g.c_tiles32 = cdiv(lv.cin, 32);
    g.c_blocks = cdiv(g.c_tiles32, kMaxTN);
    return g;
}

static HeadsBwdWs carve_heads_bwd(void* ws, const ssdk_head_level* levels, int n_levels, int batch, size_t* total, bool fast = false) {
    Carver c(ws);
    HeadsBwdWs w{};
    w.counts = c.take<int>(kMaxProblems);
    w.vtab = c.take<int>(kVtabInts);
    w.totals = c.take<int>(kMaxProblems);
    w.mode = c.take<int>(kMaxProblems);
    for (int i = 0; i < n_levels; ++i) {
        const ssdk_head_level& lv = levels[i];
        const size_t npad = (size_t)npad_of(lv), M = (size_t)batch * lv.h * lv.w;
        w.dyp[i] = c.take<float>(M * npad);
        w.wd[i] = c.take<float>((size_t)lv.cin * 9 * npad);
        w.wt[i] = c.take<float>((size_t)lv.cin * 9 * npad);
        w.row_list[i] = c.take<int>(M);
        if (deterministic()) {
            WgradGroup one{};
            one.p[0] = heads_wgrad_problem(lv, batch);
            size_wgrad_splits(one, 1, 1);
            w.dw_splits[i] = one.p[0].k_splits;
            w.dw_part[i] = c.take<float>((size_t)one.p[0].k_splits * (lv.n_score + lv.n_loc) * 9 * lv.cin);
            w.db_part[i] = c.take<float>((size_t)kColsumBlocks * (lv.n_score + lv.n_loc));
        }
        if (fast) {
            w.fast_hi[i] = c.take<__bf16>((size_t)cdiv(lv.cin, 32) * 32 * 9 * npad);
            w.fast_mid[i] = c.take<__bf16>((size_t)cdiv(lv.cin, 32) * 32 * 9 * npad);
        }
    }
    if (total) *total = c.off;
    return w;
}

// K (= pixel rows) is split so that every problem contributes >= ~256 workgroups (tiny maps are latency bound at one
// wave per SIMD: more, shorter workgroups) and no workgroup walks more than 64 slices; each split costs one 64 KB
// atomic tile, so never fewer than 2 slices per split.
// picks the LDS-DMA kernel when every problem of the group qualifies (16-byte rows, operands below 2 GiB, pixel count below 2^24)
static int launch_wgrad(WgradGroup& wg, hipStream_t s, bool fast = false, bool rows32 = false) {
    bool dma = !getenv("SSDK_CONV_NO_DMA");
    for (int i = 0; i < wg.count && dma; ++i) {
        WgradProblem& g = wg.p[i];
        const long long rows = g.seg_count ? g.seg_cap : (long long)g.B * g.Hout * g.Wout;
        const long long dy_bytes = rows * g.Npad * 4, x_bytes = (long long)g.B * g.Hin * g.Win * g.Cc * 4;
        if (g.Npad % 4 || g.Cc % 4 || ((uintptr_t)g.dy & 15) || ((uintptr_t)g.x & 15) || dy_bytes >= (1LL << 31) - 65536 || x_bytes >= (1LL << 31) - 65536 ||
            (long long)g.B * g.Hout * g.Wout >= (1 << 24)) { dma = false; break; }
        g.dy_bytes = (unsigned)dy_bytes;
        g.x_bytes = (unsigned)x_bytes;
    }
    for (int i = 0; i < wg.count; ++i)
        SSDK_REQUIRE(dma || !(wg.p[i].ordered && !wg.p[i].det_stride) , SSDK_E_UNSUPPORTED, "launch_wgrad: ordered stores need the LDS-DMA kernel");
    if (dma && fast) hipLaunchKernelGGL(igemm_wgrad_bf16x3_kernel, dim3(wg.total_blocks), dim3(kConvThreads), 0, s, wg);
    else if (dma && rows32) {
        // one launch per row-matrix width (Npad / 32 MFMAs per K step, a compile-time count: no zero rows multiplied, no branches in the K loop);
        // the levels of a model share their class count, so this is one launch
        for (int nq = 1; nq <= 4; ++nq) {
            WgradGroup sub{};
            int begin = 0;
            for (int i = 0; i < wg.count; ++i) {
                if (cdiv(wg.p[i].Npad, 32) != nq) continue;
                const int blocks = (i + 1 < wg.count ? wg.p[i + 1].block_begin : wg.total_blocks) - wg.p[i].block_begin;
                WgradProblem& q = sub.p[sub.count++];
                q = wg.p[i];
                q.block_begin = begin;
                begin += blocks;
            }
            if (!sub.count) continue;
            sub.total_blocks = begin;
            if (nq == 1) hipLaunchKernelGGL(igemm_wgrad_rows_kernel<1>, dim3(begin), dim3(kConvThreads), 0, s, sub);
            else if (nq == 2) hipLaunchKernelGGL(igemm_wgrad_rows_kernel<2>, dim3(begin), dim3(kConvThreads), 0, s, sub);
            else if (nq == 3) hipLaunchKernelGGL(igemm_wgrad_rows_kernel<3>, dim3(begin), dim3(kConvThreads), 0, s, sub);
            else hipLaunchKernelGGL(igemm_wgrad_rows_kernel<4>, dim3(begin), dim3(kConvThreads), 0, s, sub);
        }
    }
    else if (dma) hipLaunchKernelGGL(igemm_wgrad_dma_kernel, dim3(wg.total_blocks), dim3(kConvThreads), 0, s, wg);
    else hipLaunchKernelGGL(igemm_wgrad_kernel, dim3(wg.total_blocks), dim3(kConvThreads), 0, s, wg);
    SSDK_CHECK_LAUNCH("igemm_wgrad_kernel");
    return SSDK_OK;
}

// Host side of the deterministic reductions: jobs for reduce_partials_kernel, launched in groups of kMaxReduceJobs.
struct ReduceList {
    ReduceGroup g;
    int blocks;
    ReduceList() : blocks(0) { g.count = 0; }
    int add(float* dst, const float* src, long long elems, long long stride, int n_src, int accumulate, hipStream_t s, const int* mode = nullptr,
            int want_mode = 0) {
        if (!dst || elems <= 0 || n_src <= 0) return SSDK_OK;
        if (g.count == kMaxReduceJobs) { const int rc = launch(s); if (rc) return rc; }
        ReduceJob& J = g.j[g.count++];
        J.dst = dst; J.src = src; J.elems = elems; J.stride = stride; J.n_src = n_src; J.accumulate = accumulate; J.block_begin = blocks;
        J.mode = mode; J.want_mode = want_mode;
        blocks += (int)((elems + 1023) / 1024);
        return SSDK_OK;
    }
    int launch(hipStream_t s) {
        if (!g.count) return SSDK_OK;
        hipLaunchKernelGGL(reduce_partials_kernel, dim3(blocks), dim3(256), 0, s, g);
        SSDK_CHECK_LAUNCH("reduce_partials_kernel");
        g.count = 0;
        blocks = 0;
        return SSDK_OK;
    }
};
// K splits of a dense weight-gradient problem that actually get rows (the last of `k_splits` equal ranges can be empty)
static int wgrad_used_splits(const WgradProblem& g) {
    const int slices = cdiv(g.B * g.Hout * g.Wout, 32);
    const int per = cdiv(slices, g.k_splits);
    return cdiv(slices, per);
}
static void size_wgrad_splits(WgradGroup& wg, int n, int density_div) {
    int begin = 0;
    for (int i = 0; i < n; ++i) {
        WgradProblem& g = wg.p[i];
        const int slices = cdiv(cdiv(g.B * g.Hout * g.Wout, density_div), 32);
        // (small problems -- tiles x row slices <= 1024 --: 32-channel instead of 128-channel workgroups first -- a quarter of the atomics each, see narrow_for_atomics --
        // and only then split K)
        while (g.c_blocks < g.c_tiles32 && (long long)g.ksize * g.ksize * g.n_tiles * g.c_blocks * (g.seg_count ? g.segs : 1) * slices <= 1024) g.c_blocks = std::min(g.c_tiles32, g.c_blocks * 2);
        const int tiles = g.ksize * g.ksize * g.n_tiles * g.c_blocks * (g.seg_count ? g.segs : 1);
        int ks = cdiv(256, tiles);   // (512 measured 6 % slower over the step's weight-gradient launches, 128 10 % slower)
        if (cdiv(slices, 64) > ks) ks = cdiv(slices, 64);
        if (ks > slices / 2) ks = slices / 2;
        if (ks < 1) ks = 1;
        g.k_splits = ks;
        g.block_begin = begin;
        begin += tiles * ks;
    }
    wg.count = n;
    wg.total_blocks = begin;
}

// ---- the ordered pipeline (every level has anchor structure: a loc head with nb = n_loc / 4 anchor types, n_score = nb * C, C + 4 <= 128) ----
// Per level the DEVICE picks the form (anchor_plan_kernel): 2 = anchor rows (see "Ordered anchor-row backward" above) when the rows fit the
// level's T buffer, else 0 = dense.  Both forms sum in an order fixed by the launch: no atomics, no zero-fill, the same bits on every run
// and in a HIP-graph replay, whatever ssdk_set_deterministic says (it governs the legacy pipeline below and the generic convolutions).
static int heads_t_div() {
    static const int v = []() { const char* e = getenv("SSDK_HEADS_T_DIV"); const int d = e ? atoi(e) : 0; return d >= 1 ? d : 4; }();
    return v;
}
// rows of a level's T buffer: a quarter of its anchors (hard-negative mining marks ~4 %), never fewer than 4 096 (small maps always fit)
static inline long long t_rows_of(const ssdk_head_level& lv, int batch) {
    const long long all = (long long)anchor_types_of(lv) * batch * lv.h * lv.w;
    return std::min(all, std::max<long long>(4096, (all + heads_t_div() - 1) / heads_t_div()));
}
static bool ordered_heads_ok(const ssdk_head_level* levels, int n_levels, int batch) {
    if (getenv("SSDK_CONV_NO_DMA") || getenv("SSDK_HEADS_BWD_LEGACY")) return false;
    const char* f = getenv("SSDK_HEADS_BWD_MODE");
    if (f && atoi(f) == 1) return false;   // (the pixel-row form exists in the legacy pipeline only)
    for (int i = 0; i < n_levels; ++i) {
        const ssdk_head_level& lv = levels[i];
        const int nb = anchor_types_of(lv);
        if (!nb || lv.cin % 4 || ((uintptr_t)lv.x & 15) || ((uintptr_t)lv.w_score & 15) || ((uintptr_t)lv.w_loc & 15)) return false;
        const long long M = (long long)batch * lv.h * lv.w, jpad = jpad_of(lv);
        if (jpad > 128 || M >= (1 << 24) || M * jpad * 4 >= (1LL << 31) - 65536 || M * lv.cin * 4 >= (1LL << 31) - 65536 ||
            M * npad_of(lv) * 4 >= (1LL << 31) - 65536) return false;   // (every operand of the LDS-DMA weight-gradient kernel below 2 GiB)
        if (t_rows_of(lv, batch) * 9 * lv.cin >= (1LL << 32)) return false;
        if (npad_of(lv) > kPackColIters * 64) return false;
    }
    return true;
}
// K splits of the anchor-row weight gradient of a level: a fixed function of the shapes (sized for 1 / 16 of the anchors of a type carrying a
// gradient, ~64 slices of 32 rows per split, at most 8; every split stores a full copy of the level's weight gradient).  Measured on
// SSD-300 / 81 classes, batch 32 (level 0: ~55 slices per type; kernel + reduction, us): 1 split 150 + 10, 2 splits 123 + 14, 3: 158 + 14,
// 5: 163 + 15, 6: 141 + 21 (tools/r05_stats.sh with SSDK_ANCHOR_WGRAD_SLICES = 128 / 64 / 32 / 22 / 16; before the K loop lost its per-MFMA
// `q < nq` branches -- the kernel is now instantiated per row-matrix width -- the same sweep read 224 / 160 / 208 / - / 173)
static inline int anchor_wgrad_splits(const ssdk_head_level& lv, int batch) {
    const int slices = cdiv(cdiv(batch * lv.h * lv.w, 16), 32);
    static const int per = []() { const char* e = getenv("SSDK_ANCHOR_WGRAD_SLICES"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 64; }();   // (measurement knob)
    return std::max(1, std::min(8, cdiv(slices, per)));
}
struct OrderedWs {
    int* acounts; int* plan; int* mode; int* gtab; float* zeros;
    unsigned char* imask[kMaxProblems];
    int* blk[kMaxProblems]; int* apix[kMaxProblems]; int* aidx[kMaxProblems];
    float* ga[kMaxProblems]; float* T[kMaxProblems]; float* dbp[kMaxProblems];
    float* dyp[kMaxProblems]; float* wd[kMaxProblems];
    float* dw_part[kMaxProblems];
    int dense_splits[kMaxProblems], anchor_splits[kMaxProblems];
    long long tcap[kMaxProblems];
    __bf16* fast_hi[kMaxProblems];
    __bf16* fast_mid[kMaxProblems];
};
static OrderedWs carve_heads_ordered(void* ws, const ssdk_head_level* levels, int n_levels, int batch, size_t* total, bool fast) {
    Carver c(ws);
    OrderedWs w{};
    w.acounts = c.take<int>(kMaxProblems * kAT);
    w.plan = c.take<int>(kPlanInts);
    w.mode = c.take<int>(kMaxProblems);
    w.gtab = c.take<int>(kGtabInts);
    w.zeros = c.take<float>(64);
    for (int i = 0; i < n_levels; ++i) {
        const ssdk_head_level& lv = levels[i];
        const size_t M = (size_t)batch * lv.h * lv.w, nb = (size_t)anchor_types_of(lv), jpad = (size_t)jpad_of(lv), npad = (size_t)npad_of(lv);
        const size_t N = (size_t)lv.n_score + lv.n_loc, K9 = (size_t)9 * lv.cin;
        w.imask[i] = c.take<unsigned char>(M * nb);
        w.blk[i] = c.take<int>((size_t)cdiv((int)M, kAnchorBlk) * kAT);
        w.apix[i] = c.take<int>(nb * M);
        w.aidx[i] = c.take<int>(nb * M);
        w.ga[i] = c.take<float>(nb * M * jpad);
        w.tcap[i] = t_rows_of(lv, batch);
        w.T[i] = c.take<float>((size_t)w.tcap[i] * K9);
        w.dbp[i] = c.take<float>(nb * (size_t)cdiv((int)M, kGatherRows) * jpad);
        w.dyp[i] = c.take<float>(M * npad);
        w.wd[i] = c.take<float>((size_t)lv.cin * 9 * npad);
        {   // the K-split copies of the weight gradient: the dense form's or the anchor form's, whichever is larger (a level takes one)
            WgradGroup one{};
            one.p[0] = heads_wgrad_problem(lv, batch);
            size_wgrad_splits(one, 1, 1);
            w.dense_splits[i] = one.p[0].k_splits;
            w.anchor_splits[i] = anchor_wgrad_splits(lv, batch);
            w.dw_part[i] = c.take<float>((size_t)std::max(w.dense_splits[i], w.anchor_splits[i] > 1 ? w.anchor_splits[i] : 0) * N * K9);
        }
        if (fast) {
            w.fast_hi[i] = c.take<__bf16>((size_t)cdiv(lv.cin, 32) * 32 * 9 * npad);
            w.fast_mid[i] = c.take<__bf16>((size_t)cdiv(lv.cin, 32) * 32 * 9 * npad);
        }
    }
    if (total) *total = c.off;
    return w;
}

// TEST HOOK: where the ordered pipeline keeps its intermediates of level `level` inside the workspace (byte offsets), so that tests can hold
// the row matrix, the inverted index and the T rows against a CPU restatement.  out[0..7] = ga, T, aidx, apix, acounts (of this level),
// plan (of this level: T row base / tile base per type), mode (of this level), T capacity in rows.  Returns 0, or -3 when these levels do
// not take the ordered pipeline.
extern "C" int ssdk_debug_heads_bwd_layout(const ssdk_head_level* levels, int n_levels, int batch, int level, unsigned long long* out) {
    SSDK_REQUIRE(levels && out && n_levels > 0 && n_levels <= kMaxProblems && level >= 0 && level < n_levels && batch > 0, SSDK_E_INVALID, "ssdk_debug_heads_bwd_layout: bad arguments");
    if (!ordered_heads_ok(levels, n_levels, batch)) return SSDK_E_UNSUPPORTED;
    char* const base = reinterpret_cast<char*>(0x1000);   // (never dereferenced: the carver only adds offsets)
    const OrderedWs w = carve_heads_ordered(base, levels, n_levels, batch, nullptr, false);
    auto off = [&](const void* p) { return (unsigned long long)(reinterpret_cast<const char*>(p) - base); };
    out[0] = off(w.ga[level]); out[1] = off(w.T[level]); out[2] = off(w.aidx[level]); out[3] = off(w.apix[level]);
    out[4] = off(w.acounts + level * kAT); out[5] = off(w.plan + kPlanHead + level * kPlanStride); out[6] = off(w.mode + level);
    out[7] = (unsigned long long)w.tcap[level];
    return SSDK_OK;
}

static int heads_bwd_ordered(const ssdk_head_level* levels, int n_levels, int batch, const float* dscores, long long scores_batch_stride,
                             const float* dlocs, long long locs_batch_stride, void* workspace, hipStream_t s, bool fast,
                             const unsigned char* row_mask, int num_anchors) {
    OrderedWs w = carve_heads_ordered(workspace, levels, n_levels, batch, nullptr, fast);
    // 1. which anchors carry a gradient: the caller's mask where it covers the level in the level's own numbering, else derived
    AnchorGroup ag{};
    ag.count = n_levels;
    {
        const char* f = getenv("SSDK_HEADS_BWD_MODE");
        ag.force = f ? atoi(f) : -1;
    }
    ag.acounts = w.acounts; ag.plan = w.plan; ag.mode = w.mode; ag.gtab = w.gtab; ag.zeros = w.zeros;
    MaskGroup mg{};
    mg.B = batch; mg.sb = scores_batch_stride; mg.lb = locs_batch_stride;
    int blk_begin = 0, mask_blocks = 0;
    for (int i = 0; i < n_levels; ++i) {
        const ssdk_head_level& lv = levels[i];
        int rc = check_level("ssdk_heads_bwd", batch, lv);
        if (rc) return rc;
        SSDK_REQUIRE(dlocs || lv.n_loc == 0, SSDK_E_INVALID, "ssdk_heads_bwd: null dlocs");
        const int nb = anchor_types_of(lv), C = lv.n_score / nb, hw = lv.h * lv.w, LQ = loc_cols_of(lv);
        AnchorLevel& L = ag.lv[i];
        L.nb = nb; L.HW = hw; L.cap = batch * hw;
        L.blk_begin = blk_begin; L.nblk = cdiv(L.cap, kAnchorBlk);
        blk_begin += L.nblk;
        L.blk = w.blk[i]; L.apix = w.apix[i]; L.aidx = w.aidx[i];
        L.tcap = (int)std::min<long long>(w.tcap[i], 0x7fffffff);
        L.items_per_tile = lv.dx ? cdiv(9 * lv.cin, kRgChunkCols) : 0;
        // (a single head -- n_loc == 0 -- is covered when its columns start at a whole anchor: its locs_offset carries the anchor-type count)
        const bool covered = row_mask && lv.scores_offset % C == 0 && (LQ == 0 || lv.locs_offset == 4 * (lv.scores_offset / C)) &&
                             lv.scores_offset / C + (long long)hw * nb <= num_anchors;
        if (covered && !getenv("SSDK_PACK_SCAN")) {
            L.rmask = row_mask; L.a_total = num_anchors; L.a_off = (int)(lv.scores_offset / C);
        } else {
            L.rmask = w.imask[i]; L.a_total = hw * nb; L.a_off = 0;
            MaskLevel& Q = mg.lv[mg.count++];
            Q.ds = dscores + lv.scores_offset; Q.dl = LQ ? dlocs + lv.locs_offset : nullptr; Q.out = w.imask[i]; Q.nb = nb; Q.C = C; Q.HW = hw; Q.LQ = LQ;
            Q.blk_begin = mask_blocks;
            mask_blocks += cdiv(L.cap, kMaskRows);
        }
    }
    if (mg.count) {
        hipLaunchKernelGGL(anchor_mask_kernel, dim3(mask_blocks), dim3(256), 0, s, mg);
        SSDK_CHECK_LAUNCH("anchor_mask_kernel");
    }
    hipLaunchKernelGGL(anchor_count_kernel, dim3(blk_begin), dim3(kAnchorBlk), 0, s, ag);
    SSDK_CHECK_LAUNCH("anchor_count_kernel");
    hipLaunchKernelGGL(anchor_plan_kernel, dim3(1), dim3(1024), 0, s, ag);
    SSDK_CHECK_LAUNCH("anchor_plan_kernel");
    hipLaunchKernelGGL(anchor_fill_kernel, dim3(blk_begin), dim3(kAnchorBlk), 0, s, ag);
    SSDK_CHECK_LAUNCH("anchor_fill_kernel");

    // 2. the rows' values (anchor form) and the bias gradients (every form: the bias gradient IS the sum over the marked anchors)
    PackGroup pg{};
    pg.count = n_levels; pg.B = batch; pg.sb = scores_batch_stride; pg.lb = locs_batch_stride;
    int pack_blocks = 0, max_npad = 0;
    long long worst_chunks = 0;
    DbiasGroup dbg{};
    dbg.count = n_levels; dbg.acounts = w.acounts;
    for (int i = 0; i < n_levels; ++i) {
        const ssdk_head_level& lv = levels[i];
        PackLevel& L = pg.lv[i];
        L.LQ = loc_cols_of(lv);
        L.ds = dscores + lv.scores_offset; L.dl = L.LQ ? dlocs + lv.locs_offset : nullptr;
        L.n0 = lv.n_score; L.n1 = lv.n_loc; L.Npad = npad_of(lv); L.HW = lv.h * lv.w;
        L.out = w.dyp[i];
        L.nb = ag.lv[i].nb; L.C = lv.n_score / L.nb; L.Jpad = jpad_of(lv); L.cap = batch * L.HW;
        L.ga = w.ga[i]; L.apix = w.apix[i]; L.acount = w.acounts + i * kAT;
        L.mode = w.mode + i;
        L.dbp = w.dbp[i]; L.chunks_cap = cdiv(L.cap, kGatherRows);
        L.block_begin = pack_blocks;
        pack_blocks += cdiv(batch * L.HW, kPackRows);
        max_npad = std::max(max_npad, L.Npad);
        worst_chunks += (long long)L.nb * L.chunks_cap;
        DbiasLevel& D = dbg.lv[i];
        D.part = w.dbp[i]; D.db0 = lv.db_score; D.db1 = L.LQ ? lv.db_loc : nullptr; D.nb = L.nb; D.C = L.C; D.Jpad = L.Jpad; D.chunks_cap = L.chunks_cap;
        D.LQ = L.LQ;
    }
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)std::min<long long>(worst_chunks, 1024)), dim3(256), 0, s, pg, w.gtab);
    SSDK_CHECK_LAUNCH("gather_rows_kernel");
    hipLaunchKernelGGL(anchor_dbias_kernel, dim3(n_levels * kAT), dim3(128), 0, s, dbg);
    SSDK_CHECK_LAUNCH("anchor_dbias_kernel");
    {   // the dense rows [pixel][Npad] of the levels that took the dense form (a small grid: usually none does)
        const int grid = std::min(pack_blocks, 256);
        if (max_npad <= 384) hipLaunchKernelGGL(pack_store_kernel<6>, dim3(grid), dim3(256), 0, s, pg, pack_blocks);
        else if (max_npad <= 512) hipLaunchKernelGGL(pack_store_kernel<8>, dim3(grid), dim3(256), 0, s, pg, pack_blocks);
        else hipLaunchKernelGGL(pack_store_kernel<12>, dim3(grid), dim3(256), 0, s, pg, pack_blocks);
        SSDK_CHECK_LAUNCH("pack_store_kernel");
    }

    // 3. data gradients.  Anchor form: row GEMM into T, then the sum pass; dense form: the forward kernel with mirrored taps
    {
        RowGemmGroup rg{};
        DxGroup dg{};
        rg.plan = w.plan; rg.acounts = w.acounts; rg.zeros = w.zeros;
        { const char* e = getenv("SSDK_RG_PROBE"); rg.probe = e ? atoi(e) : 0; }
        rg.count = n_levels;
        int dx_blocks = 0, kq = 1;
        bool any_dx = false;
        for (int i = 0; i < n_levels; ++i) {
            const ssdk_head_level& lv = levels[i];
            RowGemmLevel& R = rg.lv[i];
            R.ga = w.ga[i]; R.ws = lv.w_score; R.wl = lv.w_loc; R.T = w.T[i]; R.LQ = loc_cols_of(lv);
            R.nb = ag.lv[i].nb; R.C = lv.n_score / R.nb; R.cap = batch * lv.h * lv.w; R.K9 = 9 * lv.cin; R.n_chunks = cdiv(9 * lv.cin, kRgChunkCols);
            kq = std::max(kq, jpad_of(lv) / 32);
            if (!lv.dx) continue;
            any_dx = true;
            DxLevel& D = dg.lv[dg.count++];
            D.aidx = w.aidx[i]; D.T = w.T[i]; D.dx = lv.dx; D.mode = w.mode + i; D.nb = R.nb; D.H = lv.h; D.W = lv.w; D.cin = lv.cin; D.cap = R.cap;
            D.blk_begin = dx_blocks;
            dx_blocks += cdiv(R.cap, 4);
        }
        for (int i = 0; i < n_levels; ++i)
            SSDK_REQUIRE(jpad_of(levels[i]) / 32 == kq, SSDK_E_UNSUPPORTED, "ssdk_heads_bwd: the levels' row widths differ (columns per anchor rounded up to 32: %d vs %d)",
                         jpad_of(levels[i]), kq * 32);
        if (any_dx) {
            // what is resident at once: 3 / 2 workgroups per CU (measured at 512 / 768 / 1 024 workgroups: 127 / 120 / 126 us)
            const int grid = kq <= 3 ? 768 : 512;
            if (kq == 1) hipLaunchKernelGGL(anchor_rowgemm_kernel<1>, dim3(grid), dim3(256), 0, s, rg);
            else if (kq == 2) hipLaunchKernelGGL(anchor_rowgemm_kernel<2>, dim3(grid), dim3(256), 0, s, rg);
            else if (kq == 3) hipLaunchKernelGGL(anchor_rowgemm_kernel<3>, dim3(grid), dim3(256), 0, s, rg);
            else hipLaunchKernelGGL(anchor_rowgemm_kernel<4>, dim3(grid), dim3(256), 0, s, rg);
            SSDK_CHECK_LAUNCH("anchor_rowgemm_kernel");
            hipLaunchKernelGGL(anchor_dx_kernel, dim3(dx_blocks), dim3(256), 0, s, dg);
            SSDK_CHECK_LAUNCH("anchor_dx_kernel");
        }
    }
    {   // dense form (mode 0)
        TransposeGroup tg{};
        int t_blocks = 0;
        ConvProblem rest[kMaxProblems], fdg[kMaxProblems];
        FastProblem ffp[kMaxProblems];
        SplitGroup fsg{};
        int n_rest = 0, n_fdg = 0, fsplit_blocks = 0;
        for (int i = 0; i < n_levels; ++i) {
            const ssdk_head_level& lv = levels[i];
            if (!lv.dx) continue;
            const int npad = npad_of(lv), hw = lv.h * lv.w;
            TransposeJob& J = tg.j[tg.count++];
            J.w0 = lv.w_score; J.w1 = lv.w_loc; J.out = w.wd[i];
            J.kind = 0; J.n0 = lv.n_score; J.n1 = lv.n_loc; J.Npad = npad; J.taps = 9; J.Cc = lv.cin; J.mode = w.mode + i;
            J.tiles_x = cdiv(npad, 32); J.tiles_y = cdiv(lv.cin, 32); J.block_begin = t_blocks;
            t_blocks += J.tiles_x * J.tiles_y * kTrDepth;
            ConvProblem g{};
            g.a = w.dyp[i]; g.a_bstride = (long long)hw * npad; g.a_pstride = npad; g.Cc = npad;
            g.B = batch; g.Hout = lv.h; g.Wout = lv.w; g.Hin = lv.h; g.Win = lv.w; g.ksize = 3; g.stride = 1; g.pad = 1;
            g.w0 = w.wd[i]; g.n0 = lv.cin; g.n1 = 0;
            g.o0 = lv.dx; g.ob0 = (long long)hw * lv.cin; g.os0 = lv.cin;
            g.mode = w.mode + i; g.want_mode = 0;
            finish_problem(g);
            if (fast && fast_conv_ok(w.dyp[i], batch, lv.h, lv.w, npad, 3, 1, lv.cin)) {
                const int rows = cdiv(lv.cin, 32) * 32, K = 9 * npad;
                ffp[n_fdg].w_hi = w.fast_hi[i]; ffp[n_fdg].w_mid = w.fast_mid[i]; ffp[n_fdg].w_bytes = (unsigned)((size_t)rows * K * 2);
                SplitJob& SJ = fsg.j[fsg.count++];
                SJ.w0 = w.wd[i]; SJ.w1 = nullptr; SJ.n0 = lv.cin; SJ.n1 = 0; SJ.n_rows = rows; SJ.K = K; SJ.hi = w.fast_hi[i]; SJ.mid = w.fast_mid[i];
                SJ.flip_taps = 9; SJ.tap_len = npad;
                SJ.block_begin = fsplit_blocks;
                fsplit_blocks += (int)(((long long)rows * K / 4 + 255) / 256);
                fdg[n_fdg++] = g;
            } else {
                rest[n_rest++] = g;
            }
        }
        if (tg.count) {
            hipLaunchKernelGGL(transpose_group_kernel, dim3(t_blocks), dim3(256), 0, s, tg);
            SSDK_CHECK_LAUNCH("transpose_group_kernel");
        }
        int rc = SSDK_OK;
        if (n_rest) rc = launch_group(rest, n_rest, true, s);
        if (!rc && n_fdg) rc = launch_fast_group(fdg, ffp, n_fdg, fsg, fsplit_blocks, s);
        if (rc) return rc;
    }

    // 4. weight gradients: dense (mode 0) and anchor rows (mode 2), every K split into its own copy; then the copies in split order
    WgradGroup wd_{}, wa_{};
    int n_wgrad = 0;
    int idx_of[kMaxProblems];
    for (int i = 0; i < n_levels; ++i) {
        const ssdk_head_level& lv = levels[i];
        if (!lv.dw_score) continue;
        SSDK_REQUIRE(lv.dw_loc || lv.n_loc == 0, SSDK_E_INVALID, "ssdk_heads_bwd: dw_loc missing");
        const long long K9 = (long long)9 * lv.cin, N = lv.n_score + lv.n_loc;
        const int LQ = loc_cols_of(lv);
        WgradProblem g = heads_wgrad_problem(lv, batch);
        g.dy = w.dyp[i];
        g.mode = w.mode + i; g.want_mode = 0;
        g.dw0 = w.dw_part[i];
        g.dw1 = w.dw_part[i] + (size_t)lv.n_score * K9;
        g.det_stride = N * K9;
        g.ordered = 1;
        wd_.p[n_wgrad] = g;
        const int nb = ag.lv[i].nb, C = lv.n_score / nb, ks = w.anchor_splits[i];
        WgradProblem a = heads_wgrad_problem(lv, batch);
        a.dy = w.ga[i]; a.Npad = jpad_of(lv); a.n0 = C; a.n1 = LQ; a.n_tiles = 1;
        a.mode = w.mode + i; a.want_mode = 2;
        a.row_list = w.apix[i]; a.row_count = nullptr;
        a.seg_count = w.acounts + i * kAT; a.segs = nb; a.seg_cap = batch * lv.h * lv.w;
        a.dw0_seg = (long long)C * K9; a.dw1_seg = (long long)LQ * K9;
        a.ordered = 1;
        if (ks > 1) {
            a.dw0 = w.dw_part[i]; a.dw1 = LQ ? w.dw_part[i] + (size_t)lv.n_score * K9 : nullptr; a.det_stride = N * K9;
        } else {
            a.dw0 = lv.dw_score; a.dw1 = LQ ? lv.dw_loc : nullptr; a.det_stride = 0;
        }
        wa_.p[n_wgrad] = a;
        idx_of[n_wgrad] = i;
        ++n_wgrad;
    }
    if (n_wgrad) {
        size_wgrad_splits(wd_, n_wgrad, 1);
        { int rc = launch_wgrad(wd_, s, fast); if (rc) return rc; }
        // anchor rows: 128-channel workgroups, the host-side split rule of anchor_wgrad_splits (a fixed function of the shapes)
        int begin = 0;
        for (int q = 0; q < n_wgrad; ++q) {
            WgradProblem& a = wa_.p[q];
            a.k_splits = w.anchor_splits[idx_of[q]];
            a.block_begin = begin;
            begin += a.ksize * a.ksize * a.n_tiles * a.c_blocks * a.segs * a.k_splits;
        }
        wa_.count = n_wgrad;
        wa_.total_blocks = begin;
        { int rc = launch_wgrad(wa_, s, fast, true); if (rc) return rc; }
        ReduceList rl;
        for (int q = 0; q < n_wgrad; ++q) {
            const int i = idx_of[q];
            const ssdk_head_level& lv = levels[i];
            const long long K9 = (long long)9 * lv.cin, N = lv.n_score + lv.n_loc;
            const WgradProblem& g = wd_.p[q];
            SSDK_REQUIRE(g.k_splits == w.dense_splits[i], SSDK_E_WORKSPACE, "ssdk_heads_bwd: the workspace was sized for another split rule");
            const int used = wgrad_used_splits(g);
            int rc = rl.add(lv.dw_score, w.dw_part[i], (long long)lv.n_score * K9, N * K9, used, 0, s, w.mode + i, 0);
            if (!rc && lv.n_loc) rc = rl.add(lv.dw_loc, w.dw_part[i] + (size_t)lv.n_score * K9, (long long)lv.n_loc * K9, N * K9, used, 0, s, w.mode + i, 0);
            const int ks = w.anchor_splits[i];
            if (!rc && ks > 1) rc = rl.add(lv.dw_score, w.dw_part[i], (long long)lv.n_score * K9, N * K9, ks, 0, s, w.mode + i, 2);
            if (!rc && ks > 1 && lv.n_loc) rc = rl.add(lv.dw_loc, w.dw_part[i] + (size_t)lv.n_score * K9, (long long)lv.n_loc * K9, N * K9, ks, 0, s, w.mode + i, 2);
            if (rc) return rc;
        }
        const int rc = rl.launch(s);
        if (rc) return rc;
    }
    return SSDK_OK;
}

static size_t heads_bwd_ws_bytes(const ssdk_head_level* levels, int n_levels, int batch, bool fast) {
    size_t total = 0;
    if (!levels || n_levels <= 0 || n_levels > kMaxProblems || batch <= 0) return 0;
    if (ordered_heads_ok(levels, n_levels, batch)) carve_heads_ordered(nullptr, levels, n_levels, batch, &total, fast);
    else carve_heads_bwd(nullptr, levels, n_levels, batch, &total, fast);
    return total;
}
extern "C" size_t ssdk_heads_bwd_workspace_bytes(const ssdk_head_level* levels, int n_levels, int batch) { return heads_bwd_ws_bytes(levels, n_levels, batch, false); }
extern "C" size_t ssdk_heads_bwd_fast_workspace_bytes(const ssdk_head_level* levels, int n_levels, int batch) { return heads_bwd_ws_bytes(levels, n_levels, batch, true); }

static int heads_bwd_impl(const ssdk_head_level* levels, int n_levels, int batch, const float* dscores, long long scores_batch_stride,
                          const float* dlocs, long long locs_batch_stride, void* workspace, size_t workspace_bytes, void* stream, bool fast,
                          const unsigned char* row_mask = nullptr, int num_anchors = 0);
extern "C" int ssdk_heads_bwd_ex(const ssdk_head_level* levels, int n_levels, int batch, const float* dscores, long long scores_batch_stride,
                                 const float* dlocs, long long locs_batch_stride, const unsigned char* row_mask, int num_anchors, int fast_terms,
                                 void* workspace, size_t workspace_bytes, void* stream) {
    SSDK_REQUIRE(fast_terms == 0 || fast_terms == 3, SSDK_E_UNSUPPORTED, "ssdk_heads_bwd_ex: fast_terms=%d (0 = fp32, 3 = split-bf16)", fast_terms);
    SSDK_REQUIRE(!row_mask || num_anchors > 0, SSDK_E_INVALID, "ssdk_heads_bwd_ex: a row mask needs num_anchors");
    return heads_bwd_impl(levels, n_levels, batch, dscores, scores_batch_stride, dlocs, locs_batch_stride, workspace, workspace_bytes, stream,
                          fast_terms == 3, row_mask, num_anchors);
}
extern "C" int ssdk_heads_bwd(const ssdk_head_level* levels, int n_levels, int batch, const float* dscores,
                              long long scores_batch_stride, const float* dlocs, long long locs_batch_stride,
                              void* workspace, size_t workspace_bytes, void* stream) {
    return heads_bwd_impl(levels, n_levels, batch, dscores, scores_batch_stride, dlocs, locs_batch_stride, workspace, workspace_bytes, stream, false);
}
extern "C" int ssdk_heads_bwd_fast(const ssdk_head_level* levels, int n_levels, int batch, const float* dscores, long long scores_batch_stride,
                                   const float* dlocs, long long locs_batch_stride, int terms, void* workspace, size_t workspace_bytes, void* stream) {
    SSDK_REQUIRE(terms == 3, SSDK_E_UNSUPPORTED, "ssdk_heads_bwd_fast: terms=%d (3 is the one form built)", terms);
    return heads_bwd_impl(levels, n_levels, batch, dscores, scores_batch_stride, dlocs, locs_batch_stride, workspace, workspace_bytes, stream, true);
}

static int heads_bwd_impl(const ssdk_head_level* levels, int n_levels, int batch, const float* dscores, long long scores_batch_stride,
                          const float* dlocs, long long locs_batch_stride, void* workspace, size_t workspace_bytes, void* stream, bool fast,
                          const unsigned char* row_mask, int num_anchors) {
    SSDK_REQUIRE(levels && n_levels > 0 && n_levels <= kMaxProblems, SSDK_E_INVALID, "ssdk_heads_bwd: n_levels=%d (1..%d)", n_levels, kMaxProblems);
    SSDK_REQUIRE(dscores, SSDK_E_INVALID, "ssdk_heads_bwd: null dscores");
    SSDK_REQUIRE(batch > 0, SSDK_E_INVALID, "ssdk_heads_bwd: batch=%d", batch);
    hipStream_t s = (hipStream_t)stream;
    {
        const size_t need = heads_bwd_ws_bytes(levels, n_levels, batch, fast);
        SSDK_REQUIRE(workspace && workspace_bytes >= need, SSDK_E_WORKSPACE, "ssdk_heads_bwd: workspace too small");
    }
    if (ordered_heads_ok(levels, n_levels, batch))
        return heads_bwd_ordered(levels, n_levels, batch, dscores, scores_batch_stride, dlocs, locs_batch_stride, workspace, s, fast, row_mask, num_anchors);

    // ---- the legacy pipeline: levels without anchor structure (score and loc towers as two single-head calls, detector.py:50-66 behind a
    // SharedConvPredictor), class counts above 124, channel counts that are not multiples of 32, operands of 2 GiB and more, or
    // SSDK_HEADS_BWD_MODE=1.  Two forms per level, picked on the device from the number of pixel rows with a gradient: 0 dense, 1 rows =
    // pixels with a gradient (scatter-added data gradient: fp32 atomics).  Deterministic mode takes the dense form on every level.
    HeadsBwdWs w = carve_heads_bwd(workspace, levels, n_levels, batch, nullptr, fast);
    LevelTotals h_totals{};
    for (int i = 0; i < n_levels; ++i) {
        int rc = check_level("ssdk_heads_bwd", batch, levels[i]);
        if (rc) return rc;
        SSDK_REQUIRE(levels[i].n_loc == 0 || dlocs, SSDK_E_INVALID, "ssdk_heads_bwd: null dlocs");
        SSDK_REQUIRE(levels[i].cin % 4 == 0 && ((uintptr_t)levels[i].x & 15) == 0, SSDK_E_UNSUPPORTED, "ssdk_heads_bwd: Cin %% 4 != 0 or x not 16-byte aligned");
        SSDK_REQUIRE(npad_of(levels[i]) <= kPackColIters * 64, SSDK_E_UNSUPPORTED, "ssdk_heads_bwd: n_score + n_loc = %d exceeds %d",
                     levels[i].n_score + levels[i].n_loc, kPackColIters * 64);
        h_totals.v[i] = batch * levels[i].h * levels[i].w;
    }
    {   // test / experiment hook: SSDK_HEADS_BWD_MODE = 0 dense, 1 pixel-sparse, unset: by density (2 = anchor rows exists in the ordered pipeline only: by density here)
        const char* f = getenv("SSDK_HEADS_BWD_MODE");
        h_totals.force = f ? atoi(f) : -1;
        if (h_totals.force > 1) h_totals.force = -1;
    }
    // Deterministic mode: the pixel-sparse form scatter-adds into dX with fp32 atomics and lists its rows in the order the pack's workgroups
    // finish -- the dense (output-stationary) data gradient and a dense weight gradient whose K splits are added in split order take
    // its place; the bias gradients are column sums of the packed rows in two fixed-order stages
    const bool det = deterministic();
    if (det) h_totals.force = 0;
    // 0. zero everything the atomics of this call add into, in one launch (two when > 32 buffers); the row counters ride along
    {
        ZeroList zl;
        zl.add(reinterpret_cast<float*>(w.counts), kMaxProblems);
        for (int i = 0; i < n_levels; ++i) {
            const ssdk_head_level& lv = levels[i];
            const size_t hw = (size_t)lv.h * lv.w;
            zl.add(lv.db_score, (size_t)lv.n_score);
            if (lv.n_loc) zl.add(lv.db_loc, (size_t)lv.n_loc);
            zl.add(lv.dx, (size_t)batch * hw * lv.cin);
            zl.add(lv.dw_score, (size_t)lv.n_score * 9 * lv.cin);
            if (lv.n_loc) zl.add(lv.dw_loc, (size_t)lv.n_loc * 9 * lv.cin);
            if (zl.a.count > kMaxZero - 5) { int rc = zl.launch(s); if (rc) return rc; }
        }
        int rc = zl.launch(s);
        if (rc) return rc;
    }
    // 1. pack dY (aligned, zero padded rows), bias gradients, list of rows that carry a gradient: all levels, one launch
    {
        PackGroup pg{};
        pg.count = n_levels; pg.B = batch; pg.sb = scores_batch_stride; pg.lb = locs_batch_stride;
        int begin = 0;
        for (int i = 0; i < n_levels; ++i) {
            const ssdk_head_level& lv = levels[i];
            PackLevel& L = pg.lv[i];
            L.ds = dscores + lv.scores_offset; L.dl = lv.n_loc ? dlocs + lv.locs_offset : nullptr;
            L.n0 = lv.n_score; L.n1 = lv.n_loc; L.Npad = npad_of(lv); L.HW = lv.h * lv.w;
            L.out = w.dyp[i]; L.db0 = det ? nullptr : lv.db_score; L.db1 = det ? nullptr : lv.db_loc; L.row_list = w.row_list[i]; L.row_count = w.counts + i;
            if (row_mask && lv.n_loc > 0 && lv.n_loc % 4 == 0 && lv.n_score % (lv.n_loc / 4) == 0) {
                // the level's anchors in the caller's [batch][num_anchors] numbering: anchor-major class-minor rows (detector.py:52-63);
                // a pixel whose anchors are all unmarked is not read
                const int rnb = lv.n_loc / 4, C = lv.n_score / rnb;
                if (rnb <= kWave && lv.scores_offset % C == 0 && lv.locs_offset == 4 * (lv.scores_offset / C) &&
                    lv.scores_offset / C + (long long)L.HW * rnb <= num_anchors) {
                    L.rmask = row_mask; L.a_total = num_anchors; L.a_off = (int)(lv.scores_offset / C); L.rnb = rnb;
                }
            }
            L.block_begin = begin;
            begin += cdiv(batch * L.HW, kPackRows);
        }
        // one instantiation per launch: column trips of the widest level (6 / 8 / 12: Npad <= 384 / 512 / 768)
        int max_npad = 0;
        for (int i = 0; i < n_levels; ++i) max_npad = std::max(max_npad, pg.lv[i].Npad);
        if (max_npad <= 384) hipLaunchKernelGGL(pack_dy_kernel<6>, dim3(begin), dim3(256), 0, s, pg);
        else if (max_npad <= 512) hipLaunchKernelGGL(pack_dy_kernel<8>, dim3(begin), dim3(256), 0, s, pg);
        else hipLaunchKernelGGL(pack_dy_kernel<12>, dim3(begin), dim3(256), 0, s, pg);
        SSDK_CHECK_LAUNCH("pack_dy_kernel");
        hipLaunchKernelGGL(decide_sparse_kernel, dim3(1), dim3(64), 0, s, w.counts, h_totals, n_levels, w.mode);
        SSDK_CHECK_LAUNCH("decide_sparse_kernel");
    }

    // 2. backward-data.  dense: dX[m][c] = sum_(tap,n) dY[m + pad - tap][n] * W[n][tap][c] (output stationary);
    //    sparse: T[row][tap*Cin + c] = dY[row][:] . W[:, tap, c] for the non-zero rows only, scatter-added into dX.
    ConvProblem dense[kMaxProblems], sparse[kMaxProblems];
    int n_dgrad = 0;
    TransposeGroup tg{};
    int t_blocks = 0;
    for (int i = 0; i < n_levels; ++i) {
        const ssdk_head_level& lv = levels[i];
        if (!lv.dx) continue;
        const int npad = npad_of(lv);
        for (int kind = 0; kind < (det ? 1 : 2); ++kind) {
            TransposeJob& J = tg.j[tg.count++];
            J.w0 = lv.w_score; J.w1 = lv.w_loc; J.out = kind == 0 ? w.wd[i] : w.wt[i];
            J.kind = kind; J.n0 = lv.n_score; J.n1 = lv.n_loc; J.Npad = npad; J.taps = 9; J.Cc = lv.cin; J.mode = w.mode + i;
            J.tiles_x = cdiv(npad, 32); J.tiles_y = cdiv(lv.cin, 32); J.block_begin = t_blocks;
            t_blocks += J.tiles_x * J.tiles_y * kTrDepth;
        }
    }
    for (int i = 0; i < n_levels; ++i) {
        const ssdk_head_level& lv = levels[i];
        if (!lv.dx) continue;
        const int npad = npad_of(lv), hw = lv.h * lv.w;
        ConvProblem g{};
        g.a = w.dyp[i]; g.a_bstride = (long long)hw * npad; g.a_pstride = npad; g.Cc = npad;
        g.B = batch; g.Hout = lv.h; g.Wout = lv.w; g.Hin = lv.h; g.Win = lv.w; g.ksize = 3; g.stride = 1; g.pad = 1;
        g.w0 = w.wd[i]; g.n0 = lv.cin; g.n1 = 0;
        g.o0 = lv.dx; g.ob0 = (long long)hw * lv.cin; g.os0 = lv.cin;
        g.mode = w.mode + i; g.want_mode = 0;
        finish_problem(g);
        dense[n_dgrad] = g;
        ConvProblem q = g;
        q.w0 = w.wt[i]; q.n0 = 9 * lv.cin; q.sc_cin = lv.cin;
        q.row_list = w.row_list[i]; q.row_count = w.counts + i; q.want_mode = 1;
        finish_problem(q);  // m_tiles for the worst case; workgroups past the real row count exit at once
        sparse[n_dgrad] = q;
        ++n_dgrad;
    }
    if (tg.count) {
        hipLaunchKernelGGL(transpose_group_kernel, dim3(t_blocks), dim3(256), 0, s, tg);
        SSDK_CHECK_LAUNCH("transpose_group_kernel");
    }
    if (n_dgrad) {
        // fast mode: the DENSE data gradient (what a focal-loss step takes on every level) as the forward convolution of the packed
        // rows with the mirrored kernel on the split-bf16 GEMM; levels it cannot take stay on the fp32 kernel
        ConvProblem fdg[kMaxProblems], rest[kMaxProblems];
        FastProblem ffp[kMaxProblems];
        SplitGroup fsg{};
        int n_fdg = 0, n_rest = 0, fsplit_blocks = 0, di = 0;
        for (int i = 0; i < n_levels; ++i) {
            const ssdk_head_level& lv = levels[i];
            if (!lv.dx) continue;
            ConvProblem g = dense[di++];
            const int npad = npad_of(lv);
            if (fast && fast_conv_ok(w.dyp[i], batch, lv.h, lv.w, npad, 3, 1, lv.cin)) {
                const int rows = cdiv(lv.cin, 32) * 32, K = 9 * npad;
                ffp[n_fdg].w_hi = w.fast_hi[i]; ffp[n_fdg].w_mid = w.fast_mid[i]; ffp[n_fdg].w_bytes = (unsigned)((size_t)rows * K * 2);
                SplitJob& J = fsg.j[fsg.count++];
                J.w0 = w.wd[i]; J.w1 = nullptr; J.n0 = lv.cin; J.n1 = 0; J.n_rows = rows; J.K = K; J.hi = w.fast_hi[i]; J.mid = w.fast_mid[i];
                J.flip_taps = 9; J.tap_len = npad;
                J.block_begin = fsplit_blocks;
                fsplit_blocks += (int)(((long long)rows * K / 4 + 255) / 256);
                fdg[n_fdg++] = g;   // (pad 1 = 3 - 1 - 1: the mirrored 3 x 3 / pad 1 convolution pads like the forward one)
            } else {
                rest[n_rest++] = g;
            }
        }
        int rc = SSDK_OK;
        if (n_rest) rc = launch_group(rest, n_rest, true, s);
        if (!rc && n_fdg) rc = launch_fast_group(fdg, ffp, n_fdg, fsg, fsplit_blocks, s);
        if (rc) return rc;
        if (!det) rc = launch_group(sparse, n_dgrad, false, s, false, true, w.vtab);
        if (rc) return rc;
    }

    // 3. backward-weights, dense (all pixels) or sparse (only the listed rows)
    WgradGroup wd_{}, ws_{};
    int n_wgrad = 0;
    for (int i = 0; i < n_levels; ++i) {
        const ssdk_head_level& lv = levels[i];
        if (!lv.dw_score) continue;
        SSDK_REQUIRE(lv.n_loc == 0 || lv.dw_loc, SSDK_E_INVALID, "ssdk_heads_bwd: dw_loc missing");
        WgradProblem g = heads_wgrad_problem(lv, batch);
        g.dy = w.dyp[i];
        g.mode = w.mode + i; g.want_mode = 0;
        if (det) {   // every K split stores its own copy; reduce_partials_kernel adds them in split order (below)
            g.dw0 = w.dw_part[i];
            g.dw1 = lv.n_loc ? w.dw_part[i] + (size_t)lv.n_score * 9 * lv.cin : nullptr;
            g.det_stride = (long long)(lv.n_score + lv.n_loc) * 9 * lv.cin;
        }
        wd_.p[n_wgrad] = g;
        g.want_mode = 1; g.row_list = w.row_list[i]; g.row_count = w.counts + i;
        ws_.p[n_wgrad] = g;
        ++n_wgrad;
    }
    if (n_wgrad) {
        size_wgrad_splits(wd_, n_wgrad, 1);
        { int rc = launch_wgrad(wd_, s, fast); if (rc) return rc; }
        if (!det) {
            size_wgrad_splits(ws_, n_wgrad, 4);  // sparse mode means < 1/4 of the rows
            int rc = launch_wgrad(ws_, s, fast); if (rc) return rc;
        }
    }
    if (det) {
        // fixed-order second halves: weight gradients = sum over the K-split copies, bias gradients = column sums of the packed rows
        ReduceList rl;
        int wi = 0;
        for (int i = 0; i < n_levels; ++i) {
            const ssdk_head_level& lv = levels[i];
            const long long K9 = (long long)9 * lv.cin, N = lv.n_score + lv.n_loc;
            if (lv.dw_score) {
                const WgradProblem& g = wd_.p[wi++];
                SSDK_REQUIRE(g.k_splits == w.dw_splits[i], SSDK_E_WORKSPACE, "ssdk_heads_bwd: the workspace was sized under another deterministic-mode setting");
                const int used = wgrad_used_splits(g);
                int rc = rl.add(lv.dw_score, w.dw_part[i], (long long)lv.n_score * K9, N * K9, used, 0, s);
                if (!rc && lv.n_loc) rc = rl.add(lv.dw_loc, w.dw_part[i] + (size_t)lv.n_score * K9, (long long)lv.n_loc * K9, N * K9, used, 0, s);
                if (rc) return rc;
            }
            if (lv.db_score || lv.db_loc) {
                const long long M = (long long)batch * lv.h * lv.w;
                const int rows_per_block = (int)std::max<long long>(64, (M + kColsumBlocks - 1) / kColsumBlocks);
                const int blocks = (int)((M + rows_per_block - 1) / rows_per_block);
                hipLaunchKernelGGL(colsum_partial_kernel, dim3(blocks), dim3(256), 0, s, w.dyp[i], M, (int)N, npad_of(lv), w.db_part[i], rows_per_block);
                SSDK_CHECK_LAUNCH("colsum_partial_kernel");
                int rc = rl.add(lv.db_score, w.db_part[i], lv.n_score, N, blocks, 0, s);
                if (!rc && lv.n_loc) rc = rl.add(lv.db_loc, w.db_part[i] + lv.n_score, lv.n_loc, N, blocks, 0, s);
                if (rc) return rc;
            }
        }
        const int rc = rl.launch(s);
        if (rc) return rc;
    }
    return SSDK_OK;
}

// ---- generic NHWC convolution (extras H2, Retina tower H3) on the same kernels --------------------------------------

static int check_conv(const char* fn, int batch, const ssdk_conv_desc& d) {
    SSDK_REQUIRE(batch > 0 && d.hin > 0 && d.win > 0 && d.cin > 0 && d.cout > 0, SSDK_E_INVALID, "%s: bad shape", fn);
    SSDK_REQUIRE((d.ksize == 1 || d.ksize == 3) && d.stride >= 1 && d.stride <= 2 && d.pad >= 0 && d.pad < d.ksize, SSDK_E_UNSUPPORTED,
                 "%s: ksize=%d stride=%d pad=%d (1x1 / 3x3, stride 1..2)", fn, d.ksize, d.stride, d.pad);
    SSDK_REQUIRE(d.x && d.w, SSDK_E_INVALID, "%s: null pointer", fn);
    SSDK_REQUIRE((d.hin + 2 * d.pad - d.ksize) / d.stride + 1 > 0 && (d.win + 2 * d.pad - d.ksize) / d.stride + 1 > 0, SSDK_E_INVALID, "%s: empty output", fn);
    return SSDK_OK;
}
static inline int out_dim(int in, int k, int s, int p) { return (in + 2 * p - k) / s + 1; }

extern "C" int ssdk_conv2d_fwd_ws(const ssdk_conv_desc* descs, int n, int batch, void* workspace, size_t workspace_bytes, void* stream) {
    SSDK_REQUIRE(descs && n > 0 && n <= kMaxProblems, SSDK_E_INVALID, "ssdk_conv2d_fwd: n=%d (1..%d)", n, kMaxProblems);
    SSDK_REQUIRE(!g_sk_host_err || *static_cast<volatile unsigned*>(g_sk_host_err) == 0u, SSDK_E_STREAMK_TIMEOUT,
                 "ssdk_conv2d_fwd: an earlier stream-K launch of this process gave up waiting for a parked partial tile -- see ssdk_heads_fwd_timeouts");
    ConvProblem probs[kMaxProblems];
    ZeroList zl;
    for (int i = 0; i < n; ++i) {
        const ssdk_conv_desc& d = descs[i];
        int rc = check_conv("ssdk_conv2d_fwd", batch, d);
        if (rc) return rc;
        SSDK_REQUIRE(d.y, SSDK_E_INVALID, "ssdk_conv2d_fwd: null output");
        const int ho = out_dim(d.hin, d.ksize, d.stride, d.pad), wo = out_dim(d.win, d.ksize, d.stride, d.pad);
        ConvProblem g{};
        g.a = d.x; g.a_bstride = (long long)d.hin * d.win * d.cin; g.a_pstride = d.cin; g.Cc = d.cin;
        g.B = batch; g.Hout = ho; g.Wout = wo; g.Hin = d.hin; g.Win = d.win; g.ksize = d.ksize; g.stride = d.stride; g.pad = d.pad;
        g.w0 = d.w; g.w1 = nullptr; g.bias0 = d.bias; g.bias1 = nullptr; g.n0 = d.cout; g.n1 = 0;
        g.o0 = d.y; g.ob0 = (long long)ho * wo * d.cout; g.os0 = d.cout; g.o1 = nullptr; g.ob1 = 0; g.os1 = 0;
        g.relu = d.relu;
        if (d.stats)
            SSDK_REQUIRE(d.cout % 4 == 0 && ((uintptr_t)d.y & 15) == 0, SSDK_E_UNSUPPORTED, "ssdk_conv2d_fwd: stats need cout %% 4 == 0 and a 16-byte aligned output");
        g.stats = d.stats;   // (known before the stream-K decision counts its units: a statistics epilogue rules the half-width last tile out; dropped again below for a split-K launch)
        finish_problem(g);
        probs[i] = g;
    }
    // One or two rounds of whole tiles on 256 CUs (the big layers of a pyramid tail): stream-K over all the launch's K slices instead of
    // splitting K with an atomic epilogue into a zeroed output -- no zero-fill launch, no atomics, BatchNorm statistics still in the epilogue
    const bool have_ws = workspace && workspace_bytes >= ssdk_heads_fwd_workspace_bytes();
    const bool streamk = have_ws && streamk_would_take(probs, n, true);
    for (int i = 0; i < n; ++i) {
        const ssdk_conv_desc& d = descs[i];
        ConvProblem& g = probs[i];
        const int ho = g.Hout, wo = g.Wout;
        bool split = !streamk && maybe_split_k(g);
        const char* f = deterministic() ? nullptr : getenv("SSDK_CONV_FORCE");
        if (f) {   // measurement knob (tools/conv_decomp_sweep.py): "<column blocks>,<K splits>"
            int nb = 0, ks = 0;
            if (!streamk && sscanf(f, "%d,%d", &nb, &ks) == 2 && nb > 0 && ks > 0) {
                g.n_blocks = std::min(nb, g.tiles_n);
                g.k_splits = (g.relu || ks < 2) ? 1 : ks;
                g.forced = 1;
                split = g.k_splits > 1;
            }
        }
        if (split) zl.add(d.y, (size_t)batch * ho * wo * d.cout);
        if (split) g.stats = nullptr;   // (in the epilogue otherwise; a split-K output is only complete after the launch: a pass of its own below)
    }
    int rc = zl.launch((hipStream_t)stream);
    if (rc) return rc;
    StreamKWs sk{};
    if (streamk) {
        Carver c(workspace);
        sk.partial = c.take<float>((size_t)(kStreamKWgs + 1) * (4 * kMaxTN * 4 * 64 * 4));
        sk.flags = c.take<unsigned>((size_t)kStreamKWgs + 2);
        sk.nwg = kStreamKWgs;
    }
    rc = launch_group(probs, n, false, (hipStream_t)stream, true, false, nullptr, streamk ? &sk : nullptr);
    if (rc) return rc;
    for (int i = 0; i < n; ++i) {
        const ssdk_conv_desc& d = descs[i];
        if (!d.stats || probs[i].stats) continue;
        const long long rows = (long long)batch * out_dim(d.hin, d.ksize, d.stride, d.pad) * out_dim(d.win, d.ksize, d.stride, d.pad);
        rc = ssdk_batchnorm_stats_accumulate(d.y, rows, d.cout, d.stats, stream);
        if (rc) return rc;
    }
    return SSDK_OK;
}

extern "C" int ssdk_conv2d_fwd(const ssdk_conv_desc* descs, int n, int batch, void* stream) {
    return ssdk_conv2d_fwd_ws(descs, n, batch, nullptr, 0, stream);
}

// The weights of n convolutions in the layout their backward-data GEMM reads (stride 1: [cin][tap][cout], mirrored-tap dgrad; strided:
// [tap][cin][cout], scatter dgrad), cin * ksize^2 * cout floats each, in ONE launch.  A training step calls it once for all the layers of
// a chain (the weights do not change between the forward and the backward pass) and hands the results to ssdk_conv2d_bwd as
// ssdk_conv_desc::w_t -- the per-layer re-layout launches in front of every backward-data GEMM (8 per SSD-300 step, 40 per RetinaNet
// tower step) disappear.
extern "C" int ssdk_conv2d_transpose_weights(const ssdk_conv_desc* descs, int n, float* const* outs, void* stream) {
    SSDK_REQUIRE(descs && outs && n > 0 && n <= kMaxTransposeJobs, SSDK_E_INVALID, "ssdk_conv2d_transpose_weights: n=%d (1..%d)", n, kMaxTransposeJobs);
    TransposeGroup tg{};
    int blocks = 0;
    for (int i = 0; i < n; ++i) {
        const ssdk_conv_desc& d = descs[i];
        SSDK_REQUIRE(d.w && outs[i] && d.cin > 0 && d.cout > 0 && d.ksize > 0 && d.stride > 0, SSDK_E_INVALID,
                     "ssdk_conv2d_transpose_weights: descriptor %d: null pointer or bad shape", i);
        TransposeJob& J = tg.j[tg.count++];
        J.w0 = d.w; J.w1 = nullptr; J.out = outs[i];
        J.kind = d.stride == 1 ? 0 : 1; J.n0 = d.cout; J.n1 = 0; J.Npad = d.cout; J.taps = d.ksize * d.ksize; J.Cc = d.cin; J.mode = nullptr;
        J.tiles_x = cdiv(d.cout, 32); J.tiles_y = cdiv(d.cin, 32); J.block_begin = blocks;
        blocks += J.tiles_x * J.tiles_y * kTrDepth;
    }
    hipLaunchKernelGGL(transpose_group_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, tg);
    SSDK_CHECK_LAUNCH("transpose_group_kernel");
    return SSDK_OK;
}

// the dense weight-gradient problem of one generic convolution (dy / outputs are filled in by the caller)
static WgradProblem conv_wgrad_problem(const ssdk_conv_desc& d, int batch) {
    WgradProblem g{};
    g.x = d.x; g.Npad = d.cout; g.Cc = d.cin;
    g.B = batch; g.Hout = out_dim(d.hin, d.ksize, d.stride, d.pad); g.Wout = out_dim(d.win, d.ksize, d.stride, d.pad);
    g.Hin = d.hin; g.Win = d.win; g.ksize = d.ksize; g.stride = d.stride; g.pad = d.pad;
    g.n0 = d.cout; g.n1 = 0;
    g.n_tiles = cdiv(d.cout, 128);
    g.c_tiles32 = cdiv(d.cin, 32);
    g.c_blocks = cdiv(g.c_tiles32, kMaxTN);
    return g;
}
static int conv_wgrad_splits(const ssdk_conv_desc& d, int batch) {
    WgradGroup one{};
    one.p[0] = conv_wgrad_problem(d, batch);
    size_wgrad_splits(one, 1, 1);
    return one.p[0].k_splits;
}

static size_t conv2d_bwd_ws_bytes(const ssdk_conv_desc* descs, int n, int batch, bool fast) {
    size_t total = 0;
    for (int i = 0; i < n; ++i) {
        const ssdk_conv_desc& d = descs[i];
        const size_t wsz = (size_t)d.cin * d.ksize * d.ksize * (size_t)d.cout;
        total += align_up(wsz * sizeof(float), 256);
        if (fast) total += 2 * align_up((size_t)cdiv(d.cin, 32) * 32 * d.ksize * d.ksize * d.cout * sizeof(__bf16), 256);   // the two split planes of the mirrored kernel
        if (d.dx && d.stride > 1 && d.ksize > 0 && batch > 0)   // the contribution rows of the strided data gradient [B * Hout * Wout][taps * Cin]
            total += align_up((size_t)batch * out_dim(d.hin, d.ksize, d.stride, d.pad) * out_dim(d.win, d.ksize, d.stride, d.pad) * d.ksize * d.ksize * d.cin * sizeof(float), 256);
        if (deterministic() && batch > 0 && d.ksize > 0 && d.stride > 0) {   // K-split copies of dw, per-workgroup column sums of db
            if (d.dw) total += align_up((size_t)conv_wgrad_splits(d, batch) * wsz * sizeof(float), 256);
            if (d.db) total += align_up((size_t)kColsumBlocks * d.cout * sizeof(float), 256);
        }
    }
    return total;
}
extern "C" size_t ssdk_conv2d_bwd_workspace_bytes(const ssdk_conv_desc* descs, int n, int batch) { return conv2d_bwd_ws_bytes(descs, n, batch, false); }
extern "C" size_t ssdk_conv2d_bwd_fast_workspace_bytes(const ssdk_conv_desc* descs, int n, int batch) { return conv2d_bwd_ws_bytes(descs, n, batch, true); }

// dy: gradient w.r.t. the convolution output (AFTER any fused ReLU has been undone by the caller), [batch,hout,wout,cout]
// with cout % 4 == 0.  dw / db are ACCUMULATED when `accumulate` != 0 (shared weights across levels), else overwritten.
static int conv2d_bwd_impl(const ssdk_conv_desc* descs, int n, int batch, int accumulate, void* workspace, size_t workspace_bytes, void* stream,
                           bool fast, void* sk_workspace = nullptr, size_t sk_workspace_bytes = 0);
extern "C" int ssdk_conv2d_bwd(const ssdk_conv_desc* descs, int n, int batch, int accumulate, void* workspace, size_t workspace_bytes,
                               void* stream) {
    return conv2d_bwd_impl(descs, n, batch, accumulate, workspace, workspace_bytes, stream, false);
}
extern "C" int ssdk_conv2d_bwd_sk(const ssdk_conv_desc* descs, int n, int batch, int accumulate, void* workspace, size_t workspace_bytes,
                                  void* sk_workspace, size_t sk_workspace_bytes, void* stream) {
    return conv2d_bwd_impl(descs, n, batch, accumulate, workspace, workspace_bytes, stream, false, sk_workspace, sk_workspace_bytes);
}
extern "C" int ssdk_conv2d_bwd_fast(const ssdk_conv_desc* descs, int n, int batch, int accumulate, int terms, void* workspace, size_t workspace_bytes,
                                    void* stream) {
    SSDK_REQUIRE(terms == 3, SSDK_E_UNSUPPORTED, "ssdk_conv2d_bwd_fast: terms=%d (3 is the one form built)", terms);
    return conv2d_bwd_impl(descs, n, batch, accumulate, workspace, workspace_bytes, stream, true);
}

static int conv2d_bwd_impl(const ssdk_conv_desc* descs, int n, int batch, int accumulate, void* workspace, size_t workspace_bytes, void* stream,
                           bool fast, void* sk_workspace, size_t sk_workspace_bytes) {
    SSDK_REQUIRE(descs && n > 0 && n <= kMaxProblems, SSDK_E_INVALID, "ssdk_conv2d_bwd: n=%d (1..%d)", n, kMaxProblems);
    SSDK_REQUIRE(!sk_workspace || !g_sk_host_err || *static_cast<volatile unsigned*>(g_sk_host_err) == 0u, SSDK_E_STREAMK_TIMEOUT,
                 "ssdk_conv2d_bwd: an earlier stream-K launch of this process gave up waiting for a parked partial tile -- see ssdk_heads_fwd_timeouts");
    SSDK_REQUIRE(workspace && workspace_bytes >= conv2d_bwd_ws_bytes(descs, n, batch, fast), SSDK_E_WORKSPACE, "ssdk_conv2d_bwd: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    Carver carve(workspace);
    ConvProblem dgrad[kMaxProblems], scat[kMaxProblems], rowsT[kMaxProblems];
    int dgrad_desc[kMaxProblems];
    StridedDxGroup sdg{};
    int n_rowsT = 0, sd_blocks = 0;
    // strided data gradients: contribution rows + sum pass (ordered, no zero-fill) in deterministic mode; otherwise the atomic scatter
    // form, which measured 15-35 us faster per SSD-300 step (3.055 against 3.07-3.09 ms: the small maps' row GEMMs underfill the chip);
    // SSDK_CONV_STRIDED_ORDERED=1 takes the ordered form everywhere (measurement knob).  Round 4's deterministic form -- the
    // output-stationary gather GEMM with three of four (pixel, tap) pairs masked -- cost 760 us per step where this costs 190.
    static const bool strided_scatter = getenv("SSDK_CONV_STRIDED_ORDERED") == nullptr;
    WgradGroup wg{};
    ZeroList zl{};
    int n_dgrad = 0, n_scat = 0, n_wgrad = 0;
    // Deterministic mode: no K split with atomics, strided data gradients in the output-stationary (gather) form instead of the
    // scatter form, weight gradients as K-split copies added in split order, bias gradients as two-stage column sums
    const bool det = deterministic();
    float* dw_part[kMaxProblems] = {};
    float* db_part[kMaxProblems] = {};
    int wg_of[kMaxProblems];
    // fast mode (ssdk_conv2d_bwd_fast): stride-1 data gradients on the split-bf16 kernel -- the forward convolution of dy with the
    // mirrored kernel, whose two bf16 planes are split from the re-laid-out weights [cin][tap][cout] with the taps reversed
    ConvProblem fdg[kMaxProblems];
    FastProblem ffp[kMaxProblems];
    SplitGroup fsg{};
    int n_fdg = 0, fsplit_blocks = 0;
    const float* fsrc[kMaxProblems];
    for (int i = 0; i < n; ++i) {
        const ssdk_conv_desc& d = descs[i];
        int rc = check_conv("ssdk_conv2d_bwd", batch, d);
        if (rc) return rc;
        SSDK_REQUIRE(d.dy, SSDK_E_INVALID, "ssdk_conv2d_bwd: null dy");
        SSDK_REQUIRE(d.cout % 4 == 0 && d.cin % 4 == 0 && ((uintptr_t)d.dy & 15) == 0 && ((uintptr_t)d.x & 15) == 0, SSDK_E_UNSUPPORTED,
                     "ssdk_conv2d_bwd: channels must be multiples of 4 and buffers 16-byte aligned");
        const int ho = out_dim(d.hin, d.ksize, d.stride, d.pad), wo = out_dim(d.win, d.ksize, d.stride, d.pad);
        const int taps = d.ksize * d.ksize;
        float* const wd_own = carve.take<float>((size_t)d.cin * taps * d.cout);
        float* wd = wd_own;
        const bool have_wt = d.w_t != nullptr;   // re-laid out beforehand by ssdk_conv2d_transpose_weights (one launch for many layers)
        if (have_wt) {
            SSDK_REQUIRE(((uintptr_t)d.w_t & 15) == 0, SSDK_E_INVALID, "ssdk_conv2d_bwd: w_t must be 16-byte aligned");
            wd = const_cast<float*>(d.w_t);
        }
        float* const rows_t = (d.dx && d.stride > 1) ? carve.take<float>((size_t)batch * ho * wo * taps * d.cin) : nullptr;
        if (d.dx && d.stride != 1 && (det || !strided_scatter)) {
            // T = dy . W as a 1 x 1 GEMM over the output pixels (weights [tap][cin][cout]: rows tap * Cin + c, K = cout), then the sum pass
            if (!have_wt) {
                hipLaunchKernelGGL(transpose_tapmajor_kernel, dim3(cdiv(d.cout, 32), cdiv(d.cin, 32), taps), dim3(256), 0, s, d.w, (const float*)nullptr,
                                   d.cout, 0, d.cout, taps, d.cin, wd);
                SSDK_CHECK_LAUNCH("transpose_tapmajor_kernel");
            }
            ConvProblem g{};
            g.a = d.dy; g.a_bstride = (long long)ho * wo * d.cout; g.a_pstride = d.cout; g.Cc = d.cout;
            g.B = batch; g.Hout = ho; g.Wout = wo; g.Hin = ho; g.Win = wo; g.ksize = 1; g.stride = 1; g.pad = 0;
            g.w0 = wd; g.n0 = taps * d.cin; g.n1 = 0;
            g.o0 = rows_t; g.ob0 = (long long)ho * wo * taps * d.cin; g.os0 = taps * d.cin;
            finish_problem(g);   // (k_splits stays 1 -- whole K chains: a split would add into T with atomics; launch_group may narrow the column blocks)
            rowsT[n_rowsT++] = g;
            StridedDxJob& J = sdg.j[sdg.count++];
            J.T = rows_t; J.dx = d.dx; J.B = batch; J.hin = d.hin; J.win = d.win; J.cin = d.cin; J.ho = ho; J.wo = wo; J.ks = d.ksize; J.stride = d.stride; J.pad = d.pad;
            J.blk_begin = sd_blocks;
            sd_blocks += (int)(((long long)batch * d.hin * d.win * (d.cin / 4) + 255) / 256);
        } else if (d.dx && d.stride == 1) {
            // output stationary: rows are INPUT pixels, A = dy [ho*wo][cout] with mirrored taps, W = wd [cin][taps*cout]
            if (!have_wt) {
                hipLaunchKernelGGL(transpose_taps_kernel, dim3(cdiv(d.cout, 32), cdiv(d.cin, 32), taps), dim3(256), 0, s, d.w, (const float*)nullptr,
                                   d.cout, 0, d.cout, taps, d.cin, wd);
                SSDK_CHECK_LAUNCH("transpose_taps_kernel");
            }
            ConvProblem g{};
            g.a = d.dy; g.a_bstride = (long long)ho * wo * d.cout; g.a_pstride = d.cout; g.Cc = d.cout;
            g.B = batch; g.Hout = d.hin; g.Wout = d.win; g.Hin = ho; g.Win = wo; g.ksize = d.ksize; g.stride = 1; g.pad = d.pad;
            g.w0 = wd; g.n0 = d.cin; g.n1 = 0;
            g.o0 = d.dx; g.ob0 = (long long)d.hin * d.win * d.cin; g.os0 = d.cin;
            finish_problem(g);
            __bf16* const plane_hi = fast ? carve.take<__bf16>((size_t)cdiv(d.cin, 32) * 32 * taps * d.cout) : nullptr;
            __bf16* const plane_mid = fast ? carve.take<__bf16>((size_t)cdiv(d.cin, 32) * 32 * taps * d.cout) : nullptr;
            // (at least ~1 GFLOP: below that the split of the weights costs more than the bf16 MFMAs save -- the same bound ops.conv2d applies to the forward)
            if (fast && 2.0 * batch * d.hin * d.win * (double)d.cin * taps * d.cout >= fast_min_flops() &&
                fast_conv_ok(d.dy, batch, ho, wo, d.cout, d.ksize, d.ksize - 1 - d.pad, d.cin)) {
                // forward form: "input" = dy, padding ksize - 1 - pad, "weights" = the mirrored kernel [cin][tap'][cout]
                g.pad = d.ksize - 1 - d.pad;
                const int rows = cdiv(d.cin, 32) * 32, K = taps * d.cout;
                int same = -1;   // (a tower layer's weights are shared by all pyramid levels: split them once)
                for (int q = 0; q < n_fdg && same < 0; ++q)
                    if (fsrc[q] == wd) same = q;
                if (same >= 0) ffp[n_fdg] = ffp[same];
                else {
                    ffp[n_fdg].w_hi = plane_hi; ffp[n_fdg].w_mid = plane_mid; ffp[n_fdg].w_bytes = (unsigned)((size_t)rows * K * 2);
                    SplitJob& J = fsg.j[fsg.count++];
                    J.w0 = wd; J.w1 = nullptr; J.n0 = d.cin; J.n1 = 0; J.n_rows = rows; J.K = K; J.hi = plane_hi; J.mid = plane_mid;
                    J.flip_taps = taps; J.tap_len = d.cout;
                    J.block_begin = fsplit_blocks;
                    fsplit_blocks += (int)(((long long)rows * K / 4 + 255) / 256);
                }
                fsrc[n_fdg] = wd;
                fdg[n_fdg++] = g;
            } else {
                dgrad_desc[n_dgrad] = i;   // (a K split of the small maps is decided below, once the launch is known not to take stream-K)
                dgrad[n_dgrad++] = g;
            }
        } else if (d.dx) {
            // strided: input stationary.  T[out pixel][tap*cin + c] = dy[out pixel][:] . W[:, tap, c], scatter-added into
            // dx at (yo*stride - pad + ky, xo*stride - pad + kx): no multiply is spent on (pixel, tap) pairs that do not exist
            if (!have_wt) {
                hipLaunchKernelGGL(transpose_tapmajor_kernel, dim3(cdiv(d.cout, 32), cdiv(d.cin, 32), taps), dim3(256), 0, s, d.w, (const float*)nullptr,
                                   d.cout, 0, d.cout, taps, d.cin, wd);
                SSDK_CHECK_LAUNCH("transpose_tapmajor_kernel");
            }
            zl.add(d.dx, (size_t)batch * d.hin * d.win * d.cin);
            ConvProblem g{};
            g.a = d.dy; g.a_bstride = (long long)ho * wo * d.cout; g.a_pstride = d.cout; g.Cc = d.cout;
            g.B = batch; g.Hout = ho; g.Wout = wo; g.Hin = d.hin; g.Win = d.win; g.ksize = d.ksize; g.stride = d.stride; g.pad = d.pad;
            g.w0 = wd; g.n0 = taps * d.cin; g.n1 = 0; g.sc_cin = d.cin;
            g.o0 = d.dx; g.ob0 = (long long)d.hin * d.win * d.cin; g.os0 = d.cin;
            finish_problem(g);
            scat[n_scat++] = g;
        }
        wg_of[i] = -1;
        if (d.dw) {
            bool seen = false;  // descriptors that share weights share dw: zero it once
            for (int q = 0; q < i; ++q) seen = seen || descs[q].dw == d.dw;
            if (!accumulate && !seen && !det) zl.add(d.dw, (size_t)d.cout * taps * d.cin);
            WgradProblem g = conv_wgrad_problem(d, batch);
            g.dy = d.dy;
            g.dw0 = d.dw; g.dw1 = nullptr;
            if (det) {   // every K split stores its own copy (added in split order below)
                dw_part[i] = carve.take<float>((size_t)conv_wgrad_splits(d, batch) * d.cout * taps * d.cin);
                g.dw0 = dw_part[i];
                g.det_stride = (long long)d.cout * taps * d.cin;
            }
            wg_of[i] = n_wgrad;
            wg.p[n_wgrad++] = g;
        }
        if (d.db && det) db_part[i] = carve.take<float>((size_t)kColsumBlocks * d.cout);
        if (d.db && !accumulate && !det) {
            bool seen = false;
            for (int q = 0; q < i; ++q) seen = seen || descs[q].db == d.db;
            if (!seen) zl.add(d.db, (size_t)d.cout);
        }
    }
    // stride-1 data gradients: a few rounds of whole tiles whose last round is partly filled (the RetinaNet towers: 2 688 tiles on 512 slots)
    // run in stream-K form, as the forward launch of the same layers -- with the caller's stream-K state (the workspace of ssdk_heads_fwd /
    // ssdk_conv2d_fwd_ws); otherwise the small maps split K (one long chain per tile: K = taps * cout) with atomics into a zeroed dx
    const bool dgrad_sk = n_dgrad && sk_workspace && sk_workspace_bytes >= ssdk_heads_fwd_workspace_bytes() && streamk_would_take(dgrad, n_dgrad, true, true);
    for (int q = 0; q < n_dgrad && !dgrad_sk; ++q) {
        const ssdk_conv_desc& d = descs[dgrad_desc[q]];
        if (maybe_split_k(dgrad[q])) zl.add(d.dx, (size_t)batch * d.hin * d.win * d.cin);
    }
    int rc = zl.launch(s);
    if (rc) return rc;
    ReduceList rl;
    for (int i = 0; i < n && det; ++i) {   // bias gradients: per-workgroup column sums, then their sum in workgroup order
        const ssdk_conv_desc& d = descs[i];
        if (!d.db) continue;
        const int ho = out_dim(d.hin, d.ksize, d.stride, d.pad), wo = out_dim(d.win, d.ksize, d.stride, d.pad);
        const long long M = (long long)batch * ho * wo;
        const int rows_per_block = (int)std::max<long long>(64, (M + kColsumBlocks - 1) / kColsumBlocks);
        const int blocks = (int)((M + rows_per_block - 1) / rows_per_block);
        hipLaunchKernelGGL(colsum_partial_kernel, dim3(blocks), dim3(256), 0, s, d.dy, M, d.cout, d.cout, db_part[i], rows_per_block);
        SSDK_CHECK_LAUNCH("colsum_partial_kernel");
        bool seen = false;   // (descriptors that share a bias: the later ones add to what the earlier ones left -- in separate launches)
        for (int q = 0; q < i; ++q) seen = seen || descs[q].db == d.db;
        if (seen) { rc = rl.launch(s); if (rc) return rc; }
        rc = rl.add(d.db, db_part[i], d.cout, d.cout, blocks, (accumulate || seen) ? 1 : 0, s);
        if (rc) return rc;
    }
    for (int i = 0; i < n && !det; ++i) {
        const ssdk_conv_desc& d = descs[i];
        if (!d.db) continue;
        const int ho = out_dim(d.hin, d.ksize, d.stride, d.pad), wo = out_dim(d.win, d.ksize, d.stride, d.pad);
        const long long M = (long long)batch * ho * wo;
        // (64 rows per workgroup until the launch has 256 of them, then more rows: every workgroup ends with an atomic per column on the
        // same cout addresses -- the rule measured for the BatchNorm statistics, norm.hip)
        const int rows_per_block = (int)std::max<long long>(64, (M + 255) / 256);
        hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)((M + rows_per_block - 1) / rows_per_block)), dim3(256), 0, s, d.dy, M, d.cout,
                           d.db, rows_per_block);
        SSDK_CHECK_LAUNCH("colsum_kernel");
    }
    if (n_dgrad) {
        StreamKWs sk{};
        const bool use_sk = dgrad_sk;
        if (use_sk) {
            Carver c(sk_workspace);
            sk.partial = c.take<float>((size_t)(kStreamKWgs + 1) * (4 * kMaxTN * 4 * 64 * 4));
            sk.flags = c.take<unsigned>((size_t)kStreamKWgs + 2);
            sk.nwg = kStreamKWgs;
        }
        rc = launch_group(dgrad, n_dgrad, true, s, use_sk, false, nullptr, use_sk ? &sk : nullptr);
        if (rc) return rc;
    }
    if (n_scat) {
        rc = launch_group(scat, n_scat, false, s, false, true);
        if (rc) return rc;
    }
    if (n_rowsT) {
        rc = launch_group(rowsT, n_rowsT, false, s, true);
        if (rc) return rc;
        hipLaunchKernelGGL(strided_dx_kernel, dim3(sd_blocks), dim3(256), 0, s, sdg);
        SSDK_CHECK_LAUNCH("strided_dx_kernel");
    }
    if (n_fdg) {   // (after the re-layout launches above: the split reads their output)
        rc = launch_fast_group(fdg, ffp, n_fdg, fsg, fsplit_blocks, s);
        if (rc) return rc;
    }
    if (n_wgrad) {
        size_wgrad_splits(wg, n_wgrad, 1);
        { int rc2 = launch_wgrad(wg, s, fast); if (rc2) return rc2; }
    }
    if (det) {
        for (int i = 0; i < n; ++i) {
            const ssdk_conv_desc& d = descs[i];
            if (!d.dw) continue;
            const WgradProblem& g = wg.p[wg_of[i]];
            bool seen = false;   // (shared weights: one reduction per descriptor, in descriptor order, in separate launches)
            for (int q = 0; q < i; ++q) seen = seen || descs[q].dw == d.dw;
            if (seen) { rc = rl.launch(s); if (rc) return rc; }
            const long long elems = (long long)d.cout * d.ksize * d.ksize * d.cin;
            rc = rl.add(d.dw, dw_part[i], elems, elems, wgrad_used_splits(g), (accumulate || seen) ? 1 : 0, s);
            if (rc) return rc;
        }
        rc = rl.launch(s);
        if (rc) return rc;
    }
    return SSDK_OK;
}
