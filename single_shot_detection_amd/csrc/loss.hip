// loss.hip -- hard-negative sampler + multibox loss forward/backward + box coder.
//
// Reference: detection/sampler.py:9-25, detection/losses/multibox_loss.py:35-94, bf/modules/losses.py:34-54,
// detection/box_coder.py:13-57, bf/utils/box_utils.py:25-36.  The reference materialises a full log_softmax of
// [B,A,C], two full argsorts of A per image, boolean-mask gathers and an index_put backward.
//
// Kernels (all HBM-bound; the only dense passes over the [B,A,C] logits are ONE read in the forward
// (hnm_rows_kernel) and ONE write of dscores in the backward -- 8*A*C bytes per image, the algorithmic minimum):
//   hnm_rows_kernel    streams the logits through a 64-row LDS tile (16-byte coalesced loads, rows of C floats are
//                      not 16-byte aligned on their own), 4 lanes per row: per-anchor log-sum-exp (kept for the loss
//                      and its backward) and the background loss -log_softmax[...,0].
//   hnm_select_kernel  one workgroup per image: exact N-th-largest selection by 4x8-bit radix histograms in LDS
//                      instead of argsort(argsort()); ties at the threshold go to the lower anchor index.
//   loss_fwd_kernel    one thread per anchor: target encode in place (to_centroids + encode_box, every anchor, as
//                      the reference mutates it), smooth-L1 over positives, and the classification term for SAMPLED
//                      rows only (a gather of x[class] when the log-sum-exp is already there; otherwise the wave
//                      cooperates on each sampled row).  Partial sums per workgroup, fixed-order final reduce.
//   loss_bwd_kernel    64-row output tiles: zero-fill, fill the sampled rows, one coalesced write.
#include <algorithm>
#include <limits.h>
#include <math.h>

#include "common.h"

namespace ssdk {

constexpr int kTileRows = 64;
constexpr int kLossThreads = 256;
constexpr float kMarkPositive = -INFINITY;  // bgloss marker: positive anchor (sampler.py:22 sets -inf for non-negatives)
constexpr float kMarkIgnore = -1.0f;        // bgloss marker: ignored anchor (class -1)
constexpr int kSelMaxChunks = 128;          // hnm_select_kernel's tie table: anchors / 1024 (beyond it: the sequential tie pass)
constexpr int kSelCache = 8;                // values per thread that hnm_select_kernel keeps in registers (8 * 1024 anchors)

struct LossState {  // lives in the workspace, written by finalize_kernel, read by the backward
    float divider;   // max(1, #positives)               multibox_loss.py:88
    float mean_div;  // #sampled rows for focal 'mean', else 1
    float scale;     // soft-target losses: 1 / mean target mass (losses.py:89, :102-103), else 1
    float out3[3];
    int npos;
    int nrows;
    int pad;
};

struct LossWs {
    float* lse;      // [B*A]
    float* bgloss;   // [B*A]
    float* partials; // [3 * kMaxPartials] per-workgroup sums: classification, localisation, soft-target mass
    int* counts;     // [3 * kMaxPartials] per-workgroup counts: positives, sampled rows, sampled rows with a positive target
    LossState* state;
};
constexpr int kMaxPartials = 2048;

static LossWs carve_loss_ws(void* ws, size_t n_rows, size_t* total) {
    Carver c(ws);
    LossWs w;
    w.lse = c.take<float>(n_rows);
    w.bgloss = c.take<float>(n_rows);
    w.partials = c.take<float>(3 * kMaxPartials);
    w.counts = c.take<int>(3 * kMaxPartials);
    w.state = c.take<LossState>(1);
    if (total) *total = c.off;
    return w;
}

// radix key of a negative's background loss: fp32 bits are order-preserving for values >= +0; +1 keeps 0 free
__device__ __forceinline__ unsigned neg_key(float x) { return x > 0.0f ? __float_as_uint(x) + 1u : 1u; }

__device__ __forceinline__ float quad_max(float v) {
    v = fmaxf(v, __shfl_xor(v, 1, kWave));
    return fmaxf(v, __shfl_xor(v, 2, kWave));
}
__device__ __forceinline__ float quad_sum(float v) {
    v += __shfl_xor(v, 1, kWave);
    return v + __shfl_xor(v, 2, kWave);
}

// ---- S1: hard negative mining ---------------------------------------------------------------------------------

// Launched as exactly one resident set of workgroups that walk the tiles side by side (tile = blockIdx.x + k * gridDim.x: the
// workgroups in flight read neighbouring tiles): no second, partly filled round of workgroups at the end (25 -> 20 us at
// batch 32; contiguous per-workgroup row ranges or a register prefetch of the next tile were both slower).
__global__ void __launch_bounds__(kLossThreads) hnm_rows_kernel(const float* __restrict__ scores, const float* __restrict__ target_cls,
                                                                int cls_stride, long long n_rows, int C, float* __restrict__ lse_out,
                                                                float* __restrict__ bgloss_out) {
    extern __shared__ __attribute__((aligned(16))) float s_tile[];  // kTileRows * C floats
    const long long n_tiles = (n_rows + kTileRows - 1) / kTileRows;
    for (long long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const long long r0 = tile * kTileRows;
        const int rows = (int)min((long long)kTileRows, n_rows - r0);
        const int nfloat = rows * C;
        const float* src = scores + r0 * C;  // 16-byte aligned: r0*C*4 = tile*256*C
        __syncthreads();
        if (nfloat >> 2) stage_tile_f4(reinterpret_cast<float4*>(s_tile), reinterpret_cast<const float4*>(src), nfloat >> 2);
        for (int t = (nfloat & ~3) + threadIdx.x; t < nfloat; t += kLossThreads) s_tile[t] = src[t];
        __syncthreads();
        const int row = threadIdx.x >> 2, q = threadIdx.x & 3;
        const float* x = s_tile + row * C;
        const bool live = row < rows;
        float m = -INFINITY;
        if (live)
            for (int c = q; c < C; c += 4) m = fmaxf(m, x[c]);
        m = quad_max(m);
        float s = 0.0f;
        if (live)
            for (int c = q; c < C; c += 4) s += __expf(x[c] - m);
        s = quad_sum(s);
        if (live && q == 0) {
            const float ls = logf(s);
            const float cls = target_cls[(r0 + row) * cls_stride];
            const float loss = -((x[0] - m) - ls);  // sampler.py:13 -log_softmax[..., NEGATIVE_CLASS]
            lse_out[r0 + row] = m + ls;
            bgloss_out[r0 + row] = cls == 0.0f ? fmaxf(loss, 0.0f) : (cls == -1.0f ? kMarkIgnore : kMarkPositive);
        }
    }
}

// One workgroup per image.  Selects the `n` largest background losses among negatives (sampler.py:17-25).
__global__ void __launch_bounds__(1024) hnm_select_kernel(const float* __restrict__ bgloss, int A, double ratio,
                                                          long long min_neg, uint8_t* __restrict__ sampled) {
    __shared__ unsigned s_hist[256];
    __shared__ int s_cnt[2];
    __shared__ unsigned s_prefix, s_need, s_wave_cnt[16], s_run;
    __shared__ unsigned s_eq[kSelMaxChunks * 16];   // tie pass: keys equal to the threshold per (chunk of 1024 anchors, wave)
    const int i = blockIdx.x, tid = threadIdx.x;
    const float* v = bgloss + (size_t)i * A;
    uint8_t* out = sampled + (size_t)i * A;
    if (tid < 2) s_cnt[tid] = 0;
    __syncthreads();
    // the first kSelCache * 1024 values live in registers for all passes; anything beyond is re-read (L2 resident)
    float cache[kSelCache];
#pragma unroll
    for (int k = 0; k < kSelCache; ++k) {
        const int a = tid + k * 1024;
        cache[k] = a < A ? v[a] : kMarkIgnore;
    }
    int npos = 0, nneg = 0;
#pragma unroll
    for (int k = 0; k < kSelCache; ++k) { npos += cache[k] == kMarkPositive; nneg += cache[k] >= 0.0f; }
    for (int a = tid + kSelCache * 1024; a < A; a += blockDim.x) {
        const float x = v[a];
        npos += x == kMarkPositive;
        nneg += x >= 0.0f;
    }
    npos = wave_allreduce(npos, OpAddI());
    nneg = wave_allreduce(nneg, OpAddI());
    if (lane_id() == 0) { atomicAdd(&s_cnt[0], npos); atomicAdd(&s_cnt[1], nneg); }
    __syncthreads();
    npos = s_cnt[0]; nneg = s_cnt[1];
    // sampler.py:20  min(clamp(P * ratio, min=min_neg), #neg); `rank < n` with a fractional n keeps ceil(n) ranks
    long long n;
    if (ratio == floor(ratio)) {
        n = (long long)npos * (long long)ratio;
        if (n < min_neg) n = min_neg;
    } else {
        float want = (float)npos * (float)ratio;  // int64 tensor * python float -> float32 tensor
        if (want < (float)min_neg) want = (float)min_neg;
        n = (long long)ceilf(want);
    }
    if (n > nneg) n = nneg;
    if (n >= nneg || n <= 0) {  // every negative, or none: no ranking needed
        const bool all = n > 0;
        for (int a = tid; a < A; a += blockDim.x) {
            const float x = v[a];
            out[a] = (x == kMarkPositive) || (all && x >= 0.0f);
        }
        return;
    }
    // radix select of the n-th largest key among negatives; key = fp32 bits (losses are >= +0) + 1
    unsigned prefix = 0, need = (unsigned)n;
    for (int shift = 24; shift >= 0; shift -= 8) {
        for (int b = tid; b < 256; b += blockDim.x) s_hist[b] = 0;
        __syncthreads();
        const unsigned himask = shift == 24 ? 0u : (0xFFFFFFFFu << (shift + 8));
#pragma unroll
        for (int k = 0; k < kSelCache; ++k) {
            const float x = cache[k];
            if (x >= 0.0f) {
                const unsigned key = neg_key(x);
                if ((key & himask) == prefix) atomicAdd(&s_hist[(key >> shift) & 255u], 1u);
            }
        }
        for (int a = tid + kSelCache * 1024; a < A; a += blockDim.x) {
            const float x = v[a];
            if (x >= 0.0f) {
                const unsigned key = neg_key(x);
                if ((key & himask) == prefix) atomicAdd(&s_hist[(key >> shift) & 255u], 1u);
            }
        }
        __syncthreads();
        if (tid < kWave) {
            // wave 0 finds the digit: lane l owns bins 4l .. 4l+3; `above` = keys in the bins above its group (suffix sum over the
            // lanes).  The bin is the largest b with (keys in bins > b) + hist[b] >= need.  (One thread walking the 256 bins was
            // 255 dependent LDS reads per pass: ~8 us of the kernel's 58, four times.)
            const int l = tid;
            const unsigned h0 = s_hist[4 * l], h1 = s_hist[4 * l + 1], h2 = s_hist[4 * l + 2], h3 = s_hist[4 * l + 3];
            const unsigned mine = h0 + h1 + h2 + h3;
            unsigned incl = mine;   // inclusive suffix sum: lanes >= l
#pragma unroll
            for (int d = 1; d < kWave; d <<= 1) {
                const unsigned t = __shfl_down(incl, d, kWave);
                if (l + d < kWave) incl += t;
            }
            unsigned acc = incl - mine;
            int found = -1;
            unsigned rest = 0;
            const unsigned hq[4] = {h0, h1, h2, h3};
#pragma unroll
            for (int q = 3; q >= 0; --q) {
                if (found < 0) {
                    if (acc + hq[q] >= need) { found = 4 * l + q; rest = need - acc; }
                    else acc += hq[q];
                }
            }
            // the lane whose group holds the crossing reports it (highest such lane = highest bin); none (need > keys: cannot
            // happen for n <= #negatives) falls back to bin 0 like the sequential walk did
            const unsigned long long hit = __ballot(found >= 0);
            if (hit) {
                const int src = 63 - __clzll((long long)hit);
                if (l == src) { s_prefix = prefix | ((unsigned)found << shift); s_need = rest; }
            } else if (l == 0) {
                s_prefix = prefix;
                s_need = need - (incl - h0);
            }
        }
        __syncthreads();
        prefix = s_prefix; need = s_need;
        __syncthreads();
    }
    // prefix = key of the n-th largest; `need` of the anchors whose key == prefix are taken, lowest index first.
    // Chunk k = anchors k*1024 .. k*1024+1023 (the register cache's mapping).  Pass 1: per (chunk, wave) the number of keys equal
    // to the threshold; wave 0 turns the table into exclusive prefixes; pass 2: every anchor knows its rank among the ties.
    // (Three barriers in all; a barrier-separated running count per chunk cost 27.)
    const unsigned thr = prefix;
    const int chunks = (A + 1023) / 1024;
    const int w = tid >> 6, lane = lane_id();
    if (chunks <= kSelMaxChunks) {
        unsigned long long bal[kSelCache];
#pragma unroll
        for (int k = 0; k < kSelCache; ++k) {
            const float x = cache[k];
            bal[k] = __ballot(x >= 0.0f && neg_key(x) == thr);
            if (lane == 0 && k < chunks) s_eq[k * 16 + w] = (unsigned)__popcll(bal[k]);
        }
        for (int k = kSelCache; k < chunks; ++k) {
            const int a = k * 1024 + tid;
            const float x = a < A ? v[a] : kMarkIgnore;
            const unsigned long long bk = __ballot(x >= 0.0f && neg_key(x) == thr);
            if (lane == 0) s_eq[k * 16 + w] = (unsigned)__popcll(bk);
        }
        __syncthreads();
        if (tid < kWave) {   // exclusive prefix over the chunks * 16 entries: each lane a contiguous run, wave scan of the run totals
            const int entries = chunks * 16, per = (entries + kWave - 1) / kWave;
            const int e0 = min(tid * per, entries), e1 = min(e0 + per, entries);
            unsigned run = 0;
            for (int e = e0; e < e1; ++e) run += s_eq[e];
            unsigned incl = run;
#pragma unroll
            for (int d = 1; d < kWave; d <<= 1) {
                const unsigned t = __shfl_up(incl, d, kWave);
                if (tid >= d) incl += t;
            }
            unsigned acc = incl - run;
            for (int e = e0; e < e1; ++e) { const unsigned c = s_eq[e]; s_eq[e] = acc; acc += c; }
        }
        __syncthreads();
        const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
        for (int k = 0; k < kSelCache; ++k) {
            const int a = k * 1024 + tid;
            if (a < A) {
                const float x = cache[k];
                const bool neg = x >= 0.0f;
                const unsigned key = neg ? neg_key(x) : 0u;
                const unsigned before = s_eq[k * 16 + w] + (unsigned)__popcll(bal[k] & below);
                out[a] = (x == kMarkPositive) || (neg && key > thr) || (neg && key == thr && before < need);
            }
        }
        for (int k = kSelCache; k < chunks; ++k) {
            const int a = k * 1024 + tid;
            const float x = a < A ? v[a] : kMarkIgnore;
            const bool neg = x >= 0.0f;
            const unsigned key = neg ? neg_key(x) : 0u;
            const unsigned long long bk = __ballot(neg && key == thr);
            if (a < A) out[a] = (x == kMarkPositive) || (neg && key > thr) || (neg && key == thr && s_eq[k * 16 + w] + (unsigned)__popcll(bk & below) < need);
        }
        return;
    }
    // (more chunks than the table holds -- A > 131 072: a barrier-separated running count per chunk)
    if (tid == 0) s_run = 0;
    __syncthreads();
    for (int base = 0; base < A; base += blockDim.x) {
        const int a = base + tid;
        float x = kMarkIgnore;
        if (a < A) x = v[a];
        const bool neg = x >= 0.0f;
        const unsigned key = neg ? neg_key(x) : 0u;
        const bool eq = neg && key == thr;
        const unsigned long long bal = __ballot(eq);
        if (lane == 0) s_wave_cnt[w] = (unsigned)__popcll(bal);
        __syncthreads();
        unsigned before = s_run;
        for (int k = 0; k < w; ++k) before += s_wave_cnt[k];
        before += (unsigned)__popcll(bal & ((1ull << lane) - 1ull));
        if (a < A) out[a] = (x == kMarkPositive) || (neg && key > thr) || (eq && before < need);
        __syncthreads();
        if (tid == 0) {
            unsigned tot = 0;
            for (int k = 0; k < (int)(blockDim.x >> 6); ++k) tot += s_wave_cnt[k];
            s_run += tot;
        }
        __syncthreads();
    }
}

__global__ void __launch_bounds__(256) naive_sampler_kernel(const float* __restrict__ target_cls, int cls_stride, long long n_rows,
                                                            uint8_t* __restrict__ sampled) {
    for (long long r = blockIdx.x * (long long)blockDim.x + threadIdx.x; r < n_rows; r += (long long)gridDim.x * blockDim.x) {
        const float cls = target_cls[r * cls_stride];
        sampled[r] = cls != 0.0f && cls != -1.0f;  // sampler.py:10
    }
}

// ---- L1/L2/L3 forward -----------------------------------------------------------------------------------------

// torch smooth_l1: z < beta ? 0.5 z^2 / beta : z - 0.5 beta
__device__ __forceinline__ float smooth_l1(float a, float b, float beta) {
    const float z = fabsf(a - b);
    return z < beta ? 0.5f * z * z / beta : z - 0.5f * beta;
}
__device__ __forceinline__ float smooth_l1_grad(float a, float b, float beta) {
    const float d = a - b;
    return d <= -beta ? -1.0f : (d >= beta ? 1.0f : d / beta);
}
__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + __expf(-x)); }

// bf/modules/losses.py:42-52 for one element; tg = one-hot target value
__device__ __forceinline__ float focal_elem(float x, float tg, float gamma, float alpha) {
    const float aw = tg * alpha + (1.0f - tg) * (1.0f - alpha);
    float pb = sigmoid_f(x);
    pb = pb * tg + (1.0f - pb) * (1.0f - tg);
    const float ce = fmaxf(x, 0.0f) - x * tg + log1pf(__expf(-fabsf(x)));
    return aw * __powf(1.0f - pb, gamma) * ce;
}
__device__ __forceinline__ float focal_elem_grad(float x, float tg, float gamma, float alpha) {
    const float aw = tg * alpha + (1.0f - tg) * (1.0f - alpha);
    const float s = sigmoid_f(x);
    const float pb = s * tg + (1.0f - s) * (1.0f - tg);
    const float ce = fmaxf(x, 0.0f) - x * tg + log1pf(__expf(-fabsf(x)));
    const float om = 1.0f - pb;
    const float dpb = s * (1.0f - s) * (2.0f * tg - 1.0f);
    return aw * (-gamma * __powf(om, gamma - 1.0f) * dpb * ce + __powf(om, gamma) * (s - tg));
}

struct LossParams {
    int cls_kind, C, lse_valid;
    float gamma, alpha, xy_scale, wh_scale, eps, beta;
    int loc_kind;
    float epsilon;  // label smoothing of the soft-target losses (losses.py:13-18)
};

// losses.py:13-18 _soften for a target row with one positive entry of value s at column pos (pos < 0: all-zero row)
__device__ __forceinline__ float soft_t(int c, int pos, float s, int C, float eps) {
    if (pos < 0) return 0.0f;
    if (eps == 0.0f) return c == pos ? s : 0.0f;
    return c == pos ? s - eps * s : eps * s / (float)(C - 1);
}

// decoded corner box of a prediction (box_coder.py:55-57 + box_utils.py:16-23); cen returns the centroid form
__device__ __forceinline__ float4 decode_corners(float4 t, float4 p, float xy_scale, float wh_scale, float4& cen) {
    cen = make_float4(p.x + p.z * t.x / xy_scale, p.y + p.w * t.y / xy_scale, p.z * expf(t.z / wh_scale), p.w * expf(t.w / wh_scale));
    return to_corners(cen);
}

// 1 - generalized_iou (box_utils.py:104-143, cartesian=False) of predicted corners P against target corners T
__device__ __forceinline__ float giou_loss(float4 P, float4 T) {
    const float inter = area4(tmaxf(P.x, T.x), tmaxf(P.y, T.y), tminf(P.z, T.z), tminf(P.w, T.w));
    const float uni = area4(P.x, P.y, P.z, P.w) + area4(T.x, T.y, T.z, T.w) - inter;
    const float enc = area4(tminf(P.x, T.x), tminf(P.y, T.y), tmaxf(P.z, T.z), tmaxf(P.w, T.w));
    return 1.0f - (inter / uni - (enc - uni) / enc);
}

// d(giou_loss)/dP: loss = 2 - inter/uni - uni/enc; max/min route the gradient to the selected argument (ties: half),
// clamp(0) passes it where its argument is >= 0 (what autograd does for box_utils.py:104-143)
__device__ __forceinline__ float4 giou_loss_grad(float4 P, float4 T) {
    const float pw = P.z - P.x, ph = P.w - P.y, cw = clamp0(pw), ch = clamp0(ph);
    const float area_p = cw * ch, area_t = area4(T.x, T.y, T.z, T.w);
    const float ix1 = tmaxf(P.x, T.x), iy1 = tmaxf(P.y, T.y), ix2 = tminf(P.z, T.z), iy2 = tminf(P.w, T.w);
    const float iw = clamp0(ix2 - ix1), ih = clamp0(iy2 - iy1), inter = iw * ih;
    const float uni = area_p + area_t - inter;
    const float ex1 = tminf(P.x, T.x), ey1 = tminf(P.y, T.y), ex2 = tmaxf(P.z, T.z), ey2 = tmaxf(P.w, T.w);
    const float ew = clamp0(ex2 - ex1), eh = clamp0(ey2 - ey1), enc = ew * eh;
    const float d_uni = inter / (uni * uni) - 1.0f / enc, d_enc = uni / (enc * enc);
    const float c_inter = -1.0f / uni - d_uni, c_area = d_uni;
    float4 d = make_float4(0.f, 0.f, 0.f, 0.f);
    auto sel = [](float a, float b, bool greater) { return greater ? (a > b ? 1.0f : (a == b ? 0.5f : 0.0f)) : (a < b ? 1.0f : (a == b ? 0.5f : 0.0f)); };
    if (pw >= 0.0f) { d.z += c_area * ch; d.x -= c_area * ch; }
    if (ph >= 0.0f) { d.w += c_area * cw; d.y -= c_area * cw; }
    if (ix2 - ix1 >= 0.0f) { const float g = c_inter * ih; d.z += g * sel(P.z, T.z, false); d.x -= g * sel(P.x, T.x, true); }
    if (iy2 - iy1 >= 0.0f) { const float g = c_inter * iw; d.w += g * sel(P.w, T.w, false); d.y -= g * sel(P.y, T.y, true); }
    if (ex2 - ex1 >= 0.0f) { const float g = d_enc * eh; d.z += g * sel(P.z, T.z, true); d.x -= g * sel(P.x, T.x, false); }
    if (ey2 - ey1 >= 0.0f) { const float g = d_enc * ew; d.w += g * sel(P.w, T.w, true); d.y -= g * sel(P.y, T.y, false); }
    return d;
}

__global__ void __launch_bounds__(kLossThreads) loss_fwd_kernel(LossParams p, const float* __restrict__ scores,
                                                                const float4* __restrict__ locs, const float4* __restrict__ anchors,
                                                                float* __restrict__ target, const uint8_t* __restrict__ sampled,
                                                                long long n_rows, int A, float* __restrict__ lse,
                                                                float* __restrict__ partials, int* __restrict__ counts) {
    __shared__ float s_red[kLossThreads / kWave];
    __shared__ int s_redi[kLossThreads / kWave];
    float cls_acc = 0.0f, loc_acc = 0.0f, t_acc = 0.0f;
    int npos = 0, nrows = 0, nposrows = 0;
    const int lane = lane_id();
    const long long stride = (long long)gridDim.x * blockDim.x;
    const long long n_iter = (n_rows + stride - 1) / stride;
    for (long long it = 0; it < n_iter; ++it) {
        const long long r = it * stride + blockIdx.x * (long long)blockDim.x + threadIdx.x;
        const bool live = r < n_rows;
        int cls = 0;
        float tscore = 0.0f;
        bool smp = false, pos = false;
        if (live) {
            float2* trow = reinterpret_cast<float2*>(target + r * 6);  // 24-byte rows: 8-byte aligned
            float2 t01 = trow[0], t23 = trow[1];
            const float2 t45 = trow[2];
            cls = (int)t45.x;  // multibox_loss.py:48 .long()
            tscore = t45.y;
            pos = cls != 0 && cls != -1;
            smp = sampled[r] != 0;
            npos += pos;
            const float4 pr = anchors[r % A];
            if (p.loc_kind == SSDK_LOC_GIOU) {  // multibox_loss.py:77-79: decoded corners vs the RAW target, no mutation
                if (pos) {
                    float4 cen;
                    loc_acc += giou_loss(decode_corners(locs[r], pr, p.xy_scale, p.wh_scale, cen), make_float4(t01.x, t01.y, t23.x, t23.y));
                }
            } else {
            // box_utils.py:33-34 to_centroids(inplace), then box_coder.py:22-30 encode_box(inplace)
            t23.x -= t01.x; t23.y -= t01.y;
            t01.x += t23.x / 2.0f; t01.y += t23.y / 2.0f;
            t01.x -= pr.x; t01.y -= pr.y;
            t01.x /= pr.z; t01.y /= pr.w;
            t01.x *= p.xy_scale; t01.y *= p.xy_scale;
            t23.x /= pr.z; t23.y /= pr.w;
            t23.x += p.eps; t23.y += p.eps;
            t23.x = logf(t23.x); t23.y = logf(t23.y);
            t23.x *= p.wh_scale; t23.y *= p.wh_scale;
            trow[0] = t01; trow[1] = t23;
            if (pos) {  // multibox_loss.py:84-86
                const float4 l = locs[r];
                loc_acc += smooth_l1(l.x, t01.x, p.beta) + smooth_l1(l.y, t01.y, p.beta) + smooth_l1(l.z, t23.x, p.beta) +
                           smooth_l1(l.w, t23.y, p.beta);
            }
            }
        }
        const bool hard_label = p.cls_kind == SSDK_CLS_CROSS_ENTROPY || p.cls_kind == SSDK_CLS_SOFTMAX_FOCAL;
        const bool want = hard_label ? (smp && cls != -1) : smp;  // CE / softmax focal: ignore_index = -1
        nrows += smp;
        if (hard_label && p.lse_valid) {
            if (want) {
                const float nlogpb = lse[r] - scores[r * p.C + cls];
                if (p.cls_kind == SSDK_CLS_CROSS_ENTROPY) {
                    cls_acc += nlogpb;
                } else {  // losses.py:56-78
                    float l = __powf(1.0f - __expf(-nlogpb), p.gamma) * nlogpb;
                    if (p.alpha >= 0.0f) l *= (cls == 0 ? 1.0f - p.alpha : p.alpha);
                    cls_acc += l;
                }
            }
        } else {
            unsigned long long todo = __ballot(want);
            while (todo) {  // the wave cooperates on each sampled row
                const int src = __ffsll((long long)todo) - 1;
                todo &= todo - 1;
                const long long rs = __shfl(r, src, kWave);
                const int cs = __shfl(cls, src, kWave);
                const float ts = __shfl(tscore, src, kWave);
                const float* x = scores + rs * p.C;
                float val;
                if (p.cls_kind != SSDK_CLS_SIGMOID_FOCAL && p.cls_kind != SSDK_CLS_BCE_SOFT) {  // softmax family: needs the row's log-sum-exp
                    float m = -INFINITY;
                    for (int c = lane; c < p.C; c += kWave) m = fmaxf(m, x[c]);
                    m = wave_allreduce(m, OpMaxF());
                    float s = 0.0f;
                    for (int c = lane; c < p.C; c += kWave) s += __expf(x[c] - m);
                    s = wave_allreduce(s, OpAddF());
                    const float l = m + logf(s);
                    if (lane == src) lse[rs] = l;
                    if (p.cls_kind == SSDK_CLS_CROSS_ENTROPY) {
                        val = l - x[cs];
                    } else if (p.cls_kind == SSDK_CLS_SOFTMAX_FOCAL) {
                        const float nlogpb = l - x[cs];
                        val = __powf(1.0f - __expf(-nlogpb), p.gamma) * nlogpb;
                        if (p.alpha >= 0.0f) val *= (cs == 0 ? 1.0f - p.alpha : p.alpha);
                    } else {  // SSDK_CLS_CE_SOFT, losses.py:80-93 on the target of multibox_loss.py:68-71
                        const int pos_c = cs != -1 ? cs : -1;
                        float f = 0.0f, rsum = 0.0f;
                        for (int c = lane; c < p.C; c += kWave) {
                            const float tc = soft_t(c, pos_c, ts, p.C, p.epsilon);
                            f += (x[c] - l) * tc;
                            rsum += tc;
                        }
                        val = -wave_allreduce(f, OpAddF());
                        rsum = wave_allreduce(rsum, OpAddF());
                        if (lane == src) t_acc += rsum;
                    }
                } else if (p.cls_kind == SSDK_CLS_BCE_SOFT) {  // losses.py:95-106 on the target of multibox_loss.py:64-67
                    const bool ps = cs != 0 && cs != -1;
                    const int pos_c = ps ? cs - 1 : -1;
                    float f = 0.0f, rsum = 0.0f;
                    for (int c = lane; c < p.C; c += kWave) {
                        const float tc = soft_t(c, pos_c, ts, p.C, p.epsilon);
                        f += fmaxf(x[c], 0.0f) - x[c] * tc + log1pf(__expf(-fabsf(x[c])));
                        rsum += tc;
                    }
                    val = wave_allreduce(f, OpAddF());
                    rsum = wave_allreduce(rsum, OpAddF()) / (float)p.C;
                    if (lane == src) { t_acc += rsum; nposrows += rsum > 0.0f; }
                } else {
                    const bool ps = cs != 0 && cs != -1;
                    float f = 0.0f;
                    for (int c = lane; c < p.C; c += kWave)
                        f += focal_elem(x[c], (ps && c == cs - 1) ? ts : 0.0f, p.gamma, p.alpha);  // multibox_loss.py:64-67
                    val = wave_allreduce(f, OpAddF());
                }
                if (lane == src) cls_acc += val;
            }
        }
    }
    const float cs_ = block_sum(cls_acc, s_red);
    const float ls_ = block_sum(loc_acc, s_red);
    const float ts_ = block_sum(t_acc, s_red);
    const int np_ = block_sum(npos, s_redi);
    const int nr_ = block_sum(nrows, s_redi);
    const int npr_ = block_sum(nposrows, s_redi);
    if (threadIdx.x == 0) {
        partials[blockIdx.x] = cs_;
        partials[kMaxPartials + blockIdx.x] = ls_;
        partials[2 * kMaxPartials + blockIdx.x] = ts_;
        // per-workgroup counts, summed by loss_finalize_kernel: a thousand workgroups adding into two words of one line serialised at
        // ~12 ns per atomic -- 21 of the kernel's 21 us at batch 32 -- and needed a zero-fill launch in front
        counts[blockIdx.x] = np_;
        counts[kMaxPartials + blockIdx.x] = nr_;
        counts[2 * kMaxPartials + blockIdx.x] = npr_;
    }
}

__global__ void __launch_bounds__(256) loss_finalize_kernel(const float* __restrict__ partials, int n_part,
                                                            const int* __restrict__ counts, int cls_kind, int reduce_mean,
                                                            float cls_w, float loc_w, LossState* __restrict__ state,
                                                            float* __restrict__ out3) {
    __shared__ double s_red[4];
    __shared__ int s_redi[4];
    double c = 0.0, l = 0.0, t = 0.0;
    int n0 = 0, n1 = 0, n2 = 0;
    for (int k = threadIdx.x; k < n_part; k += blockDim.x) {
        c += (double)partials[k]; l += (double)partials[kMaxPartials + k]; t += (double)partials[2 * kMaxPartials + k];
        n0 += counts[k]; n1 += counts[kMaxPartials + k]; n2 += counts[2 * kMaxPartials + k];
    }
    c = block_sum(c, s_red);
    l = block_sum(l, s_red);
    t = block_sum(t, s_red);
    n0 = block_sum(n0, s_redi);
    n1 = block_sum(n1, s_redi);
    n2 = block_sum(n2, s_redi);
    if (threadIdx.x == 0) {
        const int npos = n0, nrows = n1, nposrows = n2;
        // reduce_mean == 2: the plain sums, not divided by the positives (the standalone modules of bf/modules/losses.py)
        const float divider = reduce_mean == 2 ? 1.0f : (float)(npos < 1 ? 1 : npos);  // multibox_loss.py:88
        const bool focal = cls_kind == SSDK_CLS_SIGMOID_FOCAL || cls_kind == SSDK_CLS_SOFTMAX_FOCAL;
        const float mean_div = (focal && reduce_mean == 1) ? (float)nrows : 1.0f;
        float scale = 1.0f;
        if (cls_kind == SSDK_CLS_CE_SOFT) scale = 1.0f / ((float)t / (float)nrows);       // losses.py:89 target.sum(-1).mean() ** -1
        if (cls_kind == SSDK_CLS_BCE_SOFT) scale = 1.0f / ((float)t / (float)nposrows);   // losses.py:102-103
        const float class_loss = scale * (float)c / mean_div * cls_w / divider;  // :90
        const float loc_loss = (float)l * loc_w / divider;               // :89
        state->scale = scale;
        state->divider = divider; state->mean_div = mean_div; state->npos = npos; state->nrows = nrows;
        state->out3[0] = class_loss + loc_loss; state->out3[1] = class_loss; state->out3[2] = loc_loss;
        out3[0] = class_loss + loc_loss; out3[1] = class_loss; out3[2] = loc_loss;
    }
}

// ---- backward ------------------------------------------------------------------------------------------------

__global__ void __launch_bounds__(kLossThreads) loss_bwd_kernel(LossParams p, int reduce_mean, float cls_w, float loc_w,
                                                                const float* __restrict__ scores, const float4* __restrict__ locs,
                                                                const float4* __restrict__ anchors, int A,
                                                                const float* __restrict__ target, const uint8_t* __restrict__ sampled,
                                                                const float* __restrict__ grad_out, long long n_rows,
                                                                const float* __restrict__ lse, const LossState* __restrict__ state,
                                                                float* __restrict__ dscores, float4* __restrict__ dlocs,
                                                                unsigned char* __restrict__ row_mask, int grad_single) {
    extern __shared__ __attribute__((aligned(16))) float s_tile[];  // kTileRows * C floats
    __shared__ int s_cls[kTileRows];
    __shared__ float s_tscore[kTileRows];
    __shared__ int s_any;
    const float divider = state->divider;
    const float g_cls = grad_out[0] * state->scale * cls_w / divider / (reduce_mean ? state->mean_div : 1.0f);
    const float g_loc = grad_out[grad_single ? 0 : 1] * loc_w / divider;   // (grad_single: one upstream scalar, the gradient of class_loss + loc_loss)
    const long long n_tiles = (n_rows + kTileRows - 1) / kTileRows;
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    for (long long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const long long r0 = tile * kTileRows;
        const int rows = (int)min((long long)kTileRows, n_rows - r0);
        __syncthreads();
        if (threadIdx.x == 0) s_any = 0;
        __syncthreads();
        if (threadIdx.x < rows) {
            const long long r = r0 + threadIdx.x;
            const float2* trow = reinterpret_cast<const float2*>(target + r * 6);
            const float2 t01 = trow[0], t23 = trow[1], t45 = trow[2];
            const int cls = (int)t45.x;
            const bool pos = cls != 0 && cls != -1;
            const bool smp = sampled[r] != 0;
            const bool hard_label = p.cls_kind == SSDK_CLS_CROSS_ENTROPY || p.cls_kind == SSDK_CLS_SOFTMAX_FOCAL;
            const bool want = hard_label ? (smp && cls != -1) : smp;
            s_cls[threadIdx.x] = want ? cls : INT_MIN;
            s_tscore[threadIdx.x] = t45.y;
            if (want) s_any = 1;
            float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
            if (pos) {
                const float4 l = locs[r];
                if (p.loc_kind == SSDK_LOC_GIOU) {
                    const float4 pr = anchors[r % A];
                    float4 cen;
                    const float4 P = decode_corners(l, pr, p.xy_scale, p.wh_scale, cen);
                    const float4 d = giou_loss_grad(P, make_float4(t01.x, t01.y, t23.x, t23.y));
                    // corners = c -+ wh/2 ; c = p_xy + p_wh t / xy_scale ; wh = p_wh exp(t / wh_scale)
                    g = make_float4((d.x + d.z) * pr.z / p.xy_scale * g_loc, (d.y + d.w) * pr.w / p.xy_scale * g_loc,
                                    (d.z - d.x) * 0.5f * cen.z / p.wh_scale * g_loc, (d.w - d.y) * 0.5f * cen.w / p.wh_scale * g_loc);
                } else {
                    g = make_float4(smooth_l1_grad(l.x, t01.x, p.beta) * g_loc, smooth_l1_grad(l.y, t01.y, p.beta) * g_loc,
                                    smooth_l1_grad(l.z, t23.x, p.beta) * g_loc, smooth_l1_grad(l.w, t23.y, p.beta) * g_loc);
                }
            }
            dlocs[r] = g;
            // anchors whose gradient rows can be non-zero (a classification term, or a box term): what ssdk_heads_bwd_ex may rely on
            if (row_mask) row_mask[r] = (want || pos) ? 1 : 0;
        }
        __syncthreads();
        const int nfloat = rows * p.C;
        float* dst = dscores + r0 * p.C;
        if (!s_any) {
            const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int t = threadIdx.x; t < (nfloat >> 2); t += kLossThreads) reinterpret_cast<float4*>(dst)[t] = z;
            for (int t = (nfloat & ~3) + threadIdx.x; t < nfloat; t += kLossThreads) dst[t] = 0.0f;
            continue;
        }
        for (int t = threadIdx.x; t < nfloat; t += kLossThreads) s_tile[t] = 0.0f;
        __syncthreads();
        for (int row = wave; row < rows; row += kLossThreads / kWave) {
            const int cls = s_cls[row];
            if (cls == INT_MIN) continue;
            const long long r = r0 + row;
            const float* x = scores + r * p.C;
            float* o = s_tile + row * p.C;
            if (p.cls_kind == SSDK_CLS_CROSS_ENTROPY || p.cls_kind == SSDK_CLS_SOFTMAX_FOCAL) {
                const float l = lse[r];
                float coef = 1.0f;  // dL/dx_c = coef * (p_c - onehot_c)
                if (p.cls_kind == SSDK_CLS_SOFTMAX_FOCAL) {
                    const float logpb = x[cls] - l, pb = __expf(logpb), om = 1.0f - pb;
                    coef = -(p.gamma * __powf(om, p.gamma - 1.0f) * pb * logpb - __powf(om, p.gamma));
                    if (p.alpha >= 0.0f) coef *= (cls == 0 ? 1.0f - p.alpha : p.alpha);
                }
                for (int c = lane; c < p.C; c += kWave) o[c] = coef * (__expf(x[c] - l) - (c == cls ? 1.0f : 0.0f)) * g_cls;
            } else if (p.cls_kind == SSDK_CLS_CE_SOFT) {
                const float l = lse[r], ts = s_tscore[row];
                const int pos_c = cls != -1 ? cls : -1;
                float rsum = 0.0f;
                for (int c = lane; c < p.C; c += kWave) rsum += soft_t(c, pos_c, ts, p.C, p.epsilon);
                rsum = wave_allreduce(rsum, OpAddF());
                for (int c = lane; c < p.C; c += kWave) o[c] = (__expf(x[c] - l) * rsum - soft_t(c, pos_c, ts, p.C, p.epsilon)) * g_cls;
            } else if (p.cls_kind == SSDK_CLS_BCE_SOFT) {
                const bool ps = cls != 0 && cls != -1;
                const int pos_c = ps ? cls - 1 : -1;
                const float ts = s_tscore[row];
                for (int c = lane; c < p.C; c += kWave) o[c] = (sigmoid_f(x[c]) - soft_t(c, pos_c, ts, p.C, p.epsilon)) * g_cls;
            } else {
                const bool ps = cls != 0 && cls != -1;
                const float ts = s_tscore[row];
                for (int c = lane; c < p.C; c += kWave)
                    o[c] = focal_elem_grad(x[c], (ps && c == cls - 1) ? ts : 0.0f, p.gamma, p.alpha) * g_cls;
            }
        }
        __syncthreads();
        for (int t = threadIdx.x; t < (nfloat >> 2); t += kLossThreads)
            reinterpret_cast<float4*>(dst)[t] = reinterpret_cast<const float4*>(s_tile)[t];
        for (int t = (nfloat & ~3) + threadIdx.x; t < nfloat; t += kLossThreads) dst[t] = s_tile[t];
    }
}

// ---- box coder (elementwise) ------------------------------------------------------------------------------------

__global__ void __launch_bounds__(256) encode_box_kernel(const float4* __restrict__ boxes, const float4* __restrict__ priors,
                                                         float4* __restrict__ out, long long n, int A, float xy_scale,
                                                         float wh_scale, float eps, int inplace_semantics) {
    for (long long r = blockIdx.x * (long long)blockDim.x + threadIdx.x; r < n; r += (long long)gridDim.x * blockDim.x) {
        const float4 b = boxes[r], p = priors[r % A];
        float4 o;
        if (inplace_semantics) {  // box_coder.py:22-30: eps after the divide
            o.x = (b.x - p.x) / p.z * xy_scale;
            o.y = (b.y - p.y) / p.w * xy_scale;
            o.z = logf(b.z / p.z + eps) * wh_scale;
            o.w = logf(b.w / p.w + eps) * wh_scale;
        } else {  // box_coder.py:32-34: eps before the divide
            o.x = (b.x - p.x) / p.z * xy_scale;
            o.y = (b.y - p.y) / p.w * xy_scale;
            o.z = logf((b.z + eps) / p.z) * wh_scale;
            o.w = logf((b.w + eps) / p.w) * wh_scale;
        }
        out[r] = o;
    }
}

__global__ void __launch_bounds__(256) decode_box_kernel(const float4* __restrict__ locs, const float4* __restrict__ priors,
                                                         float4* __restrict__ out, long long n, int A, float xy_scale, float wh_scale,
                                                         int inplace_semantics) {
    for (long long r = blockIdx.x * (long long)blockDim.x + threadIdx.x; r < n; r += (long long)gridDim.x * blockDim.x) {
        const float4 t = locs[r], p = priors[r % A];
        if (inplace_semantics)  // box_coder.py:45-53: /= scale, *= p_wh, += p_xy ; /= scale, exp, *= p_wh
            out[r] = make_float4(t.x / xy_scale * p.z + p.x, t.y / xy_scale * p.w + p.y, expf(t.z / wh_scale) * p.z,
                                 expf(t.w / wh_scale) * p.w);
        else  // box_coder.py:55-57
            out[r] = make_float4(p.x + p.z * t.x / xy_scale, p.y + p.w * t.y / xy_scale, p.z * expf(t.z / wh_scale),
                                 p.w * expf(t.w / wh_scale));
    }
}

static inline int stream_grid(long long work_items, int per_block) {
    long long b = (work_items + per_block - 1) / per_block;
    if (b < 1) b = 1;
    return (int)(b > 2048 ? 2048 : b);
}

// Workgroups of `fn` (kLossThreads threads, `lds` dynamic bytes) that the whole device holds at once.
static int resident_blocks(const void* fn, size_t lds) {
    int dev = 0, cus = 256, per_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, kLossThreads, lds) != hipSuccess || per_cu < 1) per_cu = 4;
    return cus * per_cu;
}

}  // namespace ssdk

using namespace ssdk;

extern "C" size_t ssdk_multibox_loss_workspace_bytes(int batch, int num_anchors, int num_classes) {
    (void)num_classes;
    size_t total = 0;
    carve_loss_ws(nullptr, (size_t)batch * (size_t)num_anchors, &total);
    return total;
}

static int check_loss_common(const char* fn, const void* scores, int batch, int A, int C, void* ws, size_t ws_bytes) {
    SSDK_REQUIRE(batch > 0 && A > 0 && C > 0, SSDK_E_INVALID, "%s: batch=%d anchors=%d classes=%d", fn, batch, A, C);
    SSDK_REQUIRE(scores && ((uintptr_t)scores & 15) == 0, SSDK_E_INVALID, "%s: scores must be non-null and 16-byte aligned", fn);
    SSDK_REQUIRE((size_t)kTileRows * C * sizeof(float) <= 160 * 1024 - 4096, SSDK_E_UNSUPPORTED, "%s: num_classes=%d too large for an LDS tile", fn, C);
    SSDK_REQUIRE(ws && ws_bytes >= ssdk_multibox_loss_workspace_bytes(batch, A, C), SSDK_E_WORKSPACE, "%s: workspace too small", fn);
    return SSDK_OK;
}

extern "C" int ssdk_hard_negative_mining(const float* scores, const float* target_classes, int class_stride, int batch,
                                         int num_anchors, int num_classes, double negative_per_positive_ratio,
                                         int64_t min_negative_per_image, uint8_t* sampled, void* workspace,
                                         size_t workspace_bytes, void* stream) {
    int rc = check_loss_common("ssdk_hard_negative_mining", scores, batch, num_anchors, num_classes, workspace, workspace_bytes);
    if (rc) return rc;
    SSDK_REQUIRE(target_classes && sampled && class_stride >= 1, SSDK_E_INVALID, "ssdk_hard_negative_mining: null pointer / bad stride");
    SSDK_REQUIRE(negative_per_positive_ratio >= 0, SSDK_E_INVALID, "ssdk_hard_negative_mining: negative ratio");
    hipStream_t s = (hipStream_t)stream;
    const long long n_rows = (long long)batch * num_anchors;
    LossWs w = carve_loss_ws(workspace, (size_t)n_rows, nullptr);
    const size_t lds = (size_t)kTileRows * num_classes * sizeof(float);
    if (lds > 48 * 1024)
        SSDK_CHECK_HIP(hipFuncSetAttribute((const void*)hnm_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(hnm_rows_kernel, dim3(std::min<long long>((n_rows + kTileRows - 1) / kTileRows, resident_blocks((const void*)hnm_rows_kernel, lds))), dim3(kLossThreads), lds, s, scores, target_classes, class_stride,
                       n_rows, num_classes, w.lse, w.bgloss);
    SSDK_CHECK_LAUNCH("hnm_rows_kernel");
    hipLaunchKernelGGL(hnm_select_kernel, dim3(batch), dim3(1024), 0, s, w.bgloss, num_anchors, negative_per_positive_ratio,
                       (long long)min_negative_per_image, sampled);
    SSDK_CHECK_LAUNCH("hnm_select_kernel");
    return SSDK_OK;
}

extern "C" int ssdk_naive_sampler(const float* target_classes, int class_stride, int batch, int num_anchors, uint8_t* sampled,
                                  void* stream) {
    SSDK_REQUIRE(target_classes && sampled && batch > 0 && num_anchors > 0 && class_stride >= 1, SSDK_E_INVALID, "ssdk_naive_sampler: bad arguments");
    const long long n_rows = (long long)batch * num_anchors;
    hipLaunchKernelGGL(naive_sampler_kernel, dim3(stream_grid(n_rows, 256)), dim3(256), 0, (hipStream_t)stream, target_classes, class_stride, n_rows, sampled);
    SSDK_CHECK_LAUNCH("naive_sampler_kernel");
    return SSDK_OK;
}

static int check_loss_params(const char* fn, const ssdk_loss_params* q) {
    SSDK_REQUIRE(q, SSDK_E_INVALID, "%s: null params", fn);
    SSDK_REQUIRE(q->cls_kind >= SSDK_CLS_CROSS_ENTROPY && q->cls_kind <= SSDK_CLS_BCE_SOFT, SSDK_E_INVALID, "%s: cls_kind=%d", fn, q->cls_kind);
    SSDK_REQUIRE(q->loc_kind == SSDK_LOC_SMOOTH_L1 || q->loc_kind == SSDK_LOC_GIOU, SSDK_E_INVALID, "%s: loc_kind=%d", fn, q->loc_kind);
    SSDK_REQUIRE(q->smooth_l1_beta > 0, SSDK_E_INVALID, "%s: beta must be > 0", fn);
    SSDK_REQUIRE(q->soft_epsilon >= 0.0f && q->soft_epsilon < 1.0f, SSDK_E_INVALID, "%s: epsilon outside [0, 1) (losses.py:15)", fn);
    SSDK_REQUIRE(q->reduce_mean >= 0 && q->reduce_mean <= 2, SSDK_E_INVALID, "%s: reduce_mean=%d (0 sum, 1 mean, 2 sums not divided by the positives)", fn, q->reduce_mean);
    return SSDK_OK;
}

static LossParams to_device_params(const ssdk_loss_params* q, int C, int lse_valid) {
    LossParams p{};
    p.cls_kind = q->cls_kind; p.C = C; p.lse_valid = lse_valid;
    p.gamma = q->focal_gamma; p.alpha = q->focal_alpha; p.xy_scale = q->xy_scale; p.wh_scale = q->wh_scale; p.eps = q->eps;
    p.beta = q->smooth_l1_beta; p.loc_kind = q->loc_kind; p.epsilon = q->soft_epsilon;
    return p;
}

extern "C" int ssdk_multibox_loss_fwd(const ssdk_loss_params* params, const float* scores, const float* locs, const float* anchors,
                                      float* target, const uint8_t* sampled, int batch, int num_anchors, int num_classes, int lse_valid,
                                      float* out3, void* workspace, size_t workspace_bytes, void* stream) {
    int rc = check_loss_common("ssdk_multibox_loss_fwd", scores, batch, num_anchors, num_classes, workspace, workspace_bytes);
    if (rc) return rc;
    rc = check_loss_params("ssdk_multibox_loss_fwd", params);
    if (rc) return rc;
    SSDK_REQUIRE(locs && anchors && target && sampled && out3, SSDK_E_INVALID, "ssdk_multibox_loss_fwd: null pointer");
    SSDK_REQUIRE(((uintptr_t)locs & 15) == 0 && ((uintptr_t)anchors & 15) == 0 && ((uintptr_t)target & 7) == 0, SSDK_E_INVALID,
                 "ssdk_multibox_loss_fwd: locs/anchors need 16-byte and target 8-byte alignment");
    hipStream_t s = (hipStream_t)stream;
    const long long n_rows = (long long)batch * num_anchors;
    LossWs w = carve_loss_ws(workspace, (size_t)n_rows, nullptr);
    LossParams p = to_device_params(params, num_classes, lse_valid);
    const int grid = stream_grid(n_rows, kLossThreads);
    hipLaunchKernelGGL(loss_fwd_kernel, dim3(grid), dim3(kLossThreads), 0, s, p, scores, (const float4*)locs, (const float4*)anchors,
                       target, sampled, n_rows, num_anchors, w.lse, w.partials, w.counts);
    SSDK_CHECK_LAUNCH("loss_fwd_kernel");
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, s, w.partials, grid, w.counts, params->cls_kind, params->reduce_mean,
                       params->classification_weight, params->localization_weight, w.state, out3);
    SSDK_CHECK_LAUNCH("loss_finalize_kernel");
    return SSDK_OK;
}

extern "C" int ssdk_multibox_loss_bwd(const ssdk_loss_params* params, const float* scores, const float* locs, const float* anchors,
                                      const float* target, const uint8_t* sampled, const float* grad_out, int batch, int num_anchors,
                                      int num_classes, float* dscores, float* dlocs, void* workspace, size_t workspace_bytes, void* stream) {
    return ssdk_multibox_loss_bwd_ex(params, scores, locs, anchors, target, sampled, grad_out, 0, batch, num_anchors, num_classes, dscores, dlocs,
                                     nullptr, workspace, workspace_bytes, stream);
}

extern "C" int ssdk_multibox_loss_bwd_ex(const ssdk_loss_params* params, const float* scores, const float* locs, const float* anchors,
                                         const float* target, const uint8_t* sampled, const float* grad_out, int grad_single, int batch, int num_anchors,
                                         int num_classes, float* dscores, float* dlocs, uint8_t* row_mask, void* workspace,
                                         size_t workspace_bytes, void* stream) {
    int rc = check_loss_common("ssdk_multibox_loss_bwd", scores, batch, num_anchors, num_classes, workspace, workspace_bytes);
    if (rc) return rc;
    rc = check_loss_params("ssdk_multibox_loss_bwd", params);
    if (rc) return rc;
    SSDK_REQUIRE(locs && anchors && target && sampled && grad_out && dscores && dlocs, SSDK_E_INVALID, "ssdk_multibox_loss_bwd: null pointer");
    SSDK_REQUIRE(((uintptr_t)dscores & 15) == 0 && ((uintptr_t)dlocs & 15) == 0 && ((uintptr_t)locs & 15) == 0 && ((uintptr_t)anchors & 15) == 0,
                 SSDK_E_INVALID, "ssdk_multibox_loss_bwd: dscores/dlocs/locs/anchors must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const long long n_rows = (long long)batch * num_anchors;
    LossWs w = carve_loss_ws(workspace, (size_t)n_rows, nullptr);
    LossParams p = to_device_params(params, num_classes, 1);
    const size_t lds = (size_t)kTileRows * num_classes * sizeof(float);
    if (lds > 48 * 1024)
        SSDK_CHECK_HIP(hipFuncSetAttribute((const void*)loss_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const bool focal = params->cls_kind == SSDK_CLS_SIGMOID_FOCAL || params->cls_kind == SSDK_CLS_SOFTMAX_FOCAL;
    hipLaunchKernelGGL(loss_bwd_kernel, dim3((unsigned)std::min<long long>((n_rows + kTileRows - 1) / kTileRows, 1 << 20)), dim3(kLossThreads), lds, s, p, focal ? params->reduce_mean : 0,
                       params->classification_weight, params->localization_weight, scores, (const float4*)locs, (const float4*)anchors,
                       num_anchors, target, sampled, grad_out, n_rows, w.lse, w.state, dscores, (float4*)dlocs, row_mask, grad_single ? 1 : 0);
    SSDK_CHECK_LAUNCH("loss_bwd_kernel");
    return SSDK_OK;
}

extern "C" int ssdk_encode_box(const float* boxes, const float* priors, float* out, int batch, int num_anchors, float xy_scale,
                               float wh_scale, float eps, int inplace_semantics, void* stream) {
    SSDK_REQUIRE(boxes && priors && out && batch > 0 && num_anchors > 0, SSDK_E_INVALID, "ssdk_encode_box: bad arguments");
    const long long n = (long long)batch * num_anchors;
    hipLaunchKernelGGL(encode_box_kernel, dim3(stream_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, (const float4*)boxes,
                       (const float4*)priors, (float4*)out, n, num_anchors, xy_scale, wh_scale, eps, inplace_semantics);
    SSDK_CHECK_LAUNCH("encode_box_kernel");
    return SSDK_OK;
}

extern "C" int ssdk_decode_box(const float* locs, const float* priors, float* out, int batch, int num_anchors, float xy_scale,
                               float wh_scale, int inplace_semantics, void* stream) {
    SSDK_REQUIRE(locs && priors && out && batch > 0 && num_anchors > 0, SSDK_E_INVALID, "ssdk_decode_box: bad arguments");
    const long long n = (long long)batch * num_anchors;
    hipLaunchKernelGGL(decode_box_kernel, dim3(stream_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, (const float4*)locs,
                       (const float4*)priors, (float4*)out, n, num_anchors, xy_scale, wh_scale, inplace_semantics);
    SSDK_CHECK_LAUNCH("decode_box_kernel");
    return SSDK_OK;
}
