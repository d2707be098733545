// boxes.hip -- bf/utils/box_utils.py:16-194 as a callable device surface: to_corners, to_centroids, area,
// intersection, iou, generalized_iou and the one-problem nms wrapper.  The hot path never materialises these -- the IoU lives inside
// assign_kernel / gt_argmax_kernel (match.hip), GIoU inside loss_fwd (loss.hip), NMS inside the postprocess kernels -- but
// matcher.match_per_prediction(weights, ...) takes an IoU matrix, and callers of the reference use the module directly.
// Built -ffp-contract=off: every expression rounds op for op like the reference's separate torch ops (bit-exact against its goldens).
#include "common.h"

namespace ssdk {

// box_utils.py:16-23  to_corners: [c - wh / 2, c + wh / 2]
__global__ void __launch_bounds__(256) box_to_corners_kernel(const float4* __restrict__ in, float4* __restrict__ out, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = to_corners(in[i]);
}
// box_utils.py:25-36  to_centroids.  Out of place: [(max + min) / 2, max - min]; in place: wh = max - min, then c = min + wh / 2 --
// two different roundings of the centre, both reproduced (inplace_form selects)
__global__ void __launch_bounds__(256) box_to_centroids_kernel(const float4* __restrict__ in, float4* __restrict__ out, long long n, int inplace_form) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 b = in[i];
    float4 r;
    if (inplace_form) {
        r.z = b.z - b.x; r.w = b.w - b.y;
        r.x = b.x + r.z / 2.0f; r.y = b.y + r.w / 2.0f;
    } else {
        r.x = (b.z + b.x) / 2.0f; r.y = (b.w + b.y) / 2.0f;
        r.z = b.z - b.x; r.w = b.w - b.y;
    }
    out[i] = r;
}
// box_utils.py:38-46  area: clamp(x2 - x1, 0) * clamp(y2 - y1, 0)
__global__ void __launch_bounds__(256) box_area_kernel(const float4* __restrict__ in, float* __restrict__ out, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { const float4 b = in[i]; out[i] = area4(b.x, b.y, b.z, b.w); }
}

// box_utils.py:49-80  intersection: cat([max(a_min, b_min), min(a_max, b_max)]); zero_incorrect: rows with any(max_ < min_) -> 0
__global__ void __launch_bounds__(256) box_intersection_kernel(const float4* __restrict__ a, int na, const float4* __restrict__ b, int nb, int cartesian,
                                                               int zero_incorrect, float4* __restrict__ out) {
    const long long total = cartesian ? (long long)na * nb : na;
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const float4 A = a[cartesian ? e / nb : e], B = b[cartesian ? e % nb : e];
    float4 r = make_float4(tmaxf(A.x, B.x), tmaxf(A.y, B.y), tminf(A.z, B.z), tminf(A.w, B.w));
    if (zero_incorrect && (r.z < r.x || r.w < r.y)) r = make_float4(0.f, 0.f, 0.f, 0.f);
    out[e] = r;
}

// box_utils.py:83-101 iou / :104-143 generalized_iou
__device__ __forceinline__ float giou_corner(float4 a, float4 b) {
    const float inter = area4(tmaxf(a.x, b.x), tmaxf(a.y, b.y), tminf(a.z, b.z), tminf(a.w, b.w));
    const float uni = area4(a.x, a.y, a.z, a.w) + area4(b.x, b.y, b.z, b.w) - inter;
    const float enc = area4(tminf(a.x, b.x), tminf(a.y, b.y), tmaxf(a.z, b.z), tmaxf(a.w, b.w));
    return inter / uni - (enc - uni) / enc;
}
__global__ void __launch_bounds__(256) box_iou_kernel(const float4* __restrict__ a, int na, const float4* __restrict__ b, int nb, int cartesian,
                                                      int generalized, float* __restrict__ out) {
    const long long total = cartesian ? (long long)na * nb : na;
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const float4 A = a[cartesian ? e / nb : e], B = b[cartesian ? e % nb : e];
    out[e] = generalized ? giou_corner(A, B) : iou_corner(A, area4(A.x, A.y, A.z, A.w), B, area4(B.x, B.y, B.z, B.w));
}

// ---- box_utils.py:166-194 nms for ONE problem (the batched, fused form is ssdk_postprocess) ------------------------------------------
// One workgroup.  (1) ranks by (score descending, index ascending); with a cap the first max_per_class of that order are the subset
// (:186-188 topk(sorted=False): the SET is the reference's, its order is left to the library there -- here: that order);
// (2) hard: torchvision.ops.nms's documented contract (stable descending sort, suppress IoU > thr; boxes unclamped like orc_nms_hard;
// PARITY UNPINNED, box_utils.py:193) -- picked = positions in the array nms() passed on (the subset, or the input itself);
// soft: _soft_nms (:145-163) on that array, incl. its loop condition mask.nonzero().sum() (the SUM OF INDICES of live boxes).
constexpr int kNmsThreads = 1024;
constexpr int kNmsMax = 1 << 16;
struct NmsWs { int* order; float* sc; };

__device__ __forceinline__ float nms_iou_unclamped(float4 a, float4 b) {   // (oracle: orc_nms_hard)
    const float area_a = (a.z - a.x) * (a.w - a.y), area_b = (b.z - b.x) * (b.w - b.y);
    const float iw = clamp0(tminf(a.z, b.z) - tmaxf(a.x, b.x)), ih = clamp0(tminf(a.w, b.w) - tmaxf(a.y, b.y));
    const float inter = iw * ih;
    return inter / (area_a + area_b - inter);
}

__global__ void __launch_bounds__(kNmsThreads) nms_one_kernel(const float4* __restrict__ boxes, const float* __restrict__ scores, int n, float overlap_thr,
                                                              float score_thr, int cap, int soft, float sigma, NmsWs w, long long* __restrict__ picked,
                                                              float4* __restrict__ picked_boxes, float* __restrict__ picked_scores, int* __restrict__ count) {
    __shared__ int s_count;
    __shared__ unsigned s_alive[kNmsMax / 32];   // hard: boxes not yet suppressed; soft: the mask of _soft_nms (bits in LDS: the atomics that clear them
                                                 // would otherwise run at the L2 while the next read may still hit this CU's L1)
    __shared__ long long s_red[kNmsThreads / 64];
    __shared__ unsigned long long s_best[kNmsThreads / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int K = (cap > 0 && cap < n) ? cap : n;
    const bool capped = K < n;
    // (1) order[p] = original index of the p-th element of the array nms() works on
    if (capped || !soft) {
        for (int i = tid; i < n; i += kNmsThreads) {
            const float si = scores[i];
            int rank = 0;
            for (int j = 0; j < n; ++j) {
                const float sj = scores[j];
                rank += (sj > si || (sj == si && j < i)) ? 1 : 0;
            }
            if (rank < n) w.order[rank] = i;   // (the full order; only the first K are used when capped)
        }
    } else {
        for (int i = tid; i < n; i += kNmsThreads) w.order[i] = i;
    }
    if (tid == 0) s_count = 0;
    __syncthreads();
    if (!soft) {
        for (int q = tid; q < (K + 31) / 32; q += kNmsThreads) s_alive[q] = 0xffffffffu;
        __syncthreads();
        for (int p = 0; p < K; ++p) {
            const bool live = (s_alive[p >> 5] >> (p & 31)) & 1u;   // (uniform: every thread reads the same word, behind the last barrier)
            if (!live) continue;
            const int ip = w.order[p];
            const float4 bp = boxes[ip];
            if (tid == 0) {
                const int c = s_count++;
                picked[c] = capped ? (long long)p : (long long)ip;
                if (picked_boxes) picked_boxes[c] = bp;
                if (picked_scores) picked_scores[c] = scores[ip];
            }
            for (int q = p + 1 + tid; q < K; q += kNmsThreads)
                if (((s_alive[q >> 5] >> (q & 31)) & 1u) && nms_iou_unclamped(bp, boxes[w.order[q]]) > overlap_thr) atomicAnd(&s_alive[q >> 5], ~(1u << (q & 31)));
            __syncthreads();
        }
    } else {
        for (int q = tid; q < K; q += kNmsThreads) w.sc[q] = scores[w.order[q]];
        __syncthreads();
        // mask of the iteration = (sc > score_thr) as it stood BEFORE the previous iteration's decay (:151 tests the mask of :156 / :147)
        for (int q = tid; q < (K + 31) / 32; q += kNmsThreads) {
            unsigned m = 0;
            for (int b = 0; b < 32 && q * 32 + b < K; ++b) m |= w.sc[q * 32 + b] > score_thr ? 1u << b : 0u;
            s_alive[q] = m;
        }
        __syncthreads();
        for (int it = 0; it < K + 1; ++it) {
            long long part = 0;
            for (int q = tid; q < K; q += kNmsThreads) part += ((s_alive[q >> 5] >> (q & 31)) & 1u) ? q : 0;
            part = wave_allreduce(part, [](long long a, long long b) { return a + b; });
            // argmax of sc over ALL entries, first index on ties (:152): key = (value bits as ordered uint, ~index)
            unsigned long long best = 0ull;
            for (int q = tid; q < K; q += kNmsThreads) {
                const float v = w.sc[q];
                unsigned u = __float_as_uint(v);
                u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);   // order-preserving map of the float
                if (v != v) u = 0xffffffffu;                      // torch.argmax ranks NaN above everything (first NaN wins)
                const unsigned long long key = ((unsigned long long)u << 32) | (unsigned)(0xffffffffu - (unsigned)q);
                best = key > best ? key : best;
            }
            best = wave_allreduce(best, OpMaxU64());
            if (lane == 0) { s_red[wave] = part; s_best[wave] = best; }
            __syncthreads();
            long long idxsum = 0;
            unsigned long long bk = 0ull;
            for (int v = 0; v < kNmsThreads / 64; ++v) { idxsum += s_red[v]; bk = s_best[v] > bk ? s_best[v] : bk; }
            __syncthreads();
            if (idxsum == 0) break;
            const int bq = (int)(0xffffffffu - (unsigned)(bk & 0xffffffffull));
            const int ib = w.order[bq];
            const float4 bb = boxes[ib];
            const float area_b = area4(bb.x, bb.y, bb.z, bb.w);
            if (tid == 0) {
                w.sc[bq] = 0.0f;
                const int c = s_count++;
                picked[c] = capped ? (long long)bq : (long long)ib;
                if (picked_boxes) picked_boxes[c] = bb;
                if (picked_scores) picked_scores[c] = scores[ib];
            }
            __syncthreads();
            for (int q = tid; q < (K + 31) / 32; q += kNmsThreads) {
                unsigned m = 0;
                for (int b = 0; b < 32 && q * 32 + b < K; ++b) {
                    const int e = q * 32 + b;
                    const float v = w.sc[e];
                    if (v > score_thr) {
                        m |= 1u << b;
                        const float4 x = boxes[w.order[e]];
                        const float inter = area4(tmaxf(bb.x, x.x), tmaxf(bb.y, x.y), tminf(bb.z, x.z), tminf(bb.w, x.w));
                        const float iou = inter / (area_b + area4(x.x, x.y, x.z, x.w) - inter);
                        w.sc[e] = v * expf(-(iou * iou / sigma));
                    }
                }
                s_alive[q] = m;
            }
            __syncthreads();
        }
    }
    __syncthreads();
    if (tid == 0) *count = s_count;
}

static NmsWs carve_nms(void* ws, int n, size_t* total) {
    Carver c(ws);
    NmsWs w;
    w.order = c.take<int>((size_t)n);
    w.sc = c.take<float>((size_t)n);
    if (total) *total = c.off;
    return w;
}

}  // namespace ssdk
using namespace ssdk;

static inline unsigned blocks_for(long long n) { return (unsigned)((n + 255) / 256); }

extern "C" int ssdk_box_to_corners(const float* box, float* out, long long n, void* stream) {
    SSDK_REQUIRE(n >= 0 && (n == 0 || (box && out)), SSDK_E_INVALID, "ssdk_box_to_corners: bad arguments");
    SSDK_REQUIRE((((uintptr_t)box | (uintptr_t)out) & 15) == 0, SSDK_E_UNSUPPORTED, "ssdk_box_to_corners: boxes must be 16-byte aligned");
    if (!n) return SSDK_OK;
    hipLaunchKernelGGL(box_to_corners_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, (const float4*)box, (float4*)out, n);
    SSDK_CHECK_LAUNCH("box_to_corners_kernel");
    return SSDK_OK;
}
extern "C" int ssdk_box_to_centroids(const float* box, float* out, long long n, int inplace_form, void* stream) {
    SSDK_REQUIRE(n >= 0 && (n == 0 || (box && out)), SSDK_E_INVALID, "ssdk_box_to_centroids: bad arguments");
    SSDK_REQUIRE((((uintptr_t)box | (uintptr_t)out) & 15) == 0, SSDK_E_UNSUPPORTED, "ssdk_box_to_centroids: boxes must be 16-byte aligned");
    if (!n) return SSDK_OK;
    hipLaunchKernelGGL(box_to_centroids_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, (const float4*)box, (float4*)out, n, inplace_form);
    SSDK_CHECK_LAUNCH("box_to_centroids_kernel");
    return SSDK_OK;
}
extern "C" int ssdk_box_area(const float* box, float* out, long long n, void* stream) {
    SSDK_REQUIRE(n >= 0 && (n == 0 || (box && out)), SSDK_E_INVALID, "ssdk_box_area: bad arguments");
    SSDK_REQUIRE(((uintptr_t)box & 15) == 0, SSDK_E_UNSUPPORTED, "ssdk_box_area: boxes must be 16-byte aligned");
    if (!n) return SSDK_OK;
    hipLaunchKernelGGL(box_area_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, (const float4*)box, out, n);
    SSDK_CHECK_LAUNCH("box_area_kernel");
    return SSDK_OK;
}
static int check_pairs(const char* fn, const float* a, int na, const float* b, int nb, int cartesian, const void* out) {
    SSDK_REQUIRE(na >= 0 && nb >= 0 && (cartesian || na == nb), SSDK_E_INVALID, "%s: na=%d nb=%d (pairwise needs equal counts, box_utils.py:70)", fn, na, nb);
    SSDK_REQUIRE((na == 0 || a) && (nb == 0 || b) && ((long long)na * nb == 0 || out), SSDK_E_INVALID, "%s: null pointer", fn);
    SSDK_REQUIRE((((uintptr_t)a | (uintptr_t)b) & 15) == 0, SSDK_E_UNSUPPORTED, "%s: boxes must be 16-byte aligned", fn);
    return SSDK_OK;
}
extern "C" int ssdk_box_intersection(const float* a, int na, const float* b, int nb, int cartesian, int zero_incorrect, float* out, void* stream) {
    int rc = check_pairs("ssdk_box_intersection", a, na, b, nb, cartesian, out);
    if (rc) return rc;
    SSDK_REQUIRE(((uintptr_t)out & 15) == 0, SSDK_E_UNSUPPORTED, "ssdk_box_intersection: out must be 16-byte aligned");
    const long long total = cartesian ? (long long)na * nb : na;
    if (!total) return SSDK_OK;
    hipLaunchKernelGGL(box_intersection_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, (const float4*)a, na, (const float4*)b, nb, cartesian,
                       zero_incorrect, (float4*)out);
    SSDK_CHECK_LAUNCH("box_intersection_kernel");
    return SSDK_OK;
}
extern "C" int ssdk_box_iou(const float* a, int na, const float* b, int nb, int cartesian, int generalized, float* out, void* stream) {
    int rc = check_pairs("ssdk_box_iou", a, na, b, nb, cartesian, out);
    if (rc) return rc;
    const long long total = cartesian ? (long long)na * nb : na;
    if (!total) return SSDK_OK;
    hipLaunchKernelGGL(box_iou_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, (const float4*)a, na, (const float4*)b, nb, cartesian, generalized,
                       out);
    SSDK_CHECK_LAUNCH("box_iou_kernel");
    return SSDK_OK;
}
extern "C" size_t ssdk_nms_workspace_bytes(int n) {
    size_t total = 0;
    if (n > 0) carve_nms(nullptr, n, &total);
    return total;
}
extern "C" int ssdk_nms(const float* boxes, const float* scores, int n, float overlap_threshold, float score_threshold, int max_per_class, int soft,
                        float sigma, long long* picked, float* picked_boxes, float* picked_scores, int* count, void* workspace, size_t workspace_bytes,
                        void* stream) {
    SSDK_REQUIRE(n >= 0 && n <= kNmsMax, SSDK_E_UNSUPPORTED, "ssdk_nms: n=%d (0..%d boxes per call; the batched path is ssdk_postprocess)", n, kNmsMax);
    SSDK_REQUIRE(count, SSDK_E_INVALID, "ssdk_nms: null count");
    if (n == 0) return (int)zero_async(count, sizeof(int), (hipStream_t)stream);
    SSDK_REQUIRE(boxes && scores && picked, SSDK_E_INVALID, "ssdk_nms: null pointer");
    SSDK_REQUIRE((((uintptr_t)boxes | (uintptr_t)picked_boxes) & 15) == 0, SSDK_E_UNSUPPORTED, "ssdk_nms: boxes must be 16-byte aligned");
    SSDK_REQUIRE(workspace && workspace_bytes >= ssdk_nms_workspace_bytes(n), SSDK_E_WORKSPACE, "ssdk_nms: workspace too small");
    const NmsWs w = carve_nms(workspace, n, nullptr);
    hipLaunchKernelGGL(nms_one_kernel, dim3(1), dim3(kNmsThreads), 0, (hipStream_t)stream, (const float4*)boxes, scores, n, overlap_threshold, score_threshold,
                       max_per_class, soft, sigma, w, picked, (float4*)picked_boxes, picked_scores, count);
    SSDK_CHECK_LAUNCH("nms_one_kernel");
    return SSDK_OK;
}
