// match.hip -- IoU + matcher + target encoding, fused.
//
// Reference: detection/target_assigner.py:22-63 (python loop over images; per image a materialised [G,A] IoU
// matrix from bf/utils/box_utils.py:83-101 with ~10 [G,A,*] temporaries, then detection/matcher.py:33-56), all on
// the CPU because anchors live there -- 331 ms per SSD-300 batch of 32.
//
// Here nothing [G,A]-shaped exists.  Two launches over the whole batch:
//   1. gt_argmax_kernel   one workgroup per ground-truth box: it sweeps the anchors (16-byte coalesced loads,
//                         L2-resident: the anchor table is shared by every image) and reduces the box's best anchor as
//                         a 64-bit key (iou bits << 32 | ~anchor): argmax(dim=1) with torch's first-index tie rule
//                         (matcher.py:52).
//   2. assign_kernel      one thread per (image, anchor): re-derives the anchor's IoU with the image's boxes from
//                         LDS (cheaper than storing [G,A]), applies max(dim=0) with first-max ties, the two
//                         thresholds, then the force-match (last writer = highest box index wins, matcher.py:53-54),
//                         and writes the 24-byte target row through an LDS tile so HBM sees whole 8-byte lanes of a
//                         contiguous 6 KB block.  The target is written exactly once: B*A*24 bytes, the
//                         algorithmic minimum for this path.
// IoU is computed op for op like the reference (this TU is built -ffp-contract=off, IEEE divide) so that
// assignments are bit-exact.
#include "common.h"

namespace ssdk {

constexpr int kAssignThreads = 256;
constexpr int kGtChunk = 128;
constexpr int kSegAnchors = 512;  // anchors swept by one wave of gt_argmax_kernel

// One workgroup per ground-truth box: its 1 024 threads sweep ALL anchors and the result is stored, not accumulated -- no atomics and
// therefore no zero-fill launch in front (the first version folded (box, 512-anchor segment) partial results into a zeroed array with
// atomicMax: a third launch, and a memset node costs as much as a small kernel).
constexpr int kArgmaxThreads = 1024;   // (a box's sweep is a chain of dependent L2 round trips per thread: 8 anchors per thread at A = 8 108, not 32)
// Round 5: large anchor sets (RetinaNet: 47 961 -- 47 anchors per thread, 21 us for 144 boxes on 144 CUs) are swept by `segs` workgroups
// per box, each STORING the key of its anchor range (gt_best[g * segs + seg]); assign_kernel takes the maximum of a box's keys when it
// loads the box -- still no atomics and nothing to zero, and the same key as one sweep would produce (largest IoU, then smallest anchor).
constexpr int kArgmaxMaxSegs = 8;
static inline int argmax_segs(int A) { const int s = (A + kArgmaxThreads * 12 - 1) / (kArgmaxThreads * 12); return s < 1 ? 1 : (s > kArgmaxMaxSegs ? kArgmaxMaxSegs : s); }
__global__ void __launch_bounds__(kArgmaxThreads) gt_argmax_kernel(const float* __restrict__ gt_rows, int gt_stride,
                                                                   const float4* __restrict__ anchors, int A_all,
                                                                   unsigned long long* __restrict__ gt_best, const int32_t* __restrict__ gt_off,
                                                                   int batch) {
    __shared__ unsigned long long s_key[kArgmaxThreads / kWave];
    const int g = blockIdx.x;
    if (g >= gt_off[batch]) return;   // (a row buffer of fixed capacity, e.g. inside a captured HIP graph: rows past the last image's are padding)
    const int segs = gridDim.y, seg = blockIdx.y;
    const int seg_len = (A_all + segs - 1) / segs;
    const int a_begin = seg * seg_len, A = min(A_all, a_begin + seg_len);   // this workgroup's anchors: [a_begin, A)
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    const float* r = gt_rows + (size_t)g * gt_stride;
    const float4 gb = make_float4(r[0], r[1], r[2], r[3]);
    const float garea = area4(gb.x, gb.y, gb.z, gb.w);
    float best = 0.0f;
    int bi = -1;
    auto take = [&](int a, const float4& p) {
        const float4 c = to_corners(p);
        const float v = iou_corner(gb, garea, c, area4(c.x, c.y, c.z, c.w));
        if (bi < 0 || v > best || (v != v && best == best)) { best = v; bi = a; }
    };
    int a = a_begin + threadIdx.x;
    for (; a + 3 * kArgmaxThreads < A; a += 4 * kArgmaxThreads) {   // four loads in flight, taken in ascending anchor order (first-index ties)
        const float4 p0 = anchors[a], p1 = anchors[a + kArgmaxThreads], p2 = anchors[a + 2 * kArgmaxThreads], p3 = anchors[a + 3 * kArgmaxThreads];
        take(a, p0);
        take(a + kArgmaxThreads, p1);
        take(a + 2 * kArgmaxThreads, p2);
        take(a + 3 * kArgmaxThreads, p3);
    }
    for (; a < A; a += kArgmaxThreads) take(a, anchors[a]);
    unsigned long long key = 0ull;
    if (bi >= 0) key = ((unsigned long long)__float_as_uint(best) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)bi);
    key = wave_allreduce(key, OpMaxU64());
    if (lane == 0) s_key[wave] = key;
    __syncthreads();
    if (threadIdx.x < kWave) {
        unsigned long long k = threadIdx.x < kArgmaxThreads / kWave ? s_key[threadIdx.x] : 0ull;
        k = wave_allreduce(k, OpMaxU64());
        if (threadIdx.x == 0) gt_best[(size_t)g * segs + seg] = k;
    }
}

__global__ void __launch_bounds__(kAssignThreads) assign_kernel(const float* __restrict__ gt_rows, int gt_stride,
                                                                const int32_t* __restrict__ gt_off,
                                                                const float4* __restrict__ anchors, int A, float matched_thr,
                                                                float unmatched_thr,
                                                                const unsigned long long* __restrict__ gt_best, int segs,
                                                                float* __restrict__ target, int32_t* __restrict__ box_idx) {
    __shared__ float4 s_box[kGtChunk];
    __shared__ float s_area[kGtChunk];
    __shared__ int s_best_anchor[kGtChunk];
    __shared__ __attribute__((aligned(16))) float s_out[kAssignThreads * 6];

    const int i = blockIdx.y;
    const int a0 = blockIdx.x * kAssignThreads;
    const int a = a0 + threadIdx.x;
    const int g0 = gt_off[i], G = gt_off[i + 1] - g0;

    float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
    float carea = 0.f;
    if (a < A) {
        c = to_corners(anchors[a]);
        carea = area4(c.x, c.y, c.z, c.w);
    }
    float best = 0.0f;
    int bi = SSDK_NOT_MATCHED, forced = -1;
    for (int base = 0; base < G; base += kGtChunk) {
        const int n = min(kGtChunk, G - base);
        __syncthreads();
        if (threadIdx.x < n) {
            const int g = g0 + base + threadIdx.x;
            const float* r = gt_rows + (size_t)g * gt_stride;
            const float4 gb = make_float4(r[0], r[1], r[2], r[3]);
            s_box[threadIdx.x] = gb;
            s_area[threadIdx.x] = area4(gb.x, gb.y, gb.z, gb.w);
            unsigned long long key = gt_best[(size_t)g * segs];
            for (int q = 1; q < segs; ++q) { const unsigned long long k2 = gt_best[(size_t)g * segs + q]; key = k2 > key ? k2 : key; }
            s_best_anchor[threadIdx.x] = (int)(0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull));
        }
        __syncthreads();
        for (int k = 0; k < n; ++k) {
            const float v = iou_corner(s_box[k], s_area[k], c, carea);
            if ((base + k) == 0 || v > best || (v != v && best == best)) { best = v; bi = base + k; }  // max(dim=0)
            if (s_best_anchor[k] == a) forced = base + k;  // ascending k: the highest box index wins
        }
    }
    if (G > 0) {  // matcher.py:45-50
        if (best < unmatched_thr) bi = SSDK_NOT_MATCHED;
        else if (best < matched_thr) bi = SSDK_IGNORE;
        if (forced >= 0) bi = forced;
    }
    float* o = s_out + threadIdx.x * 6;
    if (bi >= 0) {  // target_assigner.py:52-54
        const float* r = gt_rows + (size_t)(g0 + bi) * gt_stride;
        o[0] = r[0]; o[1] = r[1]; o[2] = r[2]; o[3] = r[3]; o[4] = r[4]; o[5] = r[5];
    } else {
        o[0] = 0.f; o[1] = 0.f; o[2] = 0.f; o[3] = 0.f;
        o[4] = bi == SSDK_IGNORE ? -1.0f : 0.0f;  // :56-58 / :39
        o[5] = bi == SSDK_IGNORE ? -1.0f : 1.0f;  // :40
    }
    if (box_idx && a < A) box_idx[(size_t)i * A + a] = bi;
    __syncthreads();
    // 24-byte rows of this block are one contiguous run in HBM; (i*A + a0)*24 is always 8-byte aligned.
    const int rows = min(kAssignThreads, A - a0);
    float2* dst = reinterpret_cast<float2*>(target + ((size_t)i * A + a0) * 6);
    const float2* src = reinterpret_cast<const float2*>(s_out);
    for (int t = threadIdx.x; t < rows * 3; t += kAssignThreads) dst[t] = src[t];
}

// ---- detection/matcher.py:33-56 on a materialised weight matrix [G, A] (the API the reference exposes; the training path uses the
// fused kernels above and never stores such a matrix) ------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) mpp_row_argmax_kernel(const float* __restrict__ w, int A, unsigned long long* __restrict__ best) {
    const int g = blockIdx.y;
    const int seg = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = lane_id();
    const int a0 = seg * kSegAnchors;
    if (a0 >= A) return;
    const int a1 = min(A, a0 + kSegAnchors);
    const float* row = w + (size_t)g * A;
    float bv = 0.0f;
    int bi = -1;
    for (int a = a0 + lane; a < a1; a += kWave) {   // weights.argmax(dim=1): first maximum, NaN counts as the largest (matcher.py:52)
        const float v = row[a];
        if (bi < 0 || v > bv || (v != v && bv == bv)) { bv = v; bi = a; }
    }
    unsigned long long key = 0ull;
    if (bi >= 0) {
        // order-preserving bits for any sign: flip all bits of negatives, set the sign bit of the others; NaN above everything
        unsigned u = __float_as_uint(bv);
        u = (bv != bv) ? 0xFFFFFFFFu : ((u & 0x80000000u) ? ~u : (u | 0x80000000u));
        key = ((unsigned long long)u << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)bi);
    }
    key = wave_allreduce(key, OpMaxU64());
    if (lane == 0 && key) atomicMax(best + g, key);
}

__global__ void __launch_bounds__(kAssignThreads) mpp_assign_kernel(const float* __restrict__ w, int G, int A, float matched_thr, float unmatched_thr,
                                                                    int force, const unsigned long long* __restrict__ best,
                                                                    long long* __restrict__ box_idx) {
    __shared__ int s_best_anchor[kGtChunk];
    const int a = blockIdx.x * kAssignThreads + threadIdx.x;
    float bv = 0.0f;
    int bi = 0, forced = -1;
    for (int base = 0; base < G; base += kGtChunk) {
        const int n = min(kGtChunk, G - base);
        __syncthreads();
        if (threadIdx.x < n) s_best_anchor[threadIdx.x] = force ? (int)(0xFFFFFFFFu - (unsigned)(best[base + threadIdx.x] & 0xFFFFFFFFull)) : -1;
        __syncthreads();
        if (a < A)
            for (int k = 0; k < n; ++k) {
                const float v = w[(size_t)(base + k) * A + a];
                if ((base + k) == 0 || v > bv || (v != v && bv == bv)) { bv = v; bi = base + k; }   // weights.max(dim=0), :45
                if (s_best_anchor[k] == a) forced = base + k;                                            // :52-54, the highest box index wins
            }
    }
    if (a >= A) return;
    if (bv < unmatched_thr) bi = SSDK_NOT_MATCHED;        // :49
    else if (bv < matched_thr) bi = SSDK_IGNORE;          // :50
    if (forced >= 0) bi = forced;
    box_idx[a] = bi;
}

}  // namespace ssdk

using namespace ssdk;

extern "C" size_t ssdk_match_per_prediction_workspace_bytes(int num_boxes) {
    Carver c(nullptr);
    c.take<unsigned long long>((size_t)(num_boxes > 0 ? num_boxes : 1));
    return c.off;
}

extern "C" int ssdk_match_per_prediction(const float* weights, int num_boxes, int num_anchors, float matched_threshold,
                                         float unmatched_threshold, int force_match_for_each_target, int64_t* box_idx, void* workspace,
                                         size_t workspace_bytes, void* stream) {
    SSDK_REQUIRE(weights && box_idx && num_boxes > 0 && num_anchors > 0, SSDK_E_INVALID,
                 "ssdk_match_per_prediction: boxes=%d anchors=%d (weights.max(dim=0) of an empty matrix is an error in the reference too)",
                 num_boxes, num_anchors);
    SSDK_REQUIRE(num_boxes <= 65535, SSDK_E_INVALID, "ssdk_match_per_prediction: more than 65535 boxes");
    SSDK_REQUIRE(matched_threshold >= unmatched_threshold, SSDK_E_INVALID,
                 "ssdk_match_per_prediction: matched_threshold < unmatched_threshold (matcher.py:43)");
    SSDK_REQUIRE(workspace && workspace_bytes >= ssdk_match_per_prediction_workspace_bytes(num_boxes), SSDK_E_WORKSPACE,
                 "ssdk_match_per_prediction: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    unsigned long long* best = (unsigned long long*)workspace;
    if (force_match_for_each_target) {
        SSDK_CHECK_HIP(zero_async(best, sizeof(unsigned long long) * (size_t)num_boxes, s));
        hipLaunchKernelGGL(mpp_row_argmax_kernel, dim3(cdiv(num_anchors, kSegAnchors * 4), num_boxes), dim3(256), 0, s, weights, num_anchors, best);
        SSDK_CHECK_LAUNCH("mpp_row_argmax_kernel");
    }
    hipLaunchKernelGGL(mpp_assign_kernel, dim3(cdiv(num_anchors, kAssignThreads)), dim3(kAssignThreads), 0, s, weights, num_boxes, num_anchors,
                       matched_threshold, unmatched_threshold, force_match_for_each_target, best, (long long*)box_idx);
    SSDK_CHECK_LAUNCH("mpp_assign_kernel");
    return SSDK_OK;
}

extern "C" size_t ssdk_encode_ground_truth_workspace_bytes(int batch, int total_gt) {
    (void)batch;
    Carver c(nullptr);
    c.take<unsigned long long>((size_t)(total_gt > 0 ? total_gt : 1) * kArgmaxMaxSegs);
    return c.off;
}

extern "C" int ssdk_encode_ground_truth(const float* gt_rows, int gt_stride, const int32_t* gt_offsets, int batch,
                                        int total_gt, const float* anchors, int num_anchors, float matched_threshold,
                                        float unmatched_threshold, float* target, int32_t* box_idx, void* workspace,
                                        size_t workspace_bytes, void* stream) {
    SSDK_REQUIRE(batch > 0 && num_anchors > 0 && total_gt >= 0, SSDK_E_INVALID,
                 "ssdk_encode_ground_truth: batch=%d anchors=%d total_gt=%d", batch, num_anchors, total_gt);
    SSDK_REQUIRE(batch <= 65535 && total_gt <= 65535, SSDK_E_INVALID, "ssdk_encode_ground_truth: batch/total_gt exceed grid limits");
    SSDK_REQUIRE(gt_offsets && anchors && target && (gt_rows || total_gt == 0), SSDK_E_INVALID, "ssdk_encode_ground_truth: null pointer");
    SSDK_REQUIRE(gt_stride >= 6, SSDK_E_INVALID, "ssdk_encode_ground_truth: gt_stride=%d < 6", gt_stride);
    SSDK_REQUIRE(matched_threshold >= unmatched_threshold, SSDK_E_INVALID,
                 "ssdk_encode_ground_truth: matched_threshold < unmatched_threshold (matcher.py:43)");
    SSDK_REQUIRE(workspace && workspace_bytes >= ssdk_encode_ground_truth_workspace_bytes(batch, total_gt), SSDK_E_WORKSPACE,
                 "ssdk_encode_ground_truth: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    Carver c(workspace);
    unsigned long long* gt_best = c.take<unsigned long long>((size_t)(total_gt > 0 ? total_gt : 1) * kArgmaxMaxSegs);
    const int segs = argmax_segs(num_anchors);
    if (total_gt > 0) {
        hipLaunchKernelGGL(gt_argmax_kernel, dim3(total_gt, segs), dim3(kArgmaxThreads), 0, s, gt_rows, gt_stride, (const float4*)anchors, num_anchors, gt_best, gt_offsets, batch);
        SSDK_CHECK_LAUNCH("gt_argmax_kernel");
    }
    dim3 grid(cdiv(num_anchors, kAssignThreads), batch);
    hipLaunchKernelGGL(assign_kernel, grid, dim3(kAssignThreads), 0, s, gt_rows, gt_stride, gt_offsets, (const float4*)anchors,
                       num_anchors, matched_threshold, unmatched_threshold, gt_best, segs, target, box_idx);
    SSDK_CHECK_LAUNCH("assign_kernel");
    return SSDK_OK;
}
