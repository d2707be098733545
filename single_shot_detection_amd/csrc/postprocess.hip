// postprocess.hip -- score conversion + decode + per-class top-k + NMS + per-image top-k.
//
// Reference: detection/postprocessor.py:24-78 and bf/utils/box_utils.py:166-194.  The reference softmaxes all of
// [B,A,C], decodes all B*A boxes, moves everything to the CPU (its anchors live there) and runs a python double loop
// over (image, class) -- 2 560 iterations of mask + gather + topk + torchvision.ops.nms for a batch of 32.
//
// Here three launches cover the whole batch:
//   post_select_kernel  the one pass over the logits (64-anchor x C tiles through LDS, 16-byte coalesced loads):
//                       softmax/sigmoid, score threshold, and per-(image,class) candidate lists of 64-bit keys
//                       (score bits << 32 | ~anchor), compacted per tile in LDS so that HBM sees one atomic and one
//                       contiguous burst per (tile, class).
//   post_nms_kernel     one workgroup per (image, class): exact top-max_per_class by radix narrowing + an LDS bitonic
//                       sort of the survivors (keys are distinct, so order is total: score descending, then anchor
//                       ascending), decode of ONLY those <= 256 boxes, an IoU bit-matrix in LDS and a greedy sweep
//                       done by one wave with v_readlane (no LDS round trip on the serial chain).
//   post_merge_kernel   one workgroup per image: concatenation in class order, or top-max_total over all kept boxes
//                       (score descending; ties: lower class, then higher rank within the class first).
// Hard NMS follows torchvision.ops.nms's documented contract (bf/utils/box_utils.py:193 delegates to it):
// IoU = inter / (a + b - inter) with areas (x2-x1)*(y2-y1), suppress when IoU > threshold.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <type_traits>
#include <mutex>

#include "common.h"

namespace ssdk {

typedef unsigned long long u64;

constexpr int kPostTileRows = 64;
constexpr int kPostThreads = 256;
constexpr int kSortCap = 1024;      // survivors of the radix narrowing that are sorted in LDS
constexpr int kMaxPerClass = 256;   // NMS bit-matrix is kMaxPerClass x kMaxPerClass bits

struct PostWs {
    u64* cand;        // [B*ncls][A]   candidate keys
    int* cand_count;  // [B*ncls]
    float* pc_rows;   // [B*ncls][K][6] kept detections per (image, class), NMS order
    int* pc_count;    // [B*ncls]
    u64* merge_keys;  // [B][ncls*K]
    float* pc_score;  // [B*ncls][K] scores of pc_rows, contiguous (the merge reads only these)
    unsigned* tau;    // [B*ncls] per-(image, class) lower bound of the K-th score (score bits; 0 = none)
};

static PostWs carve_post_ws(void* ws, size_t B, size_t A, size_t ncls, size_t K, size_t* total) {
    Carver c(ws);
    PostWs w;
    w.cand = c.take<u64>(B * ncls * A);
    w.cand_count = c.take<int>(B * ncls);
    w.pc_rows = c.take<float>(B * ncls * K * 6);
    w.pc_count = c.take<int>(B * ncls);
    w.merge_keys = c.take<u64>(B * ncls * K);
    w.pc_score = c.take<float>(B * ncls * K);
    w.tau = c.take<unsigned>(B * ncls);
    if (total) *total = c.off;
    return w;
}

__device__ __forceinline__ float quad_max(float v) {
    v = fmaxf(v, __shfl_xor(v, 1, kWave));
    return fmaxf(v, __shfl_xor(v, 2, kWave));
}
__device__ __forceinline__ float quad_sum(float v) {
    v += __shfl_xor(v, 1, kWave);
    return v + __shfl_xor(v, 2, kWave);
}

// ---- P1: scores -> per-(image,class) candidate lists ----------------------------------------------------------
// LDS: tile [64*C] floats | lists [ncls][64] u64 | cnt [ncls] | base [ncls]
__global__ void __launch_bounds__(kPostThreads) post_select_kernel(const float* __restrict__ scores, int A, int C, int softmax,
                                                                   float score_thr, int tiles_per_image,
                                                                   u64* __restrict__ cand, int* __restrict__ cand_count) {
    // LDS = the 64 x C tile only (20 KB at C = 81 -> 7 workgroups per CU): the probabilities overwrite the logits in place, then
    // one wave per class reads its column (stride C: conflict-free for odd C, 2-way for C = 80), ballots p > threshold, reserves
    // the hits with one global atomic and stores them in row order.  (The first version kept per-class candidate lists of
    // worst-case size in LDS -- 41 KB more, 2 workgroups per CU -- and ran at 0.9 TB/s on the logits.)
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    const int ncls = softmax ? C - 1 : C;
    float* s_tile = reinterpret_cast<float*>(s_raw);
    int* s_cnt = reinterpret_cast<int*>(s_raw + align_up((size_t)kPostTileRows * C * 4, 16));

    const int i = blockIdx.y;
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    for (int tile = blockIdx.x; tile < tiles_per_image; tile += gridDim.x) {
        const int a0 = tile * kPostTileRows;
        const int rows = min(kPostTileRows, A - a0);
        const int nfloat = rows * C;
        const float* src = scores + ((size_t)i * A + a0) * C;
        __syncthreads();
        if ((reinterpret_cast<uintptr_t>(src) & 15) == 0) {
            if (nfloat >> 2) stage_tile_f4(reinterpret_cast<float4*>(s_tile), reinterpret_cast<const float4*>(src), nfloat >> 2);
            for (int t = (nfloat & ~3) + threadIdx.x; t < nfloat; t += kPostThreads) s_tile[t] = src[t];
        } else {
            for (int t = threadIdx.x; t < nfloat; t += kPostThreads) s_tile[t] = src[t];
        }
        __syncthreads();
        const int row = threadIdx.x >> 2, q = threadIdx.x & 3;
        float* x = s_tile + row * C;
        const bool live = row < rows;
        if (softmax) {  // postprocessor.py:43 F.softmax(dim=-1), :46-48 drop column 0
            float m = -INFINITY;
            if (live)
                for (int c = q; c < C; c += 4) m = fmaxf(m, x[c]);
            m = quad_max(m);
            float sum = 0.0f;
            if (live)
                for (int c = q; c < C; c += 4) {   // exp(x - max) is computed ONCE and kept in the tile (the kernel is bound by
                    const float e = expf(x[c] - m);   // its exp / divide arithmetic, not by the 20 KB of logits per tile)
                    x[c] = e;
                    sum += e;
                }
            sum = quad_sum(sum);
            if (live)
                for (int c = q; c < C; c += 4) x[c] = x[c] / sum;   // (dividing only the entries near the threshold, in the per-class
                                                                    //  pass, measured slower: 83 -> 93 us)
        } else if (live) {
            for (int c = q; c < C; c += 4) x[c] = 1.0f / (1.0f + expf(-x[c]));
        }
        __syncthreads();
        const unsigned akey = 0xFFFFFFFFu - (unsigned)(a0 + lane);
        const int c_off = softmax ? 1 : 0;
        // hits per class (one wave per class), then ALL the reservations of the tile in flight at once (a wave that reserves
        // class after class waits for one returning atomic per class: 20 dependent round trips), then the stores
        for (int c = wave; c < ncls; c += kPostThreads / kWave) {
            const float p = lane < rows ? s_tile[lane * C + c + c_off] : 0.0f;
            const unsigned long long mask = __ballot(lane < rows && p > score_thr);   // :63
            if (lane == 0) s_cnt[c] = __popcll(mask);
        }
        __syncthreads();
        for (int c = threadIdx.x; c < ncls; c += kPostThreads) {
            const int n = s_cnt[c];
            s_cnt[c] = n ? atomicAdd(&cand_count[i * ncls + c], n) : -1;
        }
        __syncthreads();
        for (int c = wave; c < ncls; c += kPostThreads / kWave) {
            const int base = s_cnt[c];
            if (base < 0) continue;   // (uniform)
            const float p = lane < rows ? s_tile[lane * C + c + c_off] : 0.0f;
            const bool hit = lane < rows && p > score_thr;
            const unsigned long long mask = __ballot(hit);
            if (hit) cand[((size_t)i * ncls + c) * A + base + __popcll(mask & ((1ull << lane) - 1ull))] = ((u64)__float_as_uint(p) << 32) | akey;
        }
    }
}

// descending bitonic sort of s_keys[0..m) (LDS); slots m..P-1 are filled with 0 and end up last.  Whole workgroup.
__device__ void bitonic_sort_desc(u64* s_keys, int m) {
    const int tid = threadIdx.x, nthr = blockDim.x;
    int P = 1;
    while (P < m) P <<= 1;
    for (int k = m + tid; k < P; k += nthr) s_keys[k] = 0;
    __syncthreads();
    for (int size = 2; size <= P; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = tid; t < (P >> 1); t += nthr) {
                const int lo = 2 * t - (t & (stride - 1));
                const int hi = lo + stride;
                const bool desc = (lo & size) == 0;
                const u64 a = s_keys[lo], b = s_keys[hi];
                if ((a < b) == desc) { s_keys[lo] = b; s_keys[hi] = a; }
            }
            __syncthreads();
        }
    }
}

// ---- exact top-K of distinct non-zero 64-bit keys, result sorted descending in LDS ---------------------------
// keys: global, n of them.  s_keys: LDS u64[kSortCap].  s_hist: LDS unsigned[256].  s_misc: LDS u64[4].
// Returns min(n, K); s_keys[0 .. result) holds the K largest keys in descending order.
__device__ int block_topk_sorted(const u64* __restrict__ keys, int n, int K, u64* s_keys, unsigned* s_hist, u64* s_misc) {
    const int tid = threadIdx.x, nthr = blockDim.x;
    u64 lower = 0;  // collect every key >= lower
    if (n > kSortCap) {
        u64 prefix = 0;
        unsigned above = 0;  // keys strictly above the current prefix range
        for (int shift = 56; shift >= 0; shift -= 8) {
            for (int b = tid; b < 256; b += nthr) s_hist[b] = 0;
            __syncthreads();
            for (int k0 = tid; k0 < n; k0 += 4 * nthr) {   // four keys per trip, loads first (branch-free: clamped index)
                u64 key[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) key[u] = keys[min(k0 + u * nthr, n - 1)];
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (k0 + u * nthr < n && (shift == 56 || (key[u] >> (shift + 8)) == (prefix >> (shift + 8))))
                        atomicAdd(&s_hist[(unsigned)(key[u] >> shift) & 255u], 1u);
            }
            __syncthreads();
            if (tid < kWave) {
                // wave 0 finds the digit (lane l owns bins 4l .. 4l+3, suffix sums over the lanes): the largest b with
                // above + (keys in bins > b) + hist[b] >= K.  One thread walking the bins was 255 dependent LDS reads per pass.
                const int l = tid;
                const unsigned h0 = s_hist[4 * l], h1 = s_hist[4 * l + 1], h2 = s_hist[4 * l + 2], h3 = s_hist[4 * l + 3];
                const unsigned mine = h0 + h1 + h2 + h3;
                unsigned incl = mine;
#pragma unroll
                for (int d = 1; d < kWave; d <<= 1) {
                    const unsigned t = __shfl_down(incl, d, kWave);
                    if (l + d < kWave) incl += t;
                }
                unsigned cum = above + (incl - mine);
                int found = -1;
                unsigned cum_at = 0, h_at = 0;
                const unsigned hq[4] = {h0, h1, h2, h3};
#pragma unroll
                for (int q = 3; q >= 0; --q) {
                    if (found < 0) {
                        if (cum + hq[q] >= (unsigned)K) { found = 4 * l + q; cum_at = cum; h_at = hq[q]; }
                        else cum += hq[q];
                    }
                }
                const unsigned long long hit = __ballot(found >= 0);
                if (hit) {
                    const int src = 63 - __clzll((long long)hit);
                    if (l == src) {
                        s_misc[0] = prefix | ((u64)found << shift);
                        s_misc[1] = cum_at;             // new `above`
                        s_misc[2] = cum_at + h_at;      // keys >= new prefix
                    }
                } else if (l == 0) {   // fewer than K keys in range: bin 0, like the sequential walk
                    const unsigned c0 = above + (incl - h0);
                    s_misc[0] = prefix;
                    s_misc[1] = c0;
                    s_misc[2] = c0 + h0;
                }
            }
            __syncthreads();
            prefix = s_misc[0];
            above = (unsigned)s_misc[1];
            const unsigned count_ge = (unsigned)s_misc[2];
            __syncthreads();
            if (count_ge <= (unsigned)kSortCap) break;
        }
        lower = prefix;
    }
    // collect
    int* s_n = reinterpret_cast<int*>(&s_misc[3]);
    if (tid == 0) *s_n = 0;
    __syncthreads();
    for (int k0 = tid; k0 < n; k0 += 4 * nthr) {   // (four loads in flight per thread; the sort below makes slot order irrelevant)
        u64 key[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) key[u] = keys[min(k0 + u * nthr, n - 1)];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (k0 + u * nthr < n && key[u] >= lower) {
                const int slot = atomicAdd(s_n, 1);
                if (slot < kSortCap) s_keys[slot] = key[u];
            }
    }
    __syncthreads();
    const int m = min(*s_n, kSortCap);
    bitonic_sort_desc(s_keys, m);
    return min(m, K);
}

__device__ __forceinline__ u64 readlane_u64(u64 v, int lane) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, lane);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), lane);
    return ((u64)hi << 32) | lo;
}

// ---- P2: per-(image,class) top-k + NMS -----------------------------------------------------------------------
__global__ void __launch_bounds__(kPostThreads) post_nms_kernel(const float4* __restrict__ locs, const float4* __restrict__ priors,
                                                                int A, int ncls, int K, float nms_thr, int soft, float sigma, float score_thr,
                                                                float xy_scale, float wh_scale,
                                                                const u64* __restrict__ cand, const int* __restrict__ cand_count,
                                                                float* __restrict__ pc_rows, float* __restrict__ pc_score,
                                                                int* __restrict__ pc_count, u64* __restrict__ nms_candidates) {
    __shared__ u64 s_keys[kSortCap];
    __shared__ unsigned s_hist[256];
    __shared__ u64 s_misc[4];
    __shared__ float4 s_box[kMaxPerClass];
    __shared__ float s_area[kMaxPerClass];
    __shared__ u64 s_mask[kMaxPerClass][kMaxPerClass / 64];
    __shared__ u64 s_keep[kMaxPerClass / 64];

    const int pc = blockIdx.x;  // image * ncls + class
    const int i = pc / ncls, c = pc % ncls;
    const int n = min(cand_count[pc], A);
    if (n == 0) {
        if (threadIdx.x == 0) pc_count[pc] = 0;
        return;
    }
    const int m = block_topk_sorted(cand + (size_t)pc * A, n, K, s_keys, s_hist, s_misc);  // box_utils.py:186-188
    const int W = (m + 63) >> 6;
    const int tid = threadIdx.x;
    if (soft && n <= K) {
        // no top-k happened: _soft_nms sees the candidates in boxes[mask] order = ascending anchor (its argmax tie rule and
        // its "sum of live indices" loop test depend on positions) -> re-sort by anchor
        if (tid < m) { const u64 k = s_keys[tid]; s_keys[tid] = (k << 32) | (k >> 32); }
        __syncthreads();
        bitonic_sort_desc(s_keys, m);   // descending ~anchor = ascending anchor
        if (tid < m) { const u64 k = s_keys[tid]; s_keys[tid] = (k << 32) | (k >> 32); }
        __syncthreads();
    }
    if (tid < m) {
        const unsigned a = 0xFFFFFFFFu - (unsigned)(s_keys[tid] & 0xFFFFFFFFull);
        const float4 t = locs[(size_t)i * A + a], p = priors[a];
        // box_coder.py:55-57 decode_box, then box_utils.py:16-23 to_corners (postprocessor.py:52-53)
        const float4 cen = make_float4(p.x + p.z * t.x / xy_scale, p.y + p.w * t.y / xy_scale, p.z * expf(t.z / wh_scale),
                                       p.w * expf(t.w / wh_scale));
        const float4 b = to_corners(cen);
        s_box[tid] = b;
        s_area[tid] = soft ? area4(b.x, b.y, b.z, b.w) : (b.z - b.x) * (b.w - b.y);
    }
    __syncthreads();
    if (soft) {
        // bf/utils/box_utils.py:145-163 _soft_nms on one wave; lane l owns candidates l, l+64, l+128, l+192.
        __shared__ int s_picked[kMaxPerClass];
        __shared__ int s_npicked;
        if (tid < kWave) {
            float sc[4];
            unsigned live = 0;  // the mask of :147 / :156 (refreshed AFTER the pick, BEFORE the decay -- tested one iteration late)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int idx = q * 64 + tid;
                sc[q] = idx < m ? __uint_as_float((unsigned)(s_keys[idx] >> 32)) : 0.0f;
                if (idx < m && sc[q] > score_thr) live |= 1u << q;
            }
            int npicked = 0;
            for (int it = 0; it < m; ++it) {
                int idxsum = 0;  // :151 `mask.nonzero().sum()`: the SUM OF INDICES of live candidates
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if ((live >> q) & 1u) idxsum += q * 64 + tid;
                idxsum = wave_allreduce(idxsum, OpAddI());
                if (idxsum == 0) break;
                u64 best = 0;  // :152 argmax, first maximum: (score bits, ~index)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int idx = q * 64 + tid;
                    if (idx < m) {
                        const u64 k = ((u64)__float_as_uint(sc[q]) << 32) | (u64)(0xFFFFFFFFu - (unsigned)idx);
                        best = k > best ? k : best;
                    }
                }
                best = wave_allreduce(best, OpMaxU64());
                const int bi = (int)(0xFFFFFFFFu - (unsigned)(best & 0xFFFFFFFFull));
                if (tid == 0) s_picked[npicked] = bi;
                ++npicked;
                const float4 bb = s_box[bi];
                const float ba = s_area[bi];
                live = 0;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int idx = q * 64 + tid;
                    if (idx == bi) sc[q] = 0.0f;  // :153
                    if (idx < m && sc[q] > score_thr) {  // :156 mask, then :158-160 decay of the masked candidates
                        live |= 1u << q;
                        const float4 bj = s_box[idx];
                        const float inter = area4(tmaxf(bb.x, bj.x), tmaxf(bb.y, bj.y), tminf(bb.z, bj.z), tminf(bb.w, bj.w));
                        const float iou = inter / (ba + s_area[idx] - inter);
                        sc[q] = sc[q] * expf(-(iou * iou / sigma));
                    }
                }
            }
            if (tid == 0) s_npicked = npicked;
        }
        __syncthreads();
        const int np = s_npicked;
        if (tid < np) {  // rows in pick order, ORIGINAL scores (:163 scores[picked])
            const int src = s_picked[tid];
            float* o = pc_rows + ((size_t)pc * K + tid) * 6;
            const float4 b = s_box[src];
            o[0] = b.x; o[1] = b.y; o[2] = b.z; o[3] = b.w;
            o[4] = (float)(c + 1);
            o[5] = __uint_as_float((unsigned)(s_keys[src] >> 32));
            pc_score[(size_t)pc * K + tid] = o[5];
        }
        if (tid == 0) {
            pc_count[pc] = np;
            if (nms_candidates) atomicAdd(nms_candidates + i, (u64)m);
        }
        return;
    }
    // IoU bit matrix, strict upper triangle only (j > k).  A work item is a 16-column quarter of one 64-bit word of one row, so the
    // longest dependent chain of a thread is 16 IoUs (LDS read + division each) and the triangle spreads over the 256 threads;
    // the four 16-bit parts of a word are stored side by side (little endian: part q = bits 16q .. 16q+15 of s_mask[k][w]).
    {
        unsigned short* m16 = reinterpret_cast<unsigned short*>(&s_mask[0][0]);
        for (int t = tid; t < m * W * 4; t += kPostThreads) {
            const int k = t / (W * 4), r = t % (W * 4), w = r >> 2, q = r & 3;
            const int j0 = w * 64 + q * 16;
            const int jj_begin = k + 1 > j0 ? k + 1 - j0 : 0, jj_end = m - j0 < 16 ? m - j0 : 16;
            unsigned bits = 0;
            if (jj_begin < jj_end) {
                const float4 bk = s_box[k];
                const float ak = s_area[k];
                for (int jj = jj_begin; jj < jj_end; ++jj) {
                    const int j = j0 + jj;
                    const float4 bj = s_box[j];
                    const float iw = clamp0(tminf(bk.z, bj.z) - tmaxf(bk.x, bj.x));
                    const float ih = clamp0(tminf(bk.w, bj.w) - tmaxf(bk.y, bj.y));
                    const float inter = iw * ih;
                    if (inter / (ak + s_area[j] - inter) > nms_thr) bits |= 1u << jj;
                }
            }
            m16[(k * (kMaxPerClass / 64) + w) * 4 + q] = (unsigned short)bits;
        }
    }
    __syncthreads();
    if (tid < kWave) {  // greedy sweep on one wave; lane l owns rows l, l+64, l+128, l+192
        u64 row[4][4];
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int w = 0; w < 4; ++w) row[s][w] = (s * 64 + tid < m && w < W) ? s_mask[s * 64 + tid][w] : 0ull;
        u64 removed[4] = {0, 0, 0, 0}, keep[4] = {0, 0, 0, 0};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int kend = min(64, m - s * 64);   // (<= 0 for the segments past m: no iterations)
            for (int l = 0; l < kend; ++l) {
                if (!((removed[s] >> l) & 1ull)) {
                    keep[s] |= 1ull << l;
                    // row k only has bits at j > k: words below the diagonal segment are empty, words past W do not exist
#pragma unroll
                    for (int w = 0; w < 4; ++w)
                        if (w >= s && w < W) removed[w] |= readlane_u64(row[s][w], l);
                }
            }
        }
        if (tid == 0) {
#pragma unroll
            for (int w = 0; w < 4; ++w) s_keep[w] = keep[w];
        }
    }
    __syncthreads();
    if (tid < m) {
        const int s = tid >> 6, l = tid & 63;
        if ((s_keep[s] >> l) & 1ull) {
            int pos = __popcll(s_keep[s] & ((1ull << l) - 1ull));
            for (int w = 0; w < s; ++w) pos += __popcll(s_keep[w]);
            float* o = pc_rows + ((size_t)pc * K + pos) * 6;
            const float4 b = s_box[tid];
            o[0] = b.x; o[1] = b.y; o[2] = b.z; o[3] = b.w;
            o[4] = (float)(c + 1);  // postprocessor.py:66
            o[5] = __uint_as_float((unsigned)(s_keys[tid] >> 32));
            pc_score[(size_t)pc * K + pos] = o[5];
        }
    }
    if (tid == 0) {
        int tot = 0;
        for (int w = 0; w < 4; ++w) tot += __popcll(s_keep[w]);
        pc_count[pc] = tot;
        if (nms_candidates) atomicAdd(nms_candidates + i, (u64)m);
    }
}

// ---- per-image merge: class-order concat, or top-max_total (postprocessor.py:68-74) --------------------------
__global__ void __launch_bounds__(kPostThreads) post_merge_kernel(int ncls, int K, int max_total, const float* __restrict__ pc_rows,
                                                                  const int* __restrict__ pc_count, u64* __restrict__ merge_keys,
                                                                  float* __restrict__ out, int out_cap, int* __restrict__ counts) {
    __shared__ u64 s_keys[kSortCap];
    __shared__ unsigned s_hist[256];
    __shared__ u64 s_misc[4];
    extern __shared__ int s_prefix[];  // [ncls + 1]
    const int i = blockIdx.x, tid = threadIdx.x;
    const int* cnt = pc_count + (size_t)i * ncls;
    if (tid < kWave) {   // exclusive prefix of the per-class counts: each lane a contiguous run of classes, wave scan of the run totals
        const int per = (ncls + kWave - 1) / kWave;   // (one thread adding up ncls dependent global loads was ~10 us of this kernel)
        const int c0 = min(tid * per, ncls), c1 = min(c0 + per, ncls);
        int run = 0;
        for (int c = c0; c < c1; ++c) run += cnt[c];
        int incl = run;
#pragma unroll
        for (int d = 1; d < kWave; d <<= 1) {
            const int t = __shfl_up(incl, d, kWave);
            if (tid >= d) incl += t;
        }
        int acc = incl - run;
        for (int c = c0; c < c1; ++c) { s_prefix[c] = acc; acc += cnt[c]; }
        if (tid == kWave - 1) s_prefix[ncls] = incl;
    }
    __syncthreads();
    const int T = s_prefix[ncls];
    float* o = out + (size_t)i * out_cap * 6;
    const float* rows = pc_rows + (size_t)i * ncls * K * 6;
    if (max_total <= 0 || T <= max_total) {
        const int nw = min(T, out_cap);
        for (int c = tid >> 6; c < ncls; c += kPostThreads / kWave) {  // a wave per class
            const int base = s_prefix[c], nc = cnt[c];
            for (int e = lane_id(); e < nc * 6; e += kWave) {
                const int dst = base * 6 + e;
                if (dst < nw * 6) o[dst] = rows[(size_t)c * K * 6 + e];
            }
        }
        if (tid == 0) counts[i] = nw;
        return;
    }
    u64* keys = merge_keys + (size_t)i * ncls * K;
    for (int c = tid >> 6; c < ncls; c += kPostThreads / kWave) {
        const int base = s_prefix[c], nc = cnt[c];
        for (int k = lane_id(); k < nc; k += kWave) {
            const unsigned sb = __float_as_uint(rows[((size_t)c * K + k) * 6 + 5]);
            keys[base + k] = ((u64)sb << 32) | (u64)(0xFFFFFFFFu - (unsigned)(base + k));
        }
    }
    __threadfence_block();
    __syncthreads();
    const int m = block_topk_sorted(keys, T, max_total, s_keys, s_hist, s_misc);
    const int nw = min(m, out_cap);
    for (int k = tid; k < nw; k += kPostThreads) {
        const int flat = (int)(0xFFFFFFFFu - (unsigned)(s_keys[k] & 0xFFFFFFFFull));
        int lo = 0, hi = ncls;  // class c with s_prefix[c] <= flat < s_prefix[c+1]
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (s_prefix[mid] <= flat) lo = mid; else hi = mid;
        }
        const float* r = rows + ((size_t)lo * K + (flat - s_prefix[lo])) * 6;
        float* d = o + (size_t)k * 6;
        d[0] = r[0]; d[1] = r[1]; d[2] = r[2]; d[3] = r[3]; d[4] = r[4]; d[5] = r[5];
    }
    if (tid == 0) counts[i] = nw;
}


// =====================================================================================================================
// Round-2 pipeline (hard NMS, max_per_class <= 128, num_classes <= 96): the launches above remain the general fallback.
//
//   post_select2_kernel   the pass over the logits.  Its cost is vector arithmetic, not bytes (42 M exponentials at batch 64): the row
//                         softmax keeps exp(x - max) in the LDS tile and does NOT divide -- a candidate is found by the conservative
//                         test e > thr * (1 - 2^-21) * sum (one multiply per row, one compare per element) and only the survivors get
//                         the exact IEEE e / sum and the exact `> thr` test, so the result set is the reference's.  exp itself is the
//                         two-term product form (x*log2e split hi/lo around v_exp_f32: 7 instructions, same 1-ulp error as the
//                         library's 13).  Every thread keeps a bit mask of its passing elements and walks its own set bits.  A
//                         workgroup OWNS a run of tiles and a segment of every class list (capacity = its rows): a hit's slot is the
//                         class's running count inside the workgroup (an LDS atomic) -- no global atomics, no counters to zero.
//                         For even C (RetinaNet: 80) the tile is laid out with an odd row stride, otherwise the 64 rows of one class
//                         column sit in two LDS banks.
//   sample + tau          in the worst case (every (anchor, class) pair passes the score threshold) the candidate keys were 3x the
//                         algorithmic traffic.  Now every 8th tile is selected first (mode 1), post_tau_kernel takes the K-th largest
//                         key of each (image, class) list that already holds >= K candidates -- a rigorous LOWER bound of the final
//                         K-th score, because the sample is a subset -- trims the list to those K, and the main pass (mode 2) only
//                         emits pairs at or above that bound.  Nothing that can reach the top K is dropped; lists with < K sample
//                         candidates (the trained-like case) are untouched.
//   post_nms_wave_kernel  one WAVE per (image, class): radix narrowing by the wave, survivors ranked by counting, decode of the <= K
//                         boxes into registers (lane l owns sorted entries l and l + 64), greedy sweep that computes the IoU row of an
//                         entry only while it is alive (v_readlane broadcast + two ballots), inter / union > thr decided exactly
//                         without a division.  With max_total set it runs twice (head of the best candidates per class ->
//                         post_img_tau_kernel: image-level bound -> tail for the classes that can still matter; see the kernel).
//   post_merge2_kernel    1 024 threads per image, the <= 8 192 keys of an image in registers, radix select + rank-by-counting.

constexpr int kSelMaxC = 96;
constexpr int kSelQueue = 256;       // post_select2_kernel: survivors of one wave's 16 rows that the compaction queue holds
constexpr int kSampleStride = 8;
constexpr int kWaveK = 128;          // post_nms_wave_kernel: sorted slots (max_per_class <= this)
constexpr int kMergeSlots = 8192;    // post_merge2_kernel: ncls * max_per_class <= this (8 keys per thread)
constexpr int kMergeCap = 256;       // survivors of the merge's radix narrowing (>= max_total)
constexpr int kTauCache = 1024;      // post_tau_kernel: keys of one sample list kept in LDS (8 KB per wave)
constexpr int kNmsCache = 1024;      // post_nms_wave_kernel: keys of one list kept in LDS (8 KB per wave)
constexpr int kNmsCacheHead = 1024;  // ... by the head launch (MODE 1).  (640 keys -- 7.7 KB of LDS per one-wave workgroup, 20 per CU, the 5 120 lists of
                                     // SSD-300 / 81 classes at batch 64 in ONE resident round instead of two -- measured 45.2 against 38.6 us: the lists
                                     // above 640 keys then re-read memory per radix pass, which costs more than the second round)

// exp(d) for d <= 0 (NaN stays NaN, -inf -> 0): t = d*log2e, r = the rounding error of that product + d*log2e_lo,
// exp2(t) * (1 + r*ln2).  v_exp_f32 is 1 ulp over its whole range; the library's expf adds range checks this call site does not need.
__device__ __forceinline__ float exp_nonpos(float d) {
    const float L = 1.44269502162933349609375f, Llo = 1.925963033500011e-08f;
    const float t = d * L;
    float r = fmaf(d, L, -t);
    r = fmaf(d, Llo, r);
    const float e0 = __builtin_amdgcn_exp2f(t);
    // d = -inf: e0 = 0 but r is NaN (inf - inf); the legacy multiply (0 * anything = 0) keeps the result 0 without a compare + select
    float corr;
    asm("v_mul_legacy_f32 %0, %1, %2" : "=v"(corr) : "v"(e0), "v"(r * 0.6931471805599453f));
    return e0 + corr;
}

// the digit of one radix pass: the largest bin b with above + (keys in bins > b) + hist[b] >= K.  One wave; lane l owns bins 4l..4l+3.
// Returns (all lanes): digit, cum = keys strictly above the digit's bin (incl. `above`), h = the bin's population.
__device__ __forceinline__ void wave_find_digit(const unsigned* s_hist, unsigned above, unsigned K, int* digit, unsigned* cum_out, unsigned* h_out) {
    const int l = lane_id();
    const unsigned h0 = s_hist[4 * l], h1 = s_hist[4 * l + 1], h2 = s_hist[4 * l + 2], h3 = s_hist[4 * l + 3];
    const unsigned mine = h0 + h1 + h2 + h3;
    unsigned incl = mine;
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
        const unsigned t = __shfl_down(incl, d, kWave);
        if (l + d < kWave) incl += t;
    }
    unsigned cum = above + (incl - mine);
    int found = -1;
    unsigned cum_at = 0, h_at = 0;
    const unsigned hq[4] = {h0, h1, h2, h3};
#pragma unroll
    for (int q = 3; q >= 0; --q) {
        if (found < 0) {
            if (cum + hq[q] >= K) { found = 4 * l + q; cum_at = cum; h_at = hq[q]; }
            else cum += hq[q];
        }
    }
    const unsigned long long hit = __ballot(found >= 0);
    int src = 0;
    if (hit) src = 63 - __clzll((long long)hit);
    else { found = 0; cum_at = above + (incl - h0); h_at = h0; }   // fewer than K keys in range: bin 0 (lane 0's values)
    *digit = __shfl(found, src, kWave);
    *cum_out = __shfl(cum_at, src, kWave);
    *h_out = __shfl(h_at, src, kWave);
}

// ---- P2, any max_per_class (None, or more than the 256 the bit-matrix kernel holds): greedy selection ------------------------------
// bf/utils/box_utils.py:166-194 with max_per_class=None lets every candidate of a class into NMS; a cap above 256 does not fit the
// bit matrix of post_nms_kernel.  Here: boxes of an image decoded once (all classes share them), then one workgroup per (image, class)
// repeats { best alive key -> keep it -> suppress the alive candidates it overlaps } with the alive set as a bit array in LDS.  The loop
// stops after `cap` kept boxes: more than max_total boxes of one class can never reach the final top-max_total.
__global__ void __launch_bounds__(kPostThreads) post_decode_kernel(const float4* __restrict__ locs, const float4* __restrict__ priors, int A, long long total,
                                                                   float xy_scale, float wh_scale, float4* __restrict__ boxes) {
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        const float4 l = locs[t], p = priors[t % A];
        const float4 cen = make_float4(p.x + p.z * l.x / xy_scale, p.y + p.w * l.y / xy_scale, p.z * expf(l.z / wh_scale), p.w * expf(l.w / wh_scale));
        boxes[t] = to_corners(cen);   // box_coder.py:55-57, box_utils.py:16-23
    }
}

constexpr int kAnyMaxAnchors = 1 << 17;   // alive bits in LDS: 16 KB

__global__ void __launch_bounds__(kPostThreads) post_nms_any_kernel(const float4* __restrict__ boxes, int A, int ncls, int K, int cap, float nms_thr,
                                                                    const u64* __restrict__ cand, const int* __restrict__ cand_count,
                                                                    float* __restrict__ pc_rows, float* __restrict__ pc_score, int* __restrict__ pc_count,
                                                                    u64* __restrict__ nms_candidates) {
    __shared__ unsigned s_dead[kAnyMaxAnchors / 32];   // bit k: candidate k of the list is cut off, kept already, or suppressed
    __shared__ unsigned s_hist[256];
    __shared__ u64 s_red[kPostThreads / kWave];
    __shared__ u64 s_pick;
    __shared__ u64 s_misc[2];
    const int pc = blockIdx.x, tid = threadIdx.x, lane = lane_id(), wave = tid >> 6;
    const int i = pc / ncls, c = pc % ncls;
    const int n = min(cand_count[pc], A);
    if (n == 0) {
        if (tid == 0) pc_count[pc] = 0;
        return;
    }
    const u64* keys = cand + (size_t)pc * A;
    const float4* ibox = boxes + (size_t)i * A;
    for (int w = tid; w < (n + 31) / 32; w += kPostThreads) s_dead[w] = 0u;
    __syncthreads();
    int m = n;   // boxes that enter NMS
    if (K > 0 && n > K) {   // box_utils.py:186-188: only the K best scores enter -- find the K-th largest key exactly (keys are distinct)
        u64 prefix = 0;
        unsigned above = 0;
        for (int shift = 56; shift >= 0; shift -= 8) {
            for (int b = tid; b < 256; b += kPostThreads) s_hist[b] = 0;
            __syncthreads();
            for (int k = tid; k < n; k += kPostThreads) {
                const u64 key = keys[k];
                if (shift == 56 || (key >> (shift + 8)) == (prefix >> (shift + 8))) atomicAdd(&s_hist[(unsigned)(key >> shift) & 255u], 1u);
            }
            __syncthreads();
            if (tid < kWave) {
                int digit;
                unsigned cum, h;
                wave_find_digit(s_hist, above, (unsigned)K, &digit, &cum, &h);
                if (tid == 0) { s_misc[0] = prefix | ((u64)digit << shift); s_misc[1] = cum; }
            }
            __syncthreads();
            prefix = s_misc[0];
            above = (unsigned)s_misc[1];
            __syncthreads();
        }
        for (int k = tid; k < n; k += kPostThreads)
            if (keys[k] < prefix) atomicOr(&s_dead[k >> 5], 1u << (k & 31));
        m = K;
        __syncthreads();
    }
    int kept = 0;
    while (kept < cap) {
        // best alive key
        u64 best = 0;
        for (int k = tid; k < n; k += kPostThreads) {
            if ((s_dead[k >> 5] >> (k & 31)) & 1u) continue;
            const u64 key = keys[k];
            best = key > best ? key : best;
        }
        best = wave_allreduce(best, OpMaxU64());
        if (lane == 0) s_red[wave] = best;
        __syncthreads();
        if (tid == 0) {
            u64 b = s_red[0];
            for (int w = 1; w < kPostThreads / kWave; ++w) b = s_red[w] > b ? s_red[w] : b;
            s_pick = b;
        }
        __syncthreads();
        const u64 pick = s_pick;
        if (pick == 0) break;
        const unsigned pa = 0xFFFFFFFFu - (unsigned)(pick & 0xFFFFFFFFull);
        const float4 pb = ibox[min(pa, (unsigned)(A - 1))];
        const float parea = (pb.z - pb.x) * (pb.w - pb.y);
        if (tid == 0) {
            float* o = pc_rows + ((size_t)pc * cap + kept) * 6;
            o[0] = pb.x; o[1] = pb.y; o[2] = pb.z; o[3] = pb.w;
            o[4] = (float)(c + 1);   // postprocessor.py:66
            o[5] = __uint_as_float((unsigned)(pick >> 32));
            pc_score[(size_t)pc * cap + kept] = o[5];
        }
        // the pick itself and everything it overlaps leave the alive set (torchvision.ops.nms contract: IoU > threshold)
        for (int k = tid; k < n; k += kPostThreads) {
            if ((s_dead[k >> 5] >> (k & 31)) & 1u) continue;
            const u64 key = keys[k];
            bool kill = key == pick;
            if (!kill) {
                const float4 bj = ibox[min(0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull), (unsigned)(A - 1))];
                const float iw = clamp0(tminf(pb.z, bj.z) - tmaxf(pb.x, bj.x));
                const float ih = clamp0(tminf(pb.w, bj.w) - tmaxf(pb.y, bj.y));
                const float inter = iw * ih;
                kill = inter / (parea + (bj.z - bj.x) * (bj.w - bj.y) - inter) > nms_thr;
            }
            if (kill) atomicOr(&s_dead[k >> 5], 1u << (k & 31));
        }
        ++kept;
        __syncthreads();
    }
    if (tid == 0) {
        pc_count[pc] = kept;
        if (nms_candidates) atomicAdd(nms_candidates + i, (u64)m);
    }
}

// Soft-NMS (bf/utils/box_utils.py:145-163) for any number of candidates: the greedy path's counterpart of the wave loop in post_nms_kernel.
// One workgroup per (image, class); the decaying scores live in global memory (cur[k], one float per candidate of the list; -inf = not
// among the max_per_class best, i.e. not in the array _soft_nms sees).  What the reference's positions decide is reproduced without
// sorting the list: the array order is ascending anchor (boxes[mask], postprocessor.py:63) or, after box_utils.py:186's top-k,
// descending (score, ascending anchor) -- the oracle's convention for the unordered top-k -- so
//   * argmax's first maximum (:152; NaN ranks above everything) = the entry with the largest tie value among those holding the maximum,
//     tie = ~anchor (no top-k) or the original key (top-k),
//   * `mask.nonzero().sum()` (:151, the SUM OF INDICES) is zero when nothing is live or when the only live entry is position 0 = the entry
//     with the largest tie value of all,
// and the mask tested at the top of an iteration is the one taken after the previous pick and BEFORE its decay (:156 vs :158-160).
// Rows leave in pick order with the ORIGINAL scores (:163), so a class may hand all of its entries to the merge (cap = array length).
__global__ void __launch_bounds__(kPostThreads) post_softnms_any_kernel(const float4* __restrict__ boxes, int A, int ncls, int K, int cap, float score_thr,
                                                                        float sigma, const u64* __restrict__ cand, const int* __restrict__ cand_count,
                                                                        float* __restrict__ cur_all, float* __restrict__ pc_rows, float* __restrict__ pc_score,
                                                                        int* __restrict__ pc_count, u64* __restrict__ nms_candidates) {
    __shared__ unsigned s_hist[256];
    __shared__ u64 s_misc[2];
    __shared__ u64 s_r64[kPostThreads / kWave];
    __shared__ int s_rlive[kPostThreads / kWave];
    __shared__ int s_k0, s_pick;
    const int pc = blockIdx.x, tid = threadIdx.x, lane = lane_id(), wave = tid >> 6;
    const int i = pc / ncls, c = pc % ncls;
    const int n = min(cand_count[pc], A);
    if (n == 0) {
        if (tid == 0) pc_count[pc] = 0;
        return;
    }
    const u64* keys = cand + (size_t)pc * A;
    float* cur = cur_all + (size_t)pc * A;
    const float4* ibox = boxes + (size_t)i * A;
    const bool topk = K > 0 && n > K;
    u64 prefix = 0;
    if (topk) {   // box_utils.py:186-188: the K-th largest key (keys are distinct), as in post_nms_any_kernel
        unsigned above = 0;
        for (int shift = 56; shift >= 0; shift -= 8) {
            for (int b = tid; b < 256; b += kPostThreads) s_hist[b] = 0;
            __syncthreads();
            for (int k = tid; k < n; k += kPostThreads) {
                const u64 key = keys[k];
                if (shift == 56 || (key >> (shift + 8)) == (prefix >> (shift + 8))) atomicAdd(&s_hist[(unsigned)(key >> shift) & 255u], 1u);
            }
            __syncthreads();
            if (tid < kWave) {
                int digit;
                unsigned cum, h;
                wave_find_digit(s_hist, above, (unsigned)K, &digit, &cum, &h);
                if (tid == 0) { s_misc[0] = prefix | ((u64)digit << shift); s_misc[1] = cum; }
            }
            __syncthreads();
            prefix = s_misc[0];
            above = (unsigned)s_misc[1];
            __syncthreads();
        }
    }
    const int m = topk ? K : n;
    auto tie_of = [&](u64 key) -> u64 { return topk ? key : (key & 0xFFFFFFFFull); };
    auto block_max64 = [&](u64 v) -> u64 {
        v = wave_allreduce(v, OpMaxU64());
        if (lane == 0) s_r64[wave] = v;
        __syncthreads();
        u64 r = s_r64[0];
        for (int w = 1; w < kPostThreads / kWave; ++w) r = s_r64[w] > r ? s_r64[w] : r;
        __syncthreads();
        return r;
    };
    // the array: cur = the scores, -inf outside; position 0 = the largest tie value; every entry starts live (the list only holds scores > threshold)
    u64 t0 = 0;
    for (int k = tid; k < n; k += kPostThreads) {
        const u64 key = keys[k];
        const bool in = key >= prefix;
        cur[k] = in ? __uint_as_float((unsigned)(key >> 32)) : -INFINITY;
        if (in) { const u64 t = tie_of(key); t0 = t > t0 ? t : t0; }
    }
    t0 = block_max64(t0);
    for (int k = tid; k < n; k += kPostThreads)
        if (keys[k] >= prefix && tie_of(keys[k]) == t0) s_k0 = k;
    __threadfence_block();
    __syncthreads();
    const int k0 = s_k0;
    int nlive = m, live0 = 1;   // the mask of :147
    int np = 0;
    for (int it = 0; it < m; ++it) {
        if (nlive == 0 || (nlive == 1 && live0)) break;   // :151
        // :152 argmax over the whole array: the largest value (NaN first), then the largest tie value among its holders
        unsigned pmax = 0;
        for (int k = tid; k < n; k += kPostThreads) {
            const float v = cur[k];
            if (v == -INFINITY) continue;
            unsigned u = __float_as_uint(v);
            u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
            if (v != v) u = 0xFFFFFFFFu;
            pmax = u > pmax ? u : pmax;
        }
        pmax = (unsigned)block_max64((u64)pmax);
        u64 tb = 0;
        for (int k = tid; k < n; k += kPostThreads) {
            const float v = cur[k];
            if (v == -INFINITY) continue;
            unsigned u = __float_as_uint(v);
            u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
            if (v != v) u = 0xFFFFFFFFu;
            if (u == pmax) { const u64 t = tie_of(keys[k]); tb = t > tb ? t : tb; }
        }
        tb = block_max64(tb);
        for (int k = tid; k < n; k += kPostThreads)
            if (cur[k] != -INFINITY && tie_of(keys[k]) == tb) { s_pick = k; cur[k] = 0.0f; }   // :153 (one thread: tie values are distinct)
        __threadfence_block();
        __syncthreads();
        const int bk = s_pick;
        const u64 bkey = keys[bk];
        const float4 bb = ibox[min(0xFFFFFFFFu - (unsigned)(bkey & 0xFFFFFFFFull), (unsigned)(A - 1))];
        const float ba = area4(bb.x, bb.y, bb.z, bb.w);
        if (tid == 0 && np < cap) {   // :154, :163: pick order, original score
            float* o = pc_rows + ((size_t)pc * cap + np) * 6;
            o[0] = bb.x; o[1] = bb.y; o[2] = bb.z; o[3] = bb.w;
            o[4] = (float)(c + 1);   // postprocessor.py:66
            o[5] = __uint_as_float((unsigned)(bkey >> 32));
            pc_score[(size_t)pc * cap + np] = o[5];
        }
        ++np;
        // :156 the mask, then :158-160 the decay of the masked entries
        int mylive = 0, my0 = 0;
        for (int k = tid; k < n; k += kPostThreads) {
            const float v = cur[k];
            if (v > score_thr) {   // (false for -inf, NaN and the zeros of the picked ones)
                ++mylive;
                if (k == k0) my0 = 1;
                const float4 bj = ibox[min(0xFFFFFFFFu - (unsigned)(keys[k] & 0xFFFFFFFFull), (unsigned)(A - 1))];
                const float inter = area4(tmaxf(bb.x, bj.x), tmaxf(bb.y, bj.y), tminf(bb.z, bj.z), tminf(bb.w, bj.w));
                const float iou = inter / (ba + area4(bj.x, bj.y, bj.z, bj.w) - inter);
                cur[k] = v * expf(-(iou * iou / sigma));
            }
        }
        mylive = wave_allreduce(mylive | (my0 << 24), OpAddI());   // (at most 2^17 live entries; one entry is position 0)
        if (lane == 0) s_rlive[wave] = mylive;
        __threadfence_block();
        __syncthreads();
        int tot = 0;
        for (int w = 0; w < kPostThreads / kWave; ++w) tot += s_rlive[w];
        nlive = tot & 0xFFFFFF;
        live0 = tot >> 24;
        __syncthreads();
    }
    if (tid == 0) {
        pc_count[pc] = min(np, cap);
        if (nms_candidates) atomicAdd(nms_candidates + i, (u64)m);
    }
}

// One workgroup of the select pass owns a run of `tiles_per_wg` consecutive tiles of one image AND a segment of every class list of
// that image (capacity = its rows): a hit's slot is the class's running count inside the workgroup (an LDS atomic), so nothing is
// reserved through global memory -- no returning global atomic sits on the critical path of a tile (it was 3-5 us of every tile's
// ~8), no counter has to be zeroed before a call, and the order inside a list, which nothing depends on (keys are distinct and get
// ranked later), is whatever the hardware made it.  At the end the workgroup publishes its per-class counts.
struct SelArgs {
    const float* scores;
    int A, C, ncls, c_off;
    float thr;
    int mode;          // 0: every tile, 1: tiles with tile % kSampleStride == 0, 2: the others
    int sel_tiles;     // tiles of this mode per image
    int tiles_per_wg;  // T: workgroup g takes tiles g*T .. g*T + T - 1 of this mode
    long long list_cap;   // keys per (image, class) list
    int seg_off;       // first key of this pass's segment 0 inside a list; segment g starts at seg_off + g * T * 64
    int nseg;          // segment counters per (image, class)
    int seg0;          // index of this pass's segment 0 among them
    const unsigned* tau;   // [B * ncls] score-bit lower bounds (0 = none) or NULL
    const unsigned* hotb;  // [B * ncls] score bits at or above which a key is HOT (NULL: every key is)
    int hotb_per_image;    // != 0: hotb is indexed [image * ncls + class]; 0: [class] -- one bound per class for every image (HotState)
    u64* cand;         // [B * ncls][list_cap]
    int* segcnt;       // [B * ncls][nseg]  keys in the segment
    int* seghot;       // [B * ncls][nseg]  of which hot (filled from the segment's front; the cold ones from its back)
    int stop;          // debug (SSDK_POST_STOP): 1 = staging only, 2 = no threshold test
};

// JMAX = elements per thread of the row pass (4 threads per row): ceil(C / 4) <= JMAX
template <bool SOFTMAX, bool PAD, int JMAX>
__global__ void __launch_bounds__(kPostThreads) post_select2_kernel(SelArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    const int C = a.C, Cp = PAD ? C + 1 : C, ncls = a.ncls;
    float* s_tile = reinterpret_cast<float*>(s_raw);
    float* s_pre = s_tile + align_up((size_t)kPostTileRows * Cp, 4);
    unsigned* s_taub = reinterpret_cast<unsigned*>(s_pre + ncls);
    int* s_ccnt = reinterpret_cast<int*>(s_taub + ncls);            // hot keys of the class so far (slots from the front of the segment)
    int* s_cold = s_ccnt + ncls;                                      // cold keys (slots from its back)
    unsigned* s_hotb = reinterpret_cast<unsigned*>(s_cold + ncls);
    float* s_rowmax = reinterpret_cast<float*>(s_hotb + ncls);   // [64] per row of the tile: max logit,
    float* s_rowsum = s_rowmax + kPostTileRows;                    //      sum of the softmax terms,
    float* s_rowinv = s_rowsum + kPostTileRows;                    //      its refined reciprocal
    int* s_qn = reinterpret_cast<int*>(s_rowinv + kPostTileRows);   // [4] entries in each wave's survivor queue
    unsigned short* s_queue = reinterpret_cast<unsigned short*>(s_qn + kPostThreads / kWave);   // [4][kSelQueue]: row << 8 | column
    const int i = blockIdx.y, g = blockIdx.x, tid = threadIdx.x;
    const float thr = a.thr;
    for (int c = tid; c < ncls; c += kPostThreads) {
        const unsigned tb = a.tau ? a.tau[(size_t)i * ncls + c] : 0u;
        const float teff = fmaxf(thr, __uint_as_float(tb));
        // conservative pre-test levels: softmax e > pre * sum, sigmoid x > pre (logit of the level, minus a margin far above logf's error)
        float pre;
        if (SOFTMAX) pre = teff > 0.0f ? teff * (1.0f - 1.9073486328125e-06f) : teff * (1.0f + 1.9073486328125e-06f);   // (2^-19: above the row pass' one-instruction exponential's error for every term that can pass: |t| <= log2(1 / thr))
        else pre = teff <= 0.0f ? -INFINITY : (teff >= 1.0f ? INFINITY : logf(teff / (1.0f - teff)) - 1e-3f);
        s_pre[c] = pre;
        s_taub[c] = tb;
        s_ccnt[c] = 0;
        s_cold[c] = 0;
        s_hotb[c] = a.hotb ? a.hotb[(a.hotb_per_image ? (size_t)i * ncls : (size_t)0) + c] : 0u;
    }
    __syncthreads();
    float lvl_min = INFINITY;   // (every thread the same loop: ncls broadcast reads once per workgroup)
    for (int c = 0; c < ncls; ++c) lvl_min = fminf(lvl_min, s_pre[c]);
    u64* seg = a.cand + (size_t)i * ncls * a.list_cap + a.seg_off + (size_t)g * a.tiles_per_wg * kPostTileRows;
    const unsigned cap32 = (unsigned)a.list_cap;   // (ncls * list_cap < 2^31: checked by the host)
    const int seg_last = a.tiles_per_wg * kPostTileRows - 1;   // last slot of this workgroup's segment of a class list
    const int u_end = min(a.sel_tiles, (g + 1) * a.tiles_per_wg);
    auto tile_of = [&](int u) { return a.mode == 0 ? u : (a.mode == 1 ? u * kSampleStride : u + u / (kSampleStride - 1) + 1); };
    // The NEXT tile's 20 KB are fetched into registers (<= 6 float4 per thread) while this one is processed: a workgroup that loads,
    // waits, computes and only then loads again keeps the memory system busy a third of the time (168 MB in 42 us with five workgroups
    // per CU taking turns; the row pass and the survivors are another 40).  Only for 16-byte aligned, unpadded tiles; the other layouts
    // stage as before.
    constexpr int kPre = 6;   // 64 rows x 96 classes / 4 floats / 256 threads
    float4 pre0, pre1, pre2, pre3, pre4, pre5;   // (named registers: an array here ended up in scratch memory)
    pre0 = pre1 = pre2 = pre3 = pre4 = pre5 = make_float4(0.f, 0.f, 0.f, 0.f);
    static_assert(kPre == 6, "six prefetch registers");
    bool pre_valid = false;
#define SSDK_SEL_PREFETCH(U)                                                                                                   \
    do {                                                                                                                       \
        const int a0n_ = tile_of(U) * kPostTileRows;                                                                           \
        const int rowsn_ = min(kPostTileRows, a.A - a0n_);                                                                     \
        const float* srcn_ = a.scores + ((size_t)i * a.A + a0n_) * C;                                                          \
        pre_valid = !PAD && (reinterpret_cast<uintptr_t>(srcn_) & 15) == 0 && rowsn_ * C >= 4;                                 \
        if (pre_valid) {                                                                                                       \
            const int n4_ = (rowsn_ * C) >> 2;                                                                                 \
            const float4* s4_ = reinterpret_cast<const float4*>(srcn_);                                                        \
            pre0 = s4_[min(tid, n4_ - 1)];                                                                                     \
            pre1 = s4_[min(tid + kPostThreads, n4_ - 1)];                                                                      \
            pre2 = s4_[min(tid + 2 * kPostThreads, n4_ - 1)];                                                                  \
            pre3 = s4_[min(tid + 3 * kPostThreads, n4_ - 1)];                                                                  \
            pre4 = s4_[min(tid + 4 * kPostThreads, n4_ - 1)];                                                                  \
            pre5 = s4_[min(tid + 5 * kPostThreads, n4_ - 1)];                                                                  \
        }                                                                                                                      \
    } while (0)
    if (g * a.tiles_per_wg < u_end) SSDK_SEL_PREFETCH(g * a.tiles_per_wg);
    for (int u = g * a.tiles_per_wg; u < u_end; ++u) {
        const int tile = tile_of(u);
        const int a0 = tile * kPostTileRows;
        const int rows = min(kPostTileRows, a.A - a0);
        const int nfloat = rows * C;
        const float* src = a.scores + ((size_t)i * a.A + a0) * C;
        __syncthreads();
        const bool vec = (reinterpret_cast<uintptr_t>(src) & 15) == 0;
        if (!PAD && pre_valid) {   // (uniform: the same for every thread of the workgroup)
            const int n4 = nfloat >> 2;
            float4* t4 = reinterpret_cast<float4*>(s_tile);
            t4[min(tid, n4 - 1)] = pre0;
            t4[min(tid + kPostThreads, n4 - 1)] = pre1;
            t4[min(tid + 2 * kPostThreads, n4 - 1)] = pre2;
            t4[min(tid + 3 * kPostThreads, n4 - 1)] = pre3;
            t4[min(tid + 4 * kPostThreads, n4 - 1)] = pre4;
            t4[min(tid + 5 * kPostThreads, n4 - 1)] = pre5;
            for (int t = (nfloat & ~3) + tid; t < nfloat; t += kPostThreads) s_tile[t] = src[t];
        } else if (!PAD) {
            if (vec) {
                if (nfloat >> 2) stage_tile_f4(reinterpret_cast<float4*>(s_tile), reinterpret_cast<const float4*>(src), nfloat >> 2);
                for (int t = (nfloat & ~3) + tid; t < nfloat; t += kPostThreads) s_tile[t] = src[t];
            } else {
                for (int t = tid; t < nfloat; t += kPostThreads) s_tile[t] = src[t];
            }
        } else {
            // odd row stride (even C: the four threads of a row, 16 rows per wave, would otherwise meet in 8 of the 32 banks):
            // element e of the tile goes to e + e / C
            const float invC = 1.0f / (float)C;
            if (vec) {
                const int n4 = nfloat >> 2;
                for (int base = 0; base < n4; base += 4 * kPostThreads) {
                    float4 v[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] = reinterpret_cast<const float4*>(src)[min(base + tid + k * kPostThreads, n4 - 1)];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int e = 4 * min(base + tid + k * kPostThreads, n4 - 1);
                        int row = (int)(((float)e + 0.5f) * invC);
                        int col = e - row * C;
                        const float vv[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            s_tile[row * Cp + col] = vv[j];
                            if (++col == C) { col = 0; ++row; }
                        }
                    }
                }
                for (int e = (nfloat & ~3) + tid; e < nfloat; e += kPostThreads) s_tile[e + e / C] = src[e];
            } else {
                for (int e = tid; e < nfloat; e += kPostThreads) s_tile[e + e / C] = src[e];
            }
        }
        __syncthreads();
        pre_valid = false;
        if (u + 1 < u_end) SSDK_SEL_PREFETCH(u + 1);   // in flight during the row pass and the survivors below
        if (a.stop == 1) continue;
        // --- row pass, 4 threads per row, the row's values in registers: softmax terms (written back to the tile), and a bit per
        // element that passes the conservative test.  Then every thread walks ITS set bits: the exact probability, the exact tests,
        // and the key straight into the class's segment.  (A wave-wide `if (any lane passes)` around the exact part per element ran it
        // for nearly every element: at a few % of passing pairs some lane of 64 almost always has one.)
        const int row = tid >> 2, q = tid & 3;
        const bool live = row < rows;
        float* x = s_tile + row * Cp;
        float sum = 0.0f, row_max = 0.0f;
        unsigned bits = 0;
        {
            float e[JMAX];
            float m = -INFINITY;
#pragma unroll
            for (int j = 0; j < JMAX; ++j) {
                const int c = q + 4 * j;
                e[j] = (live && c < C) ? x[c] : -INFINITY;
                m = fmaxf(m, e[j]);
            }
            if (SOFTMAX) {   // postprocessor.py:43 F.softmax(dim=-1)
                m = quad_max(m);
                // exp2((x - max) * log2 e) alone: the product's rounding error makes a term wrong by |t| * 2^-24 relative (t <= 0 its
                // exponent), i.e. a few ulp for the terms that matter and ever less in absolute terms for the small ones -- the same size
                // as what the order of the row's additions (here: four strided partial sums, in the reference torch's vector lanes)
                // already puts into the sum.  The numerator of a SURVIVOR is computed with the two-term product form below.  The row
                // pass was 14 of the kernel's ~35 vector instructions per logit, 8 of them this exponential's correction terms.
#pragma unroll
                for (int j = 0; j < JMAX; ++j) {
                    e[j] = __builtin_amdgcn_exp2f((e[j] - m) * 1.44269502162933349609375f);
                    sum += e[j];
                }
                sum = quad_sum(sum);
            }
            if (a.stop == 2) continue;
            const float lvl = SOFTMAX ? lvl_min * sum : lvl_min;   // one level for the whole image: the lowest of its classes' bounds
#pragma unroll
            for (int j = 0; j < JMAX; ++j) bits |= (e[j] > lvl) ? (1u << j) : 0u;
            if (SOFTMAX) row_max = m;
        }
        if (q == 0 && a.c_off) bits &= ~1u;   // postprocessor.py:46-48 drops the background column (c = 0 lives in q = 0, j = 0)
        if (!live) bits = 0;
        // Survivors of the conservative test (a few % of the elements on trained-like scores): the exact probability, the exact tests, the
        // key.  exp(x - max) is RECOMPUTED from the logit still in the tile (7 instructions) -- writing all 21 terms of every thread back
        // to LDS for the few that are read again cost a ds_write per element, ~10 us of LDS time per call at batch 64 -- and e / sum is the
        // compiler's IEEE division sequence (rcp, one Newton step on the reciprocal, two residual corrections of the quotient) with the
        // per-ROW part hoisted: sum is in [1, C] and e in (0, 1], so the scaling / fix-up instructions of the general sequence
        // (v_div_scale, v_div_fixup) have nothing to do.  5 instructions per survivor instead of ~12, same bits.
        float rinv = 0.0f;
        if (SOFTMAX) {
            const float r0 = __builtin_amdgcn_rcpf(sum);
            rinv = fmaf(fmaf(-sum, r0, 1.0f), r0, r0);
        }
        auto exact_p = [&](float xv, float mx, float sm, float ri) -> float {
            if (!SOFTMAX) return 1.0f / (1.0f + expf(-xv));   // the reference's value (:43), exactly
            const float v = exp_nonpos(xv - mx);
            const float q0 = v * ri;
            const float q1 = fmaf(fmaf(-sm, q0, v), ri, q0);
            return fmaf(fmaf(-sm, q1, v), ri, q1);
        };
        // About one survivor per thread on trained-like scores, but four or five in SOME lane of every wave: a per-thread loop runs as
        // long as its busiest lane.  So a wave first compacts its survivors into a queue in LDS (one returning LDS atomic per lane for
        // the base, then 2-byte entries), and then all 64 lanes take one entry each.  A wave with more survivors than the queue holds
        // (every pair passes: the worst case, where the lanes are evenly loaded anyway) keeps the per-thread loop.
        const int lane = tid & 63, wave = tid >> 6;
        if (SOFTMAX && q == 0 && live) { s_rowmax[row] = row_max; s_rowsum[row] = sum; s_rowinv[row] = rinv; }
        // queue base = the survivors of the lanes below, total = the wave's: from the five bits of the per-lane count (<= 24), a ballot and a
        // masked bit count each -- no LDS (a returning LDS atomic of 64 lanes on ONE address, behind a store and in front of a load, was
        // three dependent LDS round trips per tile and wave)
        const int mycnt = __popc(bits);
        int qbase = 0, total = 0;
#pragma unroll
        for (int b = 0; b < 5; ++b) {
            const unsigned long long mk = __ballot((mycnt >> b) & 1);
            qbase += (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mk >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mk, 0u)) << b;
            total += __popcll(mk) << b;
        }
        if (total <= kSelQueue) {
            unsigned short* qw = s_queue + wave * kSelQueue;
            while (bits) {
                const int j = __ffs(bits) - 1;
                bits &= bits - 1;
                qw[qbase++] = (unsigned short)((row << 8) | (q + 4 * j));
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            __builtin_amdgcn_wave_barrier();
            for (int sidx = lane; sidx < total; sidx += kWave) {
                const unsigned ent = qw[sidx];
                const int r = (int)(ent >> 8), c = (int)(ent & 255u), cls = c - a.c_off;
                if (c >= C) continue;   // (only when the level is negative: -inf padding never passes otherwise)
                const float p = exact_p(s_tile[r * Cp + c], SOFTMAX ? s_rowmax[r] : 0.0f, SOFTMAX ? s_rowsum[r] : 0.0f, SOFTMAX ? s_rowinv[r] : 0.0f);
                if (p > thr && __float_as_uint(p) >= s_taub[cls]) {               // :63
                    // hot keys (at or above the class's bound: the few the NMS head will want) from the front of the segment, the others from its back
                    const bool hot = __float_as_uint(p) >= s_hotb[cls];
                    const int k = atomicAdd(hot ? &s_ccnt[cls] : &s_cold[cls], 1);
                    const int slot = hot ? k : seg_last - k;
                    if (a.stop != 3) seg[(unsigned)cls * cap32 + (unsigned)slot] = ((u64)__float_as_uint(p) << 32) | (u64)(0xFFFFFFFFu - (unsigned)(a0 + r));
                }
            }
        } else {
            const u64 akey = (u64)(0xFFFFFFFFu - (unsigned)(a0 + row));
            while (bits) {
                const int j = __ffs(bits) - 1;
                bits &= bits - 1;
                const int c = q + 4 * j, cls = c - a.c_off;
                if (c >= C) continue;
                const float p = exact_p(x[c], row_max, sum, rinv);
                if (p > thr && __float_as_uint(p) >= s_taub[cls]) {               // :63
                    const bool hot = __float_as_uint(p) >= s_hotb[cls];
                    const int k = atomicAdd(hot ? &s_ccnt[cls] : &s_cold[cls], 1);
                    const int slot = hot ? k : seg_last - k;
                    if (a.stop != 3) seg[(unsigned)cls * cap32 + (unsigned)slot] = ((u64)__float_as_uint(p) << 32) | akey;
                }
            }
        }
    }
#undef SSDK_SEL_PREFETCH
    __syncthreads();
    for (int c = tid; c < ncls; c += kPostThreads) {
        const size_t at = ((size_t)i * ncls + c) * a.nseg + a.seg0 + g;
        a.segcnt[at] = s_ccnt[c] + s_cold[c];
        a.seghot[at] = s_ccnt[c];
    }
}

// The keys of one (image, class) as one index space 0 .. n-1: first the `ntop` keys of a contiguous list, then the segments the select
// workgroups filled (s_pref = exclusive prefix of their counts, s_pref[nseg] = total; nseg <= 64).  load(p) finds its segment by a
// binary search over the prefix (uniform trip count), so a wave fetches 64 keys per instruction whatever the segment sizes --
// walking the segments one after the other cost a dependent memory round trip per segment (20-40 us per launch).
struct KeySpace {
    const u64* top;
    int ntop;
    const u64* segs;
    int seg_cap, nseg, n;
    const int* s_pref;
    const int* s_hot;   // NULL: a segment's keys are its first (count) slots (the hot keys of a two-ended segment, or a front-filled one);
                        // else [nseg] hot counts: key `off` of a segment is slot off when off < hot, else slot seg_cap - 1 - (off - hot)
    __device__ __forceinline__ u64 load(int p) const {
        if (p < ntop) return top[p];
        const int q = p - ntop;
        int lo = 0, hi = nseg;
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (s_pref[mid] <= q) lo = mid; else hi = mid;
        }
        int off = q - s_pref[lo];
        if (s_hot) {
            const int h = s_hot[lo];
            off = off < h ? off : seg_cap - 1 - (off - h);
        }
        return segs[(size_t)lo * seg_cap + off];
    }
};

// The one-wave helpers below synchronise the wave with its own LDS traffic.  WG = true: the wave is the whole workgroup (__syncthreads, as
// the kernels of one wave per (image, class) always did); WG = false: the wave is one of several in a workgroup that walk different lists
// (post_finish_kernel) -- a workgroup barrier would wait for waves that never come, and a wave's own DS operations complete in order, so
// a fence that keeps the compiler from moving LDS accesses across it is all that is needed.
template <bool WG>
__device__ __forceinline__ void post_sync() {
    if (WG) {
        __syncthreads();
    } else {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
    }
}

// builds s_pref from the per-lane counts (lane g = count of segment g) and returns the total; one wave
template <bool WG = true>
__device__ __forceinline__ int wave_prefix_to_lds(int mycnt, int nseg, int* s_pref) {
    const int lane = lane_id();
    int incl = mycnt;
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
        const int t = __shfl_up(incl, d, kWave);
        if (lane >= d) incl += t;
    }
    s_pref[lane] = incl - mycnt;
    const int total = __shfl(incl, kWave - 1, kWave);
    if (lane == 0) s_pref[kWave] = total;
    if (lane >= nseg) s_pref[lane] = total;   // (so that s_pref[nseg] = total for any nseg <= 64)
    post_sync<WG>();
    return total;
}

// All n keys into an LDS array with every fetch of the wave in flight at once (n <= cap): the radix passes and the collection then run
// from LDS.  Fetching per pass made the launch as long as one wave's chain of dependent memory round trips (3 passes x 3 trips).
template <int CAP, bool WG = true>
__device__ __forceinline__ void wave_cache_keys(const KeySpace& ks, u64* s_all) {
    const int lane = lane_id();
    u64 v[CAP / kWave];
#pragma unroll
    for (int k = 0; k < CAP / kWave; ++k) v[k] = (k * kWave < ks.n) ? ks.load(min(k * kWave + lane, ks.n - 1)) : 0ull;
#pragma unroll
    for (int k = 0; k < CAP / kWave; ++k)
        if (k * kWave + lane < ks.n) s_all[k * kWave + lane] = v[k];
    post_sync<WG>();
}

// Radix narrowing by one wave: the prefix such that at most `cap` and at least K of the keys are >= it (keys distinct).
// s_all != NULL: the keys are in LDS (wave_cache_keys), else they are fetched through ks.
template <bool WG = true>
__device__ __forceinline__ u64 wave_radix_prefix(unsigned* s_hist, int K, int cap, const KeySpace& ks, const u64* s_all) {
    const int lane = lane_id();
    u64 prefix = 0;
    unsigned above = 0;
    for (int shift = 56; shift >= 0; shift -= 8) {
        for (int b = lane; b < 256; b += kWave) s_hist[b] = 0;
        post_sync<WG>();
        for (int p0 = 0; p0 < ks.n; p0 += 2 * kWave) {   // two independent fetches per trip
            const int pa = p0 + lane, pb = p0 + kWave + lane;
            const u64 ka = s_all ? s_all[min(pa, ks.n - 1)] : ks.load(min(pa, ks.n - 1));
            const u64 kb = s_all ? s_all[min(pb, ks.n - 1)] : ks.load(min(pb, ks.n - 1));
            if (pa < ks.n && (shift == 56 || (ka >> (shift + 8)) == (prefix >> (shift + 8)))) atomicAdd(&s_hist[(unsigned)(ka >> shift) & 255u], 1u);
            if (pb < ks.n && (shift == 56 || (kb >> (shift + 8)) == (prefix >> (shift + 8)))) atomicAdd(&s_hist[(unsigned)(kb >> shift) & 255u], 1u);
        }
        post_sync<WG>();
        int digit;
        unsigned cum, h;
        wave_find_digit(s_hist, above, (unsigned)K, &digit, &cum, &h);
        prefix |= (u64)digit << shift;
        above = cum;
        post_sync<WG>();
        if (cum + h <= (unsigned)cap) break;
    }
    return prefix;
}

// keys >= prefix -> out[0 ..) (any order), at most `cap` of them; returns the count.  One wave.
template <class Out>
__device__ __forceinline__ int wave_collect(const KeySpace& ks, const u64* s_all, u64 prefix, int cap, Out out) {
    const int lane = lane_id();
    int cnt = 0;   // (wave-uniform)
    for (int p0 = 0; p0 < ks.n; p0 += kWave) {
        const int p = p0 + lane;
        const u64 key = s_all ? s_all[min(p, ks.n - 1)] : ks.load(min(p, ks.n - 1));
        const bool keep = p < ks.n && key >= prefix;
        const unsigned long long mk = __ballot(keep);
        const int pos = cnt + __popcll(mk & ((1ull << lane) - 1ull));
        if (keep && pos < cap) out(pos, key);   // (keys are distinct, so pos < cap always; the test keeps the buffer intact if they are not)
        cnt = min(cnt + __popcll(mk), cap);
    }
    return cnt;
}

// The sample pass's candidates of every (image, class), gathered into one contiguous list top[pc][<= K]: all of them when there are
// at most K, else the K largest -- and then tau = the K-th largest key's score, a rigorous lower bound of the final K-th score.
__global__ void __launch_bounds__(kWave) post_tau_kernel(const u64* __restrict__ cand, long long list_cap, int seg_off, int seg_cap,
                                                         const int* __restrict__ segcnt, int nseg_all, int seg0, int nseg, int K,
                                                         u64* __restrict__ top, int* __restrict__ topcnt, unsigned* __restrict__ tau,
                                                         int hot_rank, unsigned* __restrict__ hotb, int* __restrict__ tophot, int* __restrict__ nall_out) {
    __shared__ unsigned s_hist[256];
    __shared__ int s_pref[kWave + 1];
    __shared__ u64 s_cache[kTauCache];
    __shared__ u64 s_out[kWaveK];
    const int pc = blockIdx.x, lane = threadIdx.x;
    const int mycnt = lane < nseg ? segcnt[(size_t)pc * nseg_all + seg0 + lane] : 0;
    KeySpace ks;
    ks.top = nullptr; ks.ntop = 0; ks.segs = cand + (size_t)pc * list_cap + seg_off; ks.seg_cap = seg_cap; ks.nseg = nseg; ks.s_pref = s_pref; ks.s_hot = nullptr;
    ks.n = wave_prefix_to_lds(mycnt, nseg, s_pref);
    // the sample holds every key above the score threshold of every kSampleStride-th tile, unpruned: an estimate of the whole list's
    // length that does not depend on the bound the main pass prunes with (post_density_hint)
    if (lane == 0 && nall_out) nall_out[pc] = ks.n * kSampleStride;
    u64* out = top + (size_t)pc * K;
    u64 prefix = 0;
    const u64* s_all = nullptr;
    if (ks.n > K && ks.n <= kTauCache) {
        wave_cache_keys<kTauCache>(ks, s_cache);
        s_all = s_cache;
    }
    if (ks.n > K) prefix = wave_radix_prefix(s_hist, K, K, ks, s_all);   // exactly K keys are >= prefix
    const int outn = ks.n ? wave_collect(ks, s_all, prefix, K, [&](int pos, u64 key) { s_out[pos] = key; }) : 0;
    __syncthreads();
    // The hot bound: the score of the hot_rank-th largest sample key.  The sample is every 8th tile, so about 8 * hot_rank keys of the
    // whole list lie at or above it -- enough for the NMS head's handful with a few % of exceptions (which then read the whole list),
    // and a tenth of what the list holds.  Fewer than hot_rank sample keys: no bound, every key is hot.
    unsigned hb = 0u;
    const unsigned scA = lane < outn ? (unsigned)(s_out[lane] >> 32) : 0u, scB = lane + kWave < outn ? (unsigned)(s_out[lane + kWave] >> 32) : 0u;
    if (hot_rank > 0 && outn >= hot_rank) {
        // the largest value v with at least hot_rank scores >= v, bit by bit (two ballots and a count per bit: no LDS, no cross-lane
        // reduction chains -- five rounds of a 64-bit wave maximum were 60 dependent ds_bpermute round trips, +5 us on this kernel)
        for (int bit = 30; bit >= 0; --bit) {
            const unsigned cand_v = hb | (1u << bit);
            const int cnt = __popcll(__ballot(scA >= cand_v)) + __popcll(__ballot(scB >= cand_v));
            if (cnt >= hot_rank) hb = cand_v;
        }
    }
    // top[pc] = the sample's keys with the hot ones first
    int nh = 0;
    for (int e0 = 0; e0 < outn; e0 += kWave) nh += __popcll(__ballot(e0 + lane < outn && (unsigned)(s_out[min(e0 + lane, outn - 1)] >> 32) >= hb));
    int ph = 0, pcold = nh;
    for (int e0 = 0; e0 < outn; e0 += kWave) {
        const int e = e0 + lane;
        const u64 k = s_out[min(e, outn - 1)];
        const bool live = e < outn, hot = live && (unsigned)(k >> 32) >= hb;
        const unsigned long long mh = __ballot(hot), mc = __ballot(live && !hot);
        const unsigned long long below = (1ull << lane) - 1ull;
        if (hot) out[ph + __popcll(mh & below)] = k;
        else if (live) out[pcold + __popcll(mc & below)] = k;
        ph += __popcll(mh);
        pcold += __popcll(mc);
    }
    if (lane == 0) {
        topcnt[pc] = outn;
        tophot[pc] = nh;
        hotb[pc] = hb;
        tau[pc] = ks.n > K ? (unsigned)(prefix >> 32) : 0u;
    }
}

// one wave per (image, class): top-K + decode + hard NMS.
// Everything here is bound by instruction issue (5 120 independent problems, a few thousand instructions each), so the structure is
// chosen for instruction count: survivors of the radix narrowing are ranked by counting (no sort network), lane l keeps sorted entries
// l and l + 64 (box, area) in registers, and the greedy sweep computes the IoU row of entry i ONLY when i is still alive -- all 64
// lanes test their two entries against box i (broadcast with v_readlane) and the two ballots are the row.  inter / union > thr is
// decided exactly without a division: RN(q) > thr  <=>  q > mid (or >= when thr's last mantissa bit is odd: ties round to even),
// mid = (thr + next float) / 2, and inter > mid * union is evaluated in fp64, where the 25 x 24-bit product is exact.
struct NmsSrc {
    const u64* cand;      // [npc][list_cap]
    long long list_cap;
    int seg_off, seg_cap; // the main pass's segments
    const int* segcnt;    // [npc][nseg_all]
    const int* seghot;    // [npc][nseg_all] hot keys of a segment (its first slots; the cold ones fill it from the back)
    int nseg_all, seg0, nseg;
    const u64* top;       // [npc][K] the sample pass's survivors (NULL: none), hot ones first
    const int* topcnt;
    const int* tophot;
    int* nall_out;        // [npc] (MODE 0 / 1) keys of the whole list: what the next call's plan is chosen from (post_density_hint)
};

// single-instruction min / max (fminf / fmaxf canonicalise every operand that may be a signalling NaN: four extra v_max per IoU here)
__device__ __forceinline__ float vmin(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float vmax(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

//
// MODE 0: the whole list in one pass.  With max_total set, only an image's max_total best kept boxes leave the postprocessor
// (postprocessor.py:72-74), so most of that work is for rows the merge drops.  Two passes instead:
// MODE 1 (head): NMS of the Khead best candidates only -- greedy NMS of a prefix of the sorted list IS the prefix of the full result;
//         head_last = score bits + 1 of the Khead-th candidate (0: the class has no candidate beyond the head).
//         post_img_tau_kernel then takes the max_total-th largest score among an image's kept head boxes: a rigorous lower bound of the
//         final max_total-th score, all of those boxes being truly kept (0 when there are not more than max_total of them).
// MODE 2 (tail): a class whose Khead-th score is below the bound is done; any other one repeats the NMS on its candidates at or above
//         the bound (all of them when it is 0) and overwrites the head's rows.  Exact: a kept box below the bound cannot be among the
//         max_total best, and a box's fate depends only on better boxes of its class.
// the LDS of one list's wave (the kernel of one wave per list declares them; post_finish_kernel hands every tail wave a slice of its own)
struct NmsLds {
    u64* s_keys;      // [kCap]
    u64* s_sorted;    // [kCap]
    unsigned* s_hist; // [256]
    int* s_pref;      // [kWave + 1]
    u64* s_cache;     // [kNmsCache]
    int* s_hot;       // [kWave]
};
constexpr size_t kNmsLdsBytes = (size_t)kWaveK * 8 * 2 + 256 * 4 + (kWave + 1 + 3) * 4 + (size_t)kNmsCache * 8 + kWave * 4;   // (MODE 2: kCap = kWaveK)
__device__ __forceinline__ NmsLds carve_nms_lds(unsigned char* base) {
    NmsLds l;
    l.s_cache = reinterpret_cast<u64*>(base);
    l.s_keys = l.s_cache + kNmsCache;
    l.s_sorted = l.s_keys + kWaveK;
    l.s_hist = reinterpret_cast<unsigned*>(l.s_sorted + kWaveK);
    l.s_pref = reinterpret_cast<int*>(l.s_hist + 256);
    l.s_hot = l.s_pref + (kWave + 1 + 3);
    return l;
}

// WG: see post_sync.  floor_in (MODE 2): the image's bound (post_img_tau_kernel's value).
template <bool TIE_UP, int MODE, bool WG>
__device__ __forceinline__ void nms_wave_body(const float4* __restrict__ locs, const float4* __restrict__ priors, int A, int ncls,
                                              int K, double thr_mid, float xy_scale, float wh_scale, const NmsSrc& src,
                                              float* __restrict__ pc_rows, float* __restrict__ pc_score, int* __restrict__ pc_count,
                                              int* __restrict__ pc_m, int stop, int Khead, unsigned floor_in,
                                              unsigned* __restrict__ head_last, int pc, const NmsLds& lds, unsigned* __restrict__ hot_acc = nullptr) {
    constexpr int kCap = MODE == 1 ? kWave : kWaveK;   // survivors of the radix narrowing (the head keeps one entry per lane)
    u64* const s_keys = lds.s_keys;
    u64* const s_sorted = lds.s_sorted;
    unsigned* const s_hist = lds.s_hist;
    int* const s_pref = lds.s_pref;
    u64* const s_cache = lds.s_cache;
    int* const s_hot = lds.s_hot;
    const int lane = lane_id();
    const int i = pc / ncls, c = pc % ncls;
    const int Kuse = MODE == 1 ? Khead : K;
    unsigned floor_bits = 0;
    if (MODE == 2) {
        const unsigned hl = head_last[pc];
        floor_bits = floor_in;
        if (hl == 0u || hl <= floor_bits) return;   // (hl - 1 < floor: every further candidate of this class is below the bound)
    }
    const int mycnt = lane < src.nseg ? src.segcnt[(size_t)pc * src.nseg_all + src.seg0 + lane] : 0;
    const int myhot = lane < src.nseg ? src.seghot[(size_t)pc * src.nseg_all + src.seg0 + lane] : 0;
    const int ntop_all = src.top ? src.topcnt[pc] : 0, ntop_hot = src.top ? src.tophot[pc] : 0;
    s_hot[lane] = myhot;
    KeySpace ks;
    ks.top = src.top ? src.top + (size_t)pc * K : nullptr;
    ks.segs = src.cand + (size_t)pc * src.list_cap + src.seg_off; ks.seg_cap = src.seg_cap; ks.nseg = src.nseg; ks.s_pref = s_pref;
    int n_all = ntop_all, n_hot = ntop_hot;
    {
        int ta = mycnt, th = myhot;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { ta += __shfl_xor(ta, d, kWave); th += __shfl_xor(th, d, kWave); }
        n_all += ta; n_hot += th;
    }
    // The head wants the Khead best keys of the list: they are all among the hot keys when there are at least Khead of those (every
    // cold key is below every hot one), and then it reads tens of keys instead of hundreds.  Anything else takes the whole list.
    const bool hot_view = MODE == 1 && n_hot >= Khead && n_hot < n_all;
    if (hot_view) {
        ks.ntop = ntop_hot; ks.s_hot = nullptr;
        ks.n = ntop_hot + wave_prefix_to_lds<WG>(myhot, src.nseg, s_pref);
    } else {
        ks.ntop = ntop_all; ks.s_hot = s_hot;
        ks.n = ntop_all + wave_prefix_to_lds<WG>(mycnt, src.nseg, s_pref);
    }
    const int n = ks.n;
    if (n_all == 0) {
        if (lane == 0) {
            pc_count[pc] = 0;
            pc_m[pc] = 0;
            if (MODE == 1) head_last[pc] = 0u;
            if (MODE != 2 && src.nall_out) src.nall_out[pc] = 0;
        }
        return;
    }
    // --- the (up to kCap) largest keys, unordered, into s_keys
    u64 prefix = 0;
    const u64* s_all = nullptr;
    constexpr int kCache = MODE == 1 ? kNmsCacheHead : kNmsCache;
    if (n > kCap && n <= kCache) {
        wave_cache_keys<kCache, WG>(ks, s_cache);
        s_all = s_cache;
    }
    if (n > kCap) prefix = wave_radix_prefix<WG>(s_hist, Kuse, kCap, ks, s_all);
    s_keys[lane] = 0ull;
    if (MODE != 1) s_keys[lane + 64] = 0ull;
    post_sync<WG>();
    const int cnt = wave_collect(ks, s_all, prefix, kCap, [&](int pos, u64 key) { s_keys[pos] = key; });
    post_sync<WG>();
    // --- rank by counting (keys are distinct): rank = number of larger keys; s_sorted[rank] = key
    const u64 kA = s_keys[lane], kB = MODE != 1 ? s_keys[lane + 64] : 0ull;
    s_sorted[lane] = 0ull;
    if (MODE != 1) s_sorted[lane + 64] = 0ull;
    post_sync<WG>();
    {
        int rA = 0, rB = 0;
        const int c4 = (cnt + 3) & ~3;
        for (int j = 0; j < c4; j += 4) {
            const u64 k0 = s_keys[j], k1 = s_keys[j + 1], k2 = s_keys[j + 2], k3 = s_keys[j + 3];   // (slots past cnt hold 0: never larger)
            rA += (k0 > kA) + (k1 > kA) + (k2 > kA) + (k3 > kA);
            if (MODE != 1) rB += (k0 > kB) + (k1 > kB) + (k2 > kB) + (k3 > kB);
        }
        if (lane < cnt) s_sorted[rA] = kA;
        if (MODE != 1 && lane + 64 < cnt) s_sorted[rB] = kB;
    }
    post_sync<WG>();
    int m = min(cnt, Kuse);   // box_utils.py:186-188
    if (MODE != 2 && lane == 0 && src.nall_out) src.nall_out[pc] = n_all;
    if (MODE == 1 && lane == 0) {
        pc_m[pc] = min(n_all, K);   // boxes that enter NMS in the reference, whatever part of the work the bound spares
        head_last[pc] = n_all > Khead ? (unsigned)(s_sorted[Khead - 1] >> 32) + 1u : 0u;
        // the next call's hot bound of this class (HotState): the lowest Khead-th best score over the images that have that many
        if (hot_acc && cnt >= Khead) atomicMin(hot_acc + c, (unsigned)(s_sorted[Khead - 1] >> 32));
    }
    if (MODE == 2 && floor_bits) {   // the sorted entries at or above the image's bound are a prefix
        const bool geA = lane < m && (unsigned)(s_sorted[lane] >> 32) >= floor_bits;
        const bool geB = lane + 64 < m && (unsigned)(s_sorted[lane + 64] >> 32) >= floor_bits;
        m = __popcll(__ballot(geA)) + __popcll(__ballot(geB));
    }
    if (stop == 1) { if (lane == 0) { pc_count[pc] = 0; pc_m[pc] = (int)(s_sorted[0] & 1); } return; }
    // --- decode: lane l holds sorted entries l and l + 64
    float4 boxA = make_float4(0.f, 0.f, 0.f, 0.f), boxB = boxA;
    float areaA = 0.f, areaB = 0.f, scoreA = 0.f, scoreB = 0.f;
    auto decode = [&](int e, float4& b, float& ar, float& sc) {
        const u64 key = s_sorted[e];
        const unsigned an = min(0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull), (unsigned)(A - 1));
        const float4 t = locs[(size_t)i * A + an], p = priors[an];
        // box_coder.py:55-57 decode_box, then box_utils.py:16-23 to_corners (postprocessor.py:52-53)
        const float4 cen = make_float4(p.x + p.z * t.x / xy_scale, p.y + p.w * t.y / xy_scale, p.z * expf(t.z / wh_scale), p.w * expf(t.w / wh_scale));
        b = to_corners(cen);
        ar = (b.z - b.x) * (b.w - b.y);
        sc = __uint_as_float((unsigned)(key >> 32));
    };
    if (lane < m) decode(lane, boxA, areaA, scoreA);
    if (MODE != 1 && lane + 64 < m) decode(lane + 64, boxB, areaB, scoreB);
    if (stop == 2) { if (lane == 0) { pc_count[pc] = 0; pc_m[pc] = (int)(boxA.x > 0.f); } return; }
    // --- greedy sweep with the IoU row of every surviving entry computed on the spot
    auto suppressed = [&](float bx1, float by1, float bx2, float by2, float barea, const float4& bj, float aj) -> bool {
        const float iw = vmax(vmin(bx2, bj.z) - vmax(bx1, bj.x), 0.0f);
        const float ih = vmax(vmin(by2, bj.w) - vmax(by1, bj.y), 0.0f);
        const float inter = iw * ih;
        const float uni = barea + aj - inter;
        const double lhs = (double)inter, rhs = thr_mid * (double)uni;
        const bool over = TIE_UP ? lhs >= rhs : lhs > rhs;
        return over & (uni > 0.0f);   // (& not &&: no branch)
    };
    u64 rem_lo = 0, rem_hi = 0, keep_lo = 0, keep_hi = 0;
    const bool liveA = lane < m, liveB = lane + 64 < m;
    const int mA = min(m, 64);
    for (int e = 0; e < mA; ++e) {
        if ((rem_lo >> e) & 1ull) continue;
        keep_lo |= 1ull << e;
        const float bx1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(boxA.x), e)), by1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(boxA.y), e));
        const float bx2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(boxA.z), e)), by2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(boxA.w), e));
        const float ba = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(areaA), e));
        rem_lo |= __ballot(liveA & suppressed(bx1, by1, bx2, by2, ba, boxA, areaA));
        if (MODE != 1 && m > 64) rem_hi |= __ballot(liveB & suppressed(bx1, by1, bx2, by2, ba, boxB, areaB));
    }
    for (int e = 64; MODE != 1 && e < m; ++e) {
        if ((rem_hi >> (e - 64)) & 1ull) continue;
        keep_hi |= 1ull << (e - 64);
        const float bx1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(boxB.x), e - 64)), by1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(boxB.y), e - 64));
        const float bx2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(boxB.z), e - 64)), by2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(boxB.w), e - 64));
        const float ba = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(areaB), e - 64));
        rem_hi |= __ballot(liveB & suppressed(bx1, by1, bx2, by2, ba, boxB, areaB));
    }
    if (stop == 3) { if (lane == 0) { pc_count[pc] = 0; pc_m[pc] = (int)(keep_lo & 1); } return; }
    // --- rows out, in sorted order
    const int nlo = __popcll(keep_lo);
    auto emit = [&](const float4& b, float sc, int pos) {
        float* o = pc_rows + ((size_t)pc * K + pos) * 6;
        o[0] = b.x; o[1] = b.y; o[2] = b.z; o[3] = b.w;
        o[4] = (float)(c + 1);   // postprocessor.py:66
        o[5] = sc;
        pc_score[(size_t)pc * K + pos] = sc;
    };
    const unsigned long long below = (1ull << lane) - 1ull;
    if (liveA && ((keep_lo >> lane) & 1ull)) emit(boxA, scoreA, __popcll(keep_lo & below));
    if (liveB && ((keep_hi >> lane) & 1ull)) emit(boxB, scoreB, nlo + __popcll(keep_hi & below));
    if (lane == 0) {
        pc_count[pc] = nlo + __popcll(keep_hi);
        if (MODE == 0) pc_m[pc] = m;   // boxes that entered NMS
    }
}

template <bool TIE_UP, int MODE>
__global__ void __launch_bounds__(kWave) post_nms_wave_kernel(const float4* __restrict__ locs, const float4* __restrict__ priors, int A, int ncls,
                                                              int K, double thr_mid, float xy_scale, float wh_scale, NmsSrc src,
                                                              float* __restrict__ pc_rows, float* __restrict__ pc_score, int* __restrict__ pc_count,
                                                              int* __restrict__ pc_m, int stop, int Khead, const unsigned* __restrict__ img_tau,
                                                              unsigned* __restrict__ head_last, unsigned* __restrict__ hot_acc) {
    constexpr int kCap = MODE == 1 ? kWave : kWaveK;
    __shared__ u64 s_keys[kCap];
    __shared__ u64 s_sorted[kCap];
    __shared__ unsigned s_hist[256];
    __shared__ int s_pref[kWave + 1];
    __shared__ u64 s_cache[MODE == 1 ? kNmsCacheHead : kNmsCache];
    __shared__ int s_hot[kWave];
    NmsLds lds;
    lds.s_keys = s_keys; lds.s_sorted = s_sorted; lds.s_hist = s_hist; lds.s_pref = s_pref; lds.s_cache = s_cache; lds.s_hot = s_hot;
    const int pc = blockIdx.x;
    nms_wave_body<TIE_UP, MODE, true>(locs, priors, A, ncls, K, thr_mid, xy_scale, wh_scale, src, pc_rows, pc_score, pc_count, pc_m, stop, Khead,
                                      MODE == 2 ? img_tau[pc / ncls] : 0u, head_last, pc, lds, MODE == 1 ? hot_acc : nullptr);
}

// The max_total-th largest score among an image's kept head boxes (post_nms_wave_kernel MODE 1), as float bits rounded down to 16
// significant bits; 0 when the image has
// at most max_total of them (then nothing may be dropped: the merge concatenates when the total stays within max_total).
constexpr int kImgTauPer = 24;   // values per thread of a 256-thread workgroup: ncls * Khead <= 256 * 24
// (THREADS threads of one workgroup, all of them; returns the bound to every thread)
template <int THREADS>
__device__ __forceinline__ unsigned img_tau_block(int i, int ncls, int K, int Khead, int max_total, const float* __restrict__ pc_score,
                                                  const int* __restrict__ pc_count, unsigned* s_hist, unsigned* s_misc, int* s_total) {
    constexpr int kPer = kImgTauPer * 256 / THREADS;
    const int tid = threadIdx.x;
    const int slots = ncls * Khead;
    unsigned v[kPer];   // score bits + 1; 0 = no row
    int mine = 0;
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
        const int s = tid + k * THREADS;
        v[k] = 0u;
        if (s < slots) {   // (count and score fetched side by side: a slot past the count holds stale bits, never an invalid address)
            const int c = s / Khead, r = s - c * Khead;
            const size_t pc = (size_t)i * ncls + c;
            const int have = pc_count[pc];
            const unsigned bits = __float_as_uint(pc_score[pc * K + r]);
            if (r < have) { v[k] = bits + 1u; ++mine; }
        }
    }
    if (tid == 0) *s_total = 0;
    __syncthreads();
    struct AddI { __device__ __forceinline__ int operator()(int a, int b) const { return a + b; } };
    mine = wave_allreduce(mine, AddI());
    if (lane_id() == 0) atomicAdd(s_total, mine);
    __syncthreads();
    if (*s_total <= max_total) return 0u;
    unsigned prefix = 0, above = 0;
    for (int shift = 24; shift >= 16; shift -= 8) {   // the upper 16 bits of the value only: still a lower bound, half the passes
        for (int b = tid; b < 256; b += THREADS) s_hist[b] = 0;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kPer; ++k)
            if (v[k] != 0u && (shift == 24 || (v[k] >> (shift + 8)) == (prefix >> (shift + 8)))) atomicAdd(&s_hist[(v[k] >> shift) & 255u], 1u);
        __syncthreads();
        if (tid < kWave) {
            int digit;
            unsigned cum, h;
            wave_find_digit(s_hist, above, (unsigned)max_total, &digit, &cum, &h);
            if (tid == 0) { s_misc[0] = prefix | ((unsigned)digit << shift); s_misc[1] = cum; }
        }
        __syncthreads();
        prefix = s_misc[0];
        above = s_misc[1];
        __syncthreads();
    }
    return prefix ? prefix - 1u : 0u;   // (prefix <= the max_total-th largest stored value, which is score bits + 1)
}
__global__ void __launch_bounds__(256) post_img_tau_kernel(int ncls, int K, int Khead, int max_total, const float* __restrict__ pc_score,
                                                           const int* __restrict__ pc_count, unsigned* __restrict__ img_tau) {
    __shared__ unsigned s_hist[256];
    __shared__ unsigned s_misc[2];
    __shared__ int s_total;
    const unsigned tau = img_tau_block<256>(blockIdx.x, ncls, K, Khead, max_total, pc_score, pc_count, s_hist, s_misc, &s_total);
    if (threadIdx.x == 0) img_tau[blockIdx.x] = tau;
}

// per-image merge with the image's keys in registers (ncls * K <= kMergeSlots)
struct MergeLds {
    u64* s_keys;       // [kMergeCap]
    unsigned* s_hist;  // [256]
    u64* s_misc;       // [4]
    int* s_n;
    int* s_prefix;     // [ncls + 1]
};
// (the 1 024 threads of one workgroup, all of them, image i)
__device__ __forceinline__ void merge_block(int i, int ncls, int K, int max_total, const float* __restrict__ pc_rows,
                                            const float* __restrict__ pc_score, const int* __restrict__ pc_count,
                                            const int* __restrict__ pc_m, float* __restrict__ out, int out_cap,
                                            int* __restrict__ counts, long long* __restrict__ nms_candidates,
                                            const int* __restrict__ pc_nall, unsigned* __restrict__ host_hint, const MergeLds& lds,
                                            unsigned floor_bits = 0u) {
    u64* const s_keys = lds.s_keys;
    unsigned* const s_hist = lds.s_hist;
    u64* const s_misc = lds.s_misc;
    int& s_n = *lds.s_n;
    int* const s_prefix = lds.s_prefix;
    const int tid = threadIdx.x;
    const int* cnt = pc_count + (size_t)i * ncls;
    if (tid < kWave) {
        const int per = (ncls + kWave - 1) / kWave;
        const int c0 = min(tid * per, ncls), c1 = min(c0 + per, ncls);
        int run = 0;
        for (int c = c0; c < c1; ++c) run += cnt[c];
        int incl = run;
#pragma unroll
        for (int d = 1; d < kWave; d <<= 1) {
            const int t = __shfl_up(incl, d, kWave);
            if (tid >= d) incl += t;
        }
        int acc = incl - run;
        for (int c = c0; c < c1; ++c) { s_prefix[c] = acc; acc += cnt[c]; }
        if (tid == kWave - 1) s_prefix[ncls] = incl;
    } else if (tid < 2 * kWave && nms_candidates) {   // boxes that entered NMS, summed over the image's classes
        long long run = 0;
        for (int c = tid - kWave; c < ncls; c += kWave) run += pc_m[(size_t)i * ncls + c];
        struct AddLL { __device__ __forceinline__ long long operator()(long long a, long long b) const { return a + b; } };
        run = wave_allreduce(run, AddLL());
        if (tid == kWave) nms_candidates[i] = run;
    } else if (tid >= 2 * kWave && tid < 3 * kWave && host_hint && i == 0) {
        // image 0's keys above the score threshold, all classes: a sample of how dense this workload's score lists are, left in pinned
        // host memory for the NEXT call's choice of plan (post_density_hint) -- never read back by this call
        long long run = 0;
        for (int c = tid - 2 * kWave; c < ncls; c += kWave) run += pc_nall[c];
        struct AddLL2 { __device__ __forceinline__ long long operator()(long long a, long long b) const { return a + b; } };
        run = wave_allreduce(run, AddLL2());
        if (tid == 2 * kWave) {
            const unsigned avg = (unsigned)min(run / (long long)(ncls > 0 ? ncls : 1), 0x7FFFFFFFLL);
            __hip_atomic_store(host_hint, avg + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // (+ 1: 0 means "nothing seen yet")
        }
    }
    __syncthreads();
    const int T = s_prefix[ncls];
    float* o = out + (size_t)i * out_cap * 6;
    const float* rows = pc_rows + (size_t)i * ncls * K * 6;
    if (max_total <= 0 || T <= max_total) {   // postprocessor.py:68-70: concatenation in class order
        const int nw = min(T, out_cap);
        for (int c = tid >> 6; c < ncls; c += 1024 / kWave) {
            const int base = s_prefix[c], nc = s_prefix[c + 1] - base;
            for (int e = lane_id(); e < nc * 6; e += kWave) {
                const int dst = base * 6 + e;
                if (dst < nw * 6) o[dst] = rows[(size_t)c * K * 6 + e];
            }
        }
        if (tid == 0) counts[i] = nw;
        return;
    }
    // :72-74 topk(max_total, sorted=True): key = (score bits, ~flat position); thread t holds slots t, t + 1024, ...
    const float* sc = pc_score + (size_t)i * ncls * K;
    u64 key[kMergeSlots / 1024];
    const int slots = ncls * K;
#pragma unroll
    for (int k = 0; k < kMergeSlots / 1024; ++k) {
        const int s = tid + k * 1024;
        key[k] = 0ull;
        if (s < slots) {
            const int c = s / K, r = s - c * K;
            const int base = s_prefix[c];
            if (r < s_prefix[c + 1] - base) key[k] = ((u64)__float_as_uint(sc[s]) << 32) | (u64)(0xFFFFFFFFu - (unsigned)(base + r));
        }
    }
    u64 prefix = 0;
    unsigned above = 0;
    bool narrowed = false;
    if (floor_bits) {
        // the image's bound (post_finish_kernel: a rigorous lower bound of the max_total-th final score) usually leaves fewer keys than the
        // rank stage holds: then it IS the narrowing, and the radix passes (two or three trips of histogram + digit search) are skipped
        if (tid == 0) s_n = 0;
        __syncthreads();
        int mine_ge = 0;
#pragma unroll
        for (int k = 0; k < kMergeSlots / 1024; ++k) mine_ge += (key[k] != 0ull && (unsigned)(key[k] >> 32) >= floor_bits) ? 1 : 0;
        mine_ge = wave_allreduce(mine_ge, OpAddI());
        if (lane_id() == 0 && mine_ge) atomicAdd(&s_n, mine_ge);
        __syncthreads();
        narrowed = s_n >= max_total && s_n <= kMergeCap;   // (>= max_total always holds for a valid bound; tested, not assumed)
        __syncthreads();
        if (narrowed) prefix = (u64)floor_bits << 32;
    }
    for (int shift = 56; shift >= 0 && !narrowed; shift -= 8) {
        for (int b = tid; b < 256; b += 1024) s_hist[b] = 0;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kMergeSlots / 1024; ++k)
            if (key[k] != 0ull && (shift == 56 || (key[k] >> (shift + 8)) == (prefix >> (shift + 8))))
                atomicAdd(&s_hist[(unsigned)(key[k] >> shift) & 255u], 1u);
        __syncthreads();
        if (tid < kWave) {
            int digit;
            unsigned cum, h;
            wave_find_digit(s_hist, above, (unsigned)max_total, &digit, &cum, &h);
            if (tid == 0) { s_misc[0] = prefix | ((u64)digit << shift); s_misc[1] = cum; s_misc[2] = cum + h; }
        }
        __syncthreads();
        prefix = s_misc[0];
        above = (unsigned)s_misc[1];
        const unsigned count_ge = (unsigned)s_misc[2];
        __syncthreads();
        if (count_ge <= (unsigned)kMergeCap) break;
    }
    if (tid == 0) s_n = 0;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kMergeSlots / 1024; ++k)
        if (key[k] != 0ull && key[k] >= prefix) {
            const int slot = atomicAdd(&s_n, 1);
            if (slot < kMergeCap) s_keys[slot] = key[k];
        }
    __syncthreads();
    const int m = min(s_n, kMergeCap);
    // rank by counting: 4 threads per survivor, each compares with a quarter of the others
    {
        const int e = tid >> 2, q = tid & 3;
        int rank = 0;
        const u64 mine = e < m ? s_keys[e] : 0ull;
        if (e < m)
            for (int j = q; j < m; j += 4) rank += s_keys[j] > mine;
        rank += __shfl_xor(rank, 1, kWave);
        rank += __shfl_xor(rank, 2, kWave);
        const int nw = min(min(m, max_total), out_cap);
        if (e < m && q == 0 && rank < nw) {
            const int flat = (int)(0xFFFFFFFFu - (unsigned)(mine & 0xFFFFFFFFull));
            int lo = 0, hi = ncls;  // class c with s_prefix[c] <= flat < s_prefix[c+1]
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (s_prefix[mid] <= flat) lo = mid; else hi = mid;
            }
            const float* r = rows + ((size_t)lo * K + (flat - s_prefix[lo])) * 6;
            float* d = o + (size_t)rank * 6;
            d[0] = r[0]; d[1] = r[1]; d[2] = r[2]; d[3] = r[3]; d[4] = r[4]; d[5] = r[5];
        }
        if (tid == 0) counts[i] = nw;
    }
}
__global__ void __launch_bounds__(1024) post_merge2_kernel(int ncls, int K, int max_total, const float* __restrict__ pc_rows,
                                                           const float* __restrict__ pc_score, const int* __restrict__ pc_count,
                                                           const int* __restrict__ pc_m, float* __restrict__ out, int out_cap,
                                                           int* __restrict__ counts, long long* __restrict__ nms_candidates,
                                                           const int* __restrict__ pc_nall, unsigned* __restrict__ host_hint) {
    __shared__ u64 s_keys[kMergeCap];
    __shared__ unsigned s_hist[256];
    __shared__ u64 s_misc[4];
    __shared__ int s_n;
    extern __shared__ int s_prefix[];  // [ncls + 1]
    MergeLds lds;
    lds.s_keys = s_keys; lds.s_hist = s_hist; lds.s_misc = s_misc; lds.s_n = &s_n; lds.s_prefix = s_prefix;
    merge_block(blockIdx.x, ncls, K, max_total, pc_rows, pc_score, pc_count, pc_m, out, out_cap, counts, nms_candidates, pc_nall, host_hint, lds);
}

// Round 5: what follows the NMS heads of an image, in ONE workgroup of 1 024 threads per image instead of three launches (image bound:
// 64 workgroups x 256 threads; tail: one wave per (image, class), nearly all of which found nothing to do; merge: 64 x 1 024):
//   A. the image's bound (img_tau_block);
//   B. the classes that must be redone above the bound (head_last > bound: a handful per image, often none) are listed, and eight waves
//      take them one list each (nms_wave_body MODE 2 with the wave-level synchronisation, a slice of LDS each);
//   C. the merge (merge_block), behind a workgroup barrier and an agent-scope fence -- the tail waves' rows were written through this
//      CU's L1, which may still hold the head's version of the same lines from phase A.
// An image with many classes to redo (more than eight) takes them in rounds of eight: correct, just slower than the chip-wide launch was.
constexpr int kFinishTailWaves = 8;
template <bool TIE_UP>
__global__ void __launch_bounds__(1024) post_finish_kernel(const float4* __restrict__ locs, const float4* __restrict__ priors, int A, int ncls, int K,
                                                           double thr_mid, float xy_scale, float wh_scale, NmsSrc src, float* __restrict__ pc_rows,
                                                           float* __restrict__ pc_score, int* __restrict__ pc_count, int* __restrict__ pc_m, int Khead,
                                                           int max_total, unsigned* __restrict__ img_tau, unsigned* __restrict__ head_last,
                                                           float* __restrict__ out, int out_cap, int* __restrict__ counts,
                                                           long long* __restrict__ nms_candidates, const int* __restrict__ pc_nall,
                                                           unsigned* __restrict__ host_hint, unsigned* __restrict__ hot_cur,
                                                           unsigned* __restrict__ hot_acc, float hot_margin) {
    __shared__ u64 s_keys[kMergeCap];
    __shared__ unsigned s_hist[256];
    __shared__ u64 s_misc[4];
    __shared__ unsigned s_misc32[2];
    __shared__ int s_n, s_total, s_nflag;
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];   // [kFinishTailWaves][kNmsLdsBytes], then s_flagged[ncls], then s_prefix[ncls + 1]
    int* const s_flagged = reinterpret_cast<int*>(s_dyn + (size_t)kFinishTailWaves * kNmsLdsBytes);
    int* const s_prefix = s_flagged + ncls;
    const int i = blockIdx.x, tid = threadIdx.x, wave = tid >> 6;
    // HotState: what the head launch accumulated becomes the next call's bound per class (a class no image had Khead candidates of: none)
    if (i == 0 && hot_cur && tid < ncls) {
        const unsigned v = hot_acc[tid];
        // (a margin below the minimum: on a DIFFERENT batch about one list in 65 would otherwise hold fewer than Khead keys at or above the
        // bound and take its whole list -- the slow waves that then end the head launch; a lower bound only lengthens the hot views a little)
        hot_cur[tid] = v == 0xFFFFFFFFu ? 0u : __float_as_uint(__uint_as_float(v) * hot_margin);
        hot_acc[tid] = 0xFFFFFFFFu;
    }
    // A
    const unsigned tau = img_tau_block<1024>(i, ncls, K, Khead, max_total, pc_score, pc_count, s_hist, s_misc32, &s_total);
    if (tid == 0) { img_tau[i] = tau; s_nflag = 0; }
    __syncthreads();
    // B
    for (int c = tid; c < ncls; c += 1024) {
        const unsigned hl = head_last[(size_t)i * ncls + c];
        if (hl != 0u && hl > tau) s_flagged[atomicAdd(&s_nflag, 1)] = c;
    }
    __syncthreads();
    const int nflag = s_nflag;
    if (wave < kFinishTailWaves && nflag) {
        const NmsLds lds = carve_nms_lds(s_dyn + (size_t)wave * kNmsLdsBytes);
        for (int f = wave; f < nflag; f += kFinishTailWaves)
            nms_wave_body<TIE_UP, 2, false>(locs, priors, A, ncls, K, thr_mid, xy_scale, wh_scale, src, pc_rows, pc_score, pc_count, pc_m, 0, Khead, tau,
                                            head_last, i * ncls + s_flagged[f], lds);
    }
    if (nflag) {   // (uniform)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    // C
    MergeLds ml;
    ml.s_keys = s_keys; ml.s_hist = s_hist; ml.s_misc = s_misc; ml.s_n = &s_n; ml.s_prefix = s_prefix;
    merge_block(i, ncls, K, max_total, pc_rows, pc_score, pc_count, pc_m, out, out_cap, counts, nms_candidates, pc_nall, host_hint, ml, tau);
}

}  // namespace ssdk

using namespace ssdk;

static inline int ncls_of(int C, int softmax) { return softmax ? C - 1 : C; }

// ---- plan of the round-2 pipeline: who owns which tiles and which segment of the class lists -------------------------------------
struct PostPlan {
    bool ok;            // shapes inside the pipeline's limits (hard NMS is checked by the caller)
    int tiles, ns;      // 64-row tiles per image; sample tiles (0: no sample pass)
    int Gs, Ts, Gm, Tm; // workgroups per image and tiles per workgroup of the sample / main pass
    long long list_cap; // keys per (image, class) list
    int nseg;
};

// Average keys per (image, class) list of image 0 of the last postprocess call that finished (+ 1; 0: none yet), written by
// post_merge2_kernel into pinned host memory.  The sample pass + per-class bound (two extra launches and a second read of 1 / 8 of the
// scores) pays when the lists are long -- random logits: 8 108 keys per list, 165 against 281 us at batch 64 -- and costs when they are
// short -- trained-like scores: ~580 keys per list, 130 against 117 us.  Both plans are exact, so a stale or missing hint only costs time.
static unsigned* g_post_hint = nullptr;
static bool g_post_hint_tried = false;
static unsigned* post_hint_word(hipStream_t s) {
    if (!g_post_hint && !g_post_hint_tried) {
        hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &st) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        if (st != hipStreamCaptureStatusNone) return nullptr;
        void* p = nullptr;
        if (hipHostMalloc(&p, 64, hipHostMallocMapped | hipHostMallocPortable) == hipSuccess && p) {
            *static_cast<volatile unsigned*>(p) = 0u;
            g_post_hint = static_cast<unsigned*>(p);
        } else {
            (void)hipGetLastError();
        }
        g_post_hint_tried = true;
    }
    return g_post_hint;
}
// HotState (round 5): in the plan WITHOUT a sample pass nothing told the select pass which keys the NMS head will want, so a head wave
// narrowed its whole list (~580 keys on trained-like scores, of which it takes 16).  Now every call leaves, per class, the lowest
// Khead-th best score over its images (post_nms_wave_kernel MODE 1 -> atomicMin, rolled over by post_finish_kernel), and the NEXT call's
// select pass files keys at or above it at the front of their segments ("hot"), as the sample plan's per-list bound does.  Exactness does
// not depend on the value: a head wave takes the hot view only when it holds at least Khead keys (every cold key is below every hot one
// of the same list, because one call uses one bound per class), else the whole list.  What depends on it is speed: on a stream of
// similar batches the hot view holds a few dozen keys.  One state per (device, stream) -- calls on a stream are ordered, so the bound a
// call reads is not written while it runs --, keyed by the call's shape; never created or re-keyed during stream capture.
// SSDK_POST_NO_HOT_STATE: off.
struct HotState {
    unsigned* cur;   // [kSelMaxC] score bits (0: every key is hot)
    unsigned* acc;   // [kSelMaxC] running minimum of this call (0xFFFFFFFF: nothing yet)
    int dev;
    hipStream_t stream;
    int ncls, khead, K, softmax;
    unsigned thr_bits;
    bool used;
};
static HotState* hot_state_for(hipStream_t s, int ncls, int khead, int K, int softmax, float thr) {
    static const bool off = getenv("SSDK_POST_NO_HOT_STATE") != nullptr;
    static std::mutex mu;
    static HotState tab[16];
    if (off || ncls > kSelMaxC) return nullptr;
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    const bool capturing = st != hipStreamCaptureStatusNone;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    unsigned thr_bits;
    memcpy(&thr_bits, &thr, 4);
    std::lock_guard<std::mutex> lock(mu);
    HotState* slot = nullptr;
    for (auto& t : tab)
        if (t.used && t.dev == dev && t.stream == s) { slot = &t; break; }
    if (slot && slot->ncls == ncls && slot->khead == khead && slot->K == K && slot->softmax == softmax && slot->thr_bits == thr_bits) return slot;
    if (capturing) return nullptr;   // (no allocation, no re-keying inside a capture: that call runs without a bound)
    if (!slot) {
        for (auto& t : tab)
            if (!t.used) { slot = &t; break; }
        if (!slot) return nullptr;
        void* p = nullptr;
        if (hipMalloc(&p, sizeof(unsigned) * 2 * kSelMaxC) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        slot->cur = static_cast<unsigned*>(p);
        slot->acc = slot->cur + kSelMaxC;
        slot->dev = dev; slot->stream = s; slot->used = true;
    }
    if (hipMemsetAsync(slot->cur, 0, sizeof(unsigned) * kSelMaxC, s) != hipSuccess ||
        hipMemsetAsync(slot->acc, 0xFF, sizeof(unsigned) * kSelMaxC, s) != hipSuccess) { (void)hipGetLastError(); slot->ncls = -1; return nullptr; }
    slot->ncls = ncls; slot->khead = khead; slot->K = K; slot->softmax = softmax; slot->thr_bits = thr_bits;
    return slot;
}

static bool post_wants_sample_pass(int K) {
    if (getenv("SSDK_POST_NO_SAMPLE")) return false;
    if (getenv("SSDK_POST_SAMPLE") || !g_post_hint) return true;
    const unsigned v = *static_cast<volatile unsigned*>(g_post_hint);
    // nothing seen yet: the plan that is safe for long lists.  Crossover measured at batch 64, K = 100 (tools/bench_post.py bg<shift>):
    // 1 264 keys per list: 153 with / 168 us without the sample pass; 580: 130 / 117; 38: 96 / 77
    return v == 0u || (long long)(v - 1u) > 8LL * K;
}

static PostPlan make_plan(int batch, int A, int C, int softmax, int K, int max_total, bool sample = true) {
    PostPlan p;
    memset(&p, 0, sizeof(p));
    const int ncls = ncls_of(C, softmax);
    p.ok = C <= kSelMaxC && K <= kWaveK && (long long)ncls * K <= kMergeSlots && max_total <= kMergeCap && !getenv("SSDK_POST_OLD");
    p.tiles = cdiv(A, kPostTileRows);
    const int ns = cdiv(p.tiles, kSampleStride);
    // sample pass + per-class bound only where the sample can hold well over max_per_class candidates
    p.ns = ((long long)ns * kPostTileRows >= 4LL * K && p.tiles - ns > 0 && sample) ? ns : 0;
    const int target_wgs = getenv("SSDK_POST_WGS") ? atoi(getenv("SSDK_POST_WGS")) : 768;   // (three resident workgroups per CU: measured best of 384 .. 2048 at batch 64, tools/r03_post_wgs.sh)
    int per_image = cdiv(target_wgs, batch);   // about five resident workgroups per CU over the whole grid ...
    per_image = per_image < 1 ? 1 : (per_image > kWave ? kWave : per_image);   // ... and at most one segment counter per lane of the consumer
    const int tm = p.tiles - p.ns;
    p.Gm = tm < per_image ? tm : per_image;
    p.Tm = cdiv(tm, p.Gm);
    p.Gm = cdiv(tm, p.Tm);
    if (p.ns) {
        p.Gs = p.ns < per_image ? p.ns : per_image;
        p.Ts = cdiv(p.ns, p.Gs);
        p.Gs = cdiv(p.ns, p.Ts);
    }
    p.list_cap = ((long long)p.Gs * p.Ts + (long long)p.Gm * p.Tm) * kPostTileRows;
    p.nseg = p.Gs + p.Gm;
    if ((long long)ncls * p.list_cap >= (1LL << 31)) p.ok = false;   // 32-bit key indices inside one image's lists
    return p;
}

struct PostWs2 {
    u64* cand;        // [npc][list_cap]
    int* segcnt;      // [npc][nseg]
    int* seghot;      // [npc][nseg]
    u64* top;         // [npc][K]
    int* topcnt;      // [npc]
    int* tophot;      // [npc]   hot keys at the front of top[pc]
    unsigned* tau;    // [npc]
    unsigned* hotb;   // [npc]   score bits of the hot bound (0: every key is hot)
    float* pc_rows;   // [npc][K][6]
    float* pc_score;  // [npc][K]
    int* pc_count;    // [npc]
    int* pc_m;        // [npc]
    unsigned* head_last;  // [npc]   two-pass NMS (post_nms_wave_kernel MODE 1 / 2)
    unsigned* img_tau;    // [batch]
    int* pc_nall;         // [npc]   keys of the whole list (post_density_hint)
};

// Head size of the two-pass NMS: about 2.5 x the share of max_total a class would get if the image's detections were spread evenly
// (so that the kept head boxes of an image comfortably exceed max_total), a multiple of 8 in 16 .. 64; 0 = one pass (no max_total, or a
// head that would not be much smaller than max_per_class).
static int nms_head_size(int ncls, int K, int max_total) {
    if (max_total <= 0 || getenv("SSDK_NMS_ONE_PASS")) return 0;
    int h = (int)((5LL * max_total / (2LL * ncls) + 7) / 8 * 8);
    h = h < 16 ? 16 : h;
    if (h > kWave || 2 * h > K || (long long)ncls * h <= max_total || (long long)ncls * h > 256LL * kImgTauPer) return 0;
    return h;
}

static PostWs2 carve_post_ws2(void* ws, size_t npc, size_t K, const PostPlan& p, size_t* total) {
    Carver c(ws);
    PostWs2 w;
    w.cand = c.take<u64>(npc * (size_t)p.list_cap);
    w.segcnt = c.take<int>(npc * (size_t)p.nseg);
    w.seghot = c.take<int>(npc * (size_t)p.nseg);
    w.top = c.take<u64>(npc * K);
    w.topcnt = c.take<int>(npc);
    w.tophot = c.take<int>(npc);
    w.tau = c.take<unsigned>(npc);
    w.hotb = c.take<unsigned>(npc);
    w.pc_rows = c.take<float>(npc * K * 6);
    w.pc_score = c.take<float>(npc * K);
    w.pc_count = c.take<int>(npc);
    w.pc_m = c.take<int>(npc);
    w.head_last = c.take<unsigned>(npc);
    w.img_tau = c.take<unsigned>(npc);   // (one per image needed; npc >= batch)
    w.pc_nall = c.take<int>(npc);
    if (total) *total = c.off;
    return w;
}

// per-class row capacity of the greedy path (max_per_class None or > 256): a class never contributes more than max_total rows
static inline long long any_cap(int num_anchors, int max_per_class, int max_total) {
    const long long per_class = max_per_class > 0 ? (max_per_class < num_anchors ? max_per_class : num_anchors) : num_anchors;
    return max_total > 0 && max_total < per_class ? max_total : per_class;
}

struct PostWsAny {
    u64* cand;
    int* cand_count;
    float4* boxes;
    float* pc_rows;
    float* pc_score;
    int* pc_count;
    u64* merge_keys;
    float* cur;   // soft-NMS only: the decaying scores, one per key of cand
};
static PostWsAny carve_post_any(void* ws, size_t B, size_t A, size_t ncls, size_t cap, size_t* total, bool soft = false) {
    Carver c(ws);
    PostWsAny w;
    w.cand = c.take<u64>(B * ncls * A);
    w.cand_count = c.take<int>(B * ncls);
    w.boxes = c.take<float4>(B * A);
    w.pc_rows = c.take<float>(B * ncls * cap * 6);
    w.pc_score = c.take<float>(B * ncls * cap);
    w.pc_count = c.take<int>(B * ncls);
    w.merge_keys = c.take<u64>(B * ncls * cap);
    w.cur = soft ? c.take<float>(B * ncls * A) : nullptr;
    if (total) *total = c.off;
    return w;
}

// (soft-NMS hands rows out in pick order with the original scores: the max_total best of an image can sit anywhere in a class's picks, so
// the per-class row capacity is the array length, not max_total)
static inline long long any_cap_of(int num_anchors, int max_per_class, int max_total, bool soft) {
    return soft ? any_cap(num_anchors, max_per_class, 0) : any_cap(num_anchors, max_per_class, max_total);
}

extern "C" size_t ssdk_postprocess_workspace_bytes_ex(int batch, int num_anchors, int num_classes, int softmax, int max_per_class,
                                                      int max_total, int soft_nms) {
    if (batch <= 0 || num_anchors <= 0 || num_classes <= 0) return 0;
    const size_t ncls = (size_t)ncls_of(num_classes, softmax);
    if (max_per_class <= 0 || max_per_class > kMaxPerClass) {   // None (<= 0) or beyond the bit-matrix kernel: the greedy path
        size_t t = 0;
        carve_post_any(nullptr, (size_t)batch, (size_t)num_anchors, ncls, (size_t)any_cap_of(num_anchors, max_per_class, max_total, soft_nms != 0), &t, soft_nms != 0);
        return t;
    }
    size_t total = 0, total2 = 0;
    carve_post_ws(nullptr, (size_t)batch, (size_t)num_anchors, ncls, (size_t)max_per_class, &total);
    for (int sample = 0; sample < 2; ++sample) {   // (the plan is chosen per call: the workspace fits either)
        const PostPlan p = make_plan(batch, num_anchors, num_classes, softmax, max_per_class, max_total, sample != 0);
        size_t t = 0;
        if (p.ok) carve_post_ws2(nullptr, (size_t)batch * ncls, (size_t)max_per_class, p, &t);
        total2 = t > total2 ? t : total2;
    }
    return total > total2 ? total : total2;   // (soft-NMS takes the general pipeline: the caller may ask for either)
}
extern "C" size_t ssdk_postprocess_workspace_bytes(int batch, int num_anchors, int num_classes, int softmax, int max_per_class, int max_total) {
    return ssdk_postprocess_workspace_bytes_ex(batch, num_anchors, num_classes, softmax, max_per_class, max_total, 0);
}

static int postprocess_v2(const PostPlan& p, const float* scores, const float* locs, const float* priors, int batch, int num_anchors,
                          int num_classes, int softmax, float score_threshold, int max_per_class, float nms_threshold, int max_total,
                          float xy_scale, float wh_scale, float* out, int out_cap, int32_t* counts, int64_t* nms_candidates, void* workspace,
                          hipStream_t s) {
    const int ncls = ncls_of(num_classes, softmax), npc = batch * ncls;
    const PostWs2 w = carve_post_ws2(workspace, (size_t)npc, (size_t)max_per_class, p, nullptr);
    const bool pad = (num_classes & 1) == 0;
    const int Cp = pad ? num_classes + 1 : num_classes;
    const size_t lds = (align_up((size_t)kPostTileRows * Cp, 4) + 5 * (size_t)ncls + 3 * kPostTileRows + kPostThreads / kWave) * 4 + (size_t)(kPostThreads / kWave) * kSelQueue * 2;
    SelArgs a;
    a.scores = scores; a.A = num_anchors; a.C = num_classes; a.ncls = ncls; a.c_off = softmax ? 1 : 0; a.thr = score_threshold;
    a.list_cap = p.list_cap; a.nseg = p.nseg; a.cand = w.cand; a.segcnt = w.segcnt; a.seghot = w.seghot; a.hotb = nullptr; a.hotb_per_image = 1;
    a.stop = getenv("SSDK_POST_STOP") ? atoi(getenv("SSDK_POST_STOP")) : 0;
    const int jneed = cdiv(num_classes, 4);
    const int khead_plan = (getenv("SSDK_NMS_STOP") && atoi(getenv("SSDK_NMS_STOP"))) ? 0 : nms_head_size(ncls, max_per_class, max_total);
    auto launch = [&](int mode, int sel_tiles, int G, int T, int seg_off, int seg0, const unsigned* tau, const unsigned* hotb) -> int {
        a.mode = mode; a.sel_tiles = sel_tiles; a.tiles_per_wg = T; a.seg_off = seg_off; a.seg0 = seg0; a.tau = tau; a.hotb = hotb;
        const dim3 grid(G, batch), block(kPostThreads);
#define SSDK_SEL(SM, PD, J) hipLaunchKernelGGL((post_select2_kernel<SM, PD, J>), grid, block, lds, s, a)
#define SSDK_SEL_J(SM, PD)                          \
    do {                                            \
        if (jneed <= 6) SSDK_SEL(SM, PD, 6);        \
        else if (jneed <= 12) SSDK_SEL(SM, PD, 12); \
        else if (jneed <= 21) SSDK_SEL(SM, PD, 21); \
        else SSDK_SEL(SM, PD, 24);                  \
    } while (0)
        if (softmax && pad) SSDK_SEL_J(true, true);
        else if (softmax) SSDK_SEL_J(true, false);
        else if (pad) SSDK_SEL_J(false, true);
        else SSDK_SEL_J(false, false);
#undef SSDK_SEL_J
#undef SSDK_SEL
        SSDK_CHECK_LAUNCH("post_select2_kernel");
        return SSDK_OK;
    };
    const int seg_off_main = p.Gs * p.Ts * kPostTileRows;
    HotState* hot = nullptr;
    if (p.ns) {
        int rc = launch(1, p.ns, p.Gs, p.Ts, 0, 0, nullptr, nullptr);
        if (rc) return rc;
        // hot bound = the hot_rank-th largest sample key: about kSampleStride * hot_rank keys of the whole list at or above it, against the
        // Khead the head wants (5 / 16 of Khead: 3 % of the lists come out short and take the whole list, 8 % exceed the head's 64 slots)
        const int hot_rank = (khead_plan && !getenv("SSDK_POST_NO_HOT")) ? (5 * khead_plan + 15) / 16 : 0;
        hipLaunchKernelGGL(post_tau_kernel, dim3(npc), dim3(kWave), 0, s, w.cand, p.list_cap, 0, p.Ts * kPostTileRows, w.segcnt, p.nseg, 0, p.Gs,
                           max_per_class, w.top, w.topcnt, w.tau, hot_rank, w.hotb, w.tophot, post_hint_word(s) ? w.pc_nall : nullptr);
        SSDK_CHECK_LAUNCH("post_tau_kernel");
        rc = launch(2, p.tiles - p.ns, p.Gm, p.Tm, seg_off_main, p.Gs, w.tau, hot_rank ? w.hotb : nullptr);
        if (rc) return rc;
    } else {
        // no sample pass: the hot bound per class the previous call on this stream left (HotState), if any
        if (khead_plan && !getenv("SSDK_POST_NO_FOLD") && !getenv("SSDK_POST_NO_HOT")) hot = hot_state_for(s, ncls, khead_plan, max_per_class, softmax, score_threshold);
        a.hotb_per_image = 0;
        const int rc = launch(0, p.tiles, p.Gm, p.Tm, 0, 0, nullptr, hot ? hot->cur : nullptr);
        a.hotb_per_image = 1;
        if (rc) return rc;
    }
    if (a.stop > 0) return SSDK_OK;   // debug: timing of the select stages alone (outputs are not written)
    // RN(inter / union) > thr  <=>  inter / union > (or >=) the midpoint of thr and the next float above it
    const float thr_next = nextafterf(nms_threshold, INFINITY);
    const double thr_mid = 0.5 * ((double)nms_threshold + (double)thr_next);
    unsigned thr_bits;
    memcpy(&thr_bits, &nms_threshold, 4);
    const int tie_up = (int)(thr_bits & 1u);   // a tie rounds to the even mantissa: up to thr_next when thr's is odd
    NmsSrc src;
    src.cand = w.cand; src.list_cap = p.list_cap; src.seg_off = p.ns ? seg_off_main : 0; src.seg_cap = p.Tm * kPostTileRows;
    src.segcnt = w.segcnt; src.seghot = w.seghot; src.nseg_all = p.nseg; src.seg0 = p.Gs; src.nseg = p.Gm;
    src.top = p.ns ? w.top : nullptr; src.topcnt = w.topcnt; src.tophot = w.tophot;
    unsigned* const hint = post_hint_word(s);
    src.nall_out = (hint && !p.ns) ? w.pc_nall : nullptr;   // (with a sample pass the estimate is the tau kernel's: the main pass' lists are pruned)
    const int nms_stop = getenv("SSDK_NMS_STOP") ? atoi(getenv("SSDK_NMS_STOP")) : 0;
    const int khead = nms_stop ? 0 : nms_head_size(ncls, max_per_class, max_total);
#define SSDK_NMS(TIE, MODE)                                                                                                                     \
    hipLaunchKernelGGL((post_nms_wave_kernel<TIE, MODE>), dim3(npc), dim3(kWave), 0, s, (const float4*)locs, (const float4*)priors, num_anchors, \
                       ncls, max_per_class, thr_mid, xy_scale, wh_scale, src, w.pc_rows, w.pc_score, w.pc_count, w.pc_m, nms_stop, khead,      \
                       w.img_tau, w.head_last, hot ? hot->acc : nullptr)
    if (khead && !getenv("SSDK_POST_NO_FOLD")) {
        if (tie_up) SSDK_NMS(true, 1); else SSDK_NMS(false, 1);
        SSDK_CHECK_LAUNCH("post_nms_wave_kernel (head)");
        // the image's bound, the few classes to redo above it and the merge: one workgroup per image (post_finish_kernel)
        const size_t dyn = (size_t)kFinishTailWaves * kNmsLdsBytes + sizeof(int) * (size_t)(2 * ncls + 1);
        static const float hot_margin = []() { const char* e = getenv("SSDK_POST_HOT_MARGIN"); const float v = e ? (float)atof(e) : 0.0f; return v > 0.0f && v <= 1.0f ? v : 0.9f; }();   // (measurement knob; batch of 64, same batch / two batches taking turns, us: 1.0: 92 / 103, 0.9: 94 / 98, 0.8: 95 / 100, 0.7: 97 / 102, 0.5: 103 / 107)
        static bool attr_set[2] = {false, false};
        if (!attr_set[tie_up]) {
            if (tie_up) SSDK_CHECK_HIP(hipFuncSetAttribute((const void*)post_finish_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024 - 8 * 1024)));
            else SSDK_CHECK_HIP(hipFuncSetAttribute((const void*)post_finish_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024 - 8 * 1024)));
            attr_set[tie_up] = true;
        }
        SSDK_REQUIRE(dyn <= 160 * 1024 - 8 * 1024, SSDK_E_UNSUPPORTED, "ssdk_postprocess: %d classes need %zu bytes of LDS", ncls, dyn);
#define SSDK_FINISH(TIE)                                                                                                                      \
    hipLaunchKernelGGL((post_finish_kernel<TIE>), dim3(batch), dim3(1024), dyn, s, (const float4*)locs, (const float4*)priors, num_anchors, ncls, \
                       max_per_class, thr_mid, xy_scale, wh_scale, src, w.pc_rows, w.pc_score, w.pc_count, w.pc_m, khead, max_total, w.img_tau,   \
                       w.head_last, out, out_cap, counts, (long long*)nms_candidates, w.pc_nall, hint, hot ? hot->cur : nullptr,          \
                       hot ? hot->acc : nullptr, hot_margin)
        if (tie_up) SSDK_FINISH(true); else SSDK_FINISH(false);
#undef SSDK_FINISH
        SSDK_CHECK_LAUNCH("post_finish_kernel");
        return SSDK_OK;
    } else if (khead) {   // (SSDK_POST_NO_FOLD: the three launches of rounds 2-4, a measurement knob)
        if (tie_up) SSDK_NMS(true, 1); else SSDK_NMS(false, 1);
        SSDK_CHECK_LAUNCH("post_nms_wave_kernel (head)");
        hipLaunchKernelGGL(post_img_tau_kernel, dim3(batch), dim3(256), 0, s, ncls, max_per_class, khead, max_total, w.pc_score, w.pc_count,
                           w.img_tau);
        SSDK_CHECK_LAUNCH("post_img_tau_kernel");
        if (tie_up) SSDK_NMS(true, 2); else SSDK_NMS(false, 2);
        SSDK_CHECK_LAUNCH("post_nms_wave_kernel (tail)");
    } else {
        if (tie_up) SSDK_NMS(true, 0); else SSDK_NMS(false, 0);
        SSDK_CHECK_LAUNCH("post_nms_wave_kernel");
    }
#undef SSDK_NMS
    hipLaunchKernelGGL(post_merge2_kernel, dim3(batch), dim3(1024), sizeof(int) * (size_t)(ncls + 1), s, ncls, max_per_class, max_total,
                       w.pc_rows, w.pc_score, w.pc_count, w.pc_m, out, out_cap, counts, (long long*)nms_candidates, w.pc_nall, hint);
    SSDK_CHECK_LAUNCH("post_merge2_kernel");
    return SSDK_OK;
}

extern "C" int ssdk_postprocess(const float* scores, const float* locs, const float* priors, int batch, int num_anchors,
                                int num_classes, int softmax, float score_threshold, int max_per_class, float nms_threshold,
                                int soft_nms, float soft_sigma, int max_total, float xy_scale, float wh_scale, float* out, int out_cap,
                                int32_t* counts,
                                int64_t* nms_candidates, void* workspace, size_t workspace_bytes, void* stream) {
    SSDK_REQUIRE(batch > 0 && num_anchors > 0 && num_classes > (softmax ? 1 : 0), SSDK_E_INVALID,
                 "ssdk_postprocess: batch=%d anchors=%d classes=%d", batch, num_anchors, num_classes);
    SSDK_REQUIRE(scores && locs && priors && out && counts, SSDK_E_INVALID, "ssdk_postprocess: null pointer");
    SSDK_REQUIRE(((uintptr_t)locs & 15) == 0 && ((uintptr_t)priors & 15) == 0, SSDK_E_INVALID, "ssdk_postprocess: locs/priors must be 16-byte aligned");
    const bool any_k = max_per_class <= 0 || max_per_class > kMaxPerClass;   // None, or beyond the bit-matrix kernel
    SSDK_REQUIRE(max_total <= kSortCap, SSDK_E_UNSUPPORTED, "ssdk_postprocess: max_total=%d > %d", max_total, kSortCap);
    SSDK_REQUIRE(!soft_nms || soft_sigma > 0.0f, SSDK_E_INVALID, "ssdk_postprocess: soft-NMS sigma must be > 0");
    const int ncls = ncls_of(num_classes, softmax);
    const long long per_class_cap = any_k ? any_cap_of(num_anchors, max_per_class, max_total, soft_nms != 0) : (max_per_class < num_anchors ? max_per_class : num_anchors);   // (a class never holds more rows than there are anchors)
    SSDK_REQUIRE(out_cap >= (max_total > 0 ? max_total : 1), SSDK_E_INVALID, "ssdk_postprocess: out_cap=%d too small", out_cap);
    SSDK_REQUIRE(max_total > 0 || (long long)out_cap >= (long long)ncls * per_class_cap, SSDK_E_INVALID,
                 "ssdk_postprocess: out_cap=%d < ncls * rows per class with max_total=None", out_cap);
    SSDK_REQUIRE(workspace && workspace_bytes >= ssdk_postprocess_workspace_bytes_ex(batch, num_anchors, num_classes, softmax, max_per_class, max_total, soft_nms),
                 SSDK_E_WORKSPACE, "ssdk_postprocess: workspace too small (soft-NMS without max_per_class or above %d: size it with ssdk_postprocess_workspace_bytes_ex)", kMaxPerClass);
    SSDK_REQUIRE((long long)batch * ncls < 2147483647LL && batch <= 65535, SSDK_E_INVALID, "ssdk_postprocess: grid too large");
    hipStream_t s = (hipStream_t)stream;
    const PostPlan plan = make_plan(batch, num_anchors, num_classes, softmax, max_per_class, max_total, post_wants_sample_pass(max_per_class));
    if (plan.ok && !soft_nms && !any_k)
        return postprocess_v2(plan, scores, locs, priors, batch, num_anchors, num_classes, softmax, score_threshold, max_per_class, nms_threshold,
                              max_total, xy_scale, wh_scale, out, out_cap, counts, nms_candidates, workspace, s);
    if (any_k) {
        // ---- greedy path: max_per_class None (every candidate enters NMS, box_utils.py:186 skipped) or > 256
        SSDK_REQUIRE(num_anchors <= kAnyMaxAnchors, SSDK_E_UNSUPPORTED, "ssdk_postprocess: max_per_class=None / > %d takes at most %d anchors", kMaxPerClass,
                     kAnyMaxAnchors);
        const int cap = (int)per_class_cap;
        SSDK_REQUIRE((long long)batch * ncls * cap < (1LL << 40), SSDK_E_UNSUPPORTED, "ssdk_postprocess: %lld rows per class", (long long)cap);
        PostWsAny w = carve_post_any(workspace, (size_t)batch, (size_t)num_anchors, (size_t)ncls, (size_t)cap, nullptr, soft_nms != 0);
        SSDK_CHECK_HIP(zero_async(w.cand_count, sizeof(int) * (size_t)batch * ncls, s));
        if (nms_candidates) SSDK_CHECK_HIP(zero_async(nms_candidates, sizeof(int64_t) * (size_t)batch, s));
        const int tiles = cdiv(num_anchors, kPostTileRows);
        const size_t lds = align_up((size_t)kPostTileRows * num_classes * 4, 16) + (size_t)ncls * 4;
        SSDK_REQUIRE(lds <= 160 * 1024 - 1024, SSDK_E_UNSUPPORTED, "ssdk_postprocess: num_classes=%d needs %zu bytes of LDS", num_classes, lds);
        SSDK_CHECK_HIP(hipFuncSetAttribute((const void*)post_select_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(post_select_kernel, dim3(tiles < 1024 ? tiles : 1024, batch), dim3(kPostThreads), lds, s, scores, num_anchors, num_classes, softmax,
                           score_threshold, tiles, w.cand, w.cand_count);
        SSDK_CHECK_LAUNCH("post_select_kernel");
        const long long total = (long long)batch * num_anchors;
        hipLaunchKernelGGL(post_decode_kernel, dim3((unsigned)(cdiv((int)((total + 255) / 256), 1) < 4096 ? (total + 255) / 256 : 4096)), dim3(kPostThreads), 0, s,
                           (const float4*)locs, (const float4*)priors, num_anchors, total, xy_scale, wh_scale, w.boxes);
        SSDK_CHECK_LAUNCH("post_decode_kernel");
        if (soft_nms) {
            hipLaunchKernelGGL(post_softnms_any_kernel, dim3(batch * ncls), dim3(kPostThreads), 0, s, (const float4*)w.boxes, num_anchors, ncls,
                               max_per_class > 0 ? max_per_class : 0, cap, score_threshold, soft_sigma, w.cand, w.cand_count, w.cur, w.pc_rows, w.pc_score,
                               w.pc_count, (u64*)nms_candidates);
            SSDK_CHECK_LAUNCH("post_softnms_any_kernel");
        } else {
            hipLaunchKernelGGL(post_nms_any_kernel, dim3(batch * ncls), dim3(kPostThreads), 0, s, (const float4*)w.boxes, num_anchors, ncls,
                               max_per_class > 0 ? max_per_class : 0, cap, nms_threshold, w.cand, w.cand_count, w.pc_rows, w.pc_score, w.pc_count,
                               (u64*)nms_candidates);
            SSDK_CHECK_LAUNCH("post_nms_any_kernel");
        }
        hipLaunchKernelGGL(post_merge_kernel, dim3(batch), dim3(kPostThreads), sizeof(int) * (size_t)(ncls + 1), s, ncls, cap, max_total, w.pc_rows,
                           w.pc_count, w.merge_keys, out, out_cap, counts);
        SSDK_CHECK_LAUNCH("post_merge_kernel");
        return SSDK_OK;
    }
    // ---- general pipeline: soft-NMS, max_per_class > 128, more than 96 classes
    PostWs w = carve_post_ws(workspace, (size_t)batch, (size_t)num_anchors, (size_t)ncls, (size_t)max_per_class, nullptr);
    SSDK_CHECK_HIP(zero_async(w.cand_count, sizeof(int) * (size_t)batch * ncls, s));
    if (nms_candidates) SSDK_CHECK_HIP(zero_async(nms_candidates, sizeof(int64_t) * (size_t)batch, s));
    const int tiles = cdiv(num_anchors, kPostTileRows);
    const size_t lds = align_up((size_t)kPostTileRows * num_classes * 4, 16) + (size_t)ncls * 4;
    SSDK_REQUIRE(lds <= 160 * 1024 - 1024, SSDK_E_UNSUPPORTED, "ssdk_postprocess: num_classes=%d needs %zu bytes of LDS", num_classes, lds);
    SSDK_CHECK_HIP(hipFuncSetAttribute((const void*)post_select_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int gx = tiles < 1024 ? tiles : 1024;
    hipLaunchKernelGGL(post_select_kernel, dim3(gx, batch), dim3(kPostThreads), lds, s, scores, num_anchors, num_classes, softmax,
                       score_threshold, tiles, w.cand, w.cand_count);
    SSDK_CHECK_LAUNCH("post_select_kernel");
    hipLaunchKernelGGL(post_nms_kernel, dim3(batch * ncls), dim3(kPostThreads), 0, s, (const float4*)locs, (const float4*)priors,
                       num_anchors, ncls, max_per_class, nms_threshold, soft_nms, soft_sigma, score_threshold, xy_scale, wh_scale, w.cand,
                       w.cand_count, w.pc_rows, w.pc_score, w.pc_count, (u64*)nms_candidates);
    SSDK_CHECK_LAUNCH("post_nms_kernel");
    hipLaunchKernelGGL(post_merge_kernel, dim3(batch), dim3(kPostThreads), sizeof(int) * (size_t)(ncls + 1), s, ncls, max_per_class,
                       max_total, w.pc_rows, w.pc_count, w.merge_keys, out, out_cap, counts);
    SSDK_CHECK_LAUNCH("post_merge_kernel");
    return SSDK_OK;
}
