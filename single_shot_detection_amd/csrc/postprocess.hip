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
};

static PostWs carve_post_ws(void* ws, size_t B, size_t A, size_t ncls, size_t K, size_t* total) {
    Carver c(ws);
    PostWs w;
    w.cand = c.take<u64>(B * ncls * A);
    w.cand_count = c.take<int>(B * ncls);
    w.pc_rows = c.take<float>(B * ncls * K * 6);
    w.pc_count = c.take<int>(B * ncls);
    w.merge_keys = c.take<u64>(B * ncls * K);
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
                                                                float* __restrict__ pc_rows, int* __restrict__ pc_count,
                                                                u64* __restrict__ nms_candidates) {
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

}  // namespace ssdk

using namespace ssdk;

static inline int ncls_of(int C, int softmax) { return softmax ? C - 1 : C; }

extern "C" size_t ssdk_postprocess_workspace_bytes(int batch, int num_anchors, int num_classes, int softmax, int max_per_class,
                                                   int max_total) {
    (void)max_total;
    if (batch <= 0 || num_anchors <= 0 || num_classes <= 0 || max_per_class <= 0) return 0;
    size_t total = 0;
    carve_post_ws(nullptr, (size_t)batch, (size_t)num_anchors, (size_t)ncls_of(num_classes, softmax), (size_t)max_per_class, &total);
    return total;
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
    SSDK_REQUIRE(max_per_class >= 1 && max_per_class <= kMaxPerClass, SSDK_E_UNSUPPORTED,
                 "ssdk_postprocess: max_per_class=%d outside 1..%d (None is not supported on the GPU path)", max_per_class, kMaxPerClass);
    SSDK_REQUIRE(max_total <= kSortCap, SSDK_E_UNSUPPORTED, "ssdk_postprocess: max_total=%d > %d", max_total, kSortCap);
    SSDK_REQUIRE(!soft_nms || soft_sigma > 0.0f, SSDK_E_INVALID, "ssdk_postprocess: soft-NMS sigma must be > 0");
    const int ncls = ncls_of(num_classes, softmax);
    SSDK_REQUIRE(out_cap >= (max_total > 0 ? max_total : 1), SSDK_E_INVALID, "ssdk_postprocess: out_cap=%d too small", out_cap);
    SSDK_REQUIRE(max_total > 0 || (long long)out_cap >= (long long)ncls * max_per_class, SSDK_E_INVALID,
                 "ssdk_postprocess: out_cap=%d < ncls*max_per_class with max_total=None", out_cap);
    SSDK_REQUIRE(workspace && workspace_bytes >= ssdk_postprocess_workspace_bytes(batch, num_anchors, num_classes, softmax, max_per_class, max_total),
                 SSDK_E_WORKSPACE, "ssdk_postprocess: workspace too small");
    SSDK_REQUIRE((long long)batch * ncls < 2147483647LL && batch <= 65535, SSDK_E_INVALID, "ssdk_postprocess: grid too large");
    hipStream_t s = (hipStream_t)stream;
    PostWs w = carve_post_ws(workspace, (size_t)batch, (size_t)num_anchors, (size_t)ncls, (size_t)max_per_class, nullptr);
    SSDK_CHECK_HIP(hipMemsetAsync(w.cand_count, 0, sizeof(int) * (size_t)batch * ncls, s));
    if (nms_candidates) SSDK_CHECK_HIP(hipMemsetAsync(nms_candidates, 0, sizeof(int64_t) * (size_t)batch, s));

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
                       w.cand_count, w.pc_rows, w.pc_count,
                       (u64*)nms_candidates);
    SSDK_CHECK_LAUNCH("post_nms_kernel");
    hipLaunchKernelGGL(post_merge_kernel, dim3(batch), dim3(kPostThreads), sizeof(int) * (size_t)(ncls + 1), s, ncls, max_per_class,
                       max_total, w.pc_rows, w.pc_count, w.merge_keys, out, out_cap, counts);
    SSDK_CHECK_LAUNCH("post_merge_kernel");
    return SSDK_OK;
}
