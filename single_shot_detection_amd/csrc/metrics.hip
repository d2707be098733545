// metrics.hip -- mean average precision on the device.
//
// Reference: detection/metrics/mean_average_precision.py:10-116 -- a Python loop over every prediction in descending score
// order (one box_utils.iou call, several .item() syncs and dict updates per prediction), then per class a Python loop for the
// running maximum of the precision.  bf/eval.py:63-69 moves all predictions to the host first.
//
// Here: two stable radix sorts (radix_* kernels below: 8-bit digits, per-tile digit histograms -> row scans -> stable scatter with
// wave-level ballot matching) give the predictions in (class, score desc) order -- the order of the
// per-class cumulative counts -- and in (image, class, score desc) order, in which the greedy matching of :47-70 is
// independent between (image, class) groups: one thread walks one group, against the <= G_i ground truths of that image.
// A workgroup per class then does the cumulative sums, precision / recall, the reverse NaN-propagating running maximum and
// the area / 11-point integration with block scans (common.h), in fp32 like the reference's tensors.
#include "common.h"

namespace ssdk {

constexpr int kMapThreads = 256;

struct MapWs {
    unsigned long long* key_a;   // [n]
    unsigned long long* key_b;   // [n]  sorted (class, score) keys -- read by map_ap_kernel
    unsigned long long* key_c;   // [n]  sorted image keys (scratch)
    unsigned* val_a;             // [n]
    unsigned* val_b;             // [n]
    unsigned* perm_b;            // [n]  prediction row at position j of the (class, score desc) order
    unsigned char* flag;         // [n]  by prediction row: 0 = neither, 1 = true positive, 2 = false positive
    unsigned char* matched;      // [total_gt]
    float* prec;                 // [n]  by (class, score) position
    float* rec;                  // [n]
    int* total_positive;         // [num_classes]
    unsigned* table;             // [256][kSortMaxBlocks] per-tile digit counts, then their row-wise exclusive scans
    unsigned* digit_total;       // [256]
    size_t total;
};

constexpr int kSortThreads = 256;
constexpr int kSortMaxBlocks = 4096;   // tiles of a sort pass (the tile grows with n beyond 4096 * 4096 keys)

static MapWs carve_map(void* base, long long n, long long total_gt, int num_classes) {
    Carver c(base);
    MapWs w;
    const size_t nn = (size_t)(n > 0 ? n : 1);
    w.key_a = c.take<unsigned long long>(nn);
    w.key_b = c.take<unsigned long long>(nn);
    w.key_c = c.take<unsigned long long>(nn);
    w.val_a = c.take<unsigned>(nn);
    w.val_b = c.take<unsigned>(nn);
    w.perm_b = c.take<unsigned>(nn);
    w.flag = c.take<unsigned char>(nn);
    w.matched = c.take<unsigned char>((size_t)(total_gt > 0 ? total_gt : 1));
    w.prec = c.take<float>(nn);
    w.rec = c.take<float>(nn);
    w.total_positive = c.take<int>((size_t)num_classes);
    w.table = c.take<unsigned>((size_t)256 * kSortMaxBlocks);
    w.digit_total = c.take<unsigned>(256);
    w.total = c.off;
    return w;
}

// order-preserving map of a float onto unsigned, inverted: ascending key = descending score
__device__ __forceinline__ unsigned desc_key(float s) {
    const unsigned u = __float_as_uint(s);
    return ~((u & 0x80000000u) ? ~u : (u | 0x80000000u));
}

// keys of the (class, score desc) order; counted ground truths per class (:27-35)
__global__ void map_prepare_kernel(const float* __restrict__ pred, long long n, const float* __restrict__ gt, int gstride, long long total_gt,
                                   int num_classes, unsigned long long* key, unsigned* val, int* total_positive) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const int c = (int)pred[i * 7 + 5];
        const bool ok = c >= 0 && c < num_classes;
        key[i] = ok ? ((unsigned long long)(unsigned)c << 32) | desc_key(pred[i * 7 + 6])
                    : ((unsigned long long)(unsigned)num_classes << 32) | 0xFFFFFFFFull;                        // foreign classes sort last
        val[i] = (unsigned)i;
    }
    if (i < total_gt) {
        const int c = (int)gt[i * gstride + 4];
        if (c >= 0 && c < num_classes && (gstride <= 6 || gt[i * gstride + 6] == 0.0f)) atomicAdd(&total_positive[c], 1);
    }
}

// second sort key: the image of the prediction at position j of the (class, score) order
__global__ void map_image_key_kernel(const float* __restrict__ pred, long long n, const unsigned* __restrict__ perm_b, unsigned long long* key,
                                     unsigned* val) {
    const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const unsigned r = perm_b[j];
    key[j] = (unsigned long long)(unsigned)(int)pred[(long long)r * 7];
    val[j] = r;
}

// greedy matching (:47-70): the thread at the head of an (image, class) group walks the group in score order
__global__ void map_match_kernel(const float* __restrict__ pred, long long n, const unsigned* __restrict__ perm_a, const float* __restrict__ gt,
                                 int gstride, const int* __restrict__ gt_off, int num_images, int num_classes, float thr,
                                 unsigned char* matched, unsigned char* flag) {
    const long long j0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j0 >= n) return;
    const float* p0 = pred + (long long)perm_a[j0] * 7;
    const int id = (int)p0[0], c = (int)p0[5];
    if (c < 0 || c >= num_classes) { flag[perm_a[j0]] = 0; return; }
    if (j0 > 0) {
        const float* q = pred + (long long)perm_a[j0 - 1] * 7;
        if ((int)q[0] == id && (int)q[5] == c) return;   // not the head of its group
    }
    const bool ignore_difficult = gstride > 6;
    const int g_begin = (id >= 0 && id < num_images) ? gt_off[id] : 0, g_end = (id >= 0 && id < num_images) ? gt_off[id + 1] : 0;
    for (long long j = j0; j < n; ++j) {
        const unsigned row = perm_a[j];
        const float* p = pred + (long long)row * 7;
        if ((int)p[0] != id || (int)p[5] != c) break;
        const float4 pb = make_float4(p[1], p[2], p[3], p[4]);
        const float parea = area4(pb.x, pb.y, pb.z, pb.w);
        bool have = false;
        int best_g = -1;
        float best = 0.0f;
        for (int g = g_begin; g < g_end; ++g) {
            const float* q = gt + (long long)g * gstride;
            if ((int)q[4] != c) continue;
            const float4 qb = make_float4(q[0], q[1], q[2], q[3]);
            const float v = iou_corner(pb, parea, qb, area4(qb.x, qb.y, qb.z, qb.w));
            if (!have || (v > best && best == best) || (v != v && best == best)) { best = v; best_g = g; }   // first max, NaN propagates
            have = true;
        }
        unsigned char f = 2;                                                        // :54-56, :68: false positive
        if (have && best > thr) {                                                   // :60
            if (!ignore_difficult || gt[(long long)best_g * gstride + 6] == 0.0f) { // :61
                if (!matched[best_g]) { matched[best_g] = 1; f = 1; }              // :62-64
            } else {
                f = 0;                                                              // difficult hit: neither counter moves
            }
        }
        flag[row] = f;
    }
}

struct OpTmax { __device__ __forceinline__ float operator()(float a, float b) const { return tmaxf(a, b); } };

// one workgroup per class: cumulative counts -> precision / recall -> reverse running max -> AP (:77-108)
__global__ void __launch_bounds__(kMapThreads) map_ap_kernel(const unsigned long long* __restrict__ key_b, const unsigned* __restrict__ perm_b, long long n,
                                                             const unsigned char* __restrict__ flag, const int* __restrict__ total_positive, int voc,
                                                             float* prec, float* rec, float* ap_out) {
    __shared__ unsigned long long s_scan_i[17];   // (tp, fp) counts packed as two 32-bit halves
    __shared__ float s_scan_f[17];
    __shared__ long long s_range[2];
    __shared__ int s_cnt[11];
    __shared__ float s_vocp[11];
    __shared__ float s_carry_f;

    const int c = blockIdx.x, tid = threadIdx.x;
    const int tot_i = total_positive[c];
    if (tot_i == 0) {   // class without counted ground truth: not part of the mean (:74-75, :80)
        if (tid == 0) ap_out[c] = __uint_as_float(0x7FC00000u);
        return;
    }
    if (tid < 2) {      // segment of this class in the (class, score) order: lower bounds of c << 32 and (c + 1) << 32
        const unsigned long long want = (unsigned long long)(unsigned)(c + tid) << 32;
        long long lo = 0, hi = n;
        while (lo < hi) {
            const long long mid = (lo + hi) >> 1;
            if (key_b[mid] < want) lo = mid + 1; else hi = mid;
        }
        s_range[tid] = lo;
    }
    if (tid < 11) { s_cnt[tid] = 0; s_vocp[tid] = 0.0f; }
    __syncthreads();
    const long long begin = s_range[0], m = s_range[1] - s_range[0];
    const float tot = (float)tot_i;
    if (m == 0) {       // no prediction of this class: tp = [0], fp = [1] (:81-89) -> AP = 0 either way
        if (tid == 0) ap_out[c] = 0.0f;
        return;
    }
    float thr[11];
#pragma unroll
    for (int k = 0; k < 11; ++k) thr[k] = (float)(0.0 + k * 0.1);   // torch.arange(0, 1.1, .1)
    int my_cnt[11];
#pragma unroll
    for (int k = 0; k < 11; ++k) my_cnt[k] = 0;

    // pass 1, forward: cumulative (tp, fp) -> precision, recall
    int2 carry = make_int2(0, 0);
    struct AddU64 { __device__ __forceinline__ unsigned long long operator()(unsigned long long a, unsigned long long b) const { return a + b; } };
    for (long long base = 0; base < m; base += kMapThreads) {
        const long long i = base + tid;
        unsigned long long v = 0ull;   // tp in the low half, fp in the high half (a class has < 2^31 predictions: no carry between them)
        if (i < m) {
            const unsigned char f = flag[perm_b[begin + i]];
            v = (unsigned long long)(f == 1) | ((unsigned long long)(f == 2) << 32);
        }
        unsigned long long aggv;
        const unsigned long long inclv = block_inclusive_scan(v, AddU64(), s_scan_i, &aggv);
        const int2 incl = make_int2((int)(unsigned)inclv, (int)(unsigned)(inclv >> 32)), agg = make_int2((int)(unsigned)aggv, (int)(unsigned)(aggv >> 32));
        __syncthreads();
        if (i < m) {
            const float tp = (float)(carry.x + incl.x), fp = (float)(carry.y + incl.y);
            const float p = tp / (tp + fp), r = tp / tot;      // :91, :97
            prec[begin + i] = p;
            rec[begin + i] = r;
#pragma unroll
            for (int k = 0; k < 11; ++k) my_cnt[k] += thr[k] > r;
        }
        carry = make_int2(carry.x + agg.x, carry.y + agg.y);
    }
    if (voc) {
#pragma unroll
        for (int k = 0; k < 11; ++k)
            if (my_cnt[k]) atomicAdd(&s_cnt[k], my_cnt[k]);
    }
    if (tid == 0) s_carry_f = 0.0f;   // precision[m] = 0 (:92)
    __syncthreads();

    // pass 2, backward: running maximum (:94-95), then the integral
    float acc = 0.0f;
    const long long chunks = (m + kMapThreads - 1) / kMapThreads;
    for (long long ch = chunks - 1; ch >= 0; --ch) {
        // thread t handles position i = ch * T + (T - 1 - t): scanning t upwards walks i downwards
        const long long i = ch * kMapThreads + (kMapThreads - 1 - tid);
        const float neg_inf = __uint_as_float(0xFF800000u);
        float v = i < m ? prec[begin + i] : neg_inf;
        const float incl = block_inclusive_scan(v, OpTmax(), s_scan_f, (float*)nullptr);
        const float carry_f = s_carry_f;
        __syncthreads();
        const float pm = tmaxf(incl, carry_f);
        if (i < m) {
            if (voc) {
#pragma unroll
                for (int k = 0; k < 11; ++k)
                    if ((long long)s_cnt[k] == i) s_vocp[k] = pm;
            } else {
                const float r = rec[begin + i], r_prev = i > 0 ? rec[begin + i - 1] : 0.0f;
                acc += (r - r_prev) * pm;                         // (recall[1:] - recall[:-1]) . precision (:105-106)
            }
        }
        if (tid == kMapThreads - 1) s_carry_f = pm;               // lowest position of the chunk
        __syncthreads();
    }
    if (voc) {
        if (tid == 0) {
            float s = 0.0f;
            for (int k = 0; k < 11; ++k) s += s_vocp[k];          // indexes == m select precision[m] = 0 (s_vocp stays 0)
            ap_out[c] = s / 11.0f;
        }
    } else {
        // the last term (1 - recall[m-1]) * precision[m] is (finite) * 0
        const float sum = block_sum(acc, s_scan_f);
        if (tid == 0) ap_out[c] = sum;
    }
}

// ---- stable LSD radix sort of (64-bit key, 32-bit value) pairs, 8 bits per pass ---------------------------------------------------
// Pass = three launches.  radix_hist: every tile (a contiguous run of `tile` keys, one workgroup) counts its digits -> table[digit][tile].
// radix_rowscan: one workgroup per digit turns its row into exclusive prefixes (and the row total).  radix_scatter: a tile's four waves
// own four contiguous quarters of it; a key's slot = (keys with a smaller digit anywhere) + (same digit in earlier tiles) + (same digit
// in earlier waves of the tile) + (same digit earlier in the wave's walk), the last found 64 keys at a time by ballot matching -- every
// term follows input order, so the sort is stable (equal scores keep the lower prediction row first, like the oracle's stable sort).
struct SortPass {
    const unsigned long long* kin;
    unsigned long long* kout;
    const unsigned* vin;
    unsigned* vout;
    long long n;
    int shift, tile, nblocks;
    unsigned* table;        // [256][nblocks]
    unsigned* digit_total;  // [256]
};

__global__ void __launch_bounds__(kSortThreads) radix_hist_kernel(SortPass a) {
    __shared__ unsigned s_hist[256];
    s_hist[threadIdx.x] = 0;
    __syncthreads();
    const long long b0 = (long long)blockIdx.x * a.tile, b1 = min(a.n, b0 + a.tile);
    for (long long i = b0 + threadIdx.x; i < b1; i += kSortThreads) atomicAdd(&s_hist[(unsigned)(a.kin[i] >> a.shift) & 255u], 1u);
    __syncthreads();
    a.table[(size_t)threadIdx.x * a.nblocks + blockIdx.x] = s_hist[threadIdx.x];
}

__global__ void __launch_bounds__(kSortThreads) radix_rowscan_kernel(SortPass a) {
    __shared__ unsigned s_tmp[17];
    unsigned* row = a.table + (size_t)blockIdx.x * a.nblocks;
    struct AddU { __device__ __forceinline__ unsigned operator()(unsigned x, unsigned y) const { return x + y; } };
    unsigned carry = 0;
    for (int base = 0; base < a.nblocks; base += kSortThreads) {
        const int i = base + threadIdx.x;
        const unsigned v = i < a.nblocks ? row[i] : 0u;
        unsigned agg;
        const unsigned incl = block_inclusive_scan(v, AddU(), s_tmp, &agg);
        if (i < a.nblocks) row[i] = carry + incl - v;
        carry += agg;
        __syncthreads();
    }
    if (threadIdx.x == 0) a.digit_total[blockIdx.x] = carry;
}

__global__ void __launch_bounds__(kSortThreads) radix_scatter_kernel(SortPass a) {
    __shared__ unsigned s_base[256];        // keys with a smaller digit + same digit in earlier tiles
    __shared__ volatile unsigned s_wcnt[4][256];   // per wave: its digit counts, then its running slots (volatile: lanes of a wave hand
                                                   // the slot of a digit to each other through it from one 64-key round to the next)
    __shared__ unsigned s_tmp[17];
    const int tid = threadIdx.x, lane = lane_id(), wave = tid >> 6;
    struct AddU { __device__ __forceinline__ unsigned operator()(unsigned x, unsigned y) const { return x + y; } };
    {
        const unsigned t = a.digit_total[tid];
        const unsigned incl = block_inclusive_scan(t, AddU(), s_tmp, (unsigned*)nullptr);
        s_base[tid] = incl - t + a.table[(size_t)tid * a.nblocks + blockIdx.x];
    }
    for (int w = 0; w < 4; ++w) s_wcnt[w][tid] = 0;
    __syncthreads();
    const long long b0 = (long long)blockIdx.x * a.tile, b1 = min(a.n, b0 + a.tile);
    const long long quarter = a.tile / 4;
    const long long w0 = b0 + wave * quarter, w1 = min(b1, w0 + quarter);
    for (long long i = w0 + lane; i < w1; i += kWave) atomicAdd(const_cast<unsigned*>(&s_wcnt[wave][(unsigned)(a.kin[i] >> a.shift) & 255u]), 1u);
    __syncthreads();
    {   // thread d: the four waves' starting slots for digit d
        unsigned run = s_base[tid];
        for (int w = 0; w < 4; ++w) { const unsigned c = s_wcnt[w][tid]; s_wcnt[w][tid] = run; run += c; }
    }
    __syncthreads();
    for (long long i0 = w0; i0 < w1; i0 += kWave) {
        const long long i = i0 + lane;
        const bool valid = i < w1;
        const unsigned long long key = valid ? a.kin[i] : 0ull;
        const unsigned d = (unsigned)(key >> a.shift) & 255u;
        unsigned long long m = __ballot(valid);
#pragma unroll
        for (int bit = 0; bit < 8; ++bit) {
            const bool one = (d >> bit) & 1u;
            const unsigned long long bal = __ballot(one);
            m &= one ? bal : ~bal;
        }
        const int rank = __popcll(m & ((1ull << lane) - 1ull)), cnt = __popcll(m);
        if (valid) {
            const unsigned pos = s_wcnt[wave][d] + (unsigned)rank;
            a.kout[pos] = key;
            a.vout[pos] = a.vin[i];
        }
        __builtin_amdgcn_wave_barrier();   // (every lane has read its digit's slot before the group's last lane moves it on)
        if (valid && rank == cnt - 1) s_wcnt[wave][d] = s_wcnt[wave][d] + (unsigned)cnt;
        __builtin_amdgcn_wave_barrier();
    }
}

__global__ void map_mean_kernel(const float* __restrict__ ap, const int* __restrict__ total_positive, int num_classes, double* map_out) {
    if (threadIdx.x || blockIdx.x) return;
    double s = 0.0;
    int cnt = 0;
    for (int c = 0; c < num_classes; ++c)
        if (total_positive[c]) { s += (double)ap[c]; ++cnt; }   // :111 (python floats)
    *map_out = cnt ? s / cnt : (double)__uint_as_float(0x7FC00000u);
}

}  // namespace ssdk

using namespace ssdk;

// stable sort of n (key, value) pairs on key bits [0, end_bit); the result lands in kb / vb.  ka / va are overwritten.
static int radix_sort_pairs(unsigned long long* ka, unsigned long long* kb, unsigned* va, unsigned* vb, long long n, int end_bit, unsigned* table,
                            unsigned* digit_total, hipStream_t s) {
    SortPass a;
    a.n = n;
    long long tile = 4096;
    while ((n + tile - 1) / tile > kSortMaxBlocks) tile *= 2;
    a.tile = (int)tile;
    a.nblocks = (int)((n + tile - 1) / tile);
    a.table = table;
    a.digit_total = digit_total;
    const int digits = (end_bit + 7) / 8;
    // pass p reads a (p even) or b (p odd) and writes the other: an odd number of passes ends in kb / vb.  With an even number of digits
    // the most significant one is sorted once more -- a stable sort by a digit the data is already ordered by is a copy.
    const int passes = (digits & 1) ? digits : digits + 1;
    for (int p = 0; p < passes; ++p) {
        a.shift = 8 * (p < digits ? p : digits - 1);
        a.kin = (p & 1) ? kb : ka; a.kout = (p & 1) ? ka : kb;
        a.vin = (p & 1) ? vb : va; a.vout = (p & 1) ? va : vb;
        hipLaunchKernelGGL(radix_hist_kernel, dim3(a.nblocks), dim3(kSortThreads), 0, s, a);
        SSDK_CHECK_LAUNCH("radix_hist_kernel");
        hipLaunchKernelGGL(radix_rowscan_kernel, dim3(256), dim3(kSortThreads), 0, s, a);
        SSDK_CHECK_LAUNCH("radix_rowscan_kernel");
        hipLaunchKernelGGL(radix_scatter_kernel, dim3(a.nblocks), dim3(kSortThreads), 0, s, a);
        SSDK_CHECK_LAUNCH("radix_scatter_kernel");
    }
    return SSDK_OK;
}

extern "C" size_t ssdk_mean_average_precision_workspace_bytes(long long n_pred, long long total_gt, int num_classes) {
    if (n_pred < 0 || total_gt < 0 || num_classes <= 0) return 0;
    return carve_map(nullptr, n_pred, total_gt, num_classes).total;
}

extern "C" int ssdk_mean_average_precision(const float* predictions, long long n_pred, const float* gt_rows, int gt_stride,
                                           const int* gt_offsets, int num_images, long long total_gt, int num_classes,
                                           float iou_threshold, int voc, float* ap_out, double* map_out, void* workspace,
                                           size_t workspace_bytes, void* stream) {
    SSDK_REQUIRE(n_pred >= 0 && n_pred < (1LL << 31) && total_gt >= 0 && total_gt < (1LL << 31) && num_images >= 0 && num_classes > 0 && num_classes < (1 << 20),
                 SSDK_E_INVALID, "ssdk_mean_average_precision: n_pred=%lld total_gt=%lld num_images=%d num_classes=%d", n_pred, total_gt, num_images, num_classes);
    SSDK_REQUIRE(gt_stride >= 5 && gt_offsets && ap_out && map_out && (n_pred == 0 || predictions) && (total_gt == 0 || gt_rows), SSDK_E_INVALID,
                 "ssdk_mean_average_precision: null pointer or gt_stride=%d < 5", gt_stride);
    MapWs w = carve_map(workspace, n_pred, total_gt, num_classes);
    SSDK_REQUIRE(workspace && workspace_bytes >= w.total, SSDK_E_WORKSPACE, "ssdk_mean_average_precision: workspace %zu < %zu bytes", workspace_bytes, w.total);
    hipStream_t s = (hipStream_t)stream;
    SSDK_CHECK_HIP(zero_async(w.total_positive, sizeof(int) * (size_t)num_classes, s));
    SSDK_CHECK_HIP(zero_async(w.matched, (size_t)(total_gt > 0 ? total_gt : 1), s));
    const long long work = n_pred > total_gt ? n_pred : total_gt;
    if (work > 0) {
        hipLaunchKernelGGL(map_prepare_kernel, dim3((unsigned)cdiv((int)work, 256)), dim3(256), 0, s, predictions, n_pred, gt_rows, gt_stride, total_gt, num_classes,
                           w.key_a, w.val_a, w.total_positive);
        SSDK_CHECK_LAUNCH("map_prepare_kernel");
    }
    if (n_pred > 0) {
        const int n = (int)n_pred;
        int class_bits = 1;
        while ((1 << class_bits) <= num_classes) ++class_bits;   // keys go up to num_classes << 32 (foreign classes)
        // (class, score desc): stable, so equal scores keep the lower row first
        int rc = radix_sort_pairs(w.key_a, w.key_b, w.val_a, w.perm_b, n_pred, 32 + class_bits, w.table, w.digit_total, s);
        if (rc) return rc;
        hipLaunchKernelGGL(map_image_key_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, s, predictions, n_pred, w.perm_b, w.key_a, w.val_a);
        SSDK_CHECK_LAUNCH("map_image_key_kernel");
        // stable sort by image on top: (image, class, score desc); val_b = prediction rows in that order
        rc = radix_sort_pairs(w.key_a, w.key_c, w.val_a, w.val_b, n_pred, 32, w.table, w.digit_total, s);
        if (rc) return rc;
        hipLaunchKernelGGL(map_match_kernel, dim3((unsigned)cdiv(n, 128)), dim3(128), 0, s, predictions, n_pred, w.val_b, gt_rows, gt_stride, gt_offsets, num_images,
                           num_classes, iou_threshold, w.matched, w.flag);
        SSDK_CHECK_LAUNCH("map_match_kernel");
    }
    hipLaunchKernelGGL(map_ap_kernel, dim3((unsigned)num_classes), dim3(kMapThreads), 0, s, w.key_b, w.perm_b, n_pred, w.flag, w.total_positive, voc, w.prec, w.rec, ap_out);
    SSDK_CHECK_LAUNCH("map_ap_kernel");
    hipLaunchKernelGGL(map_mean_kernel, dim3(1), dim3(64), 0, s, ap_out, w.total_positive, num_classes, map_out);
    SSDK_CHECK_LAUNCH("map_mean_kernel");
    return SSDK_OK;
}

// ---- device-resident input side: mixup (bf/core/batch_container.py:25-45) ----------------------------------------------
namespace ssdk {

// HBM-bound: 2 reads + 1 write of 16 B per lane for mixed images, a plain copy for the others
__global__ void mixup_images_kernel(const float4* __restrict__ in, float4* __restrict__ out, long long per_image4, const int* __restrict__ index,
                                    const unsigned char* __restrict__ roll, float lam, float oml) {
    const int b = blockIdx.y;
    const bool mix = roll[b] != 0;
    const float4* a = in + (long long)b * per_image4;
    const float4* o = in + (long long)index[b] * per_image4;
    float4* d = out + (long long)b * per_image4;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < per_image4; i += (long long)gridDim.x * blockDim.x) {
        float4 v = a[i];
        if (mix) {
            const float4 w = o[i];   // (this TU is built -ffp-contract=off: multiply and add round separately, like the two torch ops)
            v = make_float4(lam * v.x + oml * w.x, lam * v.y + oml * w.y, lam * v.z + oml * w.z, lam * v.w + oml * w.w);
        }
        d[i] = v;
    }
}
__global__ void mixup_images_tail_kernel(const float* __restrict__ in, float* __restrict__ out, long long per_image, long long done, const int* __restrict__ index,
                                         const unsigned char* __restrict__ roll, float lam, float oml) {
    const int b = blockIdx.y;
    const long long i = done + (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= per_image) return;
    const float v = in[(long long)b * per_image + i];
    out[(long long)b * per_image + i] = roll[b] ? lam * v + oml * in[(long long)index[b] * per_image + i] : v;
}

// one workgroup: offsets by a serial scan over the (small) batch, then the row copies
__global__ void mixup_gt_kernel(const float* __restrict__ rows_in, int stride, const int* __restrict__ off_in, int batch, const int* __restrict__ index,
                                const unsigned char* __restrict__ roll, float lam, float oml, float* __restrict__ rows_out, int* __restrict__ off_out) {
    extern __shared__ int s_off[];
    if (threadIdx.x == 0) {
        int acc = 0;
        for (int b = 0; b < batch; ++b) {
            s_off[b] = acc;
            acc += (off_in[b + 1] - off_in[b]) + (roll[b] ? off_in[index[b] + 1] - off_in[index[b]] : 0);
        }
        s_off[batch] = acc;
    }
    __syncthreads();
    for (int b = threadIdx.x; b <= batch; b += blockDim.x) off_out[b] = s_off[b];
    for (int b = 0; b < batch; ++b) {
        const int own = off_in[b + 1] - off_in[b];
        const int other = roll[b] ? off_in[index[b] + 1] - off_in[index[b]] : 0;
        const int total = (own + other) * stride;
        for (int e = threadIdx.x; e < total; e += blockDim.x) {
            const int r = e / stride, c = e % stride;
            const bool second = r >= own;
            float v = second ? rows_in[(long long)(off_in[index[b]] + r - own) * stride + c] : rows_in[(long long)(off_in[b] + r) * stride + c];
            if (c == 5 && roll[b]) v *= second ? oml : lam;   // :38, :40 (SCORE_INDEX)
            rows_out[(long long)(s_off[b] + r) * stride + c] = v;
        }
    }
}

}  // namespace ssdk

extern "C" int ssdk_mixup_images(const float* in, float* out, int batch, long long per_image, const int* index, const unsigned char* roll,
                                 double lam, void* stream) {
    SSDK_REQUIRE(in && out && index && roll && batch > 0 && batch <= 65535 && per_image > 0 && in != out, SSDK_E_INVALID,
                 "ssdk_mixup_images: batch=%d per_image=%lld (out-of-place only)", batch, per_image);
    hipStream_t s = (hipStream_t)stream;
    const float lam_f = (float)lam, oml_f = (float)(1.0 - lam);
    const bool vec = (per_image % 4 == 0) && (((uintptr_t)in | (uintptr_t)out) & 15) == 0;
    if (vec) {
        const long long n4 = per_image / 4;
        const unsigned gx = (unsigned)(n4 / 256 / 4 > 0 ? (n4 / 256 / 4 < 1024 ? n4 / 256 / 4 : 1024) : 1);
        hipLaunchKernelGGL(mixup_images_kernel, dim3(gx, (unsigned)batch), dim3(256), 0, s, reinterpret_cast<const float4*>(in), reinterpret_cast<float4*>(out), n4, index,
                           roll, lam_f, oml_f);
    } else {
        hipLaunchKernelGGL(mixup_images_tail_kernel, dim3((unsigned)((per_image + 255) / 256), (unsigned)batch), dim3(256), 0, s, in, out, per_image, 0LL, index, roll,
                           lam_f, oml_f);
    }
    SSDK_CHECK_LAUNCH("mixup_images_kernel");
    return SSDK_OK;
}

extern "C" int ssdk_mixup_ground_truth(const float* rows_in, int gt_stride, const int* offsets_in, int batch, const int* index,
                                       const unsigned char* roll, double lam, float* rows_out, int* offsets_out, void* stream) {
    SSDK_REQUIRE(offsets_in && index && roll && rows_out && offsets_out && batch > 0 && batch <= 8192 && gt_stride >= 6, SSDK_E_INVALID,
                 "ssdk_mixup_ground_truth: batch=%d gt_stride=%d", batch, gt_stride);
    hipLaunchKernelGGL(mixup_gt_kernel, dim3(1), dim3(256), sizeof(int) * (size_t)(batch + 1), (hipStream_t)stream, rows_in, gt_stride, offsets_in, batch, index, roll,
                       (float)lam, (float)(1.0 - lam), rows_out, offsets_out);
    SSDK_CHECK_LAUNCH("mixup_gt_kernel");
    return SSDK_OK;
}
