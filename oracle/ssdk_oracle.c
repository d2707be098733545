/*
 * ssdk_oracle.c -- CPU restatement of the reference's detection hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product package may import, link or call this file; it is used
 * by tests/, by __graft_entry__.smoke() and by bench.py's cpu_baseline leg as the CHECKER / baseline.
 *
 * Every function restates, in scalar fp32 C (no FMA contraction: build with -ffp-contract=off), the
 * arithmetic of the reference function it cites (paths relative to /root/reference).  Parity pinning:
 * tests/test_oracle_golden.py checks this file against golden vectors produced by running the reference
 * itself (tools/gen_golden.py).  Hard NMS is the one exception: the reference delegates it to
 * torchvision.ops.nms (bf/utils/box_utils.py:193; torchvision>=0.3.0, not vendored, not installed), so
 * orc_nms_hard restates torchvision's documented contract -- "parity unpinned" for that function only.
 *
 * Tie rules where the reference leaves order to an unstable sort/topk (documented in DESIGN.md):
 * equal keys are ordered by ascending index (i.e. a stable descending sort).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_NOT_MATCHED (-2) /* detection/matcher.py:4 */
#define ORC_IGNORE (-1)      /* detection/matcher.py:5 */

int orc_version(void) { return 1; }

int orc_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
/* bench.py's cpu_baseline probes the thread count (an oversubscribed OpenMP pool made the oracle leg wander 16x between boxes) */
void orc_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}


/* torch.max / torch.min propagate NaN */
static inline float tmaxf(float a, float b) { return (a > b || a != a) ? a : b; }
static inline float tminf(float a, float b) { return (a < b || a != a) ? a : b; }
/* Tensor.clamp_(0): NaN stays NaN */
static inline float clamp0(float v) { return v < 0.0f ? 0.0f : v; }

/* bf/utils/box_utils.py:16-23  to_corners: cat([c - wh/2, c + wh/2]) */
void orc_to_corners(const float* box, float* out, int64_t n) {
    for (int64_t i = 0; i < n; ++i) {
        const float cx = box[4 * i], cy = box[4 * i + 1], w = box[4 * i + 2], h = box[4 * i + 3];
        out[4 * i + 0] = cx - w / 2.0f;
        out[4 * i + 1] = cy - h / 2.0f;
        out[4 * i + 2] = cx + w / 2.0f;
        out[4 * i + 3] = cy + h / 2.0f;
    }
}

/* bf/utils/box_utils.py:25-36  to_centroids(inplace=True): wh = max - min; c = min + wh/2 */
void orc_to_centroids_inplace(float* box, int64_t n) {
    for (int64_t i = 0; i < n; ++i) {
        box[4 * i + 2] -= box[4 * i + 0];
        box[4 * i + 3] -= box[4 * i + 1];
        box[4 * i + 0] += box[4 * i + 2] / 2.0f;
        box[4 * i + 1] += box[4 * i + 3] / 2.0f;
    }
}

/* bf/utils/box_utils.py:38-46  area */
static inline float area4(float x1, float y1, float x2, float y2) { return clamp0(x2 - x1) * clamp0(y2 - y1); }

/* bf/utils/box_utils.py:49-80 (intersection, cartesian) + :83-101 (iou) for one pair of corner boxes */
static inline float iou_pair(const float* a, float area_a, const float* b, float area_b) {
    const float ix1 = tmaxf(a[0], b[0]), iy1 = tmaxf(a[1], b[1]);
    const float ix2 = tminf(a[2], b[2]), iy2 = tminf(a[3], b[3]);
    const float inter = area4(ix1, iy1, ix2, iy2);
    return inter / (area_a + area_b - inter);
}

/* iou(gt[G,4 @ stride], corner_anchors[A,4]) -> [G,A] */
void orc_iou(const float* gt, int G, int gstride, const float* corner, int64_t A, float* out) {
    for (int g = 0; g < G; ++g) {
        const float* a = gt + (int64_t)g * gstride;
        const float area_a = area4(a[0], a[1], a[2], a[3]);
        for (int64_t j = 0; j < A; ++j) {
            const float* b = corner + 4 * j;
            out[(int64_t)g * A + j] = iou_pair(a, area_a, b, area4(b[0], b[1], b[2], b[3]));
        }
    }
}

/* detection/matcher.py:33-56  match_per_prediction on a materialised weight matrix */
void orc_match_per_prediction(const float* w, int G, int64_t A, float matched_thr, float unmatched_thr,
                              int force_match, int64_t* box_idx) {
    for (int64_t j = 0; j < A; ++j) { /* weights.max(dim=0): first max wins, NaN propagates */
        float best = w[j];
        int64_t bi = 0;
        for (int g = 1; g < G; ++g) {
            const float v = w[(int64_t)g * A + j];
            if (v > best || (v != v && best == best)) { best = v; bi = g; }
        }
        const int below_matched = best < matched_thr, below_unmatched = best < unmatched_thr;
        if (below_unmatched) bi = ORC_NOT_MATCHED;
        else if (below_matched) bi = ORC_IGNORE;
        box_idx[j] = bi;
    }
    if (force_match) { /* :52-54 anchor_idx = argmax(dim=1); box_idx[anchor_idx] = arange(G): last writer wins */
        for (int g = 0; g < G; ++g) {
            float best = w[(int64_t)g * A];
            int64_t bj = 0;
            for (int64_t j = 1; j < A; ++j) {
                const float v = w[(int64_t)g * A + j];
                if (v > best || (v != v && best == best)) { best = v; bj = j; }
            }
            box_idx[bj] = g;
        }
    }
}

/*
 * detection/target_assigner.py:22-63  TargetAssigner.encode_ground_truth
 *   gt rows: [x1,y1,x2,y2,cls,score(,..)] at `gstride` floats; image i owns rows gt_off[i]..gt_off[i+1].
 *   anchors: [A,4] centroid form.  target: [B,A,6].  box_idx (optional): int32 [B,A], -2 for empty images.
 */
void orc_encode_ground_truth(const float* gt, const int32_t* gt_off, int B, int gstride, const float* anchors,
                             int64_t A, float matched_thr, float unmatched_thr, float* target, int32_t* box_idx_out) {
    float* corner = (float*)malloc(sizeof(float) * 4 * A);
    orc_to_corners(anchors, corner, A);
#pragma omp parallel for schedule(dynamic, 1)
    for (int i = 0; i < B; ++i) {
        float* t = target + (int64_t)i * A * 6;
        for (int64_t j = 0; j < A; ++j) { /* :38-40 zeros, class = NEGATIVE_CLASS, score = 1 */
            t[6 * j + 0] = t[6 * j + 1] = t[6 * j + 2] = t[6 * j + 3] = 0.0f;
            t[6 * j + 4] = 0.0f;
            t[6 * j + 5] = 1.0f;
        }
        const int G = gt_off[i + 1] - gt_off[i];
        if (box_idx_out) for (int64_t j = 0; j < A; ++j) box_idx_out[(int64_t)i * A + j] = ORC_NOT_MATCHED;
        if (G == 0) continue; /* :43-44 */
        const float* g0 = gt + (int64_t)gt_off[i] * gstride;
        float* w = (float*)malloc(sizeof(float) * (size_t)G * A);
        int64_t* idx = (int64_t*)malloc(sizeof(int64_t) * A);
        orc_iou(g0, G, gstride, corner, A, w);                                     /* :47 */
        orc_match_per_prediction(w, G, A, matched_thr, unmatched_thr, 1, idx);     /* :49 */
        for (int64_t j = 0; j < A; ++j) {
            if (idx[j] >= 0) { /* :50-54 */
                const float* r = g0 + idx[j] * gstride;
                t[6 * j + 0] = r[0]; t[6 * j + 1] = r[1]; t[6 * j + 2] = r[2]; t[6 * j + 3] = r[3];
                t[6 * j + 4] = r[4];
                t[6 * j + 5] = r[5];
            } else if (idx[j] == ORC_IGNORE) { /* :56-58 */
                t[6 * j + 4] = -1.0f;
                t[6 * j + 5] = -1.0f;
            }
            if (box_idx_out) box_idx_out[(int64_t)i * A + j] = (int32_t)idx[j];
        }
        free(w);
        free(idx);
    }
    free(corner);
}

/* detection/box_coder.py:22-30  encode_box(inplace=True) on centroid boxes [n,4] against priors [n,4] */
static inline void encode_inplace(float* b, const float* p, float xy_scale, float wh_scale, float eps) {
    b[0] -= p[0]; b[1] -= p[1];
    b[0] /= p[2]; b[1] /= p[3];
    b[0] *= xy_scale; b[1] *= xy_scale;
    b[2] /= p[2]; b[3] /= p[3];
    b[2] += eps; b[3] += eps;
    b[2] = logf(b[2]); b[3] = logf(b[3]);
    b[2] *= wh_scale; b[3] *= wh_scale;
}

void orc_encode_box_inplace(float* boxes, const float* priors, int B, int64_t A, float xy_scale, float wh_scale, float eps) {
    for (int i = 0; i < B; ++i)
        for (int64_t j = 0; j < A; ++j) encode_inplace(boxes + ((int64_t)i * A + j) * 4, priors + 4 * j, xy_scale, wh_scale, eps);
}

/* detection/box_coder.py:32-34  encode_box(inplace=False): eps added BEFORE the divide */
void orc_encode_box(const float* boxes, const float* priors, float* out, int B, int64_t A, float xy_scale, float wh_scale, float eps) {
    for (int i = 0; i < B; ++i)
        for (int64_t j = 0; j < A; ++j) {
            const float* b = boxes + ((int64_t)i * A + j) * 4;
            const float* p = priors + 4 * j;
            float* o = out + ((int64_t)i * A + j) * 4;
            o[0] = (b[0] - p[0]) / p[2] * xy_scale;
            o[1] = (b[1] - p[1]) / p[3] * xy_scale;
            o[2] = logf((b[2] + eps) / p[2]) * wh_scale;
            o[3] = logf((b[3] + eps) / p[3]) * wh_scale;
        }
}

/* detection/box_coder.py:55-57  decode_box (not in place) -> centroid boxes */
static inline void decode_one(const float* t, const float* p, float xy_scale, float wh_scale, float* o) {
    o[0] = p[0] + p[2] * t[0] / xy_scale;
    o[1] = p[1] + p[3] * t[1] / xy_scale;
    o[2] = p[2] * expf(t[2] / wh_scale);
    o[3] = p[3] * expf(t[3] / wh_scale);
}

void orc_decode_box(const float* locs, const float* priors, float* out, int B, int64_t A, float xy_scale, float wh_scale) {
    for (int i = 0; i < B; ++i)
        for (int64_t j = 0; j < A; ++j)
            decode_one(locs + ((int64_t)i * A + j) * 4, priors + 4 * j, xy_scale, wh_scale, out + ((int64_t)i * A + j) * 4);
}

/* F.log_softmax row statistics: returns max and log(sum exp(x - max)) */
static inline void row_lse(const float* x, int C, float* m_out, float* logsum_out) {
    float m = x[0];
    for (int c = 1; c < C; ++c) m = tmaxf(m, x[c]);
    float s = 0.0f;
    for (int c = 0; c < C; ++c) s += expf(x[c] - m);
    *m_out = m;
    *logsum_out = logf(s);
}

typedef struct { float v; int32_t i; } orc_kv;
static int cmp_desc_stable(const void* a, const void* b) {
    const orc_kv* x = (const orc_kv*)a; const orc_kv* y = (const orc_kv*)b;
    if (x->v > y->v) return -1;
    if (x->v < y->v) return 1;
    return (x->i > y->i) - (x->i < y->i);
}

/*
 * detection/sampler.py:12-25  hard_negative_mining
 *   scores [B,A,C]; cls = target[...,4] as integer (-1 ignore, 0 negative, >0 positive); mask uint8 [B,A].
 *   ratio may be fractional (python number * int64 tensor); min_neg clamps from below.
 *   bgloss_out (optional) [B,A]: -log_softmax[...,0], -inf where not negative (:16, :22).
 */
void orc_hard_negative_mining(const float* scores, const float* target, int B, int64_t A, int C, double ratio,
                              int64_t min_neg, uint8_t* mask, float* bgloss_out) {
#pragma omp parallel for schedule(dynamic, 1)
    for (int i = 0; i < B; ++i) {
        orc_kv* kv = (orc_kv*)malloc(sizeof(orc_kv) * A);
        int64_t npos = 0, nneg = 0;
        for (int64_t j = 0; j < A; ++j) {
            const float* x = scores + ((int64_t)i * A + j) * C;
            const int64_t cls = (int64_t)target[((int64_t)i * A + j) * 6 + 4];
            float m, ls;
            row_lse(x, C, &m, &ls);
            float loss = -((x[0] - m) - ls);
            const int neg = cls == 0, pos = cls != 0 && cls != -1;
            npos += pos; nneg += neg;
            if (!neg) loss = -INFINITY;
            kv[j].v = loss; kv[j].i = (int32_t)j;
            mask[(int64_t)i * A + j] = (uint8_t)pos;
            if (bgloss_out) bgloss_out[(int64_t)i * A + j] = loss;
        }
        /* :20 min(clamp(P*ratio, min=min_neg), #neg); P*ratio is integer when ratio is */
        double want = (double)npos * ratio;
        if (want < (double)min_neg) want = (double)min_neg;
        int64_t n = (int64_t)ceil(want); /* rank < want: ceil(want) ranks qualify when want is fractional */
        if (n > nneg) n = nneg;
        qsort(kv, (size_t)A, sizeof(orc_kv), cmp_desc_stable); /* rank = argsort(argsort(desc)) ; rank < n */
        for (int64_t r = 0; r < n; ++r) mask[(int64_t)i * A + kv[r].i] = 1;
        free(kv);
    }
}

/* torch smooth_l1 (beta): z<beta ? 0.5 z^2/beta : z - 0.5 beta  (aten/src/ATen/native/cpu/PointwiseOpsKernel) */
static inline float smooth_l1(float a, float b, float beta) {
    const float z = fabsf(a - b);
    return z < beta ? 0.5f * z * z / beta : z - 0.5f * beta;
}
static inline float smooth_l1_grad(float a, float b, float beta) {
    const float d = a - b;
    if (d <= -beta) return -1.0f;
    if (d >= beta) return 1.0f;
    return d / beta;
}

/*
 * detection/losses/multibox_loss.py:35-94  MultiboxLoss.forward, CrossEntropyLoss(sum, ignore_index=-1) +
 * SmoothL1Loss(sum) branch, with `sampled` supplied by the sampler (:58).  Also the backward pass that
 * autograd derives for it (upstream gradient 1 on `loss`).
 *   MUTATES target[...,0:4] exactly as :81-82 do (to_centroids + encode_box in place, every anchor).
 *   out3 = {loss, class_loss, loc_loss}; dscores [B,A,C] / dlocs [B,A,4] optional (dense, zero off-sample).
 */
void orc_multibox_loss_ce(const float* scores, const float* locs, const float* anchors, float* target,
                          const uint8_t* sampled, int B, int64_t A, int C, float cls_w, float loc_w,
                          float xy_scale, float wh_scale, float eps, float beta, double* out3, float* dscores, float* dlocs) {
    double cls_sum = 0.0, loc_sum = 0.0;
    int64_t npos = 0;
    const int64_t N = (int64_t)B * A;
#pragma omp parallel for reduction(+ : cls_sum, loc_sum, npos) schedule(static)
    for (int64_t r = 0; r < N; ++r) {
        const int64_t j = r % A;
        float* t = target + r * 6;
        const int64_t cls = (int64_t)t[4];
        const int pos = cls != 0 && cls != -1;
        npos += pos;
        if (sampled[r] && cls != -1) { /* :60-75, ignore_index = -1 */
            const float* x = scores + r * C;
            float m, ls;
            row_lse(x, C, &m, &ls);
            cls_sum += (double)(-((x[cls] - m) - ls));
        }
        orc_to_centroids_inplace(t, 1);                                  /* :81 */
        encode_inplace(t, anchors + 4 * j, xy_scale, wh_scale, eps);      /* :82 */
        if (pos) {                                                       /* :84-86 */
            const float* l = locs + r * 4;
            for (int k = 0; k < 4; ++k) loc_sum += (double)smooth_l1(l[k], t[k], beta);
        }
    }
    const float divider = (float)(npos < 1 ? 1 : npos);                   /* :88 */
    const float class_loss = (float)cls_sum * cls_w / divider;            /* :90 */
    const float loc_loss = (float)loc_sum * loc_w / divider;              /* :89 */
    out3[0] = (double)(class_loss + loc_loss);
    out3[1] = class_loss;
    out3[2] = loc_loss;
    if (!dscores && !dlocs) return;
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < N; ++r) {
        const float* t = target + r * 6;
        const int64_t cls = (int64_t)t[4];
        const int pos = cls != 0 && cls != -1;
        if (dscores) {
            float* g = dscores + r * C;
            if (sampled[r] && cls != -1) {
                const float* x = scores + r * C;
                float m, ls;
                row_lse(x, C, &m, &ls);
                for (int c = 0; c < C; ++c) g[c] = (expf((x[c] - m) - ls) - (c == cls ? 1.0f : 0.0f)) * cls_w / divider;
            } else {
                memset(g, 0, sizeof(float) * C);
            }
        }
        if (dlocs) {
            float* g = dlocs + r * 4;
            const float* l = locs + r * 4;
            for (int k = 0; k < 4; ++k) g[k] = pos ? smooth_l1_grad(l[k], t[k], beta) * loc_w / divider : 0.0f;
        }
    }
}

static inline float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

/*
 * MultiboxLoss.forward with SigmoidFocalLoss (bf/modules/losses.py:34-54; MULTICLASS one-hot target built at
 * multibox_loss.py:64-67: class_target[row, cls-1] = target_score) + SmoothL1Loss(sum).
 *   reduce_mean != 0 reproduces the reference's constructor quirk (misc_utils.py:22-29 drops reduction='sum'
 *   for classes whose __init__ takes **kwargs, so the focal loss runs with reduction='mean' over sampled rows).
 */
void orc_multibox_loss_focal(const float* scores, const float* locs, const float* anchors, float* target,
                             const uint8_t* sampled, int B, int64_t A, int C, float gamma, float alpha, int reduce_mean,
                             float cls_w, float loc_w, float xy_scale, float wh_scale, float eps, float beta,
                             double* out3, float* dscores, float* dlocs) {
    double cls_sum = 0.0, loc_sum = 0.0;
    int64_t npos = 0, nrows = 0;
    const int64_t N = (int64_t)B * A;
#pragma omp parallel for reduction(+ : cls_sum, loc_sum, npos, nrows) schedule(static)
    for (int64_t r = 0; r < N; ++r) {
        const int64_t j = r % A;
        float* t = target + r * 6;
        const int64_t cls = (int64_t)t[4];
        const float tscore = t[5];
        const int pos = cls != 0 && cls != -1;
        npos += pos;
        if (sampled[r]) {
            nrows += 1;
            const float* x = scores + r * C;
            float row = 0.0f;
            for (int c = 0; c < C; ++c) {
                const float tg = (pos && c == cls - 1) ? tscore : 0.0f;
                const float aw = tg * alpha + (1.0f - tg) * (1.0f - alpha);
                float pb = sigmoidf_(x[c]);
                pb = pb * tg + (1.0f - pb) * (1.0f - tg);
                const float ce = fmaxf(x[c], 0.0f) - x[c] * tg + log1pf(expf(-fabsf(x[c])));
                row += aw * powf(1.0f - pb, gamma) * ce;
            }
            cls_sum += (double)row;
        }
        orc_to_centroids_inplace(t, 1);
        encode_inplace(t, anchors + 4 * j, xy_scale, wh_scale, eps);
        if (pos) {
            const float* l = locs + r * 4;
            for (int k = 0; k < 4; ++k) loc_sum += (double)smooth_l1(l[k], t[k], beta);
        }
    }
    const float divider = (float)(npos < 1 ? 1 : npos);
    float red = (float)cls_sum;
    const float mean_div = reduce_mean ? (float)nrows : 1.0f; /* mean over an empty set is NaN, as in torch */
    red = red / mean_div;
    const float class_loss = red * cls_w / divider;
    const float loc_loss = (float)loc_sum * loc_w / divider;
    out3[0] = (double)(class_loss + loc_loss);
    out3[1] = class_loss;
    out3[2] = loc_loss;
    if (!dscores && !dlocs) return;
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < N; ++r) {
        const float* t = target + r * 6;
        const int64_t cls = (int64_t)t[4];
        const float tscore = t[5];
        const int pos = cls != 0 && cls != -1;
        if (dscores) {
            float* g = dscores + r * C;
            if (sampled[r]) {
                const float* x = scores + r * C;
                for (int c = 0; c < C; ++c) {
                    const float tg = (pos && c == cls - 1) ? tscore : 0.0f;
                    const float aw = tg * alpha + (1.0f - tg) * (1.0f - alpha);
                    const float s = sigmoidf_(x[c]);
                    const float pb = s * tg + (1.0f - s) * (1.0f - tg);
                    const float ce = fmaxf(x[c], 0.0f) - x[c] * tg + log1pf(expf(-fabsf(x[c])));
                    const float om = 1.0f - pb;
                    const float dpb = s * (1.0f - s) * (2.0f * tg - 1.0f);
                    const float d = aw * (-gamma * powf(om, gamma - 1.0f) * dpb * ce + powf(om, gamma) * (s - tg));
                    g[c] = d / mean_div * cls_w / divider;
                }
            } else {
                memset(g, 0, sizeof(float) * C);
            }
        }
        if (dlocs) {
            float* g = dlocs + r * 4;
            const float* l = locs + r * 4;
            for (int k = 0; k < 4; ++k) g[k] = pos ? smooth_l1_grad(l[k], t[k], beta) * loc_w / divider : 0.0f;
        }
    }
}

/* ---- postprocess ------------------------------------------------------------------------------------------ */

/*
 * Hard NMS per torchvision.ops.nms's documented contract (PARITY UNPINNED: torchvision is not vendored in
 * /root/reference; call site bf/utils/box_utils.py:193).  boxes corner form [n,4]; order = descending score,
 * ties by ascending index; keeps a box unless an earlier kept box has IoU > thr with it.
 * Returns the number kept; picked[] receives indices into boxes in kept order.
 */
int orc_nms_hard(const float* boxes, const float* scores, int n, float thr, int32_t* picked) {
    orc_kv* kv = (orc_kv*)malloc(sizeof(orc_kv) * (size_t)(n > 0 ? n : 1));
    uint8_t* dead = (uint8_t*)calloc((size_t)(n > 0 ? n : 1), 1);
    for (int i = 0; i < n; ++i) { kv[i].v = scores[i]; kv[i].i = i; }
    qsort(kv, (size_t)n, sizeof(orc_kv), cmp_desc_stable);
    int k = 0;
    for (int a = 0; a < n; ++a) {
        if (dead[a]) continue;
        const float* ba = boxes + 4 * kv[a].i;
        const float area_a = (ba[2] - ba[0]) * (ba[3] - ba[1]);
        picked[k++] = kv[a].i;
        for (int b = a + 1; b < n; ++b) {
            if (dead[b]) continue;
            const float* bb = boxes + 4 * kv[b].i;
            const float area_b = (bb[2] - bb[0]) * (bb[3] - bb[1]);
            const float iw = clamp0(tminf(ba[2], bb[2]) - tmaxf(ba[0], bb[0]));
            const float ih = clamp0(tminf(ba[3], bb[3]) - tmaxf(ba[1], bb[1]));
            const float inter = iw * ih;
            if (inter / (area_a + area_b - inter) > thr) dead[b] = 1;
        }
    }
    free(kv);
    free(dead);
    return k;
}

/*
 * bf/utils/box_utils.py:145-163  _soft_nms (gaussian).  Returns count; picked[] = indices in pick order.
 * The loop condition `mask.nonzero().sum()` (:151) is the SUM OF INDICES of live boxes, so the loop also stops
 * when the only live box is index 0 -- reproduced here.
 */
int orc_nms_soft(const float* boxes, const float* scores, int n, float score_thr, float sigma, int32_t* picked) {
    float* sc = (float*)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
    uint8_t* mask = (uint8_t*)malloc((size_t)(n > 0 ? n : 1));
    memcpy(sc, scores, sizeof(float) * (size_t)n);
    for (int i = 0; i < n; ++i) mask[i] = scores[i] > score_thr; /* :147 */
    int k = 0;
    for (;;) {
        /* :151 the mask tested here is the one computed BEFORE the previous iteration's decay (:156) */
        int64_t idxsum = 0;
        for (int i = 0; i < n; ++i) if (mask[i]) idxsum += i;
        if (idxsum == 0) break;
        /* :152 argmax: first max; torch.argmax ranks NaN above everything (first NaN wins) -- a score turns NaN when a degenerate box
         * (zero area) meets the picked box with zero intersection: 0 / 0 in :158 (tests/golden/box_utils.npz has four such boxes) */
        int best = 0;
        for (int i = 1; i < n; ++i) {
            const int best_nan = sc[best] != sc[best];
            if (best_nan) break;
            if (sc[i] != sc[i] || sc[i] > sc[best]) best = i;
        }
        sc[best] = 0.0f;
        picked[k++] = best;
        for (int i = 0; i < n; ++i) mask[i] = sc[i] > score_thr; /* :156 */
        const float* ba = boxes + 4 * best;
        const float area_a = area4(ba[0], ba[1], ba[2], ba[3]);
        for (int i = 0; i < n; ++i) {
            if (!mask[i]) continue;
            const float* bb = boxes + 4 * i;
            const float inter = area4(tmaxf(ba[0], bb[0]), tmaxf(ba[1], bb[1]), tminf(ba[2], bb[2]), tminf(ba[3], bb[3]));
            const float iou = inter / (area_a + area4(bb[0], bb[1], bb[2], bb[3]) - inter);
            sc[i] = sc[i] * expf(-(iou * iou / sigma));
        }
    }
    free(sc);
    free(mask);
    return k;
}

/*
 * detection/postprocessor.py:24-78  Postprocessor.postprocess (+ bf/utils/box_utils.py:166-194 nms wrapper).
 *   scores [B,A,C] logits, locs [B,A,4], priors [A,4] centroid form.
 *   softmax != 0: F.softmax then drop the background column (:46-48) -> class ids 1..C-1; else sigmoid -> 1..C.
 *   max_per_class <= 0 means None; max_total <= 0 means None.  soft != 0 selects _soft_nms.
 *   out rows [x1,y1,x2,y2,class,score]; image i writes at out + i*out_cap*6, counts[i] rows (out_cap must be
 *   >= min(max_total, ncls*max_per_class)).  cand_counts (optional, [B]): boxes that entered NMS for image i.
 */
void orc_postprocess(const float* scores, const float* locs, const float* priors, int B, int64_t A, int C,
                     int softmax, float score_thr, int max_per_class, float nms_thr, int soft, float sigma,
                     int max_total, float xy_scale, float wh_scale, float* out, int out_cap, int32_t* counts,
                     int64_t* cand_counts) {
    const int ncls = softmax ? C - 1 : C;
#pragma omp parallel for schedule(dynamic, 1)
    for (int i = 0; i < B; ++i) {
        float* prob = (float*)malloc(sizeof(float) * (size_t)A * ncls);   /* class-major [ncls][A] */
        float* box = (float*)malloc(sizeof(float) * (size_t)A * 4);
        for (int64_t j = 0; j < A; ++j) {
            const float* x = scores + ((int64_t)i * A + j) * C;
            if (softmax) {
                float m = x[0];
                for (int c = 1; c < C; ++c) m = tmaxf(m, x[c]);
                float s = 0.0f;
                for (int c = 0; c < C; ++c) s += expf(x[c] - m);
                for (int c = 1; c < C; ++c) prob[(int64_t)(c - 1) * A + j] = expf(x[c] - m) / s;
            } else {
                for (int c = 0; c < C; ++c) prob[(int64_t)c * A + j] = sigmoidf_(x[c]);
            }
            float cen[4];
            decode_one(locs + ((int64_t)i * A + j) * 4, priors + 4 * j, xy_scale, wh_scale, cen); /* :52 */
            orc_to_corners(cen, box + 4 * j, 1);                                                  /* :53 */
        }
        orc_kv* kv = (orc_kv*)malloc(sizeof(orc_kv) * (size_t)A);
        float* cb = (float*)malloc(sizeof(float) * (size_t)A * 4);
        float* cs = (float*)malloc(sizeof(float) * (size_t)A);
        int32_t* picked = (int32_t*)malloc(sizeof(int32_t) * (size_t)A);
        size_t cap = 1024, tot = 0;
        float* rows = (float*)malloc(sizeof(float) * 6 * cap);
        int64_t ncand = 0;
        for (int c = 0; c < ncls; ++c) { /* :61-68 */
            const float* p = prob + (int64_t)c * A;
            int n = 0;
            for (int64_t j = 0; j < A; ++j) if (p[j] > score_thr) { kv[n].v = p[j]; kv[n].i = (int32_t)j; ++n; }
            if (max_per_class > 0 && max_per_class < n) { /* box_utils.py:186-188 topk (set semantics) */
                qsort(kv, (size_t)n, sizeof(orc_kv), cmp_desc_stable);
                n = max_per_class;
            }
            for (int k = 0; k < n; ++k) { memcpy(cb + 4 * k, box + 4 * (int64_t)kv[k].i, 16); cs[k] = kv[k].v; }
            ncand += n;
            const int np = soft ? orc_nms_soft(cb, cs, n, score_thr, sigma, picked) : orc_nms_hard(cb, cs, n, nms_thr, picked);
            for (int k = 0; k < np; ++k) {
                if (tot == cap) { cap *= 2; rows = (float*)realloc(rows, sizeof(float) * 6 * cap); }
                memcpy(rows + 6 * tot, cb + 4 * picked[k], 16);
                rows[6 * tot + 4] = (float)(c + 1); /* :66 */
                rows[6 * tot + 5] = cs[picked[k]];
                ++tot;
            }
        }
        float* o = out + (int64_t)i * out_cap * 6;
        if (max_total > 0 && (size_t)max_total < tot) { /* :72-74 topk(sorted=True) */
            orc_kv* tk = (orc_kv*)malloc(sizeof(orc_kv) * tot);
            for (size_t k = 0; k < tot; ++k) { tk[k].v = rows[6 * k + 5]; tk[k].i = (int32_t)k; }
            qsort(tk, tot, sizeof(orc_kv), cmp_desc_stable);
            for (int k = 0; k < max_total; ++k) memcpy(o + 6 * k, rows + 6 * (size_t)tk[k].i, 24);
            counts[i] = max_total;
            free(tk);
        } else {
            const size_t nw = tot < (size_t)out_cap ? tot : (size_t)out_cap;
            memcpy(o, rows, 24 * nw);
            counts[i] = (int32_t)nw;
        }
        if (cand_counts) cand_counts[i] = ncand;
        free(prob); free(box); free(kv); free(cb); free(cs); free(picked); free(rows);
    }
}

/* ---- anchors ---------------------------------------------------------------------------------------------- */

/*
 * torch.linspace(start, end, steps) for float32 on CPU (aten RangeFactoriesKernel.cpp linspace_kernel):
 * step = (end - start) / (steps - 1) in fp32; element idx < steps/2 -> start + step*idx, else
 * end - step*(steps - idx - 1).  The multiply-add is contracted to an FMA in the shipped x86 builds
 * (checked against torch 2.10 outputs in tests/test_oracle_golden.py); use_fma selects that form.
 */
void orc_linspace_f32(float start, float end, int64_t steps, int use_fma, float* out) {
    if (steps == 1) { out[0] = start; return; }
    const float step = (end - start) / (float)(steps - 1);
    const int64_t half = steps / 2;
    for (int64_t i = 0; i < steps; ++i) {
        if (i < half) out[i] = use_fma ? fmaf(step, (float)i, start) : start + step * (float)i;
        else out[i] = use_fma ? fmaf(-step, (float)(steps - i - 1), end) : end - step * (float)(steps - i - 1);
    }
}

/*
 * detection/anchor_generators/ssd.py:12-53 (builder: scales = linspace(min,max,L+1) fp32) and :55-151
 * (SsdAnchorGenerator ctor + _generate_anchors) for num_branches = 1, step = None, offset = .5 (the builder never
 * forwards offsets/steps it is given except `steps`), flip = True, clip = False.
 *   ratios/nratio: the per-level aspect_ratios list BEFORE flipping.  Returns nb; writes [H,W,nb,4].
 *   min_scale/max_scale are the fp32 linspace values scales[i], scales[i+1].
 */
int orc_anchors_ssd_level(const double* ratios, int nratio, float min_scale, float max_scale, int img_w, int img_h,
                          int layer_w, int layer_h, int use_fma, float* out) {
    double ar[64];
    int nar = 0;
    for (int k = 0; k < nratio; ++k) { /* :86-92 */
        ar[nar++] = ratios[k];
        if (ratios[k] > 1.0) ar[nar++] = 1.0 / ratios[k];
    }
    const int nb = nar + 1; /* max_scale given -> +1 (:96-97) */
    /* :99-104 scales = linspace(min,max,2) -> [min, max]; :125 sizes = scales * img (fp32 * python int) */
    const float min_w = min_scale * (float)img_w, min_h = min_scale * (float)img_h;
    const float max_w = max_scale * (float)img_w, max_h = max_scale * (float)img_h;
    float hw[65][2];
    for (int k = 0; k < nar; ++k) { /* :131-133 tensor(fp32) * python float -> fp32 op with the scalar cast to fp32 */
        const float sr = (float)sqrt(ar[k]);
        hw[k][0] = min_w * sr;
        hw[k][1] = min_h / sr;
    }
    /* :135-136 math.sqrt(fp32 product) computed in double, stored to fp32 */
    hw[nar][0] = (float)sqrt((double)(min_w * max_w));
    hw[nar][1] = (float)sqrt((double)(min_h * max_h));
    const double step_w = (double)img_w / layer_w, step_h = (double)img_h / layer_h; /* :117-118 */
    float* xs = (float*)malloc(sizeof(float) * (size_t)layer_w);
    float* ys = (float*)malloc(sizeof(float) * (size_t)layer_h);
    orc_linspace_f32((float)(0.5 * step_w), (float)((0.5 + layer_w - 1) * step_w), layer_w, use_fma, xs); /* :138 */
    orc_linspace_f32((float)(0.5 * step_h), (float)((0.5 + layer_h - 1) * step_h), layer_h, use_fma, ys); /* :139 */
    for (int y = 0; y < layer_h; ++y)
        for (int x = 0; x < layer_w; ++x)
            for (int k = 0; k < nb; ++k) {
                float* o = out + (((int64_t)y * layer_w + x) * nb + k) * 4;
                o[0] = xs[x]; o[1] = ys[y]; o[2] = hw[k][0]; o[3] = hw[k][1];
            }
    free(xs); free(ys);
    return nb;
}

/* detection/anchor_generators/retina_net.py:18-54.  sizes = scale * 2**(level + x/spl) in python double (:26);
 * hws[..] = size*sqrt(ar), size/sqrt(ar) in double, stored to fp32 (:42-43); box order scale-major (:40-43). */
int orc_anchors_retina_level(const double* ratios, int nratio, int level, double scale, int scales_per_level,
                             int img_w, int img_h, int layer_w, int layer_h, int use_fma, float* out) {
    const int nb = nratio * scales_per_level;
    float hw[256][2];
    for (int j = 0; j < scales_per_level; ++j) {
        const double size = scale * pow(2.0, (double)level + (double)j / scales_per_level);
        for (int k = 0; k < nratio; ++k) {
            hw[j * nratio + k][0] = (float)(size * sqrt(ratios[k]));
            hw[j * nratio + k][1] = (float)(size / sqrt(ratios[k]));
        }
    }
    const double step_w = (double)img_w / layer_w, step_h = (double)img_h / layer_h;
    float* xs = (float*)malloc(sizeof(float) * (size_t)layer_w);
    float* ys = (float*)malloc(sizeof(float) * (size_t)layer_h);
    orc_linspace_f32((float)(0.5 * step_w), (float)((0.5 + layer_w - 1) * step_w), layer_w, use_fma, xs);
    orc_linspace_f32((float)(0.5 * step_h), (float)((0.5 + layer_h - 1) * step_h), layer_h, use_fma, ys);
    for (int y = 0; y < layer_h; ++y)
        for (int x = 0; x < layer_w; ++x)
            for (int k = 0; k < nb; ++k) {
                float* o = out + (((int64_t)y * layer_w + x) * nb + k) * 4;
                o[0] = xs[x]; o[1] = ys[y]; o[2] = hw[k][0]; o[3] = hw[k][1];
            }
    free(xs); free(ys);
    return nb;
}

/* ---- the other selectable losses ------------------------------------------------------------------- */

/* bf/modules/losses.py:13-18 _Loss._soften for a target row with ONE positive entry of value s at column pos (or none):
 * positive -> s (1 - eps), every other entry -> eps * s / (C - 1).  Row sums are preserved. */
static inline float soft_t(int c, int pos, float s, int C, float eps) {
    if (pos < 0) return 0.0f;
    if (eps == 0.0f) return c == pos ? s : 0.0f;
    return c == pos ? s - eps * s : eps * s / (float)(C - 1);
}

/* bf/utils/box_utils.py:104-143 generalized_iou (cartesian=False) for one pair of corner boxes */
static inline float giou_pair(const float* a, const float* b) {
    const float inter = area4(tmaxf(a[0], b[0]), tmaxf(a[1], b[1]), tminf(a[2], b[2]), tminf(a[3], b[3]));
    const float uni = area4(a[0], a[1], a[2], a[3]) + area4(b[0], b[1], b[2], b[3]) - inter;
    const float enc = area4(tminf(a[0], b[0]), tminf(a[1], b[1]), tmaxf(a[2], b[2]), tmaxf(a[3], b[3]));
    return inter / uni - (enc - uni) / enc;
}

/*
 * MultiboxLoss.forward (detection/losses/multibox_loss.py:35-94) for the remaining loss classes of bf/modules/losses.py:
 *   cls_kind 2  SoftmaxFocalLoss :56-78 (alpha < 0 means None); class target = integer class, ignore_index = -1;
 *               reduce_mean != 0: mean over ALL sampled rows (the constructor quirk drops reduction='sum' here too)
 *   cls_kind 3  CrossEntropyWithSoftTargetsLoss :80-93 on the soft target of multibox_loss.py:68-71
 *               (class_target[row, cls] = target score for cls != -1), epsilon smoothing, reduction 'sum'
 *   cls_kind 4  BinaryCrossEntropyWithSoftTargetsLoss :95-106 on the multiclass target of multibox_loss.py:64-67
 *               (class_target[row, cls - 1] = target score for positives), reduction 'sum'
 *   cls_kind 0  CrossEntropyLoss (as orc_multibox_loss_ce)
 *   loc_kind 0  SmoothL1Loss on encoded targets (mutates target[..., 0:4]);
 *   loc_kind 1  GeneralizedIoULoss :109-114 on decoded corner boxes vs the RAW corner target (multibox_loss.py:77-79;
 *               target is NOT mutated).
 * Gradients are finite-difference-free analytic derivatives of exactly these expressions.
 */
void orc_multibox_loss_ex(const float* scores, const float* locs, const float* anchors, float* target, const uint8_t* sampled, int B,
                          int64_t A, int C, int cls_kind, int loc_kind, float gamma, float alpha, float epsilon, int reduce_mean,
                          float cls_w, float loc_w, float xy_scale, float wh_scale, float eps, float beta, double* out3, float* dscores,
                          float* dlocs) {
    const int64_t N = (int64_t)B * A;
    double cls_sum = 0.0, loc_sum = 0.0, tsum = 0.0;
    int64_t npos = 0, nrows = 0, nposrows = 0;
    for (int64_t r = 0; r < N; ++r) {
        const int64_t j = r % A;
        float* t = target + r * 6;
        const int64_t cls = (int64_t)t[4];
        const float ts = t[5];
        const int pos = cls != 0 && cls != -1;
        npos += pos;
        if (sampled[r]) {
            nrows += 1;
            const float* x = scores + r * C;
            float m, ls;
            row_lse(x, C, &m, &ls);
            if (cls_kind == 0) {
                if (cls != -1) cls_sum += (double)(-((x[cls] - m) - ls));
            } else if (cls_kind == 2) {
                if (cls != -1) {
                    const float logpb = (x[cls] - m) - ls, pb = expf(logpb);
                    float l = -1.0f * powf(1.0f - pb, gamma) * logpb;
                    if (alpha >= 0.0f) l *= (cls == 0 ? 1.0f - alpha : alpha);
                    cls_sum += (double)l;
                }
            } else if (cls_kind == 3) {
                const int p_ = cls != -1 ? (int)cls : -1;
                float row = 0.0f, rs = 0.0f;
                for (int c = 0; c < C; ++c) {
                    const float tc = soft_t(c, p_, ts, C, epsilon);
                    row += ((x[c] - m) - ls) * tc;
                    rs += tc;
                }
                cls_sum += (double)(-row);
                tsum += (double)rs;
            } else if (cls_kind == 4) {
                const int p_ = pos ? (int)cls - 1 : -1;
                float row = 0.0f, rs = 0.0f;
                for (int c = 0; c < C; ++c) {
                    const float tc = soft_t(c, p_, ts, C, epsilon);
                    row += fmaxf(x[c], 0.0f) - x[c] * tc + log1pf(expf(-fabsf(x[c])));
                    rs += tc;
                }
                cls_sum += (double)row;
                tsum += (double)(rs / (float)C);
                nposrows += rs / (float)C > 0.0f;
            }
        }
        if (loc_kind == 0) {
            orc_to_centroids_inplace(t, 1);
            encode_inplace(t, anchors + 4 * j, xy_scale, wh_scale, eps);
            if (pos) for (int k = 0; k < 4; ++k) loc_sum += (double)smooth_l1(locs[r * 4 + k], t[k], beta);
        } else if (pos) {
            float cen[4], cor[4];
            decode_one(locs + r * 4, anchors + 4 * j, xy_scale, wh_scale, cen);
            orc_to_corners(cen, cor, 1);
            loc_sum += (double)(1.0f - giou_pair(cor, t));
        }
    }
    const float divider = (float)(npos < 1 ? 1 : npos);
    float scale = 1.0f;
    if (cls_kind == 3) scale = 1.0f / ((float)tsum / (float)nrows);                 /* target.sum(-1).mean() ** -1 */
    if (cls_kind == 4) scale = 1.0f / ((float)tsum / (float)nposrows);              /* scale.sum() / (scale > 0).sum() ** -1 */
    const float mean_div = (cls_kind == 2 && reduce_mean) ? (float)nrows : 1.0f;
    const float class_loss = scale * (float)cls_sum / mean_div * cls_w / divider;
    const float loc_loss = (float)loc_sum * loc_w / divider;
    out3[0] = (double)(class_loss + loc_loss); out3[1] = class_loss; out3[2] = loc_loss;
    if (!dscores && !dlocs) return;
    for (int64_t r = 0; r < N; ++r) {
        const int64_t j = r % A;
        const float* t = target + r * 6;
        const int64_t cls = (int64_t)t[4];
        const float ts = t[5];
        const int pos = cls != 0 && cls != -1;
        if (dscores) {
            float* g = dscores + r * C;
            memset(g, 0, sizeof(float) * C);
            if (sampled[r]) {
                const float* x = scores + r * C;
                float m, ls;
                row_lse(x, C, &m, &ls);
                const float gs = scale / mean_div * cls_w / divider;
                if ((cls_kind == 0 || cls_kind == 2) && cls != -1) {
                    float coef = 1.0f;   /* dL/dx_c = coef * (p_c - onehot) */
                    if (cls_kind == 2) {
                        const float logpb = (x[cls] - m) - ls, pb = expf(logpb), om = 1.0f - pb;
                        /* L = -(1-p)^g ln p ;  dL/dx_c = [g (1-p)^(g-1) p ln p - (1-p)^g] (onehot - p_c) */
                        coef = -(gamma * powf(om, gamma - 1.0f) * pb * logpb - powf(om, gamma));
                        if (alpha >= 0.0f) coef *= (cls == 0 ? 1.0f - alpha : alpha);
                    }
                    for (int c = 0; c < C; ++c) g[c] = coef * (expf((x[c] - m) - ls) - (c == cls ? 1.0f : 0.0f)) * gs;
                } else if (cls_kind == 3) {
                    const int p_ = cls != -1 ? (int)cls : -1;
                    float rs = 0.0f;
                    for (int c = 0; c < C; ++c) rs += soft_t(c, p_, ts, C, epsilon);
                    for (int c = 0; c < C; ++c) g[c] = (expf((x[c] - m) - ls) * rs - soft_t(c, p_, ts, C, epsilon)) * gs;
                } else if (cls_kind == 4) {
                    const int p_ = pos ? (int)cls - 1 : -1;
                    for (int c = 0; c < C; ++c) g[c] = (sigmoidf_(x[c]) - soft_t(c, p_, ts, C, epsilon)) * gs;
                }
            }
        }
        if (dlocs) {
            float* g = dlocs + r * 4;
            g[0] = g[1] = g[2] = g[3] = 0.0f;
            if (!pos) continue;
            const float gl = loc_w / divider;
            if (loc_kind == 0) {
                for (int k = 0; k < 4; ++k) g[k] = smooth_l1_grad(locs[r * 4 + k], t[k], beta) * gl;
            } else {
                const float* pr = anchors + 4 * j;
                const float* l = locs + r * 4;
                float cen[4], P[4];
                decode_one(l, pr, xy_scale, wh_scale, cen);
                orc_to_corners(cen, P, 1);
                /* loss = 2 - inter/uni - uni/enc */
                const float pw = P[2] - P[0], ph = P[3] - P[1], cw = clamp0(pw), ch = clamp0(ph);
                const float area_p = cw * ch, area_t = area4(t[0], t[1], t[2], t[3]);
                const float ix1 = tmaxf(P[0], t[0]), iy1 = tmaxf(P[1], t[1]), ix2 = tminf(P[2], t[2]), iy2 = tminf(P[3], t[3]);
                const float iw = clamp0(ix2 - ix1), ih = clamp0(iy2 - iy1), inter = iw * ih;
                const float uni = area_p + area_t - inter;
                const float ex1 = tminf(P[0], t[0]), ey1 = tminf(P[1], t[1]), ex2 = tmaxf(P[2], t[2]), ey2 = tmaxf(P[3], t[3]);
                const float ew = clamp0(ex2 - ex1), eh = clamp0(ey2 - ey1), enc = ew * eh;
                const float d_inter = -1.0f / uni, d_uni = inter / (uni * uni) - 1.0f / enc, d_enc = uni / (enc * enc);
                const float c_inter = d_inter - d_uni, c_area = d_uni;   /* uni = area_p + area_t - inter */
                float dP[4] = {0.f, 0.f, 0.f, 0.f};
                /* area_p = clamp(x2-x1) clamp(y2-y1) (clamp passes the gradient where its argument is >= 0) */
                if (pw >= 0.0f) { dP[2] += c_area * ch; dP[0] -= c_area * ch; }
                if (ph >= 0.0f) { dP[3] += c_area * cw; dP[1] -= c_area * cw; }
                /* inter = clamp(ix2-ix1) clamp(iy2-iy1); max/min route the gradient to the selected argument (ties: half) */
                if (ix2 - ix1 >= 0.0f) {
                    const float gq = c_inter * ih;
                    dP[2] += gq * (P[2] < t[2] ? 1.0f : (P[2] == t[2] ? 0.5f : 0.0f));
                    dP[0] -= gq * (P[0] > t[0] ? 1.0f : (P[0] == t[0] ? 0.5f : 0.0f));
                }
                if (iy2 - iy1 >= 0.0f) {
                    const float gq = c_inter * iw;
                    dP[3] += gq * (P[3] < t[3] ? 1.0f : (P[3] == t[3] ? 0.5f : 0.0f));
                    dP[1] -= gq * (P[1] > t[1] ? 1.0f : (P[1] == t[1] ? 0.5f : 0.0f));
                }
                if (ex2 - ex1 >= 0.0f) {
                    const float gq = d_enc * eh;
                    dP[2] += gq * (P[2] > t[2] ? 1.0f : (P[2] == t[2] ? 0.5f : 0.0f));
                    dP[0] -= gq * (P[0] < t[0] ? 1.0f : (P[0] == t[0] ? 0.5f : 0.0f));
                }
                if (ey2 - ey1 >= 0.0f) {
                    const float gq = d_enc * ew;
                    dP[3] += gq * (P[3] > t[3] ? 1.0f : (P[3] == t[3] ? 0.5f : 0.0f));
                    dP[1] -= gq * (P[1] < t[1] ? 1.0f : (P[1] == t[1] ? 0.5f : 0.0f));
                }
                /* corners = c -+ wh/2 ; c = p_xy + p_wh t_xy / xy_scale ; wh = p_wh exp(t_wh / wh_scale) */
                const float dcx = dP[0] + dP[2], dcy = dP[1] + dP[3], dw = (dP[2] - dP[0]) * 0.5f, dh = (dP[3] - dP[1]) * 0.5f;
                g[0] = dcx * pr[2] / xy_scale * gl;
                g[1] = dcy * pr[3] / xy_scale * gl;
                g[2] = dw * cen[2] / wh_scale * gl;
                g[3] = dh * cen[3] / wh_scale * gl;
            }
        }
    }
}

/*
 * detection/metrics/mean_average_precision.py:10-116  mean_average_precision
 *   pred [N,7]: image id, corner box, class id, score.  gt rows [sum G, gstride] (corner box, class at 4, difficult at 6 when
 *   gstride > 6 -- :22), gt_off[num_images + 1].  Class ids are integers in [0, num_classes).
 *   ap_out[num_classes]: AP of every class that has at least one counted (non-difficult) ground truth (:33-34), NaN elsewhere;
 *   returns the mean over those classes (:111).  Score ties (the reference's argsort is unstable, :40): lower row first.
 *   All arithmetic that the reference does in fp32 tensors (cumulative counts, precision, recall, running max, dot /
 *   11-point mean) is done in fp32 in the same order; torch.max propagates NaN (:95-96).
 */
double orc_mean_average_precision(const float* pred, int64_t n, const float* gt, int gstride, const int32_t* gt_off, int num_images,
                                 int num_classes, float iou_threshold, int voc, float* ap_out) {
    const int ignore_difficult = gstride > 6;                                    /* :22 */
    const int64_t total_gt = gt_off[num_images];
    int64_t* total_positive = (int64_t*)calloc((size_t)num_classes, sizeof(int64_t));
    for (int64_t g = 0; g < total_gt; ++g) {                                     /* :27-35 */
        const int c = (int)gt[g * gstride + 4];
        if (c < 0 || c >= num_classes) continue;
        if (!ignore_difficult || gt[g * gstride + 6] == 0.0f) total_positive[c] += 1;
    }
    orc_kv* order = (orc_kv*)malloc(sizeof(orc_kv) * (size_t)(n > 0 ? n : 1));  /* :40-41 */
    for (int64_t i = 0; i < n; ++i) { order[i].v = pred[i * 7 + 6]; order[i].i = (int32_t)i; }
    qsort(order, (size_t)n, sizeof(orc_kv), cmp_desc_stable);
    uint8_t* matched = (uint8_t*)calloc((size_t)(total_gt > 0 ? total_gt : 1), 1);
    /* per class: the running (tp, fp) lists of :51-52 in prediction order */
    int64_t* cnt = (int64_t*)calloc((size_t)num_classes, sizeof(int64_t));
    for (int64_t i = 0; i < n; ++i) { const int c = (int)pred[i * 7 + 5]; if (c >= 0 && c < num_classes) cnt[c] += 1; }
    int64_t* start = (int64_t*)malloc(sizeof(int64_t) * (size_t)(num_classes + 1));
    start[0] = 0;
    for (int c = 0; c < num_classes; ++c) start[c + 1] = start[c] + cnt[c];
    float* tp = (float*)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
    float* fp = (float*)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
    int64_t* fill = (int64_t*)calloc((size_t)num_classes, sizeof(int64_t));
    for (int64_t k = 0; k < n; ++k) {                                            /* :47-70 */
        const float* p = pred + (int64_t)order[k].i * 7;
        const int id = (int)p[0], c = (int)p[5];
        if (c < 0 || c >= num_classes) continue;
        const int64_t pos = start[c] + fill[c];
        float t = fill[c] ? tp[pos - 1] : 0.0f, f = fill[c] ? fp[pos - 1] : 0.0f;
        fill[c] += 1;
        /* IoU against the ground truths of this class in this image, in row order; first maximum; NaN propagates */
        int have = 0;
        int64_t best_g = -1;
        float best = 0.0f;
        const float parea = area4(p[1], p[2], p[3], p[4]);
        if (id >= 0 && id < num_images)
            for (int64_t g = gt_off[id]; g < gt_off[id + 1]; ++g) {
                const float* q = gt + g * gstride;
                if ((int)q[4] != c) continue;
                const float v = iou_pair(p + 1, parea, q, area4(q[0], q[1], q[2], q[3]));
                if (!have || (v > best && best == best) || (v != v && best == best)) { best = v; best_g = g; }
                have = 1;
            }
        if (!have) f += 1.0f;                                                    /* :54-56 */
        else if (best > iou_threshold) {                                         /* :60 */
            if (!ignore_difficult || gt[best_g * gstride + 6] == 0.0f) {         /* :61 */
                if (!matched[best_g]) { t += 1.0f; matched[best_g] = 1; }       /* :62-64 */
                else f += 1.0f;                                                  /* :66 */
            }
        } else f += 1.0f;                                                        /* :68 */
        tp[pos] = t;
        fp[pos] = f;
    }
    double map_sum = 0.0;
    int map_n = 0;
    for (int c = 0; c < num_classes; ++c) {
        ap_out[c] = NAN;
        if (!total_positive[c]) continue;                                        /* :74-75, :80 */
        int64_t m = cnt[c];
        const float* tpc = tp + start[c];
        const float* fpc = fp + start[c];
        float one_tp = 0.0f, one_fp = 1.0f;
        if (m == 0) { m = 1; tpc = &one_tp; fpc = &one_fp; }                     /* :81-89 */
        float* prec = (float*)malloc(sizeof(float) * (size_t)(m + 1));
        float* rec = (float*)malloc(sizeof(float) * (size_t)(m + 2));
        for (int64_t i = 0; i < m; ++i) prec[i] = tpc[i] / (tpc[i] + fpc[i]);    /* :91 */
        prec[m] = 0.0f;                                                          /* :92 */
        for (int64_t i = m; i >= 1; --i) prec[i - 1] = tmaxf(prec[i - 1], prec[i]);   /* :94-95 */
        const float tot = (float)total_positive[c];
        float ap;
        if (voc) {                                                               /* :99-103 */
            for (int64_t i = 0; i < m; ++i) rec[i] = tpc[i] / tot;
            rec[m] = 1.0f;
            float acc = 0.0f;
            for (int k = 0; k <= 10; ++k) {
                const float thr = (float)(0.0 + k * 0.1);                        /* torch.arange(0, 1.1, .1): double arithmetic, fp32 result */
                int64_t idx = 0;
                for (int64_t i = 0; i <= m; ++i) idx += thr > rec[i];
                acc += prec[idx];
            }
            ap = acc / 11.0f;
        } else {                                                                 /* :104-106 */
            rec[0] = 0.0f;
            for (int64_t i = 0; i < m; ++i) rec[i + 1] = tpc[i] / tot;
            rec[m + 1] = 1.0f;
            float acc = 0.0f;
            for (int64_t i = 0; i <= m; ++i) acc += (rec[i + 1] - rec[i]) * prec[i];
            ap = acc;
        }
        ap_out[c] = ap;
        map_sum += (double)ap;                                                   /* :108, :111: python floats */
        map_n += 1;
        free(prec);
        free(rec);
    }
    free(total_positive); free(order); free(matched); free(cnt); free(start); free(tp); free(fp); free(fill);
    return map_n ? map_sum / map_n : (double)NAN;
}
