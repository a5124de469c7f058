/*
 * rn_oracle.c -- CPU restatement of the reference's dense-head path.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library; the product path
 * (pytorch_retinanet_amd/) never does and fails loudly without its HIP library.
 *
 * Parity status: PINNED.  tests/golden/gen_golden.py imports the reference
 * itself (/root/reference, behind the test-only torchvision stand-in) in the
 * build container, asserts this restatement against it, and writes the golden
 * vectors tests/ re-checks on every run (tests/test_oracle_golden.py).
 *
 * Every function cites the reference lines it restates (paths relative to
 * /root/reference).  Element arithmetic is fp32 in the reference's evaluation
 * order; only long sums are carried in double (the reference uses torch's
 * cascaded fp32 sum, error ~1e-7 rel, so both agree to the 1e-5 tolerance).
 *
 * Build: see oracle/Makefile (gcc -O2 -fopenmp -ffp-contract=off).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define RNO_API __attribute__((visibility("default")))

typedef struct rno_level { int32_t H, W, stride, num_cell; } rno_level;

typedef struct rno_loss_params {
    float alpha, gamma, beta, logit_shift, log_eps;
    float reg_w[4];
} rno_loss_params;

typedef struct rno_detect_params {
    float score_thr, min_box, nms_thr;
    int32_t max_det;
    float reg_w[4];
} rno_detect_params;

RNO_API int rno_version(void) { return 1; }

RNO_API int rno_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

RNO_API void rno_set_num_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* ------------------------------------------------------------------------ */
/* A1: AnchorGenerator.generate_cell_anchors  retinanet/anchors.py:110-135    */
/* size-major, ratio-minor; Python double arithmetic, then .float() (:104)    */
/* ------------------------------------------------------------------------ */
RNO_API void rno_cell_anchors(const double *sizes, int ns, const double *ratios, int nr, float *out)
{
    int n = 0;
    for (int i = 0; i < ns; ++i) {
        double area = pow(sizes[i], 2.0);               /* anchors.py:127 */
        for (int j = 0; j < nr; ++j) {
            double w = sqrt(area / ratios[j]);          /* anchors.py:129 */
            double h = ratios[j] * w;                   /* anchors.py:130 */
            out[n * 4 + 0] = (float)(-w / 2.0);
            out[n * 4 + 1] = (float)(-h / 2.0);
            out[n * 4 + 2] = (float)(w / 2.0);
            out[n * 4 + 3] = (float)(h / 2.0);
            ++n;
        }
    }
}

/* ------------------------------------------------------------------------ */
/* A2-A4: _compute_grid_offsets / grid_anchors / forward                      */
/* retinanet/anchors.py:151-170, :172-197, :199-228                           */
/* shifts = arange(offset*stride, n*stride, stride) (double accumulate on CPU */
/* then fp32), anchors = fp32(shift) + fp32(cell), row-major (y,x), cell      */
/* anchors fastest, levels concatenated in order.                             */
/* ------------------------------------------------------------------------ */
RNO_API int64_t rno_anchors_count(const rno_level *lv, int L)
{
    int64_t n = 0;
    for (int l = 0; l < L; ++l) n += (int64_t)lv[l].H * lv[l].W * lv[l].num_cell;
    return n;
}

RNO_API void rno_anchors_emit(const rno_level *lv, int L, const float *const *cell, double offset, float *out)
{
    int64_t base = 0;
    for (int l = 0; l < L; ++l) {
        const int H = lv[l].H, W = lv[l].W, S = lv[l].stride, C = lv[l].num_cell;
        const double start = offset * (double)S;   /* python: offset * stride */
        const float *ca = cell[l];
#pragma omp parallel for schedule(static)
        for (int y = 0; y < H; ++y) {
            const float sy = (float)(start + (double)y * (double)S);
            for (int x = 0; x < W; ++x) {
                const float sx = (float)(start + (double)x * (double)S);
                float *o = out + (base + ((int64_t)y * W + x) * C) * 4;
                for (int c = 0; c < C; ++c) {
                    o[c * 4 + 0] = sx + ca[c * 4 + 0];
                    o[c * 4 + 1] = sy + ca[c * 4 + 1];
                    o[c * 4 + 2] = sx + ca[c * 4 + 2];
                    o[c * 4 + 3] = sy + ca[c * 4 + 3];
                }
            }
        }
        base += (int64_t)H * W * C;
    }
}

/* ------------------------------------------------------------------------ */
/* M1: torchvision box_iou as called at retinanet/box_utils.py:74             */
/*   area=(x2-x1)*(y2-y1); lt=max, rb=min; wh=clamp(rb-lt,0); inter=w*h;      */
/*   iou = inter / ((area_t + area_a) - inter)      all fp32, no fma          */
/* ------------------------------------------------------------------------ */
static inline float iou_pair(const float *t, float area_t, const float *a, float area_a)
{
    float ltx = t[0] > a[0] ? t[0] : a[0];
    float lty = t[1] > a[1] ? t[1] : a[1];
    float rbx = t[2] < a[2] ? t[2] : a[2];
    float rby = t[3] < a[3] ? t[3] : a[3];
    float w = rbx - ltx; if (!(w > 0.0f)) w = (w != w) ? w : 0.0f;   /* clamp(min=0) keeps NaN */
    float h = rby - lty; if (!(h > 0.0f)) h = (h != h) ? h : 0.0f;
    float inter = w * h;
    float uni = (area_t + area_a) - inter;
    return inter / uni;
}

/* ------------------------------------------------------------------------ */
/* M2: matcher  retinanet/box_utils.py:51-80                                   */
/* init -2 (:68); empty GT -> all -2 (:70-71); vals,idxs = iou.max(dim=0)      */
/* (:76, first max wins on ties, NaN propagates); vals<bg -> -1 (:78);         */
/* vals>fg -> idx (:79).  Thresholds compare in fp32.                          */
/* anchor_bstride = 0: one anchor set shared by every image.                   */
/* ------------------------------------------------------------------------ */
RNO_API void rno_iou_match(const float *anchors, int64_t anchor_bstride,
                           const float *gt, const int32_t *gt_off, int B, int64_t A,
                           float fg_thr, float bg_thr, int64_t *matches, int32_t *num_fg)
{
    for (int b = 0; b < B; ++b) {
        const int t0 = gt_off[b], T = gt_off[b + 1] - gt_off[b];
        const float *anc = anchors + (int64_t)b * anchor_bstride;
        int64_t *m = matches + (int64_t)b * A;
        int64_t nfg = 0;
        float *area_t = (float *)malloc(sizeof(float) * (size_t)(T > 0 ? T : 1));
        for (int t = 0; t < T; ++t) {
            const float *g = gt + (int64_t)(t0 + t) * 4;
            area_t[t] = (g[2] - g[0]) * (g[3] - g[1]);
        }
#pragma omp parallel for schedule(static) reduction(+ : nfg)
        for (int64_t a = 0; a < A; ++a) {
            if (T == 0) { m[a] = -2; continue; }
            const float *an = anc + a * 4;
            const float area_a = (an[2] - an[0]) * (an[3] - an[1]);
            float best = iou_pair(gt + (int64_t)t0 * 4, area_t[0], an, area_a);
            int bi = 0;
            for (int t = 1; t < T; ++t) {
                if (best != best) break;                       /* first NaN wins (torch.max) */
                float v = iou_pair(gt + (int64_t)(t0 + t) * 4, area_t[t], an, area_a);
                if (v > best || v != v) { best = v; bi = t; }
            }
            int64_t r = -2;
            if (best < bg_thr) r = -1;
            if (best > fg_thr) r = bi;
            m[a] = r;
            if (r >= 0) ++nfg;
        }
        if (num_fg) num_fg[b] = (int32_t)nfg;
        free(area_t);
    }
}

/* ------------------------------------------------------------------------ */
/* L1: bbox_2_activ  retinanet/box_utils.py:25-34 (+convert_xywh :11-15)       */
/* ------------------------------------------------------------------------ */
static inline void encode_box(const float *g, const float *a, const float *rw, float log_eps, float *o)
{
    float gcx = (g[0] + g[2]) / 2.0f, gcy = (g[1] + g[3]) / 2.0f;
    float gw = g[2] - g[0], gh = g[3] - g[1];
    float acx = (a[0] + a[2]) / 2.0f, acy = (a[1] + a[3]) / 2.0f;
    float aw = a[2] - a[0], ah = a[3] - a[1];
    o[0] = ((gcx - acx) / aw) * rw[0];
    o[1] = ((gcy - acy) / ah) * rw[1];
    o[2] = logf(gw / aw + log_eps) * rw[2];
    o[3] = logf(gh / ah + log_eps) * rw[3];
}

RNO_API void rno_encode(const float *gt, const float *anchors, int64_t n, const float *rw, float log_eps, float *out)
{
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) encode_box(gt + i * 4, anchors + i * 4, rw, log_eps, out + i * 4);
}

/* ------------------------------------------------------------------------ */
/* L2-L5: RetinaNetLosses  retinanet/losses.py:19-27 (smooth-L1),             */
/* :29-47 (focal, detached weights Q3, alpha reversed Q2), :49-111 (calc_loss, */
/* logit shift Q1 :84, ignore rows dropped :78-79, per-image /clamp(nfg,1)     */
/* :108-109), :113-145 (mean over images).                                     */
/* Also returns the gradients the reference's autograd would produce for       */
/* d(classification_loss)/d(cls) and d(regression_loss)/d(box):                */
/*   gcls = w * (sigmoid(x+shift) - t) / max(nfg,1) / B   (w constant, Q3)     */
/*   gbox = smoothl1'(pred - tgt)     / max(nfg,1) / B    (fg rows only)       */
/* cls [B][A][K], box [B][A][4] fp32; gt_labels are 1..K (0 is background).    */
/* per_image (nullable) receives [B][2] = (bb_loss_b, clas_loss_b) after the   */
/* per-image normalisation, i.e. calc_loss's return values.                    */
/* ------------------------------------------------------------------------ */
static inline float focal_elem(float x, float t, const rno_loss_params *p, float *grad_unscaled)
{
    const float z = x + p->logit_shift;                       /* losses.py:84 */
    const float ps = 1.0f / (1.0f + expf(-z));                /* losses.py:42 */
    float w = t * (1.0f - ps) + (1.0f - t) * ps;              /* losses.py:43 */
    const float al = (1.0f - t) * p->alpha + t * (float)(1.0 - (double)p->alpha); /* losses.py:44 */
    w = (p->gamma == 2.0f) ? w * w : powf(w, p->gamma);       /* losses.py:45 */
    w = w * al;
    /* BCE-with-logits (losses.py:46): (1-t)*z - log_sigmoid(z) */
    const float az = fabsf(z);
    const float logsig = (z < 0.0f ? z : 0.0f) - log1pf(expf(-az));
    const float bce = (1.0f - t) * z - logsig;
    *grad_unscaled = w * (ps - t);
    return w * bce;
}

RNO_API void rno_loss_fwd_bwd(const float *cls, const float *box, int B, int64_t A, int K,
                              const float *anchors, int64_t anchor_bstride,
                              const float *gt_boxes, const int64_t *gt_labels, const int32_t *gt_off,
                              const int64_t *matches, const rno_loss_params *p,
                              float *out_loss /*[2] cls, reg*/, float *per_image /*[B][2] or NULL*/,
                              float *gcls /*nullable*/, float *gbox /*nullable*/)
{
    double tot_cls = 0.0, tot_reg = 0.0;
    for (int b = 0; b < B; ++b) {
        const int64_t *m = matches + (int64_t)b * A;
        const float *anc = anchors + (int64_t)b * anchor_bstride;
        const int t0 = gt_off[b];
        int64_t nfg = 0;
        for (int64_t a = 0; a < A; ++a) nfg += (m[a] >= 0);
        const float denom = (float)(nfg > 1 ? nfg : 1);       /* clamp(min=1) losses.py:108 */
        const float gscale = 1.0f / denom / (float)B;
        double s_cls = 0.0, s_reg = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : s_cls, s_reg)
        for (int64_t a = 0; a < A; ++a) {
            const int64_t mm = m[a];
            const float *x = cls + ((int64_t)b * A + a) * K;
            float *gx = gcls ? gcls + ((int64_t)b * A + a) * K : NULL;
            float *gb = gbox ? gbox + ((int64_t)b * A + a) * 4 : NULL;
            /* regression: fg rows only (losses.py:66-72) */
            if (mm >= 0) {
                float tgt[4];
                encode_box(gt_boxes + (int64_t)(t0 + mm) * 4, anc + a * 4, p->reg_w, p->log_eps, tgt);
                const float *pb = box + ((int64_t)b * A + a) * 4;
                for (int j = 0; j < 4; ++j) {
                    const float d = pb[j] - tgt[j];
                    const float n = fabsf(d);
                    float l, g;
                    if (p->beta < 1e-5f) { l = n; g = (d > 0.0f) - (d < 0.0f); }
                    else if (n < p->beta) { l = 0.5f * (n * n) / p->beta; g = d / p->beta; }
                    else { l = n - 0.5f * p->beta; g = (d > 0.0f) - (d < 0.0f); }
                    s_reg += (double)l;
                    if (gb) gb[j] = g * gscale;
                }
            } else if (gb) {
                gb[0] = gb[1] = gb[2] = gb[3] = 0.0f;
            }
            /* classification: ignore rows (-2) dropped (losses.py:76-79) */
            if (mm == -2) {
                if (gx) memset(gx, 0, sizeof(float) * (size_t)K);
                continue;
            }
            /* label of the row: bg -> 0, fg -> gt label (1..K); one_hot[:,1:] (losses.py:96-103) */
            const int64_t lab = (mm >= 0) ? gt_labels[t0 + mm] : 0;
            for (int k = 0; k < K; ++k) {
                const float t = (lab == (int64_t)(k + 1)) ? 1.0f : 0.0f;
                float g;
                s_cls += (double)focal_elem(x[k], t, p, &g);
                if (gx) gx[k] = g * gscale;
            }
        }
        const float bb = (float)s_reg / denom, cl = (float)s_cls / denom;
        if (per_image) { per_image[b * 2 + 0] = bb; per_image[b * 2 + 1] = cl; }
        tot_reg += (double)bb;
        tot_cls += (double)cl;
    }
    out_loss[0] = (float)(tot_cls / (double)B);               /* losses.py:138 */
    out_loss[1] = (float)(tot_reg / (double)B);               /* losses.py:140 */
}

/* ------------------------------------------------------------------------ */
/* D1+D2: activ_2_bbox  retinanet/box_utils.py:37-48 (Q4: sizes use dx,dy)     */
/* then torchvision clip_boxes_to_image (retinanet/models.py:189) with the     */
/* resized, unpadded (h, w) of each image.  image_hw == NULL: no clip.         */
/* ------------------------------------------------------------------------ */
static inline void decode_box(const float *d, const float *a, const float *rw, float *o)
{
    const float dx = d[0] / rw[0], dy = d[1] / rw[1];         /* box_utils.py:43 */
    const float acx = (a[0] + a[2]) / 2.0f, acy = (a[1] + a[3]) / 2.0f;
    const float aw = a[2] - a[0], ah = a[3] - a[1];
    const float cx = aw * dx + acx, cy = ah * dy + acy;       /* box_utils.py:45 */
    const float w = aw * expf(dx), h = ah * expf(dy);         /* box_utils.py:46 (Q4) */
    o[0] = cx - w / 2.0f; o[1] = cy - h / 2.0f;               /* box_utils.py:20-21 */
    o[2] = cx + w / 2.0f; o[3] = cy + h / 2.0f;
}

static inline float clampf(float v, float lo, float hi) { v = v < lo ? lo : v; return v > hi ? hi : v; }

RNO_API void rno_decode_clip(const float *deltas, int B, int64_t A, const float *anchors, int64_t anchor_bstride,
                             const int32_t *image_hw /*[B][2] or NULL*/, const float *rw, float *out)
{
    for (int b = 0; b < B; ++b) {
        const float *anc = anchors + (int64_t)b * anchor_bstride;
#pragma omp parallel for schedule(static)
        for (int64_t a = 0; a < A; ++a) {
            float *o = out + ((int64_t)b * A + a) * 4;
            decode_box(deltas + ((int64_t)b * A + a) * 4, anc + a * 4, rw, o);
            if (image_hw) {
                const float hh = (float)image_hw[b * 2 + 0], ww = (float)image_hw[b * 2 + 1];
                o[0] = clampf(o[0], 0.0f, ww); o[2] = clampf(o[2], 0.0f, ww);
                o[1] = clampf(o[1], 0.0f, hh); o[3] = clampf(o[3], 0.0f, hh);
            }
        }
    }
}

/* ------------------------------------------------------------------------ */
/* D4: torchvision nms (retinanet/models.py:210).  Stable sort by score desc, */
/* greedy, suppress when inter/(area_i+area_j-inter) > thr.  keep[] receives   */
/* indices into the input in score order; returns the count.                   */
/* ------------------------------------------------------------------------ */
typedef struct { float s; int64_t i; } rno_si;

static void merge_sort_desc(rno_si *v, rno_si *tmp, int64_t n)
{
    if (n < 2) return;
    int64_t h = n / 2;
    merge_sort_desc(v, tmp, h);
    merge_sort_desc(v + h, tmp, n - h);
    int64_t i = 0, j = h, k = 0;
    while (i < h && j < n) tmp[k++] = (v[j].s > v[i].s) ? v[j++] : v[i++];   /* stable */
    while (i < h) tmp[k++] = v[i++];
    while (j < n) tmp[k++] = v[j++];
    memcpy(v, tmp, sizeof(rno_si) * (size_t)n);
}

RNO_API int64_t rno_nms(const float *boxes, const float *scores, int64_t n, float thr, int64_t *keep)
{
    if (n <= 0) return 0;
    rno_si *ord = (rno_si *)malloc(sizeof(rno_si) * (size_t)n * 2);
    float *area = (float *)malloc(sizeof(float) * (size_t)n);
    unsigned char *sup = (unsigned char *)calloc((size_t)n, 1);
    for (int64_t i = 0; i < n; ++i) {
        ord[i].s = scores[i]; ord[i].i = i;
        area[i] = (boxes[i * 4 + 2] - boxes[i * 4 + 0]) * (boxes[i * 4 + 3] - boxes[i * 4 + 1]);
    }
    merge_sort_desc(ord, ord + n, n);
    int64_t nk = 0;
    for (int64_t oi = 0; oi < n; ++oi) {
        const int64_t i = ord[oi].i;
        if (sup[i]) continue;
        keep[nk++] = i;
        const float *bi = boxes + i * 4;
        for (int64_t oj = oi + 1; oj < n; ++oj) {
            const int64_t j = ord[oj].i;
            if (sup[j]) continue;
            const float *bj = boxes + j * 4;
            float xx1 = bi[0] > bj[0] ? bi[0] : bj[0];
            float yy1 = bi[1] > bj[1] ? bi[1] : bj[1];
            float xx2 = bi[2] < bj[2] ? bi[2] : bj[2];
            float yy2 = bi[3] < bj[3] ? bi[3] : bj[3];
            float w = xx2 - xx1; w = w > 0.0f ? w : 0.0f;
            float h = yy2 - yy1; h = h > 0.0f ? h : 0.0f;
            float inter = w * h;
            float ovr = inter / ((area[i] + area[j]) - inter);
            if (ovr > thr) sup[j] = 1;
        }
    }
    free(ord); free(area); free(sup);
    return nk;
}

/* ------------------------------------------------------------------------ */
/* D3+D5: process_detections  retinanet/models.py:160-243                      */
/* sigmoid (:170) -> decode+clip (:187-189) -> per class: score>thr (:196),    */
/* remove_small_boxes (>=, :203), nms (:210) -> concat class-major (:222-224)  */
/* -> labels+1 (:230) -> stable sort desc, first max_det (:234-240).           */
/* Outputs are padded to max_det rows per image; out_count[b] = valid rows.    */
/* ------------------------------------------------------------------------ */
RNO_API void rno_detect(const float *cls, const float *deltas, int B, int64_t A, int K,
                        const float *anchors, int64_t anchor_bstride, const int32_t *image_hw,
                        const rno_detect_params *p,
                        float *out_boxes, float *out_scores, int64_t *out_labels, int32_t *out_count)
{
#pragma omp parallel for schedule(dynamic, 1)
    for (int b = 0; b < B; ++b) {
        float *boxes = (float *)malloc(sizeof(float) * 4 * (size_t)A);
        rno_decode_clip(deltas + (int64_t)b * A * 4, 1, A, anchors + (int64_t)b * anchor_bstride, 0,
                        image_hw ? image_hw + b * 2 : NULL, p->reg_w, boxes);
        float *cb = (float *)malloc(sizeof(float) * 4 * (size_t)A);
        float *cs = (float *)malloc(sizeof(float) * (size_t)A);
        int64_t *keep = (int64_t *)malloc(sizeof(int64_t) * (size_t)A);
        int64_t cap = 1024, n_all = 0;
        float *ab = (float *)malloc(sizeof(float) * 4 * (size_t)cap);
        float *as = (float *)malloc(sizeof(float) * (size_t)cap);
        int64_t *al = (int64_t *)malloc(sizeof(int64_t) * (size_t)cap);
        for (int k = 0; k < K; ++k) {
            int64_t n = 0;
            for (int64_t a = 0; a < A; ++a) {
                const float x = cls[((int64_t)b * A + a) * K + k];
                const float s = 1.0f / (1.0f + expf(-x));
                if (!(s > p->score_thr)) continue;
                const float *bx = boxes + a * 4;
                const float ws = bx[2] - bx[0], hs = bx[3] - bx[1];
                if (!(ws >= p->min_box && hs >= p->min_box)) continue;
                memcpy(cb + n * 4, bx, sizeof(float) * 4);
                cs[n] = s;
                ++n;
            }
            const int64_t nk = rno_nms(cb, cs, n, p->nms_thr, keep);
            if (n_all + nk > cap) {
                while (n_all + nk > cap) cap *= 2;
                ab = (float *)realloc(ab, sizeof(float) * 4 * (size_t)cap);
                as = (float *)realloc(as, sizeof(float) * (size_t)cap);
                al = (int64_t *)realloc(al, sizeof(int64_t) * (size_t)cap);
            }
            for (int64_t i = 0; i < nk; ++i) {
                memcpy(ab + (n_all + i) * 4, cb + keep[i] * 4, sizeof(float) * 4);
                as[n_all + i] = cs[keep[i]];
                al[n_all + i] = (int64_t)k + 1;
            }
            n_all += nk;
        }
        rno_si *ord = (rno_si *)malloc(sizeof(rno_si) * (size_t)(n_all > 0 ? n_all : 1) * 2);
        for (int64_t i = 0; i < n_all; ++i) { ord[i].s = as[i]; ord[i].i = i; }
        merge_sort_desc(ord, ord + n_all, n_all);
        const int64_t nout = n_all < p->max_det ? n_all : p->max_det;
        for (int64_t i = 0; i < p->max_det; ++i) {
            float *ob = out_boxes + ((int64_t)b * p->max_det + i) * 4;
            if (i < nout) {
                memcpy(ob, ab + ord[i].i * 4, sizeof(float) * 4);
                out_scores[(int64_t)b * p->max_det + i] = as[ord[i].i];
                out_labels[(int64_t)b * p->max_det + i] = al[ord[i].i];
            } else {
                ob[0] = ob[1] = ob[2] = ob[3] = 0.0f;
                out_scores[(int64_t)b * p->max_det + i] = 0.0f;
                out_labels[(int64_t)b * p->max_det + i] = 0;
            }
        }
        out_count[b] = (int32_t)nout;
        free(ord); free(ab); free(as); free(al); free(keep); free(cs); free(cb); free(boxes);
    }
}

/* ------------------------------------------------------------------------ */
/* T1: torchvision GeneralizedRCNNTransform as the reference runs it         */
/* (retinanet/models.py:116, :262, :279): per image (x - mean) / std, then    */
/* F.interpolate(..., mode="bilinear", align_corners=False,                   */
/* recompute_scale_factor=True) to out_hw (the caller computes the sizes,     */
/* floor(size * scale) in double, like torchvision), then zero padding into   */
/* one [B][3][Hp][Wp] batch.  Sampling follows ATen's upsample_bilinear2d:    */
/* scale = in / out (fp32), src = scale * (dst + 0.5) - 0.5 clamped at 0,     */
/* taps i0 = (int)src and i0 + (i0 < in - 1), weights (1 - l, l).             */
/* ------------------------------------------------------------------------ */
static void tap_axis(int dst, int in, int out, int *i0, int *i1, float *l0, float *l1)
{
    const float scale = (in == out) ? 1.0f : (float)in / (float)out;
    float src = scale * ((float)dst + 0.5f) - 0.5f;
    if (src < 0.0f) src = 0.0f;
    int a = (int)src;
    if (a > in - 1) a = in - 1;
    *i0 = a;
    *i1 = a + (a < in - 1 ? 1 : 0);
    *l1 = src - (float)a;
    *l0 = 1.0f - *l1;
}

RNO_API void rno_transform_batch(const float *const *images, const int32_t *in_hw, const int32_t *out_hw, int B,
                                 const float *mean, const float *std, int Hp, int Wp, float *out /* [B][3][Hp][Wp] */)
{
    memset(out, 0, sizeof(float) * (size_t)B * 3 * (size_t)Hp * (size_t)Wp);
    for (int b = 0; b < B; ++b) {
        const int ih = in_hw[2 * b], iw = in_hw[2 * b + 1], oh = out_hw[2 * b], ow = out_hw[2 * b + 1];
        const float *img = images[b];
#pragma omp parallel for schedule(static)
        for (int y = 0; y < oh; ++y) {
            int y0, y1; float ly0, ly1;
            tap_axis(y, ih, oh, &y0, &y1, &ly0, &ly1);
            for (int x = 0; x < ow; ++x) {
                int x0, x1; float lx0, lx1;
                tap_axis(x, iw, ow, &x0, &x1, &lx0, &lx1);
                for (int c = 0; c < 3; ++c) {
                    const float *pl = img + (size_t)c * ih * iw;
                    const float m = mean[c], s = std[c];
                    const float p00 = (pl[(size_t)y0 * iw + x0] - m) / s, p01 = (pl[(size_t)y0 * iw + x1] - m) / s;
                    const float p10 = (pl[(size_t)y1 * iw + x0] - m) / s, p11 = (pl[(size_t)y1 * iw + x1] - m) / s;
                    out[(((size_t)b * 3 + c) * Hp + y) * Wp + x] = ly0 * (lx0 * p00 + lx1 * p01) + ly1 * (lx0 * p10 + lx1 * p11);
                }
            }
        }
    }
}
