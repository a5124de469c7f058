// K4-K7 inference chain -- replaces Retinanet.process_detections
// (retinanet/models.py:160-243) and the torchvision ops it calls.
// Compiled with -ffp-contract=off: box arithmetic that feeds NMS decisions must
// round like the CPU path.
#include "rn_internal.hpp"

namespace {

// ---- K4 decode + clip -----------------------------------------------------------
// activ_2_bbox (retinanet/box_utils.py:37-48): centres from (dx,dy), sizes from
// exp(dx), exp(dy) (Q4: the reference uses [..., :2] for both), then
// clip_boxes_to_image (retinanet/models.py:189) with the resized, unpadded size.
template <int DT> struct delta4;
template <> struct delta4<RN_F32> {
    static __device__ __forceinline__ void ld(const void *p, int64_t row, float (&f)[4]) {
        const rn::f32x4 v = ((const rn::f32x4 *)p)[row];
        f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
    }
};
template <> struct delta4<RN_BF16> {
    static __device__ __forceinline__ void ld(const void *p, int64_t row, float (&f)[4]) {
        const rn::u32x2 v = ((const rn::u32x2 *)p)[row];
        f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
        f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
    }
};
template <> struct delta4<RN_F16> {
    static __device__ __forceinline__ void ld(const void *p, int64_t row, float (&f)[4]) {
        const rn::u32x2 v = ((const rn::u32x2 *)p)[row];
        f[0] = rn::half_lo(v.x); f[1] = rn::half_hi(v.x); f[2] = rn::half_lo(v.y); f[3] = rn::half_hi(v.y);
    }
};

struct RegW { float w[4]; };

__device__ __forceinline__ float clampf(float v, const float lo, const float hi)
{
    v = v < lo ? lo : v;
    return v > hi ? hi : v;
}

__device__ __forceinline__ rn::f32x4 decode_one(const float (&d)[4], const rn::f32x4 an, const RegW rw)
{
    const float dx = d[0] / rw.w[0], dy = d[1] / rw.w[1];
    const float acx = (an.x + an.z) / 2.0f, acy = (an.y + an.w) / 2.0f;
    const float aw = an.z - an.x, ah = an.w - an.y;
    const float cx = aw * dx + acx, cy = ah * dy + acy;
    const float w = aw * expf(dx), h = ah * expf(dy);
    rn::f32x4 o;
    o.x = cx - w / 2.0f; o.y = cy - h / 2.0f; o.z = cx + w / 2.0f; o.w = cy + h / 2.0f;
    return o;
}

template <int DT>
__global__ __launch_bounds__(256) void decode_clip_kernel(const void *__restrict__ deltas, const int64_t A, const int64_t R,
                                                          const rn::f32x4 *__restrict__ anchors, const int64_t anchor_bstride4,
                                                          const int32_t *__restrict__ image_hw, const RegW rw,
                                                          rn::f32x4 *__restrict__ out)
{
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < R; r += (int64_t)gridDim.x * blockDim.x) {
        const int b = (int)((uint32_t)r / (uint32_t)A);
        const int64_t ai = r - (int64_t)b * A;
        float d[4];
        delta4<DT>::ld(deltas, r, d);
        rn::f32x4 o = decode_one(d, anchors[(int64_t)b * anchor_bstride4 + ai], rw);
        if (image_hw) {
            const float hh = (float)image_hw[2 * b], ww = (float)image_hw[2 * b + 1];
            o.x = clampf(o.x, 0.0f, ww); o.z = clampf(o.z, 0.0f, ww);
            o.y = clampf(o.y, 0.0f, hh); o.w = clampf(o.w, 0.0f, hh);
        }
        out[r] = o;
    }
}


// ---- K5 score scan + candidate compaction ---------------------------------------
// Streams the class logits once (A*K*s bytes per image, 16-byte loads).  An element is
// a candidate when sigmoid(x) > score_thr (models.py:196) and its decoded box passes
// remove_small_boxes (models.py:203; boxes are per anchor, so the size test commutes
// with the per-class loop).  Candidates (~4e-4 of the elements at the reference's
// prior) are staged in LDS and flushed with ONE global atomic per (block, image).
constexpr int SCAN_THREADS = 256;
constexpr int SCAN_CAP = 2560;   // > SCAN_THREADS*8 (one iteration's worst case) with room to batch flushes

struct ScanArgs {
    const void *cls;
    const rn::f32x4 *boxes;
    int64_t A, R, rows_per_block, C;
    int32_t K, B;
    uint32_t magicK;
    float score_thr, pre_thr, min_box;
    uint64_t *cand;          // [B][C]  (inv_ordered(score) << 32) | (anchor*K + k)
    int32_t *cand_count;     // [B]
    int32_t *seg_count;      // [B][K]
};

struct ScanShared {
    uint64_t key[SCAN_CAP];
    uint16_t img[SCAN_CAP];
    int count, nb, base, fill;
};

__device__ __forceinline__ void scan_flush(ScanShared &sh, const ScanArgs &a, const int b_lo, const int b_hi)
{
    __syncthreads();
    const int n = min(sh.count, SCAN_CAP);
    for (int b = b_lo; b <= b_hi; ++b) {
        if (threadIdx.x == 0) { sh.nb = 0; sh.fill = 0; }
        __syncthreads();
        int local = 0;
        for (int i = threadIdx.x; i < n; i += SCAN_THREADS) local += (sh.img[i] == (uint16_t)(b - b_lo)) ? 1 : 0;
        if (local) atomicAdd(&sh.nb, local);
        __syncthreads();
        if (threadIdx.x == 0) sh.base = sh.nb ? atomicAdd(&a.cand_count[b], sh.nb) : 0;
        __syncthreads();
        const int base = sh.base;
        for (int i = threadIdx.x; i < n; i += SCAN_THREADS) {
            if (sh.img[i] != (uint16_t)(b - b_lo)) continue;
            const int64_t pos = (int64_t)base + atomicAdd(&sh.fill, 1);
            if (pos < a.C) {
                const uint64_t key = sh.key[i];
                a.cand[(int64_t)b * a.C + pos] = key;
                atomicAdd(&a.seg_count[(int64_t)b * a.K + (int)((uint32_t)key % (uint32_t)a.K)], 1);
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) sh.count = 0;
    __syncthreads();
}

__device__ __forceinline__ void scan_elem(ScanShared &sh, const ScanArgs &a, const float x, const uint32_t le,
                                          const int64_t r0, const int b_lo)
{
    if (!(x > a.pre_thr)) return;
    const float s = 1.0f / (1.0f + expf(-x));                 // models.py:170
    if (!(s > a.score_thr)) return;
    const uint32_t row = (a.K == 1) ? le : __umulhi(le, a.magicK);
    const uint32_t k = le - row * (uint32_t)a.K;
    const int64_t r = r0 + row;
    const int b = (int)((uint32_t)r / (uint32_t)a.A);
    const uint32_t anchor = (uint32_t)(r - (int64_t)b * a.A);
    const rn::f32x4 bx = a.boxes[r];
    if (!((bx.z - bx.x) >= a.min_box && (bx.w - bx.y) >= a.min_box)) return;
    const int idx = atomicAdd(&sh.count, 1);
    if (idx < SCAN_CAP) {
        sh.key[idx] = ((uint64_t)rn::inv_ordered(s) << 32) | (uint32_t)(anchor * (uint32_t)a.K + k);
        sh.img[idx] = (uint16_t)(b - b_lo);
    }
}

template <int DT>
__global__ __launch_bounds__(SCAN_THREADS) void score_scan_kernel(const ScanArgs a)
{
    typedef rn::dt<DT> D;
    constexpr int VEC = D::VEC;
    __shared__ ScanShared sh;
    const int64_t r0 = (int64_t)blockIdx.x * a.rows_per_block;
    const int64_t r1 = min(r0 + a.rows_per_block, a.R);
    if (r0 >= r1) return;
    const int b_lo = (int)((uint32_t)r0 / (uint32_t)a.A), b_hi = (int)((uint32_t)(r1 - 1) / (uint32_t)a.A);
    if (threadIdx.x == 0) sh.count = 0;
    __syncthreads();
    const int64_t e0 = r0 * a.K;
    const uint32_t ne = (uint32_t)((r1 - r0) * a.K);
    const uint32_t nvec = ne / VEC;
    const rn::u32x4 *src = (const rn::u32x4 *)((const typename D::elem *)a.cls + e0);
    const rn::u32x4 zero4 = {0u, 0u, 0u, 0u};
    rn::u32x4 nxt = (threadIdx.x < nvec) ? src[threadIdx.x] : zero4;
    for (uint32_t vb = 0; vb < nvec; vb += SCAN_THREADS) {          // block-uniform trip count
        const uint32_t v = vb + threadIdx.x;
        const rn::u32x4 raw = nxt;
        nxt = (v + SCAN_THREADS < nvec) ? src[v + SCAN_THREADS] : zero4;
        if (v < nvec) {
            float x[VEC];
            D::unpack(raw, x);
#pragma unroll
            for (int j = 0; j < VEC; ++j) scan_elem(sh, a, x[j], v * VEC + j, r0, b_lo);
        }
        __syncthreads();
        if (sh.count > SCAN_CAP - SCAN_THREADS * VEC) scan_flush(sh, a, b_lo, b_hi);   // uniform decision
    }
    const uint32_t le = nvec * VEC + threadIdx.x;                      // ragged tail (< VEC elements)
    if (le < ne) scan_elem(sh, a, D::ld(a.cls, e0 + le), le, r0, b_lo);
    scan_flush(sh, a, b_lo, b_hi);
}

// ---- per-image segment offsets -----------------------------------------------------
__global__ __launch_bounds__(64) void seg_offsets_kernel(const int32_t *__restrict__ cand_count, const int32_t *__restrict__ seg_count,
                                                         const int K, const int64_t C, int64_t *__restrict__ seg_start,
                                                         int32_t *__restrict__ seg_len, int32_t *__restrict__ out_status)
{
    const int b = blockIdx.x;
    if (threadIdx.x != 0) return;
    int64_t run = (int64_t)b * C;
    for (int k = 0; k < K; ++k) {
        const int c = seg_count[(int64_t)b * K + k];
        seg_start[(int64_t)b * K + k] = run;
        seg_len[(int64_t)b * K + k] = c;
        run += c;
    }
    out_status[b] = cand_count[b] > C ? 1 : 0;
}

// ---- scatter candidates into their (image, class) segments ---------------------------
__global__ __launch_bounds__(256) void seg_scatter_kernel(const uint64_t *__restrict__ cand, const int32_t *__restrict__ cand_count,
                                                          const int K, const int64_t C, const int64_t *__restrict__ seg_start,
                                                          int32_t *__restrict__ seg_fill, uint64_t *__restrict__ seg)
{
    const int b = blockIdx.y;
    const int64_t n = min((int64_t)cand_count[b], C);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t key = cand[(int64_t)b * C + i];
        const uint32_t ak = (uint32_t)key;
        const uint32_t anchor = ak / (uint32_t)K, k = ak - anchor * (uint32_t)K;
        const int pos = atomicAdd(&seg_fill[(int64_t)b * K + k], 1);
        seg[seg_start[(int64_t)b * K + k] + pos] = (key & 0xffffffff00000000ull) | anchor;
    }
}

// ---- K7 class-major merge + top max_det (models.py:222-240) ----------------------------
// Final order = stable sort by score desc of the class-major concatenation, i.e. ascending
// (inv score, class, in-class rank); in-class rank order == (inv score, anchor) order, so the
// key (inv score << 32 | class*A + anchor) reproduces it.  Only the first max_det survivors of
// each class can reach the global top max_det.
constexpr int TOPK_THREADS = 1024;
constexpr int TOPK_CAP = 4096;

__global__ __launch_bounds__(TOPK_THREADS) void topk_kernel(const uint64_t *__restrict__ kept, const int64_t *__restrict__ seg_start,
                                                            const int32_t *__restrict__ kept_count, const rn::f32x4 *__restrict__ boxes,
                                                            const int K, const int64_t A, const int max_det,
                                                            rn::f32x4 *__restrict__ out_boxes, float *__restrict__ out_scores,
                                                            int64_t *__restrict__ out_labels, int32_t *__restrict__ out_count)
{
    __shared__ uint64_t s_key[TOPK_CAP];
    const int b = blockIdx.x;
    int cur = 0;
    for (int k = 0; k < K; ++k) {
        const int m = min(kept_count[(int64_t)b * K + k], max_det);
        if (m == 0) continue;
        if (cur + m > TOPK_CAP) {            // compact to the best max_det so far (uniform branch)
            __syncthreads();
            for (int i = cur + threadIdx.x; i < TOPK_CAP; i += TOPK_THREADS) s_key[i] = ~0ull;
            __syncthreads();
            for (int kk = 2; kk <= TOPK_CAP; kk <<= 1)
                for (int j = kk >> 1; j > 0; j >>= 1) {
                    for (int i = threadIdx.x; i < TOPK_CAP; i += TOPK_THREADS) {
                        const int p = i ^ j;
                        if (p > i) {
                            const uint64_t x = s_key[i], y = s_key[p];
                            if ((x > y) == ((i & kk) == 0)) { s_key[i] = y; s_key[p] = x; }
                        }
                    }
                    __syncthreads();
                }
            cur = min(cur, max_det);
        }
        const int64_t st = seg_start[(int64_t)b * K + k];
        for (int i = threadIdx.x; i < m; i += TOPK_THREADS) {
            const uint64_t key = kept[st + i];
            s_key[cur + i] = (key & 0xffffffff00000000ull) | (uint32_t)((uint32_t)k * (uint32_t)A + (uint32_t)key);
        }
        cur += m;
    }
    __syncthreads();
    for (int i = cur + threadIdx.x; i < TOPK_CAP; i += TOPK_THREADS) s_key[i] = ~0ull;
    __syncthreads();
    for (int kk = 2; kk <= TOPK_CAP; kk <<= 1)
        for (int j = kk >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < TOPK_CAP; i += TOPK_THREADS) {
                const int p = i ^ j;
                if (p > i) {
                    const uint64_t x = s_key[i], y = s_key[p];
                    if ((x > y) == ((i & kk) == 0)) { s_key[i] = y; s_key[p] = x; }
                }
            }
            __syncthreads();
        }
    const int nout = min(cur, max_det);
    for (int i = threadIdx.x; i < max_det; i += TOPK_THREADS) {
        const int64_t o = (int64_t)b * max_det + i;
        if (i < nout) {
            const uint64_t key = s_key[i];
            const uint32_t ka = (uint32_t)key;
            const uint32_t k = ka / (uint32_t)A, anchor = ka - k * (uint32_t)A;
            out_boxes[o] = boxes[(int64_t)b * A + anchor];
            out_scores[o] = rn::score_of((uint32_t)(key >> 32));
            out_labels[o] = (int64_t)k + 1;                       // models.py:230
        } else {
            const rn::f32x4 z = {0.f, 0.f, 0.f, 0.f};
            out_boxes[o] = z; out_scores[o] = 0.0f; out_labels[o] = 0;
        }
    }
    if (threadIdx.x == 0) out_count[b] = nout;
}

size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

struct DetectWs {
    rn::f32x4 *boxes; uint64_t *cand, *seg; rn::f32x4 *sbox; uint8_t *supp;
    int32_t *cand_count, *seg_count, *seg_fill, *kept_count, *seg_len; int64_t *seg_start;
    size_t zero_bytes, total;
};

DetectWs carve(void *base, int B, int64_t A, int K, int64_t C)
{
    DetectWs w;
    unsigned char *p = (unsigned char *)base;
    size_t off = 0;
    auto take = [&](size_t bytes) { unsigned char *q = p ? p + off : nullptr; off += al256(bytes); return q; };
    const size_t BK = (size_t)B * K, BC = (size_t)B * C;
    w.boxes = (rn::f32x4 *)take((size_t)B * A * 16);
    w.cand = (uint64_t *)take(BC * 8);
    w.seg = (uint64_t *)take(BC * 8);
    w.sbox = (rn::f32x4 *)take(BC * 16);
    w.supp = (uint8_t *)take(BC);
    // zeroed every call: cand_count | seg_count | seg_fill (contiguous)
    const size_t z0 = off;
    w.cand_count = (int32_t *)take((size_t)B * 4);
    w.seg_count = (int32_t *)take(BK * 4);
    w.seg_fill = (int32_t *)take(BK * 4);
    w.zero_bytes = off - z0;
    w.kept_count = (int32_t *)take(BK * 4);
    w.seg_len = (int32_t *)take(BK * 4);
    w.seg_start = (int64_t *)take(BK * 8);
    w.total = off;
    return w;
}

}  // namespace

RN_API int rn_decode_clip(const void *deltas, int dtype, int B, int64_t A, const float *anchors, int64_t anchor_bstride,
                          const int32_t *image_hw, const float reg_w[4], float *out, void *stream)
{
    if (!deltas || !anchors || !out || !reg_w || B <= 0 || A <= 0) return RN_EINVAL;
    const int64_t R = (int64_t)B * A;
    if (R >= ((int64_t)1 << 31)) return RN_EUNSUPPORTED;
    if (!rn::aligned(deltas, dtype == RN_F32 ? 16 : 8) || !rn::aligned(anchors, 16) || !rn::aligned(out, 16) ||
        (anchor_bstride & 3))
        return RN_EALIGN;
    RegW rw = {{reg_w[0], reg_w[1], reg_w[2], reg_w[3]}};
    int64_t blocks = (R + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    const dim3 g((unsigned)blocks), blk(256);
    hipStream_t st = (hipStream_t)stream;
    switch (dtype) {
        case RN_F32: hipLaunchKernelGGL((decode_clip_kernel<RN_F32>), g, blk, 0, st, deltas, A, R, (const rn::f32x4 *)anchors, anchor_bstride / 4, image_hw, rw, (rn::f32x4 *)out); break;
        case RN_BF16: hipLaunchKernelGGL((decode_clip_kernel<RN_BF16>), g, blk, 0, st, deltas, A, R, (const rn::f32x4 *)anchors, anchor_bstride / 4, image_hw, rw, (rn::f32x4 *)out); break;
        case RN_F16: hipLaunchKernelGGL((decode_clip_kernel<RN_F16>), g, blk, 0, st, deltas, A, R, (const rn::f32x4 *)anchors, anchor_bstride / 4, image_hw, rw, (rn::f32x4 *)out); break;
        default: return RN_EINVAL;
    }
    RN_LAUNCH_CHECK();
    return RN_OK;
}

RN_API size_t rn_detect_workspace_bytes(int B, int64_t A, int K, int64_t max_candidates)
{
    if (B <= 0 || A <= 0 || K <= 0 || max_candidates <= 0) return 0;
    return carve(nullptr, B, A, K, max_candidates).total;
}

RN_API int rn_detect(const void *cls, const void *deltas, int dtype, int B, int64_t A, int K, const float *anchors,
                     int64_t anchor_bstride, const int32_t *image_hw, const rn_detect_params *params,
                     int64_t max_candidates, float *out_boxes, float *out_scores, int64_t *out_labels,
                     int32_t *out_count, int32_t *out_status, void *workspace, size_t workspace_bytes, void *stream)
{
    if (!cls || !deltas || !anchors || !params || !out_boxes || !out_scores || !out_labels || !out_count || !out_status ||
        !workspace)
        return RN_EINVAL;
    if (B <= 0 || A <= 0 || K <= 0 || max_candidates <= 0 || params->max_det <= 0) return RN_EINVAL;
    const int64_t R = (int64_t)B * A, C = max_candidates;
    // payloads are 32-bit: anchor*K + k and class*A + anchor; scan tags images with 16 bits per block
    if (R >= ((int64_t)1 << 31) || A * (int64_t)K >= ((int64_t)1 << 32) || K > 4096 || params->max_det > TOPK_THREADS ||
        (int64_t)B * C >= ((int64_t)1 << 40) || (int64_t)B * K >= ((int64_t)1 << 31))
        return RN_EUNSUPPORTED;
    if (workspace_bytes < rn_detect_workspace_bytes(B, A, K, C)) return RN_EWORKSPACE;
    if (!rn::aligned(cls, 16) || !rn::aligned(workspace, 256) || !rn::aligned(out_boxes, 16)) return RN_EALIGN;
    hipStream_t st = (hipStream_t)stream;
    DetectWs w = carve(workspace, B, A, K, C);

    int rc = rn_decode_clip(deltas, dtype, B, A, anchors, anchor_bstride, image_hw, params->reg_w, (float *)w.boxes, stream);
    if (rc != RN_OK) return rc;
    RN_HIP(hipMemsetAsync(w.cand_count, 0, w.zero_bytes, st));

    ScanArgs sa;
    sa.cls = cls; sa.boxes = w.boxes; sa.A = A; sa.R = R; sa.C = C; sa.K = K; sa.B = B;
    sa.magicK = (K > 1) ? (uint32_t)(((uint64_t)1 << 32) / (uint64_t)K) + 1u : 0u;
    sa.score_thr = params->score_thr;
    sa.min_box = params->min_box;
    {   // logit-space pre-filter with a safety margin; the exact test is still sigmoid(x) > thr
        const double t = (double)params->score_thr;
        sa.pre_thr = (t > 0.0 && t < 1.0) ? (float)(log(t / (1.0 - t)) - 1e-2) : (t <= 0.0 ? -INFINITY : 40.0f);
        if (t >= 1.0) sa.pre_thr = INFINITY;
    }
    sa.cand = w.cand; sa.cand_count = w.cand_count; sa.seg_count = w.seg_count;
    // contiguous multiple-of-8 row ranges per block; few enough rows that <= 65535 images... and magic-div range holds
    int64_t rpb = (R + 4095) / 4096;
    rpb = ((rpb + 7) / 8) * 8;
    const int64_t max_rpb = (((int64_t)1 << 32) / ((int64_t)K * K)) & ~(int64_t)7;   // le < 2^32/K for the magic divide
    if (rpb > max_rpb) rpb = max_rpb > 8 ? max_rpb : 8;
    sa.rows_per_block = rpb;
    const unsigned blocks = (unsigned)((R + rpb - 1) / rpb);
    if ((rpb + A - 1) / A + 1 > 65535) return RN_EUNSUPPORTED;
    switch (dtype) {
        case RN_F32: hipLaunchKernelGGL((score_scan_kernel<RN_F32>), dim3(blocks), dim3(SCAN_THREADS), 0, st, sa); break;
        case RN_BF16: hipLaunchKernelGGL((score_scan_kernel<RN_BF16>), dim3(blocks), dim3(SCAN_THREADS), 0, st, sa); break;
        case RN_F16: hipLaunchKernelGGL((score_scan_kernel<RN_F16>), dim3(blocks), dim3(SCAN_THREADS), 0, st, sa); break;
        default: return RN_EINVAL;
    }
    RN_LAUNCH_CHECK();
    hipLaunchKernelGGL(seg_offsets_kernel, dim3((unsigned)B), dim3(64), 0, st, w.cand_count, w.seg_count, K, C, w.seg_start,
                       w.seg_len, out_status);
    RN_LAUNCH_CHECK();
    {
        int64_t sb = (C + 255) / 256;
        if (sb > 1024) sb = 1024;
        hipLaunchKernelGGL(seg_scatter_kernel, dim3((unsigned)sb, (unsigned)B), dim3(256), 0, st, w.cand, w.cand_count, K, C,
                           w.seg_start, w.seg_fill, w.seg);
        RN_LAUNCH_CHECK();
    }
    rn::NmsLaunch na;
    na.keys = w.seg; na.kept = w.cand; na.keep_idx = nullptr; na.boxes = w.boxes;
    na.seg_start = w.seg_start; na.seg_len = w.seg_len; na.kept_count = w.kept_count;
    na.scratch_boxes = w.sbox; na.scratch_supp = w.supp;
    na.S = B * K; na.box_mode = 1; na.K = K; na.A = A; na.iou_thr = params->nms_thr;
    rc = rn::launch_nms(na, st);
    if (rc != RN_OK) return rc;
    hipLaunchKernelGGL(topk_kernel, dim3((unsigned)B), dim3(TOPK_THREADS), 0, st, w.cand, w.seg_start, w.kept_count, w.boxes,
                       K, A, params->max_det, (rn::f32x4 *)out_boxes, out_scores, out_labels, out_count);
    RN_LAUNCH_CHECK();
    return RN_OK;
}
