// K2 iou_match -- replaces matcher() (retinanet/box_utils.py:51-80) and the
// torchvision box_iou it calls (:74).  The [T,A] IoU matrix is never written:
// anchors live in registers, the images' GT boxes (+ their areas) are staged in
// LDS, and the running (max IoU, first arg-max) pair is reduced in-register.
//
// Two kernels:
//   * iou_match_batch_kernel -- the train-step shape (one anchor set shared by the batch, a few GT per
//     image): a thread owns ONE anchor for ALL images, so the anchor is loaded once per batch, the GT of the
//     whole batch sit in LDS, and a launch is A/256 workgroups (788 at A = 201 600) instead of B*A/256.
//     HBM traffic per batch: A*16 B anchors + sum(T)*16 B GT + B*A*8 B int64 matches.
//   * iou_match_tile_kernel<R> -- everything else (T up to thousands, per-image anchors): a thread owns R
//     anchors (strided by the workgroup size: coalesced 16-byte loads, 8-byte stores); GT tiles of 256 are
//     staged in LDS once per workgroup and every LDS read of a GT box is used for R pairs.
//
// Per pair the common case costs 12 VALU instructions: when every GT box of the tile and every anchor of the
// wave is a proper finite box (x2 > x1, y2 > y1; anchors may be degenerate) the union is positive, so the
// quotient of a pair WITHOUT overlap is +0 and cannot change (max, first arg-max) -- the IEEE divide (~12
// instructions) and the update sit behind a wave-uniform branch taken only when some lane overlaps.  Anything
// else (NaN / Inf / inverted boxes) takes the careful loop, which evaluates torch's semantics pair by pair.
//
// Bit-exactness with the CPU path: fp32 throughout, association
// (area_t + area_a) - inter, IEEE divide, no FMA contraction (this file is
// compiled with -ffp-contract=off), first index wins ties, NaN propagates as in
// torch.max (first NaN wins, and a NaN max is neither < bg nor > fg -> -2).
#include "rn_common.hpp"

namespace {

constexpr int MATCH_BLOCK = 256;
constexpr int GT_TILE = 256;
constexpr int BATCH_GT_MAX = 1024;          // sum(T) the batch kernel stages in LDS (20 KiB)

// ---- careful pair: any input, torch's result bit for bit --------------------------------------------------
// The IEEE divide dominates, but most pairs do not overlap: when NO lane of the wave has a non-zero (or NaN)
// intersection the quotient is known without dividing -- 0/uni is +-0 for uni != 0 (the sign never matters to
// the comparisons) and NaN for uni == 0 or NaN.
__device__ __forceinline__ float iou_pair(const rn::f32x4 t, const float area_t, const rn::f32x4 a, const float area_a)
{
    const float ltx = t.x > a.x ? t.x : a.x;
    const float lty = t.y > a.y ? t.y : a.y;
    const float rbx = t.z < a.z ? t.z : a.z;
    const float rby = t.w < a.w ? t.w : a.w;
    float w = rbx - ltx;
    if (!(w > 0.0f)) w = (w != w) ? w : 0.0f;
    float h = rby - lty;
    if (!(h > 0.0f)) h = (h != h) ? h : 0.0f;
    const float inter = w * h;
    const float uni = (area_t + area_a) - inter;
    if (__any(inter != 0.0f)) return inter / uni;          // (NaN != 0) is true: NaN takes the exact path
    return (uni != 0.0f && uni == uni) ? 0.0f : __builtin_nanf("");
}

struct Best { float v; int i; bool have; };

__device__ __forceinline__ void careful_update(Best &b, const float v, const int j)
{
    if (!b.have) {
        b.v = v; b.i = j; b.have = true;
    } else if (b.v == b.v && (v > b.v || v != v)) {
        b.v = v; b.i = j;
    }
}

// ---- fast pair: proper finite boxes only ------------------------------------------------------------------
__device__ __forceinline__ float vmaxf(const float a, const float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float vminf(const float a, const float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

__device__ __forceinline__ float inter_fast(const rn::f32x4 t, const rn::f32x4 a)
{
    const float w = vmaxf(vminf(t.z, a.z) - vmaxf(t.x, a.x), 0.0f);
    const float h = vmaxf(vminf(t.w, a.w) - vmaxf(t.y, a.y), 0.0f);
    return w * h;
}

__device__ __forceinline__ bool gt_is_proper(const rn::f32x4 g, const float area)
{
    return (g.z - g.x) > 0.0f && (g.w - g.y) > 0.0f && area < __builtin_inff();
}
__device__ __forceinline__ bool anchor_is_proper(const rn::f32x4 a, const float area)
{
    return (a.z - a.x) >= 0.0f && (a.w - a.y) >= 0.0f && area < __builtin_inff();
}

__device__ __forceinline__ int64_t classify(const float best, const int bi, const int T, const float fg_thr, const float bg_thr)
{
    int64_t r = -2;
    if (T > 0) {
        if (best < bg_thr) r = -1;
        if (best > fg_thr) r = bi;
    }
    return r;
}

// ============================================================================================================
// Shared anchors, small GT sets: one thread = one anchor x all images of the batch.
__global__ __launch_bounds__(MATCH_BLOCK) void iou_match_batch_kernel(
    const rn::f32x4 *__restrict__ anchors, const rn::f32x4 *__restrict__ gt, const int32_t *__restrict__ gt_off,
    const int B, const int64_t A, const float fg_thr, const float bg_thr, int64_t *__restrict__ matches,
    int32_t *__restrict__ num_fg)
{
    __shared__ rn::f32x4 s_box[BATCH_GT_MAX];
    __shared__ float s_area[BATCH_GT_MAX];
    __shared__ int s_off[65];
    __shared__ int s_bad;

    const int tid = threadIdx.x;
    if (tid == 0) s_bad = 0;
    if (tid <= B) s_off[tid] = gt_off[tid] - gt_off[0];
    __syncthreads();
    const int total = min(s_off[B], BATCH_GT_MAX), g0 = gt_off[0];      // (the host promised total <= BATCH_GT_MAX)
    bool ok = true;
    for (int j = tid; j < total; j += MATCH_BLOCK) {
        const rn::f32x4 g = gt[g0 + j];
        const float ar = (g.z - g.x) * (g.w - g.y);
        s_box[j] = g;
        s_area[j] = ar;
        ok = ok && gt_is_proper(g, ar);
    }
    if (!ok) s_bad = 1;                                    // benign race: every writer stores 1
    __syncthreads();

    const int64_t a_idx = (int64_t)blockIdx.x * MATCH_BLOCK + tid;
    const bool live = a_idx < A;
    rn::f32x4 an = {0.f, 0.f, 0.f, 0.f};
    if (live) an = anchors[a_idx];
    const float area_a = (an.z - an.x) * (an.w - an.y);
    const bool fast = !s_bad && __all(anchor_is_proper(an, area_a));

    for (int b = 0; b < B; ++b) {
        const int j0 = s_off[b], T = s_off[b + 1] - j0;
        float best = 0.0f;
        int bi = 0;
        if (fast) {
            for (int j = 0; j < T; ++j) {
                const rn::f32x4 g = s_box[j0 + j];
                const float inter = inter_fast(g, an);
                if (__any(inter != 0.0f)) {
                    const float v = inter / ((s_area[j0 + j] + area_a) - inter);
                    if (v > best) { best = v; bi = j; }
                }
            }
        } else {
            Best bb = {0.0f, 0, false};
            for (int j = 0; j < T; ++j) careful_update(bb, iou_pair(s_box[j0 + j], s_area[j0 + j], an, area_a), j);
            best = bb.v; bi = bb.i;
        }
        const int64_t r = classify(best, bi, T, fg_thr, bg_thr);
        if (live) matches[(int64_t)b * A + a_idx] = r;
        if (num_fg) {
            const unsigned long long fg = __ballot(live && r >= 0);
            if ((tid & (RN_WAVE - 1)) == 0 && fg) atomicAdd(&num_fg[b], __popcll(fg));
        }
    }
}

// ============================================================================================================
// General shape: R anchors per thread, GT tiles in LDS.
template <int R>
__global__ __launch_bounds__(MATCH_BLOCK) void iou_match_tile_kernel(
    const rn::f32x4 *__restrict__ anchors, const int64_t anchor_bstride4,
    const rn::f32x4 *__restrict__ gt, const int32_t *__restrict__ gt_off, const int64_t A,
    const float fg_thr, const float bg_thr, int64_t *__restrict__ matches, int32_t *__restrict__ num_fg)
{
    __shared__ rn::f32x4 s_box[GT_TILE];
    __shared__ float s_area[GT_TILE];
    __shared__ int s_bad[2];

    const int tid = threadIdx.x;
    const int b = blockIdx.y;
    const int t0 = gt_off[b];
    const int T = gt_off[b + 1] - t0;
    const int64_t a0 = (int64_t)blockIdx.x * (MATCH_BLOCK * R) + tid;

    rn::f32x4 an[R];
    float area_a[R];
    bool a_ok = true;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int64_t a_idx = a0 + (int64_t)r * MATCH_BLOCK;
        an[r] = rn::f32x4{0.f, 0.f, 0.f, 0.f};
        if (a_idx < A) an[r] = anchors[(int64_t)b * anchor_bstride4 + a_idx];
        area_a[r] = (an[r].z - an[r].x) * (an[r].w - an[r].y);
        a_ok = a_ok && anchor_is_proper(an[r], area_a[r]);
    }
    const bool wave_ok = __all(a_ok);

    Best best[R];
#pragma unroll
    for (int r = 0; r < R; ++r) best[r] = Best{0.0f, 0, false};

    if (tid < 2) s_bad[tid] = 0;
    for (int base = 0, it = 0; base < T; base += GT_TILE, ++it) {
        const int n = min(GT_TILE, T - base);
        __syncthreads();                                   // previous tile fully consumed; s_bad[it & 1] reset below is ordered
        if (tid == 0) s_bad[(it + 1) & 1] = 0;
        if (tid < n) {
            const rn::f32x4 g = gt[t0 + base + tid];
            const float ar = (g.z - g.x) * (g.w - g.y);
            s_box[tid] = g;
            s_area[tid] = ar;
            if (!gt_is_proper(g, ar)) s_bad[it & 1] = 1;
        }
        __syncthreads();
        if (wave_ok && !s_bad[it & 1]) {
            // once a tile has been processed here, `have` only means "best/bi hold torch's running result so far";
            // with proper boxes every quotient is >= +0, so starting from (0, index 0) and updating on strict > is exact
            // (a negative running maximum can only come from an earlier careful tile: this tile's first pair beats it)
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if (!best[r].have) best[r] = Best{0.0f, 0, true};
                else if (best[r].v < 0.0f) best[r] = Best{0.0f, base, true};
            }
#pragma unroll 2
            for (int j = 0; j < n; ++j) {
                const rn::f32x4 g = s_box[j];
                float inter[R];
                bool any = false;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    inter[r] = inter_fast(g, an[r]);
                    any = any || inter[r] != 0.0f;
                }
                if (__any(any)) {
                    const float ga = s_area[j];
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        // a NaN best (from an earlier careful tile) stays: (v > NaN) is false
                        const float v = inter[r] / ((ga + area_a[r]) - inter[r]);
                        if (v > best[r].v) { best[r].v = v; best[r].i = base + j; }
                    }
                }
            }
        } else {
            for (int j = 0; j < n; ++j) {
                const rn::f32x4 g = s_box[j];
                const float ga = s_area[j];
#pragma unroll
                for (int r = 0; r < R; ++r) careful_update(best[r], iou_pair(g, ga, an[r], area_a[r]), base + j);
            }
        }
    }

    int nfg = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int64_t a_idx = a0 + (int64_t)r * MATCH_BLOCK;
        const int64_t m = classify(best[r].v, best[r].i, T, fg_thr, bg_thr);
        if (a_idx < A) {
            matches[(int64_t)b * A + a_idx] = m;
            nfg += m >= 0 ? 1 : 0;
        }
    }
    if (num_fg) {
        nfg = rn::wave_sum_i(nfg);
        if ((tid & (RN_WAVE - 1)) == 0 && nfg) atomicAdd(&num_fg[b], nfg);
    }
}

}  // namespace

RN_API int rn_iou_match(const float *anchors, int64_t anchor_bstride, const float *gt_boxes, const int32_t *gt_off,
                        int B, int64_t A, float fg_thr, float bg_thr, int64_t *matches, int32_t *num_fg, void *stream)
{
    return rn_iou_match_ex(anchors, anchor_bstride, gt_boxes, gt_off, B, A, fg_thr, bg_thr, matches, num_fg, -1, stream);
}

RN_API int rn_iou_match_ex(const float *anchors, int64_t anchor_bstride, const float *gt_boxes, const int32_t *gt_off,
                           int B, int64_t A, float fg_thr, float bg_thr, int64_t *matches, int32_t *num_fg,
                           int64_t total_gt, void *stream)
{
    if (!anchors || !gt_off || !matches || B <= 0 || A <= 0 || B > 65535) return RN_EINVAL;
    if (!(fg_thr > bg_thr)) return RN_ETHRESH;
    if (!rn::aligned(anchors, 16) || (gt_boxes && !rn::aligned(gt_boxes, 16)) || (anchor_bstride & 3)) return RN_EALIGN;
    hipStream_t st = (hipStream_t)stream;
    if (num_fg) RN_HIP(hipMemsetAsync(num_fg, 0, sizeof(int32_t) * (size_t)B, st));
    // The batch kernel needs host knowledge of sum(T) (gt_off lives on the device and this call never syncs):
    // callers that know it pass total_gt >= 0; -1 means unknown -> the general kernel.
    if (anchor_bstride == 0 && B <= 64 && total_gt >= 0 && total_gt <= BATCH_GT_MAX && total_gt <= 32 * (int64_t)B) {
        const dim3 grid((unsigned)((A + MATCH_BLOCK - 1) / MATCH_BLOCK));
        hipLaunchKernelGGL(iou_match_batch_kernel, grid, dim3(MATCH_BLOCK), 0, st, (const rn::f32x4 *)anchors,
                           (const rn::f32x4 *)gt_boxes, gt_off, B, A, fg_thr, bg_thr, matches, num_fg);
    } else if (total_gt >= 0 && total_gt <= 32 * (int64_t)B) {
        const dim3 grid((unsigned)((A + MATCH_BLOCK * 2 - 1) / (MATCH_BLOCK * 2)), (unsigned)B);
        hipLaunchKernelGGL(iou_match_tile_kernel<2>, grid, dim3(MATCH_BLOCK), 0, st, (const rn::f32x4 *)anchors, anchor_bstride / 4,
                           (const rn::f32x4 *)gt_boxes, gt_off, A, fg_thr, bg_thr, matches, num_fg);
    } else {
        const dim3 grid((unsigned)((A + MATCH_BLOCK * 4 - 1) / (MATCH_BLOCK * 4)), (unsigned)B);
        hipLaunchKernelGGL(iou_match_tile_kernel<4>, grid, dim3(MATCH_BLOCK), 0, st, (const rn::f32x4 *)anchors, anchor_bstride / 4,
                           (const rn::f32x4 *)gt_boxes, gt_off, A, fg_thr, bg_thr, matches, num_fg);
    }
    RN_LAUNCH_CHECK();
    return RN_OK;
}
