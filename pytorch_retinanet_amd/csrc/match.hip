// K2 iou_match -- replaces matcher() (retinanet/box_utils.py:51-80) and the
// torchvision box_iou it calls (:74).  The [T,A] IoU matrix is never written:
// each thread owns one anchor (registers), the image's GT boxes are staged
// through LDS in tiles, and the running (max IoU, first arg-max) pair is reduced
// in-register.  HBM traffic per image: A*16 B anchors (L2-resident across images
// when shared) + T*16 B GT + A*8 B int64 matches.
//
// Bit-exactness with the CPU path: fp32 throughout, association
// (area_t + area_a) - inter, IEEE divide, no FMA contraction (this file is
// compiled with -ffp-contract=off), first index wins ties, NaN propagates as in
// torch.max (first NaN wins, and a NaN max is neither < bg nor > fg -> -2).
#include "rn_common.hpp"

namespace {

constexpr int MATCH_BLOCK = 256;
constexpr int GT_TILE = 256;

// IoU of one (GT, anchor) pair, bit-for-bit the CPU sequence.  The IEEE divide is ~15 instructions
// and dominates the kernel, but most pairs do not overlap: when NO lane of the wave has a non-zero
// (or NaN) intersection the quotient is known without dividing -- 0/uni is +-0 for uni != 0 (the sign
// never matters to the comparisons below) and NaN for uni == 0 or NaN -- so the divide sits behind a
// wave-uniform branch.
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

__global__ __launch_bounds__(MATCH_BLOCK) void iou_match_kernel(
    const rn::f32x4 *__restrict__ anchors, const int64_t anchor_bstride4,
    const rn::f32x4 *__restrict__ gt, const int32_t *__restrict__ gt_off, const int64_t A,
    const float fg_thr, const float bg_thr, int64_t *__restrict__ matches, int32_t *__restrict__ num_fg)
{
    __shared__ rn::f32x4 s_box[GT_TILE];
    __shared__ float s_area[GT_TILE];

    const int b = blockIdx.y;
    const int t0 = gt_off[b];
    const int T = gt_off[b + 1] - t0;
    const int64_t a_idx = (int64_t)blockIdx.x * MATCH_BLOCK + threadIdx.x;
    const bool live = a_idx < A;

    rn::f32x4 an = {0.f, 0.f, 0.f, 0.f};
    if (live) an = anchors[(int64_t)b * anchor_bstride4 + a_idx];
    const float area_a = (an.z - an.x) * (an.w - an.y);

    float best = 0.0f;
    int bi = 0;
    bool have = false;
    for (int base = 0; base < T; base += GT_TILE) {
        const int n = min(GT_TILE, T - base);
        __syncthreads();
        if ((int)threadIdx.x < n) {
            const rn::f32x4 g = gt[t0 + base + threadIdx.x];
            s_box[threadIdx.x] = g;
            s_area[threadIdx.x] = (g.z - g.x) * (g.w - g.y);
        }
        __syncthreads();
        for (int j = 0; j < n; ++j) {
            const float v = iou_pair(s_box[j], s_area[j], an, area_a);
            if (!have) {
                best = v; bi = base + j; have = true;
            } else if (best == best && (v > best || v != v)) {
                best = v; bi = base + j;
            }
        }
    }

    int64_t r = -2;
    if (T > 0) {
        if (best < bg_thr) r = -1;
        if (best > fg_thr) r = bi;
    }
    if (live) matches[(int64_t)b * A + a_idx] = r;

    if (num_fg) {
        const unsigned long long fg = __ballot(live && r >= 0);
        if ((threadIdx.x & (RN_WAVE - 1)) == 0 && fg) atomicAdd(&num_fg[b], __popcll(fg));
    }
}

}  // namespace

RN_API int rn_iou_match(const float *anchors, int64_t anchor_bstride, const float *gt_boxes, const int32_t *gt_off,
                        int B, int64_t A, float fg_thr, float bg_thr, int64_t *matches, int32_t *num_fg, void *stream)
{
    if (!anchors || !gt_off || !matches || B <= 0 || A <= 0 || B > 65535) return RN_EINVAL;
    if (!(fg_thr > bg_thr)) return RN_ETHRESH;
    if (!rn::aligned(anchors, 16) || (gt_boxes && !rn::aligned(gt_boxes, 16)) || (anchor_bstride & 3)) return RN_EALIGN;
    hipStream_t st = (hipStream_t)stream;
    if (num_fg) RN_HIP(hipMemsetAsync(num_fg, 0, sizeof(int32_t) * (size_t)B, st));
    const dim3 grid((unsigned)((A + MATCH_BLOCK - 1) / MATCH_BLOCK), (unsigned)B);
    hipLaunchKernelGGL(iou_match_kernel, grid, dim3(MATCH_BLOCK), 0, st, (const rn::f32x4 *)anchors, anchor_bstride / 4,
                       (const rn::f32x4 *)gt_boxes, gt_off, A, fg_thr, bg_thr, matches, num_fg);
    RN_LAUNCH_CHECK();
    return RN_OK;
}
