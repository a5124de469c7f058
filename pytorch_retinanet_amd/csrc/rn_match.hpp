// Pair arithmetic of the IoU matcher (retinanet/box_utils.py:51-80 + torchvision box_iou), shared by K2 (match.hip) and by the
// loss kernel's fused matching prologue (loss.hip).  Bit-exactness with the CPU path: fp32 throughout, association
// (area_t + area_a) - inter, IEEE divide, NO FMA contraction -- match.hip is compiled with -ffp-contract=off, loss.hip is not, so the
// one function with a multiply that feeds an add / subtract (iou_pair) switches contraction off for itself.
#pragma once
#include "rn_common.hpp"

namespace rn_match {

// ---- careful pair: any input, torch's result bit for bit --------------------------------------------------
// The IEEE divide dominates, but most pairs do not overlap: when NO lane of the wave has a non-zero (or NaN)
// intersection the quotient is known without dividing -- 0/uni is +-0 for uni != 0 (the sign never matters to
// the comparisons) and NaN for uni == 0 or NaN.
__device__ __forceinline__ float iou_pair(const rn::f32x4 t, const float area_t, const rn::f32x4 a, const float area_a)
{
#pragma clang fp contract(off)
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
// inter = max(min(t.z, a.z) - max(t.x, a.x), 0) * max(min(t.w, a.w) - max(t.y, a.y), 0) in exactly 9 instructions.
// v_max_f32 / v_min_f32 give, for finite inputs, the same values as torch's max / min / clamp up to the sign of a zero.
// One asm block because fmaxf / fminf make the compiler re-canonicalise (v_max x, x, x) every loop-invariant operand
// inside the pair loop.  The GT box is wave-uniform and sits in SGPRs (one scalar operand per VALU instruction).
struct GtBox { float x, y, z, w, area; };           // wave-uniform
__device__ __forceinline__ float inter_fast(const GtBox t, const rn::f32x4 a)
{
    float inter, t0, t1;
    asm("v_min_f32 %0, %5, %9\n\t"
        "v_max_f32 %1, %3, %7\n\t"
        "v_sub_f32 %0, %0, %1\n\t"
        "v_min_f32 %1, %6, %10\n\t"
        "v_max_f32 %2, %4, %8\n\t"
        "v_sub_f32 %1, %1, %2\n\t"
        "v_max_f32 %0, 0, %0\n\t"
        "v_max_f32 %1, 0, %1\n\t"
        "v_mul_f32 %0, %0, %1"
        : "=&v"(inter), "=&v"(t0), "=&v"(t1)
        : "s"(t.x), "s"(t.y), "s"(t.z), "s"(t.w), "v"(a.x), "v"(a.y), "v"(a.z), "v"(a.w));
    return inter;
}

// GT boxes reach the pair loop through LDS -> one box per LANE (conflict-free ds_read_b128) -> v_readlane into SGPRs.
// Reading s_box[j] directly from every lane (a same-address ds_read_b128) is serviced at ~1 lane group per cycle on
// gfx950: measured 1250 cycles per GT box and wave with 24 waves per CU -- 10x the whole pair arithmetic.
__device__ __forceinline__ float lane_f(const float v, const int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
__device__ __forceinline__ GtBox gt_of_lane(const rn::f32x4 box, const float area, const int l)
{
    return GtBox{lane_f(box.x, l), lane_f(box.y, l), lane_f(box.z, l), lane_f(box.w, l), lane_f(area, l)};
}
__device__ __forceinline__ rn::f32x4 vec(const GtBox g) { return rn::f32x4{g.x, g.y, g.z, g.w}; }

// Bounding box of a wave's anchors, wave-uniform (x0 = min x1, y0 = min y1, x1 = max x2, y1 = max y2).  A GT box that does
// not properly intersect it has w <= 0 or h <= 0 against every anchor of the wave: all its pairs have inter == 0.
struct WaveBox { float x0, y0, x1, y1; };
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, RN_WAVE));
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, RN_WAVE));
    return v;
}
__device__ __forceinline__ bool may_overlap(const WaveBox bb, const GtBox g)
{
    return g.z > bb.x0 && bb.x1 > g.x && g.w > bb.y0 && bb.y1 > g.y;
}

__device__ __forceinline__ bool may_overlap(const WaveBox bb, const rn::f32x4 g)
{
    return g.z > bb.x0 && bb.x1 > g.x && g.w > bb.y0 && bb.y1 > g.y;
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

}  // namespace rn_match
