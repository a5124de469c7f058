// K4-K7 inference chain -- replaces Retinanet.process_detections
// (retinanet/models.py:160-243) and the torchvision ops it calls.
// Compiled with -ffp-contract=off: box arithmetic that feeds NMS decisions must
// round like the CPU path.
#include "rn_common.hpp"

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
