// T1 transform_batch -- replaces the torchvision GeneralizedRCNNTransform the reference runs at
// retinanet/models.py:116, :262, :279 (normalise -> bilinear resize -> zero-padded batch), i.e. per
// image ~2 elementwise kernels + interpolate + a strided copy, plus the batch memset, the
// channels_last re-layout and autocast's fp32->bf16 cast of the conv1 input -- as ONE launch:
//
//   out[b][c][y][x] = y < oh_b && x < ow_b ? bilinear((in_b - mean_c) / std_c)(y, x) : 0
//
// Sampling is torch's upsample_bilinear2d with align_corners=False and the scale recomputed from
// the integer sizes (recompute_scale_factor=True): src = (dst + 0.5) * (in / out) - 0.5, clamped
// at 0, taps (i0, min(i0 + 1, in - 1)), weights (1 - l, l); each tap is normalised before the blend,
// like the reference (normalise first, then resize).  When in == out the weights are exactly
// (1, 0), so the identity case is a single tap.  HBM-bound: reads sum_b 3*h_b*w_b*4 bytes (each
// input pixel is touched by at most 4 neighbouring outputs, served by L1/L2), writes
// B*3*Hp*Wp*s bytes, s = output element size.
//
// Each thread produces 4 consecutive x of one row for all 3 channels, so every output layout gets
// 8- or 16-byte stores: NCHW (f32: 3 x 16 B, 16-bit: 3 x 8 B) or channels-last NHWC (f32: 3 x 16 B,
// 16-bit: 3 x 8 B contiguous).
#include "rn_common.hpp"

namespace {

constexpr int TB_MAX_IMAGES = 64;      // per launch (kernarg table)
constexpr int TB_PX = 4;

struct TransformArgs {
    const float *img[TB_MAX_IMAGES];   // [3][h][w] f32, contiguous
    int32_t ih[TB_MAX_IMAGES], iw[TB_MAX_IMAGES], oh[TB_MAX_IMAGES], ow[TB_MAX_IMAGES];
    float mean[3], std[3];
    int32_t B, Hp, Wp;
    void *out;                         // first image of this launch
};

__device__ __forceinline__ void tap_axis(const int dst, const int in, const int out, int &i0, int &i1, float &l0, float &l1)
{
    if (in == out) { i0 = i1 = dst; l0 = 1.0f; l1 = 0.0f; return; }
    const float scale = (float)in / (float)out;
    float src = scale * ((float)dst + 0.5f) - 0.5f;
    src = src < 0.0f ? 0.0f : src;
    i0 = (int)src;
    i0 = i0 < in - 1 ? i0 : in - 1;
    i1 = i0 + ((i0 < in - 1) ? 1 : 0);
    l1 = src - (float)i0;
    l0 = 1.0f - l1;
}

template <int DT> struct store4;
template <> struct store4<RN_F32> {
    static __device__ __forceinline__ void st(void *p, int64_t elem, const float (&v)[4]) {
        rn::f32x4 o; o.x = v[0]; o.y = v[1]; o.z = v[2]; o.w = v[3];
        *(rn::f32x4 *)((float *)p + elem) = o;
    }
};
template <> struct store4<RN_BF16> {
    static __device__ __forceinline__ void st(void *p, int64_t elem, const float (&v)[4]) {
        rn::u32x2 o; o.x = rn::dt<RN_BF16>::pk(v[0], v[1]); o.y = rn::dt<RN_BF16>::pk(v[2], v[3]);
        *(rn::u32x2 *)((uint16_t *)p + elem) = o;
    }
};
template <> struct store4<RN_F16> {
    static __device__ __forceinline__ void st(void *p, int64_t elem, const float (&v)[4]) {
        rn::u32x2 o; o.x = rn::dt<RN_F16>::pk(v[0], v[1]); o.y = rn::dt<RN_F16>::pk(v[2], v[3]);
        *(rn::u32x2 *)((uint16_t *)p + elem) = o;
    }
};

template <int DT, bool NHWC>
__global__ __launch_bounds__(256) void transform_batch_kernel(const TransformArgs a)
{
    const int b = blockIdx.z;
    const int y = blockIdx.y;
    const int x0 = (blockIdx.x * blockDim.x + threadIdx.x) * TB_PX;
    if (x0 >= a.Wp) return;
    const int ih = a.ih[b], iw = a.iw[b], oh = a.oh[b], ow = a.ow[b];
    float v[3][TB_PX];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int p = 0; p < TB_PX; ++p) v[c][p] = 0.0f;

    if (y < oh && x0 < ow) {
        const float *__restrict__ src = a.img[b];
        const int64_t plane = (int64_t)ih * iw;
        int y0, y1; float ly0, ly1;
        tap_axis(y, ih, oh, y0, y1, ly0, ly1);
        if (ih == oh && iw == ow) {                             // identity: one tap
#pragma unroll
            for (int p = 0; p < TB_PX; ++p) {
                const int x = x0 + p;
                if (x < ow) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) v[c][p] = (src[c * plane + (int64_t)y * iw + x] - a.mean[c]) / a.std[c];
                }
            }
        } else {
#pragma unroll
            for (int p = 0; p < TB_PX; ++p) {
                const int x = x0 + p;
                if (x < ow) {
                    int xa, xb; float lx0, lx1;
                    tap_axis(x, iw, ow, xa, xb, lx0, lx1);
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const float *pl = src + c * plane;
                        const float m = a.mean[c], s = a.std[c];
                        const float p00 = (pl[(int64_t)y0 * iw + xa] - m) / s, p01 = (pl[(int64_t)y0 * iw + xb] - m) / s;
                        const float p10 = (pl[(int64_t)y1 * iw + xa] - m) / s, p11 = (pl[(int64_t)y1 * iw + xb] - m) / s;
                        v[c][p] = ly0 * (lx0 * p00 + lx1 * p01) + ly1 * (lx0 * p10 + lx1 * p11);
                    }
                }
            }
        }
    }
    const int64_t HW = (int64_t)a.Hp * a.Wp;
    if (NHWC) {                                                 // [b][y][x][c]: 12 contiguous elements
        const int64_t e = ((int64_t)b * HW + (int64_t)y * a.Wp + x0) * 3;
        const float q0[4] = {v[0][0], v[1][0], v[2][0], v[0][1]};
        const float q1[4] = {v[1][1], v[2][1], v[0][2], v[1][2]};
        const float q2[4] = {v[2][2], v[0][3], v[1][3], v[2][3]};
        store4<DT>::st(a.out, e, q0); store4<DT>::st(a.out, e + 4, q1); store4<DT>::st(a.out, e + 8, q2);
    } else {                                                    // [b][c][y][x]
#pragma unroll
        for (int c = 0; c < 3; ++c) store4<DT>::st(a.out, ((int64_t)b * 3 + c) * HW + (int64_t)y * a.Wp + x0, v[c]);
    }
}

}  // namespace

RN_API int rn_transform_batch(const void *const *images, const int32_t *in_hw, const int32_t *out_hw, int B,
                              const float mean[3], const float std[3], int Hp, int Wp, void *out, int out_dtype,
                              int channels_last, void *stream)
{
    if (!images || !in_hw || !out_hw || !mean || !std || !out || B <= 0 || Hp <= 0 || Wp <= 0) return RN_EINVAL;
    if (out_dtype != RN_F32 && out_dtype != RN_BF16 && out_dtype != RN_F16) return RN_EINVAL;
    if (Wp % TB_PX) return RN_EUNSUPPORTED;
    if (Hp > 65535) return RN_EUNSUPPORTED;                     // gridDim.y
    if (!rn::aligned(out, 16)) return RN_EALIGN;
    for (int b = 0; b < B; ++b) {
        if (!images[b] || in_hw[2 * b] <= 0 || in_hw[2 * b + 1] <= 0 || out_hw[2 * b] <= 0 || out_hw[2 * b + 1] <= 0) return RN_EINVAL;
        if (out_hw[2 * b] > Hp || out_hw[2 * b + 1] > Wp) return RN_EINVAL;
        if (!rn::aligned(images[b], 4)) return RN_EALIGN;
        if (std[0] == 0.0f || std[1] == 0.0f || std[2] == 0.0f) return RN_EINVAL;
    }
    hipStream_t st = (hipStream_t)stream;
    const size_t esz = out_dtype == RN_F32 ? 4 : 2;
    for (int b0 = 0; b0 < B; b0 += TB_MAX_IMAGES) {
        TransformArgs a;
        a.B = (B - b0) < TB_MAX_IMAGES ? (B - b0) : TB_MAX_IMAGES;
        for (int i = 0; i < a.B; ++i) {
            a.img[i] = (const float *)images[b0 + i];
            a.ih[i] = in_hw[2 * (b0 + i)]; a.iw[i] = in_hw[2 * (b0 + i) + 1];
            a.oh[i] = out_hw[2 * (b0 + i)]; a.ow[i] = out_hw[2 * (b0 + i) + 1];
        }
        for (int c = 0; c < 3; ++c) { a.mean[c] = mean[c]; a.std[c] = std[c]; }
        a.Hp = Hp; a.Wp = Wp;
        a.out = (unsigned char *)out + (size_t)b0 * 3 * Hp * Wp * esz;
        const dim3 blk(256), grid((unsigned)((Wp / TB_PX + 255) / 256), (unsigned)Hp, (unsigned)a.B);
#define RN_TB_LAUNCH(DT)                                                                                  \
        if (channels_last) hipLaunchKernelGGL((transform_batch_kernel<DT, true>), grid, blk, 0, st, a);   \
        else hipLaunchKernelGGL((transform_batch_kernel<DT, false>), grid, blk, 0, st, a)
        switch (out_dtype) {
            case RN_F32: RN_TB_LAUNCH(RN_F32); break;
            case RN_BF16: RN_TB_LAUNCH(RN_BF16); break;
            default: RN_TB_LAUNCH(RN_F16); break;
        }
#undef RN_TB_LAUNCH
        RN_LAUNCH_CHECK();
    }
    return RN_OK;
}
