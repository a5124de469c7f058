// 3x3 / stride-2 / pad-1 max pooling of the ResNet stem (reference: retinanet/backbone.py:251, nn.MaxPool2d(3, 2, 1))
// for channels-last activations, forward and backward, with a one-byte arg-max code instead of an int64 index tensor.
//
// PyTorch's NHWC kernels write an int64 index per output element in the forward (4x the bytes of the bf16 output)
// and read it back in the backward; on the R50 stem ([8, 64, 400, 672] -> [8, 64, 200, 336]) they take 176 + 463 us.
// Here the forward stores the arg-max as ONE BYTE per output element (position 0..8 inside its window, PyTorch's
// scan rule `val > maxval || isnan(val)`: first maximum, last NaN) and the backward gathers: an input element sums
// dy over the <= 4 windows whose code names it.  Each thread owns 8 channels (one 16-byte vector) of one position.
// (Recomputing the arg-max in the backward from x and y was 3x SLOWER than PyTorch: every window maximum needs the
// scan for an earlier equal element, i.e. 8 more vector loads for most (element, window) pairs.)
//   forward : reads x (each line is touched by <= 4 windows, served by L1/L2), writes y and the codes
//   backward: reads the codes and dy of the <= 4 windows covering the position (cached), writes dx
#include "rn_common.hpp"

namespace {

template <int DT> struct v8 {
    static __device__ __forceinline__ void ld(const void *p, int64_t v, float (&f)[8]) { rn::dt<DT>::unpack(((const rn::u32x4 *)p)[v], f); }
    static __device__ __forceinline__ void st(void *p, int64_t v, const float (&f)[8]) { ((rn::u32x4 *)p)[v] = rn::dt<DT>::pack(f); }
};
template <> struct v8<RN_F32> {
    static __device__ __forceinline__ void ld(const void *p, int64_t v, float (&f)[8]) {
        const rn::f32x4 a = ((const rn::f32x4 *)p)[2 * v], b = ((const rn::f32x4 *)p)[2 * v + 1];
        f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
    }
    static __device__ __forceinline__ void st(void *p, int64_t v, const float (&f)[8]) {
        rn::f32x4 a, b;
        a.x = f[0]; a.y = f[1]; a.z = f[2]; a.w = f[3]; b.x = f[4]; b.y = f[5]; b.z = f[6]; b.w = f[7];
        ((rn::f32x4 *)p)[2 * v] = a; ((rn::f32x4 *)p)[2 * v + 1] = b;
    }
};

struct PoolShape { int N, H, W, C8, OH, OW; };

// (n, y, x, channel group) of flat element-group index i over [N][H][W][C8]: 32-bit divisions when the tensor has fewer than
// 2^31 groups (a 64-bit division by a run-time value is ~150 instructions on this ISA, three of them per thread dominated
// these kernels)
__device__ __forceinline__ void split_index(const int64_t i, const bool small, const int C8, const int W, const int H, int &cg, int &x,
                                            int &y, int &n)
{
    if (small) {
        const uint32_t u = (uint32_t)i;
        const uint32_t p = u / (uint32_t)C8, q = p / (uint32_t)W;
        cg = (int)(u - p * (uint32_t)C8);
        x = (int)(p - q * (uint32_t)W);
        n = (int)(q / (uint32_t)H);
        y = (int)(q - (uint32_t)n * (uint32_t)H);
    } else {
        cg = (int)(i % C8);
        int64_t p = i / C8;
        x = (int)(p % W); p /= W;
        y = (int)(p % H);
        n = (int)(p / H);
    }
}

// value of f after a store in DT and a load back (rn_common: bf16 RNE / f16)
template <int DT> __device__ __forceinline__ float pool_round(const float f);
template <> __device__ __forceinline__ float pool_round<RN_F32>(const float f) { return f; }
template <> __device__ __forceinline__ float pool_round<RN_BF16>(const float f) { return __uint_as_float(rn::dt<RN_BF16>::pk(f, 0.0f) << 16); }
template <> __device__ __forceinline__ float pool_round<RN_F16>(const float f) { return (float)(_Float16)f; }

// AFF: x is the INPUT of a BatchNorm + ReLU whose output was never written; every element read is first turned into
// round_DT(max(fma(x, a, b), 0)) -- what the apply pass would have stored -- so results and codes equal those of the two-pass form
// (the stem: one 275 MB read + 275 MB write less).
template <int DT, bool AFF = false>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const void *__restrict__ x, void *__restrict__ y, uint8_t *__restrict__ idx,
                                                          const PoolShape s, const float *__restrict__ coef_a = nullptr,
                                                          const float *__restrict__ coef_b = nullptr)
{
    const int64_t total = (int64_t)s.N * s.OH * s.OW * s.C8;
    // back to front: the producer (the stem's BN apply, 275 MB) wrote the end of x last, the Infinity Cache still holds it
    for (int64_t ii = (int64_t)blockIdx.x * 256 + threadIdx.x; ii < total; ii += (int64_t)gridDim.x * 256) {
        const int64_t i = total - 1 - ii;
        int cg, ox, oy, n;
        split_index(i, total < (1ll << 31), s.C8, s.OW, s.OH, cg, ox, oy, n);
        float m[8], ca[8], cb[8];
        int k[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { m[j] = -INFINITY; k[j] = -1; }
        if (AFF) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { ca[j] = coef_a[cg * 8 + j]; cb[j] = coef_b[cg * 8 + j]; }
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int iy = 2 * oy - 1 + r;
            if (iy < 0 || iy >= s.H) continue;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int ix = 2 * ox - 1 + q;
                if (ix < 0 || ix >= s.W) continue;
                float f[8];
                v8<DT>::ld(x, (((int64_t)n * s.H + iy) * s.W + ix) * s.C8 + cg, f);
                if (AFF) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) { const float t = fmaf(f[j], ca[j], cb[j]); f[j] = pool_round<DT>(t > 0.0f ? t : 0.0f); }
                }
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (k[j] < 0 || f[j] > m[j] || f[j] != f[j]) { m[j] = f[j]; k[j] = r * 3 + q; }   // PyTorch's scan: first maximum, last NaN
            }
        }
        v8<DT>::st(y, i, m);
        if (idx) {
            rn::u32x2 pk;
            pk.x = (uint32_t)k[0] | ((uint32_t)k[1] << 8) | ((uint32_t)k[2] << 16) | ((uint32_t)k[3] << 24);
            pk.y = (uint32_t)k[4] | ((uint32_t)k[5] << 8) | ((uint32_t)k[6] << 16) | ((uint32_t)k[7] << 24);
            ((rn::u32x2 *)idx)[i] = pk;
        }
    }
}

// dx of an input element = sum of dy over the (<= 4) windows whose arg-max code names it
template <int DT>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const uint8_t *__restrict__ idx, const void *__restrict__ dy,
                                                          void *__restrict__ dx, const PoolShape s)
{
    const int64_t total = (int64_t)s.N * s.H * s.W * s.C8;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        int cg, ix, iy, n;
        split_index(i, total < (1ll << 31), s.C8, s.W, s.H, cg, ix, iy, n);
        float g[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) g[j] = 0.0f;
        // windows (oy, ox) that contain (iy, ix): 2*o - 1 <= i <= 2*o + 1, at most 2 x 2 of them.  All four code words and
        // all four gradient vectors are fetched up front (clamped addresses, masked afterwards): no load waits on a compare
        const int oy0 = iy >> 1, oy1 = (iy + 1) >> 1, ox0 = ix >> 1, ox1 = (ix + 1) >> 1;
        int64_t w[4];
        bool ok[4];
        uint32_t rep[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int oy = oy0 + (k >> 1), ox = ox0 + (k & 1);
            ok[k] = oy <= oy1 && oy < s.OH && ox <= ox1 && ox < s.OW;
            const int oyc = ok[k] ? oy : oy0 < s.OH ? oy0 : s.OH - 1, oxc = ok[k] ? ox : ox0 < s.OW ? ox0 : s.OW - 1;
            w[k] = (((int64_t)n * s.OH + oyc) * s.OW + oxc) * s.C8 + cg;
            rep[k] = (uint32_t)((iy - (2 * oyc - 1)) * 3 + (ix - (2 * oxc - 1))) * 0x01010101u;
        }
        rn::u32x2 pk[4];
        float gv[4][8];
#pragma unroll
        for (int k = 0; k < 4; ++k) pk[k] = ((const rn::u32x2 *)idx)[w[k]];
#pragma unroll
        for (int k = 0; k < 4; ++k) v8<DT>::ld(dy, w[k], gv[k]);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            // byte-wise compare: zero bytes of (pk ^ rep) are the channels whose arg-max is this element
            const uint32_t d0 = pk[k].x ^ rep[k], d1 = pk[k].y ^ rep[k];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                g[j] += (ok[k] && ((d0 >> (8 * j)) & 0xffu) == 0u) ? gv[k][j] : 0.0f;
                g[j + 4] += (ok[k] && ((d1 >> (8 * j)) & 0xffu) == 0u) ? gv[k][j + 4] : 0.0f;
            }
        }
        v8<DT>::st(dx, i, g);
    }
}

// FPN top-down step (retinanet/layers.py:36,52-53: lateral + 2x nearest upsampling of the level above), channels-last:
//   out[n][y][x][c] = lat[n][y][x][c] + top[n][y / 2][x / 2][c]            (H = 2 Ht, W = 2 Wt)
// and its backward for `top`: dtop[n][yt][xt][c] = sum of the 2 x 2 block of g (summed in f32, rounded once); the lateral's
// gradient is g itself.  One pass each instead of an upsampling kernel + an add (forward) / an upsampling-backward kernel.
template <int DT>
__global__ __launch_bounds__(256) void add_up2x_kernel(const void *__restrict__ lat, const void *__restrict__ top, void *__restrict__ out,
                                                       const int N, const int H, const int W, const int C8)
{
    const int64_t total = (int64_t)N * H * W * C8;
    const int Wt = W >> 1, Ht = H >> 1;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        int cg, x, y, n;
        split_index(i, total < (1ll << 31), C8, W, H, cg, x, y, n);
        float a[8], b[8];
        v8<DT>::ld(lat, i, a);
        v8<DT>::ld(top, (((int64_t)n * Ht + (y >> 1)) * Wt + (x >> 1)) * C8 + cg, b);
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] += b[j];
        v8<DT>::st(out, i, a);
    }
}
template <int DT>
__global__ __launch_bounds__(256) void up2x_bwd_kernel(const void *__restrict__ g, void *__restrict__ dtop, const int N, const int Ht, const int Wt,
                                                       const int C8)
{
    const int64_t total = (int64_t)N * Ht * Wt * C8;
    const int W = Wt * 2, H = Ht * 2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        int cg, xt, yt, n;
        split_index(i, total < (1ll << 31), C8, Wt, Ht, cg, xt, yt, n);
        float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                float v[8];
                v8<DT>::ld(g, (((int64_t)n * H + 2 * yt + dy) * W + 2 * xt + dx) * C8 + cg, v);
#pragma unroll
                for (int j = 0; j < 8; ++j) s[j] += v[j];
            }
        v8<DT>::st(dtop, i, s);
    }
}

int pool_blocks(const int64_t n)
{
    int64_t b = (n + 255) / 256;
    if (b > 16384) b = 16384;
    return (int)(b < 1 ? 1 : b);
}

}  // namespace

RN_API int rn_maxpool3x3s2_forward(const void *x, void *y, uint8_t *argmax, int dtype, int N, int H, int W, int C, void *stream)
{
    if (!x || !y || N <= 0 || H <= 0 || W <= 0 || C <= 0) return RN_EINVAL;
    if (C % 8) return RN_EUNSUPPORTED;
    if (dtype != RN_F32 && dtype != RN_BF16 && dtype != RN_F16) return RN_EINVAL;
    if (!rn::aligned(x, 16) || !rn::aligned(y, 16) || (argmax && !rn::aligned(argmax, 8))) return RN_EALIGN;
    const PoolShape s{N, H, W, C / 8, (H - 1) / 2 + 1, (W - 1) / 2 + 1};
    const dim3 g(pool_blocks((int64_t)N * s.OH * s.OW * s.C8)), b(256);
    hipStream_t st = (hipStream_t)stream;
    switch (dtype) {
        case RN_F32: hipLaunchKernelGGL((maxpool_fwd_kernel<RN_F32>), g, b, 0, st, x, y, argmax, s, nullptr, nullptr); break;
        case RN_BF16: hipLaunchKernelGGL((maxpool_fwd_kernel<RN_BF16>), g, b, 0, st, x, y, argmax, s, nullptr, nullptr); break;
        default: hipLaunchKernelGGL((maxpool_fwd_kernel<RN_F16>), g, b, 0, st, x, y, argmax, s, nullptr, nullptr); break;
    }
    RN_LAUNCH_CHECK();
    return RN_OK;
}

RN_API int rn_bn_relu_maxpool3x3s2_forward(const void *x, const float *coef, void *y, uint8_t *argmax, int dtype, int N, int H, int W, int C,
                                           void *stream)
{
    if (!x || !y || !coef || N <= 0 || H <= 0 || W <= 0 || C <= 0) return RN_EINVAL;
    if (C % 8) return RN_EUNSUPPORTED;
    if (dtype != RN_F32 && dtype != RN_BF16 && dtype != RN_F16) return RN_EINVAL;
    if (!rn::aligned(x, 16) || !rn::aligned(y, 16) || (argmax && !rn::aligned(argmax, 8))) return RN_EALIGN;
    const PoolShape s{N, H, W, C / 8, (H - 1) / 2 + 1, (W - 1) / 2 + 1};
    const dim3 g(pool_blocks((int64_t)N * s.OH * s.OW * s.C8)), b(256);
    hipStream_t st = (hipStream_t)stream;
    switch (dtype) {
        case RN_F32: hipLaunchKernelGGL((maxpool_fwd_kernel<RN_F32, true>), g, b, 0, st, x, y, argmax, s, coef, coef + C); break;
        case RN_BF16: hipLaunchKernelGGL((maxpool_fwd_kernel<RN_BF16, true>), g, b, 0, st, x, y, argmax, s, coef, coef + C); break;
        default: hipLaunchKernelGGL((maxpool_fwd_kernel<RN_F16, true>), g, b, 0, st, x, y, argmax, s, coef, coef + C); break;
    }
    RN_LAUNCH_CHECK();
    return RN_OK;
}

RN_API int rn_maxpool3x3s2_backward(const uint8_t *argmax, const void *dy, void *dx, int dtype, int N, int H, int W, int C,
                                    void *stream)
{
    if (!argmax || !dy || !dx || N <= 0 || H <= 0 || W <= 0 || C <= 0) return RN_EINVAL;
    if (C % 8) return RN_EUNSUPPORTED;
    if (dtype != RN_F32 && dtype != RN_BF16 && dtype != RN_F16) return RN_EINVAL;
    if (!rn::aligned(argmax, 8) || !rn::aligned(dy, 16) || !rn::aligned(dx, 16)) return RN_EALIGN;
    const PoolShape s{N, H, W, C / 8, (H - 1) / 2 + 1, (W - 1) / 2 + 1};
    const dim3 g(pool_blocks((int64_t)N * H * W * s.C8)), b(256);
    hipStream_t st = (hipStream_t)stream;
    switch (dtype) {
        case RN_F32: hipLaunchKernelGGL((maxpool_bwd_kernel<RN_F32>), g, b, 0, st, argmax, dy, dx, s); break;
        case RN_BF16: hipLaunchKernelGGL((maxpool_bwd_kernel<RN_BF16>), g, b, 0, st, argmax, dy, dx, s); break;
        default: hipLaunchKernelGGL((maxpool_bwd_kernel<RN_F16>), g, b, 0, st, argmax, dy, dx, s); break;
    }
    RN_LAUNCH_CHECK();
    return RN_OK;
}

RN_API int rn_fpn_add_upsample2x(const void *lat, const void *top, void *out, int dtype, int N, int H, int W, int C, void *stream)
{
    if (!lat || !top || !out || N <= 0 || H <= 0 || W <= 0 || C <= 0) return RN_EINVAL;
    if (C % 8 || (H & 1) || (W & 1)) return RN_EUNSUPPORTED;
    if (dtype != RN_F32 && dtype != RN_BF16 && dtype != RN_F16) return RN_EINVAL;
    if (!rn::aligned(lat, 16) || !rn::aligned(top, 16) || !rn::aligned(out, 16)) return RN_EALIGN;
    const dim3 g(pool_blocks((int64_t)N * H * W * (C / 8))), b(256);
    hipStream_t st = (hipStream_t)stream;
    switch (dtype) {
        case RN_F32: hipLaunchKernelGGL((add_up2x_kernel<RN_F32>), g, b, 0, st, lat, top, out, N, H, W, C / 8); break;
        case RN_BF16: hipLaunchKernelGGL((add_up2x_kernel<RN_BF16>), g, b, 0, st, lat, top, out, N, H, W, C / 8); break;
        default: hipLaunchKernelGGL((add_up2x_kernel<RN_F16>), g, b, 0, st, lat, top, out, N, H, W, C / 8); break;
    }
    RN_LAUNCH_CHECK();
    return RN_OK;
}

RN_API int rn_fpn_upsample2x_backward(const void *g, void *dtop, int dtype, int N, int Ht, int Wt, int C, void *stream)
{
    if (!g || !dtop || N <= 0 || Ht <= 0 || Wt <= 0 || C <= 0) return RN_EINVAL;
    if (C % 8) return RN_EUNSUPPORTED;
    if (dtype != RN_F32 && dtype != RN_BF16 && dtype != RN_F16) return RN_EINVAL;
    if (!rn::aligned(g, 16) || !rn::aligned(dtop, 16)) return RN_EALIGN;
    const dim3 gr(pool_blocks((int64_t)N * Ht * Wt * (C / 8))), b(256);
    hipStream_t st = (hipStream_t)stream;
    switch (dtype) {
        case RN_F32: hipLaunchKernelGGL((up2x_bwd_kernel<RN_F32>), gr, b, 0, st, g, dtop, N, Ht, Wt, C / 8); break;
        case RN_BF16: hipLaunchKernelGGL((up2x_bwd_kernel<RN_BF16>), gr, b, 0, st, g, dtop, N, Ht, Wt, C / 8); break;
        default: hipLaunchKernelGGL((up2x_bwd_kernel<RN_F16>), gr, b, 0, st, g, dtop, N, Ht, Wt, C / 8); break;
    }
    RN_LAUNCH_CHECK();
    return RN_OK;
}
