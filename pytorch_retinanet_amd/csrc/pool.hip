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
    // A thread owns the 2 x 2 block of OUTPUT elements (2A .. 2A+1, 2B .. 2B+1): their windows span 5 x 5 input elements, which are
    // loaded (and, with AFF, normalised) once -- 25 loads for 4 outputs instead of 36.  Input elements are visited in row-major order,
    // which restricted to one window is that window's scan order: PyTorch's rule (first maximum, last NaN) is kept.
    const int HB = (s.OH + 1) >> 1, WB = (s.OW + 1) >> 1;
    const int64_t total = (int64_t)s.N * HB * WB * s.C8;
    // back to front: the producer (the stem's conv / BN apply, 275 MB) wrote the end of x last, the Infinity Cache still holds it
    for (int64_t ii = (int64_t)blockIdx.x * 256 + threadIdx.x; ii < total; ii += (int64_t)gridDim.x * 256) {
        const int64_t i = total - 1 - ii;
        int cg, B, A, n;
        split_index(i, total < (1ll << 31), s.C8, WB, HB, cg, B, A, n);
        if constexpr (AFF && DT != RN_F32) {
            // After the ReLU every value is a non-negative 16-bit float (a NaN input has become 0, as in bn_apply_kernel), so values
            // order like their bit patterns: key = bits << 16 | (15 - window position), one v_max_u32 per element and output keeps
            // the maximum and, among equal values, the FIRST position -- PyTorch's rule -- instead of a compare / select chain.
            uint32_t key[4][8];
            float ca[8], cb[8];
#pragma unroll
            for (int o = 0; o < 4; ++o)
#pragma unroll
                for (int j = 0; j < 8; ++j) key[o][j] = 0u;            // below every real key (position field >= 7)
#pragma unroll
            for (int j = 0; j < 8; ++j) { ca[j] = coef_a[cg * 8 + j]; cb[j] = coef_b[cg * 8 + j]; }
#pragma unroll
            for (int ri = 0; ri < 5; ++ri) {
                const int iy = 4 * A - 1 + ri;
                if (iy < 0 || iy >= s.H) continue;
#pragma unroll
                for (int ci = 0; ci < 5; ++ci) {
                    const int ix = 4 * B - 1 + ci;
                    if (ix < 0 || ix >= s.W) continue;
                    float f[8];
                    v8<DT>::ld(x, (((int64_t)n * s.H + iy) * s.W + ix) * s.C8 + cg, f);
                    uint32_t hb[8];                                     // value bits in the HIGH half
#pragma unroll
                    for (int j = 0; j < 8; j += 2) {
                        const float t0 = fmaf(f[j], ca[j], cb[j]), t1 = fmaf(f[j + 1], ca[j + 1], cb[j + 1]);
                        const uint32_t pkd = rn::dt<DT>::pk(t0 > 0.0f ? t0 : 0.0f, t1 > 0.0f ? t1 : 0.0f);
                        hb[j] = pkd << 16; hb[j + 1] = pkd & 0xffff0000u;
                    }
#pragma unroll
                    for (int o = 0; o < 4; ++o) {
                        const int r = ri - 2 * (o >> 1), q = ci - 2 * (o & 1);
                        if (r < 0 || r > 2 || q < 0 || q > 2) continue;
#pragma unroll
                        for (int j = 0; j < 8; ++j) { const uint32_t c = hb[j] | (uint32_t)(15 - (r * 3 + q)); key[o][j] = c > key[o][j] ? c : key[o][j]; }
                    }
                }
            }
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                const int oy = 2 * A + (o >> 1), ox = 2 * B + (o & 1);
                if (oy >= s.OH || ox >= s.OW) continue;
                const int64_t v = (((int64_t)n * s.OH + oy) * s.OW + ox) * s.C8 + cg;
                rn::u32x4 out;
                out.x = (key[o][0] >> 16) | (key[o][1] & 0xffff0000u); out.y = (key[o][2] >> 16) | (key[o][3] & 0xffff0000u);
                out.z = (key[o][4] >> 16) | (key[o][5] & 0xffff0000u); out.w = (key[o][6] >> 16) | (key[o][7] & 0xffff0000u);
                ((rn::u32x4 *)y)[v] = out;
                if (idx) {
                    rn::u32x2 pk;
                    pk.x = (15u - (key[o][0] & 15u)) | ((15u - (key[o][1] & 15u)) << 8) | ((15u - (key[o][2] & 15u)) << 16) | ((15u - (key[o][3] & 15u)) << 24);
                    pk.y = (15u - (key[o][4] & 15u)) | ((15u - (key[o][5] & 15u)) << 8) | ((15u - (key[o][6] & 15u)) << 16) | ((15u - (key[o][7] & 15u)) << 24);
                    ((rn::u32x2 *)idx)[v] = pk;
                }
            }
            continue;
        }
        float m[4][8], ca[8], cb[8];
        int k[4][8];
#pragma unroll
        for (int o = 0; o < 4; ++o)
#pragma unroll
            for (int j = 0; j < 8; ++j) { m[o][j] = -INFINITY; k[o][j] = -1; }
        if (AFF) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { ca[j] = coef_a[cg * 8 + j]; cb[j] = coef_b[cg * 8 + j]; }
        }
#pragma unroll
        for (int ri = 0; ri < 5; ++ri) {
            const int iy = 4 * A - 1 + ri;
            if (iy < 0 || iy >= s.H) continue;
#pragma unroll
            for (int ci = 0; ci < 5; ++ci) {
                const int ix = 4 * B - 1 + ci;
                if (ix < 0 || ix >= s.W) continue;
                float f[8];
                v8<DT>::ld(x, (((int64_t)n * s.H + iy) * s.W + ix) * s.C8 + cg, f);
                if (AFF) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) { const float t = fmaf(f[j], ca[j], cb[j]); f[j] = pool_round<DT>(t > 0.0f ? t : 0.0f); }
                }
#pragma unroll
                for (int o = 0; o < 4; ++o) {
                    const int oj = o >> 1, ok_ = o & 1, r = ri - 2 * oj, q = ci - 2 * ok_;     // position inside output o's window
                    if (r < 0 || r > 2 || q < 0 || q > 2) continue;                            // (compile-time after unrolling)
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        if (k[o][j] < 0 || f[j] > m[o][j] || f[j] != f[j]) { m[o][j] = f[j]; k[o][j] = r * 3 + q; }   // PyTorch's scan: first maximum, last NaN
                }
            }
        }
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            const int oy = 2 * A + (o >> 1), ox = 2 * B + (o & 1);
            if (oy >= s.OH || ox >= s.OW) continue;
            const int64_t v = (((int64_t)n * s.OH + oy) * s.OW + ox) * s.C8 + cg;
            v8<DT>::st(y, v, m[o]);
            if (idx) {
                rn::u32x2 pk;
                pk.x = (uint32_t)k[o][0] | ((uint32_t)k[o][1] << 8) | ((uint32_t)k[o][2] << 16) | ((uint32_t)k[o][3] << 24);
                pk.y = (uint32_t)k[o][4] | ((uint32_t)k[o][5] << 8) | ((uint32_t)k[o][6] << 16) | ((uint32_t)k[o][7] << 24);
                ((rn::u32x2 *)idx)[v] = pk;
            }
        }
    }
}

// dx of an input element = sum of dy over the (<= 4) windows whose arg-max code names it.  A thread owns the 2 x 2 block of input
// elements (2a .. 2a+1, 2b .. 2b+1): together they are covered by the SAME four windows (a .. a+1) x (b .. b+1), so their code words
// and gradient vectors are fetched once for four outputs (a thread per input element fetched them four times: 132 -> 7x us at the
// stem's shape).  Even rows / columns lie in one window row / column only, odd ones in two.  The sums run over the windows in
// (oy, ox) order as before: same values bit for bit.
// BN: the pooled tensor was relu(bn(z)) (the stem: retinanet/backbone.py:246-251) -- the two sums of that BatchNorm's backward over the
// gradient just formed are taken here (norm.hip's bn_bwd_partial_kernel<DT, 2>: g' = dx * [fma(z, fa, fb) alive]; sum g', sum g' * xhat),
// which saves that pass its read of dx.  256 % C8 == 0 and the grid stride is a multiple of 256: a thread's channel group is fixed.
struct PoolBnArgs { const void *z; const float *fa, *fb, *mean, *invstd; float *partial; };

template <int DT, bool BN = false>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const uint8_t *__restrict__ idx, const void *__restrict__ dy,
                                                          void *__restrict__ dx, const PoolShape s, const PoolBnArgs bn = PoolBnArgs{})
{
    const int HB = (s.H + 1) >> 1, WB = (s.W + 1) >> 1;
    const int64_t total = (int64_t)s.N * HB * WB * s.C8;
    float ssum[8], qsum[8], cfa[8], cfb[8], cmu[8], cis[8];
    if (BN) {
        const int cg0 = threadIdx.x % s.C8;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            ssum[j] = 0.0f; qsum[j] = 0.0f;
            cfa[j] = bn.fa[cg0 * 8 + j]; cfb[j] = bn.fb[cg0 * 8 + j]; cmu[j] = bn.mean[cg0 * 8 + j]; cis[j] = bn.invstd[cg0 * 8 + j];
        }
    }
    const float alive = DT == RN_F32 ? 0.0f : __uint_as_float(DT == RN_F16 ? 0x33000000u : 0x00004000u);      // norm.hip: relu_alive_threshold<DT>
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        int cg, b, a, n;
        split_index(i, total < (1ll << 31), s.C8, WB, HB, cg, b, a, n);
        // the four windows (clamped addresses, masked afterwards): all loads up front
        bool ok[4];
        rn::u32x2 pk[4];
        float gv[4][8];
        int64_t w[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int oy = a + (k >> 1), ox = b + (k & 1);
            ok[k] = oy < s.OH && ox < s.OW;
            const int oyc = oy < s.OH ? oy : s.OH - 1, oxc = ox < s.OW ? ox : s.OW - 1;
            w[k] = (((int64_t)n * s.OH + oyc) * s.OW + oxc) * s.C8 + cg;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) pk[k] = ((const rn::u32x2 *)idx)[w[k]];
#pragma unroll
        for (int k = 0; k < 4; ++k) v8<DT>::ld(dy, w[k], gv[k]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int ey = e >> 1, ex = e & 1, iy = 2 * a + ey, ix = 2 * b + ex;
            if (iy >= s.H || ix >= s.W) continue;
            float g[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) g[j] = 0.0f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int ky = k >> 1, kx = k & 1;
                // window row a + ky covers row iy = 2a + ey  <=>  ky == 0 or (ky == 1 and ey == 1); likewise for columns
                const bool covers = (ky == 0 || ey == 1) && (kx == 0 || ex == 1);
                if (!covers) continue;
                // position of (iy, ix) inside window (a + ky, b + kx): row iy - (2 (a + ky) - 1) = ey + 1 - 2 ky
                const uint32_t rep = (uint32_t)((ey + 1 - 2 * ky) * 3 + (ex + 1 - 2 * kx)) * 0x01010101u;
                const uint32_t d0 = pk[k].x ^ rep, d1 = pk[k].y ^ rep;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    g[j] += (ok[k] && ((d0 >> (8 * j)) & 0xffu) == 0u) ? gv[k][j] : 0.0f;
                    g[j + 4] += (ok[k] && ((d1 >> (8 * j)) & 0xffu) == 0u) ? gv[k][j + 4] : 0.0f;
                }
            }
            const int64_t vi = (((int64_t)n * s.H + iy) * s.W + ix) * s.C8 + cg;
            v8<DT>::st(dx, vi, g);
            if (BN) {
                float zz[8], gs[8];
                v8<DT>::ld(bn.z, vi, zz);
                if constexpr (DT != RN_F32) { rn::dt<DT>::unpack(rn::dt<DT>::pack(g), gs); }      // the sums are those of the stored gradient
                else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) gs[j] = g[j];
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float gj = (fmaf(zz[j], cfa[j], cfb[j]) > alive) ? gs[j] : 0.0f;
                    ssum[j] += gj;
                    qsum[j] = fmaf(gj, (zz[j] - cmu[j]) * cis[j], qsum[j]);
                }
            }
        }
    }
    if (BN) {
        __shared__ float red[256][17];
#pragma unroll
        for (int j = 0; j < 8; ++j) { red[threadIdx.x][j] = ssum[j]; red[threadIdx.x][8 + j] = qsum[j]; }
        __syncthreads();
        const int C = s.C8 * 8;
        for (int q = threadIdx.x; q < 2 * C; q += 256) {
            const int which = q / C, ch = q - which * C, cgq = ch >> 3, j = ch & 7;
            float t = 0.0f;
            for (int l = cgq; l < 256; l += s.C8) t += red[l][which * 8 + j];
            bn.partial[((int64_t)blockIdx.x * 2 + which) * C + ch] = t;
        }
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
    const dim3 g(pool_blocks((int64_t)N * ((s.OH + 1) / 2) * ((s.OW + 1) / 2) * s.C8)), b(256);      // a thread per 2 x 2 block of outputs
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
    const dim3 g(pool_blocks((int64_t)N * ((s.OH + 1) / 2) * ((s.OW + 1) / 2) * s.C8)), b(256);      // a thread per 2 x 2 block of outputs
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
    const dim3 g(pool_blocks((int64_t)N * ((H + 1) / 2) * ((W + 1) / 2) * s.C8)), b(256);        // a thread per 2 x 2 block of input elements
    hipStream_t st = (hipStream_t)stream;
    switch (dtype) {
        case RN_F32: hipLaunchKernelGGL((maxpool_bwd_kernel<RN_F32>), g, b, 0, st, argmax, dy, dx, s); break;
        case RN_BF16: hipLaunchKernelGGL((maxpool_bwd_kernel<RN_BF16>), g, b, 0, st, argmax, dy, dx, s); break;
        default: hipLaunchKernelGGL((maxpool_bwd_kernel<RN_F16>), g, b, 0, st, argmax, dy, dx, s); break;
    }
    RN_LAUNCH_CHECK();
    return RN_OK;
}

RN_API int rn_maxpool3x3s2_backward_bn_rows(int N, int H, int W, int C)
{
    if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 8 || 256 % (C / 8)) return 0;
    const int b = pool_blocks((int64_t)N * ((H + 1) / 2) * ((W + 1) / 2) * (C / 8));
    return b > 1024 ? 1024 : b;
}

RN_API int rn_maxpool3x3s2_backward_bn(const uint8_t *argmax, const void *dy, const void *z, const float *fwd_coef, const float *mean,
                                       const float *invstd, void *dx, float *partial, int dtype, int N, int H, int W, int C, void *stream)
{
    if (!argmax || !dy || !dx || !z || !fwd_coef || !mean || !invstd || !partial || N <= 0 || H <= 0 || W <= 0 || C <= 0) return RN_EINVAL;
    if (C % 8 || 256 % (C / 8)) return RN_EUNSUPPORTED;
    if (dtype != RN_BF16 && dtype != RN_F16) return RN_EUNSUPPORTED;
    if (!rn::aligned(argmax, 8) || !rn::aligned(dy, 16) || !rn::aligned(dx, 16) || !rn::aligned(z, 16)) return RN_EALIGN;
    const PoolShape s{N, H, W, C / 8, (H - 1) / 2 + 1, (W - 1) / 2 + 1};
    const PoolBnArgs bn{z, fwd_coef, fwd_coef + C, mean, invstd, partial};
    const dim3 g((unsigned)rn_maxpool3x3s2_backward_bn_rows(N, H, W, C)), b(256);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == RN_BF16) hipLaunchKernelGGL((maxpool_bwd_kernel<RN_BF16, true>), g, b, 0, st, argmax, dy, dx, s, bn);
    else hipLaunchKernelGGL((maxpool_bwd_kernel<RN_F16, true>), g, b, 0, st, argmax, dy, dx, s, bn);
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
