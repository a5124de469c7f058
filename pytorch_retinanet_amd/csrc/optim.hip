// Multi-tensor SGD step with fp32 master weights and a bf16 working copy.
//
// Under bf16 autocast PyTorch casts every fp32 conv weight to bf16 in the forward (94 small kernels per step on
// R50-FPN) and every bf16 weight gradient back to fp32 in the backward (92 more), then runs torch.optim.SGD's
// foreach kernels.  Keeping the conv weights of the model in bf16, their fp32 masters and momentum buffers in the
// optimizer, and doing the whole update -- torch.optim.SGD's arithmetic, in its order, in fp32 on the master --
// plus the refresh of the bf16 copy in one launch per 48 tensors removes all of that (~1.3 ms of a 38 ms step).
//   g = float(grad) + weight_decay * w;  buf = first ? g : momentum * buf + (1 - dampening) * g;
//   g = nesterov ? g + momentum * buf : buf;  w -= lr * g;  w16 = bf16(w)
// The same kernel updates plain fp32 parameters (BN, biases): no 16-bit copy, fp32 gradient.
#include "rn_common.hpp"

namespace {

constexpr int SGD_MAX_TENSORS = 48;
constexpr int SGD_BLOCKS_X = 1024;

struct SgdTable {
    float *master[SGD_MAX_TENSORS];
    float *mom[SGD_MAX_TENSORS];
    const void *grad[SGD_MAX_TENSORS];
    void *p16[SGD_MAX_TENSORS];
    int64_t n[SGD_MAX_TENSORS];
    float lr, momentum, dampening, weight_decay;
    int nesterov, first, grad16;            // grad16: gradients of tensors WITH a 16-bit copy are 16-bit as well (else f32)
    int f16;                                // the 16-bit copies (and 16-bit gradients) are fp16 instead of bf16
    // fp16 autocast with torch.amp.GradScaler (an optimizer with _step_supports_amp_scaling): the gradients are grad_scale[0] times
    // too large, and the whole step is skipped when found_inf[0] != 0 -- both device scalars, so nothing synchronises (null: 1 / 0)
    const float *grad_scale, *found_inf;
};

__device__ __forceinline__ float sgd_one(const SgdTable &t, const float gin, float &wi, float &mi)
{
    float g = gin;
    if (t.weight_decay != 0.0f) g = g + t.weight_decay * wi;
    if (t.momentum != 0.0f) {
        const float b = t.first ? g : t.momentum * mi + (1.0f - t.dampening) * g;
        mi = b;
        g = t.nesterov ? g + t.momentum * b : b;
    }
    wi = wi - t.lr * g;
    return wi;
}

template <bool F16>
__global__ __launch_bounds__(256) void sgd_master_kernel(const SgdTable t)
{
    constexpr int DT = F16 ? RN_F16 : RN_BF16;
    if (t.found_inf && *t.found_inf != 0.0f) return;             // (GradScaler: a non-finite gradient somewhere -> no parameter moves)
    const float inv_scale = t.grad_scale ? 1.0f / *t.grad_scale : 1.0f;
    const int ti = blockIdx.y;
    float *__restrict__ w = t.master[ti];
    float *__restrict__ m = t.mom[ti];
    uint16_t *__restrict__ p16 = (uint16_t *)t.p16[ti];
    const bool g16 = p16 && t.grad16;
    const bool has_m = t.momentum != 0.0f;
    const int64_t n = t.n[ti], n4 = n >> 2;
    // 4 elements per thread and iteration: 16-byte accesses on the fp32 arrays, 8-byte on the bf16 ones
    // (torch allocations are 256-byte aligned and every array of a tensor starts at its storage offset 0 or a
    // multiple of 4 elements: checked on the host, which otherwise sends the tensor through the scalar tail path)
    for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < n4; v += (int64_t)gridDim.x * 256) {
        rn::f32x4 wv = ((const rn::f32x4 *)w)[v];
        rn::f32x4 mv = {0.f, 0.f, 0.f, 0.f};
        if (has_m && !t.first) mv = ((const rn::f32x4 *)m)[v];
        float g[4];
        if (g16) {
            const rn::u32x2 gv = ((const rn::u32x2 *)t.grad[ti])[v];
            g[0] = rn::mma<DT>::lo(gv.x); g[1] = rn::mma<DT>::hi(gv.x);
            g[2] = rn::mma<DT>::lo(gv.y); g[3] = rn::mma<DT>::hi(gv.y);
        } else {
            const rn::f32x4 gv = ((const rn::f32x4 *)t.grad[ti])[v];
            g[0] = gv.x; g[1] = gv.y; g[2] = gv.z; g[3] = gv.w;
        }
        float ww[4] = {wv.x, wv.y, wv.z, wv.w}, mm[4] = {mv.x, mv.y, mv.z, mv.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) sgd_one(t, t.grad_scale ? g[j] * inv_scale : g[j], ww[j], mm[j]);
        wv.x = ww[0]; wv.y = ww[1]; wv.z = ww[2]; wv.w = ww[3];
        mv.x = mm[0]; mv.y = mm[1]; mv.z = mm[2]; mv.w = mm[3];
        ((rn::f32x4 *)w)[v] = wv;
        if (has_m) ((rn::f32x4 *)m)[v] = mv;
        if (p16) {
            rn::u32x2 o;
            o.x = rn::dt<DT>::pk(ww[0], ww[1]); o.y = rn::dt<DT>::pk(ww[2], ww[3]);
            ((rn::u32x2 *)p16)[v] = o;
        }
    }
    if (blockIdx.x == 0) {                                       // < 4 leftover elements
        const int64_t i = n4 * 4 + threadIdx.x;
        if (threadIdx.x < 4 && i < n) {
            float wi = w[i], mi = (has_m && !t.first) ? m[i] : 0.0f;
            float g = g16 ? rn::mma<DT>::lo((uint32_t)((const uint16_t *)t.grad[ti])[i]) : ((const float *)t.grad[ti])[i];
            if (t.grad_scale) g *= inv_scale;
            sgd_one(t, g, wi, mi);
            w[i] = wi;
            if (has_m) m[i] = mi;
            if (p16) p16[i] = rn::mma<DT>::dn(wi);
        }
    }
}

}  // namespace

RN_API int rn_sgd_master_step_ex(float *const *masters, float *const *momenta, const void *const *grads, void *const *params16,
                                 const int64_t *numels, int n_tensors, int grads16, int dtype16, float lr, float momentum, float dampening,
                                 float weight_decay, int nesterov, int first_step, const float *grad_scale, const float *found_inf,
                                 void *stream)
{
    if (dtype16 != RN_BF16 && dtype16 != RN_F16) return RN_EUNSUPPORTED;
    if (!masters || !momenta || !grads || !params16 || !numels || n_tensors < 0) return RN_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    for (int base = 0; base < n_tensors; base += SGD_MAX_TENSORS) {
        SgdTable t;
        const int cnt = (n_tensors - base) < SGD_MAX_TENSORS ? (n_tensors - base) : SGD_MAX_TENSORS;
        for (int i = 0; i < cnt; ++i) {
            if (!masters[base + i] || !grads[base + i] || numels[base + i] < 0 || (momentum != 0.0f && !momenta[base + i])) return RN_EINVAL;
            if (!rn::aligned(masters[base + i], 16) || !rn::aligned(grads[base + i], 16) || (momenta[base + i] && !rn::aligned(momenta[base + i], 16)) ||
                (params16[base + i] && !rn::aligned(params16[base + i], 8)))
                return RN_EALIGN;
            t.master[i] = masters[base + i]; t.mom[i] = momenta[base + i]; t.grad[i] = grads[base + i];
            t.p16[i] = params16[base + i]; t.n[i] = numels[base + i];
        }
        t.lr = lr; t.momentum = momentum; t.dampening = dampening; t.weight_decay = weight_decay;
        t.nesterov = nesterov; t.first = first_step; t.grad16 = grads16;
        t.f16 = dtype16 == RN_F16; t.grad_scale = grad_scale; t.found_inf = found_inf;
        int64_t max_n = 1;
        for (int i = 0; i < cnt; ++i) max_n = t.n[i] > max_n ? t.n[i] : max_n;
        int64_t bx = (max_n / 4 + 255) / 256;                    // one pass over the largest tensor, capped
        if (bx > SGD_BLOCKS_X) bx = SGD_BLOCKS_X;
        if (bx < 1) bx = 1;
        if (t.f16) hipLaunchKernelGGL(sgd_master_kernel<true>, dim3((unsigned)bx, (unsigned)cnt), dim3(256), 0, st, t);
        else hipLaunchKernelGGL(sgd_master_kernel<false>, dim3((unsigned)bx, (unsigned)cnt), dim3(256), 0, st, t);
        RN_LAUNCH_CHECK();
    }
    return RN_OK;
}

RN_API int rn_sgd_master_step(float *const *masters, float *const *momenta, const void *const *grads, void *const *params16,
                              const int64_t *numels, int n_tensors, int grads16, float lr, float momentum, float dampening,
                              float weight_decay, int nesterov, int first_step, void *stream)
{
    return rn_sgd_master_step_ex(masters, momenta, grads, params16, numels, n_tensors, grads16, RN_BF16, lr, momentum, dampening, weight_decay,
                                 nesterov, first_step, nullptr, nullptr, stream);
}

// ---- many small device-to-device copies in one launch ---------------------------------------------------------------------------
// The captured train step copies its inputs (8 images + 16 target tensors) into the graph's static buffers before every replay:
// 24 hipMemcpyAsync calls cost 0.19 ms of GPU time per step (11 us each, mostly fixed cost); one launch moves the same 102 MB.
namespace {
constexpr int COPY_MAX = 64;
struct CopyTable { const unsigned char *src[COPY_MAX]; unsigned char *dst[COPY_MAX]; int64_t nbytes[COPY_MAX]; };

__global__ __launch_bounds__(256) void copy_many_kernel(const CopyTable t)
{
    const int ti = blockIdx.y;
    const unsigned char *__restrict__ s = t.src[ti];
    unsigned char *__restrict__ d = t.dst[ti];
    const int64_t n = t.nbytes[ti];
    const bool vec = ((((uintptr_t)s) | ((uintptr_t)d)) & 15) == 0;
    const int64_t n16 = vec ? n >> 4 : 0;
    for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < n16; v += (int64_t)gridDim.x * 256)
        ((rn::u32x4 *)d)[v] = ((const rn::u32x4 *)s)[v];
    for (int64_t b = n16 * 16 + (int64_t)blockIdx.x * 256 + threadIdx.x; b < n; b += (int64_t)gridDim.x * 256) d[b] = s[b];
}
}  // namespace

RN_API int rn_copy_many(const void *const *srcs, void *const *dsts, const int64_t *nbytes, int n, void *stream)
{
    if (!srcs || !dsts || !nbytes || n < 0) return RN_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    for (int base = 0; base < n; base += COPY_MAX) {
        CopyTable t;
        const int cnt = (n - base) < COPY_MAX ? (n - base) : COPY_MAX;
        int64_t most = 0;
        for (int i = 0; i < COPY_MAX; ++i) {
            const int q = i < cnt ? base + i : base;
            if (!srcs[q] || !dsts[q] || nbytes[q] < 0) return RN_EINVAL;
            t.src[i] = (const unsigned char *)srcs[q]; t.dst[i] = (unsigned char *)dsts[q]; t.nbytes[i] = i < cnt ? nbytes[q] : 0;
            if (t.nbytes[i] > most) most = t.nbytes[i];
        }
        int64_t bx = (most / 16 + 255) / 256;
        bx = bx > 512 ? 512 : (bx < 1 ? 1 : bx);
        hipLaunchKernelGGL(copy_many_kernel, dim3((unsigned)bx, (unsigned)cnt), dim3(256), 0, st, t);
        RN_LAUNCH_CHECK();
    }
    return RN_OK;
}

// ---- n widening copies (dsts[i] f32 <- srcs[i] bf16 / f16, counts[i] elements) in one launch per 64 ------------------------------
// The gradient exchange keeps fp32 buckets for the bf16 working copies of the conv weights (parallel.BucketedGradAllReduce):
// torch._foreach_copy_ across dtypes is one kernel PER TENSOR (161 parameters: ~0.8 ms of 5-us launches per step).
namespace {
struct CastTable { const uint16_t *src[COPY_MAX]; float *dst[COPY_MAX]; int64_t n[COPY_MAX]; };

template <int DT> __global__ __launch_bounds__(256) void cast_many_kernel(const CastTable t)
{
    const int ti = blockIdx.y;
    const uint16_t *__restrict__ s = t.src[ti];
    float *__restrict__ d = t.dst[ti];
    const int64_t n = t.n[ti];
    const bool vec = ((((uintptr_t)s) | ((uintptr_t)d)) & 15) == 0;
    const int64_t n8 = vec ? n >> 3 : 0;
    for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < n8; v += (int64_t)gridDim.x * 256) {
        const rn::u32x4 q = ((const rn::u32x4 *)s)[v];
        const uint32_t w[4] = {q.x, q.y, q.z, q.w};
        float f[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (DT == RN_BF16) { f[2 * j] = __uint_as_float(w[j] << 16); f[2 * j + 1] = __uint_as_float(w[j] & 0xffff0000u); }
            else { f[2 * j] = (float)__builtin_bit_cast(_Float16, (uint16_t)(w[j] & 0xffffu)); f[2 * j + 1] = (float)__builtin_bit_cast(_Float16, (uint16_t)(w[j] >> 16)); }
        }
        ((rn::f32x4 *)d)[2 * v] = rn::f32x4{f[0], f[1], f[2], f[3]};
        ((rn::f32x4 *)d)[2 * v + 1] = rn::f32x4{f[4], f[5], f[6], f[7]};
    }
    for (int64_t e = n8 * 8 + (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256)
        d[e] = DT == RN_BF16 ? __uint_as_float((uint32_t)s[e] << 16) : (float)__builtin_bit_cast(_Float16, s[e]);
}
}  // namespace

RN_API int rn_cast_many_to_f32(const void *const *srcs, void *const *dsts, const int64_t *counts, int n, int src_dtype, void *stream)
{
    if (!srcs || !dsts || !counts || n < 0) return RN_EINVAL;
    if (src_dtype != RN_BF16 && src_dtype != RN_F16) return RN_EUNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    for (int base = 0; base < n; base += COPY_MAX) {
        CastTable t;
        const int cnt = (n - base) < COPY_MAX ? (n - base) : COPY_MAX;
        int64_t most = 0;
        for (int i = 0; i < COPY_MAX; ++i) {
            const int q = i < cnt ? base + i : base;
            if (!srcs[q] || !dsts[q] || counts[q] < 0) return RN_EINVAL;
            if (!rn::aligned(srcs[q], 2) || !rn::aligned(dsts[q], 4)) return RN_EALIGN;
            t.src[i] = (const uint16_t *)srcs[q]; t.dst[i] = (float *)dsts[q]; t.n[i] = i < cnt ? counts[q] : 0;
            if (t.n[i] > most) most = t.n[i];
        }
        int64_t bx = (most / 8 + 255) / 256;
        bx = bx > 512 ? 512 : (bx < 1 ? 1 : bx);
        if (src_dtype == RN_BF16) hipLaunchKernelGGL((cast_many_kernel<RN_BF16>), dim3((unsigned)bx, (unsigned)cnt), dim3(256), 0, st, t);
        else hipLaunchKernelGGL((cast_many_kernel<RN_F16>), dim3((unsigned)bx, (unsigned)cnt), dim3(256), 0, st, t);
        RN_LAUNCH_CHECK();
    }
    return RN_OK;
}

// ---- transposes of up to 16 small 16-bit matrices in one launch ----------------------------------------------------------------
// dsts[i] [cols_i][rows_i] = srcs[i] [rows_i][cols_i]^T: the data-gradient weights of a bottleneck's 1x1 convolutions (three
// `w.t().contiguous()` launches of ~5 us each per block and step before).
namespace {
constexpr int TR_MAX = 16;
struct TransposeTable { const uint16_t *src[TR_MAX]; uint16_t *dst[TR_MAX]; int rows[TR_MAX], cols[TR_MAX]; };

__global__ __launch_bounds__(256) void transpose_many_kernel(const TransposeTable t)
{
    __shared__ uint16_t tile[32][33];
    const int ti = blockIdx.y, R = t.rows[ti], Cc = t.cols[ti];
    const int tiles_c = (Cc + 31) / 32, tiles = ((R + 31) / 32) * tiles_c;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;                  // 32 x 8
    for (int tl = blockIdx.x; tl < tiles; tl += gridDim.x) {
        const int r0 = (tl / tiles_c) * 32, c0 = (tl % tiles_c) * 32;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = r0 + ty + 8 * j, c = c0 + tx;
            if (r < R && c < Cc) tile[ty + 8 * j][tx] = t.src[ti][(int64_t)r * Cc + c];
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = c0 + ty + 8 * j, r = r0 + tx;
            if (r < R && c < Cc) t.dst[ti][(int64_t)c * R + r] = tile[tx][ty + 8 * j];
        }
        __syncthreads();
    }
}

// The data-gradient weight of a level-mode 3x3 convolution (biasact._dgrad_weight): out [Cin][9][Kpad], slot k of tap t:
//   k < e = Cout - Cout % 8: w[k][8 - t][ci];  Cout % 8 != 0: slots e + s .. e + 7 (s = 8 - Cout % 8) carry channels e .. Cout - 1,
//   slots e .. e + s - 1 and everything from e + 8 (or e) up to Kpad are zero.   w [Cout][9][Cin], 16-bit elements.
__global__ __launch_bounds__(256) void levels_dgrad_weight_kernel(const uint16_t *__restrict__ w, uint16_t *__restrict__ out, const int Cout,
                                                                  const int Cin, const int Kpad)
{
    const int64_t total = (int64_t)Cin * 9 * Kpad;
    const int e = Cout - Cout % 8, sft = Cout % 8 ? 8 - Cout % 8 : 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int k = (int)(i % Kpad), t = (int)((i / Kpad) % 9), ci = (int)(i / ((int64_t)Kpad * 9));
        int ch = -1;
        if (k < e) ch = k;
        else if (sft && k >= e + sft && k < e + 8) ch = k - sft;
        out[i] = ch >= 0 ? w[((int64_t)ch * 9 + (8 - t)) * Cin + ci] : (uint16_t)0;
    }
}
}  // namespace

RN_API int rn_transpose_many(const void *const *srcs, void *const *dsts, const int *rows, const int *cols, int n, void *stream)
{
    if (!srcs || !dsts || !rows || !cols || n <= 0 || n > TR_MAX) return RN_EINVAL;
    TransposeTable t;
    int most = 1;
    for (int i = 0; i < TR_MAX; ++i) {
        const int q = i < n ? i : 0;
        if (!srcs[q] || !dsts[q] || rows[q] <= 0 || cols[q] <= 0 || srcs[q] == dsts[q]) return RN_EINVAL;
        t.src[i] = (const uint16_t *)srcs[q]; t.dst[i] = (uint16_t *)dsts[q]; t.rows[i] = rows[q]; t.cols[i] = cols[q];
        const int tl = ((rows[q] + 31) / 32) * ((cols[q] + 31) / 32);
        if (tl > most) most = tl;
    }
    hipLaunchKernelGGL(transpose_many_kernel, dim3((unsigned)(most > 1024 ? 1024 : most), (unsigned)n), dim3(256), 0, (hipStream_t)stream, t);
    RN_LAUNCH_CHECK();
    return RN_OK;
}

RN_API int rn_conv3x3_levels_dgrad_weight(const void *w, void *out, int Cout, int Cin, int Kpad, void *stream)
{
    if (!w || !out || Cout < 8 || Cin <= 0 || Kpad < Cout || w == out) return RN_EINVAL;
    const int64_t total = (int64_t)Cin * 9 * Kpad;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(levels_dgrad_weight_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const uint16_t *)w, (uint16_t *)out,
                       Cout, Cin, Kpad);
    RN_LAUNCH_CHECK();
    return RN_OK;
}
