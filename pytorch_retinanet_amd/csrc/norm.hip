// Fused BatchNorm2d (+ residual add) (+ ReLU) for channels-last activations -- conv-stack widening
// ("fused bias+ReLU / BN").  rocprof on the R50-FPN train step shows MIOpen's
// batch norm (7 kernels per layer) plus the stand-alone ReLU and residual-add kernels around it at
// ~30% of the GPU time and ~20% of the host launch time; they are all HBM-bound passes over the same
// [N*H*W, C] activation.  This file does the same arithmetic as
//   y = relu(batch_norm(x, running stats / batch stats, gamma, beta) + residual)
// (reference call sites: retinanet/backbone.py:70-80, :118-136, :248-250) in
//   forward  : 1 read of x for the statistics + 1 read of x (+ residual) and 1 write of y
//   backward : 1 read of (dy, y, x) for the two channel sums + 1 read of (dy, y, x), 1 write of dx
//              (+ 1 write of the residual branch's gradient)
// i.e. 3 + 5 (+1) tensor passes instead of 5 + 8 (+3), in 3 + 3 launches instead of ~13.
//
// Layout: x is [N, C, H, W] in channels_last memory format = dense [M = N*H*W][C]; every thread
// owns 8 consecutive channels (one 16-byte bf16 vector, two for fp32) so loads/stores are coalesced
// and the per-channel coefficients live in registers.  Statistics: per-thread fp32 sums over a
// block's rows, LDS tree over the block's row lanes, one partial per block, combined in double by a
// per-channel kernel (deterministic; no float atomics).  Requires C % 8 == 0.
#include "rn_common.hpp"
#include <cstdlib>

namespace {

constexpr int BN_BLOCK = 256;
constexpr int BN_RBLOCK_MAX = 1024;       // threads per block of the two BN reduction kernels (runtime: reduce_grid)
constexpr int BN_MAX_BLOCKS = 512;
constexpr int FIN_CH = 8;                 // channels per block of the per-channel kernels
constexpr int FIN_THREADS = 1024;         // threads per block of the per-channel kernels: with 512 partials every thread has ONE batch of loads in flight
constexpr int FIN_LANES = FIN_THREADS / FIN_CH;    // threads that split the partial sums of one channel (thread = channel + FIN_CH * lane)
constexpr int FIN_WAVES = FIN_THREADS / 64;

template <int DT> struct vec8;          // 8 consecutive channels <-> float[8]
template <> struct vec8<RN_BF16> {
    static __device__ __forceinline__ void ld(const void *p, int64_t v, float (&f)[8]) { rn::dt<RN_BF16>::unpack(((const rn::u32x4 *)p)[v], f); }
    static __device__ __forceinline__ void st(void *p, int64_t v, const float (&f)[8]) { ((rn::u32x4 *)p)[v] = rn::dt<RN_BF16>::pack(f); }
};
template <> struct vec8<RN_F16> {
    static __device__ __forceinline__ void ld(const void *p, int64_t v, float (&f)[8]) { rn::dt<RN_F16>::unpack(((const rn::u32x4 *)p)[v], f); }
    static __device__ __forceinline__ void st(void *p, int64_t v, const float (&f)[8]) { ((rn::u32x4 *)p)[v] = rn::dt<RN_F16>::pack(f); }
};
template <> struct vec8<RN_F32> {
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

// Work split shared by the two reduction kernels: thread = (channel group cg, row lane rl); blocks own
// contiguous row ranges.  C8 = C/8 channel groups; when C8 > 256 a thread loops over channel groups.
struct Split { int C8, lanes, groups_per_thread; };
__device__ __forceinline__ Split split_of(const int C)
{
    Split s;
    s.C8 = C / 8;
    if (s.C8 >= BN_BLOCK) { s.lanes = 1; s.groups_per_thread = (s.C8 + BN_BLOCK - 1) / BN_BLOCK; }
    else { s.lanes = BN_BLOCK / s.C8; s.groups_per_thread = 1; }
    return s;
}

// The BN reduction kernels also split the CHANNELS over blockIdx.y (gridDim.y slabs of C8 / gridDim.y channel groups): a
// layer with few rows and many channels (layer3 / layer4: 8 400 - 33 600 rows of 2 - 4 KiB) gets enough blocks to fill the
// chip without more per-block partials than the per-channel kernel wants to read.  Thread = (local group lg, row lane rl).
struct Slab { int C8, Cs8, cg0, lanes, groups_per_thread; };
__device__ __forceinline__ Slab slab_of(const int C)
{
    Slab s;
    s.C8 = C / 8;
    s.Cs8 = s.C8 / (int)gridDim.y;
    s.cg0 = (int)blockIdx.y * s.Cs8;
    const int T = (int)blockDim.x;                                   // 256 .. 1024 threads (reduce_grid)
    if (s.Cs8 >= T) { s.lanes = 1; s.groups_per_thread = (s.Cs8 + T - 1) / T; }
    else { s.lanes = T / s.Cs8; s.groups_per_thread = 1; }
    return s;
}
// block totals of the slab's channels -> partial[blockIdx.x][2][C]   (s, q: this thread's 8-channel sums)
__device__ __forceinline__ void slab_reduce(const Slab &sp, const int C, const int lg, const int rl, const bool valid, const float (&s)[8],
                                            const float (&q)[8], float *smem, float *__restrict__ partial)
{
    const int Cs = sp.Cs8 * 8;
    if (sp.lanes > 1) {
        if (valid) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { smem[(rl * 2 + 0) * Cs + lg * 8 + j] = s[j]; smem[(rl * 2 + 1) * Cs + lg * 8 + j] = q[j]; }
        }
        __syncthreads();
        for (int c = threadIdx.x; c < 2 * Cs; c += (int)blockDim.x) {   // c indexes [2][Cs]
            float t = 0.0f;
            for (int l = 0; l < sp.lanes; ++l) t += smem[l * 2 * Cs + c];
            const int which = c >= Cs ? 1 : 0;
            partial[((int64_t)blockIdx.x * 2 + which) * C + sp.cg0 * 8 + (c - which * Cs)] = t;
        }
    } else if (valid) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            partial[(int64_t)blockIdx.x * 2 * C + (sp.cg0 + lg) * 8 + j] = s[j];
            partial[(int64_t)blockIdx.x * 2 * C + C + (sp.cg0 + lg) * 8 + j] = q[j];
        }
    }
}


// Row traversal of the two reduction kernels (the per-thread sequence of rows; every row is visited once whatever the
// order, and the per-block partials are combined in a fixed order, so each ORDER is deterministic).
//   0: blocks own contiguous row ranges, walked front to back
//   1: sweep -- all blocks advance through the tensor together, front to back (the order the apply kernels use)
//   2: sweep, back to front: the rows the producer wrote last are read first, and the apply kernel that follows finds the
//      front of the tensor -- read last here -- in the Infinity Cache
struct RowWalk {
    int64_t base, step, n;          // row(k) = base + k * step, k = 0 .. n-1
    __device__ __forceinline__ int64_t row(const int64_t k) const { return base + k * step; }
};
__device__ __forceinline__ RowWalk row_walk(const int order, const int64_t M, const int lanes, const int rl)
{
    RowWalk w;
    const int64_t G = gridDim.x, b = blockIdx.x;
    if (order == 0) {
        const int64_t rows_per_block = (M + G - 1) / G;
        const int64_t r0 = b * rows_per_block, r1 = min(r0 + rows_per_block, M);
        w.base = r0 + rl; w.step = lanes;
        w.n = r1 > w.base ? (r1 - w.base + lanes - 1) / lanes : 0;
    } else {
        const int64_t first = b * lanes + rl, stride = G * lanes;
        w.n = M > first ? (M - first + stride - 1) / stride : 0;
        if (order == 1) { w.base = first; w.step = stride; }
        else { w.base = first + (w.n - 1) * stride; w.step = -stride; }
    }
    return w;
}

// Default 2 (in-step A/B on two boxes, 4 pairs: +0.6 ... +1.3 % images/s over order 0, order 1 in between; choosing by tensor
// size was no better).
inline int bn_order()
{
    return 2;
}

// ---------------------------------------------------------------- forward statistics
// partial[block][0][c] = sum x, partial[block][1][c] = sum x^2 over the block's rows
template <int DT>
__global__ __launch_bounds__(BN_RBLOCK_MAX) void bn_stats_partial_kernel(const void *__restrict__ x, const int64_t M, const int C,
                                                                    float *__restrict__ partial, const int order)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];      // [lanes][2][C]  (only when lanes > 1)
    const Slab sp = slab_of(C);
    for (int gi = 0; gi < sp.groups_per_thread; ++gi) {
        const int lg = (sp.lanes > 1) ? (int)(threadIdx.x % sp.Cs8) : (int)threadIdx.x + gi * (int)blockDim.x;
        const int rl = (sp.lanes > 1) ? (int)(threadIdx.x / sp.Cs8) : 0;
        const int cg = sp.cg0 + lg;
        const bool valid = lg < sp.Cs8 && rl < sp.lanes;
        float s[8], q[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { s[j] = 0.0f; q[j] = 0.0f; }
        if (valid) {
            const RowWalk w = row_walk(order, M, sp.lanes, rl);
            int64_t k = 0;
            for (; k + 3 < w.n; k += 4) {                                // 4 independent 16-byte loads in flight per thread
                float f[4][8];
#pragma unroll
                for (int u = 0; u < 4; ++u) vec8<DT>::ld(x, w.row(k + u) * sp.C8 + cg, f[u]);
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int j = 0; j < 8; ++j) { s[j] += f[u][j]; q[j] = fmaf(f[u][j], f[u][j], q[j]); }
            }
            for (; k < w.n; ++k) {
                float f[8];
                vec8<DT>::ld(x, w.row(k) * sp.C8 + cg, f);
#pragma unroll
                for (int j = 0; j < 8; ++j) { s[j] += f[j]; q[j] = fmaf(f[j], f[j], q[j]); }
            }
        }
        slab_reduce(sp, C, lg, rl, valid, s, q, smem, partial);
    }
}

// Sum of the per-block partials of one channel pair, split over FIN_LANES threads and combined in double: a wave holds 8 lanes of
// each of its 8 channels (butterfly over lane bits 3..5), the 16 waves meet in LDS.  Fixed order: deterministic.  Returns the
// totals to the lane-0 thread of each channel.  These kernels are latency chains (a launch, one round of loads that miss --
// the partials were written on other XCDs -- and the combine): 1024 threads so that the round is ONE batch of loads per thread
// (256 threads: four dependent batches, 5.9 us per call; 106 calls per R50 step).
__device__ __forceinline__ void channel_totals(const float *__restrict__ partial, const int nblocks, const int C, const int c,
                                               const int ln, double (*sh)[FIN_WAVES][FIN_CH], double &s, double &q)
{
    double ls = 0.0, lq = 0.0;
    if (c < C) {
        int b = ln;
        for (; b + 3 * FIN_LANES < nblocks; b += 4 * FIN_LANES) {        // 8 independent loads in flight
            float ps[4], pq[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { ps[u] = partial[(int64_t)(b + u * FIN_LANES) * 2 * C + c]; pq[u] = partial[(int64_t)(b + u * FIN_LANES) * 2 * C + C + c]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) { ls += (double)ps[u]; lq += (double)pq[u]; }
        }
        for (; b < nblocks; b += FIN_LANES) { ls += (double)partial[(int64_t)b * 2 * C + c]; lq += (double)partial[(int64_t)b * 2 * C + C + c]; }
    }
#pragma unroll
    for (int m = FIN_CH; m < 64; m <<= 1) { ls += __shfl_xor(ls, m); lq += __shfl_xor(lq, m); }
    const int ch = threadIdx.x % FIN_CH, wv = threadIdx.x / 64;
    if ((threadIdx.x & 63) < FIN_CH) { sh[0][wv][ch] = ls; sh[1][wv][ch] = lq; }
    __syncthreads();
    s = 0.0; q = 0.0;
    if (ln == 0)
        for (int l = 0; l < FIN_WAVES; ++l) { s += sh[0][l][ch]; q += sh[1][l][ch]; }
}

// per channel: mean / inverse std of the batch, the affine coefficients y = x*a + b, running-stat update
__global__ __launch_bounds__(FIN_THREADS) void bn_stats_final_kernel(const float *__restrict__ partial, const int nblocks, const int64_t M,
                                                             const int C, const float *__restrict__ gamma, const float *__restrict__ beta,
                                                             float *__restrict__ running_mean, float *__restrict__ running_var,
                                                             int64_t *__restrict__ num_batches_tracked, const float momentum,
                                                             const float eps, float *__restrict__ save_mean, float *__restrict__ save_invstd,
                                                             float *__restrict__ coef_a, float *__restrict__ coef_b)
{
    __shared__ double sh[2][FIN_WAVES][FIN_CH];
    const int c = blockIdx.x * FIN_CH + threadIdx.x % FIN_CH, ln = threadIdx.x / FIN_CH;
    if (blockIdx.x == 0 && threadIdx.x == 0 && num_batches_tracked) *num_batches_tracked += 1;
    double s, q;
    channel_totals(partial, nblocks, C, c, ln, sh, s, q);
    if (ln != 0 || c >= C) return;
    const double mean = s / (double)M;
    double var = q / (double)M - mean * mean;
    if (var < 0.0) var = 0.0;
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    save_mean[c] = (float)mean;
    save_invstd[c] = invstd;
    const float a = (gamma ? gamma[c] : 1.0f) * invstd;
    coef_a[c] = a;
    coef_b[c] = (beta ? beta[c] : 0.0f) - (float)mean * a;
    if (running_mean) {
        const double unbiased = (M > 1) ? var * (double)M / (double)(M - 1) : var;
        running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * (float)mean;
        running_var[c] = (1.0f - momentum) * running_var[c] + momentum * (float)unbiased;
    }
}

// eval mode: coefficients from the running statistics
__global__ __launch_bounds__(256) void bn_eval_coef_kernel(const int C, const float *__restrict__ gamma, const float *__restrict__ beta,
                                                           const float *__restrict__ running_mean, const float *__restrict__ running_var,
                                                           const float eps, float *__restrict__ save_mean, float *__restrict__ save_invstd,
                                                           float *__restrict__ coef_a, float *__restrict__ coef_b)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float invstd = 1.0f / sqrtf(running_var[c] + eps);
    const float a = (gamma ? gamma[c] : 1.0f) * invstd;
    save_mean[c] = running_mean[c];
    save_invstd[c] = invstd;
    coef_a[c] = a;
    coef_b[c] = (beta ? beta[c] : 0.0f) - running_mean[c] * a;
}

// ReLU mask of the forward output without reading it: y = round_DT(max(fma(x, a, b), 0)) > 0  <=>  fma(x, a, b)
// is above the largest value that rounds to zero in DT (same fma, same coefficients as bn_apply_kernel).
// Only for layers without a residual input (the residual is not available in backward).
template <int DT> __device__ __forceinline__ float relu_alive_threshold();
template <> __device__ __forceinline__ float relu_alive_threshold<RN_F32>() { return 0.0f; }
template <> __device__ __forceinline__ float relu_alive_threshold<RN_BF16>() { return __uint_as_float(0x00004000u); }   // 2^-134: half the smallest bf16 subnormal (ties to even -> 0)
template <> __device__ __forceinline__ float relu_alive_threshold<RN_F16>() { return __uint_as_float(0x33000000u); }    // 2^-25: half the smallest f16 subnormal

// value of f after a store in DT and a load back
template <int DT> __device__ __forceinline__ float round_dt(const float f);
template <> __device__ __forceinline__ float round_dt<RN_F32>(const float f) { return f; }
template <> __device__ __forceinline__ float round_dt<RN_BF16>(const float f) { return __uint_as_float(rn::dt<RN_BF16>::pk(f, 0.0f) << 16); }
template <> __device__ __forceinline__ float round_dt<RN_F16>(const float f) { return (float)(_Float16)f; }

// ---------------------------------------------------------------- forward apply: y = act(x*a + b (+ res))
// RA (with RES): the residual is itself a BatchNorm output that was never written -- res holds that layer's INPUT and
// (res_a, res_b) its coefficients; the value added is round_DT(fma(res, res_a, res_b)), exactly what a separate apply pass
// would have stored (the downsample branch of a bottleneck: one pass over the block's largest tensor less, read + write).
template <int DT, bool RELU, bool RES, bool RA = false>
__global__ __launch_bounds__(BN_BLOCK) void bn_apply_kernel(const void *__restrict__ x, const void *__restrict__ res, void *__restrict__ y,
                                                            const int64_t nvec, const int C8, const float *__restrict__ coef_a,
                                                            const float *__restrict__ coef_b, uint8_t *__restrict__ relu_mask,
                                                            const float *__restrict__ res_a = nullptr, const float *__restrict__ res_b = nullptr)
{
    const float alive = relu_alive_threshold<DT>();
    // C8 | 256 (every ResNet width): a thread always lands on the same channel group -> coefficients in registers
    const bool fixed = (BN_BLOCK % C8) == 0;
    float a[8], b[8], ra[8], rb[8];
    if (fixed) {
        vec8<RN_F32>::ld(coef_a, threadIdx.x % C8, a); vec8<RN_F32>::ld(coef_b, threadIdx.x % C8, b);
        if (RA) { vec8<RN_F32>::ld(res_a, threadIdx.x % C8, ra); vec8<RN_F32>::ld(res_b, threadIdx.x % C8, rb); }
    }
    for (int64_t v = (int64_t)blockIdx.x * BN_BLOCK + threadIdx.x; v < nvec; v += (int64_t)gridDim.x * BN_BLOCK) {
        float f[8], r[8];
        vec8<DT>::ld(x, v, f);
        if (RES) vec8<DT>::ld(res, v, r);
        if (!fixed) {
            const int cg = (int)(v % C8);
            vec8<RN_F32>::ld(coef_a, cg, a); vec8<RN_F32>::ld(coef_b, cg, b);
            if (RA) { vec8<RN_F32>::ld(res_a, cg, ra); vec8<RN_F32>::ld(res_b, cg, rb); }
        }
        if (RA) {
#pragma unroll
            for (int j = 0; j < 8; ++j) r[j] = round_dt<DT>(fmaf(r[j], ra[j], rb[j]));
        }
        unsigned bits = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float t = fmaf(f[j], a[j], b[j]);
            if (RES) t += r[j];
            if (RELU) { bits |= (t > alive) ? (1u << j) : 0u; t = t > 0.0f ? t : 0.0f; }
            f[j] = t;
        }
        vec8<DT>::st(y, v, f);
        if (RELU && relu_mask) relu_mask[v] = (uint8_t)bits;      // one byte per 8 channels: what backward reads instead of y
    }
}

// ---------------------------------------------------------------- backward sums
// g = dy * (y > 0) (RELU) ; partial[block][0][c] = sum g, partial[block][1][c] = sum g * xhat
// RELU: 0 = none, 1 = mask from y, 2 = mask recomputed from x (y is not read)
template <int DT, int RELU>
__global__ __launch_bounds__(BN_RBLOCK_MAX) void bn_bwd_partial_kernel(const void *__restrict__ dy, const void *__restrict__ y,
                                                                  const void *__restrict__ x, const int64_t M, const int C,
                                                                  const float *__restrict__ save_mean, const float *__restrict__ save_invstd,
                                                                  const float *__restrict__ fwd_a, const float *__restrict__ fwd_b,
                                                                  float *__restrict__ partial, const int order)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const Slab sp = slab_of(C);
    for (int gi = 0; gi < sp.groups_per_thread; ++gi) {
        const int lg = (sp.lanes > 1) ? (int)(threadIdx.x % sp.Cs8) : (int)threadIdx.x + gi * (int)blockDim.x;
        const int rl = (sp.lanes > 1) ? (int)(threadIdx.x / sp.Cs8) : 0;
        const int cg = sp.cg0 + lg;
        const bool valid = lg < sp.Cs8 && rl < sp.lanes;
        float s[8], q[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { s[j] = 0.0f; q[j] = 0.0f; }
        if (valid) {
            float mu[8], is[8], fa[8], fb[8];
            vec8<RN_F32>::ld(save_mean, cg, mu);
            vec8<RN_F32>::ld(save_invstd, cg, is);
            if (RELU == 2) { vec8<RN_F32>::ld(fwd_a, cg, fa); vec8<RN_F32>::ld(fwd_b, cg, fb); }
            const float alive = relu_alive_threshold<DT>();
            const RowWalk w = row_walk(order, M, sp.lanes, rl);
            int64_t k = 0;
            for (; k + 1 < w.n; k += 2) {                                // 2 rows x 3 tensors = 6 loads in flight per thread
                float g[2][8], yy[2][8], xx[2][8];
                unsigned mb[2] = {0xffu, 0xffu};
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int64_t v = w.row(k + u) * sp.C8 + cg;
                    vec8<DT>::ld(dy, v, g[u]);
                    vec8<DT>::ld(x, v, xx[u]);
                    if (RELU == 1) vec8<DT>::ld(y, v, yy[u]);
                    if (RELU == 3) mb[u] = ((const uint8_t *)y)[v];
                }
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const bool dead = RELU == 1 ? !(yy[u][j] > 0.0f) : (RELU == 2 ? !(fmaf(xx[u][j], fa[j], fb[j]) > alive)
                                                                  : (RELU == 3 ? !((mb[u] >> j) & 1u) : false));
                        const float gj = dead ? 0.0f : g[u][j];
                        s[j] += gj;
                        q[j] = fmaf(gj, (xx[u][j] - mu[j]) * is[j], q[j]);
                    }
            }
            for (; k < w.n; ++k) {
                const int64_t v = w.row(k) * sp.C8 + cg;
                float g[8], yy[8], xx[8];
                vec8<DT>::ld(dy, v, g);
                vec8<DT>::ld(x, v, xx);
                if (RELU == 1) vec8<DT>::ld(y, v, yy);
                const unsigned mb1 = RELU == 3 ? ((const uint8_t *)y)[v] : 0xffu;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const bool dead = RELU == 1 ? !(yy[j] > 0.0f) : (RELU == 2 ? !(fmaf(xx[j], fa[j], fb[j]) > alive)
                                                              : (RELU == 3 ? !((mb1 >> j) & 1u) : false));
                    const float gj = dead ? 0.0f : g[j];
                    s[j] += gj;
                    q[j] = fmaf(gj, (xx[j] - mu[j]) * is[j], q[j]);
                }
            }
        }
        slab_reduce(sp, C, lg, rl, valid, s, q, smem, partial);
    }
}

// per channel: dgamma, dbeta and the coefficients of dx = a*g + k0 + k1*x
//   training: dx = a*(g - mean(g) - xhat*mean(g*xhat));  eval (frozen statistics): dx = a*g
__global__ __launch_bounds__(FIN_THREADS) void bn_bwd_final_kernel(const float *__restrict__ partial, const int nblocks, const int64_t M, const int C,
                                                           const float *__restrict__ gamma, const float *__restrict__ save_mean,
                                                           const float *__restrict__ save_invstd, const int training,
                                                           float *__restrict__ dgamma, float *__restrict__ dbeta,
                                                           float *__restrict__ coef_a, float *__restrict__ coef_k0, float *__restrict__ coef_k1)
{
    __shared__ double sh[2][FIN_WAVES][FIN_CH];
    const int c = blockIdx.x * FIN_CH + threadIdx.x % FIN_CH, ln = threadIdx.x / FIN_CH;
    double s, q;
    channel_totals(partial, nblocks, C, c, ln, sh, s, q);
    if (ln != 0 || c >= C) return;
    if (dbeta) dbeta[c] = (float)s;
    if (dgamma) dgamma[c] = (float)q;
    const float a = (gamma ? gamma[c] : 1.0f) * save_invstd[c];
    coef_a[c] = a;
    if (training) {
        const double c1 = -(double)a * s / (double)M, c2 = -(double)a * q / (double)M;
        coef_k1[c] = (float)(c2 * (double)save_invstd[c]);
        coef_k0[c] = (float)(c1 - c2 * (double)save_mean[c] * (double)save_invstd[c]);
    } else {
        coef_k0[c] = 0.0f; coef_k1[c] = 0.0f;
    }
}

// dx = a*g + k0 + k1*x ; dres = g      (RELU as in bn_bwd_partial_kernel)
template <int DT, int RELU, bool RES>
__global__ __launch_bounds__(BN_BLOCK) void bn_bwd_apply_kernel(const void *__restrict__ dy, const void *__restrict__ y, const void *__restrict__ x,
                                                                void *__restrict__ dx, void *__restrict__ dres, const int64_t nvec, const int C8,
                                                                const float *__restrict__ coef_a, const float *__restrict__ coef_k0,
                                                                const float *__restrict__ coef_k1, const float *__restrict__ fwd_a,
                                                                const float *__restrict__ fwd_b)
{
    const bool fixed = (BN_BLOCK % C8) == 0;
    float a[8], k0[8], k1[8], fa[8], fb[8];
    const float alive = relu_alive_threshold<DT>();
    if (fixed) {
        const int cg = threadIdx.x % C8;
        vec8<RN_F32>::ld(coef_a, cg, a); vec8<RN_F32>::ld(coef_k0, cg, k0); vec8<RN_F32>::ld(coef_k1, cg, k1);
        if (RELU == 2) { vec8<RN_F32>::ld(fwd_a, cg, fa); vec8<RN_F32>::ld(fwd_b, cg, fb); }
    }
    for (int64_t v = (int64_t)blockIdx.x * BN_BLOCK + threadIdx.x; v < nvec; v += (int64_t)gridDim.x * BN_BLOCK) {
        float g[8], yy[8], xx[8];
        vec8<DT>::ld(dy, v, g);
        vec8<DT>::ld(x, v, xx);
        if (RELU == 1) vec8<DT>::ld(y, v, yy);
        const unsigned mb = RELU == 3 ? ((const uint8_t *)y)[v] : 0xffu;
        if (!fixed) {
            const int cg = (int)(v % C8);
            vec8<RN_F32>::ld(coef_a, cg, a); vec8<RN_F32>::ld(coef_k0, cg, k0); vec8<RN_F32>::ld(coef_k1, cg, k1);
            if (RELU == 2) { vec8<RN_F32>::ld(fwd_a, cg, fa); vec8<RN_F32>::ld(fwd_b, cg, fb); }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (RELU == 1 && !(yy[j] > 0.0f)) g[j] = 0.0f;
            if (RELU == 2 && !(fmaf(xx[j], fa[j], fb[j]) > alive)) g[j] = 0.0f;
            if (RELU == 3 && !((mb >> j) & 1u)) g[j] = 0.0f;
            xx[j] = fmaf(a[j], g[j], fmaf(k1[j], xx[j], k0[j]));
        }
        vec8<DT>::st(dx, v, xx);
        if (RES) vec8<DT>::st(dres, v, g);
    }
}

int reduce_blocks(const int64_t M)
{
    int64_t b = (M + 255) / 256;               // at least ~256 rows per block
    if (b > BN_MAX_BLOCKS) b = BN_MAX_BLOCKS;
    if (b < 1) b = 1;
    return (int)b;
}

// Grid of the two BN reduction kernels: x = row splits (each writes one partial per channel: at most 1 MiB of partials for
// the per-channel kernel to read), y = channel slabs (>= 32 groups = 512-byte row segments) when the rows alone give fewer
// than ~512 blocks.
struct ReduceGrid { int row_splits, slabs, threads; size_t lds; };
ReduceGrid reduce_grid(const int64_t M, const int C, const bool wide)
{
    ReduceGrid g;
    const int C8 = C / 8;
    int64_t cap = (int64_t)131072 / C;                               // row_splits * 2 * C floats <= 1 MiB
    cap = cap > BN_MAX_BLOCKS ? BN_MAX_BLOCKS : (cap < 16 ? 16 : cap);
    int64_t b = (M + 63) / 64;                                       // at least ~64 rows per block
    b = b > cap ? cap : (b < 1 ? 1 : b);
    g.row_splits = (int)b;
    g.slabs = 1;
    while (g.row_splits * g.slabs * 2 <= 512 && C8 % (g.slabs * 2) == 0 && C8 / (g.slabs * 2) >= 32) g.slabs *= 2;
    // 512-thread blocks for the backward reduction of residual layers (it reads a byte of ReLU mask per 16 bytes of gradient
    // and wants the extra waves: -28 % at every layer size in the step); 256 everywhere else (512 / 1024: no gain or a loss)
    const int Cs8 = C8 / g.slabs;
    g.threads = wide ? 2 * BN_BLOCK : BN_BLOCK;
    const int lanes = (Cs8 >= g.threads) ? 1 : g.threads / Cs8;
    g.lds = lanes > 1 ? sizeof(float) * (size_t)lanes * 2 * (size_t)Cs8 * 8 : 0;
    return g;
}

size_t reduce_lds(const int C)
{
    const int C8 = C / 8;
    const int lanes = (C8 >= BN_BLOCK) ? 1 : BN_BLOCK / C8;
    return lanes > 1 ? sizeof(float) * (size_t)lanes * 2 * C : 0;
}

int apply_blocks(const int64_t nvec)
{
    int64_t b = (nvec + BN_BLOCK - 1) / BN_BLOCK;
    if (b > 8192) b = 8192;
    return (int)(b < 1 ? 1 : b);
}

}  // namespace

// ---- launch helpers shared by the entry points (one per kernel: the fused calls and the pieces run the same code) ----
namespace {

int stats_partial_launch(const void *x, int dtype, int64_t M, int C, float *partial, hipStream_t st, int *nblocks)
{
    const ReduceGrid rg = reduce_grid(M, C, false);
    const dim3 grid(rg.row_splits, rg.slabs);
    const int order = bn_order();
    switch (dtype) {
        case RN_F32: hipLaunchKernelGGL((bn_stats_partial_kernel<RN_F32>), grid, dim3(rg.threads), rg.lds, st, x, M, C, partial, order); break;
        case RN_BF16: hipLaunchKernelGGL((bn_stats_partial_kernel<RN_BF16>), grid, dim3(rg.threads), rg.lds, st, x, M, C, partial, order); break;
        default: hipLaunchKernelGGL((bn_stats_partial_kernel<RN_F16>), grid, dim3(rg.threads), rg.lds, st, x, M, C, partial, order); break;
    }
    RN_LAUNCH_CHECK();
    *nblocks = rg.row_splits;
    return RN_OK;
}

int stats_final_launch(const float *partial, int nb, int64_t M, int C, const float *gamma, const float *beta, float *running_mean,
                       float *running_var, int64_t *num_batches_tracked, float momentum, float eps, float *save_mean,
                       float *save_invstd, float *ca, float *cb, hipStream_t st)
{
    hipLaunchKernelGGL(bn_stats_final_kernel, dim3((C + FIN_CH - 1) / FIN_CH), dim3(FIN_THREADS), 0, st, partial, nb, M, C, gamma, beta, running_mean,
                       running_var, num_batches_tracked, momentum, eps, save_mean, save_invstd, ca, cb);
    RN_LAUNCH_CHECK();
    return RN_OK;
}

int apply_launch(const void *x, const void *residual, void *y, int dtype, int64_t M, int C, const float *ca, const float *cb, int relu,
                 uint8_t *relu_mask, hipStream_t st, const float *res_a = nullptr, const float *res_b = nullptr)
{
    const int64_t nvec = M * (C / 8);
    const dim3 g(apply_blocks(nvec)), b(BN_BLOCK);
    const int C8 = C / 8;
    if (res_a) {                                                     // residual = an unwritten BatchNorm output (ReLU form only)
#define RN_BN_APPLY_RA(DT) hipLaunchKernelGGL((bn_apply_kernel<DT, true, true, true>), g, b, 0, st, x, residual, y, nvec, C8, ca, cb, relu_mask, res_a, res_b);
        switch (dtype) {
            case RN_F32: RN_BN_APPLY_RA(RN_F32) break;
            case RN_BF16: RN_BN_APPLY_RA(RN_BF16) break;
            default: RN_BN_APPLY_RA(RN_F16) break;
        }
#undef RN_BN_APPLY_RA
        RN_LAUNCH_CHECK();
        return RN_OK;
    }
#define RN_BN_APPLY(DT)                                                                                                          \
    if (relu) { if (residual) hipLaunchKernelGGL((bn_apply_kernel<DT, true, true>), g, b, 0, st, x, residual, y, nvec, C8, ca, cb, relu_mask);    \
                else hipLaunchKernelGGL((bn_apply_kernel<DT, true, false>), g, b, 0, st, x, residual, y, nvec, C8, ca, cb, relu_mask); }           \
    else      { if (residual) hipLaunchKernelGGL((bn_apply_kernel<DT, false, true>), g, b, 0, st, x, residual, y, nvec, C8, ca, cb, relu_mask);   \
                else hipLaunchKernelGGL((bn_apply_kernel<DT, false, false>), g, b, 0, st, x, residual, y, nvec, C8, ca, cb, relu_mask); }
    switch (dtype) {
        case RN_F32: RN_BN_APPLY(RN_F32) break;
        case RN_BF16: RN_BN_APPLY(RN_BF16) break;
        default: RN_BN_APPLY(RN_F16) break;
    }
#undef RN_BN_APPLY
    RN_LAUNCH_CHECK();
    return RN_OK;
}

// rmode: 0 none, 1 mask from y, 2 recomputed from x and the forward coefficients, 3 bits in y
int bwd_partial_launch(const void *dy, const void *y, const void *x, int dtype, int64_t M, int C, const float *save_mean,
                       const float *save_invstd, const float *fa, const float *fb, int rmode, float *partial, hipStream_t st, int *nblocks)
{
    const ReduceGrid rg = reduce_grid(M, C, rmode == 3);
    const dim3 grid(rg.row_splits, rg.slabs);
    const size_t lds = rg.lds;
    const int order = bn_order();
#define RN_BN_BWD_PART(DT)                                                                                                             \
    if (rmode == 1) hipLaunchKernelGGL((bn_bwd_partial_kernel<DT, 1>), grid, dim3(rg.threads), lds, st, dy, y, x, M, C, save_mean, save_invstd, fa, fb, partial, order); \
    else if (rmode == 2) hipLaunchKernelGGL((bn_bwd_partial_kernel<DT, 2>), grid, dim3(rg.threads), lds, st, dy, y, x, M, C, save_mean, save_invstd, fa, fb, partial, order); \
    else if (rmode == 3) hipLaunchKernelGGL((bn_bwd_partial_kernel<DT, 3>), grid, dim3(rg.threads), lds, st, dy, y, x, M, C, save_mean, save_invstd, fa, fb, partial, order); \
    else hipLaunchKernelGGL((bn_bwd_partial_kernel<DT, 0>), grid, dim3(rg.threads), lds, st, dy, y, x, M, C, save_mean, save_invstd, fa, fb, partial, order);
    switch (dtype) {
        case RN_F32: RN_BN_BWD_PART(RN_F32) break;
        case RN_BF16: RN_BN_BWD_PART(RN_BF16) break;
        default: RN_BN_BWD_PART(RN_F16) break;
    }
#undef RN_BN_BWD_PART
    RN_LAUNCH_CHECK();
    *nblocks = rg.row_splits;
    return RN_OK;
}

int bwd_final_launch(const float *partial, int nb, int64_t M, int C, const float *gamma, const float *save_mean, const float *save_invstd,
                     int training, float *dgamma, float *dbeta, float *ca, float *k0, float *k1, hipStream_t st)
{
    hipLaunchKernelGGL(bn_bwd_final_kernel, dim3((C + FIN_CH - 1) / FIN_CH), dim3(FIN_THREADS), 0, st, partial, nb, M, C, gamma, save_mean, save_invstd,
                       training, dgamma, dbeta, ca, k0, k1);
    RN_LAUNCH_CHECK();
    return RN_OK;
}

int bwd_apply_launch(const void *dy, const void *y, const void *x, void *dx, void *dresidual, int dtype, int64_t M, int C, const float *ca,
                     const float *k0, const float *k1, const float *fa, const float *fb, int rmode, hipStream_t st)
{
    const int64_t nvec = M * (C / 8);
    const dim3 g(apply_blocks(nvec)), b(BN_BLOCK);
    const int C8 = C / 8;
#define RN_BN_BWD_APPLY(DT)                                                                                                                  \
    if (rmode == 2) hipLaunchKernelGGL((bn_bwd_apply_kernel<DT, 2, false>), g, b, 0, st, dy, y, x, dx, dresidual, nvec, C8, ca, k0, k1, fa, fb);        \
    else if (rmode == 3) { if (dresidual) hipLaunchKernelGGL((bn_bwd_apply_kernel<DT, 3, true>), g, b, 0, st, dy, y, x, dx, dresidual, nvec, C8, ca, k0, k1, fa, fb);   \
                else hipLaunchKernelGGL((bn_bwd_apply_kernel<DT, 3, false>), g, b, 0, st, dy, y, x, dx, dresidual, nvec, C8, ca, k0, k1, fa, fb); }         \
    else if (rmode == 1) { if (dresidual) hipLaunchKernelGGL((bn_bwd_apply_kernel<DT, 1, true>), g, b, 0, st, dy, y, x, dx, dresidual, nvec, C8, ca, k0, k1, fa, fb);   \
                else hipLaunchKernelGGL((bn_bwd_apply_kernel<DT, 1, false>), g, b, 0, st, dy, y, x, dx, dresidual, nvec, C8, ca, k0, k1, fa, fb); }         \
    else      { if (dresidual) hipLaunchKernelGGL((bn_bwd_apply_kernel<DT, 0, true>), g, b, 0, st, dy, y, x, dx, dresidual, nvec, C8, ca, k0, k1, fa, fb);  \
                else hipLaunchKernelGGL((bn_bwd_apply_kernel<DT, 0, false>), g, b, 0, st, dy, y, x, dx, dresidual, nvec, C8, ca, k0, k1, fa, fb); }
    switch (dtype) {
        case RN_F32: RN_BN_BWD_APPLY(RN_F32) break;
        case RN_BF16: RN_BN_BWD_APPLY(RN_BF16) break;
        default: RN_BN_BWD_APPLY(RN_F16) break;
    }
#undef RN_BN_BWD_APPLY
    RN_LAUNCH_CHECK();
    return RN_OK;
}

bool dtype_ok(int dtype) { return dtype == RN_F32 || dtype == RN_BF16 || dtype == RN_F16; }

}  // namespace

// workspace: partial f32[BN_MAX_BLOCKS][2][C]
RN_API size_t rn_bn_workspace_bytes(int C) { return C > 0 ? sizeof(float) * (size_t)BN_MAX_BLOCKS * 2 * (size_t)C : 0; }

RN_API int rn_bn_stats(const void *x, int dtype, int64_t M, int C, const float *gamma, const float *beta, float *running_mean,
                       float *running_var, int64_t *num_batches_tracked, float momentum, float eps, float *save_mean,
                       float *save_invstd, float *coef, void *workspace, size_t workspace_bytes, void *stream)
{
    if (!x || !save_mean || !save_invstd || !coef || M <= 0 || C <= 0) return RN_EINVAL;
    if (C % 8) return RN_EUNSUPPORTED;
    if (!dtype_ok(dtype)) return RN_EINVAL;
    if (!rn::aligned(x, 16) || !rn::aligned(coef, 16) || !rn::aligned(save_mean, 16) || !rn::aligned(save_invstd, 16)) return RN_EALIGN;
    if (!workspace || workspace_bytes < rn_bn_workspace_bytes(C)) return RN_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    int nb = 0;
    const int rc = stats_partial_launch(x, dtype, M, C, (float *)workspace, st, &nb);
    if (rc != RN_OK) return rc;
    return stats_final_launch((const float *)workspace, nb, M, C, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps,
                              save_mean, save_invstd, coef, coef + C, st);
}

RN_API int rn_bn_stats_finalize(const float *partial, int nblocks, int64_t M, int C, const float *gamma, const float *beta,
                                float *running_mean, float *running_var, int64_t *num_batches_tracked, float momentum, float eps,
                                float *save_mean, float *save_invstd, float *coef, void *stream)
{
    if (!partial || nblocks <= 0 || !save_mean || !save_invstd || !coef || M <= 0 || C <= 0) return RN_EINVAL;
    return stats_final_launch(partial, nblocks, M, C, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps, save_mean,
                              save_invstd, coef, coef + C, (hipStream_t)stream);
}

RN_API int rn_bn_apply(const void *x, const void *residual, void *y, int dtype, int64_t M, int C, const float *coef, int relu,
                       uint8_t *relu_mask, void *stream)
{
    if (!x || !y || !coef || M <= 0 || C <= 0) return RN_EINVAL;
    if (C % 8) return RN_EUNSUPPORTED;
    if (!dtype_ok(dtype)) return RN_EINVAL;
    if (!rn::aligned(x, 16) || !rn::aligned(y, 16) || (residual && !rn::aligned(residual, 16)) || !rn::aligned(coef, 16)) return RN_EALIGN;
    return apply_launch(x, residual, y, dtype, M, C, coef, coef + C, relu, relu_mask, (hipStream_t)stream);
}

RN_API int rn_bn_apply_res_affine(const void *x, const void *residual, const float *res_coef, void *y, int dtype, int64_t M, int C,
                                  const float *coef, uint8_t *relu_mask, void *stream)
{
    if (!x || !y || !coef || !residual || !res_coef || M <= 0 || C <= 0) return RN_EINVAL;
    if (C % 8) return RN_EUNSUPPORTED;
    if (!dtype_ok(dtype)) return RN_EINVAL;
    if (!rn::aligned(x, 16) || !rn::aligned(y, 16) || !rn::aligned(residual, 16) || !rn::aligned(coef, 16) || !rn::aligned(res_coef, 16)) return RN_EALIGN;
    return apply_launch(x, residual, y, dtype, M, C, coef, coef + C, 1, relu_mask, (hipStream_t)stream, res_coef, res_coef + C);
}

RN_API int rn_bn_act_forward(const void *x, const void *residual, void *y, int dtype, int64_t M, int C,
                             const float *gamma, const float *beta, float *running_mean, float *running_var,
                             int64_t *num_batches_tracked, int training, float momentum, float eps, int relu,
                             float *save_mean, float *save_invstd, float *coef /*[2][C]*/, uint8_t *relu_mask, void *workspace,
                             size_t workspace_bytes, void *stream)
{
    if (!x || !y || !save_mean || !save_invstd || !coef || M <= 0 || C <= 0) return RN_EINVAL;
    if (C % 8) return RN_EUNSUPPORTED;
    if (!dtype_ok(dtype)) return RN_EINVAL;
    if (!rn::aligned(x, 16) || !rn::aligned(y, 16) || (residual && !rn::aligned(residual, 16)) || !rn::aligned(coef, 16) ||
        !rn::aligned(save_mean, 16) || !rn::aligned(save_invstd, 16))
        return RN_EALIGN;
    if (!training && (!running_mean || !running_var)) return RN_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    float *ca = coef, *cb = coef + C;
    if (training) {
        if (!workspace || workspace_bytes < rn_bn_workspace_bytes(C)) return RN_EWORKSPACE;
        int nb = 0;
        int rc = stats_partial_launch(x, dtype, M, C, (float *)workspace, st, &nb);
        if (rc != RN_OK) return rc;
        rc = stats_final_launch((const float *)workspace, nb, M, C, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps,
                                save_mean, save_invstd, ca, cb, st);
        if (rc != RN_OK) return rc;
    } else {
        hipLaunchKernelGGL(bn_eval_coef_kernel, dim3((C + 255) / 256), dim3(256), 0, st, C, gamma, beta, running_mean, running_var, eps,
                           save_mean, save_invstd, ca, cb);
        RN_LAUNCH_CHECK();
    }
    return apply_launch(x, residual, y, dtype, M, C, ca, cb, relu, relu_mask, st);
}

// relu: 0 none; 1: mask from y (the activation) when given, else recomputed from x and fwd_coef (no residual only);
//       2: y points to the byte mask the forward call wrote (relu_mask), one bit per element
static int bwd_rmode(const int relu, const void *y) { return !relu ? 0 : (relu == 2 ? 3 : (y ? 1 : 2)); }

RN_API int rn_bn_bwd_reduce(const void *dy, const void *y, const void *x, int dtype, int64_t M, int C, const float *gamma,
                            const float *save_mean, const float *save_invstd, const float *fwd_coef, int training, int relu,
                            float *dgamma, float *dbeta, float *coef3, void *workspace, size_t workspace_bytes, void *stream)
{
    if (!dy || !x || !save_mean || !save_invstd || !coef3 || M <= 0 || C <= 0) return RN_EINVAL;
    const int rmode = bwd_rmode(relu, y);
    if (rmode == 2 && !fwd_coef) return RN_EINVAL;
    if (rmode == 3 && !y) return RN_EINVAL;
    if (C % 8) return RN_EUNSUPPORTED;
    if (!dtype_ok(dtype)) return RN_EINVAL;
    if (!workspace || workspace_bytes < rn_bn_workspace_bytes(C)) return RN_EWORKSPACE;
    if (!rn::aligned(dy, 16) || !rn::aligned(x, 16) || (y && rmode != 3 && !rn::aligned(y, 16)) || !rn::aligned(coef3, 16) ||
        !rn::aligned(save_mean, 16) || !rn::aligned(save_invstd, 16))
        return RN_EALIGN;
    hipStream_t st = (hipStream_t)stream;
    int nb = 0;
    const int rc = bwd_partial_launch(dy, y, x, dtype, M, C, save_mean, save_invstd, fwd_coef, fwd_coef ? fwd_coef + C : nullptr, rmode,
                                      (float *)workspace, st, &nb);
    if (rc != RN_OK) return rc;
    return bwd_final_launch((const float *)workspace, nb, M, C, gamma, save_mean, save_invstd, training, dgamma, dbeta, coef3, coef3 + C,
                            coef3 + 2 * C, st);
}

RN_API int rn_bn_bwd_finalize(const float *partial, int nblocks, int64_t M, int C, const float *gamma, const float *save_mean,
                              const float *save_invstd, int training, float *dgamma, float *dbeta, float *coef3, void *stream)
{
    if (!partial || nblocks <= 0 || !save_mean || !save_invstd || !coef3 || M <= 0 || C <= 0) return RN_EINVAL;
    return bwd_final_launch(partial, nblocks, M, C, gamma, save_mean, save_invstd, training, dgamma, dbeta, coef3, coef3 + C, coef3 + 2 * C,
                            (hipStream_t)stream);
}

RN_API int rn_bn_bwd_apply(const void *dy, const void *y, const void *x, void *dx, void *dresidual, int dtype, int64_t M, int C,
                           const float *coef3, const float *fwd_coef, int relu_mode, void *stream)
{
    if (!dy || !x || !dx || !coef3 || M <= 0 || C <= 0) return RN_EINVAL;
    if (relu_mode < 0 || relu_mode > 3 || (relu_mode == 2 && (!fwd_coef || dresidual)) || ((relu_mode == 1 || relu_mode == 3) && !y)) return RN_EINVAL;
    if (C % 8) return RN_EUNSUPPORTED;
    if (!dtype_ok(dtype)) return RN_EINVAL;
    if (!rn::aligned(dy, 16) || !rn::aligned(x, 16) || !rn::aligned(dx, 16) || (y && relu_mode != 3 && !rn::aligned(y, 16)) ||
        (dresidual && !rn::aligned(dresidual, 16)) || !rn::aligned(coef3, 16))
        return RN_EALIGN;
    return bwd_apply_launch(dy, y, x, dx, dresidual, dtype, M, C, coef3, coef3 + C, coef3 + 2 * C, fwd_coef, fwd_coef ? fwd_coef + C : nullptr,
                            relu_mode, (hipStream_t)stream);
}

RN_API int rn_bn_act_backward(const void *dy, const void *y, const void *x, void *dx, void *dresidual, int dtype, int64_t M,
                              int C, const float *gamma, const float *save_mean, const float *save_invstd,
                              const float *fwd_coef /*[2][C] of the forward call, nullable*/, int training,
                              int relu, float *dgamma, float *dbeta, float *coef /*[3][C]*/, void *workspace,
                              size_t workspace_bytes, void *stream)
{
    if (!dy || !x || !dx || !save_mean || !save_invstd || !coef || M <= 0 || C <= 0) return RN_EINVAL;
    // ReLU mask: from y when given; without y it is recomputed from x and the forward coefficients, which is
    // only possible when the forward had no residual input
    const int rmode = bwd_rmode(relu, y);
    if (rmode == 2 && (!fwd_coef || dresidual)) return RN_EINVAL;
    if (rmode == 3 && !y) return RN_EINVAL;
    const float *fa = fwd_coef, *fb = fwd_coef ? fwd_coef + C : nullptr;
    if (C % 8) return RN_EUNSUPPORTED;
    if (!dtype_ok(dtype)) return RN_EINVAL;
    if (!workspace || workspace_bytes < rn_bn_workspace_bytes(C)) return RN_EWORKSPACE;
    if (!rn::aligned(dy, 16) || !rn::aligned(x, 16) || !rn::aligned(dx, 16) || (y && rmode != 3 && !rn::aligned(y, 16)) ||
        (dresidual && !rn::aligned(dresidual, 16)) || !rn::aligned(coef, 16) || !rn::aligned(save_mean, 16) ||
        !rn::aligned(save_invstd, 16))
        return RN_EALIGN;
    hipStream_t st = (hipStream_t)stream;
    float *ca = coef, *k0 = coef + C, *k1 = coef + 2 * C;
    int nb = 0;
    int rc = bwd_partial_launch(dy, y, x, dtype, M, C, save_mean, save_invstd, fa, fb, rmode, (float *)workspace, st, &nb);
    if (rc != RN_OK) return rc;
    rc = bwd_final_launch((const float *)workspace, nb, M, C, gamma, save_mean, save_invstd, training, dgamma, dbeta, ca, k0, k1, st);
    if (rc != RN_OK) return rc;
    return bwd_apply_launch(dy, y, x, dx, dresidual, dtype, M, C, ca, k0, k1, fa, fb, rmode, st);
}

// ================================================================ bias (+ ReLU) (+ position mask)
// y = mask[pos] ? act(x + bias[c]) : 0 on channels-last [M = N*H*W][C] activations, and its backward
// (dx = act'(y) * dy under the mask, dbias[c] = sum dx) -- the conv epilogue of the head towers and of
// the packed level canvas (layers.py): PyTorch-ROCm runs a biased conv as conv + broadcast add, the
// ReLU as a third kernel, and their backward as threshold + a separate channel reduction.  The mask
// (one byte per spatial position, shared by all images) zeroes the gaps between the pyramid levels
// packed into one canvas so that the next 3x3 conv sees zero padding there.
namespace {

template <int DT, bool RELU>
__global__ __launch_bounds__(BN_BLOCK) void bias_act_kernel(const void *__restrict__ x, void *__restrict__ y, const int64_t nvec,
                                                            const int C8, const int64_t HW, const float *__restrict__ bias,
                                                            const uint8_t *__restrict__ mask)
{
    for (int64_t v = (int64_t)blockIdx.x * BN_BLOCK + threadIdx.x; v < nvec; v += (int64_t)gridDim.x * BN_BLOCK) {
        const int64_t row = v / C8;
        const int cg = (int)(v - row * C8);
        float f[8];
        vec8<DT>::ld(x, v, f);
        const rn::f32x4 b0 = ((const rn::f32x4 *)bias)[2 * cg], b1 = ((const rn::f32x4 *)bias)[2 * cg + 1];
        const float b[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
        const bool keep = !mask || mask[row % HW];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float o = f[j] + b[j];
            if (RELU) o = o > 0.0f ? o : 0.0f;
            f[j] = keep ? o : 0.0f;
        }
        vec8<DT>::st(y, v, f);
    }
}

// dx (optional) + per-block partial channel sums of dx; same thread layout as bn_stats_partial_kernel
template <int DT, bool RELU, bool WRITE_DX>
__global__ __launch_bounds__(BN_BLOCK) void bias_act_bwd_kernel(const void *__restrict__ dy, const void *__restrict__ y,
                                                                void *__restrict__ dx, const int64_t M, const int C, const int64_t HW,
                                                                const uint8_t *__restrict__ mask, float *__restrict__ partial)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];      // [lanes][C]  (only when lanes > 1)
    const Split sp = split_of(C);
    const int64_t rows_per_block = (M + gridDim.x - 1) / gridDim.x;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block, r1 = min(r0 + rows_per_block, M);
    for (int gi = 0; gi < sp.groups_per_thread; ++gi) {
        const int cg = (sp.lanes > 1) ? (int)(threadIdx.x % sp.C8) : (int)threadIdx.x + gi * BN_BLOCK;
        const int rl = (sp.lanes > 1) ? (int)(threadIdx.x / sp.C8) : 0;
        float s[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) s[j] = 0.0f;
        if (cg < sp.C8 && rl < sp.lanes) {
            int64_t r = r0 + rl;
            for (; r + sp.lanes < r1; r += 2 * sp.lanes) {                // 2 rows x 2 tensors in flight per thread
                float g[2][8], yy[2][8];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int64_t v = (r + u * sp.lanes) * sp.C8 + cg;
                    vec8<DT>::ld(dy, v, g[u]);
                    if (RELU) vec8<DT>::ld(y, v, yy[u]);
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int64_t rr = r + u * sp.lanes;
                    const bool keep = !mask || mask[rr % HW];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float gj = (keep && !(RELU && !(yy[u][j] > 0.0f))) ? g[u][j] : 0.0f;
                        g[u][j] = gj;
                        s[j] += gj;
                    }
                    if (WRITE_DX) vec8<DT>::st(dx, rr * sp.C8 + cg, g[u]);
                }
            }
            for (; r < r1; r += sp.lanes) {
                const int64_t v = r * sp.C8 + cg;
                float g[8], yy[8];
                vec8<DT>::ld(dy, v, g);
                if (RELU) vec8<DT>::ld(y, v, yy);
                const bool keep = !mask || mask[r % HW];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float gj = (keep && !(RELU && !(yy[j] > 0.0f))) ? g[j] : 0.0f;
                    g[j] = gj;
                    s[j] += gj;
                }
                if (WRITE_DX) vec8<DT>::st(dx, v, g);
            }
        }
        if (sp.lanes > 1) {
            if (cg < sp.C8 && rl < sp.lanes) {
#pragma unroll
                for (int j = 0; j < 8; ++j) smem[rl * C + cg * 8 + j] = s[j];
            }
            __syncthreads();
            for (int c = threadIdx.x; c < C; c += BN_BLOCK) {
                float t = 0.0f;
                for (int l = 0; l < sp.lanes; ++l) t += smem[l * C + c];
                partial[(int64_t)blockIdx.x * C + c] = t;
            }
            __syncthreads();
        } else if (cg < sp.C8) {
#pragma unroll
            for (int j = 0; j < 8; ++j) partial[(int64_t)blockIdx.x * C + cg * 8 + j] = s[j];
        }
    }
}

__global__ __launch_bounds__(256) void bias_grad_final_kernel(const float *__restrict__ partial, const int nblocks, const int C,
                                                              float *__restrict__ dbias)
{
    constexpr int LANES = 256 / FIN_CH;                             // (256-thread blocks: few partials, few calls)
    __shared__ double sh[LANES][FIN_CH];
    const int ch = threadIdx.x % FIN_CH, ln = threadIdx.x / FIN_CH;
    const int c = blockIdx.x * FIN_CH + ch;
    double ls = 0.0;
    if (c < C)
        for (int b = ln; b < nblocks; b += LANES) ls += (double)partial[(int64_t)b * C + c];
    sh[ln][ch] = ls;
    __syncthreads();
    if (ln == 0 && c < C) {
        double s = 0.0;
        for (int l = 0; l < LANES; ++l) s += sh[l][ch];
        dbias[c] = (float)s;
    }
}

}  // namespace

RN_API int rn_bias_act_forward(const void *x, const float *bias, const uint8_t *mask, void *y, int dtype, int64_t M, int C,
                               int64_t HW, int relu, void *stream)
{
    if (!x || !bias || !y || M <= 0 || C <= 0 || (mask && HW <= 0)) return RN_EINVAL;
    if (C % 8) return RN_EUNSUPPORTED;
    if (dtype != RN_F32 && dtype != RN_BF16 && dtype != RN_F16) return RN_EINVAL;
    if (!rn::aligned(x, 16) || !rn::aligned(y, 16) || !rn::aligned(bias, 16)) return RN_EALIGN;
    hipStream_t st = (hipStream_t)stream;
    const int64_t nvec = M * (C / 8);
    const dim3 g(apply_blocks(nvec)), b(BN_BLOCK);
    const int C8 = C / 8;
    if (!mask) HW = 1;
#define RN_BA(DT) if (relu) hipLaunchKernelGGL((bias_act_kernel<DT, true>), g, b, 0, st, x, y, nvec, C8, HW, bias, mask); \
                  else hipLaunchKernelGGL((bias_act_kernel<DT, false>), g, b, 0, st, x, y, nvec, C8, HW, bias, mask);
    switch (dtype) {
        case RN_F32: RN_BA(RN_F32) break;
        case RN_BF16: RN_BA(RN_BF16) break;
        default: RN_BA(RN_F16) break;
    }
#undef RN_BA
    RN_LAUNCH_CHECK();
    return RN_OK;
}

RN_API int rn_bias_act_backward(const void *dy, const void *y, const uint8_t *mask, void *dx, float *dbias, int dtype, int64_t M,
                                int C, int64_t HW, int relu, void *workspace, size_t workspace_bytes, void *stream)
{
    if (!dy || !dbias || M <= 0 || C <= 0 || (mask && HW <= 0)) return RN_EINVAL;
    if (relu && !y) return RN_EINVAL;
    if ((relu || mask) && !dx) return RN_EINVAL;               // without them dx == dy and may be omitted
    if (C % 8) return RN_EUNSUPPORTED;
    if (dtype != RN_F32 && dtype != RN_BF16 && dtype != RN_F16) return RN_EINVAL;
    if (!workspace || workspace_bytes < rn_bn_workspace_bytes(C)) return RN_EWORKSPACE;
    if (!rn::aligned(dy, 16) || (y && !rn::aligned(y, 16)) || (dx && !rn::aligned(dx, 16))) return RN_EALIGN;
    hipStream_t st = (hipStream_t)stream;
    const int nb = reduce_blocks(M);
    const size_t lds = reduce_lds(C) / 2;
    float *partial = (float *)workspace;
    if (!mask) HW = 1;
#define RN_BAB(DT)                                                                                                                         \
    if (relu) hipLaunchKernelGGL((bias_act_bwd_kernel<DT, true, true>), dim3(nb), dim3(BN_BLOCK), lds, st, dy, y, dx, M, C, HW, mask, partial); \
    else if (dx) hipLaunchKernelGGL((bias_act_bwd_kernel<DT, false, true>), dim3(nb), dim3(BN_BLOCK), lds, st, dy, y, dx, M, C, HW, mask, partial); \
    else hipLaunchKernelGGL((bias_act_bwd_kernel<DT, false, false>), dim3(nb), dim3(BN_BLOCK), lds, st, dy, y, dx, M, C, HW, mask, partial);
    switch (dtype) {
        case RN_F32: RN_BAB(RN_F32) break;
        case RN_BF16: RN_BAB(RN_BF16) break;
        default: RN_BAB(RN_F16) break;
    }
#undef RN_BAB
    RN_LAUNCH_CHECK();
    hipLaunchKernelGGL(bias_grad_final_kernel, dim3((C + FIN_CH - 1) / FIN_CH), dim3(256), 0, st, partial, nb, C, dbias);
    RN_LAUNCH_CHECK();
    return RN_OK;
}
