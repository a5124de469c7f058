// Forward product of the NARROW 3x3 / stride-1 / pad-1 convolutions of the ResNet trunk at 64 channels -- conv2 of the layer1 bottlenecks
// (reference: retinanet/backbone.py:112,128) and, with tap-reversed role-swapped weights, their data gradient -- bf16 channels-last:
//
//   y[n][h][w][co] = sum over (dy, dx, ci) of x[n][h + dy - 1][w + dx - 1][ci] * w[co][dy][dx][ci]        (fp32 accumulation)
//
// Why a kernel of its own: the 256-channel implicit-GEMM kernels of conv.hip would idle 3/4 of every MFMA at 64 output
// channels, and a GEMM formulation re-reads every input pixel once per tap (pw_gemm's 3x3 mode: 42 flop per byte of L2 -> LDS traffic).
// CK's grouped-conv kernel, which MIOpen picks, runs the layer1 shape (8 x 200 x 336 x 64: 39.6 GFLOP, 69 MB in, 69 MB out) in 76 us.
// Here the WEIGHTS live in registers and the pixels in a ring of input rows:
//   * a wave owns 32 output channels x 64 pixels of an image row: its 32 x 576 weight block is 36 A-fragments of
//     v_mfma_f32_32x32x16_bf16 = 144 registers, loaded once per workgroup (rows = output channels, so D is [channel][pixel] and a lane ends
//     up with 4 CONSECUTIVE channels of a pixel; three lane swaps make that 16-byte chunks in store order, no LDS transpose);
//   * a workgroup (4 waves: 2 channel halves x 2 pixel halves) walks a band of output rows of one 128-pixel column strip; every input row
//     is staged ONCE per band (130 pixels x 128 bytes by LDS-DMA, 16-byte chunks XOR-swizzled by pixel pair: conflict-free ds_read_b128) into a ring
//     of four rows, the three vertical taps read three ring rows, the three horizontal taps the same row one pixel further;
//   * per output row a wave issues 72 MFMAs and 72 fragment reads (issued two K-steps ahead); the next input row is in flight (LDS-DMA into
//     the free ring slot) under them; one barrier per row.  Two workgroups per CU (68 KB of LDS each) overlap each other.
#include "rn_common.hpp"

namespace {

using rn::f32x16;
typedef __attribute__((address_space(3))) void *lds_void_ptr;

constexpr int N3_THREADS = 256;
constexpr int N3_TW = 128;                       // output pixels of a strip
constexpr int N3_PIECES = 17;                    // LDS-DMA pieces of a staged row: 8 pixels x 128 bytes each (130 pixels are read)
constexpr int N3_ROWBYTES = N3_PIECES * 1024;    // 136 staged pixels x (64 channels x 2 bytes)
constexpr int N3_LDS = 4 * N3_ROWBYTES;          // ring of four input rows: 69 632 bytes -> two workgroups per CU
#ifndef N3_AHEAD
#define N3_AHEAD 2                               // K-steps between a fragment read and its MFMAs (<= 5)
#endif

struct N3Args {
    const uint16_t *x;      // [N][H][W][64]
    const uint16_t *w;      // [64][9][64]  (channels-last memory of a [64, 64, 3, 3] weight)
    uint16_t *y;            // [N][H][W][64]
    const void *zeros;      // >= 128 zero bytes: what the staged pixels outside the image read
    const float *bias;      // [64] or null: y = act(conv + bias), act = ReLU when relu (inference: the folded BatchNorm of the layer)
    int relu;
    int N, H, W;
    int strips, bands, rows_per_band;
};

__device__ __forceinline__ int n3_swz(const int sp) { return (sp >> 1) & 7; }

// The fragment reads are inline asm, their waits counted by hand: the compiler would order every LDS read it knows of behind the LDS-DMA in
// flight (vmcnt(0)) and so serialise the fetch of the next input row with this row's MFMAs (same reason as csrc/wgrad3x3.hip).
#define N3_DS_READ(dst, addr, OFF) asm volatile("ds_read_b128 %0, %1 offset:" #OFF : "=v"(dst) : "v"(addr))
#define N3_WAIT_LGKM(N) asm volatile("s_waitcnt lgkmcnt(" #N ")" ::: "memory"); __builtin_amdgcn_sched_barrier(0);

template <int DT>
__global__ __launch_bounds__(N3_THREADS, 2) void conv3x3_narrow64_kernel(const N3Args a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cw = wave & 1, pw = wave >> 1;                     // channel half, pixel half
    int u_ = blockIdx.x;
    const int band = u_ % a.bands; u_ /= a.bands;
    const int strip = u_ % a.strips;
    const int n = u_ / a.strips;
    const int x0 = strip * N3_TW;
    const int y0 = band * a.rows_per_band, y1 = min(y0 + a.rows_per_band, a.H);
    if (y0 >= y1) return;

    // ---- staging of one input row (image row yy, pixels x0 - 1 .. x0 + 128): LDS-DMA, 16 bytes per lane; a piece lands as 1 KiB of
    // consecutive lanes, so the XOR swizzle of the 16-byte chunks (by pixel pair) is applied on the SOURCE side
    const int lpx = lane >> 3, lch = lane & 7;
    uint32_t src_off[2];                                          // byte offset inside the 8 pixels of a piece, by piece parity
#pragma unroll
    for (int par = 0; par < 2; ++par) src_off[par] = (uint32_t)(lpx * 128 + ((lch ^ ((par * 4 + (lane >> 4)) & 7)) << 4));
    const unsigned char *const zsrc = (const unsigned char *)a.zeros + lch * 16;
    auto stage = [&](const int yy) {
        const bool row_ok = (unsigned)yy < (unsigned)a.H;
        unsigned char *const rb = lds + ((yy + 4) & 3) * N3_ROWBYTES;
        const int64_t row_byte = (((int64_t)n * a.H + (row_ok ? yy : 0)) * a.W + (x0 - 1)) * 128;        // of staged pixel 0 (may be one pixel before the row)
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int u = wave + 4 * i;                           // wave-uniform
            if (u < N3_PIECES) {
                const int sp = 8 * u + lpx, xx = x0 - 1 + sp;
                const bool ok = row_ok && sp < N3_TW + 2 && (unsigned)xx < (unsigned)a.W;
                const unsigned char *const src = ok ? (const unsigned char *)a.x + (row_byte + u * 1024 + src_off[u & 1]) : zsrc;
                __builtin_amdgcn_global_load_lds((const void *)src, (lds_void_ptr)(rb + u * 1024), 16, 0, 0);
            }
        }
    };
    stage(y0 - 1); stage(y0); stage(y0 + 1);

    // ---- weights: this wave's 32 output channels x (9 taps x 64 input channels), as MFMA A-fragments (row = output channel)
    typedef typename rn::mma<DT>::frag frag8;
    frag8 wf[9][4];
    const int kh = lane >> 5;
    {
        const int co = cw * 32 + (lane & 31);
        const uint16_t *wp = a.w + (int64_t)co * 9 * 64 + kh * 8;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int kc = 0; kc < 4; ++kc) wf[t][kc] = *(const frag8 *)(wp + t * 64 + kc * 16);
    }
    // fragment addresses of pixel block 0 (block 1: + 32 pixels = + 4096 bytes, same swizzle): horizontal tap dx reads staged pixel
    // p + dx (staged pixel 0 is x0 - 1), 16-channel chunk kc
    uint32_t col[3][4];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
            const int sp = pw * 64 + (lane & 31) + dx;
            col[dx][kc] = (uint32_t)(sp * 128 + (((kc * 2 + kh) ^ n3_swz(sp)) << 4));
        }
    const uint32_t lds_base = (uint32_t)(uintptr_t)lds;
    // epilogue constants: the bias of this lane's 16 output channels (acc[.][4 g + j] is channel 32 cw + 8 g + 4 kh + j)
    float eb[16];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int j = 0; j < 4; ++j) eb[4 * g + j] = a.bias ? a.bias[cw * 32 + 8 * g + 4 * kh + j] : 0.0f;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int y = y0; y < y1; ++y) {
        const bool stage_next = (y + 2 <= y1);
        if (stage_next) stage(y + 2);                            // the row the NEXT step needs last; slot (y + 2) & 3 held row y - 2: nobody reads it now
        f32x16 acc[2];
#pragma unroll
        for (int pb = 0; pb < 2; ++pb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[pb][r] = 0.0f;
        uint32_t rbase[3];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) rbase[dy] = lds_base + (uint32_t)(((y + dy - 1 + 4) & 3) * N3_ROWBYTES);
        // 36 K-steps (tap, 16-channel chunk) x 2 pixel blocks; the two fragments of step s are read N3_AHEAD steps (2 N3_AHEAD MFMAs) ahead
        frag8 f[N3_AHEAD + 1][2];
#define N3_READ(S) { const uint32_t ad_ = rbase[((S) >> 2) / 3] + col[((S) >> 2) % 3][(S) & 3]; \
                     N3_DS_READ(f[(S) % (N3_AHEAD + 1)][0], ad_, 0); N3_DS_READ(f[(S) % (N3_AHEAD + 1)][1], ad_, 4096); }
#pragma unroll
        for (int s = 0; s < N3_AHEAD; ++s) N3_READ(s)
#pragma unroll
        for (int s = 0; s < 36; ++s) {
            if (s + N3_AHEAD < 36) { N3_READ(s + N3_AHEAD) }
            // outstanding after this step's issue: steps s .. min(s + N3_AHEAD, 35), two reads each; all but step s's may stay in flight
            constexpr int LAST = 35;
            const int later = (s + N3_AHEAD < LAST ? s + N3_AHEAD : LAST) - s;
            switch (later) {
                case 0: N3_WAIT_LGKM(0) break;
                case 1: N3_WAIT_LGKM(2) break;
                case 2: N3_WAIT_LGKM(4) break;
                case 3: N3_WAIT_LGKM(6) break;
                case 4: N3_WAIT_LGKM(8) break;
                default: N3_WAIT_LGKM(10) break;
            }
            acc[0] = rn::mma<DT>::m32(wf[s >> 2][s & 3], f[s % (N3_AHEAD + 1)][0], acc[0]);
            acc[1] = rn::mma<DT>::m32(wf[s >> 2][s & 3], f[s % (N3_AHEAD + 1)][1], acc[1]);
            __builtin_amdgcn_sched_barrier(0);
        }
#undef N3_READ
        // D[channel][pixel]: lane = pixel (lane & 31), channels 8 g + 4 (lane >> 5) + j in acc[4 g + j].  Three rounds of lane swaps turn
        // that into full 16-byte chunks laid out for the store: (1) v_permlane32_swap between the two halves of a channel-group pair
        // gives a lane 8 consecutive channels of its pixel (lanes 0..31 the even group, 32..63 the odd one); (2) the same swap between
        // the two pairs and (3) v_permlane16_swap regroup the four 16-lane rows so that ONE store instruction carries all four chunks
        // (64 contiguous bytes = this wave's 32 channels) of 16 pixels: row k of the wave writes chunk k.  (8-byte stores straight from
        // the accumulators: +4 us per launch; 32-byte pieces, rounds 1 only: +1.)
#pragma unroll
        for (int pb = 0; pb < 2; ++pb) {
            if (x0 + pw * 64 + pb * 32 >= a.W) continue;          // (wave-uniform: the tile lies right of the image -- the count below relies on it)
            if (a.bias) {                                          // (uniform)
#pragma unroll
                for (int r = 0; r < 16; ++r) { const float t = acc[pb][r] + eb[r]; acc[pb][r] = (a.relu && !(t > 0.0f)) ? 0.0f : t; }
            }
            uint32_t v[2][4];
#pragma unroll
            for (int gp = 0; gp < 2; ++gp) {
                const uint32_t ax = rn::dt<DT>::pk(acc[pb][8 * gp + 0], acc[pb][8 * gp + 1]), ay = rn::dt<DT>::pk(acc[pb][8 * gp + 2], acc[pb][8 * gp + 3]);
                const uint32_t bx = rn::dt<DT>::pk(acc[pb][8 * gp + 4], acc[pb][8 * gp + 5]), by = rn::dt<DT>::pk(acc[pb][8 * gp + 6], acc[pb][8 * gp + 7]);
                const auto sx = __builtin_amdgcn_permlane32_swap(ax, bx, false, false);
                const auto sy = __builtin_amdgcn_permlane32_swap(ay, by, false, false);
                v[gp][0] = sx[0]; v[gp][1] = sy[0]; v[gp][2] = sx[1]; v[gp][3] = sy[1];
            }
            // rows of 16 lanes: v[0] = [px 0-15 chunk 0 | px 16-31 chunk 0 | px 0-15 chunk 1 | px 16-31 chunk 1], v[1] the same for chunks 2, 3
            uint32_t lo[4], hi[4];                                  // lo: pixels 0..15 of the tile, hi: pixels 16..31; row k = chunk k
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const auto t = __builtin_amdgcn_permlane32_swap(v[0][d], v[1][d], false, false);
                const auto u = __builtin_amdgcn_permlane16_swap(t[0], t[1], false, false);
                lo[d] = u[0]; hi[d] = u[1];
            }
            const int chunk = lane >> 4;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int pxb = x0 + pw * 64 + pb * 32 + 16 * h;                // (wave-uniform)
                if (pxb >= a.W) continue;                                       // uniform branch: a half tile right of the image issues NO store (counted below)
                const int px = pxb + (lane & 15);
                if (px < a.W) {
                    uint16_t *const yp = a.y + (((int64_t)n * a.H + y) * a.W + px) * 64 + cw * 32 + 8 * chunk;
                    *(rn::u32x4 *)yp = h ? rn::u32x4{hi[0], hi[1], hi[2], hi[3]} : rn::u32x4{lo[0], lo[1], lo[2], lo[3]};
                }
            }
        }
        // The staged row has landed: vmcnt counts loads and stores in issue order on gfx9, so everything but the stores just issued
        // must have retired -- not the stores themselves, whose acknowledgement from the L2 takes longer than the step's MFMAs leave
        // to hide.  The count is the number of 16-pixel half tiles that START inside the image (each issues exactly one store
        // instruction: the uniform branch above skips the others, and a half tile that starts inside has lane 0 active), 0..4.
        {
            const int xb = x0 + pw * 64;
            const int stores = (xb < a.W ? 1 : 0) + (xb + 16 < a.W ? 1 : 0) + (xb + 32 < a.W ? 1 : 0) + (xb + 48 < a.W ? 1 : 0);
            switch (stores) {
                case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
                case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
                case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
                case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
                default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            }
        }
        // (a bare s_barrier: __syncthreads() carries a workgroup release fence = vmcnt(0), which would wait for the stores after all;
        // the ring is written by the DMA waited for above and read by the asm reads, all retired at the last K-step's lgkmcnt(0))
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }
}

}  // namespace

// x, y [N][H][W][C] bf16 channels-last; w [C][3][3][C] (channels-last memory of [C, C, 3, 3]); C = 64.
RN_API int rn_conv3x3_narrow_forward(const void *x, const void *w, const float *bias, void *y, int dtype, int N, int H, int W, int C, int relu,
                                     const void *zero_page, void *stream)
{
    if (!x || !w || !y || !zero_page || N <= 0 || H <= 0 || W <= 0) return RN_EINVAL;
    if ((dtype != RN_BF16 && dtype != RN_F16) || C != 64) return RN_EUNSUPPORTED;
    if ((int64_t)N * H * W >= ((int64_t)1 << 31)) return RN_EUNSUPPORTED;
    if (!rn::aligned(x, 16) || !rn::aligned(w, 16) || !rn::aligned(y, 16)) return RN_EALIGN;
    N3Args a;
    a.x = (const uint16_t *)x; a.w = (const uint16_t *)w; a.y = (uint16_t *)y;
    a.zeros = zero_page; a.bias = bias; a.relu = (bias && relu) ? 1 : 0;
    a.N = N; a.H = H; a.W = W;
    a.strips = (W + N3_TW - 1) / N3_TW;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    // two workgroups per CU: bands so that (images x strips x bands) fills them once (a band re-stages two halo rows)
    int bands = (2 * cus) / (N * a.strips);
    if (bands < 1) bands = 1;
    if (bands > H) bands = H;
    a.rows_per_band = (H + bands - 1) / bands;
    a.bands = (H + a.rows_per_band - 1) / a.rows_per_band;
#define N3_LAUNCH(DT)                                                                                                                  \
    {   static rn::DynLdsOptIn opt_in = {};                                                                                            \
        const int rc = opt_in.ensure((const void *)conv3x3_narrow64_kernel<DT>, N3_LDS);                                               \
        if (rc != RN_OK) return rc;                                                                                                    \
        hipLaunchKernelGGL(conv3x3_narrow64_kernel<DT>, dim3((unsigned)(N * a.strips * a.bands)), dim3(N3_THREADS), N3_LDS, (hipStream_t)stream, a); }
    if (dtype == RN_F16) N3_LAUNCH(RN_F16) else N3_LAUNCH(RN_BF16)
#undef N3_LAUNCH
    RN_LAUNCH_CHECK();
    return RN_OK;
}
