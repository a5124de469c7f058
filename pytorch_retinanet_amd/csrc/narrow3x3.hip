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
//     up with 4 CONSECUTIVE channels of a pixel: 8-byte stores, no LDS transpose);
//   * a workgroup (4 waves: 2 channel halves x 2 pixel halves) walks a band of output rows of one 128-pixel column strip; every input row
//     is staged ONCE per band (130 pixels x 128 bytes, 16-byte chunks XOR-swizzled by pixel pair: conflict-free ds_read_b128) into a ring
//     of four rows, the three vertical taps read three ring rows, the three horizontal taps the same row one pixel further;
//   * per output row a wave issues 72 MFMAs and 72 fragment reads; the next input row is in flight (global -> registers) under them and goes
//     into the free ring slot before the single barrier of the step.  Two workgroups per CU (66.5 KB of LDS each) overlap each other.
#include "rn_common.hpp"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int N3_THREADS = 256;
constexpr int N3_TW = 128;                       // output pixels of a strip
constexpr int N3_SPX = N3_TW + 2;                // staged pixels of an input row (one halo pixel each side)
constexpr int N3_ROWBYTES = N3_SPX * 128;        // 64 channels x 2 bytes per pixel
constexpr int N3_LDS = 4 * N3_ROWBYTES;          // ring of four input rows: 66 560 bytes
constexpr int N3_CHUNKS = N3_SPX * 8;            // 16-byte chunks of a staged row
constexpr int N3_ST = (N3_CHUNKS + N3_THREADS - 1) / N3_THREADS;     // chunks per thread (5)

struct N3Args {
    const uint16_t *x;      // [N][H][W][64]
    const uint16_t *w;      // [64][9][64]  (channels-last memory of a [64, 64, 3, 3] weight)
    uint16_t *y;            // [N][H][W][64]
    int N, H, W;
    int strips, bands, rows_per_band;
};

__device__ __forceinline__ int n3_swz(const int sp) { return (sp >> 1) & 7; }

__global__ __launch_bounds__(N3_THREADS, 2) void conv3x3_narrow64_kernel(const N3Args a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cw = wave & 1, pw = wave >> 1;                     // channel half, pixel half
    int u = blockIdx.x;
    const int band = u % a.bands; u /= a.bands;
    const int strip = u % a.strips;
    const int n = u / a.strips;
    const int x0 = strip * N3_TW;
    const int y0 = band * a.rows_per_band, y1 = min(y0 + a.rows_per_band, a.H);
    if (y0 >= y1) return;

    // ---- weights: this wave's 32 output channels x (9 taps x 64 input channels), as MFMA A-fragments (row = output channel)
    bf16x8 wf[9][4];
    {
        const int co = cw * 32 + (lane & 31), kh = lane >> 5;
        const uint16_t *wp = a.w + (int64_t)co * 9 * 64 + kh * 8;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int kc = 0; kc < 4; ++kc) wf[t][kc] = *(const bf16x8 *)(wp + t * 64 + kc * 16);
    }

    // ---- staging of one input row (image row yy, pixels x0 - 1 .. x0 + 128) : global -> registers -> ring slot
    rn::u32x4 st[N3_ST];
    auto load_row = [&](const int yy) {
        const bool row_ok = yy >= 0 && yy < a.H;
#pragma unroll
        for (int i = 0; i < N3_ST; ++i) {
            const int q = tid + i * N3_THREADS;
            const int sp = q >> 3, c = q & 7;
            const int xx = x0 - 1 + sp;
            const bool ok = row_ok && q < N3_CHUNKS && xx >= 0 && xx < a.W;
            const int64_t e = (((int64_t)n * a.H + (ok ? yy : 0)) * a.W + (ok ? xx : 0)) * 64 + c * 8;      // (clamped: always a valid address)
            st[i] = *(const rn::u32x4 *)(a.x + e);               // (zeroed in store_row: the select must not make the MFMAs wait for the load)
        }
    };
    auto store_row = [&](const int yy) {
        unsigned char *const rb = lds + ((yy + 4) & 3) * N3_ROWBYTES;
        const bool row_ok = yy >= 0 && yy < a.H;
#pragma unroll
        for (int i = 0; i < N3_ST; ++i) {
            const int q = tid + i * N3_THREADS;
            if (q < N3_CHUNKS) {
                const int sp = q >> 3, c = q & 7;
                const int xx = x0 - 1 + sp;
                const bool ok = row_ok && xx >= 0 && xx < a.W;
                *(rn::u32x4 *)(rb + sp * 128 + ((c ^ n3_swz(sp)) << 4)) = ok ? st[i] : rn::u32x4{0u, 0u, 0u, 0u};
            }
        }
    };
    load_row(y0 - 1); store_row(y0 - 1);
    load_row(y0);     store_row(y0);
    load_row(y0 + 1); store_row(y0 + 1);
    __syncthreads();

    // fragment addresses: pixel block pb, horizontal tap dx -> staged pixel sp = p + dx (staged pixel 0 is x0 - 1)
    const int kh = lane >> 5;
    int sp_off[2][3], sp_swz[2][3];
#pragma unroll
    for (int pb = 0; pb < 2; ++pb)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int sp = pw * 64 + pb * 32 + (lane & 31) + dx;
            sp_off[pb][dx] = sp * 128;
            sp_swz[pb][dx] = n3_swz(sp);
        }

    for (int y = y0; y < y1; ++y) {
        if (y + 2 <= y1) load_row(y + 2);                        // the row the NEXT step needs last (y + 2 = (y + 1) + 1), in flight under the MFMAs
        f32x16 acc[2];
#pragma unroll
        for (int pb = 0; pb < 2; ++pb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[pb][r] = 0.0f;
        // 36 K-steps (tap, 16-channel chunk) x 2 pixel blocks; the fragments of step s + 1 are read before the MFMAs of step s issue
        const unsigned char *rb3[3];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) rb3[dy] = lds + ((y + dy - 1 + 4) & 3) * N3_ROWBYTES;
        auto frag = [&](const int s, const int pb) -> bf16x8 {
            const int t = s >> 2, kc = s & 3, dy = t / 3, dx = t % 3;
            return *(const bf16x8 *)(rb3[dy] + sp_off[pb][dx] + (((kc * 2 + kh) ^ sp_swz[pb][dx]) << 4));
        };
        bf16x8 cur[2] = {frag(0, 0), frag(0, 1)}, nxt[2];
#pragma unroll
        for (int s = 0; s < 36; ++s) {
            if (s + 1 < 36) { nxt[0] = frag(s + 1, 0); nxt[1] = frag(s + 1, 1); }
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[s >> 2][s & 3], cur[0], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[s >> 2][s & 3], cur[1], acc[1], 0, 0, 0);
            cur[0] = nxt[0]; cur[1] = nxt[1];
        }
        // D[channel][pixel]: lane = pixel (lane & 31), channels 8 g + 4 (lane >> 5) + j in acc[4 g + j]
#pragma unroll
        for (int pb = 0; pb < 2; ++pb) {
            const int px = x0 + pw * 64 + pb * 32 + (lane & 31);
            if (px < a.W) {
                uint16_t *const yp = a.y + (((int64_t)n * a.H + y) * a.W + px) * 64 + cw * 32 + 4 * kh;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    rn::u32x2 o;
                    o.x = rn::dt<RN_BF16>::pk(acc[pb][4 * g + 0], acc[pb][4 * g + 1]);
                    o.y = rn::dt<RN_BF16>::pk(acc[pb][4 * g + 2], acc[pb][4 * g + 3]);
                    *(rn::u32x2 *)(yp + 8 * g) = o;
                }
            }
        }
        if (y + 2 <= y1) store_row(y + 2);                       // slot (y + 2) & 3 held row y - 2: nobody reads it in this step
        __syncthreads();
    }
}

}  // namespace

// x, y [N][H][W][C] bf16 channels-last; w [C][3][3][C] (channels-last memory of [C, C, 3, 3]); C = 64.
RN_API int rn_conv3x3_narrow_forward(const void *x, const void *w, void *y, int dtype, int N, int H, int W, int C, void *stream)
{
    if (!x || !w || !y || N <= 0 || H <= 0 || W <= 0) return RN_EINVAL;
    if (dtype != RN_BF16 || C != 64) return RN_EUNSUPPORTED;
    if ((int64_t)N * H * W >= ((int64_t)1 << 31)) return RN_EUNSUPPORTED;
    if (!rn::aligned(x, 16) || !rn::aligned(w, 16) || !rn::aligned(y, 16)) return RN_EALIGN;
    N3Args a;
    a.x = (const uint16_t *)x; a.w = (const uint16_t *)w; a.y = (uint16_t *)y;
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
    static rn::DynLdsOptIn opt_in = {};
    { const int rc = opt_in.ensure((const void *)conv3x3_narrow64_kernel, N3_LDS); if (rc != RN_OK) return rc; }
    hipLaunchKernelGGL(conv3x3_narrow64_kernel, dim3((unsigned)(N * a.strips * a.bands)), dim3(N3_THREADS), N3_LDS, (hipStream_t)stream, a);
    RN_LAUNCH_CHECK();
    return RN_OK;
}
