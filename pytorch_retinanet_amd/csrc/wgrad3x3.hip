// Weight gradient of the NARROW 3x3 / stride-1 / pad-1 convolutions of the ResNet trunk -- conv2 of the layer1 / layer2 / layer4
// bottlenecks (reference: retinanet/backbone.py:112,128; 64 / 128 / 512 channels) -- for bf16 channels-last tensors.
//
//   dW[n][t][c] = sum over output pixels m of G[m][n] * X[m + off_t][c]
//
// The contraction index is the pixel, so both MFMA operands are pixel-strided in memory.  MIOpen runs one workgroup per tap: each
// re-reads G and X (32 flop per byte of L2 traffic at 64 channels -> 320 TFLOP/s, 123 us + a zero fill + a cast per layer1 block);
// the 256-channel kernel of conv.hip does the same and can afford it.  Here ONE team of four waves holds all nine taps of a
// 64 (n) x 64 (c) block of dW in registers (36 tiles of 16 x 16 per wave = 144 accumulator registers): a stage is 64 consecutive
// output pixels of one image row -- their 64 x 64 gradient rows and, per kernel row, ONE strip of 66 input pixels (x0 - 1 .. x0 + 64)
// -- staged as they lie in memory (LDS-DMA, 16 bytes per lane) and read with the transposing ds_read_b64_tr_b16; the three horizontal taps are the same strip
// read one row further down.  Pixels outside the image are staged as zeros, so nothing is masked later.  Per 32-pixel k-step a
// wave issues 13 fragment reads for 36 MFMAs (wave = 16-channel column block of X, all four 16-row blocks of G).
// Wider layers are cut into 64 x 64 sub-problems (blockIdx.x / wgs_per_sub): 4 at 128 channels, 64 at 512.
// A workgroup is TWO teams (8 waves, two per SIMD) walking alternate stages with their own double buffers; at the end team 1
// hands its accumulators to team 0 through LDS and one f32 partial [64][9][64] per workgroup goes to the workspace (256 workgroups:
// 37.7 MB); a second kernel sums the partials into the bf16 gradient.
#include "rn_common.hpp"

namespace {

typedef __attribute__((ext_vector_type(4))) float f32x4v;

constexpr int W3_THREADS = 512, W3_TEAM = 256;
constexpr int W3_PX = 64;                          // output pixels per stage (two 32-deep k-steps)
constexpr int W3_SROWS = 66;                       // staged pixels per strip
constexpr int W3_GBYTES = W3_PX * 128, W3_SBYTES = 72 * 128;      // (strips padded to 72 rows: the k-step-1 reads of dx = +1 stay inside)
constexpr int W3_BUF = W3_GBYTES + 3 * W3_SBYTES + 1024;           // 36 KiB per stage (35 DMA units + a dummy one: every wave issues 9)
constexpr int W3_ACC_BYTES = 4 * 36 * 4 * 64 * 4;                  // one team's accumulators: 147 456 B
constexpr int W3_LDS = W3_ACC_BYTES > 4 * W3_BUF ? W3_ACC_BYTES : 4 * W3_BUF;
constexpr int W3_OUT = 64 * 9 * 64;                                // floats per partial

// 16-byte chunk slot of a 128-byte LDS row: rows {0..3, 8..11} (+ any shift, + 16, + 32) of one 32-byte column block -- what the
// 32 lanes of half a transposing read touch -- land on 8 different bank groups
__device__ __forceinline__ int w3_swz(const int row) { return (((row >> 1) & 1) | (((row >> 3) & 1) << 1)) << 1; }

struct W3Args {
    const uint16_t *g;      // [N][H][W][Cn] bf16: gradient at the conv output
    const uint16_t *x;      // [N][H][W][Cc] bf16: the conv input
    const uint16_t *zeros;  // >= 128 bytes of zeros (rows outside the image)
    float *partial;         // [gridDim.x][64][9][64]
    int N, H, W, Cn, Cc;
    int tiles_x, tiles, subs_c, wgs_per_sub, tiles_per_team;
};

template <int DT>
__global__ __launch_bounds__(W3_THREADS) void wgrad3x3_kernel(const W3Args a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    typedef __attribute__((address_space(3))) void *lds_void_ptr;
    const int tid = threadIdx.x & (W3_TEAM - 1), lane = tid & 63;
    const int team = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8);
    const int ct = __builtin_amdgcn_readfirstlane(tid >> 6);         // this wave's 16-channel column block of X
    const int grp = lane >> 4, q = (lane >> 2) & 3, p4 = lane & 3;   // transposing read: group grp reads pixels 8 grp + q (+ 4), columns 4 p4 ..
    const int sub = (int)blockIdx.x / a.wgs_per_sub, wsub = (int)blockIdx.x - sub * a.wgs_per_sub;
    const int n0 = (sub / a.subs_c) * 64, c0 = (sub % a.subs_c) * 64;
    const int t_beg = (wsub * 2 + team) * a.tiles_per_team, t_end = min(t_beg + a.tiles_per_team, a.tiles);
    unsigned char *const mybuf = lds + team * 2 * W3_BUF;

    f32x4v acc[4][9];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[i][t] = f32x4v{0.0f, 0.0f, 0.0f, 0.0f};

    // staging by LDS-DMA, 16 bytes per lane: a unit = one wave-wide DMA = 8 rows of 128 bytes; units 0 .. 7 are the gradient rows,
    // 8 + 9 s + k rows 8k .. 8k+7 of strip s -- the stage buffer is just 35 consecutive KiB.  Lane (row = lane >> 3, slot = lane & 7)
    // fetches the chunk that belongs in its slot (swizzle on the source side); rows outside the image fetch the zero page.
    // Addresses: a wave-uniform 64-bit base per unit (SALU) + a per-lane 32-bit offset that only depends on the unit's parity.
    const int lrow = lane >> 3, slot = lane & 7;
    uint32_t goff[2], xoff[2];                                     // bytes; [k & 1]: row = 8 k + lrow -> swizzle bits (lrow >> 1) & 1, k & 1
#pragma unroll
    for (int kp = 0; kp < 2; ++kp) {
        const int chunk = slot ^ ((((lrow >> 1) & 1) | (kp << 1)) << 1);
        goff[kp] = (uint32_t)((lrow * a.Cn + chunk * 8) * 2);
        xoff[kp] = (uint32_t)((lrow * a.Cc + chunk * 8) * 2);
    }
    const unsigned char *const zsrc = (const unsigned char *)a.zeros + slot * 16;
    auto stage = [&](const int tile, unsigned char *buf) {
        const int tx = tile % a.tiles_x, rowid = tile / a.tiles_x, y = rowid % a.H, b = rowid / a.H, x0 = tx * W3_PX;
        const int64_t rowbase = ((int64_t)b * a.H + y) * a.W;
        const unsigned char *const gbase = (const unsigned char *)(a.g + (rowbase + x0) * a.Cn + n0);
        const unsigned char *const xbase = (const unsigned char *)(a.x + (rowbase + x0 - 1) * a.Cc + c0);     // strip row 0 of kernel row 1
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const int u = i * 4 + ct;                              // wave-uniform; unit 35 is the dummy (zeros into the spare KiB)
            const int s = u < 8 ? -1 : (u - 8) / 9, k = u < 8 ? u : (u - 8) - 9 * s, row = 8 * k + lrow;
            const unsigned char *src;
            if (u < 8) {
                src = (x0 + row < a.W) ? gbase + (int64_t)(8 * k) * a.Cn * 2 + goff[k & 1] : zsrc;
            } else {
                const int yy = y + s - 1, xx = x0 - 1 + row;
                const bool ok = u < 35 && row < W3_SROWS && (unsigned)yy < (unsigned)a.H && (unsigned)xx < (unsigned)a.W;
                src = ok ? xbase + ((int64_t)(s - 1) * a.W + 8 * k) * a.Cc * 2 + xoff[k & 1] : zsrc;
            }
            __builtin_amdgcn_global_load_lds((const void *)src, (lds_void_ptr)(buf + u * 1024), 16, 0, 0);
        }
    };
    // per-lane byte offsets of the fragment reads of k-step 0 (k-step 1: + 32 rows, the swizzle does not see bit 5)
    const int prow = 8 * grp + q;
    uint32_t g_off[4], x_lo[3], x_hi[3];
#pragma unroll
    for (int i = 0; i < 4; ++i) g_off[i] = (uint32_t)(prow * 128 + (((2 * i + (p4 >> 1)) ^ w3_swz(prow)) << 4) + (p4 & 1) * 8);      // (+ 4 rows: same swizzle)
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const int r0 = prow + d, r1 = prow + d + 4;               // strip row of pixel prow for dx = d - 1 (strip row 0 = pixel x0 - 1)
        x_lo[d] = (uint32_t)(r0 * 128 + (((2 * ct + (p4 >> 1)) ^ w3_swz(r0)) << 4) + (p4 & 1) * 8);
        x_hi[d] = (uint32_t)(r1 * 128 + (((2 * ct + (p4 >> 1)) ^ w3_swz(r1)) << 4) + (p4 & 1) * 8);
    }
    const uint32_t lds_base = (uint32_t)(uintptr_t)lds + (uint32_t)(team * 2 * W3_BUF);

    // The fragment reads are inline asm: the compiler orders every LDS read it knows of behind the LDS-DMA in flight (vmcnt(0)),
    // which would serialise the next stage's fetch with this stage's MFMAs.  The waits are therefore counted by hand: a tap's
    // two reads are issued one tap ahead of the four MFMAs that consume them.
    struct U2 { unsigned long long lo, hi; };
#define W3_TR(dst, addr) asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(dst) : "v"(addr))
#define W3_FRAG(v) __builtin_bit_cast(typename rn::mma<DT>::frag, U2{v[0], v[1]})
#define W3_WAIT(N) asm volatile("s_waitcnt lgkmcnt(" #N ")" ::: "memory"); __builtin_amdgcn_sched_barrier(0);

    // Phases: the teams alternate -- in phase ph team (ph & 1) runs the MFMAs of its tile ph >> 1 while the other team issues the
    // LDS-DMA of a later tile (DMA issue costs a wave about as long as the MFMAs of a stage: in lockstep both teams would queue at
    // the MFMA pipe, then both at the address pipe).  A team's DMA runs two tiles ahead of its MFMAs, so a stage has a whole
    // iteration to land: after issuing 9 pieces a wave waits for everything but those 9.
    auto compute = [&](const int tile, const int bsel) {
        const int x0 = (tile % a.tiles_x) * W3_PX;
        const int ksteps = (x0 + 32 < a.W) ? 2 : 1;               // a row's last stage may hold 32 pixels or fewer
        for (int ks = 0; ks < ksteps; ++ks) {
            const uint32_t gb = lds_base + (uint32_t)(bsel * W3_BUF + ks * 32 * 128), sb = gb + W3_GBYTES;
            unsigned long long gf[4][2], xf[4][2];
#define W3_XREAD(T) { const uint32_t s_ = sb + (uint32_t)(((T) / 3) * W3_SBYTES); W3_TR(xf[(T) & 3][0], s_ + x_lo[(T) % 3]); W3_TR(xf[(T) & 3][1], s_ + x_hi[(T) % 3]); }
#pragma unroll
            for (int i = 0; i < 4; ++i) { W3_TR(gf[i][0], gb + g_off[i]); W3_TR(gf[i][1], gb + g_off[i] + 4 * 128); }
            W3_XREAD(0) W3_XREAD(1) W3_XREAD(2)
#pragma unroll
            for (int t = 0; t < 9; ++t) {                          // tap t's reads were issued three taps (12 MFMAs) ago
                if (t + 3 < 9) { W3_XREAD(t + 3) W3_WAIT(6) }
                else if (t + 2 < 9) { W3_WAIT(4) }
                else if (t + 1 < 9) { W3_WAIT(2) }
                else { W3_WAIT(0) }
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc[i][t] = rn::mma<DT>::m16(W3_FRAG(gf[i]), W3_FRAG(xf[t & 3]), acc[i][t]);
                __builtin_amdgcn_sched_barrier(0);
            }
#undef W3_XREAD
        }
    };
    if (t_beg < t_end) stage(t_beg, mybuf);
    if (t_beg + 1 < t_end) {
        stage(t_beg + 1, mybuf + W3_BUF);
        asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    for (int ph = 0; ph < 2 * a.tiles_per_team; ++ph) {           // (both teams make the same number of trips: the barriers are the workgroup's)
        const int i = ph >> 1;
        if ((ph & 1) == team) {
            if (t_beg + i < t_end) compute(t_beg + i, i & 1);
        } else {
            const int j = team == 0 ? i + 2 : i + 1;               // team 0 refills the buffer it has just used; team 1 the one it used a phase ago
            if (team == 0 || i > 0) {
                if (t_beg + j < t_end) {
                    stage(t_beg + j, mybuf + (j & 1) * W3_BUF);
                    asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
                } else {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
    }
#undef W3_TR
#undef W3_FRAG
#undef W3_WAIT

    // team 1 -> LDS -> team 0 (lane-contiguous: conflict-free), then one partial per workgroup.
    // D lane: column (c) = lane & 15, rows (n) 4 (lane >> 4) + j
    float *xch = (float *)lds;
    if (team == 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) xch[(((ct * 4 + i) * 9 + t) * 4 + j) * 64 + lane] = acc[i][t][j];
    }
    __syncthreads();
    if (team == 0) {
        float *out = a.partial + (int64_t)blockIdx.x * W3_OUT;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int n = i * 16 + 4 * (lane >> 4) + j, c = ct * 16 + (lane & 15);
                    out[(n * 9 + t) * 64 + c] = acc[i][t][j] + xch[(((ct * 4 + i) * 9 + t) * 4 + j) * 64 + lane];
                }
    }
}

// dw[(n0 + n)][t][c0 + c] (bf16, [Cn][3][3][Cc]) = sum over the sub-problem's workgroups of partial[..][n][t][c].
// Block = 32 float4 columns x 8 slices of the partials, combined through LDS in a fixed order.
template <int DT>
__global__ __launch_bounds__(256) void wgrad3x3_reduce_kernel(const float *__restrict__ partial, const int wgs_per_sub, const int subs_c,
                                                              const int Cc, uint16_t *__restrict__ dw)
{
    __shared__ rn::f32x4 sh[8][32];
    const int sub = blockIdx.y, j = threadIdx.x & 31, slice = threadIdx.x >> 5;
    const int i4 = blockIdx.x * 32 + j;                              // over 64 * 9 * 64 / 4 float4 groups of [n][t][c]
    const rn::f32x4 *src = (const rn::f32x4 *)(partial + (int64_t)sub * wgs_per_sub * W3_OUT) + i4;
    rn::f32x4 s = {0.f, 0.f, 0.f, 0.f};
    int w = slice;
    for (; w + 8 < wgs_per_sub; w += 16) {
        const rn::f32x4 v0 = src[(int64_t)w * (W3_OUT / 4)], v1 = src[(int64_t)(w + 8) * (W3_OUT / 4)];
        s.x += v0.x; s.y += v0.y; s.z += v0.z; s.w += v0.w;
        s.x += v1.x; s.y += v1.y; s.z += v1.z; s.w += v1.w;
    }
    for (; w < wgs_per_sub; w += 8) {
        const rn::f32x4 v = src[(int64_t)w * (W3_OUT / 4)];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    sh[slice][j] = s;
    __syncthreads();
    if (slice == 0) {
        rn::f32x4 t = sh[0][j];
#pragma unroll
        for (int l = 1; l < 8; ++l) { const rn::f32x4 v = sh[l][j]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
        const int e = i4 * 4, c = e & 63, tp = (e >> 6) % 9, n = e / (9 * 64);
        const int n0 = (sub / subs_c) * 64, c0 = (sub % subs_c) * 64;
        rn::u32x2 o;
        o.x = rn::dt<DT>::pk(t.x, t.y); o.y = rn::dt<DT>::pk(t.z, t.w);
        *(rn::u32x2 *)(dw + ((int64_t)(n0 + n) * 9 + tp) * Cc + c0 + c) = o;
    }
}

int w3_workgroups(const int subs)
{
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    const int per_sub = cus / subs;
    return per_sub < 1 ? 1 : per_sub;
}

}  // namespace

RN_API size_t rn_conv3x3_wgrad_narrow_workspace_bytes(int Cout, int Cin)
{
    if (Cout <= 0 || Cin <= 0 || Cout % 64 || Cin % 64) return 0;
    const int subs = (Cout / 64) * (Cin / 64);
    return (size_t)subs * w3_workgroups(subs) * W3_OUT * sizeof(float);
}

RN_API int rn_conv3x3_wgrad_narrow(const void *g, const void *x, void *dw, int dtype, int N, int H, int W, int Cout, int Cin,
                                   const void *zero_page, void *workspace, size_t workspace_bytes, void *stream)
{
    if (!g || !x || !dw || !workspace || !zero_page || N <= 0 || H <= 0 || W <= 0 || Cout <= 0 || Cin <= 0) return RN_EINVAL;
    if ((dtype != RN_BF16 && dtype != RN_F16) || Cout % 64 || Cin % 64) return RN_EUNSUPPORTED;
    const int subs = (Cout / 64) * (Cin / 64);
    if (subs > 4096 || (int64_t)N * H * ((W + W3_PX - 1) / W3_PX) >= ((int64_t)1 << 30)) return RN_EUNSUPPORTED;
    if (workspace_bytes < rn_conv3x3_wgrad_narrow_workspace_bytes(Cout, Cin)) return RN_EWORKSPACE;
    if (!rn::aligned(g, 16) || !rn::aligned(x, 16) || !rn::aligned(dw, 8) || !rn::aligned(workspace, 16) || !rn::aligned(zero_page, 16)) return RN_EALIGN;
    W3Args a;
    a.g = (const uint16_t *)g; a.x = (const uint16_t *)x; a.zeros = (const uint16_t *)zero_page; a.partial = (float *)workspace;
    a.N = N; a.H = H; a.W = W; a.Cn = Cout; a.Cc = Cin;
    a.tiles_x = (W + W3_PX - 1) / W3_PX; a.tiles = N * H * a.tiles_x; a.subs_c = Cin / 64;
    a.wgs_per_sub = w3_workgroups(subs);
    a.tiles_per_team = (a.tiles + 2 * a.wgs_per_sub - 1) / (2 * a.wgs_per_sub);
    hipStream_t st = (hipStream_t)stream;
#define W3_LAUNCH(DT)                                                                                                                  \
    {   static rn::DynLdsOptIn opt_in = {};                                                                                            \
        const int rc = opt_in.ensure((const void *)wgrad3x3_kernel<DT>, W3_LDS);                                                       \
        if (rc != RN_OK) return rc;                                                                                                    \
        hipLaunchKernelGGL(wgrad3x3_kernel<DT>, dim3((unsigned)(subs * a.wgs_per_sub)), dim3(W3_THREADS), W3_LDS, st, a);              \
        RN_LAUNCH_CHECK();                                                                                                             \
        hipLaunchKernelGGL(wgrad3x3_reduce_kernel<DT>, dim3(W3_OUT / 4 / 32, (unsigned)subs), dim3(256), 0, st, (const float *)workspace, \
                           a.wgs_per_sub, a.subs_c, Cin, (uint16_t *)dw);                                                              \
        RN_LAUNCH_CHECK(); }
    if (dtype == RN_F16) W3_LAUNCH(RN_F16) else W3_LAUNCH(RN_BF16)
#undef W3_LAUNCH
    return RN_OK;
}
