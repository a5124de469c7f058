// 3x3 / stride-1 / pad-1 convolution of the head towers on the packed level canvas as a hand-written MFMA implicit
// GEMM, with the bias + ReLU + position-mask epilogue fused (reference: the 3x3 conv + ReLU pairs
// of retinanet/layers.py:143-171 and :213-241).  Forward and data-gradient use this kernel (the data gradient is the
// same convolution with the taps reversed and the channel roles swapped); the weight gradient is the second kernel of
// this file (conv3x3_wgrad_kernel).
//
// Formulation.  The canvas carries a one-pixel zero border, [N][Hp][Wp][C] = [M][C] with M = N*Hp*Wp, so tap (r, s)
// of output position m is input position m + (r-1)*Wp + (s-1): no bounds logic in the loop.  Positions whose 3x3
// neighbourhood would leave their image are border / gap positions; their outputs are forced to zero by the mask,
// which is also what keeps the levels of the canvas from leaking into each other in the next layer.  Addresses are
// only clamped into the buffer.  GEMM: Y[m][n] = sum_{c, t} X[m + off_t][c] * W[n][t][c]; K = 9*Cin is walked as
// Cin/64 channel chunks x 9 taps (chunk outer: a tile's input lines stay in L2 across its taps).
//
// Kernel (bf16 in, f32 accumulate, bf16 out): one 256(m) x 256(n) output tile per workgroup, 8 waves of 128 x 64
// (4 x 2 v_mfma_f32_32x32x16_bf16 tiles, 128 accumulator registers); operands staged global -> LDS by 16-byte
// LDS-DMA (global_load_lds) into 128-byte rows whose 16-byte chunks are XOR-swizzled with (row >> 1) & 7 (source-side
// swizzle + the same XOR on the ds_read_b128: conflict-free for its 16-lane groups); 3 activation stages + 2 weight
// stages = 160 KiB of LDS, activations prefetched two K-tiles ahead, weights one.  The two waves of every SIMD run
// in PING-PONG: waves 0-3 and 4-7 are one barrier interval apart, so while one group issues its 16 MFMAs the other
// reads the fragments of its next 16 (12 ds_read_b128) and issues 4 LDS-DMA pieces; a K-tile is two (load, MFMA)
// phase pairs = 4 barriers.  LDS-DMA retirement: counted vmcnt at the end of the second load phase, one barrier
// before any wave of either group reads the tile; raw s_barrier (a __syncthreads would drain the DMA).
// Measured on MI355X, random data, [8,153,170,256] -> 256: 313 us = 785 TFLOP/s (939 without the 4th partial wave of
// tiles); MIOpen: forward 301-315 us + 32 us for the separate bias/ReLU/mask pass, data gradient 415 us.
#include "rn_common.hpp"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(3))) void *lds_void_ptr;

constexpr int CONV_BM = 256, CONV_BN = 256, CONV_BK = 64, CONV_THREADS = 512;
constexpr int CONV_LDS_BYTES = 5 * CONV_BM * CONV_BK * 2;        // 160 KiB
#define SWZ(row) (((row) >> 1) & 7)

constexpr int CONV_MAX_PROBLEMS = 4;
// Up to 4 convolutions of identical geometry in one launch (blockIdx.z = problem): the cls and the box tower run the
// same shapes side by side, and 2 x 813 tiles fill 7 waves of 256 workgroups where two launches take 2 x 4.
struct ConvArgs {
    const uint16_t *Xs[CONV_MAX_PROBLEMS];      // [M][Cin] bf16
    const uint16_t *Ws[CONV_MAX_PROBLEMS];      // [Cout][9][Cin] bf16
    const float *biases[CONV_MAX_PROBLEMS];     // [Cout] or null
    uint16_t *Ys[CONV_MAX_PROBLEMS];            // [M][Cout] bf16
    const uint8_t *mask;    // [HWp] or null (1 = keep)
    const uint16_t *zeros;  // dense mode: >= 256 B of zeros, the source of out-of-image taps
    int64_t M, HWp;
    int Cin, Cout, Wp, relu;
    int H;                  // 0: zero-bordered canvas (no bounds logic); > 0: dense [N][H][Wp] image, taps checked
    int taps;               // 9: 3x3 conv; 1: 1x1 conv = plain GEMM Y[M][Cout] = X[M][Cin] * W[Cout][Cin]^T
};
struct ConvProblem { const uint16_t *X, *W; const float *bias; uint16_t *Y; const uint8_t *mask; const uint16_t *zeros; int64_t M, HWp; int Cin, Cout, Wp, relu, H, taps; };

__device__ __forceinline__ uint16_t f2bf(const float f) { return (uint16_t)(rn::dt<RN_BF16>::pk(f, 0.0f) & 0xffffu); }

__global__ __launch_bounds__(CONV_THREADS) void conv3x3_canvas_kernel(const ConvArgs args)
{
    ConvProblem a;
    a.X = args.Xs[blockIdx.z]; a.W = args.Ws[blockIdx.z]; a.bias = args.biases[blockIdx.z]; a.Y = args.Ys[blockIdx.z];
    a.mask = args.mask; a.M = args.M; a.HWp = args.HWp; a.Cin = args.Cin; a.Cout = args.Cout; a.Wp = args.Wp; a.relu = args.relu;
    a.zeros = args.zeros; a.H = args.H; a.taps = args.taps;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];                 // [A0 A1 A2 | B0 B1] x 32 KiB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int64_t m0 = (int64_t)blockIdx.x * CONV_BM;
    const int n0 = blockIdx.y * CONV_BN;
    const int cpt = a.Cin / CONV_BK, KT = a.taps * cpt;
    constexpr int TILE = CONV_BM * CONV_BK * 2;
    unsigned char *const Abase = lds, *const Bbase = lds + 3 * TILE;

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    uint32_t a_off[4][4], b_off[2][4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        const int chunk = kk * 2 + (lane >> 5);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) { const int row = wm * 128 + mi * 32 + (lane & 31); a_off[mi][kk] = row * 128 + ((chunk ^ SWZ(row)) << 4); }
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) { const int row = wn * 64 + ni * 32 + (lane & 31); b_off[ni][kk] = row * 128 + ((chunk ^ SWZ(row)) << 4); }
    }
    const uint32_t lds_base = (uint32_t)(uintptr_t)lds;

    // dense mode: which of the 9 taps of this thread's 4 staging rows fall inside the image (bit t of tapmask[i])
    uint32_t tapmask[4] = {0x1ffu, 0x1ffu, 0x1ffu, 0x1ffu};
    if (a.H > 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t m = m0 + ((i * CONV_THREADS + tid) >> 3);
            uint32_t bits = 0;
            if (m < a.M) {
                const int pos = (int)(m % a.HWp), y = pos / a.Wp, x = pos - y * a.Wp;
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
                    bits |= (yy >= 0 && yy < a.H && xx >= 0 && xx < a.Wp) ? (1u << t) : 0u;
                }
            }
            tapmask[i] = bits;
        }
    }
    auto piece_a = [&](const int kt, const int i) {
        const int c0 = (kt / a.taps) * CONV_BK, t = kt % a.taps;
        const int off = a.taps == 1 ? 0 : (t / 3 - 1) * a.Wp + (t % 3 - 1);
        const int q = i * CONV_THREADS + tid, row = q >> 3, cp = q & 7;
        int64_t m = m0 + row + off;
        m = m < 0 ? 0 : (m >= a.M ? a.M - 1 : m);
        const uint16_t *g = a.X + m * a.Cin + c0 + ((cp ^ SWZ(row)) << 3);
        if (a.H > 0 && !((tapmask[i] >> t) & 1u)) g = a.zeros + ((cp ^ SWZ(row)) << 3);       // zero padding
        __builtin_amdgcn_global_load_lds((const void *)g, (lds_void_ptr)(Abase + (kt % 3) * TILE + q * 16), 16, 0, 0);
    };
    auto piece_b = [&](const int kt, const int i) {
        const int c0 = (kt / a.taps) * CONV_BK, t = kt % a.taps;
        const int q = i * CONV_THREADS + tid, row = q >> 3, cp = q & 7;
        const uint16_t *g = a.W + ((int64_t)(n0 + row) * a.taps + t) * a.Cin + c0 + ((cp ^ SWZ(row)) << 3);
        __builtin_amdgcn_global_load_lds((const void *)g, (lds_void_ptr)(Bbase + (kt & 1) * TILE + q * 16), 16, 0, 0);
    };
#pragma unroll
    for (int i = 0; i < 4; ++i) piece_a(0, i);
#pragma unroll
    for (int i = 0; i < 4; ++i) piece_b(0, i);
#pragma unroll
    for (int i = 0; i < 4; ++i) piece_a(1, i);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wm == 1) __builtin_amdgcn_s_barrier();                    // group 1 runs one barrier interval behind group 0

    bf16x8 fa[2][4], fb[2][2];
#define RN_DS_READ(dst, addr) asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr))
#define RN_LOAD_FRAGS(KK, SET)                                                     \
    RN_DS_READ(fb[SET][0], bbase + b_off[0][KK]); RN_DS_READ(fb[SET][1], bbase + b_off[1][KK]);   \
    RN_DS_READ(fa[SET][0], abase + a_off[0][KK]); RN_DS_READ(fa[SET][1], abase + a_off[1][KK]);   \
    RN_DS_READ(fa[SET][2], abase + a_off[2][KK]); RN_DS_READ(fa[SET][3], abase + a_off[3][KK]);
#define RN_MFMA8(SET)                                                              \
    _Pragma("unroll") for (int mi = 0; mi < 4; ++mi)                               \
        _Pragma("unroll") for (int ni = 0; ni < 2; ++ni)                           \
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[SET][mi], fb[SET][ni], acc[mi][ni], 0, 0, 0);
#define RN_MFMA_PHASE()                                                            \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0);          \
    __builtin_amdgcn_s_setprio(1);                                                 \
    RN_MFMA8(0) RN_MFMA8(1)                                                        \
    __builtin_amdgcn_s_setprio(0);                                                 \
    __builtin_amdgcn_sched_barrier(0);                                             \
    __builtin_amdgcn_s_barrier();

    for (int kt = 0; kt < KT; ++kt) {
        const uint32_t abase = lds_base + (uint32_t)((kt % 3) * TILE), bbase = lds_base + (uint32_t)(3 * TILE + (kt & 1) * TILE);
        // load phase 2kt: fragments of k-steps 0,1; the weight pieces of tile kt+1
        RN_LOAD_FRAGS(0, 0) RN_LOAD_FRAGS(1, 1)
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 1 < KT) {
#pragma unroll
            for (int i = 0; i < 4; ++i) piece_b(kt + 1, i);
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        RN_MFMA_PHASE()
        // load phase 2kt+1: fragments of k-steps 2,3; the activation pieces of tile kt+2; retire tile kt+1
        RN_LOAD_FRAGS(2, 0) RN_LOAD_FRAGS(3, 1)
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 2 < KT) {
#pragma unroll
            for (int i = 0; i < 4; ++i) piece_a(kt + 2, i);
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        RN_MFMA_PHASE()
    }
#undef RN_DS_READ
#undef RN_LOAD_FRAGS
#undef RN_MFMA8
#undef RN_MFMA_PHASE
    if (wm == 0) __builtin_amdgcn_s_barrier();                    // group 0 catches up with group 1's extra barrier

    __syncthreads();
    uint16_t *Ys = (uint16_t *)lds;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int col = wn * 64 + ni * 32 + (lane & 31);
            const float b = a.bias ? a.bias[n0 + col] : 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * 128 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                float v = acc[mi][ni][r] + b;
                if (a.relu & 1) v = v > 0.0f ? v : 0.0f;
                Ys[row * CONV_BN + col] = f2bf(v);
            }
        }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int q = i * CONV_THREADS + tid, row = q >> 5, piece = q & 31;
        const int64_t m = m0 + row;
        if (m < a.M) {
            uint4 v = *(const uint4 *)(Ys + row * CONV_BN + piece * 8);
            if (a.mask && !a.mask[m % a.HWp]) v = make_uint4(0, 0, 0, 0);
            *(uint4 *)(a.Y + m * a.Cout + n0 + piece * 8) = v;
        }
    }
}


// ================================================================================================================
// Weight gradient of the same convolution: dW[n][t][c] = sum_m G[m][n] * X[m + off_t][c]   (G = gradient at the conv
// output, zero on border / gap positions).  A GEMM whose contraction index is the POSITION m, so both operands are
// k-strided in memory ([m][channel] rows): the tiles are staged exactly as they lie in memory (64 positions x 256
// channels, 512-byte rows, LDS-DMA) and the MFMA fragments are read with the transposing ds_read_b64_tr_b16 (a 16-lane
// group reads a 4-position x 16-channel block and every lane receives one channel's 4 positions; two reads make the
// 8-deep k fragment).  16-byte chunks of a row are XOR-swizzled with (row & 3) << 2 (source side + read side), which
// spreads the 4 rows of a block over the banks: conflict-free per 32-lane half.
// Work split: one workgroup per (split of the positions, tap, problem); 256(n) x 256(c) fp32 partial tile per
// workgroup, 8 waves of 128 x 64 as in the forward kernel, same staging pipeline (3 G stages + 2 X stages, ping-pong
// wave groups, counted vmcnt).  Partials go to a workspace and a second kernel sums the splits into bf16.
constexpr int WG_POS = 64;                                       // positions per K-tile

struct WgradArgs {
    const uint16_t *Gs[CONV_MAX_PROBLEMS];      // [M][256] bf16
    const uint16_t *Xs[CONV_MAX_PROBLEMS];      // [M][256] bf16
    const uint16_t *zeros;
    float *partial;                             // [P][S][9][256][256] f32
    int64_t M;
    int Wp, S, tiles_per_split;                 // K-tiles (64 positions) per split
    int H;                                      // 0: zero-bordered canvas; > 0: dense [N][H][Wp] images, out-of-image taps read zeros
};

__global__ __launch_bounds__(CONV_THREADS) void conv3x3_wgrad_kernel(const WgradArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];                 // [G0 G1 G2 | X0 X1] x 32 KiB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int split = blockIdx.x, tap = blockIdx.y, prob = blockIdx.z;
    const uint16_t *__restrict__ G = a.Gs[prob], *__restrict__ X = a.Xs[prob];
    const int KT = a.tiles_per_split;
    const int64_t m_begin = (int64_t)split * KT * WG_POS;
    const int off = (tap / 3 - 1) * a.Wp + (tap % 3 - 1);
    constexpr int TILE = WG_POS * 512;                            // 32 KiB
    unsigned char *const Abase = lds, *const Bbase = lds + 3 * TILE;

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // transposed-read addresses: lane = 16*grp + 4*q + p;  grp & 1 selects the 16-channel half of a 32-channel
    // fragment, grp >> 1 the k half (positions +8); the lane supplies row q, channels 4p .. 4p+3 of its block.
    const int grp = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    uint32_t a_off[4], b_off[2];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        const int ch = wm * 16 + mi * 4 + 2 * (grp & 1) + (p >> 1);                  // 16-byte chunk of the channel row
        a_off[mi] = (uint32_t)((8 * (grp >> 1) + q) * 512 + ((ch ^ (q << 2)) << 4) + (p & 1) * 8);
    }
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const int ch = wn * 8 + ni * 4 + 2 * (grp & 1) + (p >> 1);
        b_off[ni] = (uint32_t)((8 * (grp >> 1) + q) * 512 + ((ch ^ (q << 2)) << 4) + (p & 1) * 8);
    }
    const uint32_t lds_base = (uint32_t)(uintptr_t)lds;

    // staging: tile row = position, 32 chunks of 16 B; LDS position (row, cp) holds global chunk cp ^ ((row & 3) << 2)
    const int dy_tap = tap / 3 - 1, dx_tap = tap % 3 - 1;
    const int64_t HW = (int64_t)a.H * a.Wp;
    auto piece = [&](const uint16_t *__restrict__ src, const int64_t shift, unsigned char *dst, const int kt, const int i) {
        const int qi = i * CONV_THREADS + tid, row = qi >> 5, cp = qi & 31;
        const int64_t m = m_begin + (int64_t)kt * WG_POS + row;
        int64_t ms = m + shift;
        ms = ms < 0 ? 0 : (ms >= a.M ? a.M - 1 : ms);
        const uint16_t *g = src + ms * 256 + ((cp ^ ((row & 3) << 2)) << 3);
        bool dead = m >= a.M;                                     // positions past the end contribute nothing
        if (a.H > 0 && shift != 0 && !dead) {                     // dense images: is the tap of this position inside its image?
            const int pos = (int)(m % HW), y = pos / a.Wp, x = pos - y * a.Wp;
            dead = (unsigned)(y + dy_tap) >= (unsigned)a.H || (unsigned)(x + dx_tap) >= (unsigned)a.Wp;
        }
        if (dead) g = a.zeros + ((cp & 15) << 3);
        __builtin_amdgcn_global_load_lds((const void *)g, (lds_void_ptr)(dst + qi * 16), 16, 0, 0);
    };
    auto piece_a = [&](const int kt, const int i) { piece(G, 0, Abase + (kt % 3) * TILE, kt, i); };
    auto piece_b = [&](const int kt, const int i) { piece(X, off, Bbase + (kt & 1) * TILE, kt, i); };
#pragma unroll
    for (int i = 0; i < 4; ++i) piece_a(0, i);
#pragma unroll
    for (int i = 0; i < 4; ++i) piece_b(0, i);
    if (KT > 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i) piece_a(1, i);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (wm == 1) __builtin_amdgcn_s_barrier();                    // ping-pong, as in the forward kernel

    unsigned long long fa[2][4][2], fb[2][2][2];                  // [k-step set][fragment][k half]
#define RN_TR_READ(dst, addr, OFF) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:" #OFF : "=v"(dst) : "v"(addr))
#define RN_LOAD_FRAGS(KOFF0, KOFF1, SET)                                             \
    RN_TR_READ(fb[SET][0][0], bbase + b_off[0], KOFF0); RN_TR_READ(fb[SET][0][1], bbase + b_off[0], KOFF1);   \
    RN_TR_READ(fb[SET][1][0], bbase + b_off[1], KOFF0); RN_TR_READ(fb[SET][1][1], bbase + b_off[1], KOFF1);   \
    RN_TR_READ(fa[SET][0][0], abase + a_off[0], KOFF0); RN_TR_READ(fa[SET][0][1], abase + a_off[0], KOFF1);   \
    RN_TR_READ(fa[SET][1][0], abase + a_off[1], KOFF0); RN_TR_READ(fa[SET][1][1], abase + a_off[1], KOFF1);   \
    RN_TR_READ(fa[SET][2][0], abase + a_off[2], KOFF0); RN_TR_READ(fa[SET][2][1], abase + a_off[2], KOFF1);   \
    RN_TR_READ(fa[SET][3][0], abase + a_off[3], KOFF0); RN_TR_READ(fa[SET][3][1], abase + a_off[3], KOFF1);
    struct U2 { unsigned long long lo, hi; };
#define RN_FRAG(v) __builtin_bit_cast(bf16x8, U2{v[0], v[1]})
#define RN_MFMA8(SET)                                                              \
    _Pragma("unroll") for (int mi = 0; mi < 4; ++mi)                               \
        _Pragma("unroll") for (int ni = 0; ni < 2; ++ni)                           \
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(RN_FRAG(fa[SET][mi]), RN_FRAG(fb[SET][ni]), acc[mi][ni], 0, 0, 0);
#define RN_MFMA_PHASE()                                                            \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0);          \
    __builtin_amdgcn_s_setprio(1);                                                 \
    RN_MFMA8(0) RN_MFMA8(1)                                                        \
    __builtin_amdgcn_s_setprio(0);                                                 \
    __builtin_amdgcn_sched_barrier(0);                                             \
    __builtin_amdgcn_s_barrier();

    for (int kt = 0; kt < KT; ++kt) {
        const uint32_t abase = lds_base + (uint32_t)((kt % 3) * TILE), bbase = lds_base + (uint32_t)(3 * TILE + (kt & 1) * TILE);
        // k-step kk covers positions 16kk .. 16kk+15 of the tile: byte offsets 16*kk*512 (+ 4*512 for the second k half)
        RN_LOAD_FRAGS(0, 2048, 0) RN_LOAD_FRAGS(8192, 10240, 1)
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 1 < KT) {
#pragma unroll
            for (int i = 0; i < 4; ++i) piece_b(kt + 1, i);
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        RN_MFMA_PHASE()
        RN_LOAD_FRAGS(16384, 18432, 0) RN_LOAD_FRAGS(24576, 26624, 1)
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 2 < KT) {
#pragma unroll
            for (int i = 0; i < 4; ++i) piece_a(kt + 2, i);
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        RN_MFMA_PHASE()
    }
#undef RN_TR_READ
#undef RN_LOAD_FRAGS
#undef RN_FRAG
#undef RN_MFMA8
#undef RN_MFMA_PHASE
    if (wm == 0) __builtin_amdgcn_s_barrier();

    // partial tile [n][c] f32 of this (problem, split, tap)
    float *__restrict__ out = a.partial + (((int64_t)prob * a.S + split) * 9 + tap) * 65536;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int col = wn * 64 + ni * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * 128 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                out[row * 256 + col] = acc[mi][ni][r];
            }
        }
}

// dW[p][n][t][c] (bf16) = sum over the splits of partial[p][s][t][n][c]
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float *__restrict__ partial, const int S, uint16_t *dw0, uint16_t *dw1,
                                                           uint16_t *dw2, uint16_t *dw3)
{
    const int prob = blockIdx.y;
    uint16_t *dw = prob == 0 ? dw0 : (prob == 1 ? dw1 : (prob == 2 ? dw2 : dw3));
    const int64_t i4 = (int64_t)blockIdx.x * 256 + threadIdx.x;          // over 9 * 256 * 256 / 4 float4 groups of [t][n][c]
    if (i4 >= 9 * 65536 / 4) return;
    rn::f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int sp = 0; sp < S; ++sp) {
        const rn::f32x4 v = ((const rn::f32x4 *)(partial + ((int64_t)prob * S + sp) * 9 * 65536))[i4];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    const int64_t e = i4 * 4;
    const int t = (int)(e / 65536), n = (int)((e % 65536) / 256), c = (int)(e % 256);
    rn::u32x2 o;
    o.x = rn::dt<RN_BF16>::pk(s.x, s.y); o.y = rn::dt<RN_BF16>::pk(s.z, s.w);
    *(rn::u32x2 *)(dw + ((int64_t)n * 9 + t) * 256 + c) = o;
}

}  // namespace

static int conv_launch(const void *const *xs, const void *const *ws, const float *const *biases, const uint8_t *mask,
                       void *const *ys, int P, int dtype, int64_t M, int64_t HWp, int Wp, int Cin, int Cout, int relu, int H,
                       const void *zeros, void *stream, int taps = 9)
{
    if (taps * (Cin / CONV_BK) < 2) return RN_EUNSUPPORTED;         // the pipeline keeps two K-tiles in flight
    if (!xs || !ws || !ys || P <= 0 || P > CONV_MAX_PROBLEMS || M <= 0 || HWp <= 0 || Wp <= 0 || Cin <= 0 || Cout <= 0) return RN_EINVAL;
    if (dtype != RN_BF16 || Cin % CONV_BK || Cout % CONV_BN) return RN_EUNSUPPORTED;
    if (H > 0 && (!zeros || !rn::aligned(zeros, 16))) return RN_EINVAL;
    ConvArgs a;
    for (int p = 0; p < CONV_MAX_PROBLEMS; ++p) {
        const int q = p < P ? p : 0;
        if (!xs[q] || !ws[q] || !ys[q]) return RN_EINVAL;
        if (!rn::aligned(xs[q], 16) || !rn::aligned(ws[q], 16) || !rn::aligned(ys[q], 16)) return RN_EALIGN;
        a.Xs[p] = (const uint16_t *)xs[q]; a.Ws[p] = (const uint16_t *)ws[q]; a.Ys[p] = (uint16_t *)ys[q];
        a.biases[p] = biases ? biases[q] : nullptr;
    }
    {   // 160 KiB of dynamic LDS needs the opt-in once per device (the attribute lives with the device's code object)
        static bool attr_set[64] = {};
        int dev = 0;
        RN_HIP(hipGetDevice(&dev));
        if (dev < 0 || dev >= 64 || !attr_set[dev]) {
            RN_HIP(hipFuncSetAttribute((const void *)conv3x3_canvas_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, CONV_LDS_BYTES));
            if (dev >= 0 && dev < 64) attr_set[dev] = true;
        }
    }
    a.mask = mask; a.M = M; a.HWp = HWp; a.Cin = Cin; a.Cout = Cout; a.Wp = Wp; a.relu = relu ? 1 : 0;
    a.H = H; a.zeros = (const uint16_t *)zeros; a.taps = taps;
    const dim3 grid((unsigned)((M + CONV_BM - 1) / CONV_BM), (unsigned)(Cout / CONV_BN), (unsigned)P);
    hipLaunchKernelGGL(conv3x3_canvas_kernel, grid, dim3(CONV_THREADS), CONV_LDS_BYTES, (hipStream_t)stream, a);
    RN_LAUNCH_CHECK();
    return RN_OK;
}

RN_API int rn_conv3x3_canvas_batched(const void *const *xs, const void *const *ws, const float *const *biases,
                                     const uint8_t *mask, void *const *ys, int P, int dtype, int64_t M, int64_t HWp, int Wp,
                                     int Cin, int Cout, int relu, void *stream)
{
    return conv_launch(xs, ws, biases, mask, ys, P, dtype, M, HWp, Wp, Cin, Cout, relu, 0, nullptr, stream);
}

RN_API int rn_conv3x3_nhwc(const void *x, const void *w, const float *bias, void *y, int dtype, int N, int H, int W, int Cin,
                           int Cout, int relu, const void *zeros, void *stream)
{
    if (!x || !w || !y || N <= 0 || H <= 0 || W <= 0) return RN_EINVAL;
    const void *xs[1] = {x}, *ws[1] = {w};
    const float *bs[1] = {bias};
    void *ys[1] = {y};
    return conv_launch(xs, ws, bs, nullptr, ys, 1, dtype, (int64_t)N * H * W, (int64_t)H * W, W, Cin, Cout, relu, H, zeros, stream);
}

RN_API int rn_conv3x3_canvas(const void *x, const void *w, const float *bias, const uint8_t *mask, void *y, int dtype,
                             int64_t M, int64_t HWp, int Wp, int Cin, int Cout, int relu, void *stream)
{
    if (!x || !w || !y) return RN_EINVAL;
    const void *xs[1] = {x}, *ws[1] = {w};
    const float *bs[1] = {bias};
    void *ys[1] = {y};
    return rn_conv3x3_canvas_batched(xs, ws, bs, mask, ys, 1, dtype, M, HWp, Wp, Cin, Cout, relu, stream);
}

RN_API int rn_conv1x1_nhwc(const void *x, const void *w, const float *bias, void *y, int dtype, int64_t M, int Cin, int Cout,
                           void *stream)
{
    if (!x || !w || !y || M <= 0) return RN_EINVAL;
    const void *xs[1] = {x}, *ws[1] = {w};
    const float *bs[1] = {bias};
    void *ys[1] = {y};
    return conv_launch(xs, ws, bs, nullptr, ys, 1, dtype, M, M, 1, Cin, Cout, 0, 0, nullptr, stream, 1);
}

RN_API size_t rn_conv3x3_wgrad_workspace_bytes(int P, int64_t M)
{
    if (P <= 0 || P > CONV_MAX_PROBLEMS || M <= 0) return 0;
    return (size_t)P * 64 * 9 * 65536 * sizeof(float);          // up to 64 splits of the positions
}

static int wgrad_launch(const void *const *gs, const void *const *xs, void *const *dws, int P, int dtype, int64_t M, int Wp, int H,
                        int Cin, int Cout, const void *zeros, void *workspace, size_t workspace_bytes, void *stream);

RN_API int rn_conv3x3_canvas_wgrad_batched(const void *const *gs, const void *const *xs, void *const *dws, int P, int dtype,
                                           int64_t M, int Wp, int Cin, int Cout, const void *zeros, void *workspace,
                                           size_t workspace_bytes, void *stream)
{
    return wgrad_launch(gs, xs, dws, P, dtype, M, Wp, 0, Cin, Cout, zeros, workspace, workspace_bytes, stream);
}

RN_API int rn_conv3x3_nhwc_wgrad(const void *g, const void *x, void *dw, int dtype, int N, int H, int W, int Cin, int Cout,
                                 const void *zeros, void *workspace, size_t workspace_bytes, void *stream)
{
    if (N <= 0 || H <= 0 || W <= 0) return RN_EINVAL;
    const void *gs[1] = {g}, *xs[1] = {x};
    void *dws[1] = {dw};
    return wgrad_launch(gs, xs, dws, 1, dtype, (int64_t)N * H * W, W, H, Cin, Cout, zeros, workspace, workspace_bytes, stream);
}

static int wgrad_launch(const void *const *gs, const void *const *xs, void *const *dws, int P, int dtype, int64_t M, int Wp, int H,
                        int Cin, int Cout, const void *zeros, void *workspace, size_t workspace_bytes, void *stream)
{
    if (!gs || !xs || !dws || !zeros || !workspace || P <= 0 || P > CONV_MAX_PROBLEMS || M <= 0 || Wp <= 0) return RN_EINVAL;
    if (dtype != RN_BF16 || Cin != 256 || Cout != 256) return RN_EUNSUPPORTED;
    if (workspace_bytes < rn_conv3x3_wgrad_workspace_bytes(P, M)) return RN_EWORKSPACE;
    WgradArgs a;
    uint16_t *dw[CONV_MAX_PROBLEMS] = {nullptr, nullptr, nullptr, nullptr};
    for (int p = 0; p < CONV_MAX_PROBLEMS; ++p) {
        const int q = p < P ? p : 0;
        if (!gs[q] || !xs[q] || !dws[q]) return RN_EINVAL;
        if (!rn::aligned(gs[q], 16) || !rn::aligned(xs[q], 16) || !rn::aligned(dws[q], 16)) return RN_EALIGN;
        a.Gs[p] = (const uint16_t *)gs[q]; a.Xs[p] = (const uint16_t *)xs[q];
        dw[p] = (uint16_t *)dws[q];
    }
    int dev = 0, cus = 0;
    RN_HIP(hipGetDevice(&dev));
    RN_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    {
        static bool attr_set[64] = {};
        if (dev < 0 || dev >= 64 || !attr_set[dev]) {
            RN_HIP(hipFuncSetAttribute((const void *)conv3x3_wgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, CONV_LDS_BYTES));
            if (dev >= 0 && dev < 64) attr_set[dev] = true;
        }
    }
    // one workgroup per (split, tap, problem): choose the split count so that the grid is about one wave of the chip
    int S = cus / (9 * P);
    if (S < 1) S = 1;
    if (S > 64) S = 64;
    const int64_t ktiles = (M + WG_POS - 1) / WG_POS;
    a.tiles_per_split = (int)((ktiles + S - 1) / S);
    S = (int)((ktiles + a.tiles_per_split - 1) / a.tiles_per_split);
    a.S = S; a.M = M; a.Wp = Wp; a.H = H; a.zeros = (const uint16_t *)zeros; a.partial = (float *)workspace;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(conv3x3_wgrad_kernel, dim3((unsigned)S, 9, (unsigned)P), dim3(CONV_THREADS), CONV_LDS_BYTES, st, a);
    RN_LAUNCH_CHECK();
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(9 * 65536 / 4 / 256, (unsigned)P), dim3(256), 0, st, (const float *)workspace, S, dw[0], dw[1], dw[2], dw[3]);
    RN_LAUNCH_CHECK();
    return RN_OK;
}
