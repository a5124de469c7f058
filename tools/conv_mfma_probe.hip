// Exploratory: hand-written MFMA implicit-GEMM for the 3x3 / stride-1 / pad-1 convolutions of the RetinaNet head
// towers on the packed level canvas ("only if rocprof shows it beating MIOpen").
// Stand-alone probe: correctness against a CPU reference on a small problem, then timing at the R50 canvas shape
// ([8, 153, 170, 256] padded canvas, 256 -> 256 channels), where MIOpen measures 361 us forward (660 TFLOP/s).
//   build: hipcc -O3 --offload-arch=gfx950 tools/conv_mfma_probe.hip -o tools/conv_mfma_probe.bin
//
// Formulation.  The canvas is stored with a one-pixel zero border, [N][Hp][Wp][C] = [M][C] with M = N*Hp*Wp, so
// tap (r, s) of output position m is input position m + (r-1)*Wp + (s-1): no bounds logic (positions whose
// neighbourhood would leave their image are border / gap positions, whose outputs are masked to zero anyway;
// addresses are only clamped into the buffer).  GEMM: Y[m][n] = sum_{t, c} X[m + off_t][c] * W[n][t][c],
// K = 9*Cin walked as 9 taps x Cin/64 chunks.  Epilogue: + bias, ReLU, position mask, bf16.
//
// Tile 128(m) x 128(n) x 64(k), 4 waves of 64x64 (2x2 MFMA 32x32x16 bf16), operands staged global -> LDS with
// 16-byte LDS-DMA (global_load_lds), two LDS buffers, XOR-swizzled 128-byte rows (source-side swizzle + the same
// XOR on the ds_read_b128), one barrier per K-tile.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(3))) void *lds_void_ptr;

constexpr int BM = 128, BN = 128, BK = 64, THREADS = 256;
// 16-byte chunk swizzle of the 128-byte LDS rows.  ds_read_b128 is serviced in 16-lane groups ({0-3,12-15,20-27}, ...):
// with the row parity already selecting the half of the 256-byte bank row, (row >> 1) & 7 gives the 8 rows of each
// parity in a group 8 different chunk slots -> conflict-free (row & 7 is 2-way: rows r and r + 8k share a slot).
#define SWZ(row) (((row) >> 1) & 7)

struct ConvArgs {
    const uint16_t *X;      // [M][Cin] bf16
    const uint16_t *W;      // [Cout][9][Cin] bf16
    const float *bias;      // [Cout] or null
    const uint8_t *mask;    // [HWp] or null (1 = keep)
    uint16_t *Y;            // [M][Cout] bf16
    int64_t M, HWp;
    int Cin, Cout, Wp, relu;
};

__device__ __forceinline__ uint16_t f2bf(float f)
{
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

__global__ __launch_bounds__(THREADS) void conv3x3_mfma_kernel(const ConvArgs a)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[2][2][BM * BK * 2];      // [buffer][A | B][16 KiB]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int64_t m0 = (int64_t)blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;
    const int cpt = a.Cin / BK, KT = 9 * cpt;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    auto stage = [&](const int kt, const int buf) {
        const int c0 = (kt / 9) * BK, t = kt % 9;      // channel chunk outer, tap inner: a tile's input lines stay in L2 across its 9 taps
        const int off = (t / 3 - 1) * a.Wp + (t % 3 - 1);
#pragma unroll
        for (int i = 0; i < 4; ++i) {                            // A: 128 rows x 8 chunks of 16 B
            const int q = i * THREADS + tid, row = q >> 3, cp = q & 7;
            int64_t m = m0 + row + off;
            m = m < 0 ? 0 : (m >= a.M ? a.M - 1 : m);
            const uint16_t *g = a.X + m * a.Cin + c0 + ((cp ^ SWZ(row)) << 3);
            __builtin_amdgcn_global_load_lds((const void *)g, (lds_void_ptr)(&lds[buf][0][q * 16]), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {                            // B: 128 output channels x 8 chunks
            const int q = i * THREADS + tid, row = q >> 3, cp = q & 7;
            const uint16_t *g = a.W + ((int64_t)(n0 + row) * 9 + t) * a.Cin + c0 + ((cp ^ SWZ(row)) << 3);
            __builtin_amdgcn_global_load_lds((const void *)g, (lds_void_ptr)(&lds[buf][1][q * 16]), 16, 0, 0);
        }
    };

    stage(0, 0);
    for (int kt = 0; kt < KT; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 1 < KT) stage(kt + 1, (kt + 1) & 1);
        const unsigned char *As = lds[kt & 1][0], *Bs = lds[kt & 1][1];
#pragma unroll
        for (int kk = 0; kk < BK / 16; ++kk) {
            bf16x8 af[2], bfr[2];
            const int chunk = kk * 2 + (lane >> 5);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                const int row = wm * 64 + mi * 32 + (lane & 31);
                af[mi] = *(const bf16x8 *)(As + row * 128 + ((chunk ^ SWZ(row)) << 4));
            }
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                const int row = wn * 64 + ni * 32 + (lane & 31);
                bfr[ni] = *(const bf16x8 *)(Bs + row * 128 + ((chunk ^ SWZ(row)) << 4));
            }
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mi], bfr[ni], acc[mi][ni], 0, 0, 0);
        }
    }

    // epilogue: bias + ReLU + mask, through LDS so that the global stores are 16-byte row pieces
    __syncthreads();
    uint16_t *Ys = (uint16_t *)&lds[0][0][0];                     // [128][128] bf16 = 32 KiB
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int col = wn * 64 + ni * 32 + (lane & 31);
            const float b = a.bias ? a.bias[n0 + col] : 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                float v = acc[mi][ni][r] + b;
                if (a.relu & 1) v = v > 0.0f ? v : 0.0f;
                Ys[row * BN + col] = f2bf(v);
            }
        }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) {                                 // 128 rows x 16 pieces of 16 B
        const int q = i * THREADS + tid, row = q >> 4, piece = q & 15;
        const int64_t m = m0 + row;
        if (m < a.M) {
            uint4 v = *(const uint4 *)(Ys + row * BN + piece * 8);
            if (a.mask && !a.mask[m % a.HWp]) v = make_uint4(0, 0, 0, 0);
            *(uint4 *)(a.Y + m * a.Cout + n0 + piece * 8) = v;
        }
    }
}

// ---- v2: 256 x 256 x 64 tile, 8 waves of 128 x 64 (4 x 2 MFMA tiles), all output channels of a 256-channel layer in
// one block (the activation tile is fetched once), 128 KiB of LDS = two stages, one block per CU.
constexpr int BM2 = 256, BN2 = 256, THREADS2 = 512;
__global__ __launch_bounds__(THREADS2) void conv3x3_mfma_v2_kernel(const ConvArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds2[];                 // [2 stages][A 32 KiB | B 32 KiB]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int64_t m0 = (int64_t)blockIdx.x * BM2;
    const int n0 = blockIdx.y * BN2;
    const int cpt = a.Cin / BK, KT = 9 * cpt;
    constexpr int STAGE = (BM2 + BN2) * BK * 2, AOFF = 0, BOFF = BM2 * BK * 2;

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    auto stage = [&](const int kt, const int buf) {
        const int c0 = (kt / 9) * BK, t = kt % 9;      // channel chunk outer, tap inner: a tile's input lines stay in L2 across its 9 taps
        const int off = (t / 3 - 1) * a.Wp + (t % 3 - 1);
        unsigned char *base = lds2 + buf * STAGE;
#pragma unroll
        for (int i = 0; i < 4; ++i) {                            // A: 256 rows x 8 chunks of 16 B
            const int q = i * THREADS2 + tid, row = q >> 3, cp = q & 7;
            int64_t m = m0 + row + off;
            m = m < 0 ? 0 : (m >= a.M ? a.M - 1 : m);
            const uint16_t *g = a.X + m * a.Cin + c0 + ((cp ^ SWZ(row)) << 3);
            __builtin_amdgcn_global_load_lds((const void *)g, (lds_void_ptr)(base + AOFF + q * 16), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {                            // B: 256 output channels x 8 chunks
            const int q = i * THREADS2 + tid, row = q >> 3, cp = q & 7;
            const uint16_t *g = a.W + ((int64_t)(n0 + row) * 9 + t) * a.Cin + c0 + ((cp ^ SWZ(row)) << 3);
            __builtin_amdgcn_global_load_lds((const void *)g, (lds_void_ptr)(base + BOFF + q * 16), 16, 0, 0);
        }
    };

    stage(0, 0);
    for (int kt = 0; kt < KT; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 1 < KT && !(a.relu & 256)) stage(kt + 1, (kt + 1) & 1);
        const unsigned char *As = lds2 + (kt & 1) * STAGE + AOFF, *Bs = lds2 + (kt & 1) * STAGE + BOFF;
        // fragments of k-step kk+1 are read (into their own registers) before the MFMAs of k-step kk are issued, so the
        // LDS latency hides behind 8 MFMAs instead of being waited for in front of every pair of them
        bf16x8 af[2][4], bfr[2][2];
        auto load_frags = [&](const int kk, bf16x8 (&fa)[4], bf16x8 (&fb)[2]) {
            const int chunk = kk * 2 + (lane >> 5);
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                const int row = wn * 64 + ni * 32 + (lane & 31);
                fb[ni] = *(const bf16x8 *)(Bs + row * 128 + ((chunk ^ SWZ(row)) << 4));
            }
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                const int row = wm * 128 + mi * 32 + (lane & 31);
                fa[mi] = *(const bf16x8 *)(As + row * 128 + ((chunk ^ SWZ(row)) << 4));
            }
        };
        load_frags(0, af[0], bfr[0]);
#pragma unroll
        for (int kk = 0; kk < BK / 16; ++kk) {
            if (kk + 1 < BK / 16) load_frags(kk + 1, af[(kk + 1) & 1], bfr[(kk + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[kk & 1][mi], bfr[kk & 1][ni], acc[mi][ni], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    __syncthreads();
    uint16_t *Ys = (uint16_t *)lds2;                               // [256][256] bf16 = 128 KiB
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
                Ys[row * BN2 + col] = f2bf(v);
            }
        }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {                                // 256 rows x 32 pieces of 16 B
        const int q = i * THREADS2 + tid, row = q >> 5, piece = q & 31;
        const int64_t m = m0 + row;
        if (m < a.M) {
            uint4 v = *(const uint4 *)(Ys + row * BN2 + piece * 8);
            if (a.mask && !a.mask[m % a.HWp]) v = make_uint4(0, 0, 0, 0);
            *(uint4 *)(a.Y + m * a.Cout + n0 + piece * 8) = v;
        }
    }
}

// ---- v3: v2's tile with the activation operand prefetched TWO K-tiles ahead (3 A stages + 2 B stages = 160 KiB of
// LDS), counted vmcnt and a raw s_barrier so that the LDS-DMA of tile kt+2 stays in flight across the barrier.
__global__ __launch_bounds__(THREADS2) void conv3x3_mfma_v3_kernel(const ConvArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds3[];                 // [A0 A1 A2 | B0 B1] x 32 KiB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int64_t m0 = (int64_t)blockIdx.x * BM2;
    const int n0 = blockIdx.y * BN2;
    const int cpt = a.Cin / BK, KT = 9 * cpt;
    constexpr int TILE = BM2 * BK * 2;                            // 32 KiB
    unsigned char *const Abase = lds3, *const Bbase = lds3 + 3 * TILE;

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // per-lane byte offsets of the MFMA fragments inside a 32 KiB operand tile, for the 4 k-steps of a K-tile
    uint32_t a_off[4][4], b_off[2][4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        const int chunk = kk * 2 + (lane >> 5);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) { const int row = wm * 128 + mi * 32 + (lane & 31); a_off[mi][kk] = row * 128 + ((chunk ^ SWZ(row)) << 4); }
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) { const int row = wn * 64 + ni * 32 + (lane & 31); b_off[ni][kk] = row * 128 + ((chunk ^ SWZ(row)) << 4); }
    }
    const uint32_t lds_base = (uint32_t)(uintptr_t)lds3;

    auto stage_a = [&](const int kt) {
        const int c0 = (kt / 9) * BK, t = kt % 9;      // channel chunk outer, tap inner: a tile's input lines stay in L2 across its 9 taps
        const int off = (t / 3 - 1) * a.Wp + (t % 3 - 1);
        unsigned char *base = Abase + (kt % 3) * TILE;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = i * THREADS2 + tid, row = q >> 3, cp = q & 7;
            int64_t m = m0 + row + off;
            m = m < 0 ? 0 : (m >= a.M ? a.M - 1 : m);
            const uint16_t *g = a.X + m * a.Cin + c0 + ((cp ^ SWZ(row)) << 3);
            __builtin_amdgcn_global_load_lds((const void *)g, (lds_void_ptr)(base + q * 16), 16, 0, 0);
        }
    };
    auto stage_b = [&](const int kt) {
        const int c0 = (kt / 9) * BK, t = kt % 9;      // channel chunk outer, tap inner: a tile's input lines stay in L2 across its 9 taps
        unsigned char *base = Bbase + (kt & 1) * TILE;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = i * THREADS2 + tid, row = q >> 3, cp = q & 7;
            const uint16_t *g = a.W + ((int64_t)(n0 + row) * 9 + t) * a.Cin + c0 + ((cp ^ SWZ(row)) << 3);
            __builtin_amdgcn_global_load_lds((const void *)g, (lds_void_ptr)(base + q * 16), 16, 0, 0);
        }
    };

    // one 16-byte-per-lane LDS-DMA of a stage (piece i of 4): an LDS-DMA costs its wave 60-185 issue cycles, so the
    // pieces of the next tiles are spread between the MFMAs instead of being issued back to back after the barrier
    auto piece_a = [&](const int kt, const int i) {
        const int c0 = (kt / 9) * BK, t = kt % 9;      // channel chunk outer, tap inner: a tile's input lines stay in L2 across its 9 taps
        const int off = (t / 3 - 1) * a.Wp + (t % 3 - 1);
        const int q = i * THREADS2 + tid, row = q >> 3, cp = q & 7;
        int64_t m = m0 + row + off;
        m = m < 0 ? 0 : (m >= a.M ? a.M - 1 : m);
        const uint16_t *g = a.X + m * a.Cin + c0 + ((cp ^ SWZ(row)) << 3);
        __builtin_amdgcn_global_load_lds((const void *)g, (lds_void_ptr)(Abase + (kt % 3) * TILE + q * 16), 16, 0, 0);
    };
    auto piece_b = [&](const int kt, const int i) {
        const int c0 = (kt / 9) * BK, t = kt % 9;      // channel chunk outer, tap inner: a tile's input lines stay in L2 across its 9 taps
        const int q = i * THREADS2 + tid, row = q >> 3, cp = q & 7;
        const uint16_t *g = a.W + ((int64_t)(n0 + row) * 9 + t) * a.Cin + c0 + ((cp ^ SWZ(row)) << 3);
        __builtin_amdgcn_global_load_lds((const void *)g, (lds_void_ptr)(Bbase + (kt & 1) * TILE + q * 16), 16, 0, 0);
    };

    stage_a(0); stage_b(0); stage_a(1);
    for (int kt = 0; kt < KT; ++kt) {
        // tile kt (A issued two iterations ago, B one) must have landed; the 4 LDS-DMAs of A(kt+1) may stay in flight
        if (kt + 1 < KT) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const bool nb = kt + 1 < KT, na = kt + 2 < KT;           // wave-uniform
        // Fragments of k-step kk+1 are read (into their own registers) before the MFMAs of k-step kk are issued, so the
        // LDS latency hides behind 8 MFMAs.  Inline asm + counted lgkmcnt: left to itself hipcc reuses one register set
        // and waits lgkmcnt(0) in front of every pair of MFMAs.
        const uint32_t abase = lds_base + (uint32_t)((kt % 3) * TILE), bbase = lds_base + (uint32_t)(3 * TILE + (kt & 1) * TILE);
        bf16x8 fa[2][4], fb[2][2];
#define RN_DS_READ(dst, addr) asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr))
#define RN_LOAD_FRAGS(KK, SET)                                                     \
        RN_DS_READ(fb[SET][0], bbase + b_off[0][KK]); RN_DS_READ(fb[SET][1], bbase + b_off[1][KK]);   \
        RN_DS_READ(fa[SET][0], abase + a_off[0][KK]); RN_DS_READ(fa[SET][1], abase + a_off[1][KK]);   \
        RN_DS_READ(fa[SET][2], abase + a_off[2][KK]); RN_DS_READ(fa[SET][3], abase + a_off[3][KK]);
#define RN_MFMA4(SET, MI0)                                                        \
        _Pragma("unroll") for (int mi = MI0; mi < MI0 + 2; ++mi)                   \
            _Pragma("unroll") for (int ni = 0; ni < 2; ++ni)                       \
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[SET][mi], fb[SET][ni], acc[mi][ni], 0, 0, 0);
        RN_LOAD_FRAGS(0, 0)
        RN_LOAD_FRAGS(1, 1)
        asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory"); __builtin_amdgcn_sched_barrier(0);
        RN_MFMA4(0, 0) __builtin_amdgcn_sched_barrier(0); if (nb) piece_b(kt + 1, 0); __builtin_amdgcn_sched_barrier(0);
        RN_MFMA4(0, 2) __builtin_amdgcn_sched_barrier(0); if (nb) piece_b(kt + 1, 1); __builtin_amdgcn_sched_barrier(0);
        RN_LOAD_FRAGS(2, 0)
        asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory"); __builtin_amdgcn_sched_barrier(0);
        RN_MFMA4(1, 0) __builtin_amdgcn_sched_barrier(0); if (nb) piece_b(kt + 1, 2); __builtin_amdgcn_sched_barrier(0);
        RN_MFMA4(1, 2) __builtin_amdgcn_sched_barrier(0); if (nb) piece_b(kt + 1, 3); __builtin_amdgcn_sched_barrier(0);
        RN_LOAD_FRAGS(3, 1)
        asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory"); __builtin_amdgcn_sched_barrier(0);
        RN_MFMA4(0, 0) __builtin_amdgcn_sched_barrier(0); if (na) piece_a(kt + 2, 0); __builtin_amdgcn_sched_barrier(0);
        RN_MFMA4(0, 2) __builtin_amdgcn_sched_barrier(0); if (na) piece_a(kt + 2, 1); __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0);
        RN_MFMA4(1, 0) __builtin_amdgcn_sched_barrier(0); if (na) piece_a(kt + 2, 2); __builtin_amdgcn_sched_barrier(0);
        RN_MFMA4(1, 2) __builtin_amdgcn_sched_barrier(0); if (na) piece_a(kt + 2, 3); __builtin_amdgcn_sched_barrier(0);
#undef RN_MFMA4
#undef RN_DS_READ
#undef RN_LOAD_FRAGS
#undef RN_MFMAS
    }

    __syncthreads();
    uint16_t *Ys = (uint16_t *)lds3;
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
                Ys[row * BN2 + col] = f2bf(v);
            }
        }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int q = i * THREADS2 + tid, row = q >> 5, piece = q & 31;
        const int64_t m = m0 + row;
        if (m < a.M) {
            uint4 v = *(const uint4 *)(Ys + row * BN2 + piece * 8);
            if (a.mask && !a.mask[m % a.HWp]) v = make_uint4(0, 0, 0, 0);
            *(uint4 *)(a.Y + m * a.Cout + n0 + piece * 8) = v;
        }
    }
}

// ---- v4: v3's tile and LDS stages, with the two waves of every SIMD in PING-PONG: waves 0-3 (group 0) and 4-7
// (group 1) run one barrier interval apart, so that while one group is in its MFMA phase (16 MFMAs) the other is in
// its load phase (12 ds_read_b128 for its next 16 MFMAs + 4 LDS-DMA pieces of a future tile).  Each K-tile is two
// (load, MFMA) phase pairs = 4 barriers.  LDS-DMA retirement: counted vmcnt at the end of the second load phase,
// one barrier before any wave of either group reads the tile.
__global__ __launch_bounds__(THREADS2) void conv3x3_mfma_v4_kernel(const ConvArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds4[];                 // [A0 A1 A2 | B0 B1] x 32 KiB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int64_t m0 = (int64_t)blockIdx.x * BM2;
    const int n0 = blockIdx.y * BN2;
    const int cpt = a.Cin / BK, KT = 9 * cpt;
    constexpr int TILE = BM2 * BK * 2;
    unsigned char *const Abase = lds4, *const Bbase = lds4 + 3 * TILE;

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
    const uint32_t lds_base = (uint32_t)(uintptr_t)lds4;

    auto piece_a = [&](const int kt, const int i) {
        const int c0 = (kt / 9) * BK, t = kt % 9;
        const int off = (t / 3 - 1) * a.Wp + (t % 3 - 1);
        const int q = i * THREADS2 + tid, row = q >> 3, cp = q & 7;
        int64_t m = m0 + row + off;
        m = m < 0 ? 0 : (m >= a.M ? a.M - 1 : m);
        const uint16_t *g = a.X + m * a.Cin + c0 + ((cp ^ SWZ(row)) << 3);
        __builtin_amdgcn_global_load_lds((const void *)g, (lds_void_ptr)(Abase + (kt % 3) * TILE + q * 16), 16, 0, 0);
    };
    auto piece_b = [&](const int kt, const int i) {
        const int c0 = (kt / 9) * BK, t = kt % 9;
        const int q = i * THREADS2 + tid, row = q >> 3, cp = q & 7;
        const uint16_t *g = a.W + ((int64_t)(n0 + row) * 9 + t) * a.Cin + c0 + ((cp ^ SWZ(row)) << 3);
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
        if (kt + 1 < KT && !(a.relu & 512)) {
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
            if (!(a.relu & 256)) {
#pragma unroll
                for (int i = 0; i < 4; ++i) piece_a(kt + 2, i);
            }
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
    uint16_t *Ys = (uint16_t *)lds4;
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
                Ys[row * BN2 + col] = f2bf(v);
            }
        }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int q = i * THREADS2 + tid, row = q >> 5, piece = q & 31;
        const int64_t m = m0 + row;
        if (m < a.M) {
            uint4 v = *(const uint4 *)(Ys + row * BN2 + piece * 8);
            if (a.mask && !a.mask[m % a.HWp]) v = make_uint4(0, 0, 0, 0);
            *(uint4 *)(a.Y + m * a.Cout + n0 + piece * 8) = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
static uint16_t h_f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }
static float h_bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }

static void fill(std::vector<uint16_t> &v, unsigned seed, float scale)
{
    uint32_t s = seed * 2654435761u + 12345u;
    for (auto &x : v) { s = s * 1664525u + 1013904223u; x = h_f2bf(((int)((s >> 8) & 0xffff) - 32768) / 32768.0f * scale); }
}

int main(int argc, char **argv)
{
    CK(hipFuncSetAttribute((const void *)conv3x3_mfma_v2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    CK(hipFuncSetAttribute((const void *)conv3x3_mfma_v3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
    CK(hipFuncSetAttribute((const void *)conv3x3_mfma_v4_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
    // ---- correctness: N=2, Hp=7, Wp=9, Cin=128, Cout=256, interior-only mask
    for (int ver = 1; ver <= 4; ++ver) {
        const int N = 2, Hp = 7, Wp = 9, Cin = 128, Cout = 256;
        const int64_t HWp = Hp * Wp, M = N * HWp;
        std::vector<uint16_t> X(M * Cin), W((size_t)Cout * 9 * Cin), Y(M * Cout);
        std::vector<float> bias(Cout);
        std::vector<uint8_t> mask(HWp, 0);
        fill(X, 1, 1.0f); fill(W, 2, 0.05f);
        for (int i = 0; i < Cout; ++i) bias[i] = 0.01f * (i % 17) - 0.05f;
        for (int y = 1; y < Hp - 1; ++y) for (int x = 1; x < Wp - 1; ++x) mask[y * Wp + x] = 1;
        for (int n = 0; n < N; ++n) for (int y = 0; y < Hp; ++y) for (int x = 0; x < Wp; ++x)          // zero border in X
            if (!mask[y * Wp + x]) for (int c = 0; c < Cin; ++c) X[((n * HWp) + y * Wp + x) * Cin + c] = 0;
        uint16_t *dX, *dW, *dY; float *dB; uint8_t *dM;
        CK(hipMalloc(&dX, X.size() * 2)); CK(hipMalloc(&dW, W.size() * 2)); CK(hipMalloc(&dY, Y.size() * 2));
        CK(hipMalloc(&dB, Cout * 4)); CK(hipMalloc(&dM, HWp));
        CK(hipMemcpy(dX, X.data(), X.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dW, W.data(), W.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(dB, bias.data(), Cout * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dM, mask.data(), HWp, hipMemcpyHostToDevice));
        ConvArgs a{dX, dW, dB, dM, dY, M, HWp, Cin, Cout, Wp, 1};
        if (ver == 1) hipLaunchKernelGGL(conv3x3_mfma_kernel, dim3((unsigned)((M + BM - 1) / BM), Cout / BN), dim3(THREADS), 0, 0, a);
        else if (ver == 2) hipLaunchKernelGGL(conv3x3_mfma_v2_kernel, dim3((unsigned)((M + BM2 - 1) / BM2), Cout / BN2), dim3(THREADS2), 131072, 0, a);
        else if (ver == 3) hipLaunchKernelGGL(conv3x3_mfma_v3_kernel, dim3((unsigned)((M + BM2 - 1) / BM2), Cout / BN2), dim3(THREADS2), 163840, 0, a);
        else hipLaunchKernelGGL(conv3x3_mfma_v4_kernel, dim3((unsigned)((M + BM2 - 1) / BM2), Cout / BN2), dim3(THREADS2), 163840, 0, a);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(Y.data(), dY, Y.size() * 2, hipMemcpyDeviceToHost));
        double maxerr = 0; int bad = 0;
        for (int64_t m = 0; m < M; ++m) {
            const int64_t p = m % HWp;
            for (int n = 0; n < Cout; ++n) {
                float ref = 0.0f;
                if (mask[p]) {
                    double s = bias[n];
                    for (int t = 0; t < 9; ++t) {
                        const int64_t mm = m + (t / 3 - 1) * Wp + (t % 3 - 1);
                        for (int c = 0; c < Cin; ++c) s += (double)h_bf2f(X[mm * Cin + c]) * (double)h_bf2f(W[((size_t)n * 9 + t) * Cin + c]);
                    }
                    ref = s > 0 ? (float)s : 0.0f;
                }
                const float got = h_bf2f(Y[m * Cout + n]);
                const double err = fabs(got - ref);
                if (err > maxerr) maxerr = err;
                if (err > 0.02 + 0.01 * fabs(ref)) { if (bad < 5) printf("  mismatch m=%ld n=%d got %f ref %f\n", (long)m, n, got, ref); ++bad; }
            }
        }
        printf("v%d correctness: max abs err %.4g, mismatches %d of %ld\n", ver, maxerr, bad, (long)(M * Cout));
        hipFree(dX); hipFree(dW); hipFree(dY); hipFree(dB); hipFree(dM);
    }
    // ---- timing at the R50 canvas shape
    {
        const int N = 8, Hp = 153, Wp = 170, Cin = 256, Cout = 256;
        const int64_t HWp = (int64_t)Hp * Wp;
        const int64_t M = (argc > 1) ? atoll(argv[1]) : N * HWp;          // argv[1]: row count override (tail-effect experiments)
        std::vector<uint16_t> X(M * Cin), W((size_t)Cout * 9 * Cin);
        fill(X, 3, 1.0f); fill(W, 4, 0.05f);
        std::vector<uint8_t> mask(HWp, 1);
        std::vector<float> bias(Cout, 0.1f);
        uint16_t *dX, *dW, *dY; float *dB; uint8_t *dM;
        CK(hipMalloc(&dX, X.size() * 2)); CK(hipMalloc(&dW, W.size() * 2)); CK(hipMalloc(&dY, (size_t)M * Cout * 2));
        CK(hipMalloc(&dB, Cout * 4)); CK(hipMalloc(&dM, HWp));
        CK(hipMemcpy(dX, X.data(), X.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dW, W.data(), W.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(dB, bias.data(), Cout * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dM, mask.data(), HWp, hipMemcpyHostToDevice));
        ConvArgs a{dX, dW, dB, dM, dY, M, HWp, Cin, Cout, Wp, 1 | (argc > 2 ? atoi(argv[2]) : 0)};
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int ver = 1; ver <= 4; ++ver) {
            const dim3 grid = ver == 1 ? dim3((unsigned)((M + BM - 1) / BM), Cout / BN) : dim3((unsigned)((M + BM2 - 1) / BM2), Cout / BN2);
            auto launch = [&]() {
                if (ver == 1) hipLaunchKernelGGL(conv3x3_mfma_kernel, grid, dim3(THREADS), 0, 0, a);
                else if (ver == 2) hipLaunchKernelGGL(conv3x3_mfma_v2_kernel, grid, dim3(THREADS2), 131072, 0, a);
                else if (ver == 3) hipLaunchKernelGGL(conv3x3_mfma_v3_kernel, grid, dim3(THREADS2), 163840, 0, a);
                else hipLaunchKernelGGL(conv3x3_mfma_v4_kernel, grid, dim3(THREADS2), 163840, 0, a);
            };
            for (int i = 0; i < 3; ++i) launch();
            CK(hipDeviceSynchronize());
            const int reps = 20;
            CK(hipEventRecord(e0));
            for (int i = 0; i < reps; ++i) launch();
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            const double us = ms * 1e3 / reps, flops = 2.0 * M * Cout * 9.0 * Cin;
            printf("v%d canvas conv M=%ld 256->256: %.1f us  %.0f TFLOP/s  (%u x %u blocks)\n", ver, (long)M, us, flops / us / 1e6, grid.x, grid.y);
        }
    }
    return 0;
}
