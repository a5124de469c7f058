// Backbone / FPN convolutions as bf16 MFMA GEMMs on channels-last activations, with the BatchNorm work that surrounds them
// fused in (reference: the Bottleneck of retinanet/backbone.py:105-136 -- conv1x1 -> bn -> relu ->
// conv3x3 -> bn -> relu -> conv1x1 -> bn -> (+ identity) -> relu -- and the 1x1 / 3x3 convs of the FPN, layers.py:44-64).
//
// Why: at the R50 shapes (8 x 200 x 336 x {64, 256} at layer1) these convolutions are HBM-bound -- 51 .. 102 flop per byte
// against the ~300 the chip needs to be compute-bound -- and so is every BatchNorm pass around them.  MIOpen runs
// conv, statistics, normalise, ReLU and their gradients as separate passes over the same 69 - 275 MB tensors (2/3 of the
// step in round 2).  Here the GEMM's operand load applies the PREVIOUS layer's BatchNorm + ReLU (forward) or BatchNorm
// backward (the gradient GEMMs), and its epilogue produces what the NEXT BatchNorm kernel needs (per-channel sums of the
// output; ReLU mask + the two sums of the BatchNorm backward; the residual branch's gradient), so those passes disappear.
//
//   pw_gemm_kernel    Y[M][N] = pro(X)[M][taps * Cin] . W[N][taps * Cin]^T      forward and data gradient
//                     (1x1 and 3x3, stride 1 / 2 through a per-row position decode; a 3x3 data gradient is the same
//                     kernel on the tap-reversed, role-swapped weight)
//   pw_wgrad_kernel   dW[N][taps][Cin] = sum_m pro(G)[m][N] . pro(X)[pos(m, tap)][Cin]        weight gradient
//
// Both stage their operands global -> registers -> (transform) -> LDS, because the transforms need the data in registers
// and the 3x3 / strided gathers need per-row validity; tiles are 128 x {64,128} x 64 with 4 waves, two workgroups per CU
// overlap each other's loads and MFMAs (the convolutions this file serves are bandwidth-bound: the matrix pipe is not the
// limit).  Numerics: bf16 operands, f32 accumulation; every fused BatchNorm expression is the one csrc/norm.hip uses
// (same fma order, statistics taken from the bf16-ROUNDED output), so fused and unfused layers agree to the bit wherever
// the GEMM's own summation order does.
#include "rn_common.hpp"
#include <type_traits>

#ifndef PW_GEMM_XCD_MAP
#define PW_GEMM_XCD_MAP 1      // pw_gemm_kernel: the column tiles of a row-tile walker on one XCD
#endif

namespace {

using rn::f32x16;

constexpr int PW_BM = 128, PW_BK = 64, PW_THREADS = 256;
constexpr int PW_MAX_GX = 512;                  // persistent row-tile walkers = BN partial rows (norm.hip: BN_MAX_BLOCKS)
#define PW_SWZ(row) (((row) >> 1) & 7)

enum { PRO_NONE = 0, PRO_AFFINE_RELU = 1, PRO_BN_BWD = 2 };
enum { EPI_STATS = 1, EPI_RESID = 2, EPI_RELU_BWD = 4, EPI_BIAS = 8 };

// norm.hip: relu_alive_threshold<DT> -- half the smallest subnormal of the element type (ties to even -> 0)
template <int DT> __device__ __forceinline__ float alive_dt() { return __uint_as_float(DT == RN_F16 ? 0x33000000u : 0x00004000u); }

struct PwArgs {
    const uint16_t *X, *X2;         // X: [rows][Cin];  PRO_BN_BWD: X = upstream gradient g, X2 = the BN input z (same shape)
    const uint8_t *xbits;           // PRO_BN_BWD, relu_mode 3: ReLU bits of the BN output [rows][Cin / 8]
    const uint16_t *W;              // [N][taps * Cin]
    uint16_t *Y;                    // [M][N]
    const float *pa, *pb, *pc;      // AFFINE_RELU: y = relu(x * pa + pb);  BN_BWD: dz = pa * g' + pc * z + pb   (a, k0, k1)
    const float *fa, *fb;           // BN_BWD, relu_mode 2: g' = g * [fma(z, fa, fb) > alive]
    float *partial;                 // EPI_STATS / EPI_RELU_BWD: [gx][2][N]
    const uint16_t *R;              // EPI_RESID: Y += R * rbits   ([M][N] and [M][N / 8]; rbits null: Y += R)
    const uint8_t *rbits;
    int rs, rH, rW;                 // rs == 2: R lives on the stride-2 grid [n][ceil(rH / 2)][ceil(rW / 2)] of the output grid [n][rH][rW] and
                                    //          joins the rows with even (y, x) only (the data gradient of a 1x1 / stride-2 convolution)
    const uint16_t *Zp;             // EPI_RELU_BWD: Y = Y * [fma(Zp, ea, eb) > alive]; sums of Y and Y * (Zp - emean) * einv
    const float *ea, *eb, *emean, *einv;
    const float *bias;              // EPI_BIAS: Y = act(acc + bias[n] (+ R)), act = ReLU when relu_out (inference: a folded BatchNorm + residual + ReLU)
    int relu_out;
    int M, Cin, N, taps;            // M output rows; K = taps * Cin
    int relu_mode;                  // PRO_BN_BWD: 0 none, 2 recomputed from X2, 3 bits
    int stride, pad, Ho, Wo, H, W_; // position decode of output row m = (n, ho, wo) -> input (n, ho * stride - pad + dy, ...)
    int gx;                         // row-tile walkers per column tile
    int xcd_cols;                   // > 0: 1-D grid of 8 * ceil(gx / 8) * xcd_cols workgroups, a walker's column tiles on one XCD (see the kernel)
    int f16;                        // host side: fp16 elements instead of bf16 (kernel instantiation)
};

// one output row of the tile as this thread sees it: image base (in rows) and the top-left input coordinate
struct RowPos { int base, y0, x0; bool ok; };

__device__ __forceinline__ RowPos decode_row(const PwArgs &a, const int m)
{
    RowPos r;
    r.ok = m < a.M;
    const int mm = r.ok ? m : 0;
    if (a.taps == 1 && a.stride == 1) { r.base = mm; r.y0 = 0; r.x0 = 0; return r; }
    const int hw = a.Ho * a.Wo;
    const int n = mm / hw, rem = mm - n * hw, ho = rem / a.Wo, wo = rem - ho * a.Wo;
    r.base = n * a.H * a.W_;
    r.y0 = ho * a.stride - a.pad;
    r.x0 = wo * a.stride - a.pad;
    return r;
}

__device__ __forceinline__ void ld8f(const float *p, float (&f)[8])
{
    const rn::f32x4 u = ((const rn::f32x4 *)p)[0], v = ((const rn::f32x4 *)p)[1];
    f[0] = u.x; f[1] = u.y; f[2] = u.z; f[3] = u.w; f[4] = v.x; f[5] = v.y; f[6] = v.z; f[7] = v.w;
}

// The operand transforms on one 16-byte vector of 8 consecutive channels (same expressions as norm.hip).
struct ProCoef { float a[8], b[8], c[8], fa[8], fb[8]; };
template <int PRO>
__device__ __forceinline__ void load_coef(ProCoef &k, const float *pa, const float *pb, const float *pc, const float *fa, const float *fb,
                                          const int relu_mode, const int ch)
{
    if (PRO == PRO_NONE) return;
    ld8f(pa + ch, k.a); ld8f(pb + ch, k.b);
    if (PRO == PRO_BN_BWD) {
        ld8f(pc + ch, k.c);
        if (relu_mode == 2) { ld8f(fa + ch, k.fa); ld8f(fb + ch, k.fb); }
    }
}
template <int DT, int PRO>
__device__ __forceinline__ rn::u32x4 transform(const rn::u32x4 x, const rn::u32x4 z, const uint32_t bits, const ProCoef &k, const int relu_mode,
                                               const bool valid)
{
    if (PRO == PRO_NONE) return valid ? x : rn::u32x4{0u, 0u, 0u, 0u};
    float f[8], g[8];
    rn::dt<DT>::unpack(x, f);
    if (PRO == PRO_AFFINE_RELU) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float t = fmaf(f[j], k.a[j], k.b[j]); f[j] = t > 0.0f ? t : 0.0f; }
    } else {
        rn::dt<DT>::unpack(z, g);
        const float alive = alive_dt<DT>();
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float gj = f[j];
            if (relu_mode == 2 && !(fmaf(g[j], k.fa[j], k.fb[j]) > alive)) gj = 0.0f;
            if (relu_mode == 3 && !((bits >> j) & 1u)) gj = 0.0f;
            f[j] = fmaf(k.a[j], gj, fmaf(k.c[j], g[j], k.b[j]));
        }
    }
    const rn::u32x4 o = rn::dt<DT>::pack(f);
    return valid ? o : rn::u32x4{0u, 0u, 0u, 0u};
}

// ------------------------------------------------------------------------------------------------------------------
// PLAIN: 1x1 / stride 1 (every bottleneck GEMM) -- no position decode anywhere, see the staging addresses below.
template <int DT, int BN, int PRO, int EPI, bool PLAIN>
__global__ __launch_bounds__(PW_THREADS, 2) void pw_gemm_kernel(const PwArgs a)
{
    constexpr int MI = 2, NI = BN / 64;                         // 2 x 2 waves of 64 x (BN / 2)
    constexpr int A_TILE = PW_BM * PW_BK * 2, B_TILE = BN * PW_BK * 2, STAGE = A_TILE + B_TILE;
    constexpr int BROWS = BN / 32;                              // weight rows staged per thread
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];      // 2 stages of [A | B]; the epilogue's f32 tile
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    // (walker, column tile) of this workgroup.  xcd_cols > 0: a 1-D grid placed by XCD (workgroups go round-robin over the 8 XCDs) -- the
    // column tiles of a walker run on ONE XCD, next to each other in time, so the walker's X rows enter that XCD's L2 once instead of once
    // per column tile from HBM (the inference conv3 of layer3: 8 column tiles of 57.8 MB of rows = 462 MB of reads for 520 MB of operands)
    int walker = blockIdx.x, ctile = blockIdx.y;
    if (a.xcd_cols > 0) {
        const int L = blockIdx.x, t = L >> 3;
        ctile = t % a.xcd_cols;
        walker = (t / a.xcd_cols) * 8 + (L & 7);
        if (walker >= a.gx) return;
    }
    const int n0 = ctile * BN;
    const int cpt = a.Cin / PW_BK, KT = a.taps * cpt, Ktot = a.taps * a.Cin;
    const int c = tid & 7, r0 = tid >> 3;                       // this thread stages 16-byte chunk c of rows r0 + 32 i
    const int MT = (a.M + PW_BM - 1) / PW_BM;

    uint32_t a_off[4], b_off[4];                                // fragment addresses of the 4 k-steps (first fragment)
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        const int chunk = kk * 2 + (lane >> 5);
        { const int row = wm * 64 + (lane & 31); a_off[kk] = row * 128 + ((chunk ^ PW_SWZ(row)) << 4); }
        { const int row = wn * (BN / 2) + (lane & 31); b_off[kk] = A_TILE + row * 128 + ((chunk ^ PW_SWZ(row)) << 4); }
    }
    uint32_t wa_off[4], wb_off[BROWS];                          // where this thread's staged chunks go
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int row = r0 + 32 * i; wa_off[i] = row * 128 + ((c ^ PW_SWZ(row)) << 4); }
#pragma unroll
    for (int i = 0; i < BROWS; ++i) { const int row = r0 + 32 * i; wb_off[i] = A_TILE + row * 128 + ((c ^ PW_SWZ(row)) << 4); }

    // column statistics of the tiles this workgroup walks (EPI_STATS / EPI_RELU_BWD): thread = 8 fixed columns
    constexpr int CG = BN / 8, RL = PW_THREADS / CG;            // column groups, row lanes of the epilogue
    const int ecg = tid % CG, erl = tid / CG;
    float ssum[8], qsum[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { ssum[j] = 0.0f; qsum[j] = 0.0f; }

    // Software pipeline over the row tiles this workgroup walks (these GEMMs are bandwidth-bound; what matters is bytes in
    // flight): the first K-tile of tile t+1 is loaded while tile t's epilogue runs, and the epilogue's own operands (residual +
    // bits, or the activation whose ReLU mask / BN sums it forms) are loaded at the top of the tile, under its MFMAs.
    constexpr int EROWS = PW_BM / RL;                           // epilogue rows per thread
    // Staging registers of one K-tile.  (A second set -- two K-tiles in flight, K-tile j in set j & 1 -- was measured in round 6: the
    // bias-only variants gain 2 - 10 % at 231 - 249 VGPRs, the residual variant spills; every shape it helps is hipBLASLt's anyway.)
    struct Stg {
        rn::u32x4 sx[4], sz[4], sw[BROWS];
        uint32_t sbits[4];
        bool sval[4];
        int sc0;                                                // first channel of the staged K-tile
    };
    Stg S0;
#pragma unroll
    for (int i = 0; i < 4; ++i) S0.sbits[i] = 0xffu;
    S0.sc0 = 0;
    RowPos rp[4];                                                // (!PLAIN: the decoded rows of the tile being requested)
    // PLAIN: the source of a staged vector is a WAVE-UNIFORM 64-bit base -- tile origin + channel chunk, scalar work -- plus a per-thread
    // 32-bit offset computed once per kernel.  The general path forms every address from the row's decoded position (a 64-bit multiply +
    // clamp chain per vector, 12 vectors per K-tile): the s_memrealtime stamps of the K loop showed 0.68 - 0.78 us of the 1.5 us per
    // K-tile in that issue phase, more than the fragment reads + MFMAs (0.39 - 0.45).  Weights: the same in both forms.
    uint32_t xoff[4], woff[BROWS];
#pragma unroll
    for (int i = 0; i < 4; ++i) xoff[i] = (uint32_t)((r0 + 32 * i) * a.Cin + c * 8) * 2u;              // (host: 128 rows of Cin channels < 2^31 bytes)
#pragma unroll
    for (int i = 0; i < BROWS; ++i) woff[i] = (uint32_t)((r0 + 32 * i) * Ktot + c * 8) * 2u;
    int stage_m0 = 0;                                            // first row of the row tile whose K-tiles are being requested
    // the prologue's per-channel coefficients live in LDS behind the staging area: [a | b | c | fa | fb][Cin] f32
    constexpr int LDS_MAIN = (2 * STAGE > PW_BM * BN * 4) ? 2 * STAGE : PW_BM * BN * 4;
    float *const s_coef = (float *)(lds + LDS_MAIN);
    if (PRO != PRO_NONE) {
        const int ncoef = PRO == PRO_AFFINE_RELU ? 2 : (a.relu_mode == 2 ? 5 : 3);
        for (int q = tid; q < ncoef * a.Cin; q += PW_THREADS) {
            const int which = q / a.Cin, ch = q - which * a.Cin;
            const float *src = which == 0 ? a.pa : (which == 1 ? a.pb : (which == 2 ? a.pc : (which == 3 ? a.fa : a.fb)));
            s_coef[q] = src[ch];
        }
        __syncthreads();
    }
    auto issue = [&](const int kt, Stg &s) {
        const int tap = kt / cpt, c0 = (kt - tap * cpt) * PW_BK;
        const int dy = a.taps == 1 ? 0 : tap / 3, dx = a.taps == 1 ? 0 : tap - (tap / 3) * 3;
        {
            const unsigned char *const wb = (const unsigned char *)a.W + ((int64_t)n0 * Ktot + kt * PW_BK) * 2;
#pragma unroll
            for (int i = 0; i < BROWS; ++i) s.sw[i] = *(const rn::u32x4 *)(wb + woff[i]);
        }
        s.sc0 = c0;
        if (PLAIN) {
            const int64_t e0 = (int64_t)stage_m0 * a.Cin + c0;
            const unsigned char *const xb = (const unsigned char *)a.X + e0 * 2;
            const bool ragged = stage_m0 + PW_BM > a.M;          // (uniform) the last row tile: rows past the end read row M - 1 and stage zeros
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bool v = !ragged || stage_m0 + r0 + 32 * i < a.M;
                const uint32_t o = v ? xoff[i] : (uint32_t)((a.M - 1 - stage_m0) * a.Cin + c * 8) * 2u;
                s.sval[i] = v;
                s.sx[i] = *(const rn::u32x4 *)(xb + o);
                if (PRO == PRO_BN_BWD) {
                    s.sz[i] = *(const rn::u32x4 *)((const unsigned char *)a.X2 + e0 * 2 + o);
                    s.sbits[i] = a.xbits ? (a.xbits + (e0 >> 3))[o >> 4] : 0xffu;
                }
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int y = rp[i].y0 + dy, x = rp[i].x0 + dx;
            const bool v = rp[i].ok && (a.taps == 1 && a.stride == 1 ? true : (y >= 0 && y < a.H && x >= 0 && x < a.W_));
            s.sval[i] = v;
            const int64_t row = v ? (int64_t)rp[i].base + (a.taps == 1 && a.stride == 1 ? 0 : y * a.W_ + x) : 0;      // clamped: always a valid address
            const int64_t e = row * a.Cin + c0 + c * 8;
            s.sx[i] = *(const rn::u32x4 *)(a.X + e);
            if (PRO == PRO_BN_BWD) {
                s.sz[i] = *(const rn::u32x4 *)(a.X2 + e);
                // (the byte is fetched whatever the mode -- a load behind a run-time condition would serialise the batch)
                s.sbits[i] = a.xbits ? a.xbits[e >> 3] : 0xffu;
            }
        }
    };
    auto commit = [&](const int stage, const Stg &s) {
        unsigned char *const sb = lds + stage * STAGE;
        ProCoef coef;                                           // from the LDS copy: nothing held in registers across the MFMAs
        if (PRO != PRO_NONE) {
            const int ch = s.sc0 + c * 8;
            ld8f(s_coef + ch, coef.a); ld8f(s_coef + a.Cin + ch, coef.b);
            if (PRO == PRO_BN_BWD) {
                ld8f(s_coef + 2 * a.Cin + ch, coef.c);
                if (a.relu_mode == 2) { ld8f(s_coef + 3 * a.Cin + ch, coef.fa); ld8f(s_coef + 4 * a.Cin + ch, coef.fb); }
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
            *(rn::u32x4 *)(sb + wa_off[i]) = transform<DT, PRO>(s.sx[i], PRO == PRO_BN_BWD ? s.sz[i] : s.sx[i], s.sbits[i], coef, a.relu_mode, s.sval[i]);
#pragma unroll
        for (int i = 0; i < BROWS; ++i) *(rn::u32x4 *)(sb + wb_off[i]) = s.sw[i];
    };
    // EPI_RELU_BWD: the four per-column vectors of the epilogue, in LDS behind the prologue coefficients: [ea | eb | emean | einv][BN]
    float *const s_epi = s_coef + (PRO == PRO_NONE ? 0 : (PRO == PRO_AFFINE_RELU ? 2 : 5) * a.Cin);
    if (EPI & EPI_RELU_BWD) {
        for (int q = tid; q < 4 * BN; q += PW_THREADS) {
            const int which = q / BN, col = q - which * BN;
            const float *src = which == 0 ? a.ea : (which == 1 ? a.eb : (which == 2 ? a.emean : a.einv));
            s_epi[q] = src[n0 + col];
        }
        __syncthreads();
    }
    const float alive = alive_dt<DT>();

    float ebias[8];                                             // EPI_BIAS: this thread's 8 fixed columns of the bias
    if (EPI & EPI_BIAS) ld8f(a.bias + n0 + ecg * 8, ebias);
    int mt = walker;
    if (mt < MT) {
#pragma unroll
        for (int i = 0; i < 4; ++i) if (!PLAIN) rp[i] = decode_row(a, mt * PW_BM + r0 + 32 * i);
        stage_m0 = mt * PW_BM;
        issue(0, S0);
    }
    for (; mt < MT; mt += a.gx) {
        const int m0 = mt * PW_BM;
        // the epilogue's operands for this tile, in flight under the tile's staging and MFMAs
        rn::u32x4 er[EROWS];
        uint32_t ebits[EROWS];
        // (the widest variant -- BatchNorm-backward prologue, 128 columns -- has no registers left for this prefetch across its
        // K loop: it spilled 41 VGPRs and ran 12 % slower; there the loads go out right after the last MFMA instead)
        constexpr bool EPI_EARLY = !(PRO == PRO_BN_BWD && BN == 128);
        auto load_epi = [&]() {
            if (EPI & (EPI_RESID | EPI_RELU_BWD)) {
#pragma unroll
                for (int i = 0; i < EROWS; ++i) {
                    const int m = m0 + erl + i * RL, mc = m < a.M ? m : a.M - 1;
                    int64_t e = (int64_t)mc * a.N + n0 + ecg * 8;
                    bool on = true;
                    if ((EPI & EPI_RESID) && a.rs == 2) {
                        const int hw = a.rH * a.rW, n = mc / hw, rem = mc - n * hw, y = rem / a.rW, x = rem - y * a.rW;
                        on = ((y | x) & 1) == 0;
                        e = (((int64_t)n * ((a.rH + 1) >> 1) + (y >> 1)) * ((a.rW + 1) >> 1) + (x >> 1)) * a.N + n0 + ecg * 8;
                    }
                    const rn::u32x4 zero4 = {0u, 0u, 0u, 0u};
                    er[i] = on ? *(const rn::u32x4 *)(((EPI & EPI_RESID) ? a.R : a.Zp) + e) : zero4;
                    ebits[i] = (EPI & EPI_RESID) ? (on ? (a.rbits ? (uint32_t)a.rbits[e >> 3] : 0xffu) : 0u) : 0u;
                }
            }
        };
        if (EPI_EARLY) load_epi();
        f32x16 acc[MI][NI];
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
        auto mma_stage = [&](const int stage) {                 // the 16 MFMAs per wave of the K-tile in LDS stage `stage`
            const unsigned char *const sb = lds + stage * STAGE;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                typedef typename rn::mma<DT>::frag frag8;
                frag8 fa[MI], fb[NI];
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) fa[mi] = *(const frag8 *)(sb + a_off[kk] + mi * 4096);
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) fb[ni] = *(const frag8 *)(sb + b_off[kk] + ni * 4096);
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
                        acc[mi][ni] = rn::mma<DT>::m32(fa[mi], fb[ni], acc[mi][ni]);
            }
        };
        auto issue_next_tile = [&]() {                          // the NEXT row tile's first K-tile (set 0), in flight under the epilogue
            if (mt + a.gx < MT) {
#pragma unroll
                for (int i = 0; i < 4; ++i) if (!PLAIN) rp[i] = decode_row(a, (mt + a.gx) * PW_BM + r0 + 32 * i);
                stage_m0 = (mt + a.gx) * PW_BM;
                issue(0, S0);
            }
        };

        commit(0, S0);
        __syncthreads();
        for (int kt = 0; kt < KT; ++kt) {
            if (kt + 1 < KT) issue(kt + 1, S0);                 // in flight under this K-tile's MFMAs
            else issue_next_tile();
            mma_stage(kt & 1);
            if (kt + 1 < KT) commit((kt + 1) & 1, S0);
            __syncthreads();
        }

        // ---- epilogue: accumulators -> f32 tile in LDS -> rows of 8-channel vectors
        if (!EPI_EARLY) load_epi();
        float *const tile = (float *)lds;                       // [128][BN]
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                const int col = wn * (BN / 2) + ni * 32 + (lane & 31);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    tile[row * BN + col] = acc[mi][ni][r];
                }
            }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < EROWS; ++i) {
            const int row = erl + i * RL;
            const int m = m0 + row;
            if (m < a.M) {
                float v[8];
                ld8f(tile + row * BN + ecg * 8, v);
                const int64_t e = (int64_t)m * a.N + n0 + ecg * 8;
                if (EPI & EPI_BIAS) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] += ebias[j];
                }
                if (EPI & EPI_RESID) {
                    float r[8];
                    rn::dt<DT>::unpack(er[i], r);
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] += ((ebits[i] >> j) & 1u) ? r[j] : 0.0f;
                }
                if ((EPI & EPI_BIAS) && a.relu_out) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = v[j] > 0.0f ? v[j] : 0.0f;
                }
                rn::u32x4 o = rn::dt<DT>::pack(v);
                if (EPI & (EPI_STATS | EPI_RELU_BWD)) {
                    rn::dt<DT>::unpack(o, v);              // the statistics are those of the stored (rounded) tensor
                    if (EPI & EPI_RELU_BWD) {
                        float z[8], ea[8], eb[8], emu[8], eis[8];
                        ld8f(s_epi + ecg * 8, ea); ld8f(s_epi + BN + ecg * 8, eb);
                        ld8f(s_epi + 2 * BN + ecg * 8, emu); ld8f(s_epi + 3 * BN + ecg * 8, eis);
                        rn::dt<DT>::unpack(er[i], z);
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            if (!(fmaf(z[j], ea[j], eb[j]) > alive)) v[j] = 0.0f;
                            ssum[j] += v[j];
                            qsum[j] = fmaf(v[j], (z[j] - emu[j]) * eis[j], qsum[j]);
                        }
                        o = rn::dt<DT>::pack(v);          // exact: v is a bf16 value or zero
                    } else {
#pragma unroll
                        for (int j = 0; j < 8; ++j) { ssum[j] += v[j]; qsum[j] = fmaf(v[j], v[j], qsum[j]); }
                    }
                }
                *(rn::u32x4 *)(a.Y + e) = o;
            }
        }
        __syncthreads();                                        // the tile is the next row tile's staging area
    }

    if (EPI & (EPI_STATS | EPI_RELU_BWD)) {
        float *const red = (float *)lds;                        // [RL][2][BN]
#pragma unroll
        for (int j = 0; j < 8; ++j) { red[(erl * 2 + 0) * BN + ecg * 8 + j] = ssum[j]; red[(erl * 2 + 1) * BN + ecg * 8 + j] = qsum[j]; }
        __syncthreads();
        for (int q = tid; q < 2 * BN; q += PW_THREADS) {
            float t = 0.0f;
            for (int l = 0; l < RL; ++l) t += red[l * 2 * BN + q];
            const int which = q >= BN ? 1 : 0;
            a.partial[((int64_t)walker * 2 + which) * a.N + n0 + (q - which * BN)] = t;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Weight gradient.  dW[n][tap][c] = sum_m G'[m][n] * X'[pos(m, tap)][c]: the contraction index is the position, so both
// operands are k-strided ([position][channel] rows); tiles are staged as they lie (64 positions x TN / TK channels) and the
// MFMA fragments are read with the transposing ds_read_b64_tr_b16 (see conv.hip's weight-gradient kernel for the lane map).
// 16-byte chunks of a row are XOR-swizzled so that the four rows of a transposed 4 x 16 block hit different banks.
struct WgArgs {
    const uint16_t *G, *G2;         // [M][N]; PRO_BN_BWD: G = upstream gradient, G2 = BN input z
    const uint8_t *gbits;
    const float *ga, *gb, *gc, *gfa, *gfb;
    int g_relu_mode;
    const uint16_t *X;              // [rows][Cin]; PRO_AFFINE_RELU: relu(x * xa + xb)
    const float *xa, *xb;
    float *partial;                 // [S][N][taps * Cin] f32
    int M, N, Cin, taps, stride, pad, Ho, Wo, H, W_;
    int S, tiles_per_split;         // K-tiles of 64 positions per split
    int f16;                        // host side: fp16 elements (kernel instantiation)
    int xcd_tiles;                  // > 0: 1-D grid of 8 * ceil(S / 8) * xcd_tiles workgroups, a split's tiles on one XCD (see the kernel)
};

template <int W> __device__ __forceinline__ int tr_swz(const int row) { return W >= 128 ? ((row & 3) << 2) : (((row >> 1) & 1) << 2); }

template <int DT, int TN, int TK, int PROG, int PROX>
__global__ __launch_bounds__(PW_THREADS, 2) void pw_wgrad_kernel(const WgArgs a)
{
    constexpr int MI = TN / 64, NI = TK / 64;                   // 2 x 2 waves of (TN / 2) x (TK / 2)
    constexpr int G_ROWB = TN * 2, X_ROWB = TK * 2;
    constexpr int G_TILE = 64 * G_ROWB;
    constexpr int GV = TN / 32, XV = TK / 32;                   // 16-byte vectors per thread and tile
    constexpr int GCH = TN / 8, XCH = TK / 8;                   // chunks per row
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];      // [G | X]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_k = a.Cin / TK, tiles_n = a.N / TN;
    int split = blockIdx.x, t = blockIdx.y;
    if (a.xcd_tiles > 0) {
        // 1-D grid placed by XCD (workgroups go round-robin over the 8 XCDs): all tiles of a position split run on ONE XCD, next to each
        // other in time, so that the split's G and X rows -- re-read once per column / row tile: 275 MB for 86 MB of operands at the
        // layer3 shapes -- come out of that XCD's L2 after the first touch.  XCD x walks splits x, x + 8, ...
        const int L = blockIdx.x, w = L >> 3;
        split = (L & 7) + 8 * (w / a.xcd_tiles);
        t = w % a.xcd_tiles;
        if (split >= a.S) return;
    }
    const int tk = t % tiles_k; t /= tiles_k;
    const int tn = t % tiles_n; t /= tiles_n;
    const int tap = t;
    const int n0 = tn * TN, c0 = tk * TK;
    const int dy = a.taps == 1 ? 0 : tap / 3, dx = a.taps == 1 ? 0 : tap - (tap / 3) * 3;
    const bool plain = a.taps == 1 && a.stride == 1;

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // transposed-read addresses (conv.hip): lane = 16 grp + 4 q + p
    const int grp = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    uint32_t g_off[MI], x_off[NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int ch = wm * (TN / 16) + mi * 4 + 2 * (grp & 1) + (p >> 1);
        g_off[mi] = (uint32_t)((8 * (grp >> 1) + q) * G_ROWB + ((ch ^ tr_swz<TN>(q)) << 4) + (p & 1) * 8);
    }
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const int ch = wn * (TK / 16) + ni * 4 + 2 * (grp & 1) + (p >> 1);
        x_off[ni] = (uint32_t)(G_TILE + (8 * (grp >> 1) + q) * X_ROWB + ((ch ^ tr_swz<TK>(q)) << 4) + (p & 1) * 8);
    }
    const int gc_ = tid % GCH, gr0 = tid / GCH;                 // G staging: chunk gc_ of rows gr0 + i * (256 / GCH)
    const int xc_ = tid % XCH, xr0 = tid / XCH;
    constexpr int GRS = PW_THREADS / GCH, XRS = PW_THREADS / XCH;

    ProCoef gcoef, xcoef;
    load_coef<PROG>(gcoef, a.ga, a.gb, a.gc, a.gfa, a.gfb, a.g_relu_mode, n0 + gc_ * 8);
    load_coef<PROX>(xcoef, a.xa, a.xb, nullptr, nullptr, nullptr, 0, c0 + xc_ * 8);

    const int KT = a.tiles_per_split;
    const int m_begin = split * KT * 64;
    rn::u32x4 sg[GV], sg2[GV], sx[XV];
    uint32_t sgb[GV];
    bool gval[GV], xval[XV];
    auto issue = [&](const int kt) {
        const int t0 = m_begin + kt * 64;
#pragma unroll
        for (int i = 0; i < GV; ++i) {
            const int m = t0 + gr0 + i * GRS;
            gval[i] = m < a.M;
            const int64_t e = (int64_t)(gval[i] ? m : 0) * a.N + n0 + gc_ * 8;
            sg[i] = *(const rn::u32x4 *)(a.G + e);
            sgb[i] = 0xffu;
            if (PROG == PRO_BN_BWD) {
                sg2[i] = *(const rn::u32x4 *)(a.G2 + e);
                if (a.g_relu_mode == 3) sgb[i] = a.gbits[e >> 3];
            }
        }
#pragma unroll
        for (int i = 0; i < XV; ++i) {
            const int m = t0 + xr0 + i * XRS;
            bool v = m < a.M;
            int64_t row = v ? m : 0;
            if (!plain) {
                const int mm = (int)row, hw = a.Ho * a.Wo;
                const int n = mm / hw, rem = mm - n * hw, ho = rem / a.Wo, wo = rem - ho * a.Wo;
                const int y = ho * a.stride - a.pad + dy, x = wo * a.stride - a.pad + dx;
                v = v && y >= 0 && y < a.H && x >= 0 && x < a.W_;
                row = v ? (int64_t)n * a.H * a.W_ + y * a.W_ + x : 0;
            }
            xval[i] = v;
            sx[i] = *(const rn::u32x4 *)(a.X + row * a.Cin + c0 + xc_ * 8);
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int i = 0; i < GV; ++i) {
            const int row = gr0 + i * GRS;
            *(rn::u32x4 *)(lds + row * G_ROWB + ((gc_ ^ tr_swz<TN>(row)) << 4)) = transform<DT, PROG>(sg[i], sg2[i], sgb[i], gcoef, a.g_relu_mode, gval[i]);
        }
#pragma unroll
        for (int i = 0; i < XV; ++i) {
            const int row = xr0 + i * XRS;
            *(rn::u32x4 *)(lds + G_TILE + row * X_ROWB + ((xc_ ^ tr_swz<TK>(row)) << 4)) = transform<DT, PROX>(sx[i], sx[i], 0xffu, xcoef, 0, xval[i]);
        }
    };
    // transposing fragment reads through the compiler's builtin (it then tracks their lgkmcnt itself; an inline-asm read is
    // invisible to the register allocator's copies, which may run before a hand-placed wait)
    typedef __attribute__((ext_vector_type(4))) short s16x4;
    typedef __attribute__((address_space(3))) s16x4 *lds_s16x4;
    auto tr_frag = [&](const uint32_t off, const uint32_t rowb) {     // 8-deep k fragment = positions +0..3 and +4..7
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(lds + off));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(lds + off + 4 * rowb));
        return __builtin_bit_cast(typename rn::mma<DT>::frag, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };
    if (m_begin < a.M) {
        issue(0);
        for (int kt = 0; kt < KT && m_begin + kt * 64 < a.M; ++kt) {
            commit();
            __syncthreads();
            if (kt + 1 < KT && m_begin + (kt + 1) * 64 < a.M) issue(kt + 1);      // in flight under the MFMAs
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                typename rn::mma<DT>::frag fg[MI], fx[NI];
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) fg[mi] = tr_frag(g_off[mi] + kk * 16 * G_ROWB, G_ROWB);
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) fx[ni] = tr_frag(x_off[ni] + kk * 16 * X_ROWB, X_ROWB);
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
                        acc[mi][ni] = rn::mma<DT>::m32(fg[mi], fx[ni], acc[mi][ni]);
            }
            __syncthreads();
        }
    }
    const int Ktot = a.taps * a.Cin;
    float *__restrict__ out = a.partial + (int64_t)split * a.N * Ktot;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int col = c0 + wn * (TK / 2) + ni * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = n0 + wm * (TN / 2) + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                out[(int64_t)row * Ktot + tap * a.Cin + col] = acc[mi][ni][r];
            }
        }
}

// ------------------------------------------------------------------------------------------------------------------
// conv3 of a bottleneck (1x1, Cm -> C4 = 4 Cm channels; retinanet/backbone.py:114,131-132), BOTH of its gradients in one
// pass over the block-output gradient.  pw_gemm_kernel<.., PRO_BN_BWD, EPI_RELU_BWD> (data gradient) and
// pw_wgrad_kernel<.., PRO_BN_BWD, PRO_AFFINE_RELU> (weight gradient) each stream g_out, z3 and the ReLU bits -- the block's
// largest tensors -- through the same bn3-backward transform; the two launches are HBM-bound on exactly those bytes (layer1:
// 142 + 117 us for 722 + 620 MB).  Here a workgroup forms each transformed tile dz3 [128 positions][64 channels] ONCE in LDS
// and uses it twice: row-major fragments for  dy2[m][cm] += dz3[m][c4] . W3[c4][cm]  and transposing reads of the same tile
// for  dW3[c4][cm] += dz3[m][c4] . a2[m][cm]  (a2 = relu(bn2(z2)), formed once per row tile from the z2 rows the ReLU-backward
// epilogue needs anyway).  The dW3 accumulators (C4 x Cm f32 per workgroup) stay in registers across all the row tiles a
// workgroup walks; one partial per workgroup, summed by pw_wgrad_reduce_many_kernel.
// Tile: 128 positions, all Cm output columns; 4 waves.  Cm = 64: 96 accumulator registers, two workgroups per CU.  Cm = 128: the
// C4 x Cm = 512 x 128 weight-gradient accumulators alone are 256 registers per lane, so the kernel runs ONE wave per SIMD on the
// full 512-register budget (accumulators in AGPRs); it is bandwidth-bound and issues its loads a K-chunk ahead, the matrix pipe
// has a 4 x margin.  LDS image of the dz3 tile: 128-byte rows, 16-byte chunks XOR-swizzled with PW_SWZ2(row) -- bits (1, 2, 3) of
// the row as bits (2, 0, 1): a bijection of (row >> 1) & 7 (ds_read_b128 fragments conflict-free, as PW_SWZ) whose bit 2
// separates rows q and q + 2 (the 4-row blocks of ds_read_b64_tr_b16 conflict-free, as tr_swz).
struct PairArgs {
    const uint16_t *G, *Z3;         // [M][C4]: block-output gradient, bn3 input
    const uint8_t *gbits;           // [M][C4 / 8]: ReLU bits of the block output
    const float *pa, *pb, *pc;      // [C4]: dz3 = pa * g' + pc * z3 + pb
    const uint16_t *Wt;             // [Cm][C4]: the data-gradient weight (w3 transposed)
    const uint16_t *Z2;             // [M][Cm]: bn2 input
    const float *ea, *eb, *emean, *einv;    // [Cm]: bn2's forward coefficients and statistics
    uint16_t *Y;                    // [M][Cm]: dy2 * [a2 alive]
    float *partial_bn;              // [gx][2][Cm]: sums of Y and Y * xhat2
    float *partial_w;               // [gx][C4][Cm]
    int M, f16;
    int cm, halves, nwalk;          // cm: row length of Wt's row index range / Z2 / Y / the partials (the conv's Cm); halves = cm / 64 workgroups
                                    // share a row tile, each owning 64 of its columns; nwalk row-tile walkers (a multiple of 8 when halves == 2)
};
#define PW_SWZ2(row) (((((row) >> 1) & 1) << 2) | (((row) >> 2) & 3))

template <int DT, int CM, int KT, int T>
__global__ __launch_bounds__(T, 2) void pw_conv3_bwd_kernel(const PairArgs a)
{
    constexpr int NW = T / 64, C4 = KT * 64;
    constexpr int WM = NW / 2, MI = 4 / WM, NI = CM / 64;        // data gradient: WM x 2 waves of (32 MI) x (CM / 2)
    // weight gradient: a K-chunk's [64 c4][CM] block is NTILES tiles of 32 x 32, one per wave; with twice as many waves as tiles the
    // two wave groups take the even / the odd chunks (accumulators: KT / GROUPS tiles per wave, nothing held twice)
    constexpr int NTILES = CM / 16, GROUPS = NW / NTILES, KW = KT / GROUPS;
    static_assert(NW == NTILES * GROUPS && KT == KW * GROUPS && MI * WM == 4 && (KT & (KT - 1)) == 0 && (KT / 2) % GROUPS == 0, "wave roles");
    constexpr int A_TILE = 128 * 128, W_TILE = CM * 128, STAGE = A_TILE + W_TILE;
    constexpr int AV = 1024 / T, ARS = T / 8, WV = CM * 8 / T;   // staged 16-byte vectors per thread (rows r0 + ARS i)
    constexpr int CG = CM / 8, RL = T / CG, EROWS = 128 / RL;    // epilogue: thread = 8 fixed columns, rows erl + RL i
    constexpr int LDS_MAIN = (2 * STAGE > 128 * CM * 4) ? 2 * STAGE : 128 * CM * 4;
    constexpr int XROWB = CM * 2, X2_OFF = LDS_MAIN, COEF_OFF = X2_OFF + 128 * XROWB;
    static_assert(WV * ARS == CM && EROWS * RL == 128 && AV * ARS == 128, "tile split");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = tid & 7, r0 = tid >> 3;
    const int MT = (a.M + 127) / 128;
    // halves == 2: the two workgroups of a row tile are 8 apart in launch order -- same XCD (workgroups go round-robin over the 8
    // XCDs), dispatched back to back -- so the second one's reads of g / z3 / bits find the first one's lines in that XCD's L2
    const int bid = blockIdx.x;
    const int half = a.halves == 2 ? (bid >> 3) & 1 : 0, walker = a.halves == 2 ? ((bid >> 4) << 3) | (bid & 7) : bid;
    const int n0 = half * CM, cmr = a.cm;
    // ... and walks the K-chunks half a turn ahead of it (chunk (kt + krot) % KT at step kt): the pair then has two DIFFERENT chunks in
    // flight -- with the same one, the pair's loads merge in L2 and two CUs have the bytes in flight of one (134 us at the layer2 shape,
    // the latency x concurrency bound).  The column halves of dy2 thus sum their K-chunks in different orders (deterministic; the first
    // half in pw_gemm_kernel's order).
    const int krot = half * (KT / 2);
    float *const s_coef = (float *)(lds + COEF_OFF);             // [pa | pb | pc][C4], then [ea | eb | emean | einv][CM]
    float *const s_epi = s_coef + 3 * C4;
    for (int q = tid; q < 3 * C4; q += T) {
        const int which = q / C4, ch = q - which * C4;
        s_coef[q] = (which == 0 ? a.pa : (which == 1 ? a.pb : a.pc))[ch];
    }
    for (int q = tid; q < 4 * CM; q += T) {
        const int which = q / CM, col = q - which * CM;
        s_epi[q] = (which == 0 ? a.ea : (which == 1 ? a.eb : (which == 2 ? a.emean : a.einv)))[n0 + col];
    }
    __syncthreads();

    // staging slots
    uint32_t wa_off[AV], ww_off[WV];
#pragma unroll
    for (int i = 0; i < AV; ++i) { const int row = r0 + ARS * i; wa_off[i] = row * 128 + ((c ^ PW_SWZ2(row)) << 4); }
#pragma unroll
    for (int i = 0; i < WV; ++i) { const int row = r0 + ARS * i; ww_off[i] = A_TILE + row * 128 + ((c ^ PW_SWZ(row)) << 4); }
    // data-gradient fragments (row-major ds_read_b128)
    const int wm = wave >> 1, wn = wave & 1;
    uint32_t a_off[4], b_off[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        const int chunk = kk * 2 + (lane >> 5);
        { const int row = wm * (32 * MI) + (lane & 31); a_off[kk] = row * 128 + ((chunk ^ PW_SWZ2(row)) << 4); }
        { const int row = wn * (CM / 2) + (lane & 31); b_off[kk] = A_TILE + row * 128 + ((chunk ^ PW_SWZ(row)) << 4); }
    }
    // weight-gradient fragments (transposing reads; lane = 16 grp + 4 q + p as in pw_wgrad_kernel): wave = one 32 x 32 tile of the
    // [64 c4][CM] block of a K-chunk -- rows wr * 32 .., columns wc * 32 ..
    const int wk = wave / NTILES, wtile = wave % NTILES, wr = wtile & 1, wc = wtile >> 1;    // rows wr * 32 .., columns wc * 32 ..
    const int grp = lane >> 4, q4 = (lane >> 2) & 3, p4 = lane & 3;
    uint32_t ga_lo, ga_hi, gx_lo;
    {
        const int rl = 8 * (grp >> 1) + q4, ch = wr * 4 + 2 * (grp & 1) + (p4 >> 1), chx = wc * 4 + 2 * (grp & 1) + (p4 >> 1);
        ga_lo = (uint32_t)(rl * 128 + ((ch ^ PW_SWZ2(rl)) << 4) + (p4 & 1) * 8);
        ga_hi = (uint32_t)((rl + 4) * 128 + ((ch ^ PW_SWZ2(rl + 4)) << 4) + (p4 & 1) * 8);
        gx_lo = (uint32_t)(X2_OFF + rl * XROWB + ((chx ^ tr_swz<CM>(q4)) << 4) + (p4 & 1) * 8);
    }
    const int ecg = tid % CG, erl = tid / CG;
    float ssum[8], qsum[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { ssum[j] = 0.0f; qsum[j] = 0.0f; }
    f32x16 accw[KW];
#pragma unroll
    for (int k = 0; k < KW; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) accw[k][r] = 0.0f;

    // DEPTH register sets of staged chunks: chunk kt of every tile lives in set kt % DEPTH and goes to LDS stage kt & 1; the loads of
    // chunk kt + DEPTH are issued when chunk kt's MFMAs start.  The kernel is bound by load latency x bytes in flight: at the layer1
    // shape (KT = 4) two chunks ahead take it from 149 to 140 us, the stream rate of its 722 MB; the KT = 8 form has no registers for
    // a second set (11 - 33 VGPRs spilled, 121 -> 140 us) and stays one chunk ahead.  `set` is a literal after unrolling.
    constexpr int DEPTH = KT == 4 ? 2 : 1;
    rn::u32x4 sx[DEPTH][AV], sz[DEPTH][AV], sw[DEPTH][WV], er[EROWS];
    uint32_t sbits[DEPTH][AV];
    bool sval[DEPTH][AV];
    int skt[DEPTH] = {};
    auto issue = [&](const int set, const int m0, const int kt) {
#pragma unroll
        for (int i = 0; i < AV; ++i) {
            const int m = m0 + r0 + ARS * i;
            sval[set][i] = m < a.M;
            const int64_t e = (int64_t)(sval[set][i] ? m : 0) * C4 + kt * 64 + c * 8;
            sx[set][i] = *(const rn::u32x4 *)(a.G + e);
            sz[set][i] = *(const rn::u32x4 *)(a.Z3 + e);
            sbits[set][i] = a.gbits[e >> 3];
        }
#pragma unroll
        for (int i = 0; i < WV; ++i) sw[set][i] = *(const rn::u32x4 *)(a.Wt + (int64_t)(n0 + r0 + ARS * i) * C4 + kt * 64 + c * 8);
        skt[set] = kt;
    };
    auto commit = [&](const int set, const int stage) {
        unsigned char *const sb = lds + stage * STAGE;
        ProCoef coef;
        const int ch = skt[set] * 64 + c * 8;
        ld8f(s_coef + ch, coef.a); ld8f(s_coef + C4 + ch, coef.b); ld8f(s_coef + 2 * C4 + ch, coef.c);
#pragma unroll
        for (int i = 0; i < AV; ++i) {
            *(rn::u32x4 *)(sb + wa_off[i]) = transform<DT, PRO_BN_BWD>(sx[set][i], sz[set][i], sbits[set][i], coef, 3, sval[set][i]);
            __builtin_amdgcn_sched_barrier(0);                      // one vector's 16 floats live at a time (registers)
        }
#pragma unroll
        for (int i = 0; i < WV; ++i) *(rn::u32x4 *)(sb + ww_off[i]) = sw[set][i];
    };
    auto load_z2 = [&](const int m0) {
#pragma unroll
        for (int i = 0; i < EROWS; ++i) {
            const int m = m0 + erl + i * RL, mc = m < a.M ? m : a.M - 1;
            er[i] = *(const rn::u32x4 *)(a.Z2 + (int64_t)mc * cmr + n0 + ecg * 8);
        }
    };
    auto commit_a2 = [&]() {                                        // a2 = relu(fma(z2, ea, eb)) rounded to DT: the X operand of the weight gradient
        ProCoef k;
        ld8f(s_epi + ecg * 8, k.a); ld8f(s_epi + CM + ecg * 8, k.b);
#pragma unroll
        for (int i = 0; i < EROWS; ++i) {
            const int row = erl + i * RL;
            *(rn::u32x4 *)(lds + X2_OFF + row * XROWB + ((ecg ^ tr_swz<CM>(row)) << 4)) = transform<DT, PRO_AFFINE_RELU>(er[i], er[i], 0xffu, k, 0, true);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    typedef __attribute__((ext_vector_type(4))) short s16x4;
    typedef __attribute__((address_space(3))) s16x4 *lds_s16x4;
    auto tr_pair = [&](const uint32_t lo_off, const uint32_t hi_off) {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(lds + lo_off));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(lds + hi_off));
        return __builtin_bit_cast(typename rn::mma<DT>::frag, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };
    const float alive = alive_dt<DT>();

    int mt = walker;
    if (mt < MT) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) issue(d, mt * 128, (krot + d) & (KT - 1));
        load_z2(mt * 128);
    }
    for (; mt < MT; mt += a.nwalk) {
        const int m0 = mt * 128;
        f32x16 accd[MI][NI];
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) accd[i][j][r] = 0.0f;
        commit(0, 0);
        __syncthreads();
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            if (kt + DEPTH < KT) issue(kt % DEPTH, m0, (kt + DEPTH + krot) & (KT - 1));
            else if (mt + a.nwalk < MT) issue(kt % DEPTH, (mt + a.nwalk) * 128, (kt + DEPTH + krot) & (KT - 1));   // the next row tile's first chunks
            const uint32_t sb = (uint32_t)((kt & 1) * STAGE);
            typedef typename rn::mma<DT>::frag frag8;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                frag8 fa[MI], fb[NI];
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) fa[mi] = *(const frag8 *)(lds + sb + a_off[kk] + mi * 4096);
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) fb[ni] = *(const frag8 *)(lds + sb + b_off[kk] + ni * 4096);
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
                        accd[mi][ni] = rn::mma<DT>::m32(fa[mi], fb[ni], accd[mi][ni]);
            }
            if (kt == 0) {                                          // z2 rows (loaded behind the previous epilogue) -> a2 tile, under the MFMAs above
                commit_a2();
                __syncthreads();
            }
            __builtin_amdgcn_sched_barrier(0);
            if (GROUPS == 1 || (kt % GROUPS) == wk) {              // (wave-uniform)
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) {                   // contraction over the tile's 128 positions, 16 per step
                    const frag8 fg = tr_pair(sb + ga_lo + kk * 2048, sb + ga_hi + kk * 2048);
                    const frag8 fx = tr_pair(gx_lo + kk * 16 * XROWB, gx_lo + kk * 16 * XROWB + 4 * XROWB);
                    accw[kt / GROUPS] = rn::mma<DT>::m32(fg, fx, accw[kt / GROUPS]);
                    if (kk & 1) __builtin_amdgcn_sched_barrier(0); // (keeps the fragment reads of all 8 steps from being hoisted: registers)
                }
            }
            if (kt + 1 < KT) commit((kt + 1) % DEPTH, (kt + 1) & 1);
            __syncthreads();
        }

        // ---- epilogue (pw_gemm_kernel's EPI_RELU_BWD): dy2 = acc * [a2 alive], its two bn2-backward sums
        float *const tile = (float *)lds;                           // [128][CM]
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                const int col = wn * (CM / 2) + ni * 32 + (lane & 31);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = wm * (32 * MI) + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    tile[row * CM + col] = accd[mi][ni][r];
                }
            }
        __syncthreads();
        {
            float ea[8], eb[8], emu[8], eis[8];
            ld8f(s_epi + ecg * 8, ea); ld8f(s_epi + CM + ecg * 8, eb);
            ld8f(s_epi + 2 * CM + ecg * 8, emu); ld8f(s_epi + 3 * CM + ecg * 8, eis);
#pragma unroll
            for (int i = 0; i < EROWS; ++i) {
                const int row = erl + i * RL, m = m0 + row;
                if (m < a.M) {
                    float v[8], z[8];
                    ld8f(tile + row * CM + ecg * 8, v);
                    rn::u32x4 o = rn::dt<DT>::pack(v);
                    rn::dt<DT>::unpack(o, v);                      // the statistics are those of the stored (rounded) tensor
                    rn::dt<DT>::unpack(er[i], z);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        if (!(fmaf(z[j], ea[j], eb[j]) > alive)) v[j] = 0.0f;
                        ssum[j] += v[j];
                        qsum[j] = fmaf(v[j], (z[j] - emu[j]) * eis[j], qsum[j]);
                    }
                    o = rn::dt<DT>::pack(v);
                    *(rn::u32x4 *)(a.Y + (int64_t)m * cmr + n0 + ecg * 8) = o;
                }
            }
        }
        if (mt + a.nwalk < MT) load_z2((mt + a.nwalk) * 128);      // consumed by the next tile's commit_a2 and epilogue
        __syncthreads();                                            // the tile is the next row tile's staging area
    }

    {   // bn2-backward partial sums of the rows this workgroup walked
        float *const red = (float *)lds;                            // [RL][2][CM]
#pragma unroll
        for (int j = 0; j < 8; ++j) { red[(erl * 2 + 0) * CM + ecg * 8 + j] = ssum[j]; red[(erl * 2 + 1) * CM + ecg * 8 + j] = qsum[j]; }
        __syncthreads();
        for (int q = tid; q < 2 * CM; q += T) {
            float t = 0.0f;
            for (int l = 0; l < RL; ++l) t += red[l * 2 * CM + q];
            const int which = q >= CM ? 1 : 0;
            a.partial_bn[((int64_t)walker * 2 + which) * cmr + n0 + (q - which * CM)] = t;
        }
    }
    float *__restrict__ outw = a.partial_w + (int64_t)walker * C4 * cmr + n0;
#pragma unroll
    for (int k = 0; k < KW; ++k) {
        const int col = wc * 32 + (lane & 31);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = ((k * GROUPS + wk + krot) & (KT - 1)) * 64 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            outw[row * cmr + col] = accw[k][r];
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// The end of one bottleneck and the start of the next in one pass (forward): the block output
//     y = relu(bn3(z3) + identity)          (norm.hip's bn_apply_kernel<RELU, RES[, RA]>; retinanet/backbone.py:132-136)
// is formed chunk by chunk (128 positions x 64 channels) from z3 and the identity, written to memory with its ReLU bits -- and, still
// in LDS, multiplied into the NEXT block's conv1 (1x1, C4 -> CN channels; backbone.py:118), whose output z1 and bn1 statistics
// partials leave from the epilogue (pw_gemm_kernel<.., PRO_NONE, EPI_STATS> on y).  The separate launches write y (275 MB at layer1)
// and read it straight back; here it is only written.  Same products in the same order as pw_gemm_kernel: z1 is bit-identical.
// 512 threads, a 128-row tile, all CN columns; DEPTH chunks of (z3, identity) in flight in registers.
struct ChainArgs {
    const uint16_t *Z3, *R;         // [M][C4]: bn3 input; the identity (RA: the INPUT of the downsample branch's BatchNorm)
    const float *oa, *ob;           // [C4]: bn3's forward coefficients
    const float *ra, *rb;           // [C4], RA only: the downsample BatchNorm's
    const uint16_t *W1;             // [CN][C4]: the next block's conv1 weight
    uint16_t *Y;                    // [M][C4]
    uint8_t *ybits;                 // [M][C4 / 8]
    uint16_t *Z1;                   // [M][CN]
    float *partial;                 // [gx][2][CN]: column sums / sums of squares of Z1 as stored
    int M, gx, f16;
};

template <int DT, int CN, int KT, bool RA>
__global__ __launch_bounds__(512, 2) void pw_block_out_conv1_kernel(const ChainArgs a)
{
    constexpr int T = 512, C4 = KT * 64;
    constexpr int NI = CN / 64;                                  // 4 x 2 waves of 32 x (CN / 2)
    constexpr int A_TILE = 128 * 128, W_TILE = CN * 128, STAGE = A_TILE + W_TILE;
    constexpr int AV = 1024 / T, ARS = T / 8, WV = CN * 8 / T;
    constexpr int CG = CN / 8, RL = T / CG, EROWS = 128 / RL;
    constexpr int LDS_MAIN = (2 * STAGE > 128 * CN * 4) ? 2 * STAGE : 128 * CN * 4;
    constexpr int NCOEF = RA ? 4 : 2, DEPTH = 2;
    static_assert(WV * ARS == CN && EROWS * RL == 128 && AV * ARS == 128 && KT % DEPTH == 0, "tile split");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = tid & 7, r0 = tid >> 3;
    const int MT = (a.M + 127) / 128;
    float *const s_coef = (float *)(lds + LDS_MAIN);             // [oa | ob | ra | rb][C4]
    for (int q = tid; q < NCOEF * C4; q += T) {
        const int which = q / C4, ch = q - which * C4;
        s_coef[q] = (which == 0 ? a.oa : (which == 1 ? a.ob : (which == 2 ? a.ra : a.rb)))[ch];
    }
    __syncthreads();
    uint32_t wa_off[AV], ww_off[WV];
#pragma unroll
    for (int i = 0; i < AV; ++i) { const int row = r0 + ARS * i; wa_off[i] = row * 128 + ((c ^ PW_SWZ(row)) << 4); }
#pragma unroll
    for (int i = 0; i < WV; ++i) { const int row = r0 + ARS * i; ww_off[i] = A_TILE + row * 128 + ((c ^ PW_SWZ(row)) << 4); }
    const int wm = wave >> 1, wn = wave & 1;
    uint32_t a_off[4], b_off[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        const int chunk = kk * 2 + (lane >> 5);
        { const int row = wm * 32 + (lane & 31); a_off[kk] = row * 128 + ((chunk ^ PW_SWZ(row)) << 4); }
        { const int row = wn * (CN / 2) + (lane & 31); b_off[kk] = A_TILE + row * 128 + ((chunk ^ PW_SWZ(row)) << 4); }
    }
    const int ecg = tid % CG, erl = tid / CG;
    float ssum[8], qsum[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { ssum[j] = 0.0f; qsum[j] = 0.0f; }
    const float alive = alive_dt<DT>();

    rn::u32x4 sx[DEPTH][AV], sr[DEPTH][AV], sw[DEPTH][WV];
    int srow[DEPTH][AV], skt[DEPTH] = {};                        // (row < 0: past the end)
    auto issue = [&](const int set, const int m0, const int kt) {
#pragma unroll
        for (int i = 0; i < AV; ++i) {
            const int m = m0 + r0 + ARS * i;
            srow[set][i] = m < a.M ? m : -1;
            const int64_t e = (int64_t)(m < a.M ? m : 0) * C4 + kt * 64 + c * 8;
            sx[set][i] = *(const rn::u32x4 *)(a.Z3 + e);
            sr[set][i] = *(const rn::u32x4 *)(a.R + e);
        }
#pragma unroll
        for (int i = 0; i < WV; ++i) sw[set][i] = *(const rn::u32x4 *)(a.W1 + (int64_t)(r0 + ARS * i) * C4 + kt * 64 + c * 8);
        skt[set] = kt;
    };
    auto commit = [&](const int set, const int stage) {          // y chunk -> memory (+ bits) and -> the LDS operand tile
        unsigned char *const sb = lds + stage * STAGE;
        const int ch = skt[set] * 64 + c * 8;
        float oa[8], ob[8], ra[8], rb[8];
        ld8f(s_coef + ch, oa); ld8f(s_coef + C4 + ch, ob);
        if (RA) { ld8f(s_coef + 2 * C4 + ch, ra); ld8f(s_coef + 3 * C4 + ch, rb); }
#pragma unroll
        for (int i = 0; i < AV; ++i) {
            float f[8], r[8];
            rn::dt<DT>::unpack(sx[set][i], f);
            rn::dt<DT>::unpack(sr[set][i], r);
            unsigned bits = 0;
            if (RA) {
                rn::u32x4 rr;
                float t8[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) t8[j] = fmaf(r[j], ra[j], rb[j]);
                rr = rn::dt<DT>::pack(t8);                        // the branch's BatchNorm output as a separate apply pass would have stored it
                rn::dt<DT>::unpack(rr, r);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float t = fmaf(f[j], oa[j], ob[j]) + r[j];
                bits |= (t > alive) ? (1u << j) : 0u;
                f[j] = t > 0.0f ? t : 0.0f;
            }
            rn::u32x4 o = rn::dt<DT>::pack(f);
            const bool ok = srow[set][i] >= 0;
            if (ok) {
                const int64_t e = (int64_t)srow[set][i] * C4 + ch;
                *(rn::u32x4 *)(a.Y + e) = o;
                a.ybits[e >> 3] = (uint8_t)bits;
            } else o = rn::u32x4{0u, 0u, 0u, 0u};
            *(rn::u32x4 *)(sb + wa_off[i]) = o;
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < WV; ++i) *(rn::u32x4 *)(sb + ww_off[i]) = sw[set][i];
    };

    int mt = blockIdx.x;
    if (mt < MT) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) issue(d, mt * 128, d);
    }
    for (; mt < MT; mt += a.gx) {
        const int m0 = mt * 128;
        f32x16 acc[NI];
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
        commit(0, 0);
        __syncthreads();
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            if (kt + DEPTH < KT) issue(kt % DEPTH, m0, kt + DEPTH);
            else if (mt + a.gx < MT) issue(kt % DEPTH, (mt + a.gx) * 128, kt + DEPTH - KT);
            const uint32_t sb = (uint32_t)((kt & 1) * STAGE);
            typedef typename rn::mma<DT>::frag frag8;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const frag8 fa = *(const frag8 *)(lds + sb + a_off[kk]);
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    const frag8 fb = *(const frag8 *)(lds + sb + b_off[kk] + ni * 4096);
                    acc[ni] = rn::mma<DT>::m32(fa, fb, acc[ni]);
                }
            }
            if (kt + 1 < KT) commit((kt + 1) % DEPTH, (kt + 1) & 1);
            __syncthreads();
        }
        // ---- epilogue (pw_gemm_kernel's EPI_STATS): z1 as stored, its column sums / sums of squares
        float *const tile = (float *)lds;                           // [128][CN]
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int col = wn * (CN / 2) + ni * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                tile[row * CN + col] = acc[ni][r];
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < EROWS; ++i) {
            const int row = erl + i * RL, m = m0 + row;
            if (m < a.M) {
                float v[8];
                ld8f(tile + row * CN + ecg * 8, v);
                const rn::u32x4 o = rn::dt<DT>::pack(v);
                rn::dt<DT>::unpack(o, v);
#pragma unroll
                for (int j = 0; j < 8; ++j) { ssum[j] += v[j]; qsum[j] = fmaf(v[j], v[j], qsum[j]); }
                *(rn::u32x4 *)(a.Z1 + (int64_t)m * CN + ecg * 8) = o;
            }
        }
        __syncthreads();
    }
    float *const red = (float *)lds;                                // [RL][2][CN]
#pragma unroll
    for (int j = 0; j < 8; ++j) { red[(erl * 2 + 0) * CN + ecg * 8 + j] = ssum[j]; red[(erl * 2 + 1) * CN + ecg * 8 + j] = qsum[j]; }
    __syncthreads();
    for (int q = tid; q < 2 * CN; q += T) {
        float t = 0.0f;
        for (int l = 0; l < RL; ++l) t += red[l * 2 * CN + q];
        const int which = q >= CN ? 1 : 0;
        a.partial[((int64_t)blockIdx.x * 2 + which) * CN + (q - which * CN)] = t;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// The start of one bottleneck's backward and of the one before it in one pass: conv1's data gradient joined by the identity branch
//     dx = dz1 . W1 + resid * bits            (pw_gemm_kernel<.., PRO_NONE, EPI_RESID>; dx IS the gradient at the previous block's output)
// and, on the tile just stored, the two sums of that previous block's bn3 backward
//     sum g', sum g' * xhat3,  g' = dx * prev_bits,  xhat3 = (prev_z3 - mean) * invstd      (norm.hip's bn_bwd_partial_kernel<DT, 3>)
// which otherwise re-read dx (275 MB at layer1) in a launch of their own.  One 128-column tile of dx per workgroup (its weight rows
// stay in LDS), 512 threads, row tiles walked with the next tile's operands in flight; same products and expressions as the two
// kernels: dx is bit-identical, the sums differ in summation order only.
struct DgradSumsArgs {
    const uint16_t *A;              // dz1 [M][CM]
    const uint16_t *Wt;             // [C4][CM]: conv1's data-gradient weight (w1 transposed)
    const uint16_t *R;              // the identity branch's gradient [M][C4] (rs == 2: on the stride-2 grid, see PwArgs)
    const uint8_t *rbits;           // [M][C4 / 8] or null (no mask)
    int rs, rH, rW;
    const uint16_t *Zp;             // previous block: bn3 input [M][C4]
    const uint8_t *pbits;           //                 ReLU bits of its output [M][C4 / 8]
    const float *pmean, *pinv;      //                 bn3 batch statistics [C4]
    uint16_t *Y;                    // dx [M][C4]
    float *partial;                 // [gx][2][C4]
    int M, C4, gx, f16;
    // FWD (conv3 of a bottleneck, forward): A = z2 [M][CM] with relu(fma(z2, fa, fb)) in the operand load (bn2's apply + ReLU), Wt = w3 [C4][CM],
    // Y = z3, partial = column sums / sums of squares of z3 as stored (pw_gemm_kernel<.., PRO_AFFINE_RELU, EPI_STATS>); no epilogue operands
    const float *fa, *fb;           // [CM]
};

template <int DT, int KC, bool FWD>
__global__ __launch_bounds__(512, 2) void pw_dgrad_sums_kernel(const DgradSumsArgs a)
{
    constexpr int T = 512, CM = KC * 64, BN = 128;
    constexpr int A_BYTES = KC * 128 * 128, B_BYTES = KC * BN * 128, TILE_OFF = A_BYTES + B_BYTES;     // [A | B | f32 tile 128 x 128]
    constexpr int AV = 2;                                        // 16-byte vectors per thread and 64-channel chunk (rows r0, r0 + 64)
    constexpr int CG = BN / 8, RL = T / CG, EROWS = 128 / RL;    // 16 column groups, 32 row lanes, 4 rows per thread
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = tid & 7, r0 = tid >> 3;
    const int n0 = blockIdx.y * BN;
    const int MT = (a.M + 127) / 128;
    // this column tile's weight rows, once: chunk kc of row r at B + kc * BN * 128 + r * 128, 16-byte pieces swizzled as PW_SWZ
    for (int q = tid; q < KC * BN * 8; q += T) {
        const int kc = q / (BN * 8), rem = q - kc * BN * 8, row = rem >> 3, cc = rem & 7;
        *(rn::u32x4 *)(lds + A_BYTES + kc * BN * 128 + row * 128 + ((cc ^ PW_SWZ(row)) << 4)) =
            *(const rn::u32x4 *)(a.Wt + (int64_t)(n0 + row) * CM + kc * 64 + cc * 8);
    }
    uint32_t wa_off[AV];
#pragma unroll
    for (int i = 0; i < AV; ++i) { const int row = r0 + 64 * i; wa_off[i] = row * 128 + ((c ^ PW_SWZ(row)) << 4); }
    const int wm = wave >> 1, wn = wave & 1;                     // 4 x 2 waves of 32 x 64
    uint32_t a_off[4], b_off[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        const int chunk = kk * 2 + (lane >> 5);
        { const int row = wm * 32 + (lane & 31); a_off[kk] = row * 128 + ((chunk ^ PW_SWZ(row)) << 4); }
        { const int row = wn * 64 + (lane & 31); b_off[kk] = A_BYTES + row * 128 + ((chunk ^ PW_SWZ(row)) << 4); }
    }
    const int ecg = tid % CG, erl = tid / CG;
    float ssum[8], qsum[8], mu[8], is[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { ssum[j] = 0.0f; qsum[j] = 0.0f; mu[j] = 0.0f; is[j] = 0.0f; }
    if (!FWD) { ld8f(a.pmean + n0 + ecg * 8, mu); ld8f(a.pinv + n0 + ecg * 8, is); }
    ProCoef acoef[KC];                                            // FWD: bn2's coefficients of this thread's 8 channels of every chunk
    if (FWD) {
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) { ld8f(a.fa + kc * 64 + c * 8, acoef[kc].a); ld8f(a.fb + kc * 64 + c * 8, acoef[kc].b); }
    }

    // two register sets of epilogue operands: the NEXT tile's are requested before this tile's epilogue runs (the tile loop is unrolled by
    // two through a generic lambda so that the set is a literal)
    rn::u32x4 sa[KC][AV], er[2][EROWS], ez[2][EROWS];
    uint32_t eb[2][EROWS], ezb[2][EROWS];
    bool sval[AV] = {false, false};                               // (FWD: rows past the end must stay zero THROUGH the transform)
    auto issue_a = [&](const int m0) {
#pragma unroll
        for (int kc = 0; kc < KC; ++kc)
#pragma unroll
            for (int i = 0; i < AV; ++i) {
                const int m = m0 + r0 + 64 * i;
                const rn::u32x4 zero4 = {0u, 0u, 0u, 0u};
                sval[i] = m < a.M;
                sa[kc][i] = m < a.M ? *(const rn::u32x4 *)(a.A + (int64_t)m * CM + kc * 64 + c * 8) : zero4;
            }
    };
    auto commit_a = [&]() {
#pragma unroll
        for (int kc = 0; kc < KC; ++kc)
#pragma unroll
            for (int i = 0; i < AV; ++i)
                *(rn::u32x4 *)(lds + kc * 128 * 128 + wa_off[i]) = FWD ? transform<DT, PRO_AFFINE_RELU>(sa[kc][i], sa[kc][i], 0xffu, acoef[kc], 0, sval[i]) : sa[kc][i];
    };
    auto load_epi = [&](auto set_c, const int m0) {              // a tile's epilogue operands (pw_gemm_kernel's EPI_RESID addressing)
        constexpr int S = decltype(set_c)::value;
        if (FWD) return;
#pragma unroll
        for (int i = 0; i < EROWS; ++i) {
            const int m = m0 + erl + i * RL, mc = m < a.M ? m : a.M - 1;
            const int64_t e0 = (int64_t)mc * a.C4 + n0 + ecg * 8;
            int64_t e = e0;
            bool on = true;
            if (a.rs == 2) {
                const int hw = a.rH * a.rW, n = mc / hw, rem = mc - n * hw, y = rem / a.rW, x = rem - y * a.rW;
                on = ((y | x) & 1) == 0;
                e = (((int64_t)n * ((a.rH + 1) >> 1) + (y >> 1)) * ((a.rW + 1) >> 1) + (x >> 1)) * a.C4 + n0 + ecg * 8;
            }
            const rn::u32x4 zero4 = {0u, 0u, 0u, 0u};
            er[S][i] = on ? *(const rn::u32x4 *)(a.R + e) : zero4;
            eb[S][i] = on ? (a.rbits ? (uint32_t)a.rbits[e >> 3] : 0xffu) : 0u;
            ez[S][i] = *(const rn::u32x4 *)(a.Zp + e0);
            ezb[S][i] = a.pbits[e0 >> 3];
        }
    };

    int mt = blockIdx.x;
    if (mt < MT) { issue_a(mt * 128); load_epi(std::integral_constant<int, 0>{}, mt * 128); }
    __syncthreads();                                               // (the weight rows)
    auto row_tile = [&](auto set_c) {
        constexpr int S = decltype(set_c)::value;
        const int m0 = mt * 128;
        commit_a();
        __syncthreads();
        if (mt + a.gx < MT) {                                       // the next tile's operand rows and epilogue operands: in flight under this tile
            issue_a((mt + a.gx) * 128);
            load_epi(std::integral_constant<int, 1 - S>{}, (mt + a.gx) * 128);
        }
        f32x16 acc[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
        typedef typename rn::mma<DT>::frag frag8;
#pragma unroll
        for (int kc = 0; kc < KC; ++kc)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const frag8 fa = *(const frag8 *)(lds + kc * 128 * 128 + a_off[kk]);
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    const frag8 fb = *(const frag8 *)(lds + kc * BN * 128 + b_off[kk] + ni * 4096);
                    acc[ni] = rn::mma<DT>::m32(fa, fb, acc[ni]);
                }
            }
        float *const tile = (float *)(lds + TILE_OFF);              // [128][128]
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int col = wn * 64 + ni * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                tile[row * BN + col] = acc[ni][r];
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < EROWS; ++i) {
            const int row = erl + i * RL, m = m0 + row;
            if (m < a.M) {
                float v[8], r[8], z[8];
                ld8f(tile + row * BN + ecg * 8, v);
                if (!FWD) {
                    rn::dt<DT>::unpack(er[S][i], r);
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] += ((eb[S][i] >> j) & 1u) ? r[j] : 0.0f;
                }
                const rn::u32x4 o = rn::dt<DT>::pack(v);
                *(rn::u32x4 *)(a.Y + (int64_t)m * a.C4 + n0 + ecg * 8) = o;
                rn::dt<DT>::unpack(o, v);                          // the sums are those of the stored tensor
                if (FWD) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) { ssum[j] += v[j]; qsum[j] = fmaf(v[j], v[j], qsum[j]); }
                } else {
                    rn::dt<DT>::unpack(ez[S][i], z);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float gj = ((ezb[S][i] >> j) & 1u) ? v[j] : 0.0f;
                        ssum[j] += gj;
                        qsum[j] = fmaf(gj, (z[j] - mu[j]) * is[j], qsum[j]);
                    }
                }
            }
        }
        __syncthreads();                                            // A and the tile are rewritten by the next row tile
    };
    while (mt < MT) {
        row_tile(std::integral_constant<int, 0>{});
        mt += a.gx;
        if (mt >= MT) break;
        row_tile(std::integral_constant<int, 1>{});
        mt += a.gx;
    }
    float *const red = (float *)(lds + TILE_OFF);                   // [RL][2][BN]
#pragma unroll
    for (int j = 0; j < 8; ++j) { red[(erl * 2 + 0) * BN + ecg * 8 + j] = ssum[j]; red[(erl * 2 + 1) * BN + ecg * 8 + j] = qsum[j]; }
    __syncthreads();
    for (int q = tid; q < 2 * BN; q += T) {
        float t = 0.0f;
        for (int l = 0; l < RL; ++l) t += red[l * 2 * BN + q];
        const int which = q >= BN ? 1 : 0;
        a.partial[((int64_t)blockIdx.x * 2 + which) * a.C4 + n0 + (q - which * BN)] = t;
    }
}

// dW (bf16) = sum over the splits of partial (f32).  A block owns 32 float4 outputs; its 8 thread rows each sum every 8th
// split (8 loads in flight per output instead of one serial chain over S), then the 8 sums are combined in a fixed order.
template <int DT>
__global__ __launch_bounds__(256) void pw_wgrad_reduce_kernel(const float *__restrict__ partial, const int S, const int64_t n4, uint16_t *__restrict__ dw)
{
    __shared__ rn::f32x4 sh[8][32];
    const int j = threadIdx.x & 31, slice = threadIdx.x >> 5;
    const int64_t i = (int64_t)blockIdx.x * 32 + j;
    rn::f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (i < n4)
        for (int sp = slice; sp < S; sp += 8) {
            const rn::f32x4 v = ((const rn::f32x4 *)partial)[sp * n4 + i];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    sh[slice][j] = s;
    __syncthreads();
    if (slice == 0 && i < n4) {
        rn::f32x4 t = sh[0][j];
#pragma unroll
        for (int l = 1; l < 8; ++l) { const rn::f32x4 v = sh[l][j]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
        rn::u32x2 o;
        o.x = rn::dt<DT>::pk(t.x, t.y); o.y = rn::dt<DT>::pk(t.z, t.w);
        ((rn::u32x2 *)dw)[i] = o;
    }
}

// The same reduction for up to 8 weight gradients in one launch (blockIdx.y = gradient): a fused bottleneck's three or four
// weight gradients are summed once at the end of its backward instead of behind each kernel (8 us of launch-bound work each).
constexpr int RED_MAX = 8;
struct ReduceTable { const float *partial[RED_MAX]; uint16_t *dw[RED_MAX]; int S[RED_MAX]; int64_t n4[RED_MAX]; };
template <int DT>
__global__ __launch_bounds__(256) void pw_wgrad_reduce_many_kernel(const ReduceTable t)
{
    __shared__ rn::f32x4 sh[8][32];
    const int it = blockIdx.y;
    const float *__restrict__ partial = t.partial[it];
    const int S = t.S[it];
    const int64_t n4 = t.n4[it];
    const int j = threadIdx.x & 31, slice = threadIdx.x >> 5;
    for (int64_t i0 = (int64_t)blockIdx.x * 32; i0 < n4; i0 += (int64_t)gridDim.x * 32) {      // (uniform trip count per block)
        const int64_t i = i0 + j;
        rn::f32x4 s = {0.f, 0.f, 0.f, 0.f};
        if (i < n4)
            for (int sp = slice; sp < S; sp += 8) {
                const rn::f32x4 v = ((const rn::f32x4 *)partial)[sp * n4 + i];
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            }
        sh[slice][j] = s;
        __syncthreads();
        if (slice == 0 && i < n4) {
            rn::f32x4 u = sh[0][j];
#pragma unroll
            for (int l = 1; l < 8; ++l) { const rn::f32x4 v = sh[l][j]; u.x += v.x; u.y += v.y; u.z += v.z; u.w += v.w; }
            rn::u32x2 o;
            o.x = rn::dt<DT>::pk(u.x, u.y); o.y = rn::dt<DT>::pk(u.z, u.w);
            ((rn::u32x2 *)t.dw[it])[i] = o;
        }
        __syncthreads();
    }
}

int cu_count()
{
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    return cus;
}

int walkers(const int M)
{
    const int MT = (M + PW_BM - 1) / PW_BM;
    if (MT <= PW_MAX_GX) return MT;
    const int rounds = (MT + PW_MAX_GX - 1) / PW_MAX_GX;
    return (MT + rounds - 1) / rounds;
}

template <int DT, int BN, int PRO, int EPI> int launch_gemm_dt(const PwArgs &a, hipStream_t st)
{
    constexpr int lds_main = 2 * (PW_BM * PW_BK * 2 + BN * PW_BK * 2) > PW_BM * BN * 4 ? 2 * (PW_BM * PW_BK * 2 + BN * PW_BK * 2) : PW_BM * BN * 4;
    const int lds = lds_main + (PRO == PRO_NONE ? 0 : (PRO == PRO_AFFINE_RELU ? 2 : 5) * a.Cin * 4) + ((EPI & EPI_RELU_BWD) ? 4 * BN * 4 : 0);
    if (lds > 160 * 1024) return RN_EUNSUPPORTED;
    const int ny = a.N / BN;
    PwArgs b = a;
    const bool by_xcd = PW_GEMM_XCD_MAP && ny >= 2;
    if (by_xcd) b.xcd_cols = ny;
    const dim3 grid = by_xcd ? dim3(8u * (unsigned)((a.gx + 7) / 8) * (unsigned)ny) : dim3((unsigned)a.gx, (unsigned)ny);
    if (a.taps == 1 && a.stride == 1) {
        static rn::DynLdsOptIn opt_in = {};
        { const int rc = opt_in.ensure((const void *)pw_gemm_kernel<DT, BN, PRO, EPI, true>, lds); if (rc != RN_OK) return rc; }
        hipLaunchKernelGGL((pw_gemm_kernel<DT, BN, PRO, EPI, true>), grid, dim3(PW_THREADS), lds, st, b);
    } else {
        static rn::DynLdsOptIn opt_in = {};
        { const int rc = opt_in.ensure((const void *)pw_gemm_kernel<DT, BN, PRO, EPI, false>, lds); if (rc != RN_OK) return rc; }
        hipLaunchKernelGGL((pw_gemm_kernel<DT, BN, PRO, EPI, false>), grid, dim3(PW_THREADS), lds, st, b);
    }
    RN_LAUNCH_CHECK();
    return RN_OK;
}
template <int BN, int PRO, int EPI> int launch_gemm(const PwArgs &a, hipStream_t st)
{
    return a.f16 ? launch_gemm_dt<RN_F16, BN, PRO, EPI>(a, st) : launch_gemm_dt<RN_BF16, BN, PRO, EPI>(a, st);
}

template <int BN, int PRO> int dispatch_epi(const PwArgs &a, const int epi, hipStream_t st)
{
    switch (epi) {
        case 0: return launch_gemm<BN, PRO, 0>(a, st);
        case EPI_STATS: return launch_gemm<BN, PRO, EPI_STATS>(a, st);
        case EPI_RESID: return launch_gemm<BN, PRO, EPI_RESID>(a, st);
        case EPI_RELU_BWD: return launch_gemm<BN, PRO, EPI_RELU_BWD>(a, st);
        case EPI_BIAS: if (PRO == PRO_NONE) return launch_gemm<BN, PRO_NONE, EPI_BIAS>(a, st); return RN_EUNSUPPORTED;
        case EPI_BIAS | EPI_RESID: if (PRO == PRO_NONE) return launch_gemm<BN, PRO_NONE, EPI_BIAS | EPI_RESID>(a, st); return RN_EUNSUPPORTED;
        default: return RN_EUNSUPPORTED;
    }
}

#ifndef PW_WGRAD_XCD_MAP
#define PW_WGRAD_XCD_MAP 1
#endif
template <int DT, int TN, int TK, int PROG, int PROX> int launch_wgrad_dt(const WgArgs &a, hipStream_t st)
{
    constexpr int lds = 64 * (TN + TK) * 2;
    static rn::DynLdsOptIn opt_in = {};
    { const int rc = opt_in.ensure((const void *)pw_wgrad_kernel<DT, TN, TK, PROG, PROX>, lds); if (rc != RN_OK) return rc; }
    const unsigned gy = (unsigned)((a.N / TN) * (a.Cin / TK) * a.taps);
    if (PW_WGRAD_XCD_MAP && gy >= 2 && gy <= 64 && a.S >= 8) {
        WgArgs b = a;
        b.xcd_tiles = (int)gy;
        hipLaunchKernelGGL((pw_wgrad_kernel<DT, TN, TK, PROG, PROX>), dim3(8u * (unsigned)((a.S + 7) / 8) * gy), dim3(PW_THREADS), lds, st, b);
    } else
    hipLaunchKernelGGL((pw_wgrad_kernel<DT, TN, TK, PROG, PROX>), dim3((unsigned)a.S, gy), dim3(PW_THREADS), lds, st, a);
    RN_LAUNCH_CHECK();
    return RN_OK;
}
template <int TN, int TK, int PROG, int PROX> int launch_wgrad(const WgArgs &a, hipStream_t st)
{
    return a.f16 ? launch_wgrad_dt<RN_F16, TN, TK, PROG, PROX>(a, st) : launch_wgrad_dt<RN_BF16, TN, TK, PROG, PROX>(a, st);
}

template <int TN, int TK> int dispatch_wgrad(const WgArgs &a, const int prog, const int prox, hipStream_t st)
{
    if (prog == PRO_NONE && prox == PRO_NONE) return launch_wgrad<TN, TK, PRO_NONE, PRO_NONE>(a, st);
    if (prog == PRO_NONE && prox == PRO_AFFINE_RELU) return launch_wgrad<TN, TK, PRO_NONE, PRO_AFFINE_RELU>(a, st);
    if (prog == PRO_BN_BWD && prox == PRO_NONE) return launch_wgrad<TN, TK, PRO_BN_BWD, PRO_NONE>(a, st);
    if (prog == PRO_BN_BWD && prox == PRO_AFFINE_RELU) return launch_wgrad<TN, TK, PRO_BN_BWD, PRO_AFFINE_RELU>(a, st);
    return RN_EUNSUPPORTED;
}

// weight-gradient tile of an [N][Cin] problem: as much of the small matrix per workgroup as 64 accumulator registers hold
void wgrad_tile(const int N, const int Cin, int &TN, int &TK, const bool g_transform)
{
    if (N >= 128 && Cin >= 128) { TN = 128; TK = 128; }
    else if (N >= 128) { TN = (N >= 256 && !g_transform) ? 256 : 128; TK = 64; }    // (BN-backward prologue: 256-wide G staging would spill)
    else if (Cin >= 128) { TN = 64; TK = Cin >= 256 ? 256 : 128; }
    else { TN = 64; TK = 64; }
    // the grid is (N / TN) x (Cin / TK) tiles: a tile that does not DIVIDE its axis would leave the rest of dW unwritten (N = 192
    // with TN = 128: rows 128..191).  N and Cin are multiples of 64 (check_geometry), so halving ends at 64 at the latest; every
    // (TN, TK) this can produce -- (128,128) (256,64) (128,64) (64,256) (64,128) (64,64) -- is instantiated below.
    while (N % TN) TN >>= 1;
    while (Cin % TK) TK >>= 1;
}

int wgrad_splits(const rn_pw_conv *d, int *tiles_per_split, const bool g_transform)
{
    int TN, TK;
    wgrad_tile(d->N, d->Cin, TN, TK, g_transform);
    const int tiles = (d->N / TN) * (d->Cin / TK) * d->taps;
    int S = (2 * cu_count() + tiles - 1) / tiles;                // about two workgroups per CU
    const int ktiles = (int)((d->M + 63) / 64);
    const int64_t split_bytes = (int64_t)d->N * d->taps * d->Cin * 4;
    // Split scan on MI355X (kernel + reduction, isolated, us): two workgroups per CU is the optimum at every width of the trunk --
    //   tiles 8 (134400 x 256 x 512): S = 32 / 64: 71 / 60;   tiles 32 (33600 x 512 x 1024): S = 8 / 16 / 32: 71 / 60 / 96;
    //   tiles 64 (8400 x 512 x 2048): S = 4 / 8 / 32: 43 / 37 / 71;   tiles 128 (8400 x 2048 x 1024): S = 2 / 4 / 32: 75 / 61 / 134
    // -- which is always 2 * CUs * 64 KiB = 32 MiB of f32 partials.  (Rounds 2-3 capped the partials at 16 MiB but never went
    // below 32 splits: layer4's GEMMs wrote and re-read 113 - 256 MB for 43 MB of operands.)
    const int cap = (int)(((int64_t)32 << 20) / split_bytes);
    if (S > cap) S = cap;
    if (S > ktiles) S = ktiles;
    if (S > 512) S = 512;
    if (S < 1) S = 1;
    const int tps = (ktiles + S - 1) / S;
    if (tiles_per_split) *tiles_per_split = tps;
    return (ktiles + tps - 1) / tps;
}

int check_geometry(const rn_pw_conv *d)
{
    if (!d || d->M <= 0 || d->Cin <= 0 || d->N <= 0) return RN_EINVAL;
    if (d->Cin % 64 || d->N % 64) return RN_EUNSUPPORTED;
    if (d->taps != 1 && d->taps != 9) return RN_EUNSUPPORTED;
    if (d->dtype != 0 && d->dtype != RN_BF16 && d->dtype != RN_F16) return RN_EUNSUPPORTED;          // (0: a caller from before the field existed = bf16)
    if (d->stride < 1 || d->pad < 0 || d->Ho <= 0 || d->Wo <= 0 || d->H <= 0 || d->W <= 0) return RN_EINVAL;
    if (d->M % (d->Ho * d->Wo)) return RN_EINVAL;
    if (d->M >= ((int64_t)1 << 31)) return RN_EUNSUPPORTED;                      // the kernels index rows with 32-bit integers
    if ((int64_t)(d->M / (d->Ho * d->Wo)) * d->H * d->W >= ((int64_t)1 << 31) || (int64_t)d->M * d->N >= ((int64_t)1 << 40)) return RN_EUNSUPPORTED;
    return RN_OK;
}

}  // namespace

constexpr int PAIR_THREADS = 512, PAIR_WGS_PER_CU = 1;
static int pair_walkers(const int64_t M, const int Cm)
{
    const int halves = Cm / 64;
    const int MT = (int)((M + 127) / 128), cap = PAIR_WGS_PER_CU * cu_count() / halves;
    int w = MT;
    if (MT > cap) { const int rounds = (MT + cap - 1) / cap; w = (MT + rounds - 1) / rounds; }
    if (halves == 2) w = (w + 7) & ~7;          // the pairing of the two column halves works on groups of 8 walkers (idle ones write zero partials)
    return w;
}
static bool pair_shape_ok(const int64_t M, const int Cm, const int C4)
{
    // (Cm, C4) = (128, 512): two workgroups per row tile, 64 columns each (one workgroup would hold 512 x 128 f32 accumulators, 256 KB:
    // 134 - 192 VGPRs spilled, 216 us against 105 + 72 for the two separate launches at the layer2 shape)
    return M > 0 && M < ((int64_t)1 << 31) / (C4 > 0 ? C4 : 1) * 8 && ((Cm == 64 && C4 == 256) || (Cm == 128 && C4 == 512));
}
template <int DT, int CM, int KT> static int launch_pair(const PairArgs &a, hipStream_t st)
{
    constexpr int stage = 128 * 128 + CM * 128, main_b = (2 * stage > 128 * CM * 4) ? 2 * stage : 128 * CM * 4;
    constexpr int lds = main_b + 128 * CM * 2 + 3 * KT * 64 * 4 + 4 * CM * 4;
    static rn::DynLdsOptIn opt_in = {};
    { const int rc = opt_in.ensure((const void *)pw_conv3_bwd_kernel<DT, CM, KT, PAIR_THREADS>, lds); if (rc != RN_OK) return rc; }
    hipLaunchKernelGGL((pw_conv3_bwd_kernel<DT, CM, KT, PAIR_THREADS>), dim3((unsigned)(a.nwalk * a.halves)), dim3(PAIR_THREADS), lds, st, a);
    RN_LAUNCH_CHECK();
    return RN_OK;
}

constexpr int CHAIN_WGS_PER_CU = 1;
static int chain_walkers(const int64_t M)
{
    const int MT = (int)((M + 127) / 128), cap = CHAIN_WGS_PER_CU * cu_count();
    if (MT <= cap) return MT;
    const int rounds = (MT + cap - 1) / cap;
    return (MT + rounds - 1) / rounds;
}
static bool chain_shape_ok(const int64_t M, const int C4, const int CN)
{
    return M > 0 && M < ((int64_t)1 << 31) / (C4 > 0 ? C4 : 1) * 8 && ((C4 == 256 && (CN == 64 || CN == 128)) || (C4 == 512 && CN == 128));
}
template <int DT, int CN, int KT, bool RA> static int launch_chain(const ChainArgs &a, hipStream_t st)
{
    constexpr int stage = 128 * 128 + CN * 128, main_b = (2 * stage > 128 * CN * 4) ? 2 * stage : 128 * CN * 4;
    constexpr int lds = main_b + (RA ? 4 : 2) * KT * 64 * 4;
    static rn::DynLdsOptIn opt_in = {};
    { const int rc = opt_in.ensure((const void *)pw_block_out_conv1_kernel<DT, CN, KT, RA>, lds); if (rc != RN_OK) return rc; }
    hipLaunchKernelGGL((pw_block_out_conv1_kernel<DT, CN, KT, RA>), dim3((unsigned)a.gx), dim3(512), lds, st, a);
    RN_LAUNCH_CHECK();
    return RN_OK;
}
template <int DT, bool RA> static int dispatch_chain(const ChainArgs &a, const int C4, const int CN, hipStream_t st)
{
    if (C4 == 256 && CN == 64) return launch_chain<DT, 64, 4, RA>(a, st);
    if (C4 == 256) return launch_chain<DT, 128, 4, RA>(a, st);
    return launch_chain<DT, 128, 8, RA>(a, st);
}

RN_API int rn_pw_block_out_conv1_walkers(int64_t M, int C4, int CN) { return chain_shape_ok(M, C4, CN) ? chain_walkers(M) : 0; }

RN_API int rn_pw_block_out_conv1(int64_t M, int C4, int CN, int dtype, const void *z3, const void *resid, const float *res_a, const float *res_b,
                                 const float *oa, const float *ob, const void *w1, void *y, uint8_t *ybits, void *z1, float *partial,
                                 void *stream)
{
    if (!z3 || !resid || !oa || !ob || !w1 || !y || !ybits || !z1 || !partial || (!res_a) != (!res_b)) return RN_EINVAL;
    if (dtype != RN_BF16 && dtype != RN_F16) return RN_EUNSUPPORTED;
    if (!chain_shape_ok(M, C4, CN)) return RN_EUNSUPPORTED;
    if (!rn::aligned(z3, 16) || !rn::aligned(resid, 16) || !rn::aligned(w1, 16) || !rn::aligned(y, 16) || !rn::aligned(z1, 16) ||
        !rn::aligned(oa, 16) || !rn::aligned(ob, 16) || (res_a && (!rn::aligned(res_a, 16) || !rn::aligned(res_b, 16))))
        return RN_EALIGN;
    ChainArgs a = {};
    a.Z3 = (const uint16_t *)z3; a.R = (const uint16_t *)resid; a.oa = oa; a.ob = ob; a.ra = res_a; a.rb = res_b;
    a.W1 = (const uint16_t *)w1; a.Y = (uint16_t *)y; a.ybits = ybits; a.Z1 = (uint16_t *)z1; a.partial = partial;
    a.M = (int)M; a.gx = chain_walkers(M); a.f16 = dtype == RN_F16;
    hipStream_t st = (hipStream_t)stream;
    if (a.f16) return res_a ? dispatch_chain<RN_F16, true>(a, C4, CN, st) : dispatch_chain<RN_F16, false>(a, C4, CN, st);
    return res_a ? dispatch_chain<RN_BF16, true>(a, C4, CN, st) : dispatch_chain<RN_BF16, false>(a, C4, CN, st);
}

static bool dgrad_sums_shape_ok(const int64_t M, const int Cm, const int C4)
{
    return M > 0 && M < ((int64_t)1 << 31) / (C4 > 0 ? C4 : 1) * 8 && (Cm == 64 || Cm == 128) && C4 > 0 && C4 % 128 == 0;
}
template <int DT, int KC, bool FWD = false> static int launch_dgrad_sums(const DgradSumsArgs &a, hipStream_t st)
{
    constexpr int lds = KC * 128 * 128 * 2 + 128 * 128 * 4;
    static rn::DynLdsOptIn opt_in = {};
    { const int rc = opt_in.ensure((const void *)pw_dgrad_sums_kernel<DT, KC, FWD>, lds); if (rc != RN_OK) return rc; }
    hipLaunchKernelGGL((pw_dgrad_sums_kernel<DT, KC, FWD>), dim3((unsigned)a.gx, (unsigned)(a.C4 / 128)), dim3(512), lds, st, a);
    RN_LAUNCH_CHECK();
    return RN_OK;
}

// one workgroup per CU over all column tiles: walkers = row-tile walkers per column tile
static int dgrad_sums_walkers(const int64_t M, const int C4)
{
    const int MT = (int)((M + 127) / 128), tiles = C4 / 128;
    int cap = cu_count() / tiles;
    if (cap < 1) cap = 1;
    if (MT <= cap) return MT;
    const int rounds = (MT + cap - 1) / cap;
    return (MT + rounds - 1) / rounds;
}

RN_API int rn_pw_dgrad_resid_sums_walkers(int64_t M, int Cm, int C4) { return dgrad_sums_shape_ok(M, Cm, C4) ? dgrad_sums_walkers(M, C4) : 0; }

RN_API int rn_pw_dgrad_resid_sums(int64_t M, int Cm, int C4, int dtype, const void *dz1, const void *w1t, const void *resid, const uint8_t *rbits,
                                  int res_stride, int res_h, int res_w, const void *prev_z3, const uint8_t *prev_bits, const float *prev_mean,
                                  const float *prev_invstd, void *dx, float *partial, void *stream)
{
    if (!dz1 || !w1t || !resid || !prev_z3 || !prev_bits || !prev_mean || !prev_invstd || !dx || !partial) return RN_EINVAL;
    if (dtype != RN_BF16 && dtype != RN_F16) return RN_EUNSUPPORTED;
    if (!dgrad_sums_shape_ok(M, Cm, C4)) return RN_EUNSUPPORTED;
    if (res_stride != 0 && res_stride != 1 && res_stride != 2) return RN_EUNSUPPORTED;
    if (res_stride == 2 && (res_h <= 0 || res_w <= 0 || M % ((int64_t)res_h * res_w))) return RN_EINVAL;
    if (!rn::aligned(dz1, 16) || !rn::aligned(w1t, 16) || !rn::aligned(resid, 16) || !rn::aligned(prev_z3, 16) || !rn::aligned(dx, 16) ||
        !rn::aligned(prev_mean, 16) || !rn::aligned(prev_invstd, 16))
        return RN_EALIGN;
    DgradSumsArgs a = {};
    a.A = (const uint16_t *)dz1; a.Wt = (const uint16_t *)w1t; a.R = (const uint16_t *)resid; a.rbits = rbits;
    a.rs = res_stride == 2 ? 2 : 1; a.rH = res_h; a.rW = res_w;
    a.Zp = (const uint16_t *)prev_z3; a.pbits = prev_bits; a.pmean = prev_mean; a.pinv = prev_invstd;
    a.Y = (uint16_t *)dx; a.partial = partial; a.M = (int)M; a.C4 = C4; a.gx = dgrad_sums_walkers(M, C4); a.f16 = dtype == RN_F16;
    hipStream_t st = (hipStream_t)stream;
    if (Cm == 64) return a.f16 ? launch_dgrad_sums<RN_F16, 1>(a, st) : launch_dgrad_sums<RN_BF16, 1>(a, st);
    return a.f16 ? launch_dgrad_sums<RN_F16, 2>(a, st) : launch_dgrad_sums<RN_BF16, 2>(a, st);
}

RN_API int rn_pw_conv3_forward_walkers(int64_t M, int Cm, int C4) { return dgrad_sums_shape_ok(M, Cm, C4) ? dgrad_sums_walkers(M, C4) : 0; }

RN_API int rn_pw_conv3_forward(int64_t M, int Cm, int C4, int dtype, const void *z2, const float *fwd_coef, const void *w3, void *z3, float *partial,
                               void *stream)
{
    if (!z2 || !fwd_coef || !w3 || !z3 || !partial) return RN_EINVAL;
    if (dtype != RN_BF16 && dtype != RN_F16) return RN_EUNSUPPORTED;
    if (!dgrad_sums_shape_ok(M, Cm, C4)) return RN_EUNSUPPORTED;
    if (!rn::aligned(z2, 16) || !rn::aligned(w3, 16) || !rn::aligned(z3, 16) || !rn::aligned(fwd_coef, 16)) return RN_EALIGN;
    DgradSumsArgs a = {};
    a.A = (const uint16_t *)z2; a.Wt = (const uint16_t *)w3; a.Y = (uint16_t *)z3; a.partial = partial; a.fa = fwd_coef; a.fb = fwd_coef + Cm;
    a.rs = 1; a.M = (int)M; a.C4 = C4; a.gx = dgrad_sums_walkers(M, C4); a.f16 = dtype == RN_F16;
    hipStream_t st = (hipStream_t)stream;
    if (Cm == 64) return a.f16 ? launch_dgrad_sums<RN_F16, 1, true>(a, st) : launch_dgrad_sums<RN_BF16, 1, true>(a, st);
    return a.f16 ? launch_dgrad_sums<RN_F16, 2, true>(a, st) : launch_dgrad_sums<RN_BF16, 2, true>(a, st);
}

RN_API int rn_pw_conv3_backward_walkers(int64_t M, int Cm, int C4) { return pair_shape_ok(M, Cm, C4) ? pair_walkers(M, Cm) : 0; }

RN_API size_t rn_pw_conv3_backward_workspace_bytes(int64_t M, int Cm, int C4)
{
    return pair_shape_ok(M, Cm, C4) ? (size_t)pair_walkers(M, Cm) * C4 * Cm * sizeof(float) : 0;
}

RN_API int rn_pw_conv3_backward(int64_t M, int Cm, int C4, int dtype, const void *g, const void *z3, const uint8_t *bits, const float *a3,
                                const float *k0, const float *k1, const void *w3t, const void *z2, const float *ea, const float *eb,
                                const float *emean, const float *einv, void *dy2, float *partial_bn, void *workspace,
                                size_t workspace_bytes, int *splits, void *stream)
{
    if (!g || !z3 || !bits || !a3 || !k0 || !k1 || !w3t || !z2 || !ea || !eb || !emean || !einv || !dy2 || !partial_bn || !workspace || !splits)
        return RN_EINVAL;
    if (dtype != RN_BF16 && dtype != RN_F16) return RN_EUNSUPPORTED;
    if (!pair_shape_ok(M, Cm, C4)) return RN_EUNSUPPORTED;
    if (!rn::aligned(g, 16) || !rn::aligned(z3, 16) || !rn::aligned(w3t, 16) || !rn::aligned(z2, 16) || !rn::aligned(dy2, 16) ||
        !rn::aligned(workspace, 16) || !rn::aligned(a3, 16) || !rn::aligned(k0, 16) || !rn::aligned(k1, 16) || !rn::aligned(ea, 16) ||
        !rn::aligned(eb, 16) || !rn::aligned(emean, 16) || !rn::aligned(einv, 16))
        return RN_EALIGN;
    if (workspace_bytes < rn_pw_conv3_backward_workspace_bytes(M, Cm, C4)) return RN_EWORKSPACE;
    PairArgs a = {};
    a.G = (const uint16_t *)g; a.Z3 = (const uint16_t *)z3; a.gbits = bits; a.pa = a3; a.pb = k0; a.pc = k1;
    a.Wt = (const uint16_t *)w3t; a.Z2 = (const uint16_t *)z2; a.ea = ea; a.eb = eb; a.emean = emean; a.einv = einv;
    a.Y = (uint16_t *)dy2; a.partial_bn = partial_bn; a.partial_w = (float *)workspace;
    a.M = (int)M; a.nwalk = pair_walkers(M, Cm); a.f16 = dtype == RN_F16; a.cm = Cm; a.halves = Cm / 64;
    *splits = a.nwalk;
    hipStream_t st = (hipStream_t)stream;
    if (Cm == 64) return a.f16 ? launch_pair<RN_F16, 64, 4>(a, st) : launch_pair<RN_BF16, 64, 4>(a, st);
    return a.f16 ? launch_pair<RN_F16, 64, 8>(a, st) : launch_pair<RN_BF16, 64, 8>(a, st);
}

RN_API int rn_pw_walkers(int64_t M) { return M > 0 && M < ((int64_t)1 << 31) ? walkers((int)M) : 0; }

RN_API int rn_pw_conv_forward(const rn_pw_conv *d, const void *x, const void *w, void *y, const rn_pw_prologue *pro,
                              const rn_pw_epilogue *epi, void *stream)
{
    const int rc = check_geometry(d);
    if (rc != RN_OK) return rc;
    if (!x || !w || !y) return RN_EINVAL;
    if (!rn::aligned(x, 16) || !rn::aligned(w, 16) || !rn::aligned(y, 16)) return RN_EALIGN;
    PwArgs a = {};
    a.X = (const uint16_t *)x; a.W = (const uint16_t *)w; a.Y = (uint16_t *)y;
    a.M = (int)d->M; a.Cin = d->Cin; a.N = d->N; a.taps = d->taps; a.stride = d->stride; a.pad = d->pad;
    a.Ho = d->Ho; a.Wo = d->Wo; a.H = d->H; a.W_ = d->W;
    a.gx = walkers(a.M);
    a.f16 = d->dtype == RN_F16;
    int p = PRO_NONE, e = 0;
    if (pro && pro->kind != PRO_NONE) {
        p = pro->kind;
        if (p == PRO_AFFINE_RELU) {
            if (!pro->a || !pro->b) return RN_EINVAL;
            a.pa = pro->a; a.pb = pro->b;
        } else if (p == PRO_BN_BWD) {
            if (!pro->a || !pro->b || !pro->c || !pro->x2) return RN_EINVAL;
            if (pro->relu_mode == 2 && (!pro->fa || !pro->fb)) return RN_EINVAL;
            if (pro->relu_mode == 3 && !pro->bits) return RN_EINVAL;
            if (pro->relu_mode != 0 && pro->relu_mode != 2 && pro->relu_mode != 3) return RN_EINVAL;
            a.pa = pro->a; a.pb = pro->b; a.pc = pro->c; a.fa = pro->fa; a.fb = pro->fb; a.X2 = (const uint16_t *)pro->x2;
            a.xbits = pro->bits; a.relu_mode = pro->relu_mode;
        } else return RN_EINVAL;
    }
    if (epi && epi->kind != 0) {
        e = epi->kind;
        if (e & EPI_BIAS) {                                       // (inference epilogue: alone or with an unmasked residual)
            if (!epi->bias || (e & ~(EPI_BIAS | EPI_RESID)) || !rn::aligned(epi->bias, 16)) return RN_EINVAL;
            a.bias = epi->bias; a.relu_out = epi->relu ? 1 : 0;
        }
        if (e == EPI_BIAS) { }
        else if (e == EPI_STATS) { if (!epi->partial) return RN_EINVAL; a.partial = epi->partial; }
        else if ((e & ~EPI_BIAS) == EPI_RESID) {
            if (!epi->resid) return RN_EINVAL;
            if (epi->res_stride != 0 && epi->res_stride != 1 && epi->res_stride != 2) return RN_EUNSUPPORTED;
            a.R = (const uint16_t *)epi->resid; a.rbits = epi->rbits; a.rs = epi->res_stride == 2 ? 2 : 1;
            if (a.rs == 2) {
                if (epi->res_h <= 0 || epi->res_w <= 0 || d->M % ((int64_t)epi->res_h * epi->res_w)) return RN_EINVAL;
                a.rH = epi->res_h; a.rW = epi->res_w;
            }
        }
        else if (e == EPI_RELU_BWD) {
            if (!epi->partial || !epi->zprev || !epi->ea || !epi->eb || !epi->emean || !epi->einv) return RN_EINVAL;
            a.partial = epi->partial; a.Zp = (const uint16_t *)epi->zprev; a.ea = epi->ea; a.eb = epi->eb; a.emean = epi->emean; a.einv = epi->einv;
        } else return RN_EINVAL;
    }
    hipStream_t st = (hipStream_t)stream;
    const bool wide = d->N % 128 == 0;
#define RN_PW_GO(BN)                                                                 \
    switch (p) {                                                                     \
        case PRO_NONE: return dispatch_epi<BN, PRO_NONE>(a, e, st);                  \
        case PRO_AFFINE_RELU: return dispatch_epi<BN, PRO_AFFINE_RELU>(a, e, st);    \
        default: return dispatch_epi<BN, PRO_BN_BWD>(a, e, st);                      \
    }
    if (wide) { RN_PW_GO(128) } else { RN_PW_GO(64) }
#undef RN_PW_GO
}

RN_API size_t rn_pw_wgrad_workspace_bytes(const rn_pw_conv *d)
{
    if (check_geometry(d) != RN_OK) return 0;
    const int S0 = wgrad_splits(d, nullptr, false), S1 = wgrad_splits(d, nullptr, true);
    return (size_t)(S0 > S1 ? S0 : S1) * d->N * d->taps * d->Cin * sizeof(float);
}

// splits >= 0: the kernel only (partials in `workspace`), *splits = their number; dw is then produced by rn_pw_wgrad_reduce_many
static int pw_wgrad_impl(const rn_pw_conv *d, const void *g, const void *x, void *dw, const rn_pw_prologue *gpro, const rn_pw_prologue *xpro,
                         void *workspace, size_t workspace_bytes, void *stream, int *splits)
{
    const int rc = check_geometry(d);
    if (rc != RN_OK) return rc;
    if (!g || !x || (!dw && !splits) || !workspace) return RN_EINVAL;
    if (!rn::aligned(g, 16) || !rn::aligned(x, 16) || (dw && !rn::aligned(dw, 16)) || !rn::aligned(workspace, 16)) return RN_EALIGN;
    if (workspace_bytes < rn_pw_wgrad_workspace_bytes(d)) return RN_EWORKSPACE;
    WgArgs a = {};
    a.G = (const uint16_t *)g; a.X = (const uint16_t *)x; a.partial = (float *)workspace;
    a.M = (int)d->M; a.N = d->N; a.Cin = d->Cin; a.taps = d->taps; a.stride = d->stride; a.pad = d->pad;
    a.Ho = d->Ho; a.Wo = d->Wo; a.H = d->H; a.W_ = d->W;
    const bool g_transform = gpro && gpro->kind != PRO_NONE;
    a.f16 = d->dtype == RN_F16;
    a.S = wgrad_splits(d, &a.tiles_per_split, g_transform);
    int pg = PRO_NONE, px = PRO_NONE;
    if (gpro && gpro->kind != PRO_NONE) {
        if (gpro->kind != PRO_BN_BWD || !gpro->a || !gpro->b || !gpro->c || !gpro->x2) return RN_EINVAL;
        if (gpro->relu_mode == 2 && (!gpro->fa || !gpro->fb)) return RN_EINVAL;
        if (gpro->relu_mode == 3 && !gpro->bits) return RN_EINVAL;
        if (gpro->relu_mode != 0 && gpro->relu_mode != 2 && gpro->relu_mode != 3) return RN_EINVAL;
        pg = PRO_BN_BWD;
        a.G2 = (const uint16_t *)gpro->x2; a.gbits = gpro->bits; a.ga = gpro->a; a.gb = gpro->b; a.gc = gpro->c; a.gfa = gpro->fa; a.gfb = gpro->fb;
        a.g_relu_mode = gpro->relu_mode;
    }
    if (xpro && xpro->kind != PRO_NONE) {
        if (xpro->kind != PRO_AFFINE_RELU || !xpro->a || !xpro->b) return RN_EINVAL;
        px = PRO_AFFINE_RELU;
        a.xa = xpro->a; a.xb = xpro->b;
    }
    hipStream_t st = (hipStream_t)stream;
    int TN, TK, r;
    wgrad_tile(d->N, d->Cin, TN, TK, g_transform);
    if (TN == 128 && TK == 128) r = dispatch_wgrad<128, 128>(a, pg, px, st);
    else if (TN == 256) r = dispatch_wgrad<256, 64>(a, pg, px, st);
    else if (TN == 128) r = dispatch_wgrad<128, 64>(a, pg, px, st);
    else if (TK == 256) r = dispatch_wgrad<64, 256>(a, pg, px, st);
    else if (TK == 128) r = dispatch_wgrad<64, 128>(a, pg, px, st);
    else r = dispatch_wgrad<64, 64>(a, pg, px, st);
    if (r != RN_OK) return r;
    if (splits) { *splits = a.S; return RN_OK; }
    const int64_t n4 = (int64_t)d->N * d->taps * d->Cin / 4;
    if (a.f16) hipLaunchKernelGGL(pw_wgrad_reduce_kernel<RN_F16>, dim3((unsigned)((n4 + 31) / 32)), dim3(256), 0, st, (const float *)workspace, a.S, n4, (uint16_t *)dw);
    else hipLaunchKernelGGL(pw_wgrad_reduce_kernel<RN_BF16>, dim3((unsigned)((n4 + 31) / 32)), dim3(256), 0, st, (const float *)workspace, a.S, n4, (uint16_t *)dw);
    RN_LAUNCH_CHECK();
    return RN_OK;
}

RN_API int rn_pw_conv_wgrad(const rn_pw_conv *d, const void *g, const void *x, void *dw, const rn_pw_prologue *gpro,
                            const rn_pw_prologue *xpro, void *workspace, size_t workspace_bytes, void *stream)
{
    return pw_wgrad_impl(d, g, x, dw, gpro, xpro, workspace, workspace_bytes, stream, nullptr);
}

RN_API int rn_pw_conv_wgrad_partial(const rn_pw_conv *d, const void *g, const void *x, const rn_pw_prologue *gpro, const rn_pw_prologue *xpro,
                                    void *workspace, size_t workspace_bytes, int *splits, void *stream)
{
    if (!splits) return RN_EINVAL;
    return pw_wgrad_impl(d, g, x, nullptr, gpro, xpro, workspace, workspace_bytes, stream, splits);
}

RN_API int rn_pw_wgrad_reduce_many_dt(const void *const *partials, const int *splits, const int64_t *n_elems, void *const *dws, int n, int dtype,
                                      void *stream)
{
    if (dtype != RN_BF16 && dtype != RN_F16) return RN_EUNSUPPORTED;
    if (!partials || !splits || !n_elems || !dws || n <= 0 || n > RED_MAX) return RN_EINVAL;
    ReduceTable t;
    int64_t most = 1;
    for (int i = 0; i < RED_MAX; ++i) {
        const int q = i < n ? i : 0;
        if (!partials[q] || !dws[q] || splits[q] <= 0 || n_elems[q] <= 0 || (n_elems[q] & 3)) return RN_EINVAL;
        if (!rn::aligned(partials[q], 16) || !rn::aligned(dws[q], 8)) return RN_EALIGN;
        t.partial[i] = (const float *)partials[q]; t.dw[i] = (uint16_t *)dws[q]; t.S[i] = splits[q]; t.n4[i] = n_elems[q] / 4;
        if (t.n4[i] > most) most = t.n4[i];
    }
    int64_t bx = (most + 31) / 32;
    if (bx > 2048) bx = 2048;
    if (dtype == RN_F16) hipLaunchKernelGGL(pw_wgrad_reduce_many_kernel<RN_F16>, dim3((unsigned)bx, (unsigned)n), dim3(256), 0, (hipStream_t)stream, t);
    else hipLaunchKernelGGL(pw_wgrad_reduce_many_kernel<RN_BF16>, dim3((unsigned)bx, (unsigned)n), dim3(256), 0, (hipStream_t)stream, t);
    RN_LAUNCH_CHECK();
    return RN_OK;
}

RN_API int rn_pw_wgrad_reduce_many(const void *const *partials, const int *splits, const int64_t *n_elems, void *const *dws, int n, void *stream)
{
    return rn_pw_wgrad_reduce_many_dt(partials, splits, n_elems, dws, n, RN_BF16, stream);
}
