// The ResNet stem convolution -- 7x7 / stride 2 / pad 3, 3 -> 64 channels (reference: retinanet/backbone.py:152, applied at
// :246) -- as hand-written MFMA kernels for bf16 channels-last tensors.  MIOpen runs it at 170 TFLOP/s (240 us forward,
// 245 us weight gradient at the bench shape [8, 3, 800, 1344]) although the layer is bound by the 275 MB it writes / reads.
//
// Layout trick.  The image is first copied into a zero-bordered NHWC4 buffer xp[B][H + 6][Wpp][4] (4th channel 0; 3 border
// pixels left / top, >= 3 right / bottom): one kernel row of one output pixel is then 7 pixels x 4 channels = 28 (padded: 32)
// CONSECUTIVE bf16 starting at pixel (2 yo + r, 2 xo) -- exactly one 32-deep k-step of v_mfma_f32_16x16x32_bf16, whose operand
// lane (pixel p, k-group g) wants the 8 consecutive k values 8g .. 8g+7: 16 contiguous bytes at pixel 2 (xo + p) + 2 g.  So the
// activation fragments are plain 16-byte global loads (no LDS, no bounds logic: the border is real zeros), the 64 x 7 x 32
// weights live in registers (28 fragments per lane), and a wave produces 16 pixels x 64 channels per 28 MFMAs + 7 loads.
//
// Forward (stem_fwd_kernel): D[channel][pixel] tiles (weights as the M operand), so a lane ends up with 4 CONSECUTIVE channels
// of one pixel per 16-channel tile; the 16 x 64 tile goes through a wave-private LDS strip and leaves as 16-byte stores.  The
// BatchNorm statistics of the (bf16-rounded) output ride along: per-lane running sums over the wave's tiles, reduced once at
// the end -> one partial row per workgroup for rn_bn_stats_finalize.
#include "rn_common.hpp"

namespace {

typedef __attribute__((ext_vector_type(4))) float f32x4v;

constexpr int STEM_THREADS = 256, STEM_WAVES = 4;
constexpr int STEM_KROW = 32;                    // k elements per kernel row: 7 pixels x 4 channels + 4 zeros
constexpr int STEM_SLACK_PX = 160;               // (a weight-gradient stage reads 134 pixels from its first one, whatever the row length)
constexpr int STEM_NB = 2;                       // fragment sets per wave (one computing, one in flight; 3 .. 8 sets at one wave per SIMD: no gain)

struct StemPadArgs {
    const uint16_t *x;      // [B][H][W][3] bf16 (channels-last memory of [B, 3, H, W])
    uint16_t *xp;           // [B][Hp2][Wpp][4]
    int B, H, W, Hp2, Wpp;
};

// one thread = one padded pixel (8 bytes)
__global__ __launch_bounds__(256) void stem_pad_kernel(const StemPadArgs a)
{
    // (+ 2 rows + 160 pixels of slack behind the last image, zeroed: the kernels' masked lanes read there, and the weight gradient multiplies
    // what they read by zeros -- it must be finite)
    const int64_t total = ((int64_t)a.B * a.Hp2 + 2) * a.Wpp + STEM_SLACK_PX;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int xq = (int)(i % a.Wpp);
        const int64_t rowid = i / a.Wpp;
        const int yq = (int)(rowid % a.Hp2), b = (int)(rowid / a.Hp2);
        const int xs = xq - 3, ys = yq - 3;
        rn::u32x2 o = {0u, 0u};
        if (b < a.B && xs >= 0 && xs < a.W && ys >= 0 && ys < a.H) {
            const uint16_t *p = a.x + (((int64_t)b * a.H + ys) * a.W + xs) * 3;
            o.x = (uint32_t)p[0] | ((uint32_t)p[1] << 16);
            o.y = (uint32_t)p[2];
        }
        ((rn::u32x2 *)a.xp)[i] = o;
    }
}

// w [64][7][7][3] bf16 (channels-last memory of [64, 3, 7, 7]) -> wk [64][7][32]: k = 4 * px + c, zeros at c = 3 and px = 7
__global__ __launch_bounds__(256) void stem_weight_pack_kernel(const uint16_t *__restrict__ w, uint16_t *__restrict__ wk)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 64 * 7 * STEM_KROW) return;
    const int k = i % STEM_KROW, r = (i / STEM_KROW) % 7, n = i / (7 * STEM_KROW);
    const int px = k >> 2, c = k & 3;
    wk[i] = (px < 7 && c < 3) ? w[((n * 7 + r) * 7 + px) * 3 + c] : (uint16_t)0;
}

struct StemArgs {
    const uint16_t *xp;     // [B][Hp2][Wpp][4]
    const uint16_t *wk;     // [64][7][32]
    uint16_t *y;            // [B][Ho][Wo][64]
    float *partial;         // [gridDim.x][2][64] sum / sum of squares of the stored outputs (null: no statistics)
    int B, Ho, Wo, Hp2, Wpp;
    int tiles_x, total_tiles, tiles_per_wg;
};


template <int DT>
__global__ __launch_bounds__(STEM_THREADS) void stem_fwd_kernel(const StemArgs a)
{
    __shared__ uint32_t s_strip[STEM_WAVES][16 * 34];             // per wave: 16 pixels x (32 dwords of 64 bf16 channels + 2 pad)
    __shared__ float s_stat[STEM_WAVES][2][64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // wave-uniform: the tile walk below is scalar work
    const int p = lane & 15, g = lane >> 4;                          // pixel of the tile / k-group (as operand), channel group (as result)

    // the weights: 4 channel tiles x 7 kernel rows, lane = (channel p of the tile, k-group g)
    typedef typename rn::mma<DT>::frag el16x8;             // (8 consecutive 16-bit elements: bf16 or fp16)
    el16x8 wf[4][7];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int r = 0; r < 7; ++r) wf[mt][r] = *(const el16x8 *)(a.wk + ((mt * 16 + p) * 7 + r) * STEM_KROW + 8 * g);

    float ssum[4][4], ssq[4][4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) { ssum[mt][r] = 0.0f; ssq[mt][r] = 0.0f; }

    const int t_beg = blockIdx.x * a.tiles_per_wg, t_end = min(t_beg + a.tiles_per_wg, a.total_tiles);
    uint32_t *strip = s_strip[wave];

    // Tile walk: this wave takes tiles t_beg + wave, + 4, ...  (b, yo, tx) advance by counters -- no division per tile --, the
    // source address is a wave-uniform base + a per-lane offset + r rows.
    const int lane_off = (2 * p + 2 * g) * 4;                         // pixel 2 (xo0 + p) + 2 g of the row, in elements
    const int64_t row_elems = (int64_t)a.Wpp * 4;
    auto tile_base = [&](const int b, const int yo, const int tx) {
        return a.xp + (((int64_t)b * a.Hp2 + 2 * yo) * a.Wpp + 32 * tx) * 4;
    };
    auto load_tile = [&](const uint16_t *base, el16x8 (&af)[7]) {
#pragma unroll
        for (int r = 0; r < 7; ++r) af[r] = *(const el16x8 *)(base + r * row_elems + lane_off);
    };
    auto advance = [&](int &b, int &yo, int &tx) {                   // + STEM_WAVES tiles (tiles_x >= 1: wraps handled in a loop)
        tx += STEM_WAVES;
        while (tx >= a.tiles_x) { tx -= a.tiles_x; if (++yo == a.Ho) { yo = 0; ++b; } }
    };

    // A ring of STEM_NB fragment sets per wave: the tiles after the current one are in flight while it is computed.  Measured at
    // the bench shape (134 400 tiles): 2 sets x 2 waves per SIMD 100 us; 3 sets 111; 4 / 6 sets at one wave per SIMD 101 / 103 --
    // not latency-bound.  Ablations: no output stores 64 us, no statistics 94, one load per tile instead of seven 85; non-temporal
    // output stores 81 (and the BatchNorm apply pass that reads the output next 108 -> 98 us).  MIOpen: 240 us + 46 us statistics.
    el16x8 ring[STEM_NB][7];
    int t = t_beg + wave;
    const int t_last = t_end - 1;
    int tx = t % a.tiles_x, yo = (t / a.tiles_x) % a.Ho, b = (t / a.tiles_x) / a.Ho;      // once per wave: compute cursor
    // load cursor.  A wave of the last workgroup may own NO tile (t > t_last): its (masked) loads must still hit the buffer -- they
    // walk the workgroup's last tile; t itself can lie past the last image (found by placing the buffers at the end of their mappings)
    int tl = t <= t_last ? t : t_last;
    int xl = tl % a.tiles_x, yl = (tl / a.tiles_x) % a.Ho, bl = (tl / a.tiles_x) / a.Ho;
    // The loop body is straight-line code -- every wave runs the same number of (load, compute) steps, loads past the wave's
    // last tile re-read that tile, stores and statistics past it are masked -- so that the compiler's s_waitcnt counting is exact:
    // with `if (tile exists)` around the loads its vmcnt for a fragment also waited for the loads issued AFTER it.
    auto load_next = [&](el16x8 (&dst)[7]) {
        load_tile(tile_base(bl, yl, xl), dst);
        if (tl + STEM_WAVES <= t_last) { tl += STEM_WAVES; advance(bl, yl, xl); }        // scalar; stays on the last tile at the end
    };
#pragma unroll
    for (int j = 0; j < STEM_NB - 1; ++j) load_next(ring[j]);
    const int n_tiles = t <= t_last ? (t_last - t) / STEM_WAVES + 1 : 0;
    const int n_iter = (a.tiles_per_wg / STEM_WAVES + STEM_NB - 1) / STEM_NB;            // the same for every wave of the grid
    int done = 0;
    for (int it = 0; it < n_iter; ++it) {
#pragma unroll
        for (int j = 0; j < STEM_NB; ++j) {
            load_next(ring[(j + STEM_NB - 1) % STEM_NB]);
            el16x8 (&af)[7] = ring[j];
            const bool valid = done < n_tiles;                                            // wave-uniform
            const int xo0 = tx * 16;
            f32x4v acc[4];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) acc[mt] = f32x4v{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int r = 0; r < 7; ++r)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) acc[mt] = rn::mma<DT>::m16(wf[mt][r], af[r], acc[mt]);
            // result lane (pixel p, group g): channels mt * 16 + 4 g + 0..3 of pixel xo0 + p
            const bool live = valid && xo0 + p < a.Wo;                                    // (lanes past the row end hold garbage, possibly NaN)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const uint32_t lo = rn::dt<DT>::pk(acc[mt][0], acc[mt][1]), hi = rn::dt<DT>::pk(acc[mt][2], acc[mt][3]);
                *(rn::u32x2 *)(strip + p * 34 + mt * 8 + g * 2) = rn::u32x2{lo, hi};
                const uint32_t lo_l = live ? lo : 0u, hi_l = live ? hi : 0u;
                const float v0 = rn::mma<DT>::lo(lo_l), v1 = rn::mma<DT>::hi(lo_l);
                const float v2 = rn::mma<DT>::lo(hi_l), v3 = rn::mma<DT>::hi(hi_l);
                ssum[mt][0] += v0; ssq[mt][0] = fmaf(v0, v0, ssq[mt][0]);
                ssum[mt][1] += v1; ssq[mt][1] = fmaf(v1, v1, ssq[mt][1]);
                ssum[mt][2] += v2; ssq[mt][2] = fmaf(v2, v2, ssq[mt][2]);
                ssum[mt][3] += v3; ssq[mt][3] = fmaf(v3, v3, ssq[mt][3]);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // 16 pixels x 128 bytes leave as two rounds of 64 lanes x 16 bytes (8 pixels each)
            uint16_t *yrow = a.y + (((int64_t)b * a.Ho + yo) * a.Wo + xo0) * 64;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int px = h * 8 + (lane >> 3), chunk = lane & 7;
                const rn::u32x2 v01 = *(const rn::u32x2 *)(strip + px * 34 + chunk * 4), v23 = *(const rn::u32x2 *)(strip + px * 34 + chunk * 4 + 2);
                const rn::u32x4 v = {v01.x, v01.y, v23.x, v23.y};
                if (valid && xo0 + px < a.Wo) __builtin_nontemporal_store(v, (rn::u32x4 *)(yrow + px * 64 + chunk * 8));   // (plain stores: 101 us, these: 81)
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (done + 1 < n_tiles) advance(b, yo, tx);                                   // scalar
            ++done;
        }
    }

    if (a.partial) {
        // sums over the 16 pixel lanes (same g), then over the waves
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float s = ssum[mt][r], q = ssq[mt][r];
#pragma unroll
                for (int d = 1; d < 16; d <<= 1) { s += __shfl_xor(s, d, RN_WAVE); q += __shfl_xor(q, d, RN_WAVE); }
                if (p == 0) { s_stat[wave][0][mt * 16 + 4 * g + r] = s; s_stat[wave][1][mt * 16 + 4 * g + r] = q; }
            }
        __syncthreads();
        if (threadIdx.x < 128) {
            const int which = threadIdx.x >> 6, c = threadIdx.x & 63;
            float tsum = 0.0f;
#pragma unroll
            for (int w = 0; w < STEM_WAVES; ++w) tsum += s_stat[w][which][c];
            a.partial[((int64_t)blockIdx.x * 2 + which) * 64 + c] = tsum;
        }
    }
}

// ---- weight gradient -----------------------------------------------------------------------------------------------------------
// dW[ch][r][k] = sum over output pixels of G[px][ch] * xp[2 yo + r][2 xo .. ][k]   (k = 4 * tap pixel + channel, as above).
// The contraction index is the output pixel, so BOTH MFMA operands are pixel-strided in memory; they are staged as they lie --
// a stage = 64 consecutive output pixels of one row: the 64 x 64 gradient rows and, per kernel row r, ONE raw strip of the padded
// image (134 pixels x 8 bytes: the 64 pixels' 32-element windows overlap, start 16 bytes apart) -- and read with the transposing
// ds_read_b64_tr_b16, whose lanes supply their own row addresses: a 16-byte row pitch over the raw strip IS the im2col matrix.
// Wave (mp, kp) of the 4 owns channel tiles {2 mp, 2 mp + 1} x k-tiles {(r, kp)}: 14 accumulator tiles, 18 fragment reads per 14
// MFMAs and 32-pixel k-step.  One workgroup walks a contiguous range of stages and writes one f32 partial [64][7][32]; a second
// kernel sums the partials into the [64][7][7][3] bf16 gradient.
constexpr int SW_STAGE = 64;                     // output pixels per stage (two 32-deep k-steps)
constexpr int SW_GPITCH = 136;                   // bytes per staged gradient row (128 + 8: the 4 rows of a transposing read land on different banks)
constexpr int SW_SPITCH = 1088;                  // bytes per staged strip (134 pixels x 8 = 1072, rounded up to 16-byte chunks of 68)
constexpr int SW_GBYTES = SW_STAGE * SW_GPITCH, SW_BUF = SW_GBYTES + 7 * SW_SPITCH;

struct StemWgradArgs {
    const uint16_t *g;      // [B][Ho][Wo][64] bf16: gradient at the conv output (BN: at the output of relu(bn1(.)) instead)
    const uint16_t *xp;     // [B][Hp2][Wpp][4]
    float *partial;         // [gridDim.x][64][7][32]
    int B, Ho, Wo, Hp2, Wpp;
    int tiles_x, total_stages, stages_per_wg;
    // BN: the BatchNorm + ReLU backward of norm.hip's bn_bwd_apply_kernel<DT, 2, false> in the operand load -- z = the conv output,
    // g' = g * [fma(z, fa, fb) alive], conv-output gradient = round_DT(fma(ba, g', fma(bk1, z, bk0))): the apply pass and its tensor go away
    const uint16_t *z;
    const float *ba, *bk0, *bk1, *fa, *fb;      // [64] each
};

template <int DT, bool BN>
__global__ __launch_bounds__(STEM_THREADS) void stem_wgrad_kernel(const StemWgradArgs a)
{
    typedef typename rn::mma<DT>::frag el16x8;             // (8 consecutive 16-bit elements: bf16 or fp16)
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * SW_BUF];
    typedef __attribute__((ext_vector_type(4))) short s16x4;
    typedef __attribute__((address_space(3))) s16x4 *lds_s16x4;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mp = wave >> 1, kp = wave & 1;
    const int grp = lane >> 4, q = (lane >> 2) & 3, p4 = lane & 3;   // transposing read: group grp reads pixels 8 grp + q (+ 4), columns 4 p4 ..

    f32x4v acc[2][7];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 7; ++r) acc[i][r] = f32x4v{0.0f, 0.0f, 0.0f, 0.0f};

    const int s_beg = blockIdx.x * a.stages_per_wg, s_end = min(s_beg + a.stages_per_wg, a.total_stages);
    // staging: thread -> two 16-byte chunks of the gradient rows (512 chunks) and two of the strips (7 x 67 = 469 chunks)
    rn::u32x4 rg[2], rs[2], rz[2];
    bool gval[2] = {false, false};
    float cba[8], cbk0[8], cbk1[8], cfa[8], cfb[8];                       // BN: this thread's 8 channels (chunk c -> channels 8 (c % 8) .., c % 8 = tid % 8)
    if (BN) {
        const int ch = (tid & 7) * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) { cba[j] = a.ba[ch + j]; cbk0[j] = a.bk0[ch + j]; cbk1[j] = a.bk1[ch + j]; cfa[j] = a.fa[ch + j]; cfb[j] = a.fb[ch + j]; }
    }
    const float alive = __uint_as_float(DT == RN_F16 ? 0x33000000u : 0x00004000u);      // norm.hip: relu_alive_threshold<DT>
    auto fetch = [&](const int s) {
        const int tx = s % a.tiles_x, rowid = s / a.tiles_x, yo = rowid % a.Ho, b = rowid / a.Ho, x0 = tx * SW_STAGE;
        const uint16_t *grow = a.g + (((int64_t)b * a.Ho + yo) * a.Wo + x0) * 64;
        const rn::u32x4 zero4 = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = i * STEM_THREADS + tid, px = c >> 3;                    // chunk c: pixel c / 8, channels 8 (c % 8) ..
            gval[i] = x0 + px < a.Wo;
            rg[i] = gval[i] ? *(const rn::u32x4 *)(grow + c * 8) : zero4; // pixels past the row end contribute nothing
            if (BN) rz[i] = gval[i] ? *(const rn::u32x4 *)(a.z + (grow - a.g) + c * 8) : zero4;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = i * STEM_THREADS + tid, r = c / 67, cc = c - r * 67;
            const uint16_t *src = a.xp + (((int64_t)b * a.Hp2 + 2 * yo + r) * a.Wpp + 2 * x0) * 4 + cc * 8;
            rs[i] = (c < 7 * 67) ? *(const rn::u32x4 *)src : zero4;
        }
    };
    auto commit = [&](unsigned char *buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = i * STEM_THREADS + tid, px = c >> 3, ch8 = c & 7;
            if (BN) {
                float g8[8], z8[8];
                rn::dt<DT>::unpack(rg[i], g8);
                rn::dt<DT>::unpack(rz[i], z8);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float gj = (fmaf(z8[j], cfa[j], cfb[j]) > alive) ? g8[j] : 0.0f;
                    g8[j] = fmaf(cba[j], gj, fmaf(cbk1[j], z8[j], cbk0[j]));
                }
                rg[i] = gval[i] ? rn::dt<DT>::pack(g8) : rn::u32x4{0u, 0u, 0u, 0u};
            }
            // (136-byte rows: 16-byte chunks are only 8-byte aligned -> two 8-byte stores)
            *(rn::u32x2 *)(buf + px * SW_GPITCH + ch8 * 16) = rn::u32x2{rg[i].x, rg[i].y};
            *(rn::u32x2 *)(buf + px * SW_GPITCH + ch8 * 16 + 8) = rn::u32x2{rg[i].z, rg[i].w};
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = i * STEM_THREADS + tid, r = c / 67, cc = c - r * 67;
            if (c < 7 * 67) *(rn::u32x4 *)(buf + SW_GBYTES + r * SW_SPITCH + cc * 16) = rs[i];
        }
    };
    auto tr_frag = [&](const unsigned char *base, const int rowb) {      // pixels +0..3 and +4..7 of this lane group's 8
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(base));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(base + 4 * rowb));
        return __builtin_bit_cast(el16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };

    if (s_beg < s_end) { fetch(s_beg); commit(lds); }
    __syncthreads();
    for (int s = s_beg; s < s_end; ++s) {
        unsigned char *cur = lds + ((s - s_beg) & 1) * SW_BUF, *nxt = lds + (((s - s_beg) & 1) ^ 1) * SW_BUF;
        if (s + 1 < s_end) fetch(s + 1);                                   // next stage's global loads under this stage's MFMAs
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int px = ks * 32 + 8 * grp + q;                           // this lane's row of the transposing reads
            el16x8 gf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) gf[i] = tr_frag(cur + px * SW_GPITCH + ((2 * mp + i) * 16 + 4 * p4) * 2, SW_GPITCH);
#pragma unroll
            for (int r = 0; r < 7; ++r) {
                const el16x8 xf = tr_frag(cur + SW_GBYTES + r * SW_SPITCH + px * 16 + (kp * 16 + 4 * p4) * 2, 16);
#pragma unroll
                for (int i = 0; i < 2; ++i) acc[i][r] = rn::mma<DT>::m16(gf[i], xf, acc[i][r]);
            }
        }
        if (s + 1 < s_end) commit(nxt);
        __syncthreads();
    }
    // D lane: column (k) = lane & 15, rows (channels) 4 (lane >> 4) + j
    float *out = a.partial + (int64_t)blockIdx.x * 64 * 7 * STEM_KROW;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 7; ++r)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int ch = (2 * mp + i) * 16 + 4 * (lane >> 4) + j;
                out[(ch * 7 + r) * STEM_KROW + kp * 16 + (lane & 15)] = acc[i][r][j];
            }
}

// dw [64][7][7][3] bf16 = sum of the partials (k = 4 * px + c; the pad columns are dropped).  Block = 16 outputs x 16 slices of
// the partials, combined through LDS in a fixed order (a thread per output summing 512 partials alone took 45 us).
template <int DT>
__global__ __launch_bounds__(256) void stem_wgrad_reduce_kernel(const float *__restrict__ partial, const int n, uint16_t *__restrict__ dw)
{
    __shared__ float sh[16][17];
    const int o = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int i = blockIdx.x * 16 + o;                                      // over [64][7][32]
    float s0 = 0.0f, s1 = 0.0f;
    if (i < 64 * 7 * STEM_KROW) {
        int b = sl;
        for (; b + 16 < n; b += 32) { s0 += partial[(int64_t)b * 64 * 7 * STEM_KROW + i]; s1 += partial[(int64_t)(b + 16) * 64 * 7 * STEM_KROW + i]; }
        for (; b < n; b += 16) s0 += partial[(int64_t)b * 64 * 7 * STEM_KROW + i];
    }
    sh[sl][o] = s0 + s1;
    __syncthreads();
    if (sl == 0 && i < 64 * 7 * STEM_KROW) {
        float t = 0.0f;
#pragma unroll
        for (int j = 0; j < 16; ++j) t += sh[j][o];
        const int k = i % STEM_KROW, r = (i / STEM_KROW) % 7, ch = i / (7 * STEM_KROW), px = k >> 2, c = k & 3;
        if (px < 7 && c < 3) dw[((ch * 7 + r) * 7 + px) * 3 + c] = rn::mma<DT>::dn(t);
    }
}

int stem_wgrad_grid(const int total_stages, int *stages_per_wg)
{
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
    int wgs = cus * 2;
    if (wgs > total_stages) wgs = total_stages;
    if (wgs < 1) wgs = 1;
    const int spw = (total_stages + wgs - 1) / wgs;
    *stages_per_wg = spw;
    return (total_stages + spw - 1) / spw;
}

int stem_grid(const int total_tiles, int *tiles_per_wg)
{
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
    int wgs = cus * 2;                                              // one resident workgroup (4 waves, one per SIMD) per CU
    if (wgs > (total_tiles + STEM_WAVES - 1) / STEM_WAVES) wgs = (total_tiles + STEM_WAVES - 1) / STEM_WAVES;
    if (wgs < 1) wgs = 1;
    int tpw = (total_tiles + wgs - 1) / wgs;
    tpw = ((tpw + STEM_WAVES - 1) / STEM_WAVES) * STEM_WAVES;
    *tiles_per_wg = tpw;
    return (total_tiles + tpw - 1) / tpw;
}

}  // namespace

RN_API size_t rn_stem_padded_bytes(int B, int H, int W)
{
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    const int Wpp = (W + 6 + 1) & ~1;
    // + 2 rows of slack: the masked lanes of a row's last tile read (never use) up to 36 pixels past the row end
    return (((size_t)B * (H + 6) + 2) * (size_t)Wpp + STEM_SLACK_PX) * 8;
}

RN_API int rn_stem_partial_rows(int B, int H, int W)
{
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    int tpw = 0;
    return stem_grid(B * Ho * ((Wo + 15) / 16), &tpw);
}

RN_API int rn_stem_conv_forward(const void *x, const void *w, void *xp, void *wk, void *y, float *partial, int dtype, int B, int H, int W,
                                void *stream)
{
    if (!x || !w || !xp || !wk || !y || B <= 0 || H <= 0 || W <= 0) return RN_EINVAL;
    if (dtype != RN_BF16 && dtype != RN_F16) return RN_EUNSUPPORTED;
    if ((int64_t)B * (H + 6) * (W + 8) >= ((int64_t)1 << 31)) return RN_EUNSUPPORTED;
    if (!rn::aligned(xp, 16) || !rn::aligned(wk, 16) || !rn::aligned(y, 16)) return RN_EALIGN;
    if (!rn::aligned(x, 2) || !rn::aligned(w, 2)) return RN_EALIGN;              // (read element by element: bf16 alignment is all they need)
    hipStream_t st = (hipStream_t)stream;
    StemPadArgs pa;
    pa.x = (const uint16_t *)x; pa.xp = (uint16_t *)xp; pa.B = B; pa.H = H; pa.W = W; pa.Hp2 = H + 6; pa.Wpp = (W + 6 + 1) & ~1;
    const int64_t px = (int64_t)B * pa.Hp2 * pa.Wpp;
    int64_t blocks = (px + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(stem_pad_kernel, dim3((unsigned)blocks), dim3(256), 0, st, pa);
    RN_LAUNCH_CHECK();
    hipLaunchKernelGGL(stem_weight_pack_kernel, dim3((64 * 7 * STEM_KROW + 255) / 256), dim3(256), 0, st, (const uint16_t *)w, (uint16_t *)wk);
    RN_LAUNCH_CHECK();
    StemArgs a;
    a.xp = (const uint16_t *)xp; a.wk = (const uint16_t *)wk; a.y = (uint16_t *)y; a.partial = partial;
    a.B = B; a.Ho = (H - 1) / 2 + 1; a.Wo = (W - 1) / 2 + 1; a.Hp2 = pa.Hp2; a.Wpp = pa.Wpp;
    a.tiles_x = (a.Wo + 15) / 16; a.total_tiles = B * a.Ho * a.tiles_x;
    const int wgs = stem_grid(a.total_tiles, &a.tiles_per_wg);
    if (dtype == RN_F16) hipLaunchKernelGGL(stem_fwd_kernel<RN_F16>, dim3((unsigned)wgs), dim3(STEM_THREADS), 0, st, a);
    else hipLaunchKernelGGL(stem_fwd_kernel<RN_BF16>, dim3((unsigned)wgs), dim3(STEM_THREADS), 0, st, a);
    RN_LAUNCH_CHECK();
    return RN_OK;
}

RN_API size_t rn_stem_wgrad_workspace_bytes(int B, int H, int W)
{
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    int spw = 0;
    const int wgs = stem_wgrad_grid(B * Ho * ((Wo + SW_STAGE - 1) / SW_STAGE), &spw);
    return (size_t)wgs * 64 * 7 * STEM_KROW * sizeof(float);
}

static int stem_wgrad_impl(const void *g, const void *z, const float *coef3, const float *fwd_coef, const void *xp, void *dw, int dtype, int B,
                           int H, int W, void *workspace, size_t workspace_bytes, void *stream);

RN_API int rn_stem_conv_wgrad(const void *g, const void *xp, void *dw, int dtype, int B, int H, int W, void *workspace,
                              size_t workspace_bytes, void *stream)
{
    return stem_wgrad_impl(g, nullptr, nullptr, nullptr, xp, dw, dtype, B, H, W, workspace, workspace_bytes, stream);
}

RN_API int rn_stem_conv_wgrad_bn(const void *g, const void *z, const float *coef3, const float *fwd_coef, const void *xp, void *dw, int dtype,
                                 int B, int H, int W, void *workspace, size_t workspace_bytes, void *stream)
{
    if (!z || !coef3 || !fwd_coef) return RN_EINVAL;
    if (!rn::aligned(z, 16)) return RN_EALIGN;
    return stem_wgrad_impl(g, z, coef3, fwd_coef, xp, dw, dtype, B, H, W, workspace, workspace_bytes, stream);
}

static int stem_wgrad_impl(const void *g, const void *z, const float *coef3, const float *fwd_coef, const void *xp, void *dw, int dtype, int B,
                           int H, int W, void *workspace, size_t workspace_bytes, void *stream)
{
    if (!g || !xp || !dw || !workspace || B <= 0 || H <= 0 || W <= 0) return RN_EINVAL;
    if (dtype != RN_BF16 && dtype != RN_F16) return RN_EUNSUPPORTED;
    if (workspace_bytes < rn_stem_wgrad_workspace_bytes(B, H, W)) return RN_EWORKSPACE;
    if (!rn::aligned(g, 16) || !rn::aligned(xp, 16) || !rn::aligned(workspace, 16)) return RN_EALIGN;
    StemWgradArgs a = {};
    a.g = (const uint16_t *)g; a.xp = (const uint16_t *)xp; a.partial = (float *)workspace;
    if (z) { a.z = (const uint16_t *)z; a.ba = coef3; a.bk0 = coef3 + 64; a.bk1 = coef3 + 128; a.fa = fwd_coef; a.fb = fwd_coef + 64; }
    a.B = B; a.Ho = (H - 1) / 2 + 1; a.Wo = (W - 1) / 2 + 1; a.Hp2 = H + 6; a.Wpp = (W + 6 + 1) & ~1;
    a.tiles_x = (a.Wo + SW_STAGE - 1) / SW_STAGE; a.total_stages = B * a.Ho * a.tiles_x;
    const int wgs = stem_wgrad_grid(a.total_stages, &a.stages_per_wg);
    hipStream_t st = (hipStream_t)stream;
    if (z) {
        if (dtype == RN_F16) hipLaunchKernelGGL((stem_wgrad_kernel<RN_F16, true>), dim3((unsigned)wgs), dim3(STEM_THREADS), 0, st, a);
        else hipLaunchKernelGGL((stem_wgrad_kernel<RN_BF16, true>), dim3((unsigned)wgs), dim3(STEM_THREADS), 0, st, a);
    } else {
        if (dtype == RN_F16) hipLaunchKernelGGL((stem_wgrad_kernel<RN_F16, false>), dim3((unsigned)wgs), dim3(STEM_THREADS), 0, st, a);
        else hipLaunchKernelGGL((stem_wgrad_kernel<RN_BF16, false>), dim3((unsigned)wgs), dim3(STEM_THREADS), 0, st, a);
    }
    RN_LAUNCH_CHECK();
    if (dtype == RN_F16) hipLaunchKernelGGL(stem_wgrad_reduce_kernel<RN_F16>, dim3((64 * 7 * STEM_KROW + 15) / 16), dim3(256), 0, st, (const float *)workspace, wgs, (uint16_t *)dw);
    else hipLaunchKernelGGL(stem_wgrad_reduce_kernel<RN_BF16>, dim3((64 * 7 * STEM_KROW + 15) / 16), dim3(256), 0, st, (const float *)workspace, wgs, (uint16_t *)dw);
    RN_LAUNCH_CHECK();
    return RN_OK;
}
