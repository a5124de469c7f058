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
// (8 x 4 v_mfma_f32_16x16x32_bf16 tiles, 128 accumulator registers); operands staged global -> LDS by 16-byte
// LDS-DMA (global_load_lds) into 128-byte rows whose 16-byte chunks are XOR-swizzled with (row >> 1) & 7 (source-side
// swizzle + the same XOR on the ds_read_b128: conflict-free for its 16-lane groups); 3 activation stages + 2 weight
// stages = 160 KiB of LDS, activations prefetched two K-tiles ahead, weights one.  The two waves of every SIMD run
// in PING-PONG: waves 0-3 and 4-7 are one barrier interval apart, so while one group issues its 32 MFMAs the other
// reads the fragments of its next 32 (12 ds_read_b128) and issues 4 LDS-DMA pieces; a K-tile is two (load, MFMA)
// phase pairs = 4 barriers.  LDS-DMA retirement: counted vmcnt at the end of the second load phase, one barrier
// before any wave of either group reads the tile; raw s_barrier (a __syncthreads would drain the DMA).
// Instruction shape: the same loop on 32x32x16 tiles (round 2) issues the same number of MFMA cycles and LDS reads but
// ran 5 - 17 % slower on random data -- the part holds a higher clock under the 16x16x32 shape (rocprof: same busy
// cycles, shorter duration).  The weight-gradient kernel below was tried on 16x16x32 as well and was 5 - 9 % SLOWER
// there (A/B on one box: towers 380 -> 398 us, class-output 955 -> 1046 us), so it keeps 32x32x16.
// One wave per SIMD instead of ping-pong (4 waves of 128 x 128, 256-register accumulators, fragment reads and LDS-DMA pieces issued
// between the MFMAs of the previous k-step, one barrier per K-tile) was prototyped in round 3 on the same LDS image: bit-identical
// results, 555 us against 395 us for the tower pair -- a lone wave pays the ~60-cycle issue cost of each of its 16 LDS-DMA pieces per
// K-tile inside its own MFMA stream; with two waves per SIMD the partner absorbs it.  Ping-pong stays.
// Staging instruction form: `buffer_load_dwordx4 ... lds` (descriptor + SGPR offset + 32-bit per-thread offset, no 64-bit VALU add per
// piece) instead of `global_load_lds_dwordx4` was tried in round 3: towers 389 -> 393 us, class-output data gradient 519 -> 571 us.
// Tile rounds: 1 626 equal tiles on 256 CUs fill 6.35 rounds and take 7; running the last 90 as 180 half tiles (128 rows, 8
// waves of 64 x 64, one extra launch) was built and tested in round 3 and changed nothing (386 -> 389 us): a half tile takes as
// long as a full one's share of the round, i.e. the launch is bound chip-wide (power / clock), not by the per-CU tile schedule.
// Measured on MI355X, random data, two [8,153,170,256] -> 256 towers per launch: forward 377 - 396 us (1070 - 1120 TFLOP/s
// on the canvas positions), data gradient 356 - 370 us; MIOpen: forward 301-315 us per tower + 32 us for the separate
// bias/ReLU/mask pass, data gradient 415 us.
#include "rn_common.hpp"
#include <type_traits>
#include <cstdlib>

#ifndef BAND_NARROW_FWD
#define BAND_NARROW_FWD 1            // the <= 64-column TO_LEVELS launches on conv3x3_band_narrow_kernel (0: the tile kernel's NARROW variant)
#endif

namespace {

using rn::f32x16;
typedef __attribute__((address_space(3))) void *lds_void_ptr;

constexpr int CONV_BM = 256, CONV_BN = 256, CONV_BK = 64, CONV_THREADS = 512;
constexpr int CONV_LDS_BYTES = 5 * CONV_BM * CONV_BK * 2;        // 160 KiB
#define SWZ(row) (((row) >> 1) & 7)

constexpr int CONV_MAX_PROBLEMS = 4;

// The pyramid levels as they sit on the canvas, and their DENSE per-level tensors [N][h][w][row_elems] (the layout the
// loss / detection kernels stream: [N][h*w*9][K] with row_elems = 9 K).  Used by the class-output conv, whose
// 9 x 90 = 810 output channels are written densely (no dead classes), and by its gradients.
// A canvas SHEET [Hp][Wp] carries `slots` images (the second image's P4 fills the space beside the first one's, ...), so the
// placement is a table: map[position on the sheet] = -1 for border / gap positions, else
//     (slot << 28) | (tensor << 24) | (y * w + x)        -- position (y, x) of image sheet * slots + slot in tensor `tensor`.
constexpr int CONV_LEVELS = 6;
struct LevelSet {
    const int32_t *map;                 // [HWp], device memory
    int32_t row_elems, slots, n_images, T;
    int32_t hw[CONV_LEVELS];            // h * w of each tensor
    uint16_t *ptr[CONV_LEVELS];
};

// Sheet index and position on the sheet of canvas position m < 2^22, without integer division: (m + 0.5) * (1 / d) is at
// least 0.5 / d away from an integer, far more than the rounding error of the two float operations at these magnitudes
__device__ __forceinline__ void sheet_coords(const int m, const int HWp, int &n, int &pos)
{
    n = (int)(((float)m + 0.5f) * (1.0f / (float)HWp));
    pos = m - n * HWp;
}

// The tensors' base pointers and sizes pinned in SGPRs (readfirstlane): indexed by a per-lane tensor number straight from the
// kernel arguments, the compiler turns `ptr[t]` into a vector load from the argument segment, and waits for it with
// s_waitcnt vmcnt(0) inside the MFMA loops.
struct LevelRegs { uint32_t lo[CONV_LEVELS], hi[CONV_LEVELS]; int32_t hw[CONV_LEVELS]; };
__device__ __forceinline__ LevelRegs level_regs(const LevelSet &ls)
{
    LevelRegs r;
#pragma unroll
    for (int l = 0; l < CONV_LEVELS; ++l) {
        const unsigned long long u = (unsigned long long)(uintptr_t)ls.ptr[l];
        r.lo[l] = __builtin_amdgcn_readfirstlane((uint32_t)u);
        r.hi[l] = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
        r.hw[l] = __builtin_amdgcn_readfirstlane(ls.hw[l]);
    }
    return r;
}

// Row of the dense tensor behind map entry `e` on sheet n; null for gaps, for the unused slots of the last sheet and for
// entries outside the tensor (a wrong table cannot make the kernels read or write out of bounds).
__device__ __forceinline__ uint16_t *level_row(const LevelSet &ls, const LevelRegs &lr, const int n, const int32_t e)
{
    if (e < 0) return nullptr;
    const int t = (e >> 24) & 15, img = n * ls.slots + ((e >> 28) & 7), local = e & 0xffffff;
    uint32_t lo = lr.lo[0], hi = lr.hi[0];
    int hw = lr.hw[0];
#pragma unroll
    for (int l = 1; l < CONV_LEVELS; ++l) {
        lo = t == l ? lr.lo[l] : lo;
        hi = t == l ? lr.hi[l] : hi;
        hw = t == l ? lr.hw[l] : hw;
    }
    if (t >= ls.T || img >= ls.n_images || local >= hw) return nullptr;
    uint16_t *base = (uint16_t *)(uintptr_t)(((unsigned long long)hi << 32) | lo);
    return base + ((int64_t)img * hw + local) * ls.row_elems;
}

// Map fetches inside the MFMA loops must not make the compiler wait for the LDS-DMA pieces in flight (its own s_waitcnt for
// a loop-carried load is vmcnt(0)): the fetch is issued by hand one step ahead (map_fetch: branch-free, the caller clamps the
// position and keeps the validity), it is older than the 4 pieces issued after it, so the loops' counted
// `s_waitcnt vmcnt(4)` retires it, and map_landed() right after that wait hands the register back to the compiler.
__device__ __forceinline__ void map_fetch(int32_t &dst, const int32_t *p) { asm volatile("global_load_dword %0, %1, off" : "=v"(dst) : "v"(p) : "memory"); }
__device__ __forceinline__ void map_landed(int32_t &dst) { asm volatile("" : "+v"(dst) :: "memory"); }

// Cross-lane pointer exchange inside the MFMA loops.  NOT __shfl: that is a ds_bpermute_b32, which the compiler counts as an
// LDS read that may alias the LDS-DMA pieces in flight and protects with s_waitcnt vmcnt(0) -- a full drain of the staging
// pipeline per call (the gathering weight-gradient kernel did that once per K-tile in round 2's first version).
// bperm_ptr: the same instruction behind asm (it does not touch LDS memory); pair_ptr: v_readlane for the two-source case.
__device__ __forceinline__ const uint16_t *bperm_ptr(const uint16_t *p, const int src_lane)
{
    const unsigned long long u = (unsigned long long)(uintptr_t)p;
    unsigned lo, hi;
    const unsigned sel = (unsigned)src_lane << 2;
    asm volatile("ds_bpermute_b32 %0, %2, %3\n\tds_bpermute_b32 %1, %2, %4\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(lo), "=&v"(hi) : "v"(sel), "v"((unsigned)u), "v"((unsigned)(u >> 32)) : "memory");
    return (const uint16_t *)(uintptr_t)(((unsigned long long)hi << 32) | lo);
}
template <int LANE_A, int LANE_B>
__device__ __forceinline__ const uint16_t *pair_ptr(const uint16_t *p, const bool second)
{
    const unsigned long long u = (unsigned long long)(uintptr_t)p;
    const unsigned lo_a = __builtin_amdgcn_readlane((unsigned)u, LANE_A), hi_a = __builtin_amdgcn_readlane((unsigned)(u >> 32), LANE_A);
    const unsigned lo_b = __builtin_amdgcn_readlane((unsigned)u, LANE_B), hi_b = __builtin_amdgcn_readlane((unsigned)(u >> 32), LANE_B);
    const unsigned long long a = ((unsigned long long)hi_a << 32) | lo_a, b = ((unsigned long long)hi_b << 32) | lo_b;
    return (const uint16_t *)(uintptr_t)(second ? b : a);
}

// Kernel modes.  CANVAS: canvas in, canvas out (the head towers, forward and data gradient).  TO_LEVELS: canvas in, dense
// per-level tensors out, any even Cout (weight rows >= Cout read zeros; the class-output conv forward).  FROM_LEVELS: the
// activation operand is gathered from dense per-level tensors whose row length need not be a multiple of 64 (the data
// gradient of the class-output conv: contraction over 9 taps x 810 channels), canvas out.
// DENSE: up to 4 convolutions with their OWN geometry and weights in one launch, activations and outputs as plain dense
// [N][h][w][C] tensors without any border (the FPN's 3x3 output convs on P3 / P4 / P5, retinanet/layers.py:34-38, 62-64): the
// tile list is the concatenation of the problems' row tiles (blockIdx.x -> problem by DenseGeom::tile_beg), and a tap that
// leaves its image reads the zero page instead -- a per-thread 9-bit validity mask for each of the 4 rows it stages, computed
// once per tile (valid taps are always inside the tensor, so no address clamping exists in this mode).
enum { MODE_CANVAS = 0, MODE_TO_LEVELS = 1, MODE_FROM_LEVELS = 2, MODE_DENSE = 3 };
struct DenseGeom {
    int64_t M[CONV_MAX_PROBLEMS];               // N * h * w
    int32_t h[CONV_MAX_PROBLEMS], w[CONV_MAX_PROBLEMS];
    int32_t tile_beg[CONV_MAX_PROBLEMS + 1];    // first row tile (weight-gradient kernel: first split) of problem p; INT_MAX past the last
};

// Up to 4 convolutions of identical geometry in one launch (blockIdx.z = problem): the cls and the box tower run the
// same shapes side by side, and 2 x 813 tiles fill 7 waves of 256 workgroups where two launches take 2 x 4.
struct ConvArgs {
    const uint16_t *Xs[CONV_MAX_PROBLEMS];      // [M][Cin] bf16 (unused in FROM_LEVELS mode)
    const uint16_t *Ws[CONV_MAX_PROBLEMS];      // [Cout][9][Cin] bf16
    const float *biases[CONV_MAX_PROBLEMS];     // [Cout] or null
    uint16_t *Ys[CONV_MAX_PROBLEMS];            // [M][Cout] bf16 (unused in TO_LEVELS mode)
    const uint8_t *mask;    // [HWp] or null (1 = keep)
    const uint16_t *zeros;  // >= 256 B of zeros: weight rows past Cout, gap positions and channel chunks past the row end
    // data-gradient epilogue of a conv whose INPUT was a ReLU output (the head towers): the result is also multiplied by
    // [relu_src > 0] (as a bit mask the forward conv of the layer below wrote) and its column sums -- the bias gradient of the layer below -- go to
    // colsum[p] as one partial row per row tile, [tiles_m][Cout] f32; both null otherwise
    const uint8_t *relu_masks[CONV_MAX_PROBLEMS];  // [M][Cout / 8] bytes: bit j of byte (m, c / 8) = [relu_src[m][c + j] > 0]
    float *colsums[CONV_MAX_PROBLEMS];
    uint8_t *relu_mask_outs[CONV_MAX_PROBLEMS];    // forward with ReLU: the same mask of the OUTPUT, written by the epilogue (or null)
    int64_t M, HWp;
    int Cin, Cout, Wp, relu;    // Cin = channels walked per tap (FROM_LEVELS: the padded row length, a multiple of 64)
    int n_base;                 // first output channel of blockIdx.y = 0 (the NARROW launch of the last column tile)
    int f16;                    // host side only: element type fp16 instead of bf16 (selects the kernel instantiation)
    LevelSet lv;
    DenseGeom dn;               // MODE_DENSE only (M, HWp, Wp, mask unused there)
    // MODE_DENSE, ksplit > 1: the K walk (channel chunks x taps) is cut into ksplit contiguous ranges, blockIdx.z = range; every workgroup
    // writes its f32 partial tile to kpartial[z][M][Cout] and dense_ksplit_reduce_kernel sums them (a convolution with few row tiles
    // -- conv2 of the layer4 bottlenecks: 8 400 positions, 66 workgroups of 72 K-tiles -- fills the chip with 3 x 66 of 24)
    int ksplit;
    float *kpartial;
    // xcd_ny > 0 (set by conv_launch_dt): a 1-D grid in x placed by XCD (workgroups go round-robin over the 8 XCDs: XCD = blockIdx.x % 8).
    // XCD c runs the CONSECUTIVE row tiles c * xcd_q + min(c, xcd_r) .. (xcd_q + (c < xcd_r) of them) with their xcd_ny column tiles next
    // to each other in time: the column tiles of a row tile read the same input rows, and a row tile's upper / lower taps are its
    // neighbours' centre rows, so an XCD's L2 takes an input line once instead of three XCDs fetching it each.
    int xcd_ny, xcd_q, xcd_r;
    // MODE_CANVAS, x_split > 0: the contraction runs over TWO activation tensors of x_ld channels each -- channel chunks 0 .. x_split - 1 from
    // Xs[p], the rest from X2s[p] -- against weights [Cout][9][Cin], Cin = both widths together: y = conv(x, w[.., :x_ld]) + conv(x2, w[.., x_ld:])
    // in one accumulator (the two towers' first-layer data gradients, which autograd would add)
    const uint16_t *X2s[CONV_MAX_PROBLEMS];
    int x_split, x_ld;
};
struct ConvProblem { const uint16_t *X, *W; const float *bias; uint16_t *Y; };

struct Walk { int tap, chunk; };      // position of the K walk

// NARROW (TO_LEVELS only): a column tile of at most 64 output channels -- the last, ragged tile of the 810-channel
// class-output conv (42 columns).  All 8 waves split the ROWS (32 each) and compute 32 x 64: a quarter of the MFMAs and a
// quarter of the weight staging of a full tile whose other 192 columns would be zeros.
template <int DT, int MODE, bool NARROW = false>
__global__ __launch_bounds__(CONV_THREADS) void conv3x3_canvas_kernel(const ConvArgs args)
{
    constexpr int MI = NARROW ? 1 : 4;                            // 32-row accumulator tiles per wave
    int prob = blockIdx.z, bx = blockIdx.x, by = blockIdx.y;
    if (args.xcd_ny > 0) {
        const int xcd = bx & 7, w = bx >> 3, mt = w / args.xcd_ny;
        by = w - mt * args.xcd_ny;
        if (mt >= args.xcd_q + (xcd < args.xcd_r)) return;
        bx = xcd * args.xcd_q + (xcd < args.xcd_r ? xcd : args.xcd_r) + mt;
    }
    int64_t m0 = (int64_t)bx * CONV_BM, M = args.M, HWp = args.HWp;
    int Wp = args.Wp, dense_h = 0;
    if (MODE == MODE_DENSE) {                                     // blockIdx.x walks the problems' row tiles one after the other
        prob = 0;
#pragma unroll
        for (int p = 1; p < CONV_MAX_PROBLEMS; ++p) prob = bx >= args.dn.tile_beg[p] ? p : prob;
        m0 = (int64_t)(bx - args.dn.tile_beg[prob]) * CONV_BM;
        M = args.dn.M[prob]; Wp = args.dn.w[prob]; dense_h = args.dn.h[prob]; HWp = (int64_t)Wp * dense_h;
    }
    ConvProblem a;
    a.X = args.Xs[prob]; a.W = args.Ws[prob]; a.bias = args.biases[prob]; a.Y = args.Ys[prob];
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];                 // [A0 A1 A2 | B0 B1] x 32 KiB
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);    // wave-uniform: LDS-DMA bases and wave roles live in SGPRs
    const int wm = NARROW ? wave : wave >> 2, wn = NARROW ? 0 : wave & 3;
    const int grp = wave >> 2;                                    // ping-pong group: waves 0-3 / 4-7 (one wave of each per SIMD)
    const int n0 = by * CONV_BN + args.n_base;
    const int cpt = args.Cin / CONV_BK, KT_ALL = 9 * cpt;
    int kt0 = 0, KT = KT_ALL;                                     // this workgroup's K-tiles: kt0 .. kt0 + KT - 1 of the walk
    if (MODE == MODE_DENSE && args.ksplit > 1) {
        const int z = blockIdx.z;
        kt0 = z * KT_ALL / args.ksplit;
        KT = (z + 1) * KT_ALL / args.ksplit - kt0;
    }
    constexpr int TILE = CONV_BM * CONV_BK * 2;
    // (Round 3 also built a BAND variant for the canvas / to-levels modes: one band of 256 + 2 positions per (channel chunk,
    // kernel row) staged once and read by the three horizontal taps with shifted fragment rows -- activation LDS-DMA traffic
    // 9 -> 3 per chunk.  Correct, and -3 % with the 32 x 32 x 16 instruction; +9 % slower than this loop with 16 x 16 x 32 (its
    // full vmcnt(0) drain at every band's last tap costs more when the matrix instruction leaves the partner wave half the
    // issue slots).  Ablations on it: no weight DMA -11 %, no band DMA -14 %, neither -28 % -- the tile is bound by the
    // fragment-read / MFMA / barrier cadence and the DMA ISSUE slots, not by staging bandwidth.  Removed; see git history.)
    unsigned char *const Abase = lds, *const Bbase = lds + 3 * TILE;
    // K walk.  CANVAS / TO_LEVELS: chunk outer, tap inner (a tile's input lines stay in L2 across its taps).
    // FROM_LEVELS: tap outer, chunk inner -- the gathered row pointers of a tap are computed once per 13 K-tiles.
    // The walk is kept as (tap, chunk, stage) counters -- no division in the loop.
    auto advance = [&](Walk &w) {
        if (MODE == MODE_FROM_LEVELS) { if (++w.chunk == cpt) { w.chunk = 0; ++w.tap; } }
        else { if (++w.tap == 9) { w.tap = 0; ++w.chunk; } }
    };

    // Matrix instruction: v_mfma_f32_16x16x32_bf16 (round 3; 32x32x16 before).  Same flop per cycle, same LDS image, same
    // number of fragment reads (12 ds_read_b128 per phase: 8 row tiles + 4 column tiles of one 32-deep k-step instead of
    // 2 x (4 + 2) of two 16-deep ones) -- but under load the chip holds a higher clock on this shape (MI355X_MICROARCH.md, DVFS
    // give-back item 7: +12 .. 15 % FLOP/s at equal cycles on random data).
    constexpr int MT = 2 * MI, NT = 4;                            // 16 x 16 accumulator tiles per wave: rows x columns
    rn::f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.0f;

    // fragment addresses: rows 32 apart share their swizzle ((row >> 1) & 7), so fragment mi / ni of a k-step is the
    // first one plus mi * 4096 bytes -- an immediate offset of the ds_read, not a register
    // (rows 16 apart share their swizzle ((row >> 1) & 7): tile mi / ni of a k-step is the first one plus mi * 2048 bytes, an
    // immediate offset of the ds_read.  Lane l reads row (l & 15), 16-byte chunk 4 s + (l >> 4) of k-step s: the 16 lanes of a
    // ds_read_b128 group land on 64 distinct banks with this swizzle, as the 32-row fragments did.)
    uint32_t a_off[2], b_off[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int chunk = ks * 4 + (lane >> 4);
        { const int row = wm * (32 * MI) + (lane & 15); a_off[ks] = row * 128 + ((chunk ^ SWZ(row)) << 4); }
        { const int row = wn * 64 + (lane & 15); b_off[ks] = row * 128 + ((chunk ^ SWZ(row)) << 4); }
    }
    const uint32_t lds_base = (uint32_t)(uintptr_t)lds;

    // FROM_LEVELS: this thread stages 4 rows of the activation tile (row = (i * 512 + tid) >> 3); their canvas coordinates,
    // and per tap the base pointer of the tapped position's dense row (null: border / gap / outside -> zeros)
    const uint16_t *grow[4] = {nullptr, nullptr, nullptr, nullptr};
    // A wave stages rows i * 64 + wave * 8 + (lane >> 3), i = 0..3: 32 distinct rows.  Lane j < 32 looks up row
    // (j >> 3) * 64 + wave * 8 + (j & 7) once per tap (9 times per tile) and the wave shares the pointers by shuffles.
    // The map entry of the NEXT tap is fetched when a tap's pointers are formed (map_fetch / map_landed above).
    LevelRegs lregs = {};
    if (MODE == MODE_TO_LEVELS || MODE == MODE_FROM_LEVELS) lregs = level_regs(args.lv);
    int g_n = 0, g_pos = 0;
    int32_t g_entry = -1;
    bool g_valid = false;
    auto tap_pos = [&](const int t) { return g_pos + (t / 3 - 1) * Wp + (t % 3 - 1); };
    if (MODE == MODE_FROM_LEVELS) {
        const int j = lane & 31;
        int64_t m = m0 + (j >> 3) * 64 + wave * 8 + (j & 7);
        m = m < M ? m : M - 1;
        sheet_coords((int)m, (int)HWp, g_n, g_pos);
        const int p0 = tap_pos(0);
        g_valid = p0 >= 0 && p0 < (int)HWp;
        g_entry = args.lv.map[g_valid ? p0 : 0];                  // (prologue: nothing in flight yet)
    }
    auto gather_tap = [&](const int t) {
        const uint16_t *mine = level_row(args.lv, lregs, g_n, g_valid ? g_entry : -1);
        const int pn = tap_pos(t + 1 < 9 ? t + 1 : t);
        g_valid = pn >= 0 && pn < (int)HWp;
        map_fetch(g_entry, args.lv.map + (g_valid ? pn : 0));
#pragma unroll
        for (int i = 0; i < 4; ++i) grow[i] = bperm_ptr(mine, i * 8 + (lane >> 3));
    };
    // Staging addresses.  A piece is one LDS-DMA wave instruction: 64 lanes x 16 bytes to consecutive LDS addresses from a
    // wave-uniform base (M0).  Fast path (every tile whose taps stay inside the buffer; weights: every tile but a ragged last
    // column tile): the source address is a wave-uniform 64-bit base -- tile origin + tap offset + channel chunk, SALU work
    // -- plus a per-thread 32-bit offset computed once per kernel; the old per-piece 64-bit multiply / clamp VALU chains
    // (~15 VALU each, 8 pieces per K-tile) made the load phases longer than the partner group's 16 MFMAs.
    uint32_t voff_a[4], voff_b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = i * CONV_THREADS + tid, row = q >> 3, cp = q & 7;
        voff_a[i] = (uint32_t)(row * (MODE == MODE_CANVAS && args.x_split > 0 ? args.x_ld : args.Cin) + ((cp ^ SWZ(row)) << 3)) * 2u;
        voff_b[i] = (uint32_t)(row * 9 * args.Cin + ((cp ^ SWZ(row)) << 3)) * 2u;
    }
    const bool a_edge = m0 - (Wp + 1) < 0 || m0 + CONV_BM + Wp + 1 > M;             // wave-uniform
    uint32_t tap_ok[4] = {0u, 0u, 0u, 0u};                         // DENSE: bit t = tap t of staged row i lies inside the row's image
    if (MODE == MODE_DENSE) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t m = m0 + ((i * CONV_THREADS + tid) >> 3);
            if (m < M) {
                const int yl = (int)m / Wp, x = (int)m - yl * Wp, y = yl % dense_h;      // (host: M < 2^31)
#pragma unroll
                for (int t = 0; t < 9; ++t)
                    tap_ok[i] |= ((unsigned)(y + t / 3 - 1) < (unsigned)dense_h && (unsigned)(x + t % 3 - 1) < (unsigned)Wp) ? 1u << t : 0u;
            }
        }
    }
    const bool b_ragged = MODE == MODE_TO_LEVELS && n0 + CONV_BN > args.Cout;
    auto lds_piece = [&](unsigned char *stage_base, const int i) {                                  // wave-uniform LDS base of piece i
        return stage_base + i * (CONV_THREADS * 16) + wave * (RN_WAVE * 16);
    };
    auto piece_a = [&](const Walk w, const int stage, const int i) {
        const int c0 = w.chunk * CONV_BK, t = w.tap;
        const int q = i * CONV_THREADS + tid, row = q >> 3, cp = q & 7;
        unsigned char *const sb = Abase + stage * TILE;
        if (MODE == MODE_FROM_LEVELS) {
            const int e = c0 + ((cp ^ SWZ(row)) << 3);                      // first channel of this 16-byte piece
            // pieces that start past the row end read zeros; the piece that straddles it (row_elems % 8 != 0) reads the row's
            // LAST 8 channels instead -- never past the row -- and the caller lays the weight's contraction axis out to
            // match: slots e .. e+7 = channels row_elems-8 .., zero weights on the repeated ones (rn_conv3x3_levels_to_canvas)
            const int es = e + 8 > args.lv.row_elems ? args.lv.row_elems - 8 : e;
            const uint16_t *g = (grow[i] && e < args.lv.row_elems) ? grow[i] + es : args.zeros + ((cp ^ SWZ(row)) << 3);
            __builtin_amdgcn_global_load_lds((const void *)g, (lds_void_ptr)lds_piece(sb, i), 16, 0, 0);
        } else {
            const int off = (t / 3 - 1) * Wp + (t % 3 - 1);
            if (MODE == MODE_DENSE) {
                const unsigned char *base = (const unsigned char *)a.X + ((m0 + off) * args.Cin + c0) * 2;
                const void *src = (tap_ok[i] >> t) & 1u ? (const void *)(base + voff_a[i]) : (const void *)args.zeros;
                __builtin_amdgcn_global_load_lds(src, (lds_void_ptr)lds_piece(sb, i), 16, 0, 0);
            } else {
                const uint16_t *X = a.X;
                int ld = args.Cin, cs = c0;
                if (MODE == MODE_CANVAS && args.x_split > 0) {             // (wave-uniform)
                    ld = args.x_ld;
                    if (w.chunk >= args.x_split) { X = args.X2s[prob]; cs = c0 - args.x_split * CONV_BK; }
                }
                if (!a_edge) {
                    const unsigned char *base = (const unsigned char *)X + ((m0 + off) * ld + cs) * 2;
                    __builtin_amdgcn_global_load_lds((const void *)(base + voff_a[i]), (lds_void_ptr)lds_piece(sb, i), 16, 0, 0);
                } else {
                    int64_t m = m0 + row + off;
                    m = m < 0 ? 0 : (m >= M ? M - 1 : m);
                    const uint16_t *g = X + m * ld + cs + ((cp ^ SWZ(row)) << 3);
                    __builtin_amdgcn_global_load_lds((const void *)g, (lds_void_ptr)lds_piece(sb, i), 16, 0, 0);
                }
            }
        }
    };
    auto piece_b = [&](const Walk w, const int stage, const int i) {
        if (NARROW && i > 0) return;                                 // 64 weight rows = the first piece of every thread
        const int c0 = w.chunk * CONV_BK, t = w.tap;
        unsigned char *const sb = Bbase + stage * TILE;
        if (!b_ragged) {
            const unsigned char *base = (const unsigned char *)a.W + (((int64_t)n0 * 9 + t) * args.Cin + c0) * 2;
            __builtin_amdgcn_global_load_lds((const void *)(base + voff_b[i]), (lds_void_ptr)lds_piece(sb, i), 16, 0, 0);
        } else {
            const int q = i * CONV_THREADS + tid, row = q >> 3, cp = q & 7;
            const uint16_t *g = a.W + ((int64_t)(n0 + row) * 9 + t) * args.Cin + c0 + ((cp ^ SWZ(row)) << 3);
            if (n0 + row >= args.Cout) g = args.zeros + ((cp ^ SWZ(row)) << 3);      // output channels past Cout
            __builtin_amdgcn_global_load_lds((const void *)g, (lds_void_ptr)lds_piece(sb, i), 16, 0, 0);
        }
    };
    typename rn::mma<DT>::frag fa[MT], fb[NT];
#define RN_DS_READ(dst, addr, OFF) asm volatile("ds_read_b128 %0, %1 offset:" #OFF : "=v"(dst) : "v"(addr))
#define RN_LOAD_FRAGS_AT(AA, KS)                                                   \
    { const uint32_t ba = bbase + b_off[KS], aa = (AA);                            \
      RN_DS_READ(fb[0], ba, 0); RN_DS_READ(fb[1], ba, 2048); RN_DS_READ(fb[2], ba, 4096); RN_DS_READ(fb[3], ba, 6144); \
      RN_DS_READ(fa[0], aa, 0); RN_DS_READ(fa[1], aa, 2048);                        \
      if (MT == 8) { RN_DS_READ(fa[MT - 6], aa, 4096); RN_DS_READ(fa[MT - 5], aa, 6144); RN_DS_READ(fa[MT - 4], aa, 8192);   \
                     RN_DS_READ(fa[MT - 3], aa, 10240); RN_DS_READ(fa[MT - 2], aa, 12288); RN_DS_READ(fa[MT - 1], aa, 14336); } }
#define RN_LOAD_FRAGS(KS) RN_LOAD_FRAGS_AT(abase + a_off[KS], KS)
#define RN_MFMA_ALL()                                                              \
    _Pragma("unroll") for (int mi = 0; mi < MT; ++mi)                              \
        _Pragma("unroll") for (int ni = 0; ni < NT; ++ni)                          \
            acc[mi][ni] = rn::mma<DT>::m16(fa[mi], fb[ni], acc[mi][ni]);
#define RN_MFMA_PHASE()                                                            \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0);          \
    __builtin_amdgcn_s_setprio(1);                                                 \
    RN_MFMA_ALL()                                                                  \
    __builtin_amdgcn_s_setprio(0);                                                 \
    __builtin_amdgcn_sched_barrier(0);                                             \
    __builtin_amdgcn_s_barrier();

    Walk wa = {0, 0}, wb = {0, 0};                                  // K-tile whose A / B pieces are issued next
    if (MODE == MODE_DENSE) { wa.tap = wb.tap = kt0 % 9; wa.chunk = wb.chunk = kt0 / 9; }
    int sa = 0;                                                     // its A stage (mod 3)
    if (MODE == MODE_FROM_LEVELS) gather_tap(0);
#pragma unroll
    for (int i = 0; i < 4; ++i) piece_a(wa, 0, i);
#pragma unroll
    for (int i = 0; i < 4; ++i) piece_b(wb, 0, i);
    advance(wa); advance(wb);                                       // -> K-tile 1
    if (MODE == MODE_FROM_LEVELS && cpt == 1) {
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");           // the fetch of tap 1 is older than tile 0's 8 pieces
        map_landed(g_entry);
        gather_tap(1);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) piece_a(wa, 1, i);
    advance(wa);                                                    // -> K-tile 2
    sa = 2;
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    if (MODE == MODE_FROM_LEVELS) map_landed(g_entry);
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();                    // group 1 runs one barrier interval behind group 0


    // The epilogue's per-column bias and the keep-mask bytes of the 16 rows this thread will store (row i * 16 + (tid >> 5)) are requested
    // INSIDE the walk, in K-tile 0's first load phase: younger than the staged pieces, older than every later piece, so the next counted wait
    // retires them under K-tile 0's MFMAs.  (In-kernel stamps, tools/probes: read inside the store loop the mask bytes were 16 dependent L2
    // round trips -- 5.4 us of a 79 us tile; requested before the FIRST staged pieces they delay the prologue by 2.6 us, right before the walk
    // by 0.9.  The tile's position on its sheet is a 32-bit modulo: the 64-bit one is a ~3 us software division per tile.)
    float bias_pre[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    uint32_t keep[16];                                              // (one register each: packed into bytes the compiler waits for every load in turn)
#pragma unroll
    for (int i = 0; i < 16; ++i) keep[i] = 1;
    auto epilogue_prefetch = [&]() {
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const int col = wn * 64 + ni * 16 + (lane & 15);
            bias_pre[ni] = (a.bias && (MODE != MODE_TO_LEVELS || n0 + col < args.Cout)) ? a.bias[n0 + col] : 0.0f;
        }
        if (MODE != MODE_TO_LEVELS && args.mask) {
            const int64_t pos0 = (int64_t)((uint32_t)m0 % (uint32_t)HWp);       // (host: M < 2^31)
            const bool one_wrap = HWp >= CONV_BM;                      // the tile crosses at most one sheet boundary
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = i * 16 + (tid >> 5);
                int64_t pos = pos0 + row;
                if (one_wrap) pos = pos >= HWp ? pos - HWp : pos; else pos = (int64_t)((uint32_t)pos % (uint32_t)HWp);
                keep[i] = args.mask[pos];                             // (pos < HWp whatever the row: no branch, the 16 loads fly together; rows past M are not stored)
            }
        }
    };
    int scur = 0;                                                   // A stage of K-tile kt (mod 3)
    for (int kt = 0; kt < KT; ++kt) {
        const uint32_t abase = lds_base + (uint32_t)(scur * TILE), bbase = lds_base + (uint32_t)(3 * TILE + (kt & 1) * TILE);
        scur = scur == 2 ? 0 : scur + 1;
        // load phase 2kt: fragments of k-steps 0,1; the weight pieces of tile kt+1
        RN_LOAD_FRAGS(0)
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 1 < KT) {
#pragma unroll
            for (int i = 0; i < 4; ++i) piece_b(wb, (kt + 1) & 1, i);
            advance(wb);
        }
        if (kt == 0) epilogue_prefetch();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        RN_MFMA_PHASE()
        // load phase 2kt+1: fragments of k-steps 2,3; the activation pieces of tile kt+2; retire tile kt+1
        RN_LOAD_FRAGS(1)
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 2 < KT) {
            if (MODE == MODE_FROM_LEVELS && wa.chunk == 0) gather_tap(wa.tap);     // wave-uniform: a new tap starts
#pragma unroll
            for (int i = 0; i < 4; ++i) piece_a(wa, sa, i);
            advance(wa);
            sa = sa == 2 ? 0 : sa + 1;
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            if (MODE == MODE_FROM_LEVELS) map_landed(g_entry);
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        RN_MFMA_PHASE()
    }

#undef RN_DS_READ
#undef RN_LOAD_FRAGS
#undef RN_LOAD_FRAGS_AT
#undef RN_MFMA_ALL
#undef RN_MFMA_PHASE
    if (grp == 0) __builtin_amdgcn_s_barrier();                    // group 0 catches up with group 1's extra barrier

    if (MODE == MODE_DENSE && args.ksplit > 1) {                   // f32 partial of this K range, straight from the accumulators
        float *__restrict__ out = args.kpartial + ((int64_t)blockIdx.z * M + m0) * args.Cout + n0;
#pragma unroll
        for (int mi = 0; mi < MT; ++mi)
#pragma unroll
            for (int ni = 0; ni < NT; ++ni) {
                const int col = wn * 64 + ni * 16 + (lane & 15);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = wm * (32 * MI) + mi * 16 + 4 * (lane >> 4) + r;
                    if (m0 + row < M) out[(int64_t)row * args.Cout + col] = acc[mi][ni][r];
                }
            }
        return;
    }
    __syncthreads();
    uint16_t *Ys = (uint16_t *)lds;                               // [256][256] bf16 output tile = 128 KiB
    uint16_t **Yrow = (uint16_t **)(lds + CONV_BM * CONV_BN * 2);  // TO_LEVELS: destination row of each tile row (2 KiB)
    float *s_cs = (float *)(lds + CONV_BM * CONV_BN * 2 + 2048);  // fused ReLU backward: column sums [16][256] f32 (16 KiB)
    uint8_t *s_rmask = lds + CONV_BM * CONV_BN * 2 + 2048 + 16384; //   and the tile's ReLU bits [256][32] bytes (8 KiB)
    uint4 *s_lut = (uint4 *)(lds + CONV_BM * CONV_BN * 2 + 2048 + 16384 + 8192);   //   and the 256 byte -> four-word-masks table (4 KiB)
    static_assert(CONV_BM * CONV_BN * 2 + 2048 + 16384 + 8192 + 4096 <= CONV_LDS_BYTES, "epilogue LDS");
    const uint8_t *relu_mask = MODE == MODE_TO_LEVELS ? nullptr : args.relu_masks[prob];
    uint4 rm_pre = make_uint4(0, 0, 0, 0);
    if (relu_mask) {                                              // fetched now, consumed after the staging below: latency hidden
        const int row = tid >> 1, half = tid & 1;
        if (m0 + row < M) rm_pre = *(const uint4 *)(relu_mask + (m0 + row) * (args.Cout >> 3) + (n0 >> 3) + half * 16);
    }
    // accumulators -> the staged tile.  VALU-bound (128 elements per thread, two waves per SIMD: 3.3 us of a 72 us tile on the stamp probe), so two
    // rows share one packed conversion, and the ReLU is one packed signed-16-bit max on the converted pair -- a negative or -0 pattern is a negative
    // integer, so max(., 0) is what round(relu(x)) gives (floor 0x8000 = no ReLU); a compare, a select and a conversion per ELEMENT before.
    // (A positive NaN passes through, as torch.relu does; the compare form used to zero it.)
    typedef short s16x2 __attribute__((ext_vector_type(2)));
    uint32_t relu_floor = (args.relu & 1) ? 0u : 0x80008000u;
    asm volatile("" : "+v"(relu_floor));                            // (opaque: one v_pk_max_i16 with a register operand, not a max with 0 + a select)
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) {
            const int col = wn * 64 + ni * 16 + (lane & 15);          // 16 x 16 result tile: column = lane & 15, rows 4 (lane >> 4) + r
            const float b = bias_pre[ni];
#pragma unroll
            for (int r = 0; r < 4; r += 2) {
                const int row = wm * (32 * MI) + mi * 16 + 4 * (lane >> 4) + r;
                const float x0 = acc[mi][ni][r] + b, x1 = acc[mi][ni][r + 1] + b;
                uint32_t raw;
                if (DT == RN_BF16) asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(raw) : "v"(x0), "v"(x1));      // (rn::dt::pk converts each half on its own + v_perm)
                else raw = rn::dt<DT>::pk(x0, x1);
                const uint32_t pk = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, raw), __builtin_bit_cast(s16x2, relu_floor)));
                Ys[row * CONV_BN + col] = (uint16_t)pk;
                Ys[(row + 1) * CONV_BN + col] = (uint16_t)(pk >> 16);
            }
        }
    if (MODE == MODE_TO_LEVELS && tid < CONV_BM) {
        const int64_t m = m0 + tid;
        uint16_t *dst = nullptr;
        if (m < M) {
            int n, pos;
            sheet_coords((int)m, (int)HWp, n, pos);
            dst = level_row(args.lv, lregs, n, args.lv.map[pos]);
        }
        Yrow[tid] = dst;
    }
    if (relu_mask) {
        *(uint4 *)(s_rmask + tid * 16) = rm_pre;
        if (tid < 256) {                                            // entry e: word j keeps its low / high half where bit 2j / 2j + 1 of e is set
            uint32_t w[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) w[j] = ((tid >> (2 * j)) & 1 ? 0x0000ffffu : 0u) | ((tid >> (2 * j + 1)) & 1 ? 0xffff0000u : 0u);
            s_lut[tid] = make_uint4(w[0], w[1], w[2], w[3]);
        }
    }
    __syncthreads();
    if (MODE == MODE_TO_LEVELS) {
        // dense rows of Cout elements start on 4-byte boundaries only (Cout even): 16-byte stores to dword-aligned
        // addresses are fine on gfx950 (tools/misalign_probe.hip); the tile's last piece may be cut by the row end
        const int ncols = min(CONV_BN, args.Cout - n0);           // > 0
#pragma unroll
        for (int i = 0; i < (NARROW ? 4 : 16); ++i) {               // NARROW: 8 pieces (64 columns) per row
            const int q = i * CONV_THREADS + tid, row = NARROW ? q >> 3 : q >> 5, piece = NARROW ? q & 7 : q & 31;
            uint16_t *dst = Yrow[row];
            if (dst && piece * 8 < ncols) {
                const uint16_t *src = Ys + row * CONV_BN + piece * 8;
                dst += n0 + piece * 8;
                if (piece * 8 + 8 <= ncols) {
                    const rn::u32x4 v = *(const rn::u32x4 *)src;
                    asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(dst), "v"(v) : "memory");
                } else {
                    for (int e = 0; e < ncols - piece * 8; e += 2) *(uint32_t *)(dst + e) = *(const uint32_t *)(src + e);
                }
            }
        }
    } else {
        uint8_t *rmask_out = MODE == MODE_TO_LEVELS ? nullptr : args.relu_mask_outs[prob];
        float cs[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int q = i * CONV_THREADS + tid, row = q >> 5, piece = q & 31;
            const int64_t m = m0 + row;
            if (m < M) {
                uint4 v = *(const uint4 *)(Ys + row * CONV_BN + piece * 8);
                if (!keep[i]) v = make_uint4(0, 0, 0, 0);
                if (relu_mask) {
                    // (the byte's four word masks come out of a 256-entry LDS table: expanded with shifts and selects they were ~24 of the
                    //  ~60 VALU instructions per piece of a loop that takes 4.9 us per tile, tools/probes)
                    const uint4 mk = s_lut[s_rmask[row * 32 + piece]];
                    uint32_t vw[4] = {v.x & mk.x, v.y & mk.y, v.z & mk.z, v.w & mk.w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {                    // two bf16 per word
                        cs[2 * j] += rn::mma<DT>::lo(vw[j]);
                        cs[2 * j + 1] += rn::mma<DT>::hi(vw[j]);
                    }
                    v = make_uint4(vw[0], vw[1], vw[2], vw[3]);
                }
                if (rmask_out) {                                     // forward: the ReLU bits of what is stored (y > 0 as floats: NaN no)
                    // (what is stored went through the ReLU: never negative, never NaN, never -0 -- so y > 0 is "the 16-bit pattern is not
                    //  zero": one packed unsigned min with 1 per word instead of two conversions and two compares)
                    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
                    const uint32_t vw[4] = {v.x, v.y, v.z, v.w};
                    uint32_t bits = 0;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const uint32_t nz = __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(us2, vw[j]), us2{1, 1}));
                        bits |= ((nz | (nz >> 15)) & 3u) << (2 * j);
                    }
                    rmask_out[m * (args.Cout >> 3) + (n0 >> 3) + piece] = (uint8_t)bits;
                }
                *(uint4 *)(a.Y + m * args.Cout + n0 + piece * 8) = v;
            }
        }
        if (relu_mask) {
            // a thread owns one 8-column piece over 16 rows; the 16 threads of a piece (tid & 31 equal) meet in LDS behind the
            // staged tile, then 256 threads write the tile's column sums
#pragma unroll
            for (int j = 0; j < 8; ++j) s_cs[(tid >> 5) * CONV_BN + (tid & 31) * 8 + j] = cs[j];
            __syncthreads();
            if (tid < CONV_BN) {
                float t = 0.0f;
#pragma unroll
                for (int r = 0; r < 16; ++r) t += s_cs[r * CONV_BN + tid];
                args.colsums[prob][(int64_t)bx * args.Cout + n0 + tid] = t;
            }
        }
    }
}


// ================================================================================================================
// Band-staged NARROW forward (canvas in, dense per-level rows out, at most 64 output columns): the box-output conv (36 channels)
// and the ragged last column tile of the class-output conv (810 = 3 x 256 + 42).  The NARROW variant of the kernel above stages a
// 256 x 64 activation tile per (channel chunk, tap) -- 36 tiles of 32 KiB per row tile for a quarter of a full tile's MFMAs: it
// is bound by operand staging (863 MB of L2 -> LDS traffic for a 98 MB input, 131 us).  Here the three horizontal taps of a
// kernel row read ONE band of 256 + 2 consecutive canvas positions (tap (r, s) of position m is position m + (r - 1) Wp + s - 1:
// band row = tile row + s), staged once per (chunk, kernel row): 12 bands of 33 KiB instead of 36 tiles.  Two band buffers + two
// buffers of the band's three weight tiles (24 KiB); one band ahead by LDS-DMA, counted vmcnt, two barriers per band; 8 waves of
// 32 rows x 64 columns, 48 MFMAs (16x16x32) per wave and band.  Same products in another order than the tile kernel (kernel
// row outer): results agree to f32 summation order.
// (a band is CONV_BM + 2 = 258 positions)
constexpr int BAND_BYTES = (CONV_BM + 8) * 128;                   // 264 rows of 64 channels: pieces are whole 8-row groups
constexpr int BANDW_BYTES = 3 * 64 * 128;                         // the band's three taps: 64 weight rows x 64 channels each
constexpr int BAND_LDS = 2 * BAND_BYTES + 2 * BANDW_BYTES;        // 116 736 bytes

template <int DT>
__global__ __launch_bounds__(CONV_THREADS) void conv3x3_band_narrow_kernel(const ConvArgs args)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t m0 = (int64_t)blockIdx.x * CONV_BM, M = args.M;
    const int Wp = args.Wp, cpt = args.Cin / CONV_BK, NB = 3 * cpt;
    const int n0 = args.n_base;
    const uint16_t *__restrict__ X = args.Xs[0], *__restrict__ Wt = args.Ws[0];
    const LevelRegs lregs = level_regs(args.lv);
    const uint32_t lds_base = (uint32_t)(uintptr_t)lds;

    rn::f32x4 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.0f;

    // fragment addresses: band row of (tile row, tap column s) = row + s
    uint32_t a_off[3][2][2], b_off[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int chunk = ks * 4 + (lane >> 4);
#pragma unroll
        for (int sft = 0; sft < 3; ++sft)
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                const int row = wave * 32 + mi * 16 + (lane & 15) + sft;
                a_off[sft][mi][ks] = row * 128 + ((chunk ^ SWZ(row)) << 4);
            }
        { const int row = lane & 15; b_off[ks] = row * 128 + ((chunk ^ SWZ(row)) << 4); }
    }

    auto stage = [&](const int b) {                                // band b and its three weight tiles -> buffers b & 1
        const int c0 = (b / 3) * CONV_BK, r = b % 3;
        const int64_t p0 = m0 + (int64_t)(r - 1) * Wp - 1;         // canvas position of band row 0
        unsigned char *const ab = lds + (b & 1) * BAND_BYTES, *const wb = lds + 2 * BAND_BYTES + (b & 1) * BANDW_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = i * CONV_THREADS + tid, j = q >> 3, cp = q & 7;
            int64_t p = p0 + j;
            p = p < 0 ? 0 : (p >= M ? M - 1 : p);                    // (only border / gap outputs see a clamped row; they are never stored)
            const uint16_t *g = X + p * args.Cin + c0 + ((cp ^ SWZ(j)) << 3);
            __builtin_amdgcn_global_load_lds((const void *)g, (lds_void_ptr)(ab + i * (CONV_THREADS * 16) + wave * (RN_WAVE * 16)), 16, 0, 0);
        }
        if (wave == 0) {                                            // band rows 256 .. 263 (257 is the last one read)
            const int j = 256 + (lane >> 3), cp = lane & 7;
            int64_t p = p0 + j;
            p = p < 0 ? 0 : (p >= M ? M - 1 : p);
            const uint16_t *g = X + p * args.Cin + c0 + ((cp ^ SWZ(j)) << 3);
            __builtin_amdgcn_global_load_lds((const void *)g, (lds_void_ptr)(ab + 4 * (CONV_THREADS * 16)), 16, 0, 0);
        }
#pragma unroll
        for (int sft = 0; sft < 3; ++sft) {
            const int row = tid >> 3, cp = tid & 7, t = 3 * r + sft;
            const uint16_t *g = (n0 + row < args.Cout) ? Wt + ((int64_t)(n0 + row) * 9 + t) * args.Cin + c0 + ((cp ^ SWZ(row)) << 3)
                                                       : args.zeros + ((cp ^ SWZ(row)) << 3);
            __builtin_amdgcn_global_load_lds((const void *)g, (lds_void_ptr)(wb + sft * 8192 + wave * (RN_WAVE * 16)), 16, 0, 0);
        }
    };

    typename rn::mma<DT>::frag fa[2], fb[4];
#define RN_DS_READ(dst, addr, OFF) asm volatile("ds_read_b128 %0, %1 offset:" #OFF : "=v"(dst) : "v"(addr))
    stage(0);
    for (int b = 0; b < NB; ++b) {
        if (b + 1 < NB) {
            stage(b + 1);
            if (wave == 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");     // band b's pieces are older than the 8 (7) just issued
            else asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        const uint32_t aband = lds_base + (uint32_t)((b & 1) * BAND_BYTES), wband = lds_base + (uint32_t)(2 * BAND_BYTES + (b & 1) * BANDW_BYTES);
#pragma unroll
        for (int sft = 0; sft < 3; ++sft)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const uint32_t ba = wband + (uint32_t)(sft * 8192) + b_off[ks];
                RN_DS_READ(fb[0], ba, 0); RN_DS_READ(fb[1], ba, 2048); RN_DS_READ(fb[2], ba, 4096); RN_DS_READ(fb[3], ba, 6144);
                RN_DS_READ(fa[0], aband + a_off[sft][0][ks], 0); RN_DS_READ(fa[1], aband + a_off[sft][1][ks], 0);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = rn::mma<DT>::m16(fa[mi], fb[ni], acc[mi][ni]);
                __builtin_amdgcn_sched_barrier(0);
            }
        __builtin_amdgcn_s_barrier();                               // buffers b & 1 are free for band b + 2
    }
#undef RN_DS_READ

    // ---- epilogue (the tile kernel's TO_LEVELS / NARROW form): bias, 16-bit tile in LDS, dense per-level rows
    __syncthreads();
    uint16_t *Ys = (uint16_t *)lds;                               // [256][64]
    uint16_t **Yrow = (uint16_t **)(lds + CONV_BM * 64 * 2);
    const float *bias = args.biases[0];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const int col = ni * 16 + (lane & 15);
            const float bv = (bias && n0 + col < args.Cout) ? bias[n0 + col] : 0.0f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = wave * 32 + mi * 16 + 4 * (lane >> 4) + r;
                Ys[row * 64 + col] = rn::mma<DT>::dn(acc[mi][ni][r] + bv);
            }
        }
    if (tid < CONV_BM) {
        const int64_t m = m0 + tid;
        uint16_t *dst = nullptr;
        if (m < M) {
            int n, pos;
            sheet_coords((int)m, (int)args.HWp, n, pos);
            dst = level_row(args.lv, lregs, n, args.lv.map[pos]);
        }
        Yrow[tid] = dst;
    }
    __syncthreads();
    const int ncols = min(64, args.Cout - n0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = i * CONV_THREADS + tid, row = q >> 3, piece = q & 7;
        uint16_t *dst = Yrow[row];
        if (dst && piece * 8 < ncols) {
            const uint16_t *src = Ys + row * 64 + piece * 8;
            dst += n0 + piece * 8;
            if (piece * 8 + 8 <= ncols) {
                const rn::u32x4 v = *(const rn::u32x4 *)src;
                asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(dst), "v"(v) : "memory");
            } else {
                for (int e = 0; e < ncols - piece * 8; e += 2) *(uint32_t *)(dst + e) = *(const uint32_t *)(src + e);
            }
        }
    }
}

// ================================================================================================================
// Band-staged 3x3 / stride-1 / pad-1 convolution of a plain dense tensor [N][H][W][Cin] -> [N][H][W][Cout], Cout a multiple of 128:
// conv2 of the layer2 bottlenecks (128 -> 128 on 134 400 positions; retinanet/backbone.py:112,128) forward and, with the flipped
// weights, data gradient -- CK's grouped-conv kernel takes 68 us there, this one 65.  As in the band kernel above a 256-position tile
// reads, per (channel chunk, kernel row), ONE band of 258 consecutive positions for its three horizontal taps; a dense tensor has no
// zero border, so a tap that leaves its image is removed AT THE FRAGMENT: lane l of a 16 x 16 x 32 A fragment carries row l & 15, and
// the row's 9-bit tap validity (computed once per tile) zeroes the lane's 16 bytes.  8 waves of 32 rows x 128 columns (64 accumulator
// registers).  Weights: one 128 x 64 tile per tap in three rotating buffers -- a tap buffer is refilled for the next band as soon as
// the barrier of the following tap step proves it consumed -- so LDS is 2 bands + 3 tap tiles = 116 KiB; one barrier per tap step.
// What bounds it: LDS-DMA fills a CU's LDS at ~38 GB/s, and a tile needs 6 x (33 + 48) KiB -- the whole 295 KB weight tensor per
// tile; 65 us with two barriers per tap and unpipelined fragment reads, 65 us with one barrier and the second k-step's reads under
// the first one's MFMAs.  The weights as direct global -> register fragment loads (waves 2 x 4, three rotating register sets,
// requested two tap steps ahead, bands alone through LDS) were built as well: 88 - 92 us.
constexpr int DBAND_LDS = 2 * BAND_BYTES + 3 * 128 * 128;          // 116 736 bytes

struct DenseBandArgs {
    const uint16_t *X, *W;          // [M][Cin]; [Cout][9][Cin]
    uint16_t *Y;                    // [M][Cout]
    const uint16_t *zeros;
    int64_t M;
    int H, Wd, Cin, Cout;
    // partial != nullptr: the per-channel sums (sum y, sum y^2) of the tile AS STORED ride in the epilogue, partial f32 [row tiles][2][Cout] --
    // the statistics pass of the BatchNorm that follows the convolution (bn2 of a bottleneck).  (The two BatchNorm-BACKWARD sums of the
    // layer in front, for the data-gradient launch, were built the same way -- Z tile and constants prefetched before the walk -- and cost
    // as much as the pass they replace: 15 us of VALU work per launch that nothing overlaps with one workgroup per CU.  Removed.)
    float *partial;
};

template <int DT>
__global__ __launch_bounds__(CONV_THREADS) void conv3x3_dense_band_kernel(const DenseBandArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t m0 = (int64_t)blockIdx.x * CONV_BM, M = a.M;
    const int n0 = blockIdx.y * 128;
    const int cpt = a.Cin / CONV_BK, NB = 3 * cpt;
    const uint32_t lds_base = (uint32_t)(uintptr_t)lds;
    unsigned char *const wbase = lds + 2 * BAND_BYTES;

    rn::f32x4 acc[2][8];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.0f;

    uint32_t a_off[3][2][2], b_off[2], tap_ok[2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {                              // which of the 9 taps of this lane's row stay inside its image
        const int64_t m = m0 + wave * 32 + mi * 16 + (lane & 15);
        tap_ok[mi] = 0u;
        if (m < M) {
            const int yl = (int)(m / a.Wd), x = (int)(m - (int64_t)yl * a.Wd), y = yl % a.H;
#pragma unroll
            for (int t = 0; t < 9; ++t)
                tap_ok[mi] |= ((unsigned)(y + t / 3 - 1) < (unsigned)a.H && (unsigned)(x + t % 3 - 1) < (unsigned)a.Wd) ? 1u << t : 0u;
        }
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int chunk = ks * 4 + (lane >> 4);
#pragma unroll
        for (int sft = 0; sft < 3; ++sft)
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                const int row = wave * 32 + mi * 16 + (lane & 15) + sft;
                a_off[sft][mi][ks] = row * 128 + ((chunk ^ SWZ(row)) << 4);
            }
        { const int row = lane & 15; b_off[ks] = row * 128 + ((chunk ^ SWZ(row)) << 4); }
    }

    auto stage_band = [&](const int b) {                            // 4 pieces per thread, 5 in wave 0
        const int c0 = (b / 3) * CONV_BK, r = b % 3;
        const int64_t p0 = m0 + (int64_t)(r - 1) * a.Wd - 1;
        unsigned char *const ab = lds + (b & 1) * BAND_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = i * CONV_THREADS + tid, j = q >> 3, cp = q & 7;
            int64_t p = p0 + j;
            p = p < 0 ? 0 : (p >= M ? M - 1 : p);
            __builtin_amdgcn_global_load_lds((const void *)(a.X + p * a.Cin + c0 + ((cp ^ SWZ(j)) << 3)),
                                             (lds_void_ptr)(ab + i * (CONV_THREADS * 16) + wave * (RN_WAVE * 16)), 16, 0, 0);
        }
        if (wave == 0) {
            const int j = 256 + (lane >> 3), cp = lane & 7;
            int64_t p = p0 + j;
            p = p < 0 ? 0 : (p >= M ? M - 1 : p);
            __builtin_amdgcn_global_load_lds((const void *)(a.X + p * a.Cin + c0 + ((cp ^ SWZ(j)) << 3)), (lds_void_ptr)(ab + 4 * (CONV_THREADS * 16)), 16, 0, 0);
        }
    };
    auto stage_tap = [&](const int b, const int sft) {              // 2 pieces per thread: weight rows n0 .. n0 + 127, tap 3 r + sft, chunk of band b
        const int c0 = (b / 3) * CONV_BK, t = 3 * (b % 3) + sft;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int q = i * CONV_THREADS + tid, row = q >> 3, cp = q & 7;
            __builtin_amdgcn_global_load_lds((const void *)(a.W + ((int64_t)(n0 + row) * 9 + t) * a.Cin + c0 + ((cp ^ SWZ(row)) << 3)),
                                             (lds_void_ptr)(wbase + sft * 16384 + i * (CONV_THREADS * 16) + wave * (RN_WAVE * 16)), 16, 0, 0);
        }
    };

    typename rn::mma<DT>::frag fa[2][2], fb[2][8];                // [k-step][fragment]: the second k-step's reads are issued under the first one's MFMAs
    const typename rn::mma<DT>::frag fzero = {};
#define RN_DS_READ(dst, addr, OFF) asm volatile("ds_read_b128 %0, %1 offset:" #OFF : "=v"(dst) : "v"(addr))
#define RN_BAND_FRAGS(KS)                                                                                           \
    { const uint32_t ba = lds_base + (uint32_t)(2 * BAND_BYTES + sft * 16384) + b_off[KS];                          \
      RN_DS_READ(fb[KS][0], ba, 0); RN_DS_READ(fb[KS][1], ba, 2048); RN_DS_READ(fb[KS][2], ba, 4096); RN_DS_READ(fb[KS][3], ba, 6144);       \
      RN_DS_READ(fb[KS][4], ba, 8192); RN_DS_READ(fb[KS][5], ba, 10240); RN_DS_READ(fb[KS][6], ba, 12288); RN_DS_READ(fb[KS][7], ba, 14336); \
      RN_DS_READ(fa[KS][0], aband + a_off[sft][0][KS], 0); RN_DS_READ(fa[KS][1], aband + a_off[sft][1][KS], 0); }
#define RN_BAND_MFMA(KS)                                                                                            \
    { if (!ok0) fa[KS][0] = fzero;                                                                                  \
      if (!ok1) fa[KS][1] = fzero;                                                                                  \
      _Pragma("unroll") for (int mi = 0; mi < 2; ++mi)                                                              \
          _Pragma("unroll") for (int ni = 0; ni < 8; ++ni) acc[mi][ni] = rn::mma<DT>::m16(fa[KS][mi], fb[KS][ni], acc[mi][ni]); }
    // One barrier per tap step.  Issue order (pieces per thread: band 4, 5 in wave 0; tap 2): step (b, 0) requests band b + 1 and tap
    // (b, 2), step (b, 1) tap (b + 1, 0), step (b, 2) tap (b + 1, 1) -- always into the buffer the PREVIOUS step has just released,
    // which the step's barrier proves.  A step's own operands are therefore older than the 2 / 6 (7) / 2 pieces that may still be out.
    stage_band(0); stage_tap(0, 0); stage_tap(0, 1);
    for (int b = 0; b < NB; ++b) {
        const bool more = b + 1 < NB;
        const uint32_t aband = lds_base + (uint32_t)((b & 1) * BAND_BYTES);
#pragma unroll
        for (int sft = 0; sft < 3; ++sft) {
            if (!more) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (sft == 1) { if (wave == 0) asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); }
            else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            if (sft == 0) { if (more) stage_band(b + 1); stage_tap(b, 2); }
            else if (more) stage_tap(b + 1, sft - 1);
            const int t = 3 * (b % 3) + sft;
            const bool ok0 = (tap_ok[0] >> t) & 1u, ok1 = (tap_ok[1] >> t) & 1u;
            RN_BAND_FRAGS(0)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            RN_BAND_FRAGS(1)
            __builtin_amdgcn_sched_barrier(0);
            RN_BAND_MFMA(0)
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            RN_BAND_MFMA(1)
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#undef RN_DS_READ
#undef RN_BAND_FRAGS
#undef RN_BAND_MFMA

    __syncthreads();
    uint16_t *Ys = (uint16_t *)lds;                               // [256][128]
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 8; ++ni) {
            const int col = ni * 16 + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = wave * 32 + mi * 16 + 4 * (lane >> 4) + r;
                Ys[row * 128 + col] = rn::mma<DT>::dn(acc[mi][ni][r]);
            }
        }
    __syncthreads();
    const int piece = tid & 15, rt = tid >> 4;                    // thread = 8 columns of rows rt, rt + 32, ..
    float cs[8], cq[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { cs[j] = 0.0f; cq[j] = 0.0f; }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = i * 32 + rt;
        const int64_t m = m0 + row;
        if (m < M) {
            const rn::u32x4 v = *(const rn::u32x4 *)(Ys + row * 128 + piece * 8);
            *(rn::u32x4 *)(a.Y + m * a.Cout + n0 + piece * 8) = v;
            if (a.partial) {
                float y[8];
                rn::dt<DT>::unpack(v, y);
#pragma unroll
                for (int j = 0; j < 8; ++j) { cs[j] += y[j]; cq[j] = fmaf(y[j], y[j], cq[j]); }
            }
        }
    }
    if (a.partial) {                                              // 32 row threads per column -> one partial row per tile (fixed order)
        float *red = (float *)(lds + 65536);                      // [32][2][128] f32 behind the staged tile
#pragma unroll
        for (int j = 0; j < 8; ++j) { red[(rt * 2 + 0) * 128 + piece * 8 + j] = cs[j]; red[(rt * 2 + 1) * 128 + piece * 8 + j] = cq[j]; }
        __syncthreads();
        if (tid < 256) {
            const int which = tid >> 7, c = tid & 127;
            float t = 0.0f;
#pragma unroll 8
            for (int r = 0; r < 32; ++r) t += red[(r * 2 + which) * 128 + c];
            a.partial[((int64_t)blockIdx.x * 2 + which) * a.Cout + n0 + c] = t;
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
__device__ __forceinline__ int wg_swz_row(const int row) { return (row & 3) << 2; }

struct WgradArgs {
    const uint16_t *Gs[CONV_MAX_PROBLEMS];      // [M][256] bf16 (GATHER: unused, the gradient rows come from `lv`)
    const uint16_t *Xs[CONV_MAX_PROBLEMS];      // [M][256] bf16
    const uint16_t *zeros;
    float *partial;                             // [P][S][9][256][256] f32
    int64_t M, HWp;
    int Wp, S, tiles_per_split;                 // K-tiles (64 positions) per split
    int ch_base;                                // GATHER: first output channel of problem 0 (the NARROW launch of the last row tile)
    int f16;                                    // host side only: fp16 elements (kernel instantiation)
    LevelSet lv;                                // GATHER: dense per-level gradient tensors [N][h][w][row_elems]; problem p = output channels 256 p ..
    DenseGeom dn;                               // DENSE: per-problem geometry; tile_beg = first split of the problem (blockIdx.x walks them all)
    int dn_tps[CONV_MAX_PROBLEMS];              //        K-tiles per split of problem p
    // xcd_units > 0: a 1-D grid of 256 workgroups placed by XCD (workgroups go round-robin over the 8 XCDs: XCD = blockIdx.x % 8).  All
    // (tap, problem) workgroups of position split s run on XCD s -- they stream the SAME position range, the 4 problems of a tap the
    // same X tiles and the 9 taps of a problem the same G tiles, in step -- so each XCD's L2 sees its split's operands once instead of
    // every workgroup fetching its own copy (the gathered class-output gradient + activations are 388 MB, the 36 copies 6.1 GB);
    // the (xcd_units - 32) units per split that do not fit XCD s's 32 CUs run on XCD 7.  Host: S <= 7, (xcd_units - 32) * S <= 32.
    int xcd_units;
};

// GATHER = false: G and X are canvases (the head towers).  GATHER = true: the gradient operand is gathered, position by
// position, from dense per-level tensors with row_elems channels (the class-output conv: 810); blockIdx.z selects the tile
// of 256 output channels, chunks past the row end read zeros (the straddling chunk pollutes only unused dW rows).
// NARROW (GATHER only): a row tile of at most 64 output channels -- the ragged last tile of the 810-channel class-output conv
// (42 rows) and the whole 36-channel box-output conv.  All 8 waves split the 256 input channels (32 each) and compute
// 64 x 32: a quarter of the MFMAs and half of the fragment reads of a full tile whose other 192 rows would be zeros.
// DENSE (not GATHER): G and X are plain dense [N][h][w][256] tensors of up to 4 problems with their own geometry (MODE_DENSE of
// the forward kernel); every problem has its own number of splits -- proportional to its positions, so that all workgroups
// walk about the same number of K-tiles -- and an X row whose tap leaves the image reads the zero page: row / column of the
// 4 rows a thread stages are recomputed per K-tile (two reciprocal multiplications each, sheet_coords).
template <int DT, bool GATHER, bool NARROW = false, bool DENSE = false>
__global__ __launch_bounds__(CONV_THREADS) void conv3x3_wgrad_kernel(const WgradArgs a)
{
    constexpr int MI = NARROW ? 2 : 4, NI = NARROW ? 1 : 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];                 // [G0 G1 G2 | X0 X1] x 32 KiB
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);    // wave-uniform (LDS-DMA bases, wave roles)
    const int wm = NARROW ? 0 : wave >> 2, wn = NARROW ? wave : wave & 3;       // NARROW: wn counts 32-channel columns
    const int pp = wave >> 2;                                     // ping-pong group: waves 0-3 / 4-7
    int split = blockIdx.x, prob = blockIdx.z, KT = a.tiles_per_split, Wp = a.Wp, dense_h = 0;
    int tap = blockIdx.y;
    if (!DENSE && a.xcd_units > 0) {
        const int L = blockIdx.x, xcd = L & 7, w = L >> 3;
        int unit;
        if (xcd < 7) { split = xcd; unit = w; if (split >= a.S || unit >= a.xcd_units) return; }
        else { const int spill = a.xcd_units - 32; if (spill <= 0) return; split = w / spill; unit = 32 + w % spill; if (split >= a.S) return; }
        tap = unit % 9; prob = unit / 9;
    }
    int64_t M = a.M;
    if (DENSE) {
        prob = 0;
#pragma unroll
        for (int p = 1; p < CONV_MAX_PROBLEMS; ++p) prob = (int)blockIdx.x >= a.dn.tile_beg[p] ? p : prob;
        split = (int)blockIdx.x - a.dn.tile_beg[prob];
        KT = a.dn_tps[prob]; Wp = a.dn.w[prob]; dense_h = a.dn.h[prob]; M = a.dn.M[prob];
    }
    const uint16_t *__restrict__ G = a.Gs[prob], *__restrict__ X = a.Xs[GATHER ? 0 : prob];
    const int64_t m_begin = (int64_t)split * KT * WG_POS;
    const int off = (tap / 3 - 1) * Wp + (tap % 3 - 1);
    constexpr int TILE = WG_POS * 512;                            // 32 KiB
    unsigned char *const Abase = lds, *const Bbase = lds + 3 * TILE;

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // transposed-read addresses: lane = 16*grp + 4*q + p;  grp & 1 selects the 16-channel half of a 32-channel
    // fragment, grp >> 1 the k half (positions +8); the lane supplies row q, channels 4p .. 4p+3 of its block.
    const int grp = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    uint32_t a_off[MI], b_off[NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int ch = wm * 16 + mi * 4 + 2 * (grp & 1) + (p >> 1);                  // 16-byte chunk of the channel row
        a_off[mi] = (uint32_t)((8 * (grp >> 1) + q) * 512 + ((ch ^ (q << 2)) << 4) + (p & 1) * 8);
    }
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const int ch = wn * (4 * NI) + ni * 4 + 2 * (grp & 1) + (p >> 1);
        b_off[ni] = (uint32_t)((8 * (grp >> 1) + q) * 512 + ((ch ^ (q << 2)) << 4) + (p & 1) * 8);
    }
    const uint32_t lds_base = (uint32_t)(uintptr_t)lds;

    // staging: tile row = position, 32 chunks of 16 B; LDS position (row, cp) holds global chunk cp ^ ((row & 3) << 2).
    // As in the forward kernel: wave-uniform 64-bit base per K-tile + a per-thread 32-bit offset for every K-tile that lies
    // inside the buffer; the clamping per-lane path only for the first / last ones.
    uint32_t voff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int qi = i * CONV_THREADS + tid, row = qi >> 5, cp = qi & 31;
        voff[i] = (uint32_t)(row * 512 + ((cp ^ ((row & 3) << 2)) << 4));
    }
    auto lds_piece = [&](unsigned char *stage_base, const int i) {
        return stage_base + i * (CONV_THREADS * 16) + wave * (RN_WAVE * 16);
    };
    auto piece_b = [&](const int kt, const int i) {
        const int64_t t0 = m_begin + (int64_t)kt * WG_POS;                                   // first position of the K-tile
        unsigned char *const sb = Bbase + (kt & 1) * TILE;
        if (DENSE) {
            const int qi = i * CONV_THREADS + tid, row = qi >> 5, cp = qi & 31;
            const int m = (int)t0 + row;                                                      // (host: M < 2^22)
            int yl, x, n, y;
            sheet_coords(m, Wp, yl, x);
            sheet_coords(yl, dense_h, n, y);
            const bool ok = m < (int)M && (unsigned)(y + tap / 3 - 1) < (unsigned)dense_h && (unsigned)(x + tap % 3 - 1) < (unsigned)Wp;
            const uint16_t *g = ok ? X + ((int64_t)m + off) * 256 + ((cp ^ wg_swz_row(row)) << 3) : a.zeros + ((cp & 15) << 3);
            __builtin_amdgcn_global_load_lds((const void *)g, (lds_void_ptr)lds_piece(sb, i), 16, 0, 0);
        } else if (t0 + off >= 0 && t0 + off + WG_POS <= M) {                                      // wave-uniform
            const unsigned char *base = (const unsigned char *)X + (t0 + off) * 512;
            __builtin_amdgcn_global_load_lds((const void *)(base + voff[i]), (lds_void_ptr)lds_piece(sb, i), 16, 0, 0);
        } else {
            const int qi = i * CONV_THREADS + tid, row = qi >> 5, cp = qi & 31;
            const int64_t m = t0 + row;
            int64_t ms = m + off;
            ms = ms < 0 ? 0 : (ms >= M ? M - 1 : ms);
            const uint16_t *g = X + ms * 256 + ((cp ^ ((row & 3) << 2)) << 3);
            if (m >= M) g = a.zeros + ((cp & 15) << 3);             // positions past the end contribute nothing
            __builtin_amdgcn_global_load_lds((const void *)g, (lds_void_ptr)lds_piece(sb, i), 16, 0, 0);
        }
    };
    // GATHER: a wave stages rows i * 16 + wave * 2 + (lane >> 5), i = 0..3, of a K-tile: 8 distinct rows.  Lane j < 8 looks up
    // row (j >> 1) * 16 + wave * 2 + (j & 1) and the wave shares the pointers by shuffles (1 lookup per lane, not 4).
    const uint16_t *grow[4] = {nullptr, nullptr, nullptr, nullptr};
    // (the map entries of K-tile kt + 1 are fetched while the pointers of K-tile kt are formed: map_fetch / map_landed)
    LevelRegs lregs = {};
    if (GATHER) lregs = level_regs(a.lv);
    int w_n = 0;
    int32_t w_entry = -1;
    bool w_valid = false;
    auto rows_pos = [&](const int kt) {                            // -> w_n, w_valid; returns the (clamped) sheet position
        const int j = lane & 7;
        const int64_t m = m_begin + (int64_t)kt * WG_POS + (j >> 1) * 16 + wave * 2 + (j & 1);
        w_valid = m < M;
        int pos;
        sheet_coords((int)(w_valid ? m : M - 1), (int)a.HWp, w_n, pos);
        return pos;
    };
    if (GATHER) w_entry = a.lv.map[rows_pos(0)];                   // (prologue: nothing in flight yet)
    auto gather_rows = [&](const int kt) {
        const uint16_t *mine = level_row(a.lv, lregs, w_n, w_valid ? w_entry : -1);
        map_fetch(w_entry, a.lv.map + rows_pos(kt + 1 < KT ? kt + 1 : kt));
        const bool second = lane >= 32;
        grow[0] = pair_ptr<0, 1>(mine, second); grow[1] = pair_ptr<2, 3>(mine, second);
        grow[2] = pair_ptr<4, 5>(mine, second); grow[3] = pair_ptr<6, 7>(mine, second);
    };
    auto piece_a = [&](const int kt, const int stage, const int i) {
        const int qi = i * CONV_THREADS + tid, row = qi >> 5, cp = qi & 31;
        const int64_t t0 = m_begin + (int64_t)kt * WG_POS;
        unsigned char *const sb = Abase + stage * TILE;
        if (GATHER) {
            const int e = a.ch_base + prob * 256 + ((cp ^ ((row & 3) << 2)) << 3);          // first output channel of this 16-byte piece
            // the piece that straddles the row end reads the row's last 8 channels instead (never past the row); the
            // reduction kernel writes its dW rows to the channels they really are
            const int es = e + 8 > a.lv.row_elems ? a.lv.row_elems - 8 : e;
            const uint16_t *g = (grow[i] && e < a.lv.row_elems) ? grow[i] + es : a.zeros + ((cp & 15) << 3);
            __builtin_amdgcn_global_load_lds((const void *)g, (lds_void_ptr)lds_piece(sb, i), 16, 0, 0);
        } else if (t0 + WG_POS <= M) {                                          // wave-uniform
            const unsigned char *base = (const unsigned char *)G + t0 * 512;
            __builtin_amdgcn_global_load_lds((const void *)(base + voff[i]), (lds_void_ptr)lds_piece(sb, i), 16, 0, 0);
        } else {
            const int64_t m = t0 + row;
            const int64_t ms = m >= M ? M - 1 : m;
            const uint16_t *g = G + ms * 256 + ((cp ^ ((row & 3) << 2)) << 3);
            if (m >= M) g = a.zeros + ((cp & 15) << 3);
            __builtin_amdgcn_global_load_lds((const void *)g, (lds_void_ptr)lds_piece(sb, i), 16, 0, 0);
        }
    };
    if (GATHER) gather_rows(0);
#pragma unroll
    for (int i = 0; i < 4; ++i) piece_a(0, 0, i);
#pragma unroll
    for (int i = 0; i < 4; ++i) piece_b(0, i);
    if (KT > 1) {
        if (GATHER) {
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");       // the fetch for K-tile 1 is older than tile 0's 8 pieces
            map_landed(w_entry);
            gather_rows(1);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) piece_a(1, 1, i);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        if (GATHER) map_landed(w_entry);
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (pp == 1) __builtin_amdgcn_s_barrier();                    // ping-pong, as in the forward kernel

    unsigned long long fa[2][MI][2], fb[2][NI][2];                // [k-step set][fragment][k half]
#define RN_TR_READ(dst, addr, OFF) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:" #OFF : "=v"(dst) : "v"(addr))
#define RN_LOAD_FRAGS(KOFF0, KOFF1, SET)                                             \
    RN_TR_READ(fb[SET][0][0], bbase + b_off[0], KOFF0); RN_TR_READ(fb[SET][0][1], bbase + b_off[0], KOFF1);   \
    if (NI == 2) { RN_TR_READ(fb[SET][NI - 1][0], bbase + b_off[NI - 1], KOFF0); RN_TR_READ(fb[SET][NI - 1][1], bbase + b_off[NI - 1], KOFF1); } \
    RN_TR_READ(fa[SET][0][0], abase + a_off[0], KOFF0); RN_TR_READ(fa[SET][0][1], abase + a_off[0], KOFF1);   \
    RN_TR_READ(fa[SET][1][0], abase + a_off[1], KOFF0); RN_TR_READ(fa[SET][1][1], abase + a_off[1], KOFF1);   \
    if (MI == 4) {                                                                                            \
        RN_TR_READ(fa[SET][MI - 2][0], abase + a_off[MI - 2], KOFF0); RN_TR_READ(fa[SET][MI - 2][1], abase + a_off[MI - 2], KOFF1);   \
        RN_TR_READ(fa[SET][MI - 1][0], abase + a_off[MI - 1], KOFF0); RN_TR_READ(fa[SET][MI - 1][1], abase + a_off[MI - 1], KOFF1); }
    struct U2 { unsigned long long lo, hi; };
#define RN_FRAG(v) __builtin_bit_cast(typename rn::mma<DT>::frag, U2{v[0], v[1]})
#define RN_MFMA8(SET)                                                              \
    _Pragma("unroll") for (int mi = 0; mi < MI; ++mi)                              \
        _Pragma("unroll") for (int ni = 0; ni < NI; ++ni)                          \
            acc[mi][ni] = rn::mma<DT>::m32(RN_FRAG(fa[SET][mi]), RN_FRAG(fb[SET][ni]), acc[mi][ni]);
#define RN_MFMA_PHASE()                                                            \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0);          \
    __builtin_amdgcn_s_setprio(1);                                                 \
    RN_MFMA8(0) RN_MFMA8(1)                                                        \
    __builtin_amdgcn_s_setprio(0);                                                 \
    __builtin_amdgcn_sched_barrier(0);                                             \
    __builtin_amdgcn_s_barrier();

    int scur = 0, sa = 2;                                           // A stages (mod 3) of K-tiles kt and kt + 2
    for (int kt = 0; kt < KT; ++kt) {
        const uint32_t abase = lds_base + (uint32_t)(scur * TILE), bbase = lds_base + (uint32_t)(3 * TILE + (kt & 1) * TILE);
        scur = scur == 2 ? 0 : scur + 1;
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
            if (GATHER) gather_rows(kt + 2);
#pragma unroll
            for (int i = 0; i < 4; ++i) piece_a(kt + 2, sa, i);
            sa = sa == 2 ? 0 : sa + 1;
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            if (GATHER) map_landed(w_entry);
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
    if (pp == 0) __builtin_amdgcn_s_barrier();

    // partial tile [n][c] f32 of this (problem, split, tap)
    float *__restrict__ out = a.partial + (DENSE ? (int64_t)blockIdx.x * 9 + tap : ((int64_t)prob * a.S + split) * 9 + tap) * 65536;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int col = wn * (32 * NI) + ni * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * 128 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                out[row * 256 + col] = acc[mi][ni][r];
            }
        }
}

// dW[p][n][t][c] (bf16) = sum over the splits of partial[p][s][t][n][c]
// (rows: output channels of problem p that exist -- 256 except for the last channel tile of the class-output conv)
template <int DT>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float *__restrict__ partial, const int S, uint16_t *dw0, uint16_t *dw1,
                                                           uint16_t *dw2, uint16_t *dw3, const int rows0, const int rows1,
                                                           const int rows2, const int rows3, const int shift_from, const int shift)
{
    const int prob = blockIdx.y;
    uint16_t *dw = prob == 0 ? dw0 : (prob == 1 ? dw1 : (prob == 2 ? dw2 : dw3));
    const int rows = prob == 0 ? rows0 : (prob == 1 ? rows1 : (prob == 2 ? rows2 : rows3));
    const int64_t i4 = (int64_t)blockIdx.x * 256 + threadIdx.x;          // over 9 * 256 * 256 / 4 float4 groups of [t][n][c]
    if (i4 >= 9 * 65536 / 4) return;
    const int64_t e = i4 * 4;
    const int t = (int)(e / 65536), n = (int)((e % 65536) / 256), c = (int)(e % 256);
    if (n >= rows && !(shift > 0 && prob == (int)gridDim.y - 1 && n >= shift_from && n < shift_from + 8)) return;   // rows nobody wrote or wants
    rn::f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int sp = 0; sp < S; ++sp) {
        const rn::f32x4 v = ((const rn::f32x4 *)(partial + ((int64_t)prob * S + sp) * 9 * 65536))[i4];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    // gathered gradients whose row length is not a multiple of 8: tile rows shift_from .. shift_from+7 of the LAST problem hold
    // the row's last 8 channels (conv3x3_wgrad_kernel<true>), i.e. tile row r there is channel r - shift; the first
    // `shift` of them repeat channels already produced
    int n_out = n;
    if (shift > 0 && prob == (int)gridDim.y - 1 && n >= shift_from) {
        if (n < shift_from + shift || n >= shift_from + 8) return;
        n_out = n - shift;
    } else if (n >= rows) return;
    rn::u32x2 o;
    o.x = rn::dt<DT>::pk(s.x, s.y); o.y = rn::dt<DT>::pk(s.z, s.w);
    *(rn::u32x2 *)(dw + ((int64_t)n_out * 9 + t) * 256 + c) = o;
}

// The same sum for the DENSE launch, whose problems own different numbers of splits: partial[split][t][n][c], splits
// beg[p] .. beg[p + 1] belong to problem p.
struct DenseSplits { int beg[CONV_MAX_PROBLEMS + 1]; };
template <int DT>
__global__ __launch_bounds__(256) void wgrad_reduce_dense_kernel(const float *__restrict__ partial, const DenseSplits sp, uint16_t *dw0,
                                                                 uint16_t *dw1, uint16_t *dw2, uint16_t *dw3)
{
    const int prob = blockIdx.y;
    uint16_t *dw = prob == 0 ? dw0 : (prob == 1 ? dw1 : (prob == 2 ? dw2 : dw3));
    const int64_t i4 = (int64_t)blockIdx.x * 256 + threadIdx.x;          // over 9 * 256 * 256 / 4 float4 groups of [t][n][c]
    if (i4 >= 9 * 65536 / 4) return;
    const int64_t e = i4 * 4;
    const int t = (int)(e / 65536), n = (int)((e % 65536) / 256), c = (int)(e % 256);
    rn::f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int q = sp.beg[prob]; q < sp.beg[prob + 1]; ++q) {
        const rn::f32x4 v = ((const rn::f32x4 *)(partial + (int64_t)q * 9 * 65536))[i4];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    rn::u32x2 o;
    o.x = rn::dt<DT>::pk(s.x, s.y); o.y = rn::dt<DT>::pk(s.z, s.w);
    *(rn::u32x2 *)(dw + ((int64_t)n * 9 + t) * 256 + c) = o;
}


// Data-gradient weights of P convs in one launch: out[ci][8 - t][co] = w[co][t][ci] (taps reversed, channel roles swapped),
// w [Cout][9][Cin], out [Cin][9][Cout], 16-bit elements; 32 x 32 tiles through LDS.  (torch: flip + transpose + contiguous,
// four small launches per weight, every step.)
struct WeightPrep { const uint16_t *w[CONV_MAX_PROBLEMS]; uint16_t *out[CONV_MAX_PROBLEMS]; int Cout, Cin; };
__global__ __launch_bounds__(256) void dgrad_weight_kernel(const WeightPrep a)
{
    __shared__ uint16_t tile[32][34];
    const int t = blockIdx.x % 9, p = blockIdx.x / 9;
    const int co0 = blockIdx.y * 32, ci0 = blockIdx.z * 32;
    const uint16_t *__restrict__ w = a.w[p];
    uint16_t *__restrict__ out = a.out[p];
    const int c = threadIdx.x & 31, r0 = threadIdx.x >> 5;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = r0 + 8 * k;                                  // co offset
        tile[r][c] = w[((int64_t)(co0 + r) * 9 + t) * a.Cin + ci0 + c];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = r0 + 8 * k;                                  // ci offset
        out[((int64_t)(ci0 + r) * 9 + (8 - t)) * a.Cout + co0 + c] = tile[c][r];
    }
}

// The same for a whole model's 3x3 convolutions of DIFFERENT widths in one launch (every dependent launch costs ~4.7 us of the
// step's timeline, and a step flipped its weights in 18 of them): 1-D grid over the (entry, tap, 32 x 32 tile) units.
constexpr int DW_MANY_MAX = 48;
struct WeightPrepMany { const uint16_t *w[DW_MANY_MAX]; uint16_t *out[DW_MANY_MAX]; int cout[DW_MANY_MAX], cin[DW_MANY_MAX], first[DW_MANY_MAX + 1]; int n; };
__global__ __launch_bounds__(256) void dgrad_weight_many_kernel(const WeightPrepMany a)
{
    __shared__ uint16_t tile[32][34];
    int e = 0;
    while (e + 1 < a.n && (int)blockIdx.x >= a.first[e + 1]) ++e;     // (block-uniform)
    const int Cout = a.cout[e], Cin = a.cin[e];
    int u = (int)blockIdx.x - a.first[e];
    const int t = u % 9; u /= 9;
    const int co0 = (u % (Cout / 32)) * 32, ci0 = (u / (Cout / 32)) * 32;
    const uint16_t *__restrict__ w = a.w[e];
    uint16_t *__restrict__ out = a.out[e];
    const int c = threadIdx.x & 31, r0 = threadIdx.x >> 5;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = r0 + 8 * k;                                  // co offset
        tile[r][c] = w[((int64_t)(co0 + r) * 9 + t) * Cin + ci0 + c];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = r0 + 8 * k;                                  // ci offset
        out[((int64_t)(ci0 + r) * 9 + (8 - t)) * Cout + co0 + c] = tile[c][r];
    }
}

// Pack the per-level tensors [n_images][h][w][C] onto the canvas sheets [N][Hp][Wp][C] (gaps zeroed) or unpack them, through
// the position map: one launch instead of a zero fill plus one strided copy per (level, slot) -- 11 launches for five levels on
// two-image sheets, four times per step.  UNIT = bytes per thread item (16, 8 or 4: C * 2 must be a multiple).
template <int UNIT, bool TO_CANVAS>
__global__ __launch_bounds__(256) void canvas_pack_kernel(const LevelSet ls, uint16_t *__restrict__ canvas, const int64_t M, const int HWp,
                                                          const int units_per_row)
{
    typedef typename std::conditional<UNIT == 16, uint4, typename std::conditional<UNIT == 8, uint2, uint32_t>::type>::type V;
    const LevelRegs lr = level_regs(ls);
    const int64_t total = M * units_per_row;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t m = i / units_per_row;
        const int u = (int)(i - m * units_per_row);
        int n, pos;
        sheet_coords((int)m, HWp, n, pos);
        uint16_t *row = level_row(ls, lr, n, ls.map[pos]);
        V *c = (V *)(canvas + m * ls.row_elems) + u;
        if (TO_CANVAS) {
            V v = {};
            if (row) v = ((const V *)row)[u];
            *c = v;
        } else if (row) {
            ((V *)row)[u] = *c;
        }
    }
}

// Column sums of up to 6 dense row-major tensors [rows_l][C] of 16-bit floats whose row length C (even) is NOT a multiple
// of the 8-element vector: the bias gradient of the class-output conv over the per-level logit gradients (C = 810: torch needs
// five reductions + fills + adds, 200 us per step, for what is one 290 MB read).  A "super-row" of S = 8 / gcd(8, C) rows is a
// whole number V = C * S / 8 of 16-byte vectors, so thread slot j of a super-row always sees the same 8 columns and sums them
// in registers; G super-rows per block iteration; block totals in a fixed order -> partial[block][C]; colsum_reduce_kernel
// adds the blocks (and thereby the levels).  Rows beyond the last whole super-row: element loop of the level's block 0.
constexpr int CSR_BLOCKS = 128;             // blocks per tensor
struct ColsumRowsArgs { const uint16_t *x[CONV_LEVELS]; int64_t rows[CONV_LEVELS]; int C, S, V, G; float *partial; };
template <int DT>
__global__ __launch_bounds__(1024) void colsum_rows_kernel(const ColsumRowsArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float s_acc[];            // [G][V * 8]
    const int l = blockIdx.y, t = threadIdx.x;
    const int j = t % a.V, g = t / a.V;
    const uint16_t *__restrict__ x = a.x[l];
    const int64_t srows = a.rows[l] / a.S;                                    // whole super-rows
    float acc[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    if (g < a.G) {
        const int64_t step = (int64_t)gridDim.x * a.G;
        int64_t sr = (int64_t)blockIdx.x * a.G + g;
        const rn::u32x4 *xv = (const rn::u32x4 *)x;
        for (; sr + step < srows; sr += 2 * step) {                           // two loads in flight
            float f0[8], f1[8];
            rn::dt<DT>::unpack(xv[sr * a.V + j], f0);
            rn::dt<DT>::unpack(xv[(sr + step) * a.V + j], f1);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += f0[e] + f1[e];
        }
        for (; sr < srows; sr += step) {
            float f0[8];
            rn::dt<DT>::unpack(xv[sr * a.V + j], f0);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += f0[e];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) s_acc[(g * a.V + j) * 8 + e] = acc[e];
    }
    __syncthreads();
    const int E = a.V * 8;                                                    // elements of a super-row = S * C
    for (int c = t; c < a.C; c += blockDim.x) {
        float tot = 0.0f;
        for (int gg = 0; gg < a.G; ++gg)
            for (int r = 0; r < a.S; ++r) tot += s_acc[gg * E + r * a.C + c];
        if (blockIdx.x == 0)                                                  // leftover rows of this tensor
            for (int64_t row = srows * a.S; row < a.rows[l]; ++row) tot += rn::dt<DT>::ld(x, row * a.C + c);
        a.partial[((int64_t)l * gridDim.x + blockIdx.x) * a.C + c] = tot;
    }
}

// Column sums of a fused data-gradient epilogue: partial [P][tiles][C] f32 (one row per row tile) -> out[p][C] f32, summed in
// double in a fixed order (deterministic).  Block = 8 channels x 32 lanes over the tiles.
__global__ __launch_bounds__(256) void colsum_reduce_kernel(const float *__restrict__ partial, const int tiles, const int C, float *out0,
                                                            float *out1, float *out2, float *out3)
{
    __shared__ double sh[32][8];
    const int p = blockIdx.y, ch = threadIdx.x & 7, ln = threadIdx.x >> 3;
    const int c = blockIdx.x * 8 + ch;
    const float *src = partial + (int64_t)p * tiles * C;
    double s = 0.0;
    if (c < C) {
        int b = ln;
        for (; b + 96 < tiles; b += 128) {
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = src[(int64_t)(b + 32 * u) * C + c];
#pragma unroll
            for (int u = 0; u < 4; ++u) s += (double)v[u];
        }
        for (; b < tiles; b += 32) s += (double)src[(int64_t)b * C + c];
    }
    sh[ln][ch] = s;
    __syncthreads();
    if (ln == 0 && c < C) {
        double t = 0.0;
        for (int l = 0; l < 32; ++l) t += sh[l][ch];
        float *out = p == 0 ? out0 : (p == 1 ? out1 : (p == 2 ? out2 : out3));
        out[c] = (float)t;
    }
}

}  // namespace

#ifndef CONV_XCD_MAP
#define CONV_XCD_MAP 1
#endif
template <int DT, int MODE, bool NARROW>
static int conv_launch_dt(const ConvArgs &a, const dim3 grid, hipStream_t st)
{
    if (MODE != MODE_DENSE && (a.M >= ((int64_t)1 << 31) || a.HWp >= ((int64_t)1 << 31))) return RN_EUNSUPPORTED;      // 32-bit position arithmetic in the kernel
    // 160 KiB of dynamic LDS needs the opt-in once per device (the attribute lives with the device's code object)
    static rn::DynLdsOptIn opt_in = {};
    { const int rc = opt_in.ensure((const void *)conv3x3_canvas_kernel<DT, MODE, NARROW>, CONV_LDS_BYTES); if (rc != RN_OK) return rc; }
    if (CONV_XCD_MAP && grid.x >= 16) {
        ConvArgs b = a;
        b.xcd_ny = (int)grid.y; b.xcd_q = (int)(grid.x / 8); b.xcd_r = (int)(grid.x % 8);
        hipLaunchKernelGGL((conv3x3_canvas_kernel<DT, MODE, NARROW>), dim3(8u * (unsigned)(b.xcd_q + (b.xcd_r > 0)) * grid.y, 1, grid.z), dim3(CONV_THREADS),
                           CONV_LDS_BYTES, st, b);
    } else
    hipLaunchKernelGGL((conv3x3_canvas_kernel<DT, MODE, NARROW>), grid, dim3(CONV_THREADS), CONV_LDS_BYTES, st, a);
    RN_LAUNCH_CHECK();
    return RN_OK;
}
template <int MODE, bool NARROW = false>
static int conv_launch_mode(const ConvArgs &a, const dim3 grid, hipStream_t st)
{
    return a.f16 ? conv_launch_dt<RN_F16, MODE, NARROW>(a, grid, st) : conv_launch_dt<RN_BF16, MODE, NARROW>(a, grid, st);
}
static inline bool conv_dtype_ok(const int dtype) { return dtype == RN_BF16 || dtype == RN_F16; }

static int fill_levels(LevelSet &ls, const rn_canvas_layout *lay, int row_elems, void *const *ptrs, int N, int Hp, int Wp)
{
    if (!lay || !lay->map || !ptrs || row_elems <= 0 || (row_elems & 1)) return RN_EINVAL;
    if (lay->T <= 0 || lay->slots <= 0 || lay->n_images <= 0 || (int64_t)N * lay->slots < lay->n_images) return RN_EINVAL;
    if (lay->T > CONV_LEVELS || lay->slots > 8) return RN_EUNSUPPORTED;
    ls.map = lay->map; ls.row_elems = row_elems; ls.slots = lay->slots; ls.n_images = lay->n_images; ls.T = lay->T;
    for (int l = 0; l < CONV_LEVELS; ++l) {
        const int q = l < lay->T ? l : 0;
        if (!ptrs[q] || lay->hw[q] <= 0 || lay->hw[q] >= (1 << 24) || lay->hw[q] > Hp * Wp) return RN_EINVAL;
        if (!rn::aligned(ptrs[q], 16)) return RN_EALIGN;
        if ((int64_t)lay->n_images * lay->hw[q] >= (1ll << 31)) return RN_EUNSUPPORTED;
        ls.hw[l] = lay->hw[q];
        ls.ptr[l] = (uint16_t *)ptrs[q];
    }
    return RN_OK;
}

RN_API int rn_conv3x3_canvas_batched_ex(const void *const *xs, const void *const *ws, const float *const *biases,
                                        const uint8_t *mask, void *const *ys, uint8_t *const *relu_mask_outs, int P, int dtype, int64_t M,
                                        int64_t HWp, int Wp, int Cin, int Cout, int relu, void *stream);

RN_API int rn_conv3x3_canvas_batched(const void *const *xs, const void *const *ws, const float *const *biases,
                                     const uint8_t *mask, void *const *ys, int P, int dtype, int64_t M, int64_t HWp, int Wp,
                                     int Cin, int Cout, int relu, void *stream)
{
    return rn_conv3x3_canvas_batched_ex(xs, ws, biases, mask, ys, nullptr, P, dtype, M, HWp, Wp, Cin, Cout, relu, stream);
}

RN_API int rn_conv3x3_canvas_batched_ex(const void *const *xs, const void *const *ws, const float *const *biases,
                                        const uint8_t *mask, void *const *ys, uint8_t *const *relu_mask_outs, int P, int dtype, int64_t M,
                                        int64_t HWp, int Wp, int Cin, int Cout, int relu, void *stream)
{
    if (relu_mask_outs && !relu) return RN_EINVAL;
    if (!xs || !ws || !ys || P <= 0 || P > CONV_MAX_PROBLEMS || M <= 0 || HWp <= 0 || Wp <= 0 || Cin <= 0 || Cout <= 0) return RN_EINVAL;
    if (!conv_dtype_ok(dtype) || Cin % CONV_BK || Cout % CONV_BN) return RN_EUNSUPPORTED;
    ConvArgs a = {};
    a.f16 = dtype == RN_F16;
    for (int p = 0; p < CONV_MAX_PROBLEMS; ++p) {
        const int q = p < P ? p : 0;
        if (!xs[q] || !ws[q] || !ys[q]) return RN_EINVAL;
        if (!rn::aligned(xs[q], 16) || !rn::aligned(ws[q], 16) || !rn::aligned(ys[q], 16)) return RN_EALIGN;
        a.Xs[p] = (const uint16_t *)xs[q]; a.Ws[p] = (const uint16_t *)ws[q]; a.Ys[p] = (uint16_t *)ys[q];
        a.biases[p] = biases ? biases[q] : nullptr;
        a.relu_mask_outs[p] = relu_mask_outs ? relu_mask_outs[q] : nullptr;
        if (relu_mask_outs && (!relu_mask_outs[q] || !rn::aligned(relu_mask_outs[q], 16))) return relu_mask_outs[q] ? RN_EALIGN : RN_EINVAL;
    }
    a.mask = mask; a.M = M; a.HWp = HWp; a.Cin = Cin; a.Cout = Cout; a.Wp = Wp; a.relu = relu ? 1 : 0; a.zeros = nullptr;
    const dim3 grid((unsigned)((M + CONV_BM - 1) / CONV_BM), (unsigned)(Cout / CONV_BN), (unsigned)P);
    return conv_launch_mode<MODE_CANVAS>(a, grid, (hipStream_t)stream);
}

RN_API int rn_conv3x3_canvas_sum2(const void *x, const void *x2, const void *w, const uint8_t *mask, void *y, int dtype, int64_t M,
                                  int64_t HWp, int Wp, int C, int Cout, void *stream)
{
    if (!x || !x2 || !w || !y || M <= 0 || HWp <= 0 || Wp <= 0 || C <= 0 || Cout <= 0) return RN_EINVAL;
    if (!conv_dtype_ok(dtype) || C % CONV_BK || Cout % CONV_BN) return RN_EUNSUPPORTED;
    if (!rn::aligned(x, 16) || !rn::aligned(x2, 16) || !rn::aligned(w, 16) || !rn::aligned(y, 16)) return RN_EALIGN;
    ConvArgs a = {};
    a.f16 = dtype == RN_F16;
    for (int p = 0; p < CONV_MAX_PROBLEMS; ++p) {
        a.Xs[p] = (const uint16_t *)x; a.X2s[p] = (const uint16_t *)x2; a.Ws[p] = (const uint16_t *)w; a.Ys[p] = (uint16_t *)y;
    }
    a.x_split = C / CONV_BK; a.x_ld = C;
    a.mask = mask; a.M = M; a.HWp = HWp; a.Cin = 2 * C; a.Cout = Cout; a.Wp = Wp; a.relu = 0; a.zeros = nullptr;
    const dim3 grid((unsigned)((M + CONV_BM - 1) / CONV_BM), (unsigned)(Cout / CONV_BN), 1);
    return conv_launch_mode<MODE_CANVAS>(a, grid, (hipStream_t)stream);
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

RN_API int rn_conv3x3_dgrad_weight_batched(const void *const *ws, void *const *outs, int P, int Cout, int Cin, void *stream)
{
    if (!ws || !outs || P <= 0 || P > CONV_MAX_PROBLEMS || Cout <= 0 || Cin <= 0) return RN_EINVAL;
    if (Cout % 32 || Cin % 32) return RN_EUNSUPPORTED;
    WeightPrep a = {};
    for (int p = 0; p < CONV_MAX_PROBLEMS; ++p) {
        const int q = p < P ? p : 0;
        if (!ws[q] || !outs[q] || ws[q] == outs[q]) return RN_EINVAL;
        a.w[p] = (const uint16_t *)ws[q]; a.out[p] = (uint16_t *)outs[q];
    }
    a.Cout = Cout; a.Cin = Cin;
    hipLaunchKernelGGL(dgrad_weight_kernel, dim3((unsigned)(9 * P), (unsigned)(Cout / 32), (unsigned)(Cin / 32)), dim3(256), 0,
                       (hipStream_t)stream, a);
    RN_LAUNCH_CHECK();
    return RN_OK;
}

RN_API int rn_conv3x3_dgrad_weight_many(const void *const *ws, void *const *outs, const int *couts, const int *cins, int n, void *stream)
{
    if (!ws || !outs || !couts || !cins || n <= 0) return RN_EINVAL;
    for (int i0 = 0; i0 < n; i0 += DW_MANY_MAX) {
        WeightPrepMany a = {};
        a.n = n - i0 < DW_MANY_MAX ? n - i0 : DW_MANY_MAX;
        int blocks = 0;
        for (int i = 0; i < a.n; ++i) {
            const int q = i0 + i;
            if (!ws[q] || !outs[q] || ws[q] == outs[q] || couts[q] <= 0 || cins[q] <= 0) return RN_EINVAL;
            if (couts[q] % 32 || cins[q] % 32) return RN_EUNSUPPORTED;
            a.w[i] = (const uint16_t *)ws[q]; a.out[i] = (uint16_t *)outs[q]; a.cout[i] = couts[q]; a.cin[i] = cins[q];
            a.first[i] = blocks;
            blocks += 9 * (couts[q] / 32) * (cins[q] / 32);
        }
        a.first[a.n] = blocks;
        hipLaunchKernelGGL(dgrad_weight_many_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
        RN_LAUNCH_CHECK();
    }
    return RN_OK;
}

RN_API int rn_canvas_pack(void *const *levels, const rn_canvas_layout *layout, void *canvas, int dtype, int N, int Hp, int Wp, int C,
                          int to_canvas, void *stream)
{
    if (!levels || !canvas || N <= 0 || Hp <= 0 || Wp <= 0 || C <= 0) return RN_EINVAL;
    if ((dtype != RN_BF16 && dtype != RN_F16) || (C & 1) || (int64_t)N * Hp * Wp >= (1 << 22)) return RN_EUNSUPPORTED;
    if (!rn::aligned(canvas, 16)) return RN_EALIGN;
    LevelSet ls = {};
    const int rc = fill_levels(ls, layout, C, levels, N, Hp, Wp);
    if (rc != RN_OK) return rc;
    const int64_t M = (int64_t)N * Hp * Wp;
    const int unit = (C % 8 == 0) ? 16 : ((C % 4 == 0) ? 8 : 4);
    const int upr = C * 2 / unit;
    int64_t blocks = (M * upr + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipStream_t st = (hipStream_t)stream;
#define RN_PACK(U)                                                                                                              \
    if (to_canvas) hipLaunchKernelGGL((canvas_pack_kernel<U, true>), dim3((unsigned)blocks), dim3(256), 0, st, ls, (uint16_t *)canvas, M, Hp * Wp, upr); \
    else hipLaunchKernelGGL((canvas_pack_kernel<U, false>), dim3((unsigned)blocks), dim3(256), 0, st, ls, (uint16_t *)canvas, M, Hp * Wp, upr);
    if (unit == 16) { RN_PACK(16) } else if (unit == 8) { RN_PACK(8) } else { RN_PACK(4) }
#undef RN_PACK
    RN_LAUNCH_CHECK();
    return RN_OK;
}

RN_API size_t rn_colsum_rows_workspace_bytes(int L, int C)
{
    if (L <= 0 || L > CONV_LEVELS || C <= 0) return 0;
    return (size_t)L * CSR_BLOCKS * (size_t)C * sizeof(float);
}

RN_API int rn_colsum_rows(const void *const *xs, const int64_t *rows, int L, int C, int dtype, float *out, void *workspace,
                          size_t workspace_bytes, void *stream)
{
    if (!xs || !rows || !out || !workspace || L <= 0 || L > CONV_LEVELS || C <= 0) return RN_EINVAL;
    if ((dtype != RN_BF16 && dtype != RN_F16) || (C & 1)) return RN_EUNSUPPORTED;
    if (workspace_bytes < rn_colsum_rows_workspace_bytes(L, C)) return RN_EWORKSPACE;
    ColsumRowsArgs a = {};
    int g8 = 8;                                                        // gcd(8, C)
    while (C % g8) g8 >>= 1;
    a.C = C; a.S = 8 / g8; a.V = C / g8;
    if (a.V > 1024) return RN_EUNSUPPORTED;
    a.G = 1024 / a.V;
    if (a.G > 8) a.G = 8;
    for (int l = 0; l < CONV_LEVELS; ++l) {
        const int q = l < L ? l : 0;
        if (!xs[q] || rows[q] < 0) return RN_EINVAL;
        if (!rn::aligned(xs[q], 16)) return RN_EALIGN;
        a.x[l] = (const uint16_t *)xs[q]; a.rows[l] = rows[q];
    }
    a.partial = (float *)workspace;
    const int threads = ((a.V * a.G + 63) / 64) * 64;
    const size_t lds = (size_t)a.G * a.V * 8 * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == RN_BF16) hipLaunchKernelGGL((colsum_rows_kernel<RN_BF16>), dim3(CSR_BLOCKS, (unsigned)L), dim3((unsigned)threads), lds, st, a);
    else hipLaunchKernelGGL((colsum_rows_kernel<RN_F16>), dim3(CSR_BLOCKS, (unsigned)L), dim3((unsigned)threads), lds, st, a);
    RN_LAUNCH_CHECK();
    hipLaunchKernelGGL(colsum_reduce_kernel, dim3((unsigned)((C + 7) / 8), 1), dim3(256), 0, st, (const float *)workspace, L * CSR_BLOCKS, C, out,
                       out, out, out);
    RN_LAUNCH_CHECK();
    return RN_OK;
}

RN_API size_t rn_conv3x3_colsum_workspace_bytes(int P, int64_t M, int Cout)
{
    if (P <= 0 || P > CONV_MAX_PROBLEMS || M <= 0 || Cout <= 0) return 0;
    return (size_t)P * (size_t)((M + CONV_BM - 1) / CONV_BM) * (size_t)Cout * sizeof(float);
}

RN_API int rn_conv3x3_canvas_dgrad_relu_batched(const void *const *gs, const void *const *ws, const uint8_t *const *relu_masks,
                                                const uint8_t *mask, void *const *ys, float *const *dbiases, int P, int dtype,
                                                int64_t M, int64_t HWp, int Wp, int Cin, int Cout, void *workspace,
                                                size_t workspace_bytes, void *stream)
{
    if (!gs || !ws || !relu_masks || !ys || !dbiases || !workspace || P <= 0 || P > CONV_MAX_PROBLEMS || M <= 0 || HWp <= 0 || Wp <= 0 ||
        Cin <= 0 || Cout <= 0)
        return RN_EINVAL;
    if (!conv_dtype_ok(dtype) || Cin % CONV_BK || Cout % CONV_BN) return RN_EUNSUPPORTED;
    if (workspace_bytes < rn_conv3x3_colsum_workspace_bytes(P, M, Cout)) return RN_EWORKSPACE;
    ConvArgs a = {};
    a.f16 = dtype == RN_F16;
    const int64_t tiles = (M + CONV_BM - 1) / CONV_BM;
    float *outs[CONV_MAX_PROBLEMS] = {nullptr, nullptr, nullptr, nullptr};
    for (int p = 0; p < CONV_MAX_PROBLEMS; ++p) {
        const int q = p < P ? p : 0;
        if (!gs[q] || !ws[q] || !ys[q] || !relu_masks[q] || !dbiases[q]) return RN_EINVAL;
        if (!rn::aligned(gs[q], 16) || !rn::aligned(ws[q], 16) || !rn::aligned(ys[q], 16) || !rn::aligned(relu_masks[q], 16)) return RN_EALIGN;
        a.Xs[p] = (const uint16_t *)gs[q]; a.Ws[p] = (const uint16_t *)ws[q]; a.Ys[p] = (uint16_t *)ys[q];
        a.biases[p] = nullptr;
        a.relu_masks[p] = relu_masks[q];
        a.colsums[p] = (float *)workspace + (int64_t)q * tiles * Cout;
        outs[p] = dbiases[q];
    }
    a.mask = mask; a.M = M; a.HWp = HWp; a.Cin = Cin; a.Cout = Cout; a.Wp = Wp; a.relu = 0; a.zeros = nullptr;
    const dim3 grid((unsigned)tiles, (unsigned)(Cout / CONV_BN), (unsigned)P);
    const int rc = conv_launch_mode<MODE_CANVAS>(a, grid, (hipStream_t)stream);
    if (rc != RN_OK) return rc;
    hipLaunchKernelGGL(colsum_reduce_kernel, dim3((unsigned)((Cout + 7) / 8), (unsigned)P), dim3(256), 0, (hipStream_t)stream,
                       (const float *)workspace, (int)tiles, Cout, outs[0], outs[1], outs[2], outs[3]);
    RN_LAUNCH_CHECK();
    return RN_OK;
}

RN_API int rn_conv3x3_canvas_to_levels(const void *x, const void *w, const float *bias, const rn_canvas_layout *layout,
                                       void *const *ys, int dtype, int N, int Hp, int Wp, int Cin, int Cout, const void *zeros,
                                       void *stream)
{
    if (!x || !w || !zeros || N <= 0 || Hp <= 2 || Wp <= 2 || Cin <= 0 || Cout <= 0) return RN_EINVAL;
    if (!conv_dtype_ok(dtype) || Cin % CONV_BK || (Cout & 1) || (int64_t)N * Hp * Wp >= (1 << 22)) return RN_EUNSUPPORTED;
    if (!rn::aligned(x, 16) || !rn::aligned(w, 16) || !rn::aligned(zeros, 16)) return RN_EALIGN;
    ConvArgs a = {};
    a.f16 = dtype == RN_F16;
    const int rc = fill_levels(a.lv, layout, Cout, ys, N, Hp, Wp);
    if (rc != RN_OK) return rc;
    for (int p = 0; p < CONV_MAX_PROBLEMS; ++p) { a.Xs[p] = (const uint16_t *)x; a.Ws[p] = (const uint16_t *)w; a.biases[p] = bias; a.Ys[p] = nullptr; }
    a.mask = nullptr; a.M = (int64_t)N * Hp * Wp; a.HWp = (int64_t)Hp * Wp; a.Cin = Cin; a.Cout = Cout; a.Wp = Wp; a.relu = 0;
    a.zeros = (const uint16_t *)zeros;
    const unsigned tiles_m = (unsigned)((a.M + CONV_BM - 1) / CONV_BM);
    const int full = Cout / CONV_BN, rest = Cout % CONV_BN;
    if (rest > 0 && rest <= 64) {
        // whole 256-column tiles, then the ragged last one (<= 64 columns; the only one of the 36-channel box-output conv) on
        // the NARROW kernel
        if (full > 0) {
            const int rc2 = conv_launch_mode<MODE_TO_LEVELS>(a, dim3(tiles_m, (unsigned)full, 1), (hipStream_t)stream);
            if (rc2 != RN_OK) return rc2;
        }
        a.n_base = full * CONV_BN;
        if (BAND_NARROW_FWD) {
            static rn::DynLdsOptIn opt_bf = {}, opt_f = {};
            hipStream_t st = (hipStream_t)stream;
            if (a.f16) {
                const int rc3 = opt_f.ensure((const void *)conv3x3_band_narrow_kernel<RN_F16>, BAND_LDS); if (rc3 != RN_OK) return rc3;
                hipLaunchKernelGGL(conv3x3_band_narrow_kernel<RN_F16>, dim3(tiles_m), dim3(CONV_THREADS), BAND_LDS, st, a);
            } else {
                const int rc3 = opt_bf.ensure((const void *)conv3x3_band_narrow_kernel<RN_BF16>, BAND_LDS); if (rc3 != RN_OK) return rc3;
                hipLaunchKernelGGL(conv3x3_band_narrow_kernel<RN_BF16>, dim3(tiles_m), dim3(CONV_THREADS), BAND_LDS, st, a);
            }
            RN_LAUNCH_CHECK();
            return RN_OK;
        }
        return conv_launch_mode<MODE_TO_LEVELS, true>(a, dim3(tiles_m, 1, 1), (hipStream_t)stream);
    }
    const dim3 grid(tiles_m, (unsigned)((Cout + CONV_BN - 1) / CONV_BN), 1);
    return conv_launch_mode<MODE_TO_LEVELS>(a, grid, (hipStream_t)stream);
}

RN_API int rn_conv3x3_levels_to_canvas(const void *const *gs, const rn_canvas_layout *layout, int row_elems, const void *w,
                                       const uint8_t *mask, void *y, int dtype, int N, int Hp, int Wp, int Kpad, int Cout,
                                       const void *zeros, void *stream)
{
    if (!gs || !w || !y || !zeros || N <= 0 || Hp <= 2 || Wp <= 2 || Kpad <= 0 || Cout <= 0) return RN_EINVAL;
    if (!conv_dtype_ok(dtype) || Kpad % CONV_BK || Cout % CONV_BN || Kpad < row_elems || Kpad - row_elems >= CONV_BK ||
        (int64_t)N * Hp * Wp >= (1 << 22) || row_elems < 8)
        return RN_EUNSUPPORTED;
    if (!rn::aligned(w, 16) || !rn::aligned(y, 16) || !rn::aligned(zeros, 16)) return RN_EALIGN;
    ConvArgs a = {};
    a.f16 = dtype == RN_F16;
    const int rc = fill_levels(a.lv, layout, row_elems, const_cast<void *const *>(gs), N, Hp, Wp);
    if (rc != RN_OK) return rc;
    for (int p = 0; p < CONV_MAX_PROBLEMS; ++p) { a.Xs[p] = nullptr; a.Ws[p] = (const uint16_t *)w; a.biases[p] = nullptr; a.Ys[p] = (uint16_t *)y; }
    a.mask = mask; a.M = (int64_t)N * Hp * Wp; a.HWp = (int64_t)Hp * Wp; a.Cin = Kpad; a.Cout = Cout; a.Wp = Wp; a.relu = 0;
    a.zeros = (const uint16_t *)zeros;
    const dim3 grid((unsigned)((a.M + CONV_BM - 1) / CONV_BM), (unsigned)(Cout / CONV_BN), 1);
    return conv_launch_mode<MODE_FROM_LEVELS>(a, grid, (hipStream_t)stream);
}

// rn_conv3x3_levels_to_canvas whose INPUT-side activation was a ReLU output (the last tower layer in front of the class- / box-
// output conv): the result is also multiplied by the ReLU bits of that activation and its column sums -- the bias gradient of the
// layer below -- go to dbias (rn_conv3x3_canvas_dgrad_relu_batched's epilogue on the gathering kernel).
RN_API int rn_conv3x3_levels_to_canvas_relu(const void *const *gs, const rn_canvas_layout *layout, int row_elems, const void *w,
                                            const uint8_t *relu_mask, const uint8_t *mask, void *y, float *dbias, int dtype, int N, int Hp,
                                            int Wp, int Kpad, int Cout, const void *zeros, void *workspace, size_t workspace_bytes,
                                            void *stream)
{
    if (!gs || !w || !y || !zeros || !relu_mask || !dbias || !workspace || N <= 0 || Hp <= 2 || Wp <= 2 || Kpad <= 0 || Cout <= 0) return RN_EINVAL;
    if (!conv_dtype_ok(dtype) || Kpad % CONV_BK || Cout % CONV_BN || Kpad < row_elems || Kpad - row_elems >= CONV_BK ||
        (int64_t)N * Hp * Wp >= (1 << 22) || row_elems < 8)
        return RN_EUNSUPPORTED;
    if (!rn::aligned(w, 16) || !rn::aligned(y, 16) || !rn::aligned(zeros, 16) || !rn::aligned(relu_mask, 16)) return RN_EALIGN;
    const int64_t M = (int64_t)N * Hp * Wp;
    if (workspace_bytes < rn_conv3x3_colsum_workspace_bytes(1, M, Cout)) return RN_EWORKSPACE;
    ConvArgs a = {};
    a.f16 = dtype == RN_F16;
    const int rc = fill_levels(a.lv, layout, row_elems, const_cast<void *const *>(gs), N, Hp, Wp);
    if (rc != RN_OK) return rc;
    for (int p = 0; p < CONV_MAX_PROBLEMS; ++p) {
        a.Xs[p] = nullptr; a.Ws[p] = (const uint16_t *)w; a.biases[p] = nullptr; a.Ys[p] = (uint16_t *)y;
        a.relu_masks[p] = relu_mask; a.colsums[p] = (float *)workspace;
    }
    a.mask = mask; a.M = M; a.HWp = (int64_t)Hp * Wp; a.Cin = Kpad; a.Cout = Cout; a.Wp = Wp; a.relu = 0;
    a.zeros = (const uint16_t *)zeros;
    const int64_t tiles = (M + CONV_BM - 1) / CONV_BM;
    const dim3 grid((unsigned)tiles, (unsigned)(Cout / CONV_BN), 1);
    const int rc2 = conv_launch_mode<MODE_FROM_LEVELS>(a, grid, (hipStream_t)stream);
    if (rc2 != RN_OK) return rc2;
    hipLaunchKernelGGL(colsum_reduce_kernel, dim3((unsigned)((Cout + 7) / 8), 1), dim3(256), 0, (hipStream_t)stream, (const float *)workspace,
                       (int)tiles, Cout, dbias, dbias, dbias, dbias);
    RN_LAUNCH_CHECK();
    return RN_OK;
}

#ifndef WGRAD_XCD_MAP
#define WGRAD_XCD_MAP 1
#endif

// Position splits of the weight-gradient kernels: one workgroup per (split, tap, problem), about one wave of the chip.
static int wgrad_splits(const int P, const int64_t M, int *tiles_per_split)
{
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
    int S = cus / (9 * P);
    if (S < 1) S = 1;
    if (S > 64) S = 64;
    const int64_t ktiles = (M + WG_POS - 1) / WG_POS;
    const int tps = (int)((ktiles + S - 1) / S);
    if (tiles_per_split) *tiles_per_split = tps;
    return (int)((ktiles + tps - 1) / tps);
}

// f32 partials of the splits the launch will really use (it used to size for the 64-split maximum: 151 MB per problem)
RN_API size_t rn_conv3x3_wgrad_workspace_bytes(int P, int64_t M)
{
    if (P <= 0 || P > CONV_MAX_PROBLEMS || M <= 0) return 0;
    // (the NARROW box-output launch runs P = 1 after a P - 1 launch in the same workspace: the P-problem figure covers both)
    const int S = wgrad_splits(P, M, nullptr), S1 = wgrad_splits(1, M, nullptr);
    return ((size_t)P * S + (size_t)S1) * 9 * 65536 * sizeof(float);
}

template <int DT, bool GATHER, bool NARROW>
static int wgrad_launch_dt(WgradArgs &a, uint16_t *const (&dw)[CONV_MAX_PROBLEMS], const int (&rows)[CONV_MAX_PROBLEMS], int P, int64_t M,
                           void *workspace, hipStream_t st, int shift_from, int shift, size_t *used_floats)
{
    static rn::DynLdsOptIn opt_in = {};
    { const int rc = opt_in.ensure((const void *)conv3x3_wgrad_kernel<DT, GATHER, NARROW>, CONV_LDS_BYTES); if (rc != RN_OK) return rc; }
    // one workgroup per (split, tap, problem): choose the split count so that the grid is about one wave of the chip
    const int S = wgrad_splits(P, M, &a.tiles_per_split);
    a.S = S; a.M = M; a.partial = (float *)workspace;
    if (used_floats) *used_floats = (size_t)P * S * 9 * 65536;
    const int U = 9 * P;
    a.xcd_units = (GATHER && !NARROW && WGRAD_XCD_MAP && S <= 7 && U >= 32 && (U - 32) * S <= 32 && U <= 36) ? U : 0;
    if (a.xcd_units > 0)
        hipLaunchKernelGGL((conv3x3_wgrad_kernel<DT, GATHER, NARROW>), dim3(256, 1, 1), dim3(CONV_THREADS), CONV_LDS_BYTES, st, a);
    else
    hipLaunchKernelGGL((conv3x3_wgrad_kernel<DT, GATHER, NARROW>), dim3((unsigned)S, 9, (unsigned)P), dim3(CONV_THREADS), CONV_LDS_BYTES, st, a);
    RN_LAUNCH_CHECK();
    hipLaunchKernelGGL(wgrad_reduce_kernel<DT>, dim3(9 * 65536 / 4 / 256, (unsigned)P), dim3(256), 0, st, (const float *)workspace, S, dw[0], dw[1],
                       dw[2], dw[3], rows[0], rows[1], rows[2], rows[3], shift_from, shift);
    RN_LAUNCH_CHECK();
    return RN_OK;
}
template <bool GATHER, bool NARROW = false>
static int wgrad_launch(WgradArgs &a, uint16_t *const (&dw)[CONV_MAX_PROBLEMS], const int (&rows)[CONV_MAX_PROBLEMS], int P, int64_t M,
                        void *workspace, hipStream_t st, int shift_from = 0, int shift = 0, size_t *used_floats = nullptr)
{
    return a.f16 ? wgrad_launch_dt<RN_F16, GATHER, NARROW>(a, dw, rows, P, M, workspace, st, shift_from, shift, used_floats)
                 : wgrad_launch_dt<RN_BF16, GATHER, NARROW>(a, dw, rows, P, M, workspace, st, shift_from, shift, used_floats);
}

RN_API int rn_conv3x3_canvas_wgrad_batched(const void *const *gs, const void *const *xs, void *const *dws, int P, int dtype,
                                           int64_t M, int Wp, int Cin, int Cout, const void *zeros, void *workspace,
                                           size_t workspace_bytes, void *stream)
{
    if (!gs || !xs || !dws || !zeros || !workspace || P <= 0 || P > CONV_MAX_PROBLEMS || M <= 0 || Wp <= 0) return RN_EINVAL;
    if (!conv_dtype_ok(dtype) || Cin != 256 || Cout != 256) return RN_EUNSUPPORTED;
    if (workspace_bytes < rn_conv3x3_wgrad_workspace_bytes(P, M)) return RN_EWORKSPACE;
    WgradArgs a = {};
    a.f16 = dtype == RN_F16;
    uint16_t *dw[CONV_MAX_PROBLEMS] = {nullptr, nullptr, nullptr, nullptr};
    const int rows[CONV_MAX_PROBLEMS] = {256, 256, 256, 256};
    for (int p = 0; p < CONV_MAX_PROBLEMS; ++p) {
        const int q = p < P ? p : 0;
        if (!gs[q] || !xs[q] || !dws[q]) return RN_EINVAL;
        if (!rn::aligned(gs[q], 16) || !rn::aligned(xs[q], 16) || !rn::aligned(dws[q], 16)) return RN_EALIGN;
        a.Gs[p] = (const uint16_t *)gs[q]; a.Xs[p] = (const uint16_t *)xs[q];
        dw[p] = (uint16_t *)dws[q];
    }
    a.Wp = Wp; a.HWp = 0; a.zeros = (const uint16_t *)zeros;
    return wgrad_launch<false>(a, dw, rows, P, M, workspace, (hipStream_t)stream);
}

RN_API int rn_conv3x3_levels_wgrad(const void *const *gs, const rn_canvas_layout *layout, int row_elems, const void *x, void *dw,
                                   int dtype, int N, int Hp, int Wp, int Cin, const void *zeros, void *workspace,
                                   size_t workspace_bytes, void *stream)
{
    if (!gs || !x || !dw || !zeros || !workspace || N <= 0 || Hp <= 2 || Wp <= 2) return RN_EINVAL;
    const int P = (row_elems + 255) / 256;
    const int64_t M = (int64_t)N * Hp * Wp;
    if (!conv_dtype_ok(dtype) || Cin != 256 || P > CONV_MAX_PROBLEMS || M >= (1 << 22) || row_elems < 8) return RN_EUNSUPPORTED;
    if (workspace_bytes < rn_conv3x3_wgrad_workspace_bytes(P, M)) return RN_EWORKSPACE;
    if (!rn::aligned(x, 16) || !rn::aligned(dw, 16) || !rn::aligned(zeros, 16)) return RN_EALIGN;
    WgradArgs a = {};
    a.f16 = dtype == RN_F16;
    const int rc = fill_levels(a.lv, layout, row_elems, const_cast<void *const *>(gs), N, Hp, Wp);
    if (rc != RN_OK) return rc;
    uint16_t *dws[CONV_MAX_PROBLEMS];
    int rows[CONV_MAX_PROBLEMS];
    for (int p = 0; p < CONV_MAX_PROBLEMS; ++p) {
        a.Gs[p] = nullptr; a.Xs[p] = (const uint16_t *)x;
        dws[p] = (uint16_t *)dw + (int64_t)p * 256 * 9 * 256;                      // dW is [row_elems][3][3][256]
        rows[p] = row_elems - p * 256 < 256 ? (row_elems - p * 256 > 0 ? row_elems - p * 256 : 0) : 256;
    }
    a.Wp = Wp; a.HWp = (int64_t)Hp * Wp; a.zeros = (const uint16_t *)zeros;
    // tile-local position and size of the straddling piece's shift (see the kernel): last problem only
    const int tail = row_elems % 8, e_last = row_elems - tail - (P - 1) * 256;
    if (P == 1 && rows[0] <= 64 && (!tail || e_last + 8 <= 64)) {
        // a conv with at most 64 output channels (the 36-channel box-output conv) on the NARROW kernel.  (Splitting the ragged last
        // row tile of the 810-channel conv off the same way was measured and is slower -- 0.94 vs 0.91 ms: a K-tile of the
        // gathering kernel costs the same 1.9 us with a quarter of the MFMAs, it is bound by staging, not by the matrix pipe.)
        size_t used = 0;
        if (P > 1) {
            const int rc2 = wgrad_launch<true>(a, dws, rows, P - 1, M, workspace, (hipStream_t)stream, 0, 0, &used);
            if (rc2 != RN_OK) return rc2;
        }
        uint16_t *dw1[CONV_MAX_PROBLEMS] = {dws[P - 1], dws[P - 1], dws[P - 1], dws[P - 1]};
        const int rows1[CONV_MAX_PROBLEMS] = {rows[P - 1], 0, 0, 0};
        a.ch_base = (P - 1) * 256;
        return wgrad_launch<true, true>(a, dw1, rows1, 1, M, (float *)workspace + used, (hipStream_t)stream, tail ? e_last : 0, tail ? 8 - tail : 0);
    }
    return wgrad_launch<true>(a, dws, rows, P, M, workspace, (hipStream_t)stream, tail ? e_last : 0, tail ? 8 - tail : 0);
}


// ---- MODE_DENSE entry points: P <= 4 convolutions 3x3 / stride 1 / pad 1 with their own [N][h_p][w_p] geometry and weights ----
static int dense_geom(DenseGeom &g, int P, int N, const int *hs, const int *wds)
{
    if (!hs || !wds || P <= 0 || P > CONV_MAX_PROBLEMS || N <= 0) return RN_EINVAL;
    for (int p = 0; p < CONV_MAX_PROBLEMS; ++p) {
        const int q = p < P ? p : 0;
        if (hs[q] <= 0 || wds[q] <= 0) return RN_EINVAL;
        const int64_t M = (int64_t)N * hs[q] * wds[q];
        if (M >= (1 << 22)) return RN_EUNSUPPORTED;                 // sheet_coords' range (weight gradient); 32-bit row arithmetic
        g.M[p] = M; g.h[p] = hs[q]; g.w[p] = wds[q];
    }
    return RN_OK;
}

RN_API int rn_conv3x3_dense_batched_act(const void *const *xs, const void *const *ws, const float *const *biases, void *const *ys, int P,
                                        int dtype, int N, const int *hs, const int *wds, int Cin, int Cout, const void *zeros, int relu,
                                        void *stream)
{
    if (!xs || !ws || !ys || !zeros || Cin <= 0 || Cout <= 0) return RN_EINVAL;
    if (!conv_dtype_ok(dtype) || Cin % CONV_BK || Cout % CONV_BN) return RN_EUNSUPPORTED;
    ConvArgs a = {};
    a.f16 = dtype == RN_F16;
    int rc = dense_geom(a.dn, P, N, hs, wds);
    if (rc != RN_OK) return rc;
    if (!rn::aligned(zeros, 16)) return RN_EALIGN;
    int tiles = 0;
    for (int p = 0; p < CONV_MAX_PROBLEMS; ++p) {
        const int q = p < P ? p : 0;
        if (!xs[q] || !ws[q] || !ys[q]) return RN_EINVAL;
        if (!rn::aligned(xs[q], 16) || !rn::aligned(ws[q], 16) || !rn::aligned(ys[q], 16)) return RN_EALIGN;
        a.Xs[p] = (const uint16_t *)xs[q]; a.Ws[p] = (const uint16_t *)ws[q]; a.Ys[p] = (uint16_t *)ys[q];
        a.biases[p] = biases ? biases[q] : nullptr;
        a.dn.tile_beg[p] = p < P ? tiles : 0x7fffffff;
        if (p < P) tiles += (int)((a.dn.M[p] + CONV_BM - 1) / CONV_BM);
    }
    a.dn.tile_beg[CONV_MAX_PROBLEMS] = 0x7fffffff;
    a.mask = nullptr; a.M = 0; a.HWp = 1; a.Wp = 1; a.Cin = Cin; a.Cout = Cout; a.relu = relu ? 1 : 0; a.zeros = (const uint16_t *)zeros;
    return conv_launch_mode<MODE_DENSE>(a, dim3((unsigned)tiles, (unsigned)(Cout / CONV_BN), 1), (hipStream_t)stream);
}

// y (16-bit) = sum over the K ranges of kpartial[z][M * Cout] (f32), fixed order
template <int DT>
__global__ __launch_bounds__(256) void dense_ksplit_reduce_kernel(const float *__restrict__ partial, const int S, const int64_t n4, uint16_t *__restrict__ y)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        rn::f32x4 s = ((const rn::f32x4 *)partial)[i];
        for (int z = 1; z < S; ++z) { const rn::f32x4 v = ((const rn::f32x4 *)partial)[z * n4 + i]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
        rn::u32x2 o;
        o.x = rn::dt<DT>::pk(s.x, s.y); o.y = rn::dt<DT>::pk(s.z, s.w);
        ((rn::u32x2 *)y)[i] = o;
    }
}

// K ranges of a single dense convolution: as many as keep the launch inside one round of the chip, at most 3 (one kernel row each)
static int dense_ksplit(const int64_t M, const int Cout)
{
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
    const int64_t wgs = ((M + CONV_BM - 1) / CONV_BM) * (Cout / CONV_BN);
    if (wgs <= 0) return 0;
    int s = (int)(cus / wgs);
    return s >= 3 ? 3 : (s >= 2 ? 2 : 1);
}

static int dense_band_launch(const void *x, const void *w, void *y, float *partial, int dtype, int N, int h, int wd, int Cin, int Cout,
                             const void *zeros, void *stream)
{
    if (!x || !w || !y || !zeros || N <= 0 || h <= 0 || wd <= 0 || Cin <= 0 || Cout <= 0) return RN_EINVAL;
    if (!conv_dtype_ok(dtype) || Cin % CONV_BK || Cout % 128) return RN_EUNSUPPORTED;
    const int64_t M = (int64_t)N * h * wd;
    if (M >= ((int64_t)1 << 31) / (Cin > Cout ? Cin : Cout)) return RN_EUNSUPPORTED;
    if (!rn::aligned(x, 16) || !rn::aligned(w, 16) || !rn::aligned(y, 16) || !rn::aligned(zeros, 16)) return RN_EALIGN;
    DenseBandArgs a;
    a.X = (const uint16_t *)x; a.W = (const uint16_t *)w; a.Y = (uint16_t *)y; a.zeros = (const uint16_t *)zeros;
    a.M = M; a.H = h; a.Wd = wd; a.Cin = Cin; a.Cout = Cout; a.partial = partial;
    const dim3 grid((unsigned)((M + CONV_BM - 1) / CONV_BM), (unsigned)(Cout / 128), 1);
    hipStream_t st = (hipStream_t)stream;
    static rn::DynLdsOptIn opt_bf = {}, opt_f = {};
    if (dtype == RN_F16) {
        const int rc = opt_f.ensure((const void *)conv3x3_dense_band_kernel<RN_F16>, DBAND_LDS); if (rc != RN_OK) return rc;
        hipLaunchKernelGGL(conv3x3_dense_band_kernel<RN_F16>, grid, dim3(CONV_THREADS), DBAND_LDS, st, a);
    } else {
        const int rc = opt_bf.ensure((const void *)conv3x3_dense_band_kernel<RN_BF16>, DBAND_LDS); if (rc != RN_OK) return rc;
        hipLaunchKernelGGL(conv3x3_dense_band_kernel<RN_BF16>, grid, dim3(CONV_THREADS), DBAND_LDS, st, a);
    }
    RN_LAUNCH_CHECK();
    return RN_OK;
}

RN_API int rn_conv3x3_dense_band(const void *x, const void *w, void *y, int dtype, int N, int h, int wd, int Cin, int Cout, const void *zeros,
                                 void *stream)
{
    return dense_band_launch(x, w, y, nullptr, dtype, N, h, wd, Cin, Cout, zeros, stream);
}
RN_API int rn_conv3x3_dense_band_tiles(int N, int h, int wd)
{
    if (N <= 0 || h <= 0 || wd <= 0) return 0;
    return (int)(((int64_t)N * h * wd + CONV_BM - 1) / CONV_BM);
}
RN_API int rn_conv3x3_dense_band_stats(const void *x, const void *w, void *y, float *partial, int dtype, int N, int h, int wd, int Cin, int Cout,
                                       const void *zeros, void *stream)
{
    if (!partial) return RN_EINVAL;
    return dense_band_launch(x, w, y, partial, dtype, N, h, wd, Cin, Cout, zeros, stream);
}

RN_API size_t rn_conv3x3_dense_splitk_workspace_bytes(int N, int h, int w, int Cout)
{
    if (N <= 0 || h <= 0 || w <= 0 || Cout <= 0 || Cout % CONV_BN) return 0;
    const int64_t M = (int64_t)N * h * w;
    const int S = dense_ksplit(M, Cout);
    return S > 1 ? (size_t)S * M * Cout * sizeof(float) : 0;
}

RN_API int rn_conv3x3_dense_splitk(const void *x, const void *w, void *y, int dtype, int N, int h, int wd, int Cin, int Cout, const void *zeros,
                                   void *workspace, size_t workspace_bytes, void *stream)
{
    if (!x || !w || !y || !zeros || !workspace || Cin <= 0 || Cout <= 0) return RN_EINVAL;
    if (!conv_dtype_ok(dtype) || Cin % CONV_BK || Cout % CONV_BN) return RN_EUNSUPPORTED;
    ConvArgs a = {};
    a.f16 = dtype == RN_F16;
    const int hs[1] = {h}, wds[1] = {wd};
    const int rc = dense_geom(a.dn, 1, N, hs, wds);
    if (rc != RN_OK) return rc;
    const int64_t M = a.dn.M[0];
    const int S = dense_ksplit(M, Cout);
    if (S < 2 || (M * Cout) % 4) return RN_EUNSUPPORTED;
    if (workspace_bytes < (size_t)S * M * Cout * sizeof(float)) return RN_EWORKSPACE;
    if (!rn::aligned(x, 16) || !rn::aligned(w, 16) || !rn::aligned(y, 16) || !rn::aligned(zeros, 16) || !rn::aligned(workspace, 16)) return RN_EALIGN;
    for (int p = 0; p < CONV_MAX_PROBLEMS; ++p) {
        a.Xs[p] = (const uint16_t *)x; a.Ws[p] = (const uint16_t *)w; a.Ys[p] = (uint16_t *)y; a.biases[p] = nullptr;
        a.dn.tile_beg[p] = p == 0 ? 0 : 0x7fffffff;
    }
    a.dn.tile_beg[CONV_MAX_PROBLEMS] = 0x7fffffff;
    const int tiles = (int)((M + CONV_BM - 1) / CONV_BM);
    a.mask = nullptr; a.M = 0; a.HWp = 1; a.Wp = 1; a.Cin = Cin; a.Cout = Cout; a.relu = 0; a.zeros = (const uint16_t *)zeros;
    a.ksplit = S; a.kpartial = (float *)workspace;
    hipStream_t st = (hipStream_t)stream;
    const int rc2 = conv_launch_mode<MODE_DENSE>(a, dim3((unsigned)tiles, (unsigned)(Cout / CONV_BN), (unsigned)S), st);
    if (rc2 != RN_OK) return rc2;
    const int64_t n4 = M * Cout / 4;
    const unsigned blocks = (unsigned)((n4 + 255) / 256 > 4096 ? 4096 : (n4 + 255) / 256);
    if (a.f16) hipLaunchKernelGGL(dense_ksplit_reduce_kernel<RN_F16>, dim3(blocks), dim3(256), 0, st, (const float *)workspace, S, n4, (uint16_t *)y);
    else hipLaunchKernelGGL(dense_ksplit_reduce_kernel<RN_BF16>, dim3(blocks), dim3(256), 0, st, (const float *)workspace, S, n4, (uint16_t *)y);
    RN_LAUNCH_CHECK();
    return RN_OK;
}

RN_API int rn_conv3x3_dense_batched(const void *const *xs, const void *const *ws, const float *const *biases, void *const *ys, int P,
                                    int dtype, int N, const int *hs, const int *wds, int Cin, int Cout, const void *zeros, void *stream)
{
    return rn_conv3x3_dense_batched_act(xs, ws, biases, ys, P, dtype, N, hs, wds, Cin, Cout, zeros, 0, stream);
}

// splits of the DENSE weight gradient: every split walks `tps` K-tiles (the last one of a problem fewer), as many splits in
// total as fit one round of the chip (cus / 9 workgroups per tap)
static int dense_splits(const DenseGeom &g, int P, int *beg, int *tps_out)
{
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
    int budget = cus / 9;
    if (budget < P) budget = P;
    int64_t kt[CONV_MAX_PROBLEMS], total = 0;
    for (int p = 0; p < P; ++p) { kt[p] = (g.M[p] + WG_POS - 1) / WG_POS; total += kt[p]; }
    int64_t tps = (total + budget - 1) / budget;
    if (tps < 1) tps = 1;
    while (true) {
        int n = 0;
        for (int p = 0; p < P; ++p) n += (int)((kt[p] + tps - 1) / tps);
        if (n <= budget) break;
        ++tps;
    }
    int n = 0;
    for (int p = 0; p < CONV_MAX_PROBLEMS; ++p) {
        beg[p] = p < P ? n : 0x7fffffff;
        if (p < P) n += (int)((kt[p] + tps - 1) / tps);
    }
    beg[CONV_MAX_PROBLEMS] = 0x7fffffff;
    *tps_out = (int)tps;
    return n;
}

RN_API size_t rn_conv3x3_dense_wgrad_workspace_bytes(int P)
{
    if (P <= 0 || P > CONV_MAX_PROBLEMS) return 0;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
    const int budget = cus / 9 < P ? P : cus / 9;
    return (size_t)budget * 9 * 65536 * sizeof(float);
}

RN_API int rn_conv3x3_dense_wgrad_batched(const void *const *gs, const void *const *xs, void *const *dws, int P, int dtype, int N,
                                          const int *hs, const int *wds, int Cin, int Cout, const void *zeros, void *workspace,
                                          size_t workspace_bytes, void *stream)
{
    if (!gs || !xs || !dws || !zeros || !workspace) return RN_EINVAL;
    if (!conv_dtype_ok(dtype) || Cin != 256 || Cout != 256) return RN_EUNSUPPORTED;
    WgradArgs a = {};
    a.f16 = dtype == RN_F16;
    int rc = dense_geom(a.dn, P, N, hs, wds);
    if (rc != RN_OK) return rc;
    if (workspace_bytes < rn_conv3x3_dense_wgrad_workspace_bytes(P)) return RN_EWORKSPACE;
    if (!rn::aligned(zeros, 16) || !rn::aligned(workspace, 16)) return RN_EALIGN;
    uint16_t *dw[CONV_MAX_PROBLEMS] = {nullptr, nullptr, nullptr, nullptr};
    for (int p = 0; p < CONV_MAX_PROBLEMS; ++p) {
        const int q = p < P ? p : 0;
        if (!gs[q] || !xs[q] || !dws[q]) return RN_EINVAL;
        if (!rn::aligned(gs[q], 16) || !rn::aligned(xs[q], 16) || !rn::aligned(dws[q], 16)) return RN_EALIGN;
        a.Gs[p] = (const uint16_t *)gs[q]; a.Xs[p] = (const uint16_t *)xs[q];
        dw[p] = (uint16_t *)dws[q];
    }
    DenseSplits sp;
    int tps = 1;
    const int total = dense_splits(a.dn, P, sp.beg, &tps);
    for (int p = 0; p <= CONV_MAX_PROBLEMS; ++p) a.dn.tile_beg[p] = sp.beg[p];
    for (int p = 0; p < CONV_MAX_PROBLEMS; ++p) a.dn_tps[p] = tps;
    sp.beg[P] = total;                                              // end of the last problem for the reduction
    a.zeros = (const uint16_t *)zeros; a.partial = (float *)workspace; a.S = total; a.M = 0; a.Wp = 1; a.HWp = 1; a.tiles_per_split = tps;
    hipStream_t st = (hipStream_t)stream;
    a.f16 = dtype == RN_F16;
#define RN_DENSE_WGRAD(DT)                                                                                                                    \
    {   static rn::DynLdsOptIn opt_in = {};                                                                                                   \
        const int rc2 = opt_in.ensure((const void *)conv3x3_wgrad_kernel<DT, false, false, true>, CONV_LDS_BYTES);                            \
        if (rc2 != RN_OK) return rc2;                                                                                                         \
        hipLaunchKernelGGL((conv3x3_wgrad_kernel<DT, false, false, true>), dim3((unsigned)total, 9, 1), dim3(CONV_THREADS), CONV_LDS_BYTES, st, a); \
        RN_LAUNCH_CHECK();                                                                                                                    \
        hipLaunchKernelGGL(wgrad_reduce_dense_kernel<DT>, dim3(9 * 65536 / 4 / 256, (unsigned)P), dim3(256), 0, st, (const float *)workspace, sp, dw[0], \
                           dw[1], dw[2], dw[3]);                                                                                              \
        RN_LAUNCH_CHECK(); }
    if (a.f16) RN_DENSE_WGRAD(RN_F16) else RN_DENSE_WGRAD(RN_BF16)
#undef RN_DENSE_WGRAD
    return RN_OK;
}
