// K2 iou_match -- replaces matcher() (retinanet/box_utils.py:51-80) and the
// torchvision box_iou it calls (:74).  The [T,A] IoU matrix is never written:
// anchors live in registers, the images' GT boxes (+ their areas) are staged in
// LDS, and the running (max IoU, first arg-max) pair is reduced in-register.
//
// Two kernels:
//   * iou_match_batch_kernel -- the train-step shape (one anchor set shared by the batch, a few GT per
//     image): a thread owns ONE anchor for a group of images (the whole batch when A alone fills the chip),
//     the group's GT sit in LDS.  HBM traffic per batch: A*16 B anchors (L2 hits for the second group) +
//     sum(T)*16 B GT + B*A*8 B int64 matches.
//   * iou_match_tile_kernel<R> -- everything else (T up to thousands, per-image anchors): a thread owns R
//     anchors (strided by the workgroup size: coalesced 16-byte loads, 8-byte stores); GT tiles of 256 are
//     staged in LDS once per workgroup and every LDS read of a GT box is used for R pairs.
//
// Fast loop: when every GT box of the tile and every anchor of the wave is a proper finite box (x2 > x1, y2 > y1;
// anchors may be degenerate) the union is positive, so the quotient of a pair WITHOUT overlap is +0 and cannot
// change (max, first arg-max).  Two wave-uniform culls follow from that: (1) a GT box that misses the bounding box
// of the wave's anchors (a wave owns consecutive anchors = one strip of the feature map) is skipped for the whole
// wave after 4 compares -- ~90 % of the GT boxes at the headline shapes; (2) otherwise the intersection costs 9
// instructions per pair and the IEEE divide (~12) + update run only when some lane of the wave overlaps.
// Anything else (NaN / Inf / inverted boxes) takes the careful loop: torch's semantics pair by pair.
//
// Bit-exactness with the CPU path: fp32 throughout, association
// (area_t + area_a) - inter, IEEE divide, no FMA contraction (this file is
// compiled with -ffp-contract=off), first index wins ties, NaN propagates as in
// torch.max (first NaN wins, and a NaN max is neither < bg nor > fg -> -2).
#include <cstdlib>

#include "rn_common.hpp"
#include "rn_match.hpp"

namespace {

constexpr int MATCH_BLOCK = 256;
constexpr int GT_TILE = 256;
constexpr int BATCH_GT_MAX = 1024;          // sum(T) the batch kernel stages in LDS (20 KiB)

using namespace rn_match;

// The loss kernel's shortcut to the rows that are not plain background: special[b][a >> 6] bit (a & 63) = [matches[b][a] != -1]
// (matched or ignored: ~0.3 % of the rows at the reference's thresholds).  Every kernel below hands each wave 64 consecutive
// anchors that start on a multiple of 64, so a word is one ballot; K3 then fetches a 64-row chunk's flags with two SCALAR loads
// instead of a 512-byte vector load of `matches`, and reads `matches` only where a bit is set.
__device__ __forceinline__ void special_word(unsigned long long *__restrict__ special, const int b, const int64_t A, const int64_t a_idx,
                                             const bool flag)
{
    const unsigned long long w = __ballot(flag);
    if ((threadIdx.x & (RN_WAVE - 1)) == 0 && (a_idx & ~(int64_t)63) < A) special[(int64_t)b * ((A + 63) >> 6) + (a_idx >> 6)] = w;
}

// ============================================================================================================
// Shared anchors, small GT sets: one thread = one anchor x `ipb` images of the batch (blockIdx.y selects the image group).
// SPARSE: `matches` is written at FLAGGED rows only (code != -1: matched or ignored, ~0.3 % of the rows) -- for the caller that
// reads it through the flag words (the loss kernel): B * A * 8 bytes of int64 stores nobody reads are 80 % of this kernel's traffic
// at the train shape (12.9 of 16.1 MB).
template <bool SPARSE>
__global__ __launch_bounds__(MATCH_BLOCK) void iou_match_batch_kernel(
    const rn::f32x4 *__restrict__ anchors, const rn::f32x4 *__restrict__ gt, const int32_t *__restrict__ gt_off,
    const int B, const int ipb, const int64_t A, const float fg_thr, const float bg_thr, int64_t *__restrict__ matches,
    int32_t *__restrict__ num_fg, unsigned long long *__restrict__ special)
{
    __shared__ rn::f32x4 s_box[BATCH_GT_MAX + RN_WAVE];   // (a wave reads 64 entries at a time, possibly past the last row)
    __shared__ float s_area[BATCH_GT_MAX + RN_WAVE];
    __shared__ int s_bad;

    const int tid = threadIdx.x, lane = tid & (RN_WAVE - 1);
    const int b0 = blockIdx.y * ipb, b1 = min(B, b0 + ipb);
    // the anchor load goes out first: its latency overlaps the GT staging below
    const int64_t a_idx = (int64_t)blockIdx.x * MATCH_BLOCK + tid;
    const bool live = a_idx < A;
    rn::f32x4 an = {0.f, 0.f, 0.f, 0.f};
    if (live) an = anchors[a_idx];
    // the group's offsets in one vector load (lane l holds gt_off[b0 + l], ipb <= 64 - 1 lanes; see iou_match_small_kernel)
    const int goff = gt_off[b0 + min(lane, b1 - b0)];
    const int g0 = __builtin_amdgcn_readlane(goff, 0);
    const int total = min(__builtin_amdgcn_readlane(goff, b1 - b0) - g0, BATCH_GT_MAX);      // (the host promised <= BATCH_GT_MAX)
    if (tid == 0) s_bad = 0;
    __syncthreads();
    bool ok = true;
    for (int j = tid; j < total; j += MATCH_BLOCK) {
        const rn::f32x4 g = gt[g0 + j];
        const float ar = (g.z - g.x) * (g.w - g.y);
        s_box[j] = g;
        s_area[j] = ar;
        ok = ok && gt_is_proper(g, ar);
    }
    if (!ok) s_bad = 1;                                    // benign race: every writer stores 1
    __syncthreads();

    const float area_a = (an.z - an.x) * (an.w - an.y);
    const bool fast = !s_bad && __all(anchor_is_proper(an, area_a));
    WaveBox bb = {0.f, 0.f, 0.f, 0.f};
    if (fast) bb = WaveBox{wave_min(an.x), wave_min(an.y), wave_max(an.z), wave_max(an.w)};

    for (int b = b0; b < b1; ++b) {
        // clamped against what was staged: an inconsistent gt_off (rows beyond the host-supplied total / beyond BATCH_GT_MAX)
        // then matches against a truncated GT set instead of indexing past the LDS arrays
        const int ob = __builtin_amdgcn_readlane(goff, b - b0), oe = __builtin_amdgcn_readlane(goff, b - b0 + 1);
        const int j0 = min(max(ob - g0, 0), total);
        const int T = min(max(oe - ob, 0), total - j0);
        float best = 0.0f;
        int bi = 0;
        Best bb2 = {0.0f, 0, false};
        for (int c = 0; c < T; c += RN_WAVE) {
            const rn::f32x4 mine = s_box[j0 + c + lane];    // lane l holds GT row c + l of this image
            const float mine_a = s_area[j0 + c + lane];
            const int m = min(RN_WAVE, T - c);
            if (fast) {
                // 64 boxes culled by one ballot against the wave's bounding box (see iou_match_small_kernel); survivors in ascending order
                unsigned long long ov = __ballot(lane < m && may_overlap(bb, mine));
                while (ov) {                                // wave-uniform
                    const int l = __builtin_ctzll(ov);
                    ov &= ov - 1ull;
                    const GtBox g = gt_of_lane(mine, mine_a, l);
                    const float inter = inter_fast(g, an);
                    if (__any(inter != 0.0f)) {
                        const float v = inter / ((g.area + area_a) - inter);
                        if (v > best) { best = v; bi = c + l; }
                    }
                }
            } else {
                for (int l = 0; l < m; ++l) {
                    const GtBox g = gt_of_lane(mine, mine_a, l);
                    careful_update(bb2, iou_pair(vec(g), g.area, an, area_a), c + l);
                }
            }
        }
        if (!fast) { best = bb2.v; bi = bb2.i; }
        const int64_t r = classify(best, bi, T, fg_thr, bg_thr);
        if (live && (!SPARSE || r != -1)) matches[(int64_t)b * A + a_idx] = r;
        if (special) special_word(special, b, A, a_idx, live && r != -1);
        if (num_fg) {
            const unsigned long long fg = __ballot(live && r >= 0);
            if ((tid & (RN_WAVE - 1)) == 0 && fg) atomicAdd(&num_fg[b], __popcll(fg));
        }
    }
}

// ============================================================================================================
// The train shape proper (round 5): shared anchors and at most 128 GT boxes in the WHOLE batch (B = 8 x T = 8 = 64).  No LDS and no
// workgroup barrier: lane j of every wave loads GT rows j and j + 64 of the batch itself (one coalesced 16-byte load each: the rows
// are L2 hits after the first wave), computes their areas, and the pair loop fetches a box with v_readlane exactly as the batch
// kernel does after its LDS staging.  One thread = one anchor x ALL images, so the grid is A / 256 workgroups (788 instead of the
// batch kernel's 1 576 at B = 8) and a wave's dependent chain is gt_off (scalar) -> GT rows -> pairs -> stores: the batch kernel's
// two __syncthreads and its LDS round trip are gone.  This kernel is launch-latency-bound whatever it does; what is
// left is the shortest chain that still produces the codes, the flag words and num_fg.
constexpr int SMALL_GT_MAX = 2 * RN_WAVE;
// bits [lo, hi) of a 64-bit word, 0 <= lo <= hi <= 64
__device__ __forceinline__ unsigned long long bits64(const int lo, const int hi)
{
    const unsigned long long upto_hi = hi >= 64 ? ~0ull : ((1ull << hi) - 1ull);
    const unsigned long long upto_lo = lo >= 64 ? ~0ull : ((1ull << lo) - 1ull);
    return upto_hi & ~upto_lo;
}
template <bool SPARSE>
__global__ __launch_bounds__(MATCH_BLOCK) void iou_match_small_kernel(
    const rn::f32x4 *__restrict__ anchors, const rn::f32x4 *__restrict__ gt, const int32_t *__restrict__ gt_off,
    const int B, const int64_t A, const float fg_thr, const float bg_thr, int64_t *__restrict__ matches,
    int32_t *__restrict__ num_fg, unsigned long long *__restrict__ special, const int n_plain, const int nsplit)
{
    const int tid = threadIdx.x, lane = tid & (RN_WAVE - 1);
    // Load balance.  A wave's time is its number of cull survivors: 4.7 of the batch's 64 boxes on P3, 39 on P6, 63 on P7 (big anchors
    // overlap everything) -- the eleven P7 waves alone ran a 9 us chain at the END of the launch.  So (1) the workgroups walk the anchor
    // list BACK TO FRONT (an FPN anchor list ends with its coarsest level), and (2) the last 1/16 of the anchor blocks is split over the
    // images as well: anchor block n_plain + t / nsplit, image t % nsplit -- `nsplit` short waves instead of one long one.  Both are
    // pure scheduling: every (anchor, image) pair is still computed by exactly one lane.
    const int blk = (int)gridDim.x - 1 - (int)blockIdx.x;
    int ablock = blk, b_lo = 0, b_hi = B;
    if (blk >= n_plain) {
        const int t = blk - n_plain;
        ablock = n_plain + t / nsplit;
        const int ipp = (B + nsplit - 1) / nsplit;
        b_lo = (t % nsplit) * ipp;
        b_hi = min(B, b_lo + ipp);
    }
    const int64_t a_idx = (int64_t)ablock * MATCH_BLOCK + tid;
    const bool live = a_idx < A;
    rn::f32x4 an = {0.f, 0.f, 0.f, 0.f};
    if (live) an = anchors[a_idx];
    // all B + 1 offsets in ONE vector load (lane l holds gt_off[l]; B <= 63): a scalar load of gt_off[b] inside the image loop costs a
    // memory round trip per image, and eight of those in a row are most of this kernel's time (round 5: 16.9 us for the batch kernel,
    // whose loop was 8 x (two dependent scalar loads -> LDS read -> ~300 ALU instructions))
    const int goff = gt_off[lane <= B ? lane : B];
    const int g0 = __builtin_amdgcn_readlane(goff, 0);
    const int total = min(max(__builtin_amdgcn_readlane(goff, B) - g0, 0), SMALL_GT_MAX);        // (the host promised <= SMALL_GT_MAX)
    rn::f32x4 m0 = {0.f, 0.f, 0.f, 0.f}, m1 = {0.f, 0.f, 0.f, 0.f};
    if (lane < total) m0 = gt[g0 + lane];
    if (lane + RN_WAVE < total) m1 = gt[g0 + RN_WAVE + lane];
    const float ar0 = (m0.z - m0.x) * (m0.w - m0.y), ar1 = (m1.z - m1.x) * (m1.w - m1.y);
    const bool gt_ok = (lane >= total || gt_is_proper(m0, ar0)) && (lane + RN_WAVE >= total || gt_is_proper(m1, ar1));
    const float area_a = (an.z - an.x) * (an.w - an.y);
    const bool fast = __all(gt_ok) && __all(anchor_is_proper(an, area_a));
    WaveBox bb = {0.f, 0.f, 0.f, 0.f};
    if (fast) bb = WaveBox{wave_min(an.x), wave_min(an.y), wave_max(an.z), wave_max(an.w)};

    // Fast path: ALL the batch's GT boxes are tested against the wave's bounding box at once -- the boxes are lane-resident, the
    // bounding box is wave-uniform: four compares and a ballot per register cull up to 128 boxes, where a per-box test (v_readlane x 5,
    // four compares, a branch) cost ~50 issue cycles for each of the 64 boxes of the train shape: 16 of this kernel's 23 us (round-5
    // ablation, profiles/r05_k2_ablation.txt: 23.1 us, 7.1 without the pair loop).  A box that misses the bounding box has inter == 0
    // with every anchor of the wave and cannot change (max, first arg-max); the survivors (~6 % at the train shape) are visited in
    // ascending order, so the first index still wins ties.
    unsigned long long ov0 = 0ull, ov1 = 0ull;
    if (fast) {
        ov0 = __ballot(lane < total && may_overlap(bb, m0));
        ov1 = __ballot(lane + RN_WAVE < total && may_overlap(bb, m1));
    }
    for (int b = b_lo; b < b_hi; ++b) {
        // clamped against what was loaded: an inconsistent gt_off matches against a truncated GT set instead of reading other rows
        const int ob = __builtin_amdgcn_readlane(goff, b), oe = __builtin_amdgcn_readlane(goff, b + 1);
        const int j0 = min(max(ob - g0, 0), total);
        const int T = min(max(oe - ob, 0), total - j0);
        float best = 0.0f;
        int bi = 0;
        Best bb2 = {0.0f, 0, false};
        if (fast) {
            // the image's rows [j0, j0 + T) of the two 64-bit survivor masks
            const int j1 = j0 + T;
            unsigned long long p0 = ov0 & bits64(min(j0, 64), min(j1, 64));
            unsigned long long p1 = ov1 & bits64(max(j0, 64) - 64, max(j1, 64) - 64);
            while (p0 | p1) {                                           // wave-uniform
                int j;
                if (p0) { j = __builtin_ctzll(p0); p0 &= p0 - 1ull; }
                else { j = 64 + __builtin_ctzll(p1); p1 &= p1 - 1ull; }
                const GtBox g = j < RN_WAVE ? gt_of_lane(m0, ar0, j) : gt_of_lane(m1, ar1, j - RN_WAVE);
                const float inter = inter_fast(g, an);
                if (__any(inter != 0.0f)) {
                    const float v = inter / ((g.area + area_a) - inter);
                    if (v > best) { best = v; bi = j - j0; }
                }
            }
        } else {
            for (int l = 0; l < T; ++l) {
                const int j = j0 + l;                                    // wave-uniform
                const GtBox g = j < RN_WAVE ? gt_of_lane(m0, ar0, j) : gt_of_lane(m1, ar1, j - RN_WAVE);
                careful_update(bb2, iou_pair(vec(g), g.area, an, area_a), l);
            }
        }
        if (!fast) { best = bb2.v; bi = bb2.i; }
        const int64_t r = classify(best, bi, T, fg_thr, bg_thr);
        if (live && (!SPARSE || r != -1)) matches[(int64_t)b * A + a_idx] = r;
        if (special) special_word(special, b, A, a_idx, live && r != -1);
        if (num_fg) {
            const unsigned long long fg = __ballot(live && r >= 0);
            if (lane == 0 && fg) atomicAdd(&num_fg[b], __popcll(fg));
        }
    }
}

// ============================================================================================================
// General shape: R anchors per thread, GT tiles in LDS.
template <int R>
__global__ __launch_bounds__(MATCH_BLOCK) void iou_match_tile_kernel(
    const rn::f32x4 *__restrict__ anchors, const int64_t anchor_bstride4,
    const rn::f32x4 *__restrict__ gt, const int32_t *__restrict__ gt_off, const int64_t A,
    const float fg_thr, const float bg_thr, int64_t *__restrict__ matches, int32_t *__restrict__ num_fg,
    unsigned long long *__restrict__ special)
{
    __shared__ rn::f32x4 s_box[GT_TILE];
    __shared__ float s_area[GT_TILE];
    __shared__ int s_bad[2];

    const int tid = threadIdx.x, lane = tid & (RN_WAVE - 1);
    const int b = blockIdx.y;
    const int t0 = gt_off[b];
    const int T = gt_off[b + 1] - t0;
    // a wave owns R * 64 CONSECUTIVE anchors (anchor r of a lane = a0 + 64 r): one contiguous strip of the feature map,
    // so the "does any lane overlap this GT box" branch below is taken as rarely as possible
    const int64_t a0 = (int64_t)blockIdx.x * (MATCH_BLOCK * R) + (tid >> 6) * (RN_WAVE * R) + (tid & (RN_WAVE - 1));

    rn::f32x4 an[R];
    float area_a[R];
    bool a_ok = true;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int64_t a_idx = a0 + (int64_t)r * RN_WAVE;
        an[r] = rn::f32x4{0.f, 0.f, 0.f, 0.f};
        if (a_idx < A) an[r] = anchors[(int64_t)b * anchor_bstride4 + a_idx];
        area_a[r] = (an[r].z - an[r].x) * (an[r].w - an[r].y);
        a_ok = a_ok && anchor_is_proper(an[r], area_a[r]);
    }
    const bool wave_ok = __all(a_ok);
    WaveBox bb = {0.f, 0.f, 0.f, 0.f};
    if (wave_ok) {
        float x0 = an[0].x, y0 = an[0].y, x1 = an[0].z, y1 = an[0].w;
#pragma unroll
        for (int r = 1; r < R; ++r) { x0 = fminf(x0, an[r].x); y0 = fminf(y0, an[r].y); x1 = fmaxf(x1, an[r].z); y1 = fmaxf(y1, an[r].w); }
        bb = WaveBox{wave_min(x0), wave_min(y0), wave_max(x1), wave_max(y1)};
    }

    Best best[R];
#pragma unroll
    for (int r = 0; r < R; ++r) best[r] = Best{0.0f, 0, false};

    if (tid < 2) s_bad[tid] = 0;
    for (int base = 0, it = 0; base < T; base += GT_TILE, ++it) {
        const int n = min(GT_TILE, T - base);
        __syncthreads();                                   // previous tile fully consumed; s_bad[it & 1] reset below is ordered
        if (tid == 0) s_bad[(it + 1) & 1] = 0;
        if (tid < n) {
            const rn::f32x4 g = gt[t0 + base + tid];
            const float ar = (g.z - g.x) * (g.w - g.y);
            s_box[tid] = g;
            s_area[tid] = ar;
            if (!gt_is_proper(g, ar)) s_bad[it & 1] = 1;
        }
        __syncthreads();
        if (wave_ok && !s_bad[it & 1]) {
            // once a tile has been processed here, `have` only means "best/bi hold torch's running result so far";
            // with proper boxes every quotient is >= +0, so starting from (0, index 0) and updating on strict > is exact
            // (a negative running maximum can only come from an earlier careful tile: this tile's first pair beats it)
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if (!best[r].have) best[r] = Best{0.0f, 0, true};
                else if (best[r].v < 0.0f) best[r] = Best{0.0f, base, true};
            }
            for (int c = 0; c < n; c += RN_WAVE) {
                const rn::f32x4 mine = s_box[c + lane];     // lane l holds GT row base + c + l
                const float mine_a = s_area[c + lane];
                const int m = min(RN_WAVE, n - c);
                // 64 GT boxes are culled against the wave's strip at once (each lane tests its own box); only the survivors --
                // ~10 % at the headline shapes -- are broadcast and paired
                unsigned long long todo = __ballot(lane < m && may_overlap(bb, mine));
                while (todo) {
                    const int l = __ffsll((long long)todo) - 1;
                    todo &= todo - 1;
                    const GtBox g = gt_of_lane(mine, mine_a, l);
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const float inter = inter_fast(g, an[r]);
                        if (__any(inter != 0.0f)) {
                            // a NaN best (from an earlier careful tile) stays: (v > NaN) is false
                            const float v = inter / ((g.area + area_a[r]) - inter);
                            if (v > best[r].v) { best[r].v = v; best[r].i = base + c + l; }
                        }
                    }
                }
            }
        } else {
            for (int c = 0; c < n; c += RN_WAVE) {
                const rn::f32x4 mine = s_box[c + lane];
                const float mine_a = s_area[c + lane];
                const int m = min(RN_WAVE, n - c);
                for (int l = 0; l < m; ++l) {
                    const GtBox g = gt_of_lane(mine, mine_a, l);
#pragma unroll
                    for (int r = 0; r < R; ++r) careful_update(best[r], iou_pair(vec(g), g.area, an[r], area_a[r]), base + c + l);
                }
            }
        }
    }

    int nfg = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int64_t a_idx = a0 + (int64_t)r * RN_WAVE;
        const int64_t m = classify(best[r].v, best[r].i, T, fg_thr, bg_thr);
        if (a_idx < A) {
            matches[(int64_t)b * A + a_idx] = m;
            nfg += m >= 0 ? 1 : 0;
        }
        if (special) special_word(special, b, A, a_idx, a_idx < A && m != -1);
    }
    if (num_fg) {                                          // one global atomic per workgroup (same-address atomics serialise)
        nfg = rn::wave_sum_i(nfg);
        __syncthreads();                                    // every wave is past its last read of s_bad
        if (tid == 0) s_bad[0] = 0;
        __syncthreads();
        if ((tid & (RN_WAVE - 1)) == 0 && nfg) atomicAdd(&s_bad[0], nfg);
        __syncthreads();
        if (tid == 0 && s_bad[0]) atomicAdd(&num_fg[b], s_bad[0]);
    }
}

// ============================================================================================================
// Large GT sets (hundreds of boxes per image): the pair work of an anchor strip is proportional to the number of GT boxes
// it overlaps, so the waves that own the big anchors of P5-P7 (which overlap nearly everything) run 10-50x longer than the
// waves of P3 -- with one wave per anchor strip the launch lasted as long as the P6/P7 strips alone (200 of 300 us at
// T = 500).  Here the GT axis is split as well: grid.z workgroups share an anchor strip, each walks every grid.z-th
// tile of CHUNK_TILE GT boxes, and the per-anchor results meet in `matches` through a 64-bit atomic max on
//     key = ordered(IoU) << 32 | ~index        (ordered(): order-preserving map of the float, NaN on top, -0 = +0)
// whose maximum is torch's (max IoU, first index; the first NaN wins).  A finalize kernel turns the keys into match
// codes.  Workgroups that took the fast loop only (proper boxes: every IoU >= +0) skip the atomic when their maximum is 0;
// the finalize kernel recomputes the rare anchor for which that shortcut could matter (a negative maximum, i.e. inverted
// boxes, or a negative foreground threshold) from scratch.  The strips are handed out from the END of the anchor array
// (the big anchors) so the long-running workgroups start first.
constexpr int CHUNK_TILE = 128;

__device__ __forceinline__ uint32_t ordered_bits(const float v)
{
    if (v != v) return 0xffffffffu;
    const uint32_t u = __float_as_uint(v == 0.0f ? 0.0f : v);
    return (u >> 31) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ordered_value(const uint32_t k)
{
    if (k == 0xffffffffu) return __builtin_nanf("");
    return __uint_as_float((k >> 31) ? (k & 0x7fffffffu) : ~k);
}

template <int R>
__global__ __launch_bounds__(MATCH_BLOCK) void iou_match_chunk_kernel(
    const rn::f32x4 *__restrict__ anchors, const int64_t anchor_bstride4,
    const rn::f32x4 *__restrict__ gt, const int32_t *__restrict__ gt_off, const int64_t A,
    unsigned long long *__restrict__ keys)
{
    __shared__ rn::f32x4 s_box[CHUNK_TILE];
    __shared__ float s_area[CHUNK_TILE];
    __shared__ int s_bad[2];

    const int tid = threadIdx.x, lane = tid & (RN_WAVE - 1);
    const int b = blockIdx.y;
    const int t0 = gt_off[b];
    const int T = gt_off[b + 1] - t0;
    const int64_t strip = (int64_t)(gridDim.x - 1 - blockIdx.x);          // big anchors (end of the array) first
    const int64_t a0 = strip * (MATCH_BLOCK * R) + (tid >> 6) * (RN_WAVE * R) + lane;

    rn::f32x4 an[R];
    float area_a[R];
    bool a_ok = true;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int64_t a_idx = a0 + (int64_t)r * RN_WAVE;
        an[r] = rn::f32x4{0.f, 0.f, 0.f, 0.f};
        if (a_idx < A) an[r] = anchors[(int64_t)b * anchor_bstride4 + a_idx];
        area_a[r] = (an[r].z - an[r].x) * (an[r].w - an[r].y);
        a_ok = a_ok && anchor_is_proper(an[r], area_a[r]);
    }
    const bool wave_ok = __all(a_ok);
    WaveBox bb = {0.f, 0.f, 0.f, 0.f};
    if (wave_ok) {
        float x0 = an[0].x, y0 = an[0].y, x1 = an[0].z, y1 = an[0].w;
#pragma unroll
        for (int r = 1; r < R; ++r) { x0 = fminf(x0, an[r].x); y0 = fminf(y0, an[r].y); x1 = fmaxf(x1, an[r].z); y1 = fmaxf(y1, an[r].w); }
        bb = WaveBox{wave_min(x0), wave_min(y0), wave_max(x1), wave_max(y1)};
    }

    Best best[R];
#pragma unroll
    for (int r = 0; r < R; ++r) best[r] = Best{0.0f, 0, false};
    bool all_fast = true;                                   // wave-uniform: every tile so far went through the fast loop

    if (tid < 2) s_bad[tid] = 0;
    for (int base = (int)blockIdx.z * CHUNK_TILE, it = 0; base < T; base += (int)gridDim.z * CHUNK_TILE, ++it) {
        const int n = min(CHUNK_TILE, T - base);
        __syncthreads();
        if (tid == 0) s_bad[(it + 1) & 1] = 0;
        if (tid < n) {
            const rn::f32x4 g = gt[t0 + base + tid];
            const float ar = (g.z - g.x) * (g.w - g.y);
            s_box[tid] = g;
            s_area[tid] = ar;
            if (!gt_is_proper(g, ar)) s_bad[it & 1] = 1;
        }
        __syncthreads();
        if (wave_ok && !s_bad[it & 1]) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if (!best[r].have) best[r] = Best{0.0f, base, true};
                else if (best[r].v < 0.0f) best[r] = Best{0.0f, base, true};
            }
            for (int c = 0; c < n; c += RN_WAVE) {
                const rn::f32x4 mine = s_box[min(c + lane, CHUNK_TILE - 1)];
                const float mine_a = s_area[min(c + lane, CHUNK_TILE - 1)];
                const int m = min(RN_WAVE, n - c);
                // 64 GT boxes are culled against the wave's strip at once (each lane tests its own box); only the survivors
                // are broadcast and paired
                unsigned long long todo = __ballot(lane < m && may_overlap(bb, mine));
                while (todo) {
                    const int l = __ffsll((long long)todo) - 1;
                    todo &= todo - 1;
                    const GtBox g = gt_of_lane(mine, mine_a, l);
                    float inter[R];
                    bool any = false;
#pragma unroll
                    for (int r = 0; r < R; ++r) { inter[r] = inter_fast(g, an[r]); any = any || inter[r] != 0.0f; }
                    if (__any(any)) {
                        // all R quotients in one basic block: the R independent divide chains overlap (the strips that come
                        // here at all are the big anchors, for which most of the R groups overlap the box anyway)
#pragma unroll
                        for (int r = 0; r < R; ++r) {
                            const float v = inter[r] / ((g.area + area_a[r]) - inter[r]);
                            if (v > best[r].v) { best[r].v = v; best[r].i = base + c + l; }
                        }
                    }
                }
            }
        } else {
            all_fast = false;
            for (int c = 0; c < n; c += RN_WAVE) {
                const rn::f32x4 mine = s_box[min(c + lane, CHUNK_TILE - 1)];
                const float mine_a = s_area[min(c + lane, CHUNK_TILE - 1)];
                const int m = min(RN_WAVE, n - c);
                for (int l = 0; l < m; ++l) {
                    const GtBox g = gt_of_lane(mine, mine_a, l);
#pragma unroll
                    for (int r = 0; r < R; ++r) careful_update(best[r], iou_pair(vec(g), g.area, an[r], area_a[r]), base + c + l);
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int64_t a_idx = a0 + (int64_t)r * RN_WAVE;
        if (a_idx >= A || !best[r].have) continue;
        if (all_fast && best[r].v == 0.0f) continue;       // (0, first index of the chunk): cannot win against any other contribution
        const unsigned long long key = ((unsigned long long)ordered_bits(best[r].v) << 32) | (uint32_t)~(uint32_t)best[r].i;
        atomicMax(&keys[(int64_t)b * A + a_idx], key);
    }
}

// keys -> match codes (+ num_fg).  No key: every IoU was +0 -> (0, index 0).
constexpr int FIN_BLOCK = 1024;
__global__ __launch_bounds__(FIN_BLOCK) void iou_match_finalize_kernel(
    const rn::f32x4 *__restrict__ anchors, const int64_t anchor_bstride4, const rn::f32x4 *__restrict__ gt,
    const int32_t *__restrict__ gt_off, const int64_t A, const float fg_thr, const float bg_thr,
    int64_t *__restrict__ matches, int32_t *__restrict__ num_fg, unsigned long long *__restrict__ special)
{
    const int b = blockIdx.y;
    const int t0 = gt_off[b], T = gt_off[b + 1] - t0;
    __shared__ int s_fg;
    if (threadIdx.x == 0) s_fg = 0;
    __syncthreads();
    const int64_t a_idx = (int64_t)blockIdx.x * FIN_BLOCK + threadIdx.x;
    const bool live = a_idx < A;
    int64_t r = -2;
    if (live) {
        const unsigned long long key = (unsigned long long)matches[(int64_t)b * A + a_idx];
        float best = 0.0f;
        int bi = 0;
        if (key) { best = ordered_value((uint32_t)(key >> 32)); bi = (int)~(uint32_t)key; }
        if (T > 0 && (best < 0.0f || (best == 0.0f && fg_thr < 0.0f))) {
            // the zero-skipping of the chunk kernel is not exact here (inverted boxes / a negative threshold): redo this anchor
            const rn::f32x4 an = anchors[(int64_t)b * anchor_bstride4 + a_idx];
            const float area_a = (an.z - an.x) * (an.w - an.y);
            Best bb = {0.0f, 0, false};
            for (int j = 0; j < T; ++j) {
                const rn::f32x4 g = gt[t0 + j];
                const float ga = (g.z - g.x) * (g.w - g.y);
                // (not the wave-uniform shortcut of iou_pair: lanes are at different j here)
                const float ltx = g.x > an.x ? g.x : an.x, lty = g.y > an.y ? g.y : an.y;
                const float rbx = g.z < an.z ? g.z : an.z, rby = g.w < an.w ? g.w : an.w;
                float w = rbx - ltx; if (!(w > 0.0f)) w = (w != w) ? w : 0.0f;
                float h = rby - lty; if (!(h > 0.0f)) h = (h != h) ? h : 0.0f;
                const float inter = w * h;
                careful_update(bb, inter / ((ga + area_a) - inter), j);
            }
            best = bb.v; bi = bb.i;
        }
        r = classify(best, bi, T, fg_thr, bg_thr);
        matches[(int64_t)b * A + a_idx] = r;
    }
    if (special) special_word(special, b, A, a_idx, live && r != -1);
    if (num_fg) {
        // thousands of foreground anchors per image here: one global atomic per 1024 anchors (same-address atomics from every
        // wave serialise at ~50 per microsecond and address: 35 us of a 69 us kernel before)
        const unsigned long long fg = __ballot(live && r >= 0);
        if ((threadIdx.x & (RN_WAVE - 1)) == 0 && fg) atomicAdd(&s_fg, __popcll(fg));
        __syncthreads();
        if (threadIdx.x == 0 && s_fg) atomicAdd(&num_fg[b], s_fg);
    }
}

// num_fg[0..B) = 0 as a KERNEL.  A hipMemsetAsync becomes a MEMSET NODE when the step is captured in a hipGraph, and on ROCm 7.0 the
// memset nodes of a replayed graph write garbage once the process has synchronised with the device and enqueued other work (round 4:
// every replay after the first torch.cuda.synchronize() scaled both losses by 1 / garbage -- num_fg; the same graph with this kernel
// is exact).  The library issues no hipMemsetAsync at all.
__global__ void zero_i32_kernel(int32_t *__restrict__ p, const int n)
{
    for (int i = threadIdx.x; i < n; i += blockDim.x) p[i] = 0;
}
__global__ __launch_bounds__(256) void zero_u64_kernel(unsigned long long *__restrict__ p, const int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) p[i] = 0ull;
}

}  // namespace

RN_API int rn_iou_match(const float *anchors, int64_t anchor_bstride, const float *gt_boxes, const int32_t *gt_off,
                        int B, int64_t A, float fg_thr, float bg_thr, int64_t *matches, int32_t *num_fg, void *stream)
{
    return rn_iou_match_ex(anchors, anchor_bstride, gt_boxes, gt_off, B, A, fg_thr, bg_thr, matches, num_fg, -1, stream);
}

RN_API int rn_iou_match_ex(const float *anchors, int64_t anchor_bstride, const float *gt_boxes, const int32_t *gt_off,
                           int B, int64_t A, float fg_thr, float bg_thr, int64_t *matches, int32_t *num_fg,
                           int64_t total_gt, void *stream)
{
    return rn_iou_match_special(anchors, anchor_bstride, gt_boxes, gt_off, B, A, fg_thr, bg_thr, matches, num_fg, nullptr, total_gt, stream);
}

RN_API size_t rn_iou_match_special_bytes(int B, int64_t A) { return (B > 0 && A > 0) ? sizeof(uint64_t) * (size_t)B * (size_t)((A + 63) >> 6) : 0; }

RN_API int rn_iou_match_special(const float *anchors, int64_t anchor_bstride, const float *gt_boxes, const int32_t *gt_off,
                                int B, int64_t A, float fg_thr, float bg_thr, int64_t *matches, int32_t *num_fg,
                                uint64_t *special_rows, int64_t total_gt, void *stream)
{
    return rn_iou_match_special_ex(anchors, anchor_bstride, gt_boxes, gt_off, B, A, fg_thr, bg_thr, matches, num_fg, special_rows, total_gt, 0, stream);
}

RN_API int rn_iou_match_special_ex(const float *anchors, int64_t anchor_bstride, const float *gt_boxes, const int32_t *gt_off,
                                   int B, int64_t A, float fg_thr, float bg_thr, int64_t *matches, int32_t *num_fg,
                                   uint64_t *special_rows, int64_t total_gt, int flags, void *stream)
{
    if (flags & ~(RN_MATCH_NUM_FG_ZEROED | RN_MATCH_FLAGGED_ONLY)) return RN_EINVAL;
    if ((flags & RN_MATCH_FLAGGED_ONLY) && !special_rows) return RN_EINVAL;       // (without the words nobody can tell which rows were written)
    const bool sparse = (flags & RN_MATCH_FLAGGED_ONLY) != 0;
    unsigned long long *special = (unsigned long long *)special_rows;
    if (special && !rn::aligned(special, 8)) return RN_EALIGN;
    if (!anchors || !gt_off || !matches || B <= 0 || A <= 0 || B > 65535) return RN_EINVAL;
    if (!(fg_thr > bg_thr)) return RN_ETHRESH;
    if (!rn::aligned(anchors, 16) || (gt_boxes && !rn::aligned(gt_boxes, 16)) || (anchor_bstride & 3)) return RN_EALIGN;
    hipStream_t st = (hipStream_t)stream;
    if (num_fg && !(flags & RN_MATCH_NUM_FG_ZEROED)) {                 // (a kernel, never hipMemsetAsync: DESIGN.md, memset nodes)
        hipLaunchKernelGGL(zero_i32_kernel, dim3(1), dim3(64), 0, st, num_fg, B);
        RN_LAUNCH_CHECK();
    }
    // The batch kernel needs host knowledge of sum(T) (gt_off lives on the device and this call never syncs):
    // callers that know it pass total_gt >= 0; -1 means unknown -> the general kernel.
    if (anchor_bstride == 0 && B <= 63 && total_gt >= 0 && total_gt <= SMALL_GT_MAX && A >= 32 * MATCH_BLOCK) {
        // a few GT boxes in the whole batch and enough anchors to fill the chip with one thread per anchor: no LDS, no barrier
        const int nblk = (int)((A + MATCH_BLOCK - 1) / MATCH_BLOCK);
        const int tail = nblk / 16 > 0 ? nblk / 16 : 1;                 // anchor blocks that are split over the images too (see the kernel)
        const int nsplit = B < 8 ? B : 8;
        const int n_plain = nblk - tail;
        const dim3 grid((unsigned)(n_plain + tail * nsplit));
        if (sparse) hipLaunchKernelGGL(iou_match_small_kernel<true>, grid, dim3(MATCH_BLOCK), 0, st, (const rn::f32x4 *)anchors,
                                       (const rn::f32x4 *)gt_boxes, gt_off, B, A, fg_thr, bg_thr, matches, num_fg, special, n_plain, nsplit);
        else hipLaunchKernelGGL(iou_match_small_kernel<false>, grid, dim3(MATCH_BLOCK), 0, st, (const rn::f32x4 *)anchors,
                                (const rn::f32x4 *)gt_boxes, gt_off, B, A, fg_thr, bg_thr, matches, num_fg, special, n_plain, nsplit);
    } else if (anchor_bstride == 0 && B <= 64 && total_gt >= 0 && total_gt <= BATCH_GT_MAX && total_gt <= 32 * (int64_t)B) {
        // images per workgroup: all of them when the anchors alone give >= 1024 workgroups, else split the batch
        const int64_t bx = (A + MATCH_BLOCK - 1) / MATCH_BLOCK;
        int by = (int)((1024 + bx - 1) / bx);
        by = by < 1 ? 1 : (by > B ? B : by);
        if ((B + by - 1) / by > RN_WAVE - 1) by = 2;                    // (a workgroup's offsets live one per lane: at most 63 images + 1)
        const int ipb = (B + by - 1) / by;
        const dim3 grid((unsigned)bx, (unsigned)((B + ipb - 1) / ipb));
        if (sparse) hipLaunchKernelGGL(iou_match_batch_kernel<true>, grid, dim3(MATCH_BLOCK), 0, st, (const rn::f32x4 *)anchors,
                                       (const rn::f32x4 *)gt_boxes, gt_off, B, ipb, A, fg_thr, bg_thr, matches, num_fg, special);
        else hipLaunchKernelGGL(iou_match_batch_kernel<false>, grid, dim3(MATCH_BLOCK), 0, st, (const rn::f32x4 *)anchors,
                                (const rn::f32x4 *)gt_boxes, gt_off, B, ipb, A, fg_thr, bg_thr, matches, num_fg, special);
    } else if (total_gt > 192 * (int64_t)B && A * (int64_t)B < ((int64_t)1 << 40)) {
        // hundreds of GT boxes per image: split the GT axis too (see iou_match_chunk_kernel); z workgroups per anchor strip
        int z = (int)((total_gt / B + CHUNK_TILE - 1) / CHUNK_TILE);
        z = z < 2 ? 2 : (z > 16 ? 16 : z);
        {   // (a kernel, not hipMemsetAsync: see zero_i32_kernel)
            const int64_t n = (int64_t)B * A;
            int64_t blocks = (n + 255) / 256;
            if (blocks > 4096) blocks = 4096;
            hipLaunchKernelGGL(zero_u64_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (unsigned long long *)matches, n);
            RN_LAUNCH_CHECK();
        }
        const dim3 grid((unsigned)((A + MATCH_BLOCK * 4 - 1) / (MATCH_BLOCK * 4)), (unsigned)B, (unsigned)z);
        hipLaunchKernelGGL(iou_match_chunk_kernel<4>, grid, dim3(MATCH_BLOCK), 0, st, (const rn::f32x4 *)anchors, anchor_bstride / 4,
                           (const rn::f32x4 *)gt_boxes, gt_off, A, (unsigned long long *)matches);
        RN_LAUNCH_CHECK();
        const dim3 fgrid((unsigned)((A + FIN_BLOCK - 1) / FIN_BLOCK), (unsigned)B);
        hipLaunchKernelGGL(iou_match_finalize_kernel, fgrid, dim3(FIN_BLOCK), 0, st, (const rn::f32x4 *)anchors, anchor_bstride / 4,
                           (const rn::f32x4 *)gt_boxes, gt_off, A, fg_thr, bg_thr, matches, num_fg, special);
    } else if (total_gt >= 0 && total_gt <= 32 * (int64_t)B) {
        const dim3 grid((unsigned)((A + MATCH_BLOCK * 2 - 1) / (MATCH_BLOCK * 2)), (unsigned)B);
        hipLaunchKernelGGL(iou_match_tile_kernel<2>, grid, dim3(MATCH_BLOCK), 0, st, (const rn::f32x4 *)anchors, anchor_bstride / 4,
                           (const rn::f32x4 *)gt_boxes, gt_off, A, fg_thr, bg_thr, matches, num_fg, special);
    } else {
        const dim3 grid((unsigned)((A + MATCH_BLOCK * 4 - 1) / (MATCH_BLOCK * 4)), (unsigned)B);
        hipLaunchKernelGGL(iou_match_tile_kernel<4>, grid, dim3(MATCH_BLOCK), 0, st, (const rn::f32x4 *)anchors, anchor_bstride / 4,
                           (const rn::f32x4 *)gt_boxes, gt_off, A, fg_thr, bg_thr, matches, num_fg, special);
    }
    RN_LAUNCH_CHECK();
    return RN_OK;
}
