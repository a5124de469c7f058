// K6 segmented sort + greedy NMS -- replaces torchvision.ops.nms as the reference
// calls it once per (image, class) at retinanet/models.py:210.  One workgroup per
// segment, all B*K segments in flight at once (the reference issues them one by
// one, each with host syncs).
//
// Semantics (bit-exact keep indices given identical boxes/scores): stable sort by
// score descending (ties: lower input index first), greedy scan, box j dropped when
//   inter / ((area_i + area_j) - inter) > thr,  w = max(0, xx2 - xx1), h likewise,
// fp32, IEEE divide, no FMA contraction (file compiled with -ffp-contract=off).
//
// Two kernels, both launched over every segment (a block exits at once when the segment is not in its class):
//   n <= 256   nms_mask_kernel: rank sort, pairwise suppression matrix as 64-bit ballot words in
//              LDS, greedy scan on SGPRs (rows fetched with v_readlane) -- no serial loop over boxes.
//   larger     nms_large_kernel, one launch for every longer segment:
//                n <= 2048: bitonic sort of the 64-bit keys in LDS (keys are unique, so "stable" = key
//                           order), boxes gathered once into LDS, greedy loop with one workgroup barrier
//                           per KEPT box;
//                beyond:    LDS-chunk sort + in-HBM merge passes, HBM-resident boxes / flags (correct for
//                           any length; slower).
// nms_prep_kernel builds the 64-bit keys for the op-level entry point rn_nms_segments.
#include "rn_internal.hpp"

__device__ unsigned long long g_nms_stamps[8192 * 8];
#define STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 8192) g_nms_stamps[blockIdx.x * 8 + (i)] = (unsigned long long)clock64(); } while (0)

namespace {

using rn::f32x4;

constexpr int MED_CAP = 2048;      // sorted and suppressed in LDS (nms_lds_body)
constexpr int BIG_THREADS = 1024;
constexpr int MASK_CAP = 256;      // boxes per suppression-matrix tile
constexpr int MASK_WORDS = MASK_CAP / 64;

__device__ __forceinline__ bool overlaps(const f32x4 bi, const float ai, const f32x4 bj, const float aj, const float thr)
{
    const float xx1 = bi.x > bj.x ? bi.x : bj.x;
    const float yy1 = bi.y > bj.y ? bi.y : bj.y;
    const float xx2 = bi.z < bj.z ? bi.z : bj.z;
    const float yy2 = bi.w < bj.w ? bi.w : bj.w;
    float w = xx2 - xx1; w = w > 0.0f ? w : 0.0f;
    float h = yy2 - yy1; h = h > 0.0f ? h : 0.0f;
    const float inter = w * h;
    const float ovr = inter / ((ai + aj) - inter);
    return ovr > thr;
}

// The same predicate for a whole wave, without the divide where it cannot matter: with t = thr * union (> 0, finite),
// |inter - t| > 2^-21 t decides it -- the rounding of t and of the quotient are 2^-24 relative each -- and only a wave in
// which some lane is closer than that (or has a non-positive / non-finite union) evaluates the exact quotient.
__device__ __forceinline__ bool overlaps_wave(const f32x4 bi, const float ai, const f32x4 bj, const float aj, const float thr, const bool live)
{
    // inter as in `overlaps` (v_max / v_min give std::max / std::min's values for non-NaN boxes; NaN boxes make `clear`
    // false below through a NaN t, and the exact path then evaluates the reference expression)
    float inter, t0, t1;
    asm("v_min_f32 %0, %5, %9\n\t"
        "v_max_f32 %1, %3, %7\n\t"
        "v_sub_f32 %0, %0, %1\n\t"
        "v_min_f32 %1, %6, %10\n\t"
        "v_max_f32 %2, %4, %8\n\t"
        "v_sub_f32 %1, %1, %2\n\t"
        "v_max_f32 %0, 0, %0\n\t"
        "v_max_f32 %1, 0, %1\n\t"
        "v_mul_f32 %0, %0, %1"
        : "=&v"(inter), "=&v"(t0), "=&v"(t1)
        : "v"(bi.x), "v"(bi.y), "v"(bi.z), "v"(bi.w), "v"(bj.x), "v"(bj.y), "v"(bj.z), "v"(bj.w));
    const float uni = (ai + aj) - inter;
    const float t = thr * uni;
    const float d = inter - t;
    const bool clear = t > 0.0f && fabsf(d) > t * 4.76837158e-7f;      // (an infinite or NaN t is never clear)
    if (__any(live && !clear)) return overlaps(bi, ai, bj, aj, thr);
    return d > 0.0f;
}

template <int THREADS>
__device__ __forceinline__ void bitonic_sort_lds(uint64_t *keys, const int n_pad)
{
    for (int k = 2; k <= n_pad; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < n_pad; i += THREADS) {
                const int p = i ^ j;
                if (p > i) {
                    const uint64_t x = keys[i], y = keys[p];
                    const bool up = (i & k) == 0;
                    if ((x > y) == up) { keys[i] = y; keys[p] = x; }
                }
            }
            __syncthreads();
        }
    }
}

// Exclusive scan of per-thread counts through LDS; returns this thread's offset and the total.
template <int THREADS>
__device__ __forceinline__ int block_excl_scan(int *s_scan, const int mine, int &total)
{
    s_scan[threadIdx.x] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int t = 0; t < THREADS; ++t) { const int c = s_scan[t]; s_scan[t] = run; run += c; }
        s_scan[THREADS] = run;
    }
    __syncthreads();
    total = s_scan[THREADS];
    return s_scan[threadIdx.x];
}

// Steps 2 and 3 of the blocked greedy scan (nms_lds_body, nms_big_body) on one tile of m <= MASK_CAP sorted boxes in LDS.
// tile_matrix: bit j of mask[i][w] (j = 64 w + bit) <=> j > i and IoU(i, j) > thr; a wave takes every NW-th row, its lanes keep column
// j = 64 w + lane of every word in registers, the row's box is one LDS broadcast, lane w collects word w (as nms_mask_kernel).
template <int NW>
__device__ __forceinline__ void tile_matrix(const f32x4 *s_box, const float *s_area, const int m, uint64_t (*s_mask)[MASK_WORDS], const float thr)
{
    const int lane = threadIdx.x & (RN_WAVE - 1), wave = threadIdx.x / RN_WAVE;
    const int nw = (m + 63) >> 6;
    f32x4 bj[MASK_WORDS];
    float aj[MASK_WORDS];
#pragma unroll
    for (int w = 0; w < MASK_WORDS; ++w) {
        const int j = min(w * 64 + lane, m - 1);
        bj[w] = s_box[j]; aj[w] = s_area[j];
    }
    for (int i = wave; i < m; i += NW) {
        const f32x4 bi = s_box[i];
        const float ai = s_area[i];
        const int w0 = (i + 1) >> 6;
        uint64_t mine = 0ull;
#pragma unroll
        for (int w = 0; w < MASK_WORDS; ++w) {
            if (w >= w0 && w < nw) {                                      // wave-uniform
                const int j = w * 64 + lane;
                const bool live = j > i && j < m;
                const bool sup = overlaps_wave(bi, ai, bj[w], aj[w], thr, live) && live;
                const unsigned long long mm = __ballot(sup);
                if (lane == w) mine = mm;
            }
        }
        if (lane < MASK_WORDS) s_mask[i][lane] = mine;
    }
}

// tile_scan (wave 0; call after a barrier): the greedy scan over the tile's matrix with the "removed" set in SGPRs, starting from
// the boxes already removed (supp[i] != 0: suppressed by a box kept in an earlier tile), visiting only rows that suppress something;
// keepw[w] = the tile's kept boxes as bits.
__device__ __forceinline__ void tile_scan(uint64_t (*s_mask)[MASK_WORDS], const uint8_t *s_supp, const int m, uint64_t *s_keepw)
{
    const int lane = threadIdx.x & (RN_WAVE - 1), wave = threadIdx.x / RN_WAVE;
    if (wave != 0) return;
    const int nw = (m + 63) >> 6;
    uint64_t row[MASK_WORDS][MASK_WORDS];
#pragma unroll
    for (int wb = 0; wb < MASK_WORDS; ++wb)
#pragma unroll
        for (int ww = 0; ww < MASK_WORDS; ++ww) row[wb][ww] = (wb * 64 + lane < m && ww < nw) ? s_mask[wb * 64 + lane][ww] : 0ull;
    uint64_t rem[MASK_WORDS];
#pragma unroll
    for (int w = 0; w < MASK_WORDS; ++w) rem[w] = __ballot(w * 64 + lane < m && s_supp[min(w * 64 + lane, m - 1)] != 0);
#pragma unroll
    for (int w = 0; w < MASK_WORDS; ++w) {
        if (w * 64 < m) {
            uint64_t any = 0;
#pragma unroll
            for (int ww = 0; ww < MASK_WORDS; ++ww) if (ww >= w) any |= row[w][ww];
            uint64_t todo = __ballot(any != 0ull);
            while (true) {
                const uint64_t live = todo & ~rem[w];
                if (!live) break;
                const int bit = __builtin_ctzll(live);
#pragma unroll
                for (int ww = 0; ww < MASK_WORDS; ++ww)
                    if (ww >= w) {
                        const unsigned lo32 = __builtin_amdgcn_readlane((unsigned)row[w][ww], bit);
                        const unsigned hi32 = __builtin_amdgcn_readlane((unsigned)(row[w][ww] >> 32), bit);
                        rem[ww] |= ((uint64_t)hi32 << 32) | lo32;
                    }
                todo &= ~((2ull << bit) - 1ull);
            }
        }
    }
    if (lane < MASK_WORDS) {
        uint64_t r = 0;
#pragma unroll
        for (int w = 0; w < MASK_WORDS; ++w) if (lane == w) r = rem[w];
        const int lo64 = lane * 64;
        const uint64_t valid = (m - lo64 >= 64) ? ~0ull : ((m - lo64 > 0) ? ((1ull << (m - lo64)) - 1ull) : 0ull);
        s_keepw[lane] = ~r & valid;
    }
}

// LDS image of nms_lds_body: box f32x4 | key u64 | area f32 | supp u8 (29 bytes per entry), scan i32[THREADS + 1], then (8-byte aligned) the
// tile's suppression matrix u64[MASK_CAP][MASK_WORDS], the kept list u16[CAP] and the tile's keep words u64[MASK_WORDS].
__host__ __device__ constexpr size_t nms_lds_mask_offset(const int cap, const int threads) { return (((size_t)cap * 29 + sizeof(int) * (threads + 1)) + 7) / 8 * 8; }
__host__ __device__ constexpr size_t nms_lds_bytes(const int cap, const int threads) { return nms_lds_mask_offset(cap, threads) + (size_t)MASK_CAP * MASK_WORDS * 8 + (size_t)cap * 2 + MASK_WORDS * 8; }

// `n` keys: the whole segment (read from a.keys), or -- `preselected` -- the n BEST keys of a longer segment of n_total entries, already in
// s_key (nms_select_body).  Returns false, having written nothing, when the scan ran out of preselected boxes before it had kept
// max_keep of them (the caller then runs the full path); true otherwise.  The return value is workgroup-uniform.
template <int THREADS, int CAP>
__device__ __forceinline__ bool nms_lds_body(const rn::NmsLaunch &a, unsigned char *smem, const int n, const bool preselected = false,
                                             const int n_total = 0)
{
    f32x4 *s_box = (f32x4 *)smem;
    uint64_t *s_key = (uint64_t *)(smem + (size_t)CAP * 16);
    float *s_area = (float *)(smem + (size_t)CAP * 24);
    uint8_t *s_supp = smem + (size_t)CAP * 28;
    int *s_scan = (int *)(smem + (size_t)CAP * 29);        // CAP % 4 == 0

    const int s = blockIdx.x;
    const int64_t start = a.seg_start[s];
    const int64_t box_base = a.box_mode ? (int64_t)(s / a.K) * a.A : start;

    int n_pad = 1;
    while (n_pad < n) n_pad <<= 1;
    if (!preselected) {
        for (int i = threadIdx.x; i < n_pad; i += THREADS) s_key[i] = (i < n) ? a.keys[start + i] : ~0ull;
    } else {
        for (int i = n + threadIdx.x; i < n_pad; i += THREADS) s_key[i] = ~0ull;      // s_key[0 .. n) holds the selected keys
    }
    __syncthreads();
    bitonic_sort_lds<THREADS>(s_key, n_pad);

    for (int i = threadIdx.x; i < n; i += THREADS) {
        const f32x4 b = a.boxes[box_base + (uint32_t)s_key[i]];
        s_box[i] = b;
        s_area[i] = (b.z - b.x) * (b.w - b.y);
        s_supp[i] = 0;
    }
    __syncthreads();

    // Blocked greedy scan, tiles of MASK_CAP sorted boxes (the box-by-box loop it replaces paid one workgroup barrier per KEPT box:
    // 2.5 ms for a 2 000-box segment of barely overlapping boxes -- an untrained head, or a crowded class):
    //   1. every box of the tile against the boxes KEPT in earlier tiles (all threads, no serial step);
    //   2. the tile's own pairwise suppression matrix as ballot words (a wave takes every NW-th row), as in nms_mask_kernel;
    //   3. wave 0 scans the matrix with the "removed" set in SGPRs, starting from the boxes step 1 removed, visiting only rows that
    //      suppress something;
    //   4. the kept boxes join the list step 1 of the later tiles reads.
    // Same predicate, same order: the keep set is the greedy one bit for bit.
    constexpr int NW = THREADS / RN_WAVE;
    uint64_t (*s_mask)[MASK_WORDS] = (uint64_t (*)[MASK_WORDS])(smem + nms_lds_mask_offset(CAP, THREADS));
    uint16_t *s_kept = (uint16_t *)(smem + nms_lds_mask_offset(CAP, THREADS) + (size_t)MASK_CAP * MASK_WORDS * 8);   // sorted positions of the kept boxes
    uint64_t *s_keepw = (uint64_t *)(smem + nms_lds_mask_offset(CAP, THREADS) + (size_t)MASK_CAP * MASK_WORDS * 8 + (size_t)CAP * 2);
    const int lane = threadIdx.x & (RN_WAVE - 1);
    int nkept = 0, n_done = 0;                                                     // n_done: sorted boxes the scan has decided
    for (int t0 = 0; t0 < n; t0 += MASK_CAP) {
        const int m = min(MASK_CAP, n - t0);
        if (nkept > 0) {                                                          // 1. (uniform)
            constexpr int G = THREADS / MASK_CAP;                                  // threads per tile box, striding the kept list
            const int j = threadIdx.x % MASK_CAP, g = threadIdx.x / MASK_CAP;      // (a wave shares g: its loop below is wave-uniform)
            const bool in = j < m;
            const f32x4 bj = s_box[t0 + (in ? j : 0)];
            const float aj = s_area[t0 + (in ? j : 0)];
            bool dead = false;
            for (int k = g; k < nkept; k += G) {
                const int p = s_kept[k];
                const bool live = in && !dead;
                if (overlaps_wave(s_box[p], s_area[p], bj, aj, a.iou_thr, live) && live) dead = true;
            }
            if (dead) s_supp[t0 + j] = 1;
            __syncthreads();
        }
        tile_matrix<NW>(s_box + t0, s_area + t0, m, s_mask, a.iou_thr);                // 2.
        __syncthreads();
        tile_scan(s_mask, s_supp + t0, m, s_keepw);                                    // 3.
        __syncthreads();
        {                                                                         // 4.
            int before = 0, total = 0;
            const int tw = (threadIdx.x % MASK_CAP) >> 6;
#pragma unroll
            for (int w = 0; w < MASK_WORDS; ++w) {
                const int c = __popcll(s_keepw[w]);
                if (w < tw) before += c;
                total += c;
            }
            if ((int)threadIdx.x < MASK_CAP && (int)threadIdx.x < m) {
                const uint64_t mine = s_keepw[tw];
                if ((mine >> lane) & 1ull) s_kept[nkept + before + __popcll(mine & ((1ull << lane) - 1ull))] = (uint16_t)(t0 + threadIdx.x);
                else s_supp[t0 + threadIdx.x] = 1;
            }
            nkept += total;
        }
        __syncthreads();
        n_done = t0 + m;
        if (a.max_keep > 0 && nkept >= a.max_keep) break;                          // (uniform) the rest of the segment cannot reach the caller's top max_keep
    }

    if (preselected && n < n_total && !(a.max_keep > 0 && nkept >= a.max_keep)) return false;      // (uniform: nkept is the same in every thread)
    const int per = (n_done + THREADS - 1) / THREADS;
    const int lo = min(n_done, (int)threadIdx.x * per), hi = min(n_done, lo + per);
    int cnt = 0;
    for (int i = lo; i < hi; ++i) cnt += s_supp[i] ? 0 : 1;
    int total;
    int off = block_excl_scan<THREADS>(s_scan, cnt, total);
    for (int i = lo; i < hi; ++i) {
        if (!s_supp[i]) {
            a.kept[start + off] = s_key[i];
            if (a.keep_idx) a.keep_idx[start + off] = (int64_t)(uint32_t)s_key[i];
            ++off;
        }
    }
    if (threadIdx.x == 0) a.kept_count[s] = total;
    return true;
}

// A segment longer than MED_CAP whose scan may stop after max_keep kept boxes (the detect chain): the greedy scan walks the boxes in
// score order and usually ends within the first few hundred -- so only the BEST <= MED_CAP keys are selected (a 2048-bin histogram of
// the score field over [min, max] finds the largest bin prefix that fits), sorted and scanned in LDS like a short segment; a.keys is only
// read.  SURVEY 8d's stress regime (~7 k candidates per class, 1 440 segments) paid a full sort of every segment through HBM merge
// passes: 949 us per batch of 16 images.  Returns false (uniform; nothing written) when the selection cannot be made (more than MED_CAP
// keys share the best bin: mass ties) or did not reach max_keep kept boxes: the caller runs nms_big_body.
template <int THREADS, int CAP>
__device__ __forceinline__ bool nms_select_body(const rn::NmsLaunch &a, unsigned char *smem)
{
    constexpr int BINS = 2048;
    static_assert(THREADS * 2 == BINS && (BINS + 8) * 4 <= CAP * 16, "two bins per thread; the histogram aliases the box area");
    uint64_t *s_key = (uint64_t *)(smem + (size_t)CAP * 16);
    int *s_hist = (int *)smem;                                   // [BINS] | [BINS .. BINS + 8): min, max, bin, count, fill
    const int s = blockIdx.x, t = threadIdx.x;
    const int n = a.seg_len[s];
    const uint64_t *keys = a.keys + a.seg_start[s];
    for (int i = t; i < BINS + 8; i += THREADS) s_hist[i] = (i == BINS) ? (int)0x7fffffff : (i == BINS + 2 ? -1 : 0);
    uint32_t lo = 0xffffffffu, hi = 0u;
    for (int i = t; i < n; i += THREADS) { const uint32_t h = (uint32_t)(keys[i] >> 32); lo = min(lo, h); hi = max(hi, h); }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { lo = min(lo, (uint32_t)__shfl_xor((int)lo, d, RN_WAVE)); hi = max(hi, (uint32_t)__shfl_xor((int)hi, d, RN_WAVE)); }
    __syncthreads();
    if ((t & (RN_WAVE - 1)) == 0 && lo <= hi) { atomicMin(&s_hist[BINS], (int)(lo >> 1)); atomicMax(&s_hist[BINS + 1], (int)(hi >> 1)); }
    __syncthreads();
    const uint32_t lo1 = (uint32_t)s_hist[BINS], range = (uint32_t)s_hist[BINS + 1] - lo1;      // in units of 2 ulps of the field (non-negative ints)
    const int shift = range >= (uint32_t)BINS ? (32 - __clz((int)range)) - 11 : 0;
    for (int i = t; i < n; i += THREADS) atomicAdd(&s_hist[(((uint32_t)(keys[i] >> 33)) - lo1) >> shift], 1);
    __syncthreads();
    // inclusive scan over the bins, two per thread (as topk_kernel): the last bin whose cumulative count still fits MED_CAP
    const int h0 = s_hist[2 * t], h1 = s_hist[2 * t + 1];
    int incl = h0 + h1;
    const int lane = t & (RN_WAVE - 1), wv = t >> 6;
#pragma unroll
    for (int d = 1; d < RN_WAVE; d <<= 1) { const int up = __shfl_up(incl, d, RN_WAVE); if (lane >= d) incl += up; }
    __shared__ int s_wtot[THREADS / RN_WAVE];
    if (lane == RN_WAVE - 1) s_wtot[wv] = incl;
    __syncthreads();
    int before = 0;
    for (int w = 0; w < wv; ++w) before += s_wtot[w];
    // cum(b) = keys in bins 0 .. b never decreases: tb = the LAST bin with cum(tb) <= CAP (none when the best bin alone exceeds CAP).
    // Thread t owns bins 2t and 2t + 1 and knows cum(2t - 1), cum(2t), cum(2t + 1); exactly one of the three tests below fires.
    const int c1 = before + incl, c0 = c1 - h1, prev = c0 - h0;
    if (c0 <= CAP && c1 > CAP) { s_hist[BINS + 2] = 2 * t; s_hist[BINS + 3] = c0; }
    if (t > 0 && prev <= CAP && c0 > CAP) { s_hist[BINS + 2] = 2 * t - 1; s_hist[BINS + 3] = prev; }
    if (t == THREADS - 1 && c1 <= CAP) { s_hist[BINS + 2] = BINS - 1; s_hist[BINS + 3] = c1; }
    __syncthreads();
    const int tb = s_hist[BINS + 2], m = s_hist[BINS + 3];
    if (tb < 0 || m <= 0) return false;                          // the best bin alone holds more than MED_CAP keys
    for (int i = t; i < n; i += THREADS) {
        const uint64_t k = keys[i];
        if ((int)((((uint32_t)(k >> 33)) - lo1) >> shift) <= tb) s_key[atomicAdd(&s_hist[BINS + 4], 1)] = k;
    }
    __syncthreads();
    return nms_lds_body<THREADS, CAP>(a, smem, m, true, n);
}

// Segments of up to MASK_CAP entries (the common case: ~100 candidates per (image, class) at the
// reference's prior): rank sort, the full pairwise suppression matrix as 64-bit ballot words, then
// a register-resident scan.  Nothing in it is serial except the n-step scan of 4 words.
__global__ __launch_bounds__(MASK_CAP) void nms_mask_kernel(const rn::NmsLaunch a)
{
    __shared__ uint64_t s_in[MASK_CAP];
    __shared__ uint64_t s_key[MASK_CAP];
    __shared__ f32x4 s_box[MASK_CAP];
    __shared__ float s_area[MASK_CAP];
    __shared__ uint64_t s_mask[MASK_CAP][MASK_WORDS];
    __shared__ uint64_t s_keep[MASK_WORDS];

    STAMP(0);
    const int s = blockIdx.x;
    const int n = a.seg_len[s];
    if (n > MASK_CAP) return;
    if (n == 0) {
        if (threadIdx.x == 0) a.kept_count[s] = 0;
        return;
    }
    const int t = threadIdx.x;
    const int lane = t & (RN_WAVE - 1), wave = t >> 6;
    const int64_t start = a.seg_start[s];
    const int64_t box_base = a.box_mode ? (int64_t)(s / a.K) * a.A : start;

    const uint64_t key = (t < n) ? a.keys[start + t] : ~0ull;
    s_in[t] = key;
    __syncthreads();
    STAMP(1);
    if (t < n) {                                       // rank sort: keys are unique
        // s_in is padded with ~0 up to MASK_CAP (never < key): eight independent LDS reads per step, no guards (a rolled loop of
        // single reads waited out one LDS latency per key: ~120 of them in a row on every thread)
        int rank = 0;
        const int n8 = (n + 7) & ~7;
        for (int j = 0; j < n8; j += 8) {
            uint64_t v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = s_in[j + u];
#pragma unroll
            for (int u = 0; u < 8; ++u) rank += (v[u] < key) ? 1 : 0;
        }
        s_key[rank] = key;
    }
    __syncthreads();
    STAMP(2);
    if (t < n) {
        const f32x4 b = a.boxes[box_base + (uint32_t)s_key[t]];
        s_box[t] = b;
        s_area[t] = (b.z - b.x) * (b.w - b.y);
    }
    __syncthreads();
    STAMP(3);
    // suppression matrix: bit j of s_mask[i][w] (j = 64w + bit) <=> j > i and IoU(i, j) > thr.  A wave takes every 4th row;
    // its lanes keep column j = 64w + lane of every word w in registers (box + area), the row's box is one LDS broadcast,
    // and lane w collects word w so a row costs one LDS store.
    const int nw = (n + 63) >> 6;
    f32x4 bj[MASK_WORDS];
    float aj[MASK_WORDS];
#pragma unroll
    for (int w = 0; w < MASK_WORDS; ++w) {
        const int j = min(w * 64 + lane, n - 1);
        bj[w] = s_box[j]; aj[w] = s_area[j];
    }
    f32x4 bi_n = s_box[min(wave, n - 1)];
    float ai_n = s_area[min(wave, n - 1)];
    for (int i = wave; i < n; i += MASK_CAP / RN_WAVE) {
        const f32x4 bi = bi_n;                               // (the next row's box is requested before this row's arithmetic)
        const float ai = ai_n;
        bi_n = s_box[min(i + MASK_CAP / RN_WAVE, n - 1)];
        ai_n = s_area[min(i + MASK_CAP / RN_WAVE, n - 1)];
        const int w0 = (i + 1) >> 6;                         // words entirely at or below the diagonal hold no j > i
        uint64_t mine = 0ull;
#pragma unroll
        for (int w = 0; w < MASK_WORDS; ++w) {
            if (w >= w0 && w < nw) {                         // wave-uniform
                const int j = w * 64 + lane;
                const bool live = j > i && j < n;
                const bool sup = overlaps_wave(bi, ai, bj[w], aj[w], a.iou_thr, live) && live;
                const unsigned long long m = __ballot(sup);
                if (lane == w) mine = m;
            }
        }
        if (lane < MASK_WORDS) s_mask[i][lane] = mine;
    }
    __syncthreads();
    STAMP(4);
    // greedy scan by wave 0: row i of the matrix sits in the registers of lane i & 63 (register set i >> 6), the "removed"
    // set is wave-uniform (SGPRs); a kept row is fetched with v_readlane -- no LDS round trip on the serial chain
    if (wave == 0) {
        uint64_t row[MASK_WORDS][MASK_WORDS];
#pragma unroll
        for (int wb = 0; wb < MASK_WORDS; ++wb)
#pragma unroll
            for (int ww = 0; ww < MASK_WORDS; ++ww) row[wb][ww] = (wb * 64 + lane < n && ww < nw) ? s_mask[wb * 64 + lane][ww] : 0ull;
        uint64_t rem[MASK_WORDS];
#pragma unroll
        for (int w = 0; w < MASK_WORDS; ++w) rem[w] = 0;
        // Only a kept row that suppresses something changes the state, so the serial chain walks those rows alone: `todo` = rows
        // of this word with a non-empty mask row, not yet visited; each step takes the lowest one that is still alive.  (A row with
        // an empty mask row is kept or dropped by what is in `rem` when the scan ends -- nothing to do for it.)  With ~120 boxes
        // per segment and a handful of overlapping pairs this is a few steps instead of one per box.
#pragma unroll
        for (int w = 0; w < MASK_WORDS; ++w) {
            if (w * 64 < n) {
                uint64_t any = 0;
#pragma unroll
                for (int ww = 0; ww < MASK_WORDS; ++ww) if (ww >= w) any |= row[w][ww];
                uint64_t todo = __ballot(any != 0ull);
                while (true) {
                    const uint64_t live = todo & ~rem[w];
                    if (!live) break;
                    const int bit = __builtin_ctzll(live);
#pragma unroll
                    for (int ww = 0; ww < MASK_WORDS; ++ww)
                        if (ww >= w) {
                            const unsigned lo = __builtin_amdgcn_readlane((unsigned)row[w][ww], bit);
                            const unsigned hi = __builtin_amdgcn_readlane((unsigned)(row[w][ww] >> 32), bit);
                            rem[ww] |= ((uint64_t)hi << 32) | lo;
                        }
                    todo &= ~((2ull << bit) - 1ull);                 // rows up to `bit` are done
                }
            }
        }
        if (lane < MASK_WORDS) {
            uint64_t r = 0;
#pragma unroll
            for (int w = 0; w < MASK_WORDS; ++w) if (lane == w) r = rem[w];
            const int lo = lane * 64;
            const uint64_t valid = (n - lo >= 64) ? ~0ull : ((n - lo > 0) ? ((1ull << (n - lo)) - 1ull) : 0ull);
            s_keep[lane] = ~r & valid;
        }
    }
    __syncthreads();
    STAMP(5);
    int before = 0, total = 0;
#pragma unroll
    for (int w = 0; w < MASK_WORDS; ++w) {
        const int c = __popcll(s_keep[w]);
        if (w < wave) before += c;
        total += c;
    }
    const uint64_t mine = s_keep[wave];
    if ((mine >> lane) & 1ull) {
        const int pos = before + __popcll(mine & ((1ull << lane) - 1ull));
        a.kept[start + pos] = s_key[t];
        if (a.keep_idx) a.keep_idx[start + pos] = (int64_t)(uint32_t)s_key[t];
    }
    if (t == 0) a.kept_count[s] = total;
    STAMP(6);
}

// Segments longer than MED_CAP: chunk sort in LDS, merge passes and NMS state in HBM.
__device__ __forceinline__ void nms_big_body(const rn::NmsLaunch &a, unsigned char *smem)
{
    __shared__ uint64_t s_key[MED_CAP];
    __shared__ int s_scan[BIG_THREADS + 1];

    const int s = blockIdx.x;
    const int n = a.seg_len[s];
    const int64_t start = a.seg_start[s];
    const int64_t box_base = a.box_mode ? (int64_t)(s / a.K) * a.A : start;
    uint64_t *buf0 = a.keys + start, *buf1 = a.kept + start;

    // phase 1: sort each MED_CAP chunk in LDS
    for (int c0 = 0; c0 < n; c0 += MED_CAP) {
        const int m = min(MED_CAP, n - c0);
        int m_pad = 1;
        while (m_pad < m) m_pad <<= 1;
        for (int i = threadIdx.x; i < m_pad; i += BIG_THREADS) s_key[i] = (i < m) ? buf0[c0 + i] : ~0ull;
        __syncthreads();
        bitonic_sort_lds<BIG_THREADS>(s_key, m_pad);
        for (int i = threadIdx.x; i < m; i += BIG_THREADS) buf0[c0 + i] = s_key[i];
        __syncthreads();
    }
    // phase 2: pairwise merges, each element finds its rank in the sibling run (keys are unique)
    uint64_t *src = buf0, *dst = buf1;
    for (int64_t w = MED_CAP; w < n; w <<= 1) {
        for (int i = threadIdx.x; i < n; i += BIG_THREADS) {
            const int64_t pair0 = ((int64_t)i / (2 * w)) * (2 * w);
            const int64_t mid = min(pair0 + w, (int64_t)n), end = min(pair0 + 2 * w, (int64_t)n);
            const uint64_t key = src[i];
            int64_t lo, hi;
            if (i < mid) { lo = mid; hi = end; } else { lo = pair0; hi = mid; }
            const int64_t sib0 = lo;
            while (lo < hi) {                          // count sibling keys < key
                const int64_t md = (lo + hi) >> 1;
                if (src[md] < key) lo = md + 1; else hi = md;
            }
            const int64_t rank_sib = lo - sib0;
            const int64_t rank_own = (i < mid) ? (i - pair0) : (i - mid);
            dst[pair0 + rank_own + rank_sib] = key;
        }
        __syncthreads();
        uint64_t *t = src; src = dst; dst = t;
    }
    if (src != buf0) {                                   // sorted keys must end in a.keys, survivors go to a.kept
        for (int i = threadIdx.x; i < n; i += BIG_THREADS) buf0[i] = src[i];
        __syncthreads();
    }

    f32x4 *g_box = a.scratch_boxes + start;
    uint8_t *g_supp = a.scratch_supp + start;
    int n_done = 0;
    // (max_keep > 0: the scan usually ends within the first tiles -- each tile gathers its own boxes and clears its own flags instead
    // of a pass over the whole segment up front; g_box then only holds the compacted kept boxes)
    if (a.max_keep <= 0) {
        for (int i = threadIdx.x; i < n; i += BIG_THREADS) {
            g_box[i] = a.boxes[box_base + (uint32_t)buf0[i]];
            g_supp[i] = 0;
        }
    }
    __syncthreads();
    // The blocked greedy scan of nms_lds_body with the sorted boxes in HBM: a tile of MASK_CAP boxes at a time in LDS; the boxes kept so
    // far are COMPACTED IN PLACE at the front of g_box (a kept box moves to a position at or below its own, and tiles are read front to
    // back, so nothing unread is overwritten) and pass through LDS 1 024 at a time for step 1.  (The box-by-box loop this replaces: one
    // barrier and an HBM round trip per kept box, 2.5 ms for a 3 000-box segment of barely overlapping boxes.)
    {
        constexpr int NW = BIG_THREADS / RN_WAVE, G = BIG_THREADS / MASK_CAP;
        f32x4 *s_tbox = (f32x4 *)smem;                                                      // [MASK_CAP]
        f32x4 *s_kbox = (f32x4 *)(smem + (size_t)MASK_CAP * 16);                            // [BIG_THREADS] kept boxes of the current chunk
        float *s_tarea = (float *)(smem + (size_t)(MASK_CAP + BIG_THREADS) * 16);           // [MASK_CAP]
        uint64_t (*s_mask)[MASK_WORDS] = (uint64_t (*)[MASK_WORDS])(smem + (size_t)(MASK_CAP + BIG_THREADS) * 16 + (size_t)MASK_CAP * 4);
        uint64_t *s_keepw = (uint64_t *)(smem + (size_t)(MASK_CAP + BIG_THREADS) * 16 + (size_t)MASK_CAP * 4 + (size_t)MASK_CAP * MASK_WORDS * 8);
        uint8_t *s_tsupp = smem + (size_t)(MASK_CAP + BIG_THREADS) * 16 + (size_t)MASK_CAP * 4 + (size_t)MASK_CAP * MASK_WORDS * 8 + MASK_WORDS * 8;
        const int lane = threadIdx.x & (RN_WAVE - 1);
        const int j = threadIdx.x % MASK_CAP, g = threadIdx.x / MASK_CAP;
        int nkept = 0;
        for (int t0 = 0; t0 < n; t0 += MASK_CAP) {
            const int m = min(MASK_CAP, n - t0);
            if ((int)threadIdx.x < m) {
                const f32x4 b = a.max_keep > 0 ? a.boxes[box_base + (uint32_t)buf0[t0 + threadIdx.x]] : g_box[t0 + threadIdx.x];
                s_tbox[threadIdx.x] = b;
                s_tarea[threadIdx.x] = (b.z - b.x) * (b.w - b.y);
                s_tsupp[threadIdx.x] = 0;
            }
            __syncthreads();
            const bool in = j < m;
            const f32x4 bj = s_tbox[in ? j : 0];
            const float aj = s_tarea[in ? j : 0];
            bool dead = false;
            for (int c0 = 0; c0 < nkept; c0 += BIG_THREADS) {                                // 1. (uniform trip count)
                const int kc = min(BIG_THREADS, nkept - c0);
                if ((int)threadIdx.x < kc) s_kbox[threadIdx.x] = g_box[c0 + threadIdx.x];
                __syncthreads();
                for (int k = g; k < kc; k += G) {
                    const f32x4 bk = s_kbox[k];
                    const bool live = in && !dead;
                    if (overlaps_wave(bk, (bk.z - bk.x) * (bk.w - bk.y), bj, aj, a.iou_thr, live) && live) dead = true;
                }
                __syncthreads();                                                             // the chunk is overwritten next
            }
            if (dead) s_tsupp[j] = 1;
            __syncthreads();
            tile_matrix<NW>(s_tbox, s_tarea, m, s_mask, a.iou_thr);                          // 2.
            __syncthreads();
            tile_scan(s_mask, s_tsupp, m, s_keepw);                                          // 3.
            __syncthreads();
            {                                                                                // 4.
                int before = 0, total = 0;
                const int tw = j >> 6;
#pragma unroll
                for (int w = 0; w < MASK_WORDS; ++w) {
                    const int c = __popcll(s_keepw[w]);
                    if (w < tw) before += c;
                    total += c;
                }
                if ((int)threadIdx.x < m) {
                    const uint64_t mine = s_keepw[tw];
                    const bool kp = (mine >> lane) & 1ull;
                    if (kp) g_box[nkept + before + __popcll(mine & ((1ull << lane) - 1ull))] = s_tbox[threadIdx.x];
                    g_supp[t0 + threadIdx.x] = kp ? 0 : 1;
                }
                nkept += total;
            }
            __syncthreads();
            n_done = t0 + m;
            if (a.max_keep > 0 && nkept >= a.max_keep) break;                                // (uniform) see nms_lds_body
        }
    }
    const int per = (n_done + BIG_THREADS - 1) / BIG_THREADS;
    const int lo = min(n_done, (int)threadIdx.x * per), hi = min(n_done, lo + per);
    int cnt = 0;
    for (int i = lo; i < hi; ++i) cnt += g_supp[i] ? 0 : 1;
    int total;
    int off = block_excl_scan<BIG_THREADS>(s_scan, cnt, total);
    for (int i = lo; i < hi; ++i) {
        if (!g_supp[i]) {
            buf1[off] = buf0[i];
            if (a.keep_idx) a.keep_idx[start + off] = (int64_t)(uint32_t)buf0[i];
            ++off;
        }
    }
    if (threadIdx.x == 0) a.kept_count[s] = total;
}

// op-boundary helpers: build keys from (scores, seg_off)
__global__ __launch_bounds__(256) void nms_prep_kernel(const float *__restrict__ scores, const int32_t *__restrict__ seg_off,
                                                       const int S, const int64_t N, uint64_t *__restrict__ keys,
                                                       int64_t *__restrict__ seg_start, int32_t *__restrict__ seg_len)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < S) { seg_start[i] = seg_off[i]; seg_len[i] = seg_off[i + 1] - seg_off[i]; }
    if (i < N) {
        int lo = 0, hi = S;                       // segment of element i: last s with seg_off[s] <= i
        while (hi - lo > 1) { const int md = (lo + hi) >> 1; if (seg_off[md] <= i) lo = md; else hi = md; }
        keys[i] = ((uint64_t)rn::inv_ordered(scores[i]) << 32) | (uint32_t)(i - seg_off[lo]);
    }
}

size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

// Segments longer than MASK_CAP in ONE launch of 1024-thread workgroups (most exit at once: such segments are rare at the
// reference's score threshold): up to MED_CAP entries sort and suppress in LDS, longer ones go through HBM.
__global__ __launch_bounds__(BIG_THREADS) void nms_large_kernel(const rn::NmsLaunch a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int n = a.seg_len[blockIdx.x];
    if (n <= MASK_CAP) return;
    if (n <= MED_CAP) { nms_lds_body<BIG_THREADS, MED_CAP>(a, smem, n); return; }
    if (a.max_keep > 0 && nms_select_body<BIG_THREADS, MED_CAP>(a, smem)) return;
    __syncthreads();
    nms_big_body(a, smem);
}

}  // namespace

int rn::launch_nms(const rn::NmsLaunch &a, hipStream_t st)
{
    if (a.S <= 0) return RN_OK;
    const size_t lds_med = nms_lds_bytes(MED_CAP, BIG_THREADS);
    {   // > 64 KiB of LDS in one workgroup (static + dynamic) needs the opt-in once per device
        static bool attr_set[64] = {};
        int dev = 0;
        RN_HIP(hipGetDevice(&dev));
        if (dev < 0 || dev >= 64 || !attr_set[dev]) {
            RN_HIP(hipFuncSetAttribute((const void *)nms_large_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_med));
            if (dev >= 0 && dev < 64) attr_set[dev] = true;
        }
    }
    hipLaunchKernelGGL(nms_mask_kernel, dim3((unsigned)a.S), dim3(MASK_CAP), 0, st, a);                 // n <= 256
    RN_LAUNCH_CHECK();
    hipLaunchKernelGGL(nms_large_kernel, dim3((unsigned)a.S), dim3(BIG_THREADS), lds_med, st, a);       // everything longer, one launch
    RN_LAUNCH_CHECK();
    return RN_OK;
}

// workspace (op mode): keys u64[N] | kept u64[N] | seg_start i64[S] | seg_len i32[S] | scratch boxes f32x4[N] | supp u8[N]
RN_API size_t rn_nms_workspace_bytes(int64_t N, int S)
{
    if (N < 0 || S < 0) return 0;
    return al256((size_t)N * 8) * 2 + al256((size_t)S * 8) + al256((size_t)S * 4) + al256((size_t)N * 16) + al256((size_t)N) + 256;
}

RN_API int rn_nms_segments(const float *boxes, const float *scores, const int32_t *seg_off, int S, int64_t N,
                           float iou_thr, int64_t *keep, int32_t *keep_count, void *workspace, size_t workspace_bytes,
                           void *stream)
{
    if (!seg_off || !keep_count || S <= 0 || N < 0) return RN_EINVAL;
    if (N > 0 && (!boxes || !scores || !keep)) return RN_EINVAL;
    if (N >= ((int64_t)1 << 31)) return RN_EUNSUPPORTED;
    if (!workspace || workspace_bytes < rn_nms_workspace_bytes(N, S)) return RN_EWORKSPACE;
    if ((boxes && !rn::aligned(boxes, 16)) || !rn::aligned(workspace, 16)) return RN_EALIGN;
    unsigned char *w = (unsigned char *)workspace;
    rn::NmsLaunch a;
    a.keys = (uint64_t *)w; w += al256((size_t)N * 8);
    a.kept = (uint64_t *)w; w += al256((size_t)N * 8);
    int64_t *seg_start = (int64_t *)w; w += al256((size_t)S * 8);
    int32_t *seg_len = (int32_t *)w; w += al256((size_t)S * 4);
    a.scratch_boxes = (rn::f32x4 *)w; w += al256((size_t)N * 16);
    a.scratch_supp = (uint8_t *)w;
    a.keep_idx = keep;
    a.boxes = (const rn::f32x4 *)boxes;
    a.seg_start = seg_start; a.seg_len = seg_len; a.kept_count = keep_count;
    a.S = S; a.box_mode = 0; a.K = 1; a.A = 0; a.iou_thr = iou_thr; a.max_keep = 0;
    hipStream_t st = (hipStream_t)stream;
    const int64_t work = N > S ? N : S;
    hipLaunchKernelGGL(nms_prep_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, st, scores, seg_off, S, N,
                       a.keys, seg_start, seg_len);
    RN_LAUNCH_CHECK();
    return rn::launch_nms(a, st);
}

RN_API int rn_debug_nms_stamps(unsigned long long *host_out, int n)
{
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_nms_stamps), sizeof(unsigned long long) * (size_t)n, 0, hipMemcpyDeviceToHost);
}
