// K4-K7 inference chain -- replaces Retinanet.process_detections
// (retinanet/models.py:160-243) and the torchvision ops it calls.
// Compiled with -ffp-contract=off: box arithmetic that feeds NMS decisions must
// round like the CPU path.
#include "rn_internal.hpp"

namespace {

// ---- K4 decode + clip -----------------------------------------------------------
// activ_2_bbox (retinanet/box_utils.py:37-48): centres from (dx,dy), sizes from
// exp(dx), exp(dy) (Q4: the reference uses [..., :2] for both), then
// clip_boxes_to_image (retinanet/models.py:189) with the resized, unpadded size.
template <int DT> struct delta4;
template <> struct delta4<RN_F32> {
    static __device__ __forceinline__ void ld(const void *p, int64_t row, float (&f)[4]) {
        const rn::f32x4 v = ((const rn::f32x4 *)p)[row];
        f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
    }
};
template <> struct delta4<RN_BF16> {
    static __device__ __forceinline__ void ld(const void *p, int64_t row, float (&f)[4]) {
        const rn::u32x2 v = ((const rn::u32x2 *)p)[row];
        f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
        f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
    }
};
template <> struct delta4<RN_F16> {
    static __device__ __forceinline__ void ld(const void *p, int64_t row, float (&f)[4]) {
        const rn::u32x2 v = ((const rn::u32x2 *)p)[row];
        f[0] = rn::half_lo(v.x); f[1] = rn::half_hi(v.x); f[2] = rn::half_lo(v.y); f[3] = rn::half_hi(v.y);
    }
};

struct RegW { float w[4]; };

__device__ __forceinline__ float clampf(float v, const float lo, const float hi)
{
    v = v < lo ? lo : v;
    return v > hi ? hi : v;
}

__device__ __forceinline__ rn::f32x4 decode_one(const float (&d)[4], const rn::f32x4 an, const RegW rw)
{
    const float dx = d[0] / rw.w[0], dy = d[1] / rw.w[1];
    const float acx = (an.x + an.z) / 2.0f, acy = (an.y + an.w) / 2.0f;
    const float aw = an.z - an.x, ah = an.w - an.y;
    const float cx = aw * dx + acx, cy = ah * dy + acy;
    const float w = aw * expf(dx), h = ah * expf(dy);
    rn::f32x4 o;
    o.x = cx - w / 2.0f; o.y = cy - h / 2.0f; o.z = cx + w / 2.0f; o.w = cy + h / 2.0f;
    return o;
}

template <int DT>
// deltas: one pyramid level [B][A_l][4]; its anchors are rows base .. base+A_l of the A per image.
__global__ __launch_bounds__(256) void decode_clip_kernel(const void *__restrict__ deltas, const int64_t A_l, const int64_t base,
                                                          const int64_t A, const int64_t R,
                                                          const rn::f32x4 *__restrict__ anchors, const int64_t anchor_bstride4,
                                                          const int32_t *__restrict__ image_hw, const RegW rw,
                                                          rn::f32x4 *__restrict__ out)
{
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < R; r += (int64_t)gridDim.x * blockDim.x) {
        const int b = (int)((uint32_t)r / (uint32_t)A_l);
        const int64_t ai = base + (r - (int64_t)b * A_l);
        float d[4];
        delta4<DT>::ld(deltas, r, d);
        rn::f32x4 o = decode_one(d, anchors[(int64_t)b * anchor_bstride4 + ai], rw);
        if (image_hw) {
            const float hh = (float)image_hw[2 * b], ww = (float)image_hw[2 * b + 1];
            o.x = clampf(o.x, 0.0f, ww); o.z = clampf(o.z, 0.0f, ww);
            o.y = clampf(o.y, 0.0f, hh); o.w = clampf(o.w, 0.0f, hh);
        }
        out[(int64_t)b * A + ai] = o;
    }
}

int launch_decode(const void *deltas, const int dtype, const int B, const int64_t A_l, const int64_t base, const int64_t A,
                  const float *anchors, const int64_t anchor_bstride, const int32_t *image_hw, const RegW rw, float *out,
                  hipStream_t st)
{
    const int64_t R = (int64_t)B * A_l;
    int64_t blocks = (R + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    const dim3 g((unsigned)blocks), blk(256);
    switch (dtype) {
        case RN_F32: hipLaunchKernelGGL((decode_clip_kernel<RN_F32>), g, blk, 0, st, deltas, A_l, base, A, R, (const rn::f32x4 *)anchors, anchor_bstride / 4, image_hw, rw, (rn::f32x4 *)out); break;
        case RN_BF16: hipLaunchKernelGGL((decode_clip_kernel<RN_BF16>), g, blk, 0, st, deltas, A_l, base, A, R, (const rn::f32x4 *)anchors, anchor_bstride / 4, image_hw, rw, (rn::f32x4 *)out); break;
        case RN_F16: hipLaunchKernelGGL((decode_clip_kernel<RN_F16>), g, blk, 0, st, deltas, A_l, base, A, R, (const rn::f32x4 *)anchors, anchor_bstride / 4, image_hw, rw, (rn::f32x4 *)out); break;
        default: return RN_EINVAL;
    }
    RN_LAUNCH_CHECK();
    return RN_OK;
}


// ---- K5 score scan + candidate compaction ---------------------------------------
// Streams the class logits once (A*K*s bytes per image, 16-byte non-temporal loads, two groups in
// flight per wave; each wave owns one contiguous range of the flattened tensor).  An element is a
// candidate when sigmoid(x) > score_thr (models.py:196).  Candidates (~4e-4 of the elements at the
// reference's prior) are staged in a wave-private LDS list and flushed with ONE global atomic per
// (wave, image): no block barrier and no dependent global load anywhere in the stream.  The
// streaming loop itself only spots 16-byte vectors that contain an element above a logit-space
// pre-threshold and parks them (raw bits + position) in a wave-private LDS queue; the exact test
// sigmoid(x) > thr, the index arithmetic and the key are done later for 64 parked vectors at a
// time with all lanes busy, instead of with 1-2 active lanes at every hit.  The box
// side of the filter (remove_small_boxes, models.py:203) runs afterwards on the candidates only
// (cand_filter_kernel), which is also where their boxes are decoded.
constexpr int SCAN_THREADS = 256;
constexpr int SCAN_WAVES = SCAN_THREADS / RN_WAVE;
constexpr int SCAN_CAP = 128;            // wave-private list entries (>= 64: one ballot's worth always fits after a flush)
constexpr int SCAN_PF = 4;
constexpr int SCAN_QCAP = 128;           // parked vectors per wave
constexpr int SCAN_SHARDS = 128;         // candidate-list shards per image

// One pyramid level of the class logits, [B][A_l][K] dense; the levels are scanned as one virtual
// sequence of 16-byte vectors (level l = vectors voff .. voff + nvec), like K3 (loss.hip).
struct ScanLevel {
    const void *cls;
    int64_t A_l, base;       // anchors of this level per image; index of its first anchor among the A
    int64_t N, nvec, voff;   // elements, whole vectors, first virtual vector
};

struct ScanArgs {
    int L;
    ScanLevel lv[RN_MAX_LEVELS];
    int64_t A, total_vec, vec_per_wave, C;
    int32_t K, B;
    int32_t S;               // candidate-list shards per image (power of two)
    int64_t Cs;              // capacity of one shard; S * Cs <= C
    float score_thr, pre_thr;
    uint64_t *cand;          // [B][S][Cs]  (inv_ordered(score) << 32) | (anchor*K + k)
    int32_t *cand_count;     // [B][S]
};

struct ScanList { uint64_t key[SCAN_CAP]; int img[SCAN_CAP]; };
struct HitQueue { rn::u32x4 raw[SCAN_QCAP]; int64_t vv[SCAN_QCAP]; };     // vv: virtual vector index (level voff + v)

// wave-level flush: one atomic per distinct image in the list.  The list of an image is split into S
// shards (wave g appends to shard g mod S): returning atomics on one address cost ~130 ns each on
// MI355X, and with one counter per image the ~8000 end-of-kernel flushes took 65 us.
__device__ __forceinline__ void scan_flush(ScanList &sl, const int fill, const ScanArgs &a, const int lane, const int shard)
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    bool done0 = lane >= fill, done1 = lane + RN_WAVE >= fill;                 // entries lane, lane+64
    const uint64_t k0 = done0 ? 0 : sl.key[lane], k1 = done1 ? 0 : sl.key[lane + RN_WAVE];
    const int b0 = done0 ? -1 : sl.img[lane], b1 = done1 ? -1 : sl.img[lane + RN_WAVE];
    while (true) {
        const unsigned long long pend = __ballot(!done0 || !done1);
        if (!pend) break;
        const int src = __ffsll((long long)pend) - 1;
        const int bsel_local = !done0 ? b0 : b1;
        const int b = __shfl(bsel_local, src, RN_WAVE);                        // image handled this round (wave-uniform)
        const bool m0 = !done0 && b0 == b, m1 = !done1 && b1 == b;
        const unsigned long long mk0 = __ballot(m0), mk1 = __ballot(m1);
        const int n0 = __popcll(mk0), n = n0 + __popcll(mk1);
        int base = 0;
        const int64_t list = (int64_t)b * a.S + shard;
        if (lane == 0) base = atomicAdd(&a.cand_count[list], n);
        base = __shfl(base, 0, RN_WAVE);
        const unsigned long long lt = (1ull << lane) - 1ull;
        if (m0) {
            const int64_t pos = (int64_t)base + __popcll(mk0 & lt);
            if (pos < a.Cs) a.cand[list * a.Cs + pos] = k0;
            done0 = true;
        }
        if (m1) {
            const int64_t pos = (int64_t)base + n0 + __popcll(mk1 & lt);
            if (pos < a.Cs) a.cand[list * a.Cs + pos] = k1;
            done1 = true;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// exact score of one element (rare path), models.py:170
__device__ __forceinline__ float scan_score(const float x) { return 1.0f / (1.0f + expf(-x)); }

// key and image of element e of a level (A_l anchors per image, the first one is anchor `base`), whose score s passed
__device__ __forceinline__ void scan_key(const ScanArgs &a, const int64_t A_l, const int64_t base, const float s, const int64_t e,
                                         uint64_t &key, int &img)
{
    const int64_t r = e / a.K;                                // row of this level's [B*A_l][K]
    const uint32_t k = (uint32_t)(e - r * a.K);
    const int b = (int)((uint32_t)r / (uint32_t)A_l);
    const uint32_t anchor = (uint32_t)(base + (r - (int64_t)b * A_l));
    key = ((uint64_t)rn::inv_ordered(s) << 32) | (uint32_t)(anchor * (uint32_t)a.K + k);
    img = b;
}

template <int DT>
__global__ __launch_bounds__(SCAN_THREADS) void score_scan_kernel(const ScanArgs a)
{
    typedef rn::dt<DT> D;
    constexpr int VEC = D::VEC;
    __shared__ ScanList s_list[SCAN_WAVES];
    __shared__ HitQueue s_queue[SCAN_WAVES];
    const int lane = threadIdx.x & (RN_WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    ScanList &sl = s_list[wave];
    HitQueue &hq = s_queue[wave];
    const int64_t gwave = (int64_t)blockIdx.x * SCAN_WAVES + wave;
    const int64_t w_beg = gwave * a.vec_per_wave;
    const int64_t w_end = min(w_beg + a.vec_per_wave, a.total_vec);
    int fill = 0, qfill = 0;                                   // wave-uniform
    const int shard = (int)(gwave & (int64_t)(a.S - 1));
    const unsigned long long lt = (1ull << lane) - 1ull;

    // append the candidates selected by `pred` (at most one per lane): ballot, maybe flush, write
    auto append = [&](const bool pred, const uint64_t key, const int img) {
        const unsigned long long mk = __ballot(pred);
        if (!mk) return;
        const int n = __popcll(mk);
        if (fill + n > SCAN_CAP) { scan_flush(sl, fill, a, lane, shard); fill = 0; }
        if (pred) {
            const int pos = fill + __popcll(mk & lt);
            sl.key[pos] = key; sl.img[pos] = img;
        }
        fill += n;
    };
    // exact test + keys of the parked vectors, one vector per lane
    auto drain = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (int q0 = 0; q0 < qfill; q0 += RN_WAVE) {
            const bool valid = q0 + lane < qfill;
            const int qi = valid ? q0 + lane : 0;
            const int64_t vv = hq.vv[qi];
            int64_t voff = 0, A_l = a.lv[0].A_l, abase = 0;   // level of this vector
#pragma unroll
            for (int l = 1; l < RN_MAX_LEVELS; ++l)
                if (l < a.L && vv >= a.lv[l].voff) { voff = a.lv[l].voff; A_l = a.lv[l].A_l; abase = a.lv[l].base; }
            const int64_t e0 = (vv - voff) * VEC;
#pragma unroll 1
            for (int j = 0; j < VEC; ++j) {                   // rolled: element j is re-read from the parked bits
                const float xj = D::ld((const void *)&hq.raw[qi], j);
                bool c = valid && (xj > a.pre_thr);
                uint64_t key = 0; int img = 0;
                if (c) {
                    const float sc = scan_score(xj);
                    c = sc > a.score_thr;
                    if (c) scan_key(a, A_l, abase, sc, e0 + j, key, img);
                }
                append(c, key, img);
            }
        }
        qfill = 0;
        __builtin_amdgcn_wave_barrier();
    };

    for (int l = 0; l < a.L; ++l) {                            // wave-uniform: the levels this wave's range touches
        const ScanLevel &lv = a.lv[l];
        const int64_t v_beg = max(w_beg, lv.voff) - lv.voff, v_end = min(w_end, lv.voff + lv.nvec) - lv.voff;
        const rn::u32x4 *src = (const rn::u32x4 *)lv.cls;
        auto do_vec = [&](const rn::u32x4 raw, const int64_t v) {
            float x[VEC];
            D::unpack(raw, x);
            bool hit = false;
#pragma unroll
            for (int j = 0; j < VEC; ++j) hit |= (x[j] > a.pre_thr);
            const unsigned long long mk = __ballot(hit);
            if (!mk) return;                                  // common case: ~0.2 candidates per 512 elements
            const int n = __popcll(mk);
            if (qfill + n > SCAN_QCAP) drain();
            if (hit) {
                const int pos = qfill + __popcll(mk & lt);
                hq.raw[pos] = raw; hq.vv[pos] = lv.voff + v;
            }
            qfill += n;
        };
        if (v_beg < v_end) {
            const int64_t last = v_end - 1;
            const int64_t groups = (v_end - v_beg) / (SCAN_PF * RN_WAVE);
            rn::u32x4 q[SCAN_PF];
#pragma unroll
            for (int u = 0; u < SCAN_PF; ++u) q[u] = __builtin_nontemporal_load(&src[min(v_beg + u * RN_WAVE + lane, last)]);
            int64_t v0 = v_beg;
            for (int64_t gi = 0; gi < groups; ++gi, v0 += SCAN_PF * RN_WAVE) {
                rn::u32x4 qn[SCAN_PF];
#pragma unroll
                for (int u = 0; u < SCAN_PF; ++u) qn[u] = __builtin_nontemporal_load(&src[min(v0 + (SCAN_PF + u) * RN_WAVE + lane, last)]);
#pragma unroll
                for (int u = 0; u < SCAN_PF; ++u) do_vec(q[u], v0 + u * RN_WAVE + lane);
#pragma unroll
                for (int u = 0; u < SCAN_PF; ++u) q[u] = qn[u];
            }
            for (int64_t vb = v0; vb < v_end; vb += RN_WAVE) {       // leftover iterations (wave-uniform trip count)
                const int64_t v = vb + lane;
                const rn::u32x4 zero4 = {0u, 0u, 0u, 0u};
                rn::u32x4 raw = (v < v_end) ? src[v] : zero4;
                if (v >= v_end) {                                     // lanes past the end must not produce candidates
                    float lowv[VEC];
#pragma unroll
                    for (int j = 0; j < VEC; ++j) lowv[j] = -INFINITY;
                    raw = D::pack(lowv);
                }
                do_vec(raw, min(v, last));
            }
        }
        // ragged tail of the level (< VEC elements): the wave whose range ends the level (wave 0 if it has no whole vector)
        const bool owns_tail = lv.nvec ? (w_beg < lv.voff + lv.nvec && lv.voff + lv.nvec <= w_end) : (gwave == 0);
        if (owns_tail && lv.nvec * VEC < lv.N) {
            const int64_t e = lv.nvec * VEC + lane;
            uint64_t key = 0; int img = 0;
            bool c = false;
            if (lane < VEC && e < lv.N) {
                const float x = D::ld(lv.cls, e);
                c = (x > a.pre_thr) && (scan_score(x) > a.score_thr);
                if (c) scan_key(a, lv.A_l, lv.base, scan_score(x), e, key, img);
            }
            append(c, key, img);
        }
    }
    if (qfill) drain();
    if (fill) scan_flush(sl, fill, a, lane, shard);
}

// ---- candidates -> per-(image, class) segments ----------------------------------------------
// Everything between the score scan and NMS, as two launches of SEG_GROUPS (16) workgroups per image -- one workgroup per
// image (round 1) kept 16 of the 256 CUs busy for 59 us on this latency-bound chain of dependent loads:
//  pass 1  (seg_count_kernel) each workgroup takes 1/16 of the image's candidate list: decode + clip the box of every
//          candidate's anchor (K4's arithmetic, decode_one), apply remove_small_boxes (models.py:203; boxes are per
//          anchor, so the size test commutes with the per-class loop), count survivors per class in LDS, mark the rest
//          dead, publish the class histogram of its slice;
//  pass 2  (seg_scatter_kernel) class totals over the 16 histograms -> segment offsets (seg_start / seg_len, what NMS
//          consumes); fill pointer of (workgroup, class) = segment start + counts of the workgroups before it; scatter
//          the slice's survivors (LDS counters hand out the slots).
// Only ~1e-3 of the anchors are ever decoded; candidates of one anchor in several classes write the
// same box.  No global atomics: the per-class counters of an image live in 3 cache lines, and
// ~1e4 global atomics on them serialise in L2 (measured 25 us per pass).
struct FilterLevel { const void *box; int64_t A_l, base; };
struct SegArgs {
    int L;
    FilterLevel lv[RN_MAX_LEVELS];
    const rn::f32x4 *anchors;
    int64_t anchor_bstride4;
    const int32_t *image_hw;
    RegW rw;
    int64_t A, C;
    int32_t K;
    float min_box;
    int32_t S;                   // shards of the candidate list (see ScanArgs)
    int64_t Cs;
    uint64_t *cand;              // [B][S][Cs] candidates (dead ones are overwritten with DEAD_KEY)
    const int32_t *cand_count;   // [B][S]
    rn::f32x4 *boxes;            // [B][A] decoded boxes, written only at candidate anchors
    uint64_t *seg;               // [B][C] (inv score << 32 | anchor), grouped by class
    int64_t *seg_start;          // [B][K]
    int32_t *seg_len;            // [B][K]
    int32_t *out_status;         // [B] 1 = more candidates than C
    int32_t *hist;               // [B][SEG_GROUPS][K] per-workgroup class counts (pass 1 -> pass 2)
};
constexpr uint64_t DEAD_KEY = ~0ull;      // not a valid key: its score field would be the smallest float
constexpr int SEG_THREADS = 256;
constexpr int SEG_GROUPS = 16;            // workgroups per image: each takes 1/16 of the image's candidate list
constexpr int SEG_UNROLL = 4;             // candidates per thread in flight (independent load chains)
constexpr int SEG_MAXK = 4096;

// Shared prologue of the two kernels: shard fills -> exclusive prefix (s_shard[0..S]), overflow flag, and this
// workgroup's slice [lo, hi) of the image's n candidates.
struct SegSlice { int64_t n, lo, hi; int over; };
__device__ __forceinline__ SegSlice seg_slice(const SegArgs &a, const int b, const int g, int *s_shard, int *s_over)
{
    const int t = threadIdx.x;
    if (t == 0) *s_over = 0;
    __syncthreads();
    if (t < RN_WAVE) {                                        // wave 0: shard fills -> prefix (S <= 128: two rounds)
        int run = 0;
        for (int s0 = 0; s0 < a.S; s0 += RN_WAVE) {
            const int sh = s0 + t;
            int c = sh < a.S ? a.cand_count[(int64_t)b * a.S + sh] : 0;
            if ((int64_t)c > a.Cs) { c = (int)a.Cs; *s_over = 1; }
            int incl = c;
#pragma unroll
            for (int dd = 1; dd < RN_WAVE; dd <<= 1) { const int up = __shfl_up(incl, dd, RN_WAVE); if (t >= dd) incl += up; }
            if (sh < a.S) s_shard[sh] = run + (incl - c);
            run += __shfl(incl, RN_WAVE - 1, RN_WAVE);
        }
        if (t == 0) s_shard[a.S] = run;
    }
    __syncthreads();
    SegSlice r;
    r.n = s_shard[a.S];
    r.lo = r.n * g / SEG_GROUPS;
    r.hi = r.n * (g + 1) / SEG_GROUPS;
    r.over = *s_over;
    return r;
}
// candidate i of the image -> its slot in the sharded list
__device__ __forceinline__ int64_t seg_slot_of(const SegArgs &a, const int *s_shard, const int64_t i)
{
    int lo = 0, hi = a.S;                                      // shard sh with s_shard[sh] <= i < s_shard[sh + 1]
    while (hi - lo > 1) { const int md = (lo + hi) >> 1; if (s_shard[md] <= i) lo = md; else hi = md; }
    return (int64_t)lo * a.Cs + (i - s_shard[lo]);
}

// Pass 1 (grid: SEG_GROUPS x B): decode + clip + remove_small_boxes of this workgroup's slice of the candidates, boxes of the
// survivors, DEAD_KEY over the rest, and the slice's per-class survivor counts -> hist[b][g][K].
template <int DT>
__global__ __launch_bounds__(SEG_THREADS) void seg_count_kernel(const SegArgs a)
{
    __shared__ int s_cnt[SEG_MAXK];
    __shared__ int s_shard[SCAN_SHARDS + 1];
    __shared__ int s_over;
    const int g = blockIdx.x, b = blockIdx.y, t = threadIdx.x;
    for (int k = t; k < a.K; k += SEG_THREADS) s_cnt[k] = 0;
    const SegSlice sl = seg_slice(a, b, g, s_shard, &s_over);
    float hh = 0.0f, ww = 0.0f;
    if (a.image_hw) { hh = (float)a.image_hw[2 * b]; ww = (float)a.image_hw[2 * b + 1]; }
    uint64_t *cand = a.cand + (int64_t)b * a.S * a.Cs;

    for (int64_t i0 = sl.lo; i0 < sl.hi; i0 += SEG_UNROLL * SEG_THREADS) {
        uint64_t key[SEG_UNROLL];
        uint32_t anchor[SEG_UNROLL], k[SEG_UNROLL];
        float d[SEG_UNROLL][4];
        rn::f32x4 an[SEG_UNROLL];
        int64_t slot[SEG_UNROLL];
#pragma unroll
        for (int u = 0; u < SEG_UNROLL; ++u) {
            slot[u] = seg_slot_of(a, s_shard, min(i0 + u * SEG_THREADS + t, sl.hi - 1));
            key[u] = cand[slot[u]];
            // Lanes past the slice re-read its LAST candidate (clamped index) only to keep the loads below unconditional -- but the
            // lane that OWNS that candidate may belong to another wave of this workgroup, which may already have marked it dead
            // (DEAD_KEY: "anchor" 2^32 / K) by the time this wave loads it: the delta / anchor loads below then ran 760 MB past their
            // tensors.  Round 4, found by the end-to-end predict line on a random-init R101 (half of all candidates dead): a lane
            // without a candidate of its own loads anchor 0.
            if (i0 + u * SEG_THREADS + t >= sl.hi) key[u] = 0;
        }
#pragma unroll
        for (int u = 0; u < SEG_UNROLL; ++u) {
            const uint32_t ak = (uint32_t)key[u];
            anchor[u] = ak / (uint32_t)a.K; k[u] = ak - anchor[u] * (uint32_t)a.K;
            const void *bp = a.lv[0].box;
            int64_t A_l = a.lv[0].A_l, base = 0;
#pragma unroll
            for (int l = 1; l < RN_MAX_LEVELS; ++l)
                if (l < a.L && (int64_t)anchor[u] >= a.lv[l].base) { bp = a.lv[l].box; A_l = a.lv[l].A_l; base = a.lv[l].base; }
            delta4<DT>::ld(bp, (int64_t)b * A_l + ((int64_t)anchor[u] - base), d[u]);
            an[u] = a.anchors[(int64_t)b * a.anchor_bstride4 + anchor[u]];
        }
#pragma unroll
        for (int u = 0; u < SEG_UNROLL; ++u) {
            const int64_t i = i0 + u * SEG_THREADS + t;
            if (i >= sl.hi) continue;
            rn::f32x4 o = decode_one(d[u], an[u], a.rw);
            if (a.image_hw) {
                o.x = clampf(o.x, 0.0f, ww); o.z = clampf(o.z, 0.0f, ww);
                o.y = clampf(o.y, 0.0f, hh); o.w = clampf(o.w, 0.0f, hh);
            }
            if ((o.z - o.x) >= a.min_box && (o.w - o.y) >= a.min_box) {
                a.boxes[(int64_t)b * a.A + anchor[u]] = o;    // (several candidates of one anchor write the same box)
                atomicAdd(&s_cnt[k[u]], 1);
            } else {
                cand[slot[u]] = DEAD_KEY;
            }
        }
    }
    __syncthreads();
    int32_t *hist = a.hist + ((int64_t)b * SEG_GROUPS + g) * a.K;
    for (int k = t; k < a.K; k += SEG_THREADS) hist[k] = s_cnt[k];
}

// Pass 2 (same grid, next launch): class totals over the image's workgroups -> segment offsets (seg_start / seg_len, what
// NMS consumes); this workgroup's fill pointer of class k = segment start + the counts of the workgroups before it;
// scatter of the slice's survivors (LDS counters hand out the slots).  The order inside a class segment is arbitrary:
// NMS sorts by (score, anchor).
template <int DT>
__global__ __launch_bounds__(SEG_THREADS) void seg_scatter_kernel(const SegArgs a)
{
    __shared__ int s_cnt[SEG_MAXK];                           // class totals, then fill pointers
    __shared__ int s_before[SEG_MAXK];                        // survivors of class k in the workgroups before this one
    __shared__ int s_shard[SCAN_SHARDS + 1];
    __shared__ int s_over;
    const int g = blockIdx.x, b = blockIdx.y, t = threadIdx.x;
    const SegSlice sl = seg_slice(a, b, g, s_shard, &s_over);
    const int32_t *hist = a.hist + (int64_t)b * SEG_GROUPS * a.K;
    for (int k = t; k < a.K; k += SEG_THREADS) {
        int tot = 0, before = 0;
#pragma unroll
        for (int gg = 0; gg < SEG_GROUPS; ++gg) { const int c = hist[gg * a.K + k]; tot += c; before += gg < g ? c : 0; }
        s_cnt[k] = tot; s_before[k] = before;
    }
    __syncthreads();
    if (t < RN_WAVE) {                                        // wave 0: exclusive scan of the class totals
        int run = 0;
        for (int k0 = 0; k0 < a.K; k0 += RN_WAVE) {
            const int kk = k0 + t;
            const int c = kk < a.K ? s_cnt[kk] : 0;
            int incl = c;
#pragma unroll
            for (int dd = 1; dd < RN_WAVE; dd <<= 1) { const int up = __shfl_up(incl, dd, RN_WAVE); if (t >= dd) incl += up; }
            if (kk < a.K) {
                if (g == 0) {
                    a.seg_start[(int64_t)b * a.K + kk] = (int64_t)b * a.C + run + (incl - c);
                    a.seg_len[(int64_t)b * a.K + kk] = c;
                }
                s_cnt[kk] = run + (incl - c) + s_before[kk];  // this workgroup's fill pointer
            }
            run += __shfl(incl, RN_WAVE - 1, RN_WAVE);
        }
        if (t == 0 && g == 0) a.out_status[b] = sl.over;
    }
    __syncthreads();
    const uint64_t *cand = a.cand + (int64_t)b * a.S * a.Cs;
    uint64_t *seg = a.seg + (int64_t)b * a.C;
    for (int64_t i0 = sl.lo; i0 < sl.hi; i0 += SEG_UNROLL * SEG_THREADS) {
        uint64_t key[SEG_UNROLL];
#pragma unroll
        for (int u = 0; u < SEG_UNROLL; ++u) key[u] = cand[seg_slot_of(a, s_shard, min(i0 + u * SEG_THREADS + t, sl.hi - 1))];
#pragma unroll
        for (int u = 0; u < SEG_UNROLL; ++u) {
            if (i0 + u * SEG_THREADS + t >= sl.hi || key[u] == DEAD_KEY) continue;
            const uint32_t ak = (uint32_t)key[u];
            const uint32_t anchor = ak / (uint32_t)a.K, k = ak - anchor * (uint32_t)a.K;
            const int pos = atomicAdd(&s_cnt[k], 1);
            seg[pos] = (key[u] & 0xffffffff00000000ull) | anchor;
        }
    }
}

// ---- K7 class-major merge + top max_det (models.py:222-240) ----------------------------
// Final order = stable sort by score desc of the class-major concatenation, i.e. ascending
// (inv score, class, in-class rank); in-class rank order == (inv score, anchor) order, so the
// key (inv score << 32 | class*A + anchor) reproduces it.  Only the first max_det survivors of
// each class can reach the global top max_det.
constexpr int TOPK_THREADS = 1024;
constexpr int TOPK_CHUNK = 1024;          // keys sorted together (one per thread)
constexpr int TOPK_GROUP = 4;             // chunks resident at once (one key per thread per chunk, in registers)
constexpr int TOPK_SURV = 2048;           // survivors (top max_det of every chunk so far)
constexpr int TOPK_MAXK = 4096;

__device__ __forceinline__ uint64_t shfl_xor_u64(const uint64_t v, const int j)
{
    const uint32_t lo = __shfl_xor((uint32_t)v, j, RN_WAVE), hi = __shfl_xor((uint32_t)(v >> 32), j, RN_WAVE);
    return ((uint64_t)hi << 32) | lo;
}

// Ascending bitonic sort of NPT*1024 keys held in registers, key r of thread t = element r*1024 + t.
// JOINT = false: NPT independent 1024-element arrays; JOINT = true: one NPT*1024-element array.
// Exchange distance j < 64 stays inside a wave (shuffles, no LDS, no barrier); 64 <= j < 1024 goes
// through LDS (two block barriers); j >= 1024 pairs two registers of the same thread.
template <int NPT, bool JOINT>
__device__ __forceinline__ void bitonic_regs(uint64_t (&k)[NPT], uint64_t *lds)
{
    const int t = threadIdx.x;
    constexpr int N = JOINT ? NPT * TOPK_CHUNK : TOPK_CHUNK;
    for (int kk = 2; kk <= N; kk <<= 1) {
        for (int j = kk >> 1; j > 0; j >>= 1) {
            if (j >= TOPK_CHUNK) {                      // JOINT only: partner is another register of this thread
#pragma unroll
                for (int r = 0; r < NPT; ++r) {
                    const int pr = r ^ (j / TOPK_CHUNK);
                    if (pr > r) {
                        const bool up = (((r * TOPK_CHUNK + t) & kk) == 0);
                        const uint64_t x = k[r], y = k[pr];
                        if ((x > y) == up) { k[r] = y; k[pr] = x; }
                    }
                }
            } else if (j >= RN_WAVE) {
#pragma unroll
                for (int r = 0; r < NPT; ++r) lds[r * TOPK_CHUNK + t] = k[r];
                __syncthreads();
#pragma unroll
                for (int r = 0; r < NPT; ++r) {
                    const int e = (JOINT ? r * TOPK_CHUNK : 0) + t;
                    const uint64_t y = lds[r * TOPK_CHUNK + (t ^ j)];
                    const bool keep_min = (((t & j) == 0) == ((e & kk) == 0));
                    k[r] = keep_min ? (k[r] < y ? k[r] : y) : (k[r] > y ? k[r] : y);
                }
                __syncthreads();
            } else {
#pragma unroll
                for (int r = 0; r < NPT; ++r) {
                    const int e = (JOINT ? r * TOPK_CHUNK : 0) + t;
                    const uint64_t y = shfl_xor_u64(k[r], j);
                    const bool keep_min = (((t & j) == 0) == ((e & kk) == 0));
                    k[r] = keep_min ? (k[r] < y ? k[r] : y) : (k[r] > y ? k[r] : y);
                }
            }
        }
    }
}

// Ascending bitonic sort of the first n (power of two, <= 1024) of one key per thread; all threads call.
__device__ __forceinline__ void bitonic_one(uint64_t &k, uint64_t *lds, const int n)
{
    const int t = threadIdx.x;
    for (int kk = 2; kk <= n; kk <<= 1) {
        for (int j = kk >> 1; j > 0; j >>= 1) {
            uint64_t y;
            if (j >= RN_WAVE) {
                lds[t] = k;
                __syncthreads();
                y = lds[t ^ j];
                __syncthreads();
            } else {
                y = shfl_xor_u64(k, j);
            }
            const bool keep_min = (((t & j) == 0) == ((t & kk) == 0));
            k = keep_min ? (k < y ? k : y) : (k > y ? k : y);
        }
    }
}

constexpr int TOPK_FAST_PT = 12;          // keys per thread the histogram path keeps in registers
constexpr int TOPK_BINS = 2048;

// One workgroup per image.  Counts / offsets of all classes are fetched in parallel (one latency),
// the first max_det survivors of every class are gathered by flat index, sorted 1024 at a time, and
// only the best max_det of every chunk survive to the final sort: O(M log^2 1024) instead of
// O(M log^2 M), for any number of candidates M.
// Usual case (M <= 12288, e.g. 90 classes x 100): no sort of the M keys at all -- a 2048-bin
// histogram of the score field finds the bin holding the max_det-th best key, the keys up to that bin
// (about max_det of them) are compacted and only those are sorted.
__global__ __launch_bounds__(TOPK_THREADS) void topk_kernel(const uint64_t *__restrict__ kept, const int64_t *__restrict__ seg_start,
                                                            const int32_t *__restrict__ kept_count, const rn::f32x4 *__restrict__ boxes,
                                                            const int K, const int64_t A, const int max_det,
                                                            rn::f32x4 *__restrict__ out_boxes, float *__restrict__ out_scores,
                                                            int64_t *__restrict__ out_labels, int32_t *__restrict__ out_count)
{
    __shared__ uint64_t s_key[TOPK_GROUP * TOPK_CHUNK];      // exchange buffer of the sorts
    __shared__ uint64_t s_surv[TOPK_SURV];
    __shared__ int s_pre[TOPK_MAXK + 1];       // exclusive prefix of min(kept_count, max_det) over classes
    const int b = blockIdx.x;
    const int t = threadIdx.x;

    for (int k = t; k < K; k += TOPK_THREADS) s_pre[k + 1] = min(kept_count[(int64_t)b * K + k], max_det);
    if (t == 0) s_pre[0] = 0;
    __syncthreads();
    if (t < RN_WAVE) {                                       // wave 0: inclusive scan, 64 classes per step (a serial loop of one thread cost K LDS round trips)
        int run = 0;
        for (int k0 = 1; k0 <= K; k0 += RN_WAVE) {
            const int k = k0 + t;
            int incl = k <= K ? s_pre[k] : 0;
#pragma unroll
            for (int dd = 1; dd < RN_WAVE; dd <<= 1) { const int up = __shfl_up(incl, dd, RN_WAVE); if (t >= dd) incl += up; }
            if (k <= K) s_pre[k] = run + incl;
            run += __shfl(incl, RN_WAVE - 1, RN_WAVE);
        }
    }
    for (int i = t; i < TOPK_SURV; i += TOPK_THREADS) s_surv[i] = ~0ull;
    __syncthreads();
    const int M = s_pre[K];
    int n_surv = 0;
    uint64_t win = ~0ull;                  // final key of output row t
    bool done = false;                     // block-uniform

    if (M <= TOPK_FAST_PT * TOPK_THREADS) {
        int *s_hist = (int *)s_key;                            // [TOPK_BINS]
        int *s_wave = s_hist + TOPK_BINS;                      // [16] wave totals | [32..] min, max, bin, nsel, fill
        uint64_t kr[TOPK_FAST_PT];
        uint32_t lo = 0xffffffffu, hi = 0u;
#pragma unroll
        for (int r = 0; r < TOPK_FAST_PT; ++r) {
            kr[r] = ~0ull;
            const int f = r * TOPK_THREADS + t;
            if (f < M) {
                int a_ = 0, b_ = K;                                 // class k with s_pre[k] <= f < s_pre[k+1]
                while (b_ - a_ > 1) { const int md = (a_ + b_) >> 1; if (s_pre[md] <= f) a_ = md; else b_ = md; }
                const uint64_t key = kept[seg_start[(int64_t)b * K + a_] + (f - s_pre[a_])];
                kr[r] = (key & 0xffffffff00000000ull) | (uint32_t)((uint32_t)a_ * (uint32_t)A + (uint32_t)key);
                const uint32_t h = (uint32_t)(key >> 32);
                lo = min(lo, h); hi = max(hi, h);
            }
        }
        for (int i = t; i < TOPK_BINS + 64; i += TOPK_THREADS) s_hist[i] = (i == TOPK_BINS + 32) ? (int)0x7fffffff : 0;
        __syncthreads();
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) { lo = min(lo, (uint32_t)__shfl_xor((int)lo, d, RN_WAVE)); hi = max(hi, (uint32_t)__shfl_xor((int)hi, d, RN_WAVE)); }
        if ((t & (RN_WAVE - 1)) == 0 && lo <= hi) {            // min/max of the score field (non-negative after >> 1)
            atomicMin(&s_wave[32], (int)(lo >> 1));
            atomicMax(&s_wave[33], (int)(hi >> 1));
        }
        __syncthreads();
        const uint32_t lo1 = (uint32_t)s_wave[32], range = (uint32_t)s_wave[33] - lo1;     // in units of 2 ulps of the field
        const int shift = range >= (uint32_t)TOPK_BINS ? (32 - __clz((int)range)) - 11 : 0;
#pragma unroll
        for (int r = 0; r < TOPK_FAST_PT; ++r)
            if (kr[r] != ~0ull) atomicAdd(&s_hist[(((uint32_t)(kr[r] >> 33)) - lo1) >> shift], 1);
        __syncthreads();
        // inclusive scan of the bins: two per thread, wave shuffles, 16 wave totals
        const int h0 = s_hist[2 * t], h1 = s_hist[2 * t + 1];
        int incl = h0 + h1;
        const int lane = t & (RN_WAVE - 1), wv = t >> 6;
#pragma unroll
        for (int d = 1; d < RN_WAVE; d <<= 1) { const int up = __shfl_up(incl, d, RN_WAVE); if (lane >= d) incl += up; }
        if (lane == RN_WAVE - 1) s_wave[wv] = incl;
        __syncthreads();
        int before = 0;
        for (int w = 0; w < wv; ++w) before += s_wave[w];
        const int c1 = before + incl, c0 = c1 - h1, cm = c0 - h0;      // cumulative after bin 2t+1, after 2t, before 2t
        const int need = min(M, max_det);
        if (need > 0) {
            if (cm < need && c0 >= need) { s_wave[34] = 2 * t; s_wave[35] = c0; }
            else if (c0 < need && c1 >= need) { s_wave[34] = 2 * t + 1; s_wave[35] = c1; }
        }
        __syncthreads();
        const int tbin = s_wave[34], nsel = s_wave[35];
        if (nsel <= TOPK_CHUNK) {                               // else: too many ties in one bin, take the general path
            if (need > 0) {
#pragma unroll
                for (int r = 0; r < TOPK_FAST_PT; ++r)
                    if (kr[r] != ~0ull && (int)((((uint32_t)(kr[r] >> 33)) - lo1) >> shift) <= tbin)
                        s_surv[atomicAdd(&s_wave[36], 1)] = kr[r];
            }
            __syncthreads();
            win = t < nsel ? s_surv[t] : ~0ull;
            int npow = RN_WAVE;
            while (npow < nsel) npow <<= 1;
            __syncthreads();                                    // s_key is reused as the exchange buffer
            bitonic_one(win, s_key, npow);
            done = true;
        } else {
            __syncthreads();
            for (int i = t; i < TOPK_SURV; i += TOPK_THREADS) s_surv[i] = ~0ull;
            __syncthreads();
        }
    }
    if (!done) {

    // chunks per round: (1 + group) * max_det survivors must fit in s_surv (max_det <= 1024 -> group >= 1)
    const int group = max(1, min(TOPK_GROUP, TOPK_SURV / max_det - 1));
    for (int g0 = 0; g0 < M; g0 += group * TOPK_CHUNK) {
        const int n_here = min(group * TOPK_CHUNK, M - g0);
        const int n_chunks = (n_here + TOPK_CHUNK - 1) / TOPK_CHUNK;
        uint64_t kr[TOPK_GROUP];
#pragma unroll
        for (int r = 0; r < TOPK_GROUP; ++r) {
            kr[r] = ~0ull;
            const int i = r * TOPK_CHUNK + t;
            if (i < n_here) {
                const int f = g0 + i;
                int lo = 0, hi = K;                                 // class k with s_pre[k] <= f < s_pre[k+1]
                while (hi - lo > 1) { const int md = (lo + hi) >> 1; if (s_pre[md] <= f) lo = md; else hi = md; }
                const uint64_t key = kept[seg_start[(int64_t)b * K + lo] + (f - s_pre[lo])];
                kr[r] = (key & 0xffffffff00000000ull) | (uint32_t)((uint32_t)lo * (uint32_t)A + (uint32_t)key);
            }
        }
        bitonic_regs<TOPK_GROUP, false>(kr, s_key);
        if (n_surv + n_chunks * max_det > TOPK_SURV) {              // uniform: compact the survivors first
            uint64_t sr[2] = {s_surv[t], s_surv[TOPK_CHUNK + t]};
            __syncthreads();
            bitonic_regs<2, true>(sr, s_key);
            s_surv[t] = (t < max_det) ? sr[0] : ~0ull;
            s_surv[TOPK_CHUNK + t] = ~0ull;
            n_surv = min(n_surv, max_det);
            __syncthreads();
        }
        if (t < max_det) {
#pragma unroll
            for (int r = 0; r < TOPK_GROUP; ++r)
                if (r < n_chunks) s_surv[n_surv + r * max_det + t] = kr[r];      // padded slots carry ~0 and sort to the end
        }
        n_surv += n_chunks * max_det;
        __syncthreads();
    }
    uint64_t sr[2] = {s_surv[t], s_surv[TOPK_CHUNK + t]};
    __syncthreads();
    bitonic_regs<2, true>(sr, s_key);
    win = sr[0];                                                 // max_det <= 1024: the winners are elements 0..1023
    }

    const int nout = min(M, max_det);
    if (t < max_det) {
        const int64_t o = (int64_t)b * max_det + t;
        if (t < nout) {
            const uint64_t key = win;
            const uint32_t ka = (uint32_t)key;
            const uint32_t k = ka / (uint32_t)A, anchor = ka - k * (uint32_t)A;
            out_boxes[o] = boxes[(int64_t)b * A + anchor];
            out_scores[o] = rn::score_of((uint32_t)(key >> 32));
            out_labels[o] = (int64_t)k + 1;                       // models.py:230
        } else {
            const rn::f32x4 z = {0.f, 0.f, 0.f, 0.f};
            out_boxes[o] = z; out_scores[o] = 0.0f; out_labels[o] = 0;
        }
    }
    if (t == 0) out_count[b] = nout;
}

__global__ void zero_bytes4_kernel(uint32_t *__restrict__ p, const int64_t n)
{
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) p[i] = 0u;
}

size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

struct DetectWs {
    rn::f32x4 *boxes; uint64_t *cand, *seg; rn::f32x4 *sbox; uint8_t *supp;
    int32_t *cand_count, *kept_count, *seg_len, *hist; int64_t *seg_start;
    size_t zero_bytes, total;
};

DetectWs carve(void *base, int B, int64_t A, int K, int64_t C)
{
    DetectWs w;
    unsigned char *p = (unsigned char *)base;
    size_t off = 0;
    auto take = [&](size_t bytes) { unsigned char *q = p ? p + off : nullptr; off += al256(bytes); return q; };
    const size_t BK = (size_t)B * K, BC = (size_t)B * C;
    w.boxes = (rn::f32x4 *)take((size_t)B * A * 16);
    w.cand = (uint64_t *)take(BC * 8);
    w.seg = (uint64_t *)take(BC * 8);
    w.sbox = (rn::f32x4 *)take(BC * 16);
    w.supp = (uint8_t *)take(BC);
    // zeroed every call
    const size_t z0 = off;
    w.cand_count = (int32_t *)take((size_t)B * SCAN_SHARDS * 4);
    w.zero_bytes = off - z0;
    w.kept_count = (int32_t *)take(BK * 4);
    w.seg_len = (int32_t *)take(BK * 4);
    w.seg_start = (int64_t *)take(BK * 8);
    w.hist = (int32_t *)take(BK * 16 * 4);                    // SEG_GROUPS = 16 workgroups per image
    w.total = off;
    return w;
}

}  // namespace

RN_API int rn_decode_clip(const void *deltas, int dtype, int B, int64_t A, const float *anchors, int64_t anchor_bstride,
                          const int32_t *image_hw, const float reg_w[4], float *out, void *stream)
{
    if (!deltas || !anchors || !out || !reg_w || B <= 0 || A <= 0) return RN_EINVAL;
    const int64_t R = (int64_t)B * A;
    if (R >= ((int64_t)1 << 31)) return RN_EUNSUPPORTED;
    if (!rn::aligned(deltas, dtype == RN_F32 ? 16 : 8) || !rn::aligned(anchors, 16) || !rn::aligned(out, 16) ||
        (anchor_bstride & 3))
        return RN_EALIGN;
    RegW rw = {{reg_w[0], reg_w[1], reg_w[2], reg_w[3]}};
    return launch_decode(deltas, dtype, B, A, 0, A, anchors, anchor_bstride, image_hw, rw, out, (hipStream_t)stream);
}

RN_API size_t rn_detect_workspace_bytes(int B, int64_t A, int K, int64_t max_candidates)
{
    if (B <= 0 || A <= 0 || K <= 0 || max_candidates <= 0) return 0;
    return carve(nullptr, B, A, K, max_candidates).total;
}

RN_API int rn_detect_levels(const void *const *cls_levels, const void *const *box_levels, const int64_t *level_anchors, int L,
                            int dtype, int B, int K, const float *anchors, int64_t anchor_bstride, const int32_t *image_hw,
                            const rn_detect_params *params, int64_t max_candidates, float *out_boxes, float *out_scores,
                            int64_t *out_labels, int32_t *out_count, int32_t *out_status, void *workspace,
                            size_t workspace_bytes, void *stream)
{
    if (!cls_levels || !box_levels || !level_anchors || !anchors || !params || !out_boxes || !out_scores || !out_labels ||
        !out_count || !out_status || !workspace)
        return RN_EINVAL;
    if (L <= 0 || L > RN_MAX_LEVELS || B <= 0 || K <= 0 || max_candidates <= 0 || params->max_det <= 0) return RN_EINVAL;
    if (dtype != RN_F32 && dtype != RN_BF16 && dtype != RN_F16) return RN_EINVAL;
    int64_t A = 0;
    for (int l = 0; l < L; ++l) {
        if (!cls_levels[l] || !box_levels[l] || level_anchors[l] <= 0) return RN_EINVAL;
        A += level_anchors[l];
    }
    const int64_t R = (int64_t)B * A, C = max_candidates;
    // payloads are 32-bit: anchor*K + k and class*A + anchor
    if (R >= ((int64_t)1 << 31) || A * (int64_t)K >= ((int64_t)1 << 32) || K > 4096 || params->max_det > TOPK_THREADS ||
        (int64_t)B * C >= ((int64_t)1 << 40) || (int64_t)B * K >= ((int64_t)1 << 31))
        return RN_EUNSUPPORTED;
    if (workspace_bytes < rn_detect_workspace_bytes(B, A, K, C)) return RN_EWORKSPACE;
    if (!rn::aligned(workspace, 256) || !rn::aligned(out_boxes, 16) || !rn::aligned(anchors, 16) || (anchor_bstride & 3))
        return RN_EALIGN;
    for (int l = 0; l < L; ++l)
        if (!rn::aligned(cls_levels[l], 16) || !rn::aligned(box_levels[l], dtype == RN_F32 ? 16 : 8)) return RN_EALIGN;
    hipStream_t st = (hipStream_t)stream;
    DetectWs w = carve(workspace, B, A, K, C);
    const RegW rw = {{params->reg_w[0], params->reg_w[1], params->reg_w[2], params->reg_w[3]}};

    ScanArgs sa;
    const int vec = (dtype == RN_F32) ? 4 : 8;
    sa.L = L;
    int64_t base = 0, voff = 0;
    int rc = RN_OK;
    SegArgs fa;
    fa.L = L;
    for (int l = 0; l < L; ++l) {
        fa.lv[l].box = box_levels[l]; fa.lv[l].A_l = level_anchors[l]; fa.lv[l].base = base;
        ScanLevel &lv = sa.lv[l];
        lv.cls = cls_levels[l]; lv.A_l = level_anchors[l]; lv.base = base;
        lv.N = (int64_t)B * level_anchors[l] * K; lv.nvec = lv.N / vec; lv.voff = voff;
        base += level_anchors[l];
        voff += lv.nvec;
    }
    sa.total_vec = voff;
    hipLaunchKernelGGL(zero_bytes4_kernel, dim3(1), dim3(256), 0, st, (uint32_t *)w.cand_count, (int64_t)(w.zero_bytes / 4));   // (no hipMemsetAsync: match.hip, zero_i32_kernel)
    RN_LAUNCH_CHECK();

    for (int l = L; l < RN_MAX_LEVELS; ++l) fa.lv[l] = fa.lv[L - 1];
    fa.anchors = (const rn::f32x4 *)anchors; fa.anchor_bstride4 = anchor_bstride / 4; fa.image_hw = image_hw; fa.rw = rw;
    fa.A = A; fa.C = C; fa.K = K; fa.min_box = params->min_box;
    fa.cand = w.cand; fa.cand_count = w.cand_count; fa.boxes = w.boxes;
    fa.seg = w.seg; fa.seg_start = w.seg_start; fa.seg_len = w.seg_len; fa.out_status = out_status; fa.hist = w.hist;
    sa.A = A; sa.C = C; sa.K = K; sa.B = B;
    // shards of the per-image candidate list; the worst-case capacity (the caller's retry) is one list so
    // that "A*K always suffices" holds however the candidates are distributed over the waves
    sa.S = (C >= A * (int64_t)K || C / SCAN_SHARDS < 64) ? 1 : SCAN_SHARDS;
    sa.Cs = C / sa.S;
    fa.S = sa.S; fa.Cs = sa.Cs;
    sa.score_thr = params->score_thr;
    {   // logit-space pre-filter with a safety margin; the exact test is still sigmoid(x) > thr
        const double t = (double)params->score_thr;
        sa.pre_thr = (t > 0.0 && t < 1.0) ? (float)(log(t / (1.0 - t)) - 1e-2) : (t <= 0.0 ? -INFINITY : 40.0f);
        if (t >= 1.0) sa.pre_thr = INFINITY;
    }
    sa.cand = w.cand; sa.cand_count = w.cand_count;
    // resident-sized grid, even split of the 16-byte vectors over the waves (whole wave-iterations)
    int dev = 0, cus = 0, per_cu = 0;
    RN_HIP(hipGetDevice(&dev));
    RN_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const int64_t nvec = sa.total_vec;
    unsigned blocks = 1;
    auto size_grid = [&](auto kernel) -> int {
        RN_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, SCAN_THREADS, 0));
        const int64_t waves = (int64_t)(cus * per_cu > 0 ? cus * per_cu : 1) * SCAN_WAVES;
        int64_t vpw = (nvec + waves - 1) / waves;
        vpw = ((vpw + RN_WAVE - 1) / RN_WAVE) * RN_WAVE;
        if (vpw < RN_WAVE) vpw = RN_WAVE;
        sa.vec_per_wave = vpw;
        int64_t need = ((nvec + vpw - 1) / vpw + SCAN_WAVES - 1) / SCAN_WAVES;
        blocks = (unsigned)(need < 1 ? 1 : need);             // >= 1: wave 0 also owns the tails of vector-less levels
        return RN_OK;
    };
    switch (dtype) {
        case RN_F32: rc = size_grid(score_scan_kernel<RN_F32>); if (rc) return rc;
            hipLaunchKernelGGL((score_scan_kernel<RN_F32>), dim3(blocks), dim3(SCAN_THREADS), 0, st, sa); break;
        case RN_BF16: rc = size_grid(score_scan_kernel<RN_BF16>); if (rc) return rc;
            hipLaunchKernelGGL((score_scan_kernel<RN_BF16>), dim3(blocks), dim3(SCAN_THREADS), 0, st, sa); break;
        case RN_F16: rc = size_grid(score_scan_kernel<RN_F16>); if (rc) return rc;
            hipLaunchKernelGGL((score_scan_kernel<RN_F16>), dim3(blocks), dim3(SCAN_THREADS), 0, st, sa); break;
        default: return RN_EINVAL;
    }
    RN_LAUNCH_CHECK();
    {
        static_assert(SEG_GROUPS == 16, "carve() sizes the class histograms for 16 workgroups per image");
        const dim3 fg(SEG_GROUPS, (unsigned)B), fb(SEG_THREADS);
        switch (dtype) {
            case RN_F32: hipLaunchKernelGGL((seg_count_kernel<RN_F32>), fg, fb, 0, st, fa); hipLaunchKernelGGL((seg_scatter_kernel<RN_F32>), fg, fb, 0, st, fa); break;
            case RN_BF16: hipLaunchKernelGGL((seg_count_kernel<RN_BF16>), fg, fb, 0, st, fa); hipLaunchKernelGGL((seg_scatter_kernel<RN_BF16>), fg, fb, 0, st, fa); break;
            default: hipLaunchKernelGGL((seg_count_kernel<RN_F16>), fg, fb, 0, st, fa); hipLaunchKernelGGL((seg_scatter_kernel<RN_F16>), fg, fb, 0, st, fa); break;
        }
        RN_LAUNCH_CHECK();
    }
    rn::NmsLaunch na;
    na.keys = w.seg; na.kept = w.cand; na.keep_idx = nullptr; na.boxes = w.boxes;
    na.seg_start = w.seg_start; na.seg_len = w.seg_len; na.kept_count = w.kept_count;
    na.scratch_boxes = w.sbox; na.scratch_supp = w.supp;
    na.S = B * K; na.box_mode = 1; na.K = K; na.A = A; na.iou_thr = params->nms_thr;
    // only the first max_det survivors of a class can reach the image's top max_det (the merge below reads min(kept_count, max_det) of
    // them): a long segment's greedy scan stops once it has kept that many -- the reference (no pre-NMS top-k, models.py:193-219) runs
    // O(n^2) over ~7 k boxes per class in SURVEY 8d's stress regime; the detections are the same rows
    na.max_keep = params->max_det;
    rc = rn::launch_nms(na, st);
    if (rc != RN_OK) return rc;
    hipLaunchKernelGGL(topk_kernel, dim3((unsigned)B), dim3(TOPK_THREADS), 0, st, w.cand, w.seg_start, w.kept_count, w.boxes,
                       K, A, params->max_det, (rn::f32x4 *)out_boxes, out_scores, out_labels, out_count);
    RN_LAUNCH_CHECK();
    return RN_OK;
}

RN_API int rn_detect(const void *cls, const void *deltas, int dtype, int B, int64_t A, int K, const float *anchors,
                     int64_t anchor_bstride, const int32_t *image_hw, const rn_detect_params *params,
                     int64_t max_candidates, float *out_boxes, float *out_scores, int64_t *out_labels,
                     int32_t *out_count, int32_t *out_status, void *workspace, size_t workspace_bytes, void *stream)
{
    if (!cls || !deltas || A <= 0) return RN_EINVAL;
    const void *c1[1] = {cls}, *d1[1] = {deltas};
    const int64_t a1[1] = {A};
    return rn_detect_levels(c1, d1, a1, 1, dtype, B, K, anchors, anchor_bstride, image_hw, params, max_candidates, out_boxes,
                            out_scores, out_labels, out_count, out_status, workspace, workspace_bytes, stream);
}
