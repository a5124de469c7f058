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
// candidate when sigmoid(x) > score_thr (models.py:196) and its decoded box passes
// remove_small_boxes (models.py:203; boxes are per anchor, so the size test commutes with the
// per-class loop).  Candidates (~4e-4 of the elements at the reference's prior) are staged in a
// wave-private LDS list and flushed with ONE global atomic per (wave, image): no block barrier
// anywhere in the stream.
constexpr int SCAN_THREADS = 256;
constexpr int SCAN_WAVES = SCAN_THREADS / RN_WAVE;
constexpr int SCAN_CAP = 128;            // wave-private list entries (>= 64: one ballot's worth always fits after a flush)
constexpr int SCAN_PF = 2;

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
    const rn::f32x4 *boxes;
    int64_t A, total_vec, vec_per_wave, C;
    int32_t K, B;
    float score_thr, pre_thr, min_box;
    uint64_t *cand;          // [B][C]  (inv_ordered(score) << 32) | (anchor*K + k)
    int32_t *cand_count;     // [B]
    int32_t *seg_count;      // [B][K]
};

struct ScanList { uint64_t key[SCAN_CAP]; int img[SCAN_CAP]; };

// wave-level flush: one atomic per distinct image in the list
__device__ __forceinline__ void scan_flush(ScanList &sl, const int fill, const ScanArgs &a, const int lane)
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
        if (lane == 0) base = atomicAdd(&a.cand_count[b], n);
        base = __shfl(base, 0, RN_WAVE);
        const unsigned long long lt = (1ull << lane) - 1ull;
        if (m0) {
            const int64_t pos = (int64_t)base + __popcll(mk0 & lt);
            if (pos < a.C) {
                a.cand[(int64_t)b * a.C + pos] = k0;
                atomicAdd(&a.seg_count[(int64_t)b * a.K + (int)((uint32_t)k0 % (uint32_t)a.K)], 1);
            }
            done0 = true;
        }
        if (m1) {
            const int64_t pos = (int64_t)base + n0 + __popcll(mk1 & lt);
            if (pos < a.C) {
                a.cand[(int64_t)b * a.C + pos] = k1;
                atomicAdd(&a.seg_count[(int64_t)b * a.K + (int)((uint32_t)k1 % (uint32_t)a.K)], 1);
            }
            done1 = true;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// exact candidate test of one element (rare path)
__device__ __forceinline__ bool scan_test(const ScanArgs &a, const ScanLevel &lv, const float x, const int64_t e, uint64_t &key,
                                          int &img)
{
    const float s = 1.0f / (1.0f + expf(-x));                 // models.py:170
    if (!(s > a.score_thr)) return false;
    const int64_t r = e / a.K;                                // row of this level's [B*A_l][K]
    const uint32_t k = (uint32_t)(e - r * a.K);
    const int b = (int)((uint32_t)r / (uint32_t)lv.A_l);
    const uint32_t anchor = (uint32_t)(lv.base + (r - (int64_t)b * lv.A_l));
    const rn::f32x4 bx = a.boxes[(int64_t)b * a.A + anchor];
    if (!((bx.z - bx.x) >= a.min_box && (bx.w - bx.y) >= a.min_box)) return false;
    key = ((uint64_t)rn::inv_ordered(s) << 32) | (uint32_t)(anchor * (uint32_t)a.K + k);
    img = b;
    return true;
}

template <int DT>
__global__ __launch_bounds__(SCAN_THREADS) void score_scan_kernel(const ScanArgs a)
{
    typedef rn::dt<DT> D;
    constexpr int VEC = D::VEC;
    __shared__ ScanList s_list[SCAN_WAVES];
    const int lane = threadIdx.x & (RN_WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    ScanList &sl = s_list[wave];
    const int64_t gwave = (int64_t)blockIdx.x * SCAN_WAVES + wave;
    const int64_t w_beg = gwave * a.vec_per_wave;
    const int64_t w_end = min(w_beg + a.vec_per_wave, a.total_vec);
    int fill = 0;                                              // wave-uniform

    // append the candidates selected by `pred` (at most one per lane): ballot, maybe flush, write
    auto append = [&](const bool pred, const uint64_t key, const int img) {
        const unsigned long long mk = __ballot(pred);
        if (!mk) return;
        const int n = __popcll(mk);
        if (fill + n > SCAN_CAP) { scan_flush(sl, fill, a, lane); fill = 0; }
        if (pred) {
            const int pos = fill + __popcll(mk & ((1ull << lane) - 1ull));
            sl.key[pos] = key; sl.img[pos] = img;
        }
        fill += n;
    };

    for (int l = 0; l < a.L; ++l) {                            // wave-uniform: the levels this wave's range touches
        const ScanLevel &lv = a.lv[l];
        const int64_t v_beg = max(w_beg, lv.voff) - lv.voff, v_end = min(w_end, lv.voff + lv.nvec) - lv.voff;
        const rn::u32x4 *src = (const rn::u32x4 *)lv.cls;
        auto do_vec = [&](const rn::u32x4 raw, const int64_t v) {
            float x[VEC];
            D::unpack(raw, x);
            bool any_lane = false;
#pragma unroll
            for (int j = 0; j < VEC; ++j) any_lane |= (x[j] > a.pre_thr);
            if (__any(any_lane)) {                            // rare: ~0.2 candidates per 512 elements
#pragma unroll 1
                for (int j = 0; j < VEC; ++j) {
                    uint64_t key = 0; int img = 0;
                    const bool c = (x[j] > a.pre_thr) && scan_test(a, lv, x[j], v * VEC + j, key, img);
                    append(c, key, img);
                }
            }
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
                c = (x > a.pre_thr) && scan_test(a, lv, x, e, key, img);
            }
            append(c, key, img);
        }
    }
    if (fill) scan_flush(sl, fill, a, lane);
}

// ---- per-image segment offsets -----------------------------------------------------
__global__ __launch_bounds__(64) void seg_offsets_kernel(const int32_t *__restrict__ cand_count, const int32_t *__restrict__ seg_count,
                                                         const int K, const int64_t C, int64_t *__restrict__ seg_start,
                                                         int32_t *__restrict__ seg_len, int32_t *__restrict__ out_status)
{
    const int b = blockIdx.x;
    if (threadIdx.x != 0) return;
    int64_t run = (int64_t)b * C;
    for (int k = 0; k < K; ++k) {
        const int c = seg_count[(int64_t)b * K + k];
        seg_start[(int64_t)b * K + k] = run;
        seg_len[(int64_t)b * K + k] = c;
        run += c;
    }
    out_status[b] = cand_count[b] > C ? 1 : 0;
}

// ---- scatter candidates into their (image, class) segments ---------------------------
__global__ __launch_bounds__(256) void seg_scatter_kernel(const uint64_t *__restrict__ cand, const int32_t *__restrict__ cand_count,
                                                          const int K, const int64_t C, const int64_t *__restrict__ seg_start,
                                                          int32_t *__restrict__ seg_fill, uint64_t *__restrict__ seg)
{
    const int b = blockIdx.y;
    const int64_t n = min((int64_t)cand_count[b], C);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t key = cand[(int64_t)b * C + i];
        const uint32_t ak = (uint32_t)key;
        const uint32_t anchor = ak / (uint32_t)K, k = ak - anchor * (uint32_t)K;
        const int pos = atomicAdd(&seg_fill[(int64_t)b * K + k], 1);
        seg[seg_start[(int64_t)b * K + k] + pos] = (key & 0xffffffff00000000ull) | anchor;
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

// One workgroup per image.  Counts / offsets of all classes are fetched in parallel (one latency),
// the first max_det survivors of every class are gathered by flat index, sorted 1024 at a time, and
// only the best max_det of every chunk survive to the final sort: O(M log^2 1024) instead of
// O(M log^2 M), for any number of candidates M.
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
    if (t == 0) {
        int run = 0;
        for (int k = 1; k <= K; ++k) { run += s_pre[k]; s_pre[k] = run; }
    }
    for (int i = t; i < TOPK_SURV; i += TOPK_THREADS) s_surv[i] = ~0ull;
    __syncthreads();
    const int M = s_pre[K];
    int n_surv = 0;

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

    const int nout = min(M, max_det);
    if (t < max_det) {
        const int64_t o = (int64_t)b * max_det + t;
        if (t < nout) {
            const uint64_t key = sr[0];                              // max_det <= 1024: the winners are elements 0..1023
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

size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

struct DetectWs {
    rn::f32x4 *boxes; uint64_t *cand, *seg; rn::f32x4 *sbox; uint8_t *supp;
    int32_t *cand_count, *seg_count, *seg_fill, *kept_count, *seg_len; int64_t *seg_start;
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
    // zeroed every call: cand_count | seg_count | seg_fill (contiguous)
    const size_t z0 = off;
    w.cand_count = (int32_t *)take((size_t)B * 4);
    w.seg_count = (int32_t *)take(BK * 4);
    w.seg_fill = (int32_t *)take(BK * 4);
    w.zero_bytes = off - z0;
    w.kept_count = (int32_t *)take(BK * 4);
    w.seg_len = (int32_t *)take(BK * 4);
    w.seg_start = (int64_t *)take(BK * 8);
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
    for (int l = 0; l < L; ++l) {
        rc = launch_decode(box_levels[l], dtype, B, level_anchors[l], base, A, anchors, anchor_bstride, image_hw, rw,
                           (float *)w.boxes, st);
        if (rc != RN_OK) return rc;
        ScanLevel &lv = sa.lv[l];
        lv.cls = cls_levels[l]; lv.A_l = level_anchors[l]; lv.base = base;
        lv.N = (int64_t)B * level_anchors[l] * K; lv.nvec = lv.N / vec; lv.voff = voff;
        base += level_anchors[l];
        voff += lv.nvec;
    }
    sa.total_vec = voff;
    RN_HIP(hipMemsetAsync(w.cand_count, 0, w.zero_bytes, st));

    sa.boxes = w.boxes; sa.A = A; sa.C = C; sa.K = K; sa.B = B;
    sa.score_thr = params->score_thr;
    sa.min_box = params->min_box;
    {   // logit-space pre-filter with a safety margin; the exact test is still sigmoid(x) > thr
        const double t = (double)params->score_thr;
        sa.pre_thr = (t > 0.0 && t < 1.0) ? (float)(log(t / (1.0 - t)) - 1e-2) : (t <= 0.0 ? -INFINITY : 40.0f);
        if (t >= 1.0) sa.pre_thr = INFINITY;
    }
    sa.cand = w.cand; sa.cand_count = w.cand_count; sa.seg_count = w.seg_count;
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
    hipLaunchKernelGGL(seg_offsets_kernel, dim3((unsigned)B), dim3(64), 0, st, w.cand_count, w.seg_count, K, C, w.seg_start,
                       w.seg_len, out_status);
    RN_LAUNCH_CHECK();
    {
        int64_t sb = (C + 255) / 256;
        if (sb > 1024) sb = 1024;
        hipLaunchKernelGGL(seg_scatter_kernel, dim3((unsigned)sb, (unsigned)B), dim3(256), 0, st, w.cand, w.cand_count, K, C,
                           w.seg_start, w.seg_fill, w.seg);
        RN_LAUNCH_CHECK();
    }
    rn::NmsLaunch na;
    na.keys = w.seg; na.kept = w.cand; na.keep_idx = nullptr; na.boxes = w.boxes;
    na.seg_start = w.seg_start; na.seg_len = w.seg_len; na.kept_count = w.kept_count;
    na.scratch_boxes = w.sbox; na.scratch_supp = w.supp;
    na.S = B * K; na.box_mode = 1; na.K = K; na.A = A; na.iou_thr = params->nms_thr;
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
