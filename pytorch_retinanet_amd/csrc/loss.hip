// K3 loss_fwd_bwd -- replaces RetinaNetLosses (retinanet/losses.py:19-145) and
// bbox_2_activ (retinanet/box_utils.py:25-34): the loss scalars AND their gradients
// from one streaming pass over the head outputs plus a sparse fix-up.
//
// Reference semantics kept: logit shift x+1 (Q1), reversed
// alpha (Q2), detached focal weight so d/dx = w*(sigmoid(x+1)-t) (Q3), ignore
// rows (-2) contribute nothing, empty GT => image contributes zero (Q7),
// per-image /clamp(num_fg,1) then mean over images (Q8), smooth-L1 beta form
// (Q10), log(gw/aw + 1e-8) (Q11).
//
// Structure (HBM-bound; per image A*K*s read + A*K*s written for the class tensor,
// A*4*s + A*4*s for the box tensor, A*8 for matches).  One kernel, each wave owns one contiguous
// element range of the flattened [B*A*K] tensor:
//
//   phase A (stream)  Treats EVERY class element as a plain background element (t = 0) of its
//                     image: no metadata, no LDS, no selects -- 10 VALU + exp/rcp/log per element,
//                     16-byte non-temporal loads / 16-byte stores, two groups of PF KiB of loads in
//                     flight per wave.  Exact for > 99% of the rows.
//   phase B (repair)  The same wave then visits the rows of its range that are NOT plain background
//                     (matched rows: one positive element each; ignored rows: whole row; about
//                     0.3% of the rows at the reference's settings; their `matches` were prefetched
//                     before phase A), rewrites those gradient elements, adds the loss corrections
//                     (correct term minus the background term phase A added, computed with the same
//                     instructions) and does the regression branch (encode + smooth-L1 + box
//                     gradient) of matched rows; every other box-gradient row is written as zeros.
//   finalize          A one-block kernel adds the per-block partial sums in double in a fixed
//                     order (deterministic; no float atomics).
#include <cstdlib>

#include "rn_common.hpp"
#include "rn_match.hpp"

namespace {

constexpr int LOSS_BLOCK = 256;
constexpr int LOSS_WAVES = LOSS_BLOCK / RN_WAVE;
constexpr int LOSS_MAX_BLOCKS = 4096;   // bound on resident blocks (256 CUs x 8, with headroom): sizes the partials workspace

// One pyramid level's head outputs: cls [B][A_l][K], box [B][A_l][4] (dense).  A single tensor
// [B][A][K] is the L = 1 case.  Levels are laid end to end in a virtual 16-byte-vector index space
// (voff = first vector of the level) that is split evenly over the waves.
struct LossLevel {
    const void *cls, *box;
    void *gcls, *gbox;
    int64_t A_l, base;       // anchors per image in this level, offset of the level in the per-image anchor axis
    int64_t N, nvec, voff;   // elements B*A_l*K, full vectors N/VEC, virtual vector offset
    int64_t per_image;       // A_l*K elements
};

struct LossArgs {
    int32_t L;
    int64_t total_vec;
    LossLevel lv[RN_MAX_LEVELS];
    const rn::f32x4 *anchors;
    int64_t anchor_bstride4;
    const rn::f32x4 *gt_boxes;
    const int64_t *gt_labels;
    const int32_t *gt_off;
    const int64_t *matches;
    const unsigned long long *special;   // nullable: [B][special_W] bit (a & 63) of word a >> 6 = [matches[b][a] != -1]  (rn_iou_match_special)
    int64_t special_W;
    const int32_t *num_fg;
    int64_t A;               // anchors per image over all levels (row length of `matches`)
    int64_t vec_per_wave;    // stream kernel: 16-byte vectors per wave (multiple of 64)
    int32_t reverse;         // stream kernel: waves take the ranges back to front (the logits the conv wrote last are read first)
    int32_t K, B;
    float inv_B;
    rn_loss_params p;
    float alpha_pos;         // weight of t=1 elements: 1-alpha (Q2)
    float2 *part_stream;     // [blocks] (cls, reg) partial sums
    unsigned *fin;           // nullable: state words of the in-kernel finalize (rn_loss_fwd_bwd_levels_fin): [0] arrivals, [2..5] two int64 sums
    float *out_loss;         // f32[2], written by workgroup 0 when `fin` is set
    const float *gscale;     // nullable device f32[1]: every GRADIENT (not the losses) is multiplied by it before the rounding to the I/O dtype
                             // (a GradScaler's scale: fp16 gradients of ~4e-10 would flush to zero if the scale came after the store)
    // fused matching (loss_stream_kernel<.., FUSED = true>): the IoU matcher of box_utils.py:51-80 runs in this kernel's prologue
    float fg_thr, bg_thr;
    int32_t *nfg_acc;        // [B] foreground counts of this launch: zero on entry (the finalize kernel re-zeroes them), device-scope atomics
    unsigned *bar;           // [1] grid barrier arrival counter: zero on entry (the finalize kernel re-zeroes it)
    int64_t *matches_out;    // nullable: [B][A] match codes, written by the wave that owns a row's first element
};

// ---- background element (t = 0) ------------------------------------------------------
// With E = exp(-z):  sigmoid(z) = 1/(1+E),  softplus(z) = z + ln(1+E)  -- no |z|, no select.
// z is clamped at -80 so E stays finite (sigmoid(-80) = 1.8e-35; its weight p^gamma underflows
// either way).  v_exp/v_log are the bare base-2 instructions (__expf/__logf would add ~10
// instructions of range handling each).  Returns wb = p^gamma * bce and g = p^gamma * p, both
// still to be multiplied by alpha * scale.
template <bool GAMMA2>
__device__ __forceinline__ void bg_elem(const float x, const rn_loss_params &p, float &wb, float &g)
{
    const float z = __builtin_amdgcn_fmed3f(x + p.logit_shift, -80.0f, __builtin_inff());
    const float den = 1.0f + __builtin_amdgcn_exp2f(z * -1.4426950408889634f);
    const float ps = __builtin_amdgcn_rcpf(den);
    const float w = GAMMA2 ? ps * ps : ((p.gamma == 0.0f) ? 1.0f : __powf(ps, p.gamma));
    const float bce = fmaf(__builtin_amdgcn_logf(den), 0.6931471805599453f, z);
    wb = w * bce;
    g = w * ps;
}

// ---- general element (fix-up kernel only): loss and d loss/dx, unscaled -------------------
template <bool GAMMA2>
__device__ __forceinline__ void focal_elem(const float x, const bool pos, const LossArgs &a, float &loss, float &grad)
{
    const float z = x + a.p.logit_shift;
    const float az = fabsf(z);
    const float e = __expf(-az);                       // exp(-|z|) in (0,1]
    const float den = 1.0f + e;
    const float r = __builtin_amdgcn_rcpf(den);
    const float er = e * r;
    const bool zp = z >= 0.0f;
    const float ps = zp ? r : er;                      // sigmoid(z)
    const float om = zp ? er : r;                      // 1 - sigmoid(z), no cancellation
    const float q = pos ? om : ps;                     // losses.py:43
    float w = GAMMA2 ? q * q : ((a.p.gamma == 0.0f) ? 1.0f : __powf(q, a.p.gamma));
    w *= pos ? a.alpha_pos : a.p.alpha;                // losses.py:44-45
    // log1p(e) = log(den) + (e - (den-1))/den   (correction recovers the bits lost in 1+e)
    const float l1p = __builtin_amdgcn_logf(den) * 0.6931471805599453f + (e - (den - 1.0f)) * r;
    const float bce = fmaxf(pos ? -z : z, 0.0f) + l1p; // (1-t)*z - log_sigmoid(z)
    loss = w * bce;
    grad = pos ? -(w * om) : (w * ps);                 // w * (sigmoid(z) - t)
}

template <int DT> struct box4;
template <> struct box4<RN_F32> {
    static __device__ __forceinline__ void ld(const void *p, int64_t row, float (&f)[4]) {
        const rn::f32x4 v = ((const rn::f32x4 *)p)[row];
        f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
    }
    static __device__ __forceinline__ void st(void *p, int64_t row, const float (&f)[4]) {
        rn::f32x4 v; v.x = f[0]; v.y = f[1]; v.z = f[2]; v.w = f[3];
        ((rn::f32x4 *)p)[row] = v;
    }
};
template <> struct box4<RN_BF16> {
    static __device__ __forceinline__ void ld(const void *p, int64_t row, float (&f)[4]) {
        const rn::u32x2 v = ((const rn::u32x2 *)p)[row];
        f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
        f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
    }
    static __device__ __forceinline__ void st(void *p, int64_t row, const float (&f)[4]) {
        rn::u32x2 v; v.x = rn::dt<RN_BF16>::pk(f[0], f[1]); v.y = rn::dt<RN_BF16>::pk(f[2], f[3]);
        ((rn::u32x2 *)p)[row] = v;
    }
};
template <> struct box4<RN_F16> {
    static __device__ __forceinline__ void ld(const void *p, int64_t row, float (&f)[4]) {
        const rn::u32x2 v = ((const rn::u32x2 *)p)[row];
        f[0] = rn::half_lo(v.x); f[1] = rn::half_hi(v.x); f[2] = rn::half_lo(v.y); f[3] = rn::half_hi(v.y);
    }
    static __device__ __forceinline__ void st(void *p, int64_t row, const float (&f)[4]) {
        rn::u32x2 v; v.x = rn::dt<RN_F16>::pk(f[0], f[1]); v.y = rn::dt<RN_F16>::pk(f[2], f[3]);
        ((rn::u32x2 *)p)[row] = v;
    }
};

// ---- regression term of one fg row: encode (box_utils.py:25-34) + smooth-L1 ----
__device__ __forceinline__ float reg_row(const rn::f32x4 g, const rn::f32x4 an, const float (&pred)[4],
                                         const rn_loss_params &p, float (&grad)[4])
{
    const float gcx = (g.x + g.z) / 2.0f, gcy = (g.y + g.w) / 2.0f, gw = g.z - g.x, gh = g.w - g.y;
    const float acx = (an.x + an.z) / 2.0f, acy = (an.y + an.w) / 2.0f, aw = an.z - an.x, ah = an.w - an.y;
    float tgt[4];
    tgt[0] = ((gcx - acx) / aw) * p.reg_w[0];
    tgt[1] = ((gcy - acy) / ah) * p.reg_w[1];
    tgt[2] = logf(gw / aw + p.log_eps) * p.reg_w[2];
    tgt[3] = logf(gh / ah + p.log_eps) * p.reg_w[3];
    float l = 0.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float d = pred[j] - tgt[j];
        const float n = fabsf(d);
        const float sg = (d > 0.0f) ? 1.0f : ((d < 0.0f) ? -1.0f : 0.0f);
        if (p.beta < 1e-5f) { l += n; grad[j] = sg; }                       // losses.py:21-22
        else if (n < p.beta) { l += 0.5f * (n * n) / p.beta; grad[j] = d / p.beta; }
        else { l += n - 0.5f * p.beta; grad[j] = sg; }
    }
    return l;
}

// num_fg[b]: K2's output, or -- fused matching -- this launch's device-scope counter (complete once the grid barrier has been passed;
// read with a device-scope atomic load: the adds were performed by other XCDs, whose L2 this one does not snoop)
template <bool FUSED>
__device__ __forceinline__ int load_nfg(const LossArgs &a, const int b)
{
    if (FUSED) return __hip_atomic_load(&a.nfg_acc[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return a.num_fg[b];
}

// alpha / (max(num_fg,1) * B) of image b; 0 for an image without GT (every row ignored, Q7)
template <bool FUSED = false>
__device__ __forceinline__ float image_gmul(const LossArgs &a, const int b)
{
    const int T = a.gt_off[b + 1] - a.gt_off[b];
    const int nf = load_nfg<FUSED>(a, b);
    return (T > 0) ? a.p.alpha * ((1.0f / (float)(nf > 1 ? nf : 1)) * a.inv_B) : 0.0f;
}

// ================================= loss kernel =======================================
// Phase A (stream): every class element of the wave's range is processed as a plain background
// element.  Phase B (epilogue): the same wave repairs the few elements of ITS OWN range that are not
// plain background (positive element of a matched row, whole ignored rows) and emits the box
// gradient / regression term of the rows whose first element it owns.  Both phases write from one
// wave, so program order gives the right final value without any cross-wave ordering.
constexpr int LIST_CAP_PLAIN = 320;  // rows of a wave's range whose repair goes through the LDS lists (5 chunks of 64; 273 at the train shape)
constexpr int LIST_CAP_FUSED = 384;  // fused matching: EVERY row of the range must fit (it runs at 5 workgroups per CU: 319 rows at the train shape)
constexpr int IGN_U = 4;            // independent element loads per lane per round in the ignored-row repair
constexpr int IGN_V = 2;            // independent 16-byte pieces per lane per round there
typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));   // a 16-byte access at a dword-aligned address

// Per-workgroup loss partials as 2^-32 fixed point (|partial| < 2^31: a workgroup sums at most a few 10^5 elements of O(10) each):
// the sum over the workgroups is then an INTEGER sum -- the same bits in any order -- so neither the finalize kernel nor the
// in-kernel finalize needs a fixed summation order to be deterministic.  Resolution 2.3e-10 per partial against losses of O(1).
// Non-finite partials (NaN / Inf logits) must stay visible: they add nothing and raise a flag instead (1 NaN, 2 +Inf, 4 -Inf);
// loss_unfix gives the value IEEE summation would have given: NaN, or the infinity when only one sign was seen.
__device__ __forceinline__ long long loss_fix(const float v, unsigned &flags)
{
    if (!(fabsf(v) < 2147483648.0f)) {                            // NaN, Inf, or a partial beyond the fixed-point range (treated as Inf)
        flags |= (v != v) ? 1u : (v > 0.0f ? 2u : 4u);
        return 0;
    }
    return __double2ll_rn((double)v * 4294967296.0);
}
constexpr int FIN_LINES = 64;                                    // cache lines of the in-kernel finalize (one per lane of the polling wave)
__device__ __forceinline__ long long shfl_xor_ll(const long long v, const int o)
{
    const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)((unsigned long long)v & 0xffffffffull), o, RN_WAVE);
    const unsigned hi = (unsigned)__shfl_xor((int)(unsigned)((unsigned long long)v >> 32), o, RN_WAVE);
    return (long long)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ float loss_unfix(const long long s, const unsigned flags)
{
    if (flags & 1u || (flags & 6u) == 6u) return __builtin_nanf("");
    if (flags & 2u) return __builtin_inff();
    if (flags & 4u) return -__builtin_inff();
    return (float)((double)s * (1.0 / 4294967296.0));
}
template <int DT, bool GAMMA2, bool WRITE_GRAD, int PF, int NT, bool FUSED = false, bool LIST = false>
__global__ __launch_bounds__(LOSS_BLOCK) __attribute__((amdgpu_waves_per_eu((FUSED || !GAMMA2) ? 5 : 6))) void loss_stream_kernel(const LossArgs a)
{
    typedef rn::dt<DT> D;
    constexpr int VEC = D::VEC;
    constexpr int LIST_CAP = FUSED ? LIST_CAP_FUSED : LIST_CAP_PLAIN;
    __shared__ float s_part[LOSS_WAVES][2];
    __shared__ signed char s_pmv[FUSED ? LOSS_WAVES : 1][FUSED ? LIST_CAP : 1];   // fused matching: match code of every row of the wave's range
    __shared__ unsigned short s_flag[LIST ? LOSS_WAVES : 1][LIST ? LIST_CAP : 1];   // (LIST) flagged rows of the wave's range (offsets from its first row), all chunks in one list
    __shared__ unsigned short s_ign_row[LOSS_WAVES][LIST_CAP];   // ignored rows of the wave's range (offsets from its first row)
    __shared__ float s_ign_gm[LOSS_WAVES][LIST_CAP];            // their alpha/(max(nfg,1)*B)
    __shared__ int s_pos_off[LOSS_WAVES][LIST_CAP];             // positive elements of matched rows: offset from the range's first element
    __shared__ float s_pos_val[LOSS_WAVES][LIST_CAP];           //   and their gradient

    const int lane = threadIdx.x & (RN_WAVE - 1);
    // readfirstlane: the wave index and everything derived from it (ranges, trip counts, image
    // seams) is wave-uniform -> SGPRs and scalar branches.
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t gwave_raw = (int64_t)blockIdx.x * LOSS_WAVES + wave;
    const int64_t gwave = (a.reverse & 1) ? (int64_t)gridDim.x * LOSS_WAVES - 1 - gwave_raw : gwave_raw;
    const int64_t gv_beg = gwave * a.vec_per_wave;                // this wave's range in the virtual vector space
    const int64_t gv_end = min(gv_beg + a.vec_per_wave, a.total_vec);  // (levels laid end to end)
    const int K = a.K;
    const float gs = a.gscale ? *a.gscale : 1.0f;                 // gradient pre-scale (wave-uniform: a scalar load)

    // ---- Fused matching (FUSED): box_utils.py:51-80 for the rows of THIS wave's range, before anything else.  At the train shape
    // (T <= 64 GT boxes per image) a row's match is ~T x 25 VALU instructions, while a separate K2 launch is bound by launch latency and
    // one memory round trip per workgroup (17 us kernel + a memset for 16 MB).  The GT boxes of an image sit one
    // per LANE (a 16-byte load each, L2 hits) and are broadcast with v_readlane; pair arithmetic and tie / NaN semantics are K2's own
    // (rn_match.hpp).  The one quantity that is global is num_fg[b] (the normaliser of every gradient of image b, losses.py:107-109):
    // the waves add their own rows' foreground counts to device-scope counters and meet at a grid barrier -- the grid is the resident
    // grid (launch_stream), so every workgroup is on the chip.  Match codes wait in LDS (one byte per row) for the repair phase.
    if (FUSED) {
        using namespace rn_match;
        int m_off = 0;
        for (int li = 0; li < a.L; ++li) {
            const LossLevel &lv = a.lv[li];
            const int64_t nvec = lv.nvec;
            if (gv_end <= lv.voff && nvec > 0) break;
            const int64_t v_beg = max(gv_beg, lv.voff) - lv.voff;
            const int64_t v_end = min(gv_end, lv.voff + nvec) - lv.voff;
            const bool active = (nvec > 0) ? (v_beg < v_end) : (gwave == 0);
            if (!active) continue;
            const int64_t e_beg = v_beg * VEC;
            const int64_t e_end = (nvec == 0 || v_end == nvec) ? lv.N : v_end * VEC;
            const int64_t row_lo = e_beg / K;
            const int64_t row_hi = e_end > e_beg ? (e_end - 1) / K : row_lo - 1;
#pragma unroll 1
            for (int64_t r0 = row_lo; r0 <= row_hi; r0 += RN_WAVE) {
                const int n = (int)min((int64_t)RN_WAVE, row_hi - r0 + 1);
                const bool valid = lane < n;
                const int64_t r = min(r0 + lane, row_hi);                          // (idle lanes repeat the last row: every lane holds a real anchor)
                const int b = (int)((uint32_t)r / (uint32_t)lv.A_l);
                const int64_t ag = lv.base + (r - (int64_t)b * lv.A_l);
                const rn::f32x4 an = a.anchors[(int64_t)b * a.anchor_bstride4 + ag];
                const float area_a = (an.z - an.x) * (an.w - an.y);
                const int b_first = __builtin_amdgcn_readfirstlane(b), b_last = __builtin_amdgcn_readlane(b, RN_WAVE - 1);   // (rows ascend with the lane)
                const bool a_ok = __all(anchor_is_proper(an, area_a));
                WaveBox bb = {0.f, 0.f, 0.f, 0.f};
                if (a_ok) bb = WaveBox{wave_min(an.x), wave_min(an.y), wave_max(an.z), wave_max(an.w)};
                int pmv = -2;
#pragma unroll 1
                for (int bq = b_first; bq <= b_last; ++bq) {                        // one image per chunk, two where a seam crosses it
                    const int t0 = a.gt_off[bq];
                    const int T = min(max(a.gt_off[bq + 1] - t0, 0), RN_WAVE);      // (the host promised <= 64 per image)
                    rn::f32x4 mine = {0.f, 0.f, 0.f, 0.f};
                    float mine_a = 0.0f;
                    if (lane < T) { mine = a.gt_boxes[t0 + lane]; mine_a = (mine.z - mine.x) * (mine.w - mine.y); }
                    float best = 0.0f;
                    int bi = 0;
                    if (a_ok && __all(lane >= T || gt_is_proper(mine, mine_a))) {
                        unsigned long long todo = __ballot(lane < T && may_overlap(bb, mine));   // 64 boxes culled against the wave's strip at once
                        while (todo) {
                            const int l = __ffsll((long long)todo) - 1;
                            todo &= todo - 1;
                            const GtBox g = gt_of_lane(mine, mine_a, l);
                            const float inter = inter_fast(g, an);
                            if (__any(inter != 0.0f)) {
                                const float v = inter / ((g.area + area_a) - inter);
                                if (v > best) { best = v; bi = l; }
                            }
                        }
                    } else {
                        Best bb2 = {0.0f, 0, false};
                        for (int l = 0; l < T; ++l) {
                            const GtBox g = gt_of_lane(mine, mine_a, l);
                            careful_update(bb2, iou_pair(vec(g), g.area, an, area_a), l);
                        }
                        best = bb2.v; bi = bb2.i;
                    }
                    const int code = (int)classify(best, bi, T, a.fg_thr, a.bg_thr);
                    if (b == bq) pmv = code;
                }
                if (valid) s_pmv[wave][m_off + (int)(r0 - row_lo) + lane] = (signed char)pmv;
                const bool own = valid && r * K >= e_beg;                          // the row's first element lies in this wave's range: counted / written once
                if (a.matches_out && own) a.matches_out[(int64_t)b * a.A + ag] = (int64_t)pmv;
#pragma unroll 1
                for (int bq = b_first; bq <= b_last; ++bq) {
                    const unsigned long long fg = __ballot(own && b == bq && pmv >= 0);
                    if (fg && lane == 0) {
                        // returning form + a use of the result: the wave waits until the add has been PERFORMED, so it is ordered before
                        // this workgroup's barrier arrival below without a release fence (an agent-scope release writes the L2 back)
                        const int old = __hip_atomic_fetch_add(&a.nfg_acc[bq], (int)__popcll(fg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        asm volatile("" ::"v"(old));
                    }
                }
            }
            m_off += (int)(row_hi - row_lo + 1);
        }
        // grid barrier: one arrival per workgroup, polled by one thread per workgroup
        __syncthreads();
        if (threadIdx.x == 0) {
            // NON-returning add: 1 536 returning adds to one address serialise at ~40 ns each (measured: +57 us per launch), the
            // fire-and-forget form at ~1.4 ns.  This wave's foreground adds above were waited for, the other waves' before the barrier
            __hip_atomic_fetch_add(a.bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // (bounded: were the grid ever not co-resident -- it is sized from the occupancy query -- the launch must fail loudly, not
            // hang the device: ~4 M polls, seconds, then a sticky flag that makes the finalize kernel return NaN losses)
            unsigned spins = 0;
            while (__hip_atomic_load(a.bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x) {
                __builtin_amdgcn_s_sleep(8);
                if (++spins > (1u << 21)) { __hip_atomic_store(a.bar + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            }
        }
        __syncthreads();
    }

    // acc is carried in double: phase A adds the background term of rows that phase B later takes out
    // again, and that cancellation must not cost precision when such a row holds large logits.
    // (One DP fma per group per lane: free.)
    double acc = 0.0;                                            // already scaled by alpha*scale
    float reg = 0.0f;
    int seg_off = 0;                                             // (fused matching) first LDS slot of the current level segment's match codes

    for (int li = 0; li < a.L; ++li) {                           // a wave's range rarely spans more than one level
        const LossLevel &lv = a.lv[li];
        const int64_t nvec = lv.nvec;                            // full vectors of this level's tensor
        if (gv_end <= lv.voff && nvec > 0) break;
        const int64_t v_beg = max(gv_beg, lv.voff) - lv.voff;
        const int64_t v_end = min(gv_end, lv.voff + nvec) - lv.voff;
        // the wave that ends at nvec also owns the level's ragged tail (< VEC elements)
        const bool active = (nvec > 0) ? (v_beg < v_end) : (gwave == 0);
        if (!active) continue;
        const rn::u32x4 *src = (const rn::u32x4 *)lv.cls;
        rn::u32x4 *dst = (rn::u32x4 *)lv.gcls;
        const int64_t e_beg = v_beg * VEC;
        const int64_t e_end = (nvec == 0 || v_end == nvec) ? lv.N : v_end * VEC;
        const int64_t row_lo = e_beg / K;                        // rows are local to the level: r = b * A_l + a_local
        const int64_t row_hi = e_end > e_beg ? (e_end - 1) / K : row_lo - 1;     // inclusive

        // ---- Phase B, part 1 (BEFORE the stream): everything the repair of this range needs is INPUT data -- match codes, labels,
        // GT boxes, the positive logits -- so the dependent load chains (3 - 4 round trips) run here, at the start of the wave,
        // where the other resident waves cover them, instead of at its end, where the whole chip waits for the slowest wave
        // (round 2: the stream alone 110 / 126 us warm / cold, with the repair behind it 135 / 144).  What has to FOLLOW the
        // stream are only the stores that overwrite gradient elements phase A writes (the positive element of a matched row,
        // the zeros of an ignored row): they wait in two wave-private LDS lists.  The box gradients (not touched by phase A) are
        // final here.  A chunk of 64 rows costs two SCALAR loads when it has no special row (5 chunks of 6): the words of
        // rn_iou_match_special; `matches` is read only by the lanes whose flag is set.
        const int64_t cap_hi = min(row_hi, row_lo + (int64_t)LIST_CAP - 1);        // rows [row_lo, cap_hi] go through the lists
        int n_pos = 0, n_ign = 0, n_ign_b = 0;                                     // (n_ign_b: ignored rows shared with a neighbouring wave, listed from the back)
        auto prep = [&]() {
            if (WRITE_GRAD) {
                // zero box gradients of the rows this wave owns, as 16-byte vectors (two 16-bit rows per lane); matched rows
                // overwrite theirs below (same wave: program order)
                constexpr int RB = 4 * (int)sizeof(typename D::elem);     // bytes per box row
                const int64_t own_lo = (e_beg + K - 1) / K;
                if (own_lo <= cap_hi) {                                   // wave-uniform
                    unsigned char *const gb0 = (unsigned char *)lv.gbox;
                    const int64_t b0 = own_lo * RB, b1 = (cap_hi + 1) * RB;
                    const int64_t v0 = (b0 + 15) >> 4, v1 = b1 >> 4;
                    for (int64_t v = v0 + lane; v < v1; v += RN_WAVE) ((rn::u32x4 *)gb0)[v] = rn::u32x4{0u, 0u, 0u, 0u};
                    if (RB == 8) {                                        // an odd first / last row: one 8-byte store each
                        if (lane == 0 && (b0 & 15)) *(rn::u32x2 *)(gb0 + b0) = rn::u32x2{0u, 0u};
                        if (lane == 1 && (b1 & 15) && (v1 << 4) >= b0) *(rn::u32x2 *)(gb0 + (v1 << 4)) = rn::u32x2{0u, 0u};
                    }
                }
            }
            if constexpr (LIST) {
            // ---- (1) which rows of the range are special: a 64-row chunk's flags are two SCALAR loads (the words of rn_iou_match_special);
            // the flagged rows of ALL chunks go into ONE compact list.  The kernel is bound by its memory INSTRUCTIONS, not its bytes
            // (an 8-byte access of three active lanes costs what a 16-byte access of 64 costs): at 500 GT boxes per image every chunk of
            // a wave holds a few special rows, and chunk by chunk each paid the whole chain below -- match code, label, GT box, anchor,
            // prediction, positive logit -- for a handful of lanes; from the list the chain runs once per 64 SPECIAL rows.
            int n_flag = 0;
            const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll 1
            for (int64_t r0 = row_lo; r0 <= cap_hi; r0 += RN_WAVE) {                 // wave-uniform
                const int n = (int)min((int64_t)RN_WAVE, cap_hi - r0 + 1);
                unsigned long long m = ~0ull;                                      // no words (old entry points): look every row up
                if (FUSED) {
                    int pmv_l = -1;
                    if (lane < n) pmv_l = s_pmv[wave][seg_off + (int)(r0 - row_lo) + lane];
                    m = __ballot(pmv_l != -1);
                } else if (a.special) {
                    const int b0 = (int)((uint32_t)r0 / (uint32_t)lv.A_l);
                    const int64_t a0 = r0 - (int64_t)b0 * lv.A_l;
                    if (a0 + n <= lv.A_l) {                                        // the chunk lies inside one image
                        const int64_t ag0 = lv.base + a0, w = ag0 >> 6;
                        const int sh = (int)(ag0 & 63);
                        const unsigned long long *p = a.special + (int64_t)b0 * a.special_W + w;
                        const unsigned long long lo = p[0];
                        const unsigned long long hi = (sh && w + 1 < a.special_W) ? p[1] : 0ull;
                        m = sh ? ((lo >> sh) | (hi << (64 - sh))) : lo;
                    } else {
                        // the chunk straddles an image seam of this level: every lane looks its own bit up (`matches` may hold
                        // NOTHING at rows without a flag: rn_iou_match_special_ex, RN_MATCH_FLAGGED_ONLY)
                        bool f = false;
                        if (lane < n) {
                            const int64_t rr = r0 + lane;
                            const int bb = (int)((uint32_t)rr / (uint32_t)lv.A_l);
                            const int64_t agl = lv.base + (rr - (int64_t)bb * lv.A_l);
                            f = (a.special[(int64_t)bb * a.special_W + (agl >> 6)] >> (agl & 63)) & 1ull;
                        }
                        m = __ballot(f);
                    }
                }
                if (n < RN_WAVE) m &= (1ull << n) - 1ull;
                if (!m) continue;
                if ((m >> lane) & 1ull) s_flag[wave][n_flag + __popcll(m & below)] = (unsigned short)((int)(r0 - row_lo) + lane);
                n_flag += __popcll(m);
            }
            if (n_flag) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
            // ---- (2) the special rows, one per lane
#pragma unroll 1
            for (int i0 = 0; i0 < n_flag; i0 += RN_WAVE) {                          // wave-uniform
                const bool flagged = i0 + lane < n_flag;
                const int ro = (int)s_flag[wave][min(i0 + lane, n_flag - 1)];
                const int64_t r = row_lo + ro;
                int pmv = -1, b = 0;
                int64_t ag = 0;
                if (flagged) {
                    b = (int)((uint32_t)r / (uint32_t)lv.A_l);
                    ag = lv.base + (r - (int64_t)b * lv.A_l);
                    pmv = FUSED ? (int)s_pmv[wave][seg_off + ro] : (int)a.matches[(int64_t)b * a.A + ag];
                }
                const bool special = flagged && pmv != -1;
                int t0 = 0, T = 0, nf = 1;
                if (special) { t0 = a.gt_off[b]; T = a.gt_off[b + 1] - t0; nf = load_nfg<FUSED>(a, b); }
                const bool live = special && T > 0;                                // images without GT: phase A writes zeros
                const bool matched = live && pmv >= 0, ign = live && pmv < 0;
                const float scale = (1.0f / (float)(nf > 1 ? nf : 1)) * a.inv_B;
                int code = -1;
                if (matched) code = (int)a.gt_labels[t0 + pmv] - 1;
                const int64_t e_pos = r * K + code;
                const bool pos_ok = matched && code >= 0 && code < K && e_pos >= e_beg && e_pos < e_end;
                float xp = 0.0f;
                if (pos_ok) xp = D::ld(lv.cls, e_pos);
                float gr = 0.0f;
                if (pos_ok) {                                                      // matched row: only the positive element differs from what phase A does
                    float wb, gbg, l;
                    bg_elem<GAMMA2>(xp, a.p, wb, gbg);
                    focal_elem<GAMMA2>(xp, true, a, l, gr);
                    acc += (double)l * (double)scale - (double)wb * (double)(a.p.alpha * scale);
                    gr *= scale * gs;
                }
                const unsigned long long pmask = __ballot(pos_ok);
                if (pos_ok) {
                    const int pos = n_pos + __popcll(pmask & below);
                    s_pos_off[wave][pos] = (int)(e_pos - e_beg);
                    s_pos_val[wave][pos] = gr;
                }
                n_pos += __popcll(pmask);
                const bool mo = matched && r * K >= e_beg;                         // the row's first element is ours: its box gradient too
                if (__any(mo)) {                                                   // wave-uniform
                    if (mo) {
                        float gb[4] = {0.0f, 0.0f, 0.0f, 0.0f}, pred[4];
                        box4<DT>::ld(lv.box, r, pred);
                        const float l = reg_row(a.gt_boxes[t0 + pmv], a.anchors[(int64_t)b * a.anchor_bstride4 + ag], pred, a.p, gb);
                        reg += l * scale;
#pragma unroll
                        for (int j = 0; j < 4; ++j) gb[j] *= scale * gs;
                        if (WRITE_GRAD) box4<DT>::st(lv.gbox, r, gb);
                    }
                }
                // ignored rows: those that lie wholly inside the range fill the list from the front (whole 16-byte pieces below), the
                // at most two that another wave shares from the back (element-wise, clipped to the range)
                const bool inner = r * K >= e_beg && (r + 1) * K <= e_end;
                const unsigned long long imask = __ballot(ign && inner), bmask = __ballot(ign && !inner);
                if (ign) {
                    const int pos = inner ? n_ign + __popcll(imask & below) : LIST_CAP - 1 - (n_ign_b + __popcll(bmask & below));
                    s_ign_row[wave][pos] = (unsigned short)ro;
                    s_ign_gm[wave][pos] = a.p.alpha * scale;
                }
                n_ign += __popcll(imask);
                n_ign_b += __popcll(bmask);
            }
            if (n_ign | n_ign_b) {                                                 // wave-uniform
                // ignored rows: their background terms (which phase A will add) come out again; reads only -- the zero stores
                // follow the stream.
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                if (VEC == 8 && !(K & 1)) {
                    // 16-bit logits, even K: a row is K / 2 whole dwords starting on a 4-byte boundary.  Whole rows move as 16-BYTE pieces
                    // (dword-aligned global_load_dwordx4: fine on gfx950, tools/misalign_probe.hip) + K / 2 mod 4 single dwords: a row of
                    // K = 90 logits is 11 + 1 accesses instead of 45 -- at 500 GT boxes per image a wave has ~14 ignored rows, whose dword
                    // accesses were as many memory instructions as a tenth of its stream, read side and store side each
                    const int K2 = K >> 1, P = K2 >> 2, R = K2 & 3;
                    const uint32_t *const src32 = (const uint32_t *)lv.cls;
                    const int total_p = n_ign * P;
#pragma unroll 1
                    for (int tb = 0; tb < total_p; tb += RN_WAVE * IGN_V) {
                        u32x4_a4 xs[IGN_V];
                        float gms[IGN_V];
                        bool ok[IGN_V];
#pragma unroll
                        for (int u = 0; u < IGN_V; ++u) {
                            const int tt = tb + u * RN_WAVE + lane;
                            const int t = min(tt, total_p - 1);
                            const int j = t / P, pi = t - j * P;
                            const int64_t d = (((row_lo + s_ign_row[wave][j]) * K) >> 1) + 4 * pi;
                            gms[u] = s_ign_gm[wave][j];
                            ok[u] = tt < total_p;
                            xs[u] = *(const u32x4_a4 *)(src32 + d);
                        }
#pragma unroll
                        for (int u = 0; u < IGN_V; ++u) {
                            if (ok[u]) {
                                float x[VEC], s = 0.0f;
                                D::unpack(rn::u32x4{xs[u].x, xs[u].y, xs[u].z, xs[u].w}, x);
#pragma unroll
                                for (int q8 = 0; q8 < VEC; ++q8) {
                                    float wb, gbg;
                                    bg_elem<GAMMA2>(x[q8], a.p, wb, gbg);
                                    s += wb;
                                }
                                acc -= (double)s * (double)gms[u];
                            }
                        }
                    }
                    // single dwords: the K / 2 mod 4 tail of every inner row, then all of the (<= 2) shared rows, clipped to the range
                    const int total_r = n_ign * R, total_b = n_ign_b * K2;
#pragma unroll 1
                    for (int tb = 0; tb < total_r + total_b; tb += RN_WAVE * IGN_U) {
                        uint32_t xs[IGN_U];
                        float gms[IGN_U];
                        bool ok[IGN_U];
#pragma unroll
                        for (int u = 0; u < IGN_U; ++u) {
                            const int tt = tb + u * RN_WAVE + lane;
                            const int t = min(tt, total_r + total_b - 1);
                            int j, k2;
                            if (t < total_r) { j = t / R; k2 = 4 * P + (t - j * R); }
                            else { const int tq = t - total_r; const int jb = tq / K2; k2 = tq - jb * K2; j = LIST_CAP - 1 - jb; }
                            const int64_t e = (row_lo + s_ign_row[wave][j]) * K + 2 * k2;
                            gms[u] = s_ign_gm[wave][j];
                            ok[u] = tt < total_r + total_b && e >= e_beg && e < e_end;
                            xs[u] = src32[e >> 1];
                        }
#pragma unroll
                        for (int u = 0; u < IGN_U; ++u) {
                            if (ok[u]) {
                                const float x0 = DT == RN_BF16 ? __uint_as_float(xs[u] << 16) : rn::half_lo(xs[u]);
                                const float x1 = DT == RN_BF16 ? __uint_as_float(xs[u] & 0xffff0000u) : rn::half_hi(xs[u]);
                                float wb0, wb1, gbg;
                                bg_elem<GAMMA2>(x0, a.p, wb0, gbg);
                                bg_elem<GAMMA2>(x1, a.p, wb1, gbg);
                                acc -= (double)wb0 * (double)gms[u];
                                acc -= (double)wb1 * (double)gms[u];
                            }
                        }
                    }
                } else {
                    const int total = (n_ign + n_ign_b) * K;
                    for (int t0 = 0; t0 < total; t0 += RN_WAVE * IGN_U) {
                        float xs[IGN_U], gms[IGN_U];
                        bool ok[IGN_U];
#pragma unroll
                        for (int u = 0; u < IGN_U; ++u) {
                            const int t = min(t0 + u * RN_WAVE + lane, total - 1);
                            const int jj = t / K, k = t - jj * K;
                            const int j = jj < n_ign ? jj : LIST_CAP - 1 - (jj - n_ign);
                            const int64_t e = (row_lo + s_ign_row[wave][j]) * K + k;
                            gms[u] = s_ign_gm[wave][j];
                            ok[u] = (t0 + u * RN_WAVE + lane < total) && e >= e_beg && e < e_end;
                            xs[u] = D::ld(lv.cls, e);
                        }
#pragma unroll
                        for (int u = 0; u < IGN_U; ++u) {
                            if (ok[u]) {
                                float wb, gbg;
                                bg_elem<GAMMA2>(xs[u], a.p, wb, gbg);
                                acc -= (double)wb * (double)gms[u];
                            }
                        }
                    }
                }
            }
            } else {
#pragma unroll 1
            for (int64_t r0 = row_lo; r0 <= cap_hi; r0 += RN_WAVE) {                 // wave-uniform
                const int n = (int)min((int64_t)RN_WAVE, cap_hi - r0 + 1);
                unsigned long long m = ~0ull;                                      // no words (old entry points): look every row up
                int pmv_l = -1;
                if (FUSED) {
                    if (lane < n) pmv_l = s_pmv[wave][seg_off + (int)(r0 - row_lo) + lane];
                    m = __ballot(pmv_l != -1);
                } else if (a.special) {
                    const int b0 = (int)((uint32_t)r0 / (uint32_t)lv.A_l);
                    const int64_t a0 = r0 - (int64_t)b0 * lv.A_l;
                    if (a0 + n <= lv.A_l) {                                        // the chunk lies inside one image
                        const int64_t ag0 = lv.base + a0, w = ag0 >> 6;
                        const int sh = (int)(ag0 & 63);
                        const unsigned long long *p = a.special + (int64_t)b0 * a.special_W + w;
                        const unsigned long long lo = p[0];
                        const unsigned long long hi = (sh && w + 1 < a.special_W) ? p[1] : 0ull;
                        m = sh ? ((lo >> sh) | (hi << (64 - sh))) : lo;
                    } else {
                        // the chunk straddles an image seam of this level: every lane looks its own bit up (`matches` may hold
                        // NOTHING at rows without a flag: rn_iou_match_special_ex, RN_MATCH_FLAGGED_ONLY)
                        bool f = false;
                        if (lane < n) {
                            const int64_t rr = r0 + lane;
                            const int bb = (int)((uint32_t)rr / (uint32_t)lv.A_l);
                            const int64_t agl = lv.base + (rr - (int64_t)bb * lv.A_l);
                            f = (a.special[(int64_t)bb * a.special_W + (agl >> 6)] >> (agl & 63)) & 1ull;
                        }
                        m = __ballot(f);
                    }
                }
                if (n < RN_WAVE) m &= (1ull << n) - 1ull;
                if (!m) continue;
                const int64_t r = r0 + lane;
                const bool flagged = (m >> lane) & 1ull;
                int pmv = -1, b = 0;
                int64_t ag = 0;
                if (flagged) {
                    b = (int)((uint32_t)r / (uint32_t)lv.A_l);
                    ag = lv.base + (r - (int64_t)b * lv.A_l);
                    pmv = FUSED ? pmv_l : (int)a.matches[(int64_t)b * a.A + ag];
                }
                const bool special = flagged && pmv != -1;
                int t0 = 0, T = 0, nf = 1;
                if (special) { t0 = a.gt_off[b]; T = a.gt_off[b + 1] - t0; nf = load_nfg<FUSED>(a, b); }
                const bool live = special && T > 0;                                // images without GT: phase A writes zeros
                const bool matched = live && pmv >= 0, ign = live && pmv < 0;
                const float scale = (1.0f / (float)(nf > 1 ? nf : 1)) * a.inv_B;
                int code = -1;
                if (matched) code = (int)a.gt_labels[t0 + pmv] - 1;
                const int64_t e_pos = r * K + code;
                const bool pos_ok = matched && code >= 0 && code < K && e_pos >= e_beg && e_pos < e_end;
                float xp = 0.0f;
                if (pos_ok) xp = D::ld(lv.cls, e_pos);
                float gr = 0.0f;
                if (pos_ok) {                                                      // matched row: only the positive element differs from what phase A does
                    float wb, gbg, l;
                    bg_elem<GAMMA2>(xp, a.p, wb, gbg);
                    focal_elem<GAMMA2>(xp, true, a, l, gr);
                    acc += (double)l * (double)scale - (double)wb * (double)(a.p.alpha * scale);
                    gr *= scale * gs;
                }
                const unsigned long long below = (1ull << lane) - 1ull;
                const unsigned long long pmask = __ballot(pos_ok);
                if (pos_ok) {
                    const int pos = n_pos + __popcll(pmask & below);
                    s_pos_off[wave][pos] = (int)(e_pos - e_beg);
                    s_pos_val[wave][pos] = gr;
                }
                n_pos += __popcll(pmask);
                const bool mo = matched && r * K >= e_beg;                         // the row's first element is ours: its box gradient too
                if (__any(mo)) {                                                   // wave-uniform
                    if (mo) {
                        float gb[4] = {0.0f, 0.0f, 0.0f, 0.0f}, pred[4];
                        box4<DT>::ld(lv.box, r, pred);
                        const float l = reg_row(a.gt_boxes[t0 + pmv], a.anchors[(int64_t)b * a.anchor_bstride4 + ag], pred, a.p, gb);
                        reg += l * scale;
#pragma unroll
                        for (int j = 0; j < 4; ++j) gb[j] *= scale * gs;
                        if (WRITE_GRAD) box4<DT>::st(lv.gbox, r, gb);
                    }
                }
                const unsigned long long imask = __ballot(ign);
                if (ign) {
                    const int pos = n_ign + __popcll(imask & below);
                    s_ign_row[wave][pos] = (unsigned short)(r - row_lo);
                    s_ign_gm[wave][pos] = a.p.alpha * scale;
                }
                n_ign += __popcll(imask);
            }
            if (n_ign) {                                                           // wave-uniform
                // ignored rows: their background terms (which phase A will add) come out again; reads only -- the zero stores
                // follow the stream.  Elements spread over all lanes, IGN_U independent loads per lane per round.
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                if (VEC == 8 && !(K & 1)) {
                    // 16-bit logits, even K: a row is K / 2 whole dwords (rows start on 4-byte boundaries, ranges on 16-byte ones), so the
                    // repair moves two elements per memory instruction -- at 500 GT boxes per image a wave has ~14 ignored rows, and their
                    // 2-byte loads / stores were as many memory instructions as a fifth of its stream (round 3: 196 us at T = 500)
                    const int K2 = K >> 1, total2 = n_ign * K2;
                    const uint32_t *const src32 = (const uint32_t *)lv.cls;
                    for (int t0 = 0; t0 < total2; t0 += RN_WAVE * IGN_U) {
                        uint32_t xs[IGN_U];
                        float gms[IGN_U];
                        bool ok[IGN_U];
#pragma unroll
                        for (int u = 0; u < IGN_U; ++u) {
                            const int t = min(t0 + u * RN_WAVE + lane, total2 - 1);
                            const int j = t / K2, k2 = t - j * K2;
                            const int64_t e = (row_lo + s_ign_row[wave][j]) * K + 2 * k2;
                            gms[u] = s_ign_gm[wave][j];
                            ok[u] = (t0 + u * RN_WAVE + lane < total2) && e >= e_beg && e < e_end;
                            xs[u] = src32[e >> 1];
                        }
#pragma unroll
                        for (int u = 0; u < IGN_U; ++u) {
                            if (ok[u]) {
                                const float x0 = DT == RN_BF16 ? __uint_as_float(xs[u] << 16) : rn::half_lo(xs[u]);
                                const float x1 = DT == RN_BF16 ? __uint_as_float(xs[u] & 0xffff0000u) : rn::half_hi(xs[u]);
                                float wb0, wb1, gbg;
                                bg_elem<GAMMA2>(x0, a.p, wb0, gbg);
                                bg_elem<GAMMA2>(x1, a.p, wb1, gbg);
                                acc -= (double)wb0 * (double)gms[u];
                                acc -= (double)wb1 * (double)gms[u];
                            }
                        }
                    }
                } else {
                    const int total = n_ign * K;
                    for (int t0 = 0; t0 < total; t0 += RN_WAVE * IGN_U) {
                        float xs[IGN_U], gms[IGN_U];
                        bool ok[IGN_U];
#pragma unroll
                        for (int u = 0; u < IGN_U; ++u) {
                            const int t = min(t0 + u * RN_WAVE + lane, total - 1);
                            const int j = t / K, k = t - j * K;
                            const int64_t e = (row_lo + s_ign_row[wave][j]) * K + k;
                            gms[u] = s_ign_gm[wave][j];
                            ok[u] = (t0 + u * RN_WAVE + lane < total) && e >= e_beg && e < e_end;
                            xs[u] = D::ld(lv.cls, e);
                        }
#pragma unroll
                        for (int u = 0; u < IGN_U; ++u) {
                            if (ok[u]) {
                                float wb, gbg;
                                bg_elem<GAMMA2>(xs[u], a.p, wb, gbg);
                                acc -= (double)wb * (double)gms[u];
                            }
                        }
                    }
                }
            }
            }
        };
        if (v_beg >= v_end) prep();

        if (v_beg < v_end) {
            int b = (int)(e_beg / lv.per_image);                      // image of the first element
            int64_t img_end_v = ((int64_t)(b + 1) * lv.per_image) / VEC;   // vectors [.., img_end_v) lie entirely in image b
            float gmul = image_gmul<FUSED>(a, b);
            float gmul_g = gmul * gs;                                 // the gradients' multiplier (loss multiplier x pre-scale)

            const int64_t last = v_end - 1;
            const int64_t groups = (v_end - v_beg) / (PF * RN_WAVE);      // full groups of PF wave-iterations
            rn::u32x4 q[PF];
#pragma unroll
            for (int u = 0; u < PF; ++u) q[u] = (NT & 1) ? __builtin_nontemporal_load(&src[min(v_beg + u * RN_WAVE + lane, last)]) : src[min(v_beg + u * RN_WAVE + lane, last)];

            int64_t v0 = v_beg;
            // the range's repair preparation (dependent load chains, a few microseconds), under the first stream loads.
            // (Measured alternatives, same box: after the stream as in round 2 +6 us cold -- the whole chip waits for the
            // slowest wave's chains --; at a per-wave pseudo-random point inside the stream +5..8 us cold: the second copy of
            // the loop costs registers and the stream its sixth wave per SIMD.)
            prep();
            for (int64_t gi = 0; gi < groups; ++gi, v0 += PF * RN_WAVE) {
                // next group's loads first (index clamped: straight-line code, countable vmcnt)
                rn::u32x4 qn[PF];
#pragma unroll
                for (int u = 0; u < PF; ++u) qn[u] = (NT & 1) ? __builtin_nontemporal_load(&src[min(v0 + (PF + u) * RN_WAVE + lane, last)]) : src[min(v0 + (PF + u) * RN_WAVE + lane, last)];

                if (v0 + PF * RN_WAVE <= img_end_v) {                    // whole group inside image b (scalar test)
                    float acc_g = 0.0f;
#pragma unroll
                    for (int u = 0; u < PF; ++u) {
                        float x[VEC], g[VEC];
                        D::unpack(q[u], x);
#pragma unroll
                        for (int j = 0; j < VEC; ++j) {
                            float wb, gg;
                            bg_elem<GAMMA2>(x[j], a.p, wb, gg);
                            acc_g += wb;
                            g[j] = gg * gmul_g;
                        }
                        if (WRITE_GRAD) { if (NT & 2) __builtin_nontemporal_store(D::pack(g), &dst[v0 + u * RN_WAVE + lane]); else dst[v0 + u * RN_WAVE + lane] = D::pack(g); }
                    }
                    acc += (double)acc_g * (double)gmul;
                } else {                                                 // an image seam crosses this group (<= B-1 times per level)
                    // The waves that meet a seam must not fall behind (the whole chip waits for the last wave): when only ONE
                    // seam lies in the group -- always, unless an image of this level is shorter than a group -- every element
                    // is "image b before element e_seam, image b + 1 from it on": two wave-uniform multipliers and a compare.
                    const int64_t e_seam = (int64_t)(b + 1) * lv.per_image;
                    const bool one_seam = lv.per_image >= (int64_t)PF * RN_WAVE * VEC && b + 1 < a.B;
                    const float gmul_next = one_seam ? image_gmul<FUSED>(a, b + 1) : 0.0f;
#pragma unroll 1
                    for (int u = 0; u < PF; ++u) {
                        const int64_t v = v0 + u * RN_WAVE + lane;
                        float x[VEC], g[VEC];
                        D::unpack(src[v], x);
                        // elements [0, k) of this vector belong to image b, the rest to image b + 1
                        const int k = one_seam ? (int)max(min(e_seam - v * VEC, (int64_t)VEC), (int64_t)0) : VEC;
                        float s_all = 0.0f, s_hi = 0.0f;
#pragma unroll
                        for (int j = 0; j < VEC; ++j) {
                            float wb, gg;
                            bg_elem<GAMMA2>(x[j], a.p, wb, gg);
                            if (one_seam) {
                                s_all += wb;
                                s_hi += j >= k ? wb : 0.0f;
                                g[j] = gg * (j >= k ? gmul_next * gs : gmul_g);
                            } else {
                                const float gm = image_gmul<FUSED>(a, (int)((v * VEC + j) / lv.per_image));
                                acc += (double)wb * (double)gm;
                                g[j] = gg * (gm * gs);
                            }
                        }
                        if (one_seam) acc += (double)s_all * (double)gmul + (double)s_hi * ((double)gmul_next - (double)gmul);
                        if (WRITE_GRAD) dst[v] = D::pack(g);
                    }
                    b = (int)(((v0 + PF * RN_WAVE) * VEC) / lv.per_image);
                    if (b > a.B - 1) b = a.B - 1;
                    img_end_v = ((int64_t)(b + 1) * lv.per_image) / VEC;
                    gmul = image_gmul<FUSED>(a, b);
                    gmul_g = gmul * gs;
                }
#pragma unroll
                for (int u = 0; u < PF; ++u) q[u] = qn[u];
            }
            // leftover vectors of the range (< one group), per-element image lookup
#pragma unroll 1
            for (int64_t v = v0 + lane; v < v_end; v += RN_WAVE) {
                float x[VEC], g[VEC];
                D::unpack(src[v], x);
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    const float gm = image_gmul<FUSED>(a, (int)((v * VEC + j) / lv.per_image));
                    float wb, gg;
                    bg_elem<GAMMA2>(x[j], a.p, wb, gg);
                    acc += (double)wb * (double)gm;
                    g[j] = gg * (gm * gs);
                }
                if (WRITE_GRAD) dst[v] = D::pack(g);
            }
        }
        // ragged tail of the level's tensor (< VEC elements), owned by the wave that ends at nvec
        if (nvec == 0 || v_end == nvec) {
            const int64_t e = nvec * VEC + lane;
            if (lane < VEC && e < lv.N) {
                const float gm = image_gmul<FUSED>(a, (int)(e / lv.per_image));
                float wb, gg;
                bg_elem<GAMMA2>(D::ld(lv.cls, e), a.p, wb, gg);
                acc += (double)wb * (double)gm;
                if (WRITE_GRAD) D::st(lv.gcls, e, gg * (gm * gs));
            }
        }

        // ---- Phase B, part 2 (after the stream): the stores that overwrite what phase A wrote, from the two lists -- no loads
        if constexpr (LIST) {
        if (WRITE_GRAD && (n_pos | n_ign | n_ign_b)) {                             // wave-uniform
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            for (int i = lane; i < n_pos; i += RN_WAVE) D::st(lv.gcls, e_beg + s_pos_off[wave][i], s_pos_val[wave][i]);
            if (VEC == 8 && !(K & 1)) {                                            // (16-byte zero pieces + single dwords: see the read side)
                const int K2 = K >> 1, P = K2 >> 2, R = K2 & 3;
                uint32_t *const dst32 = (uint32_t *)lv.gcls;
                const int total_p = n_ign * P;
                for (int t = lane; t < total_p; t += RN_WAVE) {
                    const int j = t / P, pi = t - j * P;
                    const int64_t d = (((row_lo + s_ign_row[wave][j]) * K) >> 1) + 4 * pi;
                    *(u32x4_a4 *)(dst32 + d) = u32x4_a4{0u, 0u, 0u, 0u};
                }
                const int total_r = n_ign * R, total_b = n_ign_b * K2;
                for (int t = lane; t < total_r + total_b; t += RN_WAVE) {
                    int j, k2;
                    if (t < total_r) { j = t / R; k2 = 4 * P + (t - j * R); }
                    else { const int tq = t - total_r; const int jb = tq / K2; k2 = tq - jb * K2; j = LIST_CAP - 1 - jb; }
                    const int64_t e = (row_lo + s_ign_row[wave][j]) * K + 2 * k2;
                    if (e >= e_beg && e < e_end) dst32[e >> 1] = 0u;
                }
            } else {
                const int total = (n_ign + n_ign_b) * K;
                for (int t = lane; t < total; t += RN_WAVE) {
                    const int jj = t / K, k = t - jj * K;
                    const int j = jj < n_ign ? jj : LIST_CAP - 1 - (jj - n_ign);
                    const int64_t e = (row_lo + s_ign_row[wave][j]) * K + k;
                    if (e >= e_beg && e < e_end) D::st(lv.gcls, e, 0.0f);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        } else {
        if (WRITE_GRAD && (n_pos | n_ign)) {                                       // wave-uniform
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            for (int i = lane; i < n_pos; i += RN_WAVE) D::st(lv.gcls, e_beg + s_pos_off[wave][i], s_pos_val[wave][i]);
            if (VEC == 8 && !(K & 1)) {                                            // (two 16-bit zeros per store: see the read side)
                const int K2 = K >> 1, total2 = n_ign * K2;
                for (int t = lane; t < total2; t += RN_WAVE) {
                    const int j = t / K2, k2 = t - j * K2;
                    const int64_t e = (row_lo + s_ign_row[wave][j]) * K + 2 * k2;
                    if (e >= e_beg && e < e_end) ((uint32_t *)lv.gcls)[e >> 1] = 0u;
                }
            } else {
                const int total = n_ign * K;
                for (int t = lane; t < total; t += RN_WAVE) {
                    const int j = t / K, k = t - j * K;
                    const int64_t e = (row_lo + s_ign_row[wave][j]) * K + k;
                    if (e >= e_beg && e < e_end) D::st(lv.gcls, e, 0.0f);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        }
        if (FUSED) seg_off += (int)(row_hi - row_lo + 1);       // (the host admits the fused form only when every range fits the lists)
        // rows past the lists' capacity (ranges longer than LIST_CAP rows: small K or few waves), chunk by chunk after the stream
        for (int64_t c0 = row_lo + LIST_CAP; !FUSED && c0 <= row_hi; c0 += RN_WAVE) {
            const int64_t r = c0 + lane;
            bool ignored = false;
            float ign_gm = 0.0f;
            if (r <= row_hi) {
                const int b = (int)((uint32_t)r / (uint32_t)lv.A_l);
                const int64_t ag = lv.base + (r - (int64_t)b * lv.A_l);   // anchor index within the image
                // (with the flag words `matches` is read at flagged rows only: it may be written there only, RN_MATCH_FLAGGED_ONLY)
                const bool flagged = !a.special || ((a.special[(int64_t)b * a.special_W + (ag >> 6)] >> (ag & 63)) & 1ull);
                const int64_t m = flagged ? a.matches[(int64_t)b * a.A + ag] : (int64_t)-1;
                const int64_t r_e0 = r * K;
                const bool own_row = r_e0 >= e_beg;                       // the row's first element is ours
                float gb[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                if (m != -1) {                                            // rare: ~0.3% of the rows
                    const int t0 = a.gt_off[b], T = a.gt_off[b + 1] - t0;
                    if (T > 0) {                                          // images without GT: phase A already wrote zeros
                        const int nf = a.num_fg[b];
                        const float scale = (1.0f / (float)(nf > 1 ? nf : 1)) * a.inv_B;
                        const float gmul = a.p.alpha * scale;
                        if (m >= 0) {
                            const int gi = t0 + (int)m;
                            const int code = (int)a.gt_labels[gi] - 1;
                            const int64_t e_pos = r_e0 + code;
                            if (code >= 0 && code < K && e_pos >= e_beg && e_pos < e_end) {
                                // matched row: only the positive element differs from what phase A did
                                const float x = D::ld(lv.cls, e_pos);
                                float wb, gbg, l, gr;
                                bg_elem<GAMMA2>(x, a.p, wb, gbg);
                                focal_elem<GAMMA2>(x, true, a, l, gr);
                                acc += (double)l * (double)scale - (double)wb * (double)gmul;
                                if (WRITE_GRAD) D::st(lv.gcls, e_pos, gr * (scale * gs));
                            }
                            if (own_row) {
                                float pred[4];
                                box4<DT>::ld(lv.box, r, pred);
                                const float l = reg_row(a.gt_boxes[gi], a.anchors[(int64_t)b * a.anchor_bstride4 + ag], pred, a.p, gb);
                                reg += l * scale;
#pragma unroll
                                for (int j = 0; j < 4; ++j) gb[j] *= scale * gs;
                            }
                        } else {
                            ignored = true;
                            ign_gm = gmul;
                        }
                    }
                }
                if (WRITE_GRAD && own_row) box4<DT>::st(lv.gbox, r, gb);
            }
            // ignored rows: remove their background contribution and zero their gradient.  The rows are
            // compacted into a wave-private LDS list and their elements are spread over all lanes,
            // IGN_U independent loads per lane per round (elements outside this wave's range are masked).
            const unsigned long long imask = __ballot(ignored);
            if (imask) {                                                   // wave-uniform
                if (ignored) {
                    const int pos = __popcll(imask & ((1ull << lane) - 1ull));
                    s_ign_row[wave][pos] = (unsigned char)lane;
                    s_ign_gm[wave][pos] = ign_gm;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const int total = __popcll(imask) * K;
                for (int t0 = 0; t0 < total; t0 += RN_WAVE * IGN_U) {
                    float xs[IGN_U], gms[IGN_U];
                    int64_t es[IGN_U];
                    bool ok[IGN_U];
#pragma unroll
                    for (int u = 0; u < IGN_U; ++u) {
                        const int t = min(t0 + u * RN_WAVE + lane, total - 1);
                        const int j = t / K, k = t - j * K;
                        es[u] = (c0 + s_ign_row[wave][j]) * K + k;
                        gms[u] = s_ign_gm[wave][j];
                        ok[u] = (t0 + u * RN_WAVE + lane < total) && es[u] >= e_beg && es[u] < e_end;
                        xs[u] = D::ld(lv.cls, es[u]);
                    }
#pragma unroll
                    for (int u = 0; u < IGN_U; ++u) {
                        if (ok[u]) {
                            float wb, gbg;
                            bg_elem<GAMMA2>(xs[u], a.p, wb, gbg);
                            acc -= (double)wb * (double)gms[u];
                            if (WRITE_GRAD) D::st(lv.gcls, es[u], 0.0f);
                        }
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        }
    }

    const float accf = (float)rn::wave_sum_d(acc);
    reg = rn::wave_sum(reg);
    if (lane == 0) { s_part[wave][0] = accf; s_part[wave][1] = reg; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float c = 0.0f, rg = 0.0f;
#pragma unroll
        for (int w = 0; w < LOSS_WAVES; ++w) { c += s_part[w][0]; rg += s_part[w][1]; }
        if (a.fin) {
            // The batch sums WITHOUT a finalize launch: every workgroup adds its two partials, as 2^-32 fixed point, to the two 64-bit
            // words of ONE OF 64 cache lines (line = workgroup index mod 64: 1 536 workgroups adding to one line serialise at its L2
            // channel -- measured +15 us --, 24 per line do not) and then bumps that line's arrival counter -- relaxed device-scope
            // atomics that return nothing; a line's three words are performed in program order at its channel.  Integer addition is
            // order-independent, so the result is the same whoever adds first.  Wave 0 of workgroup 0 polls the 64 counters (one per
            // lane), sums the lines when all workgroups have arrived, converts, and leaves every word zeroed for the next launch.
            // Ordering: the two sums are RETURNING adds whose results this thread waits for -- a device-scope atomic is
            // performed at the coherence point when its result comes back -- and only then is the arrival published; the poller reads
            // arrivals, then sums, with device-scope loads.  Nothing rests on fire-and-forget atomics to one line being performed in
            // program order, and no L2 write-back (an agent-scope release fence) is paid.
            unsigned *const line = a.fin + (blockIdx.x & (FIN_LINES - 1)) * 16;
            unsigned long long *const sums = (unsigned long long *)(line + 2);
            unsigned fl_c = 0u, fl_r = 0u;
            const long long fc = loss_fix(c, fl_c), fr = loss_fix(rg, fl_r);
            const unsigned long long o0 = __hip_atomic_fetch_add(sums + 0, (unsigned long long)fc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long o1 = __hip_atomic_fetch_add(sums + 1, (unsigned long long)fr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned o2 = 0u;
            if (fl_c | fl_r) o2 = __hip_atomic_fetch_or(line + 1, fl_c | (fl_r << 8), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (rare: non-finite logits)
            asm volatile("" ::"v"(o0), "v"(o1), "v"(o2));
            __hip_atomic_fetch_add(line, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (a.fin && blockIdx.x == 0 && wave == 0) {                 // (thread 0's own arrival above precedes this in program order: same wave)
        static_assert(FIN_LINES == RN_WAVE, "one line per lane");
        unsigned *const mine = a.fin + lane * 16;
        unsigned long long *const msum = (unsigned long long *)(mine + 2);
        // STICKY poison (state word 1, shared with the fused form's barrier flag): a launch whose workgroups did not all arrive leaves
        // the lines dirty -- late arrivals would add into re-zeroed words and every later launch would finish early on stale counts with
        // finite, wrong losses.  Instead the words stay as they are, the poison word is set, and every later call on this state
        // returns NaN losses until the host replaces the state buffer (ops.reset_match_state).
        unsigned *const poison = a.fin - 64 + 1;
        unsigned spins = 0;
        bool ok = __hip_atomic_load(poison, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u;
        while (ok) {
            const int arrived = rn::wave_sum_i((int)__hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            if (arrived >= (int)gridDim.x) break;
            __builtin_amdgcn_s_sleep(4);
            if (++spins > (1u << 22)) { ok = false; __hip_atomic_store(poison, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }   // (a workgroup that never arrives = a faulted launch)
        }
        if (!ok) {
            if (lane == 0) { a.out_loss[0] = __builtin_nanf(""); a.out_loss[1] = __builtin_nanf(""); }
            return;
        }
        long long sc = (long long)__hip_atomic_load(msum + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        long long sr = (long long)__hip_atomic_load(msum + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned fl = __hip_atomic_load(mine + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            sc += shfl_xor_ll(sc, o); sr += shfl_xor_ll(sr, o);
            fl |= (unsigned)__shfl_xor((int)fl, o, RN_WAVE);
        }
        __hip_atomic_store(msum + 0, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(msum + 1, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(mine + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(mine, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (lane == 0) {
            a.out_loss[0] = loss_unfix(sc, fl & 0xffu);
            a.out_loss[1] = loss_unfix(sr, (fl >> 8) & 0xffu);
        }
    }
    if (!a.fin && threadIdx.x == 0) {
        {
            float c = 0.0f, rg = 0.0f;
#pragma unroll
            for (int w = 0; w < LOSS_WAVES; ++w) { c += s_part[w][0]; rg += s_part[w][1]; }
            a.part_stream[blockIdx.x] = make_float2(c, rg);
        }
    }
}

// =========================== K3 as stream + repair (round 6) ===========================
// The same arithmetic split over two launches (rn_loss_fwd_bwd_levels_rp):
//
//   loss_bg_kernel      phase A alone: every class element as a plain background element of its image, the zero box gradients
//                       of the rows a wave owns, the partial sums into the 64 state lines.  No row logic at all: no flag words,
//                       no lists, no LDS beyond the block reduction.
//   loss_repair_kernel  walks the flag words of rn_iou_match_special (one strip of REPAIR_WORDS words = 512 anchor rows per
//                       wave), compacts the flagged rows of its strip into an LDS list and gives every LANE one special row:
//                       match code -> label / GT box -> positive logit, the chains of 64 rows in flight together instead of one
//                       64-row chunk after the other inside a streaming wave.  Matched rows: the positive element's loss
//                       correction and gradient, the regression term and the box gradient.  Ignored rows: their background terms
//                       come out of the sum again and their gradient rows are zeroed, elements spread over all lanes.  The
//                       stores overwrite what loss_bg_kernel wrote (stream order).  The workgroup that arrives last sums the 64
//                       lines, writes the two losses and leaves the state zeroed.
//
// Why: inside the streaming kernel the repair is a per-wave serial prefix whose length follows the LOCAL density of special rows
// (at 500 GT boxes per image ~25 special rows per wave on average, several times that where the boxes cluster), and the whole chip
// waits for the slowest wave; here the special rows are balanced over the chip by construction.  Every sum leaves a workgroup as
// 2^-32 fixed point converted from the DOUBLE partial, so the cancellation between a row's background term (added by the first
// kernel) and its removal (second kernel) is exact to 2.3e-10 whatever the logits.
constexpr int REPAIR_WORDS = 8;                       // flag words per wave: 512 anchor rows
constexpr int REPAIR_ROWS = REPAIR_WORDS * 64;
constexpr int REPAIR_BLOCK = 256;
constexpr int REPAIR_WAVES = REPAIR_BLOCK / RN_WAVE;
constexpr int REPAIR_IGN_U = 8;                      // independent loads per lane per round in the ignored-row pass

__device__ __forceinline__ long long loss_fix_d(const double v, unsigned &flags)
{
    if (!(fabs(v) < 2147483648.0)) {
        flags |= (v != v) ? 1u : (v > 0.0 ? 2u : 4u);
        return 0;
    }
    return __double2ll_rn(v * 4294967296.0);
}

// a workgroup's two partial sums -> its state line (returning adds: the values are PERFORMED at the device-scope coherence point
// when the results come back, which is what orders them before a later arrival count without an L2 write-back)
__device__ __forceinline__ void fin_add(unsigned *const fin, const double c, const double rg, const bool wait)
{
    unsigned *const line = fin + (blockIdx.x & (FIN_LINES - 1)) * 16;
    unsigned long long *const sums = (unsigned long long *)(line + 2);
    unsigned fl_c = 0u, fl_r = 0u;
    const long long fc = loss_fix_d(c, fl_c), fr = loss_fix_d(rg, fl_r);
    if (wait) {
        const unsigned long long o0 = __hip_atomic_fetch_add(sums + 0, (unsigned long long)fc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long o1 = __hip_atomic_fetch_add(sums + 1, (unsigned long long)fr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned o2 = 0u;
        if (fl_c | fl_r) o2 = __hip_atomic_fetch_or(line + 1, fl_c | (fl_r << 8), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("" ::"v"(o0), "v"(o1), "v"(o2));
    } else {                                                   // (the kernel's end performs them: the next kernel on the stream reads them)
        __hip_atomic_fetch_add(sums + 0, (unsigned long long)fc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(sums + 1, (unsigned long long)fr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (fl_c | fl_r) __hip_atomic_fetch_or(line + 1, fl_c | (fl_r << 8), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <int DT, bool GAMMA2, bool WRITE_GRAD, int PF, int NT>
__global__ __launch_bounds__(LOSS_BLOCK) void loss_bg_kernel(const LossArgs a)
{
    typedef rn::dt<DT> D;
    constexpr int VEC = D::VEC;
    __shared__ double s_part[LOSS_WAVES];
    const int lane = threadIdx.x & (RN_WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t gwave_raw = (int64_t)blockIdx.x * LOSS_WAVES + wave;
    const int64_t gwave = (a.reverse & 1) ? (int64_t)gridDim.x * LOSS_WAVES - 1 - gwave_raw : gwave_raw;
    const int64_t gv_beg = gwave * a.vec_per_wave;
    const int64_t gv_end = min(gv_beg + a.vec_per_wave, a.total_vec);
    const int K = a.K;
    const float gs = a.gscale ? *a.gscale : 1.0f;
    double acc = 0.0;

    for (int li = 0; li < a.L; ++li) {
        const LossLevel &lv = a.lv[li];
        const int64_t nvec = lv.nvec;
        if (gv_end <= lv.voff && nvec > 0) break;
        const int64_t v_beg = max(gv_beg, lv.voff) - lv.voff;
        const int64_t v_end = min(gv_end, lv.voff + nvec) - lv.voff;
        const bool active = (nvec > 0) ? (v_beg < v_end) : (gwave == 0);
        if (!active) continue;
        const rn::u32x4 *src = (const rn::u32x4 *)lv.cls;
        rn::u32x4 *dst = (rn::u32x4 *)lv.gcls;
        const int64_t e_beg = v_beg * VEC;
        const int64_t e_end = (nvec == 0 || v_end == nvec) ? lv.N : v_end * VEC;

        rn::u32x4 q[PF];
        const int64_t last = v_end - 1;
        if (v_beg < v_end) {
#pragma unroll
            for (int u = 0; u < PF; ++u) q[u] = (NT & 1) ? __builtin_nontemporal_load(&src[min(v_beg + u * RN_WAVE + lane, last)]) : src[min(v_beg + u * RN_WAVE + lane, last)];
        }
        if (WRITE_GRAD) {
            // zero box gradients of the rows whose first class element lies in this range (the repair kernel overwrites the matched ones)
            constexpr int RB = 4 * (int)sizeof(typename D::elem);
            const int64_t own_lo = (e_beg + K - 1) / K;
            const int64_t own_hi = e_end > e_beg ? (e_end - 1) / K : own_lo - 1;
            if (own_lo <= own_hi) {
                unsigned char *const gb0 = (unsigned char *)lv.gbox;
                const int64_t b0 = own_lo * RB, b1 = (own_hi + 1) * RB;
                const int64_t v0 = (b0 + 15) >> 4, v1 = b1 >> 4;
                for (int64_t v = v0 + lane; v < v1; v += RN_WAVE) ((rn::u32x4 *)gb0)[v] = rn::u32x4{0u, 0u, 0u, 0u};
                if (RB == 8) {
                    if (lane == 0 && (b0 & 15)) *(rn::u32x2 *)(gb0 + b0) = rn::u32x2{0u, 0u};
                    if (lane == 1 && (b1 & 15) && (v1 << 4) >= b0) *(rn::u32x2 *)(gb0 + (v1 << 4)) = rn::u32x2{0u, 0u};
                }
            }
        }
        if (v_beg < v_end) {
            int b = (int)(e_beg / lv.per_image);
            int64_t img_end_v = ((int64_t)(b + 1) * lv.per_image) / VEC;
            float gmul = image_gmul<false>(a, b);
            float gmul_g = gmul * gs;
            const int64_t groups = (v_end - v_beg) / (PF * RN_WAVE);
            int64_t v0 = v_beg;
            for (int64_t gi = 0; gi < groups; ++gi, v0 += PF * RN_WAVE) {
                rn::u32x4 qn[PF];
#pragma unroll
                for (int u = 0; u < PF; ++u) qn[u] = (NT & 1) ? __builtin_nontemporal_load(&src[min(v0 + (PF + u) * RN_WAVE + lane, last)]) : src[min(v0 + (PF + u) * RN_WAVE + lane, last)];
                if (v0 + PF * RN_WAVE <= img_end_v) {
                    float acc_g = 0.0f;
#pragma unroll
                    for (int u = 0; u < PF; ++u) {
                        float x[VEC], g[VEC];
                        D::unpack(q[u], x);
#pragma unroll
                        for (int j = 0; j < VEC; ++j) {
                            float wb, gg;
                            bg_elem<GAMMA2>(x[j], a.p, wb, gg);
                            acc_g += wb;
                            g[j] = gg * gmul_g;
                        }
                        if (WRITE_GRAD) { if (NT & 2) __builtin_nontemporal_store(D::pack(g), &dst[v0 + u * RN_WAVE + lane]); else dst[v0 + u * RN_WAVE + lane] = D::pack(g); }
                    }
                    acc += (double)acc_g * (double)gmul;
                } else {
                    const int64_t e_seam = (int64_t)(b + 1) * lv.per_image;
                    const bool one_seam = lv.per_image >= (int64_t)PF * RN_WAVE * VEC && b + 1 < a.B;
                    const float gmul_next = one_seam ? image_gmul<false>(a, b + 1) : 0.0f;
#pragma unroll 1
                    for (int u = 0; u < PF; ++u) {
                        const int64_t v = v0 + u * RN_WAVE + lane;
                        float x[VEC], g[VEC];
                        D::unpack(src[v], x);
                        const int k = one_seam ? (int)max(min(e_seam - v * VEC, (int64_t)VEC), (int64_t)0) : VEC;
                        float s_all = 0.0f, s_hi = 0.0f;
#pragma unroll
                        for (int j = 0; j < VEC; ++j) {
                            float wb, gg;
                            bg_elem<GAMMA2>(x[j], a.p, wb, gg);
                            if (one_seam) {
                                s_all += wb;
                                s_hi += j >= k ? wb : 0.0f;
                                g[j] = gg * (j >= k ? gmul_next * gs : gmul_g);
                            } else {
                                const float gm = image_gmul<false>(a, (int)((v * VEC + j) / lv.per_image));
                                acc += (double)wb * (double)gm;
                                g[j] = gg * (gm * gs);
                            }
                        }
                        if (one_seam) acc += (double)s_all * (double)gmul + (double)s_hi * ((double)gmul_next - (double)gmul);
                        if (WRITE_GRAD) dst[v] = D::pack(g);
                    }
                    b = (int)(((v0 + PF * RN_WAVE) * VEC) / lv.per_image);
                    if (b > a.B - 1) b = a.B - 1;
                    img_end_v = ((int64_t)(b + 1) * lv.per_image) / VEC;
                    gmul = image_gmul<false>(a, b);
                    gmul_g = gmul * gs;
                }
#pragma unroll
                for (int u = 0; u < PF; ++u) q[u] = qn[u];
            }
#pragma unroll 1
            for (int64_t v = v0 + lane; v < v_end; v += RN_WAVE) {
                float x[VEC], g[VEC];
                D::unpack(src[v], x);
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    const float gm = image_gmul<false>(a, (int)((v * VEC + j) / lv.per_image));
                    float wb, gg;
                    bg_elem<GAMMA2>(x[j], a.p, wb, gg);
                    acc += (double)wb * (double)gm;
                    g[j] = gg * (gm * gs);
                }
                if (WRITE_GRAD) dst[v] = D::pack(g);
            }
        }
        if (nvec == 0 || v_end == nvec) {
            const int64_t e = nvec * VEC + lane;
            if (lane < VEC && e < lv.N) {
                const float gm = image_gmul<false>(a, (int)(e / lv.per_image));
                float wb, gg;
                bg_elem<GAMMA2>(D::ld(lv.cls, e), a.p, wb, gg);
                acc += (double)wb * (double)gm;
                if (WRITE_GRAD) D::st(lv.gcls, e, gg * (gm * gs));
            }
        }
    }
    const double accw = rn::wave_sum_d(acc);
    if (lane == 0) s_part[wave] = accw;
    __syncthreads();
    if (threadIdx.x == 0) {
        double c = 0.0;
#pragma unroll
        for (int w = 0; w < LOSS_WAVES; ++w) c += s_part[w];
        fin_add(a.fin, c, 0.0, false);
    }
}

struct RepairLevelTab { const void *cls, *box; void *gcls, *gbox; };

template <int DT, bool GAMMA2, bool WRITE_GRAD>
__global__ __launch_bounds__(REPAIR_BLOCK) void loss_repair_kernel(const LossArgs a, const int strips_per_image, const int total_strips)
{
    typedef rn::dt<DT> D;
    constexpr int VEC = D::VEC;
    __shared__ unsigned short s_rows[REPAIR_WAVES][REPAIR_ROWS];     // flagged rows of the strip (offsets from its first anchor)
    __shared__ int s_ign_r[REPAIR_WAVES][REPAIR_ROWS];               // ignored rows: row index inside the level tensor (b * A_l + a_local)
    __shared__ unsigned char s_ign_l[REPAIR_WAVES][REPAIR_ROWS];     //   and their level
    __shared__ double s_part[REPAIR_WAVES][2];
    __shared__ RepairLevelTab s_lv[RN_MAX_LEVELS];
    __shared__ unsigned s_last;
    const int lane = threadIdx.x & (RN_WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (threadIdx.x < RN_MAX_LEVELS) {
        const LossLevel &lv = a.lv[threadIdx.x];
        s_lv[threadIdx.x] = RepairLevelTab{lv.cls, lv.box, lv.gcls, lv.gbox};
    }
    __syncthreads();
    const int K = a.K;
    const float gs = a.gscale ? *a.gscale : 1.0f;
    double acc = 0.0, reg = 0.0;
    const int strip = (int)blockIdx.x * REPAIR_WAVES + wave;
    if (strip < total_strips) {
        const int b = strip / strips_per_image;
        const int64_t w0 = (int64_t)(strip - b * strips_per_image) * REPAIR_WORDS;
        const int t0 = a.gt_off[b], T = a.gt_off[b + 1] - t0;
        unsigned long long word = 0ull;
        if (T > 0 && lane < REPAIR_WORDS && w0 + lane < a.special_W) word = a.special[(int64_t)b * a.special_W + w0 + lane];
        unsigned long long nz = __ballot(word != 0ull);               // (T <= 0: an image without GT contributes nothing; the stream wrote zeros)
        if (nz) {
            const int nf = a.num_fg[b];
            const float scale = (1.0f / (float)(nf > 1 ? nf : 1)) * a.inv_B;
            const float gm_ign = a.p.alpha * scale;
            const unsigned long long below = (1ull << lane) - 1ull;
            int n = 0;
            while (nz) {
                const int it = __ffsll((long long)nz) - 1;
                nz &= nz - 1;
                const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(word & 0xffffffffull), it);
                const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(word >> 32), it);
                const unsigned long long m = ((unsigned long long)hi << 32) | lo;
                if ((m >> lane) & 1ull) s_rows[wave][n + __popcll(m & below)] = (unsigned short)(it * 64 + lane);
                n += __popcll(m);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            int n_ign = 0;
#pragma unroll 1
            for (int i0 = 0; i0 < n; i0 += RN_WAVE) {
                const bool have = i0 + lane < n;
                const int64_t ag = w0 * 64 + (have ? (int64_t)s_rows[wave][i0 + lane] : 0);
                const bool ok = have && ag < a.A;
                int li = 0;
                int64_t base = 0, A_l = 1;
#pragma unroll 1
                for (int l = 0; l < a.L; ++l) {
                    const bool in = ag >= a.lv[l].base && ag < a.lv[l].base + a.lv[l].A_l;
                    if (in) { li = l; base = a.lv[l].base; A_l = a.lv[l].A_l; }
                }
                const int64_t r = (int64_t)b * A_l + (ag - base);             // row inside the level tensor
                int pmv = -1;
                rn::f32x4 an = {0.f, 0.f, 0.f, 0.f};
                float pred[4] = {0.f, 0.f, 0.f, 0.f};
                if (ok) {
                    pmv = (int)a.matches[(int64_t)b * a.A + ag];
                    an = a.anchors[(int64_t)b * a.anchor_bstride4 + ag];
                    box4<DT>::ld(s_lv[li].box, r, pred);
                }
                const bool matched = ok && pmv >= 0, ign = ok && pmv == -2;
                int code = -1;
                rn::f32x4 g = {0.f, 0.f, 1.f, 1.f};
                if (matched) { code = (int)a.gt_labels[t0 + pmv] - 1; g = a.gt_boxes[t0 + pmv]; }
                const bool pos_ok = matched && code >= 0 && code < K;
                const int64_t e_pos = r * K + code;
                float xp = 0.0f;
                if (pos_ok) xp = D::ld(s_lv[li].cls, e_pos);
                if (pos_ok) {
                    float wb, gbg, l, gr;
                    bg_elem<GAMMA2>(xp, a.p, wb, gbg);
                    focal_elem<GAMMA2>(xp, true, a, l, gr);
                    acc += (double)l * (double)scale - (double)wb * (double)gm_ign;
                    if (WRITE_GRAD) D::st(s_lv[li].gcls, e_pos, gr * (scale * gs));
                }
                if (matched) {
                    float gb[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                    const float l = reg_row(g, an, pred, a.p, gb);
                    reg += (double)(l * scale);
#pragma unroll
                    for (int j = 0; j < 4; ++j) gb[j] *= scale * gs;
                    if (WRITE_GRAD) box4<DT>::st(s_lv[li].gbox, r, gb);
                }
                const unsigned long long imask = __ballot(ign);
                if (ign) {
                    const int pos = n_ign + __popcll(imask & below);
                    s_ign_r[wave][pos] = (int)r;
                    s_ign_l[wave][pos] = (unsigned char)li;
                }
                n_ign += __popcll(imask);
            }
            if (n_ign) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                if (VEC == 8 && !(K & 1)) {
                    // 16-bit logits, even K: a row is K / 2 whole dwords -- two elements per memory instruction
                    const int K2 = K >> 1, total2 = n_ign * K2;
#pragma unroll 1
                    for (int tb = 0; tb < total2; tb += RN_WAVE * REPAIR_IGN_U) {
                        uint32_t xs[REPAIR_IGN_U];
                        uint32_t *gp[REPAIR_IGN_U];
                        bool okk[REPAIR_IGN_U];
#pragma unroll
                        for (int u = 0; u < REPAIR_IGN_U; ++u) {
                            const int tt = tb + u * RN_WAVE + lane;
                            const int t = min(tt, total2 - 1);
                            const int j = t / K2, k2 = t - j * K2;
                            const int l = s_ign_l[wave][j];
                            const int64_t d = ((int64_t)s_ign_r[wave][j] * K >> 1) + k2;
                            okk[u] = tt < total2;
                            xs[u] = ((const uint32_t *)s_lv[l].cls)[d];
                            gp[u] = (uint32_t *)s_lv[l].gcls + d;
                        }
#pragma unroll
                        for (int u = 0; u < REPAIR_IGN_U; ++u) {
                            if (okk[u]) {
                                const float x0 = DT == RN_BF16 ? __uint_as_float(xs[u] << 16) : rn::half_lo(xs[u]);
                                const float x1 = DT == RN_BF16 ? __uint_as_float(xs[u] & 0xffff0000u) : rn::half_hi(xs[u]);
                                float wb0, wb1, gbg;
                                bg_elem<GAMMA2>(x0, a.p, wb0, gbg);
                                bg_elem<GAMMA2>(x1, a.p, wb1, gbg);
                                acc -= (double)(wb0 + wb1) * (double)gm_ign;
                                if (WRITE_GRAD) *gp[u] = 0u;
                            }
                        }
                    }
                } else {
                    const int total = n_ign * K;
#pragma unroll 1
                    for (int tb = 0; tb < total; tb += RN_WAVE * REPAIR_IGN_U) {
                        float xs[REPAIR_IGN_U];
                        int64_t es[REPAIR_IGN_U];
                        int ls[REPAIR_IGN_U];
                        bool okk[REPAIR_IGN_U];
#pragma unroll
                        for (int u = 0; u < REPAIR_IGN_U; ++u) {
                            const int tt = tb + u * RN_WAVE + lane;
                            const int t = min(tt, total - 1);
                            const int j = t / K, k = t - j * K;
                            ls[u] = s_ign_l[wave][j];
                            es[u] = (int64_t)s_ign_r[wave][j] * K + k;
                            okk[u] = tt < total;
                            xs[u] = D::ld(s_lv[ls[u]].cls, es[u]);
                        }
#pragma unroll
                        for (int u = 0; u < REPAIR_IGN_U; ++u) {
                            if (okk[u]) {
                                float wb, gbg;
                                bg_elem<GAMMA2>(xs[u], a.p, wb, gbg);
                                acc -= (double)wb * (double)gm_ign;
                                if (WRITE_GRAD) D::st(s_lv[ls[u]].gcls, es[u], 0.0f);
                            }
                        }
                    }
                }
            }
        }
    }
    const double accw = rn::wave_sum_d(acc), regw = rn::wave_sum_d(reg);
    if (lane == 0) { s_part[wave][0] = accw; s_part[wave][1] = regw; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double c = 0.0, rg = 0.0;
#pragma unroll
        for (int w = 0; w < REPAIR_WAVES; ++w) { c += s_part[w][0]; rg += s_part[w][1]; }
        if (c != 0.0 || rg != 0.0) fin_add(a.fin, c, rg, true);     // (a NaN partial compares unequal: it is added)
        // arrival: the sums above have been performed (returning adds); the workgroup that takes the last ticket finishes the call
        const unsigned ticket = __hip_atomic_fetch_add(a.fin, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = (ticket == gridDim.x - 1u) ? 1u : 0u;
    }
    __syncthreads();
    if (s_last && wave == 0) {
        static_assert(FIN_LINES == RN_WAVE, "one line per lane");
        unsigned *const mine = a.fin + lane * 16;
        unsigned long long *const msum = (unsigned long long *)(mine + 2);
        long long sc = (long long)__hip_atomic_load(msum + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        long long sr = (long long)__hip_atomic_load(msum + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned fl = __hip_atomic_load(mine + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            sc += shfl_xor_ll(sc, o); sr += shfl_xor_ll(sr, o);
            fl |= (unsigned)__shfl_xor((int)fl, o, RN_WAVE);
        }
        __hip_atomic_store(msum + 0, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(msum + 1, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(mine + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(mine, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (lane == 0) {
            a.out_loss[0] = loss_unfix(sc, fl & 0xffu);
            a.out_loss[1] = loss_unfix(sr, (fl >> 8) & 0xffu);
        }
    }
}

// `bar` != null (fused matching): the launch's foreground counters move to num_fg_out and the state words go back to zero for the
// next launch on this state buffer (this kernel follows the stream kernel in stream order: plain loads / stores are enough here)
__global__ __launch_bounds__(1024) void loss_finalize_kernel(const float2 *__restrict__ partials, const int n,
                                                             float *__restrict__ out, unsigned *__restrict__ bar,
                                                             int32_t *__restrict__ nfg_acc, int32_t *__restrict__ num_fg_out, const int B)
{
    __shared__ long long s[2][1024];                 // the sums in 2^-32 fixed point, like the in-kernel finalize: bit-identical results
    if (bar) {
        for (int b = threadIdx.x; b < B; b += 1024) { num_fg_out[b] = nfg_acc[b]; nfg_acc[b] = 0; }
        if (threadIdx.x == 0) bar[0] = 0u;
    }
    const bool barrier_failed = bar && bar[1] != 0u;               // (uniform: read before the reset below can matter -- only thread 0 writes it, after the sums)
    __shared__ unsigned s_flags;
    if (threadIdx.x == 0) s_flags = 0u;
    __syncthreads();
    long long c = 0, r = 0;
    unsigned fl_c = 0u, fl_r = 0u;
    for (int i = threadIdx.x; i < n; i += 1024) { const float2 v = partials[i]; c += loss_fix(v.x, fl_c); r += loss_fix(v.y, fl_r); }
    if (fl_c | fl_r) atomicOr(&s_flags, fl_c | (fl_r << 8));
    s[0][threadIdx.x] = c; s[1][threadIdx.x] = r;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) { s[0][threadIdx.x] += s[0][threadIdx.x + o]; s[1][threadIdx.x] += s[1][threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        out[0] = barrier_failed ? __builtin_nanf("") : loss_unfix(s[0][0], s_flags & 0xffu);
        out[1] = barrier_failed ? __builtin_nanf("") : loss_unfix(s[1][0], (s_flags >> 8) & 0xffu);
        if (barrier_failed) bar[1] = 0u;
    }
}

template <int DT>
__global__ __launch_bounds__(256) void scale_inplace_kernel(void *data, const int64_t n, const float *__restrict__ scale)
{
    typedef rn::dt<DT> D;
    constexpr int VEC = D::VEC;
    const float s = *scale;
    if (s == 1.0f) return;          // the common case (loss.backward()): no traffic at all
    const int64_t nvec = n / VEC;
    rn::u32x4 *p = (rn::u32x4 *)data;
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (int64_t)gridDim.x * blockDim.x) {
        float f[VEC];
        D::unpack(p[v], f);
#pragma unroll
        for (int j = 0; j < VEC; ++j) f[j] *= s;
        p[v] = D::pack(f);
    }
    if (blockIdx.x == 0) {
        const int64_t i = nvec * VEC + threadIdx.x;
        if (i < n) D::st(data, i, D::ld(data, i) * s);
    }
}

// The same for up to 16 tensors in ONE launch (blockIdx.y = tensor): the gradients of a per-level loss call -- 5 class and
// 5 box tensors, scaled by the two upstream scalars -- cost ten 5 us launches otherwise, for a multiplication by 1.
constexpr int SCALE_MAX_TENSORS = 16;
struct ScaleTable { void *data[SCALE_MAX_TENSORS]; int64_t n[SCALE_MAX_TENSORS]; const float *scale[SCALE_MAX_TENSORS]; };
template <int DT>
__global__ __launch_bounds__(256) void scale_inplace_batched_kernel(const ScaleTable t)
{
    typedef rn::dt<DT> D;
    constexpr int VEC = D::VEC;
    const int k = blockIdx.y;
    const float s = *t.scale[k];
    if (s == 1.0f) return;
    const int64_t n = t.n[k], nvec = n / VEC;
    rn::u32x4 *p = (rn::u32x4 *)t.data[k];
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (int64_t)gridDim.x * blockDim.x) {
        float f[VEC];
        D::unpack(p[v], f);
#pragma unroll
        for (int j = 0; j < VEC; ++j) f[j] *= s;
        p[v] = D::pack(f);
    }
    if (blockIdx.x == 0) {
        const int64_t i = nvec * VEC + threadIdx.x;
        if (i < n) D::st(t.data[k], i, D::ld(t.data[k], i) * s);
    }
}

// Grid = what is co-resident (CUs x blocks/CU from the occupancy query), never more: a second,
// partially filled round of blocks would idle most of the chip for a whole block lifetime.
template <typename KernelT>
int resident_blocks(KernelT kernel, int *out, const int per_cu_cap = 1 << 30)
{
    int dev = 0, cus = 0, per_cu = 0;
    RN_HIP(hipGetDevice(&dev));
    RN_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    RN_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, LOSS_BLOCK, 0));
    if (per_cu > per_cu_cap) per_cu = per_cu_cap;
    int n = cus * per_cu;
    if (n < 1) n = 1;
    if (n > LOSS_MAX_BLOCKS) n = LOSS_MAX_BLOCKS;
    *out = n;
    return RN_OK;
}

// optional profiling: HIP events recorded on the launch stream right around the stream kernel (not the finalize)
struct ProfileEvents { hipEvent_t start, stop; };
static thread_local ProfileEvents g_prof = {nullptr, nullptr};     // set only for the duration of one rn_loss_fwd_bwd_levels_timed call

template <typename StreamT>
int launch_stream(StreamT stream_k, LossArgs &a, int vec, hipStream_t st, int *n_stream, const bool fused = false)
{
    int res = 0;
    // fused matching needs EVERY workgroup resident (grid barrier).  The occupancy query is an estimate: the value-only variant (64
    // VGPRs -> 8 workgroups per CU by registers, 8 x 19 232 B of LDS = 154 KB "fits" 160 KB) was answered 8 and the hardware held
    // fewer (LDS allocation granularity) -- the barrier timed out.  6 per CU is what the gradient variant (the measured one) runs at
    // and leaves 45 KB of LDS slack.
    int rc = resident_blocks(stream_k, &res, fused ? 6 : (1 << 30));
    if (rc != RN_OK) return rc;
    // even split of the vectors over the resident waves, in whole wave-iterations (64 vectors = 1 KiB)
    const int64_t nvec = a.total_vec;
    const int64_t waves = (int64_t)res * LOSS_WAVES;
    int64_t vpw = (nvec + waves - 1) / waves;
    // whole groups of 2 wave-iterations: the < 1 group leftover of a range goes through a slow per-element path, and one
    // slow wave-iteration at the end of EVERY wave is a tail the whole chip waits for
    vpw = ((vpw + 2 * RN_WAVE - 1) / (2 * RN_WAVE)) * (2 * RN_WAVE);
    if (vpw < 2 * RN_WAVE) vpw = 2 * RN_WAVE;
    a.vec_per_wave = vpw;
    {
        // default 1: in the train step the class-output conv has just written the logits, column pass by column pass; the
        // Infinity Cache (256 MiB) still holds the last three passes and the tail rows of the first, and a back-to-front walk
        // reads exactly those first: 131 -> 117 us on one box (round-2 A/B of the two orders); no effect on cold data
        a.reverse = 1;
    }
    int64_t need = ((nvec + vpw - 1) / vpw + LOSS_WAVES - 1) / LOSS_WAVES;
    if (need < 1) need = 1;
    if (fused) {
        // every workgroup must be resident (grid barrier), and the match codes of a wave's rows must fit its LDS list: a range of vpw
        // vectors (+ a ragged tail of < VEC elements per level) covers at most elements / K rows + 2 partial rows per level segment
        if (need > res) return RN_EUNSUPPORTED;
        if ((vpw * vec + (int64_t)a.L * (vec - 1)) / a.K + 2 * (int64_t)a.L + 1 > LIST_CAP_FUSED) return RN_EUNSUPPORTED;
    }
    if (g_prof.start) RN_HIP(hipEventRecord(g_prof.start, st));
    bool launched = false;
    if (fused) {
        // the grid barrier needs every workgroup on the chip: a COOPERATIVE launch makes the runtime check that (it refuses a grid that
        // is not co-resident: RN_EUNSUPPORTED up front, the caller takes the two-launch path) instead of this library trusting its
        // occupancy estimate and the kernel timing out into NaN losses.  Under stream capture the plain launch is kept (its grid was
        // sized from the occupancy query above, and the kernel's bounded barrier still fails loudly).
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(st, &cs) != hipSuccess) { (void)hipGetLastError(); cs = hipStreamCaptureStatusNone; }
        if (cs == hipStreamCaptureStatusNone) {
            void *kargs[] = {(void *)&a};
            const hipError_t e = hipLaunchCooperativeKernel((const void *)stream_k, dim3((unsigned)need), dim3(LOSS_BLOCK), kargs, 0, st);
            if (e == hipErrorCooperativeLaunchTooLarge || e == hipErrorInvalidConfiguration) { (void)hipGetLastError(); return RN_EUNSUPPORTED; }
            if (e != hipSuccess) return (int)e;
            launched = true;
        }
    }
    if (!launched) hipLaunchKernelGGL(stream_k, dim3((unsigned)need), dim3(LOSS_BLOCK), 0, st, a);
    RN_LAUNCH_CHECK();
    if (g_prof.stop) RN_HIP(hipEventRecord(g_prof.stop, st));
    *n_stream = (int)need;
    return RN_OK;
}

template <int DT>
int launch_loss(LossArgs &a, bool gamma2, bool wg, hipStream_t st, int *ns, const bool fused = false, const bool list = false)
{
    constexpr int VEC = rn::dt<DT>::VEC;
    // PF = 2 groups of loads in flight, non-temporal loads: the best of the (2|4|8) x (nt|plain) sweep on MI355X
    int n_stream = 0, rc;
    if (fused) {
        if (gamma2) {
            if (wg) rc = launch_stream(loss_stream_kernel<DT, true, true, 2, 1, true>, a, VEC, st, &n_stream, true);
            else rc = launch_stream(loss_stream_kernel<DT, true, false, 2, 1, true>, a, VEC, st, &n_stream, true);
        } else {
            if (wg) rc = launch_stream(loss_stream_kernel<DT, false, true, 2, 1, true>, a, VEC, st, &n_stream, true);
            else rc = launch_stream(loss_stream_kernel<DT, false, false, 2, 1, true>, a, VEC, st, &n_stream, true);
        }
    } else if (list) {
        // the special rows of ALL chunks through one compact list (many special rows per wave: RN_LOSS_FORM_LIST)
        if (gamma2) {
            if (wg) rc = launch_stream(loss_stream_kernel<DT, true, true, 2, 1, false, true>, a, VEC, st, &n_stream);
            else rc = launch_stream(loss_stream_kernel<DT, true, false, 2, 1, false, true>, a, VEC, st, &n_stream);
        } else {
            if (wg) rc = launch_stream(loss_stream_kernel<DT, false, true, 2, 1, false, true>, a, VEC, st, &n_stream);
            else rc = launch_stream(loss_stream_kernel<DT, false, false, 2, 1, false, true>, a, VEC, st, &n_stream);
        }
    } else if (gamma2) {
        if (wg) rc = launch_stream(loss_stream_kernel<DT, true, true, 2, 1>, a, VEC, st, &n_stream);
        else rc = launch_stream(loss_stream_kernel<DT, true, false, 2, 1>, a, VEC, st, &n_stream);
    } else {
        if (wg) rc = launch_stream(loss_stream_kernel<DT, false, true, 2, 1>, a, VEC, st, &n_stream);
        else rc = launch_stream(loss_stream_kernel<DT, false, false, 2, 1>, a, VEC, st, &n_stream);
    }
    if (rc != RN_OK) return rc;
    *ns = n_stream;
    return RN_OK;
}

template <int DT>
int launch_loss_rp(LossArgs &a, bool gamma2, bool wg, hipStream_t st)
{
    constexpr int VEC = rn::dt<DT>::VEC;
    int n_stream = 0, rc;
    if (gamma2) {
        if (wg) rc = launch_stream(loss_bg_kernel<DT, true, true, 2, 1>, a, VEC, st, &n_stream);
        else rc = launch_stream(loss_bg_kernel<DT, true, false, 2, 1>, a, VEC, st, &n_stream);
    } else {
        if (wg) rc = launch_stream(loss_bg_kernel<DT, false, true, 2, 1>, a, VEC, st, &n_stream);
        else rc = launch_stream(loss_bg_kernel<DT, false, false, 2, 1>, a, VEC, st, &n_stream);
    }
    if (rc != RN_OK) return rc;
    const int spi = (int)((a.special_W + REPAIR_WORDS - 1) / REPAIR_WORDS);
    const int64_t total = (int64_t)a.B * spi;
    if (total >= ((int64_t)1 << 30)) return RN_EUNSUPPORTED;
    const dim3 grid((unsigned)((total + REPAIR_WAVES - 1) / REPAIR_WAVES));
    if (gamma2) {
        if (wg) hipLaunchKernelGGL((loss_repair_kernel<DT, true, true>), grid, dim3(REPAIR_BLOCK), 0, st, a, spi, (int)total);
        else hipLaunchKernelGGL((loss_repair_kernel<DT, true, false>), grid, dim3(REPAIR_BLOCK), 0, st, a, spi, (int)total);
    } else {
        if (wg) hipLaunchKernelGGL((loss_repair_kernel<DT, false, true>), grid, dim3(REPAIR_BLOCK), 0, st, a, spi, (int)total);
        else hipLaunchKernelGGL((loss_repair_kernel<DT, false, false>), grid, dim3(REPAIR_BLOCK), 0, st, a, spi, (int)total);
    }
    RN_LAUNCH_CHECK();
    return RN_OK;
}

}  // namespace

RN_API size_t rn_loss_workspace_bytes(int B, int64_t A, int K)
{
    (void)B; (void)A; (void)K;
    return sizeof(float2) * (size_t)LOSS_MAX_BLOCKS * 2 + 16;
}

// fused matching: thresholds, optional match codes out, num_fg out, and the state words (zero on entry, zero on exit)
struct FusedMatch { float fg_thr, bg_thr; int64_t *matches_out; int32_t *num_fg_out; void *state; };

static int loss_levels_core(const void *const *cls_levels, const void *const *box_levels,
                            const int64_t *level_anchors, int L, int dtype, int B, int K,
                            const float *anchors, int64_t anchor_bstride, const float *gt_boxes,
                            const int64_t *gt_labels, const int32_t *gt_off, const int64_t *matches, const uint64_t *special_rows,
                            const int32_t *num_fg, const rn_loss_params *params, float *out_loss,
                            void *const *grad_cls_levels, void *const *grad_box_levels, void *workspace,
                            size_t workspace_bytes, void *stream, const FusedMatch *fm = nullptr, void *fin_state = nullptr,
                            const float *grad_prescale = nullptr, const int form = RN_LOSS_FORM_CHUNKS)
{
    if (form != RN_LOSS_FORM_CHUNKS && form != RN_LOSS_FORM_REPAIR_PASS && form != RN_LOSS_FORM_LIST) return RN_EINVAL;
    const bool repair_pass = form == RN_LOSS_FORM_REPAIR_PASS, list = form == RN_LOSS_FORM_LIST && !fm;
    if (fin_state && (fm || !rn::aligned(fin_state, 64))) return fm ? RN_EINVAL : RN_EALIGN;
    if (repair_pass && (!fin_state || !special_rows)) return RN_EINVAL;
    if (!cls_levels || !box_levels || !level_anchors || !anchors || !gt_off || !params || !out_loss || (!workspace && !repair_pass)) return RN_EINVAL;
    if (!fm && (!matches || !num_fg)) return RN_EINVAL;
    if (L <= 0 || L > RN_MAX_LEVELS || B <= 0 || K <= 0) return RN_EINVAL;
    if ((grad_cls_levels == nullptr) != (grad_box_levels == nullptr)) return RN_EINVAL;
    if (dtype != RN_F32 && dtype != RN_BF16 && dtype != RN_F16) return RN_EINVAL;
    if (K > 4096) return RN_EUNSUPPORTED;
    if (!repair_pass && workspace_bytes < rn_loss_workspace_bytes(B, 1, K)) return RN_EWORKSPACE;
    const size_t box_al = (dtype == RN_F32) ? 16 : 8;
    const int vec = (dtype == RN_F32) ? 4 : 8;
    if (!rn::aligned(anchors, 16) || (gt_boxes && !rn::aligned(gt_boxes, 16)) || (workspace && !rn::aligned(workspace, 16)) || (anchor_bstride & 3))
        return RN_EALIGN;
    if (grad_prescale && !rn::aligned(grad_prescale, 4)) return RN_EALIGN;

    LossArgs a;
    a.L = L;
    int64_t A = 0, voff = 0;
    for (int l = 0; l < RN_MAX_LEVELS; ++l) {
        LossLevel &lv = a.lv[l];
        if (l < L) {
            if (!cls_levels[l] || !box_levels[l] || level_anchors[l] <= 0) return RN_EINVAL;
            if (grad_cls_levels && (!grad_cls_levels[l] || !grad_box_levels[l])) return RN_EINVAL;
            if (!rn::aligned(cls_levels[l], 16) || !rn::aligned(box_levels[l], box_al) ||
                (grad_cls_levels && (!rn::aligned(grad_cls_levels[l], 16) || !rn::aligned(grad_box_levels[l], box_al))))
                return RN_EALIGN;
            lv.cls = cls_levels[l]; lv.box = box_levels[l];
            lv.gcls = grad_cls_levels ? grad_cls_levels[l] : nullptr;
            lv.gbox = grad_box_levels ? grad_box_levels[l] : nullptr;
            lv.A_l = level_anchors[l]; lv.base = A;
            lv.N = (int64_t)B * lv.A_l * K; lv.nvec = lv.N / vec; lv.voff = voff;
            lv.per_image = lv.A_l * (int64_t)K;
            if ((int64_t)B * lv.A_l >= ((int64_t)1 << 31)) return RN_EUNSUPPORTED;
            A += lv.A_l; voff += lv.nvec;
        } else {
            lv.cls = lv.box = nullptr; lv.gcls = lv.gbox = nullptr;
            lv.A_l = 1; lv.base = 0; lv.N = 0; lv.nvec = 0; lv.voff = voff; lv.per_image = K;
        }
    }
    a.total_vec = voff;
    a.anchors = (const rn::f32x4 *)anchors; a.anchor_bstride4 = anchor_bstride / 4;
    a.gt_boxes = (const rn::f32x4 *)gt_boxes; a.gt_labels = gt_labels; a.gt_off = gt_off;
    a.matches = matches; a.num_fg = num_fg;
    if (special_rows && !rn::aligned(special_rows, 8)) return RN_EALIGN;
    a.special = (const unsigned long long *)special_rows; a.special_W = (A + 63) >> 6;
    a.A = A; a.K = K; a.B = B;
    a.vec_per_wave = RN_WAVE;
    a.inv_B = 1.0f / (float)B;
    a.p = *params;
    a.alpha_pos = (float)(1.0 - (double)params->alpha);
    a.part_stream = (float2 *)workspace;
    a.fin = fin_state ? (unsigned *)fin_state + 64 : nullptr;           // 64 cache lines from byte 256 on (words 0, 1: the fused form's barrier words)
    a.out_loss = out_loss;
    a.gscale = grad_prescale;
    a.fg_thr = a.bg_thr = 0.0f; a.nfg_acc = nullptr; a.bar = nullptr; a.matches_out = nullptr;
    if (fm) {
        a.fg_thr = fm->fg_thr; a.bg_thr = fm->bg_thr; a.matches_out = fm->matches_out;
        a.bar = (unsigned *)fm->state; a.nfg_acc = (int32_t *)fm->state + 16;        // (counters one cache-line half away from the barrier word)
    }
    if (grad_cls_levels)
        for (int l = 0; l < L; ++l)
            if (grad_cls_levels[l] == cls_levels[l]) return RN_EINVAL;      // the repair phase re-reads logits the stream has passed
    const bool gamma2 = params->gamma == 2.0f;
    const bool wg = grad_cls_levels != nullptr;
    hipStream_t st = (hipStream_t)stream;
    int ns = 0, rc;
    if (repair_pass) {
        // two launches: the pure background stream, then the repair of the special rows + the finalize.  The caller's event pair
        // brackets BOTH (the whole of K3's device time): the stop event is recorded here, behind the repair kernel.
        const hipEvent_t stop = g_prof.stop;
        g_prof.stop = nullptr;
        switch (dtype) {
            case RN_F32: rc = launch_loss_rp<RN_F32>(a, gamma2, wg, st); break;
            case RN_BF16: rc = launch_loss_rp<RN_BF16>(a, gamma2, wg, st); break;
            default: rc = launch_loss_rp<RN_F16>(a, gamma2, wg, st); break;
        }
        if (rc != RN_OK) return rc;
        if (stop) RN_HIP(hipEventRecord(stop, st));
        return RN_OK;
    }
    switch (dtype) {
        case RN_F32: rc = launch_loss<RN_F32>(a, gamma2, wg, st, &ns, fm != nullptr, list); break;
        case RN_BF16: rc = launch_loss<RN_BF16>(a, gamma2, wg, st, &ns, fm != nullptr, list); break;
        default: rc = launch_loss<RN_F16>(a, gamma2, wg, st, &ns, fm != nullptr, list); break;
    }
    if (rc != RN_OK) return rc;
    if (a.fin) return RN_OK;                                            // (workgroup 0 of the stream kernel has written out_loss)
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(1024), 0, st, (const float2 *)a.part_stream, ns, out_loss, a.bar, a.nfg_acc,
                       fm ? fm->num_fg_out : nullptr, B);
    RN_LAUNCH_CHECK();
    return RN_OK;
}

RN_API int rn_loss_fwd_bwd_levels(const void *const *cls_levels, const void *const *box_levels,
                                  const int64_t *level_anchors, int L, int dtype, int B, int K,
                                  const float *anchors, int64_t anchor_bstride, const float *gt_boxes,
                                  const int64_t *gt_labels, const int32_t *gt_off, const int64_t *matches,
                                  const int32_t *num_fg, const rn_loss_params *params, float *out_loss,
                                  void *const *grad_cls_levels, void *const *grad_box_levels, void *workspace,
                                  size_t workspace_bytes, void *stream)
{
    return loss_levels_core(cls_levels, box_levels, level_anchors, L, dtype, B, K, anchors, anchor_bstride, gt_boxes, gt_labels, gt_off,
                            matches, nullptr, num_fg, params, out_loss, grad_cls_levels, grad_box_levels, workspace, workspace_bytes, stream);
}

RN_API int rn_loss_fwd_bwd_levels_ex(const void *const *cls_levels, const void *const *box_levels,
                                     const int64_t *level_anchors, int L, int dtype, int B, int K,
                                     const float *anchors, int64_t anchor_bstride, const float *gt_boxes,
                                     const int64_t *gt_labels, const int32_t *gt_off, const int64_t *matches,
                                     const uint64_t *special_rows, const int32_t *num_fg, const rn_loss_params *params,
                                     float *out_loss, void *const *grad_cls_levels, void *const *grad_box_levels, void *workspace,
                                     size_t workspace_bytes, void *stream, void *event_start, void *event_stop)
{
    g_prof.start = (hipEvent_t)event_start; g_prof.stop = (hipEvent_t)event_stop;
    const int rc = loss_levels_core(cls_levels, box_levels, level_anchors, L, dtype, B, K, anchors, anchor_bstride, gt_boxes, gt_labels,
                                    gt_off, matches, special_rows, num_fg, params, out_loss, grad_cls_levels, grad_box_levels, workspace,
                                    workspace_bytes, stream);
    g_prof.start = g_prof.stop = nullptr;
    return rc;
}

RN_API int rn_loss_fwd_bwd_levels_fin(const void *const *cls_levels, const void *const *box_levels,
                                      const int64_t *level_anchors, int L, int dtype, int B, int K,
                                      const float *anchors, int64_t anchor_bstride, const float *gt_boxes,
                                      const int64_t *gt_labels, const int32_t *gt_off, const int64_t *matches,
                                      const uint64_t *special_rows, const int32_t *num_fg, const rn_loss_params *params,
                                      float *out_loss, void *const *grad_cls_levels, void *const *grad_box_levels, void *workspace,
                                      size_t workspace_bytes, void *state, void *stream, void *event_start, void *event_stop)
{
    if (!state) return RN_EINVAL;
    g_prof.start = (hipEvent_t)event_start; g_prof.stop = (hipEvent_t)event_stop;
    const int rc = loss_levels_core(cls_levels, box_levels, level_anchors, L, dtype, B, K, anchors, anchor_bstride, gt_boxes, gt_labels,
                                    gt_off, matches, special_rows, num_fg, params, out_loss, grad_cls_levels, grad_box_levels, workspace,
                                    workspace_bytes, stream, nullptr, state);
    g_prof.start = g_prof.stop = nullptr;
    return rc;
}

RN_API int rn_loss_fwd_bwd_levels_rp(const void *const *cls_levels, const void *const *box_levels,
                                     const int64_t *level_anchors, int L, int dtype, int B, int K,
                                     const float *anchors, int64_t anchor_bstride, const float *gt_boxes,
                                     const int64_t *gt_labels, const int32_t *gt_off, const int64_t *matches,
                                     const uint64_t *special_rows, const int32_t *num_fg, const rn_loss_params *params,
                                     const float *grad_prescale, int form, float *out_loss, void *const *grad_cls_levels,
                                     void *const *grad_box_levels, void *workspace, size_t workspace_bytes, void *state,
                                     void *stream, void *event_start, void *event_stop)
{
    if (!state) return RN_EINVAL;
    g_prof.start = (hipEvent_t)event_start; g_prof.stop = (hipEvent_t)event_stop;
    const int rc = loss_levels_core(cls_levels, box_levels, level_anchors, L, dtype, B, K, anchors, anchor_bstride, gt_boxes, gt_labels,
                                    gt_off, matches, special_rows, num_fg, params, out_loss, grad_cls_levels, grad_box_levels, workspace,
                                    workspace_bytes, stream, nullptr, state, grad_prescale, form);
    g_prof.start = g_prof.stop = nullptr;
    return rc;
}

RN_API size_t rn_loss_match_state_bytes(int B) { return B > 0 ? (size_t)(16 + B) * sizeof(int32_t) : 0; }

RN_API int rn_loss_match_fwd_bwd_levels(const void *const *cls_levels, const void *const *box_levels,
                                        const int64_t *level_anchors, int L, int dtype, int B, int K,
                                        const float *anchors, int64_t anchor_bstride, const float *gt_boxes,
                                        const int64_t *gt_labels, const int32_t *gt_off, int max_gt_per_image,
                                        float fg_thr, float bg_thr, int64_t *matches_out, int32_t *num_fg_out,
                                        const rn_loss_params *params, float *out_loss, void *const *grad_cls_levels,
                                        void *const *grad_box_levels, void *workspace, size_t workspace_bytes, void *state,
                                        size_t state_bytes, void *stream, void *event_start, void *event_stop)
{
    if (!num_fg_out || !state || B <= 0 || max_gt_per_image < 0) return RN_EINVAL;
    if (!(fg_thr > bg_thr)) return RN_ETHRESH;
    if (max_gt_per_image > RN_WAVE) return RN_EUNSUPPORTED;            // one GT box per lane: larger sets go through rn_iou_match + rn_loss_fwd_bwd_levels
    if (max_gt_per_image > 0 && !gt_boxes) return RN_EINVAL;
    if (state_bytes < rn_loss_match_state_bytes(B)) return RN_EWORKSPACE;
    if (!rn::aligned(state, 64)) return RN_EALIGN;
    const FusedMatch fm = {fg_thr, bg_thr, matches_out, num_fg_out, state};
    g_prof.start = (hipEvent_t)event_start; g_prof.stop = (hipEvent_t)event_stop;
    const int rc = loss_levels_core(cls_levels, box_levels, level_anchors, L, dtype, B, K, anchors, anchor_bstride, gt_boxes, gt_labels,
                                    gt_off, nullptr, nullptr, nullptr, params, out_loss, grad_cls_levels, grad_box_levels, workspace,
                                    workspace_bytes, stream, &fm);
    g_prof.start = g_prof.stop = nullptr;
    return rc;
}

RN_API int rn_loss_fwd_bwd_levels_timed(const void *const *cls_levels, const void *const *box_levels,
                                        const int64_t *level_anchors, int L, int dtype, int B, int K,
                                        const float *anchors, int64_t anchor_bstride, const float *gt_boxes,
                                        const int64_t *gt_labels, const int32_t *gt_off, const int64_t *matches,
                                        const int32_t *num_fg, const rn_loss_params *params, float *out_loss,
                                        void *const *grad_cls_levels, void *const *grad_box_levels, void *workspace,
                                        size_t workspace_bytes, void *stream, void *event_start, void *event_stop)
{
    g_prof.start = (hipEvent_t)event_start; g_prof.stop = (hipEvent_t)event_stop;
    const int rc = rn_loss_fwd_bwd_levels(cls_levels, box_levels, level_anchors, L, dtype, B, K, anchors, anchor_bstride, gt_boxes,
                                          gt_labels, gt_off, matches, num_fg, params, out_loss, grad_cls_levels, grad_box_levels,
                                          workspace, workspace_bytes, stream);
    g_prof.start = g_prof.stop = nullptr;
    return rc;
}

RN_API int rn_loss_fwd_bwd(const void *cls, const void *box, int dtype, int B, int64_t A, int K,
                           const float *anchors, int64_t anchor_bstride, const float *gt_boxes,
                           const int64_t *gt_labels, const int32_t *gt_off, const int64_t *matches,
                           const int32_t *num_fg, const rn_loss_params *params, float *out_loss, void *grad_cls,
                           void *grad_box, void *workspace, size_t workspace_bytes, void *stream)
{
    if (!cls || !box || A <= 0) return RN_EINVAL;
    if ((grad_cls == nullptr) != (grad_box == nullptr)) return RN_EINVAL;
    const void *c1[1] = {cls}, *b1[1] = {box};
    void *gc1[1] = {grad_cls}, *gb1[1] = {grad_box};
    const int64_t a1[1] = {A};
    return rn_loss_fwd_bwd_levels(c1, b1, a1, 1, dtype, B, K, anchors, anchor_bstride, gt_boxes, gt_labels, gt_off, matches,
                                  num_fg, params, out_loss, grad_cls ? gc1 : nullptr, grad_cls ? gb1 : nullptr, workspace,
                                  workspace_bytes, stream);
}

RN_API int rn_scale_inplace(void *data, int dtype, int64_t n, const float *scale, void *stream)
{
    if (!data || !scale || n < 0) return RN_EINVAL;
    if (!rn::aligned(data, 16)) return RN_EALIGN;
    if (n == 0) return RN_OK;
    int64_t blocks = (n / 4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipStream_t st = (hipStream_t)stream;
    switch (dtype) {
        case RN_F32: hipLaunchKernelGGL((scale_inplace_kernel<RN_F32>), dim3((unsigned)blocks), dim3(256), 0, st, data, n, scale); break;
        case RN_BF16: hipLaunchKernelGGL((scale_inplace_kernel<RN_BF16>), dim3((unsigned)blocks), dim3(256), 0, st, data, n, scale); break;
        case RN_F16: hipLaunchKernelGGL((scale_inplace_kernel<RN_F16>), dim3((unsigned)blocks), dim3(256), 0, st, data, n, scale); break;
        default: return RN_EINVAL;
    }
    RN_LAUNCH_CHECK();
    return RN_OK;
}

RN_API int rn_scale_inplace_batched(void *const *data, const int64_t *n, const float *const *scales, int count, int dtype, void *stream)
{
    if (!data || !n || !scales || count < 0 || count > SCALE_MAX_TENSORS) return RN_EINVAL;
    if (dtype != RN_F32 && dtype != RN_BF16 && dtype != RN_F16) return RN_EINVAL;
    if (count == 0) return RN_OK;
    ScaleTable t = {};
    int64_t most = 0;
    for (int k = 0; k < SCALE_MAX_TENSORS; ++k) {
        const int q = k < count ? k : 0;
        if (!data[q] || !scales[q] || n[q] < 0) return RN_EINVAL;
        if (!rn::aligned(data[q], 16)) return RN_EALIGN;
        t.data[k] = data[q]; t.n[k] = n[q]; t.scale[k] = scales[q];
        if (k < count && n[q] > most) most = n[q];
    }
    int64_t blocks = (most / 4 + 255) / 256;
    blocks = blocks > 1024 ? 1024 : (blocks < 1 ? 1 : blocks);
    const dim3 grid((unsigned)blocks, (unsigned)count);
    hipStream_t st = (hipStream_t)stream;
    switch (dtype) {
        case RN_F32: hipLaunchKernelGGL((scale_inplace_batched_kernel<RN_F32>), grid, dim3(256), 0, st, t); break;
        case RN_BF16: hipLaunchKernelGGL((scale_inplace_batched_kernel<RN_BF16>), grid, dim3(256), 0, st, t); break;
        default: hipLaunchKernelGGL((scale_inplace_batched_kernel<RN_F16>), grid, dim3(256), 0, st, t); break;
    }
    RN_LAUNCH_CHECK();
    return RN_OK;
}
