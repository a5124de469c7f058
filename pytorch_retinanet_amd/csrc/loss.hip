// K3 loss_fwd_bwd -- replaces RetinaNetLosses (retinanet/losses.py:19-145) and
// bbox_2_activ (retinanet/box_utils.py:25-34): one streaming pass over the head
// outputs that yields both loss scalars AND their gradients.
//
// Reference semantics kept: logit shift x+1 (Q1), reversed
// alpha (Q2), detached focal weight so d/dx = w*(sigmoid(x+1)-t) (Q3), ignore
// rows (-2) contribute nothing, empty GT => image contributes zero (Q7),
// per-image /clamp(num_fg,1) then mean over images (Q8), smooth-L1 beta form
// (Q10), log(gw/aw + 1e-8) (Q11).
//
// Data movement (HBM-bound; per image A*K*s read + A*K*s written for the class
// tensor, A*4*s + A*4*s for the box tensor, A*8 for matches):
//   - [B,A,K] is treated as B*A rows of K contiguous elements.  Each WAVE owns a
//     contiguous range of rows (a multiple of 8 rows, so its first byte is
//     16-byte aligned for every K and element size) and streams it with 16-byte
//     loads/stores: 1 KiB per wave instruction, fully coalesced.
//   - Row metadata (target class code, 1/(max(num_fg,1)*B)) is staged per wave in LDS
//     (8 B/row) and read back per vector; 4 loads (4 KiB) per wave stay in flight so the
//     stream is bandwidth- not latency-bound; no block barrier inside the loop.
//   - ~99% of wave-iterations touch only plain background rows of one image: those run a
//     select-free body (12 VALU + exp/rcp/log per element); rows with positives, ignored
//     rows, image seams and ragged ends take the general body.
//   - Loss sums: per-lane fp32 accumulators -> wave shuffle reduction -> one
//     partial per block -> a second tiny kernel adds the partials in double in a
//     fixed order (deterministic; no float atomics).
#include "rn_common.hpp"

namespace {

constexpr int LOSS_BLOCK = 256;
constexpr int LOSS_WAVES = LOSS_BLOCK / RN_WAVE;
constexpr int LOSS_MAX_BLOCKS = 4096;   // upper bound on resident blocks (256 CUs x 8) with headroom; sizes the partials workspace

struct LossArgs {
    const void *cls, *box;
    void *gcls, *gbox;
    const rn::f32x4 *anchors;
    int64_t anchor_bstride4;
    const rn::f32x4 *gt_boxes;
    const int64_t *gt_labels;
    const int32_t *gt_off;
    const int64_t *matches;
    const int32_t *num_fg;
    int64_t A, R;            // anchors per image, total rows B*A
    int32_t K, B;
    int64_t rows_per_wave;   // multiple of 8
    uint32_t magicK;         // floor(2^32/K)+1 : (n*magicK)>>32 == n/K for n < 2^32/K
    float inv_B;
    rn_loss_params p;
    float alpha_pos;         // weight of t=1 elements: 1-alpha (Q2)
    float2 *partials;        // [gridDim.x] (cls, reg)
};

// ln(x) for x in [1, 2]: bare v_log_f32 (log2) times ln 2.  __logf() would add ~10 instructions of
// denormal-range handling per call, which dominated the loop.
__device__ __forceinline__ float ln_1to2(const float x) { return __builtin_amdgcn_logf(x) * 0.6931471805599453f; }
// max(z, 0) = (z + |z|) / 2: one add with an |.| source modifier + a multiply (fmaxf costs two
// v_max because of sNaN canonicalisation).
__device__ __forceinline__ float relu(const float z) { return (z + fabsf(z)) * 0.5f; }

// ---- per-element focal term ---------------------------------------------------
// Returns loss and d loss/dx (both unscaled).  t in {0,1} as `pos`.
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
    const float l1p = ln_1to2(den) + (e - (den - 1.0f)) * r;
    const float bce = relu(pos ? -z : z) + l1p;        // (1-t)*z - log_sigmoid(z)
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

// Window of rows whose metadata a wave keeps in LDS (8 B per row, 4 KiB per wave).
constexpr int WIN_ROWS = 512;
constexpr int PF = 4;                       // 16-byte loads in flight per lane (4 KiB per wave)

struct RowMeta { int code; float scale; };  // code: -2 ignore, -1 background, >=0 positive class; scale = 1/(max(nfg,1)*B)

// General element: any row kind; metadata looked up in the wave's LDS window.
template <int DT, bool GAMMA2, bool WRITE_GRAD>
__device__ __forceinline__ void slow_vector(const LossArgs &a, const RowMeta *meta, const int nrows, const rn::u32x4 *src_vec,
                                            const uint32_t le, rn::u32x4 *dst_vec, float &acc_cls)
{
    typedef rn::dt<DT> D;
    constexpr int VEC = D::VEC;
    const int K = a.K;
    float x[VEC], g[VEC];
    D::unpack(*src_vec, x);          // (re)loaded here: the general body is rare, its data is L2-hot
    uint32_t row = (K == 1) ? le : __umulhi(le, a.magicK);
    int k = (int)(le - row * (uint32_t)K);
    RowMeta m = meta[row];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        if (k >= K) {                       // crossed into the next row (possibly several times when K < VEC)
            k = 0;
            row = min(row + 1u, (uint32_t)nrows - 1u);
            m = meta[row];
        }
        float l, gr;
        focal_elem<GAMMA2>(x[j], m.code == k, a, l, gr);
        const bool use = m.code != -2;
        acc_cls += use ? l * m.scale : 0.0f;
        g[j] = use ? gr * m.scale : 0.0f;
        ++k;
    }
    if (WRITE_GRAD) *dst_vec = D::pack(g);
}

// Background-only vector: every element has t = 0 and the same scale (wave-uniform `gmul`).
// With E = exp(-z):  sigmoid(z) = 1/(1+E),  softplus(z) = z + ln(1+E)  -- no |z|, no select.
// z is clamped at -80 so E stays finite (sigmoid(-80) = 1.8e-35; its weight p^2 underflows to 0
// either way).  Per element: 10 VALU + v_exp + v_rcp + v_log.
template <int DT, bool GAMMA2, bool WRITE_GRAD>
__device__ __forceinline__ void fast_vector(const LossArgs &a, const rn::u32x4 raw, const float gmul, rn::u32x4 *dst_vec,
                                            float &acc_fast)
{
    typedef rn::dt<DT> D;
    constexpr int VEC = D::VEC;
    float x[VEC], g[VEC];
    D::unpack(raw, x);
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        const float z = __builtin_amdgcn_fmed3f(x[j] + a.p.logit_shift, -80.0f, __builtin_inff());
        const float den = 1.0f + __builtin_amdgcn_exp2f(z * -1.4426950408889634f);
        const float ps = __builtin_amdgcn_rcpf(den);               // sigmoid(z)
        const float w = GAMMA2 ? ps * ps : ((a.p.gamma == 0.0f) ? 1.0f : __powf(ps, a.p.gamma));
        const float bce = fmaf(__builtin_amdgcn_logf(den), 0.6931471805599453f, z);   // softplus(z)
        acc_fast = fmaf(w, bce, acc_fast);
        g[j] = (w * ps) * gmul;
    }
    if (WRITE_GRAD) *dst_vec = D::pack(g);
}

template <int DT, bool GAMMA2, bool WRITE_GRAD>
__global__ __launch_bounds__(LOSS_BLOCK) void loss_fwd_bwd_kernel(const LossArgs a)
{
    typedef rn::dt<DT> D;
    constexpr int VEC = D::VEC;
    __shared__ RowMeta s_meta[LOSS_WAVES][WIN_ROWS];
    __shared__ float s_part[LOSS_WAVES][2];

    const int lane = threadIdx.x & (RN_WAVE - 1);
    // readfirstlane: tells the compiler the wave index (and every range / trip count derived from
    // it) is wave-uniform, so loop control and address bases live in SGPRs and branches on them
    // are scalar branches instead of EXEC-masked regions.
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t gwave = (int64_t)blockIdx.x * LOSS_WAVES + wave;
    const int64_t rbeg = gwave * a.rows_per_wave;
    const int64_t rend = min(rbeg + a.rows_per_wave, a.R);
    const int K = a.K;
    RowMeta *meta = s_meta[wave];

    float acc_cls = 0.0f, acc_reg = 0.0f;

    for (int64_t w0 = rbeg; w0 < rend; w0 += WIN_ROWS) {
        const int nrows = (int)min((int64_t)WIN_ROWS, rend - w0);
        // first loads of the stream go out before the (dependent) metadata loads
        const int ne = nrows * K;                    // elements in the window
        const int nvec = ne / VEC;                   // full 16-byte vectors
        const int64_t e0 = w0 * (int64_t)K;          // 16-byte aligned (w0 % 8 == 0)
        const rn::u32x4 *src = (const rn::u32x4 *)((const typename D::elem *)a.cls + e0);
        rn::u32x4 *dst = WRITE_GRAD ? (rn::u32x4 *)((typename D::elem *)a.gcls + e0) : nullptr;
        const int iters = nvec / RN_WAVE;            // full wave-iterations (64 vectors = 1 KiB each)
        const rn::u32x4 zero4 = {0u, 0u, 0u, 0u};
        rn::u32x4 q[PF];
#pragma unroll
        for (int u = 0; u < PF; ++u) q[u] = (nvec > 0) ? src[min(u * RN_WAVE + lane, nvec - 1)] : zero4;

        // ---- phase 1: row metadata -> LDS, regression term + box gradients ------------
        for (int i = lane; i < nrows; i += RN_WAVE) {
            const int64_t r = w0 + i;
            const int b = (int)((uint32_t)r / (uint32_t)a.A);        // R < 2^31 (checked on the host)
            const int64_t ai = r - (int64_t)b * a.A;
            const int64_t m = a.matches[r];
            const int nf = a.num_fg[b];
            RowMeta rm;
            rm.scale = (1.0f / (float)(nf > 1 ? nf : 1)) * a.inv_B;
            float gb[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            if (m >= 0) {
                const int gi = a.gt_off[b] + (int)m;
                rm.code = (int)a.gt_labels[gi] - 1;
                float pred[4];
                box4<DT>::ld(a.box, r, pred);
                const float l = reg_row(a.gt_boxes[gi], a.anchors[(int64_t)b * a.anchor_bstride4 + ai], pred, a.p, gb);
                acc_reg += l * rm.scale;
#pragma unroll
                for (int j = 0; j < 4; ++j) gb[j] *= rm.scale;
            } else {
                rm.code = (m == -1) ? -1 : -2;
            }
            meta[i] = rm;
            if (WRITE_GRAD) box4<DT>::st(a.gbox, r, gb);
        }
        // LDS traffic of one wave is processed in order; the fence keeps the compiler from moving
        // the metadata reads above the writes (no block barrier: waves run independent trip counts).
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        // ---- phase 2: stream the class logits, PF loads in flight per lane ------------------
        // Straight-line chunks of PF iterations: every refill load is unconditional (index clamped
        // to the window's last vector) so the compiler can count outstanding loads and wait with
        // vmcnt(N) for just the oldest one instead of draining the ring with vmcnt(0).
        const int last_v = nvec - 1;
        int it = 0;
        for (; it + PF <= iters; it += PF) {
            const int v0 = it * RN_WAVE + lane;
            // next chunk's loads go out first: they have this whole chunk's compute to land
            rn::u32x4 qn[PF];
#pragma unroll
            for (int u = 0; u < PF; ++u) qn[u] = src[min(v0 + (PF + u) * RN_WAVE, last_v)];
            // Is every row touched by these PF iterations plain background with one scale?
            const uint32_t e_lo = (uint32_t)it * RN_WAVE * VEC, e_hi = (uint32_t)(it + PF) * RN_WAVE * VEC - 1u;
            const uint32_t row_lo = (K == 1) ? e_lo : __umulhi(e_lo, a.magicK);
            const uint32_t row_hi = (K == 1) ? e_hi : __umulhi(e_hi, a.magicK);
            const uint32_t nr = row_hi - row_lo + 1u;
            bool fast = false;
            float sc_u = 0.0f;
            if (nr <= (uint32_t)RN_WAVE) {
                const RowMeta mine = meta[row_lo + min((uint32_t)lane, nr - 1u)];
                sc_u = meta[row_lo].scale;
                fast = __all(mine.code == -1 && mine.scale == sc_u);
            }
            if (fast) {
                const float gmul = a.p.alpha * sc_u;                      // alpha / (max(nfg,1) * B)
                float acc_fast = 0.0f;
#pragma unroll
                for (int u = 0; u < PF; ++u)
                    fast_vector<DT, GAMMA2, WRITE_GRAD>(a, q[u], gmul, WRITE_GRAD ? dst + v0 + u * RN_WAVE : nullptr, acc_fast);
                acc_cls = fmaf(acc_fast, gmul, acc_cls);
            } else {
#pragma unroll 1
                for (int u = 0; u < PF; ++u)
                    slow_vector<DT, GAMMA2, WRITE_GRAD>(a, meta, nrows, src + v0 + u * RN_WAVE, (uint32_t)(v0 + u * RN_WAVE) * VEC,
                                                        WRITE_GRAD ? dst + v0 + u * RN_WAVE : nullptr, acc_cls);
            }
#pragma unroll
            for (int u = 0; u < PF; ++u) q[u] = qn[u];
        }
        // leftover full iterations (< PF), the partial one, then the ragged tail: general body
#pragma unroll 1
        for (int v = it * RN_WAVE + lane; v < nvec; v += RN_WAVE)
            slow_vector<DT, GAMMA2, WRITE_GRAD>(a, meta, nrows, src + v, (uint32_t)v * VEC, WRITE_GRAD ? dst + v : nullptr, acc_cls);
        // ---- partial last iteration (< 64 vectors) and ragged tail (< VEC elements) ---------
        {
            const uint32_t le = (uint32_t)(nvec * VEC + lane);
            if ((int)le < ne) {
                const uint32_t row = le / (uint32_t)K;
                const int k = (int)(le - row * (uint32_t)K);
                const RowMeta m = meta[row];
                float l, gr;
                focal_elem<GAMMA2>(D::ld(a.cls, e0 + le), m.code == k, a, l, gr);
                const bool use = m.code != -2;
                acc_cls += use ? l * m.scale : 0.0f;
                if (WRITE_GRAD) D::st(a.gcls, e0 + le, use ? gr * m.scale : 0.0f);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");             // metadata reads done before the next window's writes
        __builtin_amdgcn_wave_barrier();
    }

    // ---- block partial ----------------------------------------------------------
    acc_cls = rn::wave_sum(acc_cls);
    acc_reg = rn::wave_sum(acc_reg);
    if (lane == 0) { s_part[wave][0] = acc_cls; s_part[wave][1] = acc_reg; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float c = 0.0f, rg = 0.0f;
#pragma unroll
        for (int w = 0; w < LOSS_WAVES; ++w) { c += s_part[w][0]; rg += s_part[w][1]; }
        a.partials[blockIdx.x] = make_float2(c, rg);
    }
}

__global__ __launch_bounds__(256) void loss_finalize_kernel(const float2 *__restrict__ partials, const int n,
                                                            float *__restrict__ out)
{
    __shared__ double s[2][256];
    double c = 0.0, r = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) { c += (double)partials[i].x; r += (double)partials[i].y; }
    s[0][threadIdx.x] = c; s[1][threadIdx.x] = r;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) { s[0][threadIdx.x] += s[0][threadIdx.x + o]; s[1][threadIdx.x] += s[1][threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { out[0] = (float)s[0][0]; out[1] = (float)s[1][0]; }
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

// Grid = what is co-resident (CUs x blocks/CU from the occupancy query), never more: a second,
// partially filled round of blocks would idle most of the chip for a whole block lifetime.  Rows
// are split evenly over the resident waves in multiples of 8 (16-byte alignment of each range).
int loss_grid(int64_t R, int resident_blocks, int64_t *rows_per_wave)
{
    if (resident_blocks < 1) resident_blocks = 1;
    if (resident_blocks > LOSS_MAX_BLOCKS) resident_blocks = LOSS_MAX_BLOCKS;
    const int64_t waves = (int64_t)resident_blocks * LOSS_WAVES;
    int64_t rpw = (R + waves - 1) / waves;
    rpw = ((rpw + 7) / 8) * 8;
    if (rpw < 8) rpw = 8;
    *rows_per_wave = rpw;
    const int64_t need_waves = (R + rpw - 1) / rpw;
    return (int)((need_waves + LOSS_WAVES - 1) / LOSS_WAVES);
}

template <typename KernelT>
int launch_sized(KernelT kernel, LossArgs &a, hipStream_t st, int *blocks_out)
{
    int dev = 0, cus = 0, per_cu = 0;
    RN_HIP(hipGetDevice(&dev));
    RN_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    RN_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, LOSS_BLOCK, 0));
    const int blocks = loss_grid(a.R, cus * per_cu, &a.rows_per_wave);
    hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(LOSS_BLOCK), 0, st, a);
    RN_LAUNCH_CHECK();
    *blocks_out = blocks;
    return RN_OK;
}

template <int DT>
int launch_loss(LossArgs &a, bool gamma2, bool write_grad, hipStream_t st, int *blocks)
{
    if (gamma2) {
        if (write_grad) return launch_sized(loss_fwd_bwd_kernel<DT, true, true>, a, st, blocks);
        return launch_sized(loss_fwd_bwd_kernel<DT, true, false>, a, st, blocks);
    }
    if (write_grad) return launch_sized(loss_fwd_bwd_kernel<DT, false, true>, a, st, blocks);
    return launch_sized(loss_fwd_bwd_kernel<DT, false, false>, a, st, blocks);
}

}  // namespace

RN_API size_t rn_loss_workspace_bytes(int B, int64_t A, int K)
{
    (void)B; (void)A; (void)K;
    return sizeof(float2) * (size_t)LOSS_MAX_BLOCKS;
}

RN_API int rn_loss_fwd_bwd(const void *cls, const void *box, int dtype, int B, int64_t A, int K,
                           const float *anchors, int64_t anchor_bstride, const float *gt_boxes,
                           const int64_t *gt_labels, const int32_t *gt_off, const int64_t *matches,
                           const int32_t *num_fg, const rn_loss_params *params, float *out_loss, void *grad_cls,
                           void *grad_box, void *workspace, size_t workspace_bytes, void *stream)
{
    if (!cls || !box || !anchors || !gt_off || !matches || !num_fg || !params || !out_loss || !workspace) return RN_EINVAL;
    if (B <= 0 || A <= 0 || K <= 0) return RN_EINVAL;
    if ((grad_cls == nullptr) != (grad_box == nullptr)) return RN_EINVAL;
    if (K > 4096 || (int64_t)B * A >= ((int64_t)1 << 31)) return RN_EUNSUPPORTED;
    if (workspace_bytes < rn_loss_workspace_bytes(B, A, K)) return RN_EWORKSPACE;
    const size_t box_al = (dtype == RN_F32) ? 16 : 8;
    if (!rn::aligned(cls, 16) || (grad_cls && !rn::aligned(grad_cls, 16)) || !rn::aligned(box, box_al) ||
        (grad_box && !rn::aligned(grad_box, box_al)) || !rn::aligned(anchors, 16) ||
        (gt_boxes && !rn::aligned(gt_boxes, 16)) || !rn::aligned(workspace, 16) || (anchor_bstride & 3))
        return RN_EALIGN;

    LossArgs a;
    a.cls = cls; a.box = box; a.gcls = grad_cls; a.gbox = grad_box;
    a.anchors = (const rn::f32x4 *)anchors; a.anchor_bstride4 = anchor_bstride / 4;
    a.gt_boxes = (const rn::f32x4 *)gt_boxes; a.gt_labels = gt_labels; a.gt_off = gt_off;
    a.matches = matches; a.num_fg = num_fg;
    a.A = A; a.R = (int64_t)B * A; a.K = K; a.B = B;
    a.magicK = (uint32_t)(((uint64_t)1 << 32) / (uint64_t)K) + 1u;
    a.inv_B = 1.0f / (float)B;
    a.p = *params;
    a.alpha_pos = (float)(1.0 - (double)params->alpha);
    a.partials = (float2 *)workspace;
    int blocks = 0;
    const bool gamma2 = params->gamma == 2.0f;
    const bool wg = grad_cls != nullptr;
    hipStream_t st = (hipStream_t)stream;
    int rc;
    switch (dtype) {
        case RN_F32: rc = launch_loss<RN_F32>(a, gamma2, wg, st, &blocks); break;
        case RN_BF16: rc = launch_loss<RN_BF16>(a, gamma2, wg, st, &blocks); break;
        case RN_F16: rc = launch_loss<RN_F16>(a, gamma2, wg, st, &blocks); break;
        default: return RN_EINVAL;
    }
    if (rc != RN_OK) return rc;
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, st, (const float2 *)workspace, blocks, out_loss);
    RN_LAUNCH_CHECK();
    return RN_OK;
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
