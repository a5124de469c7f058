// K1 anchors_emit -- replaces retinanet/anchors.py:151-170 (_compute_grid_offsets),
// :172-197 (grid_anchors) and the per-image cat at :228 of the reference.
//
// One thread per anchor, one 16-byte store each (fully coalesced stream of A*16
// bytes, the kernel's only HBM traffic besides the L2-resident cell anchors).
// Bit-exactness: the grid shift is start + i*stride evaluated in double and
// rounded once to fp32 (what torch.arange does on the CPU path), followed by ONE
// fp32 add with the cell anchor; compiled with -ffp-contract=off.
#include "rn_common.hpp"

namespace {

struct AnchorLevels {
    int32_t L;
    int32_t H[RN_MAX_LEVELS], W[RN_MAX_LEVELS], S[RN_MAX_LEVELS], C[RN_MAX_LEVELS];
    int64_t base[RN_MAX_LEVELS + 1];          // prefix count of anchors
    const float *cell[RN_MAX_LEVELS];         // device pointers, f32[C][4]
    double offset;
};

__global__ __launch_bounds__(256) void anchors_emit_kernel(const AnchorLevels lv, rn::f32x4 *__restrict__ out)
{
    const int64_t total = lv.base[lv.L];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int l = 0;
#pragma unroll
        for (int k = 1; k < RN_MAX_LEVELS; ++k)
            if (k < lv.L && i >= lv.base[k]) l = k;
        const uint32_t local = (uint32_t)(i - lv.base[l]);
        const uint32_t C = (uint32_t)lv.C[l], W = (uint32_t)lv.W[l];
        const uint32_t loc = local / C, c = local - loc * C;
        const uint32_t y = loc / W, x = loc - y * W;
        const double S = (double)lv.S[l];
        const double start = lv.offset * S;
        const float sx = (float)(start + (double)x * S);
        const float sy = (float)(start + (double)y * S);
        const rn::f32x4 ca = ((const rn::f32x4 *)lv.cell[l])[c];
        rn::f32x4 o;
        o.x = sx + ca.x; o.y = sy + ca.y; o.z = sx + ca.z; o.w = sy + ca.w;
        out[i] = o;
    }
}

}  // namespace

RN_API int64_t rn_anchors_count(const rn_level *levels, int L)
{
    if (!levels || L <= 0 || L > RN_MAX_LEVELS) return -1;
    int64_t n = 0;
    for (int l = 0; l < L; ++l) n += (int64_t)levels[l].H * levels[l].W * levels[l].num_cell;
    return n;
}

RN_API int rn_anchors_emit(const rn_level *levels, int L, const float *const *cell_anchors,
                           double offset, float *out, void *stream)
{
    if (!levels || !cell_anchors || !out || L <= 0 || L > RN_MAX_LEVELS) return RN_EINVAL;
    AnchorLevels lv;
    lv.L = L;
    lv.offset = offset;
    lv.base[0] = 0;
    for (int l = 0; l < RN_MAX_LEVELS; ++l) {
        if (l < L) {
            if (levels[l].H <= 0 || levels[l].W <= 0 || levels[l].num_cell <= 0 || !cell_anchors[l]) return RN_EINVAL;
            if (!rn::aligned(cell_anchors[l], 16)) return RN_EALIGN;
            lv.H[l] = levels[l].H; lv.W[l] = levels[l].W; lv.S[l] = levels[l].stride; lv.C[l] = levels[l].num_cell;
            lv.cell[l] = cell_anchors[l];
            lv.base[l + 1] = lv.base[l] + (int64_t)levels[l].H * levels[l].W * levels[l].num_cell;
        } else {
            lv.H[l] = lv.W[l] = lv.S[l] = lv.C[l] = 1;
            lv.cell[l] = nullptr;
            lv.base[l + 1] = lv.base[L];
        }
    }
    if (!rn::aligned(out, 16)) return RN_EALIGN;
    const int64_t total = lv.base[L];
    if (total >= (int64_t)1 << 32) return RN_EUNSUPPORTED;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(anchors_emit_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, lv,
                       (rn::f32x4 *)out);
    RN_LAUNCH_CHECK();
    return RN_OK;
}
