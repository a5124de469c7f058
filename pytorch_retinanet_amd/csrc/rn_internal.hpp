// Internal (non-ABI) declarations shared between translation units of libretinanet_hip.so.
#pragma once
#include "rn_common.hpp"

namespace rn {

// Order-preserving map float -> uint32 (ascending), then inverted so that an
// ASCENDING sort of the result is a DESCENDING sort of the score.
static __host__ __device__ __forceinline__ uint32_t inv_ordered(const float s) {
    uint32_t u = __builtin_bit_cast(uint32_t, s);
    u ^= (u >> 31) ? 0xffffffffu : 0x80000000u;
    return ~u;
}
static __host__ __device__ __forceinline__ float score_of(const uint32_t inv) {
    uint32_t u = ~inv;
    u ^= (u >> 31) ? 0x80000000u : 0xffffffffu;
    return __builtin_bit_cast(float, u);
}

// Segmented sort + greedy NMS (nms.hip).
//   keys     u64[..]: per segment s, keys[seg_start[s] .. +seg_len[s]) = (inv_ordered(score) << 32) | payload,
//            unsorted on entry (clobbered).
//   boxes    f32x4[..]: box of an entry = boxes[box_base(s) + payload], box_base(s) = box_mode ? (s / K) * A : seg_start[s].
//   kept     u64[..]: on exit kept[seg_start[s] .. +kept_count[s]) = keys of the survivors in sorted order.
//   keep_idx i64[..] (nullable): same positions, payload only.
//   scratch_boxes f32x4[..], scratch_supp u8[..]: same indexing as keys; used by segments longer than 2048.
//   max_keep: see the field.
struct NmsLaunch {
    uint64_t *keys;
    uint64_t *kept;
    int64_t *keep_idx;
    const f32x4 *boxes;
    const int64_t *seg_start;
    const int32_t *seg_len;
    int32_t *kept_count;
    f32x4 *scratch_boxes;
    uint8_t *scratch_supp;
    int S;
    int box_mode;
    int K;
    int64_t A;
    float iou_thr;
    int max_keep;            // > 0: a segment's scan may stop once it has kept this many boxes (the detect chain: only the first max_det
                             // survivors of a class can reach the image's top max_det, models.py:222-240); kept_count is then >= max_keep
                             // for such a segment and the kept list holds its best survivors in order.  0: the full greedy scan (the op).
};
int launch_nms(const NmsLaunch &a, hipStream_t st);

}  // namespace rn
