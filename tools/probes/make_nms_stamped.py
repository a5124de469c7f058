#!/usr/bin/env python3
"""Writes scratch/nms_stamped.hip = csrc/nms.hip + clock64() stamps at the phase boundaries of nms_mask_kernel + the read-back entry
point rn_debug_nms_stamps; tools/probes/make_nms_stamped.sh compiles it into tools/probes/libnms_stamped.so (the rest of the library's
objects as built).  Used by tools/probes/nms_stamps_probe.py."""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
s = open(os.path.join(ROOT, "pytorch_retinanet_amd", "csrc", "nms.hip")).read()


def once(old, new):
    global s
    assert s.count(old) == 1, old
    s = s.replace(old, new)


once("namespace {\n\nusing rn::f32x4;",
     "__device__ unsigned long long g_nms_stamps[8192 * 8];\n"
     "#define STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 8192) g_nms_stamps[blockIdx.x * 8 + (i)] = (unsigned long long)clock64(); } while (0)\n\n"
     "namespace {\n\nusing rn::f32x4;")
once("    const int s = blockIdx.x;\n    const int n = a.seg_len[s];\n    if (n > MASK_CAP) return;",
     "    STAMP(0);\n    const int s = blockIdx.x;\n    const int n = a.seg_len[s];\n    if (n > MASK_CAP) return;")
once("    s_in[t] = key;\n    __syncthreads();\n", "    s_in[t] = key;\n    __syncthreads();\n    STAMP(1);\n")
once("    __syncthreads();\n    if (t < n) {\n        const f32x4 b = a.boxes[box_base + (uint32_t)s_key[t]];",
     "    __syncthreads();\n    STAMP(2);\n    if (t < n) {\n        const f32x4 b = a.boxes[box_base + (uint32_t)s_key[t]];")
once("    __syncthreads();\n    // suppression matrix: bit j of s_mask[i][w]", "    __syncthreads();\n    STAMP(3);\n    // suppression matrix: bit j of s_mask[i][w]")
once("        if (lane < MASK_WORDS) s_mask[i][lane] = mine;\n    }\n    __syncthreads();\n    // greedy scan by wave 0",
     "        if (lane < MASK_WORDS) s_mask[i][lane] = mine;\n    }\n    __syncthreads();\n    STAMP(4);\n    // greedy scan by wave 0")
once("    __syncthreads();\n    int before = 0, total = 0;\n#pragma unroll\n    for (int w = 0; w < MASK_WORDS; ++w) {\n        const int c = __popcll(s_keep[w]);",
     "    __syncthreads();\n    STAMP(5);\n    int before = 0, total = 0;\n#pragma unroll\n    for (int w = 0; w < MASK_WORDS; ++w) {\n        const int c = __popcll(s_keep[w]);")
once("    if (t == 0) a.kept_count[s] = total;\n}\n\n// Segments longer than MED_CAP",
     "    if (t == 0) a.kept_count[s] = total;\n    STAMP(6);\n}\n\n// Segments longer than MED_CAP")
s += ("\nRN_API int rn_debug_nms_stamps(unsigned long long *host_out, int n)\n{\n"
      "    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_nms_stamps), sizeof(unsigned long long) * (size_t)n, 0, hipMemcpyDeviceToHost);\n}\n")
os.makedirs(os.path.join(ROOT, "scratch"), exist_ok=True)
open(os.path.join(ROOT, "scratch", "nms_stamped.hip"), "w").write(s)
print("wrote scratch/nms_stamped.hip")
