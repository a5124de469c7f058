"""Writes scratch/conv_stamps.hip: csrc/conv.hip with s_memrealtime stamps in conv3x3_canvas_kernel (entry, first K-tile, end of the K walk, after the
barrier, after accumulators -> LDS, after the second barrier, exit) + the CU id, and `rn_debug_set_stamps(ptr)` to switch them on.  Build it in
place of conv.o (same flags as csrc/Makefile), link a scratch library, run tools/probes/stamps_probe.py against it.  Not part of the product."""
import os
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
s = open(os.path.join(root, "pytorch_retinanet_amd/csrc/conv.hip")).read()
old = "    const uint16_t *X2s[CONV_MAX_PROBLEMS];\n    int x_split, x_ld;\n};"
assert s.count(old) == 1
s = s.replace(old, "    const uint16_t *X2s[CONV_MAX_PROBLEMS];\n    int x_split, x_ld;\n    unsigned long long *stamps;\n};")
i0 = s.index("__global__ __launch_bounds__(CONV_THREADS) void conv3x3_canvas_kernel(const ConvArgs args)")
i1 = s.index("\n}\n", i0) + 3
k = s[i0:i1]


def rep(a, b):
    global k
    assert k.count(a) == 1, a
    k = k.replace(a, b)


rep("    constexpr int MI = NARROW ? 1 : 4;                            // 32-row accumulator tiles per wave\n",
    "    constexpr int MI = NARROW ? 1 : 4;                            // 32-row accumulator tiles per wave\n"
    "    const unsigned long long st0 = __builtin_amdgcn_s_memrealtime();\n"
    "    unsigned long long st1 = 0, st2 = 0, st3 = 0, st2b = 0, st2c = 0;\n"
    "    const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);\n")
rep("    int scur = 0;                                                   // A stage of K-tile kt (mod 3)\n",
    "    st1 = __builtin_amdgcn_s_memrealtime();\n    int scur = 0;                                                   // A stage of K-tile kt (mod 3)\n")
rep("    if (grp == 0) __builtin_amdgcn_s_barrier();                    // group 0 catches up with group 1's extra barrier\n",
    "    if (grp == 0) __builtin_amdgcn_s_barrier();                    // group 0 catches up with group 1's extra barrier\n    st2 = __builtin_amdgcn_s_memrealtime();\n")
rep("    __syncthreads();\n    uint16_t *Ys = (uint16_t *)lds;", "    __syncthreads();\n    st2b = __builtin_amdgcn_s_memrealtime();\n    uint16_t *Ys = (uint16_t *)lds;")
rep("    if (relu_mask) {\n        *(uint4 *)(s_rmask + tid * 16) = rm_pre;\n", "    st2c = __builtin_amdgcn_s_memrealtime();\n    if (relu_mask) {\n        *(uint4 *)(s_rmask + tid * 16) = rm_pre;\n")
rep("            s_lut[tid] = make_uint4(w[0], w[1], w[2], w[3]);\n        }\n    }\n    __syncthreads();\n",
    "            s_lut[tid] = make_uint4(w[0], w[1], w[2], w[3]);\n        }\n    }\n    __syncthreads();\n    st3 = __builtin_amdgcn_s_memrealtime();\n")
k = k[:-2] + """    if (args.stamps && threadIdx.x == 0) {
        const unsigned long long st4 = __builtin_amdgcn_s_memrealtime();
        unsigned hw = 0, xcc = 0;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        unsigned long long *o = args.stamps + (size_t)lin * 12;
        o[0] = st0; o[1] = st1; o[2] = st2; o[3] = st3; o[4] = st4; o[5] = hw; o[6] = xcc; o[7] = lin; o[8] = st2b; o[9] = st2c; o[10] = st2;
    }
}
"""
s = s[:i0] + k + s[i1:]
old = "template <int DT, int MODE, bool NARROW>\nstatic int conv_launch_dt(const ConvArgs &a, const dim3 grid, hipStream_t st)\n{"
assert s.count(old) == 1
s = s.replace(old, "static unsigned long long *g_stamps = nullptr;\nRN_API void rn_debug_set_stamps(void *p) { g_stamps = (unsigned long long *)p; }\n"
              "template <int DT, int MODE, bool NARROW>\nstatic int conv_launch_dt(const ConvArgs &a_in, const dim3 grid, hipStream_t st)\n{\n    ConvArgs a = a_in; a.stamps = g_stamps;")
os.makedirs(os.path.join(root, "scratch"), exist_ok=True)
open(os.path.join(root, "scratch/conv_stamps.hip"), "w").write(s)
print("wrote scratch/conv_stamps.hip")
