"""Writes scratch/pw_stamps.hip: csrc/pw.hip with s_memrealtime stamps in pw_gemm_kernel -- per row tile of a walker: tile start, first
commit + barrier done, end of the K loop, accumulators in LDS (after the barrier), epilogue rows stored, tile end, and the K loop's time
split into its four phases (request / fragment reads + MFMAs / commit / barrier, summed over the K-tiles) -- and
`rn_debug_pw_stamps(ptr, cap)` to switch them on (thread 0 of every workgroup, up to 4 tiles each, 12 words per tile).  Build it in place of pw.o
(tools/probes/make_stamped_pw.sh), run tools/probes/pw_stamps_probe.py against the scratch library.  Not part of the product."""
import os
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
s = open(os.path.join(root, "pytorch_retinanet_amd/csrc/pw.hip")).read()


def rep(a, b, cnt=1):
    global s
    assert s.count(a) == cnt, (s.count(a), a)
    s = s.replace(a, b)


rep("namespace {\n\nusing rn::f32x16;\n",
    "__device__ unsigned long long *g_pw_stamps = nullptr;\n__device__ int g_pw_stamp_cap = 0;\n"
    "RN_API void rn_debug_pw_stamps(void *p, int cap) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_pw_stamps), &p, sizeof(p)); (void)hipMemcpyToSymbol(HIP_SYMBOL(g_pw_stamp_cap), &cap, sizeof(cap)); }\n"
    "namespace {\n\nusing rn::f32x16;\n")
rep("    for (; mt < MT; mt += a.gx) {\n        const int m0 = mt * PW_BM;\n",
    "    int tile_no = 0;\n    const unsigned lin = blockIdx.x + gridDim.x * blockIdx.y;\n"
    "    for (; mt < MT; mt += a.gx) {\n        const int m0 = mt * PW_BM;\n        const unsigned long long s0 = __builtin_amdgcn_s_memrealtime();\n"
    "        unsigned long long s1 = 0, s2 = 0, s3 = 0, s4 = 0, d_issue = 0, d_mma = 0, d_commit = 0, d_bar = 0, tA, tB;\n")
# the K loop: per-phase sums
rep("        commit(0, S0);\n        __syncthreads();\n        for (int kt = 0; kt < KT; ++kt) {\n"
    "            if (kt + 1 < KT) issue(kt + 1, S0);                 // in flight under this K-tile's MFMAs\n"
    "            else issue_next_tile();\n            mma_stage(kt & 1);\n            if (kt + 1 < KT) commit((kt + 1) & 1, S0);\n            __syncthreads();\n        }\n",
    "        commit(0, S0);\n        __syncthreads();\n        s1 = __builtin_amdgcn_s_memrealtime();\n        for (int kt = 0; kt < KT; ++kt) {\n"
    "            tA = __builtin_amdgcn_s_memrealtime();\n"
    "            if (kt + 1 < KT) issue(kt + 1, S0);\n            else issue_next_tile();\n"
    "            __builtin_amdgcn_sched_barrier(0); tB = __builtin_amdgcn_s_memrealtime(); d_issue += tB - tA; tA = tB; __builtin_amdgcn_sched_barrier(0);\n"
    "            mma_stage(kt & 1);\n"
    "            __builtin_amdgcn_sched_barrier(0); asm volatile(\"s_nop 0\" ::: \"memory\"); tB = __builtin_amdgcn_s_memrealtime(); d_mma += tB - tA; tA = tB; __builtin_amdgcn_sched_barrier(0);\n"
    "            if (kt + 1 < KT) commit((kt + 1) & 1, S0);\n"
    "            __builtin_amdgcn_sched_barrier(0); asm volatile(\"s_waitcnt lgkmcnt(0)\" ::: \"memory\"); tB = __builtin_amdgcn_s_memrealtime(); d_commit += tB - tA; tA = tB; __builtin_amdgcn_sched_barrier(0);\n"
    "            __syncthreads();\n            tB = __builtin_amdgcn_s_memrealtime(); d_bar += tB - tA;\n        }\n")
rep("        // ---- epilogue: accumulators -> f32 tile in LDS -> rows of 8-channel vectors\n",
    "        s2 = __builtin_amdgcn_s_memrealtime();\n        // ---- epilogue: accumulators -> f32 tile in LDS -> rows of 8-channel vectors\n")
rep("        __syncthreads();\n#pragma unroll\n        for (int i = 0; i < EROWS; ++i) {\n            const int row = erl + i * RL;\n            const int m = m0 + row;\n            if (m < a.M) {\n                float v[8];\n                ld8f(tile + row * BN + ecg * 8, v);",
    "        __syncthreads();\n        s3 = __builtin_amdgcn_s_memrealtime();\n#pragma unroll\n        for (int i = 0; i < EROWS; ++i) {\n            const int row = erl + i * RL;\n            const int m = m0 + row;\n            if (m < a.M) {\n                float v[8];\n                ld8f(tile + row * BN + ecg * 8, v);")
rep("        __syncthreads();                                        // the tile is the next row tile's staging area\n    }\n",
    "        s4 = __builtin_amdgcn_s_memrealtime();\n        __syncthreads();                                        // the tile is the next row tile's staging area\n"
    "        if (g_pw_stamps && threadIdx.x == 0 && tile_no < 4 && (int)lin < g_pw_stamp_cap) {\n"
    "            unsigned long long *o = g_pw_stamps + ((size_t)lin * 4 + tile_no) * 12;\n"
    "            o[0] = s0; o[1] = s1; o[2] = s2; o[3] = s3; o[4] = s4; o[5] = __builtin_amdgcn_s_memrealtime(); o[6] = d_issue; o[7] = d_mma; o[8] = d_commit; o[9] = d_bar;\n"
    "        }\n        ++tile_no;\n    }\n")
os.makedirs(os.path.join(root, "scratch"), exist_ok=True)
open(os.path.join(root, "scratch/pw_stamps.hip"), "w").write(s)
print("wrote scratch/pw_stamps.hip")
