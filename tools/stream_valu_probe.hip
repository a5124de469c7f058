// Probe: how does K3's per-element arithmetic (12 VALU + exp/rcp/log on every bf16 element) overlap with a read+write stream
// of the same footprint, and do issue-order variants help?  Per-wave contiguous ranges, nt loads, 2 groups in flight (K3's
// skeleton).  VARIANT 0: loads of the next group first, then compute + store (K3 today).  1: same with s_setprio(3) around
// the memory instructions.  2: next group's loads issued between the two halves of the compute.  3: 4 groups per
// iteration (double the bytes per wave in flight).  WORK 0: copy only; 1: K3's background-element math.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void elem(const float x, const float gmul, float &acc, float &g)
{
    const float z = __builtin_amdgcn_fmed3f(x + 1.0f, -80.0f, __builtin_inff());
    const float den = 1.0f + __builtin_amdgcn_exp2f(z * -1.4426950408889634f);
    const float ps = __builtin_amdgcn_rcpf(den);
    const float w = ps * ps;
    const float bce = fmaf(__builtin_amdgcn_logf(den), 0.6931471805599453f, z);
    acc += w * bce;
    g = w * ps * gmul;
}
template <int WORK>
__device__ __forceinline__ u4 process(const u4 q, const float gmul, float &acc)
{
    if (WORK == 0) { u4 o = q; o.x += 1; return o; }
    const unsigned w[4] = {q.x, q.y, q.z, q.w};
    unsigned o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float g0, g1;
        elem(__uint_as_float(w[i] << 16), gmul, acc, g0);
        elem(__uint_as_float(w[i] & 0xffff0000u), gmul, acc, g1);
        typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
        bf2 r; r.x = (__bf16)g0; r.y = (__bf16)g1;
        o[i] = __builtin_bit_cast(unsigned, r);
    }
    u4 out; out.x = o[0]; out.y = o[1]; out.z = o[2]; out.w = o[3]; return out;
}

template <int WORK, int VARIANT>
__global__ __launch_bounds__(256) void stream_kernel(const u4 *__restrict__ src, u4 *__restrict__ dst, const int64_t nvec, const int64_t vec_per_wave,
                                                     float *sink, const float gmul)
{
    constexpr int PF = VARIANT == 3 ? 4 : 2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t gwave = (int64_t)blockIdx.x * 4 + wave;
    const int64_t beg = gwave * vec_per_wave, end = min(beg + vec_per_wave, nvec), last = end - 1;
    float acc = 0.0f;
    double dacc = 0.0;
    float gm = gmul;
    if (VARIANT == 5) {                  // K3's image_gmul in front of the stream: dependent vector loads + wait
        const int t0 = ((const volatile int *)sink)[1 + (lane & 1)], nf = ((const volatile int *)sink)[3];
        gm = gmul * (float)(t0 + 1) / (float)(nf > 1 ? nf : 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (beg < end) {
        u4 q[PF], qn[PF];
#pragma unroll
        for (int u = 0; u < PF; ++u) q[u] = __builtin_nontemporal_load(&src[min(beg + u * 64 + lane, last)]);
        for (int64_t v0 = beg; v0 < end; v0 += PF * 64) {
            if (VARIANT == 1) __builtin_amdgcn_s_setprio(3);
            if (VARIANT != 2) {
#pragma unroll
                for (int u = 0; u < PF; ++u) qn[u] = __builtin_nontemporal_load(&src[min(v0 + (PF + u) * 64 + lane, last)]);
            }
            if (VARIANT == 1) __builtin_amdgcn_s_setprio(0);
            float acc_g = 0.0f;
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                const u4 o = process<WORK>(q[u], gm, (VARIANT >= 4) ? acc_g : acc);
                if (VARIANT == 2 && u == 0) {
#pragma unroll
                    for (int k = 0; k < PF; ++k) qn[k] = __builtin_nontemporal_load(&src[min(v0 + (PF + k) * 64 + lane, last)]);
                }
                if (VARIANT == 1) __builtin_amdgcn_s_setprio(3);
                if (v0 + u * 64 + lane < end) dst[v0 + u * 64 + lane] = o;
                if (VARIANT == 1) __builtin_amdgcn_s_setprio(0);
            }
            if (VARIANT >= 4) dacc += (double)acc_g * (double)gm;
#pragma unroll
            for (int u = 0; u < PF; ++u) q[u] = qn[u];
        }
    }
    if (acc == 12345.678f || dacc == 12345.678) sink[0] = acc;
}

template <int WORK, int VARIANT> float run(const u4 *s, u4 *d, int64_t nvec, int blocks, float *sink)
{
    const int64_t waves = (int64_t)blocks * 4;
    int64_t vpw = (nvec + waves - 1) / waves; vpw = (vpw + 255) / 256 * 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((stream_kernel<WORK, VARIANT>), dim3(blocks), dim3(256), 0, 0, s, d, nvec, vpw, sink, 0.01f);
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((stream_kernel<WORK, VARIANT>), dim3(blocks), dim3(256), 0, 0, s, d, nvec, vpw, sink, 0.01f);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 20 * 1e3f;
}
int main()
{
    const int64_t bytes = 8ll * 201600 * 90 * 2, nvec = bytes / 16;
    u4 *s, *d; float *sink; hipMalloc(&s, bytes); hipMalloc(&d, bytes); hipMalloc(&sink, 64); hipMemset(sink, 0, 64);
    hipMemset(s, 0xc0, bytes);          // bf16 0xc0c0 = -6.02: a typical background logit
    if (getenv("PROBE_RANDOM")) {        // random logits ~ U(-8.5, -0.5): do data patterns matter (DRAM toggling / power)?
        std::vector<uint16_t> h(bytes / 2);
        uint32_t st = 12345;
        for (auto &v : h) { st = st * 1664525u + 1013904223u; const float f = -8.5f + 8.0f * (st >> 8) / 16777216.0f; uint32_t u; memcpy(&u, &f, 4); v = (uint16_t)(u >> 16); }
        hipMemcpy(s, h.data(), bytes, hipMemcpyHostToDevice);
    }
    for (int blocks : {1792, 2048}) {
        printf("blocks %d  copy: v0 %.1f  v1 %.1f  v2 %.1f  v3 %.1f us   |   with K3's math: v0 %.1f  v1 %.1f  v2 %.1f  v3 %.1f us\n", blocks,
               run<0, 0>(s, d, nvec, blocks, sink), run<0, 1>(s, d, nvec, blocks, sink), run<0, 2>(s, d, nvec, blocks, sink), run<0, 3>(s, d, nvec, blocks, sink),
               run<1, 0>(s, d, nvec, blocks, sink), run<1, 1>(s, d, nvec, blocks, sink), run<1, 2>(s, d, nvec, blocks, sink), run<1, 3>(s, d, nvec, blocks, sink));
        printf("blocks %d  with K3's math + double accumulation per group: %.1f us;  + dependent loads in front of the stream: %.1f us\n", blocks,
               run<1, 4>(s, d, nvec, blocks, sink), run<1, 5>(s, d, nvec, blocks, sink));
    }
    return 0;
}
