// Probe: VALU cost of K3's per-element arithmetic with NO memory traffic (registers only).
// Reports ns per 8-element vector per wave and the implied chip-wide element rate.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int MATH>
__device__ __forceinline__ u32x4 math_vec(const u32x4 raw, float& acc)
{
    const unsigned w[4] = {raw.x, raw.y, raw.z, raw.w};
    unsigned o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float g2[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float x = __uint_as_float(h ? (w[i] & 0xffff0000u) : (w[i] << 16));
            const float z = __builtin_amdgcn_fmed3f(x + 1.0f, -80.0f, __builtin_inff());
            const float t = z * -1.4426950408889634f;
            const float den = 1.0f + (MATH >= 1 ? __builtin_amdgcn_exp2f(t) : t * t);
            const float ps = MATH >= 2 ? __builtin_amdgcn_rcpf(den) : den * 0.37f;
            const float wgt = ps * ps;
            const float bce = fmaf(MATH >= 3 ? __builtin_amdgcn_logf(den) : den * 1.3f, 0.6931471805599453f, z);
            acc = fmaf(wgt, bce, acc);
            g2[h] = (wgt * ps) * 0.00013f;
        }
        typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
        bf2 r; r.x = (__bf16)g2[0]; r.y = (__bf16)g2[1];
        o[i] = __builtin_bit_cast(unsigned, r);
    }
    u32x4 out; out.x = o[0]; out.y = o[1]; out.z = o[2]; out.w = o[3];
    return out;
}

template <int MATH>
__global__ __launch_bounds__(256) void valu(const u32x4* __restrict__ src, u32x4* __restrict__ dst, int iters)
{
    u32x4 q = src[threadIdx.x];
    float acc = 0.f;
    for (int i = 0; i < iters; ++i) {
        u32x4 r = math_vec<MATH>(q, acc);
        q.x ^= r.x & 0x00010000u; q.y ^= r.y & 0x00010000u; q.z ^= r.z & 0x00010000u; q.w ^= r.w & 0x00010000u;   // keep a dependency, cheap
    }
    q.x += __float_as_uint(acc);
    dst[(size_t)blockIdx.x * 256 + threadIdx.x] = q;
}

template <int MATH>
void run(const char* name, u32x4* a, u32x4* b, int blocks, int iters)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((valu<MATH>), dim3(blocks), dim3(256), 0, 0, a, b, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL((valu<MATH>), dim3(blocks), dim3(256), 0, 0, a, b, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double waves_per_simd = blocks * 4.0 / 1024.0;
    const double ns_per_vec_per_simd = ms * 1e6 / (iters * waves_per_simd);
    const double elems = (double)blocks * 256 * 8 * iters;
    printf("%-26s blocks=%5d  %8.1f us  %6.1f ns per wave-vector per SIMD (= %5.1f cycles @2.1GHz per element-step)  %6.2f Telem/s -> 145.2M elem in %6.1f us\n",
           name, blocks, ms * 1e3, ns_per_vec_per_simd, ns_per_vec_per_simd * 2.1 / 8, elems / (ms * 1e-3) / 1e12, 145.152e6 / (elems / (ms * 1e-3)) * 1e6);
}

int main()
{
    u32x4 *a, *b; hipMalloc(&a, 4096 * 16); hipMalloc(&b, (size_t)4096 * 256 * 16);
    hipMemset(a, 0x3c, 4096 * 16);
    for (int blocks : {256, 512, 2048}) {
        run<0>("math0 (no trans)", a, b, blocks, 2000);
        run<1>("math1 (exp)", a, b, blocks, 2000);
        run<2>("math2 (exp,rcp)", a, b, blocks, 2000);
        run<3>("math3 (exp,rcp,log)", a, b, blocks, 2000);
    }
    return 0;
}
