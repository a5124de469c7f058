// Probe: does the wave -> address mapping of a streaming read+write kernel matter on MI355X?
//   mode 0: every WAVE owns one contiguous range (K3's layout: 8192 independent sequential streams)
//   mode 1: every WORKGROUP owns one contiguous range, its 4 waves interleave 1 KiB chunks inside it
//   mode 2: grid-stride (all waves of the chip interleave 1 KiB chunks)
// 310 MB in + 310 MB out of 16-byte vectors, nt loads, two loads in flight per lane, resident-sized grid.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void copy_kernel(const u4 *__restrict__ src, u4 *__restrict__ dst, const int64_t nvec, const int64_t vec_per_wave)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t gwave = (int64_t)blockIdx.x * 4 + wave, nwaves = (int64_t)gridDim.x * 4;
    if (MODE == 0) {
        const int64_t beg = gwave * vec_per_wave, end = min(beg + vec_per_wave, nvec);
        for (int64_t v = beg + lane; v < end; v += 128) {
            const u4 a = __builtin_nontemporal_load(&src[v]);
            const u4 b = (v + 64 < end) ? __builtin_nontemporal_load(&src[v + 64]) : a;
            u4 o = a; o.x += 1; dst[v] = o;
            if (v + 64 < end) { u4 p = b; p.x += 1; dst[v + 64] = p; }
        }
    } else if (MODE == 1) {
        const int64_t beg = (int64_t)blockIdx.x * 4 * vec_per_wave, end = min(beg + 4 * vec_per_wave, nvec);
        for (int64_t v = beg + wave * 128 + lane; v < end; v += 4 * 128) {
            const u4 a = __builtin_nontemporal_load(&src[v]);
            const u4 b = (v + 64 < end) ? __builtin_nontemporal_load(&src[v + 64]) : a;
            u4 o = a; o.x += 1; dst[v] = o;
            if (v + 64 < end) { u4 p = b; p.x += 1; dst[v + 64] = p; }
        }
    } else {
        for (int64_t v = gwave * 128 + lane; v < nvec; v += nwaves * 128) {
            const u4 a = __builtin_nontemporal_load(&src[v]);
            const u4 b = (v + 64 < nvec) ? __builtin_nontemporal_load(&src[v + 64]) : a;
            u4 o = a; o.x += 1; dst[v] = o;
            if (v + 64 < nvec) { u4 p = b; p.x += 1; dst[v + 64] = p; }
        }
    }
}

template <int MODE> float run(const u4 *s, u4 *d, int64_t nvec, int blocks, int reps)
{
    const int64_t waves = (int64_t)blocks * 4;
    int64_t vpw = (nvec + waves - 1) / waves; vpw = (vpw + 127) / 128 * 128;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(copy_kernel<MODE>, dim3(blocks), dim3(256), 0, 0, s, d, nvec, vpw);
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(copy_kernel<MODE>, dim3(blocks), dim3(256), 0, 0, s, d, nvec, vpw);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms / reps;
}
int main()
{
    const int64_t bytes = 8ll * 201600 * 90 * 2, nvec = bytes / 16;
    u4 *s, *d; hipMalloc(&s, bytes); hipMalloc(&d, bytes); hipMemset(s, 1, bytes);
    for (int blocks : {1536, 1792, 2048, 4096}) {
        const float t0 = run<0>(s, d, nvec, blocks, 20), t1 = run<1>(s, d, nvec, blocks, 20), t2 = run<2>(s, d, nvec, blocks, 20);
        printf("blocks %4d: per-wave ranges %.1f us (%.2f TB/s) | per-workgroup ranges, waves interleaved %.1f us (%.2f TB/s) | grid-stride %.1f us (%.2f TB/s)\n",
               blocks, t0 * 1e3, 2 * bytes / t0 / 1e9, t1 * 1e3, 2 * bytes / t1 / 1e9, t2 * 1e3, 2 * bytes / t2 / 1e9);
    }
    return 0;
}
