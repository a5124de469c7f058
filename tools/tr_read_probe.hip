// Verify the lane mapping of ds_read_b64_tr_b16 on gfx950 before building a kernel on it:
// per 16-lane group, lane 4q+p supplies the address of row q, columns 4p..4p+3 of a 4x16 block of 16-bit elements;
// lane i receives column i of the 4 rows (row q in element q).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((ext_vector_type(4))) unsigned short u16x4;
__global__ void k(uint16_t *out)
{
    __shared__ __attribute__((aligned(16))) uint16_t tile[16][32];       // value = row * 100 + col
    for (int i = threadIdx.x; i < 16 * 32; i += 64) tile[i / 32][i % 32] = (uint16_t)((i / 32) * 100 + (i % 32));
    __syncthreads();
    const int lane = threadIdx.x, grp = lane >> 4, l = lane & 15, q = l >> 2, p = l & 3;
    // group g reads the block rows 4g..4g+3, columns 0..15
    const uint32_t addr = (uint32_t)(uintptr_t)&tile[4 * grp + q][4 * p];
    u16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr));
    for (int e = 0; e < 4; ++e) out[lane * 4 + e] = v[e];
}
int main()
{
    uint16_t *d; hipMalloc(&d, 64 * 4 * 2);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    uint16_t h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int lane = 0; lane < 64; ++lane) {
        const int grp = lane >> 4, i = lane & 15;
        for (int e = 0; e < 4; ++e) { const int want = (4 * grp + e) * 100 + i; if (h[lane * 4 + e] != want) { if (bad < 8) printf("lane %d e %d got %d want %d\n", lane, e, h[lane * 4 + e], want); ++bad; } }
    }
    printf("tr read mapping: %d mismatches\n", bad);
    for (int lane = 0; lane < 20; ++lane) printf("lane %2d: %4d %4d %4d %4d\n", lane, h[lane * 4], h[lane * 4 + 1], h[lane * 4 + 2], h[lane * 4 + 3]);
    return 0;
}
