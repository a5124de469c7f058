// Probe (gfx950): do 16-byte LDS-DMA loads (global_load_lds_dwordx4), 16-byte register loads and 16-byte stores work
// from / to global addresses that are only 4-byte aligned?  (The dense [.., 810] bf16 rows of the cls logits start on
// 1620-byte multiples: 4-byte aligned.)  Prints OK / MISMATCH per access kind and misalignment, plus a rough rate.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <cstring>
typedef __attribute__((address_space(3))) void *lds_void_ptr;

__global__ void k_lds_dma(const unsigned char *src, uint32_t *out, int mis_bytes, int rowstride)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[256 * 16];
    const int tid = threadIdx.x;
    // lane q reads 16 bytes at src + (q>>3)*rowstride + (q&7)*16 + mis
    const unsigned char *g = src + (size_t)blockIdx.x * 32 * rowstride + (tid >> 3) * rowstride + (tid & 7) * 16 + mis_bytes;
    __builtin_amdgcn_global_load_lds((const void *)g, (lds_void_ptr)(lds + tid * 16), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const uint4 v = *(const uint4 *)(lds + tid * 16);
    ((uint4 *)out)[blockIdx.x * 256 + tid] = v;
}
__global__ void k_reg(const unsigned char *src, uint32_t *out, int mis_bytes, int rowstride)
{
    const int tid = threadIdx.x;
    const unsigned char *g = src + (size_t)blockIdx.x * 32 * rowstride + (tid >> 3) * rowstride + (tid & 7) * 16 + mis_bytes;
    typedef __attribute__((ext_vector_type(4))) unsigned int u4;
    u4 v;
    asm volatile("global_load_dwordx4 %0, %1, off\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(g) : "memory");
    ((u4 *)out)[blockIdx.x * 256 + tid] = v;
}
__global__ void k_store(unsigned char *dst, int mis_bytes, int rowstride)
{
    const int tid = threadIdx.x;
    unsigned char *g = dst + (size_t)blockIdx.x * 32 * rowstride + (tid >> 3) * rowstride + (tid & 7) * 16 + mis_bytes;
    typedef __attribute__((ext_vector_type(4))) unsigned int u4;
    u4 v = {blockIdx.x * 1024 + tid * 4, blockIdx.x * 1024 + tid * 4 + 1, blockIdx.x * 1024 + tid * 4 + 2, blockIdx.x * 1024 + tid * 4 + 3};
    asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(g), "v"(v) : "memory");
}
int main()
{
    const int blocks = 4096, rowstride = 1620;
    const size_t bytes = (size_t)blocks * 32 * 1664 + 4096;
    std::vector<unsigned char> h(bytes);
    for (size_t i = 0; i < bytes; ++i) h[i] = (unsigned char)((i * 2654435761u) >> 13);
    unsigned char *d; uint32_t *o;
    hipMalloc(&d, bytes); hipMalloc(&o, (size_t)blocks * 256 * 16);
    hipMemcpy(d, h.data(), bytes, hipMemcpyHostToDevice);
    std::vector<uint32_t> ho((size_t)blocks * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int kind = 0; kind < 2; ++kind)
        for (int mis : {0, 4, 8, 12, 2}) {
            for (int rs : {rowstride, 1664}) {
                hipMemset(o, 0, (size_t)blocks * 256 * 16);
                hipEventRecord(e0);
                for (int rep = 0; rep < 20; ++rep) {
                    if (kind == 0) hipLaunchKernelGGL(k_lds_dma, dim3(blocks), dim3(256), 0, 0, d, o, mis, rs);
                    else hipLaunchKernelGGL(k_reg, dim3(blocks), dim3(256), 0, 0, d, o, mis, rs);
                }
                hipEventRecord(e1);
                hipError_t err = hipDeviceSynchronize();
                float ms = 0; hipEventElapsedTime(&ms, e0, e1);
                hipMemcpy(ho.data(), o, ho.size() * 4, hipMemcpyDeviceToHost);
                size_t bad = 0;
                for (int b = 0; b < blocks; ++b)
                    for (int t = 0; t < 256; ++t) {
                        const unsigned char *ref = h.data() + (size_t)b * 32 * rs + (t >> 3) * rs + (t & 7) * 16 + mis;
                        if (memcmp(ref, &ho[((size_t)b * 256 + t) * 4], 16)) ++bad;
                    }
                printf("%s mis=%2d rowstride=%d: %s (%zu bad of %d) err=%d  %.1f us/launch\n", kind == 0 ? "lds_dma " : "reg_load", mis, rs,
                       bad ? "MISMATCH" : "OK", bad, blocks * 256, (int)err, ms * 1000 / 20);
                fflush(stdout);
            }
        }
    for (int mis : {0, 4, 8, 12}) {
        hipMemset(d, 0xee, bytes);
        hipLaunchKernelGGL(k_store, dim3(blocks), dim3(256), 0, 0, d, mis, rowstride);
        hipError_t err = hipDeviceSynchronize();
        hipMemcpy(h.data(), d, bytes, hipMemcpyDeviceToHost);
        size_t bad = 0;
        for (int b = 0; b < blocks; ++b)
            for (int t = 0; t < 256; ++t) {
                const uint32_t *p = (const uint32_t *)(h.data() + (size_t)b * 32 * rowstride + (t >> 3) * rowstride + (t & 7) * 16 + mis);
                uint32_t w[4]; memcpy(w, p, 16);
                for (int j = 0; j < 4; ++j) if (w[j] != (uint32_t)(b * 1024 + t * 4 + j)) { ++bad; break; }
            }
        printf("store    mis=%2d: %s (%zu bad) err=%d\n", mis, bad ? "MISMATCH" : "OK", bad, (int)err);
    }
    return 0;
}
