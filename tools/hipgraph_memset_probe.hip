// Stand-alone probe for the round-4 finding: a hipMemsetAsync captured into a hipGraph (a MEMSET NODE) and replayed after the process
// has synchronised with the device and done unrelated eager work (allocations, memsets, fill kernels).
//   hipcc --offload-arch=gfx950 -O2 -o tools/hipgraph_memset_probe.bin tools/hipgraph_memset_probe.hip && tools/hipgraph_memset_probe.bin
// Graph: memset(counter, 0, 32 B) -> add_kernel(counter += 1) -> snapshot_kernel(out = counter).  Every replay must leave out[i] == 1.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e__)); exit(2); } } while (0)

__global__ void add_kernel(int *p, int n) { int i = threadIdx.x; if (i < n) p[i] += 1; }
__global__ void snapshot_kernel(const int *p, int *out, int n) { int i = threadIdx.x; if (i < n) out[i] = p[i]; }
__global__ void fill_kernel(unsigned *p, size_t n, unsigned v) { for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) p[i] = v; }

static int check(const char *when, int *out_d, int n)
{
    std::vector<int> h(n);
    CK(hipMemcpy(h.data(), out_d, n * sizeof(int), hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < n; ++i) bad += h[i] != 1;
    printf("%-48s out = [%d %d %d %d ...]  %s\n", when, h[0], h[1], h[2], h[3], bad ? "WRONG" : "ok");
    return bad;
}

int main(int argc, char **argv)
{
    const int n = 8, rounds = argc > 1 ? atoi(argv[1]) : 4;
    int *counter, *out;
    CK(hipMalloc(&counter, n * sizeof(int)));
    CK(hipMalloc(&out, n * sizeof(int)));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    CK(hipMemsetAsync(counter, 0, n * sizeof(int), st));
    hipLaunchKernelGGL(add_kernel, dim3(1), dim3(64), 0, st, counter, n);
    hipLaunchKernelGGL(snapshot_kernel, dim3(1), dim3(64), 0, st, counter, out, n);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    int bad = 0;
    for (int r = 0; r < 3; ++r) CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    bad += check("3 replays, no interruption:", out, n);
    for (int round = 0; round < rounds; ++round) {
        // unrelated eager work on the null stream and on the capture stream: allocations, memsets with other values, fill kernels
        CK(hipDeviceSynchronize());
        std::vector<void *> junk;
        for (size_t bytes : {64ul, 256ul, 4096ul, 1ul << 16, 1ul << 20, 1ul << 24}) {
            for (int k = 0; k < 8; ++k) {
                void *p; CK(hipMalloc(&p, bytes));
                CK(hipMemsetAsync(p, 0xAB, bytes, k & 1 ? st : nullptr));
                hipLaunchKernelGGL(fill_kernel, dim3(64), dim3(256), 0, k & 1 ? nullptr : st, (unsigned *)p, bytes / 4, 0x7fc00000u);
                junk.push_back(p);
            }
        }
        CK(hipDeviceSynchronize());
        for (void *p : junk) CK(hipFree(p));
        CK(hipGraphLaunch(ge, st));
        CK(hipStreamSynchronize(st));
        char msg[96];
        snprintf(msg, sizeof msg, "replay after sync + eager work (round %d):", round);
        bad += check(msg, out, n);
        for (int r = 0; r < 2; ++r) CK(hipGraphLaunch(ge, st));
        CK(hipStreamSynchronize(st));
        bad += check("  + 2 more replays:", out, n);
    }
    printf(bad ? "MEMSET NODE MISBEHAVED\n" : "memset node behaved in this minimal setting\n");
    return 0;
}
