// Version / status strings of libretinanet_hip.so.
#include "rn_common.hpp"

RN_API int rn_version(void) { return RN_ABI_VERSION; }

RN_API const char *rn_status_string(int status)
{
    switch (status) {
        case RN_OK: return "ok";
        case RN_EINVAL: return "invalid argument (null pointer, non-positive size or bad enum)";
        case RN_EALIGN: return "pointer alignment requirement not met";
        case RN_EWORKSPACE: return "workspace too small";
        case RN_EUNSUPPORTED: return "shape outside the supported range";
        case RN_ETHRESH: return "match threshold must be greater than background threshold";
        default: break;
    }
    if (status > 0) return hipGetErrorString((hipError_t)status);
    return "unknown status";
}

// Node census of a captured hipGraph (hipGraph_t as an opaque pointer): counts[0] = kernel, [1] = memset, [2] = memcpy, [3] = every
// other node type.  graph.CapturedTrainStep refuses to replay a step whose graph holds a MEMSET node: on ROCm 7.0 such nodes write
// garbage once the process has synchronised with the device and enqueued other blit work (round 4; include/retinanet_hip.h).
RN_API int rn_hipgraph_node_census(void *graph, int64_t counts[4])
{
    if (!graph || !counts) return RN_EINVAL;
    counts[0] = counts[1] = counts[2] = counts[3] = 0;
    size_t n = 0;
    RN_HIP(hipGraphGetNodes((hipGraph_t)graph, nullptr, &n));
    if (n == 0) return RN_OK;
    hipGraphNode_t *nodes = new hipGraphNode_t[n];
    hipError_t e = hipGraphGetNodes((hipGraph_t)graph, nodes, &n);
    for (size_t i = 0; e == hipSuccess && i < n; ++i) {
        hipGraphNodeType t;
        e = hipGraphNodeGetType(nodes[i], &t);
        if (e != hipSuccess) break;
        if (t == hipGraphNodeTypeKernel) ++counts[0];
        else if (t == hipGraphNodeTypeMemset) ++counts[1];
        else if (t == hipGraphNodeTypeMemcpy) ++counts[2];
        else ++counts[3];
    }
    delete[] nodes;
    return e == hipSuccess ? RN_OK : (int)e;
}

// ---- memset nodes -> kernel nodes -------------------------------------------------------------------------------------------------
// Replays of a hipGraph that holds MEMSET nodes go wrong on ROCm 7.0 after the process has synchronised with the device and enqueued
// other blit work (see rn_hipgraph_node_census).  A captured step cannot always avoid them (MIOpen clears the output of some
// split-K weight-gradient algorithms with hipMemsetAsync), so the graph is repaired BEFORE it is instantiated: every memset node is
// replaced by a kernel node -- graph_fill_kernel below, same destination, value, element size, width, height and pitch -- that
// takes over the node's dependencies and dependents.
namespace {

struct GraphFill { unsigned char *dst; unsigned long long pitch, width, height; unsigned value, esize; };

__global__ __launch_bounds__(256) void graph_fill_kernel(const GraphFill f)
{
    const unsigned long long total = f.width * f.height;
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (unsigned long long)gridDim.x * 256) {
        const unsigned long long row = i / f.width, col = i - row * f.width;
        unsigned char *p = f.dst + row * f.pitch + col * f.esize;
        if (f.esize == 4) *(unsigned *)p = f.value;
        else if (f.esize == 2) *(unsigned short *)p = (unsigned short)f.value;
        else *p = (unsigned char)f.value;
    }
}

}  // namespace

RN_API int rn_hipgraph_replace_memset_nodes(void *graph_, int64_t *replaced)
{
    if (!graph_) return RN_EINVAL;
    hipGraph_t graph = (hipGraph_t)graph_;
    if (replaced) *replaced = 0;
    size_t n = 0;
    RN_HIP(hipGraphGetNodes(graph, nullptr, &n));
    if (n == 0) return RN_OK;
    hipGraphNode_t *nodes = new hipGraphNode_t[n];
    hipError_t e = hipGraphGetNodes(graph, nodes, &n);
    int64_t done = 0;
    for (size_t i = 0; e == hipSuccess && i < n; ++i) {
        hipGraphNodeType t;
        e = hipGraphNodeGetType(nodes[i], &t);
        if (e != hipSuccess || t != hipGraphNodeTypeMemset) continue;
        hipMemsetParams mp;
        if ((e = hipGraphMemsetNodeGetParams(nodes[i], &mp)) != hipSuccess) break;
        size_t nd = 0, nt = 0;
        if ((e = hipGraphNodeGetDependencies(nodes[i], nullptr, &nd)) != hipSuccess) break;
        if ((e = hipGraphNodeGetDependentNodes(nodes[i], nullptr, &nt)) != hipSuccess) break;
        hipGraphNode_t *deps = new hipGraphNode_t[nd + 1], *outs = new hipGraphNode_t[nt + 1];
        if (nd) e = hipGraphNodeGetDependencies(nodes[i], deps, &nd);
        if (e == hipSuccess && nt) e = hipGraphNodeGetDependentNodes(nodes[i], outs, &nt);
        if (e == hipSuccess) {
            GraphFill f;
            f.dst = (unsigned char *)mp.dst; f.esize = mp.elementSize ? mp.elementSize : 1; f.width = mp.width;
            f.height = mp.height ? mp.height : 1; f.pitch = f.height > 1 ? mp.pitch : 0; f.value = mp.value;
            // (wide dword fill for the common case: a 1-byte memset of zeros over a multiple of 4 bytes at an aligned address)
            if (f.esize == 1 && f.height == 1 && (f.width & 3) == 0 && (((uintptr_t)f.dst) & 3) == 0) {
                const unsigned b = f.value & 0xffu;
                f.value = b | (b << 8) | (b << 16) | (b << 24); f.esize = 4; f.width >>= 2;
            }
            const unsigned long long total = f.width * f.height;
            unsigned long long blocks = (total + 255) / 256;
            if (blocks > 2048) blocks = 2048;
            if (blocks < 1) blocks = 1;
            void *args[1] = {(void *)&f};
            hipKernelNodeParams kp = {};
            kp.func = (void *)graph_fill_kernel;
            kp.gridDim = dim3((unsigned)blocks); kp.blockDim = dim3(256); kp.sharedMemBytes = 0; kp.kernelParams = args; kp.extra = nullptr;
            hipGraphNode_t fresh;
            e = hipGraphAddKernelNode(&fresh, graph, nd ? deps : nullptr, nd, &kp);
            for (size_t k = 0; e == hipSuccess && k < nt; ++k) e = hipGraphAddDependencies(graph, &fresh, &outs[k], 1);
            if (e == hipSuccess) e = hipGraphDestroyNode(nodes[i]);
            if (e == hipSuccess) ++done;
        }
        delete[] deps; delete[] outs;
    }
    delete[] nodes;
    if (replaced) *replaced = done;
    return e == hipSuccess ? RN_OK : (int)e;
}
