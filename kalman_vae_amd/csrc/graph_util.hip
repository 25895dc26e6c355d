// Inspection of a captured HIP graph.  On ROCm 7.0 a memset node followed by a
// kernel node that depends on it is not always ordered at replay: the kernel has
// been observed to start before the fill kernel that implements the memset has
// finished (rocprofv3 kernel trace: start 117 ns before the fill's end), which
// breaks library reductions that zero a semaphore with hipMemsetAsync and then
// count finished workgroups on it.  Graphs built by this package therefore must
// not contain memset nodes; this helper lets the host code check that.
#include <hip/hip_runtime.h>

#include <vector>

#include "../../include/kvae_kalman.h"

namespace kvae {

static int count_nodes(hipGraph_t graph, hipGraphNodeType wanted, int depth) {
    size_t n = 0;
    if (hipGraphGetNodes(graph, nullptr, &n) != hipSuccess) return -1;
    if (n == 0) return 0;
    std::vector<hipGraphNode_t> nodes(n);
    if (hipGraphGetNodes(graph, nodes.data(), &n) != hipSuccess) return -1;
    int count = 0;
    for (size_t i = 0; i < n; ++i) {
        hipGraphNodeType type;
        if (hipGraphNodeGetType(nodes[i], &type) != hipSuccess) return -1;
        if (type == wanted) ++count;
        if (type == hipGraphNodeTypeGraph && depth < 8) {
            hipGraph_t child;
            if (hipGraphChildGraphNodeGetGraph(nodes[i], &child) != hipSuccess) return -1;
            const int sub = count_nodes(child, wanted, depth + 1);
            if (sub < 0) return -1;
            count += sub;
        }
    }
    return count;
}

}  // namespace kvae

extern "C" {

int kvae_graph_count_memset_nodes(void* hip_graph) {
    if (!hip_graph) return KVAE_E_BADARG;
    return kvae::count_nodes((hipGraph_t)hip_graph, hipGraphNodeTypeMemset, 0);
}

int kvae_graph_count_kernel_nodes(void* hip_graph) {
    if (!hip_graph) return KVAE_E_BADARG;
    return kvae::count_nodes((hipGraph_t)hip_graph, hipGraphNodeTypeKernel, 0);
}

}  // extern "C"
