// Fused graph-level readout of chem_tensorflow_sparse.py:220-231 (gated_regression) with utils.MLP(hid_sizes=[])
// (utils.py:39-70: a single linear layer whose PRE-activation is returned):
//     gate[v]  = sigmoid( [h_T[v] | h_0[v]] . Wg + bg )          Wg [2D,1]
//     val[v]   = h_T[v] . Wt + bt                                 Wt [D,1]
//     out[g]   = sum over the nodes v of graph g of gate[v] * val[v]            (tf.unsorted_segment_sum, :226-228)
// The reference materialises the [V,2D] concat (:222), two [V,1] matmul results, their product and the segment
// sum; here one pass reads h_T and h_0 once (HBM bound: 8*D bytes per node) and atomically adds one float per
// node into its graph's slot.  One 16-lane group per node, float4 per lane, __shfl_xor reduction.
#include "ggnn_common.h"

namespace ggnn {

__global__ __launch_bounds__(256) void gated_readout_kernel(const float* __restrict__ hT, const float* __restrict__ h0,
                                                            const int* __restrict__ graph_of, const float* __restrict__ Wg,
                                                            const float* __restrict__ bgp, const float* __restrict__ Wt,
                                                            const float* __restrict__ btp,
                                                            float* __restrict__ out, int V, int D, int G) {
    constexpr int LPN = 16;                                  // lanes per node
    const int l = threadIdx.x % LPN;
    int v = blockIdx.x * (256 / LPN) + threadIdx.x / LPN;
    const bool live = v < V;
    v = live ? v : V - 1;
    const int D4 = D >> 2;
    float sg = 0.f, st = 0.f;
    for (int c4 = l; c4 < D4; c4 += LPN) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(hT + (size_t)v * D + 4 * c4);
        const f32x4 b = *reinterpret_cast<const f32x4*>(h0 + (size_t)v * D + 4 * c4);
        const f32x4 wa = *reinterpret_cast<const f32x4*>(Wg + 4 * c4);
        const f32x4 wb = *reinterpret_cast<const f32x4*>(Wg + D + 4 * c4);
        const f32x4 wt = *reinterpret_cast<const f32x4*>(Wt + 4 * c4);
        sg += a.x * wa.x + a.y * wa.y + a.z * wa.z + a.w * wa.w + b.x * wb.x + b.y * wb.y + b.z * wb.z + b.w * wb.w;
        st += a.x * wt.x + a.y * wt.y + a.z * wt.z + a.w * wt.w;
    }
#pragma unroll
    for (int off = LPN / 2; off > 0; off >>= 1) {
        sg += __shfl_xor(sg, off, LPN);
        st += __shfl_xor(st, off, LPN);
    }
    if (live && l == 0) {
        const int g = graph_of[v];
        if ((unsigned)g < (unsigned)G) {
            const float gate = 1.0f / (1.0f + expf(-(sg + bgp[0])));
            unsafeAtomicAdd(out + g, gate * (st + btp[0]));
        }
    }
}

}  // namespace ggnn

using namespace ggnn;

extern "C" int ggnn_gated_readout_f32(const float* hT, const float* h0, const int32_t* graph_nodes_list, const float* gate_W,
                                      const float* gate_b, const float* transform_W, const float* transform_b, float* out, int V, int D,
                                      int num_graphs, ggnn_stream_t stream) {
    GGNN_CHECK_ARG(V >= 0 && D > 0 && D % 4 == 0 && num_graphs >= 0, "bad sizes V=%d D=%d G=%d", V, D, num_graphs);
    hipStream_t st = (hipStream_t)stream;
    if (num_graphs == 0) return GGNN_OK;
    GGNN_CHECK_ARG(out, "null pointer");
    GGNN_CHECK_HIP(hipMemsetAsync(out, 0, sizeof(float) * (size_t)num_graphs, st));
    if (V == 0) return GGNN_OK;
    GGNN_CHECK_ARG(hT && h0 && graph_nodes_list && gate_W && transform_W && gate_b && transform_b, "null pointer");
    GGNN_CHECK_ARG(aligned16(hT) && aligned16(h0) && aligned16(gate_W) && aligned16(transform_W), "pointers must be 16-byte aligned");
    hipLaunchKernelGGL(gated_readout_kernel, dim3((V + 15) / 16), dim3(256), 0, st, hT, h0, graph_nodes_list, gate_W, gate_b,
                       transform_W, transform_b, out, V, D, num_graphs);
    GGNN_CHECK_HIP(hipGetLastError());
    return GGNN_OK;
}
