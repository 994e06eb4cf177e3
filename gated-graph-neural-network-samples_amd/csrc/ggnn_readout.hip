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


// ---- fused readout + masked loss, forward and backward (chem_tensorflow_sparse.py:220-231 + chem_tensorflow.py:158-170) -------
// Deterministic: the nodes of a graph are consecutive (graph_nodes_list is non-decreasing: the batchers append graph after
// graph, :297-304), so out[g] is a segmented sum in node order -- no atomics -- and every cross-block reduction below runs
// in a fixed order.  Training no longer materialises the [V,2D] concat (:222), the two [V,1] projections, their product
// or any of their gradients as separate tensors.
//
//   forward   node kernel : gate[v] = sigmoid([hT|h0][v] . Wg + bg),  val[v] = hT[v] . Wt + bt          (16 lanes per node)
//             graph kernel: out[g] = sum_{v in g} gate[v] val[v] (node_mask[v]);  diff = (out - y) m;
//                           per-block partials of  sum 0.5 diff^2, sum |diff|, sum m
//             stats kernel: the three sums over the blocks, in block order
//   backward  node kernel : dO[g] = d_out[g] + d_num m^2 (out-y) + d_abs m sign(diff);  dgated = dO[g(v)] node_mask[v];
//                           dval = dgated gate;  dpre = dgated val gate (1-gate);  d_hT[v] (+)= dpre Wg[:D] + dval Wt;
//                           per-block column partials of  sum dpre hT, sum dpre h0, sum dval hT, sum dpre, sum dval
//             final kernel: the block partials summed in block order -> d_gate_W [2D], d_gate_b, d_transform_W [D], d_transform_b
constexpr int kReadoutLanes = 16;               // lanes per node
constexpr int kReadoutMaxSlots = 4;             // float4 column slots per lane: D <= 16 * 4 * 4 = 256
constexpr int kReadoutBwdBlocks = 1024;        // (four 4-wave blocks per CU: a block walks its nodes through a chain of dependent loads)

__global__ __launch_bounds__(256) void readout_node_kernel(const float* __restrict__ hT, const float* __restrict__ h0,
                                                           const float* __restrict__ Wg, const float* __restrict__ bgp,
                                                           const float* __restrict__ Wt, const float* __restrict__ btp,
                                                           float* __restrict__ gate, float* __restrict__ val, int V, int D) {
    constexpr int LPN = kReadoutLanes;
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
        gate[v] = 1.0f / (1.0f + expf(-(sg + bgp[0])));
        val[v] = st + btp[0];
    }
}

// first node of graph g in the non-decreasing graph_nodes_list (graph_ptr == NULL: binary search)
__device__ __forceinline__ int graph_begin(const int* __restrict__ graph_of, const int* __restrict__ graph_ptr, int g, int V) {
    if (graph_ptr) return graph_ptr[g];
    int lo = 0, hi = V;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (graph_of[mid] < g) lo = mid + 1; else hi = mid;
    }
    return lo;
}

__global__ __launch_bounds__(256) void readout_graph_kernel(const float* __restrict__ gate, const float* __restrict__ val,
                                                            const int* __restrict__ graph_of, const int* __restrict__ graph_ptr,
                                                            const float* __restrict__ node_mask, const float* __restrict__ target,
                                                            const float* __restrict__ mask, float* __restrict__ out,
                                                            float* __restrict__ partials, int V, int G) {
    __shared__ float red[3][256];
    const int g = blockIdx.x * 256 + threadIdx.x;
    float num = 0.f, ab = 0.f, ms = 0.f;
    if (g < G) {
        const int beg = graph_begin(graph_of, graph_ptr, g, V), end = graph_begin(graph_of, graph_ptr, g + 1, V);
        float s = 0.f;
        for (int v = beg; v < end; ++v) s += node_mask ? gate[v] * val[v] * node_mask[v] : gate[v] * val[v];
        out[g] = s;
        if (target) {
            const float m = mask ? mask[g] : 1.0f;
            const float diff = (s - target[g]) * m;                 // chem_tensorflow.py:161,164
            num = 0.5f * diff * diff; ab = fabsf(diff); ms = m;
        }
    }
    if (!partials) return;
    red[0][threadIdx.x] = num; red[1][threadIdx.x] = ab; red[2][threadIdx.x] = ms;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {                       // fixed tree: deterministic
        if ((int)threadIdx.x < off) {
            red[0][threadIdx.x] += red[0][threadIdx.x + off];
            red[1][threadIdx.x] += red[1][threadIdx.x + off];
            red[2][threadIdx.x] += red[2][threadIdx.x + off];
        }
        __syncthreads();
    }
    if (threadIdx.x < 3) partials[blockIdx.x * 3 + threadIdx.x] = red[threadIdx.x][0];
}

__global__ void readout_stats_kernel(const float* __restrict__ partials, int nblocks, float* __restrict__ stats) {
    if (threadIdx.x < 3) {
        float s = 0.f;
        for (int b = 0; b < nblocks; ++b) s += partials[b * 3 + threadIdx.x];
        stats[threadIdx.x] = s;
    }
}

__global__ __launch_bounds__(256) void readout_bwd_node_kernel(
        const float* __restrict__ hT, const float* __restrict__ h0, const int* __restrict__ graph_of,
        const float* __restrict__ node_mask, const float* __restrict__ Wg, const float* __restrict__ Wt,
        const float* __restrict__ gate, const float* __restrict__ val, const float* __restrict__ out,
        const float* __restrict__ target, const float* __restrict__ mask, const float* __restrict__ d_out,
        const float* __restrict__ d_stats, float* __restrict__ d_hT, int accumulate, float* __restrict__ partials,
        int V, int D, int G) {
    constexpr int LPN = kReadoutLanes, NPB = 256 / LPN, S = kReadoutMaxSlots;
    extern __shared__ float red_s[];                                // [NPB][3*D + 4]: group stride a multiple of 4 floats (f32x4 stores)
    const int l = threadIdx.x % LPN, grp = threadIdx.x / LPN;
    const int D4 = D >> 2;
    const float d_num = d_stats ? d_stats[0] : 0.f, d_abs = d_stats ? d_stats[1] : 0.f;
    f32x4 pa[S], pb[S], pt[S], wa[S], wt[S];
#pragma unroll
    for (int s = 0; s < S; ++s) {
        pa[s] = pb[s] = pt[s] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int c4 = l + s * LPN;
        wa[s] = c4 < D4 ? *reinterpret_cast<const f32x4*>(Wg + 4 * c4) : f32x4{0.f, 0.f, 0.f, 0.f};
        wt[s] = c4 < D4 ? *reinterpret_cast<const f32x4*>(Wt + 4 * c4) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    float ppre = 0.f, pval = 0.f;
    // nodes are dealt to (block, group) round-robin: every partial sum has a fixed set of nodes in a fixed order
    for (int v = blockIdx.x * NPB + grp; v < V; v += gridDim.x * NPB) {
        const int g = graph_of[v];
        float dO = 0.f;
        if ((unsigned)g < (unsigned)G) {
            if (d_out) dO = d_out[g];
            if (target) {
                const float m = mask ? mask[g] : 1.0f;
                const float diff = (out[g] - target[g]) * m;
                dO += d_num * diff * m + d_abs * m * (diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f));
            }
        }
        if (node_mask) dO *= node_mask[v];
        const float gt = gate[v], vl = val[v];
        const float dval = dO * gt, dpre = dO * vl * gt * (1.0f - gt);
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const int c4 = l + s * LPN;
            if (c4 < D4) {
                const size_t o = (size_t)v * D + 4 * c4;
                const f32x4 a = *reinterpret_cast<const f32x4*>(hT + o);
                const f32x4 b = *reinterpret_cast<const f32x4*>(h0 + o);
                f32x4 dh = dpre * wa[s] + dval * wt[s];
                if (accumulate) dh += *reinterpret_cast<const f32x4*>(d_hT + o);
                *reinterpret_cast<f32x4*>(d_hT + o) = dh;
                pa[s] += dpre * a; pb[s] += dpre * b; pt[s] += dval * a;
            }
        }
        ppre += dpre; pval += dval;
    }
    // block reduction over the NPB node groups, in group order
    const int W = 3 * D + 2, WL = 3 * D + 4;
    float* mine = red_s + grp * WL;
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const int c4 = l + s * LPN;
        if (c4 < D4) {
            *reinterpret_cast<f32x4*>(mine + 4 * c4) = pa[s];
            *reinterpret_cast<f32x4*>(mine + D + 4 * c4) = pb[s];
            *reinterpret_cast<f32x4*>(mine + 2 * D + 4 * c4) = pt[s];
        }
    }
    if (l == 0) { mine[3 * D] = ppre; mine[3 * D + 1] = pval; }
    __syncthreads();
    for (int i = threadIdx.x; i < W; i += 256) {
        float s = 0.f;
        for (int q = 0; q < NPB; ++q) s += red_s[q * WL + i];
        partials[(size_t)blockIdx.x * W + i] = s;
    }
}

// A block = 32 columns x the 8 chains (thread (c, q) sums the block partials b == q (mod 8) of its column in order; the chains are
// combined pairwise through LDS): the same sums in the same order as one thread walking eight chains, with 8x the threads -- the
// reduction over up to 1024 block partials was a 42 us launch on the training step's critical path.
__global__ __launch_bounds__(256) void readout_bwd_final_kernel(const float* __restrict__ partials, int nblocks, int D,
                                                                float* __restrict__ d_gate_W, float* __restrict__ d_gate_b,
                                                                float* __restrict__ d_transform_W, float* __restrict__ d_transform_b) {
    __shared__ float red[8][32];
    const int W = 3 * D + 2;
    const int c = threadIdx.x & 31, q = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + c;
    float acc = 0.f;
    if (i < W) {
        int b = q;
        for (; b + 24 < nblocks; b += 32) {
            const float v0 = partials[(size_t)b * W + i], v1 = partials[(size_t)(b + 8) * W + i];
            const float v2 = partials[(size_t)(b + 16) * W + i], v3 = partials[(size_t)(b + 24) * W + i];
            acc += v0; acc += v1; acc += v2; acc += v3;
        }
        for (; b < nblocks; b += 8) acc += partials[(size_t)b * W + i];
    }
    red[q][c] = acc;
    __syncthreads();
    if (q != 0 || i >= W) return;
    const float s = ((red[0][c] + red[1][c]) + (red[2][c] + red[3][c])) + ((red[4][c] + red[5][c]) + (red[6][c] + red[7][c]));
    if (i < 2 * D) d_gate_W[i] = s;
    else if (i < 3 * D) d_transform_W[i - 2 * D] = s;
    else if (i == 3 * D) d_gate_b[0] = s;
    else d_transform_b[0] = s;
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

extern "C" size_t ggnn_readout_workspace_bytes(int V, int D, int num_graphs) {
    if (V < 0 || D <= 0 || num_graphs < 0) return 0;
    const size_t fwd = (size_t)((num_graphs + 255) / 256) * 3 * sizeof(float);
    const size_t bwd = (size_t)kReadoutBwdBlocks * (3 * (size_t)D + 2) * sizeof(float);
    return (fwd > bwd ? fwd : bwd) + 256;
}

extern "C" int ggnn_readout_loss_fwd_f32(const float* hT, const float* h0, const int32_t* graph_nodes_list, const int32_t* graph_ptr,
                                         const float* node_mask, const float* gate_W, const float* gate_b, const float* transform_W,
                                         const float* transform_b, const float* target, const float* mask, float* out,
                                         float* node_gate, float* node_val, float* stats, void* ws, size_t ws_bytes, int V, int D,
                                         int num_graphs, ggnn_stream_t stream) {
    GGNN_CHECK_ARG(V >= 0 && D > 0 && D % 4 == 0 && num_graphs >= 0, "bad sizes V=%d D=%d G=%d", V, D, num_graphs);
    hipStream_t st = (hipStream_t)stream;
    if (stats) GGNN_CHECK_HIP(hipMemsetAsync(stats, 0, 3 * sizeof(float), st));
    if (num_graphs == 0) return GGNN_OK;
    GGNN_CHECK_ARG(out && (V == 0 || (hT && h0 && graph_nodes_list && gate_W && gate_b && transform_W && transform_b && node_gate && node_val)),
                   "null pointer");
    GGNN_CHECK_ARG(!stats || (target && ws), "stats need target values and a workspace");
    if (stats && ws_bytes < ggnn_readout_workspace_bytes(V, D, num_graphs)) return fail(GGNN_E_WORKSPACE, "readout workspace too small");
    GGNN_CHECK_ARG(aligned16(hT) && aligned16(h0) && aligned16(gate_W) && aligned16(transform_W), "pointers must be 16-byte aligned");
    if (V > 0) {
        hipLaunchKernelGGL(readout_node_kernel, dim3((V + 15) / 16), dim3(256), 0, st, hT, h0, gate_W, gate_b, transform_W, transform_b,
                           node_gate, node_val, V, D);
        GGNN_CHECK_HIP(hipGetLastError());
    }
    const int nb = (num_graphs + 255) / 256;
    float* partials = stats ? reinterpret_cast<float*>((reinterpret_cast<size_t>(ws) + 255) / 256 * 256) : nullptr;
    hipLaunchKernelGGL(readout_graph_kernel, dim3(nb), dim3(256), 0, st, (const float*)node_gate, (const float*)node_val, graph_nodes_list,
                       graph_ptr, node_mask, target, mask, out, partials, V, num_graphs);
    GGNN_CHECK_HIP(hipGetLastError());
    if (stats) {
        hipLaunchKernelGGL(readout_stats_kernel, dim3(1), dim3(64), 0, st, (const float*)partials, nb, stats);
        GGNN_CHECK_HIP(hipGetLastError());
    }
    return GGNN_OK;
}

extern "C" int ggnn_readout_loss_bwd_f32(const float* hT, const float* h0, const int32_t* graph_nodes_list, const float* node_mask,
                                         const float* gate_W, const float* transform_W, const float* node_gate, const float* node_val,
                                         const float* out, const float* target, const float* mask, const float* d_out,
                                         const float* d_stats, float* d_hT, int accumulate, float* d_gate_W, float* d_gate_b,
                                         float* d_transform_W, float* d_transform_b, void* ws, size_t ws_bytes, int V, int D,
                                         int num_graphs, ggnn_stream_t stream) {
    GGNN_CHECK_ARG(V >= 0 && D > 0 && D % 4 == 0 && num_graphs >= 0, "bad sizes V=%d D=%d G=%d", V, D, num_graphs);
    if (D > 4 * kReadoutLanes * kReadoutMaxSlots) return fail(GGNN_E_UNSUPPORTED, "readout backward supports hidden sizes up to %d", 4 * kReadoutLanes * kReadoutMaxSlots);
    GGNN_CHECK_ARG(d_gate_W && d_gate_b && d_transform_W && d_transform_b && ws, "null pointer");
    if (ws_bytes < ggnn_readout_workspace_bytes(V, D, num_graphs)) return fail(GGNN_E_WORKSPACE, "readout workspace too small");
    hipStream_t st = (hipStream_t)stream;
    float* partials = reinterpret_cast<float*>((reinterpret_cast<size_t>(ws) + 255) / 256 * 256);
    const int W = 3 * D + 2;
    int nb = 0;
    if (V > 0 && num_graphs > 0) {
        GGNN_CHECK_ARG(hT && h0 && graph_nodes_list && gate_W && transform_W && node_gate && node_val && out && d_hT, "null pointer");
        GGNN_CHECK_ARG(!d_stats || target, "d_stats needs target values");
        GGNN_CHECK_ARG(aligned16(hT) && aligned16(h0) && aligned16(gate_W) && aligned16(transform_W) && aligned16(d_hT), "pointers must be 16-byte aligned");
        nb = (V + 15) / 16;
        if (nb > kReadoutBwdBlocks) nb = kReadoutBwdBlocks;
        const size_t lds = (size_t)16 * (W + 2) * sizeof(float);         // 16 node groups x (3 D + 4) floats
        static std::atomic<unsigned long long> lds_ok{0};
        if (lds > 48 * 1024) GGNN_CHECK_HIP(allow_dynamic_lds(&readout_bwd_node_kernel, lds, lds_ok));
        hipLaunchKernelGGL(readout_bwd_node_kernel, dim3(nb), dim3(256), lds, st, hT, h0, graph_nodes_list, node_mask, gate_W, transform_W,
                           node_gate, node_val, out, target, mask, d_out, d_stats, d_hT, accumulate, partials, V, D, num_graphs);
        GGNN_CHECK_HIP(hipGetLastError());
    }
    hipLaunchKernelGGL(readout_bwd_final_kernel, dim3((W + 31) / 32), dim3(256), 0, st, (const float*)partials, nb, D, d_gate_W, d_gate_b,
                       d_transform_W, d_transform_b);
    GGNN_CHECK_HIP(hipGetLastError());
    return GGNN_OK;
}
