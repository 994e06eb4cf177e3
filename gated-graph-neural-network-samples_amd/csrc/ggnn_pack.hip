// Batch assembly from dataset-level tables (the step before the path: chem_tensorflow_sparse.py:278-350 + :120-129).
//
// A batch is a disjoint union of molecule graphs, nodes numbered graph after graph (:297-307).  Every index structure the
// propagation reads -- the per-type adjacency lists sorted by (src, dst) (:265, :343-348), the in-degree table (:310-313), the
// by-target slot lists of the segment sum, the compacted (source node, edge type) rows -- is therefore a CONCATENATION of per-graph
// pieces whose only batch-dependent part is an offset: the graph's first node, and per edge type the position of the graph's
// messages / compact rows in the batch's type-major lists.  The pieces are computed ONCE, for the whole dataset, by the general
// builders (rocPRIM sorts, scans: ggnn_build_target_csr, ggnn_build_compact_sources) run on all graphs as one giant batch;
// assembling a batch is then a handful of gather kernels driven by (graph, type) prefix sums the host gets from per-molecule count
// tables -- no sort, no scan, no device->host read, ~10 launches instead of ~120.
#include "ggnn_common.h"

namespace ggnn {

constexpr int kPackMaxTypes = 16;

struct PackTables {                      // dataset level (device pointers)
    const int* node_ptr;                 // [Gd+1] first node of every graph
    const float* feat; int A;            // [Nd, A] node annotations
    const float* nin;                    // [Nd, T] incoming edges per type
    const int* row_ptr;                  // [Nd+1] by-target slot ranges
    const int2* adj;                     // [Md] (src, dst), type-major, sorted by (src, dst) inside a type
    const int* slot_gather;              // [Md] slot -> src * T + type
    const int* slot_msg;                 // [Md] slot -> message id
    const int* slot_crow;                // [Md] slot -> compact row of the message's (src, type) pair   (NULL: no compaction)
    const int* pair_node;                // [Rd] compact row -> node                                       (NULL: no compaction)
    const int* e_off;                    // [Gd, T] first message of graph g inside the dataset's type-t list
    const int* p_off;                    // [Gd, T] first compact row of graph g inside the dataset's type-t rows
    int type_off[kPackMaxTypes + 1], type_row_off[kPackMaxTypes + 1];
    int T;
};

struct PackBatch {                       // batch level
    const int* gid;                      // [G] dataset graph ids, batch order
    const int* node_off;                 // [G+1]
    const int* slot_off;                 // [G+1]
    const int* msg_off;                  // [T][G+1] position of graph k's type-t messages inside the batch's type-t list
    const int* pair_off;                 // [T][G+1]
    int type_off[kPackMaxTypes + 1], type_row_off[kPackMaxTypes + 1];
    int G, V, M, R, D;
};

// largest k in [0, n) with tab[k] <= x   (tab non-decreasing, tab[0] <= x)
__device__ __forceinline__ int seg_of(const int* __restrict__ tab, int n, int x) {
    int lo = 0, hi = n;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (tab[mid] <= x) lo = mid; else hi = mid;
    }
    return lo;
}

__global__ void pack_nodes_kernel(PackTables ds, PackBatch b, int* __restrict__ gnl, int* __restrict__ graph_ptr,
                                  float* __restrict__ nin, int* __restrict__ row_ptr) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v <= b.G) graph_ptr[v] = b.node_off[v];
    if (v == b.V) row_ptr[v] = b.M;
    if (v >= b.V) return;
    // (empty graphs repeat an offset: the LAST k with node_off[k] <= v is the graph that owns v)
    const int k = seg_of(b.node_off, b.G + 1, v);
    const int g = b.gid[k];
    const int s = ds.node_ptr[g] + (v - b.node_off[k]);
    gnl[v] = k;
    for (int t = 0; t < ds.T; ++t) nin[(size_t)v * ds.T + t] = ds.nin[(size_t)s * ds.T + t];
    row_ptr[v] = b.slot_off[k] + (ds.row_ptr[s] - ds.row_ptr[ds.node_ptr[g]]);
}

// h0[v, :] = [annotation | 0 ...]   (:300-302), one lane per (node, float4 column)
__global__ void pack_states_kernel(PackTables ds, PackBatch b, const int* __restrict__ gnl, float* __restrict__ h0) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int D4 = b.D >> 2;
    if (i >= (long long)b.V * D4) return;
    const int v = (int)(i / D4), c4 = (int)(i - (long long)v * D4);
    const int k = gnl[v];
    const int s = ds.node_ptr[b.gid[k]] + (v - b.node_off[k]);
    float o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int col = 4 * c4 + e;
        o[e] = col < ds.A ? ds.feat[(size_t)s * ds.A + col] : 0.f;
    }
    *reinterpret_cast<float4*>(h0 + (size_t)v * b.D + 4 * c4) = make_float4(o[0], o[1], o[2], o[3]);
}

__global__ void pack_messages_kernel(PackTables ds, PackBatch b, int2* __restrict__ adj) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= b.M) return;
    int t = 0;
    while (t + 1 < ds.T && p >= b.type_off[t + 1]) ++t;
    const int w = p - b.type_off[t];                                   // position inside the batch's type-t list
    const int* mo = b.msg_off + (size_t)t * (b.G + 1);
    const int k = seg_of(mo, b.G + 1, w);
    const int g = b.gid[k];
    const int q = ds.type_off[t] + ds.e_off[(size_t)g * ds.T + t] + (w - mo[k]);
    const int shift = b.node_off[k] - ds.node_ptr[g];
    const int2 e = ds.adj[q];
    adj[p] = make_int2(e.x + shift, e.y + shift);
}

__global__ void pack_slots_kernel(PackTables ds, PackBatch b, int* __restrict__ gather_row, int* __restrict__ msg_perm,
                                  int* __restrict__ gather_c) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= b.M) return;
    const int k = seg_of(b.slot_off, b.G + 1, s);
    const int g = b.gid[k];
    const int q = ds.row_ptr[ds.node_ptr[g]] + (s - b.slot_off[k]);
    const int val = ds.slot_gather[q];
    const int src = val / ds.T, t = val - src * ds.T;
    gather_row[s] = (src - ds.node_ptr[g] + b.node_off[k]) * ds.T + t;
    const int gt = g * ds.T + t;
    msg_perm[s] = b.type_off[t] + b.msg_off[(size_t)t * (b.G + 1) + k] + (ds.slot_msg[q] - ds.type_off[t] - ds.e_off[gt]);
    if (gather_c) gather_c[s] = b.type_row_off[t] + b.pair_off[(size_t)t * (b.G + 1) + k] + (ds.slot_crow[q] - ds.type_row_off[t] - ds.p_off[gt]);
}

__global__ void pack_pairs_kernel(PackTables ds, PackBatch b, int* __restrict__ pair_node) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= b.R) return;
    int t = 0;
    while (t + 1 < ds.T && r >= b.type_row_off[t + 1]) ++t;
    const int w = r - b.type_row_off[t];
    const int* po = b.pair_off + (size_t)t * (b.G + 1);
    const int k = seg_of(po, b.G + 1, w);
    const int g = b.gid[k];
    const int q = ds.type_row_off[t] + ds.p_off[(size_t)g * ds.T + t] + (w - po[k]);
    pair_node[r] = ds.pair_node[q] - ds.node_ptr[g] + b.node_off[k];
}

// ---- the backward pass's transpose structures (ops.CompactBackward, the by-source CSR), gathered the same way -----------------------
struct PackBwdTables {
    const int* src_row_ptr;              // [Nd*T+1] by-(src*T+type) slot ranges
    const int* src_gather;               // [Md] slot -> dst node
    const int* src_msg;                  // [Md] slot -> message id
    const int* rows_rp;                  // [Rd+1] compact row -> first of its message slots (type-major slot order)
    const int* rows_gather;              // [Md] slot -> dst node
    const int* rows_msg;                 // [Md] slot -> message id
    const int* node_rp;                  // [Nd+1] node -> first of its compact rows in `node_order`
    const int* node_order;               // [Rd] compact rows by node, type ascending inside a node
};

__device__ __forceinline__ int type_of(const int* __restrict__ off, int T, int x) {
    int t = 0;
    while (t + 1 < T && x >= off[t + 1]) ++t;
    return t;
}

// dataset message id -> batch message id (graph k of the batch = dataset graph g)
__device__ __forceinline__ int remap_msg(const PackTables& ds, const PackBatch& b, int m_ds, int k, int g) {
    const int t = type_of(ds.type_off, ds.T, m_ds);
    return b.type_off[t] + b.msg_off[(size_t)t * (b.G + 1) + k] + (m_ds - ds.type_off[t] - ds.e_off[(size_t)g * ds.T + t]);
}

__global__ void pack_src_rowptr_kernel(PackTables ds, PackBwdTables bw, PackBatch b, const int* __restrict__ gnl, int* __restrict__ src_row_ptr) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long n = (long long)b.V * ds.T;
    if (i == n) src_row_ptr[i] = b.M;
    if (i >= n) return;
    const int v = (int)(i / ds.T), t = (int)(i - (long long)v * ds.T);
    const int k = gnl[v], g = b.gid[k];
    const int n0 = ds.node_ptr[g];
    const int s = n0 + (v - b.node_off[k]);
    src_row_ptr[i] = b.slot_off[k] + (bw.src_row_ptr[(size_t)s * ds.T + t] - bw.src_row_ptr[(size_t)n0 * ds.T]);
}

__global__ void pack_src_slots_kernel(PackTables ds, PackBwdTables bw, PackBatch b, int* __restrict__ src_gather, int* __restrict__ src_msg) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= b.M) return;
    const int k = seg_of(b.slot_off, b.G + 1, s);
    const int g = b.gid[k];
    const int q = bw.src_row_ptr[(size_t)ds.node_ptr[g] * ds.T] + (s - b.slot_off[k]);
    src_gather[s] = bw.src_gather[q] - ds.node_ptr[g] + b.node_off[k];
    src_msg[s] = remap_msg(ds, b, bw.src_msg[q], k, g);
}

// slots of the compact rows: type-major like the message lists, so the (type, graph) lookup of pack_messages_kernel applies
__global__ void pack_rows_slots_kernel(PackTables ds, PackBwdTables bw, PackBatch b, int* __restrict__ rows_gather, int* __restrict__ rows_msg) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= b.M) return;
    const int t = type_of(b.type_off, ds.T, p);
    const int w = p - b.type_off[t];
    const int* mo = b.msg_off + (size_t)t * (b.G + 1);
    const int k = seg_of(mo, b.G + 1, w);
    const int g = b.gid[k];
    const int q = ds.type_off[t] + ds.e_off[(size_t)g * ds.T + t] + (w - mo[k]);
    rows_gather[p] = bw.rows_gather[q] - ds.node_ptr[g] + b.node_off[k];
    rows_msg[p] = remap_msg(ds, b, bw.rows_msg[q], k, g);
}

__global__ void pack_rows_rp_kernel(PackTables ds, PackBwdTables bw, PackBatch b, int* __restrict__ rows_rp) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r == b.R) rows_rp[r] = b.M;
    if (r >= b.R) return;
    const int t = type_of(b.type_row_off, ds.T, r);
    const int w = r - b.type_row_off[t];
    const int* po = b.pair_off + (size_t)t * (b.G + 1);
    const int k = seg_of(po, b.G + 1, w);
    const int g = b.gid[k];
    const int q = ds.type_row_off[t] + ds.p_off[(size_t)g * ds.T + t] + (w - po[k]);
    // (the slot position counts messages in type-major order: the same arithmetic as a message id)
    rows_rp[r] = b.type_off[t] + b.msg_off[(size_t)t * (b.G + 1) + k] + (bw.rows_rp[q] - ds.type_off[t] - ds.e_off[(size_t)g * ds.T + t]);
}

__global__ void pack_node_rows_kernel(PackTables ds, PackBwdTables bw, PackBatch b, const int* __restrict__ gnl, const int* __restrict__ ptot,
                                      int* __restrict__ node_rp, int* __restrict__ node_order) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i <= b.V) {
        if (i == b.V) node_rp[i] = b.R;
        else {
            const int k = gnl[i], g = b.gid[k];
            const int n0 = ds.node_ptr[g];
            node_rp[i] = ptot[k] + (bw.node_rp[n0 + (i - b.node_off[k])] - bw.node_rp[n0]);
        }
    }
    if (i >= b.R) return;
    const int k = seg_of(ptot, b.G + 1, i);
    const int g = b.gid[k];
    const int r_ds = bw.node_order[bw.node_rp[ds.node_ptr[g]] + (i - ptot[k])];
    const int t = type_of(ds.type_row_off, ds.T, r_ds);
    node_order[i] = b.type_row_off[t] + b.pair_off[(size_t)t * (b.G + 1) + k] + (r_ds - ds.type_row_off[t] - ds.p_off[(size_t)g * ds.T + t]);
}

}  // namespace ggnn

using namespace ggnn;

static int fill_pack_args(PackTables& ds, PackBatch& b, const void* const* ds_tables, int A, int T, const int64_t* ds_type_off,
                          const int64_t* ds_type_row_off, const int32_t* batch_tab, int G, int V, int M, int R, int D,
                          const int64_t* type_off, const int64_t* type_row_off) {
    GGNN_CHECK_ARG(T > 0 && T <= kPackMaxTypes && A >= 0 && G >= 0 && V >= 0 && M >= 0 && R >= 0 && D > 0 && D % 4 == 0 && A <= D,
                   "bad sizes T=%d A=%d G=%d V=%d M=%d R=%d D=%d", T, A, G, V, M, R, D);
    GGNN_CHECK_ARG(ds_tables && ds_type_off && type_off && batch_tab, "null pointer");
    ds.node_ptr = static_cast<const int*>(ds_tables[0]); ds.feat = static_cast<const float*>(ds_tables[1]); ds.A = A;
    ds.nin = static_cast<const float*>(ds_tables[2]); ds.row_ptr = static_cast<const int*>(ds_tables[3]);
    ds.adj = static_cast<const int2*>(ds_tables[4]); ds.slot_gather = static_cast<const int*>(ds_tables[5]);
    ds.slot_msg = static_cast<const int*>(ds_tables[6]); ds.slot_crow = static_cast<const int*>(ds_tables[7]);
    ds.pair_node = static_cast<const int*>(ds_tables[8]); ds.e_off = static_cast<const int*>(ds_tables[9]);
    ds.p_off = static_cast<const int*>(ds_tables[10]); ds.T = T;
    const bool compact = ds.slot_crow != nullptr;
    GGNN_CHECK_ARG(!compact || (ds.pair_node && ds.p_off && ds_type_row_off && type_row_off), "compaction tables incomplete");
    b.G = G; b.V = V; b.M = M; b.R = compact ? R : 0; b.D = D;
    for (int t = 0; t <= T; ++t) {
        ds.type_off[t] = (int)ds_type_off[t]; b.type_off[t] = (int)type_off[t];
        ds.type_row_off[t] = compact ? (int)ds_type_row_off[t] : 0; b.type_row_off[t] = compact ? (int)type_row_off[t] : 0;
    }
    GGNN_CHECK_ARG(b.type_off[T] == M && (!compact || b.type_row_off[T] == R), "type offsets do not add up to M / R");
    b.gid = batch_tab; b.node_off = batch_tab + G; b.slot_off = b.node_off + (G + 1);
    b.msg_off = b.slot_off + (G + 1); b.pair_off = b.msg_off + (size_t)T * (G + 1);
    return GGNN_OK;
}

extern "C" int ggnn_assemble_batch_backward(const void* const* ds_tables, const void* const* bwd_tables, int A, int T,
                                            const int64_t* ds_type_off, const int64_t* ds_type_row_off, const int32_t* batch_tab,
                                            const int32_t* graph_nodes_list, int G, int V, int M, int R, int D,
                                            const int64_t* type_off, const int64_t* type_row_off, void* const* out, ggnn_stream_t stream) {
    PackTables ds{}; PackBatch b{};
    const int rc = fill_pack_args(ds, b, ds_tables, A, T, ds_type_off, ds_type_row_off, batch_tab, G, V, M, R, D, type_off, type_row_off);
    if (rc != GGNN_OK) return rc;
    GGNN_CHECK_ARG(bwd_tables && out && ds.slot_crow && (V == 0 || graph_nodes_list), "null pointer / no compaction tables");
    PackBwdTables bw{};
    bw.src_row_ptr = static_cast<const int*>(bwd_tables[0]); bw.src_gather = static_cast<const int*>(bwd_tables[1]);
    bw.src_msg = static_cast<const int*>(bwd_tables[2]); bw.rows_rp = static_cast<const int*>(bwd_tables[3]);
    bw.rows_gather = static_cast<const int*>(bwd_tables[4]); bw.rows_msg = static_cast<const int*>(bwd_tables[5]);
    bw.node_rp = static_cast<const int*>(bwd_tables[6]); bw.node_order = static_cast<const int*>(bwd_tables[7]);
    const int* ptot = b.pair_off + (size_t)T * (G + 1);             // [G+1] compact rows of the graphs before k, all types
    int* src_row_ptr = static_cast<int*>(out[0]); int* src_gather = static_cast<int*>(out[1]); int* src_msg = static_cast<int*>(out[2]);
    int* rows_rp = static_cast<int*>(out[3]); int* rows_gather = static_cast<int*>(out[4]); int* rows_msg = static_cast<int*>(out[5]);
    int* node_rp = static_cast<int*>(out[6]); int* node_order = static_cast<int*>(out[7]);
    GGNN_CHECK_ARG(src_row_ptr && rows_rp && node_rp && (M == 0 || (src_gather && src_msg && rows_gather && rows_msg)) && (R == 0 || node_order),
                   "null output");
    hipStream_t st = (hipStream_t)stream;
    const long long nvt = (long long)V * T + 1;
    hipLaunchKernelGGL(pack_src_rowptr_kernel, dim3((unsigned)((nvt + 255) / 256)), dim3(256), 0, st, ds, bw, b, graph_nodes_list, src_row_ptr);
    GGNN_CHECK_HIP(hipGetLastError());
    if (M) {
        hipLaunchKernelGGL(pack_src_slots_kernel, dim3((M + 255) / 256), dim3(256), 0, st, ds, bw, b, src_gather, src_msg);
        GGNN_CHECK_HIP(hipGetLastError());
        hipLaunchKernelGGL(pack_rows_slots_kernel, dim3((M + 255) / 256), dim3(256), 0, st, ds, bw, b, rows_gather, rows_msg);
        GGNN_CHECK_HIP(hipGetLastError());
    }
    hipLaunchKernelGGL(pack_rows_rp_kernel, dim3((R + 1 + 255) / 256), dim3(256), 0, st, ds, bw, b, rows_rp);
    GGNN_CHECK_HIP(hipGetLastError());
    const int nn = (V > R ? V : R) + 1;
    hipLaunchKernelGGL(pack_node_rows_kernel, dim3((nn + 255) / 256), dim3(256), 0, st, ds, bw, b, graph_nodes_list, ptot, node_rp, node_order);
    GGNN_CHECK_HIP(hipGetLastError());
    return GGNN_OK;
}

extern "C" int ggnn_assemble_batch(const void* const* ds_tables, int A, int T, const int64_t* ds_type_off, const int64_t* ds_type_row_off,
                                   const int32_t* batch_tab, int G, int V, int M, int R, int D, const int64_t* type_off,
                                   const int64_t* type_row_off, void* const* out, ggnn_stream_t stream) {
    PackTables ds{}; PackBatch b{};
    const int rc = fill_pack_args(ds, b, ds_tables, A, T, ds_type_off, ds_type_row_off, batch_tab, G, V, M, R, D, type_off, type_row_off);
    if (rc != GGNN_OK) return rc;
    GGNN_CHECK_ARG(out, "null pointer");
    const bool compact = ds.slot_crow != nullptr;
    float* h0 = static_cast<float*>(out[0]); int* gnl = static_cast<int*>(out[1]); int* graph_ptr = static_cast<int*>(out[2]);
    float* nin = static_cast<float*>(out[3]); int2* adj = static_cast<int2*>(out[4]); int* row_ptr = static_cast<int*>(out[5]);
    int* gather_row = static_cast<int*>(out[6]); int* msg_perm = static_cast<int*>(out[7]); int* pair_node = static_cast<int*>(out[8]);
    int* gather_c = static_cast<int*>(out[9]);
    GGNN_CHECK_ARG(graph_ptr && row_ptr && (V == 0 || (h0 && gnl && nin)) && (M == 0 || (adj && gather_row && msg_perm)), "null output");
    GGNN_CHECK_ARG(!compact || M == 0 || (gather_c && (R == 0 || pair_node)), "null compaction output");
    hipStream_t st = (hipStream_t)stream;
    const int nv = (V > G ? V : G) + 1;
    hipLaunchKernelGGL(pack_nodes_kernel, dim3((nv + 255) / 256), dim3(256), 0, st, ds, b, gnl, graph_ptr, nin, row_ptr);
    GGNN_CHECK_HIP(hipGetLastError());
    if (V) {
        const long long n4 = (long long)V * (D / 4);
        hipLaunchKernelGGL(pack_states_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, ds, b, (const int*)gnl, h0);
        GGNN_CHECK_HIP(hipGetLastError());
    }
    if (M) {
        hipLaunchKernelGGL(pack_messages_kernel, dim3((M + 255) / 256), dim3(256), 0, st, ds, b, adj);
        GGNN_CHECK_HIP(hipGetLastError());
        hipLaunchKernelGGL(pack_slots_kernel, dim3((M + 255) / 256), dim3(256), 0, st, ds, b, gather_row, msg_perm, compact ? gather_c : nullptr);
        GGNN_CHECK_HIP(hipGetLastError());
    }
    if (compact && R) {
        hipLaunchKernelGGL(pack_pairs_kernel, dim3((R + 255) / 256), dim3(256), 0, st, ds, b, pair_node);
        GGNN_CHECK_HIP(hipGetLastError());
    }
    return GGNN_OK;
}
