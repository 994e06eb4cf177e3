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

// The five gathers of a batch are independent of each other (every one derives the graph of its element from the prefix-sum tables),
// so they run as ONE launch: consecutive block ranges work on nodes / state rows / messages / slots / compact rows (PackPlan).  A
// batch is assembled next to a forward pass whose persistent workgroups own every CU for the length of a launch: each dependent
// launch of the packer then waits for a launch boundary of the forward stream, and five of them (plus the slot heads) were most of
// the time a batch spent coming together.
struct PackPlan { int e_nodes, e_states, e_msgs, e_slots, e_pairs; };      // first block AFTER each task's range

// compact row (slot_crow given) or transformed-state row of the message in dataset slot q of graph k (= dataset graph g)
__device__ __forceinline__ void slot_rows(const PackTables& ds, const PackBatch& b, int q, int k, int g, int& row, int& crow) {
    const int val = ds.slot_gather[q];
    const int src = val / ds.T, t = val - src * ds.T;
    row = (src - ds.node_ptr[g] + b.node_off[k]) * ds.T + t;
    crow = ds.slot_crow ? b.type_row_off[t] + b.pair_off[(size_t)t * (b.G + 1) + k] + (ds.slot_crow[q] - ds.type_row_off[t] - ds.p_off[g * ds.T + t]) : -1;
}

__device__ __forceinline__ void pack_nodes(const PackTables& ds, const PackBatch& b, int v, int* __restrict__ gnl, int* __restrict__ graph_ptr,
                                           float* __restrict__ nin, int* __restrict__ row_ptr, int4* __restrict__ heads) {
    if (v <= b.G) graph_ptr[v] = b.node_off[v];
    if (v == b.V) row_ptr[v] = b.M;
    if (v >= b.V) return;
    // (empty graphs repeat an offset: the LAST k with node_off[k] <= v is the graph that owns v)
    const int k = seg_of(b.node_off, b.G + 1, v);
    const int g = b.gid[k];
    const int s = ds.node_ptr[g] + (v - b.node_off[k]);
    gnl[v] = k;
    for (int t = 0; t < ds.T; ++t) nin[(size_t)v * ds.T + t] = ds.nin[(size_t)s * ds.T + t];
    const int q0 = ds.row_ptr[s];
    row_ptr[v] = b.slot_off[k] + (q0 - ds.row_ptr[ds.node_ptr[g]]);
    if (heads) {
        // the node's first four gather rows (ggnn_build_slot_heads) -- of the compacted rows when the batch has them
        const int n = ds.row_ptr[s + 1] - q0;
        int h[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            h[j] = -1;
            if (j < n) {
                int row, crow;
                slot_rows(ds, b, q0 + j, k, g, row, crow);
                h[j] = ds.slot_crow ? crow : row;
            }
        }
        heads[v] = make_int4(h[0], h[1], h[2], h[3]);
    }
}

// h0[v, :] = [annotation | 0 ...]   (:300-302), one lane per (node, float4 column)
__device__ __forceinline__ void pack_states(const PackTables& ds, const PackBatch& b, long long i, float* __restrict__ h0) {
    const int D4 = b.D >> 2;
    if (i >= (long long)b.V * D4) return;
    const int v = (int)(i / D4), c4 = (int)(i - (long long)v * D4);
    const int k = seg_of(b.node_off, b.G + 1, v);          // (the D/4 lanes of a row search the same path: broadcast loads)
    const int s = ds.node_ptr[b.gid[k]] + (v - b.node_off[k]);
    float o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int col = 4 * c4 + e;
        o[e] = col < ds.A ? ds.feat[(size_t)s * ds.A + col] : 0.f;
    }
    *reinterpret_cast<float4*>(h0 + (size_t)v * b.D + 4 * c4) = make_float4(o[0], o[1], o[2], o[3]);
}

__device__ __forceinline__ void pack_messages(const PackTables& ds, const PackBatch& b, int p, int2* __restrict__ adj) {
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

__device__ __forceinline__ void pack_slots(const PackTables& ds, const PackBatch& b, int s, int* __restrict__ gather_row,
                                           int* __restrict__ msg_perm, int* __restrict__ gather_c) {
    if (s >= b.M) return;
    const int k = seg_of(b.slot_off, b.G + 1, s);
    const int g = b.gid[k];
    const int q = ds.row_ptr[ds.node_ptr[g]] + (s - b.slot_off[k]);
    int row, crow;
    slot_rows(ds, b, q, k, g, row, crow);
    gather_row[s] = row;
    const int t = row % ds.T;
    msg_perm[s] = b.type_off[t] + b.msg_off[(size_t)t * (b.G + 1) + k] + (ds.slot_msg[q] - ds.type_off[t] - ds.e_off[g * ds.T + t]);
    if (gather_c) gather_c[s] = crow;
}

__device__ __forceinline__ void pack_pairs(const PackTables& ds, const PackBatch& b, int r, int* __restrict__ pair_node) {
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

__global__ __launch_bounds__(256) void pack_batch_kernel(PackTables ds, PackBatch b, PackPlan plan, float* __restrict__ h0, int* __restrict__ gnl,
                                                         int* __restrict__ graph_ptr, float* __restrict__ nin, int2* __restrict__ adj,
                                                         int* __restrict__ row_ptr, int* __restrict__ gather_row, int* __restrict__ msg_perm,
                                                         int* __restrict__ pair_node, int* __restrict__ gather_c, int4* __restrict__ heads) {
    const int blk = blockIdx.x, tid = threadIdx.x;
    if (blk < plan.e_nodes) pack_nodes(ds, b, blk * 256 + tid, gnl, graph_ptr, nin, row_ptr, heads);
    else if (blk < plan.e_states) pack_states(ds, b, (long long)(blk - plan.e_nodes) * 256 + tid, h0);
    else if (blk < plan.e_msgs) pack_messages(ds, b, (blk - plan.e_states) * 256 + tid, adj);
    else if (blk < plan.e_slots) pack_slots(ds, b, (blk - plan.e_msgs) * 256 + tid, gather_row, msg_perm, gather_c);
    else pack_pairs(ds, b, (blk - plan.e_slots) * 256 + tid, pair_node);
}

// ---- the (graph, type) prefix sums of a batch and its labels, on the device -----------------------------------------------------------
// batch_tab = gid[G] | pre[rows][G+1] with pre[r][k] = sum of counts_t[r][gid[j]] over j < k  (rows: nodes, message slots, messages per
// type, compact rows per type, compact rows -- the order ggnn_assemble_batch[_backward] read them in); one 1024-thread block per row.
// Further blocks gather the batch's labels: target_values[i][j] = targets[gid[j]][task_ids[i]] * mask, target_mask[i][j] = mask
// (chem_tensorflow_sparse.py:319-321, 335: masked labels feed 0).  Replaces ~10 small torch launches per batch by one.
__global__ __launch_bounds__(1024) void pack_tables_kernel(const int* __restrict__ counts_t, int Gd, int rows, const long long* __restrict__ gids,
                                                           int G, int* __restrict__ batch_tab, const float* __restrict__ targets,
                                                           const float* __restrict__ label_mask, int num_targets,
                                                           const long long* __restrict__ task_ids, int K, float* __restrict__ tv,
                                                           float* __restrict__ tm) {
    __shared__ int part[1024];
    const int tid = threadIdx.x;
    if ((int)blockIdx.x < rows) {
        const int row = blockIdx.x;
        const int* c = counts_t + (size_t)row * Gd;
        int* pre = batch_tab + G + (size_t)row * (G + 1);
        const int chunk = (G + 1023) / 1024;
        const int beg = min(tid * chunk, G), end = min(beg + chunk, G);
        int sum = 0;
        for (int j = beg; j < end; ++j) sum += c[gids[j]];
        part[tid] = sum;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {                  // inclusive scan of the 1024 chunk sums
            const int add = tid >= off ? part[tid - off] : 0;
            __syncthreads();
            part[tid] += add;
            __syncthreads();
        }
        int run = part[tid] - sum;
        if (tid == 0) pre[0] = 0;
        for (int j = beg; j < end; ++j) {
            run += c[gids[j]];
            pre[j + 1] = run;
            if (row == 0) batch_tab[j] = (int)gids[j];
        }
    } else {
        const long long i = (long long)(blockIdx.x - rows) * 1024 + tid;
        if (i >= (long long)K * G) return;
        const int ti = (int)(i / G), j = (int)(i - (long long)ti * G);
        const size_t src = (size_t)gids[j] * num_targets + task_ids[ti];
        const float m = label_mask ? label_mask[src] : 1.0f;
        tv[i] = targets[src] * m;
        tm[i] = m;
    }
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

__device__ __forceinline__ void pack_src_rowptr(const PackTables& ds, const PackBwdTables& bw, const PackBatch& b, long long i,
                                                const int* __restrict__ gnl, int* __restrict__ src_row_ptr) {
    const long long n = (long long)b.V * ds.T;
    if (i == n) src_row_ptr[i] = b.M;
    if (i >= n) return;
    const int v = (int)(i / ds.T), t = (int)(i - (long long)v * ds.T);
    const int k = gnl[v], g = b.gid[k];
    const int n0 = ds.node_ptr[g];
    const int s = n0 + (v - b.node_off[k]);
    src_row_ptr[i] = b.slot_off[k] + (bw.src_row_ptr[(size_t)s * ds.T + t] - bw.src_row_ptr[(size_t)n0 * ds.T]);
}

__device__ __forceinline__ void pack_src_slots(const PackTables& ds, const PackBwdTables& bw, const PackBatch& b, int s,
                                               int* __restrict__ src_gather, int* __restrict__ src_msg) {
    if (s >= b.M) return;
    const int k = seg_of(b.slot_off, b.G + 1, s);
    const int g = b.gid[k];
    const int q = bw.src_row_ptr[(size_t)ds.node_ptr[g] * ds.T] + (s - b.slot_off[k]);
    src_gather[s] = bw.src_gather[q] - ds.node_ptr[g] + b.node_off[k];
    src_msg[s] = remap_msg(ds, b, bw.src_msg[q], k, g);
}

// slots of the compact rows: type-major like the message lists, so the (type, graph) lookup of pack_messages_kernel applies
__device__ __forceinline__ void pack_rows_slots(const PackTables& ds, const PackBwdTables& bw, const PackBatch& b, int p,
                                                int* __restrict__ rows_gather, int* __restrict__ rows_msg) {
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

__device__ __forceinline__ void pack_rows_rp(const PackTables& ds, const PackBwdTables& bw, const PackBatch& b, int r, int* __restrict__ rows_rp) {
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

__device__ __forceinline__ void pack_node_rows(const PackTables& ds, const PackBwdTables& bw, const PackBatch& b, int i,
                                               const int* __restrict__ gnl, const int* __restrict__ ptot, int* __restrict__ node_rp,
                                               int* __restrict__ node_order) {
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

struct PackBwdPlan { int e_rowptr, e_src, e_rows, e_rp; };                 // first block AFTER each task's range; the rest: node rows

__global__ __launch_bounds__(256) void pack_batch_backward_kernel(PackTables ds, PackBwdTables bw, PackBatch b, PackBwdPlan plan,
                                                                  const int* __restrict__ gnl, const int* __restrict__ ptot,
                                                                  int* __restrict__ src_row_ptr, int* __restrict__ src_gather,
                                                                  int* __restrict__ src_msg, int* __restrict__ rows_rp,
                                                                  int* __restrict__ rows_gather, int* __restrict__ rows_msg,
                                                                  int* __restrict__ node_rp, int* __restrict__ node_order) {
    const int blk = blockIdx.x, tid = threadIdx.x;
    if (blk < plan.e_rowptr) pack_src_rowptr(ds, bw, b, (long long)blk * 256 + tid, gnl, src_row_ptr);
    else if (blk < plan.e_src) pack_src_slots(ds, bw, b, (blk - plan.e_rowptr) * 256 + tid, src_gather, src_msg);
    else if (blk < plan.e_rows) pack_rows_slots(ds, bw, b, (blk - plan.e_src) * 256 + tid, rows_gather, rows_msg);
    else if (blk < plan.e_rp) pack_rows_rp(ds, bw, b, (blk - plan.e_rows) * 256 + tid, rows_rp);
    else pack_node_rows(ds, bw, b, (blk - plan.e_rp) * 256 + tid, gnl, ptot, node_rp, node_order);
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
    GGNN_CHECK_ARG(nvt < (1LL << 31) - 256, "V*T overflows int32");
    PackBwdPlan plan{};
    plan.e_rowptr = (int)((nvt + 255) / 256);
    plan.e_src = plan.e_rowptr + (M + 255) / 256;
    plan.e_rows = plan.e_src + (M + 255) / 256;
    plan.e_rp = plan.e_rows + (R + 1 + 255) / 256;
    const int nn = (V > R ? V : R) + 1;
    const int blocks = plan.e_rp + (nn + 255) / 256;
    hipLaunchKernelGGL(pack_batch_backward_kernel, dim3(blocks), dim3(256), 0, st, ds, bw, b, plan, graph_nodes_list, ptot, src_row_ptr,
                       src_gather, src_msg, rows_rp, rows_gather, rows_msg, node_rp, node_order);
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
    int4* heads = reinterpret_cast<int4*>(out[10]);
    GGNN_CHECK_ARG(!heads || aligned16(heads), "slot heads must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const int nv = (V > G ? V : G) + 1;
    const long long n4 = (long long)V * (D / 4);
    GGNN_CHECK_ARG((n4 + 255) / 256 < (1LL << 30), "batch too large");
    PackPlan plan{};
    plan.e_nodes = (nv + 255) / 256;
    plan.e_states = plan.e_nodes + (int)((n4 + 255) / 256);
    plan.e_msgs = plan.e_states + (M + 255) / 256;
    plan.e_slots = plan.e_msgs + (M + 255) / 256;
    plan.e_pairs = plan.e_slots + (compact ? (R + 255) / 256 : 0);
    hipLaunchKernelGGL(pack_batch_kernel, dim3(plan.e_pairs), dim3(256), 0, st, ds, b, plan, h0, gnl, graph_ptr, nin, adj, row_ptr, gather_row,
                       msg_perm, pair_node, compact ? gather_c : nullptr, heads);
    GGNN_CHECK_HIP(hipGetLastError());
    return GGNN_OK;
}

extern "C" int ggnn_pack_batch_tables(const int32_t* counts_t, int Gd, int rows, const int64_t* gids, int G, const float* targets,
                                      const float* label_mask, int num_targets, const int64_t* task_ids, int K, int32_t* batch_tab,
                                      float* target_values, float* target_mask, ggnn_stream_t stream) {
    GGNN_CHECK_ARG(Gd >= 0 && rows > 0 && G >= 0 && K >= 0 && num_targets >= 0, "bad sizes Gd=%d rows=%d G=%d K=%d", Gd, rows, G, K);
    GGNN_CHECK_ARG(batch_tab && (G == 0 || (counts_t && gids)), "null pointer");
    GGNN_CHECK_ARG(K == 0 || G == 0 || (targets && task_ids && target_values && target_mask), "null label pointer");
    const long long nlab = (long long)K * G;
    const int blocks = rows + (int)((nlab + 1023) / 1024);
    hipLaunchKernelGGL(pack_tables_kernel, dim3(blocks), dim3(1024), 0, (hipStream_t)stream, counts_t, Gd, rows,
                       reinterpret_cast<const long long*>(gids), G, batch_tab, targets, label_mask, num_targets,
                       reinterpret_cast<const long long*>(task_ids), K, target_values, target_mask);
    GGNN_CHECK_HIP(hipGetLastError());
    return GGNN_OK;
}
