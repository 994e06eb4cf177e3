// The layer / timestep driver of chem_tensorflow_sparse.py:131-218 in native code: ONE C-ABI call enqueues the
// whole forward propagation (all layers x timesteps: message transform -> gather/segment-sum -> GRU) on the
// stream.  The Python host layer would otherwise make ~3 ctypes calls and ~5 tensor allocations per timestep
// (24 + 40 per 8-step forward, ~1.2 ms of host time against ~1.6 ms of GPU time: the loop was close to
// host-bound on slow hosts).  No allocation, no sync: scratch comes from the caller.
//
// Per timestep it launches either 3 kernels (transform, gather/segment-sum, GRU) or -- with `fuse_gather`, packed
// GRU weights, a fused hidden size and no edge bias -- 2 kernels (transform, GRU with the segment sum gathered
// inside: ggnn_gru_packed_gather_f32).
#include "ggnn_common.h"
#include <cstring>

static constexpr int kTileCounters = 1024;     // timesteps per call that get a dynamic tile counter (the rest split statically)

extern "C" size_t ggnn_sparse_propagate_workspace_bytes(int V, int D, int T, int64_t compact_rows) {
    if (V < 0 || D <= 0 || T <= 0) return 0;
    const size_t vd = (size_t)V * D * sizeof(float);
    const size_t hrows = compact_rows >= 0 ? (size_t)(compact_rows > 0 ? compact_rows : 1) * D * sizeof(float)
                                           : (size_t)V * T * D * sizeof(float);
    // transformed states + incoming + two ping-pong states + GRU scratch (un-fused sizes) + tile counters of the
    // fused GRU launches (one int32 per timestep) + 256-B alignment slack
    return hrows + 3 * vd + ggnn_gru_workspace_bytes(V, D) + kTileCounters * sizeof(int32_t) + 6 * 256;
}

static inline char* bump(char*& p, size_t bytes) {
    char* r = p;
    p += (bytes + 255) / 256 * 256;
    return r;
}

extern "C" int ggnn_sparse_propagate_f32(
        const float* h0, int V, int D, int T,
        const int32_t* row_ptr, const int32_t* gather_row, const int32_t* pair_node, const int64_t* type_row_off,
        const float* nin, int use_avg,
        int num_layers, const int32_t* layer_timesteps, const int32_t* res_ptr, const int32_t* res_idx,
        const float* const* edge_w, const float* const* edge_packed, const float* const* edge_bias,
        const float* const* Wg, const float* const* bg, const float* const* Wc, const float* const* bc,
        const float* const* gru_packed, const int32_t* gru_fmt, const int32_t* edge_fmt, int act, int fuse_gather,
        float* const* layer_out, void* ws, size_t ws_bytes, ggnn_stream_t stream) {
    GGNN_CHECK_ARG(V >= 0 && D > 0 && D % 4 == 0 && T > 0, "bad sizes V=%d D=%d T=%d", V, D, T);
    GGNN_CHECK_ARG(num_layers > 0 && layer_timesteps && res_ptr && layer_out, "bad layer description");
    GGNN_CHECK_ARG(edge_w || edge_packed, "edge weights missing");
    GGNN_CHECK_ARG(bg && bc && (gru_packed || (Wg && Wc)), "GRU weights missing");
    if (V == 0) return GGNN_OK;
    GGNN_CHECK_ARG(h0 && row_ptr && ws, "null pointer");
    const bool compact = pair_node != nullptr && type_row_off != nullptr;
    const int64_t rows = compact ? type_row_off[T] : -1;
    if (ws_bytes < ggnn_sparse_propagate_workspace_bytes(V, D, T, rows))
        return ggnn::fail(GGNN_E_WORKSPACE, "propagate workspace too small: %zu < %zu", ws_bytes,
                          ggnn_sparse_propagate_workspace_bytes(V, D, T, rows));
    const size_t vd = (size_t)V * D * sizeof(float);
    char* p = reinterpret_cast<char*>((reinterpret_cast<size_t>(ws) + 255) / 256 * 256);
    float* H = reinterpret_cast<float*>(bump(p, compact ? (size_t)(rows > 0 ? rows : 1) * D * sizeof(float)
                                                       : (size_t)V * T * D * sizeof(float)));
    float* incoming = reinterpret_cast<float*>(bump(p, vd));
    float* ping[2] = {reinterpret_cast<float*>(bump(p, vd)), nullptr};
    ping[1] = reinterpret_cast<float*>(bump(p, vd));
    int32_t* counters = reinterpret_cast<int32_t*>(bump(p, kTileCounters * sizeof(int32_t)));
    void* gru_ws = p;
    const size_t gru_ws_bytes = ggnn_gru_workspace_bytes(V, D);
    int step_no = 0;
    {
        int total_steps = 0;
        for (int l = 0; l < num_layers; ++l) total_steps += layer_timesteps[l] > 0 ? layer_timesteps[l] : 0;
        if (total_steps > kTileCounters) total_steps = kTileCounters;
        if (total_steps > 0) GGNN_CHECK_HIP(hipMemsetAsync(counters, 0, (size_t)total_steps * sizeof(int32_t), (hipStream_t)stream));
    }

    const float* states[64];                       // node_states_per_layer (:118-119, :152)
    GGNN_CHECK_ARG(num_layers < 63, "too many layers");
    states[0] = h0;
    for (int l = 0; l < num_layers; ++l) {
        const int nres = res_ptr[l + 1] - res_ptr[l];
        // any number of residual inputs in the reference (:139-145); here up to 6 (the generic GEMM's 8 K segments = 6 + messages
        // + h); the fused single-launch GRU kernels take 2, layers with more run the two-launch generic GRU
        GGNN_CHECK_ARG(nres >= 0 && nres <= 6, "layer %d has %d residual inputs (max 6)", l, nres);
        const float* xs[7];
        for (int i = 0; i < nres; ++i) {
            const int src = res_idx[res_ptr[l] + i];
            GGNN_CHECK_ARG(src >= 0 && src <= l, "layer %d: residual index %d refers to a later layer", l, src);
            xs[i] = states[src];                   // :140-145
        }
        xs[nres] = incoming;                       // :211-212 residual states first, aggregated messages last
        const int nx = nres + 1;
        GGNN_CHECK_ARG(layer_out[l], "layer_out[%d] is null", l);
        const float* cur = states[l];              // :152
        const int steps = layer_timesteps[l];
        const float* bias_l = edge_bias ? edge_bias[l] : nullptr;
        const bool packed_gru = gru_packed && gru_packed[l] && ggnn_gru_is_fused(D) && nx <= 3;
        const int fmt_l = gru_fmt ? gru_fmt[l] : GGNN_GRU_FMT_BF16X3;      // the format gru_packed[l] was packed in
        // fuse_gather = the largest number of concatenated GRU inputs (residuals + messages) for which the segment sum
        // is gathered inside the GRU kernel; 0 = never.
        const bool gather_in_gru = fuse_gather > 0 && nx <= fuse_gather && packed_gru && ggnn_gru_is_fused(D) == 1 && bias_l == nullptr &&
                                   (unsigned long long)V * T * D < (1ULL << 30);      // (its 32-bit byte offsets)
        for (int s = 0; s < steps; ++s) {          // :153
            int rc;
            if (compact) {
                const bool packed = edge_packed && edge_packed[l];
                rc = ggnn_msg_transform_compact_f32(cur, packed ? nullptr : edge_w[l], pair_node, type_row_off, H,
                                                    packed ? const_cast<float*>(edge_packed[l]) : gru_ws,
                                                    packed ? ggnn_msg_transform_compact_workspace_bytes(D, T) : gru_ws_bytes,
                                                    V, D, T, (packed && edge_fmt) ? edge_fmt[l] : GGNN_GRU_FMT_BF16X3, stream);
            } else {
                GGNN_CHECK_ARG(edge_w && edge_w[l], "dense transform needs raw edge weights");
                rc = ggnn_msg_transform_f32(cur, D, edge_w[l], H, V, D, T, stream);
            }
            if (rc) return rc;
            float* out = (s + 1 == steps) ? layer_out[l] : ping[s & 1];
            static const bool dyn_tiles = [] { const char* e = getenv("GGNN_DYN_TILES"); return !e || atoi(e) != 0; }();
            int32_t* counter = (dyn_tiles && step_no < kTileCounters) ? counters + step_no : nullptr;
            ++step_no;
            if (gather_in_gru) {
                rc = ggnn_gru_packed_gather_f32(xs, nx, cur, gru_packed[l], bg[l], bc[l], out, H, row_ptr, gather_row, nin, T,
                                                use_avg, V, D, act, fmt_l, counter, stream);
            } else {
                rc = ggnn_gather_segment_sum_f32(H, row_ptr, gather_row, nin, bias_l, use_avg, incoming, V, D, T, stream);
                if (rc) return rc;
                if (packed_gru)
                    rc = ggnn_gru_packed_f32(xs, nx, cur, gru_packed[l], bg[l], bc[l], out, nullptr, nullptr, nullptr, V, D, act,
                                             fmt_l, counter, stream);
                else {
                    GGNN_CHECK_ARG(Wg && Wc && Wg[l] && Wc[l], "layer %d needs raw GRU weights (no packed images for it)", l);
                    rc = ggnn_gru_f32(xs, nx, cur, Wg[l], bg[l], Wc[l], bc[l], out, gru_ws, gru_ws_bytes, nullptr, nullptr,
                                      nullptr, V, D, act, stream);
                }
            }
            if (rc) return rc;
            cur = out;
        }
        if (steps == 0) {                          // a layer without timesteps forwards its input state
            GGNN_CHECK_HIP(hipMemcpyAsync(layer_out[l], cur, vd, hipMemcpyDeviceToDevice, (hipStream_t)stream));
        }
        states[l + 1] = layer_out[l];
    }
    return GGNN_OK;
}
