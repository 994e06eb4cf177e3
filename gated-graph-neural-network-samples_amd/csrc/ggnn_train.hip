// The optimisation step of the default sparse model (chem_tensorflow.py:183-191 over chem_tensorflow_sparse.py:117-218) as native
// launch sequences: ggnn_sparse_train_forward_f32 enqueues the whole training forward (compacted message transform + fused GRU
// with the segment sum gathered inside, r / u / c / incoming and every timestep's state kept), ggnn_sparse_train_backward_f32 the
// whole backward pass (fused GRU backward, transpose segment sums, compacted transform on W^T, every weight-gradient product on
// a side stream, accumulated straight into the optimizer's gradient buffers).  Between the two the host runs the readout + loss
// (ggnn_readout_loss_{fwd,bwd}_f32) and, under data parallelism, the all-reduce of the mask counts.
//
// Why: the same sequence driven from Python through torch.autograd is ~250 ctypes calls, tensor allocations and autograd nodes per
// step -- 4.4 ms of host time for ~6 ms of GPU time, and with a producer thread packing the next batch (which needs the interpreter
// lock) fresh-batch epochs ran at 6.9-7.3 ms per step instead of 6.1.  Here a step is two calls; nothing is allocated (one
// caller-provided workspace, laid out by train_layout below), nothing synchronises.
//
// Cross-stream hazards: everything a side-stream product reads (dpc, dpg, r*h, incoming, states, dHc) has its own buffer per
// timestep, so the main stream never waits for the side stream inside a step; the call ends with the main stream waiting for the
// side stream's last product, which orders the next step's forward (it reuses the workspace) behind them.
#include "ggnn_split.hpp"
#include "ggnn_philox.hpp"
#include <mutex>

namespace ggnn {
namespace {

constexpr int kMaxLayers = 62;
constexpr int kMaxNx = 3;

inline size_t al256(size_t x) { return (x + 255) / 256 * 256; }

struct TrainLayout {
    size_t Hc, state, r, u, c, inc, counters;           // forward
    size_t dpc, rh, dh, dpg, dx, dHc, Z, xty;            // backward
    size_t vd, rd, total;
    int steps;
};

TrainLayout train_layout(int V, int D, int T, int64_t R, int steps) {
    TrainLayout L{};
    L.steps = steps;
    // per-timestep buffers follow each other WITHOUT padding (V*D*4 is a multiple of 16): the buffers of consecutive timesteps then
    // form one [S*V, D] matrix, which lets the weight-gradient products of a layer's timesteps run as ONE product over S*V rows
    L.vd = (size_t)V * D * sizeof(float);
    L.rd = al256((size_t)(R > 0 ? R : 1) * D * sizeof(float));
    size_t p = 0;
    auto take = [&](size_t bytes) { const size_t at = p; p += al256(bytes); return at; };
    L.Hc = take(L.rd);
    L.state = take(L.vd * (steps + 1));                  // slot 0: a copy of h0; slot k+1: the state after timestep k
    L.r = take(L.vd * steps); L.u = take(L.vd * steps); L.c = take(L.vd * steps);
    L.inc = take(L.vd * steps);
    L.counters = take((size_t)steps * sizeof(int32_t));
    L.dpc = take(L.vd * steps); L.rh = take(L.vd * steps); L.dh = take(L.vd * steps); L.dpg = take(2 * L.vd * steps);
    L.dx = take(L.vd * steps * kMaxNx);
    L.dHc = take(L.rd * steps); L.Z = take(L.rd);
    L.xty = take(ggnn_xty_workspace_bytes((V > R ? V : (int)R) * (steps > 1 ? steps : 1), 4 * D, 2 * D, T));
    L.total = p + 256;
    (void)T;
    return L;
}

// events that order the side stream behind the main stream's producers (created once per device; recording an event that an
// earlier hipStreamWaitEvent already captured is fine: a wait refers to the record that preceded it)
struct EventPool {
    hipEvent_t ev[64];
    int next = 0;
    bool ready = false;
};
std::mutex g_pool_mutex;
EventPool g_pools[16];

int next_event(hipEvent_t* out) {
    int dev = 0;
    GGNN_CHECK_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(g_pool_mutex);
    EventPool& p = g_pools[dev & 15];
    if (!p.ready) {
        for (int i = 0; i < 64; ++i) GGNN_CHECK_HIP(hipEventCreateWithFlags(&p.ev[i], hipEventDisableTiming));
        p.ready = true;
    }
    *out = p.ev[p.next];
    p.next = (p.next + 1) & 63;
    return GGNN_OK;
}

int order_after(hipStream_t waiter, hipStream_t producer) {
    if (waiter == producer) return GGNN_OK;
    hipEvent_t e;
    if (int rc = next_event(&e)) return rc;
    GGNN_CHECK_HIP(hipEventRecord(e, producer));
    GGNN_CHECK_HIP(hipStreamWaitEvent(waiter, e, 0));
    return GGNN_OK;
}

__global__ void add_inplace_kernel(float* __restrict__ dst, const float* __restrict__ src, long long n4) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    f32x4 a = reinterpret_cast<f32x4*>(dst)[i];
    const f32x4 b = reinterpret_cast<const f32x4*>(src)[i];
    a += b;
    reinterpret_cast<f32x4*>(dst)[i] = a;
}

int add_inplace(float* dst, const float* src, long long n, hipStream_t st) {
    const long long n4 = n / 4;
    if (n4 == 0) return GGNN_OK;
    hipLaunchKernelGGL(add_inplace_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, dst, src, n4);
    GGNN_CHECK_HIP(hipGetLastError());
    return GGNN_OK;
}

// ---- a step's weight images in ONE launch ------------------------------------------------------------------------------------------
// Every optimisation step changes every weight, so every step rebuilds every stage image: per layer the T edge-weight images and
// the T images of their transposes (both of the weight-dropout-MASKED weights, the mask applied on the fly: ggnn_philox.hpp), the
// 3(nx+1) images of the fused GRU and the 3(nx+1) of its backward -- ~120 images of 48 KiB for the default model.  Driven layer by
// layer from the host that was 30 launches of 4 us each at the head of the step (mask, transpose copy, four pack kernels per layer).
constexpr int kPrepMaxLayers = 16;
struct PrepArgs {
    const float* edge_w[kPrepMaxLayers]; const float* Wg[kPrepMaxLayers]; const float* Wc[kPrepMaxLayers];
    float* edge_img[kPrepMaxLayers]; float* edge_img_t[kPrepMaxLayers]; float* gru_img[kPrepMaxLayers]; float* gru_bwd_img[kPrepMaxLayers];
    unsigned long long seed[kPrepMaxLayers];
    int nx[kPrepMaxLayers];
    int gru_fmt[kPrepMaxLayers];       // operand format of layer l's GRU FORWARD images (kSplitF16x2 / kSplitBf16x3)
    int T; float keep;
};

// SF: every image in split form (ggnn_split.hpp).  The GRU FORWARD's images of layer l are in the operand format a.gru_fmt[l] (the
// caller's per-layer choice, see include/ggnn_hip.h "Operand formats"); the edge-weight images -- whose transposes multiply
// gradients of any magnitude -- and the GRU backward's are always in the exact bf16 x 3 format.
template <int D, bool SF>
__global__ __launch_bounds__(256) void train_prepare_kernel(PrepArgs a) {
    using C = StageCfg<D>;
    const int l = blockIdx.z, i = blockIdx.y;
    const int first = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
    const int T = a.T, ng = 3 * (a.nx[l] + 1);
    if (i < 2 * T) {
        // edge-weight image of type t (i < T), or of its transpose (the backward's Z = dHc W_t^T); masked like ggnn_dropout_f32 masks
        // the [T*D, D] variable: row key = t*D + (row of W_t), column = column of W_t
        const bool tr = i >= T;
        const int t = tr ? i - T : i;
        const float* W = a.edge_w[l] + (size_t)t * D * D;
        const float keep = a.keep;
        const unsigned long long seed = a.seed[l];
        auto value = [&](int k, int n) -> float {                         // image[k][n] = W_t[k][n], or W_t[n][k]
            if (n >= D) return 0.f;
            const int r = tr ? n : k, c = tr ? k : n;
            float v = W[(size_t)r * D + c];
            if (keep < 1.0f) v = dropout_apply(v, keep, seed, (unsigned long long)(t * D + r), c);
            return v;
        };
        float* img = (tr ? a.edge_img_t[l] : a.edge_img[l]) + (size_t)t * ImgCfg<D, SF>::IMG;
        if constexpr (SF) {
            pack_split_image<D>(value, img, first, stride);
        } else {
            for (int j = first; j < C::IMG; j += stride) {
                float v = 0.f;
                if (j < C::MAIN) {
                    const int e = j & 3, n = (j >> 2) % C::BN, k = 4 * ((j >> 2) / C::BN) + e;
                    v = value(k, n);
                } else if (j < C::MAIN + C::REM) {
                    const int jj = j - C::MAIN;
                    v = value(16 * C::NC + jj / C::BN, jj % C::BN);
                }
                img[j] = v;
            }
        }
    } else if (i < 2 * T + ng) {
        const int ci = i - 2 * T;
        if constexpr (SF) {
            if (a.gru_fmt[l] == kSplitF16x2)         // (block-uniform)
                gru_fwd_image_pack_split<D, kSplitF16x2>(a.Wg[l], a.Wc[l], a.nx[l], ci, a.gru_img[l] + (size_t)ci * ImgCfg<D, true, kSplitF16x2>::IMG, first, stride);
            else
                gru_fwd_image_pack_split<D, kSplitBf16x3>(a.Wg[l], a.Wc[l], a.nx[l], ci, a.gru_img[l] + (size_t)ci * ImgCfg<D, true>::IMG, first, stride);
        } else gru_fwd_image_pack<D>(a.Wg[l], a.Wc[l], a.nx[l], ci, a.gru_img[l] + (size_t)ci * ImgCfg<D, false>::IMG, first, stride);
    } else if (i < 2 * T + 2 * ng) {
        const int bi = i - 2 * T - ng;
        float* img = a.gru_bwd_img[l] + (size_t)bi * ImgCfg<D, SF>::IMG;
        if constexpr (SF) gru_bwd_image_pack_split<D>(a.Wg[l], a.Wc[l], a.nx[l], bi, img, first, stride);
        else gru_bwd_image_pack<D>(a.Wg[l], a.Wc[l], a.nx[l], bi, img, first, stride);
    }
}

// GGNN_TRAIN_MERGE_PRODUCTS=0: the GRU weight-gradient products timestep by timestep also for layers without residual inputs
bool merge_products() {
    static const bool v = [] { const char* e = getenv("GGNN_TRAIN_MERGE_PRODUCTS"); return !e || atoi(e) != 0; }();
    return v;
}

struct LayerPlan { int first_step, steps, nres; int res[kMaxNx]; };

int plan_layers(int num_layers, const int32_t* layer_timesteps, const int32_t* res_ptr, const int32_t* res_idx, LayerPlan* plan,
                int* total_steps) {
    GGNN_CHECK_ARG(num_layers > 0 && num_layers <= kMaxLayers && layer_timesteps && res_ptr, "bad layer description");
    int steps = 0;
    for (int l = 0; l < num_layers; ++l) {
        LayerPlan& P = plan[l];
        P.first_step = steps; P.steps = layer_timesteps[l];
        GGNN_CHECK_ARG(P.steps >= 1, "layer %d has %d timesteps: the native training step needs at least one per layer", l, P.steps);
        P.nres = res_ptr[l + 1] - res_ptr[l];
        GGNN_CHECK_ARG(P.nres >= 0 && P.nres + 1 <= kMaxNx, "layer %d has %d residual inputs (the fused kernels take %d)", l, P.nres, kMaxNx - 1);
        for (int i = 0; i < P.nres; ++i) {
            P.res[i] = res_idx[res_ptr[l] + i];
            GGNN_CHECK_ARG(P.res[i] >= 0 && P.res[i] <= l, "layer %d: residual index %d refers to a later layer", l, P.res[i]);
        }
        steps += P.steps;
    }
    *total_steps = steps;
    return GGNN_OK;
}

}  // namespace
}  // namespace ggnn

using namespace ggnn;

extern "C" int ggnn_sparse_train_prepare_f32(int num_layers, int T, int D, const int32_t* nx, const float* const* edge_w, float keep_prob,
                                             const uint64_t* seeds, const float* const* Wg, const float* const* Wc,
                                             const int32_t* gru_fmt, float* const* edge_packed, float* const* edge_packed_t,
                                             float* const* gru_packed, float* const* gru_bwd_packed, ggnn_stream_t stream) {
    GGNN_CHECK_ARG(num_layers > 0 && num_layers <= kPrepMaxLayers, "num_layers %d outside 1..%d", num_layers, kPrepMaxLayers);
    GGNN_CHECK_ARG(T > 0 && T <= 64 && nx && edge_w && Wg && Wc && edge_packed && edge_packed_t && gru_packed && gru_bwd_packed, "bad arguments");
    GGNN_CHECK_ARG(keep_prob > 0.0f && keep_prob <= 1.0f && (keep_prob >= 1.0f || seeds), "keep_prob %g outside (0, 1] or seeds missing", (double)keep_prob);
    if (ggnn_gru_is_fused(D) != 1 || !ggnn_gru_bwd_is_fused(D) || !ggnn_msg_transform_compact_supported(D))
        return fail(GGNN_E_UNSUPPORTED, "no whole-block stage images for hidden size %d", D);
    PrepArgs a{};
    a.T = T; a.keep = keep_prob;
    int max_images = 0;
    for (int l = 0; l < num_layers; ++l) {
        GGNN_CHECK_ARG(nx[l] >= 1 && nx[l] <= kMaxNx, "layer %d: nx %d outside 1..%d", l, nx[l], kMaxNx);
        GGNN_CHECK_ARG(edge_w[l] && Wg[l] && Wc[l] && edge_packed[l] && edge_packed_t[l] && gru_packed[l] && gru_bwd_packed[l], "layer %d: null pointer", l);
        a.edge_w[l] = edge_w[l]; a.Wg[l] = Wg[l]; a.Wc[l] = Wc[l];
        a.edge_img[l] = edge_packed[l]; a.edge_img_t[l] = edge_packed_t[l]; a.gru_img[l] = gru_packed[l]; a.gru_bwd_img[l] = gru_bwd_packed[l];
        a.seed[l] = seeds ? seeds[l] : 0ULL; a.nx[l] = nx[l];
        a.gru_fmt[l] = gru_launch_fmt(gru_fmt ? gru_fmt[l] : kSplitBf16x3);
        const int images = 2 * T + 6 * (nx[l] + 1);
        if (images > max_images) max_images = images;
    }
    const dim3 grid(8, max_images, num_layers);
    hipStream_t st = (hipStream_t)stream;
    const bool sf = split_matrix_path();
#define GGNN_PREP(DD) if (sf) hipLaunchKernelGGL((train_prepare_kernel<DD, true>), grid, dim3(256), 0, st, a); \
                      else hipLaunchKernelGGL((train_prepare_kernel<DD, false>), grid, dim3(256), 0, st, a);
    switch (D) {
        case 100: GGNN_PREP(100) break;
        case 64: GGNN_PREP(64) break;
        default: GGNN_PREP(32) break;
    }
#undef GGNN_PREP
    GGNN_CHECK_HIP(hipGetLastError());
    return GGNN_OK;
}

extern "C" size_t ggnn_sparse_train_workspace_bytes(int V, int D, int T, int64_t compact_rows, int total_steps) {
    if (V < 0 || D <= 0 || T <= 0 || total_steps <= 0) return 0;
    return train_layout(V, D, T, compact_rows, total_steps).total;
}

extern "C" int ggnn_sparse_train_forward_f32(
        const float* h0, int V, int D, int T, const int32_t* row_ptr, const int32_t* gather_row_c, const int32_t* pair_node,
        const int64_t* type_row_off, const float* nin, int use_avg, int num_layers, const int32_t* layer_timesteps,
        const int32_t* res_ptr, const int32_t* res_idx, const float* const* edge_packed, const float* const* bg,
        const float* const* bc, const float* const* gru_packed, const int32_t* gru_fmt, int act, void* ws, size_t ws_bytes,
        int64_t* final_state_offset, ggnn_stream_t stream) {
    GGNN_CHECK_ARG(V > 0 && D > 0 && D % 4 == 0 && T > 0, "bad sizes V=%d D=%d T=%d", V, D, T);
    GGNN_CHECK_ARG(h0 && row_ptr && gather_row_c && pair_node && type_row_off && ws && final_state_offset, "null pointer");
    GGNN_CHECK_ARG(edge_packed && bg && bc && gru_packed, "weights missing");
    GGNN_CHECK_ARG((reinterpret_cast<size_t>(ws) & 255) == 0, "workspace must be 256-byte aligned");
    if (ggnn_gru_is_fused(D) != 1 || !ggnn_msg_transform_compact_supported(D))
        return fail(GGNN_E_UNSUPPORTED, "native training step: hidden size %d has no gather-fused GRU / compacted transform", D);
    LayerPlan plan[kMaxLayers];
    int steps = 0;
    if (int rc = plan_layers(num_layers, layer_timesteps, res_ptr, res_idx, plan, &steps)) return rc;
    const int64_t R = type_row_off[T];
    GGNN_CHECK_ARG(R > 0, "no messages in the batch");
    const TrainLayout L = train_layout(V, D, T, R, steps);
    if (ws_bytes < L.total) return fail(GGNN_E_WORKSPACE, "training workspace too small: %zu < %zu", ws_bytes, L.total);
    char* base = static_cast<char*>(ws);
    float* Hc = reinterpret_cast<float*>(base + L.Hc);
    int32_t* counters = reinterpret_cast<int32_t*>(base + L.counters);
    hipStream_t st = (hipStream_t)stream;
    GGNN_CHECK_HIP(hipMemsetAsync(counters, 0, (size_t)steps * sizeof(int32_t), st));
    auto buf = [&](size_t off, int step) { return reinterpret_cast<float*>(base + off + (size_t)step * L.vd); };
    const float* states[kMaxLayers + 1];
    states[0] = h0;
    // (the backward's merged products read the input state of every timestep from the state array: slot 0 holds h0)
    GGNN_CHECK_HIP(hipMemcpyAsync(buf(L.state, 0), h0, L.vd, hipMemcpyDeviceToDevice, st));
    const size_t edge_img_bytes = ggnn_msg_transform_compact_workspace_bytes(D, T);
    for (int l = 0; l < num_layers; ++l) {
        const LayerPlan& P = plan[l];
        const float* xs[kMaxNx];
        for (int i = 0; i < P.nres; ++i) xs[i] = states[P.res[i]];              // :140-145
        const int nx = P.nres + 1;
        const float* cur = states[l];                                           // :152
        GGNN_CHECK_ARG(edge_packed[l] && bg[l] && bc[l] && gru_packed[l], "layer %d: weights missing", l);
        for (int s = 0; s < P.steps; ++s) {                                     // :153
            const int k = P.first_step + s;
            if (int rc = ggnn_msg_transform_compact_f32(cur, nullptr, pair_node, type_row_off, Hc, const_cast<float*>(edge_packed[l]),
                                                        edge_img_bytes, V, D, T, GGNN_GRU_FMT_BF16X3, stream)) return rc;
            float* out = buf(L.state, k + 1);
            if (int rc = ggnn_gru_packed_gather_train_f32(xs, nx, cur, gru_packed[l], bg[l], bc[l], out, Hc, row_ptr, gather_row_c,
                                                          use_avg ? nin : nullptr, T, use_avg ? 1 : 0, buf(L.r, k), buf(L.u, k),
                                                          buf(L.c, k), buf(L.inc, k), V, D, act,
                                                          gru_fmt ? gru_fmt[l] : GGNN_GRU_FMT_BF16X3, counters + k, stream)) return rc;
            cur = out;
        }
        states[l + 1] = cur;
    }
    *final_state_offset = (int64_t)(reinterpret_cast<const char*>(states[num_layers]) - base);
    return GGNN_OK;
}

extern "C" int ggnn_sparse_train_backward_f32(
        const float* h0, int V, int D, int T, const int32_t* pair_node, const int64_t* type_row_off, const float* nin, int use_avg,
        int num_layers, const int32_t* layer_timesteps, const int32_t* res_ptr, const int32_t* res_idx,
        const int32_t* rows_rp, const int32_t* rows_gather, const int32_t* rows_heads,
        const int32_t* node_rp, const int32_t* node_order, const int32_t* node_heads, const int32_t* identity_rows,
        const float* const* edge_packed_t, const float* const* gru_bwd_packed, int act,
        float* const* g_edge, float* const* g_Wg, float* const* g_bg, float* const* g_Wc, float* const* g_bc,
        float* d_final, float* const* d_state_ws, void* ws, size_t ws_bytes, ggnn_stream_t stream, ggnn_stream_t side_stream) {
    GGNN_CHECK_ARG(V > 0 && D > 0 && D % 4 == 0 && T > 0 && T <= 64, "bad sizes V=%d D=%d T=%d", V, D, T);
    GGNN_CHECK_ARG(h0 && pair_node && type_row_off && rows_rp && rows_gather && node_rp && node_order && identity_rows && d_final && ws,
                   "null pointer");
    GGNN_CHECK_ARG(edge_packed_t && gru_bwd_packed && g_edge && g_Wg && g_bg && g_Wc && g_bc && d_state_ws, "weights / gradient buffers missing");
    GGNN_CHECK_ARG(!use_avg || nin, "nin is required for mean aggregation");
    GGNN_CHECK_ARG((reinterpret_cast<size_t>(ws) & 255) == 0, "workspace must be 256-byte aligned");
    if (!ggnn_gru_bwd_is_fused(D)) return fail(GGNN_E_UNSUPPORTED, "native training step: no fused GRU backward for hidden size %d", D);
    LayerPlan plan[kMaxLayers];
    int steps = 0;
    if (int rc = plan_layers(num_layers, layer_timesteps, res_ptr, res_idx, plan, &steps)) return rc;
    const int64_t R64 = type_row_off[T];
    GGNN_CHECK_ARG(R64 > 0 && R64 < (1LL << 31), "bad compact row count");
    const int R = (int)R64;
    const TrainLayout L = train_layout(V, D, T, R, steps);
    if (ws_bytes < L.total) return fail(GGNN_E_WORKSPACE, "training workspace too small: %zu < %zu", ws_bytes, L.total);
    char* base = static_cast<char*>(ws);
    hipStream_t st = (hipStream_t)stream, side = side_stream ? (hipStream_t)side_stream : (hipStream_t)stream;
    auto buf = [&](size_t off, int step) { return reinterpret_cast<float*>(base + off + (size_t)step * L.vd); };
    float* Z = reinterpret_cast<float*>(base + L.Z);
    void* xty_ws = base + L.xty;
    const size_t xty_ws_bytes = L.total - 256 - L.xty;
    const size_t edge_img_bytes = ggnn_msg_transform_compact_workspace_bytes(D, T);
    const long long nvd = (long long)V * D;
    int32_t row_off_v[2] = {0, V};
    int32_t row_off_t[65];
    for (int t = 0; t <= T; ++t) row_off_t[t] = (int32_t)type_row_off[t];

    // forward states by layer (the forward pass's buffers) and the gradients flowing into them
    const float* states[kMaxLayers + 1];
    states[0] = buf(L.state, 0);                          // (the forward's copy of h0)
    for (int l = 0; l < num_layers; ++l) states[l + 1] = buf(L.state, plan[l].first_step + plan[l].steps);
    (void)h0;
    float* dstate[kMaxLayers + 1];
    bool has[kMaxLayers + 1];
    for (int l = 0; l < num_layers; ++l) { dstate[l] = d_state_ws[l]; has[l] = false; GGNN_CHECK_ARG(dstate[l], "d_state_ws[%d] is null", l); }
    dstate[num_layers] = d_final; has[num_layers] = true;

    // the first side-stream product of the step is ordered behind everything queued on the main stream so far (the zeroed gradients)
    if (int rc = order_after(side, st)) return rc;

    // The per-node sum that closes a timestep's transform backward (dh[v] += sum_t Z[row(v,t)]) is taken by its CONSUMER: the fused
    // GRU backward of the timestep processed next reads g[v] + the rows of Z its node's slot-head record names
    // (ggnn_gru_bwd_fused_gather_f32) -- one launch and one pass over [V,D] less per timestep.  Needs the head records and at most
    // four rows per node (T <= 4: a node has one compact row per edge type it sends on); GGNN_TRAIN_FUSE_NODE_SUM=0: stand-alone sums.
    static const bool fuse_env = [] { const char* e = getenv("GGNN_TRAIN_FUSE_NODE_SUM"); return !e || atoi(e) != 0; }();
    const bool fuse_node_sum = fuse_env && node_heads != nullptr && T <= 4;
    bool z_pending = false;                         // Z holds rows that the next GRU backward still has to add to its g

    for (int l = num_layers - 1; l >= 0; --l) {
        const LayerPlan& P = plan[l];
        const int nx = P.nres + 1;
        GGNN_CHECK_ARG(has[l + 1], "no gradient reaches the output of layer %d", l);
        GGNN_CHECK_ARG(edge_packed_t[l] && gru_bwd_packed[l] && g_edge[l] && g_Wg[l] && g_bg[l] && g_Wc[l] && g_bc[l], "layer %d: pointers missing", l);
        const float* g = dstate[l + 1];
        for (int s = P.steps - 1; s >= 0; --s) {
            const int k = P.first_step + s;
            const float* h_in = buf(L.state, k);              // input state of timestep k (slot k; slot 0 = h0)
            // where this timestep's gradients go: straight into an empty accumulator, else into a temporary that is added afterwards
            float* dh_dst = buf(L.dh, k);
            bool dh_direct = false;
            if (s == 0 && !has[l]) { dh_dst = dstate[l]; dh_direct = true; }
            float* dxp[kMaxNx];
            bool dx_direct[kMaxNx];
            for (int i = 0; i < P.nres; ++i) {
                const int src = P.res[i];
                dx_direct[i] = !has[src] && !(s == 0 && src == l && dh_direct);
                // (two residual inputs of one layer may name the same state: only the first of them can write it directly)
                for (int j = 0; j < i; ++j) if (P.res[j] == src) dx_direct[i] = false;
                dxp[i] = dx_direct[i] ? dstate[src] : reinterpret_cast<float*>(base + L.dx + ((size_t)k * kMaxNx + i) * L.vd);
            }
            float* dinc = reinterpret_cast<float*>(base + L.dx + ((size_t)k * kMaxNx + (nx - 1)) * L.vd);
            dxp[nx - 1] = dinc;
            float* dpc = buf(L.dpc, k); float* rh = buf(L.rh, k);
            float* dpg = reinterpret_cast<float*>(base + L.dpg + (size_t)k * 2 * L.vd);
            if (z_pending) {
                if (int rc = ggnn_gru_bwd_fused_gather_f32(g, Z, node_heads, h_in, buf(L.r, k), buf(L.u, k), buf(L.c, k),
                                                           const_cast<float*>(gru_bwd_packed[l]), dpc, dpg, rh, dh_dst, dxp, nin, T,
                                                           use_avg ? 1 : 0, nx, V, D, act, stream)) return rc;
                z_pending = false;
            } else {
                if (int rc = ggnn_gru_bwd_fused_f32(g, h_in, buf(L.r, k), buf(L.u, k), buf(L.c, k), nullptr, nullptr,
                                                    const_cast<float*>(gru_bwd_packed[l]), dpc, dpg, rh, dh_dst, dxp, nin, T, use_avg ? 1 : 0,
                                                    nx, V, D, act, stream)) return rc;
            }
            for (int i = 0; i < P.nres; ++i) if (dx_direct[i]) has[P.res[i]] = true;
            if (dh_direct) has[l] = true;

            // ---- GRU weight gradients, side stream: dWc += [x.. | incoming | r*h]^T dpc, dWg += [x.. | incoming | h]^T dpg
            // A layer WITHOUT residual inputs runs them once for all its timesteps: incoming, r*h, dpc, dpg and the input states of
            // consecutive timesteps are consecutive [V,D] ([V,2D]) buffers, i.e. one matrix of S*V rows each -- one product
            // instead of S, and its fixed part (prologue, partial store, reduction launch: ~20 us of ~100) paid once.
            const bool merged = P.nres == 0 && P.steps > 1 && merge_products();
            if (merged) {
                if (s == 0) {                               // (the layer's first timestep is the last one processed)
                    if (int rc = order_after(side, st)) return rc;
                    const int k0 = P.first_step;
                    int32_t row_off_m[2] = {0, V * P.steps};
                    const float* X[2] = {buf(L.inc, k0), buf(L.rh, k0)}; int32_t ldx[2] = {D, D};
                    if (int rc = ggnn_xty_acc_f32(X, 2, D, ldx, nullptr, buf(L.dpc, k0), D, g_Wc[l], g_bc[l], 1, 2 * D, D, 1, row_off_m, 1,
                                                  xty_ws, xty_ws_bytes, side)) return rc;
                    X[1] = buf(L.state, k0);
                    if (int rc = ggnn_xty_acc_f32(X, 2, D, ldx, nullptr, reinterpret_cast<float*>(base + L.dpg + (size_t)k0 * 2 * L.vd), 2 * D,
                                                  g_Wg[l], g_bg[l], 1, 2 * D, 2 * D, 1, row_off_m, 1, xty_ws, xty_ws_bytes, side)) return rc;
                }
            } else {
                if (int rc = order_after(side, st)) return rc;
                const float* X[4]; int32_t ldx[4];
                for (int i = 0; i < P.nres; ++i) { X[i] = states[P.res[i]]; ldx[i] = D; }
                X[nx - 1] = buf(L.inc, k); ldx[nx - 1] = D;
                X[nx] = rh; ldx[nx] = D;
                if (int rc = ggnn_xty_acc_f32(X, nx + 1, D, ldx, nullptr, dpc, D, g_Wc[l], g_bc[l], 1, (nx + 1) * D, D, 1, row_off_v, 1,
                                              xty_ws, xty_ws_bytes, side)) return rc;
                X[nx] = h_in;
                if (int rc = ggnn_xty_acc_f32(X, nx + 1, D, ldx, nullptr, dpg, 2 * D, g_Wg[l], g_bg[l], 1, (nx + 1) * D, 2 * D, 1, row_off_v, 1,
                                              xty_ws, xty_ws_bytes, side)) return rc;
            }

            // ---- back through the segment sum and the compacted transform (main stream)
            float* dHc = reinterpret_cast<float*>(base + L.dHc + (size_t)k * L.rd);
            if (rows_heads) {
                if (int rc = ggnn_gather_segment_sum_heads_f32(dinc, rows_rp, rows_gather, rows_heads, nullptr, nullptr, 0, dHc, R, D, 1, 0, stream)) return rc;
            } else {
                if (int rc = ggnn_gather_segment_sum_f32(dinc, rows_rp, rows_gather, nullptr, nullptr, 0, dHc, R, D, 1, stream)) return rc;
            }
            // The gradient of the step's INPUT state: not needed for the very first timestep (h0 is data: nothing upstream of it is
            // trained, and nobody reads d_state_ws[0]) -- its transform and node sum are not run at all.
            const bool first_step = l == 0 && s == 0;
            if (!first_step) {
                if (int rc = ggnn_msg_transform_compact_f32(dHc, nullptr, identity_rows, type_row_off, Z, const_cast<float*>(edge_packed_t[l]),
                                                            edge_img_bytes, R, D, T, GGNN_GRU_FMT_BF16X3, stream)) return rc;
                if (fuse_node_sum) {
                    z_pending = true;                   // (the next timestep's GRU backward adds the rows while it loads g)
                } else if (node_heads) {
                    if (int rc = ggnn_gather_segment_sum_heads_f32(Z, node_rp, node_order, node_heads, nullptr, nullptr, 0, dh_dst, V, D, 1, 1, stream)) return rc;
                } else {
                    if (int rc = ggnn_gather_segment_sum_acc_f32(Z, node_rp, node_order, dh_dst, V, D, stream)) return rc;
                }
            }
            // ---- edge-weight gradients, side stream (needs dHc: ordered behind the transform launch above, which also read it)
            if (int rc = order_after(side, st)) return rc;
            {
                const float* X[1] = {h_in}; int32_t ldx[1] = {D};
                if (int rc = ggnn_xty_acc_f32(X, 1, D, ldx, pair_node, dHc, D, g_edge[l], nullptr, 1, D, D, 0, row_off_t, T, xty_ws, xty_ws_bytes,
                                              side)) return rc;
            }
            // ---- accumulate what went to temporaries
            for (int i = 0; i < P.nres; ++i) {          // (sums into the gradient of h0 are not needed either)
                if (!dx_direct[i] && P.res[i] != 0) { if (int rc = add_inplace(dstate[P.res[i]], dxp[i], nvd, st)) return rc; }
            }
            if (s == 0 && !dh_direct && l != 0) { if (int rc = add_inplace(dstate[l], dh_dst, nvd, st)) return rc; }
            g = dh_dst;
        }
    }
    // every deferred node sum was consumed by the GRU backward of the timestep before it (the only step that defers without a
    // consumer would be the very first one, which runs no transform): a pending one here means a gradient was dropped
    GGNN_CHECK_ARG(!z_pending, "internal: a deferred node sum of the transform backward was never added (fuse_node_sum invariant)");
    return order_after(st, side);          // the caller's next launches (optimizer, next forward) see every product
}
