// K1 compacted: the per-edge-type message transform (chem_tensorflow_sparse.py:160-164) evaluated only for
// the (source node, edge type) pairs that actually emit a message.
//
// The reference transforms every MESSAGE (M rows: h[src] W_t, :161-164); transform-first over all nodes
// costs T*V rows.  Both over-count: a node with k outgoing edges of type t sends the SAME row h[v] W_t k
// times, and most (node, type) pairs have no outgoing edge at all (at QM9 shapes ~1.2*V pairs are active
// of 4*V; M ~ 2*V).  Here the active pairs are enumerated once per batch (type-major, node-ascending:
// compact row id), the transform runs on exactly those rows -- a row-gathered [R_t, D] x [D, D] FP32-MFMA
// GEMM per type, all types in one launch -- and the segment-sum gathers compact rows.
// Same arithmetic per message as the dense form (identical fmaf chains), ~3.3x fewer flops, and the
// transformed-state buffer shrinks from V*T*D to R*D floats (fits the 256 MiB Infinity Cache).
#include "ggnn_split.hpp"
#include <cstring>
#include <rocprim/device/device_scan.hpp>

namespace ggnn {

constexpr int kMaxTypesC = 64;
struct TypeRows { int row_off[kMaxTypesC + 1]; int tile_off[kMaxTypesC + 1]; int T; int num_nodes; unsigned long long* tdbg; };

static inline size_t align256c(size_t x) { return (x + 255) / 256 * 256; }

// ---- index prep ---------------------------------------------------------------------------------------
// flags[t*V + v] = (node v has an outgoing edge of type t)
__global__ void compact_flags_kernel(const int* __restrict__ src_row_ptr, int V, int T, int* flags) {
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= (long long)V * T) return;
    const int t = (int)(p / V), v = (int)(p - (long long)t * V);
    const int s = v * T + t;
    flags[p] = src_row_ptr[s + 1] > src_row_ptr[s] ? 1 : 0;
}

__global__ void compact_fill_kernel(const int* __restrict__ flags, const int* __restrict__ scan, int V, int T,
                                    int* pair_node, int* pair_id, int* type_row_off) {
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long n = (long long)V * T;
    if (p >= n) return;
    const int t = (int)(p / V), v = (int)(p - (long long)t * V);
    const int f = flags[p], c = scan[p];
    pair_id[v * T + t] = f ? c : -1;
    if (f) pair_node[c] = v;
    if (v == 0) type_row_off[t] = c;
    if (p == n - 1) type_row_off[T] = c + f;
}

__global__ void remap_rows_kernel(const int* __restrict__ gather_row, const int* __restrict__ pair_id, int* out, long long M) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < M) out[i] = pair_id[gather_row[i]];
}

static size_t scan_temp_bytes(long long n) {
    size_t bytes = 0;
    (void)rocprim::exclusive_scan(nullptr, bytes, (const int*)nullptr, (int*)nullptr, 0, (size_t)n, rocprim::plus<int>(), (hipStream_t)0);
    return bytes;
}

// ---- weights -> stage images ------------------------------------------------------------------------------
template <int D, bool SPLIT, int FMT = kSplitBf16x3>
__global__ void edge_weight_pack_kernel(const float* __restrict__ W, float* __restrict__ out) {
    const int t = blockIdx.y;
    float* img = out + (size_t)t * ImgCfg<D, SPLIT, FMT>::IMG;
    const int first = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
    if constexpr (SPLIT) pack_split_image<D, FMT>(StageValue<D>{W + (size_t)t * D * D, 0, 0, D, -1, nullptr, 0, 0, -1}, img, first, stride);
    else pack_stage_image<D>(W + (size_t)t * D * D, 0, 0, D, img, first, stride);
}

// ---- the transform: persistent workgroups, each bound to ONE edge type ---------------------------------------
// tr.tile_off[t] .. tile_off[t+1] are the workgroups of type t.  A workgroup brings its type's weight image into LDS
// once (LDS-DMA) and then every wave walks, independently of the others (no barrier after the first), over its share
// of the type's 16-row wave tiles: j-th workgroup of the type, wave w -> tiles (j*NW + w) + k * (NW * workgroups of
// the type).  The rows of tile k+1 (pair list -> gathered state rows) are fetched while tile k is multiplied, the
// pair-list entry of tile k+2 with them, so no wave waits on a dependent load chain inside the loop.
// Two workgroups fit a CU (LDS 2 x 48 KiB, <= 128 VGPRs): 4 waves per SIMD keep the matrix pipe fed while others
// store and fetch.  The host sizes the per-type workgroup counts so that every wave gets the same number of tiles
// (+-1) across ALL types (launch_compact).
// SPLIT: the product on the bf16 matrix pipe in 3-way split form (ggnn_split.hpp): the type's image is the 72 KiB split one, a
// tile's rows are split in registers before its MFMAs.
// FMT (SPLIT; round 5): operand format of the products and of the images, per launch -- the exact kSplitBf16x3 (default; the backward's
// Z = dHc W^T multiplies gradients of any magnitude and always runs in it) or kSplitF16x2 (two f16 pieces, three products, 48 KiB
// images) for a forward transform whose operands the caller has PROVEN inside that format's range (|h| <= 65504, |W| <= 255.875:
// formats.py, the same proof as the fused GRU's).
template <int D, int NW, bool SPLIT, int FMT = kSplitBf16x3>
__global__ __launch_bounds__(NW * 64, 4) /* 4 waves per SIMD = 2 workgroups per CU */ void msg_transform_compact_kernel(
        const float* __restrict__ h, const int* __restrict__ pair_node, TypeRows tr, const float* __restrict__ packed, float* __restrict__ Hc) {
    using C = StageCfg<D>;
    using I = ImgCfg<D, SPLIT, FMT>;
    constexpr int NT = C::NT;
    extern __shared__ __attribute__((aligned(16))) float img[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kq = lane >> 4;

#define K1C_T(K) if (tr.tdbg && lane == 0) tr.tdbg[((size_t)blockIdx.x * NW + wave) * 8 + (K)] = (K) == 0 || (K) == 7 ? __builtin_amdgcn_s_memrealtime() : __builtin_amdgcn_s_memtime();
    K1C_T(0)
    K1C_T(6)                  // (shader-clock twin of the start stamp; a wave has at most 3 tiles in these runs)
    int t = 0;
    while (t + 1 < tr.T && (int)blockIdx.x >= tr.tile_off[t + 1]) ++t;
    const int row_beg = tr.row_off[t], row_end = tr.row_off[t + 1];
    const int n_wt = (row_end - row_beg + 15) / 16;                     // wave tiles of this type
    const int stride = (tr.tile_off[t + 1] - tr.tile_off[t]) * NW;
    int idx = ((int)blockIdx.x - tr.tile_off[t]) * NW + wave;           // wave-uniform

    auto row_of = [&](int i) { const int r = row_beg + i * 16 + li; return r < row_end ? r : row_end - 1; };
    Frag<D> a, an;
    // the pair-list entries first (the head of the dependent chain pair -> rows), then the weight image, then the rows:
    // memory instructions return in order, so the chain is not queued behind 6 KiB of LDS-DMA per wave
    int node_0 = 0, node_n = 0;
    if (idx < n_wt) node_0 = pair_node[row_of(idx)];
    if (idx + stride < n_wt) node_n = pair_node[row_of(idx + stride)];
    if constexpr (SPLIT) dma_image_asm<I::IMG_BYTES, NW>(packed + (size_t)t * I::IMG, img, wave, lane);
    else dma_stage_image<D, NW>(packed + (size_t)t * C::IMG, img, wave, lane);
    if (idx < n_wt) load_frag<D>(a, h, node_0, kq);
    K1C_T(1)
    if constexpr (SPLIT) dma_wait();
    __syncthreads();                                        // the image has landed (vmcnt(0) + barrier)
    K1C_T(2)
    int tcount = 0;

    while (idx < n_wt) {
        const int idx_n = idx + stride;
        if (idx_n < n_wt) load_frag<D>(an, h, node_n, kq);             // rows of the next tile
        if (idx_n + stride < n_wt) node_n = pair_node[row_of(idx_n + stride)];
        f32x4 acc[NT];
        __builtin_amdgcn_sched_barrier(0);                  // the fetches are issued BEFORE the MFMA block
        if constexpr (SPLIT) {
            SFrag<D> sf;
            split_frag<D, FMT>(sf, a);
            stage_mma_split<D, NT, true, false, FMT>(acc, sf, a, img, li, kq);
        } else {
            stage_mma<D, NoHook, NT, true>(acc, a, img, li, kq);    // (first MFMA of each tile starts from C = 0)
            stage_tail_reduce<D>(acc);
        }
        const int r = row_beg + idx * 16 + li;
        if (r < row_end) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int col = nt * 16 + 4 * kq;
                if (col < D) {
                    if constexpr (SPLIT && SplitFmt<FMT>::acc_scale != 1.0f) st4_b(Hc, ((unsigned)r * (unsigned)D + col) * 4u, acc[nt] * SplitFmt<FMT>::acc_scale);
                    else st4_b(Hc, ((unsigned)r * (unsigned)D + col) * 4u, acc[nt]);
                }
            }
        }
        a = an;
        idx = idx_n;
        if (tcount < 3) { K1C_T(3 + tcount) }
        ++tcount;
    }
    K1C_T(7)
}

template <int D, bool SPLIT, int FMT = kSplitBf16x3>
static int launch_compact_m(const float* h, const float* W, const int* pair_node, TypeRows& tr, float* packed, float* Hc,
                            hipStream_t st) {
    // waves per workgroup: 8, two workgroups per CU.  (One 16-wave workgroup per CU halves the image DMA and evens out
    // the prologues -- the second workgroup of a CU otherwise starts 5 us late behind the first one's MFMA bursts --
    // but measures the same alone (32.8 us) and 2 % slower with two streams: 608 vs 621 M node-updates/s.)
    constexpr int NW = 8;
    using C = ImgCfg<D, SPLIT, FMT>;
    if (W) {      // raw [T,D,D] weights given: build the T stage images (skipped when the caller pre-packed them)
        hipLaunchKernelGGL((edge_weight_pack_kernel<D, SPLIT, FMT>), dim3(8, tr.T), dim3(256), 0, st, W, packed);
        GGNN_CHECK_HIP(hipGetLastError());
    }
    if (tr.row_off[tr.T] == 0 || h == nullptr) return GGNN_OK;
    const int V = tr.num_nodes;
    if ((unsigned long long)tr.row_off[tr.T] * D >= (1ULL << 30) || (unsigned long long)V * D >= (1ULL << 30))
        return fail(GGNN_E_UNSUPPORTED, "compacted transform indexes with 32-bit byte offsets: rows*D and V*D must be < 2^30");
    // Workgroups per type: the smallest number of rounds R (wave tiles per wave) for which all types together fit
    // two workgroups per CU, then ceil(wave tiles of the type / (R * NW)) workgroups for each type.
    long long total_wt = 0;
    for (int t = 0; t < tr.T; ++t) total_wt += (tr.row_off[t + 1] - tr.row_off[t] + 15) / 16;
    // workgroups per CU: 2 (16 waves; both forms stay below 128 registers and 2 x 72 KiB of images fit the LDS); GGNN_K1_WG_PER_CU=1 to compare
    static const int wg_per_cu = [] { const char* e = getenv("GGNN_K1_WG_PER_CU"); return e && atoi(e) == 1 ? 1 : 2; }();
    const long long budget = (long long)wg_per_cu * num_cus();
    long long R = total_wt / (budget * NW);
    if (R < 1) R = 1;
    for (;; ++R) {
        long long blocks = 0;
        for (int t = 0; t < tr.T; ++t) blocks += ((tr.row_off[t + 1] - tr.row_off[t] + 15) / 16 + R * NW - 1) / (R * NW);
        if (blocks <= budget) break;
    }
    tr.tile_off[0] = 0;
    for (int t = 0; t < tr.T; ++t)
        tr.tile_off[t + 1] = tr.tile_off[t] + (int)(((tr.row_off[t + 1] - tr.row_off[t] + 15) / 16 + R * NW - 1) / (R * NW));
    static std::atomic<unsigned long long> lds_ok{0};
    if (C::IMG_BYTES > 48 * 1024) GGNN_CHECK_HIP(allow_dynamic_lds(&msg_transform_compact_kernel<D, NW, SPLIT, FMT>, C::IMG_BYTES, lds_ok));
    { const char* e = getenv("GGNN_K1C_TPTR"); tr.tdbg = e ? (unsigned long long*)strtoull(e, nullptr, 10) : nullptr; }
    hipLaunchKernelGGL((msg_transform_compact_kernel<D, NW, SPLIT, FMT>), dim3(tr.tile_off[tr.T]), dim3(NW * 64), C::IMG_BYTES, st, h, pair_node,
                       tr, (const float*)packed, Hc);
    GGNN_CHECK_HIP(hipGetLastError());
    return GGNN_OK;
}

template <int D>
static int launch_compact(const float* h, const float* W, const int* pair_node, TypeRows& tr, float* packed, float* Hc, int fmt,
                          hipStream_t st) {
    if (SplitCfg<D>::OK && split_matrix_path())
        return gru_launch_fmt(fmt) == kSplitF16x2 ? launch_compact_m<D, true, kSplitF16x2>(h, W, pair_node, tr, packed, Hc, st)
                                                  : launch_compact_m<D, true>(h, W, pair_node, tr, packed, Hc, st);
    return launch_compact_m<D, false>(h, W, pair_node, tr, packed, Hc, st);
}

int gru_panel_supported(int D);      // ggnn_panel.hip: hidden sizes handled on column panels
int transform_panel_image_floats(int D);
int transform_panel_dispatch(const float* h, const float* W, const int* pair_node, const int* row_off, int T, int V, int D,
                             float* packed, float* Hc, int fmt, hipStream_t st);

static int stage_img_floats(int D) {
    if (gru_panel_supported(D)) return (D / 64) * transform_panel_image_floats(D);      // NP panel images per type
    const bool sp = split_matrix_path();
    switch (D) {
        case 100: return sp ? ImgCfg<100, true>::IMG : ImgCfg<100, false>::IMG;
        case 64: return sp ? ImgCfg<64, true>::IMG : ImgCfg<64, false>::IMG;
        case 32: return sp ? ImgCfg<32, true>::IMG : ImgCfg<32, false>::IMG;
        default: return 0;
    }
}

}  // namespace ggnn

using namespace ggnn;

extern "C" int ggnn_msg_transform_compact_supported(int D) { return stage_img_floats(D) > 0; }

extern "C" size_t ggnn_compact_workspace_bytes(int V, int T) {
    if (V <= 0 || T <= 0) return 256;
    const size_t n = (size_t)V * T;
    return 2 * align256c(n * sizeof(int)) + align256c(scan_temp_bytes((long long)n)) + 256;
}

extern "C" int ggnn_build_compact_sources(const int32_t* src_row_ptr, int V, int T, int32_t* pair_node, int32_t* pair_id,
                                          int32_t* type_row_off, void* ws, size_t ws_bytes, ggnn_stream_t stream) {
    GGNN_CHECK_ARG(V >= 0 && T > 0 && T <= kMaxTypesC && (long long)V * T < (1LL << 31), "bad sizes V=%d T=%d", V, T);
    GGNN_CHECK_ARG(type_row_off, "null pointer");
    hipStream_t st = (hipStream_t)stream;
    if (V == 0) {
        GGNN_CHECK_HIP(hipMemsetAsync(type_row_off, 0, sizeof(int) * (T + 1), st));
        return GGNN_OK;
    }
    GGNN_CHECK_ARG(src_row_ptr && pair_node && pair_id && ws, "null pointer");
    if (ws_bytes < ggnn_compact_workspace_bytes(V, T))
        return fail(GGNN_E_WORKSPACE, "compact workspace too small: %zu < %zu", ws_bytes, ggnn_compact_workspace_bytes(V, T));
    const long long n = (long long)V * T;
    const size_t arr = align256c((size_t)n * sizeof(int));
    char* p = reinterpret_cast<char*>(align256c(reinterpret_cast<size_t>(ws)));
    int* flags = reinterpret_cast<int*>(p);
    int* scan = reinterpret_cast<int*>(p + arr);
    void* tmp = p + 2 * arr;
    size_t tmp_bytes = scan_temp_bytes(n);
    const unsigned blocks = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(compact_flags_kernel, dim3(blocks), dim3(256), 0, st, src_row_ptr, V, T, flags);
    GGNN_CHECK_HIP(hipGetLastError());
    GGNN_CHECK_HIP(rocprim::exclusive_scan(tmp, tmp_bytes, (const int*)flags, scan, 0, (size_t)n, rocprim::plus<int>(), st));
    hipLaunchKernelGGL(compact_fill_kernel, dim3(blocks), dim3(256), 0, st, (const int*)flags, (const int*)scan, V, T, pair_node,
                       pair_id, type_row_off);
    GGNN_CHECK_HIP(hipGetLastError());
    return GGNN_OK;
}

extern "C" int ggnn_remap_gather_rows(const int32_t* gather_row, const int32_t* pair_id, int32_t* gather_row_compact, int64_t M,
                                      ggnn_stream_t stream) {
    GGNN_CHECK_ARG(M >= 0, "negative M");
    if (M == 0) return GGNN_OK;
    GGNN_CHECK_ARG(gather_row && pair_id && gather_row_compact, "null pointer");
    hipLaunchKernelGGL(remap_rows_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, (hipStream_t)stream, gather_row, pair_id,
                       gather_row_compact, (long long)M);
    GGNN_CHECK_HIP(hipGetLastError());
    return GGNN_OK;
}

extern "C" size_t ggnn_msg_transform_compact_workspace_bytes(int D, int T) {
    return (size_t)stage_img_floats(D) * sizeof(float) * (size_t)(T > 0 ? T : 0) + 256;
}

extern "C" int ggnn_edge_weights_pack_f32(const float* W, int T, int D, int fmt, float* packed, ggnn_stream_t stream) {
    GGNN_CHECK_ARG(T > 0 && T <= kMaxTypesC, "bad T=%d", T);
    GGNN_CHECK_ARG(fmt == 0 || fmt == GGNN_GRU_FMT_F16X2 || fmt == GGNN_GRU_FMT_BF16X3, "fmt %d is not a GGNN_GRU_FMT_* value", fmt);
    if (!ggnn_msg_transform_compact_supported(D))
        return fail(GGNN_E_UNSUPPORTED, "no packed edge weights for hidden size %d", D);
    GGNN_CHECK_ARG(W && packed && aligned16(packed), "null or misaligned pointer");
    TypeRows tr{};
    tr.T = T;                                   // all row counts zero: pack only
    hipStream_t st = (hipStream_t)stream;
    if (gru_panel_supported(D)) return transform_panel_dispatch(nullptr, W, nullptr, tr.row_off, T, 0, D, packed, nullptr, fmt, st);
    switch (D) {
        case 100: return launch_compact<100>(nullptr, W, nullptr, tr, packed, nullptr, fmt, st);
        case 64: return launch_compact<64>(nullptr, W, nullptr, tr, packed, nullptr, fmt, st);
        default: return launch_compact<32>(nullptr, W, nullptr, tr, packed, nullptr, fmt, st);
    }
}

// W == NULL: `ws` already holds the images written by ggnn_edge_weights_pack_f32 (inference: weights are
// constant across batches, so the pack pre-pass is paid once per weight version instead of once per call).
extern "C" int ggnn_msg_transform_compact_f32(const float* h, const float* W, const int32_t* pair_node,
                                              const int64_t* type_row_off, float* Hc, void* ws, size_t ws_bytes, int V, int D,
                                              int T, int fmt, ggnn_stream_t stream) {
    GGNN_CHECK_ARG(V >= 0 && D > 0 && D % 4 == 0 && T > 0 && T <= kMaxTypesC, "bad sizes V=%d D=%d T=%d", V, D, T);
    GGNN_CHECK_ARG(fmt == 0 || fmt == GGNN_GRU_FMT_F16X2 || fmt == GGNN_GRU_FMT_BF16X3, "fmt %d is not a GGNN_GRU_FMT_* value", fmt);
    GGNN_CHECK_ARG(type_row_off, "null pointer");
    if (!ggnn_msg_transform_compact_supported(D))
        return fail(GGNN_E_UNSUPPORTED, "compacted message transform supports hidden sizes 32, 64, 100, 128, 192, 256 (got %d)", D);
    TypeRows tr;
    tr.T = T;
    tr.num_nodes = V;
    tr.tile_off[0] = 0;
    GGNN_CHECK_ARG(type_row_off[0] == 0, "type_row_off must start at 0");
    for (int t = 0; t < T; ++t) {
        GGNN_CHECK_ARG(type_row_off[t] <= type_row_off[t + 1] && type_row_off[t + 1] < (1LL << 31), "type_row_off not monotone");
        tr.row_off[t] = (int)type_row_off[t];
        tr.tile_off[t + 1] = tr.tile_off[t] + (int)((type_row_off[t + 1] - type_row_off[t] + 127) / 128);
    }
    tr.row_off[T] = (int)type_row_off[T];
    if (tr.row_off[T] == 0) return GGNN_OK;
    GGNN_CHECK_ARG(h && pair_node && Hc && ws, "null pointer");
    GGNN_CHECK_ARG(aligned16(h) && (!W || aligned16(W)) && aligned16(Hc) && aligned16(ws), "pointers must be 16-byte aligned");
    if (ws_bytes < ggnn_msg_transform_compact_workspace_bytes(D, T))
        return fail(GGNN_E_WORKSPACE, "compact transform workspace too small");
    hipStream_t st = (hipStream_t)stream;
    float* packed = static_cast<float*>(ws);
    if (gru_panel_supported(D)) return transform_panel_dispatch(h, W, pair_node, tr.row_off, T, V, D, packed, Hc, fmt, st);
    switch (D) {
        case 100: return launch_compact<100>(h, W, pair_node, tr, packed, Hc, fmt, st);
        case 64: return launch_compact<64>(h, W, pair_node, tr, packed, Hc, fmt, st);
        default: return launch_compact<32>(h, W, pair_node, tr, packed, Hc, fmt, st);
    }
}
