// Graph-resident dense GGNN forward (chem_tensorflow_dense.py:93-117): ALL timesteps of a graph in one workgroup, one launch.
//
// The dense model's batch is b graphs of v <= 32 padded vertices (BASELINE configs[2]: b = 256, v = 29, h = 100, 4 edge types,
// 4 timesteps).  7,424 rows are 464 row tiles for 256 CUs: the per-timestep kernels (h W_e for all e, adjacency aggregation,
// fused GRU -- 12 launches) are each latency-bound at 0.2 of their rooflines.  But a graph only ever reads its OWN vertices
// (A[g] is v x v), so nothing has to leave the CU between timesteps: workgroup g keeps the states of graph g in registers and
// runs, per timestep, E transform stages, the aggregation and the six GRU stages back to back.
//
//   * wave w (w < NT = ceil(h/16)) owns output COLUMN tile w of every stage, for both 16-row tiles of the graph: 25 MFMAs per row
//     tile and stage (h = 100).  Its slice of a stage's weights -- one column tile of the k-interleaved stage image, 25 floats per
//     lane -- comes straight from L2 into registers one stage ahead (tile_mma_regs): no LDS ring, no barrier per stage.
//   * every wave holds the full activation fragments (the MFMA's B operand: state h, then acts / r*h) of both row tiles.  What a
//     stage produces per column tile is exchanged through LDS: the transformed states M_e (all E of them: the aggregation reads
//     rows of OTHER vertices), acts, r*h and the new state -- four workgroup barriers per timestep.
//   * aggregation acts[i] = sum_e sum_j A_e[i,j] (M_e[j] + b_e) (:103-112; bias on every row before A_e, :107-108) runs on the matrix
//     pipe too: the graph's 0/1 adjacency rows sit in LDS for the whole launch, the products A_e[i,j] M_e[j] are exact, and the bias
//     term is nin_e[i] b_e (the row sums of A_e, formed once).
//   * GRU (TF-1.3 GRUCell, :115): r, u = sigmoid([acts|h] Wg + bg), c = tanh([acts | r*h] Wc + bc), h' = u h + (1-u) c; the r*h tile of
//     a wave is r times chunk w of its h fragment (output tile nt == activation chunk nt, as in the fused GRU).
#include "ggnn_stage.hpp"
#include "ggnn_dense_graph.hpp"
#include "ggnn_split.hpp"

namespace ggnn {

template <int D>
__global__ void dense_gru_pack_kernel(const float* __restrict__ Wg, const float* __restrict__ Wc, float* __restrict__ out) {
    const int i = blockIdx.y;                                          // image 0..5
    const int first = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
    float* img = out + (size_t)i * StageCfg<D>::IMG;
    if (i < 4) pack_stage_image<D>(Wg, (i & 1) * D, (i >> 1) * D, 2 * D, img, first, stride);     // rows: x | h;  columns: r | u
    else pack_stage_image<D>(Wc, (i & 1) * D, 0, D, img, first, stride);
}

template <int D, int E, int NW>
__global__ __launch_bounds__(NW * 64) void ggnn_dense_graph_kernel(DenseGraphArgs a) {
    using C = StageCfg<D>;
    constexpr int NT = C::NT, NC = C::NC, NR = C::NR;
    constexpr int MP = C::BN + 4;                                      // row pitch of the LDS blocks (floats)
    constexpr int NS = E + 6;                                          // stages per timestep
    constexpr int AP = 33;                                             // pitch of an adjacency row in LDS (16 rows x one column: 16 banks)
    static_assert(NT <= NW && NS % 2 == 0, "one column tile per wave; the two weight slots alternate with a fixed phase per timestep");
    extern __shared__ __attribute__((aligned(16))) float lds[];        // Mbuf [E][32][MP] | Xbuf, Rbuf, Hbuf [32][MP] | Abuf [E][32][AP] | ...
    float* Mbuf = lds;
    float* Xbuf = lds + (size_t)E * 32 * MP;                           // exchange blocks: acts, r*h, the new state -- one each, so that a
    float* Rbuf = Xbuf + (size_t)32 * MP;                              // block is rewritten a whole timestep after it was last read and
    float* Hbuf = Rbuf + (size_t)32 * MP;                              // "everyone has read it" needs no barrier of its own
    float* Abuf = Hbuf + (size_t)32 * MP;                              // the graph's adjacency rows: read once, used by every timestep
    float* Nbuf = Abuf + (size_t)E * 32 * AP;                          // [E][32] incoming edges per type (row sums of A_e)
    float* Bbuf = Nbuf + E * 32;                                       // [E][BN] edge biases, zero-padded
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kq = lane >> 4;
    const int g = blockIdx.x;
    const int v = a.v;
    const bool mm = wave < NT;                                         // this wave owns a column tile
    const int tile = mm ? wave : 0;

    auto image = [&](int s) -> const float* {                          // stage s of a timestep
        return s < E ? a.eimg + (size_t)s * C::IMG : a.gimg + (size_t)(s - E) * C::IMG;
    };
    // fragment (rows t*16 + li, k = 16c + 4kq ..) of a [32][MP] LDS block
    auto frag_from_lds = [&](Frag<D>& f, const float* blk, int t) {
        const float* rowp = blk + (size_t)(t * 16 + li) * MP + 4 * kq;
#pragma unroll
        for (int c = 0; c < NC; ++c) f.v[c] = *reinterpret_cast<const f32x4*>(rowp + 16 * c);
#pragma unroll
        for (int q = 0; q < NR; ++q) f.r[q] = rowp[16 * NC + 4 * q - 4 * kq + kq];
    };
    // accumulator tile (lane (li,kq): row t*16+li, columns 16*tile + 4kq ..) -> LDS block
    auto tile_to_lds = [&](float* blk, int t, f32x4 val) {
        *reinterpret_cast<f32x4*>(blk + (size_t)(t * 16 + li) * MP + 16 * tile + 4 * kq) = val;
    };

    for (int idx = tid; idx < E * 32 * 32; idx += NW * 64) {
        const int j = idx & 31, i = (idx >> 5) & 31, e = idx >> 10;
        Abuf[(e * 32 + i) * AP + j] = (i < a.v && j < a.v) ? a.A[(((size_t)blockIdx.x * E + e) * a.v + i) * a.v + j] : 0.f;
    }
    for (int idx = tid; idx < E * C::BN; idx += NW * 64) {
        const int e = idx / C::BN, n = idx - e * C::BN;
        Bbuf[idx] = (a.ebias && n < D) ? a.ebias[(size_t)e * D + n] : 0.f;
    }
    __syncthreads();
    if (tid < E * 32) {
        float sum = 0.f;
        for (int j = 0; j < 32; ++j) sum += Abuf[tid * AP + j];
        Nbuf[tid] = sum;
    }
    Frag<D> hf[2], xf[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int i = t * 16 + li;
        if (i < v) load_frag<D>(hf[t], a.h0 + (size_t)g * v * D, i, kq);
        else {
#pragma unroll
            for (int c = 0; c < (NC > 0 ? NC : 1); ++c) hf[t].v[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < (NR > 0 ? NR : 1); ++q) hf[t].r[q] = 0.f;
        }
    }
    // this lane's four columns of the gate / candidate biases
    const int col0 = 16 * tile + 4 * kq;
    f32x4 b_r = {0.f, 0.f, 0.f, 0.f}, b_u = b_r, b_c = b_r;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (col0 + e < D) { b_r[e] = a.bg[col0 + e]; b_u[e] = a.bg[D + col0 + e]; b_c[e] = a.bc[col0 + e]; }
    }

    // this wave's column tile of a stage image, straight from L2: ONE per-lane byte offset for every image (scalar base + 32-bit
    // offset + immediate; per-load 64-bit lane addresses get hoisted for all ten images of a timestep and spill)
    const unsigned w_off = (unsigned)(kq * C::BN + li + tile * 16) * 16u;
    const unsigned wr_off = (unsigned)(C::MAIN + kq * C::BN + li + tile * 16) * 4u;
    auto load_w = [&](TileWeights<D>& w, const float* gimg) {
        const unsigned long long p = reinterpret_cast<unsigned long long>(gimg);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)p), hi = __builtin_amdgcn_readfirstlane((unsigned)(p >> 32));
        const float* sb = reinterpret_cast<const float*>(((unsigned long long)hi << 32) | lo);
#pragma unroll
        for (int c = 0; c < NC; ++c) w.v[c] = ld4_b(sb, w_off + (unsigned)(c * 4 * C::BN) * 16u);
#pragma unroll
        for (int q = 0; q < NR; ++q) w.r[q] = ld1_b(sb, wr_off + (unsigned)(q * 4 * C::BN) * 4u);
    };
    // both row tiles against one weight slice, the two accumulators alternating: a single accumulator would make the 25 MFMAs of a
    // tile one dependent chain (40 clocks per link instead of 32)
    auto mma_pair = [&](auto zero_c, f32x4 (&acc)[2], const Frag<D> (&f)[2], const TileWeights<D>& w) {
        constexpr bool ZERO = decltype(zero_c)::value;
        f32x4 c0 = acc[0], c1 = acc[1];
        if constexpr (ZERO) { c0 = f32x4{0.f, 0.f, 0.f, 0.f}; c1 = c0; }
#pragma unroll
        for (int c = 0; c < NC; ++c) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w.v[c][e], f[0].v[c][e], c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w.v[c][e], f[1].v[c][e], c1, 0, 0, 0);
            }
        }
#pragma unroll
        for (int q = 0; q < NR; ++q) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w.r[q], f[0].r[q], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w.r[q], f[1].r[q], c1, 0, 0, 0);
        }
        acc[0] = c0; acc[1] = c1;
    };
    TileWeights<D> tw[2];
    if (mm) load_w(tw[0], image(0));

#define GGNN_DG_T(K) if (a.tdbg && blockIdx.x == 0 && lane == 0 && (wave == 0 || wave == 6) && step < 8) \
        a.tdbg[(step * 2 + (wave ? 1 : 0)) * 8 + (K)] = __builtin_amdgcn_s_memtime();
    for (int step = 0; step < a.steps; ++step) {
        const bool last = step + 1 == a.steps;
        GGNN_DG_T(0)
        // one stage: request the next stage's weight slice, multiply both row tiles by this stage's
#define GGNN_DG_STAGE(S, ACC, FR, ZERO)                                                                    \
        {                                                                                                  \
            __builtin_amdgcn_sched_barrier(0);   /* (keeps the look-ahead at ONE stage: 25 weight registers in flight, not 250) */ \
            if (mm && !(last && (S) + 1 == NS)) load_w(tw[((S) + 1) & 1], image(((S) + 1) % NS));                         \
            __builtin_amdgcn_sched_barrier(0);                                                             \
            if (mm) mma_pair(std::integral_constant<bool, ZERO>{}, ACC, FR, tw[(S) & 1]);                  \
            __builtin_amdgcn_sched_barrier(0);                                                             \
        }
        // ---- E transform stages: M_e = h W_e, column tile `tile`, into LDS ------------------------------------------------------
        f32x4 acc[2];
#define GGNN_DG_XFORM(S)                                                                                   \
        if constexpr ((S) < E) {                                                                           \
            GGNN_DG_STAGE(S, acc, hf, true)                                                                \
            if (mm) { tile_to_lds(Mbuf + (size_t)(S) * 32 * MP, 0, acc[0]); tile_to_lds(Mbuf + (size_t)(S) * 32 * MP, 1, acc[1]); } \
        }
        GGNN_DG_XFORM(0) GGNN_DG_XFORM(1) GGNN_DG_XFORM(2) GGNN_DG_XFORM(3) GGNN_DG_XFORM(4) GGNN_DG_XFORM(5) GGNN_DG_XFORM(6) GGNN_DG_XFORM(7)
#undef GGNN_DG_XFORM
        GGNN_DG_T(1)
        __syncthreads();                                               // (1) every M_e is complete
        GGNN_DG_T(2)
        // ---- aggregation on the matrix pipe: acts^T tile = sum_e M_e^T[columns of this tile][src] . A_e^T[src][dst], K = 32 source
        //      vertices = 8 MFMAs per (edge type, row tile); the bias term sum_j A_e[i,j] b_e = nin_e[i] b_e in the epilogue.  (A
        //      lane-per-column walk over the non-zeros of each adjacency row -- the form of ggnn_dense_aggregate_f32 -- is a chain of
        //      dependent LDS reads: 9.3k clocks per timestep against 2.5k here.)
        f32x4 aa[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        if (mm) {
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const float* mcol = Mbuf + (size_t)e * 32 * MP + 16 * tile + li;          // M_e[.][column li of this tile]
                const float* arow0 = Abuf + (e * 32 + li) * AP + kq;                       // A_e[row li][.]
                const float* arow1 = arow0 + 16 * AP;
#pragma unroll
                for (int s4 = 0; s4 < 8; ++s4) {
                    const float wv = mcol[(size_t)(4 * s4 + kq) * MP];
                    aa[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv, arow0[4 * s4], aa[0], 0, 0, 0);
                    aa[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv, arow1[4 * s4], aa[1], 0, 0, 0);
                }
            }
            if (a.ebias) {
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const f32x4 be = *reinterpret_cast<const f32x4*>(Bbuf + e * C::BN + 16 * tile + 4 * kq);
                    aa[0] += Nbuf[e * 32 + li] * be;
                    aa[1] += Nbuf[e * 32 + 16 + li] * be;
                }
            }
            tile_to_lds(Xbuf, 0, aa[0]); tile_to_lds(Xbuf, 1, aa[1]);
        }
        GGNN_DG_T(3)
        __syncthreads();                                               // (2) acts complete
        frag_from_lds(xf[0], Xbuf, 0); frag_from_lds(xf[1], Xbuf, 1);
        GGNN_DG_T(4)
        // ---- gates --------------------------------------------------------------------------------------------------------------
        f32x4 ar[2], au[2], ac[2];
        GGNN_DG_STAGE(E + 0, ar, xf, true)
        GGNN_DG_STAGE(E + 1, ar, hf, false)
        GGNN_DG_STAGE(E + 2, au, xf, true)
        GGNN_DG_STAGE(E + 3, au, hf, false)
        GGNN_DG_STAGE(E + 4, ac, xf, true)                             // candidate, acts part (acts are still in xf)
        f32x4 htile[2], u4[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            // chunk `tile` of the state fragment = the state at this lane's accumulator positions (row li, columns 16*tile + 4kq ..)
            f32x4 hv = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < NC; ++c) if (c == tile) hv = hf[t].v[c];
            if constexpr (NR > 0) { if (tile == NC) { hv = f32x4{0.f, 0.f, 0.f, 0.f}; hv.x = __shfl(hf[t].r[0], li); hv.y = __shfl(hf[t].r[0], li + 16); hv.z = __shfl(hf[t].r[0], li + 32); hv.w = __shfl(hf[t].r[0], li + 48); if (kq) hv = f32x4{0.f, 0.f, 0.f, 0.f}; } }
            htile[t] = hv;
            f32x4 r4;
#pragma unroll
            for (int e = 0; e < 4; ++e) { r4[e] = sigmoid_f(ar[t][e] + b_r[e]); u4[t][e] = sigmoid_f(au[t][e] + b_u[e]); }
            ar[t] = r4 * hv;                                           // r * h tile
        }
        GGNN_DG_T(5)
        if (mm) { tile_to_lds(Rbuf, 0, ar[0]); tile_to_lds(Rbuf, 1, ar[1]); }
        __syncthreads();                                               // (3) r*h complete
        frag_from_lds(xf[0], Rbuf, 0); frag_from_lds(xf[1], Rbuf, 1);
        GGNN_DG_T(6)
        GGNN_DG_STAGE(E + 5, ac, xf, false)                            // candidate, r*h part
        GGNN_DG_T(7)
        f32x4 hn[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float c = tanh_f(ac[t][e] + b_c[e]);
                hn[t][e] = u4[t][e] * htile[t][e] + (1.0f - u4[t][e]) * c;
            }
        }
        if (last) {
            if (mm) {
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int i = t * 16 + li;
                    if (i < v && col0 < D) *reinterpret_cast<f32x4*>(a.out + ((size_t)g * v + i) * D + col0) = hn[t];
                }
            }
        } else {
            if (mm) { tile_to_lds(Hbuf, 0, hn[0]); tile_to_lds(Hbuf, 1, hn[1]); }
            __syncthreads();                                           // (4) new state complete
            frag_from_lds(hf[0], Hbuf, 0); frag_from_lds(hf[1], Hbuf, 1);
        }
#undef GGNN_DG_STAGE
    }
}

template <int D, int E>
static int launch_dense_graph(const DenseGraphArgs& a, hipStream_t st) {
    using C = StageCfg<D>;
    constexpr int NW = 8;
    const size_t ldsb = ((size_t)(E + 3) * 32 * (C::BN + 4) + (size_t)E * 32 * 33 + (size_t)E * 32 + (size_t)E * C::BN) * sizeof(float);
    static std::atomic<unsigned long long> lds_ok{0};
    if (ldsb > 64 * 1024) GGNN_CHECK_HIP((allow_dynamic_lds(&ggnn_dense_graph_kernel<D, E, NW>, ldsb, lds_ok)));
    hipLaunchKernelGGL((ggnn_dense_graph_kernel<D, E, NW>), dim3(a.b), dim3(NW * 64), ldsb, st, a);
    GGNN_CHECK_HIP(hipGetLastError());
    return GGNN_OK;
}

}  // namespace ggnn

using namespace ggnn;

extern "C" int ggnn_dense_propagate_supported(int v, int E, int D) {
    if (!(v >= 1 && v <= 32 && (E == 2 || E == 4 || E == 6 || E == 8) && (D == 100 || D == 64 || D == 32))) return 0;
    const int bn = (D + 15) / 16 * 16;
    const size_t ldsb = ((size_t)(E + 3) * 32 * (bn + 4) + (size_t)E * 32 * 33 + (size_t)E * 32 + (size_t)E * bn) * sizeof(float);   // M_e, exchange block, adjacency rows, in-degrees, biases
    return ldsb <= (size_t)160 * 1024;
}

// The E edge-weight stage images of the graph-resident kernel: always the f32 stage image (this kernel reads its weights as f32
// register tiles; ggnn_edge_weights_pack_f32 writes the format of the process's matrix path, which may be the split one).
template <int D>
__global__ void dense_edge_pack_kernel(const float* __restrict__ W, float* __restrict__ out) {
    const int t = blockIdx.y;
    pack_stage_image<D>(W + (size_t)t * D * D, 0, 0, D, out + (size_t)t * StageCfg<D>::IMG, blockIdx.x * blockDim.x + threadIdx.x,
                        gridDim.x * blockDim.x);
}

static size_t dense_edge_f32_bytes(int D, int T) {
    switch (D) {
        case 100: return (size_t)T * StageCfg<100>::IMG * sizeof(float);
        case 64: return (size_t)T * StageCfg<64>::IMG * sizeof(float);
        case 32: return (size_t)T * StageCfg<32>::IMG * sizeof(float);
        default: return 0;
    }
}
// (the packed buffers hold the f32 stage images FOLLOWED by the split ones: which kernel runs is decided per launch -- matrix
// path of the process, and whether the split kernel's LDS blocks fit the launch's number of edge types)
extern "C" size_t ggnn_dense_edge_packed_bytes(int D, int T) {
    if (T <= 0 || !dense_edge_f32_bytes(D, T)) return 0;
    return dense_edge_f32_bytes(D, T) + dense_split_edge_bytes(D, T);
}

extern "C" int ggnn_dense_edge_pack_f32(const float* W, int T, int D, float* packed, ggnn_stream_t stream) {
    GGNN_CHECK_ARG(W && packed && aligned16(packed) && T > 0 && T <= 64, "null or misaligned pointer, or T = %d outside 1..64", T);
    hipStream_t st = (hipStream_t)stream;
    switch (D) {
        case 100: hipLaunchKernelGGL((dense_edge_pack_kernel<100>), dim3(8, T), dim3(256), 0, st, W, packed); break;
        case 64: hipLaunchKernelGGL((dense_edge_pack_kernel<64>), dim3(8, T), dim3(256), 0, st, W, packed); break;
        case 32: hipLaunchKernelGGL((dense_edge_pack_kernel<32>), dim3(8, T), dim3(256), 0, st, W, packed); break;
        default: return fail(GGNN_E_UNSUPPORTED, "no graph-resident dense kernel for hidden size %d", D);
    }
    GGNN_CHECK_HIP(hipGetLastError());
    return dense_split_pack_edge(W, T, D, packed + dense_edge_f32_bytes(D, T) / sizeof(float), st);
}

static size_t dense_gru_f32_bytes(int D) {
    switch (D) {
        case 100: return (size_t)6 * StageCfg<100>::IMG * sizeof(float);
        case 64: return (size_t)6 * StageCfg<64>::IMG * sizeof(float);
        case 32: return (size_t)6 * StageCfg<32>::IMG * sizeof(float);
        default: return 0;
    }
}
extern "C" size_t ggnn_dense_gru_packed_bytes(int D) {
    return dense_gru_f32_bytes(D) ? dense_gru_f32_bytes(D) + dense_split_gru_bytes(D) : 0;
}

extern "C" int ggnn_dense_gru_pack_f32(const float* Wg, const float* Wc, int D, float* packed, ggnn_stream_t stream) {
    GGNN_CHECK_ARG(Wg && Wc && packed && aligned16(packed), "null or misaligned pointer");
    hipStream_t st = (hipStream_t)stream;
    switch (D) {
        case 100: hipLaunchKernelGGL((dense_gru_pack_kernel<100>), dim3(8, 6), dim3(256), 0, st, Wg, Wc, packed); break;
        case 64: hipLaunchKernelGGL((dense_gru_pack_kernel<64>), dim3(8, 6), dim3(256), 0, st, Wg, Wc, packed); break;
        case 32: hipLaunchKernelGGL((dense_gru_pack_kernel<32>), dim3(8, 6), dim3(256), 0, st, Wg, Wc, packed); break;
        default: return fail(GGNN_E_UNSUPPORTED, "no graph-resident dense kernel for hidden size %d", D);
    }
    GGNN_CHECK_HIP(hipGetLastError());
    return dense_split_pack_gru(Wg, Wc, D, packed + dense_gru_f32_bytes(D) / sizeof(float), st);
}

// split form (bf16 pipe, ggnn_dense_graph_split.hip) where it exists and fits; GGNN_DENSE_SPLIT=0 keeps the f32-MFMA kernel
extern "C" int ggnn_dense_propagate_is_split(int v, int E, int D) {
    static const bool want_split = [] { const char* e = getenv("GGNN_DENSE_SPLIT"); return !e || atoi(e) != 0; }();
    return (want_split && split_matrix_path() && ggnn_dense_propagate_supported(v, E, D) && dense_split_supported(v, E, D)) ? 1 : 0;
}

extern "C" int ggnn_dense_propagate_f32(const float* h0, const float* A, const float* edge_packed, const float* gru_packed,
                                        const float* edge_bias, const float* bg, const float* bc, float* out, int b, int v, int E,
                                        int D, int steps, int fmt, ggnn_stream_t stream) {
    GGNN_CHECK_ARG(b >= 0 && steps >= 1, "bad sizes b=%d steps=%d", b, steps);
    GGNN_CHECK_ARG(fmt == 0 || fmt == GGNN_GRU_FMT_F16X2 || fmt == GGNN_GRU_FMT_BF16X3, "fmt %d is not a GGNN_GRU_FMT_* value", fmt);
    if (!ggnn_dense_propagate_supported(v, E, D))
        return fail(GGNN_E_UNSUPPORTED, "graph-resident dense forward: v <= 32, E in {2,4,6,8}, hidden size 32/64/100 (got v=%d E=%d D=%d)", v, E, D);
    if (b == 0) return GGNN_OK;
    GGNN_CHECK_ARG(h0 && A && edge_packed && gru_packed && bg && bc && out, "null pointer");
    GGNN_CHECK_ARG(aligned16(h0) && aligned16(out) && aligned16(edge_packed) && aligned16(gru_packed) && (!edge_bias || aligned16(edge_bias)),
                   "pointers must be 16-byte aligned");
    DenseGraphArgs a{h0, A, edge_packed, gru_packed, edge_bias, bg, bc, out, b, v, steps, nullptr};
    { const char* e = getenv("GGNN_DG_TPTR"); a.tdbg = e ? (unsigned long long*)strtoull(e, nullptr, 10) : nullptr; }
    hipStream_t st = (hipStream_t)stream;
    if (ggnn_dense_propagate_is_split(v, E, D)) {
        DenseGraphArgs s = a;
        s.eimg = edge_packed + (dense_edge_f32_bytes(D, E) + dense_split_images_offset(D, E, fmt)) / sizeof(float);
        s.gimg = gru_packed + (dense_gru_f32_bytes(D) + dense_split_images_offset(D, 6, fmt)) / sizeof(float);
        return dense_split_launch(s, E, D, fmt, st);
    }
#define GGNN_DG_CASE(DD, EE) if (D == DD && E == EE) return launch_dense_graph<DD, EE>(a, st);
    GGNN_DG_CASE(100, 4) GGNN_DG_CASE(100, 8) GGNN_DG_CASE(100, 2) GGNN_DG_CASE(100, 6)
    GGNN_DG_CASE(64, 4) GGNN_DG_CASE(64, 8) GGNN_DG_CASE(64, 2) GGNN_DG_CASE(64, 6)
    GGNN_DG_CASE(32, 4) GGNN_DG_CASE(32, 8) GGNN_DG_CASE(32, 2) GGNN_DG_CASE(32, 6)
#undef GGNN_DG_CASE
    return fail(GGNN_E_UNSUPPORTED, "graph-resident dense forward: unsupported shape");
}
