// K7 in split form: the graph-resident dense GGNN forward (chem_tensorflow_dense.py:93-117; see ggnn_dense_graph.hip for the
// f32-MFMA kernel this one replaces on the default matrix path) with every D x D product on the bf16 matrix pipe (ggnn_split.hpp:
// f32 operands as three bf16 pieces, six products, f32 accumulation).
//
// The f32 kernel is bound by its MFMAs -- ten stages of 2 x 25 v_mfma_f32_16x16x4_f32 per wave and timestep at 3.4-3.8k clocks for
// 3.2k of matrix-pipe time -- and had no registers left for split planes: every wave held the FULL f32 activation fragments of both
// row tiles (100 registers) next to two weight slices.  Here the activations never live in registers:
//   * what a stage group consumes (the state h, the aggregated messages, r*h) sits in LDS ALREADY SPLIT, in the MFMA operand layout
//     [plane][32-chunk][lane group g][row][8 x bf16]: a wave reads one ds_read_b128 per (plane, chunk, row tile) straight into the
//     operand of six MFMAs (conflict-free as laid out: the b128 lane groups cover 16 consecutive 16-byte slots);
//   * the PRODUCER splits: a wave owns column tile `tile` of every block, i.e. the four k values 16 tile + 4 kq .. of rows li and
//     li + 16 -- exactly slots 4 (tile & 1) .. + 3 of lane group kq in chunk tile >> 1 -- splits those eight values (22 vector
//     instructions) and writes three ds_write_b64 per row tile.  Eight waves splitting the same fragments redundantly (what keeping
//     the f32 kernel's structure would have meant) is 800 vector instructions per wave and timestep; this is 66.
//   * a wave's slice of a stage's weights -- its column tile of the split image: 9 x 16 bytes + the remainder float per lane -- comes
//     from L2 into registers one stage ahead, as in the f32 kernel.
// Stage order of a timestep: E transforms, h -> r, h -> u (all on the state planes) | aggregation (f32 MFMA: the 0/1 adjacency
// times M_e is exact) | x -> r, x -> u, x -> c | gates | r*h -> c | blend.  The h products come FIRST (the f32 kernel adds them after
// the x products): one operand block live at a time.  r and u are therefore sums in the other order -- within the parity tolerances
// of tests/test_gpu_dense.py, not bit-identical to the three-launch path.
#include "ggnn_dense_graph.hpp"
#include "ggnn_split.hpp"
#include <type_traits>

namespace ggnn {

namespace {

__device__ __forceinline__ float sigm(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-kLog2e * x)); }
__device__ __forceinline__ float tanhf_(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.0f * kLog2e * x)); }

// this wave's column tile of one split stage image: NP planes (three bf16 / two f16) x NC2 chunks of 8 halves per lane + the remainder rows
template <int D, int FMT>
struct TileW {
    u32x4 p[SplitFmt<FMT>::NP][SplitCfg<D>::NC2 > 0 ? SplitCfg<D>::NC2 : 1];
    float r[StageCfg<D>::NR > 0 ? StageCfg<D>::NR : 1];
};

}  // namespace

// six GRU images in the order the kernel consumes them: (h,r) (h,u) (x,r) (x,u) (x,c) (r*h,c);  Wg rows [x | h], columns [r | u]
template <int D, int FMT>
__global__ void dense_split_pack_kernel(const float* __restrict__ W, const float* __restrict__ Wg, const float* __restrict__ Wc,
                                        float* __restrict__ out) {
    const int i = blockIdx.y;
    const int first = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
    float* img = out + (size_t)i * SplitCfg<D, FMT>::IMG;
    if (W) {                                                          // edge type i
        pack_split_image<D, FMT>(StageValue<D>{W + (size_t)i * D * D, 0, 0, D, -1, nullptr, 0, 0, -1}, img, first, stride);
    } else {
        constexpr int rows[6] = {1, 1, 0, 0, 0, 1}, cols[6] = {0, 1, 0, 1, 0, 0};
        if (i < 4) pack_split_image<D, FMT>(StageValue<D>{Wg, rows[i] * D, cols[i] * D, 2 * D, -1, nullptr, 0, 0, -1}, img, first, stride);
        else pack_split_image<D, FMT>(StageValue<D>{Wc, rows[i] * D, 0, D, -1, nullptr, 0, 0, -1}, img, first, stride);
    }
}

// FMT (round 5): operand format of every D x D product of the kernel, per launch -- the exact kSplitBf16x3 or kSplitF16x2 (two f16
// pieces, three products, two operand planes) when the caller has PROVEN its range for the launch (dense_model.py / formats.py:
// states tanh-bounded, |acts| <= v E (D max|W| S + max|b|), weights <= 255.875).  Accumulators then hold 2^8 x the sums
// (SplitFmt<FMT>::acc_scale is applied where they are consumed).
template <int D, int E, int NW, int FMT>
__global__ __launch_bounds__(NW * 64) void ggnn_dense_graph_split_kernel(DenseGraphArgs a) {
    using C = StageCfg<D>;
    using SC = SplitCfg<D, FMT>;
    constexpr int NP = SplitFmt<FMT>::NP;
    constexpr float ASC = SplitFmt<FMT>::acc_scale;
    constexpr int NT = C::NT, NC = C::NC, NR = C::NR, NC2 = SC::NC2;
    constexpr int MP = C::BN + 4;                                      // row pitch of the f32 M_e blocks (floats)
    constexpr int NS = E + 6;                                          // stages per timestep
    constexpr int AP = 33;
    constexpr int PSLOT = 32 * 4;                                      // floats of one (plane, chunk, g) slab: 32 rows x 16 bytes
    constexpr int PBLK = NP * NC2 * 4 * PSLOT + 32 * 4;                // floats of one split operand block (+ the remainder columns [32][4])
    static_assert(NT <= NW && NS % 2 == 0 && NR <= 1, "one column tile per wave; two weight slots with a fixed phase per timestep");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* Mbuf = lds;                                                 // [E][32][MP] f32: the transformed states, for the aggregation
    float* Hblk = Mbuf + (size_t)E * 32 * MP;                          // split operand blocks: state | aggregated messages | r*h
    float* Xblk = Hblk + PBLK;
    float* Rblk = Xblk + PBLK;
    float* Abuf = Rblk + PBLK;                                         // [E][32][AP] adjacency rows
    float* Nbuf = Abuf + (size_t)E * 32 * AP;                          // [E][32] in-degrees per type
    float* Bbuf = Nbuf + E * 32;                                       // [E][BN] edge biases, zero-padded
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kq = lane >> 4;
    const int g = blockIdx.x;
    const int v = a.v;
    const bool mm = wave < NT;
    const int tile = mm ? wave : 0;
    const int col0 = 16 * tile + 4 * kq;
    const bool tail = NR > 0 && tile == NC;                            // the tile of the D % 16 remainder columns (kq == 0 lanes hold them)

    auto image = [&](int s) -> const float* { return s < E ? a.eimg + (size_t)s * SC::IMG : a.gimg + (size_t)(s - E) * SC::IMG; };
    auto tile_to_lds = [&](float* blk, int t, f32x4 val) {
        *reinterpret_cast<f32x4*>(blk + (size_t)(t * 16 + li) * MP + 16 * tile + 4 * kq) = val;
    };
    // this wave's four columns of row tile t, split, into an operand block (see the file header for the slot arithmetic)
    auto tile_to_planes = [&](float* blk, int t, f32x4 val) {
        if (!mm) return;
        const int row = t * 16 + li;
        if (tail) {
            if (kq == 0) *reinterpret_cast<f32x4*>(blk + NP * NC2 * 4 * PSLOT + row * 4) = val;      // f32 remainder columns
            return;
        }
        unsigned h0, m0, l0, h1, m1, l1;
        split_pair<FMT>(val.x, val.y, h0, m0, l0);
        split_pair<FMT>(val.z, val.w, h1, m1, l1);
        const int c2 = tile >> 1;
        float* dst = blk + ((size_t)(c2 * 4 + kq)) * PSLOT + row * 4 + 2 * (tile & 1);
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        *reinterpret_cast<u32x2*>(dst) = u32x2{h0, h1};
        *reinterpret_cast<u32x2*>(dst + NC2 * 4 * PSLOT) = u32x2{m0, m1};
        if constexpr (NP > 2) *reinterpret_cast<u32x2*>(dst + 2 * NC2 * 4 * PSLOT) = u32x2{l0, l1};
    };

    for (int idx = tid; idx < E * 32 * 32; idx += NW * 64) {
        const int j = idx & 31, i = (idx >> 5) & 31, e = idx >> 10;
        Abuf[(e * 32 + i) * AP + j] = (i < a.v && j < a.v) ? a.A[(((size_t)blockIdx.x * E + e) * a.v + i) * a.v + j] : 0.f;
    }
    for (int idx = tid; idx < E * C::BN; idx += NW * 64) {
        const int e = idx / C::BN, n = idx - e * C::BN;
        Bbuf[idx] = (a.ebias && n < D) ? a.ebias[(size_t)e * D + n] : 0.f;
    }
    // the initial state: this wave's column tile of both row tiles (also kept as f32 for r*h and the blend)
    f32x4 htile[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int i = t * 16 + li;
        f32x4 hv = {0.f, 0.f, 0.f, 0.f};
        if (mm && i < v && col0 < D) hv = *reinterpret_cast<const f32x4*>(a.h0 + ((size_t)g * v + i) * D + col0);
        htile[t] = hv;
        tile_to_planes(Hblk, t, hv);
    }
    __syncthreads();
    if (tid < E * 32) {
        float sum = 0.f;
        for (int j = 0; j < 32; ++j) sum += Abuf[tid * AP + j];
        Nbuf[tid] = sum;
    }
    f32x4 b_r = {0.f, 0.f, 0.f, 0.f}, b_u = b_r, b_c = b_r;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (col0 + e < D) { b_r[e] = a.bg[col0 + e]; b_u[e] = a.bg[D + col0 + e]; b_c[e] = a.bc[col0 + e]; }
    }

    // byte offsets of this wave's weight slice inside a split image (two column halves, ggnn_split.hpp)
    const bool hb = tile >= SC::TA;
    const int nth = hb ? NT - SC::TA : SC::TA, til = hb ? tile - SC::TA : tile;
    const unsigned w_base = (hb ? (unsigned)SC::HA_BYTES : 0u) + (unsigned)(kq * nth * 16 + li + til * 16) * 16u;
    const unsigned w_cst = (unsigned)(4 * nth * 16) * 16u, w_pst = (unsigned)NC2 * w_cst;          // chunk / plane pitch in bytes
    const unsigned wr_base = (hb ? (unsigned)SC::HA_BYTES : 0u) + (unsigned)NP * w_pst + (unsigned)(kq * nth * 16 + li + til * 16) * 4u;
    auto load_w = [&](TileW<D, FMT>& w, const float* gimg) {
        const unsigned long long p = reinterpret_cast<unsigned long long>(gimg);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)p), hi = __builtin_amdgcn_readfirstlane((unsigned)(p >> 32));
        const float* sb = reinterpret_cast<const float*>(((unsigned long long)hi << 32) | lo);
#pragma unroll
        for (int pl = 0; pl < NP; ++pl)
#pragma unroll
            for (int c2 = 0; c2 < NC2; ++c2)
                w.p[pl][c2] = __builtin_bit_cast(u32x4, ld4_b(sb, w_base + (unsigned)pl * w_pst + (unsigned)c2 * w_cst));
#pragma unroll
        for (int q = 0; q < NR; ++q) w.r[q] = ld1_b(sb, wr_base + (unsigned)(q * 4 * nth * 16) * 4u);
    };
    // both row tiles of an operand block against one weight slice: per chunk 6 operand reads feed 12 MFMAs (two accumulator chains)
    auto mma_pair = [&](auto zero_c, f32x4 (&acc)[2], const float* blk, const TileW<D, FMT>& w) {
        constexpr bool ZERO = decltype(zero_c)::value;
        f32x4 c0 = acc[0], c1 = acc[1];
        if constexpr (ZERO) { c0 = f32x4{0.f, 0.f, 0.f, 0.f}; c1 = c0; }
        const u32x4* ob = reinterpret_cast<const u32x4*>(blk) + kq * 32 + li;            // (plane, chunk) slabs are 128 slots apart
#pragma unroll
        for (int c2 = 0; c2 < NC2; ++c2) {
            const u32x4 ah0 = ob[(0 * NC2 + c2) * 128], ah1 = ob[(0 * NC2 + c2) * 128 + 16];
            const u32x4 am0 = ob[(1 * NC2 + c2) * 128], am1 = ob[(1 * NC2 + c2) * 128 + 16];
            if constexpr (FMT == kSplitF16x2) {                        // three products per chunk and row tile, smallest first
                const u32x4 wh = w.p[0][c2], wm = w.p[1][c2];
                c0 = mfma_f16(wm, ah0, c0); c1 = mfma_f16(wm, ah1, c1);
                c0 = mfma_f16(wh, am0, c0); c1 = mfma_f16(wh, am1, c1);
                c0 = mfma_f16(wh, ah0, c0); c1 = mfma_f16(wh, ah1, c1);
                continue;
            }
            const u32x4 al0 = ob[((NP - 1) * NC2 + c2) * 128], al1 = ob[((NP - 1) * NC2 + c2) * 128 + 16];
            const u32x4 wh = w.p[0][c2], wm = w.p[1][c2], wl = w.p[NP - 1][c2];
            c0 = mfma_bf16(wl, ah0, c0); c1 = mfma_bf16(wl, ah1, c1);
            c0 = mfma_bf16(wm, am0, c0); c1 = mfma_bf16(wm, am1, c1);
            c0 = mfma_bf16(wm, ah0, c0); c1 = mfma_bf16(wm, ah1, c1);
            c0 = mfma_bf16(wh, al0, c0); c1 = mfma_bf16(wh, al1, c1);
            c0 = mfma_bf16(wh, am0, c0); c1 = mfma_bf16(wh, am1, c1);
            c0 = mfma_bf16(wh, ah0, c0); c1 = mfma_bf16(wh, ah1, c1);
        }
        if constexpr (NR > 0) {
            const float* rb = blk + NP * NC2 * 4 * PSLOT;                                  // [32][4] remainder columns
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w.r[0], rb[li * 4 + kq], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w.r[0], rb[(16 + li) * 4 + kq], c1, 0, 0, 0);
        }
        acc[0] = c0; acc[1] = c1;
    };
    TileW<D, FMT> tw[2];
    if (mm) load_w(tw[0], image(0));
    __syncthreads();                                                   // state planes, adjacency, in-degrees, biases

#define GGNN_DGS_T(K) if (a.tdbg && blockIdx.x == 0 && lane == 0 && (wave == 0 || wave == 6) && step < 8) \
        a.tdbg[(step * 2 + (wave ? 1 : 0)) * 8 + (K)] = __builtin_amdgcn_s_memtime();
    for (int step = 0; step < a.steps; ++step) {
        const bool last = step + 1 == a.steps;
        GGNN_DGS_T(0)
#define GGNN_DGS_STAGE(S, ACC, BLK, ZERO)                                                                  \
        {                                                                                                  \
            __builtin_amdgcn_sched_barrier(0);   /* (the look-ahead stays at ONE stage: 37 weight registers in flight) */ \
            if (mm && !(last && (S) + 1 == NS)) load_w(tw[((S) + 1) & 1], image(((S) + 1) % NS));          \
            __builtin_amdgcn_sched_barrier(0);                                                             \
            if (mm) mma_pair(std::integral_constant<bool, ZERO>{}, ACC, BLK, tw[(S) & 1]);                 \
            __builtin_amdgcn_sched_barrier(0);                                                             \
        }
        // ---- E transform stages: M_e = h W_e, column tile `tile`, into LDS (f32: the aggregation's operand) ------------------
        f32x4 acc[2];
#define GGNN_DGS_XFORM(S)                                                                                  \
        if constexpr ((S) < E) {                                                                           \
            GGNN_DGS_STAGE(S, acc, Hblk, true)                                                             \
            if (mm) { tile_to_lds(Mbuf + (size_t)(S) * 32 * MP, 0, acc[0] * ASC); tile_to_lds(Mbuf + (size_t)(S) * 32 * MP, 1, acc[1] * ASC); } \
        }
        GGNN_DGS_XFORM(0) GGNN_DGS_XFORM(1) GGNN_DGS_XFORM(2) GGNN_DGS_XFORM(3) GGNN_DGS_XFORM(4) GGNN_DGS_XFORM(5) GGNN_DGS_XFORM(6) GGNN_DGS_XFORM(7)
#undef GGNN_DGS_XFORM
        // ---- the state's share of the gates, while the state planes are the live operand ----------------------------------------
        f32x4 ar[2], au[2], ac[2];
        GGNN_DGS_STAGE(E + 0, ar, Hblk, true)
        GGNN_DGS_STAGE(E + 1, au, Hblk, true)
        GGNN_DGS_T(1)
        __syncthreads();                                               // (1) every M_e is complete
        GGNN_DGS_T(2)
        // ---- aggregation on the matrix pipe (f32: A_e is 0/1, the products are exact), as in the f32 kernel --------------------
        f32x4 aa[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        if (mm) {
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const float* mcol = Mbuf + (size_t)e * 32 * MP + 16 * tile + li;
                const float* arow0 = Abuf + (e * 32 + li) * AP + kq;
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
        }
        tile_to_planes(Xblk, 0, aa[0]); tile_to_planes(Xblk, 1, aa[1]);
        GGNN_DGS_T(3)
        __syncthreads();                                               // (2) acts complete, split
        GGNN_DGS_T(4)
        // ---- the messages' share of the gates and of the candidate ----------------------------------------------------------------
        GGNN_DGS_STAGE(E + 2, ar, Xblk, false)
        GGNN_DGS_STAGE(E + 3, au, Xblk, false)
        GGNN_DGS_STAGE(E + 4, ac, Xblk, true)
        f32x4 u4[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x4 r4;
#pragma unroll
            for (int e = 0; e < 4; ++e) { r4[e] = sigm(ar[t][e] * ASC + b_r[e]); u4[t][e] = sigm(au[t][e] * ASC + b_u[e]); }
            tile_to_planes(Rblk, t, r4 * htile[t]);                    // r * h tile
        }
        GGNN_DGS_T(5)
        __syncthreads();                                               // (3) r*h complete, split
        GGNN_DGS_T(6)
        GGNN_DGS_STAGE(E + 5, ac, Rblk, false)
        GGNN_DGS_T(7)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x4 hn;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float c = tanhf_(ac[t][e] * ASC + b_c[e]);
                hn[e] = u4[t][e] * htile[t][e] + (1.0f - u4[t][e]) * c;
            }
            if (col0 >= D) hn = f32x4{0.f, 0.f, 0.f, 0.f};              // (the padding columns of the last tile stay zero operands)
            htile[t] = hn;
            if (last) {
                const int i = t * 16 + li;
                if (mm && i < v && col0 < D) *reinterpret_cast<f32x4*>(a.out + ((size_t)g * v + i) * D + col0) = hn;
            } else {
                tile_to_planes(Hblk, t, hn);
            }
        }
        if (!last) __syncthreads();                                    // (4) new state complete, split
#undef GGNN_DGS_STAGE
    }
}

static size_t split_lds_bytes(int D, int E, int np = 3) {
    const int bn = (D + 15) / 16 * 16, nc2 = (D / 16) / 2;
    const size_t pblk = (size_t)np * nc2 * 4 * 128 + 128;
    return ((size_t)E * 32 * (bn + 4) + 3 * pblk + (size_t)E * 32 * 33 + (size_t)E * 32 + (size_t)E * bn) * sizeof(float);
}

int dense_split_supported(int v, int E, int D) {
    if (!(v >= 1 && v <= 32 && (E == 2 || E == 4 || E == 6 || E == 8) && (D == 100 || D == 64 || D == 32))) return 0;
    return split_lds_bytes(D, E) <= (size_t)160 * 1024;
}

// bytes of T images of one format
static size_t images_bytes(int D, int T, int fmt) {
    const bool f2 = fmt == kSplitF16x2;
    switch (D) {
        case 100: return (size_t)T * (f2 ? SplitCfg<100, kSplitF16x2>::IMG_BYTES : SplitCfg<100>::IMG_BYTES);
        case 64: return (size_t)T * (f2 ? SplitCfg<64, kSplitF16x2>::IMG_BYTES : SplitCfg<64>::IMG_BYTES);
        case 32: return (size_t)T * (f2 ? SplitCfg<32, kSplitF16x2>::IMG_BYTES : SplitCfg<32>::IMG_BYTES);
        default: return 0;
    }
}
// the split section of a packed buffer: the bf16x3 images followed by the f16x2 ones (the format is chosen per launch)
size_t dense_split_edge_bytes(int D, int T) { return images_bytes(D, T, kSplitBf16x3) + images_bytes(D, T, kSplitF16x2); }
size_t dense_split_gru_bytes(int D) { return dense_split_edge_bytes(D, 6); }
size_t dense_split_images_offset(int D, int T, int fmt) { return fmt == kSplitF16x2 ? images_bytes(D, T, kSplitBf16x3) : 0; }

int dense_split_pack_edge(const float* W, int T, int D, float* packed, hipStream_t st) {
    float* p2 = packed + images_bytes(D, T, kSplitBf16x3) / sizeof(float);
    switch (D) {
        case 100: hipLaunchKernelGGL((dense_split_pack_kernel<100, kSplitBf16x3>), dim3(8, T), dim3(256), 0, st, W, nullptr, nullptr, packed);
                  hipLaunchKernelGGL((dense_split_pack_kernel<100, kSplitF16x2>), dim3(8, T), dim3(256), 0, st, W, nullptr, nullptr, p2); break;
        case 64: hipLaunchKernelGGL((dense_split_pack_kernel<64, kSplitBf16x3>), dim3(8, T), dim3(256), 0, st, W, nullptr, nullptr, packed);
                 hipLaunchKernelGGL((dense_split_pack_kernel<64, kSplitF16x2>), dim3(8, T), dim3(256), 0, st, W, nullptr, nullptr, p2); break;
        case 32: hipLaunchKernelGGL((dense_split_pack_kernel<32, kSplitBf16x3>), dim3(8, T), dim3(256), 0, st, W, nullptr, nullptr, packed);
                 hipLaunchKernelGGL((dense_split_pack_kernel<32, kSplitF16x2>), dim3(8, T), dim3(256), 0, st, W, nullptr, nullptr, p2); break;
        default: return GGNN_OK;
    }
    GGNN_CHECK_HIP(hipGetLastError());
    return GGNN_OK;
}

int dense_split_pack_gru(const float* Wg, const float* Wc, int D, float* packed, hipStream_t st) {
    float* p2 = packed + images_bytes(D, 6, kSplitBf16x3) / sizeof(float);
    switch (D) {
        case 100: hipLaunchKernelGGL((dense_split_pack_kernel<100, kSplitBf16x3>), dim3(8, 6), dim3(256), 0, st, nullptr, Wg, Wc, packed);
                  hipLaunchKernelGGL((dense_split_pack_kernel<100, kSplitF16x2>), dim3(8, 6), dim3(256), 0, st, nullptr, Wg, Wc, p2); break;
        case 64: hipLaunchKernelGGL((dense_split_pack_kernel<64, kSplitBf16x3>), dim3(8, 6), dim3(256), 0, st, nullptr, Wg, Wc, packed);
                 hipLaunchKernelGGL((dense_split_pack_kernel<64, kSplitF16x2>), dim3(8, 6), dim3(256), 0, st, nullptr, Wg, Wc, p2); break;
        case 32: hipLaunchKernelGGL((dense_split_pack_kernel<32, kSplitBf16x3>), dim3(8, 6), dim3(256), 0, st, nullptr, Wg, Wc, packed);
                 hipLaunchKernelGGL((dense_split_pack_kernel<32, kSplitF16x2>), dim3(8, 6), dim3(256), 0, st, nullptr, Wg, Wc, p2); break;
        default: return GGNN_OK;
    }
    GGNN_CHECK_HIP(hipGetLastError());
    return GGNN_OK;
}

template <int D, int E, int FMT>
static int launch_split(const DenseGraphArgs& a, hipStream_t st) {
    constexpr int NW = 8;
    const size_t ldsb = split_lds_bytes(D, E, SplitFmt<FMT>::NP);
    static std::atomic<unsigned long long> lds_ok{0};
    if (ldsb > 64 * 1024) GGNN_CHECK_HIP((allow_dynamic_lds(&ggnn_dense_graph_split_kernel<D, E, NW, FMT>, ldsb, lds_ok)));
    hipLaunchKernelGGL((ggnn_dense_graph_split_kernel<D, E, NW, FMT>), dim3(a.b), dim3(NW * 64), ldsb, st, a);
    GGNN_CHECK_HIP(hipGetLastError());
    return GGNN_OK;
}

// a.eimg / a.gimg: the images of the format `fmt` (dense_split_images_offset into the split section of the packed buffers)
int dense_split_launch(const DenseGraphArgs& a, int E, int D, int fmt, hipStream_t st) {
    const bool f2 = gru_launch_fmt(fmt) == kSplitF16x2;
#define GGNN_DGS_CASE(DD, EE) if (D == DD && E == EE) return f2 ? launch_split<DD, EE, kSplitF16x2>(a, st) : launch_split<DD, EE, kSplitBf16x3>(a, st);
    GGNN_DGS_CASE(100, 4) GGNN_DGS_CASE(100, 2) GGNN_DGS_CASE(100, 6) GGNN_DGS_CASE(100, 8)
    GGNN_DGS_CASE(64, 4) GGNN_DGS_CASE(64, 8) GGNN_DGS_CASE(64, 2) GGNN_DGS_CASE(64, 6)
    GGNN_DGS_CASE(32, 4) GGNN_DGS_CASE(32, 8) GGNN_DGS_CASE(32, 2) GGNN_DGS_CASE(32, 6)
#undef GGNN_DGS_CASE
    return fail(GGNN_E_UNSUPPORTED, "graph-resident dense forward (split form): unsupported shape");
}

}  // namespace ggnn
