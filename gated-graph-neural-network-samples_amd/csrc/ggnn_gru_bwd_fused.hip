// Backward of the GRU node update (TF autodiff of GRUCell, chem_tensorflow_sparse.py:211-216 through
// chem_tensorflow.py:184) as ONE launch -- the mirror image of ggnn_gru_fused.hip:
//
//   given g = dL/dh' and the saved h, r, u, c of the forward pass
//     dpc = g (1-u) act'(c)        dpu = g (h-c) u (1-u)                                (element-wise, in registers)
//     drh = dpc Wc^T[h rows]       dpr = drh h r (1-r)                                  (GEMM -> epilogue -> next operand)
//     dh  = g u + drh r + dpr Wg_r^T[h rows] + dpu Wg_u^T[h rows]
//     dx_s = dpc Wc^T[x_s rows] + dpr Wg_r^T[x_s rows] + dpu Wg_u^T[x_s rows]           s = 0 .. nx-1
//     d_incoming = dx_{nx-1} / (sum_t nin + 1e-7)                                       (mean aggregation, :206-209)
//   and, for the weight-gradient products that follow (ggnn_xty_f32), dpc, [dpr | dpu] and r*h are written once.
//
// Like the forward kernel it chains through registers: a wave owns 16 rows; dpc / dpr / dpu live as activation fragments
// (lane (row, kq) holds columns 16c + 4kq + e), the D x D blocks of Wc^T / Wg^T stream through a 2-deep LDS-DMA ring as
// k-interleaved stage images (ggnn_stage.hpp), and because output tile nt of the MFMA core IS activation chunk c = nt, drh
// comes out of its GEMM stage already in the layout the next stages consume (only the D % 16 remainder column needs
// cross-lane reads).  3 (nx + 1) stages per 16-row tile -- the same MFMA count as the forward pass -- replace the five
// launches of the first backward (element-wise stage 1, two dX GEMMs with epilogues, two passes of stage-2 algebra) and the
// [V,K] intermediates between them.
//
// Stage order per tile:   h block of Wc^T (-> drh, dpr);  h blocks of Wg_r^T, Wg_u^T (-> dh);
//                         then per x segment: Wc^T, Wg_r^T, Wg_u^T blocks (-> dx_s)
#include "ggnn_split.hpp"
#include <type_traits>

// 1: the stage products recompute the lane parts of their LDS addresses per call (stage_mma_split's REMAT): 52-72 -> 36 B of scratch
// per lane at D = 100, R = 0 launch 167.8 -> 162.9 us (round 4, profiles/r04_experiments/remat_and_streams.txt)
#ifndef GGNN_BWD_REMAT
#define GGNN_BWD_REMAT 1
#endif

namespace ggnn {

struct GruBwdArgs {
    const float* g; const float* h; const float* r; const float* u; const float* c;
    float* dpc; float* dpg; float* rh; float* dh;
    float* dx[3];                       // nx outputs [V,D]; the last one is d_incoming (scaled when use_avg)
    const float* nin; int T; int use_avg;
    int nx; int V; int act;
    // optional: g_eff[v] = g[v] + sum over the (up to four) rows gz_heads[v] names of gz -- the per-node sum that closes the
    // PREVIOUS timestep's transform backward (dh[v] += sum_t Z[row(v,t)]), taken on load here instead of by a launch of its own
    const float* gz; const int* gz_heads;
    unsigned long long* tdbg;           // debug: s_memtime stamps of workgroup 0 (GGNN_BWD_TPTR; tools/gru_bwd_timeline.py)
};

// image i of the packed backward weights (image[k][n] = B(k, n) with out[:, n] = sum_k A[:, k] B(k, n)):
//   0: Wc^T h block      B(k,n) = Wc[nx*D + n][k]
//   1: Wg_r^T h block    B(k,n) = Wg[nx*D + n][k]          2: Wg_u^T h block   B(k,n) = Wg[nx*D + n][D + k]
//   3 + 3s + {0,1,2}: the same three for x segment s (rows s*D + n)
template <int D, bool SPLIT>
__global__ void gru_bwd_pack_kernel(const float* __restrict__ Wg, const float* __restrict__ Wc, int nx, float* __restrict__ out) {
    const int i = blockIdx.y;
    float* img = out + (size_t)i * ImgCfg<D, SPLIT>::IMG;
    const int first = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
    if constexpr (SPLIT) gru_bwd_image_pack_split<D>(Wg, Wc, nx, i, img, first, stride);
    else gru_bwd_image_pack<D>(Wg, Wc, nx, i, img, first, stride);
}

template <int D>
__device__ __forceinline__ void store_frag(float* base, int row, int kq, const Frag<D>& f) {
    constexpr int NC = StageCfg<D>::NC, NR = StageCfg<D>::NR;
    const unsigned ob = ((unsigned)row * (unsigned)D + 4u * (unsigned)kq) * 4u;
#pragma unroll
    for (int c = 0; c < NC; ++c) st4_b(base, ob + 64u * c, f.v[c]);
#pragma unroll
    for (int q = 0; q < NR; ++q)
        *reinterpret_cast<float*>(reinterpret_cast<char*>(base) + ((unsigned)row * (unsigned)D + 16u * NC + 4u * q + (unsigned)kq) * 4u) = f.r[q];
}

// PREFETCH: 0 = a tile's inputs are fetched at the top of its pass; 1 = under the last stage of the previous pass (same registers);
// 2 = a WHOLE pass ahead, into a second register set, one fragment per stage (g, u, c, h, r under stages 0..4) -- for the 4-wave
// form (NW = 4: one wave per SIMD, so a wave may use the whole 512-entry register file of its SIMD lane; the compiler places
// ~200 values in the accumulation half).  MEASURED (tools/gru_bwd_bench.py, tools/gru_bwd_timeline.py, round 3): the head of a
// pass shrinks from 22-41k to 2k clocks, but a stage of ONE tile per SIMD is 5.6k clocks of matrix-pipe work and takes 7-10k:
// the barrier that closes a stage waits for the LDS-DMA of the next image and -- the load counter is in order -- for the
// fragment issued before it, and neither a 48 KiB image nor a 6.4 KB-per-wave HBM fragment lands in 5.6k clocks.  58k clocks per
// 4 tiles against 60.5k in form 0: 200 vs 205 us.  Staging the images through registers instead of LDS-DMA (exact counter waits)
// does not help either: the image's loads are younger than the fragment's, waiting for them waits for it (245 us, and the
// register file overflows: 276 B of scratch).  Hiding the loads needs a load path whose completion is not ordered with the
// weights' -- a loader wave would do, but a fifth wave halves the register budget of the other four.
// 3 = the inputs spread over the stages that can carry them (round 4): g, u, c of the NEXT tile under the last three stages of a pass
// (one fragment each: the f32 fragments of the current tile are dead there in split form, only their planes are live), h and r of the
// CURRENT tile under stage 0, which multiplies dpc = f(g, u, c) alone -- the rest of the element-wise head (dpu, g u, h r (1 - r), the
// r*h store) runs behind stage 0.  The chip's HBM share of a CU (6 TB/s / 256 = ~12 B per clock) makes the five fragments of a pass
// (280 KB per CU) a 17-35k-clock phase when every workgroup fetches them at the top of its pass; spread over four stages they ride
// under the MFMAs.
// SPLIT: the products on the bf16 matrix pipe in 3-way split form (ggnn_split.hpp); dpc, dpr, dpu are split once, when complete,
// and their planes serve all 1 + NX stages that multiply them.
template <int D, int NX, int NW, int PREFETCH, int RING, bool SPLIT>
__global__ __launch_bounds__(NW * 64, (RING == 1 && NW == 4) ? 2 : 1) void ggnn_gru_bwd_fused_kernel(GruBwdArgs a, const float* __restrict__ packed) {
    using C = StageCfg<D>;
    using I = ImgCfg<D, SPLIT>;
    constexpr int NT = C::NT, NC = C::NC, NR = C::NR;
    constexpr int NSTAGE = 3 * (NX + 1);
    extern __shared__ __attribute__((aligned(16))) float ring[];    // [2][IMG]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kq = lane >> 4;
    const bool late = wave >= NW / 2;

    // static tickets: full rounds of NW tiles per workgroup, then the rest spread thin (tail_w tiles per workgroup)
    const int wt_total = (a.V + 15) / 16;
    const int nb = gridDim.x;
    const int full_tk = wt_total / (NW * nb) * nb;
    const int rest = wt_total - full_tk * NW;
    const int tail_w = (rest + nb - 1) / nb;
    const int n_tk = full_tk + (tail_w ? (rest + tail_w - 1) / tail_w : 0);
    auto tile_of = [&](int t) -> int {
        if (t < full_tk) return t * NW + wave;
        if (t >= n_tk) return -1;
        const int tl = full_tk * NW + (t - full_tk) * tail_w + wave;
        return (wave < tail_w && tl < wt_total) ? tl : -1;
    };

    // Raw inputs of a tile (g, u, c, h, r as activation fragments).  They are fetched one pass AHEAD -- at the start of the
    // last stage of the previous pass, when only dpu of the current tile is still live -- so that no pass starts with an
    // exposed load phase; the first tile's are fetched before the loop.
    struct Raw { Frag<D> g, u, c, h, r; };
    auto fetch_piece = [&](Raw& x, int t, int piece) {        // piece 0..4: g, u, c, h, r
        const int tile = tile_of(t);
        const int r0 = (tile >= 0 ? tile : 0) * 16 + li;
        const int rc = r0 < a.V ? r0 : a.V - 1;
        if (piece == 0) load_frag<D>(x.g, a.g, rc, kq);
        else if (piece == 1) load_frag<D>(x.u, a.u, rc, kq);
        else if (piece == 2) load_frag<D>(x.c, a.c, rc, kq);
        else if (piece == 3) load_frag<D>(x.h, a.h, rc, kq);
        else load_frag<D>(x.r, a.r, rc, kq);
    };
    auto fetch_raw = [&](Raw& x, int t, int part) {           // part 0: g, u, c, h;  part 1: r
        if (part == 0) { fetch_piece(x, t, 0); fetch_piece(x, t, 1); fetch_piece(x, t, 2); fetch_piece(x, t, 3); }
        else fetch_piece(x, t, 4);
    };
    // value of accumulator-layout tile NC (the D % 16 remainder columns 16NC .. +3, held by the kq == 0 lanes) from a fragment:
    // column 16NC + e is the remainder element of lane (li, kq' = e)
    auto rem_tile = [&](const Frag<D>& f) -> f32x4 {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if constexpr (NR > 0) {
            v.x = __shfl(f.r[0], li); v.y = __shfl(f.r[0], li + 16); v.z = __shfl(f.r[0], li + 32); v.w = __shfl(f.r[0], li + 48);
        }
        return v;
    };
    static_assert(NR <= 1, "remainder handling covers D % 16 in {0, 4}");

    int cur = 0;
    auto dma = [&](const float* src, float* dst) {
        if constexpr (SPLIT) dma_image_asm<I::IMG_BYTES, NW>(src, dst, wave, lane);
        else dma_stage_image<D, NW>(src, dst, wave, lane);
    };
    auto publish = [&]() { if constexpr (SPLIT) dma_wait(); __syncthreads(); };
    if constexpr (RING == 2) dma(packed, ring);
    Raw raw_a, raw_b;                                            // (raw_b: PREFETCH == 2 only, the tile after the current one)
    if (PREFETCH == 3) { if ((int)blockIdx.x < n_tk) { fetch_piece(raw_a, blockIdx.x, 0); fetch_piece(raw_a, blockIdx.x, 1); fetch_piece(raw_a, blockIdx.x, 2); } }
    else if (PREFETCH && (int)blockIdx.x < n_tk) { fetch_raw(raw_a, blockIdx.x, 0); fetch_raw(raw_a, blockIdx.x, 1); }
    publish();

    // A wave has tiles in a PREFIX of its workgroup's passes (full tickets, then possibly a thin tail ticket), so the passes run
    // as two loops -- with a tile, without -- instead of one loop with a branch: the prefetched raw fragments are carried from
    // pass to pass, and a diamond in the loop body would put 125 registers through phi copies.
    int tk = blockIdx.x;
    int pass_no = 0;
#define GGNN_BT(K) if (a.tdbg && blockIdx.x == 0 && lane == 0 && (wave & 3) == 0 && pass_no < 8) \
        a.tdbg[(pass_no * 2 + (wave >> 2)) * 16 + (K)] = __builtin_amdgcn_s_memtime();
    auto run_pass = [&](auto active_c, Raw& raw, Raw& raw_next) {
        const int tile = tile_of(tk);
        const bool last_pass = tk + nb >= n_tk;
        GGNN_BT(0)
        {
            constexpr bool ACT = decltype(active_c)::value;      // (two pass bodies instead of a branch per stage: ggnn_panel.hip)
            const int row = (ACT ? tile : 0) * 16 + li;
            const bool row_ok = ACT && row < a.V;

            auto mma = [&](auto zero_c, f32x4 (&acc)[NT], const Frag<D>& A, const SFrag<D>& S, const float* img) {
                if constexpr (SPLIT) stage_mma_split<D, NT, decltype(zero_c)::value, GGNN_BWD_REMAT>(acc, S, A, img, li, kq);
                else stage_mma<D, NoHook, NT, decltype(zero_c)::value>(acc, A, img, li, kq);
            };
            // keep_c: vector-memory instructions the EARLY waves issue in after() (behind the image DMA) that may stay in flight across the
            // stage's barrier -- the load counter retires in order, so "at most keep outstanding" still proves the DMA pieces landed
            auto stage = [&](auto zero_c, f32x4 (&acc)[NT], const Frag<D>& A, const SFrag<D>& S, int img_idx, auto&& before, auto&& after,
                             auto keep_c) {
                constexpr int KEEP = decltype(keep_c)::value;
                const int nidx = img_idx + 1 < NSTAGE ? img_idx + 1 : 0;
                const bool more = (img_idx + 1 < NSTAGE) || !last_pass;
                const float* nsrc = packed + (size_t)nidx * I::IMG;
                float* ndst = ring + (cur ^ 1) * I::IMG;
                before();
                if constexpr (PREFETCH == 2) {
                    if (ACT && !last_pass && img_idx < 5) fetch_piece(raw_next, tk + nb, img_idx);
                }
                if constexpr (RING == 1) {
                    // one image in LDS (two of these 4-wave workgroups share a CU and run out of phase: the DMA wait and the
                    // load / epilogue phases of one are covered by the MFMAs of the other)
                    __syncthreads();                                   // the previous stage's image has been consumed
                    if constexpr (SPLIT) { dma_image_asm<I::IMG_BYTES, NW>(packed + (size_t)img_idx * I::IMG, ring, wave, lane); dma_wait(); }
                    else dma_stage_image<D, NW>(packed + (size_t)img_idx * I::IMG, ring, wave, lane);
                    __syncthreads();                                   // (vmcnt(0) + barrier: landed)
                    if constexpr (ACT) mma(zero_c, acc, A, S, ring);
                    after();
                } else {
                    if (late && more) dma(nsrc, ndst);
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (ACT) mma(zero_c, acc, A, S, ring + cur * I::IMG);
                    __builtin_amdgcn_sched_barrier(0);
                    if (!late && more) dma(nsrc, ndst);
                    after();
                    if constexpr (KEEP > 0 && SPLIT) {
                        static_assert(KEEP <= 7, "partial wait covers one fragment");
                        // (only when the early waves DID issue their fragment behind the DMA: the fetch_next_* condition)
                        if (!late && ACT && !last_pass) {
                            if constexpr (KEEP == 7) asm volatile("s_waitcnt vmcnt(7) lgkmcnt(0)" ::: "memory");
                            else if constexpr (KEEP == 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
                            else if constexpr (KEEP == 2) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
                            else asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory");
                            __builtin_amdgcn_s_barrier();
                        } else publish();
                    } else publish();
                    cur ^= 1;
                }
            };
            auto nothing = [] {};

            // ---- element-wise head, in place on the raw fragments: g -> dpc, c -> dpu, u -> g*u, h -> h*r*(1-r) (r stays) ----------
            Frag<D> dpc, dpu, gu, rf, hrr, dpr;
            SFrag<D> sdpc, sdpr, sdpu;                              // (SPLIT) their bf16 planes
            if constexpr (ACT) {
                // (gathered g) the head record first: it is the start of a dependent chain record -> rows
                int hz0 = -1, hz1 = -1, hz2 = -1, hz3 = -1;
                if (a.gz) {
                    const int rc = row < a.V ? row : a.V - 1;
                    const int4 hd = *reinterpret_cast<const int4*>(a.gz_heads + 4 * (size_t)rc);
                    hz0 = hd.x; hz1 = hd.y; hz2 = hd.z; hz3 = hd.w;
                }
                if constexpr (!PREFETCH) { fetch_raw(raw, tk, 0); fetch_raw(raw, tk, 1); }
                if constexpr (PREFETCH == 3) { fetch_piece(raw, tk, 3); fetch_piece(raw, tk, 4); }     // h, r: consumed behind stage 0
                if (a.gz) {
                    // the sums of ggnn_gather_segment_sum_heads_f32(accumulate = 1), in its order: ((0 + z0) + z1 + z2 + z3) + g; slots 2, 3
                    // exist for few nodes (a node with three or more edge types): a second round that most tiles skip as a wave
                    Frag<D> z0, z1, zs;
#pragma unroll
                    for (int cc = 0; cc < NC; ++cc) { z0.v[cc] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
                    for (int q = 0; q < NR; ++q) z0.r[q] = 0.f;
                    if (hz0 >= 0) load_frag<D>(z0, a.gz, hz0, kq);
                    if (hz1 >= 0) load_frag<D>(z1, a.gz, hz1, kq);
#pragma unroll
                    for (int cc = 0; cc < NC; ++cc) { zs.v[cc] = 0.f + z0.v[cc]; if (hz1 >= 0) zs.v[cc] = zs.v[cc] + z1.v[cc]; }
#pragma unroll
                    for (int q = 0; q < NR; ++q) { zs.r[q] = 0.f + z0.r[q]; if (hz1 >= 0) zs.r[q] = zs.r[q] + z1.r[q]; }
                    if (hz2 >= 0) {
                        load_frag<D>(z0, a.gz, hz2, kq);
                        if (hz3 >= 0) load_frag<D>(z1, a.gz, hz3, kq);
#pragma unroll
                        for (int cc = 0; cc < NC; ++cc) { zs.v[cc] = zs.v[cc] + z0.v[cc]; if (hz3 >= 0) zs.v[cc] = zs.v[cc] + z1.v[cc]; }
#pragma unroll
                        for (int q = 0; q < NR; ++q) { zs.r[q] = zs.r[q] + z0.r[q]; if (hz3 >= 0) zs.r[q] = zs.r[q] + z1.r[q]; }
                    }
#pragma unroll
                    for (int cc = 0; cc < NC; ++cc) raw.g.v[cc] = zs.v[cc] + raw.g.v[cc];
#pragma unroll
                    for (int q = 0; q < NR; ++q) raw.g.r[q] = zs.r[q] + raw.g.r[q];
                }
                auto dact = [&](float cv) { return a.act == GGNN_ACT_TANH ? 1.0f - cv * cv : (cv > 0.f ? 1.0f : 0.f); };
                const unsigned os = ((unsigned)row * (unsigned)D + 4u * (unsigned)kq) * 4u;
                if constexpr (PREFETCH == 3) {                  // dpc alone (same expressions as below); the rest behind stage 0
#pragma unroll
                    for (int cc = 0; cc < NC; ++cc) {
                        const f32x4 gv = raw.g.v[cc], uv = raw.u.v[cc], cv = raw.c.v[cc];
                        const f32x4 omu = 1.0f - uv;
                        const f32x4 da = {dact(cv.x), dact(cv.y), dact(cv.z), dact(cv.w)};
                        dpc.v[cc] = gv * omu * da;
                        if (row_ok) st4_b(a.dpc, os + 64u * cc, dpc.v[cc]);
                    }
#pragma unroll
                    for (int q = 0; q < NR; ++q) {
                        const float gv = raw.g.r[q], uv = raw.u.r[q], cv = raw.c.r[q];
                        const float omu = 1.0f - uv;
                        dpc.r[q] = gv * omu * dact(cv);
                        if (row_ok) {
                            const unsigned o2 = ((unsigned)row * (unsigned)D + 16u * NC + 4u * q + (unsigned)kq) * 4u;
                            *reinterpret_cast<float*>(reinterpret_cast<char*>(a.dpc) + o2) = dpc.r[q];
                        }
                    }
                } else {
#pragma unroll
                for (int cc = 0; cc < NC; ++cc) {
                    const f32x4 gv = raw.g.v[cc], uv = raw.u.v[cc], cv = raw.c.v[cc], hv = raw.h.v[cc], rv = raw.r.v[cc];
                    const f32x4 omu = 1.0f - uv;
                    const f32x4 da = {dact(cv.x), dact(cv.y), dact(cv.z), dact(cv.w)};
                    dpc.v[cc] = gv * omu * da;
                    dpu.v[cc] = gv * (hv - cv) * uv * omu;
                    gu.v[cc] = gv * uv;
                    rf.v[cc] = rv;
                    hrr.v[cc] = hv * rv * (1.0f - rv);
                    if (row_ok) {
                        st4_b(a.dpc, os + 64u * cc, dpc.v[cc]);
                        st4_b(a.rh, os + 64u * cc, rv * hv);         // the last segment of the [x | r*h] operand of dWc
                    }
                }
#pragma unroll
                for (int q = 0; q < NR; ++q) {
                    const float gv = raw.g.r[q], uv = raw.u.r[q], cv = raw.c.r[q], hv = raw.h.r[q], rv = raw.r.r[q];
                    const float omu = 1.0f - uv;
                    dpc.r[q] = gv * omu * dact(cv);
                    dpu.r[q] = gv * (hv - cv) * uv * omu;
                    gu.r[q] = gv * uv; rf.r[q] = rv; hrr.r[q] = hv * rv * (1.0f - rv);
                    if (row_ok) {
                        const unsigned o2 = ((unsigned)row * (unsigned)D + 16u * NC + 4u * q + (unsigned)kq) * 4u;
                        *reinterpret_cast<float*>(reinterpret_cast<char*>(a.dpc) + o2) = dpc.r[q];
                        *reinterpret_cast<float*>(reinterpret_cast<char*>(a.rh) + o2) = rv * hv;
                    }
                }
                }
            }

            if constexpr (SPLIT && ACT) { split_frag<D>(sdpc, dpc); if constexpr (PREFETCH != 3) split_frag<D>(sdpu, dpu); }
            // ---- stage 0: drh = dpc Wc^T[h block]; dpr = drh h r (1-r); dh_part = drh r  (accumulator layout) ----------------
            f32x4 acc[NT];
            GGNN_BT(1)
            stage(std::true_type{}, acc, dpc, sdpc, 0, nothing, nothing, std::integral_constant<int, 0>{});
            GGNN_BT(2)
            if constexpr (ACT && PREFETCH == 3) {               // the rest of the head: h and r have landed under stage 0
                auto dact = [&](float cv) { return a.act == GGNN_ACT_TANH ? 1.0f - cv * cv : (cv > 0.f ? 1.0f : 0.f); };
                const unsigned os = ((unsigned)row * (unsigned)D + 4u * (unsigned)kq) * 4u;
#pragma unroll
                for (int cc = 0; cc < NC; ++cc) {
                    const f32x4 gv = raw.g.v[cc], uv = raw.u.v[cc], cv = raw.c.v[cc], hv = raw.h.v[cc], rv = raw.r.v[cc];
                    const f32x4 omu = 1.0f - uv;
                    dpu.v[cc] = gv * (hv - cv) * uv * omu;
                    gu.v[cc] = gv * uv;
                    rf.v[cc] = rv;
                    hrr.v[cc] = hv * rv * (1.0f - rv);
                    if (row_ok) st4_b(a.rh, os + 64u * cc, rv * hv);
                }
#pragma unroll
                for (int q = 0; q < NR; ++q) {
                    const float gv = raw.g.r[q], uv = raw.u.r[q], cv = raw.c.r[q], hv = raw.h.r[q], rv = raw.r.r[q];
                    const float omu = 1.0f - uv;
                    dpu.r[q] = gv * (hv - cv) * uv * omu;
                    gu.r[q] = gv * uv; rf.r[q] = rv; hrr.r[q] = hv * rv * (1.0f - rv);
                    if (row_ok) {
                        const unsigned o2 = ((unsigned)row * (unsigned)D + 16u * NC + 4u * q + (unsigned)kq) * 4u;
                        *reinterpret_cast<float*>(reinterpret_cast<char*>(a.rh) + o2) = rv * hv;
                    }
                }
                if constexpr (SPLIT) split_frag<D>(sdpu, dpu);
            }
            if constexpr (ACT) {
                // accumulator tile nt == fragment chunk nt (same lanes, same columns); the remainder tile through rem_tile().
                // dpr goes to its fragment, and the accumulator is re-used as the START value of the dh stages:
                // acc = drh*r + g*u, onto which stages 1, 2 add dpr Wg_r^T + dpu Wg_u^T
#pragma unroll
                for (int nt = 0; nt < NC; ++nt) {
                    dpr.v[nt] = acc[nt] * hrr.v[nt];
                    acc[nt] = acc[nt] * rf.v[nt] + gu.v[nt];
                }
                if constexpr (NR > 0) {
                    // remainder column 16NC + kq of dpr lives in tile NC of lane (li, kq' = 0), element kq
                    const f32x4 dprt = acc[NT - 1] * rem_tile(hrr);
                    const float t0 = __shfl(dprt.x, li), t1 = __shfl(dprt.y, li), t2 = __shfl(dprt.z, li), t3 = __shfl(dprt.w, li);
                    dpr.r[0] = kq == 0 ? t0 : (kq == 1 ? t1 : (kq == 2 ? t2 : t3));
                    acc[NT - 1] = acc[NT - 1] * rem_tile(rf) + rem_tile(gu);
                }
                if (row_ok) {                                               // dpg = [dpr | dpu], row stride 2D
                    constexpr unsigned RS = 2u * D * 4u;
                    const unsigned ob = (unsigned)row * RS + 16u * (unsigned)kq;
#pragma unroll
                    for (int cc = 0; cc < NC; ++cc) {
                        st4_b(a.dpg, ob + 64u * cc, dpr.v[cc]);
                        st4_b(a.dpg, ob + D * 4u + 64u * cc, dpu.v[cc]);
                    }
#pragma unroll
                    for (int q = 0; q < NR; ++q) {
                        const unsigned o1 = (unsigned)row * RS + (16u * NC + 4u * q + (unsigned)kq) * 4u;
                        *reinterpret_cast<float*>(reinterpret_cast<char*>(a.dpg) + o1) = dpr.r[q];
                        *reinterpret_cast<float*>(reinterpret_cast<char*>(a.dpg) + o1 + D * 4u) = dpu.r[q];
                    }
                }
            }

            if constexpr (SPLIT && ACT) split_frag<D>(sdpr, dpr);
            // ---- stages 1, 2: the h blocks of Wg^T -> dh = g u + drh r + dpr Wg_r^T + dpu Wg_u^T ------------------------------------
            GGNN_BT(3)
            stage(std::false_type{}, acc, dpr, sdpr, 1, nothing, nothing, std::integral_constant<int, 0>{});
            GGNN_BT(4)
            stage(std::false_type{}, acc, dpu, sdpu, 2, nothing, nothing, std::integral_constant<int, 0>{});
            GGNN_BT(5)
            if constexpr (ACT) {
                if (row_ok) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const int col = nt * 16 + 4 * kq;
                        if (col < D) st4_b(a.dh, ((unsigned)row * D + col) * 4u, acc[nt]);
                    }
                }
            }

            // ---- x segments: dx_s = dpc Wc^T + dpr Wg_r^T + dpu Wg_u^T; the LAST stage starts with the next tile's raw fetch -------
            // (g, u, c, h before the burst of the last stage -- dpc, dpr are dead by then; r after it, when dpu is dead too:
            //  all five at once overflow the register file by ~60 values)
            GGNN_BT(6)
            auto fetch_next_a = [&] { if (PREFETCH == 1 && ACT && !last_pass) fetch_raw(raw, tk + nb, 0); };
            auto fetch_next_b = [&] { if (PREFETCH == 1 && ACT && !last_pass) fetch_raw(raw, tk + nb, 1); };
            // (PREFETCH == 3) g, u, c of the next tile, one under each of the pass's last three stages (a wave whose next pass has no
            // tile fetches row 0 of tile 0: harmless)
            // The 25 load instructions of a fragment take ~3k clocks to ISSUE when all eight waves issue theirs at once (the CU's
            // vector-memory path takes ~16 clocks per 1-KiB instruction), and a wave's MFMAs wait behind them in program order: the
            // late waves issue before their burst (under the early waves' burst), the early waves after theirs -- the stage
            // lambda's rule for the image DMA.
            auto fetch_next_g = [&] { if (PREFETCH == 3 && ACT && !last_pass && late) fetch_piece(raw, tk + nb, 0); };
            auto fetch_next_u = [&] { if (PREFETCH == 3 && ACT && !last_pass && late) fetch_piece(raw, tk + nb, 1); };
            auto fetch_next_c = [&] { if (PREFETCH == 3 && ACT && !last_pass && late) fetch_piece(raw, tk + nb, 2); };
            auto fetch_next_g2 = [&] { if (PREFETCH == 3 && ACT && !last_pass && !late) fetch_piece(raw, tk + nb, 0); };
            auto fetch_next_u2 = [&] { if (PREFETCH == 3 && ACT && !last_pass && !late) fetch_piece(raw, tk + nb, 1); };
            auto fetch_next_c2 = [&] { if (PREFETCH == 3 && ACT && !last_pass && !late) fetch_piece(raw, tk + nb, 2); };
#define GGNN_BWD_SEG(S)                                                                                             \
            if constexpr ((S) < NX) {                                                                               \
                if constexpr ((S) == NX - 1 && PREFETCH == 3) stage(std::true_type{}, acc, dpc, sdpc, 3 + 3 * (S), fetch_next_g, fetch_next_g2, std::integral_constant<int, (NC + NR == 7 || NC + NR == 4 || NC + NR == 2) ? NC + NR : 0>{}); \
                else stage(std::true_type{}, acc, dpc, sdpc, 3 + 3 * (S), nothing, nothing, std::integral_constant<int, 0>{});                        \
                if ((S) == 0) { GGNN_BT(8) }                                                                        \
                if constexpr ((S) == NX - 1 && PREFETCH == 3) stage(std::false_type{}, acc, dpr, sdpr, 4 + 3 * (S), fetch_next_u, fetch_next_u2, std::integral_constant<int, (NC + NR == 7 || NC + NR == 4 || NC + NR == 2) ? NC + NR : 0>{}); \
                else stage(std::false_type{}, acc, dpr, sdpr, 4 + 3 * (S), nothing, nothing, std::integral_constant<int, 0>{});                       \
                if ((S) == 0) { GGNN_BT(9) }                                                                        \
                if constexpr ((S) == NX - 1 && PREFETCH == 3) stage(std::false_type{}, acc, dpu, sdpu, 5 + 3 * (S), fetch_next_c, fetch_next_c2, std::integral_constant<int, (NC + NR == 7 || NC + NR == 4 || NC + NR == 2) ? NC + NR : 0>{}); \
                else if constexpr ((S) == NX - 1) stage(std::false_type{}, acc, dpu, sdpu, 5 + 3 * (S), fetch_next_a, fetch_next_b, std::integral_constant<int, 0>{}); \
                else stage(std::false_type{}, acc, dpu, sdpu, 5 + 3 * (S), nothing, nothing, std::integral_constant<int, 0>{});                       \
                if ((S) == 0) { GGNN_BT(10) }                                                                       \
                if constexpr (ACT) {                                                                                \
                    if (row_ok) {                                                                                   \
                        float den = 1.0f;                                                                           \
                        if ((S) == NX - 1 && a.use_avg) {                                                           \
                            float deg = 0.f;                                                                        \
                            for (int t = 0; t < a.T; ++t) deg += a.nin[(size_t)row * a.T + t];                      \
                            den = deg + 1e-7f;                                                                      \
                        }                                                                                           \
                        _Pragma("unroll")                                                                           \
                        for (int nt = 0; nt < NT; ++nt) {                                                           \
                            const int col = nt * 16 + 4 * kq;                                                       \
                            if (col < D) {                                                                          \
                                f32x4 v = acc[nt];                                                                  \
                                if ((S) == NX - 1 && a.use_avg) v = v / den;                                        \
                                st4_b(a.dx[(S)], ((unsigned)row * D + col) * 4u, v);                                \
                            }                                                                                       \
                        }                                                                                           \
                    }                                                                                               \
                }                                                                                                   \
            }
            GGNN_BWD_SEG(0) GGNN_BWD_SEG(1) GGNN_BWD_SEG(2)
#undef GGNN_BWD_SEG
            GGNN_BT(7)
        }
        ++pass_no;
    };
    if constexpr (PREFETCH == 2) {
        // the two register sets swap roles from pass to pass: the loop is unrolled by two instead of copying 125 registers
        while (tk < n_tk && tile_of(tk) >= 0) {
            run_pass(std::true_type{}, raw_a, raw_b);
            tk += nb;
            if (!(tk < n_tk && tile_of(tk) >= 0)) break;
            run_pass(std::true_type{}, raw_b, raw_a);
            tk += nb;
        }
    } else {
        for (; tk < n_tk && tile_of(tk) >= 0; tk += nb) run_pass(std::true_type{}, raw_a, raw_a);
    }
    for (; tk < n_tk; tk += nb) run_pass(std::false_type{}, raw_a, raw_a);
}

// (A "pipelined" form -- g, u, c of the next tile fetched under the last three stages, h and r under stage 0, every store issued in
//  front of a stage -- was measured at 211 us against 201 us for the kernel above: 100 B of scratch, and the waits only move to where
//  the data is needed.  The exposed load phase is bandwidth, not latency: 8 waves x 5 fragments = 280 KB per CU and pass at the CU's
//  share of HBM (6 TB/s / 256 CUs = 10 B/clock) is 28k clocks; hiding it takes a full pass of prefetch distance, i.e. 125 more
//  registers per wave or 256 KB of LDS per CU.  tools/gru_bwd_timeline.py, tools/gru_bwd_bench.py)

template <int D, int NX, int NW, int PREFETCH, int RING, bool SPLIT = false>
static int launch_gru_bwd_variant(const GruBwdArgs& a, const float* packed, hipStream_t st) {
    using C = ImgCfg<D, SPLIT>;
    const size_t lds = (size_t)RING * C::IMG_BYTES;
    const int wt_total = (a.V + 15) / 16;
    int nb = num_cus() * (RING == 1 ? 2 : 1);                  // single-image form: two 4-wave workgroups per CU
    if (nb > wt_total) nb = wt_total;
    static std::atomic<unsigned long long> lds_ok{0};
    if (lds > 64 * 1024) GGNN_CHECK_HIP((allow_dynamic_lds(&ggnn_gru_bwd_fused_kernel<D, NX, NW, PREFETCH, RING, SPLIT>, lds, lds_ok)));
    hipLaunchKernelGGL((ggnn_gru_bwd_fused_kernel<D, NX, NW, PREFETCH, RING, SPLIT>), dim3(nb), dim3(NW * 64), lds, st, a, packed);
    GGNN_CHECK_HIP(hipGetLastError());
    return GGNN_OK;
}

// This source is compiled twice, like ggnn_gru_fused.hip: as itself (f32 MFMA) and through ggnn_gru_bwd_fused_split.hip
// (GGNN_GRU_BWD_TU_SPLIT: the split instantiations, built without packed-f32 vector instructions).
int gru_bwd_split_launch(int D, int nx, const GruBwdArgs& a, const float* Wg, const float* Wc, float* packed, hipStream_t st);

template <int D, int NX, bool SPLIT>
static int gru_bwd_prepare(const GruBwdArgs& a, const float* Wg, const float* Wc, float* packed, hipStream_t st, bool* done) {
    *done = true;
    if (Wg) {
        hipLaunchKernelGGL((gru_bwd_pack_kernel<D, SPLIT>), dim3(8, 3 * (NX + 1)), dim3(256), 0, st, Wg, Wc, NX, packed);
        GGNN_CHECK_HIP(hipGetLastError());
    }
    if (a.g == nullptr || a.V == 0) return GGNN_OK;
    if ((unsigned long long)a.V * 2 * D >= (1ULL << 30))
        return fail(GGNN_E_UNSUPPORTED, "fused GRU backward indexes with 32-bit byte offsets: V*2D must be < 2^30 (V=%d, D=%d)", a.V, D);
    *done = false;
    return GGNN_OK;
}

#ifdef GGNN_GRU_BWD_TU_SPLIT
template <int D, int NX>
static int launch_gru_bwd_split(const GruBwdArgs& a, const float* Wg, const float* Wc, float* packed, hipStream_t st) {
    bool done;
    if (int rc = gru_bwd_prepare<D, NX, true>(a, Wg, Wc, packed, st, &done); rc != GGNN_OK || done) return rc;
    // GGNN_BWD_FORM=1: the next tile's inputs prefetched under the last stage; 2: two 4-wave workgroups per CU with one image each
    // (round 4, MI355X, V = 100k: 164.7 us form 0, 163.8 form 2, 189.5 form 1; starting every other workgroup 5-25 us late moves
    // the launch time by -2 .. +4 us: with half the chip idle the rest clocks higher -- the launch is paced by the chip's power
    // budget, not by how its phases line up; profiles/r04_experiments/gru_bwd_forms_stagger.txt)
    // 3 (default since round 4): the inputs spread over the stages -- g, u, c of the next tile under the last three stages (early waves:
    // behind their burst, left in flight across the barrier), h and r under stage 0: 150.8 us against 162.7 for form 0, bit-identical,
    // 20 B of scratch instead of 36; what is left is the launch's 440 MB at ~3.8 TB/s inside the full rounds.  0: everything at the top.
    const int form = [] { const char* e = getenv("GGNN_BWD_FORM"); return e ? atoi(e) : 3; }();
    if (form == 1) return launch_gru_bwd_variant<D, NX, 8, 1, 2, true>(a, packed, st);
    if (form == 2) return launch_gru_bwd_variant<D, NX, 4, 0, 1, true>(a, packed, st);     // two 4-wave workgroups per CU, one image each
    if (form == 0) return launch_gru_bwd_variant<D, NX, 8, 0, 2, true>(a, packed, st);
    return launch_gru_bwd_variant<D, NX, 8, 3, 2, true>(a, packed, st);
}
template <int D>
static int gru_bwd_split_d(int nx, const GruBwdArgs& a, const float* Wg, const float* Wc, float* packed, hipStream_t st) {
    if constexpr (SplitCfg<D>::OK) {
        switch (nx) {
            case 1: return launch_gru_bwd_split<D, 1>(a, Wg, Wc, packed, st);
            case 2: return launch_gru_bwd_split<D, 2>(a, Wg, Wc, packed, st);
            case 3: return launch_gru_bwd_split<D, 3>(a, Wg, Wc, packed, st);
        }
    }
    return fail(GGNN_E_INVALID, "nx %d outside 1..3", nx);
}
int gru_bwd_split_launch(int D, int nx, const GruBwdArgs& a, const float* Wg, const float* Wc, float* packed, hipStream_t st) {
#ifdef GGNN_PROBE_NX
    return launch_gru_bwd_split<100, GGNN_PROBE_NX>(a, Wg, Wc, packed, st);
#else
    switch (D) {
        case 100: return gru_bwd_split_d<100>(nx, a, Wg, Wc, packed, st);
        case 64: return gru_bwd_split_d<64>(nx, a, Wg, Wc, packed, st);
        case 32: return gru_bwd_split_d<32>(nx, a, Wg, Wc, packed, st);
        default: return fail(GGNN_E_UNSUPPORTED, "no split-form fused GRU backward for hidden size %d", D);
    }
#endif
}
#else
template <int D, int NX>
static int launch_gru_bwd(const GruBwdArgs& a, const float* Wg, const float* Wc, float* packed, hipStream_t st) {
    if (SplitCfg<D>::OK && split_matrix_path()) return gru_bwd_split_launch(D, NX, a, Wg, Wc, packed, st);
    bool done;
    if (int rc = gru_bwd_prepare<D, NX, false>(a, Wg, Wc, packed, st, &done); rc != GGNN_OK || done) return rc;
    // GGNN_BWD_FORM: 0 = one 8-wave workgroup per CU, 2-image ring, inputs fetched at the top of a pass;
    //                1 = the same with the next tile's inputs prefetched under the last stage;
    //                2 = two 4-wave workgroups per CU, one image each
    //                3 = one 4-wave workgroup per CU (a wave per SIMD, 256 + ~200 registers), the next tile's inputs fetched a
    //                    whole pass ahead into a second register set, one fragment per stage (see the kernel's header)
    const int form = [] { const char* e = getenv("GGNN_BWD_FORM"); return e ? atoi(e) : 0; }();   // (read per call: tools/gru_bwd_bench.py)
    if (form == 1) return launch_gru_bwd_variant<D, NX, 8, 1, 2>(a, packed, st);
    if (form == 2) return launch_gru_bwd_variant<D, NX, 4, 0, 1>(a, packed, st);
    if (form == 3) return launch_gru_bwd_variant<D, NX, 4, 2, 2>(a, packed, st);
    return launch_gru_bwd_variant<D, NX, 8, 0, 2>(a, packed, st);
}

template <int D>
static int dispatch_gru_bwd(const GruBwdArgs& a, const float* Wg, const float* Wc, float* packed, hipStream_t st) {
    switch (a.nx) {
        case 1: return launch_gru_bwd<D, 1>(a, Wg, Wc, packed, st);
        case 2: return launch_gru_bwd<D, 2>(a, Wg, Wc, packed, st);
        case 3: return launch_gru_bwd<D, 3>(a, Wg, Wc, packed, st);
        default: return fail(GGNN_E_INVALID, "nx %d outside 1..3", a.nx);
    }
}

#endif   // !GGNN_GRU_BWD_TU_SPLIT

}  // namespace ggnn

#ifndef GGNN_GRU_BWD_TU_SPLIT
using namespace ggnn;

extern "C" int ggnn_gru_bwd_is_fused(int D) { return D == 100 || D == 64 || D == 32; }

extern "C" size_t ggnn_gru_bwd_packed_bytes(int D, int nx) {
    size_t img = 0;
    const bool sp = split_matrix_path();
    switch (D) {
        case 100: img = sp ? ImgCfg<100, true>::IMG : ImgCfg<100, false>::IMG; break;
        case 64: img = sp ? ImgCfg<64, true>::IMG : ImgCfg<64, false>::IMG; break;
        case 32: img = sp ? ImgCfg<32, true>::IMG : ImgCfg<32, false>::IMG; break;
        default: return 0;
    }
    return (size_t)3 * (nx + 1) * img * sizeof(float);
}

static int gru_bwd_fused_impl(const float* g, const float* gz, const int32_t* gz_heads, const float* h, const float* r, const float* u,
                              const float* c, const float* Wg, const float* Wc, float* packed, float* dpc, float* dpg, float* rh, float* dh,
                              float* const* dx, const float* nin, int T, int use_avg, int nx, int V, int D, int act, ggnn_stream_t stream);

// Wg / Wc given: (re)build the packed transposed-block images into `packed` first; g == NULL: pack only.
extern "C" int ggnn_gru_bwd_fused_f32(const float* g, const float* h, const float* r, const float* u, const float* c, const float* Wg,
                                      const float* Wc, float* packed, float* dpc, float* dpg, float* rh, float* dh,
                                      float* const* dx, const float* nin, int T, int use_avg, int nx, int V, int D, int act,
                                      ggnn_stream_t stream) {
    return gru_bwd_fused_impl(g, nullptr, nullptr, h, r, u, c, Wg, Wc, packed, dpc, dpg, rh, dh, dx, nin, T, use_avg, nx, V, D, act, stream);
}

// ... with g[v] + sum of the rows gz_heads[v] (an int4 slot-head record per node, -1 = no such slot) names of gz as the incoming
// gradient: ggnn_gather_segment_sum_heads_f32(gz, .., heads, accumulate = 1) into g followed by ggnn_gru_bwd_fused_f32, bit for bit,
// for nodes with at most four rows (more: the caller runs the stand-alone sum).  Pre-packed weights only.
extern "C" int ggnn_gru_bwd_fused_gather_f32(const float* g, const float* gz, const int32_t* gz_heads, const float* h, const float* r,
                                             const float* u, const float* c, float* packed, float* dpc, float* dpg, float* rh, float* dh,
                                             float* const* dx, const float* nin, int T, int use_avg, int nx, int V, int D, int act,
                                             ggnn_stream_t stream) {
    GGNN_CHECK_ARG(g && gz && gz_heads && aligned16(gz) && aligned16(gz_heads), "gathered gradient: null or misaligned pointer");
    return gru_bwd_fused_impl(g, gz, gz_heads, h, r, u, c, nullptr, nullptr, packed, dpc, dpg, rh, dh, dx, nin, T, use_avg, nx, V, D, act, stream);
}

static int gru_bwd_fused_impl(const float* g, const float* gz, const int32_t* gz_heads, const float* h, const float* r, const float* u, const float* c, const float* Wg,
                                      const float* Wc, float* packed, float* dpc, float* dpg, float* rh, float* dh,
                                      float* const* dx, const float* nin, int T, int use_avg, int nx, int V, int D, int act,
                                      ggnn_stream_t stream) {
    GGNN_CHECK_ARG(V >= 0 && nx >= 1 && nx <= 3, "bad sizes V=%d nx=%d", V, nx);
    if (!ggnn_gru_bwd_is_fused(D)) return fail(GGNN_E_UNSUPPORTED, "no fused GRU backward for hidden size %d", D);
    GGNN_CHECK_ARG(act == GGNN_ACT_TANH || act == GGNN_ACT_RELU, "unknown activation %d", act);
    GGNN_CHECK_ARG(packed && aligned16(packed) && (!Wg || Wc), "packed weights missing");
    GruBwdArgs a{};
    a.nx = nx; a.V = V; a.act = act; a.T = T; a.use_avg = use_avg; a.nin = nin;
    { const char* e = getenv("GGNN_BWD_TPTR"); a.tdbg = e ? (unsigned long long*)strtoull(e, nullptr, 10) : nullptr; }
    if (g) {
        GGNN_CHECK_ARG(h && r && u && c && dpc && dpg && rh && dh && dx && (!use_avg || nin), "null pointer");
        GGNN_CHECK_ARG(aligned16(g) && aligned16(h) && aligned16(r) && aligned16(u) && aligned16(c) && aligned16(dpc) && aligned16(dpg) &&
                       aligned16(rh) && aligned16(dh), "pointers must be 16-byte aligned");
        a.g = g; a.gz = gz; a.gz_heads = gz_heads; a.h = h; a.r = r; a.u = u; a.c = c; a.dpc = dpc; a.dpg = dpg; a.rh = rh; a.dh = dh;
        for (int s = 0; s < nx; ++s) { GGNN_CHECK_ARG(dx[s] && aligned16(dx[s]), "dx[%d] null or misaligned", s); a.dx[s] = dx[s]; }
        // (experiments, tools/gru_bwd_bench.py: wrong results on purpose) bit 0: every [V,D] output aliases dpc; bit 1: every input aliases g
        if (const char* e = getenv("GGNN_BWD_ALIAS")) {
            const int m = atoi(e);
            if (m & 1) { a.rh = a.dpc; a.dh = a.dpc; for (int s = 0; s < nx; ++s) a.dx[s] = a.dpc; }
            if (m & 2) { a.h = a.g; a.r = a.g; a.u = a.g; a.c = a.g; }
        }
    }
    hipStream_t st = (hipStream_t)stream;
    switch (D) {
        case 100: return dispatch_gru_bwd<100>(a, Wg, Wc, packed, st);
        case 64: return dispatch_gru_bwd<64>(a, Wg, Wc, packed, st);
        default: return dispatch_gru_bwd<32>(a, Wg, Wc, packed, st);
    }
}
#endif
