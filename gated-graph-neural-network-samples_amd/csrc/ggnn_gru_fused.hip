// K3 fused: the whole TF-1.3 GRUCell node update (chem_tensorflow_sparse.py:211-216) in ONE launch:
//     [r|u] = sigmoid([x|h] Wg + bg);  c = act([x | r*h] Wc + bc);  h' = u*h + (1-u)*c
//
// Why it chains without LDS or shuffles: in the GEMM core (ggnn_gemm.hpp) lane (row = lane&15, kq = lane>>4)
// holds output columns 16*nt + 4*kq + e of its row in accumulator tile nt -- and consumes the k indices
// 16*c + 4*kq + e of its row from activation chunk c.  Output tile nt IS activation chunk c = nt.  So
// r*h (gates output x resident h fragment) feeds the candidate GEMM straight from registers, and u, h, c
// meet in the blend in the same layout.  Only the D%16 remainder columns need 4 cross-lane reads.
// Per 16-row wave tile nothing but x, h are read and h' written: the r*h / u round trip of the two-launch
// version (4 x [V,D] fp32 of HBM traffic) and one launch boundary are gone.
//
// Weights: a tiny pre-pass (gru_pack_weights_kernel) rewrites Wg / Wc into per-stage LDS images
// (ggnn_stage.hpp; stage = one D x D block: the [x_s|h] rows of the r columns, of the u columns, then of
// Wc).  Stages are brought in by LDS-DMA into a 2-deep ring; the DMA of stage i+1 is issued before the
// MFMAs of stage i.  Workgroups are persistent over a contiguous, evenly split range of 16-row tiles.
#include "ggnn_stage.hpp"

namespace ggnn {

int gru_pack_floats(int D, int nx) {
    switch (D) {
        case 100: return 3 * (nx + 1) * StageCfg<100>::IMG;
        case 64: return 3 * (nx + 1) * StageCfg<64>::IMG;
        case 32: return 3 * (nx + 1) * StageCfg<32>::IMG;
        default: return 0;
    }
}

// stage ci of the sequence: gates (s = 0..nx) x {r,u}, then candidate (s = 0..nx)
template <int D>
__global__ void gru_pack_weights_kernel(const float* __restrict__ Wg, const float* __restrict__ Wc, int nx,
                                        float* __restrict__ out) {
    const int ci = blockIdx.y;
    const float* W; int r0, c0, ldw;
    if (ci < 2 * (nx + 1)) { W = Wg; r0 = (ci >> 1) * D; c0 = (ci & 1) * D; ldw = 2 * D; }
    else { W = Wc; r0 = (ci - 2 * (nx + 1)) * D; c0 = 0; ldw = D; }
    pack_stage_image<D>(W, r0, c0, ldw, out + (size_t)ci * StageCfg<D>::IMG, blockIdx.x * blockDim.x + threadIdx.x,
                        gridDim.x * blockDim.x);
}

// SAVE: also write r, u, c (all three) for a backward pass.
// GATHER: the LAST x segment (the aggregated messages, chem_tensorflow_sparse.py:198-212) is not read from memory but
// computed on the fly: incoming[row] = (sum over the row's message slots of Hrows[gather_row[slot]]) / (deg + 1e-7),
// same slot order and arithmetic as ggnn_gather_segment_sum_f32 (bit-identical), fetched in the shadow of the previous
// stage's MFMAs -- the separate segment-sum launch and the [V,D] round trip of `incoming` through HBM disappear.
template <int D, int NX, int NW, bool SAVE, bool GATHER>
__global__ __launch_bounds__(NW * 64) void ggnn_gru_fused_kernel(GruFusedArgs a, const float* __restrict__ packed) {
    using C = StageCfg<D>;
    constexpr int NT = C::NT, NC = C::NC, NR = C::NR;
    constexpr int NSTAGE = 3 * (NX + 1);
    constexpr int DMA_PER_WAVE = C::IMG_BYTES / (NW * 1024);
    static_assert(DMA_PER_WAVE <= C::NC * ((NT + 3) / 4), "more DMA pieces than MFMA groups to hide them behind");
    extern __shared__ __attribute__((aligned(16))) float ring[];    // [2][IMG]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kq = lane >> 4;

    const int wt_total = (a.V + 15) / 16;
    const int nb = gridDim.x;
    const int t_beg = (int)(((long long)wt_total * blockIdx.x) / nb);
    const int t_end = (int)(((long long)wt_total * (blockIdx.x + 1)) / nb);
    // passes of THIS workgroup: one with fewer tiles leaves early and frees its CU (LDS + registers) for
    // whatever the other streams have queued, instead of idling through the tail pass of its neighbours
    const int passes = (t_end - t_beg + NW - 1) / NW;

    // biases live in LDS behind the ring: the epilogues read them with ds_read instead of 7 serialized
    // global round trips per epilogue
    float* bias_s = ring + 2 * C::IMG;                              // [bg (2D) | bc (D)]
    for (int i = tid; i < 3 * D; i += NW * 64) bias_s[i] = i < 2 * D ? a.bg[i] : a.bc[i - 2 * D];

    int cur = 0;
    dma_stage_image<D, NW>(packed, ring, wave, lane);

    // x fragments rotate through two register sets (the segment after the current one is prefetched);
    // with more than one x segment they are re-read for the candidate GEMM rather than kept resident.
    // The NEXT pass's x[0] fragment (needed by its very first MFMA) is fetched during the last stage of the
    // current pass and is drained by that stage's barrier, so stage 0 starts without a vmcnt wait (which
    // would also drain the freshly issued weight DMA); h is loaded at the top of the pass and is not needed
    // before the third stage.
    auto gather_frag = [&](Frag<D>& f, int r) {
#pragma unroll
        for (int c = 0; c < NC; ++c) f.v[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < NR; ++q) f.r[q] = 0.f;
        const int beg = a.g_row_ptr[r], end = a.g_row_ptr[r + 1];
        for (int e = beg; e < end; ++e) {                       // slot order == reference accumulation order
            Frag<D> t;
            load_frag<D>(t, a.g_H, a.g_idx[e], kq);
#pragma unroll
            for (int c = 0; c < NC; ++c) f.v[c] += t.v[c];
#pragma unroll
            for (int q = 0; q < NR; ++q) f.r[q] += t.r[q];
        }
        if (a.g_use_avg) {                                      // :206-209
            float deg = 0.f;
            for (int t = 0; t < a.g_T; ++t) deg += a.g_nin[(size_t)r * a.g_T + t];
            const float den = deg + 1e-7f;
#pragma unroll
            for (int c = 0; c < NC; ++c) f.v[c] = f.v[c] / den;
#pragma unroll
            for (int q = 0; q < NR; ++q) f.r[q] = f.r[q] / den;
        }
    };
#define GGNN_LOAD_X(F, S, R) { if constexpr (GATHER && (S) == NX - 1) gather_frag(F, R); else load_frag<D>(F, a.x[S], R, kq); }
    Frag<D> hf, xf[2];
    if (t_beg + wave < t_end) {
        const int r0 = (t_beg + wave) * 16 + li;
        GGNN_LOAD_X(xf[0], 0, r0 < a.V ? r0 : a.V - 1)
    }
    __syncthreads();          // (drains the DMA: hipcc emits vmcnt(0) before the barrier while an LDS-DMA is in flight)

    for (int p = 0; p < passes; ++p) {
        const int tile = t_beg + p * NW + wave;
        const bool active = tile < t_end;                      // wave-uniform
        const int row = tile * 16 + li;
        const int rowc = active ? (row < a.V ? row : a.V - 1) : 0;
        const bool last_pass = (p + 1 == passes);
        if (active) load_frag<D>(hf, a.h, rowc, kq);

#define GGNN_T(CI, K) if (a.tdbg && blockIdx.x == 0 && lane == 0) a.tdbg[((p * NSTAGE + (CI)) * NW + wave) * 4 + (K)] = __builtin_amdgcn_s_memtime();
        // one stage: start the DMA of the next image, MFMAs on the current one, publish
#define GGNN_STAGE(CI, ACC, FRAG)                                                                        \
        {                                                                                                \
            constexpr int nci_ = (CI) + 1;                                                               \
            const bool more_ = (nci_ < NSTAGE) || !last_pass;                                            \
            const float* nsrc_ = packed + (size_t)(nci_ < NSTAGE ? nci_ : 0) * C::IMG;                   \
            float* ndst_ = ring + (cur ^ 1) * C::IMG;                                                    \
            GGNN_T(CI, 0)                                                                                \
            /* whole next image up front (spreading the 6 DMA instructions over the MFMA groups via the   \
               stage_mma hook measured slower: 152 vs 146 us at nx=1, 323 vs 277 us at nx=3) */          \
            /* (letting only one wave per SIMD pair issue the DMA measured no better: 150 vs 147 us) */  \
            if (more_) dma_stage_image<D, NW>(nsrc_, ndst_, wave, lane);                                 \
            GGNN_T(CI, 1)                                                                                \
            __builtin_amdgcn_sched_barrier(0);   /* keep the DMA issue AHEAD of the MFMA block */        \
            if (active && !(a.dbg & 1)) stage_mma<D>(ACC, FRAG, ring + cur * C::IMG, li, kq);            \
            GGNN_T(CI, 2)                                                                                \
            __syncthreads();                                                                             \
            GGNN_T(CI, 3)                                                                                \
            cur ^= 1;                                                                                    \
        }

        // ---- gates: [x | h] Wg, r columns then u columns of each K segment --------------------------
        f32x4 acc_r[NT], acc_u[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) { acc_r[nt] = f32x4{0.f, 0.f, 0.f, 0.f}; acc_u[nt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        if constexpr (NX >= 2) { if (active) GGNN_LOAD_X(xf[1], 1, rowc) }
        GGNN_STAGE(0, acc_r, xf[0]) GGNN_STAGE(1, acc_u, xf[0])
        if constexpr (NX >= 2) {
            if constexpr (NX >= 3) { if (active) GGNN_LOAD_X(xf[0], 2, rowc) }
            GGNN_STAGE(2, acc_r, xf[1]) GGNN_STAGE(3, acc_u, xf[1])
        }
        if constexpr (NX >= 3) { GGNN_STAGE(4, acc_r, xf[0]) GGNN_STAGE(5, acc_u, xf[0]) }
        if constexpr (NX >= 2) { if (active) GGNN_LOAD_X(xf[0], 0, rowc) }   // for the candidate GEMM
        GGNN_STAGE(2 * NX, acc_r, hf)
        GGNN_STAGE(2 * NX + 1, acc_u, hf)

        // ---- r = sigmoid(.), u = sigmoid(.), rh = r*h in activation-fragment layout -------------------
        Frag<D> rh;
        if (active && !(a.dbg & 2)) {
            stage_tail_reduce<D>(acc_r);                       // VALU-tail columns: add the four kq partials
            stage_tail_reduce<D>(acc_u);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int col = nt * 16 + 4 * kq;
                if (col < D) {
                    const f32x4 br = ld4(bias_s + col), bu = ld4(bias_s + D + col);
                    f32x4 r, u;
                    r.x = sigmoid_f(acc_r[nt].x + br.x); r.y = sigmoid_f(acc_r[nt].y + br.y);
                    r.z = sigmoid_f(acc_r[nt].z + br.z); r.w = sigmoid_f(acc_r[nt].w + br.w);
                    u.x = sigmoid_f(acc_u[nt].x + bu.x); u.y = sigmoid_f(acc_u[nt].y + bu.y);
                    u.z = sigmoid_f(acc_u[nt].z + bu.z); u.w = sigmoid_f(acc_u[nt].w + bu.w);
                    acc_r[nt] = r; acc_u[nt] = u;
                    if constexpr (SAVE) {
                        if (row < a.V) {
                            st4(a.save_r + ((unsigned)row * D + col), r);
                            st4(a.save_u + ((unsigned)row * D + col), u);
                        }
                    }
                }
            }
#pragma unroll
            for (int c = 0; c < NC; ++c) rh.v[c] = acc_r[c] * hf.v[c];
#pragma unroll
            for (int q = 0; q < NR; ++q) {
                // remainder column 16NC + 4q + kq lives in tile NC of lane (li, kq' = q), element e = kq
                const float t0 = __shfl(acc_r[NT - 1].x, li + 16 * q), t1 = __shfl(acc_r[NT - 1].y, li + 16 * q);
                const float t2 = __shfl(acc_r[NT - 1].z, li + 16 * q), t3 = __shfl(acc_r[NT - 1].w, li + 16 * q);
                const float rr = kq == 0 ? t0 : (kq == 1 ? t1 : (kq == 2 ? t2 : t3));
                rh.r[q] = rr * hf.r[q];
            }
        }

        // ---- candidate: [x | r*h] Wc --------------------------------------------------------------------
        f32x4 acc_c[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc_c[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (NX >= 2) { if (active) GGNN_LOAD_X(xf[1], 1, rowc) }
        GGNN_STAGE(2 * NX + 2, acc_c, xf[0])
        if constexpr (NX >= 2) {
            if constexpr (NX >= 3) { if (active) GGNN_LOAD_X(xf[0], 2, rowc) }
            GGNN_STAGE(2 * NX + 3, acc_c, xf[1])
        }
        if constexpr (NX >= 3) { GGNN_STAGE(2 * NX + 4, acc_c, xf[0]) }
        {   // next pass's fragments ride under the last stage's MFMAs (xf[0] is dead from here on)
            const int tile_n = tile + NW;
            if (!last_pass && tile_n < t_end) {
                const int rn = tile_n * 16 + li;
                GGNN_LOAD_X(xf[0], 0, rn < a.V ? rn : a.V - 1)
            }
        }
        GGNN_STAGE(3 * NX + 2, acc_c, rh)
#undef GGNN_STAGE

        // ---- c = act(.), h' = u*h + (1-u)*c ----------------------------------------------------------------
        // remainder tile: lane (li,kq) needs h cols 16NC + 4kq + e = remainder frag q = kq of lane (li, e);
        // the cross-lane reads stay outside lane-divergent control flow
        f32x4 hrem = {0.f, 0.f, 0.f, 0.f};
        if (active) {
            stage_tail_reduce<D>(acc_c);
#pragma unroll
            for (int q = 0; q < NR; ++q) {
                const float h0 = __shfl(hf.r[q], li), h1 = __shfl(hf.r[q], li + 16);
                const float h2 = __shfl(hf.r[q], li + 32), h3 = __shfl(hf.r[q], li + 48);
                if (kq == q) hrem = f32x4{h0, h1, h2, h3};
            }
        }
        if (active && row < a.V && !(a.dbg & 4)) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int col = nt * 16 + 4 * kq;
                if (col < D) {
                    f32x4 hv;
                    if (nt < NC) hv = hf.v[nt];
                    else hv = hrem;
                    const f32x4 b = ld4(bias_s + 2 * D + col);
                    f32x4 c = acc_c[nt] + b;
                    if (a.act == GGNN_ACT_TANH) { c.x = tanh_f(c.x); c.y = tanh_f(c.y); c.z = tanh_f(c.z); c.w = tanh_f(c.w); }
                    else { c.x = fmaxf(c.x, 0.f); c.y = fmaxf(c.y, 0.f); c.z = fmaxf(c.z, 0.f); c.w = fmaxf(c.w, 0.f); }
                    const f32x4 u = acc_u[nt];
                    st4(a.h_out + ((unsigned)row * D + col), u * hv + (1.0f - u) * c);
                    if constexpr (SAVE) st4(a.save_c + ((unsigned)row * D + col), c);
                }
            }
        }
    }
}

template <int D, int NX, int NW, bool SAVE, bool GATHER>
static int launch_gru_fused(const GruFusedArgs& a_in, float* packed, hipStream_t st) {
    using C = StageCfg<D>;
    GruFusedArgs a = a_in;
    { static int dbg = -1; if (dbg < 0) { const char* e = getenv("GGNN_GRU_DBG"); dbg = e ? atoi(e) : 0; } a.dbg = dbg; }
    { const char* e = getenv("GGNN_GRU_TPTR"); a.tdbg = e ? (unsigned long long*)strtoull(e, nullptr, 10) : nullptr; }
    if (a.Wg) {   // raw weights given: build the stage images first (skipped when the caller pre-packed them)
        hipLaunchKernelGGL((gru_pack_weights_kernel<D>), dim3(8, 3 * (NX + 1)), dim3(256), 0, st, a.Wg, a.Wc, NX, packed);
        GGNN_CHECK_HIP(hipGetLastError());
    }
    if (a.h == nullptr) return GGNN_OK;   // pack-only call
    if ((unsigned long long)a.V * D >= (1ULL << 32))
        return fail(GGNN_E_UNSUPPORTED, "fused GRU indexes with 32-bit element offsets: V*D must be < 2^32 (V=%d, D=%d)", a.V, D);
    const size_t lds = (size_t)2 * C::IMG_BYTES + (size_t)(3 * D * sizeof(float) + 15) / 16 * 16;   // ring + biases
    const int wt_total = (a.V + 15) / 16;
    int nb = num_cus();
    const int need = (wt_total + NW - 1) / NW;
    if (nb > need) nb = need;
    static bool attr_set = false;
    if (!attr_set && lds > 64 * 1024) {
        GGNN_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&ggnn_gru_fused_kernel<D, NX, NW, SAVE, GATHER>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    hipLaunchKernelGGL((ggnn_gru_fused_kernel<D, NX, NW, SAVE, GATHER>), dim3(nb), dim3(NW * 64), lds, st, a, (const float*)packed);
    GGNN_CHECK_HIP(hipGetLastError());
    return GGNN_OK;
}

template <int D>
static int dispatch_nx(const GruFusedArgs& a, float* packed, hipStream_t st) {
    const bool save = a.save_r || a.save_u || a.save_c;
    if (save && !(a.save_r && a.save_u && a.save_c))
        return fail(GGNN_E_INVALID, "save_r / save_u / save_c must be given together");
    if (a.g_H) {
        if (save) return fail(GGNN_E_UNSUPPORTED, "the gather-fused GRU has no save_r/u/c variant");
        switch (a.nx) {
            case 1: return launch_gru_fused<D, 1, 8, false, true>(a, packed, st);
            case 2: return launch_gru_fused<D, 2, 8, false, true>(a, packed, st);
            case 3: return launch_gru_fused<D, 3, 8, false, true>(a, packed, st);
            default: return fail(GGNN_E_INVALID, "nx %d outside 1..3", a.nx);
        }
    }
    switch (a.nx) {
        case 1: return save ? launch_gru_fused<D, 1, 8, true, false>(a, packed, st) : launch_gru_fused<D, 1, 8, false, false>(a, packed, st);
        case 2: return save ? launch_gru_fused<D, 2, 8, true, false>(a, packed, st) : launch_gru_fused<D, 2, 8, false, false>(a, packed, st);
        case 3: return save ? launch_gru_fused<D, 3, 8, true, false>(a, packed, st) : launch_gru_fused<D, 3, 8, false, false>(a, packed, st);
        default: return fail(GGNN_E_INVALID, "nx %d outside 1..3", a.nx);
    }
}

int gru_fused_supported(int D) { return D == 100 || D == 64 || D == 32; }

int gru_fused_dispatch(const GruFusedArgs& a, int D, float* packed, hipStream_t st) {
    switch (D) {
        case 100: return dispatch_nx<100>(a, packed, st);
        case 64: return dispatch_nx<64>(a, packed, st);
        case 32: return dispatch_nx<32>(a, packed, st);
        default: return fail(GGNN_E_UNSUPPORTED, "no fused GRU for hidden size %d", D);
    }
}

}  // namespace ggnn

extern "C" int ggnn_gru_is_fused(int D) { return ggnn::gru_fused_supported(D); }
