// K3w: the gather-fused GRU launch (chem_tensorflow_sparse.py:198-216) in its WIDE form -- one wave per SIMD on the whole
// 512-entry register file, NTW 16-row tiles per wave that share every weight-fragment read, gate-sequential stage order, every
// piece of side work riding inside the product units (GGNN_GRU_FORM = 6 / 62; 61: the same pass body as 8 waves x one tile; 64: as
// two 4-wave one-tile workgroups per CU on half-image rings).
//
// STATUS (round 6; profiles/r06_experiments/wide_gru.md): bit-identical to the ring forms of ggnn_gru_fused.hip, and NOT faster --
// 82-84 us (form 62) / 71-75 us (form 61) / 73 us (form 64) against 67 us for ring form 1 at the headline shape.  A pass is bound by instruction
// ISSUE (~2700 instructions per tile at 6-8 clocks each), and one wave on a SIMD issues slower than two; what this form saves (LDS
// weight reads, image DMAs and barriers per row, SGPR spills, branches) is not what bounds the launch.  The default dispatch is
// unchanged; the form stays selectable (ggnn_gru_form_set) as the measured record of that design.
//
// Why it was built.  The ring forms run two 256-register waves per SIMD, one 16-row tile each: every wave reads every 48 KiB stage
// image for its own 16 rows (one ds_read_b128 per 1.5 MFMAs in the two-piece f16 format), a pass of a CU is 8 tiles -- three passes,
// i.e. 18 image DMAs and 18+ stage barriers per launch at the reference's batch size -- and all three accumulator sets of a tile (84
// registers) are live from the first stage to the last, which pins those kernels to the 256-register wall (72 spilled SGPRs, 150
// v_readlane / v_writelane per pass).  Here a workgroup is FOUR waves, one per SIMD, and a wave owns NTW tiles at once:
//   * a weight fragment (the two / three operand planes of one (32-chunk, column tile) unit) is read from LDS ONCE and multiplies
//     the NTW tiles' fragments: 3 NTW MFMAs per 2 ds_read_b128 instead of 3;
//   * a pass of a CU is 4 NTW tiles: fewer image DMAs and stage barriers per row, and a barrier is among four waves instead of eight;
//   * the stages run GATE-SEQUENTIALLY -- (residual segments segment-major, then) x -> r, h -> r | r epilogue | x -> u, h -> u | u
//     epilogue | x -> c, r*h -> c -- so ONE accumulator set per tile is open at a time in the gate-sequential part (28 registers
//     instead of 84).  Per accumulator the chain is the ring forms': segments in order, h / r*h last, chunks in order, remainder
//     last, three (six) products per unit smallest first -- RESULTS ARE BIT-IDENTICAL to those kernels
//     (tests/test_gpu_parity.py::test_wide_gru_equals_ring_forms); the fusions hipcc applies to the ring kernels' epilogue
//     expressions are written out here (w_sigmoid4_acc, w_tanh4_acc, w_blend4): which contraction an expression gets depends on the
//     code around it.  The stage images are the ones ggnn_gru_pack_weights_f32 writes: only the ORDER in which a pass streams them changes.
//   * a wave alone on its SIMD has no partner whose MFMAs would cover its vector work, and it issues in order: the side work of a
//     pass is cut into PIECES and each piece is emitted inside the scheduling region of one product unit (wide_stage_mma's fill):
//     r / u epilogues a float4 group at a time, the state's planes a chunk at a time, the gathered rows of the pass to come ONE LOAD
//     per unit, the mean + split of the gathered segment, the image DMA a KiB piece at a time (GGNN_WIDE_DMA_PIECES), optionally the
//     candidate epilogue of the previous pass (GGNN_WIDE_CARRY, off: it spills).
// Two rules the measurements taught (both fatal for a lone wave when broken):
//   1. the load counter retires IN ORDER and everything counts in it -- a scratch reload, a state row, a gathered row is waited for
//      behind every older DMA piece (inline assembly: invisible to the compiler's count) and every older global_store's write
//      acknowledgement: no spills in the hot loop, loads before stores;
//   2. a run-time branch per float4 group (optional stores, activation) cuts an epilogue into one basic block per group, each
//      waiting out its own bias read and exp -> rcp chain: SAVE and the activation are compile-time, stores go to clamped rows.
// The r / u tail columns ride in the r images' padding columns exactly as in the ring forms (StageCfg::TAILPACK / TAILPACK3).
//
// Work split: workgroup b owns the contiguous tile range [T b / nb, T (b + 1) / nb) and walks it in P = ceil(n_b / (NW NTW)) passes;
// the n_b tiles are dealt evenly over the NW P (pass, wave) slots, so no wave ever has more than one tile above another's and a
// wave's count never grows from pass to pass.  The pass body is instantiated per tile count (NTA = 1 .. NTW).
//
// Gather (the aggregated-messages segment, :198-212) of the pass to come: slot range + in-degrees beside h -> r, the first four
// source-row indices beside x -> u, the rows of slots 0, 1 inside h -> u, slot 2 inside x -> c, slot 3 inside r*h -> c, where the sum
// is finished (any further slots synchronously), divided (one division per row) and split into the next pass's operand planes;
// slot order and adds are those of ggnn_gather_segment_sum_f32.
#include "ggnn_split.hpp"
#include <type_traits>

namespace ggnn {

int gru_wide_launch(int D, int nx, int ntw_req, const GruFusedArgs& a, float* packed, hipStream_t st);
int gru_wide_supported(int D, int nx, const GruFusedArgs& a);

namespace {

struct GruWideArgs {
    const float* x[2];           // residual segments (NX - 1 of them)
    const float* h; const float* bg; const float* bc; float* h_out;
    float* save_r; float* save_u; float* save_c; float* save_x;
    const float* g_H; const int* g_row_ptr; const int* g_idx; const float* g_nin;
    const float* packed;
    int V, act, g_T, g_use_avg;
    int* tickets;                // the ABI's tile counter (0 at launch, left non-zero): this form deals its tiles statically and only marks it
    unsigned long long* tdbg;    // (GGNN_WIDE_STAMPS builds) s_memtime stamps of workgroup 0: tools/wide_timeline.py
};

// -DGGNN_WIDE_STAMPS=1 (tools/variant_lib.sh): stamp [pass < 4][point < 32][wave] of workgroup 0; points 4 j + {0 stage start, 1 side
// work done, 2 products done, 3 barrier passed}, 24 candidate epilogue done, 25 pass done
#ifndef GGNN_WIDE_STAMPS
#define GGNN_WIDE_STAMPS 0
#endif
#if GGNN_WIDE_STAMPS
#define GGNN_WT(IDX) if (a.tdbg && blockIdx.x == 0 && lane == 0 && p < 4) a.tdbg[((p * 32) + (IDX)) * 4 + wave] = __builtin_amdgcn_s_memtime();
#else
#define GGNN_WT(IDX)
#endif

template <int I0, int I1, class F>
__device__ __forceinline__ void sfor(F&& f) {
    if constexpr (I0 < I1) { f(std::integral_constant<int, I0>{}); sfor<I0 + 1, I1>(f); }
}

template <int D>
__device__ __forceinline__ void frag_add_w(Frag<D>& f, const Frag<D>& t) {
#pragma unroll
    for (int c = 0; c < StageCfg<D>::NC; ++c) f.v[c] += t.v[c];
#pragma unroll
    for (int q = 0; q < StageCfg<D>::NR; ++q) f.r[q] += t.r[q];
}

// (the gate epilogues of ggnn_gru_fused.hip with their fusions written out: z k + b is ONE fma there, and which fusion an expression
// gets depends on the code around it -- the forms must agree bit for bit)
template <int FMT>
__device__ __forceinline__ f32x4 w_sigmoid4_acc(f32x4 z, f32x4 b_scaled) {
    constexpr float k = -kLog2e * SplitFmt<FMT>::acc_scale;
    const f32x4 e = {__builtin_fmaf(z.x, k, b_scaled.x), __builtin_fmaf(z.y, k, b_scaled.y), __builtin_fmaf(z.z, k, b_scaled.z), __builtin_fmaf(z.w, k, b_scaled.w)};
    return rcp_4(exp2_4(e) + 1.0f);
}
template <int FMT>
__device__ __forceinline__ f32x4 w_tanh4_acc(f32x4 z, f32x4 b_scaled) {
    constexpr float k = 2.0f * kLog2e * SplitFmt<FMT>::acc_scale;
    const f32x4 e = {__builtin_fmaf(z.x, k, b_scaled.x), __builtin_fmaf(z.y, k, b_scaled.y), __builtin_fmaf(z.z, k, b_scaled.z), __builtin_fmaf(z.w, k, b_scaled.w)};
    return 1.0f - 2.0f * rcp_4(exp2_4(e) + 1.0f);
}

// h' = u h + (1 - u) c in the contraction hipcc gives the ring kernels' `u * hv + (1.0f - u) * c` (AMDGPU fuses aggressively:
// (1 - u) c -> fma(-u, c, c), then the sum -> fma(u, h, .)); written out, because which fusion an expression gets depends on the
// code around it and the forms must agree bit for bit.
#ifndef GGNN_WIDE_BLEND
#define GGNN_WIDE_BLEND 1
#endif
__device__ __forceinline__ float w_blend1(float u, float h, float c) {
#if GGNN_WIDE_BLEND == 0
    return __builtin_fmaf(u, h, __builtin_fmaf(-u, c, c));
#elif GGNN_WIDE_BLEND == 1
    float t = (1.0f - u) * c; asm volatile("" : "+v"(t)); return __builtin_fmaf(u, h, t);
#elif GGNN_WIDE_BLEND == 2
    float t = u * h; asm volatile("" : "+v"(t)); return __builtin_fmaf(1.0f - u, c, t);
#else
    float t = u * h; asm volatile("" : "+v"(t)); return __builtin_fmaf(-u, c, c) + t;
#endif
}
__device__ __forceinline__ f32x4 w_blend4(f32x4 u, f32x4 h, f32x4 c) {
    return f32x4{w_blend1(u.x, h.x, c.x), w_blend1(u.y, h.y, c.y), w_blend1(u.z, h.z, c.z), w_blend1(u.w, h.w, c.w)};
}

// the operand planes of one 16-row activation fragment (p[0] = hi pieces, p[1] = mid / lo, p[2] = lo of the bf16 form)
template <int D, int FMT>
struct WPl { u32x4 p[SplitFmt<FMT>::NP][SplitCfg<D, FMT>::NC2 > 0 ? SplitCfg<D, FMT>::NC2 : 1]; };

template <int D, int FMT>
__device__ __forceinline__ void wsplit(WPl<D, FMT>& s, const Frag<D>& f) {
#pragma unroll
    for (int c2 = 0; c2 < SplitCfg<D, FMT>::NC2; ++c2) {
        const f32x4 a = f.v[2 * c2], b = f.v[2 * c2 + 1];
        unsigned h[4], m[4], l[4];
        split_pair<FMT>(a.x, a.y, h[0], m[0], l[0]);
        split_pair<FMT>(a.z, a.w, h[1], m[1], l[1]);
        split_pair<FMT>(b.x, b.y, h[2], m[2], l[2]);
        split_pair<FMT>(b.z, b.w, h[3], m[3], l[3]);
        s.p[0][c2] = u32x4{h[0], h[1], h[2], h[3]};
        s.p[1][c2] = u32x4{m[0], m[1], m[2], m[3]};
        if constexpr (SplitFmt<FMT>::NP > 2) s.p[2][c2] = u32x4{l[0], l[1], l[2], l[3]};
    }
}

// one 32-chunk of a fragment -> its operand planes
template <int D, int FMT>
__device__ __forceinline__ void wsplit_chunk(WPl<D, FMT>& s, int c2, f32x4 a, f32x4 b) {
    unsigned h[4], m[4], l[4];
    split_pair<FMT>(a.x, a.y, h[0], m[0], l[0]);
    split_pair<FMT>(a.z, a.w, h[1], m[1], l[1]);
    split_pair<FMT>(b.x, b.y, h[2], m[2], l[2]);
    split_pair<FMT>(b.z, b.w, h[3], m[3], l[3]);
    s.p[0][c2] = u32x4{h[0], h[1], h[2], h[3]};
    s.p[1][c2] = u32x4{m[0], m[1], m[2], m[3]};
    if constexpr (SplitFmt<FMT>::NP > 2) s.p[2][c2] = u32x4{l[0], l[1], l[2], l[3]};
}

// KiB piece K of this wave's share of an image DMA (dma_image_asm's transfer, one global_load_lds_dwordx4 at a time: the pieces
// ride in the units of a stage's products instead of going out in one burst in front of them)
template <int BYTES, int NW, int K>
__device__ __forceinline__ void dma_piece_asm(const float* src, float* dst, int wave, int lane) {
    constexpr int PER_WAVE = BYTES / (NW * 1024);
    static_assert(BYTES % (NW * 1024) == 0 && K < PER_WAVE, "image must split into whole KiB per wave");
    const unsigned voff = (unsigned)lane * 16u;
    const unsigned long long sb = reinterpret_cast<unsigned long long>(src) + (unsigned long long)wave * PER_WAVE * 1024 + (unsigned long long)(K & ~3) * 1024;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)sb);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(sb >> 32));
    const unsigned long long sa = ((unsigned long long)hi << 32) | lo;
    const unsigned l = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)(lds_void*)dst + (unsigned)wave * (PER_WAVE * 1024) + (unsigned)(K & ~3) * 1024u);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:%3" :: "s"(l), "v"(voff), "s"(sa), "n"((K & 3) * 1024) : "memory");
}

// one of the NC + NR loads of load_frag<D>: PART < NC the float4 of 16-chunk PART, PART = NC + q the remainder float q
template <int D, int PART>
__device__ __forceinline__ void load_frag_part(Frag<D>& f, const float* base, int row, int kq) {
    constexpr int NC = StageCfg<D>::NC;
    const unsigned ob = ((unsigned)row * (unsigned)D + 4u * (unsigned)kq) * 4u;
    __builtin_assume(ob < 0xF0000000u);
    if constexpr (PART < NC) f.v[PART] = ld4_b(base, ob + 64u * PART);
    else f.r[PART - NC] = ld1_b(base, ob - 16u * (unsigned)kq + 4u * (unsigned)(16 * NC + 4 * (PART - NC)) + 4u * (unsigned)kq);
}

struct NoFill { template <class U> __device__ __forceinline__ void operator()(U) const {} };

// units of a stage's product: NC2 * NTILES (32-chunk, column tile) units + NR * NTILES remainder units
template <int D, int FMT> __host__ __device__ constexpr int wide_units(int ntiles) { return (SplitCfg<D, FMT>::NC2 + StageCfg<D>::NR) * ntiles; }

// acc[t][nt] (+)= fragment t x split stage image for the wave's NTA tiles: stage_mma_split_at's products in its order per
// accumulator (chunks in order, then the remainder on the f32 MFMA; per unit the products smallest first), every weight fragment
// read once for all NTA tiles.  The planes of the NEXT unit are fetched at the start of the current one.
// fill(unit): a wave alone on its SIMD has no partner whose MFMAs would cover its vector work, and it issues in order -- so the side
// work of a stage (an epilogue of the previous gate, a fragment split) is cut into PIECES and piece `unit` is emitted inside unit
// `unit`'s scheduling region, where the scheduler deals its instructions between the unit's MFMAs (GGNN_WIDE_SGB: in a fixed
// pattern, NTA MFMAs | FV vector instructions | ...).
#ifndef GGNN_WIDE_SGB
#define GGNN_WIDE_SGB 1
#endif
#ifndef GGNN_WIDE_FV
#define GGNN_WIDE_FV 6
#endif
// (tiles [T0, T1) of the stage; half A of the image at img, half B at img_b -- one image, or two ring slots of the half-image form;
// U0: the number this call's first unit has for `fill`)
template <int D, int NTW, int NTA, int T0, int T1, bool ZERO, int FMT, int U0, class Fill = NoFill>
__device__ __forceinline__ void wide_stage_mma_at(f32x4 (&acc)[NTW][StageCfg<D>::NT], const WPl<D, FMT> (&a)[NTW],
                                                  const float (&ar)[NTW][StageCfg<D>::NR > 0 ? StageCfg<D>::NR : 1],
                                                  const float* img, const float* img_b, int li, int kq, const Fill& fill = Fill()) {
    constexpr int NTILES = T1 - T0;
    using S = StageCfg<D>;
    using C = SplitCfg<D, FMT>;
    constexpr int NP = C::NP;
    // (REMAT, as in stage_mma_split_at: the lane parts of the LDS addresses are recomputed per stage instead of living -- and being
    // spilled -- across the pass: a scratch reload here waits, in order, behind every DMA piece and gathered row in flight)
    asm volatile("" : "+v"(li), "+v"(kq));
    constexpr int NU = C::NC2 * NTILES;                               // units, chunk-major
    constexpr bool FILLED = !std::is_same<Fill, NoFill>::value;
    if constexpr (NU > 0) {
        const u32x4* base_a = reinterpret_cast<const u32x4*>(img) + kq * (C::TA * 16) + li;
        const u32x4* base_b = reinterpret_cast<const u32x4*>(img_b) + kq * ((S::NT - C::TA) * 16) + li;
        auto slot = [&](int u, int p) {
            const int nt = T0 + u % NTILES, c2 = u / NTILES;
            const int nth = C::nth_of(nt);
            const u32x4* b = C::half_of(nt) ? base_b : base_a;
            return b[p * (C::plane_bytes(nth) / 16) + c2 * 4 * nth * 16 + C::tile_in_half(nt) * 16];
        };
        u32x4 w[NP], n[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) { w[p] = slot(0, p); n[p] = w[p]; }
        sfor<0, NU>([&](auto uc) {
            constexpr int u = decltype(uc)::value, c2 = u / NTILES, nt = T0 + u % NTILES;
            if constexpr (u + 1 < NU) {
#pragma unroll
                for (int p = NP - 1; p >= 0; --p) n[p] = slot(u + 1, p);
            }
            if constexpr (!FILLED) __builtin_amdgcn_sched_barrier(0);
            fill(std::integral_constant<int, U0 + u>{});
            f32x4 c[NTA];
#pragma unroll
            for (int t = 0; t < NTA; ++t) c[t] = (ZERO && c2 == 0) ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[t][nt];
            if constexpr (FMT == kSplitF16x2) {
#pragma unroll
                for (int t = 0; t < NTA; ++t) c[t] = mfma_f16(w[1], a[t].p[0][c2], c[t]);
#pragma unroll
                for (int t = 0; t < NTA; ++t) c[t] = mfma_f16(w[0], a[t].p[1][c2], c[t]);
#pragma unroll
                for (int t = 0; t < NTA; ++t) c[t] = mfma_f16(w[0], a[t].p[0][c2], c[t]);
            } else {
#pragma unroll
                for (int t = 0; t < NTA; ++t) c[t] = mfma_bf16(w[2], a[t].p[0][c2], c[t]);
#pragma unroll
                for (int t = 0; t < NTA; ++t) c[t] = mfma_bf16(w[1], a[t].p[1][c2], c[t]);
#pragma unroll
                for (int t = 0; t < NTA; ++t) c[t] = mfma_bf16(w[1], a[t].p[0][c2], c[t]);
#pragma unroll
                for (int t = 0; t < NTA; ++t) c[t] = mfma_bf16(w[0], a[t].p[2][c2], c[t]);
#pragma unroll
                for (int t = 0; t < NTA; ++t) c[t] = mfma_bf16(w[0], a[t].p[1][c2], c[t]);
#pragma unroll
                for (int t = 0; t < NTA; ++t) c[t] = mfma_bf16(w[0], a[t].p[0][c2], c[t]);
            }
            if constexpr (FILLED && GGNN_WIDE_SGB) {
                if constexpr (u + 1 < NU) __builtin_amdgcn_sched_group_barrier(0x100, NP, 0);       // the next unit's planes
#pragma unroll
                for (int m = 0; m < (FMT == kSplitF16x2 ? 3 : 6); ++m) {
                    __builtin_amdgcn_sched_group_barrier(0x008, NTA, 0);                            // one product of every tile
                    __builtin_amdgcn_sched_group_barrier(0x002, GGNN_WIDE_FV, 0);                   // ... then vector instructions of the piece
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < NTA; ++t) acc[t][nt] = c[t];
#pragma unroll
            for (int p = 0; p < NP; ++p) w[p] = n[p];
        });
    }
    // the D % 16 remainder k values on the f32 MFMA (their weights one tile ahead)
    if constexpr (S::NR > 0 && NTILES > 0) {
        constexpr int NRM = S::NR * NTILES;
        auto rw = [&](int i) {
            const int nt = T0 + i % NTILES, q = i / NTILES;
            const int nth = C::nth_of(nt);
            const float* b = C::half_of(nt) ? img_b : img;
            return b[C::main_bytes(nth) / 4 + (q * 4 + kq) * nth * 16 + li + C::tile_in_half(nt) * 16];
        };
        float w0 = rw(0), w1 = NRM > 1 ? rw(1) : 0.f;
        sfor<0, NRM>([&](auto ic) {
            constexpr int i = decltype(ic)::value, q = i / NTILES, nt = T0 + i % NTILES;
            fill(std::integral_constant<int, U0 + NU + i>{});
#pragma unroll
            for (int t = 0; t < NTA; ++t) acc[t][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0, ar[t][q], acc[t][nt], 0, 0, 0);
            if constexpr (FILLED && GGNN_WIDE_SGB) {
#pragma unroll
                for (int m = 0; m < NTA; ++m) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, GGNN_WIDE_FV, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            w0 = w1;
            if constexpr (i + 2 < NRM) w1 = rw(i + 2);
        });
    }
}

template <int D, int NTW, int NTA, int NTILES, bool ZERO, int FMT, class Fill = NoFill>
__device__ __forceinline__ void wide_stage_mma(f32x4 (&acc)[NTW][StageCfg<D>::NT], const WPl<D, FMT> (&a)[NTW],
                                               const float (&ar)[NTW][StageCfg<D>::NR > 0 ? StageCfg<D>::NR : 1],
                                               const float* img, int li, int kq, const Fill& fill = Fill()) {
    wide_stage_mma_at<D, NTW, NTA, 0, NTILES, ZERO, FMT, 0>(acc, a, ar, img, img + SplitCfg<D, FMT>::HA, li, kq, fill);
}

// Stage sequence of a pass (NSTAGE = 3 (NX + 1) images), position j -> (gate g: 0 r, 1 u, 2 candidate; segment s):
//   residual segments s = 0 .. NX-2, SEGMENT-major: s -> r, s -> u, s -> c (a residual fragment is read once and dies after its third
//   stage; all three accumulator sets are open from here on when NX > 1),
//   then GATE-sequential over the gathered segment and the state: x -> r, h -> r | r epilogue | x -> u, h -> u | u epilogue |
//   x -> c, r*h -> c.  Per accumulator: segments in order, h / r*h last -- the ring forms' chains.
// GGNN_WIDE_ORDER (experiment): 1 = the last part SEGMENT-major like the ring forms -- x -> r, x -> u, x -> c | h -> r, h -> u | r*h -> c
// (all three accumulator sets open throughout; the r epilogue rides in h -> u, the u epilogue in r*h -> c).
#ifndef GGNN_WIDE_ORDER
#define GGNN_WIDE_ORDER 0
#endif
template <int NX> __host__ __device__ constexpr int wide_gate(int j) {
    if (j < 3 * (NX - 1)) return j % 3;
    const int jj = j - 3 * (NX - 1);
    return GGNN_WIDE_ORDER ? (jj < 3 ? jj : (jj == 5 ? 2 : jj - 3)) : jj / 2;
}
template <int NX> __host__ __device__ constexpr int wide_seg(int j) {
    if (j < 3 * (NX - 1)) return j / 3;
    const int jj = j - 3 * (NX - 1);
    return NX - 1 + (GGNN_WIDE_ORDER ? (jj < 3 ? 0 : 1) : jj % 2);
}

// SAVE: r, u, c and the gathered segment are written for a backward pass (training).  TANH: the candidate's activation (else ReLU).
// Both are compile-time: a run-time branch per float4 group cuts the epilogues into basic blocks of one group each, and a wave
// alone on its SIMD then waits out every bias read and every exp -> rcp chain one after the other.
// FILLS: 1 = the epilogues / splits are dealt into the MFMA stream of the stage they precede (wide_stage_mma's FILL), 0 = they
// run in front of it.
#ifndef GGNN_WIDE_DMA_PIECES
#define GGNN_WIDE_DMA_PIECES 1   // 1: the image DMA goes out a KiB piece at a time inside the product units (0: in one burst in front of them)
#endif
#ifndef GGNN_WIDE_CARRY
#define GGNN_WIDE_CARRY 0     // the candidate epilogue of a pass rides in the first stages of the next one (0: it runs behind its pass)
#endif
// NW: waves per workgroup -- 4 (one per SIMD, the whole register file each) or 8 (two per SIMD on 256 registers each: NTW = 1).
// HALF (form 64): TWO 4-wave workgroups per CU (256 registers a wave, NTW = 1), each on a two-slot ring of HALF images (slot 0:
// tiles [0, TA), slot 1: the rest; a stage = two sub-stages with a barrier each) -- ring form 1's arrangement: the two waves of a
// SIMD belong to different workgroups and meet at no barrier.
template <int D, int NX, int NTW, int FMT, bool SAVE, bool TANH, int NW = 4, bool HALF = false>
__global__ __launch_bounds__(NW * 64, HALF ? 2 : NW / 4) void ggnn_gru_wide_kernel(GruWideArgs a) {
    using C = StageCfg<D>;
    using I = ImgCfg<D, true, FMT>;
    constexpr int NT = C::NT, NC = C::NC, NR = C::NR, NRR = NR > 0 ? NR : 1;
    constexpr int NSEG = NX + 1, NSTAGE = 3 * NSEG, KI = 4;
    constexpr int JG = 3 * (NX - 1);                                 // first stage of the gate-sequential part
    extern __shared__ __attribute__((aligned(16))) float lds_[];   // [biases | ring [2][IMG]]
    const int tid = threadIdx.x;
    int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int li = lane & 15, kq = lane >> 4;               // (not const: re-made opaque per stage, see the stage loop)

    constexpr int BIAS_FLOATS = (4 * D + 63) / 64 * 64;
    float* bias_s = lds_;                             // [-log2e*bg (2D) | 2 log2e*bc (D) | bc (D)]
    float* ring = lds_ + BIAS_FLOATS;
    for (int i = tid; i < 4 * D; i += NW * 64)
        bias_s[i] = i < 2 * D ? -kLog2e * a.bg[i] : (i < 3 * D ? 2.0f * kLog2e * a.bc[i - 2 * D] : a.bc[i - 3 * D]);

    if (a.tickets && blockIdx.x == 0 && tid == 0) atomicAdd(a.tickets, (int)gridDim.x);   // (the counter's contract: non-zero after the launch)
    // ---- this workgroup's tiles and their deal over the (pass, wave) slots ---------------------------------------------------
    const int wt_total = (a.V + 15) / 16;
    const int nb = gridDim.x, bid = blockIdx.x;
    const int t_lo = (int)((long long)wt_total * bid / nb), t_hi = (int)((long long)wt_total * (bid + 1) / nb);
    const int n_b = t_hi - t_lo;
    const int P = (n_b + NW * NTW - 1) / (NW * NTW);
    if (P == 0) return;
    const int nsl = NW * P;
    const int sl_base = n_b / nsl, sl_extra = n_b % nsl;
    auto slot_first = [&](int p) { const int s = p * NW + wave; return t_lo + s * sl_base + (s < sl_extra ? s : sl_extra); };
    auto slot_count = [&](int p) { const int s = p * NW + wave; return p < P ? sl_base + (s < sl_extra ? 1 : 0) : 0; };

    // stage position -> packed image (ggnn_gru_fused.hip's image order: gates (s = 0..NX) x {r, u}, then the candidate blocks)
    auto img_of = [&](int j) -> const float* {
        const int g = wide_gate<NX>(j), s = wide_seg<NX>(j);
        return a.packed + (size_t)(g < 2 ? 2 * s + g : 2 * NSEG + s) * I::IMG;
    };
    int cur = 0;
    using SCH = SplitCfg<D, FMT>;
    static_assert(!HALF || (NW == 4 && NTW == 1), "the half-image form is four waves x one tile");
    constexpr int SLOT = HALF ? SCH::HA : I::IMG;                    // floats per ring slot
    auto dma = [&](const float* src, float* dst) { dma_image_asm<I::IMG_BYTES, NW>(src, dst, wave, lane); };
    auto dma_ha = [&](const float* src, float* dst) { dma_kib_asm<SCH::HA_BYTES / 1024, NW>(src, dst, wave, lane); };
    auto dma_hb = [&](const float* src, float* dst) { dma_kib_asm<SCH::HB_BYTES / 1024, NW>(src, dst, wave, lane); };
    auto publish = [&]() { dma_wait(); __syncthreads(); };
    if constexpr (HALF) dma_ha(img_of(0), ring);
    else dma(img_of(0), ring);

    // ---- state that crosses a pass boundary: the gathered segment and the first residual segment of the pass to come -----------
    WPl<D, FMT> xs[NTW];                              // operand planes of the gathered segment (aggregated messages)
    float xr[NTW][NRR];                               // ... and its remainder values (f32 MFMA)
    WPl<D, FMT> xq[NTW];                              // the residual segment being multiplied (NX > 1)
    float xqr[NTW][NRR];
    // gather of the aggregated-messages segment.  Its registers are LOCAL to a pass (GatherRegs below): declared outside the pass
    // loop, a conditionally executed phase ("if this wave has a tile t in the pass to come") would keep the previous pass's values
    // alive around the whole loop -- 200 registers that nothing reads.
    struct GatherRegs {
        int beg[NTW], end[NTW], i[NTW][KI];
        f32x4 n[NTW];
        float den[NTW], rcp[NTW];
        Frag<D> x[NTW], t[NTW];
        Frag<D> rf[NTW];                              // residual segment 0 of the pass to come, on its way in
    };
    auto g_ptrs = [&](GatherRegs& G, int t, int r) {  // level 1: slot range + in-degrees of row r
        G.beg[t] = ldi_b(a.g_row_ptr, (unsigned)r * 4u); G.end[t] = ldi_b(a.g_row_ptr, (unsigned)r * 4u + 4u);
        G.n[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (a.g_use_avg) {
            if (a.g_T == 4) {
                G.n[t] = ld4_b(a.g_nin, (unsigned)r * 16u);
            } else {
                float deg = 0.f;
                for (int k = 0; k < a.g_T; ++k) deg += a.g_nin[(size_t)r * a.g_T + k];
                G.n[t] = f32x4{deg, 0.f, 0.f, 0.f};
            }
        }
    };
    auto g_index = [&](GatherRegs& G, int t) {        // level 2: the first KI source rows (a slot beyond the degree -> row 0)
        G.den[t] = (((G.n[t].x + G.n[t].y) + G.n[t].z) + G.n[t].w) + 1e-7f;
        G.rcp[t] = 1.0f / G.den[t];
#pragma unroll
        for (int j = 0; j < KI; ++j) {
            G.i[t][j] = 0;
            if (G.beg[t] + j < G.end[t]) G.i[t][j] = ldi_b(a.g_idx, (unsigned)(G.beg[t] + j) * 4u);
        }
    };
    auto g_rows0 = [&](GatherRegs& G, int t) {        // level 3: slot 0 straight into x, slot 1 into the temporary
        load_frag<D>(G.x[t], a.g_H, G.i[t][0], kq);
        load_frag<D>(G.t[t], a.g_H, G.i[t][1], kq);
    };
    auto g_rows = [&](GatherRegs& G, int t, int k) {  // add slot k-1 (landed), fetch slot k
        if (G.beg[t] + k - 1 < G.end[t]) frag_add_w(G.x[t], G.t[t]);
        load_frag<D>(G.t[t], a.g_H, G.i[t][k], kq);
    };
    auto g_finish = [&](GatherRegs& G, int t) {       // slot 3, any further slots (synchronously), mean
        Frag<D>& f = G.x[t];
        if (G.beg[t] + KI - 1 < G.end[t]) frag_add_w(f, G.t[t]);
        for (int e = G.beg[t] + KI; e < G.end[t]; ++e) {
            load_frag<D>(G.t[t], a.g_H, a.g_idx[e], kq);
            frag_add_w(f, G.t[t]);
        }
        if (G.beg[t] >= G.end[t]) {                   // a node without incoming messages (slot 0 was row 0)
#pragma unroll
            for (int c = 0; c < NC; ++c) f.v[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < NR; ++q) f.r[q] = 0.f;
        }
        if (a.g_use_avg) {                            // :206-209, one division per row (Markstein: the correctly rounded quotient)
            const float den = G.den[t], r = G.rcp[t];
            auto dv = [&](float x) { const float q = x * r; return fmaf(fmaf(-den, q, x), r, q); };
#pragma unroll
            for (int c = 0; c < NC; ++c) f.v[c] = f32x4{dv(f.v[c].x), dv(f.v[c].y), dv(f.v[c].z), dv(f.v[c].w)};
#pragma unroll
            for (int q = 0; q < NR; ++q) f.r[q] = dv(f.r[q]);
        }
    };
    auto store_x = [&](const Frag<D>& f, int row_) {  // (training) the gathered segment is an operand of the weight gradients
        const unsigned ob = ((unsigned)row_ * (unsigned)D + 4u * (unsigned)kq) * 4u;
#pragma unroll
        for (int c = 0; c < NC; ++c) st4_b(a.save_x, ob + 64u * c, f.v[c]);
#pragma unroll
        for (int q = 0; q < NR; ++q) *reinterpret_cast<float*>(reinterpret_cast<char*>(a.save_x) + ob - 16u * (unsigned)kq + (16u * NC + 4u * q + (unsigned)kq) * 4u) = f.r[q];
    };
    auto commit = [&](WPl<D, FMT>& pl, float (&rem)[NRR], const Frag<D>& f) {   // f32 fragment -> operand planes + remainder values
        wsplit<D, FMT>(pl, f);
#pragma unroll
        for (int q = 0; q < NR; ++q) rem[q] = f.r[q];
    };
    // gathered fragment of tile t complete -> planes of the pass to come; its first residual segment with it
    auto x_commit = [&](GatherRegs& G, int t, int row_) {
        if constexpr (SAVE) store_x(G.x[t], row_);
        commit(xs[t], xr[t], G.x[t]);
        if constexpr (NX > 1) commit(xq[t], xqr[t], G.rf[t]);
    };
    auto clampv = [&](int r) { return r < a.V ? r : a.V - 1; };

    // ---- first pass: its x segments, synchronously -----------------------------------------------------------------------------
    {
        GatherRegs G;
        const int t0 = slot_first(0), n0 = slot_count(0);
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            if (t < n0) {
                const int r = clampv((t0 + t) * 16 + li);
                g_ptrs(G, t, r);
                if constexpr (NX > 1) load_frag<D>(G.rf[t], a.x[0], r, kq);
            }
        }
#pragma unroll
        for (int t = 0; t < NTW; ++t) if (t < n0) g_index(G, t);
#pragma unroll
        for (int t = 0; t < NTW; ++t) if (t < n0) g_rows0(G, t);
#pragma unroll
        for (int t = 0; t < NTW; ++t) if (t < n0) g_rows(G, t, 2);
#pragma unroll
        for (int t = 0; t < NTW; ++t) if (t < n0) g_rows(G, t, 3);
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            if (t < n0) { g_finish(G, t); x_commit(G, t, clampv((t0 + t) * 16 + li)); }
        }
    }
    publish();

    // ---- the candidate epilogue of a pass runs INSIDE the next pass (its pieces ride in the units of that pass's first stages) ------
    // what it needs crosses the pass boundary here; every path through the pass loop redefines all of it (no stale live ranges)
    f32x4 pc[NTW][NT], pu[NTW][NT];                   // candidate accumulators and u of the tiles whose epilogue is pending
    f32x4 ptl[NTW];                                   // their r-set tail tile (the x segments' share of the candidate's last tile)
    int prow[NTW];                                    // their (clamped) rows
    int npend = 0;                                    // tiles pending (wave-uniform)
    auto pend_clear = [&](int from) {
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            if (t >= from) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) { pc[t][nt] = f32x4{0.f, 0.f, 0.f, 0.f}; pu[t][nt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
                ptl[t] = f32x4{0.f, 0.f, 0.f, 0.f}; prow[t] = 0;
            }
        }
    };
    pend_clear(0);
    // c = act(.), h' = u*h + (1-u)*c for one float4 group (tile t, column tile nt) of the pending tiles; hq: the tile's state rows
    // (read again: the fragment died with the r epilogue), bzc: the candidate biases.  Group 0 of a tile first folds the x
    // segments' share of the last tile in (TAILPACK3) -- group NT - 1 reads it.
    auto c_group = [&](auto tc, auto ntc, const Frag<D>& hq, const f32x4 bcv) {
        constexpr int t = decltype(tc)::value, nt = decltype(ntc)::value;
        if constexpr (nt == 0 && C::TAILPACK3) {
            // the x segments' share of the candidate's last tile was accumulated two lane groups up in the r gate's last tile
            // (which the r epilogue rewrote only in its own lanes); add it to the r*h share
            constexpr int SH2 = 32 * ((D % 16) / 4);
            f32x4 ct;
            ct.x = __shfl(ptl[t].x, lane + SH2); ct.y = __shfl(ptl[t].y, lane + SH2);
            ct.z = __shfl(ptl[t].z, lane + SH2); ct.w = __shfl(ptl[t].w, lane + SH2);
            pc[t][NT - 1] = ct + pc[t][NT - 1];
        }
        f32x4 hv;
        if constexpr (nt < NC) hv = hq.v[nt < NC ? nt : 0];
        else {
            hv = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < NR; ++q) {
                const float h0 = __shfl(hq.r[q], li), h1 = __shfl(hq.r[q], li + 16);
                const float h2 = __shfl(hq.r[q], li + 32), h3 = __shfl(hq.r[q], li + 48);
                if (kq == q) hv = f32x4{h0, h1, h2, h3};
            }
        }
        // (a row beyond V is the clamped row V - 1 computed again from the same inputs: its stores write the same values to the
        // same addresses -- no predicate, no branch)
        const int col = nt * 16 + 4 * kq;
        if (col < D) {
            f32x4 c;
            if constexpr (TANH) {
                c = w_tanh4_acc<FMT>(pc[t][nt], bcv);
            } else {
                c = pc[t][nt] * SplitFmt<FMT>::acc_scale + bcv;
                c.x = fmaxf(c.x, 0.f); c.y = fmaxf(c.y, 0.f); c.z = fmaxf(c.z, 0.f); c.w = fmaxf(c.w, 0.f);
            }
            st4_b(a.h_out, ((unsigned)prow[t] * D + col) * 4u, w_blend4(pu[t][nt], hv, c));
            if constexpr (SAVE) st4_b(a.save_c, ((unsigned)prow[t] * D + col) * 4u, c);
        }
    };
    auto load_cbias = [&](f32x4 (&bzc)[NT]) {
        const float* b = TANH ? bias_s + 2 * D : bias_s + 3 * D;                     // (2 log2e * bc | bc itself)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bzc[nt] = (nt * 16 + 4 * kq < D) ? ld4(b + nt * 16 + 4 * kq) : f32x4{0.f, 0.f, 0.f, 0.f};
    };
    // the pending epilogue on its own (in front of a pass with another tile count, and behind the last pass)
    auto flush = [&]() {
        f32x4 bzc[NT];
        load_cbias(bzc);
        sfor<0, NTW>([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            if (t < npend) {
                Frag<D> hq;
                load_frag<D>(hq, a.h, prow[t], kq);
                sfor<0, NT>([&](auto ntc) { c_group(tc, ntc, hq, bzc[decltype(ntc)::value]); });
            }
        });
        npend = 0;
    };

    // ---- one pass with NTA tiles on this wave; CARRY: the NTA pending tiles' candidate epilogue rides in its first stages ---------
    auto run_pass = [&](auto nta_c, auto carry_c, const int p) {
        constexpr int NTA = decltype(nta_c)::value;
        constexpr bool CARRY = decltype(carry_c)::value;
        const int t0 = slot_first(p);
        const bool last = p + 1 >= P;
        const int t0n = last ? t0 : slot_first(p + 1), nn = last ? 0 : slot_count(p + 1);     // the pass to come (nn <= NTA)
        int rowc[NTW], rown[NTW];
#pragma unroll
        for (int t = 0; t < NTA; ++t) { rowc[t] = clampv((t0 + t) * 16 + li); rown[t] = clampv((t0n + t) * 16 + li); }

        Frag<D> hf[NTW];
        Frag<D> hb[NTW];                              // (CARRY) the pending tiles' state rows, for their blend
        Frag<D> rq[NTW];                              // the next residual segment of this pass, on its way in (NX > 2)
        WPl<D, FMT> hs[NTW], rhs[NTW];
        float hr[NTW][NRR], rhr[NTW][NRR];
        f32x4 acc_r[NTW][NT], acc_u[NTW][NT], acc_c[NTW][NT];
        GatherRegs G;                                 // (the pass to come)

        // the biases of the gate whose epilogue a stage carries, read from LDS ONCE in front of that stage's products: a read inside a
        // piece is waited for inside the piece -- a full LDS round trip per unit with no other wave to cover it
        f32x4 bz[NT];
        auto load_bias = [&](const float* b) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bz[nt] = (nt * 16 + 4 * kq < D) ? ld4(b + nt * 16 + 4 * kq) : f32x4{0.f, 0.f, 0.f, 0.f};
        };
        // ... or, inside the product units, a PIECE ahead of its use (one float4 in flight instead of NT of them held for a stage)
        f32x4 bq[NT * NTW + 1];                       // bq[k]: the bias of the epilogue piece k of the running gate
        auto bias_ahead = [&](const float* b, auto kc, auto nkc) {   // the bias of piece k (group k % NT) if k < nk
            constexpr int k = decltype(kc)::value, nk = decltype(nkc)::value, nt = k % NT;
            if constexpr (k < nk) bq[k] = (nt * 16 + 4 * kq < D) ? ld4(b + nt * 16 + 4 * kq) : f32x4{0.f, 0.f, 0.f, 0.f};
        };
        // ---- the side work of a pass, in PIECES (wide_stage_mma's fill: the pieces of a unit are emitted inside its region) ------
        auto touch4 = [](const f32x4& v) { asm volatile("" :: "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w)); };   // "this value exists HERE" (no sinking)
        constexpr int NC2 = SplitCfg<D, FMT>::NC2;
        // the state's planes: piece k = (tile k / NC2, chunk k % NC2)
        constexpr int H_PIECES = NC2 * NTA;
        auto h_piece = [&](auto kc) {
            constexpr int k = decltype(kc)::value, t = k / NC2, c2 = k % NC2;
            wsplit_chunk<D, FMT>(hs[t], c2, hf[t].v[2 * c2], hf[t].v[2 * c2 + 1]);
            if constexpr (c2 == 0) {
#pragma unroll
                for (int q = 0; q < NR; ++q) hr[t][q] = hf[t].r[q];
            }
        };
        // r = sigmoid(.) group by group, then r*h -> planes chunk by chunk, then the remainder columns:
        // pieces [0, NT NTA) | [NT NTA, (NT + NC2) NTA) | [(NT + NC2) NTA, (NT + NC2 + 1) NTA)
        constexpr int R_PIECES = (NT + NC2 + 1) * NTA;
        auto r_piece = [&](auto kc) {
            constexpr int k = decltype(kc)::value;
            if constexpr (k < NT * NTA) {
                constexpr int t = k / NT, nt = k % NT;
                if constexpr (nt == 0 && C::TAILPACK) {
                    // the u gate's last tile was accumulated in the padding columns of the r gate's last tile:
                    // column D + j of that tile (lane kq + (D%16)/4) is u column 16*NC + j (lane kq)
                    constexpr int SH = 16 * ((D % 16) / 4);
                    f32x4 ut;
                    ut.x = __shfl(acc_r[t][NT - 1].x, lane + SH); ut.y = __shfl(acc_r[t][NT - 1].y, lane + SH);
                    ut.z = __shfl(acc_r[t][NT - 1].z, lane + SH); ut.w = __shfl(acc_r[t][NT - 1].w, lane + SH);
                    acc_u[t][NT - 1] = ut;
                }
                bias_ahead(bias_s, std::integral_constant<int, k + 1>{}, std::integral_constant<int, NT * NTA>{});
                const int col = nt * 16 + 4 * kq;
                if (col < D) {
                    const f32x4 r = w_sigmoid4_acc<FMT>(acc_r[t][nt], bq[k]);
                    acc_r[t][nt] = r;
                    if constexpr (SAVE) st4_b(a.save_r, ((unsigned)rowc[t] * D + col) * 4u, r);
                }
                touch4(acc_r[t][nt]);
            } else if constexpr (k < (NT + NC2) * NTA) {
                constexpr int kk = k - NT * NTA, t = kk / NC2, c2 = kk % NC2;
                f32x4 ra = acc_r[t][2 * c2] * hf[t].v[2 * c2], rb = acc_r[t][2 * c2 + 1] * hf[t].v[2 * c2 + 1];
                // (the product is ROUNDED before it is split: an fma of r, h and the hi piece would split the unrounded product)
                asm volatile("" : "+v"(ra), "+v"(rb));
                wsplit_chunk<D, FMT>(rhs[t], c2, ra, rb);
            } else {
                constexpr int t = k - (NT + NC2) * NTA;
#pragma unroll
                for (int q = 0; q < NR; ++q) {
                    // remainder column 16NC + 4q + kq lives in tile NC of lane (li, kq' = q), element e = kq
                    const float t0_ = __shfl(acc_r[t][NT - 1].x, li + 16 * q), t1_ = __shfl(acc_r[t][NT - 1].y, li + 16 * q);
                    const float t2_ = __shfl(acc_r[t][NT - 1].z, li + 16 * q), t3_ = __shfl(acc_r[t][NT - 1].w, li + 16 * q);
                    const float rr = kq == 0 ? t0_ : (kq == 1 ? t1_ : (kq == 2 ? t2_ : t3_));
                    rhr[t][q] = rr * hf[t].r[q];
                    asm volatile("" :: "v"(rhr[t][q]));
                }
            }
        };
        // u = sigmoid(.): piece k = (tile k / NT, group k % NT)
        constexpr int U_PIECES = NT * NTA;
        auto u_piece = [&](auto kc) {
            constexpr int k = decltype(kc)::value, t = k / NT, nt = k % NT;
            bias_ahead(bias_s + D, std::integral_constant<int, k + 1>{}, std::integral_constant<int, U_PIECES>{});
            const int col = nt * 16 + 4 * kq;
            if (col < D) {
                const f32x4 u = w_sigmoid4_acc<FMT>(acc_u[t][nt], bq[k]);
                acc_u[t][nt] = u;
                if constexpr (SAVE) st4_b(a.save_u, ((unsigned)rowc[t] * D + col) * 4u, u);
            }
            touch4(acc_u[t][nt]);
        };
        // the pending candidate epilogue: piece k = (tile k / NT, group k % NT)
        constexpr int C_PIECES = CARRY ? NT * NTA : 0;
        // (its state rows are read again a piece ahead of their use, one load per piece: hb as a whole would hold 25 registers per
        // tile through the stage -- the registers whose spills a wave alone on its SIMD cannot afford: a scratch reload waits, in
        // order, behind every DMA piece and gathered row in flight)
        auto hb_load = [&](auto kc) {
            constexpr int k = decltype(kc)::value, t = k / NT, nt = k % NT;
            if constexpr (k < C_PIECES) {
                if constexpr (nt < NC) load_frag_part<D, nt < NC ? nt : 0>(hb[t], a.h, prow[t], kq);
                else sfor<0, NR>([&](auto qc) { load_frag_part<D, NC + decltype(qc)::value>(hb[t], a.h, prow[t], kq); });
            }
        };
        auto c_piece = [&](auto kc) {
            constexpr int k = decltype(kc)::value;
            bias_ahead(TANH ? bias_s + 2 * D : bias_s + 3 * D, std::integral_constant<int, k + 1>{}, std::integral_constant<int, C_PIECES>{});
            c_group(std::integral_constant<int, k / NT>{}, std::integral_constant<int, k % NT>{}, hb[k / NT], bq[k]);
        };
        // the gathered segment of the pass to come, tile t: [0] the last pipelined slot, any further ones, an empty row;
        // [1 .. NC2] the mean of a 32-chunk -> its planes; [NC2 + 1] the remainder columns (and the residual segment's planes)
        constexpr int G_PER = NC2 + 2, G_PIECES = G_PER * NTA;
        auto dvq = [&](int t, float x) { const float q = x * G.rcp[t]; return fmaf(fmaf(-G.den[t], q, x), G.rcp[t], q); };   // x / den, one division per row (Markstein)
        auto dv4 = [&](int t, f32x4 v) { return a.g_use_avg ? f32x4{dvq(t, v.x), dvq(t, v.y), dvq(t, v.z), dvq(t, v.w)} : v; };
        auto g_piece = [&](auto kc) {
            constexpr int k = decltype(kc)::value, t = k / G_PER, w = k % G_PER;
            Frag<D>& f = G.x[t];
            if constexpr (w == 0) {
                if (G.beg[t] + KI - 1 < G.end[t]) frag_add_w(f, G.t[t]);
                for (int e = G.beg[t] + KI; e < G.end[t]; ++e) {
                    load_frag<D>(G.t[t], a.g_H, a.g_idx[e], kq);
                    frag_add_w(f, G.t[t]);
                }
                if (G.beg[t] >= G.end[t]) {                   // a node without incoming messages (slot 0 was row 0)
#pragma unroll
                    for (int c = 0; c < NC; ++c) f.v[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int q = 0; q < NR; ++q) f.r[q] = 0.f;
                }
            } else if constexpr (w <= NC2) {
                constexpr int c2 = w - 1;
                const f32x4 va = dv4(t, f.v[2 * c2]), vb = dv4(t, f.v[2 * c2 + 1]);
                if constexpr (SAVE) {
                    if (t < nn) {
                        const unsigned ob = ((unsigned)rown[t] * (unsigned)D + 4u * (unsigned)kq) * 4u;
                        st4_b(a.save_x, ob + 64u * (2 * c2), va); st4_b(a.save_x, ob + 64u * (2 * c2 + 1), vb);
                    }
                }
                wsplit_chunk<D, FMT>(xs[t], c2, va, vb);
            } else {
#pragma unroll
                for (int c = 2 * NC2; c < NC; ++c) f.v[c] = dv4(t, f.v[c]);       // (no odd 16-chunk at the supported sizes)
#pragma unroll
                for (int q = 0; q < NR; ++q) {
                    const float v = a.g_use_avg ? dvq(t, f.r[q]) : f.r[q];
                    if constexpr (SAVE) {
                        if (t < nn) *reinterpret_cast<float*>(reinterpret_cast<char*>(a.save_x) + ((unsigned)rown[t] * (unsigned)D + 16u * NC + 4u * q + (unsigned)kq) * 4u) = v;
                    }
                    xr[t][q] = v;
                }
                if constexpr (NX > 1) commit(xq[t], xqr[t], G.rf[t]);
            }
        };
        static_assert(NC % 2 == 0, "whole 32-chunks");
        // the source rows of the pass to come, ONE LOAD per piece (a load rides in a product unit for free; in front of the products
        // a wave alone on its SIMD pays ~100 clocks of issue for each): piece m = (slot m / RL, tile, part of the row); slot 0 lands
        // in the sum itself, slots 1 .. 3 in the temporary, whose previous slot is added first (slot order = the reference's)
        constexpr int RLP = NC + NR, RL = RLP * NTA;
        auto gl_piece = [&](auto mc) {
            constexpr int m = decltype(mc)::value, slot = m / RL, t = (m % RL) / RLP, part = m % RLP;
            if constexpr (slot == 0) load_frag_part<D, part>(G.x[t], a.g_H, G.i[t][0], kq);
            else {
                if constexpr (part == 0 && slot >= 2) { if (G.beg[t] + slot - 1 < G.end[t]) frag_add_w(G.x[t], G.t[t]); }
                load_frag_part<D, part>(G.t[t], a.g_H, G.i[t][slot < KI ? slot : 0], kq);
            }
        };


        sfor<0, NSTAGE>([&](auto jc) {
            constexpr int j = decltype(jc)::value, g = wide_gate<NX>(j), s = wide_seg<NX>(j);
            // where the side work rides, by the order of the last part: the state's planes in the stage BEFORE h -> r, the r epilogue
            // behind h -> r (two stages | one), the u epilogue behind h -> u, the tail tile of the candidate set zeroed before its first stage
            constexpr int J_HP = GGNN_WIDE_ORDER ? JG + 2 : JG;      // h pieces (last units of this stage)
            constexpr int J_R0 = GGNN_WIDE_ORDER ? JG + 4 : JG + 2;  // r pieces, first stage
            constexpr int J_R1 = GGNN_WIDE_ORDER ? -1 : JG + 3;      // ... second stage (none in order 1)
            constexpr int J_U = GGNN_WIDE_ORDER ? JG + 5 : JG + 4;   // u pieces
            constexpr int J_C0 = GGNN_WIDE_ORDER ? JG + 2 : JG + 4;  // first candidate stage of the last part
            GGNN_WT(4 * (j < 6 ? j : 5) + 0)
            // REMAT: everything derived from the lane coordinates (LDS and row addresses) is recomputed per stage instead of living --
            // and, in this kernel at the edge of the register file, being spilled -- across the pass: a scratch reload waits, in
            // order, behind every DMA piece and gathered row in flight
            asm volatile("" : "+v"(li), "+v"(kq), "+v"(lane));
            // ---- side work in front of the stage's products: loads (they land under the products) ----
            if constexpr (j < JG && j % 3 == 0) {
                // residual segment s: its planes (segment 0 crossed the pass boundary as planes); the next one is requested
                if constexpr (s > 0) {
#pragma unroll
                    for (int t = 0; t < NTA; ++t) commit(xq[t], xqr[t], rq[t]);
                }
                if constexpr (s + 1 < NX - 1) {
#pragma unroll
                    for (int t = 0; t < NTA; ++t) load_frag<D>(rq[t], a.x[s + 1 < 2 ? s + 1 : 0], rowc[t], kq);
                }
            }
            if constexpr (j == (JG > 0 ? JG - 1 : 0)) {               // the state, a stage before its planes are made (NX = 1: with the first stage)
#pragma unroll
                for (int t = 0; t < NTA; ++t) load_frag<D>(hf[t], a.h, rowc[t], kq);
            }
            if constexpr (j == (NX > 1 ? 2 : J_C0)) {                // (before the first candidate stage: residual 0 -> c, or x -> c)
                if constexpr (C::TAILPACK3) {
#pragma unroll
                    for (int t = 0; t < NTA; ++t) acc_c[t][NT - 1] = f32x4{0.f, 0.f, 0.f, 0.f};   // (opened by the r*h stage only)
                }
            }
            if constexpr (j == J_R0) bias_ahead(bias_s, std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});       // r gate
            if constexpr (j == J_U) bias_ahead(bias_s + D, std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});    // u gate
            // the gather of the pass to come: one level of its dependent chain per stage, for every tile (a tile the pass to come
            // does not have gathers a valid row for nothing -- no branches in the product stream)
            if constexpr (j == JG + 1) {
#pragma unroll
                for (int t = 0; t < NTA; ++t) g_ptrs(G, t, rown[t]);
            }
            if constexpr (j == JG + 2) {
#pragma unroll
                for (int t = 0; t < NTA; ++t) g_index(G, t);
            }
            if constexpr (j == JG + 3 && NX > 1) {
#pragma unroll
                for (int t = 0; t < NTA; ++t) load_frag<D>(G.rf[t], a.x[0], rown[t], kq);
            }
            if constexpr (HALF) dma_hb(img_of(j) + SCH::HA, ring + SLOT);          // (this image's other half, under sub-stage A)
            else if constexpr (!GGNN_WIDE_DMA_PIECES) dma(img_of((j + 1) % NSTAGE), ring + (cur ^ 1) * I::IMG);
            __builtin_amdgcn_sched_barrier(0);
            GGNN_WT(4 * (j < 6 ? j : 5) + 1)
            // ---- the stage's products, with the pieces of the pass's side work in their units ----
            // u stages skip their last tile when it rides in the r image; so do the candidate stages of the x segments
            constexpr int ntl = ((C::TAILPACK && g == 1) || (C::TAILPACK3 && g == 2 && s < NX)) ? NT - 1 : NT;
            constexpr bool ZERO = (s == 0);
            constexpr int UN = wide_units<D, FMT>(ntl);              // units of this stage
            constexpr int UNF = wide_units<D, FMT>(NT);              // units of a full stage (x -> r, h -> r, r*h -> c)
            constexpr int UN2 = wide_units<D, FMT>(C::TAILPACK ? NT - 1 : NT);   // units of stage JG + 2 (x -> u)
            constexpr int PW = I::IMG_BYTES / (NW * 1024);           // this wave's KiB pieces of an image DMA
            constexpr int C0 = 2;                                    // first unit of the pending epilogue
            constexpr int H0 = UNF - H_PIECES;                       // first unit of the state's planes in stage JG
            constexpr int G0 = UNF - G_PIECES - 1;                   // first unit of the gathered segment's pieces in the last stage
            static_assert(C0 + C_PIECES <= H0 + UNF && R_PIECES <= (GGNN_WIDE_ORDER ? 1 : 2) * UN2 && U_PIECES <= wide_units<D, FMT>(C::TAILPACK3 ? NT - 1 : NT) &&
                          G0 >= RL + 2 && RL <= wide_units<D, FMT>(C::TAILPACK3 ? NT - 1 : NT) && PW <= wide_units<D, FMT>(NT - 1),
                          "the side work must fit the units of its stages");
            const float* img = ring + cur * SLOT;
            const float* nsrc = img_of((j + 1) % NSTAGE);            // the next image of the sequence (behind the last stage of the last
            float* ndst = ring + (cur ^ 1) * I::IMG;                 // pass: an image nobody reads -- no branch in the product stream)
            auto fill = [&](auto uc) {
                constexpr int u = decltype(uc)::value;
                // the image DMA, a KiB piece at a time: piece k in unit k UN / PW
                if constexpr (GGNN_WIDE_DMA_PIECES && !HALF) {
                    sfor<0, PW>([&](auto kc) {
                        constexpr int k = decltype(kc)::value;
                        if constexpr ((k * UN) / PW == u) dma_piece_asm<I::IMG_BYTES, NW, k>(nsrc, ndst, wave, lane);
                    });
                }
                if constexpr (j == JG) {                             // first stage of the last part: the pending epilogue (CARRY)
                    // (ALL the pending tiles' state rows before the first h' store: the load counter retires in order and stores count
                    // in it -- a row load issued behind a store is waited for behind that store's write acknowledgement)
                    if constexpr (CARRY && u < C0) sfor<0, C_PIECES>([&](auto kc) { if constexpr ((decltype(kc)::value * C0) / (C_PIECES > 0 ? C_PIECES : 1) == u) hb_load(kc); });
                    if constexpr (CARRY && u == C0 - 1) {
                        bias_ahead(TANH ? bias_s + 2 * D : bias_s + 3 * D, std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
                    }
                    if constexpr (u >= C0 && u < H0 && u - C0 < C_PIECES) c_piece(std::integral_constant<int, u - C0>{});
                }
                if constexpr (j == JG + 1) {                         // ... the rest of it
                    if constexpr (u + (H0 - C0) < C_PIECES) c_piece(std::integral_constant<int, u + (H0 - C0)>{});
                }
                if constexpr (j == J_HP) {                           // the state's planes in the LAST units of the stage before h -> r (the rows have landed)
                    if constexpr (u >= UN - H_PIECES) h_piece(std::integral_constant<int, u - (UN - H_PIECES)>{});
                }
                if constexpr (j == J_R0) {                           // the r epilogue behind h -> r
                    if constexpr (u < R_PIECES) r_piece(uc);
                }
                if constexpr (j == J_R1) {
                    if constexpr (u + UN2 < R_PIECES) r_piece(std::integral_constant<int, u + UN2>{});
                }
                if constexpr (j == J_U) {                            // the u epilogue behind h -> u
                    if constexpr (u < U_PIECES) u_piece(uc);
                }
                // the gathered segment of the pass to come: the rows of slots 0, 1 | slot 1 added, the rows of slot 2 | slot 2 added, the
                // rows of slot 3, then the finished sum -> planes
                if constexpr (j == JG + 3) {
                    sfor<0, 2 * RL>([&](auto mc) { if constexpr ((decltype(mc)::value * UN) / (2 * RL) == u) gl_piece(mc); });
                } else if constexpr (j == JG + 4) {
                    if constexpr (u >= UN - RL) gl_piece(std::integral_constant<int, 2 * RL + (u - (UN - RL))>{});
                } else if constexpr (j == JG + 5) {
                    if constexpr (u >= 1 && u - 1 < RL) gl_piece(std::integral_constant<int, 3 * RL + (u - 1)>{});
                    if constexpr (u >= G0 && u - G0 < G_PIECES) g_piece(std::integral_constant<int, u - G0>{});
                }
            };
            // tiles [t0, t1) of the stage from the image halves at ia / ib, the first unit numbered u0
            auto mma = [&](auto& acc, auto zc, auto t0c, auto t1c, auto u0c, const float* ia, const float* ib) {
                constexpr bool Z = decltype(zc)::value;
                constexpr int t0 = decltype(t0c)::value, t1 = decltype(t1c)::value, u0 = decltype(u0c)::value;
                if constexpr (s < NX - 1) wide_stage_mma_at<D, NTW, NTA, t0, t1, Z, FMT, u0>(acc, xq, xqr, ia, ib, li, kq, fill);
                else if constexpr (s == NX - 1) wide_stage_mma_at<D, NTW, NTA, t0, t1, Z, FMT, u0>(acc, xs, xr, ia, ib, li, kq, fill);
                else if constexpr (g < 2) wide_stage_mma_at<D, NTW, NTA, t0, t1, Z, FMT, u0>(acc, hs, hr, ia, ib, li, kq, fill);
                else wide_stage_mma_at<D, NTW, NTA, t0, t1, false, FMT, u0>(acc, rhs, rhr, ia, ib, li, kq, fill);
            };
            auto mma_g = [&](auto t0c, auto t1c, auto u0c, const float* ia, const float* ib) {
                if constexpr (g == 0) mma(acc_r, std::integral_constant<bool, ZERO>{}, t0c, t1c, u0c, ia, ib);
                else if constexpr (g == 1) mma(acc_u, std::integral_constant<bool, ZERO>{}, t0c, t1c, u0c, ia, ib);
                else mma(acc_c, std::integral_constant<bool, ZERO>{}, t0c, t1c, u0c, ia, ib);
            };
            using ic0 = std::integral_constant<int, 0>;
            if constexpr (!HALF) {
                mma_g(ic0{}, std::integral_constant<int, ntl>{}, ic0{}, img, img + SCH::HA);
                __builtin_amdgcn_sched_barrier(0);
                GGNN_WT(4 * (j < 6 ? j : 5) + 2)
                publish();
                GGNN_WT(4 * (j < 6 ? j : 5) + 3)
                cur ^= 1;
            } else {
                constexpr int ta = ntl < SCH::TA ? ntl : SCH::TA;                   // tiles of sub-stage A
                mma_g(ic0{}, std::integral_constant<int, ta>{}, ic0{}, ring, ring);
                __builtin_amdgcn_sched_barrier(0);
                publish();                                                            // (half B has landed; slot 0 is free)
                dma_ha(img_of((j + 1) % NSTAGE), ring);                              // the next image's first half, under sub-stage B
                __builtin_amdgcn_sched_barrier(0);
                mma_g(std::integral_constant<int, ta>{}, std::integral_constant<int, ntl>{}, std::integral_constant<int, wide_units<D, FMT>(ta)>{}, ring, ring + SLOT);
                __builtin_amdgcn_sched_barrier(0);
                GGNN_WT(4 * (j < 6 ? j : 5) + 2)
                publish();
                GGNN_WT(4 * (j < 6 ? j : 5) + 3)
            }
        });

        // this pass's candidate epilogue is now pending
#pragma unroll
        for (int t = 0; t < NTA; ++t) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) { pc[t][nt] = acc_c[t][nt]; pu[t][nt] = acc_u[t][nt]; }
            ptl[t] = acc_r[t][NT - 1]; prow[t] = rowc[t];
        }
        pend_clear(NTA);
        npend = NTA;
        GGNN_WT(25)
    };

    // a pass of a wave without tiles: its share of the image DMAs and the barriers
    auto idle_pass = [&](const int p) {
        for (int j = 0; j < NSTAGE; ++j) {
            if constexpr (HALF) {
                dma_hb(img_of(j) + SCH::HA, ring + SLOT);
                publish();
                dma_ha(img_of((j + 1) % NSTAGE), ring);
                publish();
            } else {
                dma(img_of((j + 1) % NSTAGE), ring + (cur ^ 1) * I::IMG);
                publish();
                cur ^= 1;
            }
        }
    };

    for (int p = 0; p < P; ++p) {
        const int n = slot_count(p);
        const bool carry = GGNN_WIDE_CARRY && n > 0 && npend == n;
        if (!carry && npend > 0) flush();
        if (n == 0) { pend_clear(0); idle_pass(p); continue; }
        sfor<1, NTW + 1>([&](auto nc) {
            if (n == decltype(nc)::value) {
                if (carry) run_pass(nc, std::true_type{}, p);
                else run_pass(nc, std::false_type{}, p);
            }
        });
    }
    flush();
}

template <int D, int NX, int NTW, int FMT, bool SAVE, bool TANH, int NW = 4, bool HALF = false>
int launch_gru_wide_m(const GruFusedArgs& f, float* packed, hipStream_t st) {
    using I = ImgCfg<D, true, FMT>;
    if ((unsigned long long)f.V * D >= (1ULL << 30) || (unsigned long long)f.V * f.g_T * D >= (1ULL << 30))
        return fail(GGNN_E_UNSUPPORTED, "fused GRU indexes with 32-bit byte offsets: V*D and V*T*D must be < 2^30 (V=%d, D=%d)", f.V, D);
    GruWideArgs a{};
    for (int s = 0; s < 2; ++s) a.x[s] = s + 1 < NX ? f.x[s] : nullptr;
    a.h = f.h; a.bg = f.bg; a.bc = f.bc; a.h_out = f.h_out;
    a.save_r = f.save_r; a.save_u = f.save_u; a.save_c = f.save_c; a.save_x = f.save_x;
    a.g_H = f.g_H; a.g_row_ptr = f.g_row_ptr; a.g_idx = f.g_idx; a.g_nin = f.g_nin;
    a.packed = packed; a.V = f.V; a.act = f.act; a.g_T = f.g_T; a.g_use_avg = f.g_use_avg; a.tickets = f.tickets;
#if GGNN_WIDE_STAMPS
    { const char* e = getenv("GGNN_GRU_TPTR"); a.tdbg = (e && NX == 1) ? (unsigned long long*)strtoull(e, nullptr, 10) : nullptr; }
#endif
    constexpr size_t lds = (size_t)2 * (HALF ? SplitCfg<D, FMT>::HA_BYTES : I::IMG_BYTES) + (size_t)((4 * D + 63) / 64 * 64) * sizeof(float);
    const int wt_total = (f.V + 15) / 16;
    int nb = HALF ? 2 * num_cus() : num_cus();
    if (nb > wt_total) nb = wt_total;
    static std::atomic<unsigned long long> lds_ok{0};        // (one per template instantiation)
    GGNN_CHECK_HIP(allow_dynamic_lds(&ggnn_gru_wide_kernel<D, NX, NTW, FMT, SAVE, TANH, NW, HALF>, lds, lds_ok));
    hipLaunchKernelGGL((ggnn_gru_wide_kernel<D, NX, NTW, FMT, SAVE, TANH, NW, HALF>), dim3(nb), dim3(NW * 64), lds, st, a);
    GGNN_CHECK_HIP(hipGetLastError());
    return GGNN_OK;
}

template <int D, int NX, int NTW, int FMT, int NW = 4, bool HALF = false>
int launch_gru_wide(const GruFusedArgs& f, float* packed, hipStream_t st) {
    return launch_gru_wide_m<D, NX, NTW, FMT, false, true, NW, HALF>(f, packed, st);
}

}  // namespace

// What the wide form is instantiated for: the inference launch (no r / u / c stores) of the tanh cell at hidden size 100 -- the
// benchmark's launches; training launches, ReLU cells and the other fused sizes stay on the ring forms (each instantiation of
// this kernel is 2 NTW unrolled pass bodies: minutes of compile time apiece).
// GGNN_WIDE_SET: 1 = nx 1 in the two-piece f16 format only (kernel experiments), 2 = + the residual fan-ins, 3 = + bf16x3.
#ifndef GGNN_WIDE_SET
#define GGNN_WIDE_SET 1
#endif
#ifndef GGNN_WIDE_BF16
#define GGNN_WIDE_BF16 0      // 1: also nx 1 in the exact bf16x3 format (experiments)
#endif
int gru_wide_supported(int D, int nx, const GruFusedArgs& a) {
    if (D != 100 || !a.g_H || a.save_r || a.act != GGNN_ACT_TANH || nx < 1 || nx > 3) return 0;
    const bool f2 = gru_launch_fmt(a.fmt) == kSplitF16x2;
    if (GGNN_WIDE_SET < 3 && !f2 && !(GGNN_WIDE_BF16 && nx == 1)) return 0;
    if (GGNN_WIDE_SET < 2 && nx > 1) return 0;
    return 1;
}

// ntw_req (experiments: GGNN_GRU_FORM = 60 + NTW): 0 = the default tiles per wave
int gru_wide_launch(int D, int nx, int ntw_req, const GruFusedArgs& a, float* packed, hipStream_t st) {
    if (!gru_wide_supported(D, nx, a)) return fail(GGNN_E_UNSUPPORTED, "no wide fused GRU for this launch");
    const bool f2 = gru_launch_fmt(a.fmt) == kSplitF16x2;
#ifdef GGNN_WIDE_PROBE_NTW   // register-allocation probe: one instantiation (-DGGNN_WIDE_PROBE_NTW=3 [-DGGNN_WIDE_PROBE_NX=1] [-DGGNN_WIDE_PROBE_FMT=2])
#ifndef GGNN_WIDE_PROBE_NX
#define GGNN_WIDE_PROBE_NX 1
#endif
#ifndef GGNN_WIDE_PROBE_FMT
#define GGNN_WIDE_PROBE_FMT kSplitF16x2
#endif
    (void)f2; (void)ntw_req;
    return launch_gru_wide<100, GGNN_WIDE_PROBE_NX, GGNN_WIDE_PROBE_NTW, GGNN_WIDE_PROBE_FMT>(a, packed, st);
#else
    if (f2) {
        switch (nx) {
            case 1: return ntw_req == 1 ? launch_gru_wide<100, 1, 1, kSplitF16x2, 8>(a, packed, st)
                         : ntw_req == 4 ? launch_gru_wide<100, 1, 1, kSplitF16x2, 4, true>(a, packed, st)      // (form 64: the half-image form)
                                        : launch_gru_wide<100, 1, 2, kSplitF16x2>(a, packed, st);
#if GGNN_WIDE_SET >= 2
            case 2: return launch_gru_wide<100, 2, 2, kSplitF16x2>(a, packed, st);
            case 3: return launch_gru_wide<100, 3, 2, kSplitF16x2>(a, packed, st);
#endif
        }
    }
#if GGNN_WIDE_BF16 && GGNN_WIDE_SET < 3
    else if (nx == 1) return ntw_req == 1 ? launch_gru_wide<100, 1, 1, kSplitBf16x3, 8>(a, packed, st) : launch_gru_wide<100, 1, 2, kSplitBf16x3>(a, packed, st);
#endif
#if GGNN_WIDE_SET >= 3
    else {
        switch (nx) {
            case 1: return launch_gru_wide<100, 1, 2, kSplitBf16x3>(a, packed, st);
            case 2: return launch_gru_wide<100, 2, 2, kSplitBf16x3>(a, packed, st);
            case 3: return launch_gru_wide<100, 3, 1, kSplitBf16x3>(a, packed, st);
        }
    }
#endif
    return fail(GGNN_E_INVALID, "nx %d outside 1..3", nx);
#endif
}

}  // namespace ggnn
