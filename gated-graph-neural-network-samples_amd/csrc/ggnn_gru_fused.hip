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
// Wc).  Stages are brought in by LDS-DMA into a 2-deep ring; the DMA of stage i+1 is issued during stage i.
// Workgroups are persistent: one per CU, each working through "tickets" of 8 tiles (one pass), see the kernel.
//
// Cost model: on gfx950 the vector ALU and the FP32 MFMA share their issue slot -- every VALU
// instruction of the kernel costs matrix-pipe time -- so the loop is written to need few of them: byte-offset
// addressing with scalar bases, accumulators opened with C = 0, biases folded into the exponent scaling, one
// division per gathered row.
//
// Stage ORDER of one pass (3 * (NX+1) stages): every input segment is consumed by three consecutive stages
//     x_s -> r columns, x_s -> u columns, x_s -> candidate columns          (s = 0 .. NX-1)
//     h   -> r columns, h   -> u columns, [gates epilogue], r*h -> candidate columns
// so each x fragment is read from memory once and dies after its third stage; the fragment of the next
// segment (or of the next pass's first segment) is fetched while the current one is being multiplied.
// All accumulation orders are those of the plain [x|h] / [x|r*h] products (segments in order, h last).
#include "ggnn_split.hpp"
#include <type_traits>

#ifndef GGNN_COOP_DEPTH
#define GGNN_COOP_DEPTH 1
#endif
// The s_memtime stamps behind GGNN_GRU_TPTR (tools/gru_timeline.py) cost scalar registers and branches in the hot loop -- 20 B of
// the split R >= 1 kernels' scratch: compiled in only with -DGGNN_GRU_STAMPS=1 (tools/variant_lib.sh builds such a library).
#ifndef GGNN_GRU_STAMPS
#define GGNN_GRU_STAMPS 0
#endif
#if GGNN_GRU_STAMPS
#define GGNN_TDBG(a) (a).tdbg
#else
#define GGNN_TDBG(a) ((unsigned long long*)nullptr)
#endif

namespace ggnn {

// large hidden sizes (multiples of 64 from 128): column-panel kernels, ggnn_panel.hip
int gru_panel_supported(int D);
int gru_panel_pack_floats(int D, int nx);
int gru_panel_dispatch(const GruFusedArgs& a, int D, float* packed, hipStream_t st);

// This source is compiled TWICE: as itself (the f32-MFMA instantiations, the dispatch, the C entry points) and, through
// ggnn_gru_fused_split.hip (GGNN_GRU_TU_SPLIT), for the SPLIT instantiations -- that translation unit is built without
// packed-f32 vector instructions (build.py: a v_pk_* instruction beside the partner wave's bf16 MFMAs stalls the SIMD).
int gru_split_launch(int D, int nx, bool save, bool gather, const GruFusedArgs& a, float* packed, hipStream_t st);

#ifndef GGNN_GRU_TU_SPLIT
int gru_pack_floats(int D, int nx) {
    if (gru_panel_supported(D)) return gru_panel_pack_floats(D, nx);
    // (the operand format is chosen per pack / launch: the buffer is sized for the larger, three-plane bf16 images)
    const bool sp = split_matrix_path();
    static_assert(ImgCfg<100, true>::IMG >= ImgCfg<100, true, kSplitF16x2>::IMG && ImgCfg<64, true>::IMG >= ImgCfg<64, true, kSplitF16x2>::IMG &&
                  ImgCfg<32, true>::IMG >= ImgCfg<32, true, kSplitF16x2>::IMG, "bf16x3 images are the larger ones");
    switch (D) {
        case 100: return 3 * (nx + 1) * (sp ? ImgCfg<100, true>::IMG : ImgCfg<100, false>::IMG);
        case 64: return 3 * (nx + 1) * (sp ? ImgCfg<64, true>::IMG : ImgCfg<64, false>::IMG);
        case 32: return 3 * (nx + 1) * (sp ? ImgCfg<32, true>::IMG : ImgCfg<32, false>::IMG);
        default: return 0;
    }
}
#endif

// image ci of the packed weights: gates (s = 0..nx) x {r,u}, then candidate (s = 0..nx)
template <int D, bool SPLIT, int FMT = kSplitBf16x3>
__global__ void gru_pack_weights_kernel(const float* __restrict__ Wg, const float* __restrict__ Wc, int nx,
                                        float* __restrict__ out) {
    const int ci = blockIdx.y;
    float* img = out + (size_t)ci * ImgCfg<D, SPLIT, FMT>::IMG;
    if constexpr (SPLIT) gru_fwd_image_pack_split<D, FMT>(Wg, Wc, nx, ci, img, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x);
    else gru_fwd_image_pack<D>(Wg, Wc, nx, ci, img, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x);
}

// position in the per-pass stage sequence -> packed image (segment s = pos / 3; 0,1: its r / u gate columns, 2: candidate)
template <int NX>
__host__ __device__ constexpr int gru_stage_image(int pos) {
    return (pos % 3) < 2 ? 2 * (pos / 3) + (pos % 3) : 2 * (NX + 1) + pos / 3;
}

// the gate epilogues on an accumulator that holds 1 / acc_scale x the sum (SplitFmt<FMT>::acc_scale: 2^-8 for the f16 x 2 form, whose
// weights are packed x 2^8; 1 otherwise): sigmoid4_scaled / tanh4_scaled with the scale folded into the exponent's scaling constant
template <int FMT>
__device__ __forceinline__ f32x4 sigmoid4_acc(f32x4 z, f32x4 b_scaled) {
    constexpr float k = -kLog2e * SplitFmt<FMT>::acc_scale;
    return rcp_4(exp2_4(z * k + b_scaled) + 1.0f);
}
template <int FMT>
__device__ __forceinline__ f32x4 tanh4_acc(f32x4 z, f32x4 b_scaled) {
    constexpr float k = 2.0f * kLog2e * SplitFmt<FMT>::acc_scale;
    return 1.0f - 2.0f * rcp_4(exp2_4(z * k + b_scaled) + 1.0f);
}

template <int D>
__device__ __forceinline__ void frag_zero(Frag<D>& f) {
#pragma unroll
    for (int c = 0; c < StageCfg<D>::NC; ++c) f.v[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < StageCfg<D>::NR; ++q) f.r[q] = 0.f;
}

template <int D>
__device__ __forceinline__ void frag_add(Frag<D>& f, const Frag<D>& t) {
#pragma unroll
    for (int c = 0; c < StageCfg<D>::NC; ++c) f.v[c] += t.v[c];
#pragma unroll
    for (int q = 0; q < StageCfg<D>::NR; ++q) f.r[q] += t.r[q];
}

// SAVE: also write r, u, c (all three) for a backward pass.
// GATHER: the LAST x segment (the aggregated messages, chem_tensorflow_sparse.py:198-212) is not read from memory but
// computed on the fly: incoming[row] = (sum over the row's message slots of Hrows[gather_row[slot]]) / (deg + 1e-7),
// same slot order and arithmetic as ggnn_gather_segment_sum_f32 (bit-identical) -- the separate segment-sum launch and
// the [V,D] round trip of `incoming` through HBM disappear.  The gather is a 3-level dependent chain (row_ptr ->
// gather_row -> rows); it is software-pipelined over the stage boundaries BEFORE the stage that consumes it: every
// level is issued at the start of a stage and has landed by that stage's closing barrier, so the MFMAs never wait
// for it (only slots beyond the pipelined depth, 4 per row = the largest valence in QM9, are fetched synchronously).
// SPLIT: the products run on the bf16 matrix pipe in 3-way split form (ggnn_split.hpp) -- same stages, same fragments, same
// accumulators; the stage images are the split ones and every activation fragment is split in registers before its first stage.
// SAVEX: the gathered segment is stored too (save_x; training).  The split-form inference dispatch runs <SAVE = true, SAVEX =
// false>: the instantiation with the r / u / c stores (skipped at run time) but without the save_x path comes out of the register
// allocator with the least scratch (R = 1 / 2: 8 / 36 B; with it 28 / 40 B; SAVE = false: 164 / 196 B).
// FORM: how a pass's stage images come through the LDS ring.
//   0  one 8-wave workgroup per CU, whole images, 2 slots; the barrier that closes a stage waits for EVERY load the wave has
//      issued (vmcnt(0): the next image's DMA, but also the stage's gather / fragment fetches).
//   1  (HALF) two 4-wave workgroups per CU, each with a 2-slot ring of HALF images (ggnn_split.hpp: half A = tiles [0, TA), half
//      B the rest; a stage = two sub-stages with a barrier each): the two waves of a SIMD belong to different workgroups and
//      meet at no barrier.  Measured (round 4): no faster than form 0 at R = 0, 5-12 % faster with residual inputs.
//   2  (DEEP) one 8-wave workgroup, half images, THREE slots: the DMA of sub-stage k + 2 is issued in sub-stage k, so the barrier
//      that closes sub-stage k only has to know that DMA k + 1 -- a whole sub-stage old -- has landed.  The load counter retires
//      in order; waiting until no more than this sub-stage's own DMA pieces are outstanding (they were issued BEFORE the
//      sub-stage's fetches: "vmcnt(pieces)") proves it, and leaves every gather / fragment fetch of the sub-stage in flight.
//      Ablation that led here (tools/fwd_kernels.py, GGNN_GRU_DBG): the R = 0 launch takes 34 us with the MFMAs, epilogues,
//      splits and DMA all switched off -- 18 stage barriers each waiting out one HBM round trip -- 63 us with only the MFMAs
//      added, 63 us with only the side work added, 88 us with both: the three parts ran one after the other.
// Same products in the same order per accumulator in every form: bit-identical results.
// FMT (SPLIT kernels): operand format of the split products and of the images (ggnn_split.hpp) -- kSplitF16x2 (two f16 pieces, three
// products: half the MFMAs, 48 KiB images; valid inside its operand range) or the exact kSplitBf16x3, chosen per launch by
// GruFusedArgs::fmt; f32 kernels: kSplitBf16x3 stands for "accumulators unscaled".
template <int D, int NX, int NW, bool SAVE, bool GATHER, bool SPLIT, bool SAVEX = SAVE, int FORM = 0, int FMT = kSplitBf16x3>
__global__ __launch_bounds__(NW * 64, FORM == 1 ? 2 : 1) void ggnn_gru_fused_kernel(GruFusedArgs a, const float* __restrict__ packed) {
    static_assert(SPLIT || FMT == kSplitBf16x3, "the f32-MFMA kernels have no operand format");
    using C = StageCfg<D>;
    using I = ImgCfg<D, SPLIT, FMT>;
    using SC = SplitCfg<D, FMT>;
    constexpr bool HALF = FORM != 0;                                  // stage images travel as halves
    constexpr bool DEEP = FORM == 2;
    constexpr int NSLOT = DEEP ? 3 : 2;
    static_assert(!HALF || SPLIT, "the half-stage forms are split-form kernels");
    static_assert(FORM != 1 || NW == 4, "form 1 is the four-wave kernel");
    constexpr int NT = C::NT, NC = C::NC, NR = C::NR;
    constexpr int NSTAGE = 3 * (NX + 1);
    constexpr int SLOT = HALF ? SC::HA : I::IMG;                     // floats per ring slot
    extern __shared__ __attribute__((aligned(16))) float lds_[];    // [biases | ring [2][IMG]]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kq = lane >> 4;
    if (GGNN_TDBG(a) && tid == 0) {   // debug: per-workgroup [start, end] in shader-clock and 100 MHz real-time ticks
        a.tdbg[4096 + blockIdx.x * 4 + 0] = __builtin_amdgcn_s_memtime();
        a.tdbg[4096 + blockIdx.x * 4 + 1] = __builtin_amdgcn_s_memrealtime();
    }

    // Work is handed out in TICKETS, one per pass of a workgroup.  Full rounds: ticket t = tiles [t*NW, (t+1)*NW),
    // all waves busy.  The rest (< NW*nb tiles) is spread thin: tail tickets of tail_w = ceil(rest / nb) tiles each, so
    // the last round keeps every CU busy for as short as possible instead of a few CUs for a whole pass.
    // Ticket order: workgroup b starts with ticket b; the next ones are b + nb, b + 2nb, ... (a.tickets == NULL), or
    // come from a global counter (a.tickets: a device int that is 0 at launch).  With the counter a workgroup that
    // starts late -- another stream held its CU -- simply takes fewer tickets and the launch ends when the tickets
    // run out, not when the last-started workgroup has worked through a fixed share.
    const int wt_total = (a.V + 15) / 16;
    const int nb = gridDim.x;
    const int full_tk = wt_total / (NW * nb) * nb;
    const int rest = wt_total - full_tk * NW;
    const int tail_w = (rest + nb - 1) / nb;                        // 0 (no tail) .. NW
    const int n_tk = full_tk + (tail_w ? (rest + tail_w - 1) / tail_w : 0);
    // One-tile tail tickets (tail_w == 1, the usual case) are worked COOPERATIVELY: all waves take the same 16 rows and
    // wave w computes output column tile w of every gate (25 MFMAs per stage instead of 175 on one wave while seven
    // idle); the r*h fragment, which the candidate stage needs whole, is exchanged through LDS.
    const bool coop_tail = (tail_w == 1) && (NT <= NW) && !HALF;   // (the half-stage forms run thin tail tickets on single waves)
    auto is_coop = [&](int t) -> bool { return coop_tail && t >= full_tk && t < n_tk; };
    auto tile_of = [&](int t) -> int {                              // this wave's tile of ticket t, or -1
        if (t < full_tk) return t * NW + wave;
        if (t >= n_tk) return -1;
        if (coop_tail) return full_tk * NW + (t - full_tk);          // (every wave: the same tile)
        const int tl = full_tk * NW + (t - full_tk) * tail_w + wave;
        return (wave < tail_w && tl < wt_total) ? tl : -1;
    };

    // biases live in LDS IN FRONT of the ring (small offsets: every epilogue read is one lane register + an
    // immediate; behind the ring each of the 21 addresses was a loop-invariant register of its own, and those
    // were spilled to scratch): the epilogues read them with ds_read instead of 7 serialized
    // global round trips per epilogue
    constexpr int BIAS_FLOATS = (4 * D + 4 + 63) / 64 * 64;
    float* bias_s = lds_;                             // [-log2e*bg (2D) | 2 log2e*bc (D) | bc (D) | next tickets (2)]
    float* ring = lds_ + BIAS_FLOATS;
    int* tk_slot = reinterpret_cast<int*>(bias_s + 4 * D);
    constexpr int RHP = C::BN + 4;                     // row pitch of the r*h exchange block (cooperative tail pass)
    float* rh_x = ring + NSLOT * SLOT;                     // [16][RHP], behind the ring (cooperative tail passes only: not HALF)
    for (int i = tid; i < 4 * D; i += NW * 64)
        bias_s[i] = i < 2 * D ? -kLog2e * a.bg[i] : (i < 3 * D ? 2.0f * kLog2e * a.bc[i - 2 * D] : a.bc[i - 3 * D]);
    // The two waves of a SIMD (w and w + NW/2) do not interleave on the matrix pipe: the older one issues its whole
    // MFMA burst first.  The stage loop leans on that: the waves of the first half ("early") burst first and do
    // their side work afterwards, the second half ("late") the other way round.
    // (form 1: the SIMD partner is another workgroup's wave at a phase of its own: every wave issues its side work first)
    const bool late = FORM == 1 || wave >= NW / 2;
    int tk = blockIdx.x, tk_next = blockIdx.x + nb;                 // current / next pass's ticket (workgroup-uniform)
    if (a.tickets && tid == 0) *tk_slot = nb + atomicAdd(a.tickets, 1);

    int cur = 0;
    // SPLIT: the DMA goes out through inline assembly and is waited for explicitly (dma_image_asm, ggnn_split.hpp)
    auto dma = [&](const float* src, float* dst) {
        if constexpr (SPLIT) dma_image_asm<I::IMG_BYTES, NW>(src, dst, wave, lane);
        else dma_image<I::IMG_BYTES, NW>(src, dst, wave, lane);
    };
    auto publish = [&]() { if constexpr (SPLIT) dma_wait(); __syncthreads(); };
    auto dma_ha = [&](const float* src, float* dst) { dma_kib_asm<SC::HA_BYTES / 1024, NW>(src, dst, wave, lane); };
    auto dma_hb = [&](const float* src, float* dst) { dma_kib_asm<SC::HB_BYTES / 1024, NW>(src, dst, wave, lane); };
    // own pieces of a half-image DMA (dma_kib_asm's split): what the DEEP barrier may leave outstanding
    constexpr int KA = SC::HA_BYTES / 1024, KB = SC::HB_BYTES / 1024;
    const int own_a = KA / NW + (wave < KA % NW ? 1 : 0), own_b = KB / NW + (wave < KB % NW ? 1 : 0);
    // barrier of a DEEP sub-stage: `own` = this wave's pieces of the DMA issued in the sub-stage (0: none was issued)
    auto publish_deep = [&](int own) {
        switch (own) {                                               // (wave-uniform; s_waitcnt takes an immediate)
            case 0: asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); break;
            case 1: asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory"); break;
            case 2: asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory"); break;
            case 3: asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory"); break;
            case 4: asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory"); break;
            case 5: asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)" ::: "memory"); break;
            case 6: asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory"); break;
            case 7: asm volatile("s_waitcnt vmcnt(7) lgkmcnt(0)" ::: "memory"); break;
            case 8: asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory"); break;
            case 9: asm volatile("s_waitcnt vmcnt(9) lgkmcnt(0)" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(10) lgkmcnt(0)" ::: "memory"); break;
        }
        __builtin_amdgcn_s_barrier();
    };
    static_assert(!DEEP || (SC::HA_BYTES / 1024 + NW - 1) / NW <= 10, "publish_deep covers <= 10 pieces per wave");
    if constexpr (DEEP) {                                            // sub-stages 0 and 1 of the first pass
        dma_ha(packed, ring);
        dma_hb(packed + SC::HA, ring + SLOT);
    } else if constexpr (HALF) dma_ha(packed, ring);
    else dma(packed, ring);

    // ---- the pipelined gather of the aggregated-messages segment (GATHER) --------------------------------------
    // Phases, each issued at a stage start and landed by that stage's closing barrier (U = the stage that consumes
    // the fragment, counted in the consuming pass; a negative position lies in the previous pass):
    //     U-5 row_ptr + in-degree | U-4 first KI source rows | U-3 rows of slots 0,1 | U-2 add 1, row of slot 2 |
    //     U-1 add 2, row of slot 3 | U add 3, further slots synchronously, mean
    // NX == 1 consumes at stage 0 of the NEXT pass (U = NSTAGE: the x fragment is dead after stage 2).
    constexpr int GSEG = NX - 1, GBUF = GSEG & 1;     // gathered segment and its fragment buffer
    constexpr int KI = 4;                             // slots covered by the pipeline
    constexpr int G_U = NX == 1 ? NSTAGE : 3 * (NX - 1);
    constexpr bool G_NEXT = (NX == 1);                // all phases run one pass ahead of the consumer
    int g_beg = 0, g_end = 0;
    int g_i[KI] = {0, 0, 0, 0};                       // (row 0 is always a valid row to fetch)
    f32x4 g_n = {0.f, 0.f, 0.f, 0.f};
    float g_den = 1.f, g_rcp = 1.f;
    Frag<D> gt;

    auto g_ptrs = [&](int r, bool on) {               // level 1: slot range + in-degree of row r
        const int b = ldi_b(a.g_row_ptr, (unsigned)r * 4u), e = ldi_b(a.g_row_ptr, (unsigned)r * 4u + 4u);   // r is valid even when !on
        g_beg = on ? b : 0; g_end = on ? e : 0;
        if (a.g_use_avg) {
            if (a.g_T == 4) {
                g_n = ld4_b(a.g_nin, (unsigned)r * 16u);
            } else {
                float deg = 0.f;
                for (int t = 0; t < a.g_T; ++t) deg += a.g_nin[(size_t)r * a.g_T + t];
                g_n = f32x4{deg, 0.f, 0.f, 0.f};
            }
        }
    };
    // Every phase below has a CONSUME half (it needs values an earlier stage fetched: the compiler puts its load-counter wait in
    // front of the first use) and an ISSUE half (new loads).  The DEEP form runs all consume halves of a stage, then its DMA, then
    // all issue halves -- a wait placed behind the DMA would count its pieces among the "younger" loads and stall on them; the
    // other forms run them back to back (part 0).
    auto touch = [](auto v) { asm volatile("" :: "v"(v)); };         // "this value is needed HERE"
    auto g_index = [&](int part) {                    // level 2: the first KI source rows
        // a slot beyond the row's degree points at row 0 of Hrows: the row loads below are then UNCONDITIONAL (no
        // zero-fills, no exec-mask juggling around 7 loads -- vector-ALU instructions cost matrix-pipe time) and only
        // the adds are predicated; Hrows always has at least one row
        if (part != 2) {
            touch(g_beg); touch(g_end);
            g_den = (((g_n.x + g_n.y) + g_n.z) + g_n.w) + 1e-7f;
            if constexpr (NX == 1) g_rcp = 1.0f / g_den;    // the in-degrees landed with the slot range
        }
        if (part != 1) {
#pragma unroll
            for (int j = 0; j < KI; ++j) {
                g_i[j] = 0;
                if (g_beg + j < g_end) g_i[j] = ldi_b(a.g_idx, (unsigned)(g_beg + j) * 4u);
            }
        }
    };
    auto g_rows0 = [&](Frag<D>& f, int part) {        // level 3: slot 0 straight into f, slot 1 into the temporary
        if (part != 2) { touch(g_i[0]); touch(g_i[1]); touch(g_i[2]); touch(g_i[3]); }
        if (part != 1) {
            load_frag<D>(f, a.g_H, g_i[0], kq);       // (0 + slot 0 of the segment-sum kernel: same value, up to -0)
            load_frag<D>(gt, a.g_H, g_i[1], kq);
        }
    };
    auto g_rows = [&](Frag<D>& f, int k, int part) {  // add slot k-1 (landed), fetch slot k
        if (part != 2) { if (g_beg + k - 1 < g_end) frag_add<D>(f, gt); }
        if (part != 1) load_frag<D>(gt, a.g_H, g_i[k], kq);
    };
    auto g_finish = [&](Frag<D>& f) {                 // slot 3, any further slots (synchronously), mean
        if (g_beg + KI - 1 < g_end) frag_add<D>(f, gt);
        for (int e = g_beg + KI; e < g_end; ++e) {
            load_frag<D>(gt, a.g_H, a.g_idx[e], kq);
            frag_add<D>(f, gt);
        }
        if (g_beg >= g_end) frag_zero<D>(f);          // a node without incoming messages (slot 0 was row 0)
        if (a.g_use_avg) {                                            // :206-209
            // x / den for the 25+ values of the row with ONE division: r = RN(1/den), q = RN(x r), then the FMA
            // residual step q + (x - den q) r.  With a correctly rounded r this IS the correctly rounded quotient
            // (Markstein) -- bit-identical to the `/` of the stand-alone kernel, asserted by the tests -- for 3 vector-ALU
            // instructions per value instead of ~11 (vector-ALU work is paid in matrix-pipe time).
            // (NX >= 2 keeps the plain division: one more long-lived register pushes those variants deeper into scratch)
            const float den = g_den, r = g_rcp;
            auto dv = [&](float x) {
                if constexpr (NX == 1) { const float q = x * r; return fmaf(fmaf(-den, q, x), r, q); }
                else return x / den;
            };
#pragma unroll
            for (int c = 0; c < NC; ++c) f.v[c] = f32x4{dv(f.v[c].x), dv(f.v[c].y), dv(f.v[c].z), dv(f.v[c].w)};
#pragma unroll
            for (int q = 0; q < NR; ++q) f.r[q] = dv(f.r[q]);
        }
    };

    // (training) the gathered segment is an operand of the weight gradients: written once, when it is complete
    auto store_x = [&](const Frag<D>& f, int row_) {
        const unsigned ob = ((unsigned)row_ * (unsigned)D + 4u * (unsigned)kq) * 4u;
#pragma unroll
        for (int c = 0; c < NC; ++c) st4_b(a.save_x, ob + 64u * c, f.v[c]);
#pragma unroll
        for (int q = 0; q < NR; ++q) *reinterpret_cast<float*>(reinterpret_cast<char*>(a.save_x) + ob - 16u * (unsigned)kq + (16u * NC + 4u * q + (unsigned)kq) * 4u) = f.r[q];
    };
    // x fragments rotate through two register sets: segment s lives in xf[s & 1].
    Frag<D> hf, xf[2];
    {
        const int t0 = tile_of(tk);
        const bool on = t0 >= 0;
        const int r0 = (on ? t0 : 0) * 16 + li;
        const int r0c = r0 < a.V ? r0 : a.V - 1;
        if constexpr (GATHER) {                       // first tile: the phases of the previous pass, synchronously
            if constexpr (G_NEXT) {
                g_ptrs(r0c, on);
                if (on) { g_index(0); g_rows0(xf[0], 0); g_rows(xf[0], 2, 0); g_rows(xf[0], 3, 0); g_finish(xf[0]); }
                if constexpr (SAVEX) { if (on && r0 < a.V && a.save_x) store_x(xf[0], r0); }
            } else {
                if constexpr (G_U - 5 < 0) g_ptrs(r0c, on);
                if constexpr (G_U - 4 < 0) g_index(0);        // (also for a wave without a tile: its slots -> row 0)
                load_frag<D>(xf[0], a.x[0], r0c, kq);
            }
        } else {
            load_frag<D>(xf[0], a.x[0], r0c, kq);
        }
    }
    publish();                // (f32: hipcc emits vmcnt(0) before the barrier while an LDS-DMA is in flight)

    if (a.tickets) tk_next = __builtin_amdgcn_readfirstlane(*tk_slot);

    // One pass (= one ticket) of the workgroup.  COOP is a compile-time flag: the ordinary passes and the cooperative
    // tail passes are separate instantiations, run by separate loops below, so the register allocation of the hot loop
    // does not see the tail code.
    const int n_main = coop_tail ? full_tk : n_tk;             // tickets [n_main, n_tk) are cooperative tail passes
    // The cooperative pass reads its weights either through the LDS ring like every other pass, or -- COOP_REGS --
    // straight from the images in global memory into registers (no ring, no per-stage barrier).  The register form
    // needs 2 x 25 more VGPRs; with residual inputs (NX >= 2) that pushes loop invariants of the ordinary passes into
    // scratch (measured: NX = 3 launch 224 -> 240 us), so only the single-input kernel uses it (121 -> 120 us).
    constexpr bool COOP_REGS = (NX == 1) && !SPLIT;
    const int n_dma = (coop_tail && COOP_REGS) ? n_main : n_tk; // passes that need the ring filled
    auto run_pass = [&](auto coop_c, const int p) {
        const int tile_ = tile_of(tk);
        constexpr bool coop = decltype(coop_c)::value;         // one tile for the workgroup, wave w -> column tile w
        const bool active = tile_ >= 0;                        // wave-uniform
        const int tile = active ? tile_ : 0;
        const int row = active ? tile * 16 + li : a.V;         // (>= V: nothing is stored)
        const int rowc = row < a.V ? row : a.V - 1;
        const bool last_pass = tk_next >= n_tk;
        const int tile_n = last_pass ? -1 : tile_of(tk_next);
        const bool has_next = tile_n >= 0;
        const int rown_ = (has_next ? tile_n : 0) * 16 + li;
        const int rown = rown_ < a.V ? rown_ : a.V - 1;
        // the ticket after the next one: requested now, published before the last stage's barrier, read after it
        int tk_fetch = 0;
        if (a.tickets && tid == 0) tk_fetch = atomicAdd(a.tickets, 1);

        // what is fetched at the START of stage POS (it lands in the shadow of that stage's MFMAs and is drained
        // by its closing barrier)
        // part 0: everything; 1: the consume halves of the stage's phases; 2: their issue halves (see g_index)
        auto prefetch = [&](auto posc, int part = 0) {
            constexpr int POS = decltype(posc)::value;
            constexpr int s = POS / 3, j = POS % 3;
            if constexpr (GATHER) {
                // a phase at position v < 0 (or any phase when G_NEXT) runs one pass ahead, for the next tile's row
                constexpr int P_PTR = G_U - 5, P_IDX = G_U - 4, P_R0 = G_U - 3, P_R1 = G_U - 2, P_R2 = G_U - 1;
                if constexpr (POS == (P_PTR + NSTAGE) % NSTAGE) {
                    if (part != 1) {
                        if constexpr (G_NEXT || P_PTR < 0) g_ptrs(rown, has_next);
                        else g_ptrs(rowc, active);
                    }
                }
                if constexpr (POS == (P_IDX + NSTAGE) % NSTAGE) g_index(part);
                if constexpr (POS == (P_R0 + NSTAGE) % NSTAGE) g_rows0(xf[GBUF], part);
                if constexpr (POS == (P_R1 + NSTAGE) % NSTAGE) g_rows(xf[GBUF], 2, part);
                if constexpr (POS == (P_R2 + NSTAGE) % NSTAGE) g_rows(xf[GBUF], 3, part);
            }
            if (part != 1) {
                // h: one stage before its first use (kept out of the registers until then)
                if constexpr (POS == 3 * NX - 1) load_frag<D>(hf, a.h, rowc, kq);
                // the next plain segment, one stage before its first use
                if constexpr (j == 2 && s + 1 < NX && !(GATHER && s + 1 == GSEG)) load_frag<D>(xf[(s + 1) & 1], a.x[s + 1], rowc, kq);
                // the next pass's first segment (xf[0] is dead by the last stage)
                if constexpr (POS == NSTAGE - 1 && !(GATHER && GSEG == 0)) load_frag<D>(xf[0], a.x[0], rown, kq);
            }
        };

#if GGNN_GRU_STAMPS
#define GGNN_T(CI, K) if (a.tdbg && blockIdx.x == 0 && lane == 0 && p < 4) a.tdbg[((p * NSTAGE + (CI)) * NW + wave) * 4 + (K)] = __builtin_amdgcn_s_memtime();
#define GGNN_T2(CI, K) if (a.tdbg && blockIdx.x == 0 && lane == 0 && p < 4) a.tdbg[2048 + ((p * NSTAGE + (CI)) * NW + wave) * 4 + (K)] = __builtin_amdgcn_s_memtime();
#else
#define GGNN_T(CI, K)
#define GGNN_T2(CI, K)
#endif
        // one stage: prefetches, start the DMA of the next image, MFMAs on the current one, publish
        // Cooperative tail pass: no ring, no per-stage barrier.  Wave w multiplies by column tile w only, whose weights
        // (1/NT of the image) it reads from the image in global memory into registers TWD stages ahead.
#define GGNN_COOP_STAGE(POS, ACC, FRAG)                                                                  \
        {                                                                                                \
            GGNN_T(POS, 0)                                                                               \
            if constexpr (GATHER && (POS) == G_U % NSTAGE) {                                             \
                if (active && (!G_NEXT || p > 0)) { g_finish(xf[GBUF]); if constexpr (SAVEX) { if (row < a.V && a.save_x) store_x(xf[GBUF], row); } } \
            }                                                                                            \
            if constexpr ((POS) + TWD < NSTAGE) {                                                        \
                if (wave < NT) load_tile_weights<D>(tw[((POS) + TWD) % (TWD + 1)],                       \
                                                    packed + (size_t)gru_stage_image<NX>((POS) + TWD) * I::IMG, li, kq, wave); \
            }                                                                                            \
            prefetch(std::integral_constant<int, (POS)>{});                                              \
            if constexpr ((POS) == 3 * NX) {   /* h columns of this wave's tile, for the two epilogues */ \
                if (wave < NT && wave * 16 + 4 * kq < D) hv_pre = ld4_b(a.h, ((unsigned)rowc * D + wave * 16 + 4 * kq) * 4u); \
            }                                                                                            \
            GGNN_T(POS, 1)                                                                               \
            if (active && wave < NT && !(a.dbg & 1)) {                                                   \
                if (C::TAILPACK3 && (POS) == NSTAGE - 1 && wave == NT - 1) {                             \
                    /* the candidate's last tile: x part + r*h part as two sums, like the tail-packed ordinary passes */ \
                    f32x4 t_;                                                                            \
                    tile_mma_regs<D, true>(t_, FRAG, tw[(POS) % (TWD + 1)]);                             \
                    ACC[0] = ACC[0] + t_;                                                                \
                } else tile_mma_regs<D, ((POS) < 3)>(ACC[0], FRAG, tw[(POS) % (TWD + 1)]);               \
            }                                                                                            \
            GGNN_T(POS, 2)                                                                               \
            if constexpr ((POS) == NSTAGE - 1) __syncthreads();   /* (the ticket slot written before this stage) */ \
            GGNN_T(POS, 3)                                                                               \
        }
        // Forms 1 and 2: the stage in two sub-stages.  Entering, slot `cur` holds half A of this stage's image (published).
        //   form 1: half B of this image is fetched during sub-stage A, half A of the next stage's during sub-stage B; the barriers
        //           wait for everything (vmcnt(0)).
        //   form 2: the DMA runs TWO sub-stages ahead (3 slots): sub-stage A fetches the NEXT stage's half A, sub-stage B its half
        //           B; a sub-stage is  consume | DMA | issue | MFMAs | vmcnt(own pieces) + barrier  (early waves: MFMAs first).
#define GGNN_HALF_STAGE(POS, ACC, FRAG)                                                                  \
        {                                                                                                \
            constexpr int npos_ = (POS) + 1;                                                             \
            const bool more_ = (npos_ < NSTAGE) || tk_next < n_dma;                                      \
            const float* csrc_ = packed + (size_t)gru_stage_image<NX>(POS) * I::IMG;                     \
            const float* nsrc_ = packed + (size_t)gru_stage_image<NX>(npos_ < NSTAGE ? npos_ : 0) * I::IMG; \
            constexpr int ntl_ = ((C::TAILPACK && (POS) % 3 == 1) ||                                     \
                                  (C::TAILPACK3 && (POS) % 3 == 2 && (POS) < 3 * NX)) ? NT - 1 : NT;     \
            const int nxt_ = cur + 1 < NSLOT ? cur + 1 : 0, nx2_ = nxt_ + 1 < NSLOT ? nxt_ + 1 : 0;      \
            GGNN_T(POS, 0)                                                                               \
            if constexpr (GATHER && (POS) == G_U % NSTAGE) {   /* the fragment this stage multiplies */  \
                if (active && (!G_NEXT || p > 0)) { g_finish(xf[GBUF]); if constexpr (SAVEX) { if (row < a.V && a.save_x) store_x(xf[GBUF], row); } } \
            }                                                                                            \
            if constexpr ((POS) % 3 == 0 || (POS) == NSTAGE - 1) { if (active && !(a.dbg & 16)) split_frag<D, FMT>(sf, FRAG); } \
            /* ---- sub-stage A: tiles [0, TA) from slot cur */                                          \
            bool dma_a_ = false;                                                                         \
            if constexpr (DEEP) {                                                                        \
                if (late) {                                                                              \
                    prefetch(std::integral_constant<int, (POS)>{}, 1);                                   \
                    if (more_ && !(a.dbg & 8)) { dma_ha(nsrc_, ring + nx2_ * SLOT); dma_a_ = true; }     \
                    prefetch(std::integral_constant<int, (POS)>{}, 2);                                   \
                }                                                                                        \
            } else {                                                                                     \
                if (!(a.dbg & 8)) dma_hb(csrc_ + SC::HA, ring + nxt_ * SLOT);                            \
                prefetch(std::integral_constant<int, (POS)>{});                                          \
            }                                                                                            \
            GGNN_T(POS, 1)                                                                               \
            __builtin_amdgcn_sched_barrier(0);                                                           \
            if (a.dbg & 32) __builtin_amdgcn_s_setprio(3);                                               \
            if (active && !(a.dbg & 1))                                                                  \
                stage_mma_split_at<D, (ntl_ < SC::TA ? ntl_ : SC::TA), ((POS) < 3), 0, false, FMT>(ACC, sf, FRAG, ring + cur * SLOT, ring + cur * SLOT, li, kq); \
            if (a.dbg & 32) __builtin_amdgcn_s_setprio(0);                                               \
            __builtin_amdgcn_sched_barrier(0);                                                           \
            GGNN_T(POS, 2)                                                                               \
            if constexpr (DEEP) {                                                                        \
                if (!late) {                                                                             \
                    prefetch(std::integral_constant<int, (POS)>{}, 1);                                   \
                    if (more_ && !(a.dbg & 8)) { dma_ha(nsrc_, ring + nx2_ * SLOT); dma_a_ = true; }     \
                    prefetch(std::integral_constant<int, (POS)>{}, 2);                                   \
                }                                                                                        \
                publish_deep(dma_a_ ? own_a : 0);                                                        \
            } else publish();                                                                            \
            GGNN_T(POS, 3)                                                                               \
            cur = nxt_;                                                                                  \
            /* ---- sub-stage B: tiles [TA, ntl) from the slot that has just been published */           \
            {                                                                                            \
                const int nb1_ = cur + 1 < NSLOT ? cur + 1 : 0, nb2_ = nb1_ + 1 < NSLOT ? nb1_ + 1 : 0;  \
                bool dma_b_ = false;                                                                     \
                if constexpr (DEEP) {                                                                    \
                    if (late && more_ && !(a.dbg & 8)) { dma_hb(nsrc_ + SC::HA, ring + nb2_ * SLOT); dma_b_ = true; } \
                } else {                                                                                 \
                    if (more_ && !(a.dbg & 8)) dma_ha(nsrc_, ring + nb1_ * SLOT);                        \
                }                                                                                        \
                __builtin_amdgcn_sched_barrier(0);                                                       \
                GGNN_T2(POS, 0)                                                                          \
                if (a.dbg & 32) __builtin_amdgcn_s_setprio(3);                                           \
                if (active && !(a.dbg & 1))                                                              \
                    stage_mma_split_at<D, ntl_, ((POS) < 3), SC::TA, false, FMT>(ACC, sf, FRAG, ring + cur * SLOT, ring + cur * SLOT, li, kq); \
                if (a.dbg & 32) __builtin_amdgcn_s_setprio(0);                                           \
                __builtin_amdgcn_sched_barrier(0);                                                       \
                GGNN_T2(POS, 1)                                                                          \
                if constexpr (DEEP) {                                                                    \
                    if (!late && more_ && !(a.dbg & 8)) { dma_hb(nsrc_ + SC::HA, ring + nb2_ * SLOT); dma_b_ = true; } \
                    GGNN_T2(POS, 2)                                                                      \
                    publish_deep(dma_b_ ? own_b : 0);                                                    \
                } else publish();                                                                        \
                GGNN_T2(POS, 3)                                                                          \
                cur = nb1_;                                                                              \
            }                                                                                            \
        }
#define GGNN_STAGE(POS, ACC, FRAG)                                                                       \
        if constexpr (HALF) GGNN_HALF_STAGE(POS, ACC, FRAG) else                                         \
        if constexpr (coop && COOP_REGS) GGNN_COOP_STAGE(POS, ACC, FRAG) else                            \
        {                                                                                                \
            constexpr int npos_ = (POS) + 1;                                                             \
            const bool more_ = (npos_ < NSTAGE) || tk_next < n_dma;                                      \
            const float* nsrc_ = packed + (size_t)gru_stage_image<NX>(npos_ < NSTAGE ? npos_ : 0) * I::IMG; \
            float* ndst_ = ring + (cur ^ 1) * I::IMG;                                                    \
            GGNN_T(POS, 0)                                                                               \
            if constexpr (GATHER && (POS) == G_U % NSTAGE) {   /* the fragment this stage multiplies */  \
                if (active && (!G_NEXT || p > 0)) { g_finish(xf[GBUF]); if constexpr (SAVEX) { if (row < a.V && a.save_x) store_x(xf[GBUF], row); } } \
            }                                                                                            \
            /* Side work of the stage (prefetches, the DMA of the whole next image): the LATE waves do it  \
               before their MFMA burst, the EARLY waves after theirs, so its memory instructions issue      \
               while the partner wave owns the matrix pipe (its few VALU instructions do not overlap --    \
               see the file header; worth 0.4 %).  (Spreading the DMA instructions over the MFMA groups    \
               via the stage_mma hook measured slower.) */                                                \
            if (late) {                                                                                  \
                prefetch(std::integral_constant<int, (POS)>{});                                          \
                if (more_ && !(a.dbg & 8)) dma(nsrc_, ndst_);                             \
            }                                                                                            \
            GGNN_T(POS, 1)                                                                               \
            __builtin_amdgcn_sched_barrier(0);   /* keep the side work on its side of the MFMA block */  \
            /* u-gate stages (POS % 3 == 1) skip their last tile when it rides in the r image */          \
            /* ... and so do the candidate stages of the x segments (POS % 3 == 2 below the h stages)    */ \
            constexpr int ntl_ = ((C::TAILPACK && (POS) % 3 == 1) ||                                     \
                                  (C::TAILPACK3 && (POS) % 3 == 2 && (POS) < 3 * NX)) ? NT - 1 : NT;     \
            /* stages 0..2 open the three accumulator sets: they start from the constant 0 */            \
            /* SPLIT: a fragment is split into its bf16 planes before the first of its stages */         \
            if constexpr (SPLIT && ((POS) % 3 == 0 || (POS) == NSTAGE - 1)) { if (active && !(a.dbg & 16)) split_frag<D, FMT>(sf, FRAG); } \
            if (a.dbg & 32) __builtin_amdgcn_s_setprio(3);                                               \
            if (active && !(a.dbg & 1)) {                                                                \
                const float* img_ = ring + cur * I::IMG;                                                 \
                if constexpr (!coop) {                                                                   \
                    if constexpr (SPLIT) stage_mma_split<D, ntl_, ((POS) < 3), false, FMT>(ACC, sf, FRAG, img_, li, kq); \
                    else stage_mma<D, NoHook, ntl_, ((POS) < 3)>(ACC, FRAG, img_, li, kq);               \
                } else if (C::TAILPACK3 && (POS) == NSTAGE - 1 && wave == NT - 1) {                      \
                    f32x4 t_;        /* (same association as the tail-packed ordinary passes, see GGNN_COOP_STAGE) */ \
                    if constexpr (SPLIT) stage_mma_one_split<D, true, FMT>(t_, sf, FRAG, img_, li, kq, wave); \
                    else stage_mma_one<D, true>(t_, FRAG, img_, li, kq, wave);                           \
                    ACC[0] = ACC[0] + t_;                                                                \
                } else if (wave < NT) {                                                                  \
                    if constexpr (SPLIT) stage_mma_one_split<D, ((POS) < 3), FMT>(ACC[0], sf, FRAG, img_, li, kq, wave); \
                    else stage_mma_one<D, ((POS) < 3)>(ACC[0], FRAG, img_, li, kq, wave);                \
                }                                                                                        \
            }                                                                                            \
            if (a.dbg & 32) __builtin_amdgcn_s_setprio(0);                                               \
            __builtin_amdgcn_sched_barrier(0);                                                           \
            if (!late) {                                                                                 \
                prefetch(std::integral_constant<int, (POS)>{});                                          \
                if (more_ && !(a.dbg & 8)) dma(nsrc_, ndst_);                             \
            }                                                                                            \
            GGNN_T(POS, 2)                                                                               \
            publish();                                                                                   \
            GGNN_T(POS, 3)                                                                               \
            cur ^= 1;                                                                                    \
        }

        f32x4 acc_r[NT], acc_u[NT], acc_c[NT];       // opened by stages 0, 1, 2 (first MFMA of each tile: C = 0)
        SFrag<D> sf;                                 // (SPLIT) bf16 planes of the fragment the current stages multiply
        constexpr int TWD = GGNN_COOP_DEPTH;         // (cooperative tail pass, COOP_REGS) stages of weight look-ahead
        TileWeights<D> tw[TWD + 1];                  // this wave's column-tile weights of stages POS .. POS+TWD
        f32x4 hv_pre = {0.f, 0.f, 0.f, 0.f};
        if constexpr (coop && COOP_REGS) {
            if (wave < NT) {
#pragma unroll
                for (int i = 0; i < TWD; ++i)
                    load_tile_weights<D>(tw[i], packed + (size_t)gru_stage_image<NX>(i) * C::IMG, li, kq, wave);
            }
        }
        if constexpr (C::TAILPACK) acc_u[NT - 1] = f32x4{0.f, 0.f, 0.f, 0.f};   // (never computed: it rides in acc_r)
        if constexpr (C::TAILPACK3) acc_c[NT - 1] = f32x4{0.f, 0.f, 0.f, 0.f};  // (opened by the r*h stage only)
        // ---- x segments: r, u and candidate columns of each -------------------------------------------
        GGNN_STAGE(0, acc_r, xf[0]) GGNN_STAGE(1, acc_u, xf[0]) GGNN_STAGE(2, acc_c, xf[0])
        if constexpr (NX >= 2) { GGNN_STAGE(3, acc_r, xf[1]) GGNN_STAGE(4, acc_u, xf[1]) GGNN_STAGE(5, acc_c, xf[1]) }
        if constexpr (NX >= 3) { GGNN_STAGE(6, acc_r, xf[0]) GGNN_STAGE(7, acc_u, xf[0]) GGNN_STAGE(8, acc_c, xf[0]) }
        // ---- h: r and u columns -----------------------------------------------------------------------------
        GGNN_STAGE(3 * NX, acc_r, hf)
        GGNN_STAGE(3 * NX + 1, acc_u, hf)

        // ---- r = sigmoid(.), u = sigmoid(.), rh = r*h in activation-fragment layout -------------------
        Frag<D> rh;
        if constexpr (coop) {
            // cooperative tail pass: this wave holds column tile `wave` of r and u (accumulators [0]).  Its r*h tile goes
            // to LDS, every wave then reads the whole r*h fragment back; u and the h columns stay in acc_u[0] / acc_r[0].
            const int col = wave * 16 + 4 * kq;
            if (wave < NT && col < D && !(a.dbg & 2)) {
                const f32x4 r = sigmoid4_acc<FMT>(acc_r[0], ld4(bias_s + col));
                const f32x4 u = sigmoid4_acc<FMT>(acc_u[0], ld4(bias_s + D + col));
                f32x4 hv;
                if constexpr (COOP_REGS) hv = hv_pre;
                else hv = ld4_b(a.h, ((unsigned)rowc * D + col) * 4u);
                if constexpr (SAVE) {
                    if (row < a.V && a.save_r) {
                        st4_b(a.save_r, ((unsigned)row * D + col) * 4u, r);
                        st4_b(a.save_u, ((unsigned)row * D + col) * 4u, u);
                    }
                }
                acc_u[0] = u; acc_r[0] = hv;
                *reinterpret_cast<f32x4*>(rh_x + li * RHP + col) = r * hv;
            }
            __syncthreads();
#pragma unroll
            for (int c = 0; c < NC; ++c) rh.v[c] = *reinterpret_cast<const f32x4*>(rh_x + li * RHP + 16 * c + 4 * kq);
#pragma unroll
            for (int q = 0; q < NR; ++q) rh.r[q] = rh_x[li * RHP + 16 * NC + 4 * q + kq];
        } else if (active && !(a.dbg & 2)) {                   // (constexpr-else of the cooperative branch)
            stage_tail_reduce<D>(acc_r);                       // VALU-tail columns: add the four kq partials
            stage_tail_reduce<D>(acc_u);
            if constexpr (C::TAILPACK) {
                // the u gate's last tile was accumulated in the padding columns of the r gate's last tile:
                // column D + j of that tile (lane kq + (D%16)/4) is u column 16*NC + j (lane kq)
                constexpr int SH = 16 * ((D % 16) / 4);
                f32x4 ut;
                ut.x = __shfl(acc_r[NT - 1].x, lane + SH); ut.y = __shfl(acc_r[NT - 1].y, lane + SH);
                ut.z = __shfl(acc_r[NT - 1].z, lane + SH); ut.w = __shfl(acc_r[NT - 1].w, lane + SH);
                acc_u[NT - 1] = ut;
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int col = nt * 16 + 4 * kq;
                if (col < D) {
                    // bias_s holds -log2(e)*bg: sigmoid(z + b) = 1 / (1 + 2^(-log2e z - log2e b)), the bias add folded
                    // into the scaling FMA; written on whole float4s so that the non-transcendental half packs (v_pk_*)
                    const f32x4 r = sigmoid4_acc<FMT>(acc_r[nt], ld4(bias_s + col));
                    const f32x4 u = sigmoid4_acc<FMT>(acc_u[nt], ld4(bias_s + D + col));
                    acc_r[nt] = r; acc_u[nt] = u;
                    if constexpr (SAVE) {
                        if (row < a.V && a.save_r) {
                            st4_b(a.save_r, ((unsigned)row * D + col) * 4u, r);
                            st4_b(a.save_u, ((unsigned)row * D + col) * 4u, u);
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

        // ---- candidate: the r*h rows of Wc ----------------------------------------------------------------
        if (a.tickets && tid == 0) tk_slot[(p + 1) & 1] = nb + tk_fetch;
        GGNN_STAGE(3 * NX + 2, acc_c, rh)
#undef GGNN_STAGE
#undef GGNN_HALF_STAGE
#undef GGNN_COOP_STAGE
        tk = tk_next;
        tk_next = a.tickets ? __builtin_amdgcn_readfirstlane(tk_slot[(p + 1) & 1]) : tk_next + nb;

        // ---- c = act(.), h' = u*h + (1-u)*c ----------------------------------------------------------------
        // remainder tile: lane (li,kq) needs h cols 16NC + 4kq + e = remainder frag q = kq of lane (li, e);
        // the cross-lane reads stay outside lane-divergent control flow
        if constexpr (coop) {
            const int col = wave * 16 + 4 * kq;
            if (wave < NT && col < D && row < a.V && !(a.dbg & 4)) {
                f32x4 c;
                if (a.act == GGNN_ACT_TANH) {
                    c = tanh4_acc<FMT>(acc_c[0], ld4(bias_s + 2 * D + col));
                } else {
                    c = acc_c[0] * SplitFmt<FMT>::acc_scale + ld4(bias_s + 3 * D + col);
                    c.x = fmaxf(c.x, 0.f); c.y = fmaxf(c.y, 0.f); c.z = fmaxf(c.z, 0.f); c.w = fmaxf(c.w, 0.f);
                }
                const f32x4 u = acc_u[0], hv = acc_r[0];
                st4_b(a.h_out, ((unsigned)row * D + col) * 4u, u * hv + (1.0f - u) * c);
                if constexpr (SAVE) { if (a.save_c) st4_b(a.save_c, ((unsigned)row * D + col) * 4u, c); }
            }
            return;
        }
        f32x4 hrem = {0.f, 0.f, 0.f, 0.f};
        if (active) {
            stage_tail_reduce<D>(acc_c);
            if constexpr (C::TAILPACK3) {
                // the x segments' share of the candidate's last tile was accumulated two lane groups up in the r
                // gate's last tile (which the gates epilogue rewrote only in its own lanes); add it to the r*h share
                constexpr int SH2 = 32 * ((D % 16) / 4);
                f32x4 ct;
                ct.x = __shfl(acc_r[NT - 1].x, lane + SH2); ct.y = __shfl(acc_r[NT - 1].y, lane + SH2);
                ct.z = __shfl(acc_r[NT - 1].z, lane + SH2); ct.w = __shfl(acc_r[NT - 1].w, lane + SH2);
                acc_c[NT - 1] = ct + acc_c[NT - 1];
            }
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
                    f32x4 c;
                    if (a.act == GGNN_ACT_TANH) {
                        c = tanh4_acc<FMT>(acc_c[nt], ld4(bias_s + 2 * D + col));           // (2 log2e * bc)
                    } else {
                        c = acc_c[nt] * SplitFmt<FMT>::acc_scale + ld4(bias_s + 3 * D + col);   // (bc itself)
                        c.x = fmaxf(c.x, 0.f); c.y = fmaxf(c.y, 0.f); c.z = fmaxf(c.z, 0.f); c.w = fmaxf(c.w, 0.f);
                    }
                    const f32x4 u = acc_u[nt];
                    st4_b(a.h_out, ((unsigned)row * D + col) * 4u, u * hv + (1.0f - u) * c);
                    if constexpr (SAVE) { if (a.save_c) st4_b(a.save_c, ((unsigned)row * D + col) * 4u, c); }
                }
            }
        }
    };
    // tickets come to a workgroup in increasing order: its ordinary passes first, then (at most a few) tail passes
    int p = 0;
    for (; tk < n_main; ++p) run_pass(std::false_type{}, p);
    if constexpr (!HALF) { for (; tk < n_tk; ++p) run_pass(std::true_type{}, p); }   // (cooperative tail passes: form 0 only)
    if (GGNN_TDBG(a) && tid == 0) {
        a.tdbg[4096 + blockIdx.x * 4 + 2] = __builtin_amdgcn_s_memtime();
        a.tdbg[4096 + blockIdx.x * 4 + 3] = __builtin_amdgcn_s_memrealtime();
    }
}

template <int D, int NX, int NW, bool SAVE, bool GATHER, bool SPLIT, bool SAVEX = SAVE, int FORM = 0, int FMT = kSplitBf16x3>
static int launch_gru_fused_m(const GruFusedArgs& a_in, float* packed, hipStream_t st) {
    using C = StageCfg<D>;
    using I = ImgCfg<D, SPLIT, FMT>;
    GruFusedArgs a = a_in;
    { static int dbg = -1; if (dbg < 0) { const char* e = getenv("GGNN_GRU_DBG"); dbg = e ? atoi(e) : 0; } a.dbg = dbg; }
    { const char* e = getenv("GGNN_GRU_TPTR"); a.tdbg = e ? (unsigned long long*)strtoull(e, nullptr, 10) : nullptr; }
    if (a.Wg) {   // raw weights given: build the stage images first (skipped when the caller pre-packed them)
        hipLaunchKernelGGL((gru_pack_weights_kernel<D, SPLIT, FMT>), dim3(8, 3 * (NX + 1)), dim3(256), 0, st, a.Wg, a.Wc, NX, packed);
        GGNN_CHECK_HIP(hipGetLastError());
    }
    if (a.h == nullptr) return GGNN_OK;   // pack-only call
    if ((unsigned long long)a.V * D >= (1ULL << 30) || (a.g_H && (unsigned long long)a.V * a.g_T * D >= (1ULL << 30)))
        return fail(GGNN_E_UNSUPPORTED, "fused GRU indexes with 32-bit byte offsets: V*D (and V*T*D for the gathered rows) "
                                        "must be < 2^30 (V=%d, D=%d)", a.V, D);
    constexpr size_t bias_b = (size_t)((4 * D + 4 + 63) / 64 * 64) * sizeof(float);                       // biases, ticket slots
    const size_t lds = FORM == 1 ? (size_t)2 * SplitCfg<D, FMT>::HA_BYTES + bias_b                              // two workgroups per CU
                     : FORM == 2 ? (size_t)3 * SplitCfg<D, FMT>::HA_BYTES + bias_b
                                 : (size_t)2 * I::IMG_BYTES + bias_b + (size_t)16 * (C::BN + 4) * sizeof(float);  // + r*h exchange block
    const int wt_total = (a.V + 15) / 16;
    // one workgroup per CU; with fewer than NW tiles per CU the tiles are spread over ALL CUs as thin tickets (the
    // kernel's tail rule: ceil(tiles / nb) waves busy per workgroup) rather than packed 8 to a workgroup on a few CUs --
    // a pass with one or two busy waves takes less than half the time of a full one
    int nb = FORM == 1 ? 2 * num_cus() : num_cus();
    if (nb > wt_total) nb = wt_total;
    // Few tiles (the dense model's b = 256 x v = 29: 464): one tile per workgroup, worked cooperatively by its 8 waves (25 MFMAs
    // per stage and wave, weights straight from L2), the workgroups beyond the CU count following as the first ones retire --
    // instead of two-tile tickets on which two waves of a workgroup run the full 175-MFMA stages while six idle.
    static const int coop_small = [] { const char* e = getenv("GGNN_GRU_COOP_SMALL"); return e ? atoi(e) : 1; }();
    if (FORM == 0 && coop_small && wt_total > nb && wt_total <= 2 * nb && StageCfg<D>::NT <= NW) nb = wt_total;
    static std::atomic<unsigned long long> lds_ok{0};        // (one per template instantiation)
    if (lds > 64 * 1024) GGNN_CHECK_HIP(allow_dynamic_lds(&ggnn_gru_fused_kernel<D, NX, NW, SAVE, GATHER, SPLIT, SAVEX, FORM, FMT>, lds, lds_ok));
    hipLaunchKernelGGL((ggnn_gru_fused_kernel<D, NX, NW, SAVE, GATHER, SPLIT, SAVEX, FORM, FMT>), dim3(nb), dim3(NW * 64), lds, st, a, (const float*)packed);
    GGNN_CHECK_HIP(hipGetLastError());
    return GGNN_OK;
}

#ifdef GGNN_GRU_TU_SPLIT
// SAVE is a run-time matter in the kernel's epilogues (uniform branches on the save pointers): the training instantiation
// serves inference too (it is also the one that comes out of the register allocator with less scratch).
// GGNN_GRU_FORM: ring form of the gather-fused launches (the kernel's FORM): 0 whole images / 8 waves (the form before round 4),
// 1 two 4-wave workgroups per CU on half-image rings, 2 8 waves on a 3-slot half-image ring with partial waits
// Default (-1): form 1, except the single-input launch in the exact bf16x3 format.  Round 5, two-piece f16 format
// (profiles/r05_experiments/gru_forms_f16x2.txt): without a tail round every form takes 63-64 us at R = 0; the reference's
// 100,000-node batches leave 106 of 6250 tiles for a fourth round, which costs 7.6 us as form 0's cooperative pass and 3.8 us as
// form 1's single-wave tickets on 512 independent workgroups (R = 0 launch 71.2 -> 66.9 us); with residual inputs form 1 is also
// the one the register allocator leaves without scratch (88 / 109 us against 99 / 121).  In the exact bf16x3 format form 0 was 3 %
// ahead of form 1 at R = 0 (round 4).
std::atomic<int>& gru_form_override();     // ggnn_api.hip: GGNN_GRU_FORM / ggnn_gru_form_set()
int gru_wide_launch(int D, int nx, int ntw_req, const GruFusedArgs& a, float* packed, hipStream_t st);   // ggnn_gru_wide.hip
int gru_wide_supported(int D, int nx, const GruFusedArgs& a);
static int gru_form(int nx, int fmt) {
    const int v = gru_form_override().load(std::memory_order_relaxed);
    return v >= 0 ? v : ((nx >= 2 || fmt == kSplitF16x2) ? 1 : 0);
}

template <int D, int FMT>
static int split_launch_d(int nx, bool gather, const GruFusedArgs& a, float* packed, hipStream_t st) {
    if constexpr (SplitCfg<D>::OK) {
        // form 6: the wide kernel (one wave per SIMD, several tiles per wave; ggnn_gru_wide.hip)
        // (60 + NTW: the same with NTW tiles per wave where the instantiation exists -- experiments)
        if (const int f = gru_form(nx, FMT); gather && (f == 6 || (f >= 60 && f < 70)) && gru_wide_supported(D, nx, a))
            return gru_wide_launch(D, nx, f >= 60 ? f - 60 : 0, a, packed, st);
        if (gather && gru_form(nx, FMT) == 2) {
            switch (nx) {
                case 1: return launch_gru_fused_m<D, 1, 8, true, true, true, true, 2, FMT>(a, packed, st);
                case 2: return a.save_x ? launch_gru_fused_m<D, 2, 8, true, true, true, true, 2, FMT>(a, packed, st)
                                        : launch_gru_fused_m<D, 2, 8, true, true, true, false, 2, FMT>(a, packed, st);
                case 3: return a.save_x ? launch_gru_fused_m<D, 3, 8, true, true, true, true, 2, FMT>(a, packed, st)
                                        : launch_gru_fused_m<D, 3, 8, true, true, true, false, 2, FMT>(a, packed, st);
            }
        }
        if (gather && gru_form(nx, FMT) == 1) {
            switch (nx) {
                case 1: return launch_gru_fused_m<D, 1, 4, true, true, true, true, 1, FMT>(a, packed, st);
                case 2: return a.save_x ? launch_gru_fused_m<D, 2, 4, true, true, true, true, 1, FMT>(a, packed, st)
                                        : launch_gru_fused_m<D, 2, 4, true, true, true, false, 1, FMT>(a, packed, st);
                case 3: return a.save_x ? launch_gru_fused_m<D, 3, 4, true, true, true, true, 1, FMT>(a, packed, st)
                                        : launch_gru_fused_m<D, 3, 4, true, true, true, false, 1, FMT>(a, packed, st);
            }
        }
        if (gather) {
            switch (nx) {
                case 1: return launch_gru_fused_m<D, 1, 8, true, true, true, true, 0, FMT>(a, packed, st);        // (R = 0: no scratch either way)
                case 2: return a.save_x ? launch_gru_fused_m<D, 2, 8, true, true, true, true, 0, FMT>(a, packed, st)
                                        : launch_gru_fused_m<D, 2, 8, true, true, true, false, 0, FMT>(a, packed, st);
                case 3: return a.save_x ? launch_gru_fused_m<D, 3, 8, true, true, true, true, 0, FMT>(a, packed, st)
                                        : launch_gru_fused_m<D, 3, 8, true, true, true, false, 0, FMT>(a, packed, st);
            }
        } else {
            switch (nx) {
                case 1: return launch_gru_fused_m<D, 1, 8, true, false, true, true, 0, FMT>(a, packed, st);
                case 2: return launch_gru_fused_m<D, 2, 8, true, false, true, true, 0, FMT>(a, packed, st);
                case 3: return launch_gru_fused_m<D, 3, 8, true, false, true, true, 0, FMT>(a, packed, st);
            }
        }
    }
    return fail(GGNN_E_INVALID, "nx %d outside 1..3", nx);
}
int gru_split_launch(int D, int nx, bool save, bool gather, const GruFusedArgs& a, float* packed, hipStream_t st) {
    (void)save;
#ifdef GGNN_PROBE_NX   // register-allocation probe (tools/kernel_regs.sh .../ggnn_gru_fused_split.hip . -DGGNN_PROBE_NX=2): one instantiation
#ifndef GGNN_PROBE_SAVE
#define GGNN_PROBE_SAVE true
#endif
#ifndef GGNN_PROBE_SAVEX
#define GGNN_PROBE_SAVEX GGNN_PROBE_SAVE
#endif
#ifndef GGNN_PROBE_FMT
#define GGNN_PROBE_FMT kSplitF16x2
#endif
    return launch_gru_fused_m<100, GGNN_PROBE_NX, 8, GGNN_PROBE_SAVE, true, true, GGNN_PROBE_SAVEX, 0, GGNN_PROBE_FMT>(a, packed, st);
#else
    const bool f2 = gru_launch_fmt(a.fmt) == kSplitF16x2;            // (per launch: GruFusedArgs::fmt)
    switch (D) {
        case 100: return f2 ? split_launch_d<100, kSplitF16x2>(nx, gather, a, packed, st) : split_launch_d<100, kSplitBf16x3>(nx, gather, a, packed, st);
        case 64: return f2 ? split_launch_d<64, kSplitF16x2>(nx, gather, a, packed, st) : split_launch_d<64, kSplitBf16x3>(nx, gather, a, packed, st);
        case 32: return f2 ? split_launch_d<32, kSplitF16x2>(nx, gather, a, packed, st) : split_launch_d<32, kSplitBf16x3>(nx, gather, a, packed, st);
        default: return fail(GGNN_E_UNSUPPORTED, "no split-form fused GRU for hidden size %d", D);
    }
#endif
}
#else
template <int D, int NX, int NW, bool SAVE, bool GATHER>
static int launch_gru_fused(const GruFusedArgs& a, float* packed, hipStream_t st) {
    if (SplitCfg<D>::OK && split_matrix_path()) return gru_split_launch(D, NX, SAVE, GATHER, a, packed, st);
    return launch_gru_fused_m<D, NX, NW, SAVE, GATHER, false>(a, packed, st);
}

static bool nosave_kernel() {
    static const bool v = [] { const char* e = getenv("GGNN_GRU_NOSAVE_KERNEL"); return e && atoi(e) != 0; }();
    return v;
}

template <int D>
static int dispatch_nx(const GruFusedArgs& a, float* packed, hipStream_t st) {
    const bool save = a.save_r || a.save_u || a.save_c;
    if (save && !(a.save_r && a.save_u && a.save_c))
        return fail(GGNN_E_INVALID, "save_r / save_u / save_c must be given together");
    if (a.g_H) {
        if (save && !a.save_x) return fail(GGNN_E_INVALID, "the gather-fused GRU saves r/u/c together with the gathered segment (save_x)");
        switch (a.nx) {
            case 1: return save ? launch_gru_fused<D, 1, 8, true, true>(a, packed, st) : launch_gru_fused<D, 1, 8, false, true>(a, packed, st);
            // With residual inputs the inference kernels (SAVE = false) come out of the register allocator with 36-40 B of scratch per
            // lane, the training instantiations (SAVE = true: the same code plus stores) with none -- so those run both, the
            // stores skipped at run time when no save buffers are given (a uniform branch).  GGNN_GRU_NOSAVE_KERNEL=1: the others.
            case 2: return (save || !nosave_kernel()) ? launch_gru_fused<D, 2, 8, true, true>(a, packed, st) : launch_gru_fused<D, 2, 8, false, true>(a, packed, st);
            case 3: return (save || !nosave_kernel()) ? launch_gru_fused<D, 3, 8, true, true>(a, packed, st) : launch_gru_fused<D, 3, 8, false, true>(a, packed, st);
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

// 1: whole-block stage images, with a gather-fused variant; 2: column-panel kernel (no gather-fused variant); 0: none
int gru_fused_supported(int D) { return (D == 100 || D == 64 || D == 32) ? 1 : (gru_panel_supported(D) ? 2 : 0); }

int gru_fused_dispatch(const GruFusedArgs& a, int D, float* packed, hipStream_t st) {
    if (gru_panel_supported(D)) return gru_panel_dispatch(a, D, packed, st);
    switch (D) {
        case 100: return dispatch_nx<100>(a, packed, st);
        case 64: return dispatch_nx<64>(a, packed, st);
        case 32: return dispatch_nx<32>(a, packed, st);
        default: return fail(GGNN_E_UNSUPPORTED, "no fused GRU for hidden size %d", D);
    }
}
#endif   // !GGNN_GRU_TU_SPLIT

}  // namespace ggnn

#ifndef GGNN_GRU_TU_SPLIT
extern "C" int ggnn_gru_is_fused(int D) { return ggnn::gru_fused_supported(D); }
#endif
