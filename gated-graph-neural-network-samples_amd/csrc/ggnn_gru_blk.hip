// K3, form 4 ("block-resident"): the gather-fused GRU of the single-input layers (chem_tensorflow_sparse.py:198-216, R = 0:
// 6 of the 8 launches of the default model's forward) organised like the split-form dense kernel (ggnn_dense_graph_split.hip):
//
//   * a workgroup works on a BLOCK of 64 rows at a time; wave w < 7 owns output COLUMN tile w of every stage for all four row tiles
//     of the block (the 16 x D fragment-per-wave forms of ggnn_gru_fused.hip give every wave 84 accumulator registers and the whole
//     epilogue of its rows; here a wave holds 4 row tiles x {r, u, c} x 4 = 48 and the epilogue of 16 columns);
//   * the activations a stage multiplies -- the gathered messages x, the state h, r*h -- live in LDS ALREADY SPLIT into the bf16
//     operand planes of the MFMA ([plane][32-chunk][lane group][row][8 x bf16]); one ds_read_b128 = the operand of six MFMAs.  They are
//     split ONCE, by whoever produces them: x and h by the wave that fetched the fragment (a whole 16-row fragment: split_frag, nine
//     ds_write_b128), r*h by the wave that owns the columns (eight values: three ds_write_b64 per row tile);
//   * a wave's slice of a stage's weights -- its column tile of the split stage image, 9 x 16 bytes + the remainder float per lane --
//     comes from L2 straight into registers one stage ahead: NO weight ring, no LDS-DMA, no barrier per stage.  (The ring forms
//     re-stream 400 KB of images per 128 rows through LDS; here the same 400 KB go L2 -> registers per 64 rows.)
//   * the operands of block b + 1 are produced while block b is multiplied: the x fragments (the 3-level gather chain row_ptr ->
//     gather_row -> rows, sum, mean) by waves 6 and 7, whose SIMDs carry half the matrix work of the others (wave 6 owns the short
//     tail tile, wave 7 no tile at all), the h fragments by waves 2 and 3.  Three workgroup barriers per block.
// Same packed images, same products in the same order per accumulator, same epilogue expressions as the ring forms.
#include "ggnn_split.hpp"
#include <type_traits>

namespace ggnn {

namespace {

template <int D>
__device__ __forceinline__ void bz(Frag<D>& f) {
#pragma unroll
    for (int c = 0; c < StageCfg<D>::NC; ++c) f.v[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < StageCfg<D>::NR; ++q) f.r[q] = 0.f;
}
template <int D>
__device__ __forceinline__ void ba(Frag<D>& f, const Frag<D>& t) {
#pragma unroll
    for (int c = 0; c < StageCfg<D>::NC; ++c) f.v[c] += t.v[c];
#pragma unroll
    for (int q = 0; q < StageCfg<D>::NR; ++q) f.r[q] += t.r[q];
}

template <int D>
struct SliceW {                                                        // a wave's column tile of one split stage image
    u32x4 p[3][SplitCfg<D>::NC2 > 0 ? SplitCfg<D>::NC2 : 1];
    float r[StageCfg<D>::NR > 0 ? StageCfg<D>::NR : 1];
};

}  // namespace

template <int D, bool SAVE>
__global__ __launch_bounds__(512, 1) void ggnn_gru_blk_kernel(GruFusedArgs a, const float* __restrict__ packed) {
    using C = StageCfg<D>;
    using SC = SplitCfg<D>;
    constexpr int NW = 8, RB = 64, RT = RB / 16;
    constexpr int NT = C::NT, NC = C::NC, NR = C::NR, NC2 = SC::NC2;
    static_assert(NT <= NW - 1 && NR <= 1 && C::TAILPACK3, "D = 100 geometry: 7 column tiles, one remainder MFMA, tails riding in the r images");
    constexpr int SLAB = RB * 4;                                       // floats of one (plane, chunk, lane group) slab: RB rows x 16 bytes
    constexpr int OBF = 3 * NC2 * 4 * SLAB + RB * 4;                   // floats of an operand block (+ the remainder columns [RB][4])
    constexpr int BIAS_FLOATS = (4 * D + 63) / 64 * 64;
    extern __shared__ __attribute__((aligned(16))) float lds_[];      // [biases | X[2] | H[2]]   (r*h reuses the X block it came from)
    float* bias_s = lds_;                                              // [-log2e*bg (2D) | 2 log2e*bc (D) | bc (D)]
    float* Xb = lds_ + BIAS_FLOATS;
    float* Hb = Xb + 2 * OBF;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kq = lane >> 4;
    const int tile = wave < NT ? wave : 0;                             // (wave 7 owns no column tile)
    const bool tailw = wave == NT - 1;                                 // the tile of the D % 16 remainder columns and of the riding tails
    const int col0 = 16 * tile + 4 * kq;

    for (int i = tid; i < 4 * D; i += NW * 64)
        bias_s[i] = i < 2 * D ? -kLog2e * a.bg[i] : (i < 3 * D ? 2.0f * kLog2e * a.bc[i - 2 * D] : a.bc[i - 3 * D]);

    const int nblk = (a.V + RB - 1) / RB;
    const int nb = gridDim.x;

    // ---- operand production ---------------------------------------------------------------------------------------------------
    // a whole 16-row fragment (row tile t of an operand block): split, nine ds_write_b128 + the remainder column
    auto frag_to_planes = [&](float* blk, int t, const Frag<D>& f) {
        SFrag<D> s;
        split_frag<D>(s, f);
        u32x4* ob = reinterpret_cast<u32x4*>(blk) + kq * RB + t * 16 + li;
#pragma unroll
        for (int c2 = 0; c2 < NC2; ++c2) {
            ob[(0 * NC2 + c2) * 4 * RB] = s.hi[c2];
            ob[(1 * NC2 + c2) * 4 * RB] = s.mid[c2];
            ob[(2 * NC2 + c2) * 4 * RB] = s.lo[c2];
        }
        if constexpr (NR > 0) blk[3 * NC2 * 4 * SLAB + (t * 16 + li) * 4 + kq] = f.r[0];
    };
    auto store_x = [&](const Frag<D>& f, int row_) {
        const unsigned ob = ((unsigned)row_ * (unsigned)D + 4u * (unsigned)kq) * 4u;
#pragma unroll
        for (int c = 0; c < NC; ++c) st4_b(a.save_x, ob + 64u * c, f.v[c]);
#pragma unroll
        for (int q = 0; q < NR; ++q) *reinterpret_cast<float*>(reinterpret_cast<char*>(a.save_x) + ob - 16u * (unsigned)kq + (16u * NC + 4u * q + (unsigned)kq) * 4u) = f.r[q];
    };
    // Production of the NEXT block's operands, one 16-row fragment per wave, in PHASES that the block loop calls between its stages
    // (each phase consumes what the previous one fetched and issues the next level of the dependent chain, so the fetches travel
    // under the stage in between -- ggnn_gru_fused.hip's pipelined gather, here across one block instead of one pass):
    //   x fragments (the aggregated messages: row_ptr -> gather_row -> rows, summed in slot order, / (deg + 1e-7)) of row tile
    //   0..3 by waves 2, 3, 6, 7 -- the waves of the two SIMDs that carry the short tail tile and no tile at all;
    //   h fragments of row tile 0..3 by waves 0, 1, 4, 5 (seven loads in phase 0, split and written in phase 5).
    const int prod_x = (wave == 2 || wave == 3) ? wave - 2 : (wave >= 6 ? wave - 4 : -1);      // row tile of the x fragment, or -1
    const int prod_h = (wave == 0 || wave == 1) ? wave : ((wave == 4 || wave == 5) ? wave - 2 : -1);
    Frag<D> pf, pt;                                                   // the fragment being produced, the slot in flight
    int p_beg = 0, p_end = 0, p_gi[4] = {0, 0, 0, 0};
    float p_den = 1.f;
    auto phase = [&](auto px_c, int ph, int nblk_i, int buf) {         // nblk_i: the block whose operands are produced (>= nblk: none)
        constexpr bool PX = decltype(px_c)::value;                     // this wave produces an x fragment (else an h fragment)
        if (nblk_i >= nblk) return;
        const int t = PX ? prod_x : prod_h;
        const int row = nblk_i * RB + t * 16 + li;
        const int rc = row < a.V ? row : a.V - 1;
        const bool valid = row < a.V;
        if constexpr (!PX) {
            if (ph == 0) load_frag<D>(pf, a.h, rc, kq);
            if (ph == 5) frag_to_planes(Hb + buf * OBF, t, pf);
            return;
        }
        if (ph == 0) {
            const int b = ldi_b(a.g_row_ptr, (unsigned)rc * 4u), e = ldi_b(a.g_row_ptr, (unsigned)rc * 4u + 4u);
            p_beg = valid ? b : 0; p_end = valid ? e : 0;
            p_den = 1.f;
            if (a.g_use_avg) {
                if (a.g_T == 4) {
                    const f32x4 n = ld4_b(a.g_nin, (unsigned)rc * 16u);
                    p_den = (((n.x + n.y) + n.z) + n.w) + 1e-7f;
                } else {
                    float deg = 0.f;
                    for (int tt = 0; tt < a.g_T; ++tt) deg += a.g_nin[(size_t)rc * a.g_T + tt];
                    p_den = (((deg + 0.f) + 0.f) + 0.f) + 1e-7f;
                }
            }
        } else if (ph == 1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                p_gi[j] = 0;
                if (p_beg + j < p_end) p_gi[j] = ldi_b(a.g_idx, (unsigned)(p_beg + j) * 4u);
            }
        } else if (ph == 2) {
            load_frag<D>(pf, a.g_H, p_gi[0], kq);                     // (slots beyond the degree fetch row 0: unconditional loads)
            load_frag<D>(pt, a.g_H, p_gi[1], kq);
        } else if (ph == 3) {
            if (p_beg + 1 < p_end) ba<D>(pf, pt);
            load_frag<D>(pt, a.g_H, p_gi[2], kq);
        } else if (ph == 4) {
            if (p_beg + 2 < p_end) ba<D>(pf, pt);
            load_frag<D>(pt, a.g_H, p_gi[3], kq);
        } else {
            if (p_beg + 3 < p_end) ba<D>(pf, pt);
            for (int e = p_beg + 4; e < p_end; ++e) {
                load_frag<D>(pt, a.g_H, a.g_idx[e], kq);
                ba<D>(pf, pt);
            }
            if (p_beg >= p_end) bz<D>(pf);
            if (a.g_use_avg) {                                        // one division per row (Markstein: the correctly rounded quotient)
                const float den = p_den, r = 1.0f / den;
                auto dv = [&](float x) { const float q = x * r; return fmaf(fmaf(-den, q, x), r, q); };
#pragma unroll
                for (int c = 0; c < NC; ++c) pf.v[c] = f32x4{dv(pf.v[c].x), dv(pf.v[c].y), dv(pf.v[c].z), dv(pf.v[c].w)};
#pragma unroll
                for (int q = 0; q < NR; ++q) pf.r[q] = dv(pf.r[q]);
            }
            if constexpr (SAVE) { if (valid && a.save_x) store_x(pf, row); }
            frag_to_planes(Xb + buf * OBF, t, pf);
        }
    };

    // ---- this wave's weight slices: byte offsets inside a split image (two column halves, ggnn_split.hpp) ---------------------------
    const bool hb = tile >= SC::TA;
    const int nth = hb ? NT - SC::TA : SC::TA, til = hb ? tile - SC::TA : tile;
    const unsigned w_base = (hb ? (unsigned)SC::HA_BYTES : 0u) + (unsigned)(kq * nth * 16 + li + til * 16) * 16u;
    const unsigned w_cst = (unsigned)(4 * nth * 16) * 16u, w_pst = (unsigned)NC2 * w_cst;
    const unsigned wr_base = (hb ? (unsigned)SC::HA_BYTES : 0u) + 3u * w_pst + (unsigned)(kq * nth * 16 + li + til * 16) * 4u;
    auto load_w = [&](SliceW<D>& w, int image) {
        const unsigned long long p = reinterpret_cast<unsigned long long>(packed + (size_t)image * SC::IMG);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)p), hi = __builtin_amdgcn_readfirstlane((unsigned)(p >> 32));
        const float* sb = reinterpret_cast<const float*>(((unsigned long long)hi << 32) | lo);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int c2 = 0; c2 < NC2; ++c2)
                w.p[pl][c2] = __builtin_bit_cast(u32x4, ld4_b(sb, w_base + (unsigned)pl * w_pst + (unsigned)c2 * w_cst));
#pragma unroll
        for (int q = 0; q < NR; ++q) w.r[q] = ld1_b(sb, wr_base + (unsigned)(q * 4 * nth * 16) * 4u);
    };
    // all RT row tiles of an operand block against one weight slice.  Units (chunk, row tile): the operand planes of unit u + 1 are
    // read while the six MFMAs of unit u issue (24 operand registers in flight; a scheduling barrier per unit stops the compiler from
    // hoisting all 36 reads of a stage)
    auto mma_block = [&](auto zero_c, f32x4 (&acc)[RT], const float* blk, const SliceW<D>& w) {
        constexpr bool ZERO = decltype(zero_c)::value;
        constexpr int NU = NC2 * RT;
        const u32x4* ob = reinterpret_cast<const u32x4*>(blk) + kq * RB + li;
        auto opnd = [&](int u, int p) { return ob[(p * NC2 + u / RT) * 4 * RB + (u % RT) * 16]; };
        u32x4 ah = opnd(0, 0), am = opnd(0, 1), al = opnd(0, 2);
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int c2 = u / RT, t = u % RT;
            u32x4 nh = ah, nm = am, nl = al;
            if (u + 1 < NU) { nh = opnd(u + 1, 0); nm = opnd(u + 1, 1); nl = opnd(u + 1, 2); }
            __builtin_amdgcn_sched_barrier(0);
            const u32x4 wh = w.p[0][c2], wm = w.p[1][c2], wl = w.p[2][c2];
            f32x4 c = (ZERO && c2 == 0) ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[t];
            c = mfma_bf16(wl, ah, c);
            c = mfma_bf16(wm, am, c);
            c = mfma_bf16(wm, ah, c);
            c = mfma_bf16(wh, al, c);
            c = mfma_bf16(wh, am, c);
            c = mfma_bf16(wh, ah, c);
            acc[t] = c;
            __builtin_amdgcn_sched_barrier(0);
            ah = nh; am = nm; al = nl;
        }
        if constexpr (NR > 0) {
            const float* rb = blk + 3 * NC2 * 4 * SLAB;
#pragma unroll
            for (int t = 0; t < RT; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.r[0], rb[(t * 16 + li) * 4 + kq], acc[t], 0, 0, 0);
        }
    };
    // this wave's four columns of row tile t, split, into an operand block (r*h): half of lane (li, kq)'s 16-byte slot of chunk tile >> 1
    auto tile_to_planes = [&](float* blk, int t, f32x4 val) {
        const int row = t * 16 + li;
        if (tailw) {
            if (kq == 0) *reinterpret_cast<f32x4*>(blk + 3 * NC2 * 4 * SLAB + row * 4) = val;
            return;
        }
        unsigned h0, m0, l0, h1, m1, l1;
        split_pair(val.x, val.y, h0, m0, l0);
        split_pair(val.z, val.w, h1, m1, l1);
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        float* dst = blk + ((size_t)((tile >> 1) * 4 + kq)) * SLAB + row * 4 + 2 * (tile & 1);
        *reinterpret_cast<u32x2*>(dst) = u32x2{h0, h1};
        *reinterpret_cast<u32x2*>(dst + NC2 * 4 * SLAB) = u32x2{m0, m1};
        *reinterpret_cast<u32x2*>(dst + 2 * NC2 * 4 * SLAB) = u32x2{l0, l1};
    };

    // packed image of (segment, gate): ggnn_gru_fused.hip's order for one input segment: gates (x,r) (x,u) (h,r) (h,u), candidate (x) (r*h)
    constexpr int IMG_XR = 0, IMG_XU = 1, IMG_HR = 2, IMG_HU = 3, IMG_XC = 4, IMG_RC = 5;
    // The block loop exists once per ROLE (compile-time; chosen once per wave): full-tile waves that produce an h fragment (waves 0, 1,
    // 4, 5) or an x fragment (2, 3), the tail wave (6: the r images' last tile and the last stage; x producer), the wave without a
    // tile (7: x producer).  One loop with "if (has a tile)" around every stage is a chain of diamonds whose accumulator / weight
    // values all go through merge points: 400-530 B of scratch per lane in that form.
    //   stage sequence of a block, full-tile wave: XR HR XU HU XC | gates | RC  (weight slices alternate between two register sets,
    //   the next stage's requested before this stage's products); tail wave: XR HR | gates | RC (one set, fetched per stage: the wave
    //   has the time)
    auto run = [&](auto full_c, auto tail_c, auto px_c) {
        constexpr bool FULL = decltype(full_c)::value, TAIL = decltype(tail_c)::value;
        SliceW<D> w0, w1;
        int blk = blockIdx.x;
        for (int ph = 0; ph < 6; ++ph) phase(px_c, ph, blk, 0);        // the first block's operands, phases back to back
        if constexpr (FULL) load_w(w0, IMG_XR);
        __syncthreads();
        for (int it = 0; blk < nblk; ++it, blk += nb) {
            const int buf = it & 1;
            const float* X = Xb + buf * OBF;
            const float* H = Hb + buf * OBF;
            float* RH = Xb + buf * OBF;
            const int r0 = blk * RB;
            const int nxt = blk + nb, nbuf = buf ^ 1;
            f32x4 ar[RT], au[RT], ac[RT];
#define GGNN_BLK_SB() __builtin_amdgcn_sched_barrier(0)
            phase(px_c, 0, nxt, nbuf); GGNN_BLK_SB();
            if constexpr (FULL) { load_w(w1, IMG_HR); GGNN_BLK_SB(); mma_block(std::true_type{}, ar, X, w0); GGNN_BLK_SB(); }
            if constexpr (TAIL) { load_w(w0, IMG_XR); GGNN_BLK_SB(); mma_block(std::true_type{}, ar, X, w0); GGNN_BLK_SB(); }
            phase(px_c, 1, nxt, nbuf); GGNN_BLK_SB();
            if constexpr (FULL) { load_w(w0, IMG_XU); GGNN_BLK_SB(); mma_block(std::false_type{}, ar, H, w1); GGNN_BLK_SB(); }
            if constexpr (TAIL) { load_w(w0, IMG_HR); GGNN_BLK_SB(); mma_block(std::false_type{}, ar, H, w0); GGNN_BLK_SB(); }
            phase(px_c, 2, nxt, nbuf); GGNN_BLK_SB();
            if constexpr (FULL) { load_w(w1, IMG_HU); GGNN_BLK_SB(); mma_block(std::true_type{}, au, X, w0); GGNN_BLK_SB(); }
            phase(px_c, 3, nxt, nbuf); GGNN_BLK_SB();
            if constexpr (FULL) { load_w(w0, IMG_XC); GGNN_BLK_SB(); mma_block(std::false_type{}, au, H, w1); GGNN_BLK_SB(); }
            phase(px_c, 4, nxt, nbuf); GGNN_BLK_SB();
            if constexpr (FULL) { load_w(w1, IMG_RC); GGNN_BLK_SB(); mma_block(std::true_type{}, ac, X, w0); GGNN_BLK_SB(); }
            if constexpr (TAIL) { load_w(w0, IMG_RC); GGNN_BLK_SB(); }
            phase(px_c, 5, nxt, nbuf); GGNN_BLK_SB();
            // ---- gates (this wave's columns): r, u, r*h --------------------------------------------------------------------------
            f32x4 ht[RT], rh[RT], ctx[RT];
            if constexpr (FULL || TAIL) {
#pragma unroll
                for (int t = 0; t < RT; ++t) {
                    const int row = r0 + t * 16 + li;
                    ht[t] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (col0 < D) ht[t] = ld4_b(a.h, ((unsigned)(row < a.V ? row : a.V - 1) * D + col0) * 4u);
                }
            }
            if constexpr (FULL) {
#pragma unroll
                for (int t = 0; t < RT; ++t) {
                    const int row = r0 + t * 16 + li;
                    const f32x4 r = sigmoid4_scaled(ar[t], ld4(bias_s + col0));
                    const f32x4 u = sigmoid4_scaled(au[t], ld4(bias_s + D + col0));
                    if constexpr (SAVE) {
                        if (row < a.V && a.save_r) { st4_b(a.save_r, ((unsigned)row * D + col0) * 4u, r); st4_b(a.save_u, ((unsigned)row * D + col0) * 4u, u); }
                    }
                    au[t] = u;
                    rh[t] = r * ht[t];
                }
            }
            if constexpr (TAIL) {
                // lane group 0 holds the r columns 16 NC .. + 3, group 1 the u gate's, group 2 the candidate's x share (the riding tails)
                constexpr int SH = 16 * ((D % 16) / 4), SH2 = 32 * ((D % 16) / 4);
#pragma unroll
                for (int t = 0; t < RT; ++t) {
                    const int row = r0 + t * 16 + li;
                    f32x4 ut, ct;
                    ut.x = __shfl(ar[t].x, lane + SH); ut.y = __shfl(ar[t].y, lane + SH); ut.z = __shfl(ar[t].z, lane + SH); ut.w = __shfl(ar[t].w, lane + SH);
                    ct.x = __shfl(ar[t].x, lane + SH2); ct.y = __shfl(ar[t].y, lane + SH2); ct.z = __shfl(ar[t].z, lane + SH2); ct.w = __shfl(ar[t].w, lane + SH2);
                    ctx[t] = ct;
                    f32x4 r = {0.f, 0.f, 0.f, 0.f}, u = r;
                    if (kq == 0) {
                        r = sigmoid4_scaled(ar[t], ld4(bias_s + col0));
                        u = sigmoid4_scaled(ut, ld4(bias_s + D + col0));
                        if constexpr (SAVE) {
                            if (row < a.V && a.save_r) { st4_b(a.save_r, ((unsigned)row * D + col0) * 4u, r); st4_b(a.save_u, ((unsigned)row * D + col0) * 4u, u); }
                        }
                    }
                    au[t] = u;
                    rh[t] = r * ht[t];
                }
            }
            __syncthreads();                                           // (1) every wave has finished reading X: r*h may overwrite it
            if constexpr (FULL || TAIL) {
#pragma unroll
                for (int t = 0; t < RT; ++t) tile_to_planes(RH, t, rh[t]);
            }
            __syncthreads();                                           // (2) r*h complete
            if constexpr (FULL) { load_w(w0, IMG_XR); GGNN_BLK_SB(); mma_block(std::false_type{}, ac, RH, w1); GGNN_BLK_SB(); }
            if constexpr (TAIL) { mma_block(std::true_type{}, ac, RH, w0); GGNN_BLK_SB(); }
#undef GGNN_BLK_SB
            // ---- candidate, blend, store ---------------------------------------------------------------------------------------------
            if constexpr (FULL || TAIL) {
#pragma unroll
                for (int t = 0; t < RT; ++t) {
                    const int row = r0 + t * 16 + li;
                    f32x4 pre = ac[t];
                    if constexpr (TAIL) pre = ctx[t] + ac[t];
                    if (col0 < D && row < a.V) {
                        f32x4 c;
                        if (a.act == GGNN_ACT_TANH) {
                            c = tanh4_scaled(pre, ld4(bias_s + 2 * D + col0));
                        } else {
                            c = pre + ld4(bias_s + 3 * D + col0);
                            c.x = fmaxf(c.x, 0.f); c.y = fmaxf(c.y, 0.f); c.z = fmaxf(c.z, 0.f); c.w = fmaxf(c.w, 0.f);
                        }
                        const f32x4 u = au[t], hv = ht[t];
                        st4_b(a.h_out, ((unsigned)row * D + col0) * 4u, u * hv + (1.0f - u) * c);
                        if constexpr (SAVE) { if (a.save_c) st4_b(a.save_c, ((unsigned)row * D + col0) * 4u, c); }
                    }
                }
            }
            __syncthreads();                                           // (3) this block's operands are dead, the next block's complete
        }
    };
    using T_ = std::true_type; using F_ = std::false_type;
    if (wave == NT - 1) run(F_{}, T_{}, T_{});                         // wave 6
    else if (wave >= NT) run(F_{}, F_{}, T_{});                        // wave 7
    else if (prod_x >= 0) run(T_{}, F_{}, T_{});                       // waves 2, 3
    else run(T_{}, F_{}, F_{});                                        // waves 0, 1, 4, 5
}

template <int D, bool SAVE>
static int launch_blk(const GruFusedArgs& a_in, float* packed, hipStream_t st) {
    GruFusedArgs a = a_in;
    a.dbg = 0; a.tdbg = nullptr;
    if ((unsigned long long)a.V * D >= (1ULL << 30) || (unsigned long long)a.V * a.g_T * D >= (1ULL << 30))
        return fail(GGNN_E_UNSUPPORTED, "fused GRU indexes with 32-bit byte offsets: V*D (and V*T*D for the gathered rows) must be < 2^30");
    constexpr int RB = 64, NC2 = SplitCfg<D>::NC2;
    const size_t lds = ((size_t)((4 * D + 63) / 64 * 64) + (size_t)4 * (3 * NC2 * 4 * RB * 4 + RB * 4)) * sizeof(float);
    const int nblk = (a.V + RB - 1) / RB;
    int nb = num_cus();
    if (nb > nblk) nb = nblk;
    static std::atomic<unsigned long long> lds_ok{0};
    if (lds > 64 * 1024) GGNN_CHECK_HIP(allow_dynamic_lds(&ggnn_gru_blk_kernel<D, SAVE>, lds, lds_ok));
    hipLaunchKernelGGL((ggnn_gru_blk_kernel<D, SAVE>), dim3(nb), dim3(512), lds, st, a, (const float*)packed);
    GGNN_CHECK_HIP(hipGetLastError());
    return GGNN_OK;
}

// the single-input gather-fused launch (a.nx == 1, a.g_H set, packed weights in split form, D = 100) in form 4
int gru_blk_launch(const GruFusedArgs& a, float* packed, hipStream_t st) {
    return a.save_x ? launch_blk<100, true>(a, packed, st) : launch_blk<100, false>(a, packed, st);
}

}  // namespace ggnn
