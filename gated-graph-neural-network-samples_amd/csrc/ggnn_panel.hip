// Register-chained kernels for LARGE hidden sizes (D a multiple of 64: 128, 192, 256 -- BASELINE config "100k nodes / 1M
// edges, h = 256"): the fused GRU node update (chem_tensorflow_sparse.py:211-216) and the compacted per-edge-type message
// transform (:160-164) on COLUMN PANELS of the weight blocks.
//
// Why panels.  The D = 100 kernels (ggnn_gru_fused.hip, ggnn_msg_compact.hip) keep one D x D weight block as a 48 KiB
// k-interleaved image in LDS and a wave owns 16 rows x ALL D output columns of all three gates.  At D = 256 a block is
// 256 KiB (LDS: 160 KiB per CU) and three gates x 16 column tiles are 192 accumulator registers.  Here a stage multiplies
// the wave's 16 x D activation fragment by ONE 64-column panel of one weight block: the panel image is D x 64 floats
// (64 KiB at D = 256; two of them form the LDS-DMA ring) and a stage's accumulators are 4 tiles = 16 registers.
//
// GRU pass of a wave over its 16 rows (NS = NX + 1 input segments x_0 .. x_{NX-1}, h; NP = D / 64 panels):
//   phase R   for every segment s, every panel p:   acc_r[p] += seg_s x Wg[s rows, r columns of panel p]
//             r = sigmoid(acc_r + bg_r);  rh = r * h   -- in activation-fragment layout (output tile nt == k chunk nt, the
//             property the D = 100 kernel chains on), so r*h feeds the candidate stages from registers
//   phase UC  for every panel p:   u_p = sum_s seg_s x Wg[s, u columns of p],  c_p = sum_s x_s x Wc[s, p] + rh x Wc[h rows, p]
//             u = sigmoid(.), c = act(.), h'[:, panel p] = u h + (1 - u) c   -> stored; nothing but x, h is read and h' written
// Only ONE activation fragment is resident besides r*h: the fragment of the next stage's segment is (re)loaded from
// memory (L2 hits after the first touch) right after the wave's MFMA burst of the current stage, i.e. while the partner
// wave of the SIMD owns the matrix pipe.  Per wave: r*h 64 + fragment 64 + accumulators 32 + weight operands 32 VGPRs.
// Every product chain accumulates its segments in [x_0 | .. | h] order like the plain [x|h] Wg / [x|r*h] Wc products.
//
// The transform is the same stage loop without the chain: a persistent workgroup is bound to ONE (edge type, panel),
// keeps that panel image in LDS and walks 16-row tiles of the type's active (source node, type) pairs.
#include "ggnn_split.hpp"
// 1: panel_part_mma_split recomputes the lane part of its LDS / L2 addresses in every call (the lane coordinates pass through an empty asm).
// Without it hipcc keeps a per-(stage, part) address register set alive across the whole pass and the split-form GRU comes out with
// 152-492 B of scratch per lane at 256 registers; with it: no scratch, 208-228 registers; D = 256 launch 507.7 -> 497.5 us (round 4).
#ifndef GGNN_PANEL_REMAT
#define GGNN_PANEL_REMAT 1
#endif
#ifndef GGNN_PANEL_PLANES
#define GGNN_PANEL_PLANES 1        // the panel GRU keeps a fragment's planes resident over the stages that multiply it in a row
#endif
#ifndef GGNN_PANEL_DMA_FIRST
#define GGNN_PANEL_DMA_FIRST 1     // every wave of the panel GRU issues its DMA pieces before its burst (0: the early waves behind theirs,
                                   // the D = 100 kernels' rule): a part issued behind a burst lands late in the round -- 497 -> 485-487 us at D = 256
#endif
#include <type_traits>

namespace ggnn {

template <int D>
struct PanelCfg {
    static_assert(D % 64 == 0, "panel kernels need a hidden size that is a multiple of 64");
    static constexpr int PT = 4;                 // column tiles per panel
    static constexpr int BN = 64;                // columns per panel
    static constexpr int NP = D / BN;            // panels per D-column weight block
    static constexpr int NC = D / 16;            // k chunks == column tiles of a gate
    static constexpr int IMG = D * BN;           // floats per panel image
    static constexpr int IMG_BYTES = IMG * 4;
};

// image[c][kq][n][e] = W[r0 + 16c + 4kq + e][c0 + n]   (n < 64): the k-interleaved layout of ggnn_stage.hpp
template <int D>
__device__ __forceinline__ void pack_panel_image(const float* __restrict__ W, int r0, int c0, int ldw, float* __restrict__ img,
                                                 int first, int stride) {
    using C = PanelCfg<D>;
    for (int i = first; i < C::IMG; i += stride) {
        const int e = i & 3, n = (i >> 2) % C::BN, ck = (i >> 2) / C::BN;
        img[i] = W[(size_t)(r0 + 4 * ck + e) * ldw + c0 + n];
    }
}

// ---- the transform's panel in 3-way split form (ggnn_split.hpp): image = three bf16 planes [c2][g][n < 64][8 x bf16] -----------
template <int D>
struct PanelSplitCfg {
    static constexpr int NC2 = D / 32;
    static constexpr int PLANE_BYTES = NC2 * 4 * 64 * 16;          // D * 128
    static constexpr int IMG_BYTES = 3 * PLANE_BYTES;              // 96 KiB at D = 256: ONE image per workgroup, no ring
    static constexpr int IMG = IMG_BYTES / 4;
};

template <int D>
__device__ __forceinline__ void pack_panel_split_image(const float* __restrict__ W, int r0, int c0, int ldw, float* __restrict__ img,
                                                       int first, int stride) {
    using C = PanelSplitCfg<D>;
    constexpr int PW = C::PLANE_BYTES / 4;
    for (int i = first; i < C::IMG; i += stride) {
        const int plane = i / PW, w = i % PW;
        const int slot = w >> 2, pr = w & 3;
        const int n = slot % 64, cg = slot / 64, c2 = cg >> 2, g = cg & 3;
        const int j0 = 2 * pr;
        const int k0 = 32 * c2 + 16 * (j0 >> 2) + 4 * g + (j0 & 3);
        const float v0 = W[(size_t)(r0 + k0) * ldw + c0 + n], v1 = W[(size_t)(r0 + k0 + 1) * ldw + c0 + n];
        img[i] = __uint_as_float(split_piece_bits(v0, plane) | (split_piece_bits(v1, plane) << 16));
    }
}

// acc[0..3] (+)= fragment x split panel image.  The fragment stays f32; the 8 values of a 32-chunk are split right before the
// chunk's 24 MFMAs (the next chunk's while the current one multiplies); weight planes rotate through 12 registers (stage_mma_split).
template <int D, bool ZERO>
__device__ __forceinline__ void panel_mma_split(f32x4 (&acc)[4], const Frag<D>& a, const float* img, int li, int kq) {
    using C = PanelSplitCfg<D>;
    constexpr int NU = C::NC2 * 4;
    constexpr int PL = C::PLANE_BYTES / 16;
    const u32x4* base = reinterpret_cast<const u32x4*>(img) + kq * 64 + li;
    auto slot = [&](int u, int p) { return base[p * PL + (u / 4) * 4 * 64 + (u % 4) * 16]; };
    auto planes = [&](int c2, u32x4& hi, u32x4& mid, u32x4& lo) {
        const f32x4 x = a.v[2 * c2], y = a.v[2 * c2 + 1];
        unsigned h[4], m[4], l[4];
        split_pair(x.x, x.y, h[0], m[0], l[0]); split_pair(x.z, x.w, h[1], m[1], l[1]);
        split_pair(y.x, y.y, h[2], m[2], l[2]); split_pair(y.z, y.w, h[3], m[3], l[3]);
        hi = u32x4{h[0], h[1], h[2], h[3]}; mid = u32x4{m[0], m[1], m[2], m[3]}; lo = u32x4{l[0], l[1], l[2], l[3]};
    };
    u32x4 ah, am, al, nh, nm, nl;
    planes(0, ah, am, al);
    u32x4 wh = slot(0, 0), wm = slot(0, 1), wl = slot(0, 2);
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int c2 = u / 4, j = u % 4;
        const bool more = u + 1 < NU;
        if (j == 0 && c2 + 1 < C::NC2) planes(c2 + 1, nh, nm, nl);      // (vector work of the next chunk, under this chunk's MFMAs)
        f32x4 c = (ZERO && c2 == 0) ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[j];
        c = mfma_bf16(wl, ah, c);
        __builtin_amdgcn_sched_barrier(0);
        if (more) wl = slot(u + 1, 2);
        c = mfma_bf16(wm, am, c);
        c = mfma_bf16(wm, ah, c);
        __builtin_amdgcn_sched_barrier(0);
        if (more) wm = slot(u + 1, 1);
        c = mfma_bf16(wh, al, c);
        c = mfma_bf16(wh, am, c);
        c = mfma_bf16(wh, ah, c);
        __builtin_amdgcn_sched_barrier(0);
        if (more) wh = slot(u + 1, 0);
        acc[j] = c;
        if (j == 3) { ah = nh; am = nm; al = nl; }
    }
}

// ---- the GRU's panels in split form: image [c2][plane][g][n < 64][8 x bf16], CHUNK-major, so that it can be brought in in PARTS --
// A split 64-column panel of a 256-row block is 96 KiB: two of them do not fit the LDS.  The ring therefore holds PARTS of an
// image (D = 256: 2 x 4 chunks = 48 KiB each; 192: 3 x 2 chunks = 24 KiB; 128: the whole 48 KiB image) and a stage is PARTS
// DMA / MFMA / barrier rounds over the same accumulators.
// FMT (ggnn_split.hpp): kSplitBf16x3 = three planes per chunk (12 KiB); kSplitF16x2 (the panel GRU since the end of round 4) = two
// (8 KiB): a D = 256 image is 64 KiB and two WHOLE images fit the ring -- no parts at any width.
template <int D, int FMT = kSplitBf16x3>
struct PanelGruSplitCfg {
    static constexpr int NC2 = D / 32;
    static constexpr int NPL = SplitFmt<FMT>::NP;                  // planes per chunk
    static constexpr int CHUNK_BYTES = NPL * 4 * 64 * 16;          // 12 KiB per 32-chunk (two planes: 8)
    static constexpr int PARTS = FMT == kSplitF16x2 ? 1 : (D == 256 ? 2 : (D == 192 ? 3 : 1));
    static constexpr int CP = NC2 / PARTS;                         // chunks per part
    static constexpr int PART_BYTES = CP * CHUNK_BYTES;
    static constexpr int PART = PART_BYTES / 4;
    static constexpr int IMG_BYTES = NC2 * CHUNK_BYTES;
    static constexpr int IMG = IMG_BYTES / 4;
    static_assert(NC2 % PARTS == 0 && PART_BYTES % 8192 == 0, "parts are whole chunks and whole KiB per wave of an 8-wave workgroup");
};

template <int D, int FMT = kSplitBf16x3>
__device__ __forceinline__ void pack_panel_gru_split_image(const float* __restrict__ W, int r0, int c0, int ldw, float* __restrict__ img,
                                                           int first, int stride) {
    using C = PanelGruSplitCfg<D, FMT>;
    for (int i = first; i < C::IMG; i += stride) {
        const int slot = i >> 2, pr = i & 3;                       // 16-byte slot ((c2*NPL + plane)*4 + g)*64 + n
        const int n = slot % 64, g = (slot / 64) % 4, plane = (slot / 256) % C::NPL, c2 = slot / (256 * C::NPL);
        const int j0 = 2 * pr;
        const int k0 = 32 * c2 + 16 * (j0 >> 2) + 4 * g + (j0 & 3);
        const float v0 = W[(size_t)(r0 + k0) * ldw + c0 + n], v1 = W[(size_t)(r0 + k0 + 1) * ldw + c0 + n];
        img[i] = __uint_as_float(split_piece_bits<FMT>(v0, plane) | (split_piece_bits<FMT>(v1, plane) << 16));
    }
}

template <int FMT = kSplitBf16x3>
__device__ __forceinline__ void frag_planes(f32x4 x, f32x4 y, u32x4& hi, u32x4& mid, u32x4& lo) {
    // The fragment is the same for every stage of its segment, so the compiler would split it ONCE and keep all planes live across
    // the stages (96 registers at D = 256: ~450 B of scratch).  The empty asm makes the inputs opaque: the split is redone per stage,
    // 44 vector instructions per 24 MFMAs, and only one chunk's planes are live.
    asm volatile("" : "+v"(x.x), "+v"(x.y), "+v"(x.z), "+v"(x.w), "+v"(y.x), "+v"(y.y), "+v"(y.z), "+v"(y.w));
    unsigned h[4], m[4], l[4];
    split_pair<FMT>(x.x, x.y, h[0], m[0], l[0]); split_pair<FMT>(x.z, x.w, h[1], m[1], l[1]);
    split_pair<FMT>(y.x, y.y, h[2], m[2], l[2]); split_pair<FMT>(y.z, y.w, h[3], m[3], l[3]);
    hi = u32x4{h[0], h[1], h[2], h[3]}; mid = u32x4{m[0], m[1], m[2], m[3]}; lo = u32x4{l[0], l[1], l[2], l[3]};
}

// acc[0..3] (+)= chunks [part*CP, (part+1)*CP) of the fragment x the same chunks of a split panel image; `chunks` points at the
// first of them (in LDS: a ring slot; GLOBAL: the image in L2, cooperative tail pass).  The fragment's 8 values of a chunk are split
// right before the chunk's 24 MFMAs (the next chunk's under the current one's); weight planes rotate through 12 registers.
// FMT = kSplitF16x2: three products per unit (w_lo a_hi, w_hi a_lo, w_hi a_hi), both planes of the next unit fetched a unit ahead.
template <int D, bool ZERO, bool GLOBAL, int part, int FMT = kSplitBf16x3>
__device__ __forceinline__ void panel_part_mma_split(f32x4 (&acc)[4], const Frag<D>& a, const float* chunks, int li, int kq) {
    using C = PanelGruSplitCfg<D, FMT>;
    constexpr int NU = C::CP * 4;
#if GGNN_PANEL_REMAT
    asm volatile("" : "+v"(li), "+v"(kq));      // (the lane part of the address is recomputed per call, see stage_mma_split_at's REMAT)
#endif
    const unsigned voff = (unsigned)(kq * 64 + li) * 16u;
    const unsigned long long gb = reinterpret_cast<unsigned long long>(chunks);
    const float* sbase = chunks;
    if constexpr (GLOBAL) {     // wave-uniform base in scalar registers + one 32-bit per-lane offset (see panel_mma_global)
        const unsigned glo = __builtin_amdgcn_readfirstlane((unsigned)gb), ghi = __builtin_amdgcn_readfirstlane((unsigned)(gb >> 32));
        sbase = reinterpret_cast<const float*>(((unsigned long long)ghi << 32) | glo);
    }
    auto slot = [&](int u, int p) -> u32x4 {                        // unit u = (chunk cc, tile j), plane p
        const unsigned off = voff + (unsigned)((((u / 4) * C::NPL + p) * 4) * 64 + (u % 4) * 16) * 16u;
        if constexpr (GLOBAL) return __builtin_bit_cast(u32x4, ld4_b(sbase, off));
        else return *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(chunks) + off);
    };
    u32x4 ah, am, al;
    if constexpr (FMT == kSplitF16x2) {
        u32x4 wh = slot(0, 0), wm = slot(0, 1), nh = wh, nm = wm;
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int cc = u / 4, j = u % 4;
            const bool more = u + 1 < NU;
            if (j == 0) frag_planes<FMT>(a.v[2 * (part * C::CP + cc)], a.v[2 * (part * C::CP + cc) + 1], ah, am, al);
            f32x4 c = (ZERO && cc == 0) ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[j];
            if (more) { nm = slot(u + 1, 1); nh = slot(u + 1, 0); }
            __builtin_amdgcn_sched_barrier(0);
            c = mfma_f16(wm, ah, c);
            c = mfma_f16(wh, am, c);
            c = mfma_f16(wh, ah, c);
            __builtin_amdgcn_sched_barrier(0);
            wm = nm; wh = nh;
            acc[j] = c;
        }
        return;
    }
    u32x4 wh = slot(0, 0), wm = slot(0, 1), wl = slot(0, 2);
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int cc = u / 4, j = u % 4;
        const bool more = u + 1 < NU;
        if (j == 0) frag_planes(a.v[2 * (part * C::CP + cc)], a.v[2 * (part * C::CP + cc) + 1], ah, am, al);
        f32x4 c = (ZERO && cc == 0) ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[j];
        c = mfma_bf16(wl, ah, c);
        __builtin_amdgcn_sched_barrier(0);
        if (more) wl = slot(u + 1, 2);
        c = mfma_bf16(wm, am, c);
        c = mfma_bf16(wm, ah, c);
        __builtin_amdgcn_sched_barrier(0);
        if (more) wm = slot(u + 1, 1);
        c = mfma_bf16(wh, al, c);
        c = mfma_bf16(wh, am, c);
        c = mfma_bf16(wh, ah, c);
        __builtin_amdgcn_sched_barrier(0);
        if (more) wh = slot(u + 1, 0);
        acc[j] = c;
    }
}

// acc[0..3] (+)= fragment x panel image: per k chunk 4 ds_read_b128 feed 16 MFMAs; one chunk of read-ahead
template <int D, bool ZERO>
__device__ __forceinline__ void panel_mma(f32x4 (&acc)[4], const Frag<D>& a, const float* img, int li, int kq) {
    using C = PanelCfg<D>;
    const f32x4* base = reinterpret_cast<const f32x4*>(img) + kq * C::BN + li;
    f32x4 w[2][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) w[0][j] = base[j * 16];
#pragma unroll
    for (int c = 0; c < C::NC; ++c) {
        if (c + 1 < C::NC) {
#pragma unroll
            for (int j = 0; j < 4; ++j) w[(c + 1) & 1][j] = base[(c + 1) * 4 * C::BN + j * 16];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 cin = (ZERO && c == 0 && e == 0) ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[j];
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[c & 1][j][e], a.v[c][e], cin, 0, 0, 0);
            }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// The same product with the panel image read straight from GLOBAL memory (L2-resident) instead of LDS, three chunks of
// read-ahead: the cooperative tail pass, where every wave multiplies by a DIFFERENT panel at the same time (the ring holds one
// image per stage).  Same k order per output element as panel_mma.
template <int D, bool ZERO>
__device__ __forceinline__ void panel_mma_global(f32x4 (&acc)[4], const Frag<D>& a, const float* __restrict__ gimg, int li, int kq) {
    using C = PanelCfg<D>;
    // wave-uniform image base in scalar registers + ONE 32-bit per-lane byte offset; the chunk advances the scalar base and the
    // tile is an instruction immediate (per-lane 64-bit addresses for the 64 loads of a stage get hoisted out of the pass loop
    // by the compiler -- hundreds of register pairs, i.e. kilobytes of scratch)
    const unsigned long long gb = reinterpret_cast<unsigned long long>(gimg);
    const unsigned glo = __builtin_amdgcn_readfirstlane((unsigned)gb), ghi = __builtin_amdgcn_readfirstlane((unsigned)(gb >> 32));
    const float* sbase = reinterpret_cast<const float*>(((unsigned long long)ghi << 32) | glo);
    const unsigned voff = (unsigned)(kq * C::BN + li) * 16u;
    auto chunk = [&](int c, int j) { return ld4_b(sbase + (size_t)c * 4 * C::BN * 4, voff + 256u * j); };
    constexpr int AHEAD = 3;
    f32x4 w[AHEAD + 1][4];
#pragma unroll
    for (int c = 0; c < AHEAD && c < C::NC; ++c)
#pragma unroll
        for (int j = 0; j < 4; ++j) w[c][j] = chunk(c, j);
#pragma unroll
    for (int c = 0; c < C::NC; ++c) {
        if (c + AHEAD < C::NC) {
#pragma unroll
            for (int j = 0; j < 4; ++j) w[(c + AHEAD) % (AHEAD + 1)][j] = chunk(c + AHEAD, j);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 cin = (ZERO && c == 0 && e == 0) ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[j];
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[c % (AHEAD + 1)][j][e], a.v[c][e], cin, 0, 0, 0);
            }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// LDS-DMA of `BYTES` (a multiple of NW KiB) from src to LDS dst by an NW-wave workgroup (see dma_stage_image)
template <int BYTES, int NW>
__device__ __forceinline__ void dma_block(const float* src, float* dst, int wave, int lane) {
    constexpr int PER_WAVE = BYTES / (NW * 1024);
    static_assert(BYTES % (NW * 1024) == 0, "block must split into whole KiB per wave");
    char* d = reinterpret_cast<char*>(dst) + (size_t)wave * PER_WAVE * 1024;
    const unsigned voff = (unsigned)lane * 16u;
#pragma unroll
    for (int i0 = 0; i0 < PER_WAVE; i0 += 4) {
        const unsigned long long sb = reinterpret_cast<unsigned long long>(src) + (unsigned long long)wave * PER_WAVE * 1024 + (unsigned long long)i0 * 1024;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)sb);
        const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(sb >> 32));
        const char* s = reinterpret_cast<const char*>(((unsigned long long)hi << 32) | lo);
        lds_void* dl = (lds_void*)(d + i0 * 1024);
        if (i0 + 0 < PER_WAVE) __builtin_amdgcn_global_load_lds((glb_void*)(s + voff), dl, 16, 0, 0);
        if (i0 + 1 < PER_WAVE) __builtin_amdgcn_global_load_lds((glb_void*)(s + voff), dl, 16, 1024, 0);
        if (i0 + 2 < PER_WAVE) __builtin_amdgcn_global_load_lds((glb_void*)(s + voff), dl, 16, 2048, 0);
        if (i0 + 3 < PER_WAVE) __builtin_amdgcn_global_load_lds((glb_void*)(s + voff), dl, 16, 3072, 0);
    }
}

// ---- packed GRU weights: one image per stage, in stage order ---------------------------------------------------------------
//   phase R :  (s, p)            s < NS, p < NP     Wg rows [sD, (s+1)D), columns [64p, 64p+64)
//   phase UC:  p < NP:  for s < NX: U (Wg rows s, cols D + 64p ..), C (Wc rows s, cols 64p ..);  U (Wg rows h);  C (Wc rows h = r*h)
__host__ __device__ constexpr int panel_gru_images(int D, int nx) { return 3 * (nx + 1) * (D / 64); }

template <int D, bool SPLIT, int FMT = kSplitBf16x3>
__global__ void gru_panel_pack_kernel(const float* __restrict__ Wg, const float* __restrict__ Wc, int nx, float* __restrict__ out) {
    using C = PanelCfg<D>;
    const int ns = nx + 1, NP = C::NP;
    const int i = blockIdx.y;
    const float* W; int r0, c0, ldw;
    if (i < ns * NP) { W = Wg; r0 = (i / NP) * D; c0 = (i % NP) * C::BN; ldw = 2 * D; }
    else {
        const int j = i - ns * NP, p = j / (2 * ns), q = j % (2 * ns);          // q: U x_0, C x_0, .., U h, C rh
        const int s = q / 2;
        if (q % 2 == 0) { W = Wg; r0 = s * D; c0 = D + p * C::BN; ldw = 2 * D; }
        else { W = Wc; r0 = s * D; c0 = p * C::BN; ldw = D; }
    }
    const int first = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
    if constexpr (SPLIT) pack_panel_gru_split_image<D, FMT>(W, r0, c0, ldw, out + (size_t)i * PanelGruSplitCfg<D, FMT>::IMG, first, stride);
    else pack_panel_image<D>(W, r0, c0, ldw, out + (size_t)i * C::IMG, first, stride);
}

// the same product on RESIDENT planes of the fragment (split once by the caller: the ring transform, and the panel GRU's stages
// that multiply one fragment several times in a row)
template <int D, bool ZERO, int part, int FMT = kSplitBf16x3>
__device__ __forceinline__ void panel_part_mma_planes(f32x4 (&acc)[4], const u32x4 (&ph)[PanelGruSplitCfg<D>::NC2], const u32x4 (&pm)[PanelGruSplitCfg<D>::NC2],
                                                      const u32x4 (&pl)[PanelGruSplitCfg<D>::NC2], const float* chunks, int li, int kq) {
    using C = PanelGruSplitCfg<D, FMT>;
    constexpr int NU = C::CP * 4;
    asm volatile("" : "+v"(li), "+v"(kq));      // (the lane part of the address is recomputed per call: GGNN_PANEL_REMAT's reason)
    const unsigned voff = (unsigned)(kq * 64 + li) * 16u;
    auto slot = [&](int u, int p) -> u32x4 {                        // unit u = (chunk cc, tile j), plane p
        const unsigned off = voff + (unsigned)((((u / 4) * C::NPL + p) * 4) * 64 + (u % 4) * 16) * 16u;
        return *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(chunks) + off);
    };
    if constexpr (FMT == kSplitF16x2) {                             // (pm: the lo pieces; pl unused)
        u32x4 wh = slot(0, 0), wm = slot(0, 1), nh = wh, nm = wm;
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int cc = part * C::CP + u / 4, j = u % 4;
            const bool more = u + 1 < NU;
            f32x4 c = (ZERO && u / 4 == 0) ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[j];
            if (more) { nm = slot(u + 1, 1); nh = slot(u + 1, 0); }
            __builtin_amdgcn_sched_barrier(0);
            c = mfma_f16(wm, ph[cc], c);
            c = mfma_f16(wh, pm[cc], c);
            c = mfma_f16(wh, ph[cc], c);
            __builtin_amdgcn_sched_barrier(0);
            wm = nm; wh = nh;
            acc[j] = c;
        }
        return;
    }
    u32x4 wh = slot(0, 0), wm = slot(0, 1), wl = slot(0, 2);
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int cc = part * C::CP + u / 4, j = u % 4;
        const bool more = u + 1 < NU;
        f32x4 c = (ZERO && u / 4 == 0) ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[j];
        c = mfma_bf16(wl, ph[cc], c);
        __builtin_amdgcn_sched_barrier(0);
        if (more) wl = slot(u + 1, 2);
        c = mfma_bf16(wm, pm[cc], c);
        c = mfma_bf16(wm, ph[cc], c);
        __builtin_amdgcn_sched_barrier(0);
        if (more) wm = slot(u + 1, 1);
        c = mfma_bf16(wh, pl[cc], c);
        c = mfma_bf16(wh, pm[cc], c);
        c = mfma_bf16(wh, ph[cc], c);
        __builtin_amdgcn_sched_barrier(0);
        if (more) wh = slot(u + 1, 0);
        acc[j] = c;
    }
}

// SPLIT: the products on the bf16 pipe in 3-way split form; an image comes through the ring in PanelGruSplitCfg::PARTS parts.
// FMT (SPLIT): operand format of the products and images (ggnn_split.hpp), per launch (GruFusedArgs::fmt): kSplitF16x2 (two f16
// pieces, three products, whole 64 KiB images through the ring; the caller's proven operand range) or the exact kSplitBf16x3.
template <int FMT>
__device__ __forceinline__ f32x4 panel_sigmoid4(f32x4 z, f32x4 b_scaled) {      // (sigmoid4_scaled on an accumulator of 1 / acc_scale x the sum)
    constexpr float k = -kLog2e * SplitFmt<FMT>::acc_scale;
    return rcp_4(exp2_4(z * k + b_scaled) + 1.0f);
}
template <int FMT>
__device__ __forceinline__ f32x4 panel_tanh4(f32x4 z, f32x4 b_scaled) {
    constexpr float k = 2.0f * kLog2e * SplitFmt<FMT>::acc_scale;
    return 1.0f - 2.0f * rcp_4(exp2_4(z * k + b_scaled) + 1.0f);
}

template <int D, int NX, int NW, bool SAVE, bool SPLIT, int FMT = kSplitBf16x3>
__global__ __launch_bounds__(NW * 64) void ggnn_gru_panel_kernel(GruFusedArgs a, const float* __restrict__ packed) {
    static_assert(SPLIT || FMT == kSplitBf16x3, "the f32-MFMA kernels have no operand format");
    using C = PanelCfg<D>;
    using SC = PanelGruSplitCfg<D, FMT>;
    constexpr int IMGF = SPLIT ? SC::IMG : C::IMG;                   // floats per image in `packed`
    constexpr int SLOTF = SPLIT ? SC::PART : C::IMG;                 // floats per ring slot
    constexpr int PARTS = SPLIT ? SC::PARTS : 1;
    constexpr int NP = C::NP, NC = C::NC, NS = NX + 1;
    constexpr int NSTAGE = 3 * NS * NP;
    constexpr int R_STAGES = NS * NP;
    extern __shared__ __attribute__((aligned(16))) float lds_[];    // [biases | ring [2][IMG]]
    constexpr int BIAS_FLOATS = (4 * D + 63) / 64 * 64;
    float* bias_s = lds_;                      // [-log2e*bg (2D) | 2 log2e*bc (D) | bc (D)]
    float* ring = lds_ + BIAS_FLOATS;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kq = lane >> 4;
    const bool late = wave >= NW / 2;          // (the two waves of a SIMD: w and w + NW/2; see ggnn_gru_fused.hip)

    for (int i = tid; i < 4 * D; i += NW * 64)
        bias_s[i] = i < 2 * D ? -kLog2e * a.bg[i] : (i < 3 * D ? 2.0f * kLog2e * a.bc[i - 2 * D] : a.bc[i - 3 * D]);

    // tickets: full rounds of NW tiles per workgroup, then the rest spread thin over all workgroups (tail_w tiles each)
    const int wt_total = (a.V + 15) / 16;
    const int nb = gridDim.x;
    const int full_tk = wt_total / (NW * nb) * nb;
    const int rest = wt_total - full_tk * NW;
    const int tail_w = (rest + nb - 1) / nb;
    const int n_tk = full_tk + (tail_w ? (rest + tail_w - 1) / tail_w : 0);
    // One-tile tail tickets (tail_w == 1, the usual case) are worked COOPERATIVELY: wave p < NP takes panel p of the tile (phase R
    // for its 64 columns, r*h exchanged through LDS, then phase UC for its panel), weights straight from the images in L2 --
    // 3 NS stages on each of NP waves instead of 3 NS NP stages on one wave while seven idle.
    const bool coop_tail = (tail_w == 1) && (NP <= NW);
    const int n_main = coop_tail ? full_tk : n_tk;              // tickets [n_main, n_tk) are cooperative tail passes
    auto tile_of = [&](int t) -> int {
        if (t < full_tk) return t * NW + wave;
        if (t >= n_tk) return -1;
        if (coop_tail) return full_tk * NW + (t - full_tk);     // (every wave: the same tile)
        const int tl = full_tk * NW + (t - full_tk) * tail_w + wave;
        return (wave < tail_w && tl < wt_total) ? tl : -1;
    };
    auto row_of = [&](int t) -> int {          // clamped row of this lane in ticket t (always a valid row to read)
        const int tile = tile_of(t);
        const int r = (tile >= 0 ? tile : 0) * 16 + li;
        return r < a.V ? r : a.V - 1;
    };

    int cur = 0;
    auto dma = [&](const float* src, float* dst) {
        if constexpr (SPLIT) dma_image_asm<SC::PART_BYTES, NW>(src, dst, wave, lane);
        else dma_block<C::IMG_BYTES, NW>(src, dst, wave, lane);
    };
    auto publish = [&]() { if constexpr (SPLIT) dma_wait(); __syncthreads(); };
    dma(packed, ring);
    Frag<D> af;                                // the ONE resident activation fragment (segment of the current stage)
    int tk = blockIdx.x;
    if (tk < n_main) load_frag<D>(af, a.x[0], row_of(tk), kq);
    publish();

    for (; tk < n_main; tk += nb) {
        const int tile = tile_of(tk);
        const bool active = tile >= 0;                                  // wave-uniform
        const int row = active ? tile * 16 + li : a.V;                  // (>= V: nothing is stored)
        const int rowc = row < a.V ? row : a.V - 1;
        const bool last_pass = tk + nb >= n_main;             // (of the passes that use the ring)
        const int rown = last_pass ? 0 : row_of(tk + nb);

        // The pass body exists twice: for a wave WITH a tile, and for a wave without one (thin tail tickets), which only takes
        // part in the image DMA and the barriers.  (One body with `if (active)` around every MFMA block turns each stage into
        // a diamond whose accumulator phis the register allocator does not coalesce: 296 instead of 230 registers at D = 256.)
        auto run_pass = [&](auto active_c) {
            constexpr bool ACT = decltype(active_c)::value;
            // one stage: [late waves: DMA of the next image] MFMAs [early waves: DMA]; then `after()` (this wave's loads for
            // the NEXT stage: they fly while the partner wave multiplies) and the barrier that publishes the next image
            // Resident planes of the fragment the next stages multiply (GGNN_PANEL_PLANES, split form): phase R multiplies a segment's
            // fragment by NP panels in a row, phase UC multiplies an x segment's by its u and candidate panels back to back -- the
            // fragment is split ONCE for such a run (10 of the 24 splits of an NX = 1 pass go) and its f32 registers are free for
            // the next fragment's loads; stages that multiply a fragment once (h -> u, r*h -> c) split chunk by chunk as before.
            constexpr bool PLANES = SPLIT && GGNN_PANEL_PLANES;
            u32x4 qh[PLANES ? SC::NC2 : 1], qm[PLANES ? SC::NC2 : 1], ql[PLANES ? SC::NC2 : 1];
            auto make_planes = [&](const Frag<D>& A) {
                if constexpr (PLANES && ACT) {
#pragma unroll
                    for (int c2 = 0; c2 < SC::NC2; ++c2) {
                        const f32x4 x = A.v[2 * c2], y = A.v[2 * c2 + 1];
                        unsigned hh[4], mm[4], ll[4];
                        split_pair<FMT>(x.x, x.y, hh[0], mm[0], ll[0]); split_pair<FMT>(x.z, x.w, hh[1], mm[1], ll[1]);
                        split_pair<FMT>(y.x, y.y, hh[2], mm[2], ll[2]); split_pair<FMT>(y.z, y.w, hh[3], mm[3], ll[3]);
                        qh[c2] = u32x4{hh[0], hh[1], hh[2], hh[3]}; qm[c2] = u32x4{mm[0], mm[1], mm[2], mm[3]}; ql[c2] = u32x4{ll[0], ll[1], ll[2], ll[3]};
                    }
                }
            };
            auto stage = [&](auto zero_c, f32x4 (&acc)[4], const Frag<D>& A, int img_idx, auto&& after, auto planes_c) {
                constexpr bool USEP = PLANES && decltype(planes_c)::value;
                const int nidx = img_idx + 1 < NSTAGE ? img_idx + 1 : 0;
                const bool more = (img_idx + 1 < NSTAGE) || !last_pass;
                auto round = [&](auto part_c) {
                    constexpr int part = decltype(part_c)::value;
                    // what the ring's other slot receives during this round: the next part of this image, or the first of the next
                    const bool more_p = part + 1 < PARTS || more;
                    const float* nsrc = part + 1 < PARTS ? packed + (size_t)img_idx * IMGF + (size_t)(part + 1) * SLOTF
                                                         : packed + (size_t)nidx * IMGF;
                    float* ndst = ring + (cur ^ 1) * SLOTF;
                    const bool dma_first = late || GGNN_PANEL_DMA_FIRST;
                    if (dma_first && more_p) dma(nsrc, ndst);
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (ACT) {
                        if constexpr (USEP) panel_part_mma_planes<D, decltype(zero_c)::value && part == 0, part, FMT>(acc, qh, qm, ql, ring + cur * SLOTF, li, kq);
                        else if constexpr (SPLIT) panel_part_mma_split<D, decltype(zero_c)::value && part == 0, false, part, FMT>(acc, A, ring + cur * SLOTF, li, kq);
                        else panel_mma<D, decltype(zero_c)::value>(acc, A, ring + cur * SLOTF, li, kq);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (!dma_first && more_p) dma(nsrc, ndst);
                    if constexpr (ACT) { if constexpr (part == PARTS - 1) after(); }
                    publish();
                    cur ^= 1;
                };
                round(std::integral_constant<int, 0>{});
                if constexpr (PARTS > 1) round(std::integral_constant<int, 1>{});
                if constexpr (PARTS > 2) round(std::integral_constant<int, 2>{});
            };
            auto nothing = [] {};

            // ---- phase R: r columns, all panels, segment by segment ---------------------------------------------------
            f32x4 acc_r[NP][4];
#define GGNN_R_STAGE(S, P)                                                                                          \
            if constexpr ((S) < NS && (P) < NP) {                                                                   \
                auto next_seg = [&] {   /* after the last panel of a segment: fetch the next segment's fragment */   \
                    if constexpr ((P) == NP - 1 && (S) + 1 < NS)                                                     \
                        load_frag<D>(af, (S) + 1 < NX ? a.x[(S) + 1 < NX ? (S) + 1 : 0] : a.h, rowc, kq);          \
                };                                                                                                  \
                if constexpr ((P) == 0) make_planes(af);                                                            \
                stage(std::integral_constant<bool, (S) == 0>{}, acc_r[(P) < NP ? (P) : 0], af, (S) * NP + (P), next_seg, std::true_type{}); \
            }
#define GGNN_R_SEG(S) GGNN_R_STAGE(S, 0) GGNN_R_STAGE(S, 1) GGNN_R_STAGE(S, 2) GGNN_R_STAGE(S, 3)
            GGNN_R_SEG(0) GGNN_R_SEG(1) GGNN_R_SEG(2) GGNN_R_SEG(3)
#undef GGNN_R_SEG
#undef GGNN_R_STAGE
            static_assert(NP <= 4 && NS <= 4, "stage macros cover up to 4 panels x 4 segments");

            // ---- r = sigmoid(.), rh = r * h in activation-fragment layout (af holds h here) ---------------------------------
            Frag<D> rh;
            if constexpr (ACT) {
#pragma unroll
                for (int nt = 0; nt < NC; ++nt) {
                    const int col = nt * 16 + 4 * kq;
                    const f32x4 r = panel_sigmoid4<FMT>(acc_r[nt / 4][nt % 4], ld4(bias_s + col));
                    if constexpr (SAVE) { if (row < a.V) st4_b(a.save_r, ((unsigned)row * D + col) * 4u, r); }
                    rh.v[nt] = r * af.v[nt];
                    __builtin_amdgcn_sched_barrier(0);      // one tile at a time: no 16-deep batch of bias reads in flight
                }
                load_frag<D>(af, a.x[0], rowc, kq);                     // first fragment of phase UC
            }

            // ---- phase UC: u and candidate columns, panel by panel (run-time loop: the panel only enters addresses) ----------
#pragma unroll 1
            for (int p = 0; p < NP; ++p) {
                f32x4 acc_u[4], acc_c[4];
                const int img0 = R_STAGES + p * 2 * NS;
                const bool last_panel = p + 1 == NP;
#define GGNN_X_STAGES(S)                                                                                            \
                if constexpr ((S) < NX) {                                                                           \
                    make_planes(af);                                                                                \
                    stage(std::integral_constant<bool, (S) == 0>{}, acc_u, af, img0 + 2 * (S), nothing, std::true_type{}); \
                    auto next_seg = [&] { load_frag<D>(af, (S) + 1 < NX ? a.x[(S) + 1 < NX ? (S) + 1 : 0] : a.h, rowc, kq); }; \
                    stage(std::integral_constant<bool, (S) == 0>{}, acc_c, af, img0 + 2 * (S) + 1, next_seg, std::true_type{}); \
                }
                GGNN_X_STAGES(0) GGNN_X_STAGES(1) GGNN_X_STAGES(2)
#undef GGNN_X_STAGES
                // h -> u columns; afterwards the fragment registers are free: the next panel's (or next pass's) x_0 goes there
                auto after_h = [&] { load_frag<D>(af, a.x[0], last_panel ? rown : rowc, kq); };
                stage(std::false_type{}, acc_u, af, img0 + 2 * NX, after_h, std::false_type{});
                stage(std::false_type{}, acc_c, rh, img0 + 2 * NX + 1, nothing, std::false_type{});
                if constexpr (ACT) {
                    if (row < a.V) {
                        // (the h columns of this panel are fetched here, not a stage ahead: 16 more live registers across the
                        //  r*h stage; the ~0.5k clocks of L2 latency per 65k-clock panel are covered by the partner wave)
                        f32x4 hcol[4];
#pragma unroll
                        for (int nt = 0; nt < 4; ++nt) hcol[nt] = ld4_b(a.h, ((unsigned)row * D + p * 64 + nt * 16 + 4 * kq) * 4u);
#pragma unroll
                        for (int nt = 0; nt < 4; ++nt) {
                            const int col = p * 64 + nt * 16 + 4 * kq;
                            const f32x4 u = panel_sigmoid4<FMT>(acc_u[nt], ld4(bias_s + D + col));
                            f32x4 c;
                            if (a.act == GGNN_ACT_TANH) {
                                c = panel_tanh4<FMT>(acc_c[nt], ld4(bias_s + 2 * D + col));
                            } else {
                                c = acc_c[nt] * SplitFmt<FMT>::acc_scale + ld4(bias_s + 3 * D + col);
                                c.x = fmaxf(c.x, 0.f); c.y = fmaxf(c.y, 0.f); c.z = fmaxf(c.z, 0.f); c.w = fmaxf(c.w, 0.f);
                            }
                            st4_b(a.h_out, ((unsigned)row * D + col) * 4u, u * hcol[nt] + (1.0f - u) * c);
                            if constexpr (SAVE) {
                                st4_b(a.save_u, ((unsigned)row * D + col) * 4u, u);
                                st4_b(a.save_c, ((unsigned)row * D + col) * 4u, c);
                            }
                        }
                    }
                }
            }
        };
        if (active) run_pass(std::true_type{});
        else run_pass(std::false_type{});
    }

    // ---- cooperative tail passes (the ring is idle from here on; its first bytes carry the r*h exchange block) ---------------
    constexpr int RHP = D + 4;                                  // row pitch of the exchange block [16][RHP]
    for (; tk < n_tk; tk += nb) {
        const int tile = tile_of(tk);                           // the same tile for every wave
        const int row = tile * 16 + li;
        const int rowc = row < a.V ? row : a.V - 1;
        const int p = wave;                                     // this wave's panel
        const bool on = p < NP;
        float* rh_x = ring;
        auto gimg = [&](int img_idx) { return packed + (size_t)img_idx * IMGF; };
        auto gmma = [&](auto zero_c, f32x4 (&acc)[4], const Frag<D>& A, const float* img) {     // whole image, straight from L2
            if constexpr (SPLIT) {
                panel_part_mma_split<D, decltype(zero_c)::value, true, 0, FMT>(acc, A, img, li, kq);
                if constexpr (SC::PARTS > 1) panel_part_mma_split<D, false, true, 1, FMT>(acc, A, img + (size_t)SC::PART, li, kq);
                if constexpr (SC::PARTS > 2) panel_part_mma_split<D, false, true, 2, FMT>(acc, A, img + (size_t)2 * SC::PART, li, kq);
            } else panel_mma_global<D, decltype(zero_c)::value>(acc, A, img, li, kq);
        };
        Frag<D> rh;
        f32x4 accr[4];
        if (on) {
            // phase R for the panel's 64 columns: segments in order, h last
#define GGNN_CR_STAGE(S)                                                                                            \
            if constexpr ((S) < NS) {                                                                               \
                load_frag<D>(af, (S) < NX ? a.x[(S) < NX ? (S) : 0] : a.h, rowc, kq);                               \
                gmma(std::integral_constant<bool, (S) == 0>{}, accr, af, gimg((S) * NP + p));                       \
            }
            GGNN_CR_STAGE(0) GGNN_CR_STAGE(1) GGNN_CR_STAGE(2) GGNN_CR_STAGE(3)
#undef GGNN_CR_STAGE
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const int col = p * 64 + nt * 16 + 4 * kq;
                const f32x4 r = panel_sigmoid4<FMT>(accr[nt], ld4(bias_s + col));
                const f32x4 hv = ld4_b(a.h, ((unsigned)rowc * D + col) * 4u);
                if constexpr (SAVE) { if (row < a.V) st4_b(a.save_r, ((unsigned)row * D + col) * 4u, r); }
                *reinterpret_cast<f32x4*>(rh_x + li * RHP + col) = r * hv;
            }
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < NC; ++c) rh.v[c] = *reinterpret_cast<const f32x4*>(rh_x + li * RHP + 16 * c + 4 * kq);
        __syncthreads();                                        // (the block is rewritten by the next cooperative pass)
        if (on) {
            f32x4 acc_u[4], acc_c[4];
            const int img0 = R_STAGES + p * 2 * NS;
#define GGNN_CX_STAGES(S)                                                                                           \
            if constexpr ((S) < NX) {                                                                               \
                load_frag<D>(af, a.x[(S)], rowc, kq);                                                               \
                gmma(std::integral_constant<bool, (S) == 0>{}, acc_u, af, gimg(img0 + 2 * (S)));                    \
                gmma(std::integral_constant<bool, (S) == 0>{}, acc_c, af, gimg(img0 + 2 * (S) + 1));                \
            }
            GGNN_CX_STAGES(0) GGNN_CX_STAGES(1) GGNN_CX_STAGES(2)
#undef GGNN_CX_STAGES
            load_frag<D>(af, a.h, rowc, kq);
            gmma(std::false_type{}, acc_u, af, gimg(img0 + 2 * NX));
            gmma(std::false_type{}, acc_c, rh, gimg(img0 + 2 * NX + 1));
            if (row < a.V) {
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    const int col = p * 64 + nt * 16 + 4 * kq;
                    const f32x4 hcol = ld4_b(a.h, ((unsigned)row * D + col) * 4u);
                    const f32x4 u = panel_sigmoid4<FMT>(acc_u[nt], ld4(bias_s + D + col));
                    f32x4 c;
                    if (a.act == GGNN_ACT_TANH) {
                        c = panel_tanh4<FMT>(acc_c[nt], ld4(bias_s + 2 * D + col));
                    } else {
                        c = acc_c[nt] * SplitFmt<FMT>::acc_scale + ld4(bias_s + 3 * D + col);
                        c.x = fmaxf(c.x, 0.f); c.y = fmaxf(c.y, 0.f); c.z = fmaxf(c.z, 0.f); c.w = fmaxf(c.w, 0.f);
                    }
                    st4_b(a.h_out, ((unsigned)row * D + col) * 4u, u * hcol + (1.0f - u) * c);
                    if constexpr (SAVE) {
                        st4_b(a.save_u, ((unsigned)row * D + col) * 4u, u);
                        st4_b(a.save_c, ((unsigned)row * D + col) * 4u, c);
                    }
                }
            }
        }
    }
}

template <int D, int NX, bool SAVE, bool SPLIT, int FMT = kSplitBf16x3>
static int launch_gru_panel_m(const GruFusedArgs& a_in, float* packed, hipStream_t st) {
    using C = PanelCfg<D>;
    constexpr int NW = 8;
    GruFusedArgs a = a_in;
    if (a.Wg) {   // raw weights given: build the stage images first
        hipLaunchKernelGGL((gru_panel_pack_kernel<D, SPLIT, FMT>), dim3(8, panel_gru_images(D, NX)), dim3(256), 0, st, a.Wg, a.Wc, NX, packed);
        GGNN_CHECK_HIP(hipGetLastError());
    }
    if (a.h == nullptr) return GGNN_OK;   // pack-only call
    if ((unsigned long long)a.V * D >= (1ULL << 30))
        return fail(GGNN_E_UNSUPPORTED, "fused GRU indexes with 32-bit byte offsets: V*D must be < 2^30 (V=%d, D=%d)", a.V, D);
    // ring: two images (f32) / two parts of an image (split); the cooperative tail's r*h exchange block [16][D + 4] shares it
    size_t ringb = (size_t)2 * (SPLIT ? PanelGruSplitCfg<D, FMT>::PART_BYTES : C::IMG_BYTES);
    if (ringb < (size_t)16 * (D + 4) * sizeof(float)) ringb = (size_t)16 * (D + 4) * sizeof(float);
    const size_t lds = ringb + (size_t)((4 * D + 63) / 64 * 64) * sizeof(float);
    const int wt_total = (a.V + 15) / 16;
    int nb = num_cus();
    if (nb > wt_total) nb = wt_total;
    static std::atomic<unsigned long long> lds_ok{0};
    if (lds > 64 * 1024) GGNN_CHECK_HIP(allow_dynamic_lds(&ggnn_gru_panel_kernel<D, NX, NW, SAVE, SPLIT, FMT>, lds, lds_ok));
    hipLaunchKernelGGL((ggnn_gru_panel_kernel<D, NX, NW, SAVE, SPLIT, FMT>), dim3(nb), dim3(NW * 64), lds, st, a, (const float*)packed);
    GGNN_CHECK_HIP(hipGetLastError());
    return GGNN_OK;
}

template <int D, int NX, bool SAVE>
static int launch_gru_panel(const GruFusedArgs& a, float* packed, hipStream_t st) {
    if (split_matrix_path()) return gru_launch_fmt(a.fmt) == kSplitF16x2 ? launch_gru_panel_m<D, NX, SAVE, true, kSplitF16x2>(a, packed, st)
                                                                 : launch_gru_panel_m<D, NX, SAVE, true>(a, packed, st);
    return launch_gru_panel_m<D, NX, SAVE, false>(a, packed, st);
}

template <int D>
static int dispatch_panel_nx(const GruFusedArgs& a, float* packed, hipStream_t st) {
    const bool save = a.save_r || a.save_u || a.save_c;
    if (save && !(a.save_r && a.save_u && a.save_c))
        return fail(GGNN_E_INVALID, "save_r / save_u / save_c must be given together");
    if (a.g_H) return fail(GGNN_E_UNSUPPORTED, "hidden size %d has no gather-fused GRU (use ggnn_gather_segment_sum_f32 + ggnn_gru_packed_f32)", D);
    switch (a.nx) {
        case 1: return save ? launch_gru_panel<D, 1, true>(a, packed, st) : launch_gru_panel<D, 1, false>(a, packed, st);
        case 2: return save ? launch_gru_panel<D, 2, true>(a, packed, st) : launch_gru_panel<D, 2, false>(a, packed, st);
        case 3: return save ? launch_gru_panel<D, 3, true>(a, packed, st) : launch_gru_panel<D, 3, false>(a, packed, st);
        default: return fail(GGNN_E_INVALID, "nx %d outside 1..3", a.nx);
    }
}

int gru_panel_supported(int D) { return D == 128 || D == 192 || D == 256; }

int gru_panel_pack_floats(int D, int nx) {
    if (!gru_panel_supported(D)) return 0;
    // (sized for either operand format of the split form: the larger of the two image sizes)
    auto mx = [](int a, int b) { return a > b ? a : b; };
    const int img = !split_matrix_path() ? D * 64
                  : (D == 128 ? mx(PanelGruSplitCfg<128>::IMG, PanelGruSplitCfg<128, kSplitF16x2>::IMG)
                     : (D == 192 ? mx(PanelGruSplitCfg<192>::IMG, PanelGruSplitCfg<192, kSplitF16x2>::IMG)
                                 : mx(PanelGruSplitCfg<256>::IMG, PanelGruSplitCfg<256, kSplitF16x2>::IMG)));
    return panel_gru_images(D, nx) * img;
}

int gru_panel_dispatch(const GruFusedArgs& a, int D, float* packed, hipStream_t st) {
    switch (D) {
        case 128: return dispatch_panel_nx<128>(a, packed, st);
        case 192: return dispatch_panel_nx<192>(a, packed, st);
        case 256: return dispatch_panel_nx<256>(a, packed, st);
        default: return fail(GGNN_E_UNSUPPORTED, "no panel GRU for hidden size %d", D);
    }
}

// ---- compacted message transform on panels ------------------------------------------------------------------------------
constexpr int kMaxTypesP = 64;
struct PanelRows {
    int row_off[kMaxTypesP + 1];       // compact rows of type t: row_off[t] .. row_off[t+1]-1
    int wg_off[kMaxTypesP + 1];        // workgroups of type t (all its panels): wg_off[t] .. wg_off[t+1]-1, a multiple of NP each
    int T;
};

// workgroup (type t, panel p, j-th of the type's workgroups on that panel): keeps image (t, p) in LDS and walks 16-row tiles
// j*NW + wave, + stride, ... of the type's active pairs; the rows of tile k+1 are fetched under the MFMAs of tile k.
template <int D, int NW, bool SPLIT>
__global__ __launch_bounds__(NW * 64) void msg_transform_panel_kernel(const float* __restrict__ h, const int* __restrict__ pair_node,
                                                                      PanelRows pr, const float* __restrict__ packed,
                                                                      float* __restrict__ Hc) {
    using C = PanelCfg<D>;
    constexpr int IMGF = SPLIT ? PanelSplitCfg<D>::IMG : C::IMG;
    constexpr int NP = C::NP;
    extern __shared__ __attribute__((aligned(16))) float img[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kq = lane >> 4;
    int t = 0;
    while (t + 1 < pr.T && (int)blockIdx.x >= pr.wg_off[t + 1]) ++t;
    const int wg_in_type = (int)blockIdx.x - pr.wg_off[t];
    const int per_panel = (pr.wg_off[t + 1] - pr.wg_off[t]) / NP;       // workgroups of this type on each panel
    const int p = wg_in_type / per_panel, j = wg_in_type % per_panel;
    const int row_beg = pr.row_off[t], row_end = pr.row_off[t + 1];
    const int n_wt = (row_end - row_beg + 15) / 16;
    const int stride = per_panel * NW;
    int idx = j * NW + wave;

    auto row_of = [&](int i) { const int r = row_beg + i * 16 + li; return r < row_end ? r : row_end - 1; };
    Frag<D> a, an;
    int node_0 = 0, node_n = 0;
    if (idx < n_wt) node_0 = pair_node[row_of(idx)];
    if (idx + stride < n_wt) node_n = pair_node[row_of(idx + stride)];
    if constexpr (SPLIT) dma_image_asm<PanelSplitCfg<D>::IMG_BYTES, NW>(packed + (size_t)(t * NP + p) * IMGF, img, wave, lane);
    else dma_block<C::IMG_BYTES, NW>(packed + (size_t)(t * NP + p) * IMGF, img, wave, lane);
    if (idx < n_wt) load_frag<D>(a, h, node_0, kq);
    if constexpr (SPLIT) dma_wait();
    __syncthreads();

    while (idx < n_wt) {
        const int idx_n = idx + stride;
        if (idx_n < n_wt) load_frag<D>(an, h, node_n, kq);
        if (idx_n + stride < n_wt) node_n = pair_node[row_of(idx_n + stride)];
        f32x4 acc[4];
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (SPLIT) panel_mma_split<D, true>(acc, a, img, li, kq);
        else panel_mma<D, true>(acc, a, img, li, kq);
        const int r = row_beg + idx * 16 + li;
        if (r < row_end) {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) st4_b(Hc, ((unsigned)r * (unsigned)D + p * 64 + nt * 16 + 4 * kq) * 4u, acc[nt]);
        }
        a = an;
        idx = idx_n;
    }
}

// ---- the same transform with the panel images STREAMED and the rows read once (round 4, split form) ------------------------------
// msg_transform_panel_kernel keeps ONE (type, panel) image per workgroup, so every state row is fetched once per panel (D = 256: four
// times; 1.11 GB of traffic for 0.48 GB algorithmic) and split once per panel.  Here a workgroup is bound to a TYPE: a pass takes
// one 16-row tile per wave, splits its fragment ONCE into resident bf16 planes (96 registers at D = 256; the f32 fragment's registers
// then receive the next tile's rows) and walks the type's NP panels, whose images come through a two-slot LDS ring in
// PanelGruSplitCfg::PARTS parts -- the panel GRU's chunk-major image format and ring (48 KiB parts at D = 256).  Per pass and CU
// NP x 96 KiB of weights from L2 instead of (NP - 1) x 128 KiB of rows from HBM / Infinity Cache, and 352 instead of NP x 352
// split instructions per tile.
// pr.wg_off counts the workgroups of a type (NOT times NP); packed: [T][NP] images in pack_panel_gru_split_image's format
// FMT (round 5): operand format of the products and of the images, per launch (kSplitF16x2: two planes, whole images through the
// ring -- no parts -- for a forward transform whose states and edge weights the caller has proven in range; formats.py).
template <int D, int NW, int FMT = kSplitBf16x3>
__global__ __launch_bounds__(NW * 64) void msg_transform_ring_kernel(const float* __restrict__ h, const int* __restrict__ pair_node,
                                                                     PanelRows pr, const float* __restrict__ packed, float* __restrict__ Hc) {
    using C = PanelCfg<D>;
    using SC = PanelGruSplitCfg<D, FMT>;
    constexpr int NP = C::NP, PARTS = SC::PARTS, SLOTF = SC::PART, IMGF = SC::IMG, NC2 = SC::NC2;
    extern __shared__ __attribute__((aligned(16))) float ring[];     // [2][PART]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kq = lane >> 4;
    int t = 0;
    while (t + 1 < pr.T && (int)blockIdx.x >= pr.wg_off[t + 1]) ++t;
    const int j = (int)blockIdx.x - pr.wg_off[t];
    const int stride = (pr.wg_off[t + 1] - pr.wg_off[t]) * NW;
    const int row_beg = pr.row_off[t], row_end = pr.row_off[t + 1];
    const int n_wt = (row_end - row_beg + 15) / 16;
    const int n_pass = j * NW < n_wt ? (n_wt - j * NW + stride - 1) / stride : 0;     // passes of this workgroup (its wave 0 has a tile in each)
    int idx = j * NW + wave;
    const float* timg = packed + (size_t)t * NP * IMGF;

    auto row_of = [&](int i) { const int r = row_beg + i * 16 + li; return r < row_end ? r : row_end - 1; };
    int cur = 0;
    auto dma = [&](const float* src, float* dst) { dma_image_asm<SC::PART_BYTES, NW>(src, dst, wave, lane); };
    auto publish = [&]() { dma_wait(); __syncthreads(); };
    // the barrier of a round in which this wave issued `keep` vector-memory instructions BEHIND its DMA pieces (the next tile's rows,
    // a panel's stores): the counter retires in order, so "at most keep outstanding" proves the pieces landed and leaves the rest in
    // flight across the barrier (they have until the next round's barrier)
    auto publish_keep = [&](int keep) {
        switch (keep) {                                              // (wave-uniform; s_waitcnt takes an immediate)
#define GGNN_WC(N) case N: asm volatile("s_waitcnt vmcnt(" #N ") lgkmcnt(0)" ::: "memory"); break;
            GGNN_WC(0) GGNN_WC(1) GGNN_WC(2) GGNN_WC(3) GGNN_WC(4) GGNN_WC(5) GGNN_WC(6) GGNN_WC(7) GGNN_WC(8) GGNN_WC(9) GGNN_WC(10)
            GGNN_WC(11) GGNN_WC(12) GGNN_WC(13) GGNN_WC(14) GGNN_WC(15) GGNN_WC(16) GGNN_WC(17) GGNN_WC(18) GGNN_WC(19) GGNN_WC(20)
            GGNN_WC(21) GGNN_WC(22) GGNN_WC(23) GGNN_WC(24) GGNN_WC(25) GGNN_WC(26)
#undef GGNN_WC
            default: asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); break;
        }
        __builtin_amdgcn_s_barrier();
    };
    static_assert(C::NC == 8 || C::NC == 12 || C::NC == 16, "publish_keep knows the row-load counts of D = 128 / 192 / 256");
    Frag<D> a;
    int node_n = 0;
    if (n_pass > 0) dma(timg, ring);
    if (idx < n_wt) load_frag<D>(a, h, pair_node[row_of(idx)], kq);
    if (idx + stride < n_wt) node_n = pair_node[row_of(idx + stride)];
    publish();

    // the rounds of one pass: NP panels x PARTS parts; ACT: this wave has a tile (else it only feeds the ring and meets the barriers)
    auto run_pass = [&](auto active_c, bool last_pass) {
        constexpr bool ACT = decltype(active_c)::value;
        u32x4 ph[NC2], pm[NC2], pl[NC2];
        const int r = row_beg + idx * 16 + li;
        if constexpr (ACT) {
#pragma unroll
            for (int c2 = 0; c2 < NC2; ++c2) {
                const f32x4 x = a.v[2 * c2], y = a.v[2 * c2 + 1];
                unsigned hh[4], mm[4], ll[4];
                split_pair<FMT>(x.x, x.y, hh[0], mm[0], ll[0]); split_pair<FMT>(x.z, x.w, hh[1], mm[1], ll[1]);
                split_pair<FMT>(y.x, y.y, hh[2], mm[2], ll[2]); split_pair<FMT>(y.z, y.w, hh[3], mm[3], ll[3]);
                ph[c2] = u32x4{hh[0], hh[1], hh[2], hh[3]}; pm[c2] = u32x4{mm[0], mm[1], mm[2], mm[3]}; pl[c2] = u32x4{ll[0], ll[1], ll[2], ll[3]};
            }
        }
        // the next tile's rows go into the fragment's registers (dead now) in the pass's first round, BEHIND the round's DMA
        const bool has_next = ACT && idx + stride < n_wt;
        const int node_next = node_n;
        if constexpr (ACT) { if (idx + 2 * stride < n_wt) node_n = pair_node[row_of(idx + 2 * stride)]; }    // (older than every DMA of the pass)
        f32x4 acc[4];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            auto round = [&](auto part_c) {
                constexpr int part = decltype(part_c)::value;
                const bool more = part + 1 < PARTS || p + 1 < NP || !last_pass;
                const float* nsrc = part + 1 < PARTS ? timg + (size_t)p * IMGF + (size_t)(part + 1) * SLOTF
                                                     : timg + (size_t)(p + 1 < NP ? p + 1 : 0) * IMGF;
                // Every wave issues its pieces of the next part FIRST (a round is ~3k clocks of MFMAs for the SIMD's two waves: a part
                // issued behind a burst would not land before the round's barrier), then -- first round of a pass -- the next tile's rows.
                int keep = 0;
                if (more) dma(nsrc, ring + (cur ^ 1) * SLOTF);
                if constexpr (ACT) { if (p == 0 && part == 0 && has_next) { load_frag<D>(a, h, node_next, kq); keep += C::NC; } }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (ACT) panel_part_mma_planes<D, part == 0, part, FMT>(acc, ph, pm, pl, ring + cur * SLOTF, li, kq);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (ACT && part == PARTS - 1) {
                    // (every tile has a valid first row: the four stores are issued by every wave with a tile)
                    if (r < row_end) {
#pragma unroll
                        for (int nt = 0; nt < 4; ++nt) st4_b(Hc, ((unsigned)r * (unsigned)D + p * 64 + nt * 16 + 4 * kq) * 4u, acc[nt] * SplitFmt<FMT>::acc_scale);
                    }
                    keep += 4;
                }
                // the rows / the stores were issued BEHIND the round's DMA pieces: they may stay in flight across the barrier
                publish_keep(keep);
                cur ^= 1;
            };
            round(std::integral_constant<int, 0>{});
            if constexpr (PARTS > 1) round(std::integral_constant<int, 1>{});
            if constexpr (PARTS > 2) round(std::integral_constant<int, 2>{});
        }
        idx += stride;
    };
    int pass = 0;
    for (; pass < n_pass && idx < n_wt; ++pass) run_pass(std::true_type{}, pass + 1 == n_pass);
    for (; pass < n_pass; ++pass) run_pass(std::false_type{}, pass + 1 == n_pass);
}

// GGNN_PANEL_TRANSFORM: 1 (default) the ring kernel above (split path only), 0 the stationary-image kernel.  Read once: the packed
// edge-weight images are in the format of the kernel that will multiply them.
static bool transform_ring() {
    static const bool v = [] { const char* e = getenv("GGNN_PANEL_TRANSFORM"); return !e || atoi(e) != 0; }();
    return v && split_matrix_path();
}

template <int D, bool SPLIT, int FMT = kSplitBf16x3>
__global__ void edge_weight_panel_pack_kernel(const float* __restrict__ W, float* __restrict__ out, int ring_format) {
    using C = PanelCfg<D>;
    const int t = blockIdx.y / C::NP, p = blockIdx.y % C::NP;
    const int first = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
    if constexpr (SPLIT) {
        if (ring_format) pack_panel_gru_split_image<D, FMT>(W + (size_t)t * D * D, 0, p * C::BN, D, out + (size_t)blockIdx.y * PanelGruSplitCfg<D, FMT>::IMG, first, stride);
        else pack_panel_split_image<D>(W + (size_t)t * D * D, 0, p * C::BN, D, out + (size_t)blockIdx.y * PanelSplitCfg<D>::IMG, first, stride);
    } else pack_panel_image<D>(W + (size_t)t * D * D, 0, p * C::BN, D, out + (size_t)blockIdx.y * C::IMG, first, stride);
}
static_assert(PanelGruSplitCfg<256>::IMG == PanelSplitCfg<256>::IMG && PanelGruSplitCfg<192>::IMG == PanelSplitCfg<192>::IMG &&
              PanelGruSplitCfg<128>::IMG == PanelSplitCfg<128>::IMG, "both split image formats of the transform have one size");

// floats of ONE (type, panel) image of the transform, in the process's matrix path
int transform_panel_image_floats(int D) {
    const bool sp = split_matrix_path();
    switch (D) {
        case 128: return sp ? PanelSplitCfg<128>::IMG : PanelCfg<128>::IMG;
        case 192: return sp ? PanelSplitCfg<192>::IMG : PanelCfg<192>::IMG;
        case 256: return sp ? PanelSplitCfg<256>::IMG : PanelCfg<256>::IMG;
        default: return 0;
    }
}

// FMT: only the ring kernel (split path) has the two-piece f16 form; the stationary-image kernel and the f32 path ignore it
template <int D, bool SPLIT, int FMT = kSplitBf16x3>
static int launch_transform_panel_m(const float* h, const float* W, const int* pair_node, const int* row_off, int T, int V,
                                  float* packed, float* Hc, hipStream_t st) {
    using C = PanelCfg<D>;
    constexpr int NW = 8, NP = C::NP;
    if (W) {
        hipLaunchKernelGGL((edge_weight_panel_pack_kernel<D, SPLIT, FMT>), dim3(8, T * NP), dim3(256), 0, st, W, packed, (int)(SPLIT && transform_ring()));
        GGNN_CHECK_HIP(hipGetLastError());
    }
    const int R = row_off[T];
    if (R == 0 || h == nullptr) return GGNN_OK;
    if ((unsigned long long)R * D >= (1ULL << 30) || (unsigned long long)V * D >= (1ULL << 30))
        return fail(GGNN_E_UNSUPPORTED, "compacted transform indexes with 32-bit byte offsets: rows*D and V*D must be < 2^30");
    // one workgroup per CU (64 KiB image + ~176 VGPRs per wave); every type gets workgroups in proportion to its rows,
    // the same number on each of its NP panels, at least one per panel, no more than it has 8-tile rounds
    PanelRows pr{};
    pr.T = T;
    const int budget = num_cus() / NP > 0 ? num_cus() / NP : 1;          // workgroups per panel, all types together
    pr.wg_off[0] = 0;
    for (int t = 0; t < T; ++t) {
        pr.row_off[t] = row_off[t];
        const long long rows = row_off[t + 1] - row_off[t];
        long long n = rows * budget / R;
        const long long rounds = (rows + 16 * NW - 1) / (16 * NW);
        if (n > rounds) n = rounds;
        if (rows > 0 && n < 1) n = 1;
        pr.wg_off[t + 1] = pr.wg_off[t] + (int)n * NP;
    }
    pr.row_off[T] = R;
    if constexpr (SPLIT) {
        if (transform_ring()) {
            // workgroups per type in proportion to its rows (one per CU all types together), no more than it has 8-tile passes
            PanelRows rr{};
            rr.T = T; rr.wg_off[0] = 0;
            for (int t = 0; t < T; ++t) {
                rr.row_off[t] = row_off[t];
                const long long rows = row_off[t + 1] - row_off[t];
                long long n = rows * num_cus() / R;
                const long long rounds = (rows + 16 * NW - 1) / (16 * NW);
                if (n > rounds) n = rounds;
                if (rows > 0 && n < 1) n = 1;
                rr.wg_off[t + 1] = rr.wg_off[t] + (int)n;
            }
            rr.row_off[T] = R;
            static std::atomic<unsigned long long> lds_ok_r{0};
            constexpr size_t lds_r = (size_t)2 * PanelGruSplitCfg<D, FMT>::PART_BYTES;
            if (lds_r > 48 * 1024) GGNN_CHECK_HIP(allow_dynamic_lds(&msg_transform_ring_kernel<D, NW, FMT>, lds_r, lds_ok_r));
            hipLaunchKernelGGL((msg_transform_ring_kernel<D, NW, FMT>), dim3(rr.wg_off[T]), dim3(NW * 64), lds_r, st, h, pair_node, rr,
                               (const float*)packed, Hc);
            GGNN_CHECK_HIP(hipGetLastError());
            return GGNN_OK;
        }
    }
    static std::atomic<unsigned long long> lds_ok{0};
    constexpr size_t lds = SPLIT ? PanelSplitCfg<D>::IMG_BYTES : C::IMG_BYTES;
    if (lds > 48 * 1024) GGNN_CHECK_HIP(allow_dynamic_lds(&msg_transform_panel_kernel<D, NW, SPLIT>, lds, lds_ok));
    hipLaunchKernelGGL((msg_transform_panel_kernel<D, NW, SPLIT>), dim3(pr.wg_off[T]), dim3(NW * 64), lds, st, h, pair_node, pr,
                       (const float*)packed, Hc);
    GGNN_CHECK_HIP(hipGetLastError());
    return GGNN_OK;
}

template <int D>
static int launch_transform_panel(const float* h, const float* W, const int* pair_node, const int* row_off, int T, int V,
                                  float* packed, float* Hc, int fmt, hipStream_t st) {
    if (split_matrix_path())
        return (gru_launch_fmt(fmt) == kSplitF16x2 && transform_ring())
                   ? launch_transform_panel_m<D, true, kSplitF16x2>(h, W, pair_node, row_off, T, V, packed, Hc, st)
                   : launch_transform_panel_m<D, true>(h, W, pair_node, row_off, T, V, packed, Hc, st);
    return launch_transform_panel_m<D, false>(h, W, pair_node, row_off, T, V, packed, Hc, st);
}

int transform_panel_dispatch(const float* h, const float* W, const int* pair_node, const int* row_off, int T, int V, int D,
                             float* packed, float* Hc, int fmt, hipStream_t st) {
    if (T > kMaxTypesP) return fail(GGNN_E_UNSUPPORTED, "more than %d edge types", kMaxTypesP);
    switch (D) {
        case 128: return launch_transform_panel<128>(h, W, pair_node, row_off, T, V, packed, Hc, fmt, st);
        case 192: return launch_transform_panel<192>(h, W, pair_node, row_off, T, V, packed, Hc, fmt, st);
        case 256: return launch_transform_panel<256>(h, W, pair_node, row_off, T, V, packed, Hc, fmt, st);
        default: return fail(GGNN_E_UNSUPPORTED, "no panel transform for hidden size %d", D);
    }
}

}  // namespace ggnn
