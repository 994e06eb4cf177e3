// Shared building blocks of the register-chained kernels (fused GRU, compacted message transform):
// the D x D weight-stage LDS image, its LDS-DMA, activation fragments and the stage MFMA loop.
//
// Stage image (one D x D weight block, columns padded to BN = 16*ceil(D/16)), k-interleaved:
//     main[c][kq][n][e] = W[16c + 4kq + e][n]      c < D/16, kq < 4, n < BN, e < 4   (float4 per (c,kq,n))
//     rem [q][kq][n]    = W[16*(D/16) + 4q + kq][n] q < (D%16)/4
// One ds_read_b128 of main[c][kq][16*nt + (lane&15)] is the weight operand of the 4 MFMAs (e = 0..3) of
// k-chunk c for output tile nt; a 16-lane ds_read_b128 group covers 16 distinct 16-byte slots
// (conflict-free without padding).  The image is linear, so it is brought into LDS by LDS-DMA
// (global_load_lds_dwordx4) with no staging registers.
#pragma once
#include "ggnn_gemm.hpp"

#ifndef GGNN_VALU_TAIL
#define GGNN_VALU_TAIL 0     // 1: compute the D%16 == 4 remainder columns on the vector ALU (see stage_mma)
#endif

namespace ggnn {

template <int D>
struct StageCfg {
    static constexpr int NT = (D + 15) / 16;
    static constexpr int NC = D / 16;
    static constexpr int NR = (D % 16) / 4;
    static constexpr int BN = NT * 16;
    static constexpr int MAIN = NC * 4 * BN * 4;        // floats in the b128 part
    static constexpr int REM = NR * 4 * BN;             // floats in the remainder part
    static constexpr int IMG_BYTES = ((MAIN + REM) * 4 + 8191) / 8192 * 8192;   // whole KiB per wave of an 8-wave group
    static constexpr int IMG = IMG_BYTES / 4;
    // the last tile holds D % 16 <= 8 valid columns: a second block's tail fits beside it (see pack_stage_image)
    static constexpr bool TAILPACK = (D % 16 != 0) && (D % 16 <= 8) && (D % 4 == 0);
    // ... and a third block's tail when the three fit one tile (D % 16 == 4, e.g. D = 100)
    static constexpr bool TAILPACK3 = TAILPACK && (3 * (D % 16) <= 16);
};

// Writes the stage image of the D x D block W[r0 .. r0+D-1][c0 .. c0+D-1] (row stride ldw) to img.
// c_alt >= 0: the first D % 16 PADDING columns of the last tile (n = D .. D + D%16 - 1) are not zero but the columns
// c_alt .. c_alt + D%16 - 1 of the same rows -- the last, partly filled tile of ANOTHER D x D block that shares these
// rows rides along in this image's padding (StageCfg::TAILPACK; the fused GRU packs the u-gate tail into the r image).
// W2 / c_alt2 >= 0: likewise the NEXT D%16 padding columns (n = D + D%16 ..) come from columns c_alt2 .. of rows
// r0_2 .. of a second matrix W2 (row stride ldw2) -- the fused GRU's candidate-gate tail, StageCfg::TAILPACK3.
template <int D>
__device__ __forceinline__ void pack_stage_image(const float* __restrict__ W, int r0, int c0, int ldw,
                                                 float* __restrict__ img, int first, int stride, int c_alt = -1,
                                                 const float* __restrict__ W2 = nullptr, int r0_2 = 0, int ldw2 = 0, int c_alt2 = -1) {
    using C = StageCfg<D>;
    constexpr int TC = D % 16;
    auto value = [&](int k, int n) -> float {
        if (n < D) return W[(size_t)(r0 + k) * ldw + c0 + n];
        if (c_alt >= 0 && n < D + TC) return W[(size_t)(r0 + k) * ldw + c_alt + (n - D)];
        if (W2 && c_alt2 >= 0 && n >= D + TC && n < D + 2 * TC) return W2[(size_t)(r0_2 + k) * ldw2 + c_alt2 + (n - D - TC)];
        return 0.f;
    };
    for (int i = first; i < C::IMG; i += stride) {
        float v = 0.f;
        if (i < C::MAIN) {
            const int e = i & 3, n = (i >> 2) % C::BN, ck = (i >> 2) / C::BN;     // ck = c*4 + kq
            v = value(4 * ck + e, n);                                              // k = 16c + 4kq + e
        } else if (i < C::MAIN + C::REM) {
            const int j = i - C::MAIN;
            const int n = j % C::BN, qk = j / C::BN;                               // qk = q*4 + kq
            v = value(16 * C::NC + qk, n);
        }
        img[i] = v;
    }
}

// Image ci of the fused GRU's packed weights (ggnn_gru_fused.hip): gates (s = 0..nx) x {r,u}, then candidate (s = 0..nx).
// r images carry the u gate's last (partly filled) tile in their padding columns (the u stages then skip it); the r images of
// the x segments also carry the candidate's last tile (same input rows; NOT the h segment, whose candidate rows multiply r*h).
template <int D>
__device__ __forceinline__ void gru_fwd_image_pack(const float* __restrict__ Wg, const float* __restrict__ Wc, int nx, int ci,
                                                   float* __restrict__ img, int first, int stride) {
    const float* W; int r0, c0, ldw;
    int c_alt = -1, c_alt2 = -1, r0_2 = 0;
    if (ci < 2 * (nx + 1)) {
        W = Wg; r0 = (ci >> 1) * D; c0 = (ci & 1) * D; ldw = 2 * D;
        if (StageCfg<D>::TAILPACK && (ci & 1) == 0) c_alt = D + (D / 16) * 16;
        if (StageCfg<D>::TAILPACK3 && (ci & 1) == 0 && (ci >> 1) < nx) { c_alt2 = (D / 16) * 16; r0_2 = (ci >> 1) * D; }
    } else { W = Wc; r0 = (ci - 2 * (nx + 1)) * D; c0 = 0; ldw = D; }
    pack_stage_image<D>(W, r0, c0, ldw, img, first, stride, c_alt, Wc, r0_2, D, c_alt2);
}

// Image i of the fused GRU backward's packed weights (ggnn_gru_bwd_fused.hip; image[k][n] = B(k, n), out[:, n] = sum_k A[:, k] B(k, n)):
//   0: Wc^T h block  B(k,n) = Wc[nx*D + n][k]    1: Wg_r^T h block  B(k,n) = Wg[nx*D + n][k]    2: Wg_u^T h block  B(k,n) = Wg[nx*D + n][D + k]
//   3 + 3s + {0,1,2}: the same three for x segment s (rows s*D + n)
template <int D>
__device__ __forceinline__ void gru_bwd_image_pack(const float* __restrict__ Wg, const float* __restrict__ Wc, int nx, int i,
                                                   float* __restrict__ img, int first, int stride) {
    using C = StageCfg<D>;
    const int seg = i < 3 ? nx : (i - 3) / 3, which = i < 3 ? i : (i - 3) % 3;
    const float* W = which == 0 ? Wc : Wg;
    const int ldw = which == 0 ? D : 2 * D;
    const int c0 = which == 2 ? D : 0;
    for (int j = first; j < C::IMG; j += stride) {
        float v = 0.f;
        int k = -1, n = 0;
        if (j < C::MAIN) {
            const int e = j & 3; n = (j >> 2) % C::BN; const int ck = (j >> 2) / C::BN;
            k = 4 * ck + e;
        } else if (j < C::MAIN + C::REM) {
            const int jj = j - C::MAIN;
            n = jj % C::BN; k = 16 * C::NC + jj / C::BN;
        }
        if (k >= 0 && n < D) v = W[(size_t)(seg * D + n) * ldw + c0 + k];          // transposed read
        img[j] = v;
    }
}

template <int D>
struct Frag {
    f32x4 v[StageCfg<D>::NC > 0 ? StageCfg<D>::NC : 1];
    float r[StageCfg<D>::NR > 0 ? StageCfg<D>::NR : 1];
};

// Accesses at (wave-uniform base) + (32-bit BYTE offset): this shape selects the scalar-base + 32-bit VGPR offset
// addressing mode of global_load / global_store.  An ELEMENT offset does not -- zext(i) << 2 is not zext(i << 2), so
// the compiler builds a 64-bit address per access with a vector-ALU instruction (v_lshl_add_u64), and vector-ALU
// instructions are paid in matrix-pipe time.  The addressed array must be < 4 GiB.
__device__ __forceinline__ f32x4 ld4_b(const float* base, unsigned byte_off) {
    return *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(base) + byte_off);
}
__device__ __forceinline__ float ld1_b(const float* base, unsigned byte_off) {
    return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off);
}
__device__ __forceinline__ int ldi_b(const int* base, unsigned byte_off) {
    return *reinterpret_cast<const int*>(reinterpret_cast<const char*>(base) + byte_off);
}
__device__ __forceinline__ void st4_b(float* base, unsigned byte_off, f32x4 v) {
    *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(base) + byte_off) = v;
}

// lane (row = lane&15, kq = lane>>4): float4 A[row][16c + 4kq ..] per chunk + one float per remainder MFMA
template <int D>
__device__ __forceinline__ void load_frag(Frag<D>& f, const float* base, int row, int kq) {
    constexpr int NC = StageCfg<D>::NC, NR = StageCfg<D>::NR;
    // 32-bit byte offsets (rows * D * 4 < 2^32): one VGPR offset + scalar base + immediate per access
    const unsigned ob = ((unsigned)row * (unsigned)D + 4u * (unsigned)kq) * 4u;
    __builtin_assume(ob < 0xF0000000u);          // (so that the +64c below can fold into the instruction's immediate)
#pragma unroll
    for (int c = 0; c < NC; ++c) f.v[c] = ld4_b(base, ob + 64u * c);
#pragma unroll
    for (int q = 0; q < NR; ++q) f.r[q] = ld1_b(base, ob - 16u * (unsigned)kq + 4u * (unsigned)(16 * NC + 4 * q) + 4u * (unsigned)kq);
}

// acc[nt] += A-fragment x stage image.  Per k-chunk c and group of <= 4 tiles: 4 ds_read_b128 feed 16 MFMAs;
// tiles are walked in the inner loop so consecutive MFMAs hit different accumulators (40-cycle dependent
// latency of v_mfma_f32_16x16x4_f32 vs 32-cycle issue).
struct NoHook { __device__ __forceinline__ void operator()(int) const {} };

// `hook(gi)` is called at the start of MFMA group gi (compile-time gi after unrolling): the fused GRU uses it
// to spread the LDS-DMA instructions of the NEXT stage over the MFMA stream instead of issuing them in one
// burst (each global_load_lds costs ~60-180 issue cycles during which the matrix pipe would sit idle).
// NTILES < NT: only the first NTILES output tiles are computed (the fused GRU's u-gate stage, whose last tile is
// computed by the r-gate stage: StageCfg::TAILPACK).
// ZERO: the accumulators are not read: the first MFMA of every tile takes the constant 0 as its C operand (saves
// zeroing 4*NT registers per accumulator set with vector-ALU moves).
template <int D, class Hook = NoHook, int NTILES = StageCfg<D>::NT, bool ZERO = false>
__device__ __forceinline__ void stage_mma(f32x4 (&acc)[StageCfg<D>::NT], const Frag<D>& a, const float* img, int li, int kq,
                                          const Hook& hook = Hook()) {
    using C = StageCfg<D>;
    const f32x4* base = reinterpret_cast<const f32x4*>(img) + kq * C::BN + li;
    // VALU tail (D % 16 == 4, e.g. D = 100): the last output tile would hold only 4 valid columns -- 12.5 % of the
    // stage's MFMAs spent on zero padding.  Those 4 columns are instead accumulated on the vector ALU, which is
    // idle while the matrix pipe works: acc[NT-1] holds THIS LANE's partial sums over its own k indices
    // (k = 16c + 4kq + e) for columns 16*NC .. +3, read from the same LDS image (a wave-wide broadcast read per
    // (c, column)); stage_tail_reduce() adds the four kq lanes once per accumulator lifetime.
    // STATUS: compiled out (GGNN_VALU_TAIL = 0).  With hipcc 7.2 the extra live ranges push the fused GRU from
    // 198 VGPRs / no scratch to 256 VGPRs + 84-320 B of scratch per lane, which costs more than the 14 % of
    // MFMAs it saves; kept for a round that hand-allocates the registers.
    constexpr bool VT = (C::NR == 1) && (GGNN_VALU_TAIL != 0);
    constexpr int NTM = VT ? C::NT - 1 : NTILES;             // tiles on the matrix pipe
    // Explicit one-group-ahead software pipeline with bounded register use: the weight operands of group
    // gi+1 (<= 4 tiles, 16 VGPRs) are read while the <= 16 MFMAs of group gi issue; a scheduling barrier per
    // group stops the compiler from hoisting further reads (which drove the kernel into scratch spills).
#ifndef GGNN_STAGE_G
#define GGNN_STAGE_G 4
#endif
    constexpr int G = GGNN_STAGE_G;
    constexpr int GPC = (NTM + G - 1) / G;                   // groups per k-chunk
    constexpr int NG = C::NC * GPC;
    const f32x4* tbase = reinterpret_cast<const f32x4*>(img) + kq * C::BN + 16 * C::NC;   // tail columns, no lane offset
    f32x4 w[2][G];
    float wr[C::NR > 0 ? C::NR : 1][C::NT];
    static_assert(NG > 0 || C::NR == 0, "remainder weights are fetched inside the main loop");
    if constexpr (NG > 0) {
#pragma unroll
        for (int j = 0; j < G; ++j)
            if (j < NTM) w[0][j] = base[j * 16];
    }
#pragma unroll
    for (int gi = 0; gi < NG; ++gi) {
        const int c = gi / GPC, g0 = (gi % GPC) * G;
        hook(gi);
        if (gi + 1 < NG) {
            const int cn = (gi + 1) / GPC, gn = ((gi + 1) % GPC) * G;
#pragma unroll
            for (int j = 0; j < G; ++j)
                if (gn + j < NTM) w[(gi + 1) & 1][j] = base[cn * 4 * C::BN + (gn + j) * 16];
            __builtin_amdgcn_sched_barrier(0);               // reads of group gi+1 are issued BEFORE group gi's MFMAs
        } else {
            // last group: the weights of the remainder MFMAs (k = 16*NC ..) are read under its MFMAs, not after them
#pragma unroll
            for (int q = 0; q < C::NR; ++q)
#pragma unroll
                for (int nt = 0; nt < NTM; ++nt) wr[q][nt] = img[C::MAIN + (q * 4 + kq) * C::BN + li + nt * 16];
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int j = 0; j < G; ++j)
                if (g0 + j < NTM) {
                    const f32x4 cin = (ZERO && c == 0 && e == 0) ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[g0 + j];
                    acc[g0 + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[gi & 1][j][e], a.v[c][e], cin, 0, 0, 0);
                }
        __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (VT) {
        // The tail runs AFTER this wave's MFMA burst (the weight registers are dead by then, so no extra register
        // pressure): the two waves of a SIMD issue their bursts back to back (older first), so this VALU work
        // overlaps the partner wave's MFMAs.
        f32x4& t = acc[C::NT - 1];
#pragma unroll
        for (int c = 0; c < C::NC; ++c) {
            const f32x4 av = a.v[c];
            const f32x4 t0 = tbase[c * 4 * C::BN + 0], t1 = tbase[c * 4 * C::BN + 1];
            const f32x4 t2 = tbase[c * 4 * C::BN + 2], t3 = tbase[c * 4 * C::BN + 3];
            t.x += av.x * t0.x + av.y * t0.y + av.z * t0.z + av.w * t0.w;
            t.y += av.x * t1.x + av.y * t1.y + av.z * t1.z + av.w * t1.w;
            t.z += av.x * t2.x + av.y * t2.y + av.z * t2.z + av.w * t2.w;
            t.w += av.x * t3.x + av.y * t3.y + av.z * t3.z + av.w * t3.w;
            __builtin_amdgcn_sched_barrier(0);               // bound the read-ahead (16 VGPRs per chunk)
        }
    }
#pragma unroll
    for (int q = 0; q < C::NR; ++q) {
#pragma unroll
        for (int nt = 0; nt < NTM; ++nt)
            acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[q][nt], a.r[q], acc[nt], 0, 0, 0);
        if constexpr (VT) {                                  // remainder k = 16*NC + kq against the 4 tail columns
            const f32x4 tr = *reinterpret_cast<const f32x4*>(img + C::MAIN + (q * 4 + kq) * C::BN + 16 * C::NC);
            acc[C::NT - 1] += a.r[q] * tr;
        }
    }
}

// ONE output tile (`tile`, wave-uniform, run time) of the same product: acc (+)= A-fragment x columns 16*tile .. of the
// stage image -- the cooperative tail pass of the fused GRU, where each wave of a workgroup owns one tile of the row block.
// Same k order per output element as stage_mma.
template <int D, bool ZERO>
__device__ __forceinline__ void stage_mma_one(f32x4& acc, const Frag<D>& a, const float* img, int li, int kq, int tile) {
    using C = StageCfg<D>;
    const f32x4* base = reinterpret_cast<const f32x4*>(img) + kq * C::BN + li + tile * 16;
    f32x4 cin = acc;
    if constexpr (ZERO) cin = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 w = base[0];
#pragma unroll
    for (int c = 0; c < C::NC; ++c) {
        const f32x4 wn = base[(c + 1 < C::NC ? c + 1 : c) * 4 * C::BN];       // next chunk's weights under this chunk's MFMAs
#pragma unroll
        for (int e = 0; e < 4; ++e) cin = __builtin_amdgcn_mfma_f32_16x16x4f32(w[e], a.v[c][e], cin, 0, 0, 0);
        w = wn;
    }
#pragma unroll
    for (int q = 0; q < C::NR; ++q) {
        const float wr = img[C::MAIN + (q * 4 + kq) * C::BN + li + tile * 16];
        cin = __builtin_amdgcn_mfma_f32_16x16x4f32(wr, a.r[q], cin, 0, 0, 0);
    }
    acc = cin;
}

// The same tile product with the weights of the column tile held in REGISTERS, fetched straight from the stage image
// in global memory (25 floats per lane at D = 100): the cooperative tail pass touches 1/NT of an image per wave, so
// staging the whole 48 KiB image through LDS (and a workgroup barrier per stage) buys nothing there -- each wave
// requests its slice two stages ahead and no stage of that pass waits on a DMA.
template <int D>
struct TileWeights {
    f32x4 v[StageCfg<D>::NC > 0 ? StageCfg<D>::NC : 1];
    float r[StageCfg<D>::NR > 0 ? StageCfg<D>::NR : 1];
};

template <int D>
__device__ __forceinline__ void load_tile_weights(TileWeights<D>& w, const float* __restrict__ gimg, int li, int kq, int tile) {
    using C = StageCfg<D>;
    const f32x4* base = reinterpret_cast<const f32x4*>(gimg) + kq * C::BN + li + tile * 16;
#pragma unroll
    for (int c = 0; c < C::NC; ++c) w.v[c] = base[c * 4 * C::BN];
#pragma unroll
    for (int q = 0; q < C::NR; ++q) w.r[q] = gimg[C::MAIN + (q * 4 + kq) * C::BN + li + tile * 16];
}

template <int D, bool ZERO>
__device__ __forceinline__ void tile_mma_regs(f32x4& acc, const Frag<D>& a, const TileWeights<D>& w) {
    using C = StageCfg<D>;
    f32x4 cin = acc;
    if constexpr (ZERO) cin = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < C::NC; ++c) {
#pragma unroll
        for (int e = 0; e < 4; ++e) cin = __builtin_amdgcn_mfma_f32_16x16x4f32(w.v[c][e], a.v[c][e], cin, 0, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < C::NR; ++q) cin = __builtin_amdgcn_mfma_f32_16x16x4f32(w.r[q], a.r[q], cin, 0, 0, 0);
    acc = cin;
}

// Completes a VALU-tail accumulator (see stage_mma): after the LAST stage that accumulates into `acc`, the partial
// sums of the four kq lanes of each row are added, so every lane of the row holds columns 16*NC..+3 -- exactly
// what the epilogues expect from lane kq == 0 of the last tile.  No-op for hidden sizes without a 4-column remainder.
template <int D>
__device__ __forceinline__ void stage_tail_reduce(f32x4 (&acc)[StageCfg<D>::NT]) {
    if constexpr (StageCfg<D>::NR == 1 && GGNN_VALU_TAIL != 0) {
        f32x4& t = acc[StageCfg<D>::NT - 1];
#pragma unroll
        for (int off = 16; off <= 32; off <<= 1) {
            t.x += __shfl_xor(t.x, off); t.y += __shfl_xor(t.y, off);
            t.z += __shfl_xor(t.z, off); t.w += __shfl_xor(t.w, off);
        }
    }
}

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

// LDS-DMA of one stage image (IMG_BYTES at src) into LDS at dst by an NW-wave workgroup: wave w moves the
// KiB blocks [w*PER_WAVE, (w+1)*PER_WAVE).  `wave` must be wave-uniform (readfirstlane).
template <int D, int NW>
__device__ __forceinline__ void dma_stage_image(const float* src, float* dst, int wave, int lane) {
    using C = StageCfg<D>;
    constexpr int PER_WAVE = C::IMG_BYTES / (NW * 1024);
    static_assert(C::IMG_BYTES % (NW * 1024) == 0, "image must split into whole KiB per wave");
    // scalar (wave-uniform) base + one 32-bit per-lane offset: the DMA addresses of all stages share a single
    // VGPR (64-bit per-lane address pairs per DMA instruction cost 72 VGPRs in the fused GRU and spilled it)
    char* d = reinterpret_cast<char*>(dst) + (size_t)wave * PER_WAVE * 1024;
    const unsigned voff = (unsigned)lane * 16u;
#pragma unroll
    for (int i0 = 0; i0 < PER_WAVE; i0 += 4) {
        // pin the 4-KiB group base into SGPRs so the access selects the saddr + 32-bit voffset form; the
        // instruction's immediate offset (applied to the global AND the LDS address) walks the KiB blocks
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

// One KiB piece (piece < IMG_BYTES / (NW*1024)) of the same transfer, for callers that interleave the DMA
// instructions with other work.  Pieces 4j .. 4j+3 share one pinned scalar base + immediate offsets.
template <int D, int NW>
__device__ __forceinline__ void dma_stage_piece(const float* src, float* dst, int wave, int lane, int piece) {
    using C = StageCfg<D>;
    constexpr int PER_WAVE = C::IMG_BYTES / (NW * 1024);
    const int i0 = piece & ~3;
    const unsigned long long sb = reinterpret_cast<unsigned long long>(src) + (unsigned long long)wave * PER_WAVE * 1024 + (unsigned long long)i0 * 1024;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)sb);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(sb >> 32));
    const char* s = reinterpret_cast<const char*>(((unsigned long long)hi << 32) | lo);
    lds_void* dl = (lds_void*)(reinterpret_cast<char*>(dst) + (size_t)wave * PER_WAVE * 1024 + i0 * 1024);
    const unsigned voff = (unsigned)lane * 16u;
    switch (piece & 3) {
        case 0: __builtin_amdgcn_global_load_lds((glb_void*)(s + voff), dl, 16, 0, 0); break;
        case 1: __builtin_amdgcn_global_load_lds((glb_void*)(s + voff), dl, 16, 1024, 0); break;
        case 2: __builtin_amdgcn_global_load_lds((glb_void*)(s + voff), dl, 16, 2048, 0); break;
        default: __builtin_amdgcn_global_load_lds((glb_void*)(s + voff), dl, 16, 3072, 0); break;
    }
}

}  // namespace ggnn
