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
};

// Writes the stage image of the D x D block W[r0 .. r0+D-1][c0 .. c0+D-1] (row stride ldw) to img.
template <int D>
__device__ __forceinline__ void pack_stage_image(const float* __restrict__ W, int r0, int c0, int ldw,
                                                 float* __restrict__ img, int first, int stride) {
    using C = StageCfg<D>;
    for (int i = first; i < C::IMG; i += stride) {
        float v = 0.f;
        if (i < C::MAIN) {
            const int e = i & 3, n = (i >> 2) % C::BN, ck = (i >> 2) / C::BN;     // ck = c*4 + kq
            const int k = 4 * ck + e;                                              // = 16c + 4kq + e
            if (n < D) v = W[(size_t)(r0 + k) * ldw + c0 + n];
        } else if (i < C::MAIN + C::REM) {
            const int j = i - C::MAIN;
            const int n = j % C::BN, qk = j / C::BN;                               // qk = q*4 + kq
            const int k = 16 * C::NC + qk;
            if (n < D) v = W[(size_t)(r0 + k) * ldw + c0 + n];
        }
        img[i] = v;
    }
}

template <int D>
struct Frag {
    f32x4 v[StageCfg<D>::NC > 0 ? StageCfg<D>::NC : 1];
    float r[StageCfg<D>::NR > 0 ? StageCfg<D>::NR : 1];
};

// lane (row = lane&15, kq = lane>>4): float4 A[row][16c + 4kq ..] per chunk + one float per remainder MFMA
template <int D>
__device__ __forceinline__ void load_frag(Frag<D>& f, const float* base, int row, int kq) {
    constexpr int NC = StageCfg<D>::NC, NR = StageCfg<D>::NR;
    const float* p = base + (size_t)row * D + 4 * kq;
#pragma unroll
    for (int c = 0; c < NC; ++c) f.v[c] = ld4(p + 16 * c);
#pragma unroll
    for (int q = 0; q < NR; ++q) f.r[q] = base[(size_t)row * D + 16 * NC + 4 * q + kq];
}

// acc[nt] += A-fragment x stage image.  Per k-chunk c and group of <= 4 tiles: 4 ds_read_b128 feed 16 MFMAs;
// tiles are walked in the inner loop so consecutive MFMAs hit different accumulators (40-cycle dependent
// latency of v_mfma_f32_16x16x4_f32 vs 32-cycle issue).
template <int D>
__device__ __forceinline__ void stage_mma(f32x4 (&acc)[StageCfg<D>::NT], const Frag<D>& a, const float* img, int li, int kq) {
    using C = StageCfg<D>;
    const f32x4* base = reinterpret_cast<const f32x4*>(img) + kq * C::BN + li;
    constexpr int G = 4;
#pragma unroll
    for (int c = 0; c < C::NC; ++c) {
#pragma unroll
        for (int g0 = 0; g0 < C::NT; g0 += G) {
            f32x4 w[G];
#pragma unroll
            for (int j = 0; j < G; ++j)
                if (g0 + j < C::NT) w[j] = base[c * 4 * C::BN + (g0 + j) * 16];
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int j = 0; j < G; ++j)
                    if (g0 + j < C::NT)
                        acc[g0 + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[j][e], a.v[c][e], acc[g0 + j], 0, 0, 0);
        }
    }
#pragma unroll
    for (int q = 0; q < C::NR; ++q) {
        const float* rb = img + C::MAIN + (q * 4 + kq) * C::BN + li;
        float w[C::NT];
#pragma unroll
        for (int nt = 0; nt < C::NT; ++nt) w[nt] = rb[nt * 16];
#pragma unroll
        for (int nt = 0; nt < C::NT; ++nt)
            acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[nt], a.r[q], acc[nt], 0, 0, 0);
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
    const char* s = reinterpret_cast<const char*>(src) + (size_t)wave * PER_WAVE * 1024 + lane * 16;
    char* d = reinterpret_cast<char*>(dst) + (size_t)wave * PER_WAVE * 1024;
#pragma unroll
    for (int i = 0; i < PER_WAVE; ++i)
        __builtin_amdgcn_global_load_lds((glb_void*)(s + i * 1024), (lds_void*)(d + i * 1024), 16, 0, 0);
}

}  // namespace ggnn
