// FP32 products on the BF16 matrix pipe: the "3-way split" form of the stage product of ggnn_stage.hpp.
//
// gfx950 multiplies f32 operands on the matrix cores at the f32 VECTOR rate (v_mfma_f32_16x16x4_f32: 1024 MACs per 32
// clocks per SIMD, 157 TF) -- 1/16 of what the same cores do with bf16 operands (v_mfma_f32_16x16x32_bf16: 8192 MACs per
// 16 clocks, 2.5 PF), and there is no TF32-like middle.  An f32 value is EXACTLY the sum of three bf16 values
//     a = a_hi + a_mid + a_lo        (8 + 8 + 8 significand bits; bf16 has the f32 exponent range)
// so the f32 product a*b is the sum of nine bf16 x bf16 products, each of which is exact in f32 (16-bit significands).
// Dropping the three smallest (a_mid*b_lo, a_lo*b_mid, a_lo*b_lo: < 2^-23 |a b| together) leaves SIX products
//     a_hi b_hi + (a_hi b_mid + a_mid b_hi) + (a_mid b_mid + a_hi b_lo + a_lo b_hi)
// accumulated in f32 by the MFMA: 6 MFMAs of k = 32 do the work of 8 MFMAs of k = 4 x 32 clocks -- 96 clocks instead of
// 256 per 16 x 16 x 32 block -- with an error bound no worse than the f32 FMA chain's (one rounding per k there, one per
// 32-k dot product here; tests/test_gpu_split_precision.py measures both against an f64 product).
//
// Layout.  v_mfma_f32_16x16x32_bf16: lane (i = lane & 15, g = lane >> 4) holds, for BOTH operands, the 8 k values
// 8g .. 8g+7 of its row / column; the output mapping is that of the f32 form (lane holds 4 consecutive output columns of
// one row when the WEIGHT fragment is the first operand).  k order inside a 32-chunk is free as long as both operands
// agree, so slot j of lane group g in chunk c2 is
//     k = 32 c2 + 16 (j >> 2) + 4 g + (j & 3)
// i.e. the float4 pair Frag::v[2 c2], Frag::v[2 c2 + 1] of the f32 activation fragment: fragments are loaded, gathered and
// chained (output tile nt == activation chunk nt) exactly as in the f32 kernels, and split in registers.  The D % 16
// remainder k indices (4 at D = 100) stay on the f32 MFMA.
//
// Stage image (one D x D block; BN = 16 ceil(D/16) columns), stored as TWO column halves that are each a complete image of
// their own (round 4: the two-workgroups-per-CU fused GRU streams a stage through its LDS ring half by half; every other
// kernel brings the whole image in at once and never notices -- the (unit -> address) map below is resolved at compile time):
//     half A = output tiles [0, TA), half B = tiles [TA, NT);  inside a half of BNH columns:
//     plane p in {hi, mid, lo}:  [c2][g][n][8 x bf16]   one ds_read_b128 = the weight operand of (c2, tile of n)
//     rem (f32): [q][g][n] = W[16 NC + 4q + g][n]       as in the f32 image
// half A is padded to whole KiB (the half-stage DMA moves KiB pieces), the image to 8 KiB.
#pragma once
#include "ggnn_stage.hpp"
#include <type_traits>

namespace ggnn {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

// Operand formats of a split product.
//   kSplitBf16x3 (3): the exact three-piece bf16 split above, six products (every kernel but the fused GRU forward).
//   kSplitF16x2 (2), round 4: TWO f16 pieces per operand and THREE products on v_mfma_f32_16x16x32_f16 -- the bf16 instruction's
//     rate and lane layout, so half the MFMAs, two operand planes instead of three (a D = 100 stage image: 48 instead of 72 KiB)
//     and 5 instead of 11 vector instructions per split value pair:
//         a = a_hi + a_lo + e,   a_hi = RN_f16(a),  a_lo = RN_f16(a - a_hi),  |e| <= 2^-22 |a|       (f16: 11 significand bits)
//         a b ~ a_hi b_hi + a_hi b_lo + a_lo b_hi                    (a_lo b_lo < 2^-22 |a b| is dropped)
//     NOT exact -- each operand is rounded to 22 bits -- but f32-faithful where it is used: against f64 the fused GRU's error is
//     BELOW the six-product form's and the f32 MFMA's (MI355X, tools/split_probe.py: h' rms 6.3e-8 / 7.3e-8 / 8.4e-8; numpy:
//     oracle.split3_f16_matmul, tests/test_oracle.py) -- the f32 chain rounds its running sum K times, this form once per 32 terms.
//     Range and scaling: f16 reaches 65504 and its normals stop at 6.1e-5.  WEIGHTS are packed x 2^8 (exact; free: once per weight
//     version), which lifts the lo pieces of everything above 2^-11 out of the subnormal range (the lo piece of an unscaled 0.1
//     sits at 5e-5 and would keep 9 bits: 2 x the error) and bounds |w| to 255.875; ACTIVATIONS are not scaled (numpy: scaling them
//     too buys 1 %) and must stay within +-65504.  Neither bound is enforced in the kernel: it is the caller's precondition, and a
//     value beyond it turns into Inf / NaN pieces, i.e. non-finite results (GGNN_F16_CLAMP below).  The MFMA keeps f16 subnormal inputs
//     (tools/f16_mfma_denorm_probe.hip).  Accumulators hold 2^8 x the sums: the remainder weights (f32 MFMA) are packed x 2^8 and
//     the consumer's epilogue scales by acc_scale, folded into a constant it multiplies by anyway.
//     Used by the fused GRU FORWARD (kernel, both packers) when the CALL asks for it (GruFusedArgs::fmt == GGNN_GRU_FMT_F16X2, the
//     `gru_fmt` argument of the C entry points, round 5): the operand range above is the caller's precondition -- the host layer
//     (formats.py) proves it from max|h0| and the weights' maxima or selects the exact format.
constexpr int kSplitBf16x3 = 3, kSplitF16x2 = 2;
template <int FMT> struct SplitFmt;
template <> struct SplitFmt<kSplitBf16x3> {
    static constexpr int NP = 3;                                   // operand planes: hi, mid, lo
    static constexpr float w_scale = 1.0f, rem_scale = 1.0f, acc_scale = 1.0f;
};
template <> struct SplitFmt<kSplitF16x2> {
    static constexpr int NP = 2;                                   // operand planes: hi, lo
    static constexpr float w_scale = 256.0f;                       // weights as packed
    static constexpr float rem_scale = 256.0f;                     // remainder weights (f32 MFMA; their activations are unscaled)
    static constexpr float acc_scale = 1.0f / 256.0f;              // accumulator -> sum
};
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
// 1: the f16 split of an activation pair in 5 hand-placed instructions (2 v_med3_f32, v_cvt_pk_f16_f32, v_fma_mixlo/hi_f16);
// 0: left to hipcc (10: it converts each value twice)
#ifndef GGNN_F16_SPLIT_ASM
#define GGNN_F16_SPLIT_ASM 1
#endif
// 0 (round 5): NO clamp of the operands before the f16 split.  The format's operand range is the caller's PRECONDITION (formats.py
// proves it per launch); a value beyond it becomes Inf in its hi piece and -Inf / NaN in its lo piece, so a precondition-violating call
// yields non-finite results instead of plausible wrong ones, and a NaN operand stays NaN (round 4 clamped with v_med3_f32 -- 2 more
// vector instructions per value pair -- which returned a finite, saturated operand for NaN / Inf: it could mask a diverged run).
// 1: the round-4 clamp (experiments; profiles/r05_experiments/gru_valu_variants.txt: 66.8 vs 66.0 us).
#ifndef GGNN_F16_CLAMP
#define GGNN_F16_CLAMP 0
#endif

// Process default of the HOST policy (GGNN_GRU_FMT; 2 = f16x2 where proven safe, 3 = always bf16x3).  Kernels take the format per launch.
int gru_fwd_fmt();
// the format a launch runs in: anything but GGNN_GRU_FMT_F16X2 is the exact bf16x3 split
inline int gru_launch_fmt(int fmt) { return fmt == kSplitF16x2 ? kSplitF16x2 : kSplitBf16x3; }

// Process-wide choice of the matrix path of the fused kernels (read once): GGNN_MATRIX=f32 selects the f32 MFMA forms.
bool split_matrix_path();

template <int D, int FMT = kSplitBf16x3>
struct SplitCfg {
    using S = StageCfg<D>;
    static constexpr bool OK = (S::NC % 2 == 0) && S::NC > 0;
    static constexpr int NC2 = S::NC / 2;
    static constexpr int TA = (S::NT + 1) / 2;                       // tiles in half A (D = 100: 4 of 7)
    static constexpr int NP = SplitFmt<FMT>::NP;                     // operand planes
    // geometry of a half holding NTH tiles
    static constexpr int plane_bytes(int nth) { return NC2 * 4 * nth * 16 * 16; }
    static constexpr int main_bytes(int nth) { return NP * plane_bytes(nth); }
    static constexpr int rem_bytes(int nth) { return S::NR * 4 * nth * 16 * 4; }
    static constexpr int half_bytes(int nth) { return (main_bytes(nth) + rem_bytes(nth) + 1023) / 1024 * 1024; }
    static constexpr int HA_BYTES = half_bytes(TA);                  // half A, whole KiB (D = 100: 37 KiB)
    static constexpr int HB_BYTES = half_bytes(S::NT - TA);          // half B, whole KiB (28 KiB)
    static constexpr int HA = HA_BYTES / 4;                          // float offset of half B inside the image
    static constexpr int IMG_BYTES = (HA_BYTES + HB_BYTES + 8191) / 8192 * 8192;
    static constexpr int IMG = IMG_BYTES / 4;
    // (unit-independent parts of the address of tile nt's operand: which half, its column count, the tile's index inside it)
    static constexpr int half_of(int nt) { return nt < TA ? 0 : 1; }
    static constexpr int nth_of(int nt) { return nt < TA ? TA : S::NT - TA; }
    static constexpr int tile_in_half(int nt) { return nt < TA ? nt : nt - TA; }
};

// image geometry of a fused kernel: the f32 stage image or the split one
template <int D, bool SPLIT, int FMT = kSplitBf16x3>
struct ImgCfg {
    static constexpr int IMG_BYTES = SPLIT ? SplitCfg<D, FMT>::IMG_BYTES : StageCfg<D>::IMG_BYTES;
    static constexpr int IMG = IMG_BYTES / 4;
};

__device__ __forceinline__ float trunc_bf16_f(float x) { return __uint_as_float(__float_as_uint(x) & 0xffff0000u); }

// piece p (0 hi, 1 mid, 2 lo) of x as bf16 bits.  Truncation: every piece takes the next 8 significand bits, the three
// together all 24 -- the split is exact and each residual subtraction is exact.
// FMT = kSplitF16x2: piece p (0 hi, 1 lo) of the WEIGHT x as f16 bits: x 2^8, round to nearest; the residual subtraction is exact.
template <int FMT = kSplitBf16x3>
__device__ __forceinline__ unsigned split_piece_bits(float x, int p) {
    if constexpr (FMT == kSplitF16x2) {
#if GGNN_F16_CLAMP
        const float xs = fminf(fmaxf(x * SplitFmt<FMT>::w_scale, -65504.0f), 65504.0f);
#else
        const float xs = x * SplitFmt<FMT>::w_scale;      // (|x| > 255.875: Inf / NaN pieces -- loud, see GGNN_F16_CLAMP)
#endif
        const _Float16 h16 = (_Float16)xs;
        if (p == 0) return (unsigned)__builtin_bit_cast(unsigned short, h16);
        return (unsigned)__builtin_bit_cast(unsigned short, (_Float16)(xs - (float)h16));
    }
    const float hi = trunc_bf16_f(x);
    if (p == 0) return __float_as_uint(hi) >> 16;
    const float r1 = x - hi, mid = trunc_bf16_f(r1);
    if (p == 1) return __float_as_uint(mid) >> 16;
    return __float_as_uint(r1 - mid) >> 16;
}

// The (k, n) -> value rule of a stage image (pack_stage_image's, see there): the block itself, plus the partly filled last
// tiles of up to two other blocks riding in the padding columns.
template <int D>
struct StageValue {
    const float* W; int r0, c0, ldw, c_alt;
    const float* W2; int r0_2, ldw2, c_alt2;
    __device__ __forceinline__ float operator()(int k, int n) const {
        constexpr int TC = D % 16;
        if (n < D) return W[(size_t)(r0 + k) * ldw + c0 + n];
        if (c_alt >= 0 && n < D + TC) return W[(size_t)(r0 + k) * ldw + c_alt + (n - D)];
        if (W2 && c_alt2 >= 0 && n >= D + TC && n < D + 2 * TC) return W2[(size_t)(r0_2 + k) * ldw2 + c_alt2 + (n - D - TC)];
        return 0.f;
    }
};

// transposed block (the backward kernels' images): B(k, n) = W[(r0 + n) * ldw + c0 + k]
template <int D>
struct StageValueT {
    const float* W; int r0, c0, ldw;
    __device__ __forceinline__ float operator()(int k, int n) const { return n < D ? W[(size_t)(r0 + n) * ldw + c0 + k] : 0.f; }
};

template <int D, int FMT = kSplitBf16x3, class Value>
__device__ __forceinline__ void pack_split_image(const Value& value, float* __restrict__ img, int first, int stride) {
    using C = SplitCfg<D, FMT>;
    using S = StageCfg<D>;
    for (int i = first; i < C::IMG; i += stride) {
        unsigned out = 0u;
        const int half = i < C::HA ? 0 : 1;
        const int nth = half ? S::NT - C::TA : C::TA, bnh = nth * 16, n0 = half ? C::TA * 16 : 0;
        const int w_ = i - (half ? C::HA : 0);                       // word inside the half
        const int PW = C::plane_bytes(nth) / 4, MW = C::main_bytes(nth) / 4, RW = C::rem_bytes(nth) / 4;
        if (w_ < MW) {
            const int plane = w_ / PW, w = w_ % PW;
            const int slot = w >> 2, pr = w & 3;                 // 16-byte slot (c2, g, n); bf16 pair (j = 2 pr, 2 pr + 1)
            const int n = n0 + slot % bnh, cg = slot / bnh, c2 = cg >> 2, g = cg & 3;
            const int j0 = 2 * pr;
            const int k0 = 32 * c2 + 16 * (j0 >> 2) + 4 * g + (j0 & 3);
            out = split_piece_bits<FMT>(value(k0, n), plane) | (split_piece_bits<FMT>(value(k0 + 1, n), plane) << 16);
        } else if (w_ < MW + RW) {
            const int j = w_ - MW;
            out = __float_as_uint(value(16 * S::NC + j / bnh, n0 + j % bnh) * SplitFmt<FMT>::rem_scale);       // j / bnh = q*4 + g
        }
        img[i] = __uint_as_float(out);
    }
}

// Image ci of the fused GRU's packed weights in split form: the same blocks and tail-riding columns as gru_fwd_image_pack.
template <int D, int FMT = kSplitBf16x3>
__device__ __forceinline__ void gru_fwd_image_pack_split(const float* __restrict__ Wg, const float* __restrict__ Wc, int nx, int ci,
                                                         float* __restrict__ img, int first, int stride) {
    StageValue<D> v{nullptr, 0, 0, 0, -1, Wc, 0, D, -1};
    if (ci < 2 * (nx + 1)) {
        v.W = Wg; v.r0 = (ci >> 1) * D; v.c0 = (ci & 1) * D; v.ldw = 2 * D;
        if (StageCfg<D>::TAILPACK && (ci & 1) == 0) v.c_alt = D + (D / 16) * 16;
        if (StageCfg<D>::TAILPACK3 && (ci & 1) == 0 && (ci >> 1) < nx) { v.c_alt2 = (D / 16) * 16; v.r0_2 = (ci >> 1) * D; }
    } else { v.W = Wc; v.r0 = (ci - 2 * (nx + 1)) * D; v.c0 = 0; v.ldw = D; }
    pack_split_image<D, FMT>(v, img, first, stride);
}

// Image i of the fused GRU backward's packed weights in split form (blocks as gru_bwd_image_pack: transposed reads)
template <int D>
__device__ __forceinline__ void gru_bwd_image_pack_split(const float* __restrict__ Wg, const float* __restrict__ Wc, int nx, int i,
                                                         float* __restrict__ img, int first, int stride) {
    const int seg = i < 3 ? nx : (i - 3) / 3, which = i < 3 ? i : (i - 3) % 3;
    pack_split_image<D>(StageValueT<D>{which == 0 ? Wc : Wg, seg * D, which == 2 ? D : 0, which == 0 ? D : 2 * D}, img, first, stride);
}

// the three bf16 planes of an activation fragment's 32-chunks (the remainder floats stay in the Frag); kSplitF16x2: hi = the hi
// pieces, mid = the lo pieces, lo unused
template <int D>
struct SFrag {
    u32x4 hi[SplitCfg<D>::NC2 > 0 ? SplitCfg<D>::NC2 : 1];
    u32x4 mid[SplitCfg<D>::NC2 > 0 ? SplitCfg<D>::NC2 : 1];
    u32x4 lo[SplitCfg<D>::NC2 > 0 ? SplitCfg<D>::NC2 : 1];
};

// two floats -> one register of each plane (element 0 in the low half): 3 v_perm + 4 v_and + 4 v_sub
// kSplitF16x2: h = the f16 hi pieces, m = the f16 lo pieces of the two values (l unused)
template <int FMT = kSplitBf16x3>
__device__ __forceinline__ void split_pair(float a0, float a1, unsigned& h, unsigned& m, unsigned& l) {
    if constexpr (FMT == kSplitF16x2) {
#if GGNN_F16_CLAMP
        const float t0 = __builtin_amdgcn_fmed3f(a0, -65504.0f, 65504.0f), t1 = __builtin_amdgcn_fmed3f(a1, -65504.0f, 65504.0f);
#else
        const float t0 = a0, t1 = a1;
#endif
#if GGNN_F16_SPLIT_ASM
        // h = (RN_f16(t0), RN_f16(t1));  m = (RN_f16(t0 - h.lo), RN_f16(t1 - h.hi)): the mixed-precision FMA takes the f16 halves of h
        // as its third source (op_sel_hi marks a source as f16, op_sel picks the half), computes t - h exactly in f32 and rounds once
        asm("v_cvt_pk_f16_f32 %0, %2, %3\n\t"
            "v_fma_mixlo_f16 %1, %2, 1.0, -%0 op_sel_hi:[0,0,1]\n\t"
            "v_fma_mixhi_f16 %1, %3, 1.0, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
            : "=&v"(h), "=&v"(m) : "v"(t0), "v"(t1));
#else
        const f16x2 hv = {(_Float16)t0, (_Float16)t1};
        h = __builtin_bit_cast(unsigned, hv);
        const f16x2 lv = {(_Float16)(t0 - (float)hv.x), (_Float16)(t1 - (float)hv.y)};
        m = __builtin_bit_cast(unsigned, lv);
#endif
        l = 0u;
        return;
    }
    const unsigned u0 = __float_as_uint(a0), u1 = __float_as_uint(a1);
    h = __builtin_amdgcn_perm(u1, u0, 0x07060302u);
    const float r0 = a0 - __uint_as_float(u0 & 0xffff0000u), r1 = a1 - __uint_as_float(u1 & 0xffff0000u);
    const unsigned v0 = __float_as_uint(r0), v1 = __float_as_uint(r1);
    m = __builtin_amdgcn_perm(v1, v0, 0x07060302u);
    const float s0 = r0 - __uint_as_float(v0 & 0xffff0000u), s1 = r1 - __uint_as_float(v1 & 0xffff0000u);
    l = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302u);
}

template <int D, int FMT = kSplitBf16x3>
__device__ __forceinline__ void split_frag(SFrag<D>& s, const Frag<D>& f) {
#pragma unroll
    for (int c2 = 0; c2 < SplitCfg<D>::NC2; ++c2) {
        const f32x4 a = f.v[2 * c2], b = f.v[2 * c2 + 1];
        unsigned h[4], m[4], l[4];
        split_pair<FMT>(a.x, a.y, h[0], m[0], l[0]);
        split_pair<FMT>(a.z, a.w, h[1], m[1], l[1]);
        split_pair<FMT>(b.x, b.y, h[2], m[2], l[2]);
        split_pair<FMT>(b.z, b.w, h[3], m[3], l[3]);
        s.hi[c2] = u32x4{h[0], h[1], h[2], h[3]};
        s.mid[c2] = u32x4{m[0], m[1], m[2], m[3]};
        s.lo[c2] = u32x4{l[0], l[1], l[2], l[3]};
    }
}

__device__ __forceinline__ f32x4 mfma_f16(u32x4 w, u32x4 a, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, w), __builtin_bit_cast(f16x8, a), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma_bf16(u32x4 w, u32x4 a, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, a), c, 0, 0, 0);
}

// the six products of one (32-chunk, tile), smallest terms first; FIRST: the accumulator is opened with C = 0
// kSplitF16x2: the three products (wm, am: the lo pieces; wl, al unused)
template <bool FIRST, int FMT = kSplitBf16x3>
__device__ __forceinline__ f32x4 split_products(f32x4 acc, u32x4 wh, u32x4 wm, u32x4 wl, u32x4 ah, u32x4 am, u32x4 al) {
    f32x4 c = FIRST ? f32x4{0.f, 0.f, 0.f, 0.f} : acc;
    if constexpr (FMT == kSplitF16x2) {
        c = mfma_f16(wm, ah, c);
        c = mfma_f16(wh, am, c);
        c = mfma_f16(wh, ah, c);
        return c;
    }
    c = mfma_bf16(wl, ah, c);
    c = mfma_bf16(wm, am, c);
    c = mfma_bf16(wm, ah, c);
    c = mfma_bf16(wh, al, c);
    c = mfma_bf16(wh, am, c);
    c = mfma_bf16(wh, ah, c);
    return c;
}

// acc[nt] (+)= A-fragment x split stage image: the counterpart of stage_mma (same tiles, same remainder handling).
// One (32-chunk, tile) unit = 3 ds_read_b128 (the lo / mid / hi planes of the weight fragment) feeding 6 MFMAs.  The planes are
// consumed in the order lo (1 product), mid (2), hi (3) and each plane's registers are refilled with the NEXT unit's fragment as
// soon as its last product has issued: 12 weight registers in flight instead of 24 for a double buffer, the lo and mid fragments
// of the next unit a full unit ahead, the hi fragment three MFMAs ahead (the partner wave's MFMAs cover what that leaves).
// T0: first tile (the half-stage kernels multiply a stage in two calls, tiles [0, TA) from half A and [TA, NTILES) from half B,
// each half at `img` / `img_b`; everyone else passes the whole image: img_b = img + HA).  Per accumulator the chain is the same
// in both forms: chunks in order, then the remainder.
#ifndef GGNN_SPLIT_PF
#define GGNN_SPLIT_PF 0
#endif
#ifndef GGNN_SPLIT_WH2
#define GGNN_SPLIT_WH2 0     // 1: the hi plane of the NEXT unit is fetched at the start of the current one into a second register
#endif                       //    set (9 MFMAs ahead instead of 3; +4 registers) -- experiment, see DESIGN.md K3
// REMAT: the lane parts of the LDS addresses are recomputed in every call (the lane coordinates pass through an empty asm, so the
// compiler cannot keep the ~10 per-slot address registers of a multi-stage kernel alive across its stages -- a kernel at the
// 256-register limit spills exactly those and reloads them in front of the remainder MFMAs).
template <int D, int NTILES = StageCfg<D>::NT, bool ZERO = false, int T0 = 0, bool REMAT = false, int FMT = kSplitBf16x3>
__device__ __forceinline__ void stage_mma_split_at(f32x4 (&acc)[StageCfg<D>::NT], const SFrag<D>& a, const Frag<D>& af,
                                                   const float* img, const float* img_b, int li, int kq) {
    using S = StageCfg<D>;
    using C = SplitCfg<D, FMT>;
    if constexpr (REMAT) asm volatile("" : "+v"(li), "+v"(kq));
    constexpr int NTW = NTILES - T0;                                  // tiles walked
    constexpr int NU = C::NC2 * NTW;                                  // units, chunk-major
    if constexpr (NU > 0) {
        // lane part of the address, per half (the halves differ in their column count)
        const u32x4* base_a = reinterpret_cast<const u32x4*>(img) + kq * (C::TA * 16) + li;
        const u32x4* base_b = reinterpret_cast<const u32x4*>(img_b) + kq * ((S::NT - C::TA) * 16) + li;
        auto slot = [&](int u, int p) {
            const int nt = T0 + u % NTW, c2 = u / NTW;
            const int nth = C::nth_of(nt);
            const u32x4* b = C::half_of(nt) ? base_b : base_a;
            return b[p * (C::plane_bytes(nth) / 16) + c2 * 4 * nth * 16 + C::tile_in_half(nt) * 16];
        };
      if constexpr (FMT == kSplitF16x2) {
        // three products per unit, smallest first; both planes of the NEXT unit are fetched at the start of the current one (a whole
        // unit = 3 MFMAs ahead of their first use; 16 weight registers in flight, and the lo plane of the activation is gone)
        u32x4 wh = slot(0, 0), wm = slot(0, 1), nh = wh, nm = wm;
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int c2 = u / NTW, nt = T0 + u % NTW;
            const bool more = u + 1 < NU;
            f32x4 c = (ZERO && c2 == 0) ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[nt];
            if (more) { nm = slot(u + 1, 1); nh = slot(u + 1, 0); }
            __builtin_amdgcn_sched_barrier(0);
            c = mfma_f16(wm, a.hi[c2], c);
            c = mfma_f16(wh, a.mid[c2], c);
            c = mfma_f16(wh, a.hi[c2], c);
            __builtin_amdgcn_sched_barrier(0);
            wm = nm; wh = nh;
            acc[nt] = c;
        }
      } else {
        u32x4 wh = slot(0, 0), wm = slot(0, 1), wl = slot(0, 2);
#if GGNN_SPLIT_WH2
        u32x4 wh_n = wh;
#endif
#if GGNN_SPLIT_PF          // (experiment) all three planes of the NEXT unit fetched at the start of the current one: 6+ MFMAs ahead, +12 registers
        u32x4 nh = wh, nm = wm, nl = wl;
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int c2 = u / NTW, nt = T0 + u % NTW;
            const bool more = u + 1 < NU;
            f32x4 c = (ZERO && c2 == 0) ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[nt];
            if (more) { nl = slot(u + 1, 2); nm = slot(u + 1, 1); nh = slot(u + 1, 0); }
            __builtin_amdgcn_sched_barrier(0);
            c = mfma_bf16(wl, a.hi[c2], c);
            c = mfma_bf16(wm, a.mid[c2], c);
            c = mfma_bf16(wm, a.hi[c2], c);
            c = mfma_bf16(wh, a.lo[c2], c);
            c = mfma_bf16(wh, a.mid[c2], c);
            c = mfma_bf16(wh, a.hi[c2], c);
            __builtin_amdgcn_sched_barrier(0);
            wl = nl; wm = nm; wh = nh;
            acc[nt] = c;
        }
#else
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int c2 = u / NTW, nt = T0 + u % NTW;
            const bool more = u + 1 < NU;
            f32x4 c = (ZERO && c2 == 0) ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[nt];
#if GGNN_SPLIT_WH2
            if (more) wh_n = slot(u + 1, 0);
            __builtin_amdgcn_sched_barrier(0);
#endif
            c = mfma_bf16(wl, a.hi[c2], c);
            __builtin_amdgcn_sched_barrier(0);
            if (more) wl = slot(u + 1, 2);
            c = mfma_bf16(wm, a.mid[c2], c);
            c = mfma_bf16(wm, a.hi[c2], c);
            __builtin_amdgcn_sched_barrier(0);
            if (more) wm = slot(u + 1, 1);
            c = mfma_bf16(wh, a.lo[c2], c);
            c = mfma_bf16(wh, a.mid[c2], c);
            c = mfma_bf16(wh, a.hi[c2], c);
            __builtin_amdgcn_sched_barrier(0);
#if GGNN_SPLIT_WH2
            wh = wh_n;
#else
            if (more) wh = slot(u + 1, 0);
#endif
            acc[nt] = c;
        }
#endif
      }
    }
    // the D % 16 remainder k values on the f32 MFMA, their weights two tiles ahead (2 registers in flight)
    if constexpr (S::NR > 0 && NTW > 0) {
        constexpr int NRM = S::NR * NTW;
        auto rw = [&](int i) {
            const int nt = T0 + i % NTW, q = i / NTW;
            const int nth = C::nth_of(nt);
            const float* b = C::half_of(nt) ? img_b : img;
            return b[C::main_bytes(nth) / 4 + (q * 4 + kq) * nth * 16 + li + C::tile_in_half(nt) * 16];
        };
        float w0 = rw(0), w1 = NRM > 1 ? rw(1) : 0.f;
#pragma unroll
        for (int i = 0; i < NRM; ++i) {
            const int q = i / NTW, nt = T0 + i % NTW;
            acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0, af.r[q], acc[nt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            w0 = w1;
            if (i + 2 < NRM) w1 = rw(i + 2);
        }
    }
}

template <int D, int NTILES = StageCfg<D>::NT, bool ZERO = false, bool REMAT = false, int FMT = kSplitBf16x3>
__device__ __forceinline__ void stage_mma_split(f32x4 (&acc)[StageCfg<D>::NT], const SFrag<D>& a, const Frag<D>& af,
                                                const float* img, int li, int kq) {
    stage_mma_split_at<D, NTILES, ZERO, 0, REMAT, FMT>(acc, a, af, img, img + SplitCfg<D, FMT>::HA, li, kq);
}

// ONE output tile (wave-uniform, run time) of the same product: the cooperative tail pass
#define NC2_PLANE_SLOTS(NC2_, NTH_) ((NC2_) * 4 * (NTH_) * 16)
template <int D, bool ZERO, int FMT = kSplitBf16x3>
__device__ __forceinline__ void stage_mma_one_split(f32x4& acc, const SFrag<D>& a, const Frag<D>& af, const float* img, int li,
                                                    int kq, int tile) {
    using S = StageCfg<D>;
    using C = SplitCfg<D, FMT>;
    const bool hb = tile >= C::TA;                                    // (wave-uniform)
    const int nth = hb ? S::NT - C::TA : C::TA, til = hb ? tile - C::TA : tile;
    const float* him = hb ? img + C::HA : img;
    const int PL = NC2_PLANE_SLOTS(C::NC2, nth);                      // plane pitch in 16-byte slots
    const int CP = 4 * nth * 16;                                      // chunk pitch in slots
    const u32x4* base = reinterpret_cast<const u32x4*>(him) + kq * nth * 16 + li + til * 16;
    f32x4 cin = acc;
    if constexpr (ZERO) cin = f32x4{0.f, 0.f, 0.f, 0.f};
    u32x4 w0 = base[0], w1 = base[PL], w2 = C::NP > 2 ? base[(C::NP - 1) * PL] : w1;
#pragma unroll
    for (int c2 = 0; c2 < C::NC2; ++c2) {
        const int cn = c2 + 1 < C::NC2 ? c2 + 1 : c2;
        const u32x4 n0 = base[cn * CP], n1 = base[PL + cn * CP], n2 = C::NP > 2 ? base[(C::NP - 1) * PL + cn * CP] : n1;
        cin = split_products<false, FMT>(cin, w0, w1, w2, a.hi[c2], a.mid[c2], a.lo[c2]);
        w0 = n0; w1 = n1; w2 = n2;
    }
#pragma unroll
    for (int q = 0; q < S::NR; ++q) {
        const float wr = him[C::NP * PL * 4 + (q * 4 + kq) * nth * 16 + li + til * 16];
        cin = __builtin_amdgcn_mfma_f32_16x16x4f32(wr, af.r[q], cin, 0, 0, 0);
    }
    acc = cin;
}

// LDS-DMA of BYTES (a multiple of NW KiB) from src into LDS at dst by an NW-wave workgroup (dma_stage_image for any image size)
template <int BYTES, int NW>
__device__ __forceinline__ void dma_image(const float* src, float* dst, int wave, int lane) {
    constexpr int PER_WAVE = BYTES / (NW * 1024);
    static_assert(BYTES % (NW * 1024) == 0, "image must split into whole KiB per wave");
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

// The same transfer issued through inline assembly, so that the COMPILER does not know an LDS-DMA is in flight: with the builtin
// it puts s_waitcnt vmcnt(0) in front of every LDS read that follows (any ds_read might alias the destination) and in front of
// every barrier -- a wave that issues its DMA pieces and then starts its MFMA burst first waits for the pieces to land, which
// at 16-clock MFMAs is a sizeable part of a stage.  The kernel waits explicitly instead: dma_wait() before the barrier that
// publishes the image.  (The compiler's own vmcnt bookkeeping stays safe: the counter is in-order and these loads only add to
// it, so its waits can only be longer than needed.)
template <int BYTES, int NW>
__device__ __forceinline__ void dma_image_asm(const float* src, float* dst, int wave, int lane) {
    constexpr int PER_WAVE = BYTES / (NW * 1024);
    static_assert(BYTES % (NW * 1024) == 0, "image must split into whole KiB per wave");
    const unsigned voff = (unsigned)lane * 16u;
    const unsigned lbase = (unsigned)(unsigned long)(lds_void*)dst + (unsigned)wave * (PER_WAVE * 1024);
    const unsigned long long sb0 = reinterpret_cast<unsigned long long>(src) + (unsigned long long)wave * PER_WAVE * 1024;
#pragma unroll
    for (int i0 = 0; i0 < PER_WAVE; i0 += 4) {
        const unsigned long long sb = sb0 + (unsigned long long)i0 * 1024;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)sb);
        const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(sb >> 32));
        const unsigned long long s = ((unsigned long long)hi << 32) | lo;
        const unsigned l = __builtin_amdgcn_readfirstlane(lbase + (unsigned)i0 * 1024u);
        constexpr int n = PER_WAVE;
        if (i0 + 3 < n)
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\t"
                         "global_load_lds_dwordx4 %1, %2 offset:2048\n\tglobal_load_lds_dwordx4 %1, %2 offset:3072" :: "s"(l), "v"(voff), "s"(s) : "memory");
        else if (i0 + 2 < n)
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\t"
                         "global_load_lds_dwordx4 %1, %2 offset:2048" :: "s"(l), "v"(voff), "s"(s) : "memory");
        else if (i0 + 1 < n)
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024"
                         :: "s"(l), "v"(voff), "s"(s) : "memory");
        else
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(l), "v"(voff), "s"(s) : "memory");
    }
}
// KIB pieces of 1 KiB (any count) from src into LDS at dst by an NW-wave workgroup, issued like dma_image_asm: wave w moves the
// pieces [w Q, (w + 1) Q) with Q = KIB / NW, and the KIB % NW pieces behind them go one each to the first waves.
template <int KIB, int NW>
__device__ __forceinline__ void dma_kib_asm(const float* src, float* dst, int wave, int lane) {
    constexpr int Q = KIB / NW, R = KIB % NW;
    const unsigned voff = (unsigned)lane * 16u;
    const unsigned lds0 = (unsigned)(unsigned long)(lds_void*)dst;
    const unsigned long long src0 = reinterpret_cast<unsigned long long>(src);
    auto group = [&](unsigned byte_off, auto nc) {                    // <= 4 consecutive KiB at byte_off (wave-uniform)
        constexpr int n = decltype(nc)::value;
        const unsigned long long sb = src0 + byte_off;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)sb);
        const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(sb >> 32));
        const unsigned long long s = ((unsigned long long)hi << 32) | lo;
        const unsigned l = __builtin_amdgcn_readfirstlane(lds0 + byte_off);
        if constexpr (n == 4)
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\t"
                         "global_load_lds_dwordx4 %1, %2 offset:2048\n\tglobal_load_lds_dwordx4 %1, %2 offset:3072" :: "s"(l), "v"(voff), "s"(s) : "memory");
        else if constexpr (n == 3)
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\t"
                         "global_load_lds_dwordx4 %1, %2 offset:2048" :: "s"(l), "v"(voff), "s"(s) : "memory");
        else if constexpr (n == 2)
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024"
                         :: "s"(l), "v"(voff), "s"(s) : "memory");
        else
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(l), "v"(voff), "s"(s) : "memory");
    };
#pragma unroll
    for (int i0 = 0; i0 < Q; i0 += 4) {
        const unsigned off = ((unsigned)wave * Q + (unsigned)i0) * 1024u;
        if (Q - i0 >= 4) group(off, std::integral_constant<int, 4>{});
        else if (Q - i0 == 3) group(off, std::integral_constant<int, 3>{});
        else if (Q - i0 == 2) group(off, std::integral_constant<int, 2>{});
        else group(off, std::integral_constant<int, 1>{});
    }
    if constexpr (R > 0) {
        if (wave < R) group(((unsigned)(NW * Q) + (unsigned)wave) * 1024u, std::integral_constant<int, 1>{});
    }
}
// every load this wave has issued (DMA pieces included) has landed
__device__ __forceinline__ void dma_wait() { __builtin_amdgcn_s_waitcnt(0x0F70); }

}  // namespace ggnn
