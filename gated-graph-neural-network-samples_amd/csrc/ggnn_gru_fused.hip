// K3 fused: the whole TF-1.3 GRUCell node update (chem_tensorflow_sparse.py:211-216) in ONE launch:
//     [r|u] = sigmoid([x|h] Wg + bg);  c = act([x | r*h] Wc + bc);  h' = u*h + (1-u)*c
//
// Why it chains without LDS or shuffles: in the GEMM core (ggnn_gemm.hpp) lane (row = lane&15, kq = lane>>4)
// holds output columns 16*nt + 4*kq + e of its row in accumulator tile nt -- and consumes the k indices
// 16*c + 4*kq + e of its row from activation chunk c.  Output tile nt IS activation chunk c = nt.  So
// r*h (gates output x resident h fragment) feeds the candidate GEMM straight from registers, and u, h, c
// meet in the blend in the same layout.  Only the D%16 remainder columns need 4 cross-lane reads.
//
// Per 16-row wave tile nothing but x, h are read and h' written: the r*h / u round trip of the two-launch
// version (4 x [V,D] fp32 of HBM traffic) and one launch boundary are gone.
//
// Weights ([x|h] rows of Wg split in r/u column halves, then Wc) stream through a double-buffered LDS
// chunk ring, one D x D chunk (padded to NT*16 columns) per stage, shared by the NW waves of the
// workgroup; the next chunk's global loads are issued before the current chunk's MFMAs.
// Workgroups are persistent over a contiguous, evenly split range of 16-row tiles (all workgroups run
// the same number of passes).
#include "ggnn_gemm.hpp"

namespace ggnn {

template <int D>
struct GruCfg {
    static constexpr int NT = (D + 15) / 16;
    static constexpr int NC = D / 16;
    static constexpr int NR = (D % 16) / 4;
    static constexpr int BN = NT * 16;
    static constexpr int LDB = BN + 4;
    static constexpr int CHUNK = D * LDB;              // floats per LDS chunk
};

template <int D>
struct Frag {
    f32x4 v[GruCfg<D>::NC > 0 ? GruCfg<D>::NC : 1];
    float r[GruCfg<D>::NR > 0 ? GruCfg<D>::NR : 1];
};

template <int D>
__device__ __forceinline__ void load_frag(Frag<D>& f, const float* base, int row, int kq) {
    constexpr int NC = GruCfg<D>::NC, NR = GruCfg<D>::NR;
    const float* p = base + (size_t)row * D + 4 * kq;
#pragma unroll
    for (int c = 0; c < NC; ++c) f.v[c] = ld4(p + 16 * c);
#pragma unroll
    for (int q = 0; q < NR; ++q) f.r[q] = base[(size_t)row * D + 16 * NC + 4 * q + kq];
}

template <int D>
__device__ __forceinline__ void chunk_mma(f32x4 (&acc)[GruCfg<D>::NT], const Frag<D>& a, const float* Bs, int li, int kq) {
    using C = GruCfg<D>;
#pragma unroll
    for (int c = 0; c < C::NC; ++c) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float* brow = Bs + (16 * c + 4 * kq + e) * C::LDB + li;
            float w[C::NT];
#pragma unroll
            for (int nt = 0; nt < C::NT; ++nt) w[nt] = brow[nt * 16];
#pragma unroll
            for (int nt = 0; nt < C::NT; ++nt)
                acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[nt], a.v[c][e], acc[nt], 0, 0, 0);
        }
    }
#pragma unroll
    for (int q = 0; q < C::NR; ++q) {
        const float* brow = Bs + (16 * C::NC + 4 * q + kq) * C::LDB + li;
        float w[C::NT];
#pragma unroll
        for (int nt = 0; nt < C::NT; ++nt) w[nt] = brow[nt * 16];
#pragma unroll
        for (int nt = 0; nt < C::NT; ++nt)
            acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[nt], a.r[q], acc[nt], 0, 0, 0);
    }
}

// element e (0..3, lane-varying) of a float4 without dynamic register indexing
__device__ __forceinline__ float pick(const f32x4& v, int e) {
    return e == 0 ? v.x : (e == 1 ? v.y : (e == 2 ? v.z : v.w));
}

template <int D, int NW>
__global__ __launch_bounds__(NW * 64) void ggnn_gru_fused_kernel(GruFusedArgs a) {
    using C = GruCfg<D>;
    constexpr int NT = C::NT, NC = C::NC, NR = C::NR, LDB = C::LDB, THREADS = NW * 64;
    constexpr int BVEC = (D * NT * 4 + THREADS - 1) / THREADS;      // float4 per thread per chunk
    extern __shared__ __attribute__((aligned(16))) float Bs[];     // [2][D][LDB]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kq = lane >> 4;
    const int nx = a.nx;
    const int nchunks = 3 * (nx + 1);

    // even split of the 16-row tiles over the grid; every workgroup runs the same number of passes
    const int wt_total = (a.V + 15) / 16;
    const int nb = gridDim.x;
    const int t_beg = (int)(((long long)wt_total * blockIdx.x) / nb);
    const int t_end = (int)(((long long)wt_total * (blockIdx.x + 1)) / nb);
    const int max_tiles = (wt_total + nb - 1) / nb;
    const int passes = (max_tiles + NW - 1) / NW;

    // ---- weight chunk streaming ------------------------------------------------------------------
    f32x4 breg[BVEC];
    auto load_b = [&](int ci) {
        const float* W; int r0, c0, ldw;
        if (ci < 2 * (nx + 1)) { W = a.Wg; r0 = (ci >> 1) * D; c0 = (ci & 1) * D; ldw = 2 * D; }
        else { W = a.Wc; r0 = (ci - 2 * (nx + 1)) * D; c0 = 0; ldw = D; }
#pragma unroll
        for (int q = 0; q < BVEC; ++q) {
            const int idx = tid + q * THREADS;
            const int kk = idx / (NT * 4), v4 = idx % (NT * 4);
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (kk < D && 4 * v4 < D) v = ld4(W + (size_t)(r0 + kk) * ldw + c0 + 4 * v4);
            breg[q] = v;
        }
    };
    auto store_b = [&](float* dst) {
#pragma unroll
        for (int q = 0; q < BVEC; ++q) {
            const int idx = tid + q * THREADS;
            const int kk = idx / (NT * 4), v4 = idx % (NT * 4);
            if (kk < D) st4(dst + kk * LDB + 4 * v4, breg[q]);
        }
    };

    int cur = 0;
    load_b(0);
    store_b(Bs);
    __syncthreads();

    for (int p = 0; p < passes; ++p) {
        const int tile = t_beg + p * NW + wave;
        const bool active = tile < t_end;                      // wave-uniform
        const int row = tile * 16 + li;
        const int rowc = active ? (row < a.V ? row : a.V - 1) : 0;
        const bool last_pass = (p + 1 == passes);

        Frag<D> hf, af, an;
        f32x4 acc_r[NT], acc_u[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) { acc_r[nt] = f32x4{0.f, 0.f, 0.f, 0.f}; acc_u[nt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        if (active) {
            load_frag<D>(hf, a.h, rowc, kq);
            if (nx > 0) load_frag<D>(af, a.x[0], rowc, kq);
        }

        // one stage: prefetch next chunk's weights, MFMAs on the current chunk, publish the next chunk
        int ci = 0;
#define GGNN_STAGE(ACC, FRAG)                                                          \
        {                                                                              \
            const int nci = ci + 1;                                                    \
            const bool more = (nci < nchunks) || !last_pass;                           \
            if (more) load_b(nci < nchunks ? nci : 0);                                 \
            if (active) chunk_mma<D>(ACC, FRAG, Bs + cur * C::CHUNK, li, kq);          \
            if (more) store_b(Bs + (cur ^ 1) * C::CHUNK);                              \
            __syncthreads();                                                           \
            cur ^= 1; ci = nci;                                                        \
        }

        // ---- gates: [x | h] Wg, r columns then u columns of each K segment --------------------------
        for (int s = 0; s <= nx; ++s) {
            if (s == nx) af = hf;
            if (active && s + 1 < nx) load_frag<D>(an, a.x[s + 1], rowc, kq);
            GGNN_STAGE(acc_r, af)
            GGNN_STAGE(acc_u, af)
            if (s + 1 < nx) af = an;
        }

        // ---- r = sigmoid(.), u = sigmoid(.), rh = r*h in activation-fragment layout -------------------
        Frag<D> rh;
        if (active) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int col = nt * 16 + 4 * kq;
                if (col < D) {
                    const f32x4 br = ld4(a.bg + col), bu = ld4(a.bg + D + col);
                    f32x4 r, u;
                    r.x = sigmoid_f(acc_r[nt].x + br.x); r.y = sigmoid_f(acc_r[nt].y + br.y);
                    r.z = sigmoid_f(acc_r[nt].z + br.z); r.w = sigmoid_f(acc_r[nt].w + br.w);
                    u.x = sigmoid_f(acc_u[nt].x + bu.x); u.y = sigmoid_f(acc_u[nt].y + bu.y);
                    u.z = sigmoid_f(acc_u[nt].z + bu.z); u.w = sigmoid_f(acc_u[nt].w + bu.w);
                    acc_r[nt] = r; acc_u[nt] = u;
                    if (row < a.V) {
                        if (a.save_r) st4(a.save_r + (size_t)row * D + col, r);
                        if (a.save_u) st4(a.save_u + (size_t)row * D + col, u);
                    }
                }
            }
#pragma unroll
            for (int c = 0; c < NC; ++c) rh.v[c] = acc_r[c] * hf.v[c];
#pragma unroll
            for (int q = 0; q < NR; ++q) {
                // remainder column 16NC + 4q + kq lives in tile NC of lane (li, kq' = q), element e = kq
                float t0 = __shfl(acc_r[NT - 1].x, li + 16 * q), t1 = __shfl(acc_r[NT - 1].y, li + 16 * q);
                float t2 = __shfl(acc_r[NT - 1].z, li + 16 * q), t3 = __shfl(acc_r[NT - 1].w, li + 16 * q);
                const float rr = kq == 0 ? t0 : (kq == 1 ? t1 : (kq == 2 ? t2 : t3));
                rh.r[q] = rr * hf.r[q];
            }
        }

        // ---- candidate: [x | r*h] Wc --------------------------------------------------------------------
        f32x4 acc_c[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc_c[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (active && nx > 0) load_frag<D>(af, a.x[0], rowc, kq);
        for (int s = 0; s <= nx; ++s) {
            if (s == nx) af = rh;
            if (active && s + 1 < nx) load_frag<D>(an, a.x[s + 1], rowc, kq);
            GGNN_STAGE(acc_c, af)
            if (s + 1 < nx) af = an;
        }
#undef GGNN_STAGE

        // ---- c = act(.), h' = u*h + (1-u)*c ----------------------------------------------------------------
        // remainder tile: lane (li,kq) needs h cols 16NC + 4kq + e = remainder frag q = kq of lane (li, e);
        // the cross-lane reads stay outside lane-divergent control flow
        f32x4 hrem = {0.f, 0.f, 0.f, 0.f};
        if (active) {
#pragma unroll
            for (int q = 0; q < NR; ++q) {
                const float h0 = __shfl(hf.r[q], li), h1 = __shfl(hf.r[q], li + 16);
                const float h2 = __shfl(hf.r[q], li + 32), h3 = __shfl(hf.r[q], li + 48);
                if (kq == q) hrem = f32x4{h0, h1, h2, h3};
            }
        }
        if (active && row < a.V) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int col = nt * 16 + 4 * kq;
                if (col < D) {
                    f32x4 hv;
                    if (nt < NC) hv = hf.v[nt];
                    else hv = hrem;
                    const f32x4 b = ld4(a.bc + col);
                    f32x4 c = acc_c[nt] + b;
                    if (a.act == GGNN_ACT_TANH) { c.x = tanhf(c.x); c.y = tanhf(c.y); c.z = tanhf(c.z); c.w = tanhf(c.w); }
                    else { c.x = fmaxf(c.x, 0.f); c.y = fmaxf(c.y, 0.f); c.z = fmaxf(c.z, 0.f); c.w = fmaxf(c.w, 0.f); }
                    const f32x4 u = acc_u[nt];
                    st4(a.h_out + (size_t)row * D + col, u * hv + (1.0f - u) * c);
                    if (a.save_c) st4(a.save_c + (size_t)row * D + col, c);
                }
            }
        }
    }
}

template <int D, int NW>
static int launch_gru_fused(const GruFusedArgs& a, hipStream_t st) {
    using C = GruCfg<D>;
    const size_t lds = (size_t)2 * C::CHUNK * sizeof(float);
    const int wt_total = (a.V + 15) / 16;
    int nb = num_cus();
    const int need = (wt_total + NW - 1) / NW;
    if (nb > need) nb = need;
    static bool attr_set = false;
    if (!attr_set && lds > 64 * 1024) {
        GGNN_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&ggnn_gru_fused_kernel<D, NW>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    hipLaunchKernelGGL((ggnn_gru_fused_kernel<D, NW>), dim3(nb), dim3(NW * 64), lds, st, a);
    GGNN_CHECK_HIP(hipGetLastError());
    return GGNN_OK;
}

// Returns 1 if a fused instantiation exists for this hidden size.
// (D = 128 would spill: 256 VGPRs + scratch; it and larger sizes take the two-launch path.)
int gru_fused_supported(int D) { return D == 100 || D == 64 || D == 32; }

int gru_fused_dispatch(const GruFusedArgs& a, int D, hipStream_t st) {
    switch (D) {
        case 100: return launch_gru_fused<100, 8>(a, st);
        case 64: return launch_gru_fused<64, 8>(a, st);
        case 32: return launch_gru_fused<32, 8>(a, st);
        default: return fail(GGNN_E_UNSUPPORTED, "no fused GRU for hidden size %d", D);
    }
}

}  // namespace ggnn

extern "C" int ggnn_gru_is_fused(int D) { return ggnn::gru_fused_supported(D); }
