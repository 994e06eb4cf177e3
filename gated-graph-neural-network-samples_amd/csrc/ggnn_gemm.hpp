// FP32-MFMA GEMM core for the GGNN hot path on gfx950 (CDNA4), written for skinny-K problems:
//   C[M,N] = [A_0 | A_1 | ... | A_{nseg-1}] x B        M = nodes (1e5), K = nseg*D (100..400),
//                                                        N = T*D / 2D / D (100..1024)
// Design (see DESIGN.md "K1/K3"):
//   * v_mfma_f32_16x16x4_f32 (exact f32, 157 TF peak).  The WEIGHT fragment is the first operand and
//     the ACTIVATION fragment the second, so every lane ends up with 4 CONSECUTIVE output columns of
//     ONE output row (float4 epilogue loads/stores, no LDS transpose).
//   * activations are never staged through LDS: each wave owns its rows exclusively, so lane
//     (row = lane&15, kq = lane>>4) loads float4 A[row][16c + 4kq .. +3] straight into VGPRs.  MFMA
//     (c,e) then contracts k = 16c + 4kq + e in lane group kq -- a permutation of k inside a
//     16-chunk, matched by the row the weight fragment is read from.  The A-operand concat
//     ([residuals | incoming | h], chem_tensorflow_sparse.py:211-212) is a list of segment
//     pointers; nothing is materialised.
//   * the weight chunk (KC x BN) lives in LDS with leading dimension BN+4 (== 4 mod 8): the four k
//     rows a 32-lane half touches land on disjoint banks -> conflict-free ds_read_b32.
//   * one stage = one KC-wide slice of one segment.  With a single stage (the message transform,
//     K = D) the weight panel is loaded ONCE per workgroup and the workgroup is persistent over row
//     tiles with the next tile's activation fragments prefetched into a second register set.
//   * XCD-aware block mapping: the column groups of one row tile run on the same XCD so the
//     activation rows they share are served by that XCD's L2.
#pragma once
#include "ggnn_common.h"

namespace ggnn {

constexpr int kGemmMaxSeg = 8;   // K segments of one product: up to 6 residual inputs + aggregated messages + h (the fused kernels: 3 + h)
constexpr int kGruFusedMaxNx = 3;

struct GemmOperands {
    const float* A[kGemmMaxSeg];     // segment base pointers, each [M, D] with row stride lda[s]
    int lda[kGemmMaxSeg];
    int nseg;              // K = nseg * D
    int D;
    const float* B;        // B(k,n) = B[(n / b_blk_cols) * b_blk_stride + k * ldb + (n % b_blk_cols)]
    int ldb;
    int b_blk_cols;        // N for a plain [K,N] matrix; D for the reference's [T,D,D] edge weights
    long b_blk_stride;
    int M, N;
};

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

// Epilogue transcendentals on the hardware exp2 / rcp units (v_exp_f32, v_rcp_f32: ~1 ulp each): the
// libm expf / tanhf / IEEE division sequences cost ~30-40 VALU instructions per element, which at 84
// elements per lane per 16-row tile was ~18 % of the fused GRU's time with the matrix pipe idle.
// Absolute error of sigmoid / tanh stays < 3e-7 (outputs are in (-1,1)); parity tolerance is 1e-6.
__device__ __forceinline__ float sigmoid_f(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanh_f(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x)); }

// The same two functions on four values with the bias folded into the exponent's scaling FMA (the caller pre-scales
// the bias): 2 packed FMA/ADD pairs + 4 v_exp + 4 v_rcp per float4 instead of 12 scalar ALU ops + 8 transcendentals.
constexpr float kLog2e = 1.4426950408889634f;
__device__ __forceinline__ f32x4 exp2_4(f32x4 t) {
    return f32x4{__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y), __builtin_amdgcn_exp2f(t.z), __builtin_amdgcn_exp2f(t.w)};
}
__device__ __forceinline__ f32x4 rcp_4(f32x4 d) {
    return f32x4{__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y), __builtin_amdgcn_rcpf(d.z), __builtin_amdgcn_rcpf(d.w)};
}
// sigmoid(z + b) with b_scaled = -log2(e) * b
__device__ __forceinline__ f32x4 sigmoid4_scaled(f32x4 z, f32x4 b_scaled) {
    return rcp_4(exp2_4(z * (-kLog2e) + b_scaled) + 1.0f);
}
// tanh(z + b) with b_scaled = 2 log2(e) * b
__device__ __forceinline__ f32x4 tanh4_scaled(f32x4 z, f32x4 b_scaled) {
    return 1.0f - 2.0f * rcp_4(exp2_4(z * (2.0f * kLog2e) + b_scaled) + 1.0f);
}

// ---- epilogues: called once per (row, 4 consecutive columns) ---------------------------------
struct EpiStore {
    float* C; int ldc;
    __device__ __forceinline__ void operator()(int row, int col, f32x4 v) const {
        st4(C + (size_t)row * ldc + col, v);
    }
};

// GRU gates (TF-1.3 GRUCell): g = sigmoid(acc + bg); columns [0,D) are r -> rh = r*h, columns
// [D,2D) are u.  D % 4 == 0, so a float4 never straddles the r/u boundary.
struct EpiGruGates {
    const float* bg; const float* h; float* rh; float* u; float* save_r; int D;
    __device__ __forceinline__ void operator()(int row, int col, f32x4 v) const {
        const f32x4 b = ld4(bg + col);
        f32x4 g;
        g.x = sigmoid_f(v.x + b.x); g.y = sigmoid_f(v.y + b.y);
        g.z = sigmoid_f(v.z + b.z); g.w = sigmoid_f(v.w + b.w);
        if (col < D) {
            const size_t o = (size_t)row * D + col;
            const f32x4 hv = ld4(h + o);
            st4(rh + o, g * hv);
            if (save_r) st4(save_r + o, g);
        } else {
            st4(u + (size_t)row * D + (col - D), g);
        }
    }
};

// GRU candidate + blend: c = act(acc + bc); h' = u*h + (1-u)*c
struct EpiGruCand {
    const float* bc; const float* h; const float* u; float* h_out; float* save_c; int D; int act;
    __device__ __forceinline__ void operator()(int row, int col, f32x4 v) const {
        const f32x4 b = ld4(bc + col);
        const size_t o = (size_t)row * D + col;
        f32x4 c = v + b;
        if (act == GGNN_ACT_TANH) {
            c.x = tanh_f(c.x); c.y = tanh_f(c.y); c.z = tanh_f(c.z); c.w = tanh_f(c.w);
        } else {
            c.x = fmaxf(c.x, 0.f); c.y = fmaxf(c.y, 0.f); c.z = fmaxf(c.z, 0.f); c.w = fmaxf(c.w, 0.f);
        }
        const f32x4 hv = ld4(h + o);
        const f32x4 uv = ld4(u + o);
        st4(h_out + o, uv * hv + (1.0f - uv) * c);
        if (save_c) st4(save_c + o, c);
    }
};

// out = act(acc + b): BasicRNNCell (chem_tensorflow_sparse.py:109-110) with act = tanh/relu, and a plain biased
// projection with act = GGNN_ACT_NONE (the hidden projection of CudnnCompatibleGRUCell).
#define GGNN_ACT_NONE 2
struct EpiBiasAct {
    const float* b; float* out; int ldo; int act;
    __device__ __forceinline__ void operator()(int row, int col, f32x4 v) const {
        f32x4 c = v + ld4(b + col);
        if (act == GGNN_ACT_TANH) { c.x = tanh_f(c.x); c.y = tanh_f(c.y); c.z = tanh_f(c.z); c.w = tanh_f(c.w); }
        else if (act == GGNN_ACT_RELU) { c.x = fmaxf(c.x, 0.f); c.y = fmaxf(c.y, 0.f); c.z = fmaxf(c.z, 0.f); c.w = fmaxf(c.w, 0.f); }
        st4(out + (size_t)row * ldo + col, c);
    }
};

// CudnnCompatibleGRUCell candidate (chem_tensorflow_sparse.py:105-108):
//   c = tanh(x Wcx + bcx + r * (h Wch + bch)); h' = u*h + (1-u)*c     (hc = h Wch + bch precomputed)
struct EpiCudnnCand {
    const float* bcx; const float* r; const float* hc; const float* h; const float* u; float* h_out; int D; float* save_c;
    __device__ __forceinline__ void operator()(int row, int col, f32x4 v) const {
        const size_t o = (size_t)row * D + col;
        f32x4 c = v + ld4(bcx + col) + ld4(r + o) * ld4(hc + o);
        c.x = tanh_f(c.x); c.y = tanh_f(c.y); c.z = tanh_f(c.z); c.w = tanh_f(c.w);
        const f32x4 uv = ld4(u + o);
        st4(h_out + o, uv * ld4(h + o) + (1.0f - uv) * c);
        if (save_c) st4(save_c + o, c);
    }
};

// ---- dX = dY W^T with the GRU gate algebra in the epilogue ------------------------------------------------------------------
// Candidate:  P = dpc Wc^T  [V, (nx+1) D].  Columns [0, nx*D): dx (the x part, stored; the gates pass adds to it).
//             Columns [nx*D, (nx+1)*D): drh ->  dh += drh * r ;  dpr = drh * h * r (1 - r) -> dpg[:, 0:D]          (stage 2)
struct EpiBwdCand {
    float* dx; int ld_dx; int xcols;                        // xcols = nx * D
    const float* h; const float* r; float* dh; float* dpg; int D;
    __device__ __forceinline__ void operator()(int row, int col, f32x4 v) const {
        if (col < xcols) { st4(dx + (size_t)row * ld_dx + col, v); return; }
        const int c = col - xcols;
        const size_t o = (size_t)row * D + c;
        const f32x4 hv = ld4(h + o), rv = ld4(r + o);
        st4(dh + o, ld4(dh + o) + v * rv);
        st4(dpg + (size_t)row * 2 * D + c, v * hv * rv * (1.0f - rv));
    }
};

// Gates:  Q = dpg Wg^T  [V, (nx+1) D].  Columns of the residual inputs: dx += Q.  Columns of the aggregated messages (the
//         LAST x segment): d_incoming = (dx + Q) * inv_deg[row] (mean aggregation, chem_tensorflow_sparse.py:206-209) -> dinc.
//         Columns of h: dh += Q.
struct EpiBwdGates {
    float* dx; int ld_dx; int xcols; int inc0;              // inc0 = (nx-1)*D: first column of the aggregated-messages segment
    float* dinc; const float* nin; int T; int use_avg;
    float* dh; int D;
    __device__ __forceinline__ void operator()(int row, int col, f32x4 v) const {
        if (col >= xcols) {
            const size_t o = (size_t)row * D + (col - xcols);
            st4(dh + o, ld4(dh + o) + v);
            return;
        }
        const f32x4 t = ld4(dx + (size_t)row * ld_dx + col) + v;
        if (col < inc0) { st4(dx + (size_t)row * ld_dx + col, t); return; }
        f32x4 o = t;
        if (use_avg) {
            float deg = 0.f;
            for (int k = 0; k < T; ++k) deg += nin[(size_t)row * T + k];
            o = t / (deg + 1e-7f);
        }
        st4(dinc + (size_t)row * D + (col - inc0), o);
    }
};


// General dX epilogue of the backward pass: Q = dY W^T with columns [x_0 .. x_{nx-1} | h] (the h block may be absent):
//   residual columns -> dx (= or +=), the LAST x segment -> dinc = (acc_dx ? dx + Q : Q) (/ (sum_t nin + 1e-7)), h columns -> dh (= or +=)
struct EpiBwdDx {
    float* dx; int ld_dx; int xcols; int inc0;
    float* dinc; const float* nin; int T; int use_avg;
    float* dh; int D; int acc_dx; int acc_dh;
    __device__ __forceinline__ void operator()(int row, int col, f32x4 v) const {
        if (col >= xcols) {
            const size_t o = (size_t)row * D + (col - xcols);
            st4(dh + o, acc_dh ? ld4(dh + o) + v : v);
            return;
        }
        f32x4 t = v;
        if (acc_dx) t = t + ld4(dx + (size_t)row * ld_dx + col);
        if (col < inc0) { st4(dx + (size_t)row * ld_dx + col, t); return; }
        if (use_avg) {
            float deg = 0.f;
            for (int k = 0; k < T; ++k) deg += nin[(size_t)row * T + k];
            t = t / (deg + 1e-7f);
        }
        st4(dinc + (size_t)row * D + (col - inc0), t);
    }
};

struct GruFusedArgs {
    const float* x[3];
    int nx;
    const float* h;
    const float* Wg; const float* bg; const float* Wc; const float* bc;
    float* h_out; float* save_r; float* save_u; float* save_c;
    float* save_x;         // gather-fused variant with save_r/u/c: the gathered last segment (incoming), an operand of the weight gradients
    int V; int act;
    // gather-fused variant: the last x segment = segment sum of g_H rows (NULL: plain loads from x[nx-1])
    const float* g_H; const int* g_row_ptr; const int* g_idx; const float* g_nin; int g_T; int g_use_avg;
    int* tickets;          // NULL: static tile split; else a device int32, 0 at launch: tiles are handed out dynamically
    int fmt;               // operand format of the split-form products and of `packed` (GGNN_GRU_FMT_*; 0 = BF16X3), per launch
    int dbg;               // ablation bitmask from GGNN_GRU_DBG (0 in production)
    unsigned long long* tdbg;   // per-stage s_memtime stamps (GGNN_GRU_TPTR, debug only)
};

int gru_fused_supported(int D);
int gru_pack_floats(int D, int nx);      // floats of scratch the fused kernel needs for its packed weight images
int gru_fused_dispatch(const GruFusedArgs& a, int D, float* packed, hipStream_t st);

// ---- the kernel ---------------------------------------------------------------------------------
// KC: K-slice per stage (D % KC == 0, KC % 4 == 0).  MT: 16-row tiles per wave.  NT: 16-column tiles
// per workgroup column group (all waves share the columns).  NW: waves per workgroup.
template <int KC, int MT, int NT, int NW>
struct GemmCfg {
    static constexpr int BN = NT * 16;
    static constexpr int LDB = BN + 4;          // == 4 (mod 8): conflict-free weight-fragment reads
    static constexpr int BM = NW * MT * 16;
    static constexpr int NC = KC / 16;          // full 16-wide k chunks (4 MFMAs each)
    static constexpr int NR = (KC % 16) / 4;    // remainder MFMAs (k = 16*NC + 4q + kq)
    static constexpr int THREADS = NW * 64;
    static constexpr int BVEC = (KC * (BN / 4) + THREADS - 1) / THREADS;   // float4 per thread per B chunk
    static constexpr size_t LDS_BYTES = (size_t)KC * LDB * sizeof(float);
};

template <int KC, int MT, int NT, int NW, class Epi>
__global__ __launch_bounds__(NW * 64) void ggnn_gemm_kernel(GemmOperands g, Epi epi, int ncg,
                                                            int row_tiles, int workers) {
    using Cfg = GemmCfg<KC, MT, NT, NW>;
    constexpr int BN = Cfg::BN, LDB = Cfg::LDB, BM = Cfg::BM, NC = Cfg::NC, NR = Cfg::NR;
    extern __shared__ __attribute__((aligned(16))) float Bs[];   // [KC][LDB]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kq = lane >> 4;

    // XCD-aware mapping (speed only): the ncg column groups of a row worker share an XCD.
    const int bid = blockIdx.x;
    const int xcd = bid % kNumXcd, slot = bid / kNumXcd;
    const int cg = slot % ncg;
    const int worker = (slot / ncg) * kNumXcd + xcd;
    if (worker >= workers) return;
    const int n0 = cg * BN;

    const int spseg = g.D / KC;                 // stages per segment
    const int nstages = g.nseg * spseg;
    const bool b_resident = (nstages == 1);

    // ---- weight chunk: global -> registers -> LDS -------------------------------------------
    f32x4 breg[Cfg::BVEC];
    auto load_b = [&](int stage) {
        const int kg0 = stage * KC;             // global k of the chunk's first row
#pragma unroll
        for (int q = 0; q < Cfg::BVEC; ++q) {
            const int idx = tid + q * Cfg::THREADS;
            const int kk = idx / (BN / 4), v4 = idx % (BN / 4);
            const int n = n0 + 4 * v4;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (kk < KC && n < g.N) {
                const int blk = n / g.b_blk_cols, nn = n % g.b_blk_cols;
                v = ld4(g.B + (size_t)blk * g.b_blk_stride + (size_t)(kg0 + kk) * g.ldb + nn);
            }
            breg[q] = v;
        }
    };
    auto store_b = [&]() {
#pragma unroll
        for (int q = 0; q < Cfg::BVEC; ++q) {
            const int idx = tid + q * Cfg::THREADS;
            const int kk = idx / (BN / 4), v4 = idx % (BN / 4);
            if (kk < KC) st4(Bs + kk * LDB + 4 * v4, breg[q]);
        }
    };

    // ---- activation fragments: global -> registers ------------------------------------------
    struct AFrag { f32x4 v[MT][NC > 0 ? NC : 1]; float r[MT][NR > 0 ? NR : 1]; };
    auto load_a = [&](AFrag& a, int rt, int stage) {
        const int seg = stage / spseg, koff = (stage % spseg) * KC;
        const float* base = g.A[seg];
        const int lda = g.lda[seg];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            int row = rt * BM + (wave * MT + mt) * 16 + li;
            row = row < g.M ? row : g.M - 1;    // clamped rows are computed and discarded
            const float* p = base + (size_t)row * lda + koff + 4 * kq;
#pragma unroll
            for (int c = 0; c < NC; ++c) a.v[mt][c] = ld4(p + 16 * c);
#pragma unroll
            for (int q = 0; q < NR; ++q) a.r[mt][q] = p[16 * NC + 4 * q - 4 * kq + kq];
        }
    };

    AFrag a_cur, a_nxt;
    int rt = worker;
    if (rt >= row_tiles) return;
    if (b_resident) load_b(0);
    load_a(a_cur, rt, 0);
    if (b_resident) { store_b(); __syncthreads(); }

    for (; rt < row_tiles; rt += workers) {
        f32x4 acc[MT][NT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

        for (int s = 0; s < nstages; ++s) {
            if (!b_resident) load_b(s);                  // weights first (older in the vmcnt queue) ...
            // ... then the NEXT item's activations, which stay in flight across the MFMA block
            int rt_n = rt, s_n = s + 1;
            if (s_n == nstages) { s_n = 0; rt_n = rt + workers; }
            const bool has_next = rt_n < row_tiles;
            if (has_next) load_a(a_nxt, rt_n, s_n);
            if (!b_resident) {
                __syncthreads();                          // WAR: previous stage finished reading Bs
                store_b();
                __syncthreads();
            }
#pragma unroll
            for (int c = 0; c < NC; ++c) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float* brow = Bs + (16 * c + 4 * kq + e) * LDB + li;
                    float w[NT];
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) w[nt] = brow[nt * 16];
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[nt], a_cur.v[mt][c][e],
                                                                               acc[mt][nt], 0, 0, 0);
                }
            }
#pragma unroll
            for (int q = 0; q < NR; ++q) {
                const float* brow = Bs + (16 * NC + 4 * q + kq) * LDB + li;
                float w[NT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) w[nt] = brow[nt * 16];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[nt], a_cur.r[mt][q],
                                                                           acc[mt][nt], 0, 0, 0);
            }
            if (has_next) a_cur = a_nxt;
        }

        // epilogue: lane (li,kq) holds C[row = tile row li][col = n0 + nt*16 + 4kq .. +3]
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int row = rt * BM + (wave * MT + mt) * 16 + li;
            if (row < g.M) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int col = n0 + nt * 16 + 4 * kq;
                    if (col < g.N) epi(row, col, acc[mt][nt]);
                }
            }
        }
    }
}

}  // namespace ggnn
