// Launchers + C-ABI entry points for the FP32-MFMA GEMM family:
//   ggnn_msg_transform_f32  (K1, chem_tensorflow_sparse.py:160-164)
//   ggnn_gru_f32            (K3, chem_tensorflow_sparse.py:211-216 / TF-1.3 GRUCell)
//   ggnn_gemm_f32           (plain C = [A0|A1|..] x B, used by the host layer's backward pass)
#include "ggnn_gemm.hpp"

namespace ggnn {

static inline int round_up8(int x) { return (x + 7) / 8 * 8; }

// Resident workgroups per CU targeted by the persistent (single-stage) variant.
constexpr int kPersistBlocksPerCU = 2;

template <int KC, int MT, int NT, int NW, class Epi>
static int launch_gemm(const GemmOperands& g, const Epi& epi, hipStream_t st) {
    using Cfg = GemmCfg<KC, MT, NT, NW>;
    const int ncg = (g.N + Cfg::BN - 1) / Cfg::BN;
    const int row_tiles = (g.M + Cfg::BM - 1) / Cfg::BM;
    if (row_tiles == 0 || ncg == 0) return GGNN_OK;
    const int nstages = g.nseg * (g.D / KC);
    int workers = round_up8(row_tiles);
    if (nstages == 1) {
        int target = num_cus() * kPersistBlocksPerCU / ncg;
        target = target / 8 * 8;
        if (target < 8) target = 8;
        if (workers > target) workers = target;
    }
    const dim3 grid((unsigned)(ncg * workers));
    hipLaunchKernelGGL((ggnn_gemm_kernel<KC, MT, NT, NW, Epi>), grid, dim3(Cfg::THREADS), Cfg::LDS_BYTES, st,
                       g, epi, ncg, row_tiles, workers);
    GGNN_CHECK_HIP(hipGetLastError());
    return GGNN_OK;
}

// padded column count if N is covered by groups of nt*16 columns
static inline int padded_cols(int N, int nt) { const int bn = nt * 16; return (N + bn - 1) / bn * bn; }

template <class Epi>
static int dispatch_gemm(const GemmOperands& g, const Epi& epi, hipStream_t st) {
    constexpr int MT = 1, NW = 8;
    const int D = g.D, N = g.N;
    if (D % 100 == 0) {
        if (padded_cols(N, 5) <= padded_cols(N, 7)) return launch_gemm<100, MT, 5, NW>(g, epi, st);
        return launch_gemm<100, MT, 7, NW>(g, epi, st);
    }
    if (D % 64 == 0) {
        if (padded_cols(N, 8) <= padded_cols(N, 4)) return launch_gemm<64, MT, 8, NW>(g, epi, st);
        return launch_gemm<64, MT, 4, NW>(g, epi, st);
    }
    if (D % 32 == 0) {
        const int p8 = padded_cols(N, 8), p4 = padded_cols(N, 4), p2 = padded_cols(N, 2);
        if (p8 <= p4 && p8 <= p2) return launch_gemm<32, MT, 8, NW>(g, epi, st);
        if (p4 <= p2) return launch_gemm<32, MT, 4, NW>(g, epi, st);
        return launch_gemm<32, MT, 2, NW>(g, epi, st);
    }
    return fail(GGNN_E_UNSUPPORTED, "hidden size %d unsupported (need a multiple of 100, 64 or 32)", D);
}

static int check_common(int V, int D) {
    GGNN_CHECK_ARG(V >= 0, "negative node count %d", V);
    GGNN_CHECK_ARG(D > 0 && D % 4 == 0, "hidden size %d must be a positive multiple of 4", D);
    return GGNN_OK;
}

}  // namespace ggnn

using namespace ggnn;

extern "C" int ggnn_msg_transform_f32(const float* h, int ldh, const float* W, float* H, int V, int D, int T,
                                      ggnn_stream_t stream) {
    if (int rc = check_common(V, D)) return rc;
    GGNN_CHECK_ARG(T > 0, "num_edge_types %d must be positive", T);
    if (V == 0) return GGNN_OK;
    GGNN_CHECK_ARG(h && W && H, "null pointer");
    GGNN_CHECK_ARG(ldh >= D && ldh % 4 == 0, "ldh %d must be >= D and a multiple of 4", ldh);
    GGNN_CHECK_ARG(aligned16(h) && aligned16(W) && aligned16(H), "pointers must be 16-byte aligned");
    GemmOperands g{};
    g.A[0] = h; g.lda[0] = ldh; g.nseg = 1; g.D = D;
    g.B = W; g.ldb = D; g.b_blk_cols = D; g.b_blk_stride = (long)D * D;   // [T,D,D]: column block t is W[t]
    g.M = V; g.N = T * D;
    EpiStore epi{H, T * D};
    return dispatch_gemm(g, epi, (hipStream_t)stream);
}

extern "C" int ggnn_gemm_f32(const float* const* a_segs, int nseg, int D, int lda, const float* B, int ldb, float* C,
                             int ldc, int M, int N, ggnn_stream_t stream) {
    if (int rc = check_common(M, D)) return rc;
    GGNN_CHECK_ARG(nseg >= 1 && nseg <= kGemmMaxSeg, "nseg %d outside 1..%d", nseg, kGemmMaxSeg);
    GGNN_CHECK_ARG(lda >= D && lda % 4 == 0, "lda %d must be >= D and a multiple of 4", lda);
    GGNN_CHECK_ARG(N > 0 && N % 4 == 0 && ldb >= N && ldb % 4 == 0 && ldc >= N && ldc % 4 == 0,
                   "N/ldb/ldc must be multiples of 4 with ldb,ldc >= N");
    if (M == 0) return GGNN_OK;
    GGNN_CHECK_ARG(a_segs && B && C, "null pointer");
    GemmOperands g{};
    for (int s = 0; s < nseg; ++s) {
        GGNN_CHECK_ARG(a_segs[s] && aligned16(a_segs[s]), "segment %d null or misaligned", s);
        g.A[s] = a_segs[s]; g.lda[s] = lda;
    }
    g.nseg = nseg; g.D = D; g.B = B; g.ldb = ldb; g.b_blk_cols = N; g.b_blk_stride = 0; g.M = M; g.N = N;
    EpiStore epi{C, ldc};
    return dispatch_gemm(g, epi, (hipStream_t)stream);
}

extern "C" size_t ggnn_gru_workspace_bytes(int V, int D) {
    if (V < 0 || D <= 0) return 0;
    // un-fused path: r*h and u; fused path: the packed weight images (up to nx = 3)
    return (size_t)2 * (size_t)V * (size_t)D * sizeof(float) + (size_t)gru_pack_floats(D, 3) * sizeof(float);
}

static int gru_args_check(const float* const* x_segs, int nx, const float* h, int V, int D) {
    if (int rc = check_common(V, D)) return rc;
    // the reference concatenates any number of residual inputs (chem_tensorflow_sparse.py:139-145,211-212); the generic kernels
    // take kGemmMaxSeg - 1 = 7 input segments (6 residual inputs + the aggregated messages), the fused single-launch kernels 3
    GGNN_CHECK_ARG(nx >= 1 && nx <= kGemmMaxSeg - 1, "nx %d outside 1..%d (residual inputs + aggregated messages)", nx, kGemmMaxSeg - 1);
    if (V == 0) return GGNN_OK;
    GGNN_CHECK_ARG(x_segs && h && aligned16(h), "null or misaligned pointer");
    for (int s = 0; s < nx; ++s) GGNN_CHECK_ARG(x_segs[s] && aligned16(x_segs[s]), "x segment %d null or misaligned", s);
    return GGNN_OK;
}

extern "C" int ggnn_gru_gates_f32(const float* const* x_segs, int nx, const float* h, const float* Wg, const float* bg,
                                  float* rh, float* u, float* save_r, int V, int D, ggnn_stream_t stream) {
    if (int rc = gru_args_check(x_segs, nx, h, V, D)) return rc;
    if (V == 0) return GGNN_OK;
    GGNN_CHECK_ARG(Wg && bg && rh && u, "null pointer");
    GGNN_CHECK_ARG(aligned16(Wg) && aligned16(bg) && aligned16(rh) && aligned16(u) && (!save_r || aligned16(save_r)),
                   "pointers must be 16-byte aligned");
    GemmOperands g{};
    for (int s = 0; s < nx; ++s) { g.A[s] = x_segs[s]; g.lda[s] = D; }
    g.A[nx] = h; g.lda[nx] = D;
    g.nseg = nx + 1; g.D = D; g.M = V;
    g.B = Wg; g.ldb = 2 * D; g.b_blk_cols = 2 * D; g.b_blk_stride = 0; g.N = 2 * D;
    EpiGruGates eg{bg, h, rh, u, save_r, D};
    return dispatch_gemm(g, eg, (hipStream_t)stream);
}

extern "C" int ggnn_gru_candidate_f32(const float* const* x_segs, int nx, const float* rh, const float* h,
                                      const float* u, const float* Wc, const float* bc, float* h_out, float* save_c,
                                      int V, int D, int act, ggnn_stream_t stream) {
    if (int rc = gru_args_check(x_segs, nx, h, V, D)) return rc;
    GGNN_CHECK_ARG(act == GGNN_ACT_TANH || act == GGNN_ACT_RELU, "unknown activation %d", act);
    if (V == 0) return GGNN_OK;
    GGNN_CHECK_ARG(rh && u && Wc && bc && h_out, "null pointer");
    GGNN_CHECK_ARG(h_out != h && h_out != rh && h_out != u, "h_out must not alias h, r*h or u");
    GGNN_CHECK_ARG(aligned16(rh) && aligned16(u) && aligned16(Wc) && aligned16(bc) && aligned16(h_out) &&
                   (!save_c || aligned16(save_c)), "pointers must be 16-byte aligned");
    GemmOperands g{};
    for (int s = 0; s < nx; ++s) { g.A[s] = x_segs[s]; g.lda[s] = D; }
    g.A[nx] = rh; g.lda[nx] = D;
    g.nseg = nx + 1; g.D = D; g.M = V;
    g.B = Wc; g.ldb = D; g.b_blk_cols = D; g.b_blk_stride = 0; g.N = D;
    EpiGruCand ec{bc, h, u, h_out, save_c, D, act};
    return dispatch_gemm(g, ec, (hipStream_t)stream);
}

extern "C" int ggnn_gru_f32(const float* const* x_segs, int nx, const float* h, const float* Wg, const float* bg,
                            const float* Wc, const float* bc, float* h_out, void* ws, size_t ws_bytes,
                            float* save_r, float* save_u, float* save_c, int V, int D, int act,
                            ggnn_stream_t stream) {
    if (int rc = gru_args_check(x_segs, nx, h, V, D)) return rc;
    if (V == 0) return GGNN_OK;
    GGNN_CHECK_ARG(ws && aligned16(ws), "workspace null or misaligned");
    if (ws_bytes < ggnn_gru_workspace_bytes(V, D))
        return fail(GGNN_E_WORKSPACE, "GRU workspace too small: %zu < %zu", ws_bytes, ggnn_gru_workspace_bytes(V, D));
    GGNN_CHECK_ARG(act == GGNN_ACT_TANH || act == GGNN_ACT_RELU, "unknown activation %d", act);
    GGNN_CHECK_ARG(Wg && bg && Wc && bc && h_out && h_out != h, "null pointer or h_out aliases h");
    if (gru_fused_supported(D) && nx <= kGruFusedMaxNx) {     // one launch: gates -> r*h -> candidate -> blend chained in registers
        GGNN_CHECK_ARG(aligned16(Wg) && aligned16(bg) && aligned16(Wc) && aligned16(bc) && aligned16(h_out),
                       "pointers must be 16-byte aligned");
        GruFusedArgs a{};
        for (int s = 0; s < nx; ++s) { a.x[s] = x_segs[s]; GGNN_CHECK_ARG(x_segs[s] != h_out, "h_out aliases an input"); }
        a.nx = nx; a.h = h; a.Wg = Wg; a.bg = bg; a.Wc = Wc; a.bc = bc; a.h_out = h_out;
        a.save_r = save_r; a.save_u = save_u; a.save_c = save_c; a.V = V; a.act = act;
        return gru_fused_dispatch(a, D, static_cast<float*>(ws), (hipStream_t)stream);
    }
    float* rh = static_cast<float*>(ws);
    float* u = save_u ? save_u : rh + (size_t)V * D;
    if (int rc = ggnn_gru_gates_f32(x_segs, nx, h, Wg, bg, rh, u, save_r, V, D, stream)) return rc;
    return ggnn_gru_candidate_f32(x_segs, nx, rh, h, u, Wc, bc, h_out, save_c, V, D, act, stream);
}

// ---- pre-packed weights: build the fused kernel's stage images once per weight version ----------------
extern "C" size_t ggnn_gru_packed_bytes(int D, int nx) {
    return (size_t)gru_pack_floats(D, nx) * sizeof(float);
}

static int gru_fmt_check(int gru_fmt) {
    GGNN_CHECK_ARG(gru_fmt == 0 || gru_fmt == GGNN_GRU_FMT_F16X2 || gru_fmt == GGNN_GRU_FMT_BF16X3,
                   "gru_fmt %d is not a GGNN_GRU_FMT_* value", gru_fmt);
    return GGNN_OK;
}

extern "C" int ggnn_gru_pack_weights_f32(const float* Wg, const float* Wc, int nx, int D, int gru_fmt, float* packed,
                                         ggnn_stream_t stream) {
    if (int rc = gru_fmt_check(gru_fmt)) return rc;
    GGNN_CHECK_ARG(nx >= 1 && nx <= 3, "nx %d outside 1..3", nx);
    if (!gru_fused_supported(D)) return fail(GGNN_E_UNSUPPORTED, "no fused GRU (hence no packed weights) for hidden size %d", D);
    GGNN_CHECK_ARG(Wg && Wc && packed && aligned16(packed), "null or misaligned pointer");
    GruFusedArgs a{};
    a.nx = nx; a.Wg = Wg; a.Wc = Wc; a.h = nullptr; a.fmt = gru_fmt;          // h == nullptr: pack only
    return gru_fused_dispatch(a, D, packed, (hipStream_t)stream);
}

extern "C" int ggnn_gru_packed_f32(const float* const* x_segs, int nx, const float* h, const float* packed, const float* bg,
                                   const float* bc, float* h_out, float* save_r, float* save_u, float* save_c, int V, int D,
                                   int act, int gru_fmt, int32_t* tile_counter, ggnn_stream_t stream) {
    if (int rc = gru_args_check(x_segs, nx, h, V, D)) return rc;
    if (int rc = gru_fmt_check(gru_fmt)) return rc;
    GGNN_CHECK_ARG(nx <= kGruFusedMaxNx, "nx %d: the fused GRU takes at most %d input segments (ggnn_gru_f32 takes more)", nx, kGruFusedMaxNx);
    GGNN_CHECK_ARG(act == GGNN_ACT_TANH || act == GGNN_ACT_RELU, "unknown activation %d", act);
    if (!gru_fused_supported(D)) return fail(GGNN_E_UNSUPPORTED, "no fused GRU for hidden size %d", D);
    if (V == 0) return GGNN_OK;
    GGNN_CHECK_ARG(packed && bg && bc && h_out && h_out != h, "null pointer or h_out aliases h");
    GGNN_CHECK_ARG(aligned16(packed) && aligned16(bg) && aligned16(bc) && aligned16(h_out), "pointers must be 16-byte aligned");
    GruFusedArgs a{};
    for (int s = 0; s < nx; ++s) { a.x[s] = x_segs[s]; GGNN_CHECK_ARG(x_segs[s] != h_out, "h_out aliases an input"); }
    a.nx = nx; a.h = h; a.Wg = nullptr; a.Wc = nullptr; a.bg = bg; a.bc = bc; a.h_out = h_out;
    a.save_r = save_r; a.save_u = save_u; a.save_c = save_c; a.V = V; a.act = act; a.tickets = tile_counter; a.fmt = gru_fmt;
    return gru_fused_dispatch(a, D, const_cast<float*>(packed), (hipStream_t)stream);
}

// ---- the other two cell types of chem_tensorflow_sparse.py:102-112 ------------------------------------------
// BasicRNNCell: h' = act([x | h] W + b)
extern "C" int ggnn_rnn_f32(const float* const* x_segs, int nx, const float* h, const float* W, const float* b, float* h_out,
                            int V, int D, int act, ggnn_stream_t stream) {
    if (int rc = gru_args_check(x_segs, nx, h, V, D)) return rc;
    GGNN_CHECK_ARG(act == GGNN_ACT_TANH || act == GGNN_ACT_RELU, "unknown activation %d", act);
    if (V == 0) return GGNN_OK;
    GGNN_CHECK_ARG(W && b && h_out && h_out != h && aligned16(W) && aligned16(b) && aligned16(h_out), "null/misaligned/aliasing pointer");
    GemmOperands g{};
    for (int s = 0; s < nx; ++s) { g.A[s] = x_segs[s]; g.lda[s] = D; }
    g.A[nx] = h; g.lda[nx] = D;
    g.nseg = nx + 1; g.D = D; g.M = V;
    g.B = W; g.ldb = D; g.b_blk_cols = D; g.b_blk_stride = 0; g.N = D;
    EpiBiasAct e{b, h_out, D, act};
    return dispatch_gemm(g, e, (hipStream_t)stream);
}

// CudnnCompatibleGRUCell: [r|u] = sigmoid([x|h] Wg + bg); c = tanh(x Wcx + bcx + r*(h Wch + bch)); h' = u*h + (1-u)*c
// ws: 4*V*D floats (r*h (unused by-product), u, r, h Wch + bch).
extern "C" size_t ggnn_cudnn_gru_workspace_bytes(int V, int D) {
    return V < 0 || D <= 0 ? 0 : (size_t)4 * V * D * sizeof(float);
}

static int cudnn_gru_impl(const float* const* x_segs, int nx, const float* h, const float* Wg, const float* bg,
                          const float* Wcx, const float* bcx, const float* Wch, const float* bch, float* h_out,
                          void* ws, size_t ws_bytes, int V, int D, float* save_c, ggnn_stream_t stream) {
    if (int rc = gru_args_check(x_segs, nx, h, V, D)) return rc;
    if (V == 0) return GGNN_OK;
    GGNN_CHECK_ARG(Wg && bg && Wcx && bcx && Wch && bch && h_out && ws && h_out != h, "null pointer or h_out aliases h");
    GGNN_CHECK_ARG(aligned16(Wg) && aligned16(bg) && aligned16(Wcx) && aligned16(bcx) && aligned16(Wch) && aligned16(bch) &&
                   aligned16(h_out) && aligned16(ws), "pointers must be 16-byte aligned");
    if (ws_bytes < ggnn_cudnn_gru_workspace_bytes(V, D)) return fail(GGNN_E_WORKSPACE, "cudnn-GRU workspace too small");
    const size_t vd = (size_t)V * D;
    float* rh = static_cast<float*>(ws);
    float* u = rh + vd; float* r = u + vd; float* hc = r + vd;
    if (int rc = ggnn_gru_gates_f32(x_segs, nx, h, Wg, bg, rh, u, r, V, D, stream)) return rc;
    hipStream_t st = (hipStream_t)stream;
    {   // hc = h Wch + bch
        GemmOperands g{};
        g.A[0] = h; g.lda[0] = D; g.nseg = 1; g.D = D; g.M = V;
        g.B = Wch; g.ldb = D; g.b_blk_cols = D; g.b_blk_stride = 0; g.N = D;
        EpiBiasAct e{bch, hc, D, GGNN_ACT_NONE};
        if (int rc = dispatch_gemm(g, e, st)) return rc;
    }
    GemmOperands g{};
    for (int s = 0; s < nx; ++s) { g.A[s] = x_segs[s]; g.lda[s] = D; }
    g.nseg = nx; g.D = D; g.M = V;
    g.B = Wcx; g.ldb = D; g.b_blk_cols = D; g.b_blk_stride = 0; g.N = D;
    EpiCudnnCand e{bcx, r, hc, h, u, h_out, D, save_c};
    return dispatch_gemm(g, e, st);
}

extern "C" int ggnn_cudnn_gru_f32(const float* const* x_segs, int nx, const float* h, const float* Wg, const float* bg,
                                  const float* Wcx, const float* bcx, const float* Wch, const float* bch, float* h_out,
                                  void* ws, size_t ws_bytes, int V, int D, ggnn_stream_t stream) {
    return cudnn_gru_impl(x_segs, nx, h, Wg, bg, Wcx, bcx, Wch, bch, h_out, ws, ws_bytes, V, D, nullptr, stream);
}

// Training form: also keeps the candidate c; after the call ws holds [r*h | u | r | h Wch + bch] ([V,D] each) for the backward pass.
extern "C" int ggnn_cudnn_gru_train_f32(const float* const* x_segs, int nx, const float* h, const float* Wg, const float* bg,
                                        const float* Wcx, const float* bcx, const float* Wch, const float* bch, float* h_out,
                                        float* save_c, void* ws, size_t ws_bytes, int V, int D, ggnn_stream_t stream) {
    GGNN_CHECK_ARG(save_c && aligned16(save_c), "save_c null or misaligned");
    return cudnn_gru_impl(x_segs, nx, h, Wg, bg, Wcx, bcx, Wch, bch, h_out, ws, ws_bytes, V, D, save_c, stream);
}

// General dX product of the backward pass with its epilogue (EpiBwdDx): Q = dY WT, dY [V, nseg_y * D] given as nseg_y column
// segments of one row-major matrix (row stride ldy), WT [nseg_y * D, K] row-major.  Columns [0, xcols) of Q are the x segments
// (xcols = nx * D; with split_inc the last D of them are the aggregated messages -> dinc, divided by the in-degree with
// use_avg; without it all of them go to dx); columns
// [xcols, K) (K == xcols + D, or K == xcols: no h block) -> dh.  acc_dx / acc_dh: add to the existing dx / dh instead of writing.
extern "C" int ggnn_bwd_dx_f32(const float* dY, int ldy, int nseg_y, const float* WT, int K, float* dx, int xcols, int split_inc,
                               float* dinc, const float* nin, int T, int use_avg, float* dh, int acc_dx, int acc_dh, int V, int D,
                               ggnn_stream_t stream) {
    if (int rc = check_common(V, D)) return rc;
    GGNN_CHECK_ARG(nseg_y >= 1 && nseg_y <= 4 && ldy >= nseg_y * D && ldy % 4 == 0, "bad dY layout");
    GGNN_CHECK_ARG(xcols >= 0 && xcols % D == 0 && (K == xcols || K == xcols + D) && K > 0, "K / xcols mismatch");
    if (V == 0) return GGNN_OK;
    const bool inc = split_inc && xcols > 0;
    GGNN_CHECK_ARG(dY && WT && (K == xcols || dh) && (!inc || dinc) && ((inc && xcols <= D && !acc_dx) || xcols == 0 || dx) &&
                   (!(inc && use_avg) || nin), "null pointer");
    GGNN_CHECK_ARG(aligned16(dY) && aligned16(WT), "pointers must be 16-byte aligned");
    GemmOperands g{};
    for (int s = 0; s < nseg_y; ++s) { g.A[s] = dY + (size_t)s * D; g.lda[s] = ldy; }
    g.nseg = nseg_y; g.D = D; g.M = V;
    g.B = WT; g.ldb = K; g.b_blk_cols = K; g.b_blk_stride = 0; g.N = K;
    const int inc0 = inc ? xcols - D : xcols;                // first column of the aggregated-messages segment (xcols: none)
    EpiBwdDx e{dx, xcols > 0 ? xcols : 4, xcols, inc0, dinc, nin, T, use_avg, dh, D, acc_dx, acc_dh};   // dx: [V, xcols]
    return dispatch_gemm(g, e, (hipStream_t)stream);
}

// GRU with the segment sum fused in: the aggregated-messages input (the LAST x segment) is not passed but gathered
// inside the kernel from the transformed-state rows.  x_segs holds the nx-1 residual segments (may be NULL for nx = 1).
extern "C" int ggnn_gru_packed_gather_f32(const float* const* x_segs, int nx, const float* h, const float* packed,
                                          const float* bg, const float* bc, float* h_out, const float* Hrows,
                                          const int32_t* row_ptr, const int32_t* gather_row, const float* nin, int T, int use_avg,
                                          int V, int D, int act, int gru_fmt, int32_t* tile_counter, ggnn_stream_t stream) {
    return ggnn_gru_packed_gather_train_f32(x_segs, nx, h, packed, bg, bc, h_out, Hrows, row_ptr, gather_row, nin, T, use_avg,
                                            nullptr, nullptr, nullptr, nullptr, V, D, act, gru_fmt, tile_counter, stream);
}

// ... the training form: also writes r, u, c and the gathered segment `incoming` (all four or none) for the backward pass
extern "C" int ggnn_gru_packed_gather_train_f32(const float* const* x_segs, int nx, const float* h, const float* packed,
                                                const float* bg, const float* bc, float* h_out, const float* Hrows,
                                                const int32_t* row_ptr, const int32_t* gather_row, const float* nin, int T,
                                                int use_avg, float* save_r, float* save_u, float* save_c, float* save_incoming,
                                                int V, int D, int act, int gru_fmt, int32_t* tile_counter, ggnn_stream_t stream) {
    if (int rc = check_common(V, D)) return rc;
    if (int rc = gru_fmt_check(gru_fmt)) return rc;
    const bool save = save_r || save_u || save_c || save_incoming;
    GGNN_CHECK_ARG(!save || (save_r && save_u && save_c && save_incoming && aligned16(save_r) && aligned16(save_u) && aligned16(save_c) &&
                             aligned16(save_incoming)), "save_r / save_u / save_c / save_incoming must be given together, 16-byte aligned");
    GGNN_CHECK_ARG(nx >= 1 && nx <= 3, "nx %d outside 1..3", nx);
    GGNN_CHECK_ARG(act == GGNN_ACT_TANH || act == GGNN_ACT_RELU, "unknown activation %d", act);
    if (!gru_fused_supported(D)) return fail(GGNN_E_UNSUPPORTED, "no fused GRU for hidden size %d", D);
    if (V == 0) return GGNN_OK;
    GGNN_CHECK_ARG(h && packed && bg && bc && h_out && h_out != h && Hrows && row_ptr && T > 0, "null pointer or h_out aliases h");
    GGNN_CHECK_ARG(!use_avg || nin, "nin is required for mean aggregation");
    GGNN_CHECK_ARG(nx == 1 || x_segs, "residual segments missing");
    GGNN_CHECK_ARG(aligned16(h) && aligned16(packed) && aligned16(bg) && aligned16(bc) && aligned16(h_out) && aligned16(Hrows),
                   "pointers must be 16-byte aligned");
    GruFusedArgs a{};
    for (int s = 0; s + 1 < nx; ++s) {
        GGNN_CHECK_ARG(x_segs[s] && aligned16(x_segs[s]) && x_segs[s] != h_out, "residual segment %d null, misaligned or aliasing h_out", s);
        a.x[s] = x_segs[s];
    }
    a.nx = nx; a.h = h; a.bg = bg; a.bc = bc; a.h_out = h_out; a.V = V; a.act = act;
    a.save_r = save_r; a.save_u = save_u; a.save_c = save_c; a.save_x = save_incoming;
    a.g_H = Hrows; a.g_row_ptr = row_ptr; a.g_idx = gather_row; a.g_nin = nin; a.g_T = T; a.g_use_avg = use_avg;
    a.tickets = tile_counter; a.fmt = gru_fmt;
    return gru_fused_dispatch(a, D, const_cast<float*>(packed), (hipStream_t)stream);
}

// ---- backward dX products with fused GRU gate algebra (epilogues in ggnn_gemm.hpp) ------------------------------------------
// WcT: Wc^T [D, (nx+1) D] row-major; dpc [V,D].  dx [V, nx*D] written; dh [V,D] += drh*r; dpg[:, 0:D] = drh*h*r*(1-r).
extern "C" int ggnn_gru_bwd_dx_cand_f32(const float* dpc, const float* WcT, const float* h, const float* r, float* dx, float* dh,
                                        float* dpg, int nx, int V, int D, ggnn_stream_t stream) {
    if (int rc = check_common(V, D)) return rc;
    GGNN_CHECK_ARG(nx >= 1 && nx <= kGemmMaxSeg - 1, "nx %d outside 1..%d", nx, kGemmMaxSeg - 1);
    if (V == 0) return GGNN_OK;
    GGNN_CHECK_ARG(dpc && WcT && h && r && dx && dh && dpg, "null pointer");
    GGNN_CHECK_ARG(aligned16(dpc) && aligned16(WcT) && aligned16(h) && aligned16(r) && aligned16(dx) && aligned16(dh) && aligned16(dpg),
                   "pointers must be 16-byte aligned");
    const int K = (nx + 1) * D;
    GemmOperands g{};
    g.A[0] = dpc; g.lda[0] = D; g.nseg = 1; g.D = D; g.M = V;
    g.B = WcT; g.ldb = K; g.b_blk_cols = K; g.b_blk_stride = 0; g.N = K;
    EpiBwdCand e{dx, nx * D, nx * D, h, r, dh, dpg, D};
    return dispatch_gemm(g, e, (hipStream_t)stream);
}

// WgT: Wg^T [2D, (nx+1) D]; dpg [V,2D].  dx[:, residual columns] += ; dinc [V,D] = (dx[:, last segment] + Q) (/ (deg + 1e-7));
// dh += Q[:, h columns].
extern "C" int ggnn_gru_bwd_dx_gates_f32(const float* dpg, const float* WgT, float* dx, float* dinc, const float* nin, int T,
                                         int use_avg, float* dh, int nx, int V, int D, ggnn_stream_t stream) {
    if (int rc = check_common(V, D)) return rc;
    GGNN_CHECK_ARG(nx >= 1 && nx <= kGemmMaxSeg - 1, "nx %d outside 1..%d", nx, kGemmMaxSeg - 1);
    if (V == 0) return GGNN_OK;
    GGNN_CHECK_ARG(dpg && WgT && dx && dinc && dh && (!use_avg || nin), "null pointer");
    GGNN_CHECK_ARG(aligned16(dpg) && aligned16(WgT) && aligned16(dx) && aligned16(dinc) && aligned16(dh), "pointers must be 16-byte aligned");
    const int K = (nx + 1) * D;
    GemmOperands g{};
    g.A[0] = dpg; g.lda[0] = 2 * D; g.A[1] = dpg + D; g.lda[1] = 2 * D; g.nseg = 2; g.D = D; g.M = V;
    g.B = WgT; g.ldb = K; g.b_blk_cols = K; g.b_blk_stride = 0; g.N = K;
    EpiBwdGates e{dx, nx * D, nx * D, (nx - 1) * D, dinc, nin, T, use_avg, dh, D};
    return dispatch_gemm(g, e, (hipStream_t)stream);
}
