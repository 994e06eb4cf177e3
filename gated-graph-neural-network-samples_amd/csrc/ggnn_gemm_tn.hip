// Weight-gradient product of the backward pass (TF autodiff of tf.matmul, chem_tensorflow.py:184):
//     C[K,N] = A[M,K]^T B[M,N]        M ~ 1e5 rows (nodes), K, N = 100 .. 400
// -- a reduction over all M rows into a small matrix.  Vendor BLAS runs this shape on ceil(K/64)*ceil(N/64) ~ 12
// workgroups (no split along M); here the rows are split over the whole GPU:
//   grid = (K/16 rounded up to 4-tile groups) x S row ranges; a workgroup of 4 waves owns 4 consecutive 16-column tiles
//   of A^T (wave w: tile w) x ALL N columns, and walks its row range in steps of 4 rows (one v_mfma_f32_16x16x4_f32 per
//   output tile and step).  The A operand is read straight from global memory (lane (i, k): A[r0+k][k0+i], 64 B runs);
//   the B rows are shared by the 4 waves and go through a double-buffered 16-row LDS slab.
//   Each workgroup writes its partial [64, N] block; a second kernel adds the S partials in a FIXED order
//   (deterministic -- no atomics).
#include "ggnn_common.h"

namespace ggnn {

constexpr int kTnWaves = 4;        // waves per workgroup = 16-column tiles of A^T per workgroup
constexpr int kTnSlab = 16;        // rows per LDS slab (4 MFMA steps)

template <int NT>                   // 16-column tiles of B (N <= 16*NT)
__global__ __launch_bounds__(kTnWaves * 64) void gemm_tn_kernel(const float* __restrict__ A, int lda, const float* __restrict__ B,
                                                                int ldb, float* __restrict__ part, int M, int K, int N,
                                                                int rows_per_split) {
    extern __shared__ __attribute__((aligned(16))) float slab[];            // [2][kTnSlab][NP], NP = 16*NT + 4 (pad)
    constexpr int NP = 16 * NT + 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kq = lane >> 4;
    const int kgrp = blockIdx.x, split = blockIdx.y;
    const int k0 = (kgrp * kTnWaves + wave) * 16;                            // this wave's columns of A
    const int r_beg = split * rows_per_split;
    const int r_end = min(M, r_beg + rows_per_split);
    const bool kcol_ok = k0 + li < K;

    f32x4 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto stage = [&](int buf, int r0) {                                      // B rows r0 .. r0+15 -> slab[buf] (zero beyond M / N)
        float* dst = slab + buf * kTnSlab * NP;
        for (int i = tid; i < kTnSlab * (NP / 4); i += kTnWaves * 64) {
            const int rr = i / (NP / 4), c4 = i % (NP / 4);
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (r0 + rr < r_end && 4 * c4 < N) v = *reinterpret_cast<const f32x4*>(B + (size_t)(r0 + rr) * ldb + 4 * c4);
            *reinterpret_cast<f32x4*>(dst + rr * NP + 4 * c4) = v;
        }
    };
    auto load_a = [&](float (&a)[4], int r0) {                               // 4 steps x (row r0 + 4s + kq, column k0 + li)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int r = r0 + 4 * s + kq;
            a[s] = (kcol_ok && r < r_end) ? A[(size_t)r * lda + k0 + li] : 0.f;
        }
    };

    float a_cur[4], a_nxt[4];
    int buf = 0;
    if (r_beg < r_end) { stage(0, r_beg); load_a(a_cur, r_beg); }
    __syncthreads();
    for (int r0 = r_beg; r0 < r_end; r0 += kTnSlab) {
        const bool more = r0 + kTnSlab < r_end;
        if (more) { stage(buf ^ 1, r0 + kTnSlab); load_a(a_nxt, r0 + kTnSlab); }
        const float* sb = slab + buf * kTnSlab * NP;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const float* brow = sb + (4 * s + kq) * NP + li;
            float b[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) b[nt] = brow[16 * nt];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[s], b[nt], acc[nt], 0, 0, 0);
        }
        __syncthreads();                                                     // slab[buf] consumed, slab[buf^1] written
        buf ^= 1;
#pragma unroll
        for (int s = 0; s < 4; ++s) a_cur[s] = a_nxt[s];
    }
    // accumulator tile nt: lane (li, kq) holds C[k0 + 4*kq + e][16*nt + li], e = 0..3   (M index = first operand)
    float* out = part + (size_t)split * K * N;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n = 16 * nt + li;
        if (n < N) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = k0 + 4 * kq + e;
                if (k < K) out[(size_t)k * N + n] = acc[nt][e];
            }
        }
    }
}

// C[i] = sum over the S partials, in order
__global__ void gemm_tn_reduce_kernel(const float* __restrict__ part, float* __restrict__ C, int KN, int S) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= KN) return;
    float s = 0.f;
    for (int p = 0; p < S; ++p) s += part[(size_t)p * KN + i];
    C[i] = s;
}

static int tn_splits(int M, int K) {
    const int kgroups = (K + 16 * kTnWaves - 1) / (16 * kTnWaves);
    int S = (3 * num_cus() + kgroups - 1) / kgroups;                         // ~3 workgroups per CU
    const int max_s = (M + 4 * kTnSlab - 1) / (4 * kTnSlab);                  // at least 4 slabs per split
    if (S > max_s) S = max_s;
    return S < 1 ? 1 : S;
}

template <int NT>
static int launch_tn(const float* A, int lda, const float* B, int ldb, float* C, float* part, int M, int K, int N, int S,
                     hipStream_t st) {
    const int kgroups = (K + 16 * kTnWaves - 1) / (16 * kTnWaves);
    int rows = (M + S - 1) / S;
    rows = (rows + kTnSlab - 1) / kTnSlab * kTnSlab;
    const size_t lds = (size_t)2 * kTnSlab * (16 * NT + 4) * sizeof(float);
    static std::atomic<unsigned long long> lds_ok{0};
    if (lds > 64 * 1024) GGNN_CHECK_HIP(allow_dynamic_lds(&gemm_tn_kernel<NT>, lds, lds_ok));
    hipLaunchKernelGGL((gemm_tn_kernel<NT>), dim3(kgroups, S), dim3(kTnWaves * 64), lds, st, A, lda, B, ldb, part, M, K, N, rows);
    GGNN_CHECK_HIP(hipGetLastError());
    hipLaunchKernelGGL(gemm_tn_reduce_kernel, dim3((K * N + 255) / 256), dim3(256), 0, st, (const float*)part, C, K * N, S);
    GGNN_CHECK_HIP(hipGetLastError());
    return GGNN_OK;
}

}  // namespace ggnn

using namespace ggnn;

extern "C" size_t ggnn_gemm_tn_workspace_bytes(int M, int K, int N) {
    if (M <= 0 || K <= 0 || N <= 0) return 256;
    return (size_t)tn_splits(M, K) * K * N * sizeof(float) + 256;
}

extern "C" int ggnn_gemm_tn_f32(const float* A, int lda, const float* B, int ldb, float* C, int M, int K, int N, void* ws,
                                size_t ws_bytes, ggnn_stream_t stream) {
    GGNN_CHECK_ARG(M >= 0 && K > 0 && N > 0 && N % 4 == 0 && N <= 512, "bad sizes M=%d K=%d N=%d (N a multiple of 4, <= 512)", M, K, N);
    GGNN_CHECK_ARG(lda >= K && ldb >= N && ldb % 4 == 0, "lda %d / ldb %d too small or ldb not a multiple of 4", lda, ldb);
    GGNN_CHECK_ARG(C, "null pointer");
    hipStream_t st = (hipStream_t)stream;
    if (M == 0) {
        GGNN_CHECK_HIP(hipMemsetAsync(C, 0, (size_t)K * N * sizeof(float), st));
        return GGNN_OK;
    }
    GGNN_CHECK_ARG(A && B && ws && aligned16(B) && aligned16(ws), "null or misaligned pointer");
    if (ws_bytes < ggnn_gemm_tn_workspace_bytes(M, K, N))
        return fail(GGNN_E_WORKSPACE, "gemm_tn workspace too small: %zu < %zu", ws_bytes, ggnn_gemm_tn_workspace_bytes(M, K, N));
    float* part = static_cast<float*>(ws);
    const int S = tn_splits(M, K);
    const int nt = (N + 15) / 16;
    if (nt <= 7) return launch_tn<7>(A, lda, B, ldb, C, part, M, K, N, S, st);
    if (nt <= 13) return launch_tn<13>(A, lda, B, ldb, C, part, M, K, N, S, st);
    if (nt <= 19) return launch_tn<19>(A, lda, B, ldb, C, part, M, K, N, S, st);
    if (nt <= 25) return launch_tn<25>(A, lda, B, ldb, C, part, M, K, N, S, st);
    return launch_tn<32>(A, lda, B, ldb, C, part, M, K, N, S, st);
}
