// Dense-adjacency aggregation of chem_tensorflow_dense.py:103-112:
//     acts[g,i,:] = sum_e sum_j A[g,e,i,j] * ( (h W_e)[g,j,:] + b_e )
// (A[g,e,dst,src] in {0,1}, chem_tensorflow_dense.py:30-36; the bias is added to EVERY row of m_e,
// padded vertices included, before A_e is applied, :107-108).
//
// The two dense contractions around it run on the MFMA kernels (h W_e for all e = ggnn_msg_transform_f32,
// the GRU = ggnn_gru_f32); this batched [v,v]x[v,D] product is ~10 % of the step's flops at v <= 29 and its
// operand is a 0/1 matrix with ~2 non-zeros per row.  One WAVE per output row (g,i): the row A[g,e,i,:] is read
// with one coalesced load per 64 columns, its non-zeros are found with a ballot, and only the matching rows of
// (h W_e) are fetched -- straight from L2 / HBM, 16 B per lane, no LDS staging of rows that are never used.
// Accumulation order (e ascending, j ascending) is that of the reference's batched matmul + reduce over e.
#include "ggnn_common.h"

namespace ggnn {

constexpr int kDenseWaves = 4;     // output rows per workgroup

__global__ __launch_bounds__(kDenseWaves * 64) void dense_aggregate_kernel(const float* __restrict__ A, const float* __restrict__ H,
                                                                           const float* __restrict__ bias, float* __restrict__ acts,
                                                                           int rows, int v, int E, int D) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * kDenseWaves + (threadIdx.x >> 6);          // wave-uniform
    if (row >= rows) return;
    const int g = row / v, i = row - g * v;
    const int D4 = D >> 2;
    for (int c0 = 0; c0 < D4; c0 += 64) {                                    // (one trip for D <= 256)
        const int c4 = c0 + lane;
        const bool on = c4 < D4;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int e = 0; e < E; ++e) {
            const float* arow = A + (((size_t)g * E + e) * v + i) * v;
            f32x4 b = {0.f, 0.f, 0.f, 0.f};
            if (bias && on) b = *reinterpret_cast<const f32x4*>(bias + (size_t)e * D + 4 * c4);
            for (int j0 = 0; j0 < v; j0 += 64) {
                const float a = (j0 + lane < v) ? arow[j0 + lane] : 0.f;
                unsigned long long nz = __ballot(a != 0.f);
                while (nz) {                                                 // wave-uniform walk over the non-zeros
                    const int j = __builtin_ctzll(nz);
                    nz &= nz - 1;
                    const float aj = __shfl(a, j);
                    if (on) {
                        const f32x4 hv = *reinterpret_cast<const f32x4*>(H + (((size_t)g * v + j0 + j) * E + e) * D + 4 * c4);
                        acc += aj * (hv + b);
                    }
                }
            }
        }
        if (on) *reinterpret_cast<f32x4*>(acts + (size_t)row * D + 4 * c4) = acc;
    }
}

}  // namespace ggnn

using namespace ggnn;

extern "C" int ggnn_dense_aggregate_f32(const float* A, const float* H, const float* bias, float* acts, int b, int v,
                                        int E, int D, ggnn_stream_t stream) {
    GGNN_CHECK_ARG(b >= 0 && v > 0 && E > 0 && D > 0 && D % 4 == 0, "bad sizes b=%d v=%d e=%d D=%d", b, v, E, D);
    if (b == 0) return GGNN_OK;
    GGNN_CHECK_ARG(A && H && acts, "null pointer");
    GGNN_CHECK_ARG(aligned16(H) && aligned16(acts) && (!bias || aligned16(bias)), "pointers must be 16-byte aligned");
    GGNN_CHECK_ARG((long long)b * v < (1LL << 31), "b*v too large");
    const int rows = b * v;
    hipLaunchKernelGGL(dense_aggregate_kernel, dim3((rows + kDenseWaves - 1) / kDenseWaves), dim3(kDenseWaves * 64), 0,
                       (hipStream_t)stream, A, H, bias, acts, rows, v, E, D);
    GGNN_CHECK_HIP(hipGetLastError());
    return GGNN_OK;
}
