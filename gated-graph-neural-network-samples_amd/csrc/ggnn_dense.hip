// Dense-adjacency aggregation of chem_tensorflow_dense.py:103-112:
//     acts[g,i,:] = sum_e sum_j A[g,e,i,j] * ( (h W_e)[g,j,:] + b_e )
// (A[g,e,dst,src] in {0,1}, chem_tensorflow_dense.py:30-36; the bias is added to EVERY row of m_e,
// padded vertices included, before A_e is applied, :107-108).
//
// The two dense contractions around it run on the MFMA kernels (h W_e for all e = ggnn_msg_transform_f32,
// the GRU = ggnn_gru_f32); this batched [v,v]x[v,D] product is ~10 % of the step's flops at v <= 29 and its
// operand is a 0/1 matrix with ~2 non-zeros per row, so it is done from LDS on the vector ALUs, one
// workgroup per graph, skipping zero entries (wave-uniform per output row).
#include "ggnn_common.h"

namespace ggnn {

// LDS: A_g [e][v][v] floats, then H_g rows [jc][e*D] floats (jc = rows per chunk)
__global__ __launch_bounds__(256) void dense_aggregate_kernel(const float* __restrict__ A, const float* __restrict__ H,
                                                              const float* __restrict__ bias, float* __restrict__ acts,
                                                              int v, int E, int D, int jc) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int g = blockIdx.x;
    const int D4 = D >> 2, ED = E * D, ED4 = ED >> 2;
    const int a_elems = (E * v * v + 3) / 4 * 4;
    float* As = smem;
    float* Hs = smem + a_elems;
    const float* Ag = A + (size_t)g * E * v * v;
    for (int i = threadIdx.x; i < E * v * v; i += blockDim.x) As[i] = Ag[i];

    const int nout = v * D4;                       // float4 outputs of this graph
    // each thread keeps up to 4 outputs (v*D/4 <= 1024 for the chem buckets); larger graphs loop
    for (int o0 = 0; o0 < nout; o0 += 4 * blockDim.x) {
        f32x4 acc[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int j0 = 0; j0 < v; j0 += jc) {
            const int jn = min(jc, v - j0);
            __syncthreads();                        // previous chunk consumed / As written
            const f32x4* Hg = reinterpret_cast<const f32x4*>(H + ((size_t)g * v + j0) * ED);
            for (int i = threadIdx.x; i < jn * ED4; i += blockDim.x) reinterpret_cast<f32x4*>(Hs)[i] = Hg[i];
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int o = o0 + q * blockDim.x + threadIdx.x;
                if (o < nout) {
                    const int i = o / D4, c4 = o - i * D4;
                    for (int e = 0; e < E; ++e) {
                        const float* arow = As + (e * v + i) * v + j0;
                        f32x4 b = {0.f, 0.f, 0.f, 0.f};
                        if (bias) b = *reinterpret_cast<const f32x4*>(bias + (size_t)e * D + 4 * c4);
                        for (int j = 0; j < jn; ++j) {
                            const float a = arow[j];
                            if (a != 0.f)
                                acc[q] += a * (*reinterpret_cast<const f32x4*>(Hs + (size_t)j * ED + e * D + 4 * c4) + b);
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int o = o0 + q * blockDim.x + threadIdx.x;
            if (o < nout) {
                const int i = o / D4, c4 = o - i * D4;
                *reinterpret_cast<f32x4*>(acts + ((size_t)g * v + i) * D + 4 * c4) = acc[q];
            }
        }
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
    const size_t a_bytes = (size_t)((E * v * v + 3) / 4 * 4) * sizeof(float);
    const size_t row_bytes = (size_t)E * D * sizeof(float);
    const size_t budget = 64 * 1024;
    if (a_bytes + row_bytes > budget)
        return fail(GGNN_E_UNSUPPORTED, "dense aggregate: e*v*v + e*D too large for LDS (v=%d e=%d D=%d)", v, E, D);
    int jc = (int)((budget - a_bytes) / row_bytes);
    if (jc > v) jc = v;
    const size_t lds = a_bytes + (size_t)jc * row_bytes;
    hipLaunchKernelGGL(dense_aggregate_kernel, dim3(b), dim3(256), lds, (hipStream_t)stream, A, H, bias, acts, v, E, D, jc);
    GGNN_CHECK_HIP(hipGetLastError());
    return GGNN_OK;
}
