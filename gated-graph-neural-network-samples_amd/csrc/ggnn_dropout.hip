// tf.nn.dropout (chem_tensorflow_sparse.py:91,113-114; chem_tensorflow_dense.py:104; utils.py:68) with a COUNTER-BASED mask.
//
//   out[r, c] = x[r, c] / keep * floor(keep + U(seed, key(r), c))          (nn_ops.py dropout @ TF r1.3)
//
// TensorFlow draws U from the graph-seeded Philox stream, which is neither reproducible outside TF nor shareable between
// processes.  Here U is a pure function of (seed, row key, column): Philox4x32-10 keyed by the 64-bit `seed`, counter =
// (row key lo, row key hi, column quad, 0), the four outputs feeding columns 4q .. 4q+3.  Consequences:
//   * data parallelism: every rank derives the SAME weight masks from (random_seed, optimisation step, site) without any
//     generator state that other draws could de-synchronise (round-2 review: a [V,D] state-dropout draw shifted the default
//     generator by a rank-dependent amount);
//   * a node's state-dropout mask depends on the node's identity (row_key = dataset graph id, node within graph), not on
//     the batch it was packed into: a sharded step equals the single-process step on the union batch;
//   * the backward pass re-derives the mask (no [V,D] mask tensor is kept): dx = dropout(dy) with the same arguments.
#include "ggnn_common.h"
#include "ggnn_philox.hpp"

namespace ggnn {
namespace {

// one thread per (row, column quad); a quad's four columns are contiguous -> 16-byte accesses when cols % 4 == 0
template <bool VEC>
__global__ void __launch_bounds__(256) dropout_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                      const int64_t* __restrict__ row_key, int64_t row_key_base, uint32_t k0,
                                                      uint32_t k1, float keep, int64_t rows, int cols, int quads) {
    const int64_t total = rows * (int64_t)quads;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / quads;
        const int q = (int)(i - r * quads);
        const uint64_t key = (uint64_t)(row_key ? row_key[r] : row_key_base + r);
        uint32_t u[4];
        philox4x32_10((uint32_t)key, (uint32_t)(key >> 32), (uint32_t)q, 0u, k0, k1, u);
        const int64_t off = r * (int64_t)cols + 4 * q;
        if (VEC) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(x + off);
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = v[e] / keep * floorf(keep + (float)(u[e] >> 8) * (1.0f / 16777216.0f));
            *reinterpret_cast<f32x4*>(out + off) = o;
        } else {
            const int n = min(4, cols - 4 * q);
            for (int e = 0; e < n; ++e) out[off + e] = x[off + e] / keep * floorf(keep + (float)(u[e] >> 8) * (1.0f / 16777216.0f));
        }
    }
}

}  // namespace
}  // namespace ggnn

extern "C" int ggnn_dropout_f32(const float* x, float* out, const int64_t* row_key, int64_t row_key_base, uint64_t seed,
                                float keep_prob, int64_t rows, int cols, ggnn_stream_t stream) {
    using namespace ggnn;
    GGNN_CHECK_ARG(rows >= 0 && cols >= 0, "ggnn_dropout_f32: negative size");
    GGNN_CHECK_ARG(keep_prob > 0.0f && keep_prob <= 1.0f, "ggnn_dropout_f32: keep_prob %g outside (0, 1]", (double)keep_prob);
    if (rows == 0 || cols == 0) return GGNN_OK;
    GGNN_CHECK_ARG(x && out, "ggnn_dropout_f32: null pointer");
    const int quads = (cols + 3) / 4;
    const int64_t total = rows * (int64_t)quads;
    const int blocks = (int)std::min<int64_t>((total + 255) / 256, (int64_t)num_cus() * 16);
    const bool vec = cols % 4 == 0 && aligned16(x) && aligned16(out);
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (vec)
        hipLaunchKernelGGL(dropout_kernel<true>, dim3(blocks), dim3(256), 0, st, x, out, row_key, row_key_base, k0, k1, keep_prob, rows,
                           cols, quads);
    else
        hipLaunchKernelGGL(dropout_kernel<false>, dim3(blocks), dim3(256), 0, st, x, out, row_key, row_key_base, k0, k1, keep_prob, rows,
                           cols, quads);
    GGNN_CHECK_HIP(hipGetLastError());
    return GGNN_OK;
}
