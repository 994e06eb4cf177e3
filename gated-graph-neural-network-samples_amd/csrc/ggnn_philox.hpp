// Philox4x32-10 (Salmon et al., SC'11; Random123) -- the counter-based generator behind ggnn_dropout_f32 and the masked weight
// images of the native training step.  oracle/ggnn_oracle.py::philox4x32_10 is its NumPy twin (Random123's known answers).
#pragma once
#include <cstdint>
#include <hip/hip_runtime.h>

namespace ggnn {

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                              uint32_t out[4]) {
#pragma unroll
    for (int round = 0; round < 10; ++round) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// tf.nn.dropout's factor for element (row key, column) under (seed, keep): x / keep * floor(keep + U) == x * dropout_keep(...) / keep
// is NOT how it is applied (the division comes first); callers use dropout_apply.
__device__ __forceinline__ float dropout_apply(float x, float keep, uint64_t seed, uint64_t row_key, int col) {
    uint32_t u[4];
    philox4x32_10((uint32_t)row_key, (uint32_t)(row_key >> 32), (uint32_t)(col >> 2), 0u, (uint32_t)seed, (uint32_t)(seed >> 32), u);
    return x / keep * floorf(keep + (float)(u[col & 3] >> 8) * (1.0f / 16777216.0f));
}

}  // namespace ggnn
