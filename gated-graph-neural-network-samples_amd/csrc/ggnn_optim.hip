// The optimiser side of the reference's train step (chem_tensorflow.py:183-191) in two launches for ALL variables:
//   per-variable tf.clip_by_norm:  g <- g * clip / max(||g||_2, clip)
//   tf.train.AdamOptimizer (TF-1.3 ApplyAdam):  m <- b1 m + (1-b1) g;  v <- b2 v + (1-b2) g^2;  p <- p - lr_t m / (sqrt(v) + eps)
// over FLAT gradient / slot buffers in which every variable starts on a 1024-float boundary, so that a block of 1024 floats
// belongs to exactly one variable (the parameters themselves stay in the model's own tensors, reached through a pointer table).  (As ~60 torch multi-tensor / element-wise launches this was ~0.3 ms of a
// 8 ms training step.)  Deterministic: fixed-order reductions.
#include "ggnn_common.h"

namespace ggnn {

constexpr int kOptBlock = 1024;          // floats per block (256 threads x float4)

// partial[b] = sum of squares of block b of the flat gradient
__global__ __launch_bounds__(256) void sumsq_blocks_kernel(const f32x4* __restrict__ g, float* __restrict__ partial) {
    __shared__ float red[256];
    const f32x4 v = g[(size_t)blockIdx.x * 256 + threadIdx.x];
    red[threadIdx.x] = (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

// block b belongs to variable block_var[b], whose blocks are var_first[t] .. var_first[t+1]-1; var_active[t] == 0: no gradient
__global__ __launch_bounds__(256) void clip_adam_kernel(float* const* __restrict__ p_ptr, const int* __restrict__ var_numel,
                                                        const f32x4* __restrict__ g, f32x4* __restrict__ m,
                                                        f32x4* __restrict__ v, const float* __restrict__ partial,
                                                        const int* __restrict__ block_var, const int* __restrict__ var_first,
                                                        const int* __restrict__ var_active, float clip, float lr_t, float b1,
                                                        float b2, float eps) {
    __shared__ float red[256];
    const int t = block_var[blockIdx.x];
    if (!var_active[t]) return;
    const int b0 = var_first[t], b1i = var_first[t + 1];
    float s = 0.f;
    for (int b = b0 + (int)threadIdx.x; b < b1i; b += 256) s += partial[b];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    const float norm = sqrtf(red[0]);
    const float scale = clip > 0.f ? clip / fmaxf(norm, clip) : 1.0f;          // chem_tensorflow.py:186-190
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const f32x4 gv = g[i] * scale;
    const f32x4 mv = m[i] * b1 + gv * (1.0f - b1);
    const f32x4 vv = v[i] * b2 + gv * gv * (1.0f - b2);
    m[i] = mv; v[i] = vv;
    f32x4 d;
    d.x = mv.x / (sqrtf(vv.x) + eps); d.y = mv.y / (sqrtf(vv.y) + eps);
    d.z = mv.z / (sqrtf(vv.z) + eps); d.w = mv.w / (sqrtf(vv.w) + eps);
    // the parameters stay where the model created them (one tensor per variable, 16-byte aligned): element `local` of variable t
    float* pp = p_ptr[t];
    const int local = ((int)blockIdx.x - b0) * kOptBlock + 4 * (int)threadIdx.x, n = var_numel[t];
    if (local + 3 < n) {
        f32x4* q = reinterpret_cast<f32x4*>(pp + local);
        *q = *q - lr_t * d;
    } else {
        const float dd[4] = {d.x, d.y, d.z, d.w};
        for (int e = 0; e < 4; ++e)
            if (local + e < n) pp[local + e] -= lr_t * dd[e];
    }
}

}  // namespace ggnn

using namespace ggnn;

extern "C" int ggnn_optim_block_floats(void) { return kOptBlock; }

// grads / m / v: flat buffers of nblocks * 1024 floats; param_ptrs: DEVICE array of the variables' base pointers (16-byte
// aligned); partial: nblocks floats of scratch; tables: device int32.
extern "C" int ggnn_clip_adam_f32(float* const* param_ptrs, const int32_t* var_numel, const float* grads, float* m, float* v,
                                  float* partial, const int32_t* block_var, const int32_t* var_first, const int32_t* var_active,
                                  int nblocks, float clip_norm, float lr_t, float beta1, float beta2, float epsilon,
                                  ggnn_stream_t stream) {
    GGNN_CHECK_ARG(nblocks >= 0, "negative size");
    if (nblocks == 0) return GGNN_OK;
    GGNN_CHECK_ARG(param_ptrs && var_numel && grads && m && v && partial && block_var && var_first && var_active, "null pointer");
    GGNN_CHECK_ARG(aligned16(grads) && aligned16(m) && aligned16(v), "buffers must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(sumsq_blocks_kernel, dim3(nblocks), dim3(256), 0, st, reinterpret_cast<const f32x4*>(grads), partial);
    GGNN_CHECK_HIP(hipGetLastError());
    hipLaunchKernelGGL(clip_adam_kernel, dim3(nblocks), dim3(256), 0, st, param_ptrs, var_numel,
                       reinterpret_cast<const f32x4*>(grads), reinterpret_cast<f32x4*>(m), reinterpret_cast<f32x4*>(v),
                       (const float*)partial, block_var, var_first, var_active, clip_norm, lr_t, beta1, beta2, epsilon);
    GGNN_CHECK_HIP(hipGetLastError());
    return GGNN_OK;
}
