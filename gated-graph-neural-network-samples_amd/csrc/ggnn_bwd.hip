// Element-wise stages of the GRU backward pass (what TF autodiff derives from GRUCell, chem_tensorflow.py:184),
// fused into two HBM-bound passes (float4 per lane) instead of ~20 separate torch element-wise launches:
//
//   stage 1 (needs only the upstream gradient g = dL/dh'):
//       dpc = g*(1-u)*act'(c)            -> dpc [V,D]            (pre-activation gradient of the candidate)
//       dpu = g*(h-c)*u*(1-u)            -> dpg[:, D:2D]         (pre-activation gradient of the u gate)
//       dh  = g*u                        -> dh  [V,D]
//       rh  = r*h                        -> a_c[:, col0:col0+D]  (the [x | r*h] operand of the dWc product)
//   stage 2 (after drh = (dpc Wc^T)[:, h-part] is known):
//       dh += drh*r ;  dpr = drh*h*r*(1-r) -> dpg[:, 0:D]
#include "ggnn_common.h"

namespace ggnn {

__global__ __launch_bounds__(256) void gru_bwd_stage1_kernel(const f32x4* __restrict__ g, const f32x4* __restrict__ h,
                                                             const f32x4* __restrict__ r, const f32x4* __restrict__ u,
                                                             const f32x4* __restrict__ c, int act, f32x4* __restrict__ dpc,
                                                             float* __restrict__ dpg, float* __restrict__ dh,
                                                             float* __restrict__ a_c, int lda, int col0, long long n4, int D4) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const long long row = i / D4;
    const int c4 = (int)(i - row * D4);
    const f32x4 gv = g[i], hv = h[i], rv = r[i], uv = u[i], cv = c[i];
    const f32x4 omu = 1.0f - uv;
    f32x4 d = gv * omu;
    if (act == GGNN_ACT_TANH) d = d * (1.0f - cv * cv);
    else { d.x = cv.x > 0.f ? d.x : 0.f; d.y = cv.y > 0.f ? d.y : 0.f; d.z = cv.z > 0.f ? d.z : 0.f; d.w = cv.w > 0.f ? d.w : 0.f; }
    dpc[i] = d;
    const int D = D4 * 4;
    *reinterpret_cast<f32x4*>(dpg + row * 2 * D + D + 4 * c4) = gv * (hv - cv) * uv * omu;
    *reinterpret_cast<f32x4*>(dh + row * D + 4 * c4) = gv * uv;
    *reinterpret_cast<f32x4*>(a_c + row * lda + col0 + 4 * c4) = rv * hv;
}

__global__ __launch_bounds__(256) void gru_bwd_stage2_kernel(const float* __restrict__ drh, int ld_drh, const f32x4* __restrict__ h,
                                                             const f32x4* __restrict__ r, float* __restrict__ dh,
                                                             float* __restrict__ dpg, long long n4, int D4) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const long long row = i / D4;
    const int c4 = (int)(i - row * D4);
    const int D = D4 * 4;
    const f32x4 dv = *reinterpret_cast<const f32x4*>(drh + row * ld_drh + 4 * c4);
    const f32x4 hv = h[i], rv = r[i];
    f32x4* dhp = reinterpret_cast<f32x4*>(dh + row * D + 4 * c4);
    *dhp = *dhp + dv * rv;
    *reinterpret_cast<f32x4*>(dpg + row * 2 * D + 4 * c4) = dv * hv * rv * (1.0f - rv);
}

// dP = g * act'(out) for out = act(P): tanh' = 1 - out^2, relu' = [out > 0]   (BasicRNNCell backward, chem_tensorflow_sparse.py:109-110)
__global__ __launch_bounds__(256) void act_bwd_kernel(const f32x4* __restrict__ g, const f32x4* __restrict__ out, int act,
                                                      f32x4* __restrict__ dP, long long n4) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const f32x4 gv = g[i], o = out[i];
    f32x4 d;
    if (act == GGNN_ACT_TANH) d = gv * (1.0f - o * o);
    else { d.x = o.x > 0.f ? gv.x : 0.f; d.y = o.y > 0.f ? gv.y : 0.f; d.z = o.z > 0.f ? gv.z : 0.f; d.w = o.w > 0.f ? gv.w : 0.f; }
    dP[i] = d;
}

// CudnnCompatibleGRUCell backward, element-wise part (chem_tensorflow_sparse.py:105-108):
//   c = tanh(x Wcx + bcx + r * hc), hc = h Wch + bch, h' = u h + (1-u) c
//   dpc = g (1-u) (1-c^2);  dpu = g (h-c) u (1-u);  dh = g u;  dhc = dpc r;  dpr = dpc hc r (1-r);  dpg = [dpr | dpu]
__global__ __launch_bounds__(256) void cudnn_bwd_stage_kernel(const f32x4* __restrict__ g, const f32x4* __restrict__ h,
                                                              const f32x4* __restrict__ r, const f32x4* __restrict__ u,
                                                              const f32x4* __restrict__ c, const f32x4* __restrict__ hc,
                                                              f32x4* __restrict__ dpc, float* __restrict__ dpg, f32x4* __restrict__ dh,
                                                              f32x4* __restrict__ dhc, long long n4, int D4) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const long long row = i / D4;
    const int c4 = (int)(i - row * D4);
    const int D = D4 * 4;
    const f32x4 gv = g[i], hv = h[i], rv = r[i], uv = u[i], cv = c[i], hcv = hc[i];
    const f32x4 omu = 1.0f - uv;
    const f32x4 d = gv * omu * (1.0f - cv * cv);
    dpc[i] = d;
    dh[i] = gv * uv;
    dhc[i] = d * rv;
    *reinterpret_cast<f32x4*>(dpg + row * 2 * D + 4 * c4) = d * hcv * rv * (1.0f - rv);
    *reinterpret_cast<f32x4*>(dpg + row * 2 * D + D + 4 * c4) = gv * (hv - cv) * uv * omu;
}

}  // namespace ggnn

using namespace ggnn;

extern "C" int ggnn_gru_bwd_stage1_f32(const float* g, const float* h, const float* r, const float* u, const float* c, int act,
                                       float* dpc, float* dpg, float* dh, float* a_c, int lda, int col0, int V, int D,
                                       ggnn_stream_t stream) {
    GGNN_CHECK_ARG(V >= 0 && D > 0 && D % 4 == 0 && lda % 4 == 0 && col0 % 4 == 0 && lda >= col0 + D, "bad sizes");
    if (V == 0) return GGNN_OK;
    GGNN_CHECK_ARG(g && h && r && u && c && dpc && dpg && dh && a_c, "null pointer");
    GGNN_CHECK_ARG(aligned16(g) && aligned16(h) && aligned16(r) && aligned16(u) && aligned16(c) && aligned16(dpc) &&
                   aligned16(dpg) && aligned16(dh) && aligned16(a_c), "pointers must be 16-byte aligned");
    const long long n4 = (long long)V * (D / 4);
    hipLaunchKernelGGL(gru_bwd_stage1_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const f32x4*)g, (const f32x4*)h, (const f32x4*)r, (const f32x4*)u, (const f32x4*)c, act, (f32x4*)dpc, dpg,
                       dh, a_c, lda, col0, n4, D / 4);
    GGNN_CHECK_HIP(hipGetLastError());
    return GGNN_OK;
}

extern "C" int ggnn_gru_bwd_stage2_f32(const float* drh, int ld_drh, const float* h, const float* r, float* dh, float* dpg,
                                       int V, int D, ggnn_stream_t stream) {
    GGNN_CHECK_ARG(V >= 0 && D > 0 && D % 4 == 0 && ld_drh % 4 == 0 && ld_drh >= D, "bad sizes");
    if (V == 0) return GGNN_OK;
    GGNN_CHECK_ARG(drh && h && r && dh && dpg, "null pointer");
    GGNN_CHECK_ARG(aligned16(drh) && aligned16(h) && aligned16(r) && aligned16(dh) && aligned16(dpg), "pointers must be 16-byte aligned");
    const long long n4 = (long long)V * (D / 4);
    hipLaunchKernelGGL(gru_bwd_stage2_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, drh, ld_drh,
                       (const f32x4*)h, (const f32x4*)r, dh, dpg, n4, D / 4);
    GGNN_CHECK_HIP(hipGetLastError());
    return GGNN_OK;
}

extern "C" int ggnn_act_bwd_f32(const float* g, const float* out, int act, float* dP, int V, int D, ggnn_stream_t stream) {
    GGNN_CHECK_ARG(V >= 0 && D > 0 && D % 4 == 0, "bad sizes");
    GGNN_CHECK_ARG(act == GGNN_ACT_TANH || act == GGNN_ACT_RELU, "unknown activation %d", act);
    if (V == 0) return GGNN_OK;
    GGNN_CHECK_ARG(g && out && dP && aligned16(g) && aligned16(out) && aligned16(dP), "null or misaligned pointer");
    const long long n4 = (long long)V * (D / 4);
    hipLaunchKernelGGL(act_bwd_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const f32x4*)g,
                       (const f32x4*)out, act, (f32x4*)dP, n4);
    GGNN_CHECK_HIP(hipGetLastError());
    return GGNN_OK;
}

extern "C" int ggnn_cudnn_gru_bwd_stage_f32(const float* g, const float* h, const float* r, const float* u, const float* c,
                                            const float* hc, float* dpc, float* dpg, float* dh, float* dhc, int V, int D,
                                            ggnn_stream_t stream) {
    GGNN_CHECK_ARG(V >= 0 && D > 0 && D % 4 == 0, "bad sizes");
    if (V == 0) return GGNN_OK;
    GGNN_CHECK_ARG(g && h && r && u && c && hc && dpc && dpg && dh && dhc, "null pointer");
    GGNN_CHECK_ARG(aligned16(g) && aligned16(h) && aligned16(r) && aligned16(u) && aligned16(c) && aligned16(hc) && aligned16(dpc) &&
                   aligned16(dpg) && aligned16(dh) && aligned16(dhc), "pointers must be 16-byte aligned");
    const long long n4 = (long long)V * (D / 4);
    hipLaunchKernelGGL(cudnn_bwd_stage_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const f32x4*)g,
                       (const f32x4*)h, (const f32x4*)r, (const f32x4*)u, (const f32x4*)c, (const f32x4*)hc, (f32x4*)dpc, dpg,
                       (f32x4*)dh, (f32x4*)dhc, n4, D / 4);
    GGNN_CHECK_HIP(hipGetLastError());
    return GGNN_OK;
}
