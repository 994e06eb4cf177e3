// Measurement aid (not on the product path): what dense bf16 MFMA rate does THIS chip sustain under its power limit?
// bench.py prints it next to the roofline fraction, whose denominator is the 2.5 PF the data sheet states at 2.4 GHz: under a stream
// of v_mfma_f32_16x16x32_bf16 on random operands an MI355X holds ~1.95 GHz at its socket power limit (round 4,
// profiles/r04_experiments/mfma_power_probe.txt: 1974 TF; all-zero operands: 2446 TF at 2.37 GHz).  tools/mfma_power_probe.hip is the
// stand-alone version with more operand patterns.
#include "ggnn_common.h"
#include <vector>

namespace ggnn {
namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 pb_bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned pb_u32x4;

__device__ __forceinline__ f32x4 pb_mm(pb_u32x4 a, pb_u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(pb_bf16x8, a), __builtin_bit_cast(pb_bf16x8, b), c, 0, 0, 0);
}

// operand registers: 12 sets of 64 lanes x 16 bytes.  mode 0: zeros; 1: random bf16 in (-2, 2); 2: planes of a 3-way split (set % 3 =
// hi / mid / lo: exponents 8 / 16 lower)
__global__ void probe_fill_kernel(unsigned* __restrict__ dst, int mode) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;               // 12 * 64 * 4 words
    if (i >= 12 * 256) return;
    unsigned x = (unsigned)i * 2654435761u + 12345u;
    auto next = [&]() { x ^= x << 13; x ^= x >> 17; x ^= x << 5; return x; };
    const int plane = mode == 2 ? (i / 256) % 3 : 0;
    auto piece = [&]() { const unsigned r = next(); return ((r & 1u) << 15) | ((120u - 8u * plane + ((r >> 1) & 7u)) << 7) | ((r >> 4) & 127u); };
    const unsigned lo = piece(), hi = piece();
    dst[i] = mode == 0 ? 0u : (lo | (hi << 16));
}

// SPLIT: the operand pattern of ggnn_split.hpp's unit -- (wl,ah) (wm,am) (wm,ah) (wh,al) (wh,am) (wh,ah), chains of six on one
// accumulator -- else all pairs of 6 x 6 operand sets round-robin over 8 accumulators
template <bool SPLIT>
__global__ __launch_bounds__(512, 1) void probe_mfma_kernel(const pb_u32x4* __restrict__ src, float* __restrict__ sink,
                                                            unsigned long long* __restrict__ clk, int iters) {
    const int lane = threadIdx.x & 63;
    pb_u32x4 A[6], B[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) { A[i] = src[i * 64 + lane]; B[i] = src[(6 + i) * 64 + lane]; }
    f32x4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    unsigned long long t0 = 0, r0 = 0;
    if (threadIdx.x == 0) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    for (int it = 0; it < iters; ++it) {                              // 48 MFMAs per iteration
        if constexpr (SPLIT) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int w = (u & 1) * 3, a = ((u >> 1) & 1) * 3;
                f32x4 c = acc[u];
                c = pb_mm(A[w + 2], B[a + 0], c); c = pb_mm(A[w + 1], B[a + 1], c); c = pb_mm(A[w + 1], B[a + 0], c);
                c = pb_mm(A[w + 0], B[a + 2], c); c = pb_mm(A[w + 0], B[a + 1], c); c = pb_mm(A[w + 0], B[a + 0], c);
                acc[u] = c;
            }
        } else {
#pragma unroll
            for (int k = 0; k < 48; ++k) acc[k & 7] = pb_mm(A[k % 6], B[(k / 6 + k) % 6], acc[k & 7]);
        }
    }
    if (threadIdx.x == 0) {
        clk[blockIdx.x * 2] = __builtin_amdgcn_s_memtime() - t0;
        clk[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memrealtime() - r0;      // 100 MHz
    }
    f32x4 s = acc[0];
#pragma unroll
    for (int i = 1; i < 8; ++i) s += acc[i];
    if (s.x == 123.456f) sink[threadIdx.x] = s.x + s.y + s.z + s.w;          // (keeps the MFMAs alive)
}

}  // namespace
}  // namespace ggnn

using namespace ggnn;

extern "C" size_t ggnn_probe_mfma_workspace_bytes(void) { return (size_t)12 * 256 * 4 + 4096 + (size_t)num_cus() * 16 + 256; }

// mode 0 zeros | 1 random operands | 2 the split product's operand pattern and planes.  Runs `launches` launches of ~16 ms back to
// back on `stream` (the chip's power management settles within the first few) and reports the LAST one: dense bf16 TFLOP/s of the
// whole chip and the mean shader clock inside the launch.  Synchronises the stream.
extern "C" int ggnn_probe_mfma_rate(int mode, int launches, void* ws, size_t ws_bytes, double* tflops, double* shader_mhz,
                                    ggnn_stream_t stream) {
    GGNN_CHECK_ARG(mode >= 0 && mode <= 2 && launches >= 1 && ws && tflops && shader_mhz, "bad probe arguments");
    GGNN_CHECK_ARG(ws_bytes >= ggnn_probe_mfma_workspace_bytes() && aligned16(ws), "probe workspace too small or misaligned");
    hipStream_t st = (hipStream_t)stream;
    const int ncu = num_cus(), iters = 40000;
    unsigned* ops = reinterpret_cast<unsigned*>(ws);
    float* sink = reinterpret_cast<float*>(ops + 12 * 256);
    unsigned long long* clk = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(sink) + 4096);
    hipLaunchKernelGGL(probe_fill_kernel, dim3(12), dim3(256), 0, st, ops, mode);
    hipEvent_t e0, e1;
    GGNN_CHECK_HIP(hipEventCreate(&e0)); GGNN_CHECK_HIP(hipEventCreate(&e1));
    for (int rep = 0; rep < launches; ++rep) {
        if (rep == launches - 1) GGNN_CHECK_HIP(hipEventRecord(e0, st));
        if (mode == 2) hipLaunchKernelGGL(probe_mfma_kernel<true>, dim3(ncu), dim3(512), 0, st, (const pb_u32x4*)ops, sink, clk, iters);
        else hipLaunchKernelGGL(probe_mfma_kernel<false>, dim3(ncu), dim3(512), 0, st, (const pb_u32x4*)ops, sink, clk, iters);
    }
    GGNN_CHECK_HIP(hipEventRecord(e1, st));
    GGNN_CHECK_HIP(hipGetLastError());
    GGNN_CHECK_HIP(hipStreamSynchronize(st));
    float ms = 0.f;
    GGNN_CHECK_HIP(hipEventElapsedTime(&ms, e0, e1));
    hipEventDestroy(e0); hipEventDestroy(e1);
    std::vector<unsigned long long> hc((size_t)ncu * 2);
    GGNN_CHECK_HIP(hipMemcpy(hc.data(), clk, hc.size() * 8, hipMemcpyDeviceToHost));
    double mhz = 0.0;
    for (int b = 0; b < ncu; ++b) mhz += (double)hc[2 * b] / ((double)hc[2 * b + 1] / 100.0);
    *shader_mhz = mhz / ncu;
    *tflops = 2.0 * 16 * 16 * 32 * 48.0 * iters * 8 * ncu / (ms * 1e-3) / 1e12;
    return GGNN_OK;
}
