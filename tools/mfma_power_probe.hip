// What bf16 MFMA rate does the chip SUSTAIN under its power limit, as a function of operand data?
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_power_probe.hip -o tools/_bin/mfma_power_probe ; tools/_bin/mfma_power_probe
// Every CU runs 8 waves (2 per SIMD) of v_mfma_f32_16x16x32_bf16 back to back on register operands for ~1.5 s per mode:
//   zero    all-zero operands
//   same    random operands, the SAME pair in every MFMA of the stream (inputs do not toggle between instructions)
//   rand    random operands, a different pair in every MFMA (8 A x 8 B registers sets, all pairs)
//   split   the operand pattern of the 3-way split product: per unit (wl,ah) (wm,am) (wm,ah) (wh,al) (wh,am) (wh,ah)
//   split2  the same six products ordered (wl,ah) (wm,ah) (wm,am) (wh,am) (wh,al) (wh,ah) (fewer operand changes)
// Prints achieved dense TFLOP/s and the shader clock (s_memtime against the 100 MHz s_memrealtime).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstring>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

__device__ __forceinline__ f32x4 mm(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

template <int MODE>
__global__ __launch_bounds__(512, 1) void probe(const u32x4* __restrict__ src, float* __restrict__ out, unsigned long long* clk, int iters) {
    const int lane = threadIdx.x & 63;
    u32x4 A[6], B[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) { A[i] = src[(i * 64 + lane)]; B[i] = src[((6 + i) * 64 + lane)]; }
    f32x4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    unsigned long long t0 = 0, r0 = 0;
    if (threadIdx.x == 0) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    for (int it = 0; it < iters; ++it) {
        // 48 MFMAs per iteration, 8 accumulators round-robin (dependent distance 8)
        if constexpr (MODE == 1) {                // same operands
#pragma unroll
            for (int k = 0; k < 48; ++k) acc[k & 7] = mm(A[0], B[0], acc[k & 7]);
        } else if constexpr (MODE == 2 || MODE == 0) {   // all pairs of 6 x 6 + 12 more
#pragma unroll
            for (int k = 0; k < 48; ++k) acc[k & 7] = mm(A[k % 6], B[(k / 6 + k) % 6], acc[k & 7]);
        } else if constexpr (MODE == 3) {         // split pattern: unit u uses weights (A[3j..3j+2] = hi, mid, lo) and activation planes B[0..2] / B[3..5]
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int w = (u & 1) * 3, a = ((u >> 1) & 1) * 3;
                f32x4 c = acc[u];
                c = mm(A[w + 2], B[a + 0], c); c = mm(A[w + 1], B[a + 1], c); c = mm(A[w + 1], B[a + 0], c);
                c = mm(A[w + 0], B[a + 2], c); c = mm(A[w + 0], B[a + 1], c); c = mm(A[w + 0], B[a + 0], c);
                acc[u] = c;
            }
        } else {                                  // split2
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int w = (u & 1) * 3, a = ((u >> 1) & 1) * 3;
                f32x4 c = acc[u];
                c = mm(A[w + 2], B[a + 0], c); c = mm(A[w + 1], B[a + 0], c); c = mm(A[w + 1], B[a + 1], c);
                c = mm(A[w + 0], B[a + 1], c); c = mm(A[w + 0], B[a + 2], c); c = mm(A[w + 0], B[a + 0], c);
                acc[u] = c;
            }
        }
    }
    if (threadIdx.x == 0) {
        clk[blockIdx.x * 2] = __builtin_amdgcn_s_memtime() - t0;
        clk[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
    f32x4 s = acc[0];
#pragma unroll
    for (int i = 1; i < 8; ++i) s += acc[i];
    if (s.x == 123.456f) out[threadIdx.x] = s.x + s.y + s.z + s.w;      // (keeps the MFMAs alive)
}

int main() {
    int ncu = 256; hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0); ncu = pr.multiProcessorCount;
    std::vector<unsigned> hrand(12 * 64 * 4), hzero(12 * 64 * 4, 0u), hsplit(12 * 64 * 4);
    srand(1);
    for (auto& v : hrand) {   // two random bf16 in (-2, 2): random sign / mantissa, exponent 120..127
        unsigned lo = ((rand() & 1) << 15) | ((120 + rand() % 8) << 7) | (rand() & 127), hi = ((rand() & 1) << 15) | ((120 + rand() % 8) << 7) | (rand() & 127);
        v = lo | (hi << 16);
    }
    // split-like data: planes hi / mid / lo of random f32 values (mid, lo: exponents 8 / 16 lower)
    for (int set = 0; set < 12; ++set) for (int i = 0; i < 256; ++i) {
        const int plane = set % 3;
        auto piece = [&]() { return (unsigned)(((rand() & 1) << 15) | ((120 - 8 * plane + rand() % 8) << 7) | (rand() & 127)); };
        hsplit[set * 256 + i] = piece() | (piece() << 16);
    }
    u32x4* d; float* o; unsigned long long* c;
    hipMalloc(&d, hrand.size() * 4); hipMalloc(&o, 4096); hipMalloc(&c, ncu * 16);
    const char* names[] = {"zero", "same", "rand", "split", "split2"};
    const int iters = 40000;                       // 48 * 40000 MFMAs per wave ~ 16 ms at 2 GHz with 2 waves per SIMD
    for (int mode = 0; mode < 5; ++mode) {
        const std::vector<unsigned>& h = mode == 0 ? hzero : (mode >= 3 ? hsplit : hrand);
        hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        float ms = 0.f; std::vector<unsigned long long> hc(ncu * 2);
        for (int rep = 0; rep < 90; ++rep) {       // ~1.5 s of load; the last launch is the one reported
            if (rep == 89) hipEventRecord(e0);
            switch (mode) {
                case 0: hipLaunchKernelGGL(probe<0>, dim3(ncu), dim3(512), 0, 0, d, o, c, iters); break;
                case 1: hipLaunchKernelGGL(probe<1>, dim3(ncu), dim3(512), 0, 0, d, o, c, iters); break;
                case 2: hipLaunchKernelGGL(probe<2>, dim3(ncu), dim3(512), 0, 0, d, o, c, iters); break;
                case 3: hipLaunchKernelGGL(probe<3>, dim3(ncu), dim3(512), 0, 0, d, o, c, iters); break;
                default: hipLaunchKernelGGL(probe<4>, dim3(ncu), dim3(512), 0, 0, d, o, c, iters); break;
            }
            if (rep == 89) hipEventRecord(e1);
        }
        hipDeviceSynchronize(); hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(hc.data(), c, ncu * 16, hipMemcpyDeviceToHost);
        double mhz = 0; for (int b = 0; b < ncu; ++b) mhz += (double)hc[2 * b] / ((double)hc[2 * b + 1] / 100.0); mhz /= ncu;
        const double flops = 2.0 * 16 * 16 * 32 * 48.0 * iters * 8 * ncu;
        printf("%-6s  %8.3f ms  %7.1f TFLOP/s dense bf16  (%.3f of 2500)  shader clock %.0f MHz  clocks per MFMA per SIMD %.2f\n", names[mode], ms,
               flops / (ms * 1e-3) / 1e12, flops / (ms * 1e-3) / 2.5e15, mhz, (double)hc[0] / (48.0 * iters * 2));
        fflush(stdout);
    }
    return 0;
}
