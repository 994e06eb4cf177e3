// Sustained shader clock of gfx950 under different instruction mixes: s_memtime (shader clocks) against s_memrealtime (100 MHz).
//   hipcc --offload-arch=gfx950 -O3 tools/clockprobe/clockprobe.hip -o tools/clockprobe/clockprobe && tools/clockprobe/clockprobe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>   // 0: fp32 MFMA back to back (4 independent accumulators), 1: v_fma_f32 stream, 2: idle spin on s_sleep, 3: bf16 MFMA
__global__ __launch_bounds__(512) void probe(unsigned long long* out, int iters, float seed) {
    f32x4 acc[4] = {{seed, 0, 0, 0}, {0, seed, 0, 0}, {0, 0, seed, 0}, {0, 0, 0, seed}};
    float a = seed + threadIdx.x, b = seed * 0.5f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
        if constexpr (MODE == 0) {
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[k & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[k & 3], 0, 0, 0);
        } else if constexpr (MODE == 1) {
#pragma unroll
            for (int k = 0; k < 64; ++k) acc[k & 3] = acc[k & 3] * a + b;
        } else if constexpr (MODE == 2) {
            __builtin_amdgcn_s_sleep(64);
        } else {
            typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
            bf16x8 x, y;
#pragma unroll
            for (int k = 0; k < 8; ++k) { x[k] = (__bf16)a; y[k] = (__bf16)b; }
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[k & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, acc[k & 3], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int k = 0; k < 4; ++k) s += acc[k].x + acc[k].y + acc[k].z + acc[k].w;
    if (threadIdx.x == 0) { out[blockIdx.x * 3 + 0] = t1 - t0; out[blockIdx.x * 3 + 1] = r1 - r0; out[blockIdx.x * 3 + 2] = (unsigned long long)(s != 12345.f); }
}

template <int MODE>
static void run(const char* name, int iters, int blocks, double flop_per_iter_per_wave) {
    unsigned long long* d;
    hipMalloc(&d, blocks * 3 * sizeof(unsigned long long));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(512), 0, 0, d, iters, 1.0f);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(blocks * 3);
        hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
        double sc = 0, rc = 0;
        for (int b = 0; b < blocks; ++b) { sc += h[b * 3]; rc += h[b * 3 + 1]; }
        sc /= blocks; rc /= blocks;
        const double us = rc / 100.0;     // s_memrealtime: 100 MHz
        const double tf = flop_per_iter_per_wave * iters * blocks * 8 / (ms * 1e-3) / 1e12;
        printf("%-10s rep %d: %8.1f us by events, %8.1f us by realtime, shader clocks %.0f -> %.3f GHz%s", name, rep, ms * 1e3, us, sc, sc / us / 1e3,
               flop_per_iter_per_wave > 0 ? "" : "\n");
        if (flop_per_iter_per_wave > 0) printf(", %.1f TFLOP/s\n", tf);
    }
    hipFree(d);
}

int main() {
    const int blocks = 256 * 2;    // 2 x 8 waves per CU: 4 waves per SIMD
    run<2>("idle", 2000, blocks, 0);
    run<0>("mfma_f32", 20000, blocks, 16.0 * 2 * 16 * 16 * 4);
    run<1>("valu_fma", 20000, blocks, 64.0 * 4 * 64 * 2);
    run<3>("mfma_bf16", 20000, blocks, 16.0 * 2 * 16 * 16 * 32);
    run<0>("mfma_f32", 100000, blocks, 16.0 * 2 * 16 * 16 * 4);
    return 0;
}
